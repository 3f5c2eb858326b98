// csrc/raster01.hip -- 01_resize.py: cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA) when the longest side exceeds max_dimension
// (01:15-19).  PARITY UNPINNED against OpenCV itself (cv2 absent); bit-exact against oracle/src/orc_raster.cpp resize_area, which
// restates OpenCV 4.x resizeAreaFast_ / resizeArea_.
//
// OpenCV builds per-axis tables of (source index, alpha) and reduces a source row at a time; the arithmetic of ONE destination pixel is nevertheless
// closed: its x entries are [left partial cell] [whole cells] [right partial cell] in that order, its y entries the same, every product and every sum
// rounded to float one by one.  One thread per destination pixel re-creates exactly that sequence from (dx, dy) alone -- no tables, no row buffers --
// and reads its source cell (scale_x * scale_y pixels, neighbours share only the partial border cells) once: the kernel is bound by reading the
// source image from HBM (cn bytes per source pixel) and writes cn bytes per destination pixel.
#include "orip_ctx.h"
#include <cfloat>

namespace {
struct AxisSpan {          // the table entries of one destination index (computeResizeAreaTab), in order: left partial, n_mid whole cells, right partial
    int s0;                // first whole cell
    int n_mid;
    float a_left, a_mid, a_right;
    bool left, right;
    int s_right;
};

__device__ __forceinline__ AxisSpan axis_span(int d, int ssize, double scale) {
    AxisSpan A;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    A.s0 = s1; A.n_mid = s2 - s1; A.s_right = s2;
    A.left = (s1 - f1 > 1e-3);
    A.a_left = (float)((s1 - f1) / cell);
    A.a_mid = (float)(1.0 / cell);
    A.right = (f2 - s2 > 1e-3);
    A.a_right = (float)(fmin(fmin(f2 - s2, 1.), cell) / cell);
    return A;
}

template <int CN>
__device__ __forceinline__ void row_acc(const uint8_t* __restrict__ S, const AxisSpan& X, float* buf) {
#pragma unroll
    for (int c = 0; c < CN; c++) buf[c] = 0.f;
    if (X.left) {
#pragma unroll
        for (int c = 0; c < CN; c++) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)S[(size_t)(X.s0 - 1) * CN + c], X.a_left));
    }
    for (int i = 0; i < X.n_mid; i++) {
#pragma unroll
        for (int c = 0; c < CN; c++) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)S[(size_t)(X.s0 + i) * CN + c], X.a_mid));
    }
    if (X.right) {
#pragma unroll
        for (int c = 0; c < CN; c++) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)S[(size_t)X.s_right * CN + c], X.a_right));
    }
}

__device__ __forceinline__ uint8_t round_u8(float v) { const int r = __float2int_rn(v); return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); }

template <int CN>
__global__ __launch_bounds__(256) void k_resize_area(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw,
                                                     double scale_x, double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const AxisSpan X = axis_span(dx, sw, scale_x), Y = axis_span(dy, sh, scale_y);
    float sum[CN], buf[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = 0.f;
    if (Y.left) {
        row_acc<CN>(src + (size_t)(Y.s0 - 1) * sw * CN, X, buf);
#pragma unroll
        for (int c = 0; c < CN; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(Y.a_left, buf[c]));
    }
    for (int j = 0; j < Y.n_mid; j++) {
        row_acc<CN>(src + (size_t)(Y.s0 + j) * sw * CN, X, buf);
#pragma unroll
        for (int c = 0; c < CN; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(Y.a_mid, buf[c]));
    }
    if (Y.right) {
        row_acc<CN>(src + (size_t)Y.s_right * sw * CN, X, buf);
#pragma unroll
        for (int c = 0; c < CN; c++) sum[c] = __fadd_rn(sum[c], __fmul_rn(Y.a_right, buf[c]));
    }
#pragma unroll
    for (int c = 0; c < CN; c++) dst[((size_t)dy * dw + dx) * CN + c] = round_u8(sum[c]);
}

// both ratios integers (resizeAreaFast_): integer cell sums; 2 x 2 -> (sum + 2) >> 2, otherwise sum * float(1 / area) rounded half-to-even
template <int CN>
__global__ __launch_bounds__(256) void k_resize_area_int(const uint8_t* __restrict__ src, int sw, uint8_t* __restrict__ dst, int dh, int dw, int isx, int isy, float inv) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    int sum[CN];
#pragma unroll
    for (int c = 0; c < CN; c++) sum[c] = 0;
    for (int sy = 0; sy < isy; sy++) {
        const uint8_t* S = src + ((size_t)(dy * isy + sy) * sw + (size_t)dx * isx) * CN;
        for (int sx = 0; sx < isx; sx++) {
#pragma unroll
            for (int c = 0; c < CN; c++) sum[c] += S[sx * CN + c];
        }
    }
    const bool two = isx == 2 && isy == 2;
#pragma unroll
    for (int c = 0; c < CN; c++) dst[((size_t)dy * dw + dx) * CN + c] = two ? (uint8_t)((sum[c] + 2) >> 2) : round_u8(__fmul_rn((float)sum[c], inv));
}

template <int CN>
void launch_resize(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const dim3 grid(cdiv(dw, 64), cdiv(dh, 4)), block(256);
    if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON)
        hipLaunchKernelGGL(k_resize_area_int<CN>, grid, block, 0, s, src, sw, dst, dh, dw, isx, isy, 1.f / (float)(isx * isy));
    else
        hipLaunchKernelGGL(k_resize_area<CN>, grid, block, 0, s, src, sh, sw, dst, dh, dw, scale_x, scale_y);
}
}  // namespace

// src u8 [H,W,cn] on the host -> INTER_AREA shrink to [newH,newW,cn].  dst (host) may be null; as_image != 0 (cn == 3) leaves the result as the
// context's image, exactly as orip_set_image of the resized pixels would (resized.png never has to exist for the resident chain).
extern "C" int orip_resize_area(orip_ctx* c, const uint8_t* src, int H, int W, int cn, int newH, int newW, uint8_t* dst, int as_image) {
    orip_enter(c);
    if (!src || H <= 0 || W <= 0 || cn < 1 || cn > 4) ORIP_FAIL(c, "bad source image %dx%dx%d", W, H, cn);
    if (newH <= 0 || newW <= 0 || newH > H || newW > W) ORIP_FAIL(c, "INTER_AREA path shrinks only: %dx%d -> %dx%d", W, H, newW, newH);
    if (as_image && cn != 3) ORIP_FAIL(c, "the context image has 3 channels, got %d", cn);
    if (!dst && !as_image) ORIP_FAIL(c, "no destination");
    hipStream_t s = LN(c).stream;
    const size_t nsrc = (size_t)H * W * cn, ndst = (size_t)newH * newW * cn;
    HIPC(c, c->resize_src.ensure(nsrc + 16));
    uint8_t* out;
    if (as_image) {
        c->mask_bits = nullptr;
        ORIP_TRY(orip_raster02_lab_tables(c));
        HIPC(c, c->image.ensure(ndst + 16));
        out = c->image.as<uint8_t>();
    } else {
        HIPC(c, c->resize_dst.ensure(ndst + 16));
        out = c->resize_dst.as<uint8_t>();
    }
    HIPC(c, hipMemcpyAsync(c->resize_src.p, src, nsrc, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(c, "k_resize_area");
        const uint8_t* in = c->resize_src.as<uint8_t>();
        switch (cn) {
            case 1: launch_resize<1>(s, in, H, W, out, newH, newW); break;
            case 2: launch_resize<2>(s, in, H, W, out, newH, newW); break;
            case 3: launch_resize<3>(s, in, H, W, out, newH, newW); break;
            default: launch_resize<4>(s, in, H, W, out, newH, newW); break;
        }
    }
    HIPC(c, hipGetLastError());
    if (dst) HIPC(c, hipMemcpyAsync(dst, out, ndst, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    if (as_image) { c->H = newH; c->W = newW; }
    return 0;
}
