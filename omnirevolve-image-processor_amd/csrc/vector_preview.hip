// csrc/vector_preview.hip -- the rasters behind the preview stages 06 / 09 / 11 (06_preview_scaled.py:76-88 _draw_layer, 09_preview_intra.py:71-88
// _draw_lines / _draw_taps, 11_preview_cross.py the same on the cross lists): polylines drawn `thickness` px wide with cv2.LINE_AA and taps as
// filled anti-aliased discs on the full canvas.  Visual QA only, off the hot path.
//
// PARITY UNPINNED: OpenCV's anti-aliased line (a fixed-point Wu variant) is not restated.  What is drawn instead is documented here and in
// oracle/oracle.py: preview_cover, which the tests hold this file to bit by bit:
//   coverage of pixel (x, y) by a segment  = clamp(thickness / 2 + 0.5 - d, 0, 1), d = Euclidean distance from the pixel centre to the segment
//   (anti-aliasing off: 1 where d <= thickness / 2);  by a tap = clamp(radius + 0.5 - d, 0, 1), d = distance to its centre (off: d <= radius);
//   a plane holds, per pixel, the LARGEST coverage of any primitive, as round(255 a) -- the reference draws one line after the other, so where lines
//   overlap its anti-aliased fringes get darker than here; the host composes colour = (background (255 - A) + colour A + 127) / 255.
// All arithmetic in float64 in the same order on both sides (no fused multiply-add: -ffp-contract=off).
#include "vec_common.h"

namespace {
__device__ __forceinline__ int cover_u8(double a) { a = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a); return (int)floor(__dadd_rn(__dmul_rn(a, 255.0), 0.5)); }

// one thread per point k of the list: the segment k -> k + 1 when both belong to one polyline
__global__ __launch_bounds__(256) void k_prev_lines(const int64_t* __restrict__ off, const int2* __restrict__ pts, int64_t n_polys, int64_t total, int W, int H,
                                                    double half, int aa, int* __restrict__ plane) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k + 1 >= total) return;
    int64_t lo = 0, hi = n_polys - 1;                      // polyline of point k: the last i with off[i] <= k
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (off[mid] <= k) lo = mid; else hi = mid - 1; }
    if (k + 1 >= off[lo + 1]) return;                      // k is the last point of its polyline
    const int2 p0 = pts[k], p1 = pts[k + 1];
    const int r = (int)ceil(half + 0.5);
    const int xa = max(0, min(p0.x, p1.x) - r), xb = min(W - 1, max(p0.x, p1.x) + r), ya = max(0, min(p0.y, p1.y) - r), yb = min(H - 1, max(p0.y, p1.y) + r);
    const double vx = (double)p1.x - (double)p0.x, vy = (double)p1.y - (double)p0.y;
    const double L2 = __dadd_rn(__dmul_rn(vx, vx), __dmul_rn(vy, vy));
    for (int y = ya; y <= yb; y++) for (int x = xa; x <= xb; x++) {
        const double wx = (double)x - (double)p0.x, wy = (double)y - (double)p0.y;
        double t = 0.0;
        if (L2 > 0.0) { t = __ddiv_rn(__dadd_rn(__dmul_rn(wx, vx), __dmul_rn(wy, vy)), L2); t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t); }
        const double dx = __dsub_rn(wx, __dmul_rn(t, vx)), dy = __dsub_rn(wy, __dmul_rn(t, vy));
        const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
        const int A = aa ? cover_u8(__dsub_rn(__dadd_rn(half, 0.5), d)) : (d <= half ? 255 : 0);
        if (A > 0) atomicMax(&plane[(size_t)y * W + x], A);
    }
}
__global__ __launch_bounds__(256) void k_prev_discs(const int2* __restrict__ taps, int64_t n, int W, int H, double radius, int aa, int* __restrict__ plane) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int2 c = taps[i];
    const int r = (int)ceil(radius + 0.5);
    for (int y = max(0, c.y - r); y <= min(H - 1, c.y + r); y++) for (int x = max(0, c.x - r); x <= min(W - 1, c.x + r); x++) {
        const double dx = (double)x - (double)c.x, dy = (double)y - (double)c.y;
        const double d = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
        const int A = aa ? cover_u8(__dsub_rn(__dadd_rn(radius, 0.5), d)) : (d <= radius ? 255 : 0);
        if (A > 0) atomicMax(&plane[(size_t)y * W + x], A);
    }
}
__global__ __launch_bounds__(256) void k_prev_pack(const int* __restrict__ plane, int64_t n, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint8_t)plane[i];
}
}  // namespace

extern "C" int orip_preview_cover(orip_ctx* c, int slot, int layer, int taps_which, int W, int H, int thickness, int radius, int antialias, uint8_t* line_cov, uint8_t* tap_cov) {
    orip_enter(c);
    if (slot < 0 || slot >= ORIP_SLOT_COUNT || layer < 0 || layer >= ORIP_MAX_LAYERS || taps_which < -1 || taps_which > 1 || W <= 0 || H <= 0 || thickness < 1 || radius < 0)
        ORIP_FAIL(c, "bad arguments");
    const int64_t np = (int64_t)W * H;
    HIPC(c, LN(c).canvas.ensure((size_t)np * 4 + 64));
    HIPC(c, LN(c).vtmp[0].ensure((size_t)np + 64));
    int* plane = LN(c).canvas.as<int>(); uint8_t* packed = LN(c).vtmp[0].as<uint8_t>();
    if (line_cov) {
        DPolys& P = c->polys[slot][layer];
        HIPC(c, hipMemsetAsync(plane, 0, (size_t)np * 4, LN(c).stream));
        if (P.n > 0 && P.total > 1) {
            if (is_coded(P)) ORIP_TRY(orip_polys_materialize(c, P));          // a preview draws every point of the list, as the reference does
            hipLaunchKernelGGL(k_prev_lines, dim3((unsigned)cdiv(P.total, 256)), dim3(256), 0, LN(c).stream, P.off.as<int64_t>(), reinterpret_cast<const int2*>(P.pts.p), P.n, P.total, W, H,
                               (double)thickness * 0.5, antialias, plane);
        }
        hipLaunchKernelGGL(k_prev_pack, dim3((unsigned)cdiv(np, 256)), dim3(256), 0, LN(c).stream, plane, np, packed);
        HIPC(c, hipMemcpyAsync(line_cov, packed, (size_t)np, hipMemcpyDeviceToHost, LN(c).stream));
        HIPC(c, hipStreamSynchronize(LN(c).stream));
    }
    if (tap_cov) {
        HIPC(c, hipMemsetAsync(plane, 0, (size_t)np * 4, LN(c).stream));
        if (taps_which >= 0) {
            DTaps& T = c->taps[taps_which][layer];
            if (T.n > 0) hipLaunchKernelGGL(k_prev_discs, dim3((unsigned)cdiv(T.n, 256)), dim3(256), 0, LN(c).stream, T.xy.as<int2>(), (int64_t)T.n, W, H, (double)radius, antialias, plane);
        }
        hipLaunchKernelGGL(k_prev_pack, dim3((unsigned)cdiv(np, 256)), dim3(256), 0, LN(c).stream, plane, np, packed);
        HIPC(c, hipMemcpyAsync(tap_cov, packed, (size_t)np, hipMemcpyDeviceToHost, LN(c).stream));
        HIPC(c, hipStreamSynchronize(LN(c).stream));
    }
    HIPC(c, hipGetLastError());
    return 0;
}
