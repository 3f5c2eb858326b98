// csrc/raster02.hip -- stage 02 (02_color_extract.py) on gfx950:
//   lab_assign   : BGR u8 -> Lab u8 (cv2.cvtColor, 02:35) -> nearest centre (02:53-55) -> dark->light label (02:120-127)
//   lab_gather   : Lab of the k-means subsample (02:39-44)
//   kmeans_fit   : cv2.kmeans with KMEANS_PP_CENTERS (02:46-49), one persistent workgroup
//   morph_pass   : erode / dilate with an arbitrary <=7x7 structuring element (02:151-154, 03:25-30)
// All float arithmetic mirrors numpy / OpenCV evaluation order: compiled with -ffp-contract=off and written
// with explicit __fmul_rn/__fadd_rn where the order matters.
#include "orip_ctx.h"
#include <cmath>
#include <algorithm>

// ------------------------------------------------------------------------------------------------
// Lab tables, built on the host once per context.
// ------------------------------------------------------------------------------------------------
struct LabTabs { uint16_t gamma[256]; uint16_t cbrt[3072]; int32_t coef[9]; };

int orip_raster02_lab_tables(orip_ctx* c) {
    if (c->tabs_ready) return 0;
    static LabTabs t;
    for (int i = 0; i < 256; i++) {
        double x = i / 255.0;
        double g = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
        long v = std::lrint(255.0 * 8.0 * g);
        t.gamma[i] = (uint16_t)std::min<long>(std::max<long>(v, 0), 65535);
    }
    for (int i = 0; i < 3072; i++) {
        double x = i / (255.0 * 8.0);
        double y = x < 0.008856 ? x * 7.787 + 0.13793103448275862 : std::cbrt(x);
        long v = std::lrint(32768.0 * y);
        t.cbrt[i] = (uint16_t)std::min<long>(std::max<long>(v, 0), 65535);
    }
    const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    const double D65[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t.coef[i * 3 + j] = (int32_t)std::lrint(4096.0 * M[i * 3 + j] / D65[i]);
    HIPC(c, c->lab_tabs.ensure(sizeof(LabTabs)));
    HIPC(c, hipMemcpyAsync(c->lab_tabs.p, &t, sizeof(LabTabs), hipMemcpyHostToDevice, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    c->tabs_ready = true;
    return 0;
}

struct LabLds { uint16_t gamma[256]; uint16_t cbrt[3072]; };

__device__ __forceinline__ void lab_lds_load(LabLds* s, const LabTabs* t) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s->gamma[i] = t->gamma[i];
    for (int i = threadIdx.x; i < 3072; i += blockDim.x) s->cbrt[i] = t->cbrt[i];
    __syncthreads();
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void bgr2lab_px(const LabLds* s, const int32_t* C, int b8, int g8, int r8, int& L, int& a, int& b) {
    int B = s->gamma[b8], G = s->gamma[g8], R = s->gamma[r8];
    int fX = s->cbrt[descale(R * C[0] + G * C[1] + B * C[2], 12)];
    int fY = s->cbrt[descale(R * C[3] + G * C[4] + B * C[5], 12)];
    int fZ = s->cbrt[descale(R * C[6] + G * C[7] + B * C[8], 12)];
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    L = min(max(descale(Lscale * fY + Lshift, 15), 0), 255);
    a = min(max(descale(500 * (fX - fY) + 128 * (1 << 15), 15), 0), 255);
    b = min(max(descale(200 * (fY - fZ) + 128 * (1 << 15), 15), 0), 255);
}

struct Centers { float c[ORIP_MAX_LAYERS * 3]; uint8_t lut[ORIP_MAX_LAYERS]; int K; };

__device__ __forceinline__ int nearest_center(const Centers& cs, int L, int a, int b) {
    float p0 = (float)L, p1 = (float)a, p2 = (float)b;
    float best = 0.f; int kb = 0;
    for (int k = 0; k < cs.K; k++) {
        float d0 = __fsub_rn(p0, cs.c[3 * k]), d1 = __fsub_rn(p1, cs.c[3 * k + 1]), d2 = __fsub_rn(p2, cs.c[3 * k + 2]);
        float s = __fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2));
        if (k == 0 || s < best) { best = s; kb = k; }
    }
    return cs.lut[kb];
}

// 4 pixels per thread: 12 B in (3 dwords), 4 B out (1 dword).  Algorithmic traffic 3+1 B/px.
__global__ __launch_bounds__(256) void k_lab_assign(const u8* __restrict__ bgr, u8* __restrict__ labels, int64_t npx,
                                                     const LabTabs* __restrict__ tabs, Centers cs) {
    __shared__ LabLds s;
    __shared__ int32_t C[9];
    if (threadIdx.x < 9) C[threadIdx.x] = tabs->coef[threadIdx.x];
    lab_lds_load(&s, tabs);
    int64_t ngrp = npx >> 2;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngrp; g += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(bgr + g * 12);
        uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
        u8 px[12] = {(u8)w0, (u8)(w0 >> 8), (u8)(w0 >> 16), (u8)(w0 >> 24), (u8)w1, (u8)(w1 >> 8), (u8)(w1 >> 16), (u8)(w1 >> 24),
                     (u8)w2, (u8)(w2 >> 8), (u8)(w2 >> 16), (u8)(w2 >> 24)};
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int L, a, b; bgr2lab_px(&s, C, px[3 * j], px[3 * j + 1], px[3 * j + 2], L, a, b);
            out |= (uint32_t)nearest_center(cs, L, a, b) << (8 * j);
        }
        reinterpret_cast<uint32_t*>(labels)[g] = out;
    }
    // tail (npx % 4)
    if (blockIdx.x == 0 && threadIdx.x < (npx & 3)) {
        int64_t i = (npx & ~(int64_t)3) + threadIdx.x;
        int L, a, b; bgr2lab_px(&s, C, bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], L, a, b);
        labels[i] = (u8)nearest_center(cs, L, a, b);
    }
}

// Lab of selected pixels (idx == nullptr: pixel i itself)
__global__ __launch_bounds__(256) void k_lab_gather(const u8* __restrict__ bgr, const int64_t* __restrict__ idx, int64_t n,
                                                     u8* __restrict__ lab, const LabTabs* __restrict__ tabs) {
    __shared__ LabLds s;
    __shared__ int32_t C[9];
    if (threadIdx.x < 9) C[threadIdx.x] = tabs->coef[threadIdx.x];
    lab_lds_load(&s, tabs);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = idx ? idx[i] : i;
        int L, a, b; bgr2lab_px(&s, C, bgr[3 * p], bgr[3 * p + 1], bgr[3 * p + 2], L, a, b);
        lab[3 * i] = (u8)L; lab[3 * i + 1] = (u8)a; lab[3 * i + 2] = (u8)b;
    }
}

// process_colors.py kmeans_palette (:31-46) clusters RGB bytes, not Lab: the sampled pixels in R, G, B order
__global__ __launch_bounds__(256) void k_rgb_gather(const u8* __restrict__ bgr, const int64_t* __restrict__ idx, int64_t n, u8* __restrict__ rgb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx ? idx[i] : i;
        rgb[3 * i] = bgr[3 * p + 2]; rgb[3 * i + 1] = bgr[3 * p + 1]; rgb[3 * i + 2] = bgr[3 * p];
    }
}
// process_colors.py assign_labels (:69-77): nearest palette colour in RGB with the reference's int16 arithmetic -- the squares of differences
// above 181 wrap to negative values before they are summed (in int64), and np.argmin takes the first minimum.  Four pixels per thread (12 bytes in,
// 4 bytes out), the palette in LDS.
__global__ __launch_bounds__(256) void k_assign_palette(const u8* __restrict__ bgr, int64_t npx, const u8* __restrict__ pal_rgb, int K, u8* __restrict__ labels) {
    __shared__ int pr[ORIP_MAX_LAYERS], pg[ORIP_MAX_LAYERS], pb[ORIP_MAX_LAYERS];
    if (threadIdx.x < K) { pr[threadIdx.x] = pal_rgb[3 * threadIdx.x]; pg[threadIdx.x] = pal_rgb[3 * threadIdx.x + 1]; pb[threadIdx.x] = pal_rgb[3 * threadIdx.x + 2]; }
    __syncthreads();
    auto sq16 = [](int d) { return (int)(int16_t)(uint16_t)(d * d); };
    auto nearest = [&](int b, int g, int r) {
        int best = 0x7fffffff, bi = 0;
        for (int k = 0; k < K; k++) { const int d = sq16(r - pr[k]) + sq16(g - pg[k]) + sq16(b - pb[k]); if (d < best) { best = d; bi = k; } }
        return bi;
    };
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;                  // group of four pixels
    if (4 * q + 3 < npx) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(bgr) + 3 * q;    // hipMalloc'd, 12-byte groups: dword aligned
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const unsigned l0 = nearest(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);
        const unsigned l1 = nearest(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
        const unsigned l2 = nearest((w1 >> 16) & 255, w1 >> 24, w2 & 255);
        const unsigned l3 = nearest((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
        reinterpret_cast<uint32_t*>(labels)[q] = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
    } else {
        for (int64_t i = 4 * q; i < npx; i++) labels[i] = (u8)nearest(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
    }
}

__global__ void k_count_labels(const u8* __restrict__ labels, int64_t n, unsigned long long* counts) {
    __shared__ unsigned int h[ORIP_MAX_LAYERS];
    if (threadIdx.x < ORIP_MAX_LAYERS) h[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[labels[i] & 15], 1u);
    __syncthreads();
    if (threadIdx.x < ORIP_MAX_LAYERS && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// cv2.kmeans as ONE persistent 1024-thread workgroup.  Samples are u8 Lab triples,
// so kmeans++ works on exact integers; the sequential scans of the CPU algorithm are replaced by
// order-independent exact equivalents (see DESIGN.md "kmeans_fit").
// ------------------------------------------------------------------------------------------------
#define KM_T 1024
struct KmState { unsigned long long rng; };

__device__ __forceinline__ unsigned km_next(unsigned long long& st) {
    st = (unsigned long long)(unsigned)st * 4164903690ULL + (unsigned)(st >> 32);
    return (unsigned)st;
}
__device__ __forceinline__ double km_double(unsigned long long& st) {
    unsigned t = km_next(st);
    unsigned long long v = ((unsigned long long)t << 32) | km_next(st);
    return (double)v * 5.4210108624275221700372640043497e-20;
}

__device__ __forceinline__ long long block_sum_ll(long long v, long long* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long r = 0;
    for (int w = 0; w < KM_T / 64; w++) r += red[w];
    return r;
}
// deterministic (fixed tree) double sum
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    for (int w = 0; w < KM_T / 64; w++) r += red[w];
    return r;
}

__device__ __forceinline__ int isq3(const u8* a, const u8* b) {
    int d0 = (int)a[0] - b[0], d1 = (int)a[1] - b[1], d2 = (int)a[2] - b[2];
    return d0 * d0 + d1 * d1 + d2 * d2;
}
// hal::normL2Sqr_(sample, centre, 3) in float: s=0; s+=t*t per component
__device__ __forceinline__ float fsq3(const u8* a, const float* c) {
    float s = 0.f;
    for (int j = 0; j < 3; j++) { float t = __fsub_rn((float)a[j], c[j]); s = __fadd_rn(s, __fmul_rn(t, t)); }
    return s;
}

#ifdef ORIP_VARIANTS      // replaced variant (ORIP_KMEANS_1WG): variants build only (make variants)
__global__ __launch_bounds__(KM_T) void k_kmeans_fit(const u8* __restrict__ data, int N, int K, int attempts, int maxCount,
                                                      double epsilon, int32_t* __restrict__ dist0, int32_t* __restrict__ dist1,
                                                      int32_t* __restrict__ dist2, int32_t* __restrict__ labels,
                                                      float* __restrict__ centers_out, double* __restrict__ compact_out,
                                                      int* __restrict__ status) {
    __shared__ long long red[KM_T / 64];
    __shared__ double redd[KM_T / 64];
    __shared__ long long part[KM_T];
    __shared__ float centers[ORIP_MAX_LAYERS * 3], old_centers[ORIP_MAX_LAYERS * 3];
    __shared__ int csum[KM_T / 64][ORIP_MAX_LAYERS * 4];
    __shared__ long long tot[ORIP_MAX_LAYERS * 4];
    __shared__ int sh_ci, sh_flag;
    __shared__ int pp_idx[ORIP_MAX_LAYERS];
    __shared__ unsigned long long sh_key;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int chunk = (N + KM_T - 1) / KM_T;
    const int lo = min(N, tid * chunk), hi = min(N, lo + chunk);
    unsigned long long rng = 0xffffffffULL;   // identical in every thread
    double best_compact = 1.79769313486231570815e+308;
    int32_t *dist = dist0, *tdist = dist1, *tdist2 = dist2;

    for (int a = 0; a < attempts; a++) {
        double compactness = 0;
        for (int iter = 0;;) {
            double max_shift = iter == 0 ? 1.79769313486231570815e+308 : 0.0;
            // swap(centers, old_centers)
            __syncthreads();
            if (tid < K * 3) { float t = centers[tid]; centers[tid] = old_centers[tid]; old_centers[tid] = t; }
            __syncthreads();
            if (iter == 0) {
                // ---------------- generateCentersPP ----------------
                int c0 = (int)(km_next(rng) % (unsigned)N);
                if (tid == 0) pp_idx[0] = c0;
                long long ls = 0;
                for (int i = tid; i < N; i += KM_T) { int d = isq3(data + 3 * i, data + 3 * c0); dist[i] = d; ls += d; }
                long long sum0 = block_sum_ll(ls, red);
                for (int k = 1; k < K; k++) {
                    long long bestSum = 0x7fffffffffffffffLL; int bestCenter = -1;
                    for (int j = 0; j < 3; j++) {
                        double p = km_double(rng) * (double)sum0;
                        // ci = first index with inclusive prefix >= p, else N-1   (exact)
                        long long cs = 0;
                        for (int i = lo; i < hi; i++) cs += dist[i];
                        part[tid] = cs;
                        __syncthreads();
                        if (tid == 0) {
                            long long run = 0; int ci = N - 1; int t;
                            for (t = 0; t < KM_T; t++) { if ((double)(run + part[t]) >= p) break; run += part[t]; }
                            if (t < KM_T) {
                                int l2 = min(N, t * chunk), h2 = min(N, l2 + chunk);
                                for (int i = l2; i < h2; i++) { run += dist[i]; if ((double)run >= p) { ci = i; break; } }
                            }
                            // p <= 0 before any subtraction cannot happen (p>0 unless sum0==0); sum0==0: loop never breaks -> N-1
                            if (ci > N - 1) ci = N - 1;
                            sh_ci = ci;
                        }
                        __syncthreads();
                        int ci = sh_ci;
                        long long s = 0;
                        for (int i = tid; i < N; i += KM_T) { int d = min(isq3(data + 3 * i, data + 3 * ci), dist[i]); tdist2[i] = d; s += d; }
                        long long S = block_sum_ll(s, red);
                        if (S < bestSum) { bestSum = S; bestCenter = ci; int32_t* t = tdist; tdist = tdist2; tdist2 = t; }
                    }
                    if (tid == 0) pp_idx[k] = bestCenter;
                    sum0 = bestSum;
                    { int32_t* t = dist; dist = tdist; tdist = t; }
                    __syncthreads();
                }
                __syncthreads();
                if (tid < K * 3) centers[tid] = (float)data[3 * pp_idx[tid / 3] + tid % 3];
                __syncthreads();
            } else {
                // ---------------- recompute centres from labels ----------------
                for (int i = tid; i < (KM_T / 64) * ORIP_MAX_LAYERS * 4; i += KM_T) (&csum[0][0])[i] = 0;
                __syncthreads();
                for (int i = tid; i < N; i += KM_T) {
                    int k = labels[i];
                    atomicAdd(&csum[wave][k * 4 + 0], (int)data[3 * i]);
                    atomicAdd(&csum[wave][k * 4 + 1], (int)data[3 * i + 1]);
                    atomicAdd(&csum[wave][k * 4 + 2], (int)data[3 * i + 2]);
                    atomicAdd(&csum[wave][k * 4 + 3], 1);
                }
                __syncthreads();
                if (tid < K * 4) { long long t = 0; for (int w = 0; w < KM_T / 64; w++) t += csum[w][tid]; tot[tid] = t; }
                __syncthreads();
                // float accumulation in sample order is exact (== integer sum) while every partial sum < 2^24
                if (tid == 0) {
                    int bad = 0;
                    for (int k = 0; k < K; k++) for (int j = 0; j < 3; j++) if (tot[k * 4 + j] >= (1LL << 24)) bad = 1;
                    sh_flag = bad;
                }
                __syncthreads();
                if (sh_flag) {
                    // rare slow path: literal sequential float32 accumulation by one lane
                    if (tid == 0) {
                        float acc[ORIP_MAX_LAYERS * 3];
                        for (int q = 0; q < K * 3; q++) acc[q] = 0.f;
                        for (int i = 0; i < N; i++) { int k = labels[i]; for (int j = 0; j < 3; j++) acc[k * 3 + j] = __fadd_rn(acc[k * 3 + j], (float)data[3 * i + j]); }
                        for (int q = 0; q < K * 3; q++) centers[q] = acc[q];
                    }
                } else if (tid < K * 3) centers[tid] = (float)tot[(tid / 3) * 4 + tid % 3];
                __syncthreads();
                // empty-cluster repair (sequential over k, as the reference)
                for (int k = 0; k < K; k++) {
                    if (tot[k * 4 + 3] != 0) continue;      // uniform branch (shared)
                    int max_k = 0;
                    for (int k1 = 1; k1 < K; k1++) if (tot[max_k * 4 + 3] < tot[k1 * 4 + 3]) max_k = k1;
                    float nb[3]; float scale = 1.f / (float)tot[max_k * 4 + 3];
                    for (int j = 0; j < 3; j++) nb[j] = __fmul_rn(centers[max_k * 3 + j], scale);
                    // farthest point: max_dist <= dist  => last index among maxima
                    unsigned long long key = 0;
                    for (int i = tid; i < N; i += KM_T) {
                        if (labels[i] != max_k) continue;
                        float d = fsq3(data + 3 * i, nb);
                        unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
                        if (kk >= key) key = kk;
                    }
                    if (tid == 0) sh_key = 0;
                    __syncthreads();
                    atomicMax(&sh_key, key);
                    __syncthreads();
                    int far_i = (int)(sh_key & 0xffffffffu);
                    __syncthreads();
                    if (tid == 0) {
                        tot[max_k * 4 + 3]--; tot[k * 4 + 3]++; labels[far_i] = k;
                        for (int j = 0; j < 3; j++) {
                            float v = (float)data[3 * far_i + j];
                            centers[max_k * 3 + j] = __fsub_rn(centers[max_k * 3 + j], v);
                            centers[k * 3 + j] = __fadd_rn(centers[k * 3 + j], v);
                        }
                    }
                    __syncthreads();
                }
                if (tid == 0) {
                    for (int k = 0; k < K; k++) {
                        float scale = 1.f / (float)tot[k * 4 + 3];
                        for (int j = 0; j < 3; j++) centers[k * 3 + j] = __fmul_rn(centers[k * 3 + j], scale);
                        if (iter > 0) {
                            double d = 0;
                            for (int j = 0; j < 3; j++) { double t = (double)__fsub_rn(centers[k * 3 + j], old_centers[k * 3 + j]); d = __dadd_rn(d, __dmul_rn(t, t)); }
                            max_shift = fmax(max_shift, d);
                        }
                    }
                    redd[0] = max_shift;
                }
                __syncthreads();
                max_shift = redd[0];
                __syncthreads();
            }
            ++iter;
            bool last = (iter == max(maxCount, 2)) || (max_shift <= epsilon);
            if (last) {
                double s = 0;
                for (int i = tid; i < N; i += KM_T) s += (double)fsq3(data + 3 * i, &centers[3 * labels[i]]);
                compactness = block_sum_d(s, redd);
                break;
            } else {
                for (int i = tid; i < N; i += KM_T) {
                    float md = 0.f; int kb = 0;
                    for (int k = 0; k < K; k++) { float d = fsq3(data + 3 * i, &centers[3 * k]); if (k == 0 || md > d) { md = d; kb = k; } }
                    labels[i] = kb;
                }
                __syncthreads();
            }
        }
        if (compactness < best_compact) {
            best_compact = compactness;
            if (tid < K * 3) centers_out[tid] = centers[tid];
        }
        __syncthreads();
    }
    if (tid == 0) { *compact_out = best_compact; *status = 0; }
}
#endif

// ------------------------------------------------------------------------------------------------
// The same fit spread over KMB_B workgroups with a device-wide barrier between phases (the single workgroup above spends 13 ms
// of pure ALU time on one CU at the head of the whole path).  Every workgroup executes the same control flow: all decisions
// are taken from values that every block reads from global memory after a barrier.  Integer sums are order-free; the only
// floating-point reduction (compactness) is still evaluated by workgroup 0 with the tree of the 1024-thread version.
// ------------------------------------------------------------------------------------------------
#define KMB_B 64
#define KMB_T 256
struct KmGlobal {
    unsigned bar_count, bar_gen;
    long long acc[8];                              // rotating accumulators of the integer passes
    long long acc3[8][4];                          // the same for passes that sum three values at once (the three trials of a ++ centre)
    int ci3[4];                                    // the three trial centres of a ++ step
    long long tot[2][ORIP_MAX_LAYERS * 4];         // cluster sums (L, a, b, count), double-buffered by iteration parity
    unsigned long long key;                        // farthest-point search of the empty-cluster repair
    int ci;
    double compact;
    float slow_centers[ORIP_MAX_LAYERS * 3];
    long long blockpart[KMB_B];
};
__device__ __forceinline__ void km_grid_sync(KmGlobal* G) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = atomicAdd(&G->bar_gen, 0u);
        if (atomicAdd(&G->bar_count, 1u) == KMB_B - 1) { atomicExch(&G->bar_count, 0u); __threadfence(); atomicAdd(&G->bar_gen, 1u); }
        else while (atomicAdd(&G->bar_gen, 0u) == gen) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
}
__device__ __forceinline__ long long kmb_block_sum(long long v, long long* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long r = 0;
    for (int w = 0; w < KMB_T / 64; w++) r += red[w];
    return r;
}
// Attempts are independent but for the random draws, and every attempt makes the same number of them (one index, 3 (K - 1) doubles): `groups` groups of KMB_B
// workgroups take the attempts a = group, group + groups, ... side by side (the fit is bound by its grid barriers, not by the card), each with its own
// barrier, accumulators and work arrays; the host takes the first attempt with the smallest compactness, as the sequential loop does (02:46-49).
struct KmResult { float cen[ORIP_MAX_LAYERS * 3]; double compact; int attempt; int status; int pad[2]; };
__global__ __launch_bounds__(KMB_T) void k_kmeans_fit_mb(const u8* __restrict__ data, int N, int K, int attempts, int groups, int maxCount, double epsilon,
                                                          int32_t* __restrict__ dist_all, int32_t* __restrict__ labels_all,
                                                          double* __restrict__ dd_all, long long* __restrict__ parts_all, KmResult* __restrict__ res_all, KmGlobal* __restrict__ G_all) {
    __shared__ long long red[KMB_T / 64];
    __shared__ float centers[ORIP_MAX_LAYERS * 3], old_centers[ORIP_MAX_LAYERS * 3];
    __shared__ int csum[KMB_T / 64][ORIP_MAX_LAYERS * 4];
    __shared__ long long tot[ORIP_MAX_LAYERS * 4];
    __shared__ int pp_idx[ORIP_MAX_LAYERS];
    __shared__ double sh_shift;
    __shared__ double ptree[1024];
    const int tid = threadIdx.x, wave = tid >> 6, grp = blockIdx.x / KMB_B, bid = blockIdx.x % KMB_B;
    const int gtid = bid * KMB_T + tid, gsz = KMB_B * KMB_T;
    const int chunk = (N + gsz - 1) / gsz;
    const int lo = min(N, gtid * chunk), hi = min(N, lo + chunk);
    unsigned long long rng = 0xffffffffULL;   // identical in every thread
    double best_compact = 1.79769313486231570815e+308; int best_attempt = -1;
    int32_t* dist = dist_all + (size_t)grp * N; int32_t* labels = labels_all + (size_t)grp * N;      // (the trials of a ++ centre no longer write candidate distance arrays)
    double* dd = dd_all + (size_t)grp * N; long long* parts = parts_all + (size_t)grp * KMB_B * KMB_T;
    KmGlobal* G = G_all + grp; KmResult* res = res_all + grp;
    unsigned pass = 0;                        // selects the accumulator; slot pass+4 is cleared for later use
    // grid-wide integer sum of v (one value per thread); two barriers apart accumulators never collide
    auto grid_sum = [&](long long v) -> long long {
        long long bs = kmb_block_sum(v, red);
        const unsigned slot = pass & 7u;
        if (tid == 0) { atomicAdd((unsigned long long*)&G->acc[slot], (unsigned long long)bs); if (bid == 0) { const unsigned z = (pass + 4u) & 7u; G->acc[z] = 0; G->acc3[z][0] = G->acc3[z][1] = G->acc3[z][2] = 0; } }
        km_grid_sync(G);
        long long r = *(volatile long long*)&G->acc[slot];
        pass++;
        return r;
    };
    auto grid_sum3 = [&](long long v0, long long v1, long long v2, long long (&r)[3]) {
        const long long b0 = kmb_block_sum(v0, red), b1 = kmb_block_sum(v1, red), b2 = kmb_block_sum(v2, red);
        const unsigned slot = pass & 7u;
        if (tid == 0) {
            atomicAdd((unsigned long long*)&G->acc3[slot][0], (unsigned long long)b0); atomicAdd((unsigned long long*)&G->acc3[slot][1], (unsigned long long)b1);
            atomicAdd((unsigned long long*)&G->acc3[slot][2], (unsigned long long)b2);
            if (bid == 0) { const unsigned z = (pass + 4u) & 7u; G->acc[z] = 0; G->acc3[z][0] = G->acc3[z][1] = G->acc3[z][2] = 0; }
        }
        km_grid_sync(G);
        for (int j = 0; j < 3; j++) r[j] = *(volatile long long*)&G->acc3[slot][j];
        pass++;
    };
    for (int a = 0; a < attempts; a++) {
        if (a % groups != grp) { for (int q = 0; q < 1 + 6 * (K - 1); q++) km_next(rng); continue; }      // another group's attempt: its draws
        double compactness = 0;
        for (int iter = 0;;) {
            double max_shift = iter == 0 ? 1.79769313486231570815e+308 : 0.0;
            __syncthreads();
            if (tid < K * 3) { float t = centers[tid]; centers[tid] = old_centers[tid]; old_centers[tid] = t; }
            __syncthreads();
            if (iter == 0) {
                // ---------------- generateCentersPP ----------------
                if (bid == 0 && tid < 2 * ORIP_MAX_LAYERS * 4) (&G->tot[0][0])[tid] = 0;     // both sum buffers start an attempt empty (barriers follow)
                int c0 = (int)(km_next(rng) % (unsigned)N);
                if (tid == 0) pp_idx[0] = c0;
                long long ls = 0;
                for (int i = gtid; i < N; i += gsz) { int d = isq3(data + 3 * i, data + 3 * c0); dist[i] = d; ls += d; }
                long long sum0 = grid_sum(ls);
                // The three trials of a centre (02:47 -> cv::generateCentersPP, 3 trials per centre) share the distances they start from, so they run
                // side by side: ONE pass of partial sums, the three descents in one phase, the three candidate sums in one reduction -- 3 grid
                // barriers per centre instead of 9 (the fit is barrier-bound: 200 000 samples on 16 384 threads).  Every thread keeps its own
                // contiguous chunk [lo, hi) from the update of the distances through the partial sums to the candidate sums, so no barrier is
                // needed between a centre's update and the next centre's sums.  Same draws in the same order, same integer arithmetic.
                for (int k = 1; k < K; k++) {
                    double pj[3];
                    for (int j = 0; j < 3; j++) pj[j] = km_double(rng) * (double)sum0;
                    long long cs = 0;
                    for (int i = lo; i < hi; i++) cs += dist[i];
                    parts[gtid] = cs;
                    long long bs = kmb_block_sum(cs, red);
                    if (tid == 0) G->blockpart[bid] = bs;
                    km_grid_sync(G);
                    if (bid == 0 && tid < 64) {
                        for (int j = 0; j < 3; j++) {
                            const double p = pj[j];
                            // ci = first index with inclusive prefix >= p, else N-1.  First wave of block 0 descends: block totals, the threads of the crossing
                            // block, then the samples of the crossing chunk; every test is "(double)(inclusive integer prefix) >= p" on exact integers
                            auto wave_incl = [&](long long v) { for (int o = 1; o < 64; o <<= 1) { long long t = __shfl_up(v, o, 64); if (tid >= o) v += t; } return v; };
                            auto first_cross = [&](long long base, long long v, bool valid, long long& before) -> int {   // lane index of the first crossing, -1: none
                                const long long inc = wave_incl(valid ? v : 0);
                                const unsigned long long m = __ballot(valid && (double)(base + inc) >= p);
                                const int f = m ? __ffsll((long long)m) - 1 : -1;
                                const int src = f >= 0 ? f : 63;
                                const long long incf = __shfl(inc, src, 64), vf = __shfl(valid ? v : 0, src, 64);
                                before = base + (f >= 0 ? incf - vf : incf);        // prefix in front of the crossing lane / whole window when none
                                return f;
                            };
                            long long run = 0; int ci = N - 1;
                            int b = -1;
                            for (int b0 = 0; b0 < KMB_B && b < 0; b0 += 64) {
                                long long before; const bool valid = b0 + tid < KMB_B;
                                const int f = first_cross(run, valid ? *(volatile long long*)&G->blockpart[b0 + tid] : 0, valid, before);
                                run = before; if (f >= 0) b = b0 + f;
                            }
                            if (b >= 0) {
                                int t = -1;
                                for (int t0 = 0; t0 < KMB_T && t < 0; t0 += 64) {
                                    long long before;
                                    const int f = first_cross(run, *(volatile long long*)&parts[b * KMB_T + t0 + tid], true, before);
                                    run = before; if (f >= 0) t = t0 + f;
                                }
                                if (t >= 0) {
                                    const int l2 = min(N, (b * KMB_T + t) * chunk), h2 = min(N, l2 + chunk);
                                    for (int i0 = l2; i0 < h2 && ci == N - 1; i0 += 64) {
                                        long long before; const bool valid = i0 + tid < h2;
                                        const int f = first_cross(run, valid ? (long long)dist[i0 + tid] : 0, valid, before);
                                        run = before; if (f >= 0) { ci = i0 + f; break; }
                                    }
                                }
                            }
                            if (ci > N - 1) ci = N - 1;
                            if (tid == 0) G->ci3[j] = ci;
                        }
                    }
                    km_grid_sync(G);
                    int cj[3]; for (int j = 0; j < 3; j++) cj[j] = *(volatile int*)&G->ci3[j];
                    long long sj[3] = {0, 0, 0};
                    for (int i = lo; i < hi; i++) {
                        const int d0 = dist[i];
                        for (int j = 0; j < 3; j++) sj[j] += min(isq3(data + 3 * i, data + 3 * cj[j]), d0);
                    }
                    long long Sj[3]; grid_sum3(sj[0], sj[1], sj[2], Sj);
                    long long bestSum = 0x7fffffffffffffffLL; int bestCenter = -1;
                    for (int j = 0; j < 3; j++) if (Sj[j] < bestSum) { bestSum = Sj[j]; bestCenter = cj[j]; }
                    if (tid == 0) pp_idx[k] = bestCenter;
                    sum0 = bestSum;
                    if (k + 1 < K) for (int i = lo; i < hi; i++) dist[i] = min(isq3(data + 3 * i, data + 3 * bestCenter), dist[i]);      // own chunk: read next by this thread only, then (after a barrier) by the descent
                    __syncthreads();
                }
                __syncthreads();
                if (tid < K * 3) centers[tid] = (float)data[3 * pp_idx[tid / 3] + tid % 3];
                __syncthreads();
            } else {
                // ---------------- recompute centres from labels ----------------
                const int par = iter & 1;
                for (int i = tid; i < (KMB_T / 64) * ORIP_MAX_LAYERS * 4; i += KMB_T) (&csum[0][0])[i] = 0;
                __syncthreads();
                for (int i = gtid; i < N; i += gsz) {
                    int k = labels[i];
                    atomicAdd(&csum[wave][k * 4 + 0], (int)data[3 * i]);
                    atomicAdd(&csum[wave][k * 4 + 1], (int)data[3 * i + 1]);
                    atomicAdd(&csum[wave][k * 4 + 2], (int)data[3 * i + 2]);
                    atomicAdd(&csum[wave][k * 4 + 3], 1);
                }
                __syncthreads();
                if (tid < K * 4) {
                    long long t = 0; for (int w = 0; w < KMB_T / 64; w++) t += csum[w][tid];
                    atomicAdd((unsigned long long*)&G->tot[par][tid], (unsigned long long)t);
                    if (bid == 0) G->tot[par ^ 1][tid] = 0;        // the other buffer was last read two barriers ago
                }
                if (bid == 0 && tid == 0) G->key = 0;
                km_grid_sync(G);
                if (tid < K * 4) tot[tid] = *(volatile long long*)&G->tot[par][tid];
                __syncthreads();
                // float accumulation in sample order is exact (== integer sum) while every partial sum < 2^24
                bool bad = false;
                for (int k = 0; k < K; k++) for (int j = 0; j < 3; j++) if (tot[k * 4 + j] >= (1LL << 24)) bad = true;
                if (bad) {
                    // rare slow path: literal sequential float32 accumulation by one lane
                    if (bid == 0 && tid == 0) {
                        float acc[ORIP_MAX_LAYERS * 3];
                        for (int q = 0; q < K * 3; q++) acc[q] = 0.f;
                        for (int i = 0; i < N; i++) { int k = labels[i]; for (int j = 0; j < 3; j++) acc[k * 3 + j] = __fadd_rn(acc[k * 3 + j], (float)data[3 * i + j]); }
                        for (int q = 0; q < K * 3; q++) G->slow_centers[q] = acc[q];
                    }
                    km_grid_sync(G);
                    if (tid < K * 3) centers[tid] = *(volatile float*)&G->slow_centers[tid];
                } else if (tid < K * 3) centers[tid] = (float)tot[(tid / 3) * 4 + tid % 3];
                __syncthreads();
                // empty-cluster repair (sequential over k, as the reference)
                for (int k = 0; k < K; k++) {
                    if (tot[k * 4 + 3] != 0) continue;      // uniform: every block holds the same totals
                    int max_k = 0;
                    for (int k1 = 1; k1 < K; k1++) if (tot[max_k * 4 + 3] < tot[k1 * 4 + 3]) max_k = k1;
                    float nb[3]; float scale = 1.f / (float)tot[max_k * 4 + 3];
                    for (int j = 0; j < 3; j++) nb[j] = __fmul_rn(centers[max_k * 3 + j], scale);
                    // farthest point: max_dist <= dist  => last index among maxima
                    unsigned long long key = 0;
                    for (int i = gtid; i < N; i += gsz) {
                        if (labels[i] != max_k) continue;
                        float d = fsq3(data + 3 * i, nb);
                        unsigned long long kk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
                        if (kk >= key) key = kk;
                    }
                    atomicMax(&G->key, key);
                    km_grid_sync(G);
                    const int far_i = (int)(*(volatile unsigned long long*)&G->key & 0xffffffffu);
                    __syncthreads();
                    if (tid == 0) {
                        tot[max_k * 4 + 3]--; tot[k * 4 + 3]++;
                        if (bid == 0) labels[far_i] = k;
                        for (int j = 0; j < 3; j++) {
                            float v = (float)data[3 * far_i + j];
                            centers[max_k * 3 + j] = __fsub_rn(centers[max_k * 3 + j], v);
                            centers[k * 3 + j] = __fadd_rn(centers[k * 3 + j], v);
                        }
                    }
                    km_grid_sync(G);                           // everybody has read the key before it is cleared
                    if (bid == 0 && tid == 0) G->key = 0;
                    km_grid_sync(G);
                }
                if (tid == 0) {
                    for (int k = 0; k < K; k++) {
                        float scale = 1.f / (float)tot[k * 4 + 3];
                        for (int j = 0; j < 3; j++) centers[k * 3 + j] = __fmul_rn(centers[k * 3 + j], scale);
                        if (iter > 0) {
                            double d = 0;
                            for (int j = 0; j < 3; j++) { double t = (double)__fsub_rn(centers[k * 3 + j], old_centers[k * 3 + j]); d = __dadd_rn(d, __dmul_rn(t, t)); }
                            max_shift = fmax(max_shift, d);
                        }
                    }
                    sh_shift = max_shift;
                }
                __syncthreads();
                max_shift = sh_shift;
                __syncthreads();
            }
            ++iter;
            bool last = (iter == max(maxCount, 2)) || (max_shift <= epsilon);
            if (last) {
                for (int i = gtid; i < N; i += gsz) dd[i] = (double)fsq3(data + 3 * i, &centers[3 * labels[i]]);
                km_grid_sync(G);
                if (bid == 0) {
                    // the reduction tree of the 1024-thread version: strided partials, shfl_down tree per 64, then 16 sequential adds
                    for (int v = tid; v < 1024; v += KMB_T) { double s = 0; for (int i = v; i < N; i += 1024) s += dd[i]; ptree[v] = s; }
                    __syncthreads();
                    for (int o = 32; o > 0; o >>= 1) {
                        for (int q = tid; q < 16 * o; q += KMB_T) { const int w = q / o, l = q % o; ptree[w * 64 + l] += ptree[w * 64 + l + o]; }
                        __syncthreads();
                    }
                    if (tid == 0) { double r = 0; for (int w = 0; w < 16; w++) r += ptree[w * 64]; G->compact = r; }
                }
                km_grid_sync(G);
                compactness = *(volatile double*)&G->compact;
                break;
            } else {
                for (int i = gtid; i < N; i += gsz) {
                    float md = 0.f; int kb = 0;
                    for (int k = 0; k < K; k++) { float d = fsq3(data + 3 * i, &centers[3 * k]); if (k == 0 || md > d) { md = d; kb = k; } }
                    labels[i] = kb;
                }
                km_grid_sync(G);
            }
        }
        if (compactness < best_compact) {
            best_compact = compactness; best_attempt = a;
            if (bid == 0 && tid < K * 3) res->cen[tid] = centers[tid];
        }
        __syncthreads();
    }
    if (bid == 0 && tid == 0) { res->compact = best_compact; res->attempt = best_attempt; res->status = 0; }
}

// ------------------------------------------------------------------------------------------------
// Morphology: one erode/dilate pass with a k x k structuring element given as a bit mask (bit i*k+j).
// Out-of-image neighbours never win (cv2 default border).  If labels_mode, the source is the label
// map and the pixel value is (label == layer) ? 255 : 0  -- fuses `mask=(labels==k)*255` (02:150).
// grid.z = layer.
// ------------------------------------------------------------------------------------------------
#define MT_X 64
#define MT_Y 16
__global__ __launch_bounds__(256) void k_morph_pass(const u8* __restrict__ src, u8* __restrict__ dst, int H, int W, int k,
                                                     unsigned long long se, int dilate, int labels_mode) {
    __shared__ u8 tile[MT_Y + 6][MT_X + 8];
    const int r = k >> 1;
    const int layer = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const u8* s = labels_mode ? src : src + plane * layer;
    u8* d = dst + plane * layer;
    const int x0 = blockIdx.x * MT_X, y0 = blockIdx.y * MT_Y;
    const u8 neutral = dilate ? 0 : 255;
    for (int i = threadIdx.x; i < (MT_Y + 2 * r) * (MT_X + 2 * r); i += blockDim.x) {
        int ty = i / (MT_X + 2 * r), tx = i % (MT_X + 2 * r);
        int y = y0 + ty - r, x = x0 + tx - r;
        u8 v = neutral;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            u8 raw = s[(size_t)y * W + x];
            v = labels_mode ? (raw == layer ? 255 : 0) : raw;
        }
        tile[ty][tx] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MT_Y * MT_X; i += blockDim.x) {
        int ty = i / MT_X, tx = i % MT_X;
        int y = y0 + ty, x = x0 + tx;
        if (y >= H || x >= W) continue;
        int v = neutral;
        for (int a = 0; a < k; a++)
            for (int b = 0; b < k; b++)
                if ((se >> (a * k + b)) & 1ULL) { int t = tile[ty + a][tx + b]; v = dilate ? max(v, t) : min(v, t); }
        d[(size_t)y * W + x] = (u8)v;
    }
}

static unsigned long long make_se_bits(int shape, int k) {
    unsigned long long se = 0;
    if (shape == 0) { for (int i = 0; i < k * k; i++) se |= 1ULL << i; return se; }
    int r = k / 2, c = k / 2; double inv_r2 = r ? 1.0 / ((double)r * r) : 0;
    for (int i = 0; i < k; i++) {
        int dy = i - r, j1 = 0, j2 = 0;
        if (std::abs(dy) <= r) { int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2)); j1 = std::max(c - dx, 0); j2 = std::min(c + dx + 1, k); }
        for (int j = j1; j < j2; j++) se |= 1ULL << (i * k + j);
    }
    return se;
}

// ------------------------------------------------------------------------------------------------
// Binary fast path of the same chain: a 0/255 mask packs to one bit per pixel (64 pixels per word, 2 MB per 4096^2 layer, so the
// whole chain of passes runs out of L2), dilation = OR and erosion = AND of shifted rows.  Out-of-image pixels are neutral
// (erode: ones, dilate: zeros), exactly as in k_morph_pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bits_pack(const u8* __restrict__ src, unsigned long long* __restrict__ bits, int H, int W, int Ww, int labels_mode) {
    const int layer = blockIdx.z;
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= (size_t)H * Ww) return;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww);
    const u8* row = (labels_mode ? src : src + (size_t)H * W * layer) + (size_t)y * W;
    unsigned long long b = 0;
    const int x0 = xw * 64, n = min(64, W - x0);
    for (int j = 0; j < n; j++) { const u8 v = row[x0 + j]; if (labels_mode ? (v == layer) : (v != 0)) b |= 1ULL << j; }
    bits[(size_t)H * Ww * layer + wi] = b;
}
// labels -> K bit planes in ONE pass over the label plane (rows that are multiples of 64 wide): a thread takes the 64 labels of a word as four 16-byte
// loads and emits the word of every layer -- per layer and 4 labels an exact "byte equals l" test on the 32-bit word (zero-byte detection of w ^ l l l l) and a
// multiply that gathers the four flag bits.  k_bits_pack read the plane once per layer, a byte per load and lane (0.25 ms at 4096^2 x 8 in front of the walks).
__global__ __launch_bounds__(256) void k_bits_pack_labels(const u8* __restrict__ labels, unsigned long long* __restrict__ bits, int H, int W, int Ww, int K) {
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x, nw = (size_t)H * Ww;
    if (wi >= nw) return;
    const uint4* p = reinterpret_cast<const uint4*>(labels + wi * 64);          // W == 64 * Ww: word wi holds the pixels 64 wi .. 64 wi + 63 of the plane
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint4 v = p[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    for (int l = 0; l < K; l++) {
        const uint32_t rep = 0x01010101u * (uint32_t)l;
        unsigned long long b = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t x = w[q] ^ rep;
            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);          // 0x80 in every byte of x that is zero, 0 elsewhere
            const uint32_t nib = ((z >> 7) * 0x00204081u) >> 21 & 0xfu;                        // bits 0, 8, 16, 24 -> bits 0 .. 3
            b |= (unsigned long long)nib << (4 * q);
        }
        bits[nw * l + wi] = b;
    }
}
__global__ __launch_bounds__(256) void k_bits_unpack(const unsigned long long* __restrict__ bits, u8* __restrict__ dst, int H, int W, int Ww) {
    const int layer = blockIdx.z;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;                // four pixels per thread
    const size_t plane = (size_t)H * W;
    if (p * 4 >= plane) return;
    u8* d = dst + plane * layer;
    for (int j = 0; j < 4; j++) {
        const size_t q = p * 4 + j; if (q >= plane) break;
        const int y = (int)(q / W), x = (int)(q % W);
        d[q] = ((bits[((size_t)layer * H + y) * Ww + (x >> 6)] >> (x & 63)) & 1ULL) ? 255 : 0;
    }
}
__global__ __launch_bounds__(256) void k_morph_bits(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int H, int W, int Ww, int k,
                                                     unsigned long long se, int dilate) {
    const int layer = blockIdx.z;
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= (size_t)H * Ww) return;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww), r = k >> 1;
    const unsigned long long* s = src + (size_t)H * Ww * layer;
    const unsigned long long neutral = dilate ? 0ULL : ~0ULL;
    const int tail = W - (Ww - 1) * 64;                                       // valid bits of the last word of a row
    const unsigned long long tail_mask = tail >= 64 ? ~0ULL : ((1ULL << tail) - 1ULL);
    auto word = [&](int yy, int xx) -> unsigned long long {                   // row yy, word xx with out-of-image bits made neutral
        if (yy < 0 || yy >= H || xx < 0 || xx >= Ww) return neutral;
        unsigned long long w = s[(size_t)yy * Ww + xx];
        if (xx == Ww - 1) w = dilate ? (w & tail_mask) : (w | ~tail_mask);
        return w;
    };
    unsigned long long out = neutral;
    for (int a = 0; a < k; a++) {
        if (!((se >> (a * k)) & ((1ULL << k) - 1ULL))) continue;
        const int yy = y + a - r;
        const unsigned long long L = word(yy, xw - 1), M = word(yy, xw), Rw = word(yy, xw + 1);
        for (int b = 0; b < k; b++) {
            if (!((se >> (a * k + b)) & 1ULL)) continue;
            const int sft = b - r;                                             // output bit x takes source bit x + sft
            unsigned long long v = sft == 0 ? M : (sft > 0 ? ((M >> sft) | (Rw << (64 - sft))) : ((M << -sft) | (L >> (64 + sft))));
            out = dilate ? (out | v) : (out & v);
        }
    }
    dst[(size_t)H * Ww * layer + wi] = out;
}
// 1 if any byte of the planes is neither 0 nor 255
__global__ __launch_bounds__(256) void k_not_binary(const u8* __restrict__ src, size_t n, int* __restrict__ flag) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    bool bad = false;
    for (int j = 0; j < 16 && i + j < n; j++) { const u8 v = src[i + j]; bad |= (v != 0 && v != 255); }
    if (bad) *flag = 1;
}

// open (erode^n, dilate^n) then close (dilate^n, erode^n); src plane(s) -> dst plane(s); uses tmpA as ping-pong
// unpack == false: when the chain ran on bit planes, leave the result there (c->morphed_bits) and do not write the byte planes (stage 03)
int orip_morph_open_close(orip_ctx* c, const u8* src, u8* dst, int K, int shape, int k, int open_iters, int close_iters, bool labels_mode, bool unpack = true) {
    if (k < 1 || k > 7 || (k & 1) == 0) ORIP_FAIL(c, "structuring element size %d unsupported (odd, 1..7)", k);
    int H = c->H, W = c->W; size_t plane = (size_t)H * W;
    unsigned long long se = make_se_bits(shape, k);
    std::vector<int> passes;   // 0 erode, 1 dilate
    for (int i = 0; i < std::max(open_iters, 0); i++) passes.push_back(0);
    for (int i = 0; i < std::max(open_iters, 0); i++) passes.push_back(1);
    for (int i = 0; i < std::max(close_iters, 0); i++) passes.push_back(1);
    for (int i = 0; i < std::max(close_iters, 0); i++) passes.push_back(0);
    if (passes.empty()) passes.push_back(-1);   // identity copy via 1x1 "erode"
    // binary input (label masks always are; explicit masks are checked): the chain runs on bit planes
    const bool have_bits = !labels_mode && c->mask_bits != nullptr && src == c->masks.as<u8>() && !getenv("ORIP_MORPH_BYTES");   // stage 02 left them
    bool binary = labels_mode || have_bits;
    if (!binary && !getenv("ORIP_MORPH_BYTES")) {
        int* d_flag = LN(c).flags.as<int>() + 12;
        HIPC(c, hipMemsetAsync(d_flag, 0, 4, LN(c).stream));
        hipLaunchKernelGGL(k_not_binary, dim3((unsigned)cdiv((int64_t)(plane * K), 4096)), dim3(256), 0, LN(c).stream, src, plane * K, d_flag);
        int h = 1; HIPC(c, hipMemcpyAsync(&h, d_flag, 4, hipMemcpyDeviceToHost, LN(c).stream)); HIPC(c, hipStreamSynchronize(LN(c).stream));
        binary = h == 0;
    }
    if (binary && passes[0] >= 0 && !getenv("ORIP_MORPH_BYTES")) {
        const int Ww = (W + 63) >> 6; const size_t nw = (size_t)H * Ww;
        HIPC(c, c->tmpA.ensure(std::max(plane * K, nw * K * 16 + 64)));
        unsigned long long* A = c->tmpA.as<unsigned long long>(); unsigned long long* B = A + nw * K;
        dim3 gw((unsigned)cdiv((int64_t)nw, 256), 1, K), block(256);
        if (have_bits) { if (c->mask_bits == (const void*)B) std::swap(A, B); }        // the planes are already there (first or second half)
        else if (labels_mode && (W & 63) == 0 && !getenv("ORIP_PACK_BYTES")) hipLaunchKernelGGL(k_bits_pack_labels, dim3(gw.x), block, 0, LN(c).stream, src, A, H, W, Ww, K);
        else hipLaunchKernelGGL(k_bits_pack, gw, block, 0, LN(c).stream, src, A, H, W, Ww, labels_mode ? 1 : 0);
        c->mask_bits = nullptr;
        for (size_t i = 0; i < passes.size(); i++) {
            ProfScope ps(c, "k_morph_bits");
            hipLaunchKernelGGL(k_morph_bits, gw, block, 0, LN(c).stream, A, B, H, W, Ww, k, se, passes[i] == 1 ? 1 : 0);
            std::swap(A, B);
        }
        if (unpack && (W & 63) == 0 && !getenv("ORIP_PACK_BYTES")) hipLaunchKernelGGL(k_bits_expand16, dim3((unsigned)cdiv((int64_t)nw * 4, 256), 1, K), block, 0, LN(c).stream, A, dst, nw);
        else if (unpack) hipLaunchKernelGGL(k_bits_unpack, dim3((unsigned)cdiv((int64_t)plane, 1024), 1, K), block, 0, LN(c).stream, A, dst, H, W, Ww);
        else c->morphed_bits = A;
        HIPC(c, hipGetLastError());
        if (labels_mode && dst == c->masks.as<u8>()) c->mask_bits = A;      // stage 03 can start from the bit planes
        return 0;
    }
    c->mask_bits = nullptr;                       // the byte passes ping-pong through tmpA
    HIPC(c, c->tmpA.ensure(plane * K));
    dim3 grid(cdiv(W, MT_X), cdiv(H, MT_Y), K), block(256);
    const u8* cur = src; bool lm = labels_mode;
    int np_ = (int)passes.size();
    for (int i = 0; i < np_; i++) {
        // choose outputs so that the last pass lands in dst
        u8* out = ((np_ - 1 - i) % 2 == 0) ? dst : c->tmpA.as<u8>();
        if (out == cur) ORIP_FAIL(c, "internal: in-place morph pass");
        int kk = passes[i] < 0 ? 1 : k; unsigned long long ss = passes[i] < 0 ? 1ULL : se;
        {
            ProfScope ps(c, "k_morph_pass");
            hipLaunchKernelGGL(k_morph_pass, grid, block, 0, LN(c).stream, cur, out, H, W, kk, ss, passes[i] == 1 ? 1 : 0, lm ? 1 : 0);
        }
        cur = out; lm = false;
    }
    HIPC(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------
extern "C" int orip_set_image(orip_ctx* c, const uint8_t* bgr, int H, int W) {
    orip_enter(c);
    c->mask_bits = nullptr;
    if (!bgr || H <= 0 || W <= 0) ORIP_FAIL(c, "bad image %dx%d", W, H);
    ORIP_TRY(orip_raster02_lab_tables(c));
    c->H = H; c->W = W;
    HIPC(c, c->image.ensure((size_t)H * W * 3 + 16));
    HIPC(c, hipMemcpyAsync(c->image.p, bgr, (size_t)H * W * 3, hipMemcpyHostToDevice, LN(c).stream));
    return 0;
}

extern "C" int orip_lab_of(orip_ctx* c, const int64_t* idx, int64_t n, uint8_t* lab_out) {
    orip_enter(c);
    if (!c->image.p) ORIP_FAIL(c, "no image set");
    if (!idx) n = (int64_t)c->H * c->W;
    HIPC(c, c->tmpB.ensure((size_t)n * 3 + 16));
    if (idx) { HIPC(c, c->tmpC.ensure((size_t)n * 8)); HIPC(c, hipMemcpyAsync(c->tmpC.p, idx, (size_t)n * 8, hipMemcpyHostToDevice, LN(c).stream)); }
    hipLaunchKernelGGL(k_lab_gather, dim3(std::min<int64_t>(2048, cdiv(n, 256))), dim3(256), 0, LN(c).stream, c->image.as<u8>(),
                       idx ? c->tmpC.as<int64_t>() : nullptr, n, c->tmpB.as<u8>(), c->lab_tabs.as<LabTabs>());
    HIPC(c, hipGetLastError());
    HIPC(c, hipMemcpyAsync(lab_out, c->tmpB.p, (size_t)n * 3, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}

static int kmeans_fit_impl(orip_ctx* c, bool rgb, const int64_t* sample_idx, int64_t n_idx, int K, int attempts, int max_iter, double eps,
                           float* centers_out, double* compactness_out) {
    orip_enter(c);
    c->mask_bits = nullptr;
    if (!c->image.p) ORIP_FAIL(c, "no image set");
    if (K < 1 || K > ORIP_MAX_LAYERS) ORIP_FAIL(c, "K=%d out of range 1..%d", K, ORIP_MAX_LAYERS);
    int64_t N = sample_idx ? n_idx : (int64_t)c->H * c->W;
    if (N < K || N > 0x7fffffff) ORIP_FAIL(c, "bad sample count %lld", (long long)N);
    HIPC(c, c->tmpB.ensure((size_t)N * 3 + 16));
    if (sample_idx) { HIPC(c, c->tmpC.ensure((size_t)N * 8)); HIPC(c, hipMemcpyAsync(c->tmpC.p, sample_idx, (size_t)N * 8, hipMemcpyHostToDevice, LN(c).stream)); }
    if (rgb) {
        hipLaunchKernelGGL(k_rgb_gather, dim3(std::min<int64_t>(2048, cdiv(N, 256))), dim3(256), 0, LN(c).stream, c->image.as<u8>(),
                           sample_idx ? c->tmpC.as<int64_t>() : nullptr, N, c->tmpB.as<u8>());
    } else {
        ProfScope ps(c, "k_lab_gather");
        hipLaunchKernelGGL(k_lab_gather, dim3(std::min<int64_t>(2048, cdiv(N, 256))), dim3(256), 0, LN(c).stream, c->image.as<u8>(),
                           sample_idx ? c->tmpC.as<int64_t>() : nullptr, N, c->tmpB.as<u8>(), c->lab_tabs.as<LabTabs>());
    }
    HIPC(c, c->tmpD.ensure((size_t)N * 4 * 4 + 256));
    int32_t* base = c->tmpD.as<int32_t>();
    HIPC(c, LN(c).flags.ensure(1024));
    attempts = std::max(attempts, 1);
    double epsilon = std::max(eps, 0.0); epsilon *= epsilon;
    int maxCount = std::min(std::max(max_iter, 2), 100);
    if (K == 1) { attempts = 1; maxCount = 2; }
    float* d_centers = (float*)((char*)LN(c).flags.p + 256);
    double* d_comp = (double*)((char*)LN(c).flags.p + 512);
    int* d_status = (int*)LN(c).flags.p;
    HIPC(c, hipMemsetAsync(LN(c).flags.p, 0xff, 4, LN(c).stream));
#ifdef ORIP_VARIANTS
    if (ORIP_VARIANT("ORIP_KMEANS_1WG")) {          // the single-workgroup version (test hook: both must give the same centres)
        ProfScope ps(c, "k_kmeans_fit");
        hipLaunchKernelGGL(k_kmeans_fit, dim3(1), dim3(KM_T), 0, LN(c).stream, c->tmpB.as<u8>(), (int)N, K, attempts, maxCount, epsilon,
                           base, base + N, base + 2 * N, base + 3 * N, d_centers, d_comp, d_status);
    } else
#endif
    {
        const int groups = std::min(attempts, 4);
        HIPC(c, c->tmpD.ensure((size_t)N * 4 * 2 * groups + 256));
        base = c->tmpD.as<int32_t>();
        const size_t per = (size_t)N * 8 + (size_t)KMB_B * KMB_T * 8;
        HIPC(c, c->tmpA.ensure(per * groups + (sizeof(KmGlobal) + sizeof(KmResult)) * groups + 256));
        double* dd = c->tmpA.as<double>(); long long* parts = (long long*)(dd + (size_t)N * groups);
        KmGlobal* G = (KmGlobal*)(parts + (size_t)KMB_B * KMB_T * groups); KmResult* res = (KmResult*)(G + groups);
        HIPC(c, hipMemsetAsync(G, 0, sizeof(KmGlobal) * groups, LN(c).stream));
        HIPC(c, hipMemsetAsync(res, 0xff, sizeof(KmResult) * groups, LN(c).stream));
        {
            ProfScope ps(c, "k_kmeans_fit");
            hipLaunchKernelGGL(k_kmeans_fit_mb, dim3(KMB_B * groups), dim3(KMB_T), 0, LN(c).stream, c->tmpB.as<u8>(), (int)N, K, attempts, groups, maxCount, epsilon,
                               base, base + (size_t)N * groups, dd, parts, res, G);
        }
        HIPC(c, hipGetLastError());
        KmResult hr[4];
        HIPC(c, hipMemcpyAsync(hr, res, sizeof(KmResult) * groups, hipMemcpyDeviceToHost, LN(c).stream));
        HIPC(c, hipStreamSynchronize(LN(c).stream));
        int bg = -1;
        for (int g = 0; g < groups; g++) {
            if (hr[g].status != 0) ORIP_FAIL(c, "kmeans kernel did not complete (group %d, status %d)", g, hr[g].status);
            if (bg < 0 || hr[g].compact < hr[bg].compact || (hr[g].compact == hr[bg].compact && hr[g].attempt < hr[bg].attempt)) bg = g;
        }
        memcpy(centers_out, hr[bg].cen, sizeof(float) * K * 3);
        if (compactness_out) *compactness_out = hr[bg].compact;
        return 0;
    }
    HIPC(c, hipGetLastError());
    struct { float cen[ORIP_MAX_LAYERS * 3]; } hc; double comp; int st;
    HIPC(c, hipMemcpyAsync(hc.cen, d_centers, sizeof(float) * K * 3, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipMemcpyAsync(&comp, d_comp, 8, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    if (st != 0) ORIP_FAIL(c, "kmeans kernel did not complete (status %d)", st);
    memcpy(centers_out, hc.cen, sizeof(float) * K * 3);
    if (compactness_out) *compactness_out = comp;
    return 0;
}

extern "C" int orip_kmeans_fit(orip_ctx* c, const int64_t* sample_idx, int64_t n_idx, int K, int attempts, int max_iter, double eps,
                               float* centers_out, double* compactness_out) {
    return kmeans_fit_impl(c, false, sample_idx, n_idx, K, attempts, max_iter, eps, centers_out, compactness_out);
}
extern "C" int orip_kmeans_fit_rgb(orip_ctx* c, const int64_t* sample_idx, int64_t n_idx, int K, int attempts, int max_iter, double eps,
                                   float* centers_out, double* compactness_out) {
    return kmeans_fit_impl(c, true, sample_idx, n_idx, K, attempts, max_iter, eps, centers_out, compactness_out);
}

// process_colors.py assign_labels: labels u8 [H,W] of the image against an RGB palette, resident (orip_get_labels) and optionally on the host
extern "C" int orip_assign_palette(orip_ctx* c, const uint8_t* palette_rgb, int K, uint8_t* labels_out, int64_t* counts_out) {
    orip_enter(c);
    c->mask_bits = nullptr;
    if (!c->image.p) ORIP_FAIL(c, "no image set");
    if (!palette_rgb || K < 1 || K > ORIP_MAX_LAYERS) ORIP_FAIL(c, "K=%d out of range 1..%d", K, ORIP_MAX_LAYERS);
    const int64_t npx = (int64_t)c->H * c->W;
    hipStream_t s = LN(c).stream;
    HIPC(c, c->labels.ensure((size_t)npx + 16));
    HIPC(c, LN(c).flags.ensure(1024));
    u8* d_pal = (u8*)LN(c).flags.p + 768;
    HIPC(c, hipMemcpyAsync(d_pal, palette_rgb, (size_t)K * 3, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(c, "k_assign_palette");
        hipLaunchKernelGGL(k_assign_palette, dim3((unsigned)cdiv(cdiv(npx, 4), 256)), dim3(256), 0, s, c->image.as<u8>(), npx, d_pal, K, c->labels.as<u8>());
    }
    HIPC(c, hipGetLastError());
    if (counts_out) {
        unsigned long long* d_counts = (unsigned long long*)((char*)LN(c).flags.p + 512);
        HIPC(c, hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * ORIP_MAX_LAYERS, s));
        hipLaunchKernelGGL(k_count_labels, dim3(512), dim3(256), 0, s, c->labels.as<u8>(), npx, d_counts);
        unsigned long long h[ORIP_MAX_LAYERS];
        HIPC(c, hipMemcpyAsync(h, d_counts, sizeof(h), hipMemcpyDeviceToHost, s));
        HIPC(c, hipStreamSynchronize(s));
        for (int k = 0; k < K; k++) counts_out[k] = (int64_t)h[k];
    }
    if (labels_out) HIPC(c, hipMemcpyAsync(labels_out, c->labels.p, (size_t)npx, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    return 0;
}

extern "C" int orip_extract_layers(orip_ctx* c, const float* centers, int K, int open_iters, int close_iters,
                                   float* centers_sorted_out, int64_t* counts_out) {
    orip_enter(c);
    if (!c->image.p) ORIP_FAIL(c, "no image set");
    if (K < 1 || K > ORIP_MAX_LAYERS) ORIP_FAIL(c, "K=%d out of range", K);
    int H = c->H, W = c->W; int64_t npx = (int64_t)H * W;
    // order = argsort(L) (stable), lut[order] = arange (02:121-127)
    std::vector<int> order(K);
    for (int i = 0; i < K; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return centers[3 * a] < centers[3 * b]; });
    Centers cs; cs.K = K;
    for (int i = 0; i < K * 3; i++) cs.c[i] = centers[i];
    for (int r = 0; r < K; r++) cs.lut[order[r]] = (uint8_t)r;
    if (centers_sorted_out) for (int r = 0; r < K; r++) for (int j = 0; j < 3; j++) centers_sorted_out[3 * r + j] = centers[3 * order[r] + j];
    c->K = K;
    HIPC(c, c->labels.ensure((size_t)npx + 16));
    HIPC(c, c->masks.ensure((size_t)npx * K));
    {
        ProfScope ps(c, "k_lab_assign");
        hipLaunchKernelGGL(k_lab_assign, dim3(std::min<int64_t>(4096, std::max<int64_t>(1, cdiv(npx / 4, 256)))), dim3(256), 0, LN(c).stream,
                           c->image.as<u8>(), c->labels.as<u8>(), npx, c->lab_tabs.as<LabTabs>(), cs);
    }
    HIPC(c, hipGetLastError());
    if (counts_out) {
        HIPC(c, LN(c).flags.ensure(1024));
        unsigned long long* d_cnt = (unsigned long long*)((char*)LN(c).flags.p + 768);
        HIPC(c, hipMemsetAsync(d_cnt, 0, ORIP_MAX_LAYERS * 8, LN(c).stream));
        hipLaunchKernelGGL(k_count_labels, dim3(1024), dim3(256), 0, LN(c).stream, c->labels.as<u8>(), npx, d_cnt);
        unsigned long long h[ORIP_MAX_LAYERS];
        HIPC(c, hipMemcpyAsync(h, d_cnt, ORIP_MAX_LAYERS * 8, hipMemcpyDeviceToHost, LN(c).stream));
        HIPC(c, hipStreamSynchronize(LN(c).stream));
        for (int k = 0; k < K; k++) counts_out[k] = (int64_t)h[k];
    }
    return orip_morph_open_close(c, c->labels.as<u8>(), c->masks.as<u8>(), K, 0, 3, open_iters, close_iters, true);
}

extern "C" int orip_get_labels(orip_ctx* c, uint8_t* out) {
    orip_enter(c);
    if (!c->labels.p) ORIP_FAIL(c, "no labels resident");
    HIPC(c, hipMemcpyAsync(out, c->labels.p, (size_t)c->H * c->W, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_get_mask(orip_ctx* c, int layer, uint8_t* out) {
    orip_enter(c);
    if (!c->masks.p || layer < 0 || layer >= c->K) ORIP_FAIL(c, "no mask for layer %d", layer);
    size_t plane = (size_t)c->H * c->W;
    HIPC(c, hipMemcpyAsync(out, c->masks.as<u8>() + plane * layer, plane, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_set_masks(orip_ctx* c, const uint8_t* masks, int K, int H, int W) {
    orip_enter(c);
    c->mask_bits = nullptr;
    if (K < 1 || K > ORIP_MAX_LAYERS || H <= 0 || W <= 0) ORIP_FAIL(c, "bad shape");
    c->H = H; c->W = W; c->K = K;
    HIPC(c, c->masks.ensure((size_t)H * W * K));
    HIPC(c, hipMemcpyAsync(c->masks.p, masks, (size_t)H * W * K, hipMemcpyHostToDevice, LN(c).stream));
    return 0;
}

extern "C" int orip_keep_layers(orip_ctx* c, const int32_t* layers, int n) {
    orip_enter(c);
    c->edge_bits = nullptr;
    c->mask_bits = nullptr;
    if (!c->masks.p || n < 1 || n > c->K) ORIP_FAIL(c, "bad layer subset (n=%d, K=%d)", n, c->K);
    size_t plane = (size_t)c->H * c->W;
    for (int i = 0; i < n; i++) {
        if (layers[i] < i || layers[i] >= c->K || (i && layers[i] <= layers[i - 1])) ORIP_FAIL(c, "layer subset must be strictly increasing and within range");
        if (layers[i] != i) HIPC(c, hipMemcpyAsync(c->masks.as<u8>() + plane * i, c->masks.as<u8>() + plane * layers[i], plane, hipMemcpyDeviceToDevice, LN(c).stream));
    }
    c->K = n;
    return 0;
}
