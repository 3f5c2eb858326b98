// csrc/stream.hip -- the data-parallel part of 13_build_stream.py: direction codes of every move of a plot (pen-up travels and the
// segments of every polyline), shared/omnirevolve_plotter_stream_creator_helper.py bresenham_dir_codes (:183-207).
//
// The reference walks each segment with an error accumulator (one iteration per step, serial).  The accumulator has a closed form: with
// dx = |x1 - x0| >= dy = |y1 - y0| the major axis moves in every iteration and the minor axis has moved
//     m(j) = max(0, ceil((2 j dy - dx) / (2 dx)))          times after j iterations
// (the minor move of iteration k happens iff m(k) < (2 (k + 1) dy - dx) / (2 dx), which is the helper's strict test `e2 < dx` with the
// error written out; symmetric for dy > dx with its strict `e2 > -dy`).  Step k of a segment is therefore independent of every other step:
// one thread per step over the concatenated moves of the whole plot, 64-bit integer arithmetic, no serial chain.
// Codes (helper :24-25): 0 +Y, 1 NE, 2 +X, 3 SE, 4 -Y, 5 SW, 6 -X, 7 NW.
#include "orip_ctx.h"
#include <rocprim/rocprim.hpp>

__global__ __launch_bounds__(256) void k_seg_counts(const int4* __restrict__ segs, int64_t n, unsigned long long* __restrict__ cnt) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    if (i == n) { cnt[i] = 0; return; }
    const int4 s = segs[i];
    const long long dx = llabs((long long)s.z - s.x), dy = llabs((long long)s.w - s.y);
    cnt[i] = (unsigned long long)(dx > dy ? dx : dy);
}

__device__ __forceinline__ long long ceil_div_pos(long long a, long long b) {      // ceil(a / b), b > 0, clamped at 0 from below
    return a <= 0 ? 0 : (a + b - 1) / b;
}

__global__ __launch_bounds__(256) void k_seg_codes(const int4* __restrict__ segs, int64_t n, const unsigned long long* __restrict__ off, unsigned long long total,
                                                   uint8_t* __restrict__ codes) {
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    // segment of step t: the last i with off[i] <= t (segments without steps have off[i] == off[i + 1] and are skipped by the search)
    int64_t lo = 0, hi = n;
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
    const int4 s = segs[lo];
    const long long k = (long long)(t - off[lo]);
    const long long dx = llabs((long long)s.z - s.x), dy = llabs((long long)s.w - s.y);
    const bool xpos = s.x < s.z, ypos = s.y < s.w;                  // helper: sx = 1 if x0 < x1 else -1 (same for y)
    bool mx, my;
    if (dx >= dy) { mx = true; my = ceil_div_pos(2 * (k + 1) * dy - dx, 2 * dx) != ceil_div_pos(2 * k * dy - dx, 2 * dx); }
    else { my = true; mx = ceil_div_pos(2 * (k + 1) * dx - dy, 2 * dy) != ceil_div_pos(2 * k * dx - dy, 2 * dy); }
    int c;
    if (mx && my) c = xpos ? (ypos ? 1 : 3) : (ypos ? 7 : 5);
    else if (mx) c = xpos ? 2 : 6;
    else c = ypos ? 0 : 4;
    codes[t] = (uint8_t)c;
}

// Direction codes of n moves (x0, y0, x1, y1), resident until the next call; *total = number of steps.
extern "C" int orip_stream_codes(orip_ctx* c, const int32_t* segs, int64_t n, int64_t* total) {
    orip_enter(c);
    if (!total || n < 0 || (n > 0 && !segs)) ORIP_FAIL(c, "bad arguments");
    *total = 0; c->stream_n = 0; c->stream_total = 0;
    if (n == 0) return 0;
    hipStream_t s = LN(c).stream;
    HIPC(c, c->stream_segs.ensure((size_t)n * 16 + 64));
    HIPC(c, c->stream_off.ensure((size_t)(n + 1) * 16 + 64));
    unsigned long long* cnt = c->stream_off.as<unsigned long long>() + (n + 1); unsigned long long* off = c->stream_off.as<unsigned long long>();
    HIPC(c, hipMemcpyAsync(c->stream_segs.p, segs, (size_t)n * 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_seg_counts, dim3(cdiv(n + 1, 256)), dim3(256), 0, s, c->stream_segs.as<int4>(), n, cnt);
    size_t bytes = 0;
    HIPC(c, rocprim::exclusive_scan(nullptr, bytes, cnt, off, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    HIPC(c, LN(c).tmpF.ensure(bytes + 16));
    HIPC(c, rocprim::exclusive_scan(LN(c).tmpF.p, bytes, cnt, off, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long h_total = 0;
    HIPC(c, hipMemcpyAsync(&h_total, off + n, 8, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    HIPC(c, c->stream_codes.ensure((size_t)h_total + 64));
    if (h_total) {
        ProfScope ps(c, "k_seg_codes");
        hipLaunchKernelGGL(k_seg_codes, dim3((unsigned)((h_total + 255) / 256)), dim3(256), 0, s, c->stream_segs.as<int4>(), n, off, h_total, c->stream_codes.as<uint8_t>());
    }
    HIPC(c, hipGetLastError());
    HIPC(c, hipStreamSynchronize(s));
    c->stream_n = n; c->stream_total = (int64_t)h_total; *total = (int64_t)h_total;
    return 0;
}

extern "C" int orip_stream_codes_fetch(orip_ctx* c, int64_t* off_out, uint8_t* codes_out) {
    orip_enter(c);
    if (!off_out) ORIP_FAIL(c, "bad arguments");
    if (c->stream_n == 0) { off_out[0] = 0; return 0; }
    hipStream_t s = LN(c).stream;
    HIPC(c, hipMemcpyAsync(off_out, c->stream_off.p, (size_t)(c->stream_n + 1) * 8, hipMemcpyDeviceToHost, s));
    if (c->stream_total && codes_out) HIPC(c, hipMemcpyAsync(codes_out, c->stream_codes.p, (size_t)c->stream_total, hipMemcpyDeviceToHost, s));
    HIPC(c, hipStreamSynchronize(s));
    return 0;
}
