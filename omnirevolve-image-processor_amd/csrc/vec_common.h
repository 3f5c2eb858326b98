// csrc/vec_common.h -- device-side building blocks shared by the vector stages (05, 07, 08, 10, 12).
#pragma once
#include "orip_ctx.h"
#include "vec_serial.h"
#include "vsrc.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

namespace {   // every TU that includes this header gets its own copy of the kernels (internal linkage)

// ---- rocPRIM wrappers (temporary storage in ctx->tmpF) ----
// exclusive prefix sums of up to 32 768 integers by ONE workgroup (thread t owns the items [t * per, (t + 1) * per)): the lists of this path are a few
// thousand polylines long, and rocPRIM's scan is two dispatches (look-back state, scan) at ~50 us each on a queue that shares the card.  in == out is fine.
// (256 threads: a 1024-thread workgroup waits for a CU with sixteen free wave slots, which costs ~0.5 ms on a card the other layers keep full.)
template <class T>
__global__ __launch_bounds__(256) void k_scan_small(const T* in, T* out, unsigned n, unsigned per) {
    __shared__ T wsum[16];
    const unsigned tid = threadIdx.x, lo = tid * per, hi = min(n, lo + per);
    T mine = 0;
    for (unsigned i = lo; i < hi; i++) mine += in[i];
    T inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o, 64); if ((int)(tid & 63u) >= o) inc += t; }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = inc;
    __syncthreads();
    T base = 0;
    for (unsigned w = 0; w < (tid >> 6); w++) base += wsum[w];
    T run = base + inc - mine;
    __syncthreads();
    for (unsigned i = lo; i < hi; i++) { const T v = in[i]; out[i] = run; run += v; }
}
template <class T>
static int vscan_excl(orip_ctx* c, const T* in, T* out, size_t n) {
    if (n == 0) return 0;
    if (n <= 32768) {
        hipLaunchKernelGGL(k_scan_small<T>, dim3(1), dim3(256), 0, LN(c).stream, in, out, (unsigned)n, (unsigned)cdiv((int64_t)n, 256));
        HIPC(c, hipGetLastError());
        return 0;
    }
    size_t bytes = 0;
    HIPC(c, rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), LN(c).stream));
    HIPC(c, LN(c).tmpF.ensure(bytes + 16));
    HIPC(c, rocprim::exclusive_scan(LN(c).tmpF.p, bytes, in, out, T(0), n, rocprim::plus<T>(), LN(c).stream));
    return 0;
}
template <class K, class V>
static int vsort_pairs(orip_ctx* c, const K* kin, K* kout, const V* vin, V* vout, size_t n, int begin_bit, int end_bit, bool desc = false) {
    if (n == 0) return 0;
    size_t bytes = 0;
    if (!desc) HIPC(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, LN(c).stream));
    else HIPC(c, rocprim::radix_sort_pairs_desc(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, LN(c).stream));
    HIPC(c, LN(c).tmpF.ensure(bytes + 16));
    if (!desc) HIPC(c, rocprim::radix_sort_pairs(LN(c).tmpF.p, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, LN(c).stream));
    else HIPC(c, rocprim::radix_sort_pairs_desc(LN(c).tmpF.p, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, LN(c).stream));
    return 0;
}
template <class T>
static int vread(orip_ctx* c, T* host, const T* dev, size_t n = 1) {
    HIPC(c, hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}

// ---- per-polyline features ----
struct PolyFeat {
    int32_t x0, y0, x1, y1;     // bbox
    int32_t sx, sy, ex, ey;     // first / last point (of the OPEN view when open_view)
    int64_t n;                  // points (of the open view when open_view)
    float per;                  // numpy pairwise float32 perimeter (KIND 0) or 12:_poly_len (KIND 1)
    double arc;                 // cv::arcLength (closed flag given by caller), exact double sum
    uint8_t closed;             // first == last on the ORIGINAL polyline (n >= 2)
};

// (stage 08's prefetch keeps the segment lengths of polylines above ORIP_LONG_CUM = 128 points (k_seglen) and takes the perimeters of open views above
// this threshold from them: it must not be below ORIP_LONG_CUM)
#define ORIP_LONG_POLY 192
// ---- where a list's points come from (vsrc.h): explicit array or the layer's walk records ----
static inline ESrc esrc_of(const DPolys& P) { return ESrc{P.off.as<int64_t>(), reinterpret_cast<const int2*>(P.pts.p)}; }
static int vsrc_of(orip_ctx* c, const DPolys& P, VSrc& out) {
    const WalkStore& WS = c->wstore[P.vlayer];
    if (P.vepoch != WS.epoch) ORIP_FAIL(c, "the walk records of layer %d this list was built on have been replaced by a newer orip_contours_layer", P.vlayer);
    out.off = P.off.as<int64_t>(); out.view = P.vident ? nullptr : P.vview.as<VView>(); out.walk = WS.walk.as<VWalk>();
    if (P.scaled && P.vsepoch != WS.sepoch) ORIP_FAIL(c, "the scaled walk records of layer %d this list was built on have been replaced by a newer orip_scale_vectors", P.vlayer);
    out.g.piece = WS.piece.as<VPiece>();
    out.g.own = P.scaled ? WS.own_s.as<int2>() : WS.own.as<int2>(); out.g.lxy = P.scaled ? WS.lxy_s.as<int2>() : WS.lxy.as<int2>();
    return 0;
}
static inline bool is_coded(const DPolys& P) { return P.virt && !P.pts_ok; }
// runs BODY once with SRC bound to the list's point source (VSrc for a walk-coded list whose points are not expanded, else ESrc)
#define ORIP_WITH_SRC(c, P, SRC, BODY)                                                         \
    do {                                                                                       \
        if (is_coded(P)) { VSrc SRC; ORIP_TRY(vsrc_of(c, P, SRC)); BODY }                      \
        else { const ESrc SRC = esrc_of(P); BODY }                                             \
    } while (0)
template <class Cur> struct CurPt {
    Cur& c;
    __device__ __forceinline__ vs::IPt operator()(int64_t i) const { const int2 p = c.at(i); return vs::IPt{p.x, p.y}; }
};

// what: bit0 perimeter KIND0, bit1 perimeter KIND1 (hypot), bit2 arcLength closed, bit3 arcLength open, bit4 open view (_ensure_open)
// the reversed polyline as a point getter (pt(i) = point n - 1 - i)
template <class Cur> struct RevPt {
    Cur& c; int64_t n;
    __device__ __forceinline__ vs::IPt operator()(int64_t i) const { const int2 p = c.at(n - 1 - i); return vs::IPt{p.x, p.y}; }
};
// what: ... bit5 (with bit0): per_rev[i] = the same perimeter over the REVERSED open polyline (numpy's pairwise sum depends on the order)
template <class Src>
__global__ __launch_bounds__(128) void k_poly_features(Src src, int64_t n_polys, int what, PolyFeat* __restrict__ out, float* __restrict__ per_rev) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_polys) return;
    auto cu = src.cur(i);
    int64_t n = src.len(i);
    PolyFeat f;
    const int2 pf = cu.at(0); int2 pl = n >= 1 ? cu.at(n - 1) : pf;
    f.closed = (n >= 2 && pf.x == pl.x && pf.y == pl.y) ? 1 : 0;
    if ((what & 16) && f.closed) { n -= 1; pl = cu.at(n - 1); }
    f.n = n;
    f.sx = pf.x; f.sy = pf.y; f.ex = pl.x; f.ey = pl.y;
    f.per = 0.f; f.arc = 0.0; f.x0 = f.x1 = pf.x; f.y0 = f.y1 = pf.y;
    if (n > ORIP_LONG_POLY) { out[i] = f; return; }      // bbox / sums of long polylines: k_poly_features_long (one block each)
    int32_t x0 = pf.x, x1 = pf.x, y0 = pf.y, y1 = pf.y;
    for (int64_t k = 1; k < n; k++) { const int2 q = cu.at(k); x0 = min(x0, q.x); x1 = max(x1, q.x); y0 = min(y0, q.y); y1 = max(y1, q.y); }
    f.x0 = x0; f.y0 = y0; f.x1 = x1; f.y1 = y1;
    const CurPt<decltype(cu)> pt{cu};
    if (what & 1) f.per = vs::pairwise_seglen_sum_p<0>(pt, n);
    if ((what & 33) == 33) { const RevPt<decltype(cu)> rp{cu, n}; per_rev[i] = vs::pairwise_seglen_sum_p<0>(rp, n); }
    if (what & 2) f.per = vs::pairwise_seglen_sum_p<1>(pt, n);
    if (what & 4) f.arc = vs::arc_length_p(pt, n, true);
    if (what & 8) f.arc = vs::arc_length_p(pt, n, false);
    out[i] = f;
}


// Long polylines (n > ORIP_LONG_POLY): one 256-thread block per polyline.  bbox and cv::arcLength are plain parallel
// reductions (the double sum of float edge lengths is exact at these magnitudes, so its order is free).  The numpy float32
// pairwise perimeter keeps numpy's exact tree: every leaf of the tree has 64..128 elements (n2 = n/2 - (n/2)%8 >= 64 for
// n > 128), so each multiple of 64 lies in exactly one leaf; the thread that holds the first multiple of 64 of a leaf sums
// that leaf in numpy's 8-accumulator order, and thread 0 then combines the leaf sums with the explicit-stack traversal.
// Evaluates numpy's pairwise tree below the node (s0, n0) from the leaf sums; `part`/`depth_left` let the root traversal stop at
// nodes that other threads have already reduced (code = path bits from the root).
__device__ float pairwise_subtree(const float* __restrict__ leafsum, int64_t s0, int64_t n0, const float* part, int depth_left) {
    int64_t fs[28], fn[28]; int fstate[28], fdep[28]; unsigned fcode[28]; float fleft[28];     // depth <= log2(2^31 / 64) + 2
    int sp = 1; fs[0] = s0; fn[0] = n0; fstate[0] = 0; fdep[0] = depth_left; fcode[0] = 0;
    float ret = 0.f;
    while (sp > 0) {
        int t = sp - 1;
        if (fn[t] <= 128) { ret = leafsum[(fs[t] + 63) >> 6]; sp--; continue; }
        if (part && fdep[t] == 0) { ret = part[fcode[t]]; sp--; continue; }
        int64_t n2 = fn[t] / 2; n2 -= n2 % 8;
        if (fstate[t] == 0) { fstate[t] = 1; fs[sp] = fs[t]; fn[sp] = n2; fstate[sp] = 0; fdep[sp] = fdep[t] - 1; fcode[sp] = fcode[t] << 1; sp++; }
        else if (fstate[t] == 1) { fleft[t] = ret; fstate[t] = 2; fs[sp] = fs[t] + n2; fn[sp] = fn[t] - n2; fstate[sp] = 0; fdep[sp] = fdep[t] - 1; fcode[sp] = (fcode[t] << 1) | 1u; sp++; }
        else { ret = fleft[t] + ret; sp--; }
    }
    return ret;
}
// one numpy leaf (8 <= n <= 128 elements from s) summed by 8 lanes: lane j owns accumulator r[j]; the xor tree reproduces
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) (float addition commutes), the n % 8 tail is added in order
template <int KIND>
__device__ __forceinline__ float pairwise_leaf_g8(const int32_t* xy, int64_t s, int64_t n, int j) {
    auto el = [&](int64_t i) { return KIND == 0 ? vs::seg_len_f32(xy, s + i) : vs::seg_hypot_f32(xy, s + i); };
    const int64_t lim = n - (n % 8);
    float r = el(j);
    for (int64_t i = 8 + j; i < lim; i += 8) r += el(i);
    r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
    for (int64_t i = lim; i < n; i++) r += el(i);
    return r;
}
#define ORIP_PW_DEPTH 8
__global__ __launch_bounds__(256) void k_len_keys(const int64_t* __restrict__ off, int64_t n, unsigned* __restrict__ key, unsigned* __restrict__ val) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { int64_t m = off[i + 1] - off[i]; key[i] = (unsigned)(m > 0xffffffffLL ? 0xffffffffLL : m); val[i] = (unsigned)i; }
}
// leaf of the REVERSED sequence: element i' of the reversed polyline's segment lengths is forward segment ns - 1 - i'
__device__ __forceinline__ float pairwise_leaf_g8_rev(const int32_t* xy, int64_t ns, int64_t s, int64_t n, int j) {
    auto el = [&](int64_t i) { return vs::seg_len_f32(xy, ns - 1 - (s + i)); };
    const int64_t lim = n - (n % 8);
    float r = el(j);
    for (int64_t i = 8 + j; i < lim; i += 8) r += el(i);
    r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
    for (int64_t i = lim; i < n; i++) r += el(i);
    return r;
}
// the same leaves over STORED segment lengths (sl[k] = float32 length of segment k; prefetch08: k_seglen)
__device__ __forceinline__ float pairwise_leaf_f(const float* sl, int64_t s, int64_t n, int j) {
    const int64_t lim = n - (n % 8);
    float r = sl[s + j];
    for (int64_t i = 8 + j; i < lim; i += 8) r += sl[s + i];
    r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
    for (int64_t i = lim; i < n; i++) r += sl[s + i];
    return r;
}
__device__ __forceinline__ float pairwise_leaf_f_rev(const float* sl, int64_t ns, int64_t s, int64_t n, int j) {
    const int64_t lim = n - (n % 8);
    float r = sl[ns - 1 - (s + j)];
    for (int64_t i = 8 + j; i < lim; i += 8) r += sl[ns - 1 - (s + i)];
    r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64);
    for (int64_t i = lim; i < n; i++) r += sl[ns - 1 - (s + i)];
    return r;
}
// All leaves of numpy's pairwise trees of all long polylines in ONE launch, from stored segment lengths (prefetch08): 8 lanes per slot of the leaf table
// (slot (off[i] >> 6) + 2 i + m belongs to the multiple 64 m of polyline i; the leaf that holds element 64 m owns it when 64 m is its first multiple of 64).
// The same leaf shape serves the forward sum and the sum over the reversed sequence (element i' of the reversed polyline = forward segment ns - 1 - i').
// k_poly_features_long then only combines the leaves (what & 64): with one block per polyline staging the lengths through LDS the launch was as long as
// ~7 rounds of 186 k-element polylines at five blocks per CU.
__device__ __forceinline__ void perim_leaves_seg_block(int64_t vblock, int64_t* i_first, const int64_t* __restrict__ off, int64_t n_polys, const PolyFeat* __restrict__ feat,
                                                       const float* __restrict__ seg, float* __restrict__ leafbuf, float* __restrict__ leafbuf_rev, int64_t nslots) {
    const int64_t q = (vblock * 256 + threadIdx.x) >> 3; const int j = threadIdx.x & 7;
    if (threadIdx.x == 0) {                               // polyline of the block's first slot: the last i with (off[i] >> 6) + 2 i <= q; the other 31 slots walk on from it
        int64_t lo = 0, hi = n_polys - 1;
        while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if ((off[mid] >> 6) + 2 * mid <= q) lo = mid; else hi = mid - 1; }
        *i_first = lo;
    }
    __syncthreads();
    if (q >= nslots) return;
    int64_t i = *i_first;
    while (i + 1 < n_polys && (off[i + 1] >> 6) + 2 * (i + 1) <= q) i++;
    const int64_t n = feat[i].n;
    if (n <= ORIP_LONG_POLY) return;
    const int64_t ns = n - 1, pm = (q - ((off[i] >> 6) + 2 * i)) << 6;
    if (pm >= ns) return;
    int64_t s = 0, len = ns;                              // the leaf of the tree over ns elements that holds element pm
    while (len > 128) { int64_t n2 = len / 2; n2 -= n2 % 8; if (pm < s + n2) len = n2; else { s += n2; len -= n2; } }
    if ((((s + 63) >> 6) << 6) != pm) return;
    const float* sl = seg + off[i];
    const float v = pairwise_leaf_f(sl, s, len, j);
    if (j == 0) leafbuf[q] = v;
    if (leafbuf_rev) { const float r = pairwise_leaf_f_rev(sl, ns, s, len, j); if (j == 0) leafbuf_rev[q] = r; }
}
__global__ __launch_bounds__(256) void k_perim_leaves_seg(const int64_t* __restrict__ off, int64_t n_polys, const PolyFeat* __restrict__ feat, const float* __restrict__ seg,
                                                          float* __restrict__ leafbuf, float* __restrict__ leafbuf_rev, int64_t nslots) {
    __shared__ int64_t i_first;
    perim_leaves_seg_block((int64_t)blockIdx.x, &i_first, off, n_polys, feat, seg, leafbuf, leafbuf_rev, nslots);
}
#define ORIP_PF_MARGIN 132      // points staged on either side of a turn's 2048: a leaf has at most 128 elements and owns a multiple of 64 of the turn
template <class Src, bool FROM_SEG = false>
__global__ __launch_bounds__(256) void k_poly_features_long(Src src, int64_t n_polys, int what,
                                                             PolyFeat* __restrict__ out, float* __restrict__ leafbuf, const unsigned* __restrict__ order,
                                                             float* __restrict__ per_rev, float* __restrict__ leafbuf_rev, const float* __restrict__ seg = nullptr) {
    __shared__ int rx0[256], rx1[256], ry0[256], ry1[256];
    __shared__ double rarc[256];
    __shared__ float part[2 << ORIP_PW_DEPTH];
    __shared__ int2 stage[2048 + 2 * ORIP_PF_MARGIN + 8];
    const bool want_rev = (what & 33) == 33;
    for (int64_t rr = blockIdx.x; rr < n_polys; rr += gridDim.x) {
        const int64_t i = order[rr];               // longest first: a block that draws a long polyline late would be the tail of the launch
        PolyFeat f = out[i];
        const int64_t n = f.n;                     // already the open view when requested
        if (n <= ORIP_LONG_POLY) continue;         // uniform for the block
        auto cu = src.cur(i);
        auto P2 = [&](int64_t k) { return cu.at(k); };
        const int tid = threadIdx.x;
        int x0 = f.sx, x1 = f.sx, y0 = f.sy, y1 = f.sy; double arc = 0.0;
        const bool closed_arc = (what & 4) != 0, any_arc = (what & 12) != 0, any_per = (what & 3) != 0;
        if (any_arc) {
            // arc length next to the bounding box: four loads in flight per thread; a thread adds its terms in the order of its k
            auto seg = [&](int64_t k, const int2 a, const int2 b) {       // b: predecessor of point k
                x0 = min(x0, a.x); x1 = max(x1, a.x); y0 = min(y0, a.y); y1 = max(y1, a.y);
                float dx = (float)a.x - (float)b.x, dy = (float)a.y - (float)b.y;
                arc += (double)sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
            };
            auto pred = [&](int64_t k) -> int64_t { return k == 0 ? (closed_arc ? n - 1 : 0) : k - 1; };
            // A wave takes four consecutive windows of 64 points per turn and fetches every point once (a cursor call is ~25 instructions):
            // the predecessor of point k sits in the lane below, that of a window's first point in the last lane of the window before,
            // and only the first point of a turn needs one extra fetch.
            const int lane = tid & 63;
            for (int64_t base = (int64_t)(tid >> 6) * 256; base < n; base += 1024) {
                int2 p[4];
#pragma unroll
                for (int w = 0; w < 4; w++) { const int64_t k = base + 64 * w + lane; p[w] = k < n ? P2(k) : make_int2(0, 0); }
                int2 first = P2(pred(base));
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    int2 b;
                    b.x = __builtin_amdgcn_update_dpp(0, p[w].x, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                    b.y = __builtin_amdgcn_update_dpp(0, p[w].y, 0x138, 0xf, 0xf, true);
                    if (lane == 0) b = first;
                    const int64_t k = base + 64 * w + lane;
                    if (k < n) seg(k, p[w], b);
                    first = make_int2(__builtin_amdgcn_readlane(p[w].x, 63), __builtin_amdgcn_readlane(p[w].y, 63));
                }
            }
        } else if (!any_per) {
            // bounding box only: four independent 8-byte loads per turn keep the memory pipeline busy (the loop is latency-bound otherwise)
            int64_t k = tid;
            for (; k + 768 < n; k += 1024) {
                const int2 a = P2(k), b = P2(k + 256), cc = P2(k + 512), d = P2(k + 768);
                x0 = min(min(x0, a.x), min(min(b.x, cc.x), d.x)); x1 = max(max(x1, a.x), max(max(b.x, cc.x), d.x));
                y0 = min(min(y0, a.y), min(min(b.y, cc.y), d.y)); y1 = max(max(y1, a.y), max(max(b.y, cc.y), d.y));
            }
            for (; k < n; k += 256) { const int2 a = P2(k); x0 = min(x0, a.x); x1 = max(x1, a.x); y0 = min(y0, a.y); y1 = max(y1, a.y); }
        }
        float per = 0.f, perR = 0.f;
        if (any_per) {
            const int64_t ns = n - 1;               // number of segments
            float* ls = leafbuf + (src.off[i] >> 6) + 2 * i;
            float* lsR = want_rev ? leafbuf_rev + (src.off[i] >> 6) + 2 * i : nullptr;
            const int grp = tid >> 3, j = tid & 7;  // 32 groups of 8 lanes, one leaf per group, turn and direction
            // A turn covers the 32 multiples of 64 in [r0, r0 + 2048).  The leaves of numpy's tree that own them lie inside
            // [r0 - 63, r0 + 2047 + 128]; the leaves of the REVERSED sequence that own the multiples of 64 of the mirrored interval
            // [ns - r0 - 2048, ns - r0) map to forward segments inside [r0 - 129, r0 + 2048 + 128).  So one stretch of points, staged
            // in LDS by all threads (independent coalesced loads), serves both directions -- and the bounding box (a point is read once).
            // The points of the NEXT turn are requested before the leaves of this turn are summed and only land in LDS after them.
            constexpr int NX = (2048 + 2 * ORIP_PF_MARGIN + 255) / 256;
            int2 nxt[NX];
            // FROM_SEG: the stretch holds the stored LENGTHS of the segments [lo, hi - 1) instead of the points [lo, hi) (the bounding box is in place already)
            const float* sgp = FROM_SEG ? seg + src.off[i] : nullptr;
            float* stagef = reinterpret_cast<float*>(stage);
            auto request = [&](int64_t r0) {
                const int64_t lo = max((int64_t)0, r0 - ORIP_PF_MARGIN), hi = min(n, r0 + 2048 + ORIP_PF_MARGIN);
#pragma unroll
                for (int u = 0; u < NX; u++) {
                    const int64_t q = lo + tid + 256 * u;
                    if (FROM_SEG) nxt[u].x = q < hi - 1 ? __float_as_int(sgp[q]) : 0;
                    else nxt[u] = q < hi ? P2(q) : make_int2(0, 0);
                }
            };
            if (!(what & 64)) request(0);
            for (int64_t r0 = 0; r0 < n && !(what & 64); r0 += 32 * 64) {      // (the last turn may hold points only: the bounding box wants them all; what & 64: the leaf sums are in place, k_perim_leaves_seg)
                const int64_t lo = max((int64_t)0, r0 - ORIP_PF_MARGIN), hi = min(n, r0 + 2048 + ORIP_PF_MARGIN);       // points [lo, hi)
                __syncthreads();
#pragma unroll
                for (int u = 0; u < NX; u++) {
                    const int64_t q = lo + tid + 256 * u;
                    if (FROM_SEG) { if (q < hi - 1) stagef[tid + 256 * u] = __int_as_float(nxt[u].x); }
                    else if (q < hi) {
                        stage[tid + 256 * u] = nxt[u];
                        if (q >= r0 && q < r0 + 2048) { x0 = min(x0, nxt[u].x); x1 = max(x1, nxt[u].x); y0 = min(y0, nxt[u].y); y1 = max(y1, nxt[u].y); }
                    }
                }
                __syncthreads();
                if (r0 + 32 * 64 < n) request(r0 + 32 * 64);
                const int32_t* sp = reinterpret_cast<const int32_t*>(stage) - 2 * lo;        // sp[2 * k] = x of point k
                const float* sf = stagef - lo;                                               // sf[k] = length of segment k
                auto leaf_of = [&](int64_t pm, int64_t& s, int64_t& len) {                   // the leaf of numpy's tree over ns elements that holds element pm
                    s = 0; len = ns;
                    while (len > 128) { int64_t n2 = len / 2; n2 -= n2 % 8; if (pm < s + n2) len = n2; else { s += n2; len -= n2; } }
                };
                const int64_t pm = r0 + (int64_t)grp * 64;
                if (pm < ns) {
                    int64_t s, len; leaf_of(pm, s, len);
                    if (((s + 63) >> 6) << 6 == pm) {   // every multiple of 64 lies in exactly one leaf; its first one owns the leaf
                        float v = FROM_SEG ? pairwise_leaf_f(sf, s, len, j) : ((what & 1) ? pairwise_leaf_g8<0>(sp, s, len, j) : pairwise_leaf_g8<1>(sp, s, len, j));
                        if (j == 0) ls[pm >> 6] = v;
                    }
                }
                if (want_rev) {
                    // multiples of 64 of the reversed index space inside the mirrored interval [max(0, ns - r0 - 2048), ns - r0)
                    const int64_t ilo = max((int64_t)0, ns - r0 - 2048), ihi = ns - r0;
                    const int64_t pmr = (((ilo + 63) >> 6) << 6) + (int64_t)grp * 64;
                    if (pmr < ihi) {
                        int64_t s, len; leaf_of(pmr, s, len);
                        if (((s + 63) >> 6) << 6 == pmr) {
                            float v = FROM_SEG ? pairwise_leaf_f_rev(sf, ns, s, len, j) : pairwise_leaf_g8_rev(sp, ns, s, len, j);
                            if (j == 0) lsR[pmr >> 6] = v;
                        }
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            // numpy's tree, level by level.  Node `code` (heap numbering, root 1) at depth d is reached by the d path bits of code - 2^d
            // (0 = left half of n2 = n/2 - (n/2)%8 elements).  Depth ORIP_PW_DEPTH: thread t reduces the subtree below its node from
            // the leaf sums; the levels above combine left + right in LDS, a node that is itself a leaf takes its leaf sum.
            auto node_of = [&](int d, int t, int64_t& s, int64_t& len) -> bool {      // false: an ancestor is already a leaf
                s = 0; len = ns;
                for (int lvl = d - 1; lvl >= 0; lvl--) {
                    if (len <= 128) return false;
                    int64_t n2 = len / 2; n2 -= n2 % 8;
                    if ((t >> lvl) & 1) { s += n2; len -= n2; } else len = n2;
                }
                return true;
            };
            for (int dir = 0; dir < (want_rev ? 2 : 1); dir++) {
                const float* lsd = dir ? lsR : ls;
                {
                    int64_t s, len;
                    if (node_of(ORIP_PW_DEPTH, tid, s, len)) part[(1 << ORIP_PW_DEPTH) + tid] = pairwise_subtree(lsd, s, len, nullptr, 0);
                }
                for (int d = ORIP_PW_DEPTH - 1; d >= 0; d--) {
                    __syncthreads();
                    if (tid < (1 << d)) {
                        int64_t s, len; const int code = (1 << d) + tid;
                        if (node_of(d, tid, s, len)) part[code] = (len <= 128) ? lsd[(s + 63) >> 6] : part[2 * code] + part[2 * code + 1];
                    }
                }
                __syncthreads();
                if (tid == 0) { if (dir) perR = part[1]; else per = part[1]; }
                __syncthreads();
            }
        }
        rx0[tid] = x0; rx1[tid] = x1; ry0[tid] = y0; ry1[tid] = y1; rarc[tid] = arc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) { rx0[tid] = min(rx0[tid], rx0[tid + s]); rx1[tid] = max(rx1[tid], rx1[tid + s]); ry0[tid] = min(ry0[tid], ry0[tid + s]); ry1[tid] = max(ry1[tid], ry1[tid + s]); rarc[tid] += rarc[tid + s]; }
            __syncthreads();
        }
        if (tid == 0) {
            if (!FROM_SEG) { f.x0 = rx0[0]; f.x1 = rx1[0]; f.y0 = ry0[0]; f.y1 = ry1[0]; }       // FROM_SEG: the box came with f (k_cumlen_long2 wrote it)
            f.arc = rarc[0]; f.per = per; out[i] = f; if (want_rev) per_rev[i] = perR;
        }
        __syncthreads();
    }
}
// features of every polyline of a list: short ones one lane each, long ones one block each
// cv::arcLength(contour, closed = true) (07:50) of the long contours of a walk-coded list whose polylines are whole walks, WITHOUT visiting their points:
// a walk is its own points plus tail pieces that run through consecutive log entries, the last ones lap after lap around one cycle (walker.h: VWalk /
// VPiece), so its perimeter is the own segments + per piece the segments of one lap (x laps) and of the partial lap + the junctions.  The reference adds the
// float32 segment lengths into a double; every length is a multiple of 2^-23 and the total stays below 2^22, so every partial sum is exact and the order (and
// the multiplication by the lap count) cannot change the result -- the same argument k_poly_features_long's parallel sum rests on.  One wave per walk; the pass
// over 2.8e8 points it replaces sat on the chain in front of stage 07's greedy order with 1 - 4 ms.
__global__ __launch_bounds__(64) void k_walk_arcs(VSrc src, int64_t n_polys, PolyFeat* __restrict__ feat) {
    const int lane = threadIdx.x;
    auto len2 = [](const int2 a, const int2 b) -> double {
        const float dx = (float)a.x - (float)b.x, dy = (float)a.y - (float)b.y;
        return (double)sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    };
    auto wave_sum = [](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    for (int64_t i = blockIdx.x; i < n_polys; i += gridDim.x) {
        if (feat[i].n <= ORIP_LONG_POLY) continue;           // the short ones have their sum from k_poly_features
        const VWalk w = src.walk[i];
        const int2* own = src.g.own + w.own_off;
        double acc = 0.0;
        for (unsigned t = 1u + (unsigned)lane; t <= w.n_own; t += 64u) acc += len2(own[t], own[t - 1u]);
        const unsigned closing = w.flags & 1u;
        const unsigned T = w.len - closing - (w.n_own + 1u);      // tail points
        int2 last = own[w.n_own];
        for (unsigned j = 0; j < w.n_piece && T > 0u; j++) {
            const VPiece q = src.g.piece[w.piece_off + j];
            const unsigned cnt = (j + 1u < w.n_piece ? src.g.piece[w.piece_off + j + 1u].u0 : T) - q.u0;
            if (cnt == 0u) continue;
            const int2* L = src.g.lxy + q.ent;
            if (lane == 0) acc += len2(L[0], last);               // the junction into the piece
            if (q.lam == 0u) {
                for (unsigned e = (unsigned)lane; e + 1u < cnt; e += 64u) acc += len2(L[e + 1u], L[e]);
                last = L[cnt - 1u];
            } else {
                // points m = 0 .. cnt - 1 sit at entry m mod lam: step m wraps iff m mod lam == 0, every lap is the lam - 1 inner steps + the wrap
                const unsigned laps = (cnt - 1u) / q.lam, r = (cnt - 1u) % q.lam;
                double full = 0.0, part = 0.0;
                for (unsigned e = (unsigned)lane; e + 1u < q.lam; e += 64u) { const double d = len2(L[e + 1u], L[e]); full += d; if (e < r) part += d; }
                full = wave_sum(full);
                if (lane == 0) acc += (double)laps * (full + len2(L[0], L[q.lam - 1u]));
                acc += part;
                last = L[r];
            }
        }
        if (lane == 0) acc += len2(own[0], last);                 // to the closing point when there is one (then the wrap is 0), else the closed contour's wrap
        acc = wave_sum(acc);
        if (lane == 0) feat[i].arc = acc;
    }
}
// the long polylines' part of vfeatures_src
template <class Src>
static int vfeatures_long(orip_ctx* c, const Src& src, int64_t n, int64_t total, int what, PolyFeat* feat, float* per_rev) {
    if (n == 0 || total <= ORIP_LONG_POLY) return 0;
    const size_t nleaf = (size_t)(total >> 6) + 2 * (size_t)n + 8;
    HIPC(c, LN(c).vtmp[11].ensure(nleaf * sizeof(float) * ((what & 32) ? 2 : 1) + (size_t)n * 16 + 64));
    float* leafbuf = LN(c).vtmp[11].as<float>(); float* leafbuf_rev = (what & 32) ? leafbuf + nleaf : nullptr;
    unsigned* kin = reinterpret_cast<unsigned*>(leafbuf + nleaf * ((what & 32) ? 2 : 1)); unsigned* kout = kin + n; unsigned* vin = kout + n; unsigned* vout = vin + n;
    hipLaunchKernelGGL(k_len_keys, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, src.off, n, kin, vin);
    ORIP_TRY((vsort_pairs<unsigned, unsigned>(c, kin, kout, vin, vout, (size_t)n, 0, 32, true)));
    ProfScope ps(c, "k_poly_features_long");
    hipLaunchKernelGGL((k_poly_features_long<Src, false>), dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(256), 0, LN(c).stream, src, n, what, feat, leafbuf, vout, per_rev, leafbuf_rev, (const float*)nullptr);
    HIPC(c, hipGetLastError());
    return 0;
}
template <class Src>
static int vfeatures_src(orip_ctx* c, const Src& src, int64_t n, int64_t total, int what, PolyFeat* feat, float* per_rev = nullptr) {
    if (n == 0) return 0;
    if (!per_rev) what &= ~32;
    hipLaunchKernelGGL(k_poly_features<Src>, dim3(cdiv(n, 128)), dim3(128), 0, LN(c).stream, src, n, what, feat, per_rev);
    ORIP_TRY(vfeatures_long(c, src, n, total, what, feat, per_rev));
    HIPC(c, hipGetLastError());
    return 0;
}
static int vfeatures(orip_ctx* c, const DPolys& P, int what, PolyFeat* feat) {
    ORIP_WITH_SRC(c, P, src, { ORIP_TRY(vfeatures_src(c, src, P.n, P.total, what, feat)); });
    return 0;
}

// ---- greedy nearest-neighbour ordering (07:55-79 / 08:223-248 / 10:69-97), one 1024-thread block per list ----
// rule07: closed contours are entered at their start only and the cursor returns to their start (07:60-62, 80-83).
struct NNEnds { int32_t sx, sy, ex, ey; uint8_t closed; };
__device__ __forceinline__ float nn_d2(int32_t ax, int32_t ay, int32_t bx, int32_t by) {
    float dx = __fsub_rn((float)ax, (float)bx), dy = __fsub_rn((float)ay, (float)by);
    return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
}
// sel[0] = seed polyline, sel[1] = coordinate-range flags (k_argmax_feat: bit 0 some coordinate beyond int16 -> no LDS variant, bit 1 beyond 15 bits -> no grid variant; both written on the device just before: the host does not wait for
// them).  The candidates are all enqueued and each decides from sel[1] whether it is the one that runs: (flags & skip_if) != 0 -> not
// this one; need_any != 0 && (flags & need_any) == 0 -> not this one either.
#define ORIP_NN_GATE(sel, skip_if, need_any) const int fl_ = (sel)[1]; if ((fl_ & (skip_if)) != 0 || ((need_any) != 0 && (fl_ & (need_any)) == 0)) return; const int seed = (sel)[0];
__global__ __launch_bounds__(1024) void k_greedy_nn(const NNEnds* __restrict__ ends, int n, const int* __restrict__ sel, int skip_if, int need_any, int rule07, uint8_t* __restrict__ used,
                                                     int32_t* __restrict__ order, uint8_t* __restrict__ flips) {
    ORIP_NN_GATE(sel, skip_if, need_any)
    __shared__ unsigned long long wbest[16];
    __shared__ int cxs, cys;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += (int)blockDim.x) used[i] = (i == seed);
    if (tid == 0) {
        order[0] = seed; flips[0] = 0;
        NNEnds e = ends[seed];
        if (rule07 && e.closed) { cxs = e.sx; cys = e.sy; } else { cxs = e.ex; cys = e.ey; }
    }
    __syncthreads();
    for (int step = 1; step < n; step++) {
        const int cx = cxs, cy = cys;
        unsigned long long best = ~0ULL;
        for (int i = tid; i < n; i += (int)blockDim.x) {
            if (used[i]) continue;
            NNEnds e = ends[i];
            float ds = nn_d2(e.sx, e.sy, cx, cy);
            float v = ds;
            if (!(rule07 && e.closed)) { float de = nn_d2(e.ex, e.ey, cx, cy); if (!(ds <= de)) v = de; }
            unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)i;
            if (key < best) best = key;
        }
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(best, o, 64); if (t < best) best = t; }
        if ((tid & 63) == 0) wbest[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = wbest[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); w++) if (wbest[w] < b) b = wbest[w];
            int bi = (int)(b & 0xffffffffu);
            NNEnds e = ends[bi];
            float ds = nn_d2(e.sx, e.sy, cx, cy), de = nn_d2(e.ex, e.ey, cx, cy);
            bool cl = rule07 && e.closed;
            bool flip = cl ? false : !(ds <= de);
            used[bi] = 1; order[step] = bi; flips[step] = flip ? 1 : 0;
            if (cl) { cxs = e.sx; cys = e.sy; }
            else if (flip) { cxs = e.sx; cys = e.sy; } else { cxs = e.ex; cys = e.ey; }
        }
        __syncthreads();
    }
}

// LDS-resident variant: end points as int16 quads + a state byte per polyline live in LDS (n <= 16000), so a greedy step costs
// two barriers and a few LDS reads instead of global-memory round trips.  Same selection rule, same tie-break.
__global__ __launch_bounds__(1024) void k_greedy_nn_lds(const NNEnds* __restrict__ ends, int n, const int* __restrict__ sel, int skip_if, int need_any, int rule07,
                                                         int32_t* __restrict__ order, uint8_t* __restrict__ flips) {
    ORIP_NN_GATE(sel, skip_if, need_any)
    extern __shared__ __align__(16) unsigned char smem[];
    short4* P = reinterpret_cast<short4*>(smem);                 // (sx, sy, ex, ey)
    uint8_t* stt = smem + (size_t)n * sizeof(short4);            // bit0 used, bit1 closed
    __shared__ unsigned long long wbest[16];
    __shared__ int cxs, cys;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) {
        NNEnds e = ends[i];
        P[i] = make_short4((short)e.sx, (short)e.sy, (short)e.ex, (short)e.ey);
        stt[i] = (uint8_t)((i == seed ? 1 : 0) | ((rule07 && e.closed) ? 2 : 0));
    }
    if (tid == 0) {
        order[0] = seed; flips[0] = 0;
        NNEnds e = ends[seed];
        if (rule07 && e.closed) { cxs = e.sx; cys = e.sy; } else { cxs = e.ex; cys = e.ey; }
    }
    __syncthreads();
    for (int step = 1; step < n; step++) {
        const int cx = cxs, cy = cys;
        unsigned long long best = ~0ULL;
        for (int i = tid; i < n; i += 1024) {
            uint8_t f = stt[i];
            if (f & 1) continue;
            short4 e = P[i];
            float ds = nn_d2(e.x, e.y, cx, cy);
            float v = ds;
            if (!(f & 2)) { float de = nn_d2(e.z, e.w, cx, cy); if (!(ds <= de)) v = de; }
            unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)i;
            if (key < best) best = key;
        }
        for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_down(best, o, 64); if (t < best) best = t; }
        if ((tid & 63) == 0) wbest[tid >> 6] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = wbest[0];
            for (int w = 1; w < 16; w++) if (wbest[w] < b) b = wbest[w];
            int bi = (int)(b & 0xffffffffu);
            short4 e = P[bi]; uint8_t f = stt[bi];
            float ds = nn_d2(e.x, e.y, cx, cy), de = nn_d2(e.z, e.w, cx, cy);
            bool cl = (f & 2) != 0;
            bool flip = cl ? false : !(ds <= de);
            stt[bi] = f | 1; order[step] = bi; flips[step] = flip ? 1 : 0;
            if (cl || flip) { cxs = e.x; cys = e.y; } else { cxs = e.z; cys = e.w; }
        }
        __syncthreads();
    }
}
// minimum of a 64-bit key over the wavefront with DPP row shifts / broadcasts (no LDS round trips); every lane gets the result
__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long v) {
#define ORIP_DPP_MIN(ctrl, rmask) { \
        unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(unsigned)v, ctrl, rmask, 0xf, false); \
        unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(unsigned)(v >> 32), ctrl, rmask, 0xf, false); \
        unsigned long long t_ = ((unsigned long long)hi_ << 32) | lo_; if (t_ < v) v = t_; }
    ORIP_DPP_MIN(0x111, 0xf) ORIP_DPP_MIN(0x112, 0xf) ORIP_DPP_MIN(0x114, 0xf) ORIP_DPP_MIN(0x118, 0xf)      // row_shr 1, 2, 4, 8
    ORIP_DPP_MIN(0x142, 0xa) ORIP_DPP_MIN(0x143, 0xc)                                                          // row_bcast 15, 31
#undef ORIP_DPP_MIN
    unsigned rl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), rh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)rh << 32) | rl;
}
// Grid-pruned variant (same selection rule, same tie-break, n <= 11000 or so and int16 coordinates): the entry points (start of
// every polyline, end of every polyline that may be entered reversed) are bucketed into a G x G grid held in LDS next to the end
// points.  A greedy step scans the (2r+1)^2 cells around the cursor, r = 1, 3, 7, ...; it is final as soon as the best squared
// distance is below the squared gap between the cursor and the nearest unscanned cell (every unscanned entry is at least that far,
// so it can neither win nor tie), or the window covers the grid.  The chain of steps is strictly serial and a step looks at a few
// dozen entries, so ONE wavefront runs it: no barriers, no cross-wave exchange, and no other wave competing for the SIMD.
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_NN_OLDGRID / ORIP_NN_DBG): variants build only (make variants)
__global__ __launch_bounds__(64) void k_greedy_nn_grid(const NNEnds* __restrict__ ends, int n, const int* __restrict__ sel, int skip_if, int need_any, int rule07, int G,
                                                        int32_t* __restrict__ order, uint8_t* __restrict__ flips, unsigned long long* __restrict__ dbg) {
    ORIP_NN_GATE(sel, skip_if, need_any)
    extern __shared__ __align__(16) unsigned char smem[];
    // end points relative to the bounding-box origin (0 <= v < 2^14: differences, hence all float distances, are unchanged);
    // bit 15 of .x = used, bit 15 of .y = closed under rule07 (entered at the start only)
    ushort4* P = reinterpret_cast<ushort4*>(smem);                                 // (sx, sy, ex, ey)
    unsigned* cst = reinterpret_cast<unsigned*>(P + n);                            // G*G + 1 cell starts
    uint16_t* Eid = reinterpret_cast<uint16_t*>(cst + (G * G + 1));                // entries sorted by cell: idx << 1 | end
    __shared__ uint16_t ring[64];                                                  // (index << 1 | flip) of the last steps: results leave in batches of 64,
                                                                                   // a store per step would stall the chain on every later s_waitcnt vmcnt
    const int lane = threadIdx.x;
    // ---- bounding box
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -0x7fffffff, mxy = -0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        NNEnds e = ends[i];
        mnx = min(mnx, min(e.sx, e.ex)); mxx = max(mxx, max(e.sx, e.ex)); mny = min(mny, min(e.sy, e.ey)); mxy = max(mxy, max(e.sy, e.ey));
    }
    for (int o = 32; o > 0; o >>= 1) { mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64)); }
    const int ox = mnx, oy = mny;
    for (int i = lane; i < n; i += 64) {
        NNEnds e = ends[i];
        P[i] = make_ushort4((unsigned short)((e.sx - ox) | (i == seed ? 0x8000 : 0)), (unsigned short)((e.sy - oy) | ((rule07 && e.closed) ? 0x8000 : 0)),
                            (unsigned short)(e.ex - ox), (unsigned short)(e.ey - oy));
    }
    for (int i = lane; i <= G * G; i += 64) cst[i] = 0;
    __syncthreads();
    int sh = 0; while (((max(mxx - mnx, mxy - mny)) >> sh) >= G) sh++;                // power-of-two cells: v >> sh < G for every end point
    const int cs = 1 << sh;
    // ---- counting sort of the entries by cell
    for (int i = lane; i < n; i += 64) {
        const ushort4 e = P[i];
        atomicAdd(&cst[((e.y & 0x7fff) >> sh) * G + ((e.x & 0x7fff) >> sh)], 1u);
        if (!(e.y & 0x8000)) atomicAdd(&cst[(e.w >> sh) * G + (e.z >> sh)], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the G*G counts: a run of consecutive cells per lane
        const int per = (G * G + 63) / 64, c0 = lane * per, c1 = min(G * G, c0 + per);
        unsigned s = 0;
        for (int cc = c0; cc < c1; cc++) s += cst[cc];
        unsigned inc = s;
        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        unsigned run = inc - s;
        for (int cc = c0; cc < c1; cc++) { unsigned v = cst[cc]; cst[cc] = run; run += v; }
        if (lane == 63) cst[G * G] = inc;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {         // scatter; cst[c] ends up as the END of cell c, i.e. start(c) = c ? cst[c-1] : 0
        const ushort4 e = P[i];
        Eid[atomicAdd(&cst[((e.y & 0x7fff) >> sh) * G + ((e.x & 0x7fff) >> sh)], 1u)] = (uint16_t)(i << 1);
        if (!(e.y & 0x8000)) Eid[atomicAdd(&cst[(e.w >> sh) * G + (e.z >> sh)], 1u)] = (uint16_t)((i << 1) | 1);
    }
    int cx, cy;                                   // cursor, relative to the origin as well
    { const ushort4 e = P[seed]; if (e.y & 0x8000) { cx = e.x & 0x7fff; cy = e.y & 0x7fff; } else { cx = e.z; cy = e.w; } }
    if (lane == 0) ring[0] = (uint16_t)(seed << 1);
    __syncthreads();
    int prev = seed;
    unsigned long long d_rounds = 0, d_scanned = 0, d_full = 0;
    for (int step = 1; step < n; step++) {
        const int gx = cx >> sh, gy = cy >> sh;
        unsigned long long best = ~0ULL;
        for (int r = 1;; r = 2 * r + 1) {
            const int x0 = max(0, gx - r), x1 = min(G - 1, gx + r), y0 = max(0, gy - r), y1 = min(G - 1, gy + r);
            unsigned long long mine = ~0ULL;
            auto consider = [&](unsigned q) {
                const unsigned id = Eid[q]; const int i = (int)(id >> 1);
                const ushort4 e = P[i];
                if ((e.x & 0x8000) || i == prev) return;
                float v = (id & 1) ? nn_d2((int)e.z, (int)e.w, cx, cy) : nn_d2((int)(e.x & 0x7fff), (int)(e.y & 0x7fff), cx, cy);
                unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)i;
                if (key < mine) mine = key;
            };
            if (y1 - y0 <= 2) {
                // up to three rows: lanes 0..2 fetch the entry ranges of the rows, then all lanes share the concatenated entries
                unsigned lo_l = 0, n_l = 0;
                if (lane <= y1 - y0) { const int c0 = (y0 + lane) * G + x0, c1 = (y0 + lane) * G + x1; lo_l = c0 ? cst[c0 - 1] : 0u; n_l = cst[c1] - lo_l; }
                const unsigned lo0 = (unsigned)__builtin_amdgcn_readlane((int)lo_l, 0), n0 = (unsigned)__builtin_amdgcn_readlane((int)n_l, 0);
                const unsigned lo1 = (unsigned)__builtin_amdgcn_readlane((int)lo_l, 1), n1 = (unsigned)__builtin_amdgcn_readlane((int)n_l, 1);
                const unsigned lo2 = (unsigned)__builtin_amdgcn_readlane((int)lo_l, 2), n2 = (unsigned)__builtin_amdgcn_readlane((int)n_l, 2);
                const unsigned total = n0 + n1 + n2;
                if (dbg) d_scanned += total;
                for (unsigned t = lane; t < total; t += 64) consider(t < n0 ? lo0 + t : (t - n0 < n1 ? lo1 + (t - n0) : lo2 + (t - n0 - n1)));
            } else
            for (int row = y0; row <= y1; row++) {
                const int c0 = row * G + x0, c1 = row * G + x1;
                const unsigned lo = c0 ? cst[c0 - 1] : 0u, hi = cst[c1];
                if (dbg) d_scanned += hi - lo;
                for (unsigned q = lo + lane; q < hi; q += 64) consider(q);
            }
            mine = wave_min_key(mine);
            best = mine;
            if (dbg) d_rounds++;
            if (x0 == 0 && y0 == 0 && x1 == G - 1 && y1 == G - 1) { if (dbg) d_full++; break; }     // everything scanned
            if (best != ~0ULL) {
                int gap = 0x7fff;                                                    // distance to the nearest unscanned cell, over the open sides
                if (x0 > 0) gap = min(gap, cx - (x0 << sh) + 1);
                if (x1 < G - 1) gap = min(gap, ((x1 + 1) << sh) - cx);
                if (y0 > 0) gap = min(gap, cy - (y0 << sh) + 1);
                if (y1 < G - 1) gap = min(gap, ((y1 + 1) << sh) - cy);
                const float bd = __uint_as_float((unsigned)(best >> 32));
                if ((double)bd * (1.0 + 1e-6) < (double)(gap * gap)) break;          // gap < 2^15: the square is exact in int
            }
        }
        const int bi = (int)(best & 0xffffffffu);
        const ushort4 e = P[bi];
        const int sx = e.x & 0x7fff, sy = e.y & 0x7fff;
        float ds = nn_d2(sx, sy, cx, cy), de = nn_d2((int)e.z, (int)e.w, cx, cy);
        const bool cl = (e.y & 0x8000) != 0;
        const bool flip = cl ? false : !(ds <= de);
        if (lane == 0) { P[bi].x = (unsigned short)(e.x | 0x8000); ring[step & 63] = (uint16_t)((bi << 1) | (flip ? 1 : 0)); }
        if ((step & 63) == 63) { const unsigned v = ring[lane]; order[step - 63 + lane] = (int32_t)(v >> 1); flips[step - 63 + lane] = (uint8_t)(v & 1u); }
        if (cl || flip) { cx = sx; cy = sy; } else { cx = e.z; cy = e.w; }
        prev = bi;
    }
    { const int done = n & ~63; if (done + lane < n) { const unsigned v = ring[lane]; order[done + lane] = (int32_t)(v >> 1); flips[done + lane] = (uint8_t)(v & 1u); } }
    if (dbg && lane == 0) { dbg[0] = d_rounds; dbg[1] = d_scanned; dbg[2] = d_full; dbg[3] = (unsigned long long)cs; }
}
#endif

// ---- descriptor-driven gather: output polyline k = src points [begin[k], begin[k]+len[k]) (reversed if rev[k]) ----
struct GatherDesc { int64_t begin; int64_t len; int32_t rev; int32_t src; };     // src: index of the source polyline (walk-coded sources are addressed by polyline, not by point)
__global__ __launch_bounds__(256) void k_gather_lens(const GatherDesc* __restrict__ d, int64_t n, int64_t* __restrict__ lens) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) lens[i] = d[i].len;
    if (i == n) lens[i] = 0;
}
// flat form: a block copies 4096 consecutive OUTPUT points whatever polylines they belong to (one block per polyline would leave
// the long polylines as the tail of the launch); the polyline of a point is found by bisecting the output offsets once per block
// and walking forward from there
__global__ __launch_bounds__(256) void k_gather_pts(const GatherDesc* __restrict__ d, int64_t n, const int32_t* __restrict__ src,
                                                     const int64_t* __restrict__ out_off, int32_t* __restrict__ dst, int64_t total) {
    const int64_t start = (int64_t)blockIdx.x * 4096;
    if (start >= total) return;
    int64_t lo = 0, hi = n - 1;                      // last polyline k with out_off[k] <= start
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (out_off[mid] <= start) lo = mid; else hi = mid - 1; }
    int64_t k = lo;
    const int2* s2 = reinterpret_cast<const int2*>(src); int2* o2 = reinterpret_cast<int2*>(dst);
    for (int64_t idx = start + threadIdx.x; idx < min(total, start + 4096); idx += 256) {
        while (out_off[k + 1] <= idx) k++;           // empty polylines are skipped too
        const GatherDesc g = d[k];
        const int64_t j = idx - out_off[k];
        o2[idx] = s2[g.begin + (g.rev ? g.len - 1 - j : j)];
    }
}
// Builds dst (DPolys) from descriptors (device array of n descs).  lens/off scratch in ctx->tmpE.
static int vgather(orip_ctx* c, const GatherDesc* d, int64_t n, const int32_t* src, DPolys& dst, int64_t known_total = -1) {
    dst.n = n; dst.total = 0; dst.set_explicit();
    HIPC(c, dst.off.ensure((size_t)(n + 1) * 8 + 64));
    if (n == 0) { HIPC(c, hipMemsetAsync(dst.off.p, 0, 8, LN(c).stream)); return 0; }
    HIPC(c, LN(c).tmpE.ensure((size_t)(n + 1) * 8 + 64));
    hipLaunchKernelGGL(k_gather_lens, dim3(cdiv(n + 1, 256)), dim3(256), 0, LN(c).stream, d, n, LN(c).tmpE.as<int64_t>());
    ORIP_TRY(vscan_excl<int64_t>(c, LN(c).tmpE.as<int64_t>(), dst.off.as<int64_t>(), (size_t)n + 1));
    int64_t total = known_total;
    if (total < 0) ORIP_TRY(vread(c, &total, dst.off.as<int64_t>() + n));
    dst.total = total;
    HIPC(c, dst.pts.ensure((size_t)std::max<int64_t>(total, 1) * 8 + 64));
    if (total > 0) hipLaunchKernelGGL(k_gather_pts, dim3((unsigned)cdiv(total, 4096)), dim3(256), 0, LN(c).stream, d, n, src, dst.off.as<int64_t>(), dst.pts.as<int32_t>(), total);
    HIPC(c, hipGetLastError());
    return 0;
}
// The same selection over a walk-coded source moves no points: output polyline k is a VIEW (walker.h) of the walk behind source
// polyline d[k].src -- its first d[k].len points, reversed if d[k].rev -- composed with the view the source polyline already is.
__global__ __launch_bounds__(256) void k_view_select(const GatherDesc* __restrict__ d, int64_t n, const VView* __restrict__ sview, const VWalk* __restrict__ walk,
                                                      VView* __restrict__ out, int64_t* __restrict__ lens) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k == n) lens[k] = 0;
    if (k >= n) return;
    const GatherDesc g = d[k];
    VView s;
    if (sview) s = sview[g.src]; else { s.wid = (unsigned)g.src; s.first = 0u; s.len = walk[g.src].len; s.rev = 0u; }
    VView o; o.wid = s.wid; o.len = (unsigned)g.len; o.rev = s.rev ^ (g.rev ? 1u : 0u);
    o.first = s.rev ? s.first + s.len - (unsigned)g.len : s.first;
    out[k] = o; lens[k] = g.len;
}
// known_total >= 0: the caller knows the number of points selected (e.g. a permutation of the whole source list): no host read
static int vgather_views(orip_ctx* c, const GatherDesc* d, int64_t n, const DPolys& src, DPolys& dst, int64_t known_total = -1) {
    VSrc vs_; ORIP_TRY(vsrc_of(c, src, vs_));
    dst.n = n; dst.total = 0;
    dst.virt = true; dst.pts_ok = false; dst.vident = false; dst.vlayer = src.vlayer; dst.vepoch = src.vepoch;
    dst.scaled = src.scaled; dst.vsepoch = src.vsepoch; dst.pf_tag = src.pf_tag;      // the views keep naming the source's walks: what was computed per walk and direction stays addressable
    HIPC(c, dst.off.ensure((size_t)(n + 1) * 8 + 64));
    if (n == 0) { HIPC(c, hipMemsetAsync(dst.off.p, 0, 8, LN(c).stream)); return 0; }
    HIPC(c, dst.vview.ensure((size_t)n * sizeof(VView) + 64));
    HIPC(c, LN(c).tmpE.ensure((size_t)(n + 1) * 8 + 64));
    hipLaunchKernelGGL(k_view_select, dim3(cdiv(n + 1, 256)), dim3(256), 0, LN(c).stream, d, n, vs_.view, vs_.walk, dst.vview.as<VView>(), LN(c).tmpE.as<int64_t>());
    ORIP_TRY(vscan_excl<int64_t>(c, LN(c).tmpE.as<int64_t>(), dst.off.as<int64_t>(), (size_t)n + 1));
    int64_t total = known_total;
    if (total < 0) ORIP_TRY(vread(c, &total, dst.off.as<int64_t>() + n));
    dst.total = total;
    HIPC(c, hipGetLastError());
    return 0;
}
// selection out of a list of either kind
static int vgather_list(orip_ctx* c, const GatherDesc* d, int64_t n, const DPolys& src, DPolys& dst, int64_t known_total = -1) {
    if (is_coded(src)) return vgather_views(c, d, n, src, dst, known_total);
    return vgather(c, d, n, src.pts.as<int32_t>(), dst, known_total);
}
// explicit points of a list of either kind: 4096 consecutive output points per block (as k_gather_pts)
template <class Src>
__global__ __launch_bounds__(256) void k_expand_pts(Src src, int64_t n, int2* __restrict__ dst, int64_t total) {
    const int64_t start = (int64_t)blockIdx.x * 4096;
    if (start >= total) return;
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (src.off[mid] <= start) lo = mid; else hi = mid - 1; }
    int64_t k = lo, kc = -1;
    auto cu = src.cur(k);
    for (int64_t idx = start + threadIdx.x; idx < min(total, start + 4096); idx += 256) {
        while (src.off[k + 1] <= idx) k++;
        if (k != kc) { cu = src.cur(k); kc = k; }
        dst[idx] = cu.at(idx - src.off[k]);
    }
}

// order/flip -> descriptors over a source list
__global__ __launch_bounds__(256) void k_desc_from_order(const int64_t* __restrict__ off, const int32_t* __restrict__ order, const uint8_t* __restrict__ flips,
                                                          int64_t n, int open_view, const PolyFeat* __restrict__ feat, GatherDesc* __restrict__ d) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int i = order ? order[k] : (int)k;
    GatherDesc g; g.begin = off[i]; g.len = open_view ? feat[i].n : (off[i + 1] - off[i]); g.rev = flips ? flips[k] : 0; g.src = i;
    d[k] = g;
}

// argmax with first-max tie-break over a float / double field of PolyFeat (seed of the greedy orders); tiny: single block
// (one 256-thread block: a 1024-thread block waits for a CU with sixteen free wave slots -- half a millisecond next to the other layers' work, in front of the greedy chain)
__global__ __launch_bounds__(256) void k_argmax_feat(const PolyFeat* __restrict__ f, int n, int use_arc, int* __restrict__ out, const NNEnds* __restrict__ e = nullptr) {
    __shared__ double bv[256]; __shared__ int bi[256];
    __shared__ int bad_s;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    double v = -1.0; int idx = 0x7fffffff; int bad = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double x = use_arc ? f[i].arc : (double)f[i].per; if (x > v) { v = x; idx = i; }
        if (e) {      // the coordinate-range flags of the greedy kernels in the same pass (k_ends_fit16's: bit 0 beyond int16, bit 1 beyond 15 bits): out[1]
            const NNEnds q = e[i];
            auto outside = [&](int lo, int hi) { return q.sx < lo || q.sx > hi || q.sy < lo || q.sy > hi || q.ex < lo || q.ex > hi || q.ey < lo || q.ey > hi; };
            if (outside(-32768, 32767)) bad |= 1;
            if (outside(-16384, 16383)) bad |= 2;
        }
    }
    if (bad) atomicOr(&bad_s, bad);
    bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            double o = bv[threadIdx.x + s]; int oi = bi[threadIdx.x + s];
            if (o > bv[threadIdx.x] || (o == bv[threadIdx.x] && oi < bi[threadIdx.x])) { bv[threadIdx.x] = o; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out = bi[0]; if (e) out[1] = bad_s; }
}
template <class Src>
__global__ __launch_bounds__(256) void k_ends_from_feat(const PolyFeat* __restrict__ f, int64_t n, int rule07, Src src, NNEnds* __restrict__ e) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    NNEnds q; q.sx = f[i].sx; q.sy = f[i].sy; q.ex = f[i].ex; q.ey = f[i].ey; q.closed = f[i].closed;
    if (rule07 && f[i].closed) {   // _ends (07:12-17): a closed contour ends at its second-to-last point
        int64_t m = src.len(i);
        if (m > 1) { const int2 p = src.cur(i).at(m - 2); q.ex = p.x; q.ey = p.y; }
    }
    e[i] = q;
}

// ---- the common step of k_greedy_nn_fast as ONE asm statement that runs step after step (r03).
// The compiled step is ~260 instructions on its usual path and stalls a dozen times on scalar instructions that consume vector results (cell
// ranges read out lane by lane, the gap test, the winner's end points): 1 750 cycles.  The usual path is narrow -- the 3x3 window away from the
// first / last cell row, 1..128 candidates, a unique nearest one that passes the gap test (94 % of the steps of the bench image) -- and this
// loop takes exactly that path in ~100 instructions: the six range words are read out back to back (one stall), the gap threshold is
// computed while the LDS reads are in flight, validity / used flags are vector selects, every lane settles reading direction and next cursor
// of its own candidate, the winner lane itself writes the used flag (exec = the one-bit tie mask), the result leaves through v_writelane.
// Anything else (empty or crowded window, a tie, a failed gap test, the border rows) leaves the loop BEFORE the step has changed anything;
// the caller then takes that one step with the compiled code.  Same arithmetic as the compiled step (unfused float ops, the same integer test).
// Returns 0: step == n; 1: 64 results are in `ringv` (step is a multiple of 64); 2..7: the step at `step` is the caller's (the reason: see the exits).
__device__ __forceinline__ int nn_asm_steps(int& cx, int& cy, int& step, unsigned& ringv, int n, int sh, int G, unsigned lds_p, unsigned lds_cst, unsigned lds_eid,
                                            unsigned n_ent_m1, int rowoff, int isend, int lane, int& dbg_cnt) {
    int ev; int s_cnt = 0;
    int s_cx = __builtin_amdgcn_readfirstlane(cx), s_cy = __builtin_amdgcn_readfirstlane(cy), s_step = __builtin_amdgcn_readfirstlane(step);
    const int s_n = __builtin_amdgcn_readfirstlane(n), s_sh = __builtin_amdgcn_readfirstlane(sh), s_G = __builtin_amdgcn_readfirstlane(G), s_Gm1 = s_G - 1;
    const unsigned s_p = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_p), s_cst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_cst),
                   s_eid = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_eid), s_nem1 = (unsigned)__builtin_amdgcn_readfirstlane((int)n_ent_m1);
    // ordinal T inside the window -> byte address Q of its entry (clamped into the table): row 0 holds the ordinals [0, n0), row 1 [n0, n01), row 2 the rest
#define ORIP_NN_Q(T, Q, TMP1, TMP2)                                                                                    \
        "v_cmp_gt_u32_e64 s[94:95], s73, " T "\n\t"                 /* (a vector compare's SGPR result is read two instructions later at the earliest) */ \
        "v_cmp_gt_u32 vcc, s68, " T "\n\t"                                                                             \
        "v_add_u32 " Q ", s67, " T "\n\t"                                                                              \
        "v_add_u32 " TMP1 ", s75, " T "\n\t"                                                                           \
        "v_add_u32 " TMP2 ", s76, " T "\n\t"                                                                           \
        "v_cndmask_b32_e64 " TMP1 ", " TMP2 ", " TMP1 ", s[94:95]\n\t"                                                 \
        "v_cndmask_b32 " Q ", " TMP1 ", " Q ", vcc\n\t"                                                                \
        "v_min_u32 " Q ", %[nem1], " Q "\n\t"                                                                          \
        "v_lshl_add_u32 " Q ", " Q ", 1, %[eidb]\n\t"
    // entry word IDW (index << 1 | end) -> its end bit, the address PA of the polyline's end points, and their read into v[E0:E1] issued
#define ORIP_NN_FETCH(IDW, ENDBIT, PA, E0, E1)                                                                         \
        "v_lshrrev_b32 v50, 1, " IDW "\n\t"                                                                            \
        "v_lshl_add_u32 " PA ", v50, 3, %[pb]\n\t"                                                                     \
        "ds_read_b64 v[" E0 ":" E1 "], " PA "\n\t"                                                                     \
        "v_and_b32 " ENDBIT ", 1, " IDW "\n\t"
    // key K of the candidate (squared distance pattern of the entry's end point; ~0 when the polyline is used or the ordinal lies beyond the window)
#define ORIP_NN_KEY(ENDBIT, K, E0, E1, VALID)                                                                          \
        "v_cmp_eq_u32_e64 s[94:95], 0, " ENDBIT "\n\t"                                                                 \
        "v_and_b32 v54, 0x7fff7fff, v" E0 "\n\t"                                                                       \
        "v_and_b32 v55, 0x8000, v" E0 "\n\t"                                                                           \
        "v_cndmask_b32_e64 v50, v" E1 ", v54, s[94:95]\n\t"                                                            \
        "v_cvt_f32_u32_sdwa v56, v50 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t"                          \
        "v_cvt_f32_u32_sdwa v57, v50 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"                          \
        "v_sub_f32 v56, v56, v40\n\t"                                                                                  \
        "v_sub_f32 v57, v57, v41\n\t"                                                                                  \
        "v_mul_f32 v56, v56, v56\n\t"                                                                                  \
        "v_mul_f32 v57, v57, v57\n\t"                                                                                  \
        "v_add_f32 " K ", v56, v57\n\t"                                                                                \
        "v_cmp_eq_u32 vcc, 0, v55\n\t"                                                                                 \
        "v_cndmask_b32 " K ", -1, " K ", vcc\n\t"                                                                      \
        "v_cndmask_b32_e64 " K ", -1, " K ", " VALID "\n\t"
    // (entry word, K) in v[IK0:IK1], end points v[E0:E1], address PA: better than the best so far (v[58:59], v[52:53], v51)?  Smaller key, then smaller entry word.
#define ORIP_NN_MERGE(IK0, IK1, E0, E1, PA)                                                                            \
        "v_cmp_lt_u64 vcc, v[" IK0 ":" IK1 "], v[58:59]\n\t"                                                           \
        "v_cndmask_b32 v58, v58, v" IK0 ", vcc\n\t"                                                                    \
        "v_cndmask_b32 v59, v59, v" IK1 ", vcc\n\t"                                                                    \
        "v_cndmask_b32 v52, v52, v" E0 ", vcc\n\t"                                                                     \
        "v_cndmask_b32 v53, v53, v" E1 ", vcc\n\t"                                                                     \
        "v_cndmask_b32 v51, v51, " PA ", vcc\n\t"
    // minimum of v47 over the wave into lane 63 (the compiler's sequence for the same reduction; a DPP source is read two instructions after it was written)
#define ORIP_NN_MIN6                                                                                                   \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                                      \
        "s_nop 1\n\t"                                                                                                  \
        "v_min_u32_dpp v47, v47, v47 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                                      \
        "s_nop 1\n\t"
    asm volatile(
        "s_mov_b32 s89, m0\n\t"
        "s_and_b32 m0, %[step], 63\n\t"
        "s_lshl_b32 s87, 1, %[sh]\n\t"                              // cell
        "s_add_i32 s88, s87, -1\n\t"                                // cell - 1
        "s_add_i32 s79, s87, 1\n\t"
        "s_mul_i32 s59, s79, s79\n\t"
        "s_lshr_b32 s80, s59, 18\n\t"
        "s_sub_i32 s59, s59, s80\n\t"
        "s_add_i32 s59, s59, -1\n\t"                              // the gap test's threshold for the smallest gap a 3x3 window can have (cell + 1)
        "s_mov_b32 s81, -1\n\t"                                     // cell the range words in s67 .. s76 belong to: none yet
        "s_mov_b32 s58, 0\n\t"                                      // 1: the lanes hold the candidates of that cell's window (v45, v49, v51, v[52:53], s[92:93])
        "v_cvt_f32_i32 v40, %[cx]\n\t"
        "v_cvt_f32_i32 v41, %[cy]\n\t"
        "L_step%=:\n\t"
        "s_lshr_b32 s60, %[cx], %[sh]\n\t"
        "s_lshr_b32 s61, %[cy], %[sh]\n\t"
        "s_lshl_b32 s79, s61, 16\n\t"
        "s_or_b32 s79, s79, s60\n\t"
        "s_cmp_eq_u32 s79, s81\n\t"
        "s_cbranch_scc1 L_samecell%=\n\t"
        // ---- another cell: the window's range words
        "s_mov_b32 s81, s79\n\t"
        "s_mov_b32 s58, 0\n\t"
        "s_sub_i32 s62, s60, 1\n\t"
        "s_max_i32 s62, s62, 0\n\t"                                 // x0
        "s_add_i32 s63, s60, 1\n\t"
        "s_min_i32 s63, s63, %[Gm1]\n\t"
        "s_add_i32 s63, s63, 1\n\t"                                 // x1 + 1
        "s_sub_i32 s64, s61, 1\n\t"
        "s_max_i32 s64, s64, 0\n\t"                                 // y0
        "s_add_i32 s65, s61, 1\n\t"
        "s_min_i32 s65, s65, %[Gm1]\n\t"
        "s_sub_i32 s65, s65, s64\n\t"                               // y1 - y0: 2, or 1 in the first / last cell row
        "s_sub_i32 s66, s63, s62\n\t"
        "v_add_u32 v42, s64, %[rowoff]\n\t"                         // lanes 0..5: row of the range word, ...
        "v_mul_u32_u24 v43, s66, %[isend]\n\t"
        "v_add_u32 v43, s62, v43\n\t"                               // ... its cell column (x0: start of the row's range, x1 + 1: its end)
        "v_mad_u32_u24 v42, v42, %[G], v43\n\t"
        "v_lshl_add_u32 v42, v42, 2, %[cstb]\n\t"
        "ds_read_b32 v44, v42\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readlane_b32 s67, v44, 0\n\t"
        "v_readlane_b32 s68, v44, 1\n\t"
        "v_readlane_b32 s69, v44, 2\n\t"
        "v_readlane_b32 s70, v44, 3\n\t"
        "v_readlane_b32 s71, v44, 4\n\t"
        "v_readlane_b32 s72, v44, 5\n\t"
        "s_sub_i32 s68, s68, s67\n\t"                               // n0
        "s_sub_i32 s70, s70, s69\n\t"                               // n1
        "s_sub_i32 s72, s72, s71\n\t"                               // n2 ...
        "s_cmp_lt_u32 s65, 2\n\t"
        "s_cselect_b32 s72, 0, s72\n\t"                             // ... none when the third row lies outside the grid
        "s_add_i32 s73, s68, s70\n\t"                               // n01
        "s_add_i32 s74, s73, s72\n\t"                               // total
        "s_sub_i32 s75, s69, s68\n\t"                               // lo1 - n0
        "s_sub_i32 s76, s71, s73\n\t"                               // lo2 - n01
        "s_nop 1\n\t"
        "s_branch L_ranges%=\n\t"
        "L_samecell%=:\n\t"                                          // the lanes may still hold this window's candidates: then no LDS read at all
        "s_cmp_eq_u32 s58, 1\n\t"
        "s_cbranch_scc1 L_hit1%=\n\t"
        "s_cmp_eq_u32 s58, 2\n\t"
        "s_cbranch_scc1 L_hit2%=\n\t"
        "L_ranges%=:\n\t"
        "s_cmp_eq_u32 s74, 0\n\t"
        "s_cbranch_scc1 L_fb3%=\n\t"
        "s_cmp_gt_u32 s74, 64\n\t"
        "s_cbranch_scc1 L_many%=\n\t"
        // ---- up to 64 candidates: one per lane, kept in the lanes while the cursor stays in the cell
        ORIP_NN_Q("%[lane]", "v46", "v47", "v48")
        "ds_read_u16 v49, v46\n\t"
        "v_cmp_gt_u32_e64 s[92:93], s74, %[lane]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        ORIP_NN_FETCH("v49", "v45", "v51", "52", "53")
        "s_mov_b32 s58, 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "L_key%=:\n\t"
        ORIP_NN_KEY("v45", "v59", "52", "53", "s[92:93]")
        "v_mov_b32 v58, v49\n\t"
        "s_branch L_best%=\n\t"
        // ---- 65 .. 128 candidates: two per lane (A: v66, v70, v65, v[68:69], s[92:93]; B: v76, v71, v73, v[74:75], s[96:97]), kept like the single ones
        "L_many%=:\n\t"
        "s_cmp_gt_u32 s74, 128\n\t"
        "s_cbranch_scc1 L_loop%=\n\t"
        "v_add_u32 v43, 64, %[lane]\n\t"
        ORIP_NN_Q("%[lane]", "v46", "v47", "v48")
        "ds_read_u16 v66, v46\n\t"
        ORIP_NN_Q("v43", "v72", "v47", "v48")
        "ds_read_u16 v76, v72\n\t"
        "v_cmp_gt_u32_e64 s[92:93], s74, %[lane]\n\t"
        "v_cmp_gt_u32_e64 s[96:97], s74, v43\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        ORIP_NN_FETCH("v66", "v70", "v65", "68", "69")
        ORIP_NN_FETCH("v76", "v71", "v73", "74", "75")
        "s_mov_b32 s58, 2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "L_key2%=:\n\t"
        ORIP_NN_KEY("v70", "v67", "68", "69", "s[92:93]")
        ORIP_NN_KEY("v71", "v77", "74", "75", "s[96:97]")
        "v_mov_b32 v58, v66\n\t"
        "v_mov_b32 v59, v67\n\t"
        "v_mov_b32 v52, v68\n\t"
        "v_mov_b32 v53, v69\n\t"
        "v_mov_b32 v51, v65\n\t"
        ORIP_NN_MERGE("76", "77", "74", "75", "v73")
        "s_branch L_best%=\n\t"
        // ---- more than 128: 128 per turn, the two halves' LDS reads in flight together; nothing is kept
        "L_loop%=:\n\t"
        "s_add_i32 %[cnt], %[cnt], 0x100000\n\t"
        "s_mov_b32 s58, 0\n\t"
        "s_mov_b32 s98, 0\n\t"
        "v_mov_b32 v58, -1\n\t"
        "v_mov_b32 v59, -1\n\t"
        "v_mov_b32 v52, 0\n\t"
        "v_mov_b32 v53, 0\n\t"
        "v_mov_b32 v51, 0\n\t"
        "L_pair%=:\n\t"
        "v_add_u32 v42, s98, %[lane]\n\t"
        "v_add_u32 v43, 64, v42\n\t"
        ORIP_NN_Q("v42", "v46", "v47", "v48")
        "ds_read_u16 v66, v46\n\t"
        ORIP_NN_Q("v43", "v72", "v47", "v48")
        "ds_read_u16 v76, v72\n\t"
        "v_cmp_gt_u32_e64 s[92:93], s74, v42\n\t"
        "v_cmp_gt_u32_e64 s[96:97], s74, v43\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        ORIP_NN_FETCH("v66", "v70", "v65", "68", "69")
        ORIP_NN_FETCH("v76", "v71", "v73", "74", "75")
        "s_waitcnt lgkmcnt(0)\n\t"
        ORIP_NN_KEY("v70", "v67", "68", "69", "s[92:93]")
        ORIP_NN_KEY("v71", "v77", "74", "75", "s[96:97]")
        ORIP_NN_MERGE("66", "67", "68", "69", "v65")
        ORIP_NN_MERGE("76", "77", "74", "75", "v73")
        "s_add_i32 s98, s98, 128\n\t"
        "s_cmp_lt_u32 s98, s74\n\t"
        "s_cbranch_scc1 L_pair%=\n\t"
        "L_best%=:\n\t"
        // ---- every lane: the next cursor if its candidate wins; the wave: the smallest key.  The winner is read backwards exactly when its
        // END entry won: had the start been as near or nearer it would hold a key as small or smaller (the gap test says every entry nearer
        // than the gap was scanned), and on equal keys the smaller entry word -- the start -- is taken below.  So entry word == index << 1 | flip.
        "v_mov_b32 v47, v59\n\t"
        "v_and_b32 v54, 0x7fff7fff, v52\n\t"
        "v_and_b32 v64, 1, v58\n\t"
        "v_min_u32_dpp v47, v47, v47 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_cmp_ne_u32 vcc, 0, v64\n\t"
        "v_cmp_gt_i32_e64 s[90:91], 0, v52\n\t"                     // closed (bit 31)
        "v_min_u32_dpp v47, v47, v47 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 v62, v53, v54, vcc\n\t"
        "v_or_b32 v60, 0x8000, v52\n\t"
        "v_min_u32_dpp v47, v47, v47 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 v62, v62, v54, s[90:91]\n\t"             // next cursor: the start when closed or read backwards, else the end
        "s_nop 0\n\t"
        "v_min_u32_dpp v47, v47, v47 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp v47, v47, v47 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp v47, v47, v47 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 s77, v47, 63\n\t"
        "s_cmp_eq_u32 s77, -1\n\t"
        "s_cbranch_scc1 L_fb5%=\n\t"
        "v_cvt_u32_f32 v48, s77\n\t"
        "v_add_u32 v48, 1, v48\n\t"
        "v_cmp_ge_u32 vcc, s59, v48\n\t"
        "s_and_b64 s[94:95], vcc, exec\n\t"
        "s_cbranch_scc0 L_gap%=\n\t"
        "L_gapok%=:\n\t"
        "v_cmp_eq_u32_e64 s[82:83], s77, v59\n\t"
        "s_bcnt1_i32_b64 s79, s[82:83]\n\t"
        "s_cmp_eq_u32 s79, 1\n\t"
        "s_cbranch_scc1 L_win%=\n\t"
        // several lanes at the smallest distance: the smallest entry word among them (07:67 -- the first polyline in list order, its start first)
        "v_cndmask_b32_e64 v47, -1, v58, s[82:83]\n\t"
        ORIP_NN_MIN6
        "v_readlane_b32 s79, v47, 63\n\t"
        "s_nop 1\n\t"
        "v_cmp_eq_u32_e64 s[94:95], s79, v58\n\t"
        "s_and_b64 s[82:83], s[82:83], s[94:95]\n\t"
        "L_win%=:\n\t"
        "s_ff1_i32_b64 s84, s[82:83]\n\t"
        "v_readlane_b32 s85, v58, s84\n\t"
        "v_readlane_b32 s86, v62, s84\n\t"
        "s_mov_b64 s[90:91], exec\n\t"
        "s_mov_b64 exec, s[82:83]\n\t"
        "ds_write_b32 v51, v60\n\t"                                 // the used flag, by the winning lane
        "s_mov_b64 exec, s[90:91]\n\t"
        "v_writelane_b32 %[ringv], s85, m0\n\t"
        // the lanes' copies of the winner's end points (its other entry may sit in this window too) take the flag as well
        "s_lshr_b32 s79, s85, 1\n\t"
        "s_cmp_eq_u32 s58, 2\n\t"
        "s_cbranch_scc1 L_upd2%=\n\t"
        "v_lshrrev_b32 v50, 1, v49\n\t"
        "v_or_b32 v55, 0x8000, v52\n\t"
        "v_cmp_eq_u32 vcc, s79, v50\n\t"
        "v_cndmask_b32 v52, v52, v55, vcc\n\t"
        "s_branch L_next%=\n\t"
        "L_upd2%=:\n\t"
        "v_lshrrev_b32 v50, 1, v66\n\t"
        "v_or_b32 v55, 0x8000, v68\n\t"
        "v_cmp_eq_u32 vcc, s79, v50\n\t"
        "v_cndmask_b32 v68, v68, v55, vcc\n\t"
        "v_lshrrev_b32 v50, 1, v76\n\t"
        "v_or_b32 v55, 0x8000, v74\n\t"
        "v_cmp_eq_u32 vcc, s79, v50\n\t"
        "v_cndmask_b32 v74, v74, v55, vcc\n\t"
        "L_next%=:\n\t"
        "s_and_b32 %[cx], s86, 0xffff\n\t"
        "s_lshr_b32 %[cy], s86, 16\n\t"
        "v_cvt_f32_i32 v40, %[cx]\n\t"
        "v_cvt_f32_i32 v41, %[cy]\n\t"
        "s_add_i32 %[step], %[step], 1\n\t"
        "s_add_i32 m0, m0, 1\n\t"
        "s_cmp_eq_u32 m0, 64\n\t"
        "s_cbranch_scc1 L_flush%=\n\t"
        "s_cmp_ge_u32 %[step], %[n]\n\t"
        "s_cbranch_scc1 L_done%=\n\t"
        "s_branch L_step%=\n\t"
        // ---- the nearest candidate is farther than cell + 1: the exact gap = cell + min over x and y of min(l + 1, cell - l), l = cursor inside its cell
        "L_gap%=:\n\t"
        "s_and_b32 s79, %[cx], s88\n\t"
        "s_sub_i32 s80, s87, s79\n\t"
        "s_add_i32 s79, s79, 1\n\t"
        "s_min_i32 s79, s79, s80\n\t"
        "s_and_b32 s80, %[cy], s88\n\t"
        "s_sub_i32 s78, s87, s80\n\t"
        "s_add_i32 s80, s80, 1\n\t"
        "s_min_i32 s80, s80, s78\n\t"
        "s_min_i32 s79, s79, s80\n\t"
        "s_add_i32 s79, s79, s87\n\t"
        "s_mul_i32 s78, s79, s79\n\t"
        "s_lshr_b32 s80, s78, 18\n\t"
        "s_sub_i32 s78, s78, s80\n\t"
        "s_add_i32 s78, s78, -1\n\t"
        "v_cmp_ge_u32 vcc, s78, v48\n\t"
        "s_and_b64 s[94:95], vcc, exec\n\t"
        "s_cbranch_scc1 L_gapok%=\n\t"
        "s_mov_b32 %[ev], 6\n\t"                                       // 6: the gap test wants a wider window
        "s_branch L_out%=\n\t"
        "L_hit1%=:\n\t"
        "s_add_i32 %[cnt], %[cnt], 1\n\t"
        "s_branch L_key%=\n\t"
        "L_hit2%=:\n\t"
        "s_add_i32 %[cnt], %[cnt], 0x400\n\t"
        "s_branch L_key2%=\n\t"
        "L_fb3%=:\n\t"                                                 // 3: empty window
        "s_mov_b32 %[ev], 3\n\t"
        "s_branch L_out%=\n\t"
        "L_fb5%=:\n\t"                                                 // 5: every candidate used
        "s_mov_b32 %[ev], 5\n\t"
        "s_branch L_out%=\n\t"
        "L_flush%=:\n\t"
        "s_mov_b32 %[ev], 1\n\t"
        "s_branch L_out%=\n\t"
        "L_done%=:\n\t"
        "s_mov_b32 %[ev], 0\n\t"
        "L_out%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 m0, s89\n\t"
        : [ev] "=&s"(ev), [cx] "+s"(s_cx), [cy] "+s"(s_cy), [step] "+s"(s_step), [ringv] "+v"(ringv), [cnt] "+s"(s_cnt)
        : [n] "s"(s_n), [sh] "s"(s_sh), [G] "s"(s_G), [Gm1] "s"(s_Gm1), [cstb] "s"(s_cst), [eidb] "s"(s_eid), [pb] "s"(s_p), [nem1] "s"(s_nem1),
          [rowoff] "v"(rowoff), [isend] "v"(isend), [lane] "v"(lane)
        : "vcc", "scc", "memory",
          "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81",
          "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60",
          "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77");
#undef ORIP_NN_Q
#undef ORIP_NN_FETCH
#undef ORIP_NN_KEY
#undef ORIP_NN_MERGE
#undef ORIP_NN_MIN6
    cx = s_cx; cy = s_cy; step = s_step; dbg_cnt = s_cnt;
    return ev;
}

// The same search with a step written for the way a lone wave executes (one instruction per ~4.5 cycles, +16..20 cycles whenever the scalar
// unit consumes a value produced by a vector instruction, every exec-mask juggle of divergent control flow a handful of both):
// k_greedy_nn_grid's step compiles to ~350 instructions with divergent loops around uniform values = 1.1 us per step.  Here
//   * everything that is the same in all lanes (cursor, window, cell ranges, winner) is kept in SGPRs explicitly (v_readfirstlane);
//   * the candidates of the 3x3 window are evaluated without branches: every lane maps its ordinal to an entry with selects, entries
//     beyond the end take the pattern 0xffffffff; a candidate is a 2-byte entry + one 8-byte LDS read (both end points packed);
//   * the minimum runs over the 32-bit float pattern of the squared distance (6 DPP steps); the index tie-break of the reference (first
//     polyline in list order wins) only runs when two lanes hold the same pattern; the winner's end points come out of the winning lane's
//     registers (v_readlane), not from another LDS round trip;
//   * "no unscanned cell can be nearer" is an integer test: floor(d2) + 1 <= gap^2 - gap^2 / 2^18 - 1 (the three float roundings of a
//     squared distance stay below 2^-22 relative): conservative, so at worst one more round is scanned, never a wrong winner;
//   * results leave through a VGPR (one lane per step, 64 at a time).
__global__ __launch_bounds__(64) void k_greedy_nn_fast(const NNEnds* __restrict__ ends, int n, const int* __restrict__ sel, int skip_if, int need_any, int rule07, int G,
                                                        int32_t* __restrict__ order, uint8_t* __restrict__ flips, int no_asm, unsigned long long* __restrict__ dbg) {
    ORIP_NN_GATE(sel, skip_if, need_any)
    extern __shared__ __align__(16) unsigned char smem[];
    uint2* P = reinterpret_cast<uint2*>(smem);                                     // .x = sx | sy << 16, .y = ex | ey << 16; bit 15 of sx: used, bit 15 of sy: closed (rule07)
    unsigned* cst = reinterpret_cast<unsigned*>(P + n);                            // cst[0] = 0, cst[c + 1] = end of cell c
    uint16_t* Eid = reinterpret_cast<uint16_t*>(cst + (G * G + 2));                // entries sorted by cell: idx << 1 | end
    const int lane = threadIdx.x;
#define NNU(x) __builtin_amdgcn_readfirstlane((int)(x))
    int mnx = 0x7fffffff, mny = 0x7fffffff, mxx = -0x7fffffff, mxy = -0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        NNEnds e = ends[i];
        mnx = min(mnx, min(e.sx, e.ex)); mxx = max(mxx, max(e.sx, e.ex)); mny = min(mny, min(e.sy, e.ey)); mxy = max(mxy, max(e.sy, e.ey));
    }
    for (int o = 32; o > 0; o >>= 1) { mnx = min(mnx, __shfl_xor(mnx, o, 64)); mny = min(mny, __shfl_xor(mny, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64)); }
    const int ox = mnx, oy = mny;
    for (int i = lane; i < n; i += 64) {
        NNEnds e = ends[i];
        P[i] = make_uint2((unsigned)((e.sx - ox) | (i == seed ? 0x8000 : 0)) | ((unsigned)((e.sy - oy) | ((rule07 && e.closed) ? 0x8000 : 0)) << 16),
                          (unsigned)(e.ex - ox) | ((unsigned)(e.ey - oy) << 16));
    }
    unsigned* cnt = cst + 1;                                                       // counts, then cell ends
    for (int i = lane; i <= G * G + 1; i += 64) cst[i] = 0;
    __syncthreads();
    int sh = 0; while (((max(mxx - mnx, mxy - mny)) >> sh) >= G) sh++;
    sh = NNU(sh);
    for (int i = lane; i < n; i += 64) {
        const uint2 e = P[i];
        atomicAdd(&cnt[(((e.x >> 16) & 0x7fff) >> sh) * G + ((e.x & 0x7fff) >> sh)], 1u);
        if (!(e.x & 0x80000000u)) atomicAdd(&cnt[((e.y >> 16) >> sh) * G + ((e.y & 0xffff) >> sh)], 1u);
    }
    __syncthreads();
    {
        const int per = (G * G + 63) / 64, c0 = lane * per, c1 = min(G * G, c0 + per);
        unsigned sm = 0;
        for (int cc = c0; cc < c1; cc++) sm += cnt[cc];
        unsigned inc = sm;
        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        unsigned run = inc - sm;
        for (int cc = c0; cc < c1; cc++) { unsigned v = cnt[cc]; cnt[cc] = run; run += v; }     // starts for now
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {         // scatter; cnt[c] ends up as the END of cell c, so start(c) = cst[c], end(c) = cst[c + 1]
        const uint2 e = P[i];
        Eid[atomicAdd(&cnt[(((e.x >> 16) & 0x7fff) >> sh) * G + ((e.x & 0x7fff) >> sh)], 1u)] = (uint16_t)(i << 1);
        if (!(e.x & 0x80000000u)) Eid[atomicAdd(&cnt[((e.y >> 16) >> sh) * G + ((e.y & 0xffff) >> sh)], 1u)] = (uint16_t)((i << 1) | 1);
    }
    __syncthreads();
    const unsigned n_ent = (unsigned)NNU(cst[G * G]);
    int cx, cy;
    { const uint2 e = P[seed]; const bool cl = (e.x & 0x80000000u) != 0; cx = NNU(cl ? (e.x & 0x7fff) : (e.y & 0xffff)); cy = NNU(cl ? ((e.x >> 16) & 0x7fff) : (e.y >> 16)); }
    unsigned ringv = lane == 0 ? (unsigned)(seed << 1) : 0u;                      // lane (step & 63): index << 1 | flip of that step
    const int Gm1 = G - 1;
    const bool use_asm = G >= 4 && !no_asm;
    const unsigned lds_p = (unsigned)(uintptr_t)P, lds_cst = (unsigned)(uintptr_t)cst, lds_eid = (unsigned)(uintptr_t)Eid;
    const int rowoff = lane < 6 ? (lane >> 1) : 0, isend = lane < 6 ? (lane & 1) : 0;
    int step = 1, r_first = 1;
    unsigned long long d_fb = 0, d_calls = 0, t_asm = 0, t_gen = 0;      // ORIP_NN_DBG2: steps taken by the compiled code, asm entries, cycles in either
    while (step < n) {
        if (use_asm) {
            const unsigned long long t_0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
            int cnt = 0;
            const int ev = nn_asm_steps(cx, cy, step, ringv, n, sh, G, lds_p, lds_cst, lds_eid, n_ent - 1u, rowoff, isend, lane, cnt);
            if (dbg) { t_asm += __builtin_amdgcn_s_memtime() - t_0; d_calls++; if (lane == 0) { if (ev >= 2) dbg[2 + ev]++; dbg[4] += (unsigned)cnt & 0x3ffu; dbg[6] += ((unsigned)cnt >> 10) & 0x3ffu; dbg[9] += (unsigned)cnt >> 20; } }
            if (ev == 1) { order[step - 64 + lane] = (int32_t)(ringv >> 1); flips[step - 64 + lane] = (uint8_t)(ringv & 1u); continue; }
            if (ev == 0) break;
            r_first = 3;                          // the 3x3 window has just been found wanting (empty, all used, or the nearest lies beyond the gap): the next one
        }
        // ---- one step with the compiled code: whatever the loop above does not take
        const unsigned long long t_g0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        d_fb++;
        const int gx = cx >> sh, gy = cy >> sh;
        const float fx = (float)cx, fy = (float)cy;
        unsigned wi = 0, w0 = 0, w1 = 0;
        for (int r = r_first;; r = 2 * r + 1) {
            const int x0 = max(0, gx - r), x1 = min(Gm1, gx + r), y0 = max(0, gy - r), y1 = min(Gm1, gy + r);
            unsigned myk = ~0u, myi = 0x7fffffffu, my0 = 0, my1 = 0;
            // entry `q` (clamped into the table) as a candidate; valid == false: counts as infinitely far
            auto consider = [&](unsigned q, bool valid) {
                const unsigned id = Eid[q < n_ent ? q : n_ent - 1u]; const unsigned i = id >> 1;
                const uint2 e = P[i];
                const unsigned xy = (id & 1u) ? e.y : (e.x & 0x7fff7fffu);
                const float dx = __fsub_rn((float)(xy & 0xffffu), fx), dy = __fsub_rn((float)(xy >> 16), fy);
                unsigned k = __float_as_uint(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                k = (valid && !(e.x & 0x8000u)) ? k : ~0u;                           // used polylines (the previous one among them) do not count
                const unsigned long long key = ((unsigned long long)k << 32) | i, mine = ((unsigned long long)myk << 32) | myi;
                const bool better = key < mine;
                myk = better ? k : myk; myi = better ? i : myi; my0 = better ? e.x : my0; my1 = better ? e.y : my1;
            };
            if (y1 - y0 <= 2) {
                // lanes 0..5: start / end of the entry range of the (up to) three rows
                const int row = y0 + (lane >> 1);
                const int ci = row * G + ((lane & 1) ? x1 + 1 : x0);
                const unsigned bnd = (lane < 6 && row <= y1) ? cst[ci] : 0u;
                const unsigned lo0 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 0), n0 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 1) - lo0;
                const unsigned lo1 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 2), n1 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 3) - lo1;
                const unsigned lo2 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 4), n2 = (unsigned)__builtin_amdgcn_readlane((int)bnd, 5) - lo2;
                const unsigned n01 = n0 + n1, total = n01 + n2;
                for (unsigned t0 = 0; t0 < total; t0 += 64) {                        // uniform trip count, no exec masking
                    const unsigned t = t0 + lane;
                    unsigned q = lo0 + t;
                    q = t >= n0 ? lo1 + (t - n0) : q;
                    q = t >= n01 ? lo2 + (t - n01) : q;
                    consider(q, t < total);
                }
            } else {
                for (int row = y0; row <= y1; row++) {
                    const unsigned lo = (unsigned)NNU(cst[row * G + x0]), hi = (unsigned)NNU(cst[row * G + x1 + 1]);
                    for (unsigned q0 = lo; q0 < hi; q0 += 64) consider(q0 + lane, q0 + lane < hi);
                }
            }
            // minimum distance pattern over the wave
            unsigned m = myk;
#define ORIP_DPP_MINU(ctrl, rmask) { const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)m, ctrl, rmask, 0xf, false); m = t_ < m ? t_ : m; }
            ORIP_DPP_MINU(0x111, 0xf) ORIP_DPP_MINU(0x112, 0xf) ORIP_DPP_MINU(0x114, 0xf) ORIP_DPP_MINU(0x118, 0xf) ORIP_DPP_MINU(0x142, 0xa) ORIP_DPP_MINU(0x143, 0xc)
#undef ORIP_DPP_MINU
            const unsigned mink = (unsigned)__builtin_amdgcn_readlane((int)m, 63);
            bool final_ = x0 == 0 && y0 == 0 && x1 == Gm1 && y1 == Gm1;            // everything scanned
            if (!final_ && mink != ~0u) {
                int gap;                                                             // distance to the nearest unscanned cell, over the open sides
                if (r == 1) {
                    // the 3x3 window (94 % of the rounds): its sides lie one cell beyond the cursor's cell, so the gap is a function of the cursor's
                    // position inside its cell.  A side on the border of the grid counts as open here -- a smaller gap is conservative.
                    const int cell = 1 << sh, lx = cx & (cell - 1), ly = cy & (cell - 1);
                    gap = cell + min(min(lx + 1, cell - lx), min(ly + 1, cell - ly));
                } else {
                    gap = 0x7fff;
                    if (x0 > 0) gap = min(gap, cx - (x0 << sh) + 1);
                    if (x1 < Gm1) gap = min(gap, ((x1 + 1) << sh) - cx);
                    if (y0 > 0) gap = min(gap, cy - (y0 << sh) + 1);
                    if (y1 < Gm1) gap = min(gap, ((y1 + 1) << sh) - cy);
                }
                const unsigned g2 = (unsigned)(gap * gap);                           // gap < 2^15: exact
                const unsigned bfl = (unsigned)NNU((unsigned)__uint_as_float(mink)); // floor of the best squared distance (< 2^31)
                final_ = bfl + 1u <= g2 - (g2 >> 18) - 1u && g2 > 1u;
            }
            if (final_) {
                unsigned long long tie = __ballot(myk == mink);
                if (tie & (tie - 1)) {                                               // several lanes hold this distance: the smallest index wins
                    unsigned ci2 = myk == mink ? myi : 0x7fffffffu;
                    for (int o = 32; o > 0; o >>= 1) { const unsigned t_ = (unsigned)__shfl_xor((int)ci2, o, 64); ci2 = t_ < ci2 ? t_ : ci2; }
                    tie = __ballot(myk == mink && myi == ci2);
                }
                // every lane settles the reading direction and the next cursor of ITS candidate (a dozen vector instructions); the winner's come
                // out with two v_readlane -- instead of three, followed by the same arithmetic on scalars that wait for them
                const unsigned sxy = my0 & 0x7fff7fffu;
                const float ds = nn_d2((int)(sxy & 0xffffu), (int)(sxy >> 16), cx, cy), de = nn_d2((int)(my1 & 0xffffu), (int)(my1 >> 16), cx, cy);
                const bool cl = (my0 & 0x80000000u) != 0;
                const bool flip = cl ? false : !(ds <= de);
                const unsigned ncur = (cl || flip) ? sxy : my1;
                const unsigned pack = (myi << 1) | (flip ? 1u : 0u);
                const int win_lane = __ffsll((long long)tie) - 1;
                wi = (unsigned)__builtin_amdgcn_readlane((int)pack, win_lane);
                w1 = (unsigned)__builtin_amdgcn_readlane((int)ncur, win_lane);
                w0 = (unsigned)__builtin_amdgcn_readlane((int)my0, win_lane);
                break;
            }
        }
        P[wi >> 1].x = w0 | 0x8000u;                                 // the used flag, through the type the entries are read as (every lane writes the same word)
        ringv = lane == (step & 63) ? wi : ringv;
        if ((step & 63) == 63) { order[step - 63 + lane] = (int32_t)(ringv >> 1); flips[step - 63 + lane] = (uint8_t)(ringv & 1u); }
        cx = (int)(w1 & 0xffffu); cy = (int)(w1 >> 16);
        step++;
        if (dbg) t_gen += __builtin_amdgcn_s_memtime() - t_g0;
    }
    if (dbg && lane == 0) { dbg[0] = d_fb; dbg[1] = d_calls; dbg[2] = t_asm; dbg[3] = t_gen; }
    { const int done = n & ~63; if (done + lane < n) { order[done + lane] = (int32_t)(ringv >> 1); flips[done + lane] = (uint8_t)(ringv & 1u); } }
#undef NNU
}

// Greedy reorder of a whole DPolys list into dst.  kind: 7 -> 07 rules (arcLength closed seed), 8 -> 08 (_poly_perimeter seed), 10 -> 10 (arcLength open seed)
// under_greedy (optional): called right after the greedy kernel has been enqueued, with the features of the source list (device
// array; complete once the lane's event ev2 has fired -- the host no longer waits in front of the chain, so the hook's stream must).  The chain of greedy steps keeps ONE wave busy for milliseconds, so work that does not
// depend on the order can be issued to the lane's side stream from there (stage 08's front: vector08.hip: prefetch08).
typedef int (*ReorderHook)(orip_ctx* c, void* arg, DPolys& src, const PolyFeat* feat);
static int vreorder(orip_ctx* c, DPolys& src, DPolys& dst, int kind, ReorderHook under_greedy = nullptr, void* hook_arg = nullptr) {
    int64_t n = src.n;
    if (n == 0) { dst.n = 0; dst.total = 0; dst.set_explicit(); HIPC(c, dst.off.ensure(64)); HIPC(c, hipMemsetAsync(dst.off.p, 0, 8, LN(c).stream)); return 0; }
    if (n > 0x7fffffff) ORIP_FAIL(c, "too many polylines");
    HIPC(c, LN(c).vtmp[6].ensure((size_t)n * (sizeof(PolyFeat) + sizeof(NNEnds) + sizeof(GatherDesc) + 4 + 2) + 256));
    PolyFeat* feat = LN(c).vtmp[6].as<PolyFeat>();
    NNEnds* ends = (NNEnds*)(feat + n);
    GatherDesc* desc = (GatherDesc*)(ends + n);
    int32_t* order = (int32_t*)(desc + n);
    uint8_t* flips = (uint8_t*)(order + n); uint8_t* used = flips + n;
    int what = kind == 7 ? 4 : (kind == 8 ? 1 : 8);
    if (kind == 7 && is_coded(src) && src.vident && !getenv("ORIP_ARC_POINTS")) {      // whole walks: the long contours' arc lengths from the walk records
        VSrc vs_; ORIP_TRY(vsrc_of(c, src, vs_));
        hipLaunchKernelGGL(k_poly_features<VSrc>, dim3(cdiv(n, 128)), dim3(128), 0, LN(c).stream, vs_, n, what, feat, (float*)nullptr);
        if (src.total > ORIP_LONG_POLY) { ProfScope ps(c, "k_walk_arcs"); hipLaunchKernelGGL(k_walk_arcs, dim3((unsigned)std::min<int64_t>(n, 16384)), dim3(64), 0, LN(c).stream, vs_, n, feat); }
        HIPC(c, hipGetLastError());
    } else ORIP_TRY(vfeatures(c, src, what, feat));
    ORIP_WITH_SRC(c, src, ps, { hipLaunchKernelGGL(k_ends_from_feat<decltype(ps)>, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, feat, n, kind == 7 ? 1 : 0, ps, ends); });
    int* d_seed = LN(c).flags.as<int>() + 32;
    hipLaunchKernelGGL(k_argmax_feat, dim3(1), dim3(256), 0, LN(c).stream, feat, (int)n, kind == 8 ? 0 : 1, d_seed, ends);      // seed and coordinate-range flags in one pass
    const size_t lds = (size_t)n * 9 + 16;
    // grid side: as fine as LDS allows (cells are powers of two, so twice the side is four times fewer candidates per window),
    // but not many more cells than polylines
    int G = 8; while (G < 128 && (size_t)(G + 8) * (G + 8) <= 4 * (size_t)n && (size_t)n * 12 + (size_t)((G + 8) * (G + 8) + 1) * 4 + 64 <= 158 * 1024) G += 8;
    const size_t lds_grid = (size_t)n * 12 + (size_t)(G * G + 1) * 4 + 64;        // (+4 for k_greedy_nn_fast: inside the 64 spare bytes of the 158 KB check)
    static std::once_flag attr_once;                // several layer threads may arrive here together
    static std::atomic<int> attr_err{0};
    std::call_once(attr_once, [] {
        orip_max_lds(k_greedy_nn_lds, 150 * 1024, attr_err);
#ifdef ORIP_VARIANTS
        orip_max_lds(k_greedy_nn_grid, 158 * 1024, attr_err);
#endif
        orip_max_lds(k_greedy_nn_fast, 158 * 1024, attr_err);
    });
    if (attr_err.load()) ORIP_FAIL(c, "hipFuncSetAttribute(greedy kernels) failed: %s", hipGetErrorString((hipError_t)attr_err.load()));
    // Seed and coordinate-range flags stay on the device: every kernel that may have to run is enqueued and picks itself from the flags
    // (bit 0: a coordinate beyond int16 -> the global-memory kernel; bit 1: beyond 15 bits -> no grid).  No host round trip in front of the chain.
    if (under_greedy) HIPC(c, hipEventRecord(LN(c).ev2, LN(c).stream));      // everything the hook's side-stream work reads (features, ends) is complete at this point of the stream
    const bool grid_ok = n >= 64 && n <= 16000 && lds_grid <= 158 * 1024 && !getenv("ORIP_NN_NOGRID");
    const bool lds_ok = n <= 16000;
    const int r07 = kind == 7 ? 1 : 0;
    {
        ProfScope ps(c, "k_greedy_nn");
        if (grid_ok) {
#ifdef ORIP_VARIANTS
            unsigned long long* dbg = ORIP_VARIANT("ORIP_NN_DBG") ? LN(c).flags.as<unsigned long long>() + 64 : nullptr;
            if (dbg || ORIP_VARIANT("ORIP_NN_OLDGRID")) {
                hipLaunchKernelGGL(k_greedy_nn_grid, dim3(1), dim3(64), lds_grid, LN(c).stream, ends, (int)n, d_seed, 3, 0, r07, G, order, flips, dbg);
                if (dbg) { unsigned long long h[4]; hipStreamSynchronize(LN(c).stream); hipMemcpy(h, dbg, 32, hipMemcpyDeviceToHost); fprintf(stderr, "[nn dbg] kind %d n %lld G %d cell %llu: rounds %llu scanned %llu full %llu\n", kind, (long long)n, G, h[3], h[0], h[1], h[2]); }
            } else
#endif
            {
                unsigned long long* dbg2 = getenv("ORIP_NN_DBG2") ? LN(c).flags.as<unsigned long long>() + 64 : nullptr;
                if (dbg2) hipMemsetAsync(dbg2, 0, 80, LN(c).stream);
                hipLaunchKernelGGL(k_greedy_nn_fast, dim3(1), dim3(64), lds_grid + 4, LN(c).stream, ends, (int)n, d_seed, 3, 0, r07, G, order, flips, getenv("ORIP_NN_NOASM") ? 1 : 0, dbg2);
                if (dbg2) { unsigned long long h[10]; hipStreamSynchronize(LN(c).stream); hipMemcpy(h, dbg2, 80, hipMemcpyDeviceToHost); fprintf(stderr, "[nn dbg2] kind %d n %lld G %d: %llu steps by the compiled code (empty %llu, all used %llu, gap %llu; asm steps from cached candidates: one per lane %llu, two per lane %llu; with more than 128 candidates %llu), %llu asm entries, cycles asm %llu compiled %llu\n", kind, (long long)n, G, h[0], h[5], h[7], h[8], h[4], h[6], h[9], h[1], h[2], h[3]); hipMemsetAsync(dbg2, 0, 80, LN(c).stream); }
            }
        }
        // Behind the grid kernel only ONE more launch, and a light one (256 threads, no dynamic LDS): a kernel that merely checks its flag and
        // returns still waits for a CU with room for its whole workgroup -- 0.5 ms for 1024 threads or 150 KB of LDS next to the other layers' work.
        // The grid kernel bows out for coordinates beyond 15 bits only (never on a canvas below 32768 px): the global-memory kernel takes those.
        if (lds_ok && !grid_ok) hipLaunchKernelGGL(k_greedy_nn_lds, dim3(1), dim3(1024), lds, LN(c).stream, ends, (int)n, d_seed, 1, 0, r07, order, flips);
        hipLaunchKernelGGL(k_greedy_nn, dim3(1), dim3(lds_ok ? 256 : 1024), 0, LN(c).stream, ends, (int)n, d_seed, 0, grid_ok ? 3 : (lds_ok ? 1 : 0), r07, used, order, flips);
    }
    if (under_greedy) ORIP_TRY(under_greedy(c, hook_arg, src, feat));
    hipLaunchKernelGGL(k_desc_from_order, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, src.off.as<int64_t>(), order, flips, n, 0, feat, desc);
    HIPC(c, hipGetLastError());
    return vgather_list(c, desc, n, src, dst, src.total);      // every polyline of the source, whole: the same number of points
}

}  // namespace
