// csrc/vector08.hip -- stage 08 (08_dedup_layer_basic.py process_layer, 08:484-557) on gfx950.
//
// Stage A (greedy virtual draw, 08:117-183) is NOT sequential on the GPU.  In the reference every sample of every
// polyline is pushed to the tail and later popped (hash add + thick-line stamp) whether or not it was accepted, so the
// stamp sequence depends only on the resampled geometry and on the processing order (perimeter, descending).  Giving
// every popped sample its global sequence number g, sample (r, j) sees exactly the stamps with g < base[r] + npop(r, j).
// The canvas therefore stores, per pixel, the SMALLEST sequence number of any capsule covering it (atomicMin), and all
// samples of all polylines of the layer are tested in parallel.  Self-collision (_PointHash, 08:68-99) is a sorted
// (polyline, cell) bucket list scanned in pop order.
// Stage B (_post_skeleton_merge, 08:376-469) runs all clusters at once on one padded canvas: clusters are >= 76 px apart
// in one axis, so per-ROI rasterise / thin / label equals whole-canvas rasterise / thin / label (DESIGN.md "stage 08-B").
#include "vec_common.h"
#include <chrono>
#include <string>
#include <type_traits>
#define PAD8 64

int orip_runs_to_polys(orip_ctx* c, const int2* spt, const uint8_t* sflag, unsigned n_slots, DPolys& dst);

namespace {

// ================================================================= A0 / A7: _split_small_and_taps (08:198-216)
template <class Src>
__global__ __launch_bounds__(128) void k_split_small08(Src src, int64_t n_polys, orip_params08 P, const PolyFeat* __restrict__ feat,
                                                        unsigned* __restrict__ is_tap, unsigned* __restrict__ is_keep, int2* __restrict__ tap_xy, GatherDesc* __restrict__ kd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_polys) return;
    if (i == n_polys) { is_tap[i] = 0; is_keep[i] = 0; return; }
    const int64_t n = src.len(i);
    unsigned tap = 0, keep = 0;
    GatherDesc g; g.begin = src.off[i]; g.len = n; g.rev = 0; g.src = (int32_t)i;
    if (n >= 2) {
        const int32_t x0 = feat[i].x0, x1 = feat[i].x1, y0 = feat[i].y0, y1 = feat[i].y1;      // bbox from vfeatures (long polylines: block-parallel)
        double d = (double)max(x1 - x0, y1 - y0);
        if (d <= P.tap_diam && d <= P.tap_max_dim && n <= (int64_t)P.tap_max_v) {      // the vertex test is evaluated last in the reference but decides alone
            double per; float cx, cy, r;
            if constexpr (std::is_same<Src, ESrc>::value) {
                const int32_t* p = reinterpret_cast<const int32_t*>(src.pts + src.off[i]);
                per = (double)vs::pairwise_seglen_sum<0>(p, n);
                if (per <= P.tap_max_per) vs::min_enclosing_circle(p, n, cx, cy, r);
            } else {                                  // a tap candidate has at most tap_max_v <= 64 vertices (checked by the host): private copy
                auto cu = src.cur(i);
                LocalPts<decltype(cu), 64> lp; lp.load(cu, (int)n);
                per = (double)vs::pairwise_seglen_sum<0>(lp.xy, n);
                if (per <= P.tap_max_per) vs::min_enclosing_circle(lp.xy, n, cx, cy, r);
            }
            if (per <= P.tap_max_per) { tap = 1; tap_xy[i] = make_int2((int)vs::round_half_even((double)cx), (int)vs::round_half_even((double)cy)); }
        }
        if (!tap && !(d < P.min_keep)) {
            keep = 1;
            if (feat[i].closed) g.len = n - 1;      // _ensure_open
        }
    }
    is_tap[i] = tap; is_keep[i] = keep; kd[i] = g;
}
__global__ __launch_bounds__(256) void k_compact_desc(const unsigned* __restrict__ flag, const unsigned* __restrict__ scan, int64_t n, const GatherDesc* __restrict__ in,
                                                       GatherDesc* __restrict__ out, const int2* __restrict__ tap_xy, int2* __restrict__ taps_out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    if (out) out[scan[i]] = in[i];
    if (taps_out) taps_out[scan[i]] = tap_xy[i];
}

// ================================================================= A2: resample (08:53-64)
// (preprocessor conditionals cannot sit inside ORIP_WITH_SRC's macro argument: the variants build's launch of the serial-chain form is a macro of its own)
#ifdef ORIP_VARIANTS
#define ORIP_CUM_CHAIN_LAUNCH(SRC_T) if (ORIP_VARIANT("ORIP_CUM_CHAIN")) { chain = true; hipLaunchKernelGGL((k_cumlen_long2<SRC_T, true>), dim3((unsigned)std::min<int64_t>(nk, 8192), 1), dim3(64), 0, LN(c).stream, sv, nk, step, cum, (int64_t)0, info, ord, 0, (const float*)nullptr); }
#else
#define ORIP_CUM_CHAIN_LAUNCH(SRC_T)
#endif
#define ORIP_LONG_CUM 128      // polylines above this many points get a wavefront for their cumulative lengths (k_cumlen_long2)
struct RsInfo { int64_t n_eff; double total; unsigned m; unsigned pass; };
// sequential float32 cumsum per polyline (np.cumsum): one lane per short polyline; long polylines (k_cumlen_long2) use one
// wavefront: 64 segment lengths are computed / loaded by the lanes, the strictly sequential chain of float adds then runs
// over them with v_readlane (the order of additions, and therefore every rounding, is the reference's)
__device__ __forceinline__ void rs_finish(RsInfo& r, float acc, int64_t n, double step) {
    r.total = (double)acc;
    if (r.total <= step) { r.pass = 1; r.m = (unsigned)n; }
    else r.m = (unsigned)ceil(r.total / step);
    if (r.m < 2) r.m = 0;                                                             // len(S) < 2 -> nothing is drawn or stamped (08:130)
}
template <class Src>
__global__ __launch_bounds__(128) void k_cumlen(Src src, int64_t n_polys, double step, float* __restrict__ cum, RsInfo* __restrict__ info) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_polys) return;
    auto cu = src.cur(i); int64_t n = src.len(i);
    float* s = cum + src.off[i];
    RsInfo r; r.n_eff = n; r.total = 0; r.m = 0; r.pass = 0;
    const int2 pf = cu.at(0);
    auto same_as_first = [&](int64_t k) { const int2 q = cu.at(k); return q.x == pf.x && q.y == pf.y; };
    if (n >= 2 && same_as_first(n - 1)) n -= 1;         // _ensure_open inside _virtual_draw (08:127)
    r.n_eff = n;
    if (n >= 2) {
        if (n > 2 && same_as_first(n - 1)) n -= 1;    // _is_closed inside _resample_arclen (08:56)
        r.n_eff = n;
        if (n <= ORIP_LONG_CUM) {
            const CurPt<decltype(cu)> pt{cu};
            float acc = 0.f; s[0] = 0.f;
            for (int64_t k = 0; k + 1 < n; k++) { float sl = vs::seg_len_f32_p(pt, k); acc = (k == 0) ? sl : acc + sl; s[k + 1] = acc; }
            rs_finish(r, acc, n, step);
        }
    }
    info[i] = r;
}

// ---- float32 np.cumsum without the serial chain (r03).  While the running sum p stays inside one binade [2^e, 2^(e+1)) its ulp u is fixed and
// p is a multiple of u, so fl(p + d) = p + R(d) with R(d) = d rounded to a multiple of u: an INTEGER increment that does not depend on p -- except
// (i) when d lies exactly half-way between two multiples of u (round-half-even looks at p's last bit) and (ii) when the sum reaches 2^(e+1) (the ulp
// doubles).  A window of 64 lengths is therefore one integer wave scan; the first lane where (i) or (ii) happens does ONE real float add from its
// neighbour's exact sum, and the lanes behind it are scanned again in the new binade.  A polyline crosses a binade ~18 times and meets a tie only
// where the low bits of a length happen to be 10..0 at the current ulp; every other window costs one scan instead of 63 dependent adds.
// State: E = biased exponent of p (0: p == 0), M = 24-bit significand; both wave-uniform.  Lengths are finite and >= 0.
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
#define ORIP_DPP_ADD(ctrl, rowmask) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false);
    ORIP_DPP_ADD(0x111, 0xf) ORIP_DPP_ADD(0x112, 0xf) ORIP_DPP_ADD(0x114, 0xf) ORIP_DPP_ADD(0x118, 0xf)      // row_shr 1, 2, 4, 8
    ORIP_DPP_ADD(0x142, 0xa) ORIP_DPP_ADD(0x143, 0xc)                                                          // row_bcast 15, 31
#undef ORIP_DPP_ADD
    return v;
}
__device__ __forceinline__ unsigned wave_incl_scan_max_u32(unsigned v) {                  // running maximum over the lanes, the same six DPP steps
#define ORIP_DPP_MAX(ctrl, rowmask) { const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false); v = t_ > v ? t_ : v; }
    ORIP_DPP_MAX(0x111, 0xf) ORIP_DPP_MAX(0x112, 0xf) ORIP_DPP_MAX(0x114, 0xf) ORIP_DPP_MAX(0x118, 0xf)
    ORIP_DPP_MAX(0x142, 0xa) ORIP_DPP_MAX(0x143, 0xc)
#undef ORIP_DPP_MAX
    return v;
}
__device__ __forceinline__ float cum_window(float dval, int lane, unsigned& E, unsigned& M) {
    const unsigned b = __float_as_uint(dval);
    const unsigned Ed = b >> 23, Md = Ed ? ((b & 0x7fffffu) | 0x800000u) : 0u;
    unsigned out = 0u; int first = 0;                 // lanes below `first` are final
    for (;;) {
        const int sh = (int)E - (int)Ed;
        const unsigned sc = (unsigned)(sh < 0 ? 0 : (sh > 31 ? 31 : sh));
        const unsigned rem = Md & ((1u << sc) - 1u), half = (1u << sc) >> 1;
        const bool live = lane >= first;
        const bool ev = live && (sh < 0 || (sc > 0u && rem == half));                 // d >= 2p, or a tie at this ulp
        const unsigned r = (live && sh >= 0) ? (Md >> sc) + ((sc > 0u && rem > half) ? 1u : 0u) : 0u;
        const unsigned S = wave_incl_scan_u32(r);
        const unsigned long long em = __ballot(ev || (live && M + S >= 0x1000000u));
        const int f = em ? __builtin_ctzll(em) : 64;
        if (live && lane < f) out = (E << 23) | ((M + S) & 0x7fffffu);
        if (f == 64) { M += (unsigned)__builtin_amdgcn_readlane((int)S, 63); break; }
        const unsigned Mp = M + (f > first ? (unsigned)__builtin_amdgcn_readlane((int)S, f - 1) : 0u);
        const float pprev = __uint_as_float((E << 23) | (Mp & 0x7fffffu));             // E == 0: M == 0, p == +0
        const float pnew = __fadd_rn(pprev, __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)b, f)));
        const unsigned nb = __float_as_uint(pnew);
        if (lane == f) out = nb;
        E = nb >> 23; M = E ? ((nb & 0x7fffffu) | 0x800000u) : 0u;
        first = f + 1;
        if (first == 64) break;
    }
    return __uint_as_float(out);
}
__device__ __forceinline__ float cum_state_value(unsigned E, unsigned M) { return __uint_as_float((E << 23) | (M & 0x7fffffu)); }
// ---- both reading directions of every polyline in one launch (prefetch08): the reversed polyline has the same segment lengths in
// reverse order, and its float32 running sum is a second, independent serial chain -- two chains interleave in one wavefront for the
// price of one (a dependent add waits ~10 cycles for its predecessor anyway).  Forward = the polyline as split_small keeps it (opened
// when closed); reversed = all its points backwards (stage 07 never flips a closed contour, so closed ones get no reversed entry).
// cum / info of the reversed reading live `rev_off` floats / `n_polys` entries behind the forward ones.
template <class Src>
__global__ __launch_bounds__(128) void k_cumlen2(Src src, const PolyFeat* __restrict__ feat07, int64_t n_polys, double step, float* __restrict__ cum, int64_t rev_off, RsInfo* __restrict__ info) {
    // one thread per polyline AND reading direction (the first n_polys threads read forwards): the launch is a few dozen blocks whose time is the longest
    // thread's loop, so two loops in a row per thread were twice that
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= 2 * n_polys) return;
    const int64_t i = gi < n_polys ? gi : gi - n_polys; const int dir0 = gi < n_polys ? 0 : 1;
    auto cu = src.cur(i); const int64_t nfull = src.len(i);
    const bool closed = feat07[i].closed != 0;
    for (int dir = dir0; dir <= dir0; dir++) {
        int64_t n = (dir == 0 && closed && nfull > 0) ? nfull - 1 : nfull;         // the view: opened forward, whole reversed
        float* s = cum + (dir ? rev_off : 0) + src.off[i];
        auto P = [&](int64_t k) { return dir ? cu.at(nfull - 1 - k) : cu.at(k); };
        RsInfo r; r.n_eff = n; r.total = 0; r.m = 0; r.pass = 0;
        if (dir == 1 && closed) { r.n_eff = 0; info[n_polys + i] = r; continue; }
        const int2 pf = P(0);
        auto same_as_first = [&](int64_t k) { const int2 q = P(k); return q.x == pf.x && q.y == pf.y; };
        if (n >= 2 && same_as_first(n - 1)) n -= 1;         // _ensure_open inside _virtual_draw (08:127)
        r.n_eff = n;
        if (n >= 2) {
            if (n > 2 && same_as_first(n - 1)) n -= 1;    // _is_closed inside _resample_arclen (08:56)
            r.n_eff = n;
            if (n <= ORIP_LONG_CUM) {
                float acc = 0.f; s[0] = 0.f;
                int2 a = P(0);
                for (int64_t k = 0; k + 1 < n; k++) {
                    const int2 b = P(k + 1);
                    float dx = (float)b.x - (float)a.x, dy = (float)b.y - (float)a.y; float qx = dx * dx, qy = dy * dy; const float sl = sqrtf(qx + qy);
                    acc = (k == 0) ? sl : acc + sl; s[k + 1] = acc; a = b;
                }
                rs_finish(r, acc, n, step);
            }
        }
        info[dir ? n_polys + i : i] = r;
    }
}
// lane j <- lane j + 1 of v; lane 63 <- `last` (the successor of a window's last point is the first point of the next window)
__device__ __forceinline__ int2 lane_succ(const int2 v, const int2 last, int lane) {
    int2 r;
    r.x = __builtin_amdgcn_update_dpp(0, v.x, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    r.y = __builtin_amdgcn_update_dpp(0, v.y, 0x130, 0xf, 0xf, true);
    if (lane == 63) r = last;
    return r;
}
// One wavefront reads one long polyline in one direction: slot-th of n_slots waves of that direction, longest polylines first (ord).
// seg != nullptr (prefetch08): the float32 length of every segment is already there (k_seglen), so a reading costs 4 bytes per segment instead of
// turning (polyline, index) into a point again (~25 instructions; the launches are bound by instruction issue).
template <class Src, bool CHAIN>
__device__ __forceinline__ void cumlen_long_wave(const Src& src, int64_t n_polys, double step, float* __restrict__ cum, int64_t rev_off, RsInfo* __restrict__ info,
                                                 const unsigned* __restrict__ ord, const bool rev, const float* __restrict__ seg, int64_t slot, int64_t n_slots, const int lane) {
    for (int64_t rr = slot; rr < n_polys; rr += n_slots) {
        const int64_t i = ord[rr];
        RsInfo r = info[rev ? n_polys + i : i];
        if (r.n_eff <= ORIP_LONG_CUM) continue;
        auto cu = src.cur(i); const int64_t nfull = src.len(i);
        float* s = cum + (rev ? rev_off : 0) + src.off[i];
        const int64_t ns = r.n_eff - 1;                                             // segments; points 0 .. ns of this reading
        const float* sg = seg ? seg + src.off[i] : nullptr;                         // sg[k]: segment k of the FORWARD polyline, k < nfull - 1
        const bool from_seg = sg != nullptr;
        float acc = 0.f; unsigned cE = 0u, cM = 0u;
        if (lane == 0) s[0] = 0.f;
        // A turn is 4 windows of 64 segment lengths.  Every point is fetched ONCE: the far end of segment k is the point in the next lane, the far end of a
        // window's last segment the first point of the next window, of a turn's last segment one extra point.  The points (or stored lengths) of the
        // next turn are requested before this turn's sums run.
        auto P = [&](int64_t k) { return cu.at(rev ? nfull - 1 - k : k); };
        auto request = [&](int64_t base, int2 (&p)[5], float (&fl)[4]) {
            if (from_seg) {
#pragma unroll
                for (int w = 0; w < 4; w++) { const int64_t k = base + 64 * w + lane; fl[w] = k < ns ? sg[rev ? nfull - 2 - k : k] : 0.f; }
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++) { const int64_t k = base + 64 * w + lane; p[w] = k <= ns ? P(k) : make_int2(0, 0); }
                p[4] = base + 256 <= ns ? P(base + 256) : make_int2(0, 0);
            }
        };
        auto lengths = [&](int64_t base, const int2 (&p)[5], const float (&fl)[4], float (&sl)[4]) {
            if (from_seg) {
#pragma unroll
                for (int w = 0; w < 4; w++) sl[w] = fl[w];
                return;
            }
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int64_t k = base + 64 * w + lane;
                const int2 nx0 = w < 3 ? make_int2(__builtin_amdgcn_readlane(p[w + 1].x, 0), __builtin_amdgcn_readlane(p[w + 1].y, 0)) : p[4];
                const int2 b2 = lane_succ(p[w], nx0, lane);
                float dx = (float)b2.x - (float)p[w].x, dy = (float)b2.y - (float)p[w].y; float qx = dx * dx, qy = dy * dy;
                const float L = sqrtf(qx + qy);                                        // seg_len_f32
                sl[w] = k < ns ? L : 0.f;                                              // beyond the last segment of this reading: +0
            }
        };
        int2 rp[5]; float rf[4] = {0.f, 0.f, 0.f, 0.f}; float cur[4];
        request(0, rp, rf); lengths(0, rp, rf, cur);
        for (int64_t base = 0; base < ns; base += 256) {
            request(base + 256, rp, rf);
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int64_t k = base + 64 * w + lane;
                float pv;
                if (CHAIN) {                                          // the strictly sequential float sums as 63 wave-shifted adds
                    float dv = (lane == 0) ? __fadd_rn(acc, cur[w]) : cur[w];
                    pv = dv;
#pragma unroll
                    for (int j = 1; j < 64; j++)
                        pv = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pv), 0x138 /* wave_shr:1 */, 0xf, 0xf, true)), dv);
                    acc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv), 63));
                } else { pv = cum_window(cur[w], lane, cE, cM); acc = cum_state_value(cE, cM); }
                if (k < ns) s[k + 1] = pv;
            }
            lengths(base + 256, rp, rf, cur);
        }
        if (lane == 0) { rs_finish(r, acc, r.n_eff, step); info[rev ? n_polys + i : i] = r; }
    }
}
template <class Src, bool CHAIN>
__global__ __launch_bounds__(64) void k_cumlen_long2(Src src, int64_t n_polys, double step, float* __restrict__ cum, int64_t rev_off, RsInfo* __restrict__ info, const unsigned* __restrict__ ord,
                                                     int dir0, const float* __restrict__ seg) {
    cumlen_long_wave<Src, CHAIN>(src, n_polys, step, cum, rev_off, info, ord, ((blockIdx.y + (unsigned)dir0) & 1u) != 0, seg, blockIdx.x, gridDim.x, threadIdx.x);
}
// prefetch08: float32 length of EVERY segment of the long polylines (seg[off[i] + k] = |P(k + 1) - P(k)|, k < len(i) - 1) and the bounding box of their open
// views (points [0, bb[i].n); bb[i] holds the first point's box on entry: k_poly_features), fully parallel: a wave takes 64 windows of 64 consecutive points
// of the FLAT point list, advancing by 63, so the far end of a lane's segment is the point in the next lane and every lane's cursor stays on consecutive
// points of (mostly) one polyline.  Both readings' cumulative lengths then run side by side from these lengths (one launch) instead of
// the reversed reading behind the forward one, and so do the perimeter leaves (k_perim_leaves_seg).
template <class Src>
__global__ __launch_bounds__(256) void k_seglen(Src src, int64_t n_polys, int64_t total, float* __restrict__ seg, PolyFeat* __restrict__ bb) {
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (63 * 64);
    if (base >= total) return;                                 // (the whole wave)
    int64_t g = base + lane;
    int64_t i = 0;
    { const int64_t gg = g < total ? g : total - 1; int64_t hi = n_polys - 1;        // polyline of the lane's first point: the last i with off[i] <= g
      while (i < hi) { const int64_t mid = (i + hi + 1) >> 1; if (src.off[mid] <= gg) i = mid; else hi = mid - 1; } }
    int64_t o0 = src.off[i], o1 = src.off[i + 1];
    auto cu = src.cur(i);
    bool is_long = o1 - o0 > ORIP_LONG_CUM;
    int64_t vn = is_long ? bb[i].n : 0; if (vn <= ORIP_LONG_POLY) vn = 0;             // box wanted for points [0, vn) of this polyline
    int bx0 = 0x7fffffff, bx1 = -0x7fffffff, by0 = 0x7fffffff, by1 = -0x7fffffff; bool has = false;
    for (int t = 0; t < 64 && base + 63 * t < total; t++, g += 63) {
        const bool valid = g < total;
        if (valid && g >= o1) {                                // the lane enters another polyline (rare: the long ones hold thousands of points)
            if (has) { atomicMin(&bb[i].x0, bx0); atomicMax(&bb[i].x1, bx1); atomicMin(&bb[i].y0, by0); atomicMax(&bb[i].y1, by1); }
            bx0 = by0 = 0x7fffffff; bx1 = by1 = -0x7fffffff; has = false;
            do { i++; o0 = o1; o1 = src.off[i + 1]; } while (g >= o1);
            cu = src.cur(i); is_long = o1 - o0 > ORIP_LONG_CUM;
            vn = is_long ? bb[i].n : 0; if (vn <= ORIP_LONG_POLY) vn = 0;
        }
        const bool on = valid && is_long;
        int2 p = make_int2(0, 0);
        if (on) p = cu.at(g - o0);
        const int2 q = lane_succ(p, make_int2(0, 0), lane);
        if (on && lane < 63) {
            if (g + 1 < o1) { float dx = (float)q.x - (float)p.x, dy = (float)q.y - (float)p.y; float qx = dx * dx, qy = dy * dy; seg[g] = sqrtf(qx + qy); }     // seg_len_f32
            if (g - o0 < vn) { bx0 = min(bx0, p.x); bx1 = max(bx1, p.x); by0 = min(by0, p.y); by1 = max(by1, p.y); has = true; }
        }
    }
    if (__all(i == __shfl(i, 0, 64))) {                        // the usual case: one polyline under the whole wave at the end
        for (int o = 32; o > 0; o >>= 1) { bx0 = min(bx0, __shfl_xor(bx0, o, 64)); bx1 = max(bx1, __shfl_xor(bx1, o, 64)); by0 = min(by0, __shfl_xor(by0, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64)); }
        if (lane == 0 && bx0 <= bx1) { atomicMin(&bb[i].x0, bx0); atomicMax(&bb[i].x1, bx1); atomicMin(&bb[i].y0, by0); atomicMax(&bb[i].y1, by1); }
    } else if (has) { atomicMin(&bb[i].x0, bx0); atomicMax(&bb[i].x1, bx1); atomicMin(&bb[i].y0, by0); atomicMax(&bb[i].y1, by1); }
}
// any_out: set when a sampled polyline reaches beyond the canvas (its samples lie inside the box of its points): only then can a sample be
// off-canvas, and only then does "the previous in-canvas sample" (k_capprev) differ from "the previous sample"
__global__ __launch_bounds__(256) void k_rank_counts(const RsInfo* __restrict__ info, const unsigned* __restrict__ ord, int64_t n, unsigned* __restrict__ mr,
                                                      const PolyFeat* __restrict__ feat, int W, int H, unsigned* __restrict__ any_out) {
    int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        const unsigned i = ord[r]; const unsigned m = info[i].m;
        mr[r] = m;
        if (m) { const PolyFeat f = feat[i]; if (f.x0 < 0 || f.y0 < 0 || f.x1 >= W || f.y1 >= H) atomicOr(any_out, 1u); }
    }
    if (r == n) mr[r] = 0;
}
__device__ __forceinline__ int64_t ub_u32v(const unsigned* a, int64_t n, unsigned v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
struct SampleArrs { double* sx; double* sy; double* dprev; int* xi; int* yi; unsigned* rank; uint8_t* inc; };
// 32-bit cell key (column, row) of the point hash (A5); computed while the samples are produced
__device__ __forceinline__ unsigned cell_key32(long long cx, long long cy) { return ((unsigned)((cx + 32768) & 0xffff) << 16) | (unsigned)((cy + 32768) & 0xffff); }
// rank (polyline) and segment of sample g, as k_samples needs them.  Both are monotone in g, so the values of the first sample of a
// 256-sample block and of the next block bound the searches of every sample in between: k_sample_hints does the two full binary
// searches once per block, k_samples only searches between the hints (mostly zero to a few steps instead of ~28 dependent loads).
__device__ __forceinline__ int64_t sample_rank(const unsigned* __restrict__ sbase, const RsInfo* __restrict__ info, const unsigned* __restrict__ ord, int64_t lo, int64_t hi, unsigned g) {
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (sbase[mid] <= g) lo = mid + 1; else hi = mid; }      // first rank in [lo, hi) whose base is > g
    int64_t r = lo - 1;
    while (info[ord[r]].m == 0) r--;
    return r;
}
__device__ __forceinline__ float sample_t(unsigned j, double step) {
    float t0 = 0.0f, t1 = (float)(0.0 + step), delta = __fsub_rn(t1, t0);
    return j == 0 ? t0 : (j == 1 ? t1 : __fadd_rn(t0, __fmul_rn((float)j, delta)));
}
// searchsorted(s, t, 'right') - 1 on s[0..n_eff), clipped to [0, n_eff-2], given klo <= result <= khi
__device__ __forceinline__ int64_t sample_seg(const float* __restrict__ s, int64_t n_eff, double t, int64_t klo, int64_t khi) {
    int64_t lo = klo + 1, hi = khi + 2;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if ((double)s[mid] <= t) lo = mid + 1; else hi = mid; }
    int64_t k = lo - 1; if (k < 0) k = 0; if (k > n_eff - 2) k = n_eff - 2;
    return k;
}
__global__ __launch_bounds__(256) void k_sample_hints(const int64_t* __restrict__ off /* where polyline i's cumulative lengths start in cum */, const float* __restrict__ cum, const RsInfo* __restrict__ info, const unsigned* __restrict__ ord,
                                                       const unsigned* __restrict__ sbase, int64_t n_rank, unsigned MS, double step, unsigned nb, int2* __restrict__ hints) {
    unsigned b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const unsigned g = b * 256u;
    int64_t r = sample_rank(sbase, info, ord, 0, n_rank, g);
    unsigned i = ord[r]; RsInfo ri = info[i];
    int64_t k = 0;
    if (!ri.pass) k = sample_seg(cum + off[i], ri.n_eff, (double)sample_t(g - sbase[r], step), -1, ri.n_eff - 2);
    hints[b] = make_int2((int)r, (int)k);
}
template <class Src>
__global__ __launch_bounds__(256) void k_samples(Src src, const int64_t* __restrict__ cumoff, const float* __restrict__ cum,
                                                  const RsInfo* __restrict__ info, const unsigned* __restrict__ ord, const unsigned* __restrict__ sbase, int64_t n_rank,
                                                  unsigned MS, double step, int W, int H, SampleArrs A, double inv_cell, unsigned* __restrict__ ckeys, unsigned* __restrict__ cvals,
                                                  const int2* __restrict__ hints, unsigned nhb, unsigned long long* __restrict__ pixbits, int Wq, unsigned* __restrict__ firstseq) {
    // FOUR consecutive samples per thread.  A sample costs a chain of ~18 dependent loads (rank, polyline, a bisection of its cumulative lengths, the
    // segment's end points), and with one sample per thread the kernel sat at 1.5 TB/s with every wave slot taken.  Consecutive samples of a polyline
    // lie a few segments apart (8 px of arc length against segments of 2 .. 3 px), so the second to fourth find their segment with ONE round of eight
    // independent loads from where the previous one stood, and their predecessor (the tail bookkeeping's distance, 08:141,147) is in registers.
    constexpr int S = 4;
    __shared__ double shx[256], shy[256];                  // the thread's last sample: the predecessor of the next thread's first one
    const unsigned g0 = (blockIdx.x * 256 + threadIdx.x) * S;
    const bool act = g0 < MS;
    double x0 = 0.0, y0 = 0.0, xl = 0.0, yl = 0.0; unsigned j0 = 0;
    if (act) {
        const unsigned hb = g0 >> 8;                       // hints: rank and segment of every 256th sample (k_sample_hints)
        const int2 h0 = hints[hb];
        const bool last = hb + 1 == nhb;
        const int2 h1 = last ? make_int2((int)n_rank - 1, 0) : hints[hb + 1];
        int64_t r = sample_rank(sbase, info, ord, h0.x + 1, (int64_t)h1.x + 1, g0);      // sbase[h0.x] <= g0 already
        unsigned i = ord[r]; unsigned j = g0 - sbase[r];
        auto cu = src.cur(i); const float* s = cum + cumoff[i];
        RsInfo ri = info[i];
        int64_t kprev = -2;                                // segment of the previous sample of this polyline taken by this thread (-2: none)
        double px = 0.0, py = 0.0;
        j0 = j;
        // position of sample jj of the current polyline, its segment known to lie in [klo, khi]
        auto pos_at = [&](int64_t k, double t, double& ox, double& oy) {
            double sk = (double)s[k], sk1 = (double)s[k + 1];
            double u = __ddiv_rn(__dsub_rn(t, sk), fmax(1e-6, __dsub_rn(sk1, sk)));
            double a = __dsub_rn(1.0, u);
            const int2 p0 = cu.at(k), p1 = cu.at(k + 1);
            ox = __dadd_rn(__dmul_rn((double)(float)p0.x, a), __dmul_rn((double)(float)p1.x, u));
            oy = __dadd_rn(__dmul_rn((double)(float)p0.y, a), __dmul_rn((double)(float)p1.y, u));
        };
#pragma unroll 1
        for (int u = 0; u < S; u++) {
            const unsigned g = g0 + (unsigned)u;
            if (g >= MS) break;
            if (u > 0 && j >= ri.m) {                      // the polyline is used up: on to the next one that has samples
                do { r++; i = ord[r]; ri = info[i]; } while (ri.m == 0);
                j = 0; cu = src.cur(i); s = cum + cumoff[i]; kprev = -2;
            }
            double x, y;
            if (ri.pass) { const int2 q = cu.at(j); x = (double)(float)q.x; y = (double)(float)q.y; }
            else {
                const double t = (double)sample_t(j, step);
                int64_t k;
                if (kprev < -1) {
                    const bool first = u == 0;
                    k = sample_seg(s, ri.n_eff, t, (first && r == h0.x) ? h0.y : -1, (first && !last && r == h1.x) ? h1.y : ri.n_eff - 2);
                } else {
                    // searchsorted(s, t, 'right') - 1, clipped, from the previous sample's segment on: eight lengths per round
                    k = kprev < 0 ? 0 : kprev;
                    const int64_t kmax = ri.n_eff - 2;
                    while (k < kmax) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = (k + 1 + q <= kmax + 1) ? s[k + 1 + q] : __int_as_float(0x7f800000);
                        int cnt = 0; bool run = true;
#pragma unroll
                        for (int q = 0; q < 8; q++) { run = run && ((double)v[q] <= t); cnt += run ? 1 : 0; }
                        k += cnt;
                        if (cnt < 8) break;
                    }
                    if (k > kmax) k = kmax;
                }
                kprev = k;
                pos_at(k, t, x, y);
            }
            long long xi = vs::round_half_even(x), yi = vs::round_half_even(y);
            A.sx[g] = x; A.sy[g] = y; A.rank[g] = (unsigned)r;
            bool in = xi >= 0 && yi >= 0 && xi < W && yi < H;
            A.xi[g] = (int)xi; A.yi[g] = (int)yi; A.inc[g] = in ? 1 : 0;
            if (pixbits && in) {       // the canvas is read at sample pixels only (k_caps_stamp_bits): mark the pixel, give it its "never stamped" value
                unsigned long long* wp = &pixbits[(size_t)yi * Wq + (xi >> 6)]; const unsigned long long bit = 1ULL << (xi & 63);
                if (!(*wp & bit) && !(atomicOr(wp, bit) & bit)) firstseq[(size_t)yi * W + xi] = 0xffffffffu;      // whoever sets the bit initialises the pixel: one write per distinct pixel, not per sample
            }
            if (ckeys) { ckeys[g] = cell_key32((long long)floor(__dmul_rn(x, inv_cell)), (long long)floor(__dmul_rn(y, inv_cell))); cvals[g] = g; }
            // distance to the predecessor on the same polyline, exactly as the tail bookkeeping evaluates it (08:141,147)
            if (j == 0) A.dprev[g] = 0.0;
            else if (u > 0) A.dprev[g] = vs::norm2_f64(x - px, y - py);
            else { x0 = x; y0 = y; }                       // the predecessor is the previous thread's last sample: after the barrier
            px = x; py = y; xl = x; yl = y;
            j++;
        }
        if (j0 > 0 && threadIdx.x == 0) {                  // the first thread of a block computes its predecessor again
            // (the loop above has moved on: look the polyline of sample g0 up again)
            int64_t r2 = sample_rank(sbase, info, ord, h0.x + 1, (int64_t)h1.x + 1, g0);
            const unsigned i2 = ord[r2]; cu = src.cur(i2); s = cum + cumoff[i2]; ri = info[i2];
            double qx, qy;
            if (ri.pass) { const int2 q = cu.at(j0 - 1); qx = (double)(float)q.x; qy = (double)(float)q.y; }
            else { const double t = (double)sample_t(j0 - 1, step); pos_at(sample_seg(s, ri.n_eff, t, -1, ri.n_eff - 2), t, qx, qy); }
            A.dprev[g0] = vs::norm2_f64(x0 - qx, y0 - qy);
        }
    }
    shx[threadIdx.x] = xl; shy[threadIdx.x] = yl;
    __syncthreads();
    if (act && j0 > 0 && threadIdx.x > 0) A.dprev[g0] = vs::norm2_f64(x0 - shx[threadIdx.x - 1], y0 - shy[threadIdx.x - 1]);
}

// distance of every sample to its predecessor on the same polyline, exactly as the tail bookkeeping evaluates it (08:141,147)
__global__ __launch_bounds__(256) void k_sample_dist(const unsigned* __restrict__ sbase, unsigned MS, SampleArrs A) {
    unsigned g = blockIdx.x * 256 + threadIdx.x;
    if (g >= MS) return;
    unsigned b = sbase[A.rank[g]];
    A.dprev[g] = (g > b) ? vs::norm2_f64(A.sx[g] - A.sx[g - 1], A.sy[g] - A.sy[g - 1]) : 0.0;
}

// ================================================================= A3: tail simulation (08:139-155)
// The tail length is a float64 running sum with data-dependent pops: strictly sequential per polyline.  One WAVEFRONT per
// polyline: the distances are loaded 64 at a time (coalesced) and the wave-uniform recurrence picks them out of the lanes with
// v_readlane, for the push stream and for the pop stream; the pop counts go back to memory 64 at a time.
__device__ __forceinline__ double rl_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_TAIL_OLDSIM): variants build only (make variants)
__global__ __launch_bounds__(64) void k_tail_sim(const unsigned* __restrict__ sbase, int64_t n_rank, double tail_len_px, SampleArrs A, unsigned* __restrict__ npop, const unsigned* __restrict__ only) {
    __shared__ double ring[2048];
    __shared__ unsigned npst[1024 + 64];
    const unsigned lane = threadIdx.x;
    for (int64_t r = blockIdx.x; r < n_rank; r += gridDim.x) {
        if (only && !only[r]) continue;       // only the polylines the parallel version could not decide
        const unsigned b = sbase[r], e = sbase[r + 1];
        if (e <= b) continue;
        const double* D = A.dprev + b; unsigned* NP = npop + b;
        const unsigned m = e - b;
        // The distances pass through an LDS ring of two 1024-sample chunks (the chunk of j and the one before it) and the pop counts
        // through an LDS stage: global memory is touched once per chunk, in bulk.  (Windows loaded from or stored to global memory
        // inside the sample loop made the wave wait for a memory round trip every 64 samples: most of the kernel's time.)  The 64-value
        // windows the recurrence picks from come out of the ring; a head that trails by more than a chunk reads global memory.
        constexpr unsigned C = 1024u, RM = 2u * C - 1u;
        auto fill = [&](unsigned c0) {           // 16 independent loads per lane in flight, then the LDS writes (slots past m: the last value, never used)
            double tmp[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const unsigned idx = c0 + lane + 64u * u; tmp[u] = D[idx < m ? idx : m - 1]; }
#pragma unroll
            for (int u = 0; u < 16; u++) ring[(c0 + lane + 64u * u) & RM] = tmp[u];
        };
        auto flush = [&](unsigned c0, unsigned cnt) { for (unsigned t = lane; t < cnt; t += 64u) NP[c0 + t] = npst[t]; };
        fill(0);
        unsigned head = 0, jw = 0, hw = 0xffffffffu, cs = 0;
        double dj = ring[lane], dh = 0.0;
        double tail_len = 0.0; unsigned nv = 0;
        for (unsigned j = 0; j < m; j++) {
            if (j - jw == 64u) {                     // next push window
                npst[jw - cs + lane] = nv;
                jw += 64u;
                if (jw - cs == C) { flush(cs, C); cs += C; fill(cs); }
                dj = ring[(jw + lane) & RM];
            }
            if (j > head) tail_len = __dadd_rn(tail_len, rl_f64(dj, (int)(j - jw)));
            while (head <= j && __builtin_amdgcn_ballot_w64(tail_len > tail_len_px) != 0) {
                head++;
                if (head <= j) {
                    const unsigned w = head & ~63u;
                    if (w != hw) {
                        hw = w;
                        if (w + C >= cs) dh = ring[(w + lane) & RM];                       // the chunk of j or the one before it
                        else { const unsigned idx = w + (unsigned)lane; dh = D[idx < m ? idx : m - 1]; }
                    }
                    tail_len = __dsub_rn(tail_len, rl_f64(dh, (int)(head - w)));
                } else tail_len = 0.0;
            }
            nv = ((unsigned)lane == (j & 63u)) ? head : nv;
        }
        npst[jw - cs + lane] = nv;
        flush(cs, m - cs);
    }
}
#endif
// ---- the sequential simulation, replayed.  k_tail_par leaves for every sample the head the queue WOULD have if every comparison were
// decided by exact arithmetic; the reference decides them with a float64 running sum whose roundings depend on the whole history of pushes
// and pops.  Given the heads, that history is a fixed list of operations (+d[j], then -d[h] for every popped h), and its value after every
// operation is a prefix sum with SEQUENTIAL rounding -- which 64 lanes evaluate as 63 wave-shifted adds (lane i is final after step i,
// exactly as k_cumlen_long2 could for float32).  So a wavefront replays 64 operations at a time instead of deciding one comparison per
// ~400 cycles, then checks the predicted heads against the reference's loop conditions with the running values it now has (after the last
// pop: not > T; before it: > T).  Samples up to the first one that fails the check are final; that one is decided by the plain loop, and
// the replay goes on from there with heads that can only have moved forward (running maximum).  Whatever the prediction was, a sample
// is only ever committed when the reference's own conditions hold on the reference's own running value: the result is the sequential one.
__device__ __forceinline__ double dpp_shr1_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void k_tail_replay(const unsigned* __restrict__ sbase, int64_t n_rank, double T, SampleArrs A, unsigned* __restrict__ npop, const unsigned* __restrict__ only) {
    constexpr unsigned C = 1024u, RM = 2u * C - 1u;          // distances of the current chunk of C samples and of the one before it stay in LDS
    __shared__ double ops[64], rr[64];
    __shared__ double Dl[2 * C];
    __shared__ unsigned NPl[C];
    const int lane = threadIdx.x;
    for (int64_t r = blockIdx.x; r < n_rank; r += gridDim.x) {
        if (only && !only[r]) continue;
        const unsigned b = sbase[r], e = sbase[r + 1];
        if (e <= b) continue;
        const double* D = A.dprev + b; unsigned* NP = npop + b;
        const unsigned m = e - b;
        unsigned head = 0; double racc = 0.0;
        unsigned j0 = 0, cb = 0;
        // one memory round trip per chunk: 32 independent loads per lane in flight, then the LDS writes (a load inside the rounds below
        // would cost the lone wave a round trip per 64 operations: most of the kernel's time)
        auto fill = [&](unsigned c0) {
            double td[16]; unsigned tn[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const unsigned idx = c0 + (unsigned)lane + 64u * u; td[u] = D[idx < m ? idx : m - 1u]; tn[u] = NP[idx < m ? idx : m - 1u]; }
#pragma unroll
            for (int u = 0; u < 16; u++) { const unsigned idx = c0 + (unsigned)lane + 64u * u; Dl[idx & RM] = td[u]; NPl[idx & (C - 1u)] = tn[u]; }
        };
        auto dist = [&](unsigned idx) -> double { return (idx + C >= cb && idx < cb + C) ? Dl[idx & RM] : D[idx]; };      // [cb - C, cb + C) is in LDS
        // the plain loop for one sample (08:139-155), every lane the same
        auto plain = [&](unsigned s) {
            if (s > head) racc = __dadd_rn(racc, dist(s));
            while (head <= s && racc > T) { head++; if (head <= s) racc = __dsub_rn(racc, dist(head)); else racc = 0.0; }
            if (lane == 0) NP[s] = head;
        };
        fill(0);
        __syncthreads();
        while (j0 < m) {
            if (j0 >= cb + C) { __syncthreads(); cb += C; fill(cb); __syncthreads(); }
            const unsigned s = j0 + (unsigned)lane; const bool valid = s < m && s < cb + C;
            unsigned hp = valid ? NPl[s & (C - 1u)] : 0u;
            hp = hp > head ? hp : head;
            hp = wave_incl_scan_max_u32(hp);                                               // heads never move back: running maximum (DPP steps: a ds_bpermute
                                                                                           // round trip per step was a quarter of the round)
            unsigned prevh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hp, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); if (lane == 0) prevh = head;
            const unsigned np = hp - prevh;
            const unsigned inc = valid ? 1u + np : 0u;
            unsigned off = wave_incl_scan_u32(inc);
            const unsigned long long fitm = __ballot(valid && off <= 64u);                 // (off is increasing over the valid lanes: a prefix)
            const int m_fit = __popcll(fitm);
            if (m_fit == 0) { plain(j0); j0++; continue; }                                 // a sample with more than 63 pops: the plain loop
            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)off, m_fit - 1);
            off -= inc;                                                                    // exclusive
            if (lane < m_fit) {
                ops[off] = s > prevh ? dist(s) : 0.0;                                      // the push adds nothing to an empty queue
                for (unsigned t = 0; t < np; t++) ops[off + 1u + t] = -dist(prevh + 1u + t);
            }
            __syncthreads();
            const double v = (unsigned)lane < total ? ops[lane] : 0.0;
            double d = lane == 0 ? __dadd_rn(racc, v) : v;
            double pre = d;
#pragma unroll
            for (int q = 1; q < 64; q++) pre = __dadd_rn(dpp_shr1_f64(pre), d);
            rr[lane] = pre;
            __syncthreads();
            bool bad = false;
            if (lane < m_fit) {
                const double after = rr[off + np];
                bad = after > T || (np > 0u && !(rr[off + np - 1u] > T)) || hp > s;     // (a head beyond its own sample would be the emptied queue: plain loop)
            }
            const unsigned long long badm = __ballot(bad);
            const int ncommit = badm ? __ffsll((long long)badm) - 1 : m_fit;
            if (lane < ncommit) NP[s] = hp;
            if (ncommit > 0) {
                head = (unsigned)__builtin_amdgcn_readlane((int)hp, ncommit - 1);
                const unsigned last_op = (unsigned)__builtin_amdgcn_readlane((int)(off + np), ncommit - 1);
                racc = rr[last_op];
            }
            __syncthreads();
            j0 += (unsigned)ncommit;
            if (badm) { plain(j0); j0++; }
        }
        __syncthreads();
    }
}

// Parallel form of the same simulation.  After sample j is pushed the queue holds samples head..j and tail_len is the sum of the
// distances D[head+1..j]; the pops leave the smallest head with that sum <= tail_len_px (the sums shrink as head grows and a head
// never moves back because D >= 0).  With S = per-polyline inclusive prefix sums of D (rocPRIM scan-by-key) the sum is S[j] - S[head],
// found by binary search.  The reference compares a float64 running sum with its own rounding history; both that sum and S[j]-S[h]
// are within ~1e-8 px of the real sum for polylines shorter than 2^22 px (ulp(2^22) * <64 additions per scan path; 2 ulp(256) per
// push/pop over < 2^20 samples), so a comparison that clears the threshold by more than ORIP_TAIL_EPS is the reference's decision.
// Any sample that is closer marks its polyline, and marked polylines are redone by the sequential k_tail_sim.
#define ORIP_TAIL_EPS 1e-6
__global__ __launch_bounds__(256) void k_tail_par(const unsigned* __restrict__ sbase, const unsigned* __restrict__ rank, const double* __restrict__ S, unsigned MS, double T,
                                                   unsigned* __restrict__ npop, unsigned* __restrict__ redo) {
    __shared__ double win[512];
    const unsigned g0 = blockIdx.x * 256, w0 = g0 >= 256 ? g0 - 256 : 0;      // window = S[w0 .. g0 + 255]
    for (unsigned t = threadIdx.x; t < 512; t += 256) { const unsigned idx = w0 + t; win[t] = (idx < MS && idx < g0 + 256) ? S[idx] : 0.0; }
    __syncthreads();
    unsigned g = g0 + threadIdx.x;
    if (g >= MS) return;
    const unsigned r = rank[g], b = sbase[r];
    const double Sj = S[g];
    bool unsure = !(Sj < 4194304.0) || (g - b) >= (1u << 20);
    // smallest h in [b, g] with Sj - S[h] <= T.  The tail covers a few dozen samples, so the answer almost always lies in the block's LDS
    // window (the 256 sums before the block + its own); otherwise gallop back through global memory, then bisect.
    unsigned lo = b, hi = g;                 // answer in [lo, hi]; S[hi] satisfies (Sj - S[g] = 0 <= T)
    const unsigned wlo = max(b, w0);         // first index of my polyline inside the window
    if (wlo == b || !(Sj - win[wlo - w0] <= T)) {
        if (wlo > b) lo = wlo + 1; else lo = b;
        if (wlo > b) { /* S[wlo] fails: answer in (wlo, g] */ }
        else if (Sj - win[b - w0] <= T) hi = b;                           // the whole prefix fits
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (Sj - win[mid - w0] <= T) hi = mid; else lo = mid + 1; }
    } else {
        hi = wlo;                            // S[wlo] still satisfies: continue below the window in global memory
        for (unsigned stepb = 1; hi > b; stepb <<= 1) {
            const unsigned p = (hi - b > stepb) ? hi - stepb : b;
            if (Sj - S[p] <= T) { hi = p; if (p == b) break; } else { lo = p + 1; break; }
        }
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (Sj - S[mid] <= T) hi = mid; else lo = mid + 1; }
    }
    const unsigned h = lo;
    if (!(Sj - S[h] <= T - ORIP_TAIL_EPS)) unsure = true;
    if (h > b && !(Sj - S[h - 1] > T + ORIP_TAIL_EPS)) unsure = true;
    npop[g] = h - b;
    if (unsure) redo[r] = 1u;
}
// previous in-canvas sample of the same polyline (the far end of the capsule stamped when sample j is popped, 08:151-155); -1: none, -2: j is off-canvas
// lastin[g] = 1 + index of the last in-canvas sample at or before g inside its polyline (0: none): a max-scan by polyline
struct IncIndex {
    const uint8_t* inc;
    __device__ unsigned operator()(unsigned g) const { return inc[g] ? g + 1u : 0u; }
};
__global__ __launch_bounds__(256) void k_capprev(const unsigned* __restrict__ sbase, unsigned MS, SampleArrs A, const unsigned* __restrict__ lastin, int* __restrict__ capprev) {
    unsigned g = blockIdx.x * 256 + threadIdx.x;
    if (g >= MS) return;
    if (!A.inc[g]) { capprev[g] = -2; return; }
    const unsigned b = sbase[A.rank[g]];
    const unsigned l = g > b ? lastin[g - 1] : 0u;
    capprev[g] = l ? (int)(l - 1u - b) : -1;
}

// ================================================================= A4: capsule de-duplication + min-sequence stamping
__device__ __forceinline__ unsigned long long cap_key(int x0, int y0, int x1, int y1) {
    unsigned long long a = ((unsigned long long)(unsigned)x0 << 14) | (unsigned)y0, b = ((unsigned long long)(unsigned)x1 << 14) | (unsigned)y1;
    if (b < a) { unsigned long long t = a; a = b; b = t; }
    return ((a << 28) | b) + 1ULL;     // 0 is the empty marker
}
__device__ __forceinline__ unsigned long long hash64(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// one 16-byte slot per capsule: key and first sequence number arrive in one memory sector (the table is far larger than the caches and
// every probe is a random access: two arrays meant two sectors per probe)
struct __attribute__((aligned(16))) CapSlot { unsigned long long key; unsigned val; unsigned pad; };
__global__ __launch_bounds__(256) void k_caps_init(CapSlot* __restrict__ tab, unsigned long long tsize) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < tsize) reinterpret_cast<uint4*>(tab)[i] = make_uint4(0u, 0u, 0xffffffffu, 0u);
}
__global__ __launch_bounds__(256) void k_caps_insert(SampleArrs A, const unsigned* __restrict__ sbase, const int* __restrict__ capprev, unsigned MS,
                                                      CapSlot* tab, unsigned long long tmask, int max_probe, int* __restrict__ overflow) {
    // Four samples per thread, a block's 1024 samples apart by 256: the chain rank -> base -> pixels -> slot is four dependent loads deep, and with one
    // sample per thread the kernel waits for them one after the other (1 TB/s of the card's 8 with every wave slot full); four independent chains per
    // thread keep four times as many loads in flight.
    constexpr int S = 4;
    const unsigned g0 = blockIdx.x * (256 * S) + threadIdx.x;
    unsigned g[S], bb[S]; int cp[S]; bool on[S];
#pragma unroll
    for (int u = 0; u < S; u++) { g[u] = g0 + 256u * u; on[u] = g[u] < MS; bb[u] = on[u] ? A.rank[g[u]] : 0u; }
#pragma unroll
    for (int u = 0; u < S; u++) if (on[u]) bb[u] = sbase[bb[u]];
#pragma unroll
    for (int u = 0; u < S; u++) {
        cp[u] = !on[u] ? -1 : (capprev ? capprev[g[u]] : (g[u] > bb[u] ? (int)(g[u] - bb[u]) - 1 : -1));      // capprev == nullptr: every sample is on the canvas, so the capsule runs from the previous sample
        on[u] = cp[u] >= 0;
    }
    unsigned long long key[S], h[S];
#pragma unroll
    for (int u = 0; u < S; u++) {
        key[u] = 0; h[u] = 0;
        if (on[u]) { key[u] = cap_key(A.xi[bb[u] + cp[u]], A.yi[bb[u] + cp[u]], A.xi[g[u]], A.yi[g[u]]); h[u] = hash64(key[u]) & tmask; }
    }
    uint4 sl[S];
#pragma unroll
    for (int u = 0; u < S; u++) sl[u] = on[u] ? *reinterpret_cast<const uint4*>(&tab[h[u]]) : make_uint4(0, 0, 0, 0);      // first probes of all four in flight together
#pragma unroll
    for (int u = 0; u < S; u++) {
        if (!on[u]) continue;
        uint4 s = sl[u]; unsigned long long hh = h[u];
        for (int probe = 0;; probe++) {
            if (probe >= max_probe) { *overflow = 1; break; }      // table too small for the number of distinct capsules: the host retries larger
            if (probe) s = *reinterpret_cast<const uint4*>(&tab[hh]);      // key and value in one 16-byte load (every probe reads another slot)
            unsigned long long cur = ((unsigned long long)s.y << 32) | s.x;
            if (cur == 0) { unsigned long long old = atomicCAS(&tab[hh].key, 0ULL, key[u]); if (old == 0 || old == key[u]) cur = key[u]; else cur = old; }
            if (cur == key[u]) { if (s.z > g[u]) atomicMin(&tab[hh].val, g[u]); break; }    // the minimum only decreases: a stale read can only cost a useless atomic
            hh = (hh + 1) & tmask;
        }
    }
}
__global__ __launch_bounds__(256) void k_caps_stamp(const CapSlot* __restrict__ tab, unsigned long long tsize,
                                                     int rad, unsigned* __restrict__ firstseq, int W, int H) {
    const int lane = threadIdx.x & 63;
    const long long r2 = (long long)rad * rad;
    unsigned long long wave = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * 256) >> 6;
    for (unsigned long long s0 = wave * 64; s0 < tsize; s0 += nwaves * 64) {
        const uint4 sl = (s0 + lane < tsize) ? reinterpret_cast<const uint4*>(tab)[s0 + lane] : make_uint4(0u, 0u, 0u, 0u);
        unsigned long long k = ((unsigned long long)sl.y << 32) | sl.x;
        unsigned v = sl.z;
        unsigned long long occ = __ballot(k != 0);
        while (occ) {
            int src = __ffsll((long long)occ) - 1; occ &= occ - 1;
            unsigned long long kk = __shfl(k, src, 64) - 1ULL; unsigned seq = __shfl(v, src, 64);
            unsigned long long a = kk >> 28, b = kk & ((1ULL << 28) - 1);
            int x0 = (int)(a >> 14), y0 = (int)(a & 16383), x1 = (int)(b >> 14), y1 = (int)(b & 16383);
            int bx0 = max(0, min(x0, x1) - rad), bx1 = min(W - 1, max(x0, x1) + rad), by0 = max(0, min(y0, y1) - rad), by1 = min(H - 1, max(y0, y1) + rad);
            int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            for (int i = lane; i < bw * bh; i += 64) {
                int x = bx0 + i % bw, y = by0 + i / bw;
                if (vs::in_capsule(x, y, x0, y0, x1, y1, r2)) atomicMin(&firstseq[(size_t)y * W + x], seq);
            }
        }
    }
}

// The canvas is only ever READ at the pixels of samples (k_accept_pre: "was my pixel stamped before my own pops?"), and those are a thin
// set: the rounded sample positions, i.e. pixels on the paths.  k_samples sets one bit per sample pixel in a bit plane of the canvas
// (12.5 MB, cache-resident) and gives those pixels their "never stamped" value; a capsule then visits the words of the plane its box
// covers and tests / stamps only the set bits -- ~100 pixels instead of the ~1800 of its box, and no 400 MB clear of the canvas.
__global__ __launch_bounds__(256) void k_caps_stamp_bits(const CapSlot* __restrict__ tab, unsigned long long tsize, int rad, unsigned* __restrict__ firstseq, int W, int H,
                                                          const unsigned long long* __restrict__ pixbits, int Wq, unsigned* __restrict__ n_distinct) {
    const int lane = threadIdx.x & 63;
    const long long r2 = (long long)rad * rad;
    unsigned long long wave = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * 256) >> 6;
    unsigned mine = 0;
    for (unsigned long long s0 = wave * 64; s0 < tsize; s0 += nwaves * 64) {
        const uint4 sl = (s0 + lane < tsize) ? reinterpret_cast<const uint4*>(tab)[s0 + lane] : make_uint4(0u, 0u, 0u, 0u);
        unsigned long long k = ((unsigned long long)sl.y << 32) | sl.x;
        unsigned v = sl.z;
        unsigned long long occ = __ballot(k != 0);
        mine += (unsigned)__popcll(occ);
        while (occ) {
            int src = __ffsll((long long)occ) - 1; occ &= occ - 1;
            unsigned long long kk = __shfl(k, src, 64) - 1ULL; unsigned seq = __shfl(v, src, 64);
            unsigned long long a = kk >> 28, b = kk & ((1ULL << 28) - 1);
            int x0 = (int)(a >> 14), y0 = (int)(a & 16383), x1 = (int)(b >> 14), y1 = (int)(b & 16383);
            int bx0 = max(0, min(x0, x1) - rad), bx1 = min(W - 1, max(x0, x1) + rad), by0 = max(0, min(y0, y1) - rad), by1 = min(H - 1, max(y0, y1) + rad);
            const int w0 = bx0 >> 6, nw = (bx1 >> 6) - w0 + 1, bh = by1 - by0 + 1;
            for (int i = lane; i < nw * bh; i += 64) {
                const int y = by0 + i / nw, wq = w0 + i % nw;
                unsigned long long bits = pixbits[(size_t)y * Wq + wq];
                const int xb = wq << 6;
                if (xb < bx0) bits &= ~0ULL << (bx0 - xb);                         // the part of the word inside the box
                if (xb + 63 > bx1) bits &= ~0ULL >> (xb + 63 - bx1);
                while (bits) {
                    const int j = __ffsll((long long)bits) - 1; bits &= bits - 1;
                    const int x = xb + j;
                    if (vs::in_capsule(x, y, x0, y0, x1, y1, r2)) { unsigned* q = &firstseq[(size_t)y * W + x]; if (*q > seq) atomicMin(q, seq); }   // (minima only decrease: a stale read costs a useless atomic at worst)
                }
            }
        }
    }
    if (lane == 0 && mine) atomicAdd(n_distinct, mine);
}

// ================================================================= A5: _PointHash.near (08:85-93)
// The samples of a polyline are contiguous (rank-major), so the hash of a polyline is its own sample range sorted by cell: a
// segmented sort on the 32-bit cell key (column, row).  The sort is stable, so every bucket lists its samples in pop order.
__device__ __forceinline__ unsigned cell_key(long long cx, long long cy) {
    return ((unsigned)((cx + 32768) & 0xffff) << 16) | (unsigned)((cy + 32768) & 0xffff);
}
__global__ __launch_bounds__(256) void k_cell_keys(SampleArrs A, unsigned MS, double inv, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    unsigned g = blockIdx.x * 256 + threadIdx.x;
    if (g >= MS) return;
    long long cx = (long long)floor(__dmul_rn(A.sx[g], inv)), cy = (long long)floor(__dmul_rn(A.sy[g], inv));
    keys[g] = cell_key(cx, cy); vals[g] = g;
}
// Two passes: the cheap test (own sample on the canvas, first stamp of its pixel earlier than its own pops) streams over all samples
// and collects the survivors; the hash-bucket searches (dozens of dependent loads) then run over the dense survivor list, so a wave
// is not held up by one lane that has to search.
__global__ __launch_bounds__(256) void k_accept_pre(SampleArrs A, const unsigned* __restrict__ sbase, const unsigned* __restrict__ npop, unsigned MS,
                                                     const unsigned* __restrict__ firstseq, int W, int2* __restrict__ spt, uint8_t* __restrict__ sflag,
                                                     unsigned* __restrict__ surv, unsigned* __restrict__ n_surv, unsigned long long* __restrict__ work) {
    // four samples per thread, 256 apart (as k_caps_insert: the chains rank -> base and pixel -> canvas word are waited for, not the bandwidth)
    constexpr int S = 4;
    const unsigned g0 = blockIdx.x * (256 * S) + threadIdx.x;
    unsigned g[S], bb[S], np[S]; int xi[S], yi[S]; bool ok[S], on[S];
#pragma unroll
    for (int u = 0; u < S; u++) { g[u] = g0 + 256u * u; on[u] = g[u] < MS; bb[u] = on[u] ? A.rank[g[u]] : 0u; }
#pragma unroll
    for (int u = 0; u < S; u++) {
        np[u] = 0; xi[u] = 0; yi[u] = 0; ok[u] = false;
        if (on[u]) { bb[u] = sbase[bb[u]]; ok[u] = A.inc[g[u]] != 0; np[u] = npop[g[u]]; xi[u] = A.xi[g[u]]; yi[u] = A.yi[g[u]]; }
    }
    unsigned fs[S];
#pragma unroll
    for (int u = 0; u < S; u++) fs[u] = (on[u] && ok[u]) ? firstseq[(size_t)yi[u] * W + xi[u]] : 0xffffffffu;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < S; u++) {
        bool need = false; unsigned mynp = 0;
        if (on[u]) {
            const unsigned j = g[u] - bb[u];
            const unsigned limit = bb[u] + np[u];         // own samples with global index < limit have been popped (hashed + stamped)
            bool k = ok[u];
            if (k && fs[u] < limit) k = false;
            spt[g[u]] = make_int2((int)A.sx[g[u]], (int)A.sy[g[u]]);
            sflag[g[u]] = (k ? 1 : 0) | (j == 0 ? 2 : 0);
            need = k && np[u] > 0; mynp = need ? np[u] : 0u;
        }
        const unsigned long long m = __ballot(need);
        if (m) {
            unsigned long long wsum = mynp;                    // popped own samples the survivors of this wave have to be compared with
            for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o, 64);
            unsigned base = 0;
            if (lane == 0) { base = atomicAdd(n_surv, (unsigned)__popcll(m)); atomicAdd(work, wsum); }
            base = (unsigned)__shfl((int)base, 0, 64);
            if (need) surv[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = g[u];
        }
    }
}
// _PointHash.near without the hash: a survivor is compared with ALL popped samples of its own polyline, 64 at a time.  Equal to the
// hash answer whenever the cell is at least the radius (every point within R then lies in the 3 x 3 cells the reference looks at), and
// cheap whenever the survivors are few and early in their polylines -- the bench image: 56 k survivors of 7.4e7 samples, all within
// the first lap of their walk; the bucket sort of ALL samples this replaces was the largest kernel of stage 08-A.  The host picks
// this path from the work sum k_accept_pre leaves (sum of popped samples over the survivors) and keeps the sorted buckets otherwise.
__global__ __launch_bounds__(256) void k_accept_brute(SampleArrs A, const unsigned* __restrict__ sbase, const unsigned* __restrict__ npop, double R2,
                                                       const unsigned* __restrict__ surv, const unsigned* __restrict__ n_surv, uint8_t* __restrict__ sflag) {
    const unsigned ns = *n_surv;
    const int lane = threadIdx.x & 63;
    for (unsigned t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ns; t += gridDim.x * 4) {
        const unsigned g = surv[t];
        const unsigned b = sbase[A.rank[g]], np = npop[g];
        const double x = A.sx[g], y = A.sy[g];
        bool rej = false;
        for (unsigned q0 = 0; q0 < np && !rej; q0 += 64) {
            const unsigned q = q0 + (unsigned)lane; bool hit = false;
            if (q < np) {
                double ddx = __dsub_rn(A.sx[b + q], x), ddy = __dsub_rn(A.sy[b + q], y);
                hit = __dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy)) <= R2;
            }
            if (__ballot(hit)) rej = true;
        }
        if (rej && lane == 0) sflag[g] &= (uint8_t)~1u;
    }
}
// one wavefront per survivor: 65-ary lower-bound searches and 64-wide scans of the three buckets of a column (they are neighbours in
// key order).  A bucket lists the polyline's own samples in pop order, so "popped before me" is simply g2 < limit; the reference
// stops at the first later sample, here later samples are just not counted -- the answer (any earlier sample within R) is the same.
__global__ __launch_bounds__(256) void k_accept(SampleArrs A, const unsigned* __restrict__ sbase, const unsigned* __restrict__ npop, double inv, double R2,
                                                 const unsigned* __restrict__ skeys, const unsigned* __restrict__ svals,
                                                 const unsigned* __restrict__ surv, const unsigned* __restrict__ n_surv, uint8_t* __restrict__ sflag) {
    const unsigned ns = *n_surv;
    const int lane = threadIdx.x & 63;
    for (unsigned t = blockIdx.x * 4 + (threadIdx.x >> 6); t < ns; t += gridDim.x * 4) {
        const unsigned g = surv[t];
        const unsigned r = A.rank[g], b = sbase[r];
        const double x = A.sx[g], y = A.sy[g];
        const unsigned limit = b + npop[g];
        const long long cx = (long long)floor(__dmul_rn(x, inv)), cy = (long long)floor(__dmul_rn(y, inv));
        const long long seg_end = sbase[r + 1];
        bool rej = false;
        for (int dx = -1; dx <= 1 && !rej; dx++) {
            const unsigned key_lo = cell_key(cx + dx, cy - 1), key_hi = cell_key(cx + dx, cy + 1);
            long long lo = b, hi = seg_end;                       // first entry >= key_lo
            while (hi - lo > 0) {
                const long long w = (hi - lo + 64) / 65;          // 64 probes split [lo, hi) into 65 parts
                const long long pos = lo + (long long)(lane + 1) * w - 1;
                const bool below = pos < hi && skeys[pos] < key_lo;
                const int cnt = __popcll(__ballot(below));        // probes are increasing: the `below` lanes are a prefix
                const long long nlo = lo + (long long)cnt * w;
                const long long nhi = (cnt < 64) ? min(hi, lo + (long long)(cnt + 1) * w - 1) : hi;
                lo = min(nlo, hi); hi = nhi;
            }
            for (long long q = lo; q < seg_end; q += 64) {
                const long long idx = q + lane;
                bool in = false, hit = false;
                if (idx < seg_end) {
                    const unsigned k = skeys[idx];
                    in = k <= key_hi;
                    if (in) {
                        const unsigned g2 = svals[idx];
                        if (g2 < limit) {
                            double ddx = __dsub_rn(A.sx[g2], x), ddy = __dsub_rn(A.sy[g2], y);
                            hit = __dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy)) <= R2;
                        }
                    }
                }
                if (__ballot(hit)) { rej = true; break; }
                if (__ballot(in) != ~0ull) break;
            }
        }
        if (rej && lane == 0) sflag[g] &= (uint8_t)~1u;
    }
}

// ================================================================= B: _post_skeleton_merge
__device__ __forceinline__ int ufind(const int* L, int a) { int p = L[a]; while (p != a) { a = p; p = L[a]; } return a; }
__device__ __forceinline__ void uunite(int* L, int a, int b) {
    bool done;
    do {
        a = ufind(L, a); b = ufind(L, b);
        if (a < b) { int old = atomicMin(&L[b], a); done = (old == b); b = old; }
        else if (b < a) { int old = atomicMin(&L[a], b); done = (old == a); a = old; }
        else done = true;
    } while (!done);
}
__global__ __launch_bounds__(256) void k_iota(int* a, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) a[i] = i; }
__global__ __launch_bounds__(256) void k_bbox_pairs(const PolyFeat* __restrict__ f, int n, int exp, int* __restrict__ par) {
    // bboxes expanded by exp on each side overlap  <=>  not separated (08:41-42)
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        int ax0 = f[i].x0 - exp, ay0 = f[i].y0 - exp, ax1 = f[i].x1 + exp, ay1 = f[i].y1 + exp;
        for (int j = i + 1 + threadIdx.x; j < n; j += 256) {
            int bx0 = f[j].x0 - exp, by0 = f[j].y0 - exp, bx1 = f[j].x1 + exp, by1 = f[j].y1 + exp;
            if (!(ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0)) uunite(par, i, j);
        }
    }
}
struct GroupInfo { int x0, y0, x1, y1; unsigned long long longest; unsigned long long near0, near1; int rank; int a0x, a0y, a1x, a1y; };
__global__ __launch_bounds__(256) void k_group_init(GroupInfo* g, int n) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    GroupInfo q; q.x0 = q.y0 = 0x7fffffff; q.x1 = q.y1 = -0x7fffffff; q.longest = ~0ULL; q.near0 = q.near1 = ~0ULL; q.rank = -1; q.a0x = q.a0y = q.a1x = q.a1y = 0;
    g[i] = q;
}
__global__ __launch_bounds__(256) void k_group_accum(const PolyFeat* __restrict__ f, int n, int exp, int* __restrict__ par, GroupInfo* __restrict__ g, unsigned* __restrict__ is_root) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i > n) return;
    if (i == n) { is_root[i] = 0; return; }
    int r = ufind(par, i); par[i] = r;
    is_root[i] = (r == i) ? 1u : 0u;
    atomicMin(&g[r].x0, f[i].x0 - exp); atomicMin(&g[r].y0, f[i].y0 - exp); atomicMax(&g[r].x1, f[i].x1 + exp); atomicMax(&g[r].y1, f[i].y1 + exp);
    unsigned long long key = ((unsigned long long)(~__float_as_uint(f[i].per)) << 32) | (unsigned)i;     // longest, first index on ties (08:391)
    atomicMin(&g[r].longest, key);
}
__global__ __launch_bounds__(256) void k_group_finish(const PolyFeat* __restrict__ f, int n, const unsigned* __restrict__ is_root, const unsigned* __restrict__ root_scan, GroupInfo* __restrict__ g) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n || !is_root[i]) return;
    g[i].rank = (int)root_scan[i];
    int l = (int)(g[i].longest & 0xffffffffu);
    g[i].a0x = f[l].sx; g[i].a0y = f[l].sy; g[i].a1x = f[l].ex; g[i].a1y = f[l].ey;
}
// raster: gid[pixel] = group root + 1 for every pixel within r of a segment of a line of the group; one wave per segment
__global__ __launch_bounds__(256) void k_stamp_groups(const int64_t* __restrict__ off, const int32_t* __restrict__ pts, int64_t n_polys, int64_t n_pts, const int* __restrict__ par,
                                                       int rad, unsigned* __restrict__ gid, int Wp, int Hp) {
    const int lane = threadIdx.x & 63; const long long r2 = (long long)rad * rad;
    long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long long)gridDim.x * 256) >> 6;
    for (long long i = wave; i + 1 < n_pts; i += nw) {
        // polyline of point i: last off <= i
        long long lo = 0, hi = n_polys;
        while (lo < hi) { long long mid = (lo + hi) >> 1; if (off[mid + 1] <= i) lo = mid + 1; else hi = mid; }
        if (i + 1 >= off[lo + 1]) continue;                 // i is the last point of its polyline
        unsigned val = (unsigned)par[lo] + 1u;
        int x0 = pts[2 * i] + PAD8, y0 = pts[2 * i + 1] + PAD8, x1 = pts[2 * i + 2] + PAD8, y1 = pts[2 * i + 3] + PAD8;
        int bx0 = max(0, min(x0, x1) - rad), bx1 = min(Wp - 1, max(x0, x1) + rad), by0 = max(0, min(y0, y1) - rad), by1 = min(Hp - 1, max(y0, y1) + rad);
        int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
        if (bw <= 0 || bh <= 0) continue;
        for (int q = lane; q < bw * bh; q += 64) {
            int x = bx0 + q % bw, y = by0 + q / bw;
            if (vs::in_capsule(x, y, x0, y0, x1, y1, r2)) gid[(size_t)y * Wp + x] = val;
        }
    }
}
// mask of the stamped raster + list of the 64x4 tiles that hold foreground (thinning only ever clears pixels, so the other tiles stay empty)
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_gid_to_mask(const unsigned* __restrict__ gid, u8* __restrict__ m, int H, int W, unsigned* __restrict__ tiles, unsigned* __restrict__ ntiles) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int fg = 0;
    if (x < W && y < H) { size_t o = (size_t)y * W + x; fg = gid[o] ? 1 : 0; m[o] = fg ? 255 : 0; }
    if (__syncthreads_or(fg) && threadIdx.x == 0) tiles[atomicAdd(ntiles, 1u)] = blockIdx.y * gridDim.x + blockIdx.x;
}
#endif
// standard-orientation Zhang-Suen sub-iteration (08:349-366) over the listed tiles
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_zs_sub(const u8* __restrict__ s, u8* __restrict__ d, int H, int W, int sub, int* __restrict__ changed,
                                                 const unsigned* __restrict__ tiles, const unsigned* __restrict__ ntiles, int gx) {
    const unsigned nt = *ntiles;
    for (unsigned ti = blockIdx.x; ti < nt; ti += gridDim.x) {
        const unsigned t = tiles[ti];
        int x = (int)(t % gx) * 64 + (threadIdx.x & 63), y = (int)(t / gx) * 4 + (threadIdx.x >> 6);
        if (x >= W || y >= H) continue;
        size_t o = (size_t)y * W + x;
        u8 v = s[o];
        if (v) {
            auto g = [&](int dy, int dx) -> int { int yy = y + dy, xx = x + dx; return (yy >= 0 && yy < H && xx >= 0 && xx < W && s[(size_t)yy * W + xx]) ? 1 : 0; };
            int P2 = g(-1, 0), P3 = g(-1, 1), P4 = g(0, 1), P5 = g(1, 1), P6 = g(1, 0), P7 = g(1, -1), P8 = g(0, -1), P9 = g(-1, -1);
            int Bn = P2 + P3 + P4 + P5 + P6 + P7 + P8 + P9;
            int A = (!P2 && P3) + (!P3 && P4) + (!P4 && P5) + (!P5 && P6) + (!P6 && P7) + (!P7 && P8) + (!P8 && P9) + (!P9 && P2);
            bool cnd = sub == 0 ? (P2 * P4 * P6 == 0 && P4 * P6 * P8 == 0) : (P2 * P4 * P8 == 0 && P2 * P6 * P8 == 0);
            if (A == 1 && Bn >= 2 && Bn <= 6 && cnd) { v = 0; *changed = 1; }
        }
        d[o] = v ? 255 : 0;
    }
}
#endif
// ---- the same thinning on bit planes (one bit per pixel, 64 pixels per word; the padded canvas is 12.8 MB, i.e. cache-resident).
// A sub-iteration evaluates the Zhang-Suen conditions for 64 pixels at once with bit-sliced logic: the eight neighbour planes come
// from the three rows by word shifts, B = P2+...+P9 from a carry-save adder tree, A == 1 ("exactly one 0->1 transition") from a
// one/two accumulator.  Same conditions as k_zs_sub (08:349-366), out-of-image pixels are background.
__global__ __launch_bounds__(256) void k_gid_to_bits(const unsigned* __restrict__ gid, unsigned long long* __restrict__ bits, int H, int W, int Ww) {
    // a wave packs 64 consecutive words: one coalesced 256-byte read + one ballot per word, then one coalesced write of the 64 words
    const size_t w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64, nw = (size_t)H * Ww;
    if (w0 >= nw) return;
    const int lane = threadIdx.x & 63;
    unsigned long long mine = 0;
    for (int j = 0; j < 64; j++) {
        const size_t wi = w0 + j;
        bool fg = false;
        if (wi < nw) { const int y = (int)(wi / Ww), x = (int)(wi % Ww) * 64 + lane; fg = x < W && gid[(size_t)y * W + x] != 0; }
        const unsigned long long b = __ballot(fg);
        if (lane == j) mine = b;
    }
    if (w0 + lane < nw) bits[w0 + lane] = mine;
}
__global__ __launch_bounds__(256) void k_bits_to_mask(const unsigned long long* __restrict__ bits, u8* __restrict__ m, int H, int W, int Ww) {
    const size_t w0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64, nw = (size_t)H * Ww;
    if (w0 >= nw) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long mine = (w0 + lane < nw) ? bits[w0 + lane] : 0ULL;
    for (int j = 0; j < 64; j++) {
        const size_t wi = w0 + j; if (wi >= nw) break;
        const unsigned long long b = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
        const int y = (int)(wi / Ww), x = (int)(wi % Ww) * 64 + lane;
        if (x < W) m[(size_t)y * W + x] = ((b >> lane) & 1ULL) ? 255 : 0;
    }
}
// one Zhang-Suen sub-iteration on a 64-pixel word, bit-sliced: M = the word, the other eight = its neighbour words; returns the pixels it deletes
__device__ __forceinline__ unsigned long long zs_word_del(unsigned long long M, unsigned long long U, unsigned long long UL, unsigned long long UR, unsigned long long ML,
                                                          unsigned long long MR, unsigned long long D, unsigned long long DL, unsigned long long DR, int sub) {
    // neighbour planes in the reference's numbering: P2 = north, then clockwise
    const unsigned long long P2 = U, P3 = (U >> 1) | (UR << 63), P4 = (M >> 1) | (MR << 63), P5 = (D >> 1) | (DR << 63);
    const unsigned long long P6 = D, P7 = (D << 1) | (DL >> 63), P8 = (M << 1) | (ML >> 63), P9 = (U << 1) | (UL >> 63);
    // B = number of foreground neighbours, bit-sliced (b0 ones, b1 twos, b2 fours, b3 eights)
    auto FA = [](unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long& sum, unsigned long long& carry) { const unsigned long long t = a ^ b; sum = t ^ c; carry = (a & b) | (t & c); };
    unsigned long long s1, c1, s2, c2, s4, c4, s5, c5;
    FA(P2, P3, P4, s1, c1); FA(P5, P6, P7, s2, c2);
    const unsigned long long s3 = P8 ^ P9, c3 = P8 & P9;
    FA(s1, s2, s3, s4, c4);
    FA(c1, c2, c3, s5, c5);
    const unsigned long long b0 = s4, b1 = s5 ^ c4, c6 = s5 & c4, b2 = c5 ^ c6, b3 = c5 & c6;
    const unsigned long long Bok = (b1 | b2) & ~b3 & ~(b2 & b1 & b0);          // 2 <= B <= 6
    // A = number of 0 -> 1 transitions in P2, P3, ..., P9, P2: exactly one
    unsigned long long one = 0, two = 0;
    auto TR = [&](unsigned long long a, unsigned long long b) { const unsigned long long t = ~a & b; two |= one & t; one |= t; };
    TR(P2, P3); TR(P3, P4); TR(P4, P5); TR(P5, P6); TR(P6, P7); TR(P7, P8); TR(P8, P9); TR(P9, P2);
    const unsigned long long Aok = one & ~two;
    const unsigned long long cnd = sub == 0 ? (~(P2 & P4 & P6) & ~(P4 & P6 & P8)) : (~(P2 & P4 & P8) & ~(P2 & P6 & P8));
    return M & Aok & Bok & cnd;
}
__global__ __launch_bounds__(256) void k_zs_bits(const unsigned long long* __restrict__ s, unsigned long long* __restrict__ d, int H, int Ww, int sub, int* __restrict__ changed) {
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= (size_t)H * Ww) return;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww);
    const unsigned long long M = s[wi];
    if (!M) { d[wi] = 0; return; }
    auto W64 = [&](int yy, int xx) -> unsigned long long { return (yy < 0 || yy >= H || xx < 0 || xx >= Ww) ? 0ULL : s[(size_t)yy * Ww + xx]; };
    const unsigned long long del = zs_word_del(M, W64(y - 1, xw), W64(y - 1, xw - 1), W64(y - 1, xw + 1), W64(y, xw - 1), W64(y, xw + 1), W64(y + 1, xw), W64(y + 1, xw - 1), W64(y + 1, xw + 1), sub);
    if (del) *changed = 1;
    d[wi] = M & ~del;
}
// `iters` whole iterations (two sub-iterations each) in ONE launch: a block keeps a tile of 64 rows x 2 words plus a halo of ZS_HALO rows / one word on
// every side in LDS and runs the sub-iterations there.  A sub-iteration reads the 3x3 neighbourhood, so after t of them the tile is exact everywhere at
// least t pixels inside the staged region: with iters <= ZS_HALO / 2 the core is exact after all of them, whatever the neighbouring tiles do meanwhile
// (they read the same source plane s; the result goes to d).  changed[b] is set when iteration b deletes a pixel of some core.  Twelve iterations were
// 24 dispatches of ~30 us on the layer's chain; most tiles of a canvas of thin lines are empty and leave after the staging.
#define ZS_HALO 24
#define ZS_TR 64
#define ZS_ROWS (ZS_TR + 2 * ZS_HALO)
__global__ __launch_bounds__(256) void k_zs_tile(const unsigned long long* __restrict__ s, unsigned long long* __restrict__ d, int H, int Ww, int iters, int* __restrict__ changed) {
    __shared__ unsigned long long T[2][ZS_ROWS][4];
    __shared__ int any_s;
    const int tid = threadIdx.x;
    const int y0 = blockIdx.y * ZS_TR - ZS_HALO, x0 = blockIdx.x * 2 - 1;          // first staged row / word
    if (tid == 0) any_s = 0;
    __syncthreads();
    int any = 0;
    for (int i = tid; i < ZS_ROWS * 4; i += 256) {
        const int r = i >> 2, wx = i & 3, y = y0 + r, xw = x0 + wx;
        const unsigned long long v = (y < 0 || y >= H || xw < 0 || xw >= Ww) ? 0ULL : s[(size_t)y * Ww + xw];
        T[0][r][wx] = v; any |= v != 0;
    }
    if (any) any_s = 1;
    __syncthreads();
    const bool empty = !any_s;
    int cur = 0;
    if (!empty) {
        for (int t = 0; t < 2 * iters; t++) {
            int del_core = 0;
            for (int i = tid; i < ZS_ROWS * 4; i += 256) {
                const int r = i >> 2, wx = i & 3;
                const unsigned long long M = T[cur][r][wx];
                unsigned long long out = 0;
                if (M) {
                    auto G = [&](int rr, int ww) -> unsigned long long { return (rr < 0 || rr >= ZS_ROWS || ww < 0 || ww > 3) ? 0ULL : T[cur][rr][ww]; };
                    const unsigned long long del = zs_word_del(M, G(r - 1, wx), G(r - 1, wx - 1), G(r - 1, wx + 1), G(r, wx - 1), G(r, wx + 1), G(r + 1, wx), G(r + 1, wx - 1), G(r + 1, wx + 1), t & 1);
                    out = M & ~del;
                    if (del && r >= ZS_HALO && r < ZS_HALO + ZS_TR && (wx == 1 || wx == 2)) del_core = 1;
                }
                T[cur ^ 1][r][wx] = out;
            }
            if (del_core) changed[t >> 1] = 1;
            cur ^= 1;
            __syncthreads();
        }
    }
    for (int i = tid; i < ZS_TR * 2; i += 256) {
        const int r = ZS_HALO + (i >> 1), wx = 1 + (i & 1), y = y0 + r, xw = x0 + wx;
        if (y < H && xw < Ww) d[(size_t)y * Ww + xw] = empty ? 0ULL : T[cur][r][wx];
    }
}
// plain (linear id) union-find CCL on the padded raster
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_ccl2_init(const u8* __restrict__ s, int* __restrict__ L, int H, int W) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    int id = y * W + x; if (s[id]) L[id] = id;          // background parents are never read: not written either
}
#endif
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_ccl2_merge(const u8* __restrict__ s, int* __restrict__ L, int H, int W) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    int id = y * W + x;
    if (!s[id]) return;
    if (x > 0 && s[id - 1]) uunite(L, id, id - 1);
    if (y > 0) {
        if (x > 0 && s[id - W - 1]) uunite(L, id, id - W - 1);
        if (s[id - W]) uunite(L, id, id - W);
        if (x + 1 < W && s[id - W + 1]) uunite(L, id, id - W + 1);
    }
}
#endif
#ifdef ORIP_VARIANTS      // replaced variant (ORIP_THIN_BYTES): variants build only (make variants)
__global__ __launch_bounds__(256) void k_ccl2_flatten(const u8* __restrict__ s, int* __restrict__ L, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n && s[i]) L[i] = ufind(L, i); }
#endif
// the same union-find driven from the thinned bit plane: a thread owns a 64-pixel word, returns at once when it is empty (the
// skeleton fills ~1 % of the canvas) and walks its set bits otherwise.  mode 0: init, 1: merge, 2: flatten.
__global__ __launch_bounds__(256) void k_ccl2_bits(const unsigned long long* __restrict__ bits, int* __restrict__ L, int H, int W, int Ww, int mode) {
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (wi >= (size_t)H * Ww) return;
    unsigned long long m = bits[wi];
    if (!m) return;
    const int y = (int)(wi / Ww), xw = (int)(wi % Ww), x0 = xw * 64;
    if (mode != 1) {
        while (m) { const int j = __ffsll((long long)m) - 1; m &= m - 1; const int id = y * W + x0 + j; L[id] = mode == 0 ? id : ufind(L, id); }
        return;
    }
    const unsigned long long cur = m;
    const unsigned long long left = xw > 0 ? bits[wi - 1] : 0ULL;
    unsigned long long U = 0, UL = 0, UR = 0;
    if (y > 0) { U = bits[wi - Ww]; if (xw > 0) UL = bits[wi - Ww - 1]; if (xw + 1 < Ww) UR = bits[wi - Ww + 1]; }
    const unsigned long long hasW = (cur << 1) | (left >> 63), hasNW = (U << 1) | (UL >> 63), hasNE = (U >> 1) | (UR << 63);
    while (m) {
        const int j = __ffsll((long long)m) - 1; m &= m - 1;
        const int id = y * W + x0 + j;
        if ((hasW >> j) & 1ULL) uunite(L, id, id - 1);
        if ((hasNW >> j) & 1ULL) uunite(L, id, id - W - 1);
        if ((U >> j) & 1ULL) uunite(L, id, id - W);
        if ((hasNE >> j) & 1ULL) uunite(L, id, id - W + 1);
    }
}
// ordered compaction of skeleton pixels (count / write), 1024 px per block
__global__ __launch_bounds__(256) void k_sk_count(const u8* __restrict__ s, int64_t n, unsigned* __restrict__ counts) {
    __shared__ unsigned ws[4];
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; unsigned c = 0;
    for (int j = 0; j < 4; j++) if (i + j < n && s[i + j]) c++;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void k_sk_write(const u8* __restrict__ s, const int* __restrict__ L, int64_t n, const unsigned* __restrict__ boff, unsigned* __restrict__ keys, unsigned* __restrict__ lin) {
    __shared__ unsigned ws[4];
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; unsigned f[4], c = 0;
    for (int j = 0; j < 4; j++) { f[j] = (i + j < n && s[i + j]) ? 1u : 0u; c += f[j]; }
    unsigned inc = c; const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned base = boff[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += ws[w];
    unsigned pos = base + inc - c;
    for (int j = 0; j < 4; j++) if (f[j]) { keys[pos] = (unsigned)L[i + j]; lin[pos] = (unsigned)(i + j); pos++; }
}
// the same ordered list from the thinned BIT plane (a word per thread, [Hp][Wwp] words; pixel index on the padded raster = y * Wp + x): 1 bit instead of
// 1 byte per canvas pixel read, and almost every word is empty
__global__ __launch_bounds__(256) void k_sk_count_bits(const unsigned long long* __restrict__ b, size_t nwords, unsigned* __restrict__ counts) {
    __shared__ unsigned ws[4];
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned c = wi < nwords ? (unsigned)__popcll(b[wi]) : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(256) void k_sk_write_bits(const unsigned long long* __restrict__ b, const int* __restrict__ L, size_t nwords, int Wp, int Wwp, const unsigned* __restrict__ boff,
                                                       unsigned* __restrict__ keys, unsigned* __restrict__ lin) {
    __shared__ unsigned ws[4];
    const size_t wi = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long m = wi < nwords ? b[wi] : 0ULL;
    const unsigned c = (unsigned)__popcll(m);
    unsigned inc = c; const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (!m) return;
    unsigned pos = boff[blockIdx.x] + inc - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pos += ws[w];
    const size_t p0 = (wi / Wwp) * (size_t)Wp + (wi % Wwp) * 64;
    while (m) { const int j = __ffsll((long long)m) - 1; m &= m - 1; const size_t p = p0 + j; keys[pos] = (unsigned)L[p]; lin[pos] = (unsigned)p; pos++; }
}
__global__ __launch_bounds__(256) void k_heads2(const unsigned* __restrict__ keys, int64_t m, unsigned* __restrict__ head) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i > m) return;
    head[i] = (i < m && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_comp_starts2(const unsigned* __restrict__ head, const unsigned* __restrict__ hs, int64_t m, unsigned* __restrict__ cs, unsigned nc) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) cs[nc] = (unsigned)m;
    if (i < m && head[i]) cs[hs[i]] = (unsigned)i;
}
// anchors: nearest skeleton pixel of the group to a0 / a1 (first in raster order on ties, 08:428-432)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o, 64); if (t < v) v = t; }
    return v;
}
__global__ __launch_bounds__(256) void k_nearest_anchor(const unsigned* __restrict__ lin, int64_t m, const unsigned* __restrict__ gid, int Wp, GroupInfo* __restrict__ g) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < m;
    unsigned grp = 0; unsigned long long k0 = ~0ULL, k1 = ~0ULL;
    if (valid) {
        unsigned p = lin[i]; int x = (int)(p % Wp) - PAD8, y = (int)(p / Wp) - PAD8;
        grp = gid[p] - 1;
        const GroupInfo* G = g + grp;
        long long d0 = (long long)(y - G->a0y) * (y - G->a0y) + (long long)(x - G->a0x) * (x - G->a0x);
        long long d1 = (long long)(y - G->a1y) * (y - G->a1y) + (long long)(x - G->a1x) * (x - G->a1x);
        k0 = ((unsigned long long)d0 << 27) | p; k1 = ((unsigned long long)d1 << 27) | p;
    }
    // pixels next to each other in raster order mostly share their group: one atomic per (wave, group) instead of one per pixel
    unsigned long long rem = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (rem) {
        int L = __ffsll((long long)rem) - 1;
        unsigned gL = (unsigned)__shfl((int)grp, L, 64);
        bool same = valid && grp == gL;
        unsigned long long m0 = wave_min_u64(same ? k0 : ~0ULL), m1 = wave_min_u64(same ? k1 : ~0ULL);
        if (lane == L) {       // the minima only ever decrease: a stale read can only let a useless atomic through, never drop a winner
            GroupInfo* G = g + gL;
            if (m0 < *(volatile unsigned long long*)&G->near0) atomicMin(&G->near0, m0);
            if (m1 < *(volatile unsigned long long*)&G->near1) atomicMin(&G->near1, m1);
        }
        rem &= ~__ballot(same);
    }
}
// per component: sort key (group rank, ROI-relative block-raster key of its first block)
__global__ __launch_bounds__(128) void k_comp_keys(const unsigned* __restrict__ cs, unsigned nc, const unsigned* __restrict__ lin, const unsigned* __restrict__ gid, int Wp,
                                                    const GroupInfo* __restrict__ g, unsigned long long* __restrict__ ckey, unsigned* __restrict__ cidx) {
    unsigned c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= nc) return;
    unsigned b = cs[c], e = cs[c + 1];
    const GroupInfo* G = g + (gid[lin[b]] - 1);
    int w = max(1, G->x1 - G->x0); int wb = (w + 1) >> 1;
    unsigned best = 0xffffffffu;
    for (unsigned q = b; q < e; q++) {
        unsigned p = lin[q]; int x = (int)(p % Wp) - PAD8 - G->x0, y = (int)(p / Wp) - PAD8 - G->y0;
        unsigned k = (unsigned)((y >> 1) * wb + (x >> 1));
        best = min(best, k);
    }
    ckey[c] = ((unsigned long long)(unsigned)G->rank << 32) | best; cidx[c] = c;
}

__device__ const int OFY[8] = {-1, -1, -1, 0, 1, 1, 1, 0};     // _OFFS (dy,dx), 08:252
__device__ const int OFX[8] = {-1, 0, 1, 1, 1, 0, -1, -1};

// ---- _component_best_path (08:295-317) + resample + RDP (08:444-463), one wavefront per skeleton component ----
// Components are contiguous ranges [cs[c], cs[c+1]) of the raster-ordered pixel list `lin`; a pixel's position in that list is its
// compact id (cid canvas), its index inside the range its local id.  nbr[q*8+k] = compact id of the k-th _OFFS neighbour (or ~0).
// The BFS keeps the reference's FIFO order exactly: the queue is consumed eight nodes (64 (node, direction) pairs) at a time, a pixel
// reached by several pairs of one chunk goes to the lowest pair, and winners are appended in pair order.
__global__ __launch_bounds__(256) void k_cid_fill(const unsigned* __restrict__ lin, unsigned m, unsigned* __restrict__ cid) {
    unsigned q = blockIdx.x * 256 + threadIdx.x; if (q < m) cid[lin[q]] = q;
}
__global__ __launch_bounds__(256) void k_nbr_build(const unsigned* __restrict__ lin, unsigned m, const u8* __restrict__ sk, const unsigned* __restrict__ cid, int Wp, int Hp, unsigned* __restrict__ nbr) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; if (t >= (size_t)m * 8) return;
    unsigned q = (unsigned)(t >> 3); int k = (int)(t & 7);
    unsigned p = lin[q]; int y = (int)(p / Wp) + OFY[k], x = (int)(p % Wp) + OFX[k];
    unsigned v = ~0u;
    if (y >= 0 && y < Hp && x >= 0 && x < Wp) { size_t j = (size_t)y * Wp + x; if (sk[j]) v = cid[j]; }
    nbr[t] = v;
}
// class lists: 0 = fits the small LDS layout, 1 = the large one, 2 = global scratch
__global__ __launch_bounds__(256) void k_comp_classes(const unsigned* __restrict__ corder, unsigned nc, const unsigned* __restrict__ cs, unsigned cap0, unsigned cap1, int need,
                                                      unsigned* __restrict__ counts, unsigned* __restrict__ l0, unsigned* __restrict__ l1, unsigned* __restrict__ l2, unsigned* __restrict__ outcnt) {
    unsigned oi = blockIdx.x * 256 + threadIdx.x; if (oi >= nc) return;
    unsigned c = corder[oi]; unsigned s = cs[c + 1] - cs[c];
    outcnt[oi] = 0;
    if ((int)s < need) return;                              // a path cannot be longer than its component
    if (s <= cap0) l0[atomicAdd(&counts[0], 1u)] = oi;
    else if (s <= cap1) l1[atomicAdd(&counts[1], 1u)] = oi;
    else l2[atomicAdd(&counts[2], 1u)] = oi;
}

template <bool LDSV> struct CompWork;
template <> struct CompWork<true> {
    typedef uint16_t Id; typedef ushort2 Stk;
    static constexpr unsigned NONE = 0xffffu;
    Id* nb; Id* prev; Id* que; float* cum; u8* seen; float2* P; Stk* stk; u8* keep;
    __device__ __forceinline__ unsigned nbr_of(unsigned u, int k) const { return nb[u * 8 + k]; }
};
template <> struct CompWork<false> {
    typedef uint32_t Id; typedef int2 Stk;
    static constexpr unsigned NONE = 0xffffffffu;
    const unsigned* nbr; unsigned b;
    Id* prev; Id* que; float* cum; u8* seen; float2* P; Stk* stk; u8* keep;
    __device__ __forceinline__ unsigned nbr_of(unsigned u, int k) const { unsigned v = nbr[(size_t)(b + u) * 8 + k]; return v == ~0u ? NONE : v - b; }
};
// FIFO BFS from src over local ids; stops when goal is dequeued (goal == NONE: full sweep).  Returns the last dequeued node.
template <class WK> __device__ unsigned bfs_wave(WK& w, unsigned src, unsigned goal, u8 stamp, int lane) {
    if (lane == 0) { w.que[0] = (typename WK::Id)src; w.seen[src] = stamp; w.prev[src] = (typename WK::Id)WK::NONE; }
    __syncthreads();
    unsigned head = 0, tail = 1;
    const int slot = lane >> 3, dir = lane & 7;
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (head < tail) {
        unsigned nn = min(8u, tail - head);
        unsigned u = (unsigned)slot < nn ? (unsigned)w.que[head + slot] : WK::NONE;
        bool hit = false;
        if (goal != WK::NONE) {
            unsigned long long gm = __ballot((unsigned)slot < nn && u == goal);
            if (gm) { nn = (unsigned)((__ffsll((long long)gm) - 1) >> 3); hit = true; }
        }
        bool act = (unsigned)slot < nn;
        unsigned v = act ? w.nbr_of(u, dir) : WK::NONE;
        bool nw = act && v != WK::NONE && w.seen[v] != stamp;
        unsigned long long cand = __ballot(nw), win = 0;
        while (cand) {
            int L = __ffsll((long long)cand) - 1;
            unsigned vL = (unsigned)__shfl((int)v, L, 64);
            unsigned long long dup = __ballot(nw && v == vL);
            win |= 1ull << L; cand &= ~dup;
        }
        if ((win >> lane) & 1ull) {
            unsigned pos = tail + (unsigned)__popcll(win & lt);
            w.que[pos] = (typename WK::Id)v; w.seen[v] = stamp; w.prev[v] = (typename WK::Id)u;
        }
        tail += (unsigned)__popcll(win);
        head += nn;
        __syncthreads();
        if (hit) return goal;
    }
    return (unsigned)w.que[tail - 1];
}

template <bool LDSV>
__device__ void comp_path_wave(CompWork<LDSV>& w, unsigned oi, unsigned b, unsigned S, unsigned a0c, unsigned a1c, const unsigned* __restrict__ lin, int Wp,
                               int min_len, double step, float eps, unsigned pcap, int2* __restrict__ outpts, unsigned* __restrict__ outcnt, int lane) {
    typedef CompWork<LDSV> WK;
    const unsigned NONE = WK::NONE;
    const unsigned e = b + S;
    const bool ha = a0c >= b && a0c < e, hb = a1c >= b && a1c < e;       // "comp[a0]" (08:300): the anchor pixel lies in this component
    const unsigned a0 = a0c - b, a1 = a1c - b;
    const int need = max(2, min_len);
    int plen = 0; unsigned pv = NONE;
    if (ha && hb) {
        if (a0 == a1) plen = 1;
        else {
            bfs_wave(w, a0, a1, 1, lane);
            if (w.seen[a1] == 1) pv = a1;
        }
    }
    // length of the prev-chain ending in pv, written backwards into the tail of the queue buffer (which the path then occupies)
    auto backtrack = [&](unsigned endn) -> int {
        int cnt = 0;
        if (lane == 0) { unsigned p = endn; unsigned pos = S; while (p != NONE) { w.que[--pos] = (typename WK::Id)p; p = (unsigned)w.prev[p]; cnt++; } }
        cnt = __shfl(cnt, 0, 64);
        __syncthreads();
        return cnt;
    };
    if (pv != NONE) { plen = backtrack(pv); }
    if (plen < need) plen = 0;
    if (plen == 0) {
        unsigned u = bfs_wave(w, 0u, NONE, 2, lane);                     // seed = first pixel in raster order (08:306)
        unsigned v = bfs_wave(w, u, NONE, 3, lane);
        plen = (u == v) ? 1 : backtrack(v);                              // the sweep from u is _bfs_path's own search, cut at v
        if (plen < need) plen = 0;
    }
    if (plen < 2) return;
    const typename WK::Id* path = w.que + (S - plen);
    auto PX = [&](int k) -> float { return (float)((int)(lin[b + path[k]] % (unsigned)Wp) - PAD8); };
    auto PY = [&](int k) -> float { return (float)((int)(lin[b + path[k]] / (unsigned)Wp) - PAD8); };
    // float32 segment lengths in parallel, then the sequential float32 cumsum (08:444-446)
    for (int k = 1 + lane; k < plen; k += 64) { float dx = PX(k) - PX(k - 1), dy = PY(k) - PY(k - 1); w.cum[k] = sqrtf(dx * dx + dy * dy); }
    __syncthreads();
    if (lane == 0) { float acc = w.cum[1]; w.cum[0] = 0.f; for (int k = 2; k < plen; k++) { acc = acc + w.cum[k]; w.cum[k] = acc; } }
    __syncthreads();
    const float total = w.cum[plen - 1];
    int m;
    if ((double)total <= step) {
        m = plen;
        if ((unsigned)m > pcap) return;      // cannot happen: pcap >= step + 2
        for (int k = lane; k < plen; k += 64) w.P[k] = make_float2(PX(k), PY(k));
    } else {
        m = (int)ceil((double)total / step);
        if ((unsigned)m > pcap) return;      // cannot happen: pcap >= sqrt(2) S / step + 2
        const float t0 = 0.0f, t1 = (float)(0.0 + step), delta = t1 - t0;
        for (int i = lane; i < m; i += 64) {
            float tf = i == 0 ? t0 : (i == 1 ? t1 : t0 + (float)i * delta);
            double t = (double)tf;
            int lo = 0, hi = plen - 2;                                   // k = #{ j in [1, plen-2] : s[j] <= t }  (searchsorted right - 1, clipped)
            while (lo < hi) { int mid = (lo + hi + 1) >> 1; if ((double)w.cum[mid] <= t) lo = mid; else hi = mid - 1; }
            int k = lo;
            double sk = (double)w.cum[k], sk1 = (double)w.cum[k + 1];
            float ax = PX(k), ay = PY(k), bx = PX(k + 1), by = PY(k + 1);
            double u = (t - sk) / fmax(1e-6, sk1 - sk);
            double a = 1.0 - u;
            w.P[i] = make_float2((float)((double)ax * a + (double)bx * u), (float)((double)ay * a + (double)by * u));
        }
    }
    if (m < 2) return;
    // RDP, explicit LIFO stack (08:453-462); the farthest point of a span is found 64 points at a time
    for (int i = lane; i < m; i += 64) w.keep[i] = (i == 0 || i == m - 1) ? 1 : 0;
    int sp = 0;
    if (lane == 0) { w.stk[0].x = 0; w.stk[0].y = (decltype(w.stk[0].y))(m - 1); }
    sp = 1;
    __syncthreads();
    while (sp > 0) {
        --sp;
        const int s = (int)w.stk[sp].x, en = (int)w.stk[sp].y;
        __syncthreads();
        if (en <= s + 1) continue;
        float ax = w.P[s].x, ay = w.P[s].y, bx = w.P[en].x, by = w.P[en].y;
        float segx = bx - ax, segy = by - ay, nx = -segy, ny = segx;
        float q = segx * segx + segy * segy;
        double seg_len = (double)sqrtf(q) + 1e-12; float seg_len_f = (float)seg_len;
        float bestd = -1.f; int bi = 0x7fffffff;
        for (int i = s + 1 + lane; i < en; i += 64) {
            float dx = w.P[i].x - ax, dy = w.P[i].y - ay;
            float t0 = dx * nx, t1 = dy * ny;
            float d = fabsf(t0 + t1) / seg_len_f;
            if (d > bestd) { bestd = d; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            float od = __shfl_xor(bestd, o, 64); int oi2 = __shfl_xor(bi, o, 64);
            if (od > bestd || (od == bestd && oi2 < bi)) { bestd = od; bi = oi2; }
        }
        if (bestd > eps) {
            if (lane == 0) {
                w.keep[bi] = 1;
                w.stk[sp].x = (decltype(w.stk[0].x))s; w.stk[sp].y = (decltype(w.stk[0].y))bi;
                w.stk[sp + 1].x = (decltype(w.stk[0].x))bi; w.stk[sp + 1].y = (decltype(w.stk[0].y))en;
            }
            sp += 2;
        }
        __syncthreads();
    }
    int2* o = outpts + b; unsigned cnt = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < m; i0 += 64) {
        int i = i0 + lane; bool kp = i < m && w.keep[i];
        unsigned long long bm = __ballot(kp);
        if (kp) { float2 p = w.P[i]; o[cnt + (unsigned)__popcll(bm & lt)] = make_int2((int)p.x, (int)p.y); }
        cnt += (unsigned)__popcll(bm);
    }
    if (lane == 0) outcnt[oi] = cnt;
}

struct CompArgs {
    const unsigned* corder; const unsigned* cs; const unsigned* lin; const unsigned* gid; const GroupInfo* g; const unsigned* cid; const unsigned* nbr;
    int Wp; int min_len; double step; float eps; int2* outpts; unsigned* outcnt;
};
__device__ __forceinline__ void comp_anchors(const CompArgs& A, unsigned b, unsigned& a0c, unsigned& a1c) {
    const GroupInfo* G = A.g + (A.gid[A.lin[b]] - 1);
    a0c = (G->near0 == ~0ULL) ? ~0u : A.cid[(unsigned)(G->near0 & ((1ULL << 27) - 1))];
    a1c = (G->near1 == ~0ULL) ? ~0u : A.cid[(unsigned)(G->near1 & ((1ULL << 27) - 1))];
}
// LDS-resident components (cap nodes, pcap resample points per block)
__global__ __launch_bounds__(64) void k_comp_paths_lds(CompArgs A, const unsigned* __restrict__ list, const unsigned* __restrict__ count, unsigned cap, unsigned pcap) {
    extern __shared__ __align__(16) unsigned char smem[];
    CompWork<true> w;
    w.cum = reinterpret_cast<float*>(smem);
    w.P = reinterpret_cast<float2*>(w.cum + cap);
    w.stk = reinterpret_cast<ushort2*>(w.P + pcap);
    w.nb = reinterpret_cast<uint16_t*>(w.stk + pcap);
    w.prev = w.nb + (size_t)cap * 8; w.que = w.prev + cap;
    w.seen = reinterpret_cast<u8*>(w.que + cap); w.keep = w.seen + cap;
    const int lane = threadIdx.x;
    const unsigned n = *count;
    for (unsigned li = blockIdx.x; li < n; li += gridDim.x) {
        unsigned oi = list[li]; unsigned c = A.corder[oi]; unsigned b = A.cs[c], S = A.cs[c + 1] - b;
        for (unsigned t = lane; t < S * 8; t += 64) { unsigned v = A.nbr[(size_t)b * 8 + t]; w.nb[t] = v == ~0u ? (uint16_t)0xffffu : (uint16_t)(v - b); }
        for (unsigned t = lane; t < S; t += 64) w.seen[t] = 0;
        __syncthreads();
        unsigned a0c, a1c; comp_anchors(A, b, a0c, a1c);
        comp_path_wave<true>(w, oi, b, S, a0c, a1c, A.lin, A.Wp, A.min_len, A.step, A.eps, pcap, A.outpts, A.outcnt, lane);
        __syncthreads();
    }
}
// components too large for LDS: same code over per-pixel scratch in global memory
struct CompScratch { unsigned* prev; unsigned* que; float* cum; u8* seen; float2* P; int2* stk; u8* keep; };
__global__ __launch_bounds__(64) void k_comp_paths_glb(CompArgs A, const unsigned* __restrict__ list, const unsigned* __restrict__ count, CompScratch X) {
    const int lane = threadIdx.x;
    const unsigned n = *count;
    for (unsigned li = blockIdx.x; li < n; li += gridDim.x) {
        unsigned oi = list[li]; unsigned c = A.corder[oi]; unsigned b = A.cs[c], S = A.cs[c + 1] - b;
        CompWork<false> w; w.nbr = A.nbr; w.b = b;
        w.prev = X.prev + b; w.que = X.que + b; w.cum = X.cum + b; w.seen = X.seen + b; w.P = X.P + b; w.stk = X.stk + b; w.keep = X.keep + b;
        for (unsigned t = lane; t < S; t += 64) w.seen[t] = 0;
        __syncthreads();
        unsigned a0c, a1c; comp_anchors(A, b, a0c, a1c);
        comp_path_wave<false>(w, oi, b, S, a0c, a1c, A.lin, A.Wp, A.min_len, A.step, A.eps, S, A.outpts, A.outcnt, lane);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_path_desc(const unsigned* __restrict__ corder, const unsigned* __restrict__ cs, const unsigned* __restrict__ outcnt, const unsigned* __restrict__ flag,
                                                    const unsigned* __restrict__ scan, unsigned nc, GatherDesc* __restrict__ d) {
    unsigned oi = blockIdx.x * 256 + threadIdx.x; if (oi >= nc || !flag[oi]) return;
    GatherDesc g; g.begin = cs[corder[oi]]; g.len = outcnt[oi]; g.rev = 0; g.src = 0;
    d[scan[oi]] = g;
}
__global__ __launch_bounds__(256) void k_flag_nonzero(const unsigned* __restrict__ v, unsigned n, unsigned* __restrict__ f) {
    unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n) f[i] = v[i] >= 2 ? 1u : 0u; if (i == n) f[i] = 0;
}
__global__ __launch_bounds__(256) void k_concat_taps(const int2* a, int64_t na, const int2* b, int64_t nb, int2* out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < na) out[i] = a[i]; else if (i < na + nb) out[i] = b[i - na];
}

// ================================================================= prefetch of the order-independent part of the front (under stage 07's greedy)
// Stage 07 only permutes and flips the scaled contours (07:55-95), and it does so with a serial chain of greedy steps that keeps one
// wavefront busy for milliseconds.  What stage 08 computes PER POLYLINE before anything depends on the order -- bounding box and numpy
// perimeter of the opened polyline (A0 / A1), its float32 cumulative lengths and sample count (A2) -- depends on the direction the
// polyline is read in, nothing else.  So both directions are computed on the lane's side stream while the chain runs, and stage 08
// picks per polyline by stage 07's flip flag.  (Closed contours are never flipped, 07:60-62: their reversed entries are unused.)
__global__ __launch_bounds__(256) void k_pf_views(const PolyFeat* __restrict__ feat07, const int64_t* __restrict__ off, int64_t n, VView* __restrict__ vf, VView* __restrict__ vr,
                                                   int64_t* __restrict__ lf, int64_t* __restrict__ lr) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == n) { lf[i] = 0; lr[i] = 0; }
    if (i >= n) return;
    const unsigned len = (unsigned)(off[i + 1] - off[i]);
    VView a; a.wid = (unsigned)i; a.first = 0u; a.len = (feat07[i].closed && len > 0u) ? len - 1u : len; a.rev = 0u;     // _ensure_open (08:48-51), as split_small leaves the kept polylines
    VView b; b.wid = (unsigned)i; b.first = 0u; b.len = len; b.rev = 1u;
    vf[i] = a; vr[i] = b; lf[i] = a.len; lr[i] = b.len;
}
__global__ __launch_bounds__(256) void k_pf_pick_feat(const VView* __restrict__ sview, int64_t n, const PolyFeat* __restrict__ pf, const float* __restrict__ per_rev, PolyFeat* __restrict__ out) {
    int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const VView v = sview[k];
    PolyFeat f = pf[v.wid];
    if (v.rev) { const int32_t ax = f.sx, ay = f.sy; f.sx = f.ex; f.sy = f.ey; f.ex = ax; f.ey = ay; f.per = per_rev[v.wid]; }      // the reversed polyline: same box, same points, ends swapped, its own pairwise sum
    out[k] = f;
}
__global__ __launch_bounds__(256) void k_pf_pick_info(const VView* __restrict__ kview, int64_t nk, const RsInfo* __restrict__ pinfo, int64_t npf, const int64_t* __restrict__ off_f,
                                                       const int64_t* __restrict__ off_r, int64_t tot_f, RsInfo* __restrict__ info, int64_t* __restrict__ cumoff) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nk) return;
    const VView v = kview[j];
    info[j] = pinfo[v.rev ? npf + (int64_t)v.wid : (int64_t)v.wid];
    cumoff[j] = v.rev ? tot_f + off_r[v.wid] : off_f[v.wid];       // (both readings of a polyline sit at its offset in the scaled list)
}
struct StreamSwap {       // everything issued while this lives goes to the lane's side stream
    LaneRes& l;
    explicit StreamSwap(LaneRes& lane) : l(lane) { std::swap(l.stream, l.stream2); }
    ~StreamSwap() { std::swap(l.stream, l.stream2); }
};
static std::atomic<uint64_t> g_pf_tag{1};
static int prefetch08(orip_ctx* c, const orip_params08& P, DPolys& S, const PolyFeat* feat07) {
    LaneRes::Prefetch08& F = LN(c).pf08;
    F.valid = false;
    const int64_t n = S.n, total = S.total;
    if (n <= 0 || total <= 0 || total > 0x3fffffff) return 0;
    const double step = std::max(1.0, P.sample_step);
    HIPC(c, F.feat.ensure((size_t)n * (sizeof(PolyFeat) + 4) + 64));
    HIPC(c, F.info.ensure((size_t)2 * n * sizeof(RsInfo) + 64));
    HIPC(c, F.cum.ensure((size_t)2 * total * 4 + 64));
    HIPC(c, F.ord.ensure((size_t)n * 16 + 64));
    HIPC(c, F.seg.ensure((size_t)total * 4 + 64));
    {
        StreamSwap sw(LN(c));                       // LN(c).stream is the side stream from here to the end of the block
        HIPC(c, hipStreamWaitEvent(LN(c).stream, LN(c).ev2, 0));      // stage 07's features (feat07) and its use of the shared scratch end here (vreorder)
        PolyFeat* ff = F.feat.as<PolyFeat>(); float* per_rev = reinterpret_cast<float*>(ff + n); RsInfo* inf = F.info.as<RsInfo>(); float* cum = F.cum.as<float>();
        VSrc sS; ORIP_TRY(vsrc_of(c, S, sS));
        // per-polyline fields first (open view, end points; bounding box and perimeters of the short ones): one thread per polyline
        hipLaunchKernelGGL(k_poly_features<VSrc>, dim3(cdiv(n, 128)), dim3(128), 0, LN(c).stream, sS, n, 1 | 16 | 32, ff, per_rev);
        // A2 (the long polylines): cumulative lengths of both readings, longest first.  k_seglen fetches the points (once each) and leaves every segment's
        // float32 length in F.seg and the open view's bounding box in ff; both readings and the perimeter sums (A0 / A1, forwards and backwards) then
        // read 4 bytes per segment instead of turning (polyline, index) into a point again.
        unsigned* kin = F.ord.as<unsigned>(); unsigned* kout = kin + n; unsigned* vin = kout + n; unsigned* ordl = vin + n;
        float* seg = F.seg.as<float>();
        hipLaunchKernelGGL(k_len_keys, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, S.off.as<int64_t>(), n, kin, vin);
        ORIP_TRY((vsort_pairs<unsigned, unsigned>(c, kin, kout, vin, ordl, (size_t)n, 0, 32, true)));
        // (the sort borrows the lane's scan / sort scratch: the main stream, which sits in the greedy chain for milliseconds yet, takes it back behind this point)
        HIPC(c, hipEventRecord(LN(c).ev4, LN(c).stream));
        HIPC(c, hipStreamWaitEvent(LN(c).stream2 /* the main stream while the swap lives */, LN(c).ev4, 0));
        // What stage 08 asks for first (split_small: boxes and perimeters) goes first and gets an event of its own (ev4); the cumulative lengths, which A2 picks
        // up a dozen launches and a host read later, follow (ev3).
        if (total > ORIP_LONG_CUM) {
            { ProfScope ps(c, "k_seglen"); hipLaunchKernelGGL(k_seglen<VSrc>, dim3((unsigned)cdiv(total, 4 * 63 * 64)), dim3(256), 0, LN(c).stream, sS, n, total, seg, ff); }
            if (total > ORIP_LONG_POLY) { ProfScope ps(c, "k_poly_features_long");
                const size_t nleaf = (size_t)(total >> 6) + 2 * (size_t)n + 8;
                HIPC(c, LN(c).vtmp[11].ensure(nleaf * sizeof(float) * 2 + 64));
                float* leafbuf = LN(c).vtmp[11].as<float>(); float* leafbuf_rev = leafbuf + nleaf;
                hipLaunchKernelGGL(k_perim_leaves_seg, dim3((unsigned)cdiv((int64_t)nleaf * 8, 256)), dim3(256), 0, LN(c).stream, sS.off, n, ff, seg, leafbuf, leafbuf_rev, (int64_t)nleaf);
                hipLaunchKernelGGL((k_poly_features_long<VSrc, true>), dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(256), 0, LN(c).stream, sS, n, 1 | 16 | 32 | 64, ff, leafbuf, ordl, per_rev, leafbuf_rev, seg); }
        }
        HIPC(c, hipEventRecord(LN(c).ev4, LN(c).stream));
        { ProfScope ps(c, "k_cumlen"); hipLaunchKernelGGL(k_cumlen2<VSrc>, dim3(cdiv(2 * n, 128)), dim3(128), 0, LN(c).stream, sS, feat07, n, step, cum, total, inf); }
        if (total > ORIP_LONG_CUM) { ProfScope ps(c, "k_cumlen_long"); const dim3 grid((unsigned)std::min<int64_t>(n, 8192), 2);       // both readings side by side
#ifdef ORIP_VARIANTS
            if (ORIP_VARIANT("ORIP_CUM_CHAIN")) hipLaunchKernelGGL((k_cumlen_long2<VSrc, true>), grid, dim3(64), 0, LN(c).stream, sS, n, step, cum, total, inf, ordl, 0, (const float*)seg);
            else
#endif
            hipLaunchKernelGGL((k_cumlen_long2<VSrc, false>), grid, dim3(64), 0, LN(c).stream, sS, n, step, cum, total, inf, ordl, 0, (const float*)seg); }
        HIPC(c, hipGetLastError());
        HIPC(c, hipEventRecord(LN(c).ev3, LN(c).stream));
    }
    F.pending = true;             // nobody has waited yet: split_small (ev4), A2 (ev3), or the next call on the lane (orip_pf08_drain)
    F.valid = true; F.tag = g_pf_tag.fetch_add(1); F.n = n; F.tot_f = total; F.step = step; F.src_off = S.off.as<int64_t>();
    return 0;
}

// split_small_and_taps on a DPolys -> kept (opened) + taps appended to tapbuf at tap_base
__global__ __launch_bounds__(256) void k_compact_feat(const unsigned* __restrict__ flag, const unsigned* __restrict__ scan, int64_t n, const PolyFeat* __restrict__ in, PolyFeat* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && flag[i]) out[scan[i]] = in[i];
}
// kept_feat (optional, room for src.n entries): features of the kept polylines' open views (bbox + numpy perimeter), so the caller
// does not have to read the points again
int split_small(orip_ctx* c, DPolys& src, const orip_params08& P, DPolys& kept, DBuf& tapbuf, int64_t tap_base, int64_t* n_taps_out, PolyFeat* kept_feat = nullptr) {
    kept.n = 0; kept.total = 0; kept.set_explicit(); *n_taps_out = 0;
    HIPC(c, kept.off.ensure(64)); HIPC(c, hipMemsetAsync(kept.off.p, 0, 8, LN(c).stream));
    int64_t n = src.n;
    if (n == 0) return 0;
    HIPC(c, LN(c).vtmp[2].ensure((size_t)(n + 1) * (16 + 8 + 2 * sizeof(GatherDesc)) + 256));
    unsigned* is_tap = LN(c).vtmp[2].as<unsigned>(); unsigned* is_keep = is_tap + (n + 1); unsigned* tap_scan = is_keep + (n + 1); unsigned* keep_scan = tap_scan + (n + 1);
    int2* tap_xy = (int2*)(keep_scan + (n + 1)); GatherDesc* kd = (GatherDesc*)(tap_xy + (n + 1)); GatherDesc* kd2 = kd + (n + 1);
    HIPC(c, LN(c).vtmp[10].ensure((size_t)n * sizeof(PolyFeat) + 64));
    PolyFeat* sfeat = LN(c).vtmp[10].as<PolyFeat>();
    if (is_coded(src) && P.tap_max_v > 64) ORIP_TRY(orip_polys_materialize(c, src));      // the walk-coded tap test copies <= 64 vertices (default tap_max_vertices: 50)
    LaneRes::Prefetch08& F = LN(c).pf08;
    if (kept_feat && is_coded(src) && src.pf_tag && F.valid && F.tag == src.pf_tag && !src.vident) {      // computed under stage 07's greedy, per walk and direction
        if (F.pending) HIPC(c, hipStreamWaitEvent(LN(c).stream, LN(c).ev4, 0));
        hipLaunchKernelGGL(k_pf_pick_feat, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, src.vview.as<VView>(), n, F.feat.as<PolyFeat>(), reinterpret_cast<const float*>(F.feat.as<PolyFeat>() + F.n), sfeat);
    } else { HIPC(c, orip_pf08_drain(c)); ORIP_TRY(vfeatures(c, src, kept_feat ? (1 | 16) : 0, sfeat)); }      // (the prefetch shares vfeatures' scratch)
    { ProfScope ps(c, "k_split_small08"); ORIP_WITH_SRC(c, src, sv, { hipLaunchKernelGGL(k_split_small08<decltype(sv)>, dim3(cdiv(n + 1, 128)), dim3(128), 0, LN(c).stream, sv, n, P, sfeat, is_tap, is_keep, tap_xy, kd); }); }
    ORIP_TRY(vscan_excl<unsigned>(c, is_tap, tap_scan, (size_t)n + 1));
    ORIP_TRY(vscan_excl<unsigned>(c, is_keep, keep_scan, (size_t)n + 1));
    unsigned nt = 0, nk = 0;
    HIPC(c, hipMemcpyAsync(&nt, tap_scan + n, 4, hipMemcpyDeviceToHost, LN(c).stream));      // both counts, one wait
    ORIP_TRY(vread(c, &nk, keep_scan + n));
    if (nt) {
        HIPC(c, tapbuf.ensure((size_t)(tap_base + nt) * 8 + 64, LN(c).stream, true));
        hipLaunchKernelGGL(k_compact_desc, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, is_tap, tap_scan, n, (const GatherDesc*)nullptr, (GatherDesc*)nullptr, tap_xy, tapbuf.as<int2>() + tap_base);
    }
    *n_taps_out = nt;
    if (nk) {
        hipLaunchKernelGGL(k_compact_desc, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, is_keep, keep_scan, n, kd, kd2, (const int2*)nullptr, (int2*)nullptr);
        ORIP_TRY(vgather_list(c, kd2, nk, src, kept));
        if (kept_feat) hipLaunchKernelGGL(k_compact_feat, dim3(cdiv(n, 256)), dim3(256), 0, LN(c).stream, is_keep, keep_scan, n, sfeat, kept_feat);
    }
    HIPC(c, hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void k_fill_per(const PolyFeat* __restrict__ f, int64_t n, float* __restrict__ k, unsigned* __restrict__ v) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { k[i] = f[i].per; v[i] = (unsigned)i; }
}


}  // namespace

int orip_prefetch08(orip_ctx* c, void* prm, DPolys& scaled, const void* feat07) {
    return prefetch08(c, *static_cast<const orip_params08*>(prm), scaled, static_cast<const PolyFeat*>(feat07));
}

extern "C" int orip_dedup_layer(orip_ctx* c, int layer, const orip_params08* prm) {
    orip_enter(c);
    if (!prm || layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad arguments");
    const orip_params08 P = *prm;
    const int W = P.W, H = P.H;
    ORIP_LANE_NODRAIN(c, layer + 1);       // stage 08 waits for the parts of its prefetch where it picks them up (split_small, A2)
    if (W <= 0 || H <= 0 || W > 16383 || H > 16383) ORIP_FAIL(c, "canvas %dx%d out of range", W, H);
    if (!(P.sample_step * 2.0 < P.max_jump)) ORIP_FAIL(c, "dedup_sample_step must be < max_join_jump_px / 2 (stage-A segments are assumed jump-free)");
    DPolys& S = c->polys[ORIP_SLOT_SORTED][layer]; DPolys& OUT = c->polys[ORIP_SLOT_LINES_INTRA][layer]; DTaps& TOUT = c->taps[ORIP_TAPS_INTRA][layer];
    OUT.n = 0; OUT.total = 0; OUT.set_explicit(); TOUT.n = 0;
    HIPC(c, OUT.off.ensure(64)); HIPC(c, hipMemsetAsync(OUT.off.p, 0, 8, LN(c).stream));
    HIPC(c, TOUT.xy.ensure(64));
    if (S.n == 0) return 0;
    struct Ref { DPolys& p; }; Ref kept0{LN(c).tp[0]}, cleaned{LN(c).tp[1]}, lines2{LN(c).tp[2]}, merged{LN(c).tp[3]};
    for (Ref* r : {&kept0, &cleaned, &lines2, &merged}) { r->p.n = 0; r->p.total = 0; r->p.set_explicit(); }
    int64_t nt0 = 0, nt2 = 0;
    bool caps_counted = false;
    const bool tdbg = getenv("ORIP_TIME08") != nullptr;      // debug: per-phase wall times of this layer (adds stream syncs)
    std::string tlog; auto tprev = std::chrono::steady_clock::now();
    auto tick = [&](const char* name) {
        if (!tdbg) return;
        hipStreamSynchronize(LN(c).stream);
        auto t = std::chrono::steady_clock::now(); char b[64];
        snprintf(b, sizeof b, " %s %.2f", name, std::chrono::duration<double, std::milli>(t - tprev).count()); tlog += b; tprev = t;
    };
    // ---- A0
    HIPC(c, LN(c).vtmp[6].ensure((size_t)S.n * sizeof(PolyFeat) + 64));
    PolyFeat* feat = LN(c).vtmp[6].as<PolyFeat>();       // open-view features of the kept polylines (perimeter: A1)
    ORIP_TRY(split_small(c, S, P, kept0.p, TOUT.xy, 0, &nt0, feat));
    const int64_t nk = kept0.p.n;
    tick("split");
    if (nk > 0) {
        if (kept0.p.total > 0x7fffffff) ORIP_FAIL(c, "layer too large");
        // ---- A1: order by perimeter, descending, stable
        tick("feat");
        HIPC(c, LN(c).vtmp[0].ensure((size_t)nk * 16 + (size_t)(nk + 1) * 8 + (size_t)nk * sizeof(RsInfo) + 256));
        float* kin = LN(c).vtmp[0].as<float>(); float* kout = kin + nk; unsigned* vin = (unsigned*)(kout + nk); unsigned* ord = vin + nk;
        unsigned* mr = ord + nk; unsigned* sbase = mr + (nk + 1); RsInfo* info = (RsInfo*)(sbase + (nk + 1) + 2);   // (6 nk + 4) dwords: 8-byte aligned
        hipLaunchKernelGGL(k_fill_per, dim3(cdiv(nk, 256)), dim3(256), 0, LN(c).stream, feat, nk, kin, vin);
        ORIP_TRY((vsort_pairs<float, unsigned>(c, kin, kout, vin, ord, (size_t)nk, 0, 32, true)));
        tick("A0-1");
        // ---- A2: resample
        const double step = std::max(1.0, P.sample_step);
        const LaneRes::Prefetch08& F = LN(c).pf08;
        const bool picked = is_coded(kept0.p) && kept0.p.pf_tag && F.valid && F.tag == kept0.p.pf_tag && F.step == step && !kept0.p.vident;
        HIPC(c, orip_pf08_drain(c));          // the cumulative lengths of the prefetch (ev3), if nobody has waited for them yet
        HIPC(c, LN(c).vtmp[1].ensure((picked ? 0 : (size_t)kept0.p.total * 4) + (size_t)(nk + 1) * 8 + 128));
        int64_t* cumoff = LN(c).vtmp[1].as<int64_t>(); float* cum = reinterpret_cast<float*>(cumoff + (nk + 2));
        if (picked) {       // cumulative lengths and sample counts were taken under stage 07's greedy, per walk and direction: pick this list's
            cum = F.cum.as<float>();
            hipLaunchKernelGGL(k_pf_pick_info, dim3(cdiv(nk, 256)), dim3(256), 0, LN(c).stream, kept0.p.vview.as<VView>(), nk, F.info.as<RsInfo>(), F.n, F.src_off, F.src_off, F.tot_f, info, cumoff);
        } else {
        HIPC(c, hipMemcpyAsync(cumoff, kept0.p.off.p, (size_t)(nk + 1) * 8, hipMemcpyDeviceToDevice, LN(c).stream));
        { ProfScope ps(c, "k_cumlen"); ORIP_WITH_SRC(c, kept0.p, sv, { hipLaunchKernelGGL(k_cumlen<decltype(sv)>, dim3(cdiv(nk, 128)), dim3(128), 0, LN(c).stream, sv, nk, step, cum, info); }); }
        if (kept0.p.total > ORIP_LONG_CUM) { ProfScope ps(c, "k_cumlen_long"); ORIP_WITH_SRC(c, kept0.p, sv, {
                bool chain = false;
                ORIP_CUM_CHAIN_LAUNCH(decltype(sv))
                if (!chain) hipLaunchKernelGGL((k_cumlen_long2<decltype(sv), false>), dim3((unsigned)std::min<int64_t>(nk, 8192), 1), dim3(64), 0, LN(c).stream, sv, nk, step, cum, (int64_t)0, info, ord, 0, (const float*)nullptr); }); }
        }
        tick("cumlen");
        HIPC(c, hipMemsetAsync(sbase + nk + 1, 0, 4, LN(c).stream));
        hipLaunchKernelGGL(k_rank_counts, dim3(cdiv(nk + 1, 256)), dim3(256), 0, LN(c).stream, info, ord, nk, mr, feat, W, H, sbase + nk + 1);
        ORIP_TRY(vscan_excl<unsigned>(c, mr, sbase, (size_t)nk + 1));
        unsigned ms_out[2] = {0, 0};
        ORIP_TRY(vread(c, ms_out, sbase + nk, 2));                // the sample count and, with it, whether any polyline leaves the canvas
        const unsigned MS = ms_out[0]; const bool any_out = ms_out[1] != 0 || getenv("ORIP_CAPPREV_SCAN");
        if (MS > 0) {
            if (MS > 0x7ffffff0u) ORIP_FAIL(c, "too many samples");
            if (tdbg) { char b[48]; snprintf(b, sizeof b, " [MS %u]", MS); tlog += b; }
            HIPC(c, LN(c).vtmp[3].ensure((size_t)MS * (8 + 8 + 8 + 4 + 4 + 4 + 1 + 4 + 4 + 8 + 1) + 1024));
            SampleArrs A; A.sx = LN(c).vtmp[3].as<double>(); A.sy = A.sx + MS; A.dprev = A.sy + MS; A.xi = (int*)(A.dprev + MS); A.yi = A.xi + MS; A.rank = (unsigned*)(A.yi + MS);
            unsigned* npop = A.rank + MS; int* capprev = (int*)(npop + MS); int2* spt = (int2*)(capprev + MS + (MS & 1)); A.inc = (uint8_t*)(spt + MS); uint8_t* sflag = A.inc + MS;
            const unsigned nb = (unsigned)cdiv(MS, 256);
            HIPC(c, LN(c).vtmp[5].ensure((size_t)MS * 24 + 64 + (size_t)(nb + 1) * 8));
            unsigned* ckin = LN(c).vtmp[5].as<unsigned>(); unsigned* ckout = ckin + MS; unsigned* cvin = ckout + MS; unsigned* cvout = cvin + MS;
            const double cell = P.grid_stride > 0 ? P.grid_stride : std::max(4.0, P.col_rad); const double inv = 1.0 / cell;
            int2* hints = (int2*)(LN(c).vtmp[5].as<uint8_t>() + (((size_t)MS * 24 + 63) & ~(size_t)63));
            // canvas of first stamps: read at sample pixels only, so k_samples initialises exactly those and marks them in a bit plane (ORIP_CAPS_FULL: whole canvas, as before)
            const bool bits_path = !getenv("ORIP_CAPS_FULL");
            const int Wq = (W + 63) >> 6;
            HIPC(c, LN(c).canvas.ensure((size_t)W * H * 4 + 64));
            unsigned* firstseq = LN(c).canvas.as<unsigned>();
            unsigned long long* pixbits = nullptr;
            if (bits_path) {
                HIPC(c, LN(c).pixbits.ensure((size_t)Wq * H * 8 + 64));
                pixbits = LN(c).pixbits.as<unsigned long long>();
                HIPC(c, hipMemsetAsync(pixbits, 0, (size_t)Wq * H * 8, LN(c).stream));
            } else HIPC(c, hipMemsetAsync(firstseq, 0xff, (size_t)W * H * 4, LN(c).stream));
            hipLaunchKernelGGL(k_sample_hints, dim3(cdiv(nb, 256)), dim3(256), 0, LN(c).stream, cumoff, cum, info, ord, sbase, nk, MS, step, nb, hints);
            { ProfScope ps(c, "k_samples"); ORIP_WITH_SRC(c, kept0.p, sv, { hipLaunchKernelGGL(k_samples<decltype(sv)>, dim3(cdiv(nb, 4)), dim3(256), 0, LN(c).stream, sv, cumoff, cum, info, ord, sbase, nk, MS, step, W, H, A, inv, (unsigned*)nullptr, (unsigned*)nullptr, hints, (unsigned)nb, pixbits, Wq, firstseq); }); }
            tick("samples");
            // ---- A3
            {
                HIPC(c, LN(c).vtmp[8].ensure((size_t)MS * 8 + (size_t)(nk + 1) * 4 + 64));
                double* S = LN(c).vtmp[8].as<double>(); unsigned* redo = (unsigned*)(S + MS);
                HIPC(c, hipMemsetAsync(redo, 0, (size_t)(nk + 1) * 4, LN(c).stream));
                size_t bytes = 0;
                HIPC(c, rocprim::inclusive_scan_by_key(nullptr, bytes, A.rank, A.dprev, S, (size_t)MS, rocprim::plus<double>(), rocprim::equal_to<unsigned>(), LN(c).stream));
                HIPC(c, LN(c).tmpF.ensure(bytes + 16));
                HIPC(c, rocprim::inclusive_scan_by_key(LN(c).tmpF.p, bytes, A.rank, A.dprev, S, (size_t)MS, rocprim::plus<double>(), rocprim::equal_to<unsigned>(), LN(c).stream));
                ProfScope ps(c, "k_tail_sim");
                hipLaunchKernelGGL(k_tail_par, dim3(cdiv(MS, 256)), dim3(256), 0, LN(c).stream, sbase, A.rank, S, MS, P.tail_len_px, npop, redo);
                if (getenv("ORIP_TAIL_DBG")) { std::vector<unsigned> h(nk), sb(nk + 1); hipStreamSynchronize(LN(c).stream); hipMemcpy(h.data(), redo, nk * 4, hipMemcpyDeviceToHost); hipMemcpy(sb.data(), sbase, (nk + 1) * 4, hipMemcpyDeviceToHost); unsigned long long nf = 0, sf = 0, mx = 0; for (int64_t q = 0; q < nk; q++) if (h[q]) { nf++; sf += sb[q + 1] - sb[q]; mx = std::max<unsigned long long>(mx, sb[q + 1] - sb[q]); } fprintf(stderr, "[tail dbg] layer %d: %llu of %lld polylines redone, %llu of %u samples, longest redone %llu\n", layer, nf, (long long)nk, sf, MS, mx); }
                const unsigned* only = getenv("ORIP_TAIL_SEQ") ? nullptr : redo;          // test hook: force the sequential simulation everywhere
                // the sequential redo only feeds the acceptance test (A6): it runs on the lane's side stream under the capsule / hash work
                HIPC(c, hipEventRecord(LN(c).ev2, LN(c).stream));
                HIPC(c, hipStreamWaitEvent(LN(c).stream2, LN(c).ev2, 0));
#ifdef ORIP_VARIANTS
                if (ORIP_VARIANT("ORIP_TAIL_OLDSIM")) hipLaunchKernelGGL(k_tail_sim, dim3((unsigned)std::min<int64_t>(nk, 65535)), dim3(64), 0, LN(c).stream2, sbase, nk, P.tail_len_px, A, npop, only);
                else
#endif
                hipLaunchKernelGGL(k_tail_replay, dim3((unsigned)std::min<int64_t>(nk, 65535)), dim3(64), 0, LN(c).stream2, sbase, nk, P.tail_len_px, A, npop, only);
                HIPC(c, hipEventRecord(LN(c).ev3, LN(c).stream2));
            }
            if (any_out)
            {
                HIPC(c, LN(c).vtmp[8].ensure((size_t)MS * 8 + (size_t)(nk + 1) * 4 + 64));
                unsigned* lastin = LN(c).vtmp[8].as<unsigned>();            // the prefix sums of the tail simulation are no longer needed
                auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<unsigned>(0u), IncIndex{A.inc});
                size_t bytes = 0;
                HIPC(c, rocprim::inclusive_scan_by_key(nullptr, bytes, A.rank, vin, lastin, (size_t)MS, rocprim::maximum<unsigned>(), rocprim::equal_to<unsigned>(), LN(c).stream));
                HIPC(c, LN(c).tmpF.ensure(bytes + 16));
                HIPC(c, rocprim::inclusive_scan_by_key(LN(c).tmpF.p, bytes, A.rank, vin, lastin, (size_t)MS, rocprim::maximum<unsigned>(), rocprim::equal_to<unsigned>(), LN(c).stream));
                hipLaunchKernelGGL(k_capprev, dim3(cdiv(MS, 256)), dim3(256), 0, LN(c).stream, sbase, MS, A, lastin, capprev);
            }
            else capprev = nullptr;       // every sample is on the canvas: "the previous in-canvas sample" is simply the previous one (k_caps_insert)
            tick("tail");
            // ---- A4: de-duplicated capsules -> min-sequence canvas
            // The table only has to hold the DISTINCT capsules (retraced paths repeat theirs many times).  Their number is not known in advance:
            // start from what this lane saw last time (a resident chain repeats itself; 3 slots per capsule), else from a quarter of the sample
            // count, with bounded probing, and grow on overflow; 2 * MS slots always suffice.  A small table is a cache-resident one.
            unsigned long long tfull = 1024; while (tfull < 2ull * MS) tfull <<= 1;
            unsigned long long tsize = 1024;
            if (LN(c).caps_hint) { while (tsize < 3ull * LN(c).caps_hint) tsize <<= 1; } else { while (tsize < MS / 4ull) tsize <<= 1; }
            tsize = std::min(tsize, tfull);
            if (getenv("ORIP_CAPS_TINY")) tsize = 1024;            // test hook: exercise the growth path
            CapSlot* tab = nullptr;
            int* d_ovf = LN(c).flags.as<int>() + 62;
            unsigned* d_dist = LN(c).flags.as<unsigned>() + 126;
            for (;; tsize = std::min(tfull, tsize * 4)) {
                HIPC(c, LN(c).vtmp[4].ensure((size_t)tsize * 16 + 64));
                tab = LN(c).vtmp[4].as<CapSlot>();
                hipLaunchKernelGGL(k_caps_init, dim3((unsigned)cdiv(tsize, 256)), dim3(256), 0, LN(c).stream, tab, tsize);
                HIPC(c, hipMemsetAsync(d_ovf, 0, 4, LN(c).stream));
                const int max_probe = tsize >= tfull ? 0x7fffffff : 96;
                { ProfScope ps(c, "k_caps_insert"); hipLaunchKernelGGL(k_caps_insert, dim3(cdiv(MS, 1024)), dim3(256), 0, LN(c).stream, A, sbase, capprev, MS, tab, tsize - 1, max_probe, d_ovf); }
                int ovf = 0; ORIP_TRY(vread(c, &ovf, d_ovf));
                if (!ovf) break;
            }
            HIPC(c, hipMemsetAsync(d_dist, 0, 4, LN(c).stream));
            {
                ProfScope ps(c, "k_caps_stamp");
                const dim3 sg((unsigned)std::min<unsigned long long>(tsize / 64 / 4 + 1, 16384));
                if (bits_path) hipLaunchKernelGGL(k_caps_stamp_bits, sg, dim3(256), 0, LN(c).stream, tab, tsize, P.brush_forbid / 2, firstseq, W, H, pixbits, Wq, d_dist);
                else hipLaunchKernelGGL(k_caps_stamp, sg, dim3(256), 0, LN(c).stream, tab, tsize, P.brush_forbid / 2, firstseq, W, H);
            }
            caps_counted = bits_path;
            tick("caps");
            // ---- A5 / A6: cheap test of every sample, then _PointHash.near for the survivors
            {
                HIPC(c, hipStreamWaitEvent(LN(c).stream, LN(c).ev3, 0));       // pop counts of the redone polylines
                unsigned* surv = LN(c).vtmp[8].as<unsigned>();                  // (the scan results kept there have been consumed by k_capprev)
                unsigned* d_ns = LN(c).flags.as<unsigned>() + 60;               // survivor count; the work sum is an 8-byte word of its own
                unsigned long long* d_work = reinterpret_cast<unsigned long long*>(LN(c).flags.as<unsigned>() + 124);
                HIPC(c, hipMemsetAsync(d_ns, 0, 4, LN(c).stream));
                HIPC(c, hipMemsetAsync(d_work, 0, 8, LN(c).stream));
                { ProfScope ps(c, "k_accept"); hipLaunchKernelGGL(k_accept_pre, dim3(cdiv(MS, 1024)), dim3(256), 0, LN(c).stream, A, sbase, npop, MS, firstseq, W, spt, sflag, surv, d_ns, d_work); }
                unsigned long long h_work = 0; ORIP_TRY(vread(c, &h_work, d_work));
                const double R2 = P.col_rad * P.col_rad;
                // without the hash when it gives the hash's answer (cell >= radius) and costs less than sorting every sample into buckets
                const bool brute = cell >= P.col_rad && h_work <= 64ull * (unsigned long long)MS && !getenv("ORIP_HASH_SORT");
                if (tdbg) { char b2[64]; snprintf(b2, sizeof b2, " [near work %llu %s]", h_work, brute ? "direct" : "buckets"); tlog += b2; }
                if (brute) {
                    ProfScope ps(c, "k_accept");
                    hipLaunchKernelGGL(k_accept_brute, dim3((unsigned)std::min<unsigned>(cdiv(MS, 256), 16384u)), dim3(256), 0, LN(c).stream, A, sbase, npop, R2, surv, d_ns, sflag);
                } else {
                    // (polyline, cell) buckets in pop order: the samples of a polyline are contiguous, so its hash is its own range sorted by cell key
                    hipLaunchKernelGGL(k_cell_keys, dim3(cdiv(MS, 256)), dim3(256), 0, LN(c).stream, A, MS, inv, ckin, cvin);
                    {
                        ProfScope ps(c, "sort_cells");
                        size_t bytes = 0;
                        HIPC(c, rocprim::segmented_radix_sort_pairs(nullptr, bytes, ckin, ckout, cvin, cvout, (unsigned)MS, (unsigned)nk, sbase, sbase + 1, 0u, 32u, LN(c).stream));
                        HIPC(c, LN(c).tmpF.ensure(bytes + 16));
                        HIPC(c, rocprim::segmented_radix_sort_pairs(LN(c).tmpF.p, bytes, ckin, ckout, cvin, cvout, (unsigned)MS, (unsigned)nk, sbase, sbase + 1, 0u, 32u, LN(c).stream));
                    }
                    ProfScope ps(c, "k_accept");
                    hipLaunchKernelGGL(k_accept, dim3((unsigned)std::min<unsigned>(cdiv(MS, 256), 16384u)), dim3(256), 0, LN(c).stream, A, sbase, npop, inv, R2, ckout, cvout, surv, d_ns, sflag);
                }
            }
            HIPC(c, hipGetLastError());
            tick("accept");
            ORIP_TRY(orip_runs_to_polys(c, spt, sflag, MS, cleaned.p));
            tick("runs");
        }
        // ---- A7
        ORIP_TRY(split_small(c, cleaned.p, P, lines2.p, TOUT.xy, nt0, &nt2));
    }
    TOUT.n = nt0 + nt2;
    tick("A7");
    // ---- B
    DPolys* fin = &lines2.p;
    const int64_t n2 = lines2.p.n;
    if (P.post_on && n2 > 0) {
        if (n2 > 0x3fffffff) ORIP_FAIL(c, "too many lines");
        const int Wp = W + 2 * PAD8, Hp = H + 2 * PAD8; const size_t Np = (size_t)Wp * Hp;
        if (Np >= (1ull << 27)) ORIP_FAIL(c, "canvas too large for stage 08-B index packing");
        const int exp = P.post_brush * 2 + 6, rad = std::max(1, P.post_brush) / 2;
        HIPC(c, LN(c).vtmp[6].ensure((size_t)n2 * sizeof(PolyFeat) + (size_t)(n2 + 1) * (4 + 4 + 4) + (size_t)n2 * sizeof(GroupInfo) + 256));
        PolyFeat* f2 = LN(c).vtmp[6].as<PolyFeat>(); int* par = (int*)(f2 + n2); unsigned* is_root = (unsigned*)(par + (n2 + 1)); unsigned* root_scan = is_root + (n2 + 1);
        GroupInfo* grp = (GroupInfo*)(root_scan + (n2 + 1) + ((3 * (n2 + 1)) & 1));
        ORIP_TRY(vfeatures(c, lines2.p, 1, f2));
        hipLaunchKernelGGL(k_iota, dim3(cdiv(n2, 256)), dim3(256), 0, LN(c).stream, par, (int)n2);
        { ProfScope ps(c, "k_bbox_pairs"); hipLaunchKernelGGL(k_bbox_pairs, dim3((unsigned)std::min<int64_t>(n2, 8192)), dim3(256), 0, LN(c).stream, f2, (int)n2, exp, par); }
        hipLaunchKernelGGL(k_group_init, dim3(cdiv(n2, 256)), dim3(256), 0, LN(c).stream, grp, (int)n2);
        hipLaunchKernelGGL(k_group_accum, dim3(cdiv(n2 + 1, 256)), dim3(256), 0, LN(c).stream, f2, (int)n2, exp, par, grp, is_root);
        ORIP_TRY(vscan_excl<unsigned>(c, is_root, root_scan, (size_t)n2 + 1));
        hipLaunchKernelGGL(k_group_finish, dim3(cdiv(n2, 256)), dim3(256), 0, LN(c).stream, f2, (int)n2, is_root, root_scan, grp);
        tick("groups");
        // raster
        HIPC(c, LN(c).canvas.ensure(Np * 4 + 64));
        unsigned* gid = LN(c).canvas.as<unsigned>();
        HIPC(c, hipMemsetAsync(gid, 0, Np * 4, LN(c).stream));
        { ProfScope ps(c, "k_stamp_groups"); hipLaunchKernelGGL(k_stamp_groups, dim3(8192), dim3(256), 0, LN(c).stream, lines2.p.off.as<int64_t>(), lines2.p.pts.as<int32_t>(), n2, lines2.p.total, par, rad, gid, Wp, Hp); }
        dim3 g2(cdiv(Wp, 64), cdiv(Hp, 4)), blk(256);
        const size_t ntile_max = (size_t)g2.x * g2.y;
        const int Wwp = (Wp + 63) >> 6; const size_t nwords = (size_t)Hp * Wwp;
        HIPC(c, LN(c).vtmp[9].ensure(Np * 2 + ntile_max * 4 + nwords * 16 + 256));
        u8* skA = LN(c).vtmp[9].as<u8>(); u8* skB = skA + Np; unsigned* tiles = (unsigned*)(skB + ((Np + 15) & ~(size_t)15));
        unsigned long long* bA = (unsigned long long*)(tiles + ((ntile_max + 3) & ~(size_t)3)); unsigned long long* bB = bA + nwords;
        int* d_changed = LN(c).flags.as<int>() + 48; unsigned* d_ntiles = LN(c).flags.as<unsigned>() + 52; (void)d_changed; (void)d_ntiles;      // (byte-plane thinning: variants build)
        if (!ORIP_VARIANT("ORIP_THIN_BYTES")) {
            const dim3 gwd((unsigned)cdiv((int64_t)nwords, 256));
            hipLaunchKernelGGL(k_gid_to_bits, gwd, blk, 0, LN(c).stream, gid, bA, Hp, Wp, Wwp);      // 4 waves x 64 words per block
            tick("raster");
            // Twelve iterations before the first round trip to the host (16-px lines thin in 9 .. 12), four per round trip after that, each iteration with
            // its own flag: an iteration after the first unchanged one changes nothing either, so running to the end of a batch leaves the image the
            // reference's loop stops with (48 iterations at most: the same cap)
            int* d_chg = LN(c).flags.as<int>() + 240;
            for (int it = 0; it < 48; ) {
                const int nb = it == 0 ? 12 : 4;
                HIPC(c, hipMemsetAsync(d_chg, 0, 48, LN(c).stream));
                if (!getenv("ORIP_ZS_LAUNCHES")) {          // the whole batch in one launch, tile by tile in LDS (ORIP_ZS_LAUNCHES=1: one launch per sub-iteration, as before)
                    ProfScope ps(c, "k_zs_sub");
                    hipLaunchKernelGGL(k_zs_tile, dim3((unsigned)cdiv(Wwp, 2), (unsigned)cdiv(Hp, ZS_TR)), blk, 0, LN(c).stream, bA, bB, Hp, Wwp, nb, d_chg);
                    std::swap(bA, bB);
                } else
                for (int b = 0; b < nb; b++) {
                    { ProfScope ps(c, "k_zs_sub"); hipLaunchKernelGGL(k_zs_bits, gwd, blk, 0, LN(c).stream, bA, bB, Hp, Wwp, 0, d_chg + b); }
                    { ProfScope ps(c, "k_zs_sub"); hipLaunchKernelGGL(k_zs_bits, gwd, blk, 0, LN(c).stream, bB, bA, Hp, Wwp, 1, d_chg + b); }
                }
                int ch[12] = {0}; ORIP_TRY(vread(c, ch, d_chg, 12));
                bool all = true; for (int b = 0; b < nb; b++) all = all && ch[b] != 0;
                if (!all) break;
                it += nb;
            }
            hipLaunchKernelGGL(k_bits_to_mask, gwd, blk, 0, LN(c).stream, bA, skA, Hp, Wp, Wwp);
        }
#ifdef ORIP_VARIANTS
        else {
        HIPC(c, hipMemsetAsync(d_ntiles, 0, 4, LN(c).stream));
        HIPC(c, hipMemsetAsync(skB, 0, Np, LN(c).stream));
        hipLaunchKernelGGL(k_gid_to_mask, g2, blk, 0, LN(c).stream, gid, skA, Hp, Wp, tiles, d_ntiles);
        tick("raster");
        const dim3 gz((unsigned)std::min<size_t>(ntile_max, 16384));
        for (int it = 0; it < 48; it++) {
            HIPC(c, hipMemsetAsync(d_changed, 0, 4, LN(c).stream));
            { ProfScope ps(c, "k_zs_sub"); hipLaunchKernelGGL(k_zs_sub, gz, blk, 0, LN(c).stream, skA, skB, Hp, Wp, 0, d_changed, tiles, d_ntiles, (int)g2.x); }
            { ProfScope ps(c, "k_zs_sub"); hipLaunchKernelGGL(k_zs_sub, gz, blk, 0, LN(c).stream, skB, skA, Hp, Wp, 1, d_changed, tiles, d_ntiles, (int)g2.x); }
            int ch = 0; ORIP_TRY(vread(c, &ch, d_changed));
            if (!ch) break;
        }
        }
#endif
        tick("thin");
        // components
        HIPC(c, LN(c).vtmp[10].ensure(Np * 4 + 64));
        int* L2 = LN(c).vtmp[10].as<int>();
        if (!ORIP_VARIANT("ORIP_THIN_BYTES")) {      // bA holds the thinned bit plane
            const dim3 gwd((unsigned)cdiv((int64_t)nwords, 256));
            hipLaunchKernelGGL(k_ccl2_bits, gwd, blk, 0, LN(c).stream, bA, L2, Hp, Wp, Wwp, 0);
            { ProfScope ps(c, "k_ccl2_merge"); hipLaunchKernelGGL(k_ccl2_bits, gwd, blk, 0, LN(c).stream, bA, L2, Hp, Wp, Wwp, 1); }
            hipLaunchKernelGGL(k_ccl2_bits, gwd, blk, 0, LN(c).stream, bA, L2, Hp, Wp, Wwp, 2);
        }
#ifdef ORIP_VARIANTS
        else {
        hipLaunchKernelGGL(k_ccl2_init, g2, blk, 0, LN(c).stream, skA, L2, Hp, Wp);
        { ProfScope ps(c, "k_ccl2_merge"); hipLaunchKernelGGL(k_ccl2_merge, g2, blk, 0, LN(c).stream, skA, L2, Hp, Wp); }
        hipLaunchKernelGGL(k_ccl2_flatten, dim3(cdiv(Np, 256)), blk, 0, LN(c).stream, skA, L2, (int)Np);
        }
#endif
        tick("c:ccl");
        const bool sk_bits = !ORIP_VARIANT("ORIP_THIN_BYTES") && !getenv("ORIP_SK_BYTES");          // the thinned bit plane is in bA
        const int nblk = sk_bits ? (int)cdiv((int64_t)nwords, 256) : cdiv((int64_t)Np, 1024);
        HIPC(c, LN(c).vtmp[0].ensure((size_t)(nblk + 1) * 8 + 64));
        unsigned* bc = LN(c).vtmp[0].as<unsigned>(); unsigned* bo = bc + (nblk + 1);
        HIPC(c, hipMemsetAsync(bc + nblk, 0, 4, LN(c).stream));
        if (sk_bits) hipLaunchKernelGGL(k_sk_count_bits, dim3(nblk), blk, 0, LN(c).stream, bA, nwords, bc);
        else hipLaunchKernelGGL(k_sk_count, dim3(nblk), blk, 0, LN(c).stream, skA, (int64_t)Np, bc);
        ORIP_TRY(vscan_excl<unsigned>(c, bc, bo, (size_t)nblk + 1));
        unsigned M = 0; ORIP_TRY(vread(c, &M, bo + nblk));
        if (M > 0) {
            HIPC(c, LN(c).vtmp[1].ensure((size_t)M * 16 + 64));
            unsigned* kin = LN(c).vtmp[1].as<unsigned>(); unsigned* lin_in = kin + M; unsigned* keys = lin_in + M; unsigned* lin = keys + M;
            if (sk_bits) hipLaunchKernelGGL(k_sk_write_bits, dim3(nblk), blk, 0, LN(c).stream, bA, L2, nwords, Wp, Wwp, bo, kin, lin_in);
            else hipLaunchKernelGGL(k_sk_write, dim3(nblk), blk, 0, LN(c).stream, skA, L2, (int64_t)Np, bo, kin, lin_in);
            ORIP_TRY((vsort_pairs<unsigned, unsigned>(c, kin, keys, lin_in, lin, (size_t)M, 0, 27)));
            HIPC(c, LN(c).vtmp[3].ensure((size_t)(M + 1) * 8 + 64));
            unsigned* head = LN(c).vtmp[3].as<unsigned>(); unsigned* hs = head + (M + 1);
            hipLaunchKernelGGL(k_heads2, dim3(cdiv(M + 1, 256)), blk, 0, LN(c).stream, keys, (int64_t)M, head);
            ORIP_TRY(vscan_excl<unsigned>(c, head, hs, (size_t)M + 1));
            unsigned NC = 0; ORIP_TRY(vread(c, &NC, hs + M));
            HIPC(c, LN(c).vtmp[4].ensure((size_t)(NC + 1) * (4 + 8 + 8 + 4 + 4 + 4 + 4 + 4) + (size_t)NC * sizeof(GatherDesc) + 256));
            unsigned long long* ckin = LN(c).vtmp[4].as<unsigned long long>(); unsigned long long* ckout = ckin + (NC + 1);
            unsigned* cs = (unsigned*)(ckout + (NC + 1)); unsigned* cidx = cs + (NC + 2); unsigned* corder = cidx + (NC + 1); unsigned* outcnt = corder + (NC + 1);
            unsigned* oflag = outcnt + (NC + 1); unsigned* oscan = oflag + (NC + 1); GatherDesc* pd = (GatherDesc*)(oscan + (NC + 1) + ((6 * (NC + 1) + 1) & 1) + 2);
            hipLaunchKernelGGL(k_comp_starts2, dim3(cdiv(M, 256)), blk, 0, LN(c).stream, head, hs, (int64_t)M, cs, NC);
            tick("c:sort");
            hipLaunchKernelGGL(k_nearest_anchor, dim3(cdiv(M, 256)), blk, 0, LN(c).stream, lin, (int64_t)M, gid, Wp, grp);
            tick("c:anchor");
            hipLaunchKernelGGL(k_comp_keys, dim3(cdiv(NC, 128)), dim3(128), 0, LN(c).stream, cs, NC, lin, gid, Wp, grp, ckin, cidx);
            ORIP_TRY((vsort_pairs<unsigned long long, unsigned>(c, ckin, ckout, cidx, corder, (size_t)NC, 0, 64)));
            tick("comps");
            // per-component path, resample, RDP
            {
                const double stp = P.post_step;
                if (!(stp >= 1.0)) ORIP_FAIL(c, "postmerge_resample_step must be >= 1");
                const double ratio = std::min(1.0, 1.41422 / stp);          // resample points per component pixel
                auto pcap_of = [&](unsigned cap) { return (unsigned)(cap * ratio) + (unsigned)stp + 4u; };
                auto cap_of = [&](size_t budget) {                             // bytes: 25/node + 13/resample point
                    unsigned cap = (unsigned)((budget - 13.0 * (stp + 4.0) - 64.0) / (25.0 + 13.0 * ratio));
                    return std::min(cap, 65000u) & ~7u;
                };
                const size_t lds0 = 32 * 1024, lds1 = 160 * 1024;
                unsigned cap0 = cap_of(lds0), cap1 = cap_of(lds1);
                if (const char* ov = getenv("ORIP_COMP_CAPS")) {          // test hook: force components into the larger classes
                    unsigned a = 0, b2 = 0; if (sscanf(ov, "%u,%u", &a, &b2) == 2 && a >= 8 && a <= b2) { cap0 = std::min(cap0, a & ~7u); cap1 = std::min(cap1, b2 & ~7u); }
                }
                auto lds_bytes = [&](unsigned cap) { return (size_t)cap * 25 + (size_t)pcap_of(cap) * 13 + 16; };
                // scratch: cid canvas (reuses the BFS canvas), nbr, class lists, global-class work arrays
                HIPC(c, LN(c).vtmp[5].ensure(Np * 4 + (size_t)M * (32 + 4 + 4 + 4 + 8 + 8 + 1 + 1 + 8) + (size_t)(NC + 1) * 12 + 1024));
                unsigned char* bump = LN(c).vtmp[5].as<unsigned char>();
                auto take = [&](size_t bytes) { unsigned char* r = bump; bump += (bytes + 15) & ~(size_t)15; return r; };
                unsigned* cid = (unsigned*)take(Np * 4); unsigned* nbr = (unsigned*)take((size_t)M * 32);
                CompScratch X; X.prev = (unsigned*)take((size_t)M * 4); X.que = (unsigned*)take((size_t)M * 4); X.cum = (float*)take((size_t)M * 4);
                X.P = (float2*)take((size_t)M * 8); X.stk = (int2*)take((size_t)M * 8); int2* outpts = (int2*)take((size_t)M * 8);
                unsigned* l0 = (unsigned*)take((size_t)(NC + 1) * 4); unsigned* l1 = (unsigned*)take((size_t)(NC + 1) * 4); unsigned* l2 = (unsigned*)take((size_t)(NC + 1) * 4);
                X.seen = (u8*)take(M); X.keep = (u8*)take(M);
                unsigned* counts = LN(c).flags.as<unsigned>() + 56;
                HIPC(c, hipMemsetAsync(counts, 0, 12, LN(c).stream));
                hipLaunchKernelGGL(k_cid_fill, dim3(cdiv(M, 256)), blk, 0, LN(c).stream, lin, M, cid);
                hipLaunchKernelGGL(k_nbr_build, dim3((unsigned)cdiv((int64_t)M * 8, 256)), blk, 0, LN(c).stream, lin, M, skA, cid, Wp, Hp, nbr);
                hipLaunchKernelGGL(k_comp_classes, dim3(cdiv(NC, 256)), blk, 0, LN(c).stream, corder, NC, cs, cap0, cap1, std::max(2, P.post_minlen), counts, l0, l1, l2, outcnt);
                CompArgs A; A.corder = corder; A.cs = cs; A.lin = lin; A.gid = gid; A.g = grp; A.cid = cid; A.nbr = nbr; A.Wp = Wp; A.min_len = P.post_minlen; A.step = stp;
                A.eps = (float)P.post_eps; A.outpts = outpts; A.outcnt = outcnt;
                static std::once_flag attr_once;            // several layer threads may arrive here together
                static std::atomic<int> attr_err{0};
                std::call_once(attr_once, [&] { orip_max_lds(k_comp_paths_lds, (int)lds1, attr_err); });
                if (attr_err.load()) ORIP_FAIL(c, "hipFuncSetAttribute(k_comp_paths_lds) failed: %s", hipGetErrorString((hipError_t)attr_err.load()));
                ProfScope ps(c, "k_comp_paths");
                // the few large components are long serial chains: they start on the side stream while the many small ones run here
                HIPC(c, hipEventRecord(LN(c).ev2, LN(c).stream));
                HIPC(c, hipStreamWaitEvent(LN(c).stream2, LN(c).ev2, 0));
                hipLaunchKernelGGL(k_comp_paths_lds, dim3(std::min(NC, 1024u)), dim3(64), lds_bytes(cap1), LN(c).stream2, A, l1, counts + 1, cap1, pcap_of(cap1));
                hipLaunchKernelGGL(k_comp_paths_glb, dim3(std::min(NC, 1024u)), dim3(64), 0, LN(c).stream2, A, l2, counts + 2, X);
                HIPC(c, hipEventRecord(LN(c).ev3, LN(c).stream2));
                hipLaunchKernelGGL(k_comp_paths_lds, dim3(std::min(NC, 8192u)), dim3(64), lds_bytes(cap0), LN(c).stream, A, l0, counts + 0, cap0, pcap_of(cap0));
                HIPC(c, hipStreamWaitEvent(LN(c).stream, LN(c).ev3, 0));
                tick("paths");
                hipLaunchKernelGGL(k_flag_nonzero, dim3(cdiv(NC + 1, 256)), blk, 0, LN(c).stream, outcnt, NC, oflag);
                ORIP_TRY(vscan_excl<unsigned>(c, oflag, oscan, (size_t)NC + 1));
                unsigned NP = 0; ORIP_TRY(vread(c, &NP, oscan + NC));
                merged.p.n = 0; merged.p.total = 0; merged.p.set_explicit();
                HIPC(c, merged.p.off.ensure(64)); HIPC(c, hipMemsetAsync(merged.p.off.p, 0, 8, LN(c).stream));
                if (NP) {
                    hipLaunchKernelGGL(k_path_desc, dim3(cdiv(NC, 256)), blk, 0, LN(c).stream, corder, cs, outcnt, oflag, oscan, NC, pd);
                    ORIP_TRY(vgather(c, pd, NP, reinterpret_cast<const int32_t*>(outpts), merged.p));
                }
            }
        } else { merged.p.n = 0; merged.p.total = 0; merged.p.set_explicit(); HIPC(c, merged.p.off.ensure(64)); HIPC(c, hipMemsetAsync(merged.p.off.p, 0, 8, LN(c).stream)); }
        HIPC(c, hipGetLastError());
        fin = &merged.p;
    }
    tick("gather");
    // ---- C
    ORIP_TRY(vreorder(c, *fin, OUT, 8));
    unsigned h_dist = 0;
    if (caps_counted) HIPC(c, hipMemcpyAsync(&h_dist, LN(c).flags.as<unsigned>() + 126, 4, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    if (caps_counted) { LN(c).caps_hint = h_dist; if (tdbg) { char b[48]; snprintf(b, sizeof b, " [caps distinct %u]", h_dist); tlog += b; } }
    tick("reorder");
    if (tdbg) fprintf(stderr, "[time08] layer %d (in %lld polys %lld pts, kept %lld pts, cleaned %lld/%lld, lines2 %lld/%lld):%s\n", layer, (long long)S.n, (long long)S.total, (long long)kept0.p.total, (long long)cleaned.p.n, (long long)cleaned.p.total, (long long)lines2.p.n, (long long)lines2.p.total, tlog.c_str());
    return 0;
}

// The front of one layer's pipeline in ONE call: orip_contours_layer -> orip_scale_vectors -> [orip_sort_contours -> [orip_dedup_layer]] (upto = 5, 7
// or 8) on the layer's lane.  Same results as the four calls; what goes away are the returns to the (Python) caller between the stages of a resident
// chain -- three hand-overs per layer, each 0.2-0.5 ms of idle stream on the critical layer.
extern "C" int orip_layer_front(orip_ctx* c, int layer, float sx, float sy, float dx, float dy, int upto, const orip_params08* prm) {
    orip_enter(c);
    if (upto >= 8 && !prm) ORIP_FAIL(c, "stage 08 needs its parameters");
    ORIP_TRY(orip_contours_layer_impl(c, layer, false));
    ORIP_TRY(orip_scale_vectors_impl(c, layer, sx, sy, dx, dy, upto < 7));
    if (upto >= 7) ORIP_TRY(orip_sort_contours_impl(c, layer, upto < 8, upto >= 8 ? prm : nullptr));
    if (upto >= 8) ORIP_TRY(orip_dedup_layer(c, layer, prm));
    return 0;
}

