// csrc/walker.h -- the centerline walker of stage 04 (04_find_contours.py trace_centerlines, 04:137-205).
//
// One WAVEFRONT walks one connected component with the reference's exact serial semantics:
//   * the 8 neighbours of the current pixel are probed by lanes 0..7 in one load each; the NEIGH8-ordered choice is a
//     ballot + find-first-set, so a step costs one memory round trip instead of up to eight dependent ones;
//   * the raster-ordered scans for the next endpoint / leftover pixel test 64 list entries per step (ballot);
//   * every decision is taken from wave-uniform values (ballot masks, broadcasts), lane 0 does all stores;
//   * no per-step fence: the pixel marked at step t is the current pixel of step t+1 (never one of its own neighbours), its
//     predecessor is excluded as `prev`, and older marks precede a probe load whose result was already waited for (vmcnt
//     retires in issue order); one fence per walk orders the marks before the next raster scan.
// The same source is compiled with g++ by tests/host/walk_harness.cpp, where a "wave" is emulated by plain loops, so
// the walk logic (phases, guards, cycle fast-forward) is unit-tested on the CPU as well (test infrastructure).
#pragma once
#include <cstdint>
#include "../../include/orip.h"
#if defined(__HIPCC__)
#define ORIP_HD __host__ __device__
#else
#define ORIP_HD
#endif
typedef uint8_t u8;
#define ST_FG 1
#define ST_VIS 2
#define ST_END 4
#define ST_JUN 8

struct WalkArgs {
    int H, W; int64_t plane;
    u8* st;                            // [K,H,W] state bytes
    const unsigned* keys; const unsigned* lin; const unsigned* comp_start; unsigned nc;
    long long total_fg[ORIP_MAX_LAYERS];
    // count pass outputs / write pass inputs
    unsigned long long* comp_pts; unsigned* comp_paths;            // per component (kept paths only)
    const unsigned long long* pts_base; const unsigned* path_base; // exclusive scans over components (global)
    unsigned long long layer_pts_base[ORIP_MAX_LAYERS]; unsigned layer_path_base[ORIP_MAX_LAYERS];
    int32_t* pts[ORIP_MAX_LAYERS]; int64_t* off[ORIP_MAX_LAYERS];
    // cycle expansion descriptors: (layer, dst point index, period, count)
    unsigned long long* desc; unsigned* n_desc; unsigned desc_cap;
    const unsigned* comp_order;        // optional: component processed by wave i (largest first), or nullptr
    // bounce memo (count pass builds it, write pass replays per-walk recipes); all optional (nullptr = plain Brent)
    const int* qidx;                   // [K,H,W]: index of a skeleton pixel in lin[] / keys[]
    unsigned* memo;                    // [M*8]: log index + 1 of the state (pixel, incoming direction), 0 = unknown
    unsigned* logbuf;                  // 3 words per entry: (lin << 3 | dir), cyc_begin, end  -- region of component c: [6*b + 64*c, +6*fg+64)
    unsigned* recipe;                  // 3 words per start pixel q: own steps before the jump, log index + 1, remaining steps
};

namespace walk_detail {
#if defined(__HIP_DEVICE_COMPILE__)
struct Wave {
    int lane;
    __device__ Wave() : lane((int)(threadIdx.x & 63)) {}
    __device__ bool leader() const { return lane == 0; }
    __device__ unsigned l0() const { return (unsigned)lane; }
    __device__ unsigned nl() const { return 64u; }
    // probe the 8 neighbours of (px,py); returns masks over NEIGH8 indices
    __device__ void probe(const u8* st, int W, int H, int px, int py, int pvx, int pvy, unsigned& m_any, unsigned& m_unvis, u8& myv) const {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        u8 v = 0; bool any = false;
        if (lane < 8) {
            int xx = px + dxs[lane], yy = py + dys[lane];
            if (xx >= 0 && xx < W && yy >= 0 && yy < H) { v = st[(size_t)yy * W + xx]; any = (v & ST_FG) && !(xx == pvx && yy == pvy); }
        }
        myv = v;
        m_any = (unsigned)(__ballot(any) & 0xffu);
        m_unvis = (unsigned)(__ballot(any && !(v & ST_VIS)) & 0xffu);
    }
    __device__ u8 value_of(u8 myv, int k) const { return (u8)__shfl((int)myv, k, 64); }
    // next q in [q0, e) whose pixel satisfies: (state & need) == need && !(state & ST_VIS)
    __device__ unsigned scan(const unsigned* lin, const u8* st, unsigned q0, unsigned e, u8 need) const {
        for (unsigned q = q0; q < e; q += 64) {
            unsigned qq = q + lane; bool ok = false;
            if (qq < e) { u8 v = st[lin[qq]]; ok = ((v & need) == need) && !(v & ST_VIS); }
            unsigned long long m = __ballot(ok);
            if (m) return q + (unsigned)(__ffsll((long long)m) - 1);
        }
        return e;
    }
    __device__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0); }
    __device__ unsigned fetch_inc(unsigned* p) const { unsigned r = 0; if (lane == 0) r = atomicAdd(p, 1u); return (unsigned)__shfl((int)r, 0, 64); }
};
#else
struct Wave {
    u8 nv[8];
    bool leader() const { return true; }
    unsigned l0() const { return 0u; }
    unsigned nl() const { return 1u; }
    void probe(const u8* st, int W, int H, int px, int py, int pvx, int pvy, unsigned& m_any, unsigned& m_unvis, u8& myv) {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        m_any = m_unvis = 0; myv = 0;
        for (int k = 0; k < 8; k++) {
            int xx = px + dxs[k], yy = py + dys[k]; nv[k] = 0;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            u8 v = st[(size_t)yy * W + xx]; nv[k] = v;
            if ((v & ST_FG) && !(xx == pvx && yy == pvy)) { m_any |= 1u << k; if (!(v & ST_VIS)) m_unvis |= 1u << k; }
        }
    }
    u8 value_of(u8, int k) const { return nv[k]; }
    unsigned scan(const unsigned* lin, const u8* st, unsigned q0, unsigned e, u8 need) const {
        for (unsigned q = q0; q < e; q++) { u8 v = st[lin[q]]; if (((v & need) == need) && !(v & ST_VIS)) return q; }
        return e;
    }
    void fence() const {}
    unsigned fetch_inc(unsigned* p) const { return (*p)++; }
};
#endif
ORIP_HD inline int ffs8(unsigned m) { int k = 0; while (!((m >> k) & 1u)) k++; return k; }
}  // namespace walk_detail

template <bool WRITE>
ORIP_HD inline void walk_component(const WalkArgs& A, unsigned c) {
    using namespace walk_detail;
    const int NBX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};   // NEIGH8 (dx,dy), 04:12
    const int NBY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    Wave wv;
    const unsigned b = A.comp_start[c], e = A.comp_start[c + 1];
    const int layer = (int)(A.keys[b] >> 26);
    u8* st = A.st + A.plane * layer;
    const int W = A.W, H = A.H;
    const long long fg_comp = (long long)(e - b), total_fg = A.total_fg[layer];
    unsigned long long n_pts = 0; unsigned n_paths = 0;
    int32_t* out = nullptr; int64_t* off = nullptr; unsigned long long wpos = 0, wend = 0; unsigned wpath = 0;
    if (WRITE) {
        out = A.pts[layer]; off = A.off[layer];
        wpos = A.pts_base[c] - A.layer_pts_base[layer];
        wpath = A.path_base[c] - A.layer_path_base[layer];
        wend = A.pts_base[c + 1] - A.layer_pts_base[layer];
    }
    // Paths shorter than 5 points are dropped after the fact, so their points must never land outside this component's own
    // output range [wpos0, wend) (the range holds exactly the kept paths; anything inside it is overwritten by a later kept path).
    auto emit = [&](unsigned long long pos, int x, int y) { if (WRITE && pos < wend && wv.leader()) { out[2 * pos] = x; out[2 * pos + 1] = y; } };
    auto mark = [&](size_t j, u8 v) { if (wv.leader()) st[j] = (u8)(v | ST_VIS); };
    // ---- phase 1: walks from endpoints (04:144-171)
    for (unsigned q = wv.scan(A.lin, st, b, e, ST_FG | ST_END); q < e; q = wv.scan(A.lin, st, q + 1, e, ST_FG | ST_END)) {
        unsigned s = A.lin[q];
        int px = (int)(s % W), py = (int)(s / W), pvx = -1, pvy = -1;
        unsigned long long len = 1; emit(wpos, px, py);
        mark(s, ST_FG | ST_END);
        long long guard = 0;
        while (true) {
            unsigned m_any, m_unvis; u8 myv;
            wv.probe(st, W, H, px, py, pvx, pvy, m_any, m_unvis, myv);
            if (!m_unvis) break;
            int k = ffs8(m_unvis);
            int nx = px + NBX[k], ny = py + NBY[k];
            u8 v = wv.value_of(myv, k);
            emit(wpos + len, nx, ny); len++;
            mark((size_t)ny * W + nx, v);
            pvx = px; pvy = py; px = nx; py = ny;
            if (v & (ST_JUN | ST_END)) break;
            guard++;
            if (guard > total_fg * 2) break;
        }
        wv.fence();       // marks of this walk are complete before the next scan
        if (len >= 5) {   // >=2 to be a path (04:168) and >=5 to survive vectorize_layer (04:224)
            n_pts += len; n_paths++;
            if (WRITE) { wpos += len; if (wv.leader()) off[wpath + 1] = (int64_t)wpos; wpath++; }
        }
    }
    // ---- phase 2: leftovers / cycles (04:174-205)
    // Bounce memo.  Once a walk has no unvisited neighbour left it follows the deterministic map (prev,cur) -> next over visited
    // pixels.  That map never changes again for states that had no unvisited neighbour (the visited set only grows), so a
    // trajectory recorded by an earlier walk of this component stays valid: a later walk that reaches a recorded state jumps
    // straight to "the rest is that trajectory" instead of re-walking transient + cycle.  The count pass records trajectories
    // (log), indexes their states (memo) and leaves one recipe per walk; the write pass replays the recipe.
    const unsigned log_base = 6u * b + 64u * c, log_cap = 6u * (e - b) + 64u;
    unsigned logcur = 0;
    const bool use_memo = A.memo != nullptr && A.logbuf != nullptr && A.recipe != nullptr && A.qidx != nullptr;
    const int* qidx = use_memo ? A.qidx + A.plane * layer : nullptr;
    auto log_pos = [&](unsigned i, unsigned long long R, int& ox, int& oy) {   // position R steps after logged state i
        unsigned cb = A.logbuf[3ull * i + 1], en = A.logbuf[3ull * i + 2];
        unsigned long long f = (unsigned long long)i + R;
        if (f >= en) f = cb + (f - en) % (unsigned long long)(en - cb);
        unsigned l = A.logbuf[3ull * f] >> 3;
        ox = (int)(l % (unsigned)W); oy = (int)(l / (unsigned)W);
    };
    for (unsigned q = wv.scan(A.lin, st, b, e, ST_FG); q < e; q = wv.scan(A.lin, st, q + 1, e, ST_FG)) {
        unsigned s = A.lin[q];
        const int x0 = (int)(s % W), y0 = (int)(s / W);
        int px = x0, py = y0, pvx = -1, pvy = -1;
        unsigned long long len = 1; emit(wpos, px, py);
        { u8 sv = st[s]; mark(s, sv); }
        long long guard = 0;
        unsigned rec_t = 0, rec_i1 = 0, rec_R = 0;
        if (WRITE && use_memo) { rec_t = A.recipe[3ull * q]; rec_i1 = A.recipe[3ull * q + 1]; rec_R = A.recipe[3ull * q + 2]; }
        const bool replay = WRITE && rec_i1 != 0;
        unsigned nofresh = 0; bool log_ok = true;
        // Brent cycle detection on the (prev,cur) state; reset whenever a fresh pixel is consumed
        int tpx = -2, tpy = -2, tcx = -2, tcy = -2; long long power = 1, lam = 0;
        while (true) {
            if (replay && len - 1 == rec_t) {     // the count pass jumped here: the rest is a recorded trajectory
                unsigned slot = wv.fetch_inc(A.n_desc);
                if (slot < A.desc_cap && wv.leader()) {
                    unsigned long long* d = A.desc + 4ull * slot;
                    d[0] = (unsigned long long)layer | (1ull << 32); d[1] = wpos + len; d[2] = (unsigned long long)(rec_i1 - 1); d[3] = (unsigned long long)rec_R;
                }
                log_pos(rec_i1 - 1, rec_R, px, py);
                len += rec_R;
                break;
            }
            unsigned m_any, m_unvis; u8 myv;
            wv.probe(st, W, H, px, py, pvx, pvy, m_any, m_unvis, myv);
            bool fresh = m_unvis != 0;
            unsigned m = fresh ? m_unvis : m_any;
            if (!m) break;
            int k = ffs8(m);
            int nx = px + NBX[k], ny = py + NBY[k];
            emit(wpos + len, nx, ny); len++;
            if (fresh) mark((size_t)ny * W + nx, wv.value_of(myv, k));
            pvx = px; pvy = py; px = nx; py = ny;
            if (px == x0 && py == y0) break;
            guard++;
            if (guard > fg_comp * 4) break;
            if (fresh) { tpx = pvx; tpy = pvy; tcx = px; tcy = py; power = 1; lam = 0; nofresh = 0; log_ok = true; }
            else if (!replay) {
                if (!WRITE && use_memo) {
                    const unsigned lin_cur = (unsigned)py * (unsigned)W + (unsigned)px;
                    const unsigned qn = (unsigned)qidx[lin_cur];
                    const unsigned mi = A.memo[8ull * qn + (unsigned)k];
                    if (mi) {                      // recorded state: everything that follows is known
                        const unsigned long long R = (unsigned long long)(fg_comp * 4 + 1 - guard);
                        if (wv.leader()) { A.recipe[3ull * q] = (unsigned)(len - 1); A.recipe[3ull * q + 1] = mi; A.recipe[3ull * q + 2] = (unsigned)R; }
                        log_pos(mi - 1, R, px, py);
                        len += R;
                        break;
                    }
                    if (log_ok && logcur + nofresh < log_cap) { if (wv.leader()) A.logbuf[3ull * (log_base + logcur + nofresh)] = (lin_cur << 3) | (unsigned)k; }
                    else log_ok = false;
                    nofresh++;
                }
                lam++;
                if (pvx == tpx && pvy == tpy && px == tcx && py == tcy) {
                    // the last `lam` steps repeat forever (no fresh pixel is reachable from the cycle and the start is not on it):
                    // remaining steps until the guard fires
                    long long remaining = fg_comp * 4 + 1 - guard;
                    bool recorded = false;
                    if (!WRITE && use_memo && log_ok && nofresh >= (unsigned)lam) {
                        // commit the trajectory [begin, end): transient then the cycle [end - lam, end); index its states
                        const unsigned begin = log_base + logcur, end = begin + nofresh, cb = end - (unsigned)lam;
                        wv.fence();
                        for (unsigned t = wv.l0(); t < nofresh; t += wv.nl()) {
                            const unsigned idx = begin + t;
                            A.logbuf[3ull * idx + 1] = cb; A.logbuf[3ull * idx + 2] = end;
                            const unsigned ld = A.logbuf[3ull * idx];
                            A.memo[8ull * (unsigned)qidx[ld >> 3] + (ld & 7u)] = idx + 1;
                        }
                        wv.fence();
                        logcur += nofresh;
                        if (wv.leader()) { A.recipe[3ull * q] = (unsigned)(len - 1); A.recipe[3ull * q + 1] = end; /* (end-1)+1 */ A.recipe[3ull * q + 2] = (unsigned)remaining; }
                        recorded = true;
                    }
                    if (remaining > 0) {
                        if (WRITE) {      // no recipe for this walk (log overflow in the count pass): the tail is periodic within its own output
                            unsigned slot = wv.fetch_inc(A.n_desc);
                            if (slot < A.desc_cap && wv.leader()) {
                                unsigned long long* d = A.desc + 4ull * slot;
                                d[0] = (unsigned long long)layer; d[1] = wpos + len; d[2] = (unsigned long long)lam; d[3] = (unsigned long long)remaining;
                            }
                        }
                        // position after the remaining steps = cycle point (remaining mod lam) steps ahead; needed for the closing test
                        long long adv = remaining % lam;
                        for (long long t = 0; t < adv; t++) {
                            unsigned ma, mu; u8 mv;
                            wv.probe(st, W, H, px, py, pvx, pvy, ma, mu, mv);
                            int kk = ffs8(ma);
                            int bx = px + NBX[kk], by = py + NBY[kk];
                            pvx = px; pvy = py; px = bx; py = by;
                        }
                        len += (unsigned long long)remaining;
                    }
                    (void)recorded;
                    break;
                }
                if (lam == power) { tpx = pvx; tpy = pvy; tcx = px; tcy = py; power <<= 1; lam = 0; }
            }
        }
        wv.fence();
        if (len >= 2) {
            int ddx = x0 - px, ddy = y0 - py;
            if (ddx * ddx + ddy * ddy < 3) {   // hypot < 1.5 on integers  <=>  d2 in {0,1,2}
                emit(wpos + len, x0, y0); len++;
            }
            if (len >= 5) {
                n_pts += len; n_paths++;
                if (WRITE) { wpos += len; if (wv.leader()) off[wpath + 1] = (int64_t)wpos; wpath++; }
            }
        }
    }
    if (!WRITE && wv.leader()) { A.comp_pts[c] = n_pts; A.comp_paths[c] = n_paths; }
}
