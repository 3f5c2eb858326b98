// csrc/walker.h -- the centerline walker of stage 04 (04_find_contours.py trace_centerlines, 04:137-205).
//
// trace_component: ONE WAVEFRONT walks one connected component with the reference's exact serial semantics and records
// what it did instead of writing points:
//   * the 8 neighbours of the current pixel are probed by lanes 0..7 from a 64x64 LDS window of the state plane; the
//     NEIGH8-ordered choice is a ballot + find-first-set; the raster-ordered scans for the next endpoint / leftover pixel
//     test 64 list entries per step; every decision is wave-uniform, lane 0 does all stores;
//   * every step is logged as a 3-bit direction code (step log), every walk leaves one WalkInfo record;
//   * bounce memo: once a walk has no unvisited neighbour left it follows the deterministic map (prev,cur) -> next over
//     visited pixels.  For a state without unvisited neighbours that map never changes again (the visited set only grows),
//     so a trajectory recorded once stays valid.  No-fresh states are logged (state log) and indexed (memo); reaching a state
//     of the walk's own current run closes a cycle exactly (no Brent overhead), reaching a committed state of an earlier walk
//     means "the rest is that trajectory".  Either way the guard-bounded tail (up to 4*fg+1 points) is not walked.
// write_walk: one wavefront per recorded walk turns the record into points: a wave prefix sum over the direction codes for the
// walk's own steps, and an indexed copy out of the state log for the tail.  No second serial pass, no visited state.
//
// The same source is compiled with g++ by tests/host/walk_harness.cpp, where a "wave" is emulated by plain loops, so the walk
// logic (phases, guards, memo, cycle closing) is unit-tested on the CPU as well (test infrastructure).
#pragma once
#include <cstdint>
#include "../../include/orip.h"
#if defined(__HIPCC__)
#define ORIP_HD __host__ __device__
#else
#define ORIP_HD
struct int2 { int x, y; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
#endif
typedef uint8_t u8;
#define ST_FG 1
#define ST_VIS 2
#define ST_END 4
#define ST_JUN 8

struct WalkInfo {            // one per potential walk: slot 2b+(q-b) for the endpoint walk starting at list index q, 2b+fg+(q-b) for a phase-2 walk
    unsigned len_kept;       // total points if the path is kept (>= 5 points, 04:224), else 0
    unsigned n_own;          // steps walked (points after the start pixel) before the recorded tail
    unsigned step_begin;     // first direction code in the step log
    unsigned log_i1;         // tail: state-log index + 1 of the state the walk stood on when it jumped (0 = no tail)
    unsigned R;              // tail length in points
    unsigned flags;          // bit0: append the start point again (04:203-204)
};

struct WalkArgs {
    int H, W; int64_t plane;
    u8* st;                            // [K,H,W] state bytes
    const unsigned* keys; const unsigned* lin; const unsigned* comp_start; unsigned nc;
    long long total_fg[ORIP_MAX_LAYERS];
    const unsigned* comp_order;        // optional: component processed by wave i (largest first), or nullptr
    unsigned* memo;                    // [K,H,W,8]: state-log index + 1 of the state (pixel, incoming direction), 0 = unknown
    unsigned* logbuf;                  // state log, 4 words per entry: (lin << 3 | dir), cont, end (0 while provisional), begin.  After entry end-1 the
                                       // trajectory continues at entry `cont`: inside the record (cont >= begin) it is a cycle, otherwise the record is a
                                       // transient that runs into an older record
    u8* steplog;                       // direction code of every step
    unsigned cap_factor;               // regions of component c (b = comp_start[c], fg = size): state log [F*b + 64*c, + F*fg + 64), step log [F*b + 256*c, + F*fg + 256)
    WalkInfo* winfo;                   // [2*M]
    int* overflow;                     // set when a region was too small (host retries with a larger cap_factor)
    // write pass
    const unsigned long long* pts_off; const unsigned* path_off;   // exclusive scans over winfo (len_kept, kept)
    unsigned long long layer_pts_base[ORIP_MAX_LAYERS]; unsigned layer_path_base[ORIP_MAX_LAYERS];
    int32_t* pts[ORIP_MAX_LAYERS]; int64_t* off[ORIP_MAX_LAYERS];
    unsigned long long* dbg;           // optional counters, 8 per component
};

namespace walk_detail {
#if defined(__HIP_DEVICE_COMPILE__)
#define WT 64
struct Wave {
    int lane;
    u8 (*tile)[WT + 4];
    int tx0, ty0; bool have; unsigned nload;
    __device__ Wave() : lane((int)(threadIdx.x & 63)), tx0(0), ty0(0), have(false), nload(0) {
        __shared__ u8 lds_tile[WT][WT + 4];
        tile = lds_tile;
    }
    __device__ bool leader() const { return lane == 0; }
    __device__ unsigned l0() const { return (unsigned)lane; }
    __device__ unsigned nl() const { return 64u; }
    __device__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0); }
    // lane 0 loads, everybody gets the value: lane 0 is also the only lane that stores these words, and a lane always sees its own earlier stores
    __device__ unsigned ld0(const unsigned* p) const { unsigned v = 0; if (lane == 0) v = *p; return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ void load_tile(const u8* st, int W, int H, int cx, int cy) {
        fence();                                                      // earlier marks have reached memory
        tx0 = ((cx - WT / 2) >> 2) << 2; ty0 = cy - WT / 2;          // 4-byte aligned columns
        const int y = ty0 + lane;
        u8* row = tile[lane];
        if (y < 0 || y >= H) { for (int j = 0; j < WT; j += 4) *reinterpret_cast<uint32_t*>(row + j) = 0u; }
        else if ((W & 3) == 0 && tx0 >= 0 && tx0 + WT <= W) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(st + (size_t)y * W + tx0);
#pragma unroll
            for (int j = 0; j < WT / 4; j++) *reinterpret_cast<uint32_t*>(row + 4 * j) = src[j];
        } else {
            for (int j = 0; j < WT; j++) { int x = tx0 + j; row[j] = (x >= 0 && x < W) ? st[(size_t)y * W + x] : (u8)0; }
        }
        have = true; nload++;
        fence();
    }
    // probe the 8 neighbours of (px,py); returns masks over NEIGH8 indices
    __device__ void probe(const u8* st, int W, int H, int px, int py, int pvx, int pvy, unsigned& m_any, unsigned& m_unvis, u8& myv) {
        if (!have || px - tx0 < 1 || px - tx0 > WT - 2 || py - ty0 < 1 || py - ty0 > WT - 2) load_tile(st, W, H, px, py);
        u8 v = 0; bool any = false;
        if (lane < 8) {
            int xx = px + (int)((0x9224u >> (2 * lane)) & 3u) - 1, yy = py + (int)((0xA940u >> (2 * lane)) & 3u) - 1;
            v = tile[yy - ty0][xx - tx0];                         // out-of-image cells of the window hold 0
            any = (v & ST_FG) && !(xx == pvx && yy == pvy);
        }
        myv = v;
        m_any = (unsigned)(__ballot(any) & 0xffu);
        m_unvis = (unsigned)(__ballot(any && !(v & ST_VIS)) & 0xffu);
    }
    __device__ u8 value_of(u8 myv, int k) const { return (u8)__shfl((int)myv, k, 64); }
    // lane 0 marks pixel (x,y) visited: global memory and, when inside, the window
    __device__ void mark(u8* st, int W, int x, int y, u8 v) {
        if (lane == 0) {
            st[(size_t)y * W + x] = (u8)(v | ST_VIS);
            if (have && x >= tx0 && x < tx0 + WT && y >= ty0 && y < ty0 + WT) tile[y - ty0][x - tx0] = (u8)(v | ST_VIS);
        }
    }
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
    // inclusive prefix sums of (dx,dy) over the lanes; (ox,oy) = the lane's prefix, (tx,ty) = wave total
    __device__ void scan2(int dx, int dy, int& ox, int& oy, int& tx, int& ty) const {
        for (int o = 1; o < 64; o <<= 1) { int ax = __shfl_up(dx, o, 64), ay = __shfl_up(dy, o, 64); if (lane >= o) { dx += ax; dy += ay; } }
        ox = dx; oy = dy; tx = __shfl(dx, 63, 64); ty = __shfl(dy, 63, 64);
    }
};
#else
struct Wave {
    u8 nv[8]; unsigned nload = 0;
    bool leader() const { return true; }
    unsigned l0() const { return 0u; }
    unsigned nl() const { return 1u; }
    void fence() const {}
    unsigned ld0(const unsigned* p) const { return *p; }
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
    void mark(u8* st, int W, int x, int y, u8 v) { st[(size_t)y * W + x] = (u8)(v | ST_VIS); }
    unsigned scan(const unsigned* lin, const u8* st, unsigned q0, unsigned e, u8 need) const {
        for (unsigned q = q0; q < e; q++) { u8 v = st[lin[q]]; if (((v & need) == need) && !(v & ST_VIS)) return q; }
        return e;
    }
    void scan2(int dx, int dy, int& ox, int& oy, int& tx, int& ty) const { ox = dx; oy = dy; tx = dx; ty = dy; }
};
#endif
ORIP_HD inline int ffs8(unsigned m) { int k = 0; while (!((m >> k) & 1u)) k++; return k; }
// NEIGH8 (dx,dy) of 04:12 as packed 2-bit fields (value + 1)
ORIP_HD inline int nbx(int k) { return (int)((0x9224u >> (2 * k)) & 3u) - 1; }
ORIP_HD inline int nby(int k) { return (int)((0xA940u >> (2 * k)) & 3u) - 1; }
}  // namespace walk_detail

// Entry reached from the record of entry `rec` at raw position f (f >= rec): follow continuations until f lies inside a record.
template <class WaveT>
ORIP_HD inline unsigned long long log_resolve(const unsigned* logbuf, const WaveT& wv, unsigned rec, unsigned long long f) {
    while (true) {
        const unsigned cont = wv.ld0(&logbuf[4ull * rec + 1]), en = wv.ld0(&logbuf[4ull * rec + 2]), begin = wv.ld0(&logbuf[4ull * rec + 3]);
        if (f < en) return f;
        if (cont >= begin) return (unsigned long long)cont + (f - en) % (unsigned long long)(en - cont);   // cycle
        f = (unsigned long long)cont + (f - en); rec = cont;                                               // transient -> older record
    }
}

ORIP_HD inline void trace_component(const WalkArgs& A, unsigned c) {
    using namespace walk_detail;
    Wave wv;
    const unsigned b = A.comp_start[c], e = A.comp_start[c + 1], fg = e - b;
    const int layer = (int)(A.keys[b] >> 26);
    u8* st = A.st + A.plane * layer;
    unsigned* memo = A.memo + (size_t)A.plane * layer * 8;
    const int W = A.W, H = A.H;
    const long long fg_comp = (long long)fg, total_fg = A.total_fg[layer];
    const unsigned F = A.cap_factor;
    const unsigned log_base = F * b + 64u * c, log_cap = F * fg + 64u;
    const unsigned step_base = F * b + 256u * c, step_cap = F * fg + 256u;
    unsigned logcur = 0, stepcur = 0;
    bool over = false;
    unsigned long long d_w1 = 0, d_s1 = 0, d_w2 = 0, d_s2 = 0, d_hit = 0, d_det = 0;
    auto put_step = [&](int k) { if (stepcur < step_cap) { if (wv.leader()) A.steplog[(size_t)step_base + stepcur] = (u8)k; } else over = true; stepcur++; };
    auto finish = [&](unsigned slot, unsigned long long len, unsigned n_own, unsigned sbeg, unsigned log_i1, unsigned R, unsigned flags) {
        if (wv.leader()) { WalkInfo wi; wi.len_kept = len >= 5 ? (unsigned)len : 0u; wi.n_own = n_own; wi.step_begin = sbeg; wi.log_i1 = log_i1; wi.R = R; wi.flags = flags; A.winfo[slot] = wi; }
    };
    // ---- phase 1: walks from endpoints (04:144-171); >= 2 points to be a path (04:168), >= 5 to survive vectorize_layer (04:224)
    for (unsigned q = wv.scan(A.lin, st, b, e, ST_FG | ST_END); q < e; q = wv.scan(A.lin, st, q + 1, e, ST_FG | ST_END)) {
        unsigned s = A.lin[q];
        int px = (int)(s % W), py = (int)(s / W), pvx = -1, pvy = -1;
        unsigned long long len = 1; const unsigned sbeg = step_base + stepcur;
        wv.mark(st, W, px, py, ST_FG | ST_END);
        long long guard = 0; d_w1++;
        while (true) {
            unsigned m_any, m_unvis; u8 myv;
            wv.probe(st, W, H, px, py, pvx, pvy, m_any, m_unvis, myv);
            if (!m_unvis) break;
            int k = ffs8(m_unvis);
            int nx = px + nbx(k), ny = py + nby(k);
            u8 v = wv.value_of(myv, k);
            put_step(k); len++; d_s1++;
            wv.mark(st, W, nx, ny, v);
            pvx = px; pvy = py; px = nx; py = ny;
            if (v & (ST_JUN | ST_END)) break;
            guard++;
            if (guard > total_fg * 2) break;
        }
        wv.fence();       // marks of this walk are complete before the next scan
        finish(2u * b + (q - b), len, (unsigned)(len - 1), sbeg, 0u, 0u, 0u);
    }
    // ---- phase 2: leftovers / cycles (04:174-205)
    auto log_pos = [&](unsigned i, unsigned long long R, int& ox, int& oy) {   // position R steps after logged state i
        unsigned long long f = log_resolve(A.logbuf, wv, i, (unsigned long long)i + R);
        unsigned l = wv.ld0(&A.logbuf[4ull * f]) >> 3;
        ox = (int)(l % (unsigned)W); oy = (int)(l / (unsigned)W);
    };
    for (unsigned q = wv.scan(A.lin, st, b, e, ST_FG); q < e; q = wv.scan(A.lin, st, q + 1, e, ST_FG)) {
        unsigned s = A.lin[q];
        const int x0 = (int)(s % W), y0 = (int)(s / W);
        int px = x0, py = y0, pvx = -1, pvy = -1;
        unsigned long long len = 1; const unsigned sbeg = step_base + stepcur;
        { u8 sv = st[s]; wv.mark(st, W, px, py, sv); }
        long long guard = 0; d_w2++;
        unsigned n_own = 0, tail_i1 = 0, tail_R = 0;
        unsigned nofresh = 0;                 // no-fresh states logged since the last fresh pixel: entries [run_begin, run_begin + nofresh)
        while (true) {
            unsigned m_any, m_unvis; u8 myv;
            wv.probe(st, W, H, px, py, pvx, pvy, m_any, m_unvis, myv);
            bool fresh = m_unvis != 0;
            unsigned m = fresh ? m_unvis : m_any;
            if (!m) break;
            int k = ffs8(m);
            int nx = px + nbx(k), ny = py + nby(k);
            put_step(k); len++; n_own++; d_s2++;
            if (fresh) wv.mark(st, W, nx, ny, wv.value_of(myv, k));
            pvx = px; pvy = py; px = nx; py = ny;
            if (px == x0 && py == y0) break;
            guard++;
            if (guard > fg_comp * 4) break;
            if (fresh) { nofresh = 0; continue; }
            const unsigned run_begin = log_base + logcur;
            const unsigned state = (((unsigned)py * (unsigned)W + (unsigned)px) << 3) | (unsigned)k;
            const unsigned mi = wv.ld0(&memo[state]);
            bool jumped = false;
            if (mi) {
                const unsigned i = mi - 1;
                const unsigned ld = wv.ld0(&A.logbuf[4ull * i]), en = wv.ld0(&A.logbuf[4ull * i + 2]);
                if (ld == state) {
                    const unsigned long long R = (unsigned long long)(fg_comp * 4 + 1 - guard);     // points still to come until the guard fires
                    if (en != 0) {                                    // committed trajectory of an earlier walk
                        d_hit++;
                        // this run's own no-fresh states become a transient record that runs into entry i, so later walks can jump from them too
                        if (nofresh) { const unsigned end = run_begin + nofresh; if (wv.leader()) for (unsigned t = 0; t < nofresh; t++) { unsigned* p = A.logbuf + 4ull * (run_begin + t); p[1] = i; p[2] = end; p[3] = run_begin; } logcur += nofresh; }
                        tail_i1 = mi; tail_R = (unsigned)R; jumped = true;
                    } else if (i >= run_begin && i < run_begin + nofresh) {   // a state of this very run: the cycle [i, run_begin + nofresh) is closed
                        d_det++;
                        const unsigned end = run_begin + nofresh;
                        if (wv.leader()) for (unsigned t = 0; t < nofresh; t++) { unsigned* p = A.logbuf + 4ull * (run_begin + t); p[1] = i; p[2] = end; p[3] = run_begin; }
                        logcur += nofresh;
                        tail_i1 = mi; tail_R = (unsigned)R; jumped = true;
                    }
                    if (jumped) { log_pos(i, R, px, py); len += R; }
                }
            }
            if (jumped) break;
            if (logcur + nofresh < log_cap) {                         // log the state (provisional until its run closes a cycle)
                const unsigned idx = run_begin + nofresh;
                if (wv.leader()) { A.logbuf[4ull * idx] = state; A.logbuf[4ull * idx + 2] = 0u; memo[state] = idx + 1; }
            } else over = true;
            nofresh++;
        }
        wv.fence();
        unsigned flags = 0;
        if (len >= 2) {
            int ddx = x0 - px, ddy = y0 - py;
            if (ddx * ddx + ddy * ddy < 3) { flags = 1; len++; }   // hypot < 1.5 on integers  <=>  d2 in {0,1,2}: the start is appended again
        }
        finish(2u * b + fg + (q - b), len >= 2 ? len : 0, n_own, sbeg, tail_i1, tail_R, flags);
    }
    if (over && wv.leader()) *A.overflow = 1;
    if (A.dbg && wv.leader()) { unsigned long long* d = A.dbg + 8ull * c; d[0] = d_w1; d[1] = d_s1; d[2] = d_w2; d[3] = d_s2; d[4] = d_hit; d[5] = d_det; d[6] = wv.nload; d[7] = (unsigned long long)fg_comp; }
}

// Points of one recorded walk (winfo slot `slot`), written by one wavefront.
ORIP_HD inline void write_walk(const WalkArgs& A, unsigned slot) {
    using namespace walk_detail;
    const WalkInfo wi = A.winfo[slot];
    if (!wi.len_kept) return;
    Wave wv;
    // component of the slot: slots [2b, 2e) belong to the component whose list range is [b, e)
    unsigned lo = 0, hi = A.nc;
    while (lo < hi) { unsigned mid = (lo + hi) >> 1; if (2u * A.comp_start[mid + 1] <= slot) lo = mid + 1; else hi = mid; }
    const unsigned c = lo, b = A.comp_start[c], fg = A.comp_start[c + 1] - b;
    const unsigned rel = slot - 2u * b;
    const unsigned q = b + (rel >= fg ? rel - fg : rel);
    const int layer = (int)(A.keys[b] >> 26);
    const int W = A.W;
    const unsigned s = A.lin[q];
    const int x0 = (int)(s % (unsigned)W), y0 = (int)(s / (unsigned)W);
    int2* out = reinterpret_cast<int2*>(A.pts[layer]) + (A.pts_off[slot] - A.layer_pts_base[layer]);
    if (wv.leader()) {
        out[0] = make_int2(x0, y0);
        A.off[layer][A.path_off[slot] - A.layer_path_base[layer] + 1] = (int64_t)(A.pts_off[slot] - A.layer_pts_base[layer] + wi.len_kept);
        if (wi.flags & 1u) out[wi.len_kept - 1] = make_int2(x0, y0);
    }
    // own steps: prefix sums of the direction vectors
    int cx = x0, cy = y0;
    for (unsigned base = 0; base < wi.n_own; base += wv.nl()) {
        unsigned t = base + wv.l0();
        int dx = 0, dy = 0;
        if (t < wi.n_own) { int k = A.steplog[(size_t)wi.step_begin + t]; dx = nbx(k); dy = nby(k); }
        int ox, oy, tx, ty;
        wv.scan2(dx, dy, ox, oy, tx, ty);
        if (t < wi.n_own) out[1 + t] = make_int2(cx + ox, cy + oy);
        cx += tx; cy += ty;
    }
    // tail: recorded trajectory, entry i+1+j, wrapping from `end` to `cyc_begin`
    if (wi.log_i1) {
        // the tail is the trajectory after entry i: a chain of record pieces (rest of i's record, then the records it continues
        // into) that ends in a cycle.  The chain is walked once per wave; each piece is a contiguous run of log entries.
        const unsigned i = wi.log_i1 - 1;
        int2* o2 = out + 1 + wi.n_own;
        unsigned long long done = 0, f0 = (unsigned long long)i + 1; unsigned rec = i;
        while (done < wi.R) {
            const unsigned cont = A.logbuf[4ull * rec + 1], en = A.logbuf[4ull * rec + 2], begin = A.logbuf[4ull * rec + 3];
            if (f0 < en) {
                unsigned long long take = en - f0; if (take > wi.R - done) take = wi.R - done;
                for (unsigned long long j = wv.l0(); j < take; j += wv.nl()) { unsigned l = A.logbuf[4ull * (f0 + j)] >> 3; o2[done + j] = make_int2((int)(l % (unsigned)W), (int)(l / (unsigned)W)); }
                done += take;
                if (done >= wi.R) break;
            }
            if (cont >= begin) {           // cycle [cont, en): the rest of the tail
                const unsigned long long lam = en - cont, rest = wi.R - done;
                for (unsigned long long j = wv.l0(); j < rest; j += wv.nl()) { unsigned l = A.logbuf[4ull * (cont + j % lam)] >> 3; o2[done + j] = make_int2((int)(l % (unsigned)W), (int)(l / (unsigned)W)); }
                break;
            }
            f0 = cont; rec = cont;         // transient record exhausted: continue in the older record
        }
    }
}
