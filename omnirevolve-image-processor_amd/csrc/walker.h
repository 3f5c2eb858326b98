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

#ifndef ORIP_WALK_LEAD
#define ORIP_WALK_LEAD 16       // px by which a reloaded window is shifted in the direction of motion
#endif
#ifndef ORIP_WALK_BATCH
#define ORIP_WALK_BATCH 4u      // first look-up of a no-fresh run after this many pending states; the batch then doubles (at most one state per lane)
#endif
namespace walk_detail {
#if defined(__HIP_DEVICE_COMPILE__)
#define WT 64
#define WTP (WT + 4)
// One wavefront walks one component.  The walk itself is a strictly serial chain executed by a wave that has its SIMD to itself,
// so its speed is the number of instructions per step: the cursor is kept as global coordinates + linear index + offset inside a
// 64x64 LDS window of the state bytes, the eight neighbours are read by lanes 0..7 through a per-lane constant offset, and the
// previous pixel is excluded by its lane number (NEIGH8 is point-symmetric: the way back from a step in direction k is 7 - k).
struct Wave {
    int lane;
    u8* tile;                   // [WT][WTP]
    int tx0, ty0; bool have; unsigned nload;
    int px, py; unsigned pl;    // cursor
    int li;                     // cursor offset inside the window (set by probe)
    int noff;                   // this lane's neighbour offset inside the window (lanes 0..7)
    u8 myv;                     // state byte of this lane's neighbour (after probe)
    u8* st; int W, H;
    // Global stores inside the step loop would make every later s_waitcnt vmcnt(0) of the loop wait for a full store round trip, so
    // the loop only writes LDS: visited marks go to the window at once and to `mlist` (flushed to memory, all lanes, before anything
    // reads the state bytes from memory again); direction codes go to `sbuf` and reach the step log 64 at a time.
    unsigned* mlist; int nm;    // pending marks: (linear index << 4) | state byte
    int lead_x = 0, lead_y = 0; // window placement ahead of the cursor (ORIP_WALK_LEAD px in the direction of the last step)
    u8* sbuf; unsigned codes_done;
    __device__ Wave() : lane((int)(threadIdx.x & 63)), tx0(0), ty0(0), have(false), nload(0), px(0), py(0), pl(0), li(0), myv(0), st(nullptr), W(0), H(0) {
        __shared__ u8 lds_tile[WT * WTP];
        __shared__ unsigned lds_marks[64];
        __shared__ u8 lds_codes[64];
        tile = lds_tile; mlist = lds_marks; nm = 0; sbuf = lds_codes; codes_done = 0;
        const int k = lane & 7;
        noff = ((int)((0xA940u >> (2 * k)) & 3u) - 1) * WTP + (int)((0x9224u >> (2 * k)) & 3u) - 1;
    }
    __device__ void init(u8* st_, int W_, int H_) { st = st_; W = W_; H = H_; }
    __device__ bool leader() const { return lane == 0; }
    __device__ unsigned l0() const { return (unsigned)lane; }
    __device__ unsigned nl() const { return 64u; }
    __device__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0); }
    // lane 0 loads, everybody gets the value
    __device__ unsigned ld0(const unsigned* p) const { unsigned v = 0; if (lane == 0) v = *p; return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
    // one 16-byte record (state, cont, end, begin) with a single load
    __device__ void ld0_rec(const unsigned* p, unsigned& cont, unsigned& en, unsigned& begin) const {
        uint4 v = make_uint4(0, 0, 0, 0); if (lane == 0) v = *reinterpret_cast<const uint4*>(p);
        cont = (unsigned)__builtin_amdgcn_readfirstlane((int)v.y); en = (unsigned)__builtin_amdgcn_readfirstlane((int)v.z); begin = (unsigned)__builtin_amdgcn_readfirstlane((int)v.w);
    }
    __device__ void flush_marks() {
        if (nm) { if (lane < nm) { const unsigned e = mlist[lane]; st[e >> 4] = (u8)(e & 15u); } nm = 0; }
    }
    // marks of the walk so far are in memory (before the state bytes are read from memory: next scan, next window)
    __device__ void sync_marks() { flush_marks(); fence(); }
    __device__ void begin_codes() { codes_done = 0; }
    __device__ void put_code(u8* slog, unsigned room, unsigned idx, int k) {
        if (lane == 0) sbuf[idx & 63u] = (u8)k;
        if ((idx & 63u) == 63u) { const unsigned p = (idx & ~63u) + (unsigned)lane; if (p < room) slog[p] = sbuf[lane]; codes_done = idx + 1u; }
    }
    __device__ void finish_codes(u8* slog, unsigned room, unsigned steps) {
        if (steps > codes_done) { const unsigned p = codes_done + (unsigned)lane; if (p < steps && p < room) slog[p] = sbuf[p & 63u]; }
    }
    __device__ void load_tile(int cx, int cy) {
        flush_marks();
        fence();                                                      // earlier marks have reached memory
        tx0 = ((cx - WT / 2 + lead_x) >> 2) << 2; ty0 = cy - WT / 2 + lead_y;          // 4-byte aligned columns; the window leads in the direction of the last step
        const int y = ty0 + lane;
        u8* row = tile + lane * WTP;
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
    __device__ void set_cursor(int x, int y) { px = x; py = y; pl = (unsigned)y * (unsigned)W + (unsigned)x; lead_x = lead_y = 0; }
    // marks the cursor pixel itself (start of a walk): global memory and, when inside, the window
    __device__ void mark_cursor(u8 v) {
        if (lane == 0) {
            st[pl] = (u8)(v | ST_VIS);
            const int lx = px - tx0, ly = py - ty0;
            if (have && (unsigned)lx < (unsigned)WT && (unsigned)ly < (unsigned)WT) tile[ly * WTP + lx] = (u8)(v | ST_VIS);
        }
    }
    // neighbours of the cursor; kopp = NEIGH8 index of the previous pixel (8: none).  Masks over NEIGH8 indices.
    __device__ void probe(int kopp, unsigned& m_any, unsigned& m_unvis) {
        int lx = px - tx0, ly = py - ty0;
        if (!have || (unsigned)(lx - 1) > (unsigned)(WT - 3) || (unsigned)(ly - 1) > (unsigned)(WT - 3)) { load_tile(px, py); lx = px - tx0; ly = py - ty0; }
        li = ly * WTP + lx;
        unsigned v = 0;
        if (lane < 8) v = tile[li + noff];                            // out-of-image cells of the window hold 0; lanes >= 8 contribute nothing
        myv = (u8)v;
        // two ballots on single-bit tests, the rest on the scalar unit
        const unsigned m_fg = (unsigned)__ballot((v & ST_FG) != 0), m_vis = (unsigned)__ballot((v & ST_VIS) != 0);
        m_any = m_fg & ~(1u << kopp);                                  // kopp == 8 clears nothing
        m_unvis = m_any & ~m_vis;
    }
    __device__ u8 value_of(int k) const { return (u8)__builtin_amdgcn_readlane((int)myv, k); }
    // moves the cursor to neighbour k (after probe); mark: the neighbour becomes visited (it lies inside the window: the cursor is interior)
    __device__ void step(int k, bool mark) {
        const int dx = (int)((0x9224u >> (2 * k)) & 3u) - 1, dy = (int)((0xA940u >> (2 * k)) & 3u) - 1;
        px += dx; py += dy; pl = (unsigned)((int)pl + dy * W + dx);
        lead_x = dx * ORIP_WALK_LEAD; lead_y = dy * ORIP_WALK_LEAD;
        if (mark) {
            const u8 v = (u8)(value_of(k) | ST_VIS);
            if (lane == 0) { tile[li + dy * WTP + dx] = v; mlist[nm] = (pl << 4) | v; }
            if (++nm == 64) flush_marks();
        }
    }
    __device__ unsigned bcast(unsigned v, int src) const { return (unsigned)__shfl((int)v, src, 64); }
    __device__ int first(bool pred) const { unsigned long long m = __ballot(pred); return m ? __ffsll((long long)m) - 1 : -1; }
    // next q in [q0, e) whose pixel satisfies: (state & need) == need && !(state & ST_VIS)
    __device__ unsigned scan(const unsigned* lin, unsigned q0, unsigned e, u8 need) const {
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
    int px = 0, py = 0; unsigned pl = 0;
    u8* st = nullptr; int W = 0, H = 0;
    void init(u8* st_, int W_, int H_) { st = st_; W = W_; H = H_; }
    bool leader() const { return true; }
    unsigned l0() const { return 0u; }
    unsigned nl() const { return 1u; }
    void fence() const {}
    unsigned ld0(const unsigned* p) const { return *p; }
    void ld0_rec(const unsigned* p, unsigned& cont, unsigned& en, unsigned& begin) const { cont = p[1]; en = p[2]; begin = p[3]; }
    void set_cursor(int x, int y) { px = x; py = y; pl = (unsigned)y * (unsigned)W + (unsigned)x; }
    void mark_cursor(u8 v) { st[pl] = (u8)(v | ST_VIS); }
    void sync_marks() {}
    void begin_codes() {}
    void put_code(u8* slog, unsigned, unsigned idx, int k) { slog[idx] = (u8)k; }
    void finish_codes(u8*, unsigned, unsigned) {}
    void probe(int kopp, unsigned& m_any, unsigned& m_unvis) {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        m_any = m_unvis = 0;
        for (int k = 0; k < 8; k++) {
            int xx = px + dxs[k], yy = py + dys[k]; nv[k] = 0;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            u8 v = st[(size_t)yy * W + xx]; nv[k] = v;
            if ((v & ST_FG) && k != kopp) { m_any |= 1u << k; if (!(v & ST_VIS)) m_unvis |= 1u << k; }
        }
    }
    u8 value_of(int k) const { return nv[k]; }
    void step(int k, bool mark) {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        px += dxs[k]; py += dys[k]; pl = (unsigned)py * (unsigned)W + (unsigned)px;
        if (mark) st[pl] = (u8)(nv[k] | ST_VIS);
    }
    unsigned bcast(unsigned v, int) const { return v; }
    int first(bool pred) const { return pred ? 0 : -1; }
    unsigned scan(const unsigned* lin, unsigned q0, unsigned e, u8 need) const {
        for (unsigned q = q0; q < e; q++) { u8 v = st[lin[q]]; if (((v & need) == need) && !(v & ST_VIS)) return q; }
        return e;
    }
    void scan2(int dx, int dy, int& ox, int& oy, int& tx, int& ty) const { ox = dx; oy = dy; tx = dx; ty = dy; }
};
#endif
ORIP_HD inline int ffs8(unsigned m) { return __builtin_ctz(m); }
// NEIGH8 (dx,dy) of 04:12 as packed 2-bit fields (value + 1)
ORIP_HD inline int nbx(int k) { return (int)((0x9224u >> (2 * k)) & 3u) - 1; }
ORIP_HD inline int nby(int k) { return (int)((0xA940u >> (2 * k)) & 3u) - 1; }
}  // namespace walk_detail

// Entry reached from the record of entry `rec` at raw position f (f >= rec): follow continuations until f lies inside a record.
template <class WaveT>
ORIP_HD inline unsigned long long log_resolve(const unsigned* logbuf, const WaveT& wv, unsigned rec, unsigned long long f) {
    while (true) {
        unsigned cont, en, begin; wv.ld0_rec(&logbuf[4ull * rec], cont, en, begin);
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
    wv.init(st, W, H);
    const long long total_fg = A.total_fg[layer];
    const unsigned F = A.cap_factor;
    const unsigned log_base = F * b + 64u * c, log_cap = F * fg + 64u;
    const unsigned step_base = F * b + 256u * c, step_cap = F * fg + 256u;
    unsigned logcur = 0, stepcur = 0;
    bool over = false;
    unsigned long long d_w1 = 0, d_s1 = 0, d_w2 = 0, d_s2 = 0, d_hit = 0, d_det = 0;
    auto finish = [&](unsigned slot, unsigned long long len, unsigned n_own, unsigned sbeg, unsigned log_i1, unsigned R, unsigned flags) {
        if (wv.leader()) { WalkInfo wi; wi.len_kept = len >= 5 ? (unsigned)len : 0u; wi.n_own = n_own; wi.step_begin = sbeg; wi.log_i1 = log_i1; wi.R = R; wi.flags = flags; A.winfo[slot] = wi; }
    };
    // ---- phase 1: walks from endpoints (04:144-171); >= 2 points to be a path (04:168), >= 5 to survive vectorize_layer (04:224)
    const unsigned long long g1 = (unsigned long long)(total_fg * 2);
    for (unsigned q = wv.scan(A.lin, b, e, ST_FG | ST_END); q < e; q = wv.scan(A.lin, q + 1, e, ST_FG | ST_END)) {
        unsigned s = A.lin[q];
        wv.set_cursor((int)(s % W), (int)(s / W));
        const unsigned sbeg = step_base + stepcur;
        u8* slog = A.steplog + (size_t)sbeg; const unsigned room = step_cap - stepcur;
        wv.mark_cursor(ST_FG | ST_END);
        wv.begin_codes();
        unsigned steps = 0; int kopp = 8; d_w1++;
        while (true) {
            unsigned m_any, m_unvis;
            wv.probe(kopp, m_any, m_unvis);
            if (!m_unvis) break;
            const int k = ffs8(m_unvis);
            const u8 v = wv.value_of(k);
            if (steps < room) wv.put_code(slog, room, steps, k); else over = true;
            steps++;
            wv.step(k, true);
            kopp = 7 - k;
            if (v & (ST_JUN | ST_END)) break;
            if ((unsigned long long)steps > g1) break;       // guard (04:163): counts the steps that went on
        }
        stepcur += steps; d_s1 += steps;
        wv.finish_codes(slog, room, steps);
        wv.sync_marks();  // marks of this walk are complete before the next scan
        finish(2u * b + (q - b), 1ull + steps, steps, sbeg, 0u, 0u, 0u);
    }
    // ---- phase 2: leftovers / cycles (04:174-205)
    auto log_pos = [&](unsigned i, unsigned long long R, int& ox, int& oy) {   // position R steps after logged state i
        unsigned long long f = log_resolve(A.logbuf, wv, i, (unsigned long long)i + R);
        unsigned l = wv.ld0(&A.logbuf[4ull * f]) >> 3;
        ox = (int)(l % (unsigned)W); oy = (int)(l / (unsigned)W);
    };
    // A walk that finds no fresh neighbour is on a trajectory that only depends on its state (pixel, incoming direction), so its
    // states are looked up in / added to the memo.  The memo lives in HBM and a look-up per step would put one full memory latency
    // on every step of a strictly serial chain; instead the walk runs ahead on the LDS window and parks its no-fresh states in the
    // lanes (lane j = j-th pending state).  flush() then does what the reference order requires for all of them at once: the first
    // pending state that is already known (to an older record, to this run, or to an earlier pending state) ends the walk there and
    // the steps taken after it are dropped; otherwise all of them become provisional entries of this run.
    const unsigned g2 = fg * 4u;                                  // guard of a leftover walk (04:199)
    const unsigned nbatch0 = wv.nl() < ORIP_WALK_BATCH ? wv.nl() : ORIP_WALK_BATCH;
    for (unsigned q = wv.scan(A.lin, b, e, ST_FG); q < e && !over; q = wv.scan(A.lin, q + 1, e, ST_FG)) {
        unsigned s = A.lin[q];
        const int x0 = (int)(s % W), y0 = (int)(s / W);
        wv.set_cursor(x0, y0);
        const unsigned sbeg = step_base + stepcur;
        u8* slog = A.steplog + (size_t)sbeg; const unsigned room = step_cap - stepcur;
        { u8 sv = st[s]; wv.mark_cursor(sv); }
        wv.begin_codes();
        d_w2++;
        unsigned steps = 0;                   // steps taken = own points - 1 = value of the reference's guard counter at its check
        unsigned long long tail_len = 0; unsigned tail_i1 = 0, tail_R = 0;
        unsigned nofresh = 0;                 // no-fresh states logged since the last fresh pixel: entries [run_begin, run_begin + nofresh)
        unsigned nb = 0, myS = 0, steps_b = 0;                     // pending states; steps before the first of them
        unsigned nbatch = nbatch0;                                 // memo hits come early in a run or not for a while: 4, 8, 16, ... pending states per look-up
        // 0: nothing known, pending states committed; 1: the walk ended in a jump; 2: log overflow
        auto flush = [&]() -> int {
            if (!nb) return 0;
            const unsigned run_begin = log_base + logcur;
            const unsigned me = wv.l0();
            const bool act = me < nb;
            unsigned mi = 0, ld = 0, en = 0;
            if (act) mi = memo[myS];
            if (act && mi) { ld = A.logbuf[4ull * (mi - 1)]; en = A.logbuf[4ull * (mi - 1) + 2]; }
            const bool hit_old = act && mi && ld == myS && (en != 0 || (mi - 1 >= run_begin && mi - 1 < run_begin + nofresh));
            bool dup = false; unsigned dsrc = 0;                                             // latest earlier pending state equal to mine
            for (unsigned jp = 0; jp + 1 < nb; jp++) { const unsigned sj = wv.bcast(myS, (int)jp); if (act && me > jp && myS == sj) { dup = true; dsrc = jp; } }
            const int js = wv.first(hit_old || dup);
            const unsigned ncommit = js < 0 ? nb : (unsigned)js;
            if (logcur + nofresh + ncommit > log_cap) { over = true; nb = 0; return 2; }
            if (me < ncommit) { const unsigned idx = run_begin + nofresh + me; A.logbuf[4ull * idx] = myS; A.logbuf[4ull * idx + 2] = 0u; memo[myS] = idx + 1; }
            if (js < 0) { wv.fence(); nofresh += nb; nb = 0; return 0; }
            // the reference would have stopped at pending state js: roll the step counter back to it
            const unsigned ev_mi = wv.bcast(dup ? run_begin + nofresh + dsrc + 1u : mi, js), ev_en = wv.bcast(dup ? 0u : en, js);
            nofresh += (unsigned)js;
            steps = steps_b + (unsigned)js + 1;
            const unsigned i = ev_mi - 1;
            const unsigned long long R = (unsigned long long)g2 + 1ull - steps;             // points still to come until the guard fires
            if (ev_en != 0) d_hit++; else d_det++;
            // committed trajectory of an earlier walk: this run's own no-fresh states become a transient record that runs into entry i, so
            // later walks can jump from them too.  A state of this very run: the cycle [i, run_begin + nofresh) is closed.
            if (nofresh) {
                const unsigned end = run_begin + nofresh;
                for (unsigned t = me; t < nofresh; t += wv.nl()) { unsigned* p = A.logbuf + 4ull * (run_begin + t); p[1] = i; p[2] = end; p[3] = run_begin; }
                logcur += nofresh;
            }
            wv.fence();                                       // provisional entries and the record are in memory before anything reads the log
            tail_i1 = ev_mi; tail_R = (unsigned)R; tail_len = R;
            int ex, ey; log_pos(i, R, ex, ey); wv.set_cursor(ex, ey);
            nb = 0;
            return 1;
        };
        int ended = 0, kopp = 8;
        while (true) {
            unsigned m_any, m_unvis;
            wv.probe(kopp, m_any, m_unvis);
            const bool fresh = m_unvis != 0;
            const unsigned m = fresh ? m_unvis : m_any;
            if (!m) break;
            if (fresh && nb) { ended = flush(); if (ended) break; }
            const int k = ffs8(m);
            if (!fresh && nb == 0) steps_b = steps;
            if (steps < room) wv.put_code(slog, room, steps, k); else over = true;
            steps++;
            wv.step(k, fresh);
            kopp = 7 - k;
            if (wv.px == x0 && wv.py == y0) break;
            if (steps > g2) break;
            if (fresh) { nofresh = 0; nbatch = nbatch0; continue; }
            if (wv.l0() == nb) myS = (wv.pl << 3) | (unsigned)k;
            nb++;
            if (nb == nbatch) { ended = flush(); if (ended) break; nbatch = nbatch * 2u < wv.nl() ? nbatch * 2u : wv.nl(); }
        }
        if (!ended && nb) ended = flush();
        wv.finish_codes(slog, room, steps);
        wv.sync_marks();
        if (ended == 2) break;
        stepcur += steps; d_s2 += steps;
        unsigned long long len = 1ull + steps + tail_len;
        unsigned flags = 0;
        if (len >= 2) {
            int ddx = x0 - wv.px, ddy = y0 - wv.py;
            if (ddx * ddx + ddy * ddy < 3) { flags = 1; len++; }   // hypot < 1.5 on integers  <=>  d2 in {0,1,2}: the start is appended again
        }
        finish(2u * b + fg + (q - b), len >= 2 ? len : 0, steps, sbeg, tail_i1, tail_R, flags);
    }
    if (over && wv.leader()) *A.overflow = 1;
    if (A.dbg && wv.leader()) { unsigned long long* d = A.dbg + 8ull * c; d[0] = d_w1; d[1] = d_s1; d[2] = d_w2; d[3] = d_s2; d[4] = d_hit; d[5] = d_det; d[6] = wv.nload; d[7] = (unsigned long long)fg; }
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
