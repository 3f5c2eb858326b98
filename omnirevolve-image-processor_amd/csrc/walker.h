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
// write_chunk: the point array of a layer is cut into equal chunks, one wavefront each; a chunk turns the records of the walks it covers
// into points: a wave prefix sum over the direction codes for a walk's own steps, an indexed copy out of the state log for its tail.
// No second serial pass, no visited state.
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
// state byte of a skeleton pixel (state plane in memory and, plus two flag bits, the walker's LDS window): the two bits every step tests are
// the top ones, so that "foreground" is `byte < 0` and "foreground, not visited" is `byte < -64` on the sign-extended byte (walker.h: Wave::run)
#define ST_FG 0x80
#define ST_VIS 0x40
#define ST_JUN 2
#define ST_END 1
// Forced stretches.  A skeleton pixel with exactly two neighbours leaves a walk that arrives from one of them no choice (04:150,180:
// the only neighbour that is not `prev`), whatever has been visited.  raster04.hip lists the maximal chains of such pixels that are at
// least ORIP_CHAIN_MIN long (cpix: their pixel indices in chain order, a sentinel on either side; cref: the position of a chain pixel in
// cpix) and flags them in the state plane: ST_CHAIN on every pixel of a listed chain, ST_CHAIN_END (the window's "stop here" bit) on its
// two end pixels.  A leftover walk that steps onto an end pixel leaves the stepping loop there and takes the stretch ahead in one go
// (trace_component: chain_jump): 64 pixels per round instead of one step per ~350 cycles.
#define ST_DEG2 0x04
#define ST_CHAIN 0x08
#define ST_CHAIN_END 0x10
#define ORIP_CHAIN_SENTINEL 0xffffffffu
#ifndef ORIP_CHAIN_MIN
#define ORIP_CHAIN_MIN 24
#endif

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
    unsigned* log_used;                // optional: state-log entries of component c in use when its trace ended (the walk-coded lists read their pixels)
    const unsigned* cref;              // optional, [K,H,W]: position of a chain pixel in cpix (only read where the state byte carries ST_CHAIN)
    const unsigned* cpix;              // chain pixels (index inside their layer's plane), chain by chain, ORIP_CHAIN_SENTINEL before and after each
    // write pass
    const unsigned long long* pts_off; const unsigned* path_off;   // exclusive scans over winfo (len_kept, kept)
    unsigned long long layer_pts_base[ORIP_MAX_LAYERS]; unsigned layer_path_base[ORIP_MAX_LAYERS];
    int32_t* pts[ORIP_MAX_LAYERS]; int64_t* off[ORIP_MAX_LAYERS];
    unsigned long long* dbg;           // optional counters, 32 per component
};

#ifndef ORIP_WALK_LEAD
#define ORIP_WALK_LEAD 16       // px by which a reloaded window is shifted in the direction of motion
#endif
#ifndef ORIP_WALK_BATCH
#define ORIP_WALK_BATCH 16u     // first look-up of a no-fresh run after this many pending states; the batch then doubles (at most one state per lane).  16 since the duplicate search of a batch has its hashed pre-check (r03: 77.1 vs 77.8 ms per step with 4)
#endif
namespace walk_detail {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ORIP_WALK_PROF)
#define WPROF_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
#else
#define WPROF_NOW() 0ull
#endif
#define WPROF_ADD(acc, t0) do { (acc) += WPROF_NOW() - (t0); } while (0)

// Why the stepping of a leftover walk comes back to its caller (Wave::run)
enum { EV_DEAD = 1,      // the cursor has no neighbour to go to (or, on the GPU, stands on a flagged window cell: ring / start pixel)
       EV_PEND = 2,      // a fresh step lies ahead while no-fresh states are pending: look them up first
       EV_LIMIT = 3,     // `steps` reached `limit` (code window full, guard, log room)
       EV_HOME = 4,      // back on the start pixel (04:196)
       EV_BATCH = 5 };   // nb reached nbatch
struct Hot { unsigned steps, limit, nb, nbatch, allow; };   // allow = NEIGH8 directions open to the next step: 0xff without the way back

#if defined(__HIP_DEVICE_COMPILE__)
#define WT 64
#define WTP (WT + 4)
// window bytes (LDS) = state bytes (bit7 foreground, bit6 visited, bit1 junction, bit0 endpoint) plus two flags:
//   bit5 ring cell of the window, bit4 start pixel of the walk (the cursor may not probe from / walk through such a cell)
#define WB_FG ST_FG
#define WB_VIS ST_VIS
#define WB_RING 0x20
#define WB_HOME 0x10
#define WB_JUN ST_JUN
#define WB_END ST_END
// One wavefront walks one component.  The walk is a strictly serial chain executed by a wave that has its SIMD to itself.  Measured on
// MI355X for such a wave (tools/walkbench): every instruction ~4.5 cycles, a not-taken branch +10, a taken one 20, a scalar instruction
// that consumes an SGPR written by a vector instruction (ballot, v_readlane) +16..20, an LDS read 44.  r01's compiled loop spent ~830
// cycles per step (~95 instructions, 15 branches).  The stepping loop of the leftover walks is therefore ONE hand-written asm
// statement (Wave::run, ~39 instructions, 4 not-taken branches per step, ~300 cycles in the same benchmark):
//   * the cursor is an LDS offset `li` plus the global linear index `pl`, both advanced by per-lane constants fetched with v_readlane;
//   * lanes 0..7 read the eight neighbours from a 64x64 LDS window; the read of the NEXT probe is issued as soon as the step is
//     decided (two copies of the loop body with the registers swapped), so its latency hides behind the bookkeeping of this step;
//   * lane 8 reads the cursor cell itself with another compare threshold: bit 8 of the second ballot says whether the cell carries a
//     flag (window ring -> the window must be re-placed; start pixel -> the walk is home), folded into the "no neighbour" exit;
//   * the visited mark is written back by the lane that read the neighbour (every lane rewrites its byte, lane k with the bit set):
//     no exec-mask changes; every step leaves ONE record ((pl << 3 | k) << 2 | fresh << 1) in lane (step & 63) of a VGPR
//     (v_writelane): direction codes, visited marks for the state plane and pending no-fresh states are all decoded from it by the
//     64 lanes in parallel when the loop is left (at least every 64 steps).
struct Wave {
    int lane;
    u8* tile;                   // [WT][WTP]
    unsigned lds_base;          // LDS byte address of tile[0]
    int tx0, ty0; bool have; unsigned nload;
    unsigned pl;                // cursor: global linear index
    int li;                     // cursor offset inside the window (valid while !reload)
    bool reload;                // the window has to be (re)placed before the next probe
    int lastk;                  // direction of the last step (8: none): a reloaded window leads that way
    int noff, noffa, dplv;      // this lane's neighbour offset: inside the window, the same + lds_base, in the image (lanes 0..7; lanes >= 8: the cursor cell)
    unsigned sel;               // WB_VIS << lane for lanes 0..7, 0 for the others: (sel >> k) & WB_VIS is the mark in lane k only
    int thr;                    // compare threshold of "foreground and not visited": -64; lane 8 (cursor cell): -48 = "carries no flag"
    int va, v;                  // window address this lane read in the last probe (relative to tile) and the byte it got (sign-extended)
    int rec;                    // lane j: record of the latest step t with (t & 63) == j
    unsigned home;              // start pixel of the leftover walk in progress (its window cell carries WB_HOME), ~0u: none
    unsigned codes_done, marks_done;    // steps of the walk whose code / visited mark has left the wave
    u8* st; int W, H;
    unsigned long long t_tile = 0;
    mutable unsigned long long n_exact = 0;
    u8* dtab;                   // [8192]: scratch of dup_distance's pre-check
    __device__ Wave() : lane((int)(threadIdx.x & 63)), tx0(0), ty0(0), have(false), nload(0), pl(0), li(0), reload(true), lastk(8), noff(0), noffa(0), dplv(0), sel(0), thr(-64),
                        va(0), v(0), rec(0), home(~0u), codes_done(0), marks_done(0), st(nullptr), W(0), H(0) {
        __shared__ u8 lds_tile[WT * WTP];
        __shared__ u8 lds_dup[8192];
        tile = lds_tile; lds_base = (unsigned)(uintptr_t)lds_tile; dtab = lds_dup;
    }
    __device__ void init(u8* st_, int W_, int H_) {
        st = st_; W = W_; H = H_;
        if (lane < 8) {
            const int dx = (int)((0x9224u >> (2 * lane)) & 3u) - 1, dy = (int)((0xA940u >> (2 * lane)) & 3u) - 1;
            noff = dy * WTP + dx; dplv = dy * W + dx; sel = (unsigned)WB_VIS << lane;
        }
        if (lane == 8) thr = -48;
        noffa = noff + (int)lds_base;
    }
    __device__ bool leader() const { return lane == 0; }
    __device__ unsigned l0() const { return (unsigned)lane; }
    __device__ unsigned nl() const { return 64u; }
    __device__ void fence() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_s_waitcnt(0); }
    // lane 0 loads, everybody gets the value
    __device__ unsigned ld0(const unsigned* p) const { unsigned x = 0; if (lane == 0) x = *p; return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
    // one 16-byte record (state, cont, end, begin) with a single load
    __device__ void ld0_rec(const unsigned* p, unsigned& cont, unsigned& en, unsigned& begin) const {
        uint4 x = make_uint4(0, 0, 0, 0); if (lane == 0) x = *reinterpret_cast<const uint4*>(p);
        cont = (unsigned)__builtin_amdgcn_readfirstlane((int)x.y); en = (unsigned)__builtin_amdgcn_readfirstlane((int)x.z); begin = (unsigned)__builtin_amdgcn_readfirstlane((int)x.w);
    }
    // The visited bits of the state plane are set with atomic ORs (no result, nothing to wait for) and every read of the plane by this
    // wave goes to L2 (agent-scope relaxed loads): an atomic executes in L2 and does not refresh a line the CU's L1 may still hold.
    __device__ u8 ld_state(unsigned lin) const { return __hip_atomic_load(st + lin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ void or_visited(unsigned lin) const {
        const uintptr_t a = (uintptr_t)(st + lin);
        atomicOr(reinterpret_cast<unsigned*>(a & ~(uintptr_t)3), (unsigned)ST_VIS << (8u * (unsigned)(a & 3)));
    }
    // step index of the record this lane holds (the latest step t < steps with (t & 63) == lane); >= steps: none
    __device__ unsigned my_step(unsigned steps) const { return steps - 1u - ((steps - 1u - (unsigned)lane) & 63u); }
    // visited marks of the steps [marks_done, steps) -> state plane
    __device__ void flush_marks(unsigned steps) {
        if (steps > marks_done) {
            const unsigned t = my_step(steps);
            if (t < steps && t >= marks_done && (rec & 2)) or_visited((unsigned)rec >> 5);
            marks_done = steps;
        }
    }
    // marks of the walk so far are in memory (before the state bytes are read from memory: next scan, next window)
    __device__ void sync_marks(unsigned steps) { flush_marks(steps); fence(); }
    __device__ void begin_walk(unsigned home_) { codes_done = 0; marks_done = 0; home = home_; }
    // direction codes of the steps [codes_done, steps) -> step log (called when a window of 64 steps is full, and at the end of a walk)
    __device__ void flush_codes(u8* slog, unsigned room, unsigned steps) {
        if (steps > codes_done) {
            const unsigned t = my_step(steps);
            if (t < steps && t >= codes_done && t < room) slog[t] = (u8)(((unsigned)rec >> 2) & 7u);
            codes_done = steps;
        }
        flush_marks(steps);
    }
    // state ((pl << 3) | k) of the step with index `first + lane` (one of the last 64 steps)
    __device__ unsigned pending(unsigned first) const { return (unsigned)__shfl(rec, (int)((first + (unsigned)lane) & 63u), 64) >> 2; }
    __device__ void load_tile(int cx, int cy, unsigned steps) {
        const unsigned long long t_0 = WPROF_NOW();
        flush_marks(steps);
        fence();                                                      // earlier marks have reached memory
        const int lead_x = lastk < 8 ? ((int)((0x9224u >> (2 * lastk)) & 3u) - 1) * ORIP_WALK_LEAD : 0;
        const int lead_y = lastk < 8 ? ((int)((0xA940u >> (2 * lastk)) & 3u) - 1) * ORIP_WALK_LEAD : 0;
        // 16-byte aligned columns; the window leads in the direction of the last step.  Rounding down moves the cursor up to 15 columns to the
        // right inside the window, so the nominal column is 24, not 32: the cursor lands in columns [8, 55], never on the ring (a cursor that
        // still stood on the ring after a reload would re-place the window for ever)
        tx0 = ((cx - (WT / 2 - 8) + lead_x) >> 4) << 4; ty0 = cy - WT / 2 + lead_y;
        const int y = ty0 + lane;
        u8* row = tile + lane * WTP;
        const uint32_t ring_row = (lane == 0 || lane == WT - 1) ? 0x20202020u : 0u;     // first and last row: every cell
        if (y < 0 || y >= H) { for (int j = 0; j < WT; j += 4) *reinterpret_cast<uint32_t*>(row + j) = 0x20202020u; }   // outside the image: not foreground; the flag is harmless
        else if ((W & 15) == 0 && tx0 >= 0 && tx0 + WT <= W && (tx0 & 15) == 0 && ((uintptr_t)st & 15) == 0) {
            // 64 bytes per row as four 16-byte loads that bypass the CU's L1 (the plane is updated with atomics, see or_visited)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u* src = reinterpret_cast<const v4u*>(st + (size_t)y * W + tx0);
            v4u q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = __builtin_nontemporal_load(src + j);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t w4[4] = {q[j].x | ring_row, q[j].y | ring_row, q[j].z | ring_row, q[j].w | ring_row};
                if (j == 0) w4[0] |= 0x20u;
                if (j == 3) w4[3] |= 0x20000000u;
#pragma unroll
                for (int t = 0; t < 4; t++) *reinterpret_cast<uint32_t*>(row + 16 * j + 4 * t) = w4[t];
            }
        } else if ((W & 3) == 0 && tx0 >= 0 && tx0 + WT <= W && ((uintptr_t)st & 3) == 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(st + (size_t)y * W + tx0);
            uint32_t w[WT / 4];
#pragma unroll
            for (int j = 0; j < WT / 4; j++) w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < WT / 4; j++) {
                uint32_t x = w[j] | ring_row;
                if (j == 0) x |= 0x20u;
                if (j == WT / 4 - 1) x |= 0x20000000u;
                *reinterpret_cast<uint32_t*>(row + 4 * j) = x;
            }
        } else {
            for (int j = 0; j < WT; j++) {
                int x = tx0 + j; u8 bte = (x >= 0 && x < W) ? ld_state((unsigned)((size_t)y * W + x)) : (u8)0;
                if (ring_row || j == 0 || j == WT - 1) bte |= WB_RING;
                row[j] = bte;
            }
        }
        have = true; nload++;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // (LDS is in order for one wave: the flag below lands after the rows)
        if (home != ~0u && steps > 0 && lane == 0) {                  // the start pixel of the walk inside the new window
            const int hx = (int)(home % (unsigned)W) - tx0, hy = (int)(home / (unsigned)W) - ty0;
            if ((unsigned)hx < (unsigned)WT && (unsigned)hy < (unsigned)WT) tile[hy * WTP + hx] |= WB_HOME;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        WPROF_ADD(t_tile, t_0);
    }
    __device__ void set_cursor(unsigned lin) { pl = lin; reload = true; lastk = 8; }
    // window offset of the cursor; reloads the window unless the cursor is strictly inside the current one
    __device__ void place(unsigned steps) {
        const int px = (int)(pl % (unsigned)W), py = (int)(pl / (unsigned)W);
        int lx = px - tx0, ly = py - ty0;
        if (!have || (unsigned)(lx - 1) > (unsigned)(WT - 3) || (unsigned)(ly - 1) > (unsigned)(WT - 3)) { load_tile(px, py, steps); lx = px - tx0; ly = py - ty0; }
        li = ly * WTP + lx; reload = false;
    }
    // marks the cursor pixel itself (start of a walk): state plane and, when inside, the window.  gv: its state byte (global coding)
    __device__ void mark_cursor(u8 gv) {
        if (lane == 0) {
            st[pl] = (u8)(gv | ST_VIS);
            if (have) {
                const int lx = (int)(pl % (unsigned)W) - tx0, ly = (int)(pl / (unsigned)W) - ty0;
                if ((unsigned)lx < (unsigned)WT && (unsigned)ly < (unsigned)WT) tile[ly * WTP + lx] |= WB_VIS;
            }
        }
        fence();
    }
    // the walk has left its start pixel: from now on stepping onto it ends the walk (the cell lies next to the cursor, inside the window)
    __device__ void flag_home() {
        if (lane == 0 && home != ~0u) {
            const int hx = (int)(home % (unsigned)W) - tx0, hy = (int)(home / (unsigned)W) - ty0;
            if (have && (unsigned)hx < (unsigned)WT && (unsigned)hy < (unsigned)WT) tile[hy * WTP + hx] |= WB_HOME;
        }
        fence();
    }
    // neighbours of the cursor; allow: NEIGH8 directions that may be taken (all but the way back).  Masks over NEIGH8 indices.
    __device__ void probe(unsigned allow, unsigned steps, unsigned& m_any, unsigned& m_unvis) {
        if (reload) place(steps);
        va = li + noff;
        v = (int)(signed char)tile[va];
        const unsigned m_fg = (unsigned)__builtin_amdgcn_ballot_w64(v < 0), m_un = (unsigned)__builtin_amdgcn_ballot_w64(v < -64);
        m_any = m_fg & allow & 0xffu;                                  // lanes >= 8 (reading the cursor cell) never count
        m_unvis = m_any & m_un;
    }
    // moves the cursor to neighbour k (after probe), records the step and returns the neighbour's state byte (global coding); fresh: it becomes visited
    __device__ u8 step(int k, bool fresh, unsigned steps) {
        const int vk = __builtin_amdgcn_readlane(v, k);
        li += __builtin_amdgcn_readlane(noff, k);
        pl = (unsigned)((int)pl + __builtin_amdgcn_readlane(dplv, k));
        lastk = k;
        const int r = (int)((((pl << 3) | (unsigned)k) << 2) | (fresh ? 2u : 0u));
        rec = lane == (int)(steps & 63u) ? r : rec;
        if (fresh) tile[va] = (u8)(v | (int)((sel >> k) & WB_VIS));   // every lane writes its byte back, lane k with the visited bit: no exec juggling
        reload = (vk & WB_RING) != 0;
        return (u8)((vk & (ST_FG | ST_VIS | ST_END | ST_JUN)) | (fresh ? ST_VIS : 0));
    }
    // after EV_DEAD: 2 = the cursor stands on the start pixel of the walk, 1 = it stood on the window ring (window re-placed), 0 = neither,
    // 3 = it stands on the end pixel of a listed chain (ST_CHAIN_END shares the window's "stop here" bit with the start-pixel flag): the
    // caller jumps or calls unflag_cursor()
    __device__ int resume_flagged(unsigned steps) {
        const u8 cell = tile[li];
        if (cell & WB_HOME) {
            if (pl == home) return 2;
            if (cell & ST_CHAIN) return 3;
            if (lane == 0) tile[li] = (u8)(cell & ~WB_HOME);          // the start pixel of an earlier walk: the flag is stale
            fence();
            if (!(cell & WB_RING)) return 1;
        }
        if (cell & WB_RING) { reload = true; place(steps); return 1; }
        return 0;
    }
    // The cursor stands on a flagged pixel of a listed chain.  If it got there along the chain (the cell it came from carries ST_CHAIN) and
    // no other neighbour does, it is leaving through an end pixel: nothing lies ahead, and the caller saves the three memory round trips
    // it would take to find that out from the chain list.  (A neighbour ahead with ST_CHAIN: an interior pixel that carries the stale
    // start flag of an earlier walk -- a jump from there is as good as one from an end.)  Not decidable on the window ring: false.
    __device__ bool came_along_chain() const {
        if (reload || lastk >= 8) return false;
        const u8 cell = tile[li];
        if (cell & WB_RING) return false;
        const u8 nbv = lane < 8 ? tile[li + noff] : (u8)0;
        const unsigned chain_nb = (unsigned)__ballot((nbv & ST_FG) && (nbv & ST_CHAIN)) & 0xffu;
        const unsigned back = 0x80u >> lastk;                          // NEIGH8 index 7 - lastk: where the cursor came from
        return (chain_nb & back) != 0u && (chain_nb & ~back) == 0u;
    }
    // the stepping loop may pass the flagged cell the cursor stands on (this window only: a reloaded window carries the flag again)
    __device__ void unflag_cursor(unsigned steps) {
        if (reload) place(steps);
        const u8 cell = tile[li];
        if (lane == 0) tile[li] = (u8)(cell & ~WB_HOME);
        fence();
        if (cell & WB_RING) { reload = true; place(steps); if (lane == 0) tile[li] = (u8)(tile[li] & ~WB_HOME); fence(); }
    }
    __device__ void log_code(u8*, unsigned, unsigned, int) {}      // the record of the step carries its code
    __device__ unsigned bcast(unsigned x, int src) const { return (unsigned)__shfl((int)x, src, 64); }
    __device__ int first(bool pred) const { unsigned long long m = __ballot(pred); return m ? __ffsll((long long)m) - 1 : -1; }
    // key: this lane's pending state + 1 (0: none).  Smallest d in [1, n) with key(lane - d) == key(lane), 0 if there is none: the keys
    // slide down the wave one lane per round (DPP wave_shr:1, zeros come in at lane 0), four rounds per loop trip; no LDS, no branches.
    // Two pending states of one look-up are almost never equal (a cycle shorter than the batch), and the exact search costs n rounds of a
    // lone wave (4.3 of 57 M cycles of the largest component of the bench image): every lane first drops its lane
    // number into a hashed LDS byte and reads it back -- a lane that finds its own number in every case has no equal (and no colliding)
    // partner, and only a look-up where some lane does not runs the exact rounds.  The table is never cleared: a lane reads what the
    // lanes of THIS look-up wrote last.
    __device__ unsigned dup_distance(unsigned key, unsigned n) const {
        if (n > 2u) {
            const unsigned slot = (key * 2654435761u) >> 19;
            if (key != 0u) dtab[slot] = (u8)lane;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const bool clash = key != 0u && dtab[slot] != (u8)lane;
            if (!__ballot(clash)) return 0u;
        }
        n_exact++;
        unsigned t = key, best = ~0u;
        for (unsigned d = 1; d < n; d += 4) {
#pragma unroll
            for (unsigned u = 0; u < 4; u++) {
                t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
                const unsigned cand = t == key ? d + u : ~0u;
                best = cand < best ? cand : best;
            }
        }
        return (key != 0 && best != ~0u) ? best : 0u;
    }
    // next q in [q0, e) whose pixel satisfies: (state & need) == need && !(state & ST_VIS).  The scans of a component move forward through its
    // pixel list in small steps, so the list entries are kept 64 at a time in a VGPR (lin_blk = entries [lin_q0, lin_q0 + 64)): a scan then costs
    // one memory round trip (the state bytes) instead of two dependent ones.
    // The hit's pixel and state byte come out of the lanes' registers (scan_lin, scan_st): the walk that starts there needs both, and
    // fetching them again would be two more dependent round trips per walk.  The block after the current one is requested when a block
    // is entered (its latency ends behind whatever the walks in between wait for).
    unsigned lin_blk = 0, lin_q0 = ~0u, lin_nxt = 0, lin_nq0 = ~0u, scan_lin = 0; u8 scan_st = 0;
    __device__ unsigned scan(const unsigned* lin, unsigned q0, unsigned e, u8 need) {
        for (unsigned q = q0; q < e; ) {
            if (q < lin_q0 || q >= lin_q0 + 64u) {
                if (q == lin_nq0) lin_blk = lin_nxt; else lin_blk = (q + (unsigned)lane < e) ? lin[q + lane] : 0u;
                lin_q0 = q;
                lin_nq0 = q + 64u; lin_nxt = (lin_nq0 + (unsigned)lane < e) ? lin[lin_nq0 + lane] : 0u;
            }
            const unsigned qq = lin_q0 + (unsigned)lane; bool ok = false; u8 x = 0;
            if (qq >= q && qq < e) { x = ld_state(lin_blk); ok = ((x & need) == need) && !(x & ST_VIS); }
            unsigned long long m = __ballot(ok);
            if (m) {
                const int hl = __ffsll((long long)m) - 1;
                scan_lin = (unsigned)__builtin_amdgcn_readlane((int)lin_blk, hl); scan_st = (u8)__builtin_amdgcn_readlane((int)x, hl);
                return lin_q0 + (unsigned)hl;
            }
            q = lin_q0 + 64u;
        }
        return e;
    }
    // inclusive prefix sums of (dx,dy) over the lanes; (ox,oy) = the lane's prefix, (tx,ty) = wave total
    __device__ void scan2(int dx, int dy, int& ox, int& oy, int& tx, int& ty) const {
        for (int o = 1; o < 64; o <<= 1) { int ax = __shfl_up(dx, o, 64), ay = __shfl_up(dy, o, 64); if (lane >= o) { dx += ax; dy += ay; } }
        ox = dx; oy = dy; tx = __shfl(dx, 63, 64); ty = __shfl(dy, 63, 64);
    }
    // ---- the stepping loop of a leftover walk (04:176-199) until something needs the caller: see EV_*.
    // In: cursor placed (li valid, not on a flagged cell unless the caller wants EV_DEAD at once), h.steps < h.limit <= next multiple of 64.
    // Per step: probe; m = unvisited neighbours else any neighbour but the way back; none -> EV_DEAD; fresh with pending states -> EV_PEND;
    // step to the first one in NEIGH8 order; record; mark it when fresh, else count it as pending; steps == limit -> EV_LIMIT;
    // nb == nbatch -> EV_BATCH.  A flagged cursor cell (ring, start pixel) makes the next probe come back empty -> EV_DEAD.
    __device__ int run(Hot& h, u8*, unsigned) {
        if (h.steps == 0) {
            // the first step of a walk goes through probe() / step(): its start cell only gets the WB_HOME flag once the cursor has left it
            unsigned f_any, f_un;
            probe(h.allow, 0u, f_any, f_un);
            const unsigned fm = f_un ? f_un : f_any;
            if (!fm) return EV_DEAD;
            const int fk = __builtin_ctz(fm);
            step(fk, f_un != 0, 0u);
            h.steps = 1; h.allow = 0xffu & ~(0x80u >> fk); h.nb = f_un ? 0u : 1u;
            flag_home();
            if (h.steps >= h.limit) return EV_LIMIT;
            if (h.nb >= h.nbatch) return EV_BATCH;
            // (a first step onto the window ring: the loop below comes back with EV_DEAD at once and the caller re-places the window)
        }
        int ev;
        int r_va, r_v, r_vb, r_w, r_t2;
        unsigned m_any, m_un, m, frmask, fr2, fr40, r, tmp, k, m0save;
        int s_li = __builtin_amdgcn_readfirstlane(li);
        unsigned s_pl = (unsigned)__builtin_amdgcn_readfirstlane((int)pl), s_allow = (unsigned)__builtin_amdgcn_readfirstlane((int)h.allow);
        unsigned s_steps = (unsigned)__builtin_amdgcn_readfirstlane((int)h.steps), s_nb = (unsigned)__builtin_amdgcn_readfirstlane((int)h.nb);
        const unsigned s_limit = (unsigned)__builtin_amdgcn_readfirstlane((int)h.limit), s_nbatch = (unsigned)__builtin_amdgcn_readfirstlane((int)h.nbatch);
#define ORIP_WALK_HALF(TAG, VA, V, VB, VW)                                                                            \
        "L_" TAG "%=:\n\t"                                                                                            \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
        "v_cmp_gt_i32 vcc, 0, %[" V "]\n\t"                         /* foreground */                                  \
        "s_and_b32 %[m_any], vcc_lo, %[allow]\n\t"                                                                    \
        "v_cmp_gt_i32 vcc, %[thr], %[" V "]\n\t"                    /* lanes 0..7: foreground, not visited; lane 8: cursor cell unflagged */ \
        "s_bitcmp1_b32 vcc_lo, 8\n\t"                                                                                 \
        "s_cselect_b32 %[m_any], %[m_any], 0\n\t"                                                                     \
        "s_and_b32 %[m_un], %[m_any], vcc_lo\n\t"                                                                     \
        "s_cselect_b32 %[m], %[m_un], %[m_any]\n\t"                                                                   \
        "s_cselect_b32 %[frmask], -1, 0\n\t"                                                                          \
        "s_cmp_eq_u32 %[m], 0\n\t"                                                                                    \
        "s_cbranch_scc1 L_dead%=\n\t"                                                                                 \
        "s_and_b32 %[tmp], %[frmask], %[nb]\n\t"                                                                      \
        "s_cbranch_scc1 L_pend%=\n\t"                                                                                 \
        "s_ff1_i32_b32 %[k], %[m]\n\t"                                                                                \
        "s_and_b32 %[fr40], %[frmask], 64\n\t"                                                                        \
        "v_readlane_b32 %[tmp], %[noff], %[k]\n\t"                                                                    \
        "v_lshrrev_b32 %[t2], %[k], %[sel]\n\t"                                                                       \
        "s_add_i32 %[li], %[li], %[tmp]\n\t"                                                                          \
        "v_and_or_b32 %[t2], %[t2], %[fr40], %[" V "]\n\t"                                                            \
        "v_add_u32 %[" VB "], %[li], %[noffa]\n\t"                                                                    \
        "ds_write_b8 %[" VA "], %[t2]\n\t"                                                                            \
        "ds_read_i8 %[" VW "], %[" VB "]\n\t"                                                                         \
        "v_readlane_b32 %[tmp], %[dplv], %[k]\n\t"                                                                    \
        "s_and_b32 %[fr2], %[frmask], 2\n\t"                                                                          \
        "s_add_i32 %[pl], %[pl], %[tmp]\n\t"                                                                          \
        "s_lshl3_add_u32 %[r], %[pl], %[k]\n\t"                                                                       \
        "s_lshl2_add_u32 %[r], %[r], %[fr2]\n\t"                                                                      \
        "v_writelane_b32 %[rec], %[r], m0\n\t"                                                                        \
        "s_add_i32 %[nb], %[nb], 1\n\t"                                                                               \
        "s_andn2_b32 %[nb], %[nb], %[frmask]\n\t"                                                                     \
        "s_lshr_b32 %[tmp], 0x80, %[k]\n\t"                                                                           \
        "s_andn2_b32 %[allow], 0xff, %[tmp]\n\t"                                                                      \
        "s_add_i32 %[steps], %[steps], 1\n\t"                                                                         \
        "s_add_i32 m0, m0, 1\n\t"                                                                                     \
        "s_cmp_ge_u32 %[steps], %[limit]\n\t"                                                                         \
        "s_cbranch_scc1 L_limit%=\n\t"                                                                                \
        "s_cmp_ge_u32 %[nb], %[nbatch]\n\t"                                                                           \
        "s_cbranch_scc1 L_batch%=\n\t"
        asm volatile(
            "s_mov_b32 %[m0save], m0\n\t"
            "s_and_b32 m0, %[steps], 63\n\t"
            "v_add_u32 %[va], %[li], %[noffa]\n\t"
            "ds_read_i8 %[v], %[va]\n\t"
            ORIP_WALK_HALF("A", "va", "v", "vb", "w")
            ORIP_WALK_HALF("B", "vb", "w", "va", "v")
            "s_branch L_A%=\n\t"
            "L_batch%=:\n\t"
            "s_mov_b32 %[ev], 5\n\t"
            "s_branch L_out%=\n\t"
            "L_dead%=:\n\t"
            "s_mov_b32 %[ev], 1\n\t"
            "s_branch L_out%=\n\t"
            "L_pend%=:\n\t"
            "s_mov_b32 %[ev], 2\n\t"
            "s_branch L_out%=\n\t"
            "L_limit%=:\n\t"
            "s_mov_b32 %[ev], 3\n\t"
            "L_out%=:\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b32 m0, %[m0save]\n\t"
            : [ev] "=&s"(ev), [li] "+s"(s_li), [pl] "+s"(s_pl), [allow] "+s"(s_allow), [steps] "+s"(s_steps), [nb] "+s"(s_nb),
              [rec] "+v"(rec), [va] "=&v"(r_va), [v] "=&v"(r_v), [vb] "=&v"(r_vb), [w] "=&v"(r_w), [t2] "=&v"(r_t2),
              [m_any] "=&s"(m_any), [m_un] "=&s"(m_un), [m] "=&s"(m), [frmask] "=&s"(frmask), [fr2] "=&s"(fr2), [fr40] "=&s"(fr40), [r] "=&s"(r), [tmp] "=&s"(tmp),
              [k] "=&s"(k), [m0save] "=&s"(m0save)
            : [limit] "s"(s_limit), [nbatch] "s"(s_nbatch), [noff] "v"(noff), [noffa] "v"(noffa), [dplv] "v"(dplv), [sel] "v"(sel), [thr] "v"(thr)
            : "vcc", "scc", "memory");
#undef ORIP_WALK_HALF
        li = s_li; pl = s_pl; h.allow = s_allow; h.steps = s_steps; h.nb = s_nb;
        const unsigned back = ~s_allow & 0xffu;                       // bit kopp = 7 - k of the last step
        lastk = back ? 7 - (int)__builtin_ctz(back) : 8;
        return ev;
    }
};
#else
struct Wave {
    u8 nv[8]; unsigned nload = 0; unsigned long long t_tile = 0, n_exact = 0;
    unsigned pl = 0, home = ~0u;
    unsigned pend_state = 0;
    u8* st = nullptr; int W = 0, H = 0;
    void init(u8* st_, int W_, int H_) { st = st_; W = W_; H = H_; }
    bool leader() const { return true; }
    unsigned l0() const { return 0u; }
    unsigned nl() const { return 1u; }
    void fence() const {}
    unsigned ld0(const unsigned* p) const { return *p; }
    void ld0_rec(const unsigned* p, unsigned& cont, unsigned& en, unsigned& begin) const { cont = p[1]; en = p[2]; begin = p[3]; }
    u8 ld_state(unsigned lin) const { return st[lin]; }
    void set_cursor(unsigned lin) { pl = lin; }
    void mark_cursor(u8 v) { st[pl] = (u8)(v | ST_VIS); }
    void flag_home() {}
    void sync_marks(unsigned) {}
    void begin_walk(unsigned home_) { home = home_; }
    void flush_codes(u8*, unsigned, unsigned) {}
    unsigned pending(unsigned) const { return pend_state; }
    int resume_flagged(unsigned) { return 0; }
    void log_code(u8* slog, unsigned room, unsigned idx, int k) { if (idx < room) slog[idx] = (u8)k; }
    void probe(unsigned allow, unsigned, unsigned& m_any, unsigned& m_unvis) {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        const int px = (int)(pl % (unsigned)W), py = (int)(pl / (unsigned)W);
        m_any = m_unvis = 0;
        for (int k = 0; k < 8; k++) {
            int xx = px + dxs[k], yy = py + dys[k]; nv[k] = 0;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            u8 v = st[(size_t)yy * W + xx]; nv[k] = v;
            if ((v & ST_FG) && ((allow >> k) & 1u)) { m_any |= 1u << k; if (!(v & ST_VIS)) m_unvis |= 1u << k; }
        }
    }
    u8 step(int k, bool mark, unsigned) {
        const int dxs[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dys[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
        pl = (unsigned)((int)pl + dys[k] * W + dxs[k]);
        if (mark) st[pl] = (u8)(nv[k] | ST_VIS);
        return (u8)(nv[k] | (mark ? ST_VIS : 0));
    }
    // the same loop as the GPU's Wave::run, one lane: codes go straight to the step log, one pending state at most (nbatch == 1)
    int run(Hot& h, u8* slog, unsigned room) {
        while (true) {
            unsigned m_any, m_un;
            probe(h.allow, h.steps, m_any, m_un);
            const bool fresh = m_un != 0;
            const unsigned m = fresh ? m_un : m_any;
            if (!m) return EV_DEAD;
            if (fresh && h.nb) return EV_PEND;
            const int k = __builtin_ctz(m);
            log_code(slog, room, h.steps, k);
            step(k, fresh, h.steps);
            h.steps++;
            h.allow = 0xffu & ~(0x80u >> k);
            if (!fresh) { pend_state = (pl << 3) | (unsigned)k; h.nb++; }
            if (pl == home) return EV_HOME;
            if (h.steps >= h.limit) return EV_LIMIT;
            if (h.nb >= h.nbatch) return EV_BATCH;
        }
    }
    unsigned bcast(unsigned v, int) const { return v; }
    int first(bool pred) const { return pred ? 0 : -1; }
    unsigned dup_distance(unsigned, unsigned) const { return 0u; }       // one pending state at most
    unsigned scan_lin = 0; u8 scan_st = 0;
    unsigned scan(const unsigned* lin, unsigned q0, unsigned e, u8 need) {
        for (unsigned q = q0; q < e; q++) { u8 v = st[lin[q]]; if (((v & need) == need) && !(v & ST_VIS)) { scan_lin = lin[q]; scan_st = v; return q; } }
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

// reader of log records for code that runs one thread per walk (not one wave)
struct PlainReader {
    ORIP_HD void ld0_rec(const unsigned* p, unsigned& cont, unsigned& en, unsigned& begin) const { cont = p[1]; en = p[2]; begin = p[3]; }
};

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

// 04:201-204: a path whose end lies within 1.5 px of its start gets the start appended again (integers: squared distance 0, 1 or 2)
ORIP_HD inline bool close_to(int W, unsigned a, unsigned b) {
    const int ddx = (int)(a % (unsigned)W) - (int)(b % (unsigned)W), ddy = (int)(a / (unsigned)W) - (int)(b / (unsigned)W);
    return ddx * ddx + ddy * ddy < 3;
}

// A walk that ended by jumping into a recorded trajectory (flags bit 1) has its end point R steps down that trajectory: following the
// records costs a few dependent loads, which the serial trace does not wait for -- one thread per such walk does it afterwards
// (k_winfo_lens on the GPU) and settles the closing point: returns the final WalkInfo length and flags.
template <class WaveT>
ORIP_HD inline void walk_close_tail(const WalkArgs& A, const WaveT& wv, unsigned slot, WalkInfo& wi) {
    if (!(wi.flags & 2u)) return;
    unsigned lo = 0, hi = A.nc;
    while (lo < hi) { unsigned mid = (lo + hi) >> 1; if (2u * A.comp_start[mid + 1] <= slot) lo = mid + 1; else hi = mid; }
    const unsigned b = A.comp_start[lo], fg = A.comp_start[lo + 1] - b, rel = slot - 2u * b;
    const unsigned s = A.lin[b + (rel >= fg ? rel - fg : rel)];
    const unsigned i = wi.log_i1 - 1;
    const unsigned long long f = log_resolve(A.logbuf, wv, i, (unsigned long long)i + wi.R);
    const unsigned endp = A.logbuf[4ull * f] >> 3;
    wi.flags = 0;
    if (close_to(A.W, s, endp)) { wi.flags = 1; wi.len_kept++; }
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
    bool over = false, stalled = false;
    unsigned long long d_w1 = 0, d_s1 = 0, d_w2 = 0, d_s2 = 0, d_hit = 0, d_det = 0, d_jump = 0;
    unsigned long long t_scan = 0, t_flush = 0, n_flush = 0, n_ev = 0, t_f3 = 0, t_chain = 0, n_chain = 0, n_round = 0, t_run = 0, t_walk = 0; const unsigned long long t_begin = WPROF_NOW();   // ORIP_WALK_PROF builds only
    auto finish = [&](unsigned slot, unsigned long long len, unsigned n_own, unsigned sbeg, unsigned log_i1, unsigned R, unsigned flags) {
        if (wv.leader()) { WalkInfo wi; wi.len_kept = len >= 5 ? (unsigned)len : 0u; wi.n_own = n_own; wi.step_begin = sbeg; wi.log_i1 = log_i1; wi.R = R; wi.flags = flags; A.winfo[slot] = wi; }
    };
    auto tscan = [&](unsigned q0, u8 need) { const unsigned long long t_0 = WPROF_NOW(); const unsigned r = wv.scan(A.lin, q0, e, need); WPROF_ADD(t_scan, t_0); return r; };
    // ---- phase 1: walks from endpoints (04:144-171); >= 2 points to be a path (04:168), >= 5 to survive vectorize_layer (04:224)
    const unsigned long long g1 = (unsigned long long)(total_fg * 2);
    for (unsigned q = tscan(b, ST_FG | ST_END); q < e; q = tscan(q + 1, ST_FG | ST_END)) {
        unsigned s = wv.scan_lin;
        wv.set_cursor(s);
        const unsigned sbeg = step_base + stepcur;
        u8* slog = A.steplog + (size_t)sbeg; const unsigned room = step_cap - stepcur;
        wv.begin_walk(~0u);
        wv.mark_cursor(ST_FG | ST_END);
        unsigned steps = 0, allow = 0xffu; d_w1++;
        while (true) {
            unsigned m_any, m_unvis;
            wv.probe(allow, steps, m_any, m_unvis);
            if (!m_unvis) break;
            const int k = ffs8(m_unvis);
            if (steps >= room) over = true;
            wv.log_code(slog, room, steps, k);
            const u8 v = wv.step(k, true, steps);
            steps++;
            if ((steps & 63u) == 0) wv.flush_codes(slog, room, steps);
            allow = 0xffu & ~(0x80u >> k);                        // not the way back: NEIGH8 index 7 - k
            if (v & (ST_JUN | ST_END)) break;
            if ((unsigned long long)steps > g1) break;       // guard (04:163): counts the steps that went on
        }
        stepcur += steps; d_s1 += steps;
        wv.flush_codes(slog, room, steps);
        wv.sync_marks(steps);  // marks of this walk are complete before the next scan
        finish(2u * b + (q - b), 1ull + steps, steps, sbeg, 0u, 0u, 0u);
    }
    // ---- phase 2: leftovers / cycles (04:174-205)
    // A walk that finds no fresh neighbour is on a trajectory that only depends on its state (pixel, incoming direction), so its
    // states are looked up in / added to the memo.  The memo lives in HBM and a look-up per step would put one full memory latency
    // on every step of a strictly serial chain; instead the walk runs ahead on the LDS window (Wave::run) and keeps its no-fresh
    // states pending (they are the records of its last nb steps).  flush() then does what the reference order requires for all of
    // them at once: the first pending state that is already known (to an older record, to this run, or to an earlier pending state)
    // ends the walk there and the steps taken after it are dropped; otherwise all of them become provisional entries of this run.
    const unsigned g2 = fg * 4u;                                  // guard of a leftover walk (04:199)
    const unsigned nbatch0 = wv.nl() < ORIP_WALK_BATCH ? wv.nl() : ORIP_WALK_BATCH;
    for (unsigned q = tscan(b, ST_FG); q < e && !over && !stalled; q = tscan(q + 1, ST_FG)) {
        const unsigned long long t_w0 = WPROF_NOW();
        unsigned s = wv.scan_lin;
        wv.set_cursor(s);
        const unsigned sbeg = step_base + stepcur;
        u8* slog = A.steplog + (size_t)sbeg; const unsigned room = step_cap - stepcur;
        wv.begin_walk(s);
        wv.mark_cursor(wv.scan_st);
        d_w2++;
        Hot h; h.steps = 0;                   // steps taken = own points - 1 = value of the reference's guard counter at its check
        h.limit = 0; h.nb = 0;                // pending no-fresh states: those of the last nb steps
        h.nbatch = nbatch0;                   // memo hits come early in a run or not for a while: 4, 8, 16, ... pending states per look-up
        h.allow = 0xffu;
        unsigned long long tail_len = 0; unsigned tail_i1 = 0, tail_R = 0;
        unsigned nofresh = 0;                 // no-fresh states logged since the last fresh pixel: entries [run_begin, run_begin + nofresh)
        unsigned flush_mark = 0;              // h.steps when the last look-up returned: a later pending state that starts beyond it follows a fresh step
        unsigned drop = 0;                    // 1: the walk ended on the step of the last pending state (start pixel / guard), which the reference does not look at
        // Provisional entries of a run that did not end in a known trajectory are taken out of the memo again, so that a non-zero memo
        // word always names a valid entry: below run_begin a committed one, from run_begin on one of the run in progress.  A look-up is
        // then ONE load (the word itself), not the word plus the log entry it points to -- half the memory latency of the serial chain.
        auto discard_run = [&]() {
            if (nofresh) {
                const unsigned run_begin = log_base + logcur;
                for (unsigned t = wv.l0(); t < nofresh; t += wv.nl()) memo[A.logbuf[4ull * (run_begin + t)]] = 0u;
                wv.fence();
                nofresh = 0;
            }
        };
        // 0: nothing known, pending states committed; 1: the walk ended in a jump; 2: log overflow.
        // ext: the pending states are not the records of the last h.nb steps but come from the caller, lane j holding the j-th (chain_jump)
        auto flush_ = [&](bool ext, unsigned ext_state) -> int {
            const unsigned nb = h.nb - drop;
            if (!nb) { h.nb = 0; return 0; }
            const unsigned steps_b = h.steps - h.nb;                  // steps before the first pending state
            if (steps_b > flush_mark) { discard_run(); h.nbatch = nbatch0; }      // a fresh pixel since the last look-up: a new run
            const unsigned myS = ext ? ext_state : wv.pending(steps_b);           // lane j: j-th pending state
            const unsigned run_begin = log_base + logcur;
            const unsigned me = wv.l0();
            const bool act = me < nb;
            wv.fence();                                               // entries written by the previous look-up (long done by now)
            unsigned mi = 0;
            if (act) mi = memo[myS];
            const unsigned long long t_c = WPROF_NOW();
            const unsigned dist = wv.dup_distance(act ? myS + 1u : 0u, nb);                  // distance to the latest earlier pending state equal to mine (0: none)
            const bool dup = act && dist != 0; const unsigned dsrc = me - dist;
            t_f3 += WPROF_NOW() - t_c;
            const bool hit_old = act && mi != 0;
            const int js = wv.first(hit_old || dup);
            const unsigned ncommit = js < 0 ? nb : (unsigned)js;
            if (logcur + nofresh + ncommit > log_cap) { over = true; h.nb = 0; return 2; }
            if (me < ncommit) { const unsigned idx = run_begin + nofresh + me; A.logbuf[4ull * idx] = myS; memo[myS] = idx + 1; }
            if (js < 0) { nofresh += nb; h.nb = 0; flush_mark = h.steps; return 0; }
            // the reference would have stopped at pending state js: roll the step counter back to it
            const unsigned ev_mi = wv.bcast(dup ? run_begin + nofresh + dsrc + 1u : mi, js);
            nofresh += (unsigned)js;
            h.steps = steps_b + (unsigned)js + 1;
            const unsigned i = ev_mi - 1;
            const unsigned long long R = (unsigned long long)g2 + 1ull - h.steps;           // points still to come until the guard fires
            if (i < run_begin) d_hit++; else d_det++;
            // committed trajectory of an earlier walk: this run's own no-fresh states become a transient record that runs into entry i, so
            // later walks can jump from them too.  A state of this very run: the cycle [i, run_begin + nofresh) is closed.
            if (nofresh) {
                const unsigned end = run_begin + nofresh;
                for (unsigned t = me; t < nofresh; t += wv.nl()) { unsigned* p = A.logbuf + 4ull * (run_begin + t); p[1] = i; p[2] = end; p[3] = run_begin; }
                logcur += nofresh;
                nofresh = 0;                                  // committed: nothing to discard
            }
            tail_i1 = ev_mi; tail_R = (unsigned)R; tail_len = R;
            h.nb = 0;
            return 1;
        };
        auto flush = [&]() -> int { const unsigned long long t_0 = WPROF_NOW(); n_flush++; const int r = flush_(false, 0u); WPROF_ADD(t_flush, t_0); return r; };
        int ended = 0;
        unsigned stall = 0, stall_steps = ~0u;
#if defined(__HIP_DEVICE_COMPILE__)
        // The cursor stands on the end pixel of a listed chain (it has just stepped onto it).  The pixels ahead along the chain are forced
        // steps: taken 64 per round by the lanes -- as long as they are all fresh or all visited (the run is cut where that changes, and
        // before the walk's start pixel), in the reference's order of events: pending no-fresh states are looked up before a fresh pixel is
        // taken (EV_PEND), no-fresh steps are looked up state by state (the first known one ends the walk there), fresh pixels are marked.
        // Leaves the cursor on the last pixel taken; the window is re-placed from the state plane.  Returns `ended` (0: walk goes on).
        unsigned last_jump_end = ~0u;
        auto chain_jump = [&]() -> int {
            if (!A.cref || wv.pl == last_jump_end || wv.lastk >= 8 || wv.came_along_chain()) { wv.unflag_cursor(h.steps); return 0; }
            const unsigned lane = wv.l0();
            wv.flush_codes(slog, room, h.steps);                          // the records of the steps so far leave the wave: codes, and the visited
            wv.fence();                                                   // marks the state bytes read below must already show
            int res = 0; bool moved = false;
            unsigned idx = wv.ld0(A.cref + (size_t)A.plane * layer + wv.pl);
            // the chain on both sides of the cursor in one round trip: lane j reads the pixel j + 1 places up and j + 1 places down
            unsigned q_up = A.cpix[idx + lane + 1u], q_dn = A.cpix[idx - lane - 1u];
            // direction along the chain: away from where the walk came from (its neighbour in the chain, or a pixel outside the chain)
            const unsigned prev_px = (unsigned)((int)wv.pl - (nby(wv.lastk) * W + nbx(wv.lastk)));
            const unsigned c_m = wv.bcast(q_dn, 0), c_p = wv.bcast(q_up, 0);
            int dir;
            if (c_p == prev_px) dir = -1; else if (c_m == prev_px) dir = 1;                 // inside the chain: keep going
            else dir = c_m == ORIP_CHAIN_SENTINEL ? 1 : -1;                                  // entered from outside at an end pixel: inwards
            bool first_round = true;
            while (true) {
                const long long room_left = (long long)(room < g2 ? room : g2) - 80ll - (long long)h.steps;       // stay clear of the guard and of the end of the step log
                if (room_left < 8) break;
                const unsigned q = first_round ? (dir > 0 ? q_up : q_dn) : A.cpix[(unsigned)((int)idx + dir * (int)(lane + 1u))];      // the next 64 chain pixels (lanes past the sentinel read the neighbouring chain: masked)
                first_round = false;
                const int sent = wv.first(q == ORIP_CHAIN_SENTINEL);
                const bool valid = sent < 0 || (int)lane < sent;
                u8 sb = 0; if (valid) sb = wv.ld_state(q);
                const bool fr = !(sb & ST_VIS);
                const bool fr0 = wv.bcast(fr ? 1u : 0u, 0) != 0u;
                const bool stop = !valid || fr != fr0 || q == wv.home;
                int r = wv.first(stop); if (r < 0) r = 64;
                bool go_on = r == 64 || wv.bcast((valid && q != wv.home) ? 1u : 0u, r < 64 ? r : 0) != 0u;      // cut by a change of freshness only: another round
                if ((long long)r > room_left) { r = (int)room_left; go_on = false; }
                if (r < (moved ? 1 : 8)) break;                                   // not worth leaving the stepping loop for
                // direction code of step j: from the pixel before (lane j - 1's, or the cursor) to q
                unsigned qp = (unsigned)__shfl_up((int)q, 1, 64); if (lane == 0) qp = wv.pl;
                const int dd = (int)q - (int)qp; const int dy = dd > 1 ? 1 : (dd < -1 ? -1 : 0); const int dx = dd - dy * W;
                const int kk = (dy + 1) * 3 + (dx + 1); const int k = kk > 4 ? kk - 1 : kk;
                const unsigned myst = (q << 3) | (unsigned)k;
                const unsigned nbp = h.nb;
                if (!fr0 && nbp + (unsigned)r <= wv.nl()) {
                    // the states still pending and the stretch's own are consecutive no-fresh states of the walk: one look-up for all of them
                    const unsigned pend = nbp ? wv.pending(h.steps - nbp) : 0u;
                    const unsigned ext = (unsigned)__shfl((int)myst, (int)((lane - nbp) & 63u), 64);
                    if ((int)lane < r && h.steps + lane < room) slog[h.steps + lane] = (u8)k;
                    moved = true;
                    h.steps += (unsigned)r; h.nb = nbp + (unsigned)r;
                    n_flush++;
                    res = flush_(true, lane < nbp ? pend : ext);
                    if (res) break;
                } else {
                    if (nbp) { res = flush(); if (res) break; }                   // states still pending are looked up before anything new happens (EV_PEND / EV_BATCH)
                    if ((int)lane < r && h.steps + lane < room) slog[h.steps + lane] = (u8)k;
                    moved = true;
                    if (fr0) {
                        if ((int)lane < r) wv.or_visited(q);
                        h.steps += (unsigned)r;
                    } else {
                        h.steps += (unsigned)r; h.nb = (unsigned)r;
                        n_flush++;
                        res = flush_(true, myst);
                        if (res) break;
                    }
                }
                d_jump += (unsigned)r; n_round++;
                wv.pl = wv.bcast(q, r - 1); wv.lastk = (int)wv.bcast((unsigned)k, r - 1); h.allow = 0xffu & ~(0x80u >> wv.lastk);
                idx = (unsigned)((int)idx + dir * r);
                if (!go_on) break;
            }
            if (moved) {
                wv.codes_done = h.steps; wv.marks_done = h.steps; last_jump_end = wv.pl;
                if (!res) { wv.have = false; wv.reload = true; wv.place(h.steps); }      // a fresh window around the new cursor (the old one lacks the marks of the stretch)
            } else if (!res) wv.unflag_cursor(h.steps);
            return res;
        };
#endif
        WPROF_ADD(t_walk, t_w0);
        while (true) {
            if (h.steps >= room) { over = true; break; }
            unsigned lim = (h.steps & ~63u) + 64u;                    // the records of 64 steps fit the lanes: codes and marks leave at every multiple of 64
            if (lim > g2 + 1u) lim = g2 + 1u;
            if (lim > room) lim = room;
            h.limit = lim;
            const unsigned long long t_r0 = WPROF_NOW();
            int ev = wv.run(h, slog, room);
            WPROF_ADD(t_run, t_r0);
            n_ev++;
            if (ev == EV_DEAD) {
                int f = wv.resume_flagged(h.steps);                   // GPU: a flagged window cell stops the loop the same way
#if defined(__HIP_DEVICE_COMPILE__)
                if (f == 3) { const unsigned long long t_0 = WPROF_NOW(); n_chain++; ended = chain_jump(); WPROF_ADD(t_chain, t_0); if (ended) break; f = 1; }       // end pixel of a listed chain: the forced stretch ahead in one go (or not: then the flag is out of the way)
#endif
                if (f == 1) {                                         // window re-placed (or a stale flag cleared): go on
                    if (h.steps == stall_steps && ++stall > 8u) { stalled = true; break; }     // ... unless nothing moves: a bug, never a hang
                    if (h.steps != stall_steps) { stall_steps = h.steps; stall = 0; }
                    continue;
                }
                if (f == 0) break;                                    // no neighbour at all (04:187-188)
                ev = EV_HOME;
            }
            if (ev == EV_HOME) { drop = h.nb ? 1u : 0u; break; }      // back on the start pixel (04:196): checked before the state of that step counts
            if (ev == EV_PEND) { ended = flush(); if (ended) break; continue; }
            if (ev == EV_LIMIT) {
                if ((h.steps & 63u) == 0) wv.flush_codes(slog, room, h.steps);
                if (h.steps > g2) { drop = h.nb ? 1u : 0u; break; }   // guard (04:199), also checked before the state of that step counts
                if (h.nb < h.nbatch) continue;                        // (the step that filled the code window may have filled the batch too)
            }
            // EV_BATCH
            ended = flush(); if (ended) break;
            h.nbatch = h.nbatch * 2u < wv.nl() ? h.nbatch * 2u : wv.nl();
        }
        const unsigned long long t_w1 = WPROF_NOW();
        if (!ended && h.nb > drop) ended = flush();
        wv.flush_codes(slog, room, h.steps);
        if (ended != 1) discard_run();                                // a run that never reached a known trajectory stays out of the memo
        wv.sync_marks(h.steps);
        if (ended == 2) break;
        const unsigned steps = h.steps;
        stepcur += steps; d_s2 += steps;
        unsigned long long len = 1ull + steps + tail_len;
        unsigned flags = 0;
        if (ended == 1) flags = 2;                                    // the end point lies in the recorded trajectory: walk_close_tail decides about the closing point
        else if (len >= 2 && close_to(W, s, wv.pl)) { flags = 1; len++; }
        finish(2u * b + fg + (q - b), len >= 2 ? len : 0, steps, sbeg, tail_i1, tail_R, flags);
        WPROF_ADD(t_walk, t_w1);
    }
    if (A.log_used && wv.leader()) A.log_used[c] = logcur;
    if (over && wv.leader()) *A.overflow = 1;
    if (stalled && wv.leader()) *A.overflow = 2;                   // internal error: the host reports it instead of retrying
    if (A.dbg && wv.leader()) {
        unsigned long long* d = A.dbg + 32ull * c; d[0] = d_w1; d[1] = d_s1; d[2] = d_w2; d[3] = d_s2; d[4] = d_hit; d[5] = d_det; d[6] = wv.nload; d[7] = (unsigned long long)fg;
        d[8] = WPROF_NOW() - t_begin; d[9] = wv.t_tile; d[10] = t_scan; d[11] = t_flush; d[12] = n_flush; d[13] = n_ev; d[14] = t_f3; d[15] = d_jump; d[16] = t_chain; d[17] = n_chain; d[18] = n_round; d[19] = t_run; d[20] = wv.n_exact; d[21] = t_walk;     // cycle counts: ORIP_WALK_PROF builds only
    }
}

// Points [p0, p1) of a layer's contour list (positions in the concatenated point array of its kept walks), written by one wavefront.
// The output is cut into equal chunks instead of giving one wave to each walk: a few walks carry millions of points (bounce tails up to
// the guard), and one wave writing such a walk alone was the whole duration of the pass.  kept[i] = walk slot of path i, in output order,
// kept_off[i] = index of its first point.
ORIP_HD inline void write_chunk(const WalkArgs& A, int layer, const unsigned* kept, const unsigned long long* kept_off, unsigned n_kept, unsigned long long p0, unsigned long long p1) {
    using namespace walk_detail;
    Wave wv;
    const int W = A.W;
    int2* outp = reinterpret_cast<int2*>(A.pts[layer]);
    // path holding point p0: the last i with start(i) <= p0
    unsigned lo = 0, hi = n_kept;
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (kept_off[mid] <= p0) lo = mid; else hi = mid; }
    for (unsigned i = lo; i < n_kept; i++) {
        const unsigned slot = kept[i];
        const unsigned long long ws = kept_off[i];
        if (ws >= p1) break;
        const WalkInfo wi = A.winfo[slot];
        const unsigned long long L = wi.len_kept;
        const unsigned long long ta = (p0 > ws ? p0 : ws) - ws, tb = (p1 < ws + L ? p1 : ws + L) - ws;      // this walk's points [ta, tb)
        if (ta >= tb) continue;
        int2* out = outp + ws;
        // start pixel of the walk: slots [2b, 2e) belong to the component whose list range is [b, e)
        unsigned clo = 0, chi = A.nc;
        while (clo < chi) { unsigned mid = (clo + chi) >> 1; if (2u * A.comp_start[mid + 1] <= slot) clo = mid + 1; else chi = mid; }
        const unsigned b = A.comp_start[clo], fg = A.comp_start[clo + 1] - b, rel = slot - 2u * b;
        const unsigned s = A.lin[b + (rel >= fg ? rel - fg : rel)];
        const int x0 = (int)(s % (unsigned)W), y0 = (int)(s / (unsigned)W);
        if (wv.leader()) {
            if (ta == 0) { out[0] = make_int2(x0, y0); A.off[layer][i + 1] = (int64_t)(ws + L); }
            if ((wi.flags & 1u) && tb == L) out[L - 1] = make_int2(x0, y0);
        }
        // ---- own steps: point t (1 <= t <= n_own) = start + sum of the direction vectors of codes [0, t)
        const unsigned long long own_a = ta > 1 ? ta : 1, own_b = tb < (unsigned long long)wi.n_own + 1 ? tb : (unsigned long long)wi.n_own + 1;
        if (own_a < own_b) {
            int cx = x0, cy = y0;
            for (unsigned long long c0 = 0; c0 < own_a - 1; c0 += wv.nl()) {            // codes before the chunk: only their sum
                const unsigned long long t = c0 + wv.l0();
                int dx = 0, dy = 0;
                if (t < own_a - 1) { const int k = A.steplog[(size_t)wi.step_begin + t]; dx = nbx(k); dy = nby(k); }
                int ox, oy, tx, ty; wv.scan2(dx, dy, ox, oy, tx, ty);
                cx += tx; cy += ty;
            }
            for (unsigned long long t0 = own_a; t0 < own_b; t0 += wv.nl()) {
                const unsigned long long t = t0 + wv.l0();
                int dx = 0, dy = 0;
                if (t < own_b) { const int k = A.steplog[(size_t)wi.step_begin + t - 1]; dx = nbx(k); dy = nby(k); }
                int ox, oy, tx, ty; wv.scan2(dx, dy, ox, oy, tx, ty);
                if (t < own_b) out[t] = make_int2(cx + ox, cy + oy);
                cx += tx; cy += ty;
            }
        }
        // ---- tail: tail point u (0 <= u < R) is entry i + 1 + u of the recorded trajectory, which is a chain of record pieces ending in a cycle
        if (wi.log_i1) {
            const unsigned long long tail0 = (unsigned long long)wi.n_own + 1;
            unsigned long long ua = (ta > tail0 ? ta : tail0) - tail0, ub = (tb < tail0 + wi.R ? tb : tail0 + wi.R);
            if (ub > tail0 && ua < ub - tail0) {
                ub -= tail0;
                int2* o2 = out + tail0;
                unsigned long long done = 0, f0 = (unsigned long long)(wi.log_i1 - 1) + 1; unsigned rec = wi.log_i1 - 1;
                const bool pow2 = (W & (W - 1)) == 0; const int wsh = pow2 ? __builtin_ctz((unsigned)W) : 0;      // x = l % W, y = l / W without a division where W allows
                auto put = [&](unsigned long long u, unsigned long long entry) {
                    const unsigned l = A.logbuf[4ull * entry] >> 3;
                    o2[u] = pow2 ? make_int2((int)(l & (unsigned)(W - 1)), (int)(l >> wsh)) : make_int2((int)(l % (unsigned)W), (int)(l / (unsigned)W));
                };
                while (done < ub) {
                    const unsigned cont = A.logbuf[4ull * rec + 1], en = A.logbuf[4ull * rec + 2], begin = A.logbuf[4ull * rec + 3];
                    if (f0 < en) {                               // the rest of this record: tail points [done, done + take)
                        const unsigned long long take = en - f0;
                        const unsigned long long a = ua > done ? ua : done, bnd = ub < done + take ? ub : done + take;
                        for (unsigned long long u = a + wv.l0(); u < bnd; u += wv.nl()) put(u, f0 + (u - done));
                        done += take;
                        if (done >= ub) break;
                    }
                    if (cont >= begin) {                         // cycle [cont, en): everything that is left
                        const unsigned lam = en - cont;
                        const unsigned long long a = ua > done ? ua : done;
                        unsigned ph = (unsigned)((a - done) % lam);                         // phase of point a inside the cycle (once per chunk)
                        unsigned mine = (ph + wv.l0()) % lam; const unsigned stride = wv.nl() % lam;
                        for (unsigned long long u = a + wv.l0(); u < ub; u += wv.nl()) { put(u, (unsigned long long)cont + mine); mine += stride; if (mine >= lam) mine -= lam; }
                        break;
                    }
                    f0 = cont; rec = cont;                       // transient record exhausted: continue in the older record
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Walk-coded contours.  A kept walk is: its start pixel, n_own steps (explicit points, a few per skeleton pixel) and, for a
// leftover walk that ran into a recorded trajectory, R tail points that are a chain of log-record pieces ending in a cycle that is
// repeated until the guard (04:199) fires -- up to 4 * fg points for a handful of distinct ones.  Stage 04 leaves these records;
// the point lists of 04 / 05 / 07 and the front of 08 are VIEWS over them (vsrc.h), and write_chunk above is only the reference
// expansion (tests/host).  tail point u (0 <= u < R) of a walk = pixel of log entry E(u); E is piecewise:
//   piece j covers [u0_j, u0_{j+1}):  E(u) = ent_j + (u - u0_j)           (lam_j == 0, the rest of a record)
//                                     E(u) = ent_j + (u - u0_j) % lam_j   (lam_j  > 0, the closing cycle: always the last piece)
// ------------------------------------------------------------------------------------------------
struct VWalk { unsigned own_off, n_own, piece_off, n_piece, len, flags, pad0, pad1; };     // flags bit0: the last point is the start again (04:203-204)
struct VPiece { unsigned u0, ent, lam, magic; };                                          // ent: index into the layer's log buffer; magic = floor(2^32 / lam) for d % lam (vsrc.h)
struct VView { unsigned wid, first, len, rev; };                                          // polyline = points first .. first+len-1 of walk wid, reversed if rev

// pieces of the tail of a kept walk; emit(j, u0, entry, lam) with GLOBAL entry indices (as the trace addressed the log).  Returns their number.
template <class WaveT, class Emit>
ORIP_HD inline unsigned vwalk_pieces(const unsigned* logbuf, const WaveT& wv, const WalkInfo& wi, Emit&& emit) {
    if (!wi.log_i1) return 0u;
    unsigned n = 0; unsigned long long done = 0, f0 = wi.log_i1; unsigned rec = wi.log_i1 - 1;
    while (done < wi.R) {
        unsigned cont, en, begin; wv.ld0_rec(&logbuf[4ull * rec], cont, en, begin);
        if (f0 < en) { emit(n++, (unsigned)done, (unsigned)f0, 0u); done += en - f0; if (done >= wi.R) break; }      // the rest of this record
        if (cont >= begin) { emit(n++, (unsigned)done, cont, en - cont); break; }                                    // cycle [cont, en): everything that is left
        f0 = cont; rec = cont;                                                                                       // transient record exhausted: on into the older record
    }
    return n;
}

// Own points [q0, q1) of a layer (positions in the concatenation of the kept walks' start + own-step points: walk i holds n_own + 1 of
// them from vw[i].own_off), written by one wavefront: prefix sums over the direction codes, as write_chunk does for a walk's own steps.
ORIP_HD inline void own_chunk(const WalkArgs& A, const unsigned* kept, const VWalk* vw, unsigned n_kept, int2* own, unsigned q0, unsigned q1) {
    using namespace walk_detail;
    Wave wv;
    const int W = A.W;
    unsigned lo = 0, hi = n_kept;
    while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (vw[mid].own_off <= q0) lo = mid; else hi = mid; }
    for (unsigned i = lo; i < n_kept; i++) {
        const VWalk w = vw[i];
        if (w.own_off >= q1) break;
        const unsigned L = w.n_own + 1u;
        const unsigned ta = (q0 > w.own_off ? q0 : w.own_off) - w.own_off, tb = (q1 < w.own_off + L ? q1 : w.own_off + L) - w.own_off;     // this walk's own points [ta, tb)
        if (ta >= tb) continue;
        const unsigned slot = kept[i];
        const WalkInfo wi = A.winfo[slot];
        unsigned clo = 0, chi = A.nc;
        while (clo < chi) { unsigned mid = (clo + chi) >> 1; if (2u * A.comp_start[mid + 1] <= slot) clo = mid + 1; else chi = mid; }
        const unsigned b = A.comp_start[clo], fg = A.comp_start[clo + 1] - b, rel = slot - 2u * b;
        const unsigned s = A.lin[b + (rel >= fg ? rel - fg : rel)];
        const int x0 = (int)(s % (unsigned)W), y0 = (int)(s / (unsigned)W);
        int2* out = own + w.own_off;
        if (wv.leader() && ta == 0) out[0] = make_int2(x0, y0);
        const unsigned own_a = ta > 1u ? ta : 1u, own_b = tb;
        if (own_a < own_b) {
            int cx = x0, cy = y0;
            for (unsigned c0 = 0; c0 < own_a - 1u; c0 += wv.nl()) {              // codes before the chunk: only their sum
                const unsigned t = c0 + wv.l0();
                int dx = 0, dy = 0;
                if (t < own_a - 1u) { const int k = A.steplog[(size_t)wi.step_begin + t]; dx = nbx(k); dy = nby(k); }
                int ox, oy, tx, ty; wv.scan2(dx, dy, ox, oy, tx, ty);
                cx += tx; cy += ty;
            }
            for (unsigned t0 = own_a; t0 < own_b; t0 += wv.nl()) {
                const unsigned t = t0 + wv.l0();
                int dx = 0, dy = 0;
                if (t < own_b) { const int k = A.steplog[(size_t)wi.step_begin + t - 1]; dx = nbx(k); dy = nby(k); }
                int ox, oy, tx, ty; wv.scan2(dx, dy, ox, oy, tx, ty);
                if (t < own_b) out[t] = make_int2(cx + ox, cy + oy);
                cx += tx; cy += ty;
            }
        }
    }
}
