// csrc/orip_ctx.h -- device context of liborip.so (gfx950 only).
#pragma once
// Kernel variants that newer ones have REPLACED and that nothing falls back to (the one-workgroup k-means fit, byte-plane thinning, the first grid greedy
// kernel, the sequential tail simulation, the serial float chain of the cumulative lengths) are compiled into a variants build only (`make variants` ->
// liborip_variants.so, -DORIP_VARIANTS): there their switches (ORIP_KMEANS_1WG, ORIP_THIN_BYTES, ORIP_NN_OLDGRID / ORIP_NN_DBG, ORIP_TAIL_OLDSIM,
// ORIP_CUM_CHAIN) select them for the agreement tests; in the default library the switches read as not set.  (The other switches force paths the default
// library needs anyway: byte morphology / CCL / NMS for inputs the bit-plane kernels do not take, the LDS-less greedy / plot-order kernels for large lists.)
#ifdef ORIP_VARIANTS
#define ORIP_VARIANT(name) (getenv(name) != nullptr)
#else
#define ORIP_VARIANT(name) false
#endif

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <atomic>
#include "../../include/orip.h"

typedef uint8_t u8;

#define ORIP_FAIL(ctx, ...)                                              \
    do {                                                                 \
        char _b[512];                                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                           \
        { std::lock_guard<std::mutex> _g((ctx)->mu); (ctx)->err = std::string(__func__) + ": " + _b; } \
        return -1;                                                       \
    } while (0)

#define HIPC(ctx, call)                                                                        \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess) ORIP_FAIL(ctx, "%s -> %s", #call, hipGetErrorString(_e));        \
    } while (0)

#define ORIP_TRY(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)

extern int orip_alloc_dbg;      // ORIP_ALLOC_DBG: log every (re)allocation -- in steady state there must be none (hipFree waits for the whole device)
// Growable device buffer; contents are NOT preserved across growth unless keep=true.
struct DBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, hipStream_t s = 0, bool keep = false) {
        if (bytes <= cap) return hipSuccess;
        if (orip_alloc_dbg) fprintf(stderr, "[alloc] %p: %zu -> %zu bytes%s\n", (void*)this, cap, bytes, p ? " (hipFree: device-wide wait)" : "");
        size_t ncap = bytes + bytes / 4 + 256;
        void* np_ = nullptr;
        hipError_t e = hipMalloc(&np_, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(np_, p, cap, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { hipFree(np_); return e; }
        }
        if (p) hipFree(p);
        p = np_; cap = ncap;
        return hipSuccess;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// Device polyline list: off int64[n+1], pts int32[2*total].
// A list can also be WALK-CODED (virt): its points are not stored but generated from the walk records stage 04 leaves for the layer
// (WalkStore below: own points, bounce tails as (log range, cycle) pieces) through a view per polyline (walk, first point, length,
// reversed) and an optional stage-05 scale -- see vsrc.h.  The walker re-walks the same pixels up to 4 * fg times, so
// the explicit form of a heavy layer is 2.8e8 points (2.25 GB) for ~1e6 distinct ones; stages 05 / 07 / 08 read the coded form and the
// explicit form only exists where somebody asks for it (orip_get_polys, a consumer that is not view-aware: orip_polys_materialize).
struct DPolys {
    DBuf off, pts;
    int64_t n = 0, total = 0;
    bool virt = false;          // walk-coded: `off` is valid, `pts` only when pts_ok
    bool pts_ok = true;
    DBuf vview;                 // VView[n]; unused when vident (polyline i = walk i, whole, forward)
    bool vident = true;
    int vlayer = 0;             // whose WalkStore
    uint64_t vepoch = 0;        // WalkStore::epoch the list was built on: a later trace of the layer makes it stale
    uint64_t vsepoch = 0;       // scaled lists: WalkStore::sepoch of the scaled tables they read
    uint64_t pf_tag = 0;        // != 0: the list is a permutation-with-flips of the list LaneRes::pf08 (same tag) was computed on
    bool scaled = false;        // reads the scaled tables of the WalkStore
    void set_explicit() { virt = false; pts_ok = true; pf_tag = 0; }
};
// What stage 04 leaves per layer (raster04.hip: trace_finish) and every walk-coded list of the layer reads
struct WalkStore {
    DBuf log;                   // the trace's state log (walker.h: 4 words per entry)
    DBuf walk, piece, own;      // VWalk[n], VPiece[...], int2 own points (start + own steps of every kept walk)
    DBuf lxy;                   // int2 pixel of every log entry in use (same index as the log)
    DBuf ent_idx, cnt;          // the indices of those entries, in any order; cnt[0] = how many (device side: nobody on the host needs it)
    DBuf own_s, lxy_s;          // the same two tables after _scale_one (05:82-96) with (sx, sy, dx, dy): what a scaled list reads
    float sx = 1.f, sy = 1.f, dx = 0.f, dy = 0.f;
    int64_t n = 0, n_own = 0, ent_cap = 0;
    int W = 0;
    uint64_t epoch = 0;         // bumped by every trace of the layer
    uint64_t sepoch = 0;        // bumped whenever own_s / lxy_s are rewritten
};
struct DTaps {
    DBuf xy;  // int32[2*n]
    int64_t n = 0;
};

struct ProfEntry { double ms = 0; int64_t launches = 0; };

// Per-lane resources.  Lane 0 serves the raster stages and the cross-layer stage 10; lane l+1 serves the per-layer vector
// stages (05, 07, 08, 12) of layer l, so that different layers can be driven concurrently from different host threads,
// each on its own HIP stream with its own scratch (the serial kernels of one layer then overlap with those of the others).
struct LaneRes {
    hipStream_t stream = 0;
    hipStream_t stream2 = 0;              // side stream of the lane (work that may overlap the main chain), fenced with ev2 / ev3
    hipEvent_t ev2 = nullptr, ev3 = nullptr, ev4 = nullptr;   // (ev4 / ev3: features / everything of stage 08's prefetch)
    DBuf vtmp[12], tmpE, tmpF, flags, canvas;
    DBuf pixbits;                         // stage 08-A: one bit per canvas pixel that is the rounded position of a sample
    unsigned caps_hint = 0;               // distinct capsules of the lane's last stage-08-A run (sizes the next run's table)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DPolys tp[6];   // persistent temporaries of the vector stages (no hipFree in steady state: hipFree synchronises the device)
    // stage 08's order-independent front, computed on the side stream while stage 07's greedy chain runs (vector08.hip: prefetch08)
    struct Prefetch08 {
        bool pending = false;          // side-stream work the lane's main stream has not waited for yet (orip_pf08_drain)
        bool valid = false; uint64_t tag = 0; int64_t n = 0; int64_t tot_f = 0; double step = 0;
        const int64_t* src_off = nullptr;   // offsets of the list it was computed on (device): both readings of polyline i keep their cumulative lengths at src_off[i]
        DBuf seg;                      // float32 length of every segment of the list (k_seglen -> both readings, perimeter sums)
        DBuf feat, info, cum, ord;     // PolyFeat[n] + reversed perimeters float[n]; RsInfo[2n]: forward at i, reversed at n + i; cum: forward readings, then (tot_f on) reversed
    } pf08;
};
extern thread_local int orip_tls_lane;
#define LN(c) ((c)->ln[orip_tls_lane])
// HIP's current device is per host thread: every entry point selects the context's GPU for the calling thread (the layer pipelines
// call in from pool threads, which would otherwise allocate and launch on device 0 of a multi-GPU node)
struct orip_ctx;
void orip_enter(orip_ctx* c);
#define ORIP_LANE_CROSS (ORIP_MAX_LAYERS + 1)
void orip_contours_free(orip_ctx* c);
// A lane (stream + scratch that grows with hipFree / hipMalloc) serves ONE call at a time: two host threads on one lane would free
// buffers under each other's kernels (the r01 memory access fault of the sharded path: a stage-12 call addressed by the GLOBAL layer id
// landed on the lane of another layer's running 04->08 pipeline).  LaneGuard claims the lane for the calling thread and the entry
// points fail loudly when it is taken; nested claims of the lane the thread already holds are free.
struct LaneGuard {
    orip_ctx* c; int prev, lane; bool ok, owner;
    LaneGuard(orip_ctx* ctx, int lane_id);
    ~LaneGuard();
};
#define ORIP_LANE_NODRAIN(ctx, lane_id)                                                                                      \
    LaneGuard _lane_guard((ctx), (lane_id));                                                                                 \
    if (!_lane_guard.ok) ORIP_FAIL(ctx, "lane %d is busy: another call is using this layer's stream and scratch", (int)(lane_id))
// Stage 08's prefetch (vector08.hip: prefetch08) may still be running on the lane's side stream when stage 07 returns: stage 08 waits for its parts where it
// consumes them; every other call that claims the lane puts its main stream behind the whole of it first (it reads the scaled list and the lane's scratch).
#define ORIP_LANE(ctx, lane_id)                                                                                              \
    ORIP_LANE_NODRAIN(ctx, lane_id);                                                                                         \
    HIPC(ctx, orip_pf08_drain(ctx))

struct orip_ctx {
    int device = 0;
    LaneRes ln[ORIP_MAX_LAYERS + 2];      // 0: raster stages; l + 1: layer l; ORIP_LANE_CROSS: stage 10
    std::atomic<int> lane_owner[ORIP_MAX_LAYERS + 2];   // 1 while a call holds the lane (LaneGuard); lane 0 is not claimed
    orip_params10 p10{}; bool p10_ready = false;   // stage 10 between orip_dedup_cross_begin and the per-layer calls
    size_t hw_cross_off = 0, hw_cross_pts = 0;      // largest kept-line list of stage 10 so far: the buffers it swaps with the layers' LINES_CROSS slots never have to grow again
    bool cross_unordered[ORIP_MAX_LAYERS] = {false};   // LINES_CROSS of the layer still waits for its travel reorder (orip_dedup_cross_layer_deferred)
    void* prep04 = nullptr;               // stage-04 state between orip_contours_prepare and orip_contours_layer (raster04.hip)
    std::mutex mu;
    std::string err;
    // image / raster state
    int H = 0, W = 0, K = 0;
    DBuf image;     // u8 [H,W,3] BGR
    DBuf labels;    // u8 [H,W]
    DBuf masks;     // u8 [K,H,W]
    DBuf edges;     // u8 [K,H,W]
    DBuf skel;      // u8 [K,H,W]
    DBuf tmpA, tmpB, tmpC, tmpD;   // raster scratch
    DBuf cref, cpix;               // forced stretches of the skeletons (walker.h: ST_CHAIN): position plane and chain pixel lists
    DBuf lab_tabs;  // u16 gamma[256] + u16 cbrt[3072] + i32 coeffs[9]
    bool tabs_ready = false;
    const void* edge_bits = nullptr;   // bit planes of `edges` left in lane 0's scratch by stage 03 (nullptr: not available); consumed by stage 04
    const void* morphed_bits = nullptr; // bit planes of the opened / closed masks left in tmpA for stage 03's NMS kernel (nullptr: byte planes in tmpB)
    const void* mask_bits = nullptr;   // bit planes of `masks` left in tmpA by stage 02 (nullptr: not available); consumed by stage 03
    // vector state
    DPolys polys[ORIP_SLOT_COUNT][ORIP_MAX_LAYERS];
    WalkStore wstore[ORIP_MAX_LAYERS];
    DTaps taps[2][ORIP_MAX_LAYERS];
    DBuf ops[ORIP_MAX_LAYERS];
    int64_t n_ops[ORIP_MAX_LAYERS] = {0};
    // multi-GPU exchange (comm.hip): RCCL communicator of this process, device row for the list sizes
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    DBuf comm_sizes;
    // 13_build_stream: moves and their direction codes (stream.hip), resident between orip_stream_codes and the fetch
    DBuf stream_segs, stream_off, stream_codes; int64_t stream_n = 0, stream_total = 0;
    DBuf resize_src, resize_dst;                       // raster01.hip staging
    int memo_pre_K = 0, memo_pre_H = 0, memo_pre_W = 0; // orip_contours_reserve cleared this many memo planes of an H x W image
    // profiling
    bool prof_on = false;
    std::map<std::string, ProfEntry> prof;
};

inline LaneGuard::LaneGuard(orip_ctx* ctx, int lane_id) : c(ctx), prev(orip_tls_lane), lane(lane_id), ok(true), owner(false) {
    if (prev != lane) { int expect = 0; ok = c->lane_owner[lane].compare_exchange_strong(expect, 1); owner = ok; }
    if (ok) orip_tls_lane = lane;
}
inline hipError_t orip_pf08_drain(orip_ctx* c) {           // the claimed lane's main stream goes on behind its pending prefetch
    LaneRes& l = LN(c);
    if (!l.pf08.pending) return hipSuccess;
    l.pf08.pending = false;
    return hipStreamWaitEvent(l.stream, l.ev3, 0);
}
inline LaneGuard::~LaneGuard() { if (ok) orip_tls_lane = prev; if (owner) c->lane_owner[lane].store(0); }

#if defined(__HIPCC__)
// one bit per pixel -> 0 / 255 bytes, 16 pixels (ONE 16-byte store) per thread: rows that are multiples of 64 wide (nw words per plane = H * W / 64),
// blockIdx.z = plane.  Four bits become four bytes by a multiply that drops bit i at position 8 i (n * (1 + 2^7 + 2^14 + 2^21), no carries) and a mask.
static __global__ __launch_bounds__(256) void k_bits_expand16(const unsigned long long* __restrict__ bits, uint8_t* __restrict__ dst, size_t nw) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nw * 4) return;
    const unsigned long long w = bits[nw * blockIdx.z + (t >> 2)];
    const unsigned n16 = (unsigned)(w >> ((t & 3) * 16)) & 0xffffu;
    auto four = [](unsigned n4) { return (((n4 & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu; };
    uint4 o; o.x = four(n16); o.y = four(n16 >> 4); o.z = four(n16 >> 8); o.w = four(n16 >> 12);
    reinterpret_cast<uint4*>(dst + nw * 64 * blockIdx.z)[t] = o;
}
#endif
// Time one kernel launch with HIP events on ctx->stream when profiling is enabled (bench.py roofline leg).
struct ProfScope {
    orip_ctx* c; const char* name;
    bool armed = false;
    ProfScope(orip_ctx* ctx, const char* n) : c(ctx), name(n) { if (c->prof_on) armed = hipEventRecord(LN(c).ev0, LN(c).stream) == hipSuccess; }
    ~ProfScope() {
        if (!armed) return;
        float ms = 0;
        if (hipEventRecord(LN(c).ev1, LN(c).stream) != hipSuccess || hipEventSynchronize(LN(c).ev1) != hipSuccess ||
            hipEventElapsedTime(&ms, LN(c).ev0, LN(c).ev1) != hipSuccess) return;          // a failed timing is dropped, never recorded as 0 ms
        std::lock_guard<std::mutex> g(c->mu);
        auto& e = c->prof[name]; e.ms += ms; e.launches++;
    }
};

// raise a kernel's dynamic-LDS limit; the result is kept so that every caller (not only the one thread that ran the std::call_once) sees a failure
template <class F> static inline void orip_max_lds(F* kernel, int bytes, std::atomic<int>& err) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) err.store((int)e);
}
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// stage entry points implemented across the .hip files
int orip_raster02_lab_tables(orip_ctx* c);
// the per-layer stages without the closing stream wait (orip_layer_front chains them on the layer's stream)
int orip_contours_layer_impl(orip_ctx* c, int layer, bool sync);
int orip_scale_vectors_impl(orip_ctx* c, int layer, float sx, float sy, float dx, float dy, bool sync);
int orip_sort_contours_impl(orip_ctx* c, int layer, bool sync, const orip_params08* prm_for_prefetch = nullptr);
int orip_prefetch08(orip_ctx* c, void* prm, DPolys& scaled, const void* feat07);
// explicit points of a walk-coded list (no-op for explicit lists); on the calling lane's stream, not synchronised (vector.hip)
int orip_polys_materialize(orip_ctx* c, DPolys& P);
