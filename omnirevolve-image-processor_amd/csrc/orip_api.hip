// csrc/orip_api.hip -- context life-cycle, slot transfers and profiling hooks of liborip.so.
#include "orip_ctx.h"
#include <cstdlib>

thread_local int orip_tls_lane = 0;
int orip_alloc_dbg = 0;

void orip_enter(orip_ctx* c) {
    if (c) hipSetDevice(c->device);      // cheap (thread-local in the runtime); not cached here: the host application may switch devices too
}

extern "C" int orip_create(int device_id, orip_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    // one hardware queue per lane where the runtime still accepts it (HIP multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues by
    // default, and kernels that share a queue run one after the other): only effective when this is the process's first HIP call
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    orip_alloc_dbg = getenv("ORIP_ALLOC_DBG") != nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -2;   // no GPU: fail loudly, there is no CPU path
    if (device_id < 0 || device_id >= ndev) return -3;
    if (hipSetDevice(device_id) != hipSuccess) return -4;
    orip_ctx* c = new orip_ctx();
    c->device = device_id;
    for (auto& o : c->lane_owner) o.store(0);
    for (auto& l : c->ln) {
        if (hipStreamCreate(&l.stream) != hipSuccess) { delete c; return -5; }
        hipEventCreate(&l.ev0); hipEventCreate(&l.ev1);
        if (hipStreamCreate(&l.stream2) != hipSuccess) { delete c; return -5; }
        hipEventCreateWithFlags(&l.ev2, hipEventDisableTiming); hipEventCreateWithFlags(&l.ev3, hipEventDisableTiming); hipEventCreateWithFlags(&l.ev4, hipEventDisableTiming);
        if (l.flags.ensure(4096) != hipSuccess) { delete c; return -6; }
        hipMemsetAsync(l.flags.p, 0, 4096, l.stream);
        hipStreamSynchronize(l.stream);
    }
    *out = c;
    return 0;
}

extern "C" void orip_destroy(orip_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    DBuf* bufs[] = {&c->image, &c->labels, &c->masks, &c->edges, &c->skel, &c->tmpA, &c->tmpB, &c->tmpC, &c->tmpD, &c->lab_tabs, &c->cref, &c->cpix};
    for (DBuf* b : bufs) b->release();
    for (auto& l : c->ln) {
        l.tmpE.release(); l.tmpF.release(); l.flags.release(); l.canvas.release(); l.pixbits.release();
        for (auto& v : l.vtmp) v.release();
        for (DBuf* b : {&l.pf08.seg, &l.pf08.feat, &l.pf08.info, &l.pf08.cum, &l.pf08.ord}) b->release();
        if (l.ev0) hipEventDestroy(l.ev0);
        if (l.ev1) hipEventDestroy(l.ev1);
        if (l.ev2) hipEventDestroy(l.ev2);
        if (l.ev3) hipEventDestroy(l.ev3);
        if (l.ev4) hipEventDestroy(l.ev4);
        if (l.stream2) hipStreamDestroy(l.stream2);
        if (l.stream) hipStreamDestroy(l.stream);
    }
    for (int s = 0; s < ORIP_SLOT_COUNT; s++) for (int l = 0; l < ORIP_MAX_LAYERS; l++) { c->polys[s][l].off.release(); c->polys[s][l].pts.release(); c->polys[s][l].vview.release(); }
    for (auto& l : c->ln) for (auto& t : l.tp) { t.off.release(); t.pts.release(); t.vview.release(); }
    for (auto& w : c->wstore) for (DBuf* b : {&w.log, &w.walk, &w.piece, &w.own, &w.lxy, &w.ent_idx, &w.cnt, &w.own_s, &w.lxy_s}) b->release();
    for (int s = 0; s < 2; s++) for (int l = 0; l < ORIP_MAX_LAYERS; l++) c->taps[s][l].xy.release();
    for (int l = 0; l < ORIP_MAX_LAYERS; l++) c->ops[l].release();
    orip_contours_free(c);
    orip_comm_destroy(c);
    c->comm_sizes.release();
    c->stream_segs.release(); c->stream_off.release(); c->stream_codes.release();
    c->resize_src.release(); c->resize_dst.release();
    delete c;
}

extern "C" const char* orip_last_error(orip_ctx* c) {
    orip_enter(c); return c ? c->err.c_str() : "null context"; }

extern "C" int orip_sync(orip_ctx* c) {
    orip_enter(c);
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}

extern "C" int orip_prof_enable(orip_ctx* c, int on) {
    orip_enter(c); c->prof_on = on != 0; return 0; }
extern "C" int orip_prof_reset(orip_ctx* c) {
    orip_enter(c); c->prof.clear(); return 0; }
extern "C" int orip_prof_get(orip_ctx* c, const char* kernel, double* total_ms, int64_t* launches) {
    orip_enter(c);
    auto it = c->prof.find(kernel);
    if (it == c->prof.end()) { *total_ms = 0; *launches = 0; return 0; }
    *total_ms = it->second.ms; *launches = it->second.launches;
    return 0;
}

extern "C" int orip_set_layer_count(orip_ctx* c, int K) {
    orip_enter(c);
    if (K < 1 || K > ORIP_MAX_LAYERS) ORIP_FAIL(c, "K=%d out of range", K);
    c->K = K;
    return 0;
}

static int check_slot(orip_ctx* c, int slot, int layer) {
    if (slot < 0 || slot >= ORIP_SLOT_COUNT || layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad slot %d / layer %d", slot, layer);
    return 0;
}

extern "C" int orip_polys_size(orip_ctx* c, int slot, int layer, int64_t* n, int64_t* total) {
    orip_enter(c);
    ORIP_TRY(check_slot(c, slot, layer));
    *n = c->polys[slot][layer].n; *total = c->polys[slot][layer].total;
    return 0;
}
extern "C" int orip_get_polys(orip_ctx* c, int slot, int layer, int64_t* off, int32_t* pts) {
    orip_enter(c);
    ORIP_TRY(check_slot(c, slot, layer));
    DPolys& P = c->polys[slot][layer];
    if (P.n == 0) { off[0] = 0; return 0; }
    if (pts) ORIP_TRY(orip_polys_materialize(c, P));      // a walk-coded list (stages 04 / 05 / 07 of a resident chain) is expanded here, on request
    HIPC(c, hipMemcpyAsync(off, P.off.p, (size_t)(P.n + 1) * 8, hipMemcpyDeviceToHost, LN(c).stream));
    if (pts && P.total) HIPC(c, hipMemcpyAsync(pts, P.pts.p, (size_t)P.total * 8, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_set_polys(orip_ctx* c, int slot, int layer, int64_t n, const int64_t* off, const int32_t* pts) {
    orip_enter(c);
    ORIP_TRY(check_slot(c, slot, layer));
    if (n < 0) ORIP_FAIL(c, "negative count");
    DPolys& P = c->polys[slot][layer];
    int64_t total = n ? off[n] : 0;
    { LaneRes::Prefetch08& F = c->ln[layer + 1].pf08;       // a prefetch of stage 08 still reading this layer's lists on its side stream: let it finish first
      if (F.pending) { HIPC(c, hipEventSynchronize(c->ln[layer + 1].ev3)); } }
    HIPC(c, P.off.ensure((size_t)(n + 1) * 8 + 64));
    HIPC(c, P.pts.ensure((size_t)std::max<int64_t>(total, 1) * 8 + 64));
    if (n) HIPC(c, hipMemcpyAsync(P.off.p, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, LN(c).stream));
    else HIPC(c, hipMemsetAsync(P.off.p, 0, 8, LN(c).stream));
    if (total) HIPC(c, hipMemcpyAsync(P.pts.p, pts, (size_t)total * 8, hipMemcpyHostToDevice, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    P.set_explicit();
    P.n = n; P.total = total;      // the raster layer count (c->K) is NOT touched: list slots are addressed up to ORIP_MAX_LAYERS
    return 0;
}
extern "C" int orip_taps_size(orip_ctx* c, int which, int layer, int64_t* n) {
    orip_enter(c);
    if (which < 0 || which > 1 || layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad taps slot");
    *n = c->taps[which][layer].n;
    return 0;
}
extern "C" int orip_has_variants(void) {
#ifdef ORIP_VARIANTS
    return 1;
#else
    return 0;
#endif
}
extern "C" int orip_get_taps(orip_ctx* c, int which, int layer, int32_t* xy) {
    orip_enter(c);
    if (which < 0 || which > 1 || layer < 0 || layer >= ORIP_MAX_LAYERS) ORIP_FAIL(c, "bad taps slot");
    DTaps& T = c->taps[which][layer];
    if (!T.n) return 0;
    HIPC(c, hipMemcpyAsync(xy, T.xy.p, (size_t)T.n * 8, hipMemcpyDeviceToHost, LN(c).stream));
    HIPC(c, hipStreamSynchronize(LN(c).stream));
    return 0;
}
extern "C" int orip_set_taps(orip_ctx* c, int which, int layer, int64_t n, const int32_t* xy) {
    orip_enter(c);
    if (which < 0 || which > 1 || layer < 0 || layer >= ORIP_MAX_LAYERS || n < 0) ORIP_FAIL(c, "bad taps slot");
    DTaps& T = c->taps[which][layer];
    HIPC(c, T.xy.ensure((size_t)std::max<int64_t>(n, 1) * 8 + 64));
    if (n) { HIPC(c, hipMemcpyAsync(T.xy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice, LN(c).stream)); HIPC(c, hipStreamSynchronize(LN(c).stream)); }
    T.n = n;
    return 0;
}
