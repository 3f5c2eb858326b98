// csrc/comm.hip -- the one exchange step of the path across GPUs: stage 10 walks the layers dark -> light against ONE cumulative
// raster (10:215,236-267), so every rank needs every layer's stage-08 output (lines_intra, taps_intra).  The lists are already resident in
// the owner's slots; they go to the other ranks with RCCL broadcasts over xGMI, device buffer to device buffer, on the stage-10 lane's
// stream -- no host staging, no torch.  Payloads are tiny (tens of KB per layer), so the step is latency-bound: one size row, then one
// grouped broadcast of the three arrays.  The reference has no counterpart (it is a single process).
#include "orip_ctx.h"
#include <rccl/rccl.h>

#define NCCLC(ctx, call)                                                                       \
    do {                                                                                       \
        ncclResult_t _r = (call);                                                              \
        if (_r != ncclSuccess) ORIP_FAIL(ctx, "%s -> %s", #call, ncclGetErrorString(_r));      \
    } while (0)

static_assert(sizeof(ncclUniqueId) == ORIP_COMM_ID_BYTES, "ORIP_COMM_ID_BYTES must match ncclUniqueId");

extern "C" int orip_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return -1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -2;
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int orip_comm_init(orip_ctx* c, const uint8_t* id_bytes, int rank, int world) {
    orip_enter(c);
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) ORIP_FAIL(c, "bad communicator arguments (rank %d of %d)", rank, world);
    if (c->comm) ORIP_FAIL(c, "communicator already initialised");
    ncclUniqueId id; memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCLC(c, ncclCommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    HIPC(c, c->comm_sizes.ensure(64));
    return 0;
}

extern "C" int orip_comm_destroy(orip_ctx* c) {
    orip_enter(c);
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    return 0;
}

// LINES_INTRA / TAPS_INTRA of slot `my_slot` from rank `root` to every rank (each rank names its own slot: the owner keeps the layer
// under a local index, the others receive into a spare slot).  Collective: every rank calls it for the same layers in the same order.
extern "C" int orip_bcast_layer(orip_ctx* c, int root, int my_slot) {
    orip_enter(c);
    if (!c->comm) ORIP_FAIL(c, "orip_comm_init has not run");
    if (my_slot < 0 || my_slot >= ORIP_MAX_LAYERS || root < 0 || root >= c->comm_world) ORIP_FAIL(c, "bad slot %d / root %d", my_slot, root);
    ORIP_LANE(c, ORIP_LANE_CROSS);
    ncclComm_t comm = (ncclComm_t)c->comm;
    hipStream_t s = LN(c).stream;
    DPolys& P = c->polys[ORIP_SLOT_LINES_INTRA][my_slot]; DTaps& T = c->taps[ORIP_TAPS_INTRA][my_slot];
    // A rank that fails after the size row has gone round must not simply return: the others would wait in the payload broadcasts
    // for ever.  Everything that can fail locally (expanding a walk-coded list, the size row's sanity, the allocations) is done
    // BEFORE this rank joins the grouped broadcast; if any of it fails the communicator is aborted, which ends the peers' pending
    // collectives with an error instead of a hang.
    auto bail = [&](const char* what) -> int {
        ncclCommAbort(comm); c->comm = nullptr;
        ORIP_FAIL(c, "%s: communicator aborted so that the other ranks fail instead of hanging", what);
    };
    int64_t sizes[3] = {0, 0, 0};
    if (c->comm_rank == root) {
        if (orip_polys_materialize(c, P) != 0) return bail("expanding the layer's list failed on the root");
        sizes[0] = P.n; sizes[1] = P.total; sizes[2] = T.n;
        if (hipMemcpyAsync(c->comm_sizes.p, sizes, sizeof(sizes), hipMemcpyHostToDevice, s) != hipSuccess) return bail("staging the size row failed");
    }
    NCCLC(c, ncclBroadcast(c->comm_sizes.p, c->comm_sizes.p, 3, ncclInt64, root, comm, s));
    if (c->comm_rank != root) {
        if (hipMemcpyAsync(sizes, c->comm_sizes.p, sizeof(sizes), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail("reading the size row failed");
        if (sizes[0] < 0 || sizes[1] < 0 || sizes[2] < 0 || sizes[0] > (int64_t)1 << 40 || sizes[1] > (int64_t)1 << 40) return bail("corrupt size row");
        if (P.off.ensure((size_t)(sizes[0] + 1) * 8 + 64) != hipSuccess || P.pts.ensure((size_t)std::max<int64_t>(sizes[1], 1) * 8 + 64) != hipSuccess ||
            T.xy.ensure((size_t)std::max<int64_t>(sizes[2], 1) * 8 + 64) != hipSuccess) return bail("no memory for the incoming layer");
        P.set_explicit();
        if (sizes[0] == 0 && hipMemsetAsync(P.off.p, 0, 8, s) != hipSuccess) return bail("clearing the offsets failed");
    }
    NCCLC(c, ncclGroupStart());
    if (sizes[0]) NCCLC(c, ncclBroadcast(P.off.p, P.off.p, (size_t)(sizes[0] + 1), ncclInt64, root, comm, s));
    if (sizes[1]) NCCLC(c, ncclBroadcast(P.pts.p, P.pts.p, (size_t)sizes[1] * 2, ncclInt32, root, comm, s));
    if (sizes[2]) NCCLC(c, ncclBroadcast(T.xy.p, T.xy.p, (size_t)sizes[2] * 2, ncclInt32, root, comm, s));
    NCCLC(c, ncclGroupEnd());
    HIPC(c, hipStreamSynchronize(s));
    P.n = sizes[0]; P.total = sizes[1]; T.n = sizes[2];
    return 0;
}
