"""Multi-GPU layout of the path: one process per GPU.  The exchange itself runs inside liborip.so (csrc/comm.hip:
RCCL broadcasts over xGMI between the device-resident list slots, no host staging, no torch); torch.distributed is plumbing only
(the launcher's rendezvous, the barrier of bench.py, and the hand-over of the RCCL unique id).  The gloo functions below move the
same lists through host memory: they are the CPU test double of the exchange (tests/test_parallel_gloo.py) and the rehearsal path
when several ranks have to share one card (RCCL refuses two ranks on one GPU).

The path shards by COLOUR LAYER.  Stage 02 (k-means fit on the fixed-seed subsample + assignment) is deterministic and
cheap, so every rank runs it on the whole image instead of row-sharding it and all-gathering the label map.  From the
mask morphology through stage 08 every layer is independent (03:45, 04:234, 05:114, 07:99, 08:561), so rank r owns the
cluster layers {l : l % world == r}.  Stage 10 walks the layers dark->light against ONE cumulative raster
(10:215,236-267) -- the only real exchange step of the path: the per-layer (lines_intra, taps_intra) lists are
all-gathered (RCCL all_gather over xGMI with backend "nccl"; gloo in the CPU tests), stage 10 is then replicated on
every rank (deterministic), and each rank orders (stage 12) the layers it owns.  run_path_sharded streams that exchange
layer by layer (broadcast_layer) so that stage 10 overlaps the per-layer pipelines; exchange_layer_lists is the one-shot form.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

Lists = Tuple[List[np.ndarray], List[Tuple[int, int]]]


def owned_layers(K: int, rank: int, world: int) -> List[int]:
    return [l for l in range(K) if l % world == rank]


def pack_layers(local: Dict[int, Lists], K: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> (sizes int64 [K,3] = (#lines, #points, #taps) rows of owned layers, payload int32 flat)."""
    sizes = np.zeros((K, 3), np.int64)
    parts: List[np.ndarray] = []
    for l in sorted(local):
        lines, taps = local[l]
        flat = [np.asarray(p).reshape(-1, 2).astype(np.int32) for p in lines]
        lens = np.array([len(p) for p in flat], np.int32)
        pts = np.concatenate(flat, 0).reshape(-1) if flat else np.zeros(0, np.int32)
        tp = np.asarray(list(taps), np.int32).reshape(-1)
        sizes[l] = (len(flat), len(pts) // 2, len(tp) // 2)
        parts += [lens, pts.astype(np.int32), tp]
    payload = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)
    return sizes, payload


def unpack_layers(sizes: np.ndarray, payloads: Sequence[np.ndarray], K: int, world: int) -> Dict[int, Lists]:
    out: Dict[int, Lists] = {}
    for r in range(world):
        buf = payloads[r]; pos = 0
        for l in owned_layers(K, r, world):
            nl, npts, nt = (int(v) for v in sizes[l])
            lens = buf[pos:pos + nl]; pos += nl
            pts = buf[pos:pos + 2 * npts].reshape(-1, 2); pos += 2 * npts
            tp = buf[pos:pos + 2 * nt].reshape(-1, 2); pos += 2 * nt
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            out[l] = ([pts[off[i]:off[i + 1]].reshape(-1, 1, 2).copy() for i in range(nl)], [(int(x), int(y)) for x, y in tp])
    return out


def exchange_layer_lists(local: Dict[int, Lists], K: int, device=None) -> Dict[int, Lists]:
    """All-gather(v) of the per-layer (lines, taps) lists: sizes first (all_reduce of a [K,3] table whose rows are written by
    the owning rank only), then one padded int32 payload per rank (all_gather).  `device`: torch device of the collective
    buffers ("cuda:<local_rank>" for RCCL, "cpu" for gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return dict(local)
    dev = torch.device(device if device is not None else "cpu")
    sizes, payload = pack_layers(local, K)
    t_sizes = torch.from_numpy(sizes).to(dev)
    dist.all_reduce(t_sizes, op=dist.ReduceOp.SUM)
    sizes = t_sizes.cpu().numpy()
    per_rank = [int(sum(sizes[l, 0] + 2 * sizes[l, 1] + 2 * sizes[l, 2] for l in owned_layers(K, r, world))) for r in range(world)]
    cap = max(1, max(per_rank))
    send = torch.zeros(cap, dtype=torch.int32, device=dev)
    if len(payload):
        send[:len(payload)] = torch.from_numpy(payload).to(dev)
    recv = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(recv, send)
    payloads = [recv[r][:per_rank[r]].cpu().numpy() for r in range(world)]
    return unpack_layers(sizes, payloads, K, world)


def broadcast_layer(lists, owner: int, device=None) -> Lists:
    """One layer's (lines, taps) from rank `owner` to everybody: a [3] size row, then one int32 payload.  `lists` is only read
    on the owner.  Collective: every rank calls it for the same layers in the same order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return lists
    rank = dist.get_rank()
    dev = torch.device(device if device is not None else "cpu")
    meta = np.zeros(3, np.int64); payload = np.zeros(0, np.int32)
    if rank == owner:
        sizes, payload = pack_layers({0: lists}, 1)
        meta = sizes[0].copy()
    t_meta = torch.from_numpy(meta).to(dev)
    dist.broadcast(t_meta, src=owner)
    meta = t_meta.cpu().numpy()
    n = int(meta[0] + 2 * meta[1] + 2 * meta[2])
    buf = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    if rank == owner and n:
        buf[:n] = torch.from_numpy(payload).to(dev)
    dist.broadcast(buf, src=owner)
    if rank == owner:
        return lists
    return unpack_layers(meta.reshape(1, 3), [buf[:n].cpu().numpy()], 1, 1)[0]


class RcclComm:
    """The product exchange: orip_bcast_layer (RCCL, device to device).  The unique id travels through the launcher's process group."""
    kind = "rccl (orip_bcast_layer, device to device)"

    def __init__(self, dev, rank: int, world: int):
        import torch.distributed as dist
        box = [dev.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        dev.comm_init(box[0], rank, world)
        self.dev = dev

    def bcast(self, owner: int, my_slot: int, i_am_owner: bool):
        self.dev.bcast_layer(owner, my_slot)

    def close(self):
        self.dev.comm_destroy()


class GlooComm:
    """The same lists through torch.distributed broadcasts: packed on the host, broadcast on `coll_device` (this rank's GPU under the
    nccl backend = RCCL over xGMI; host memory under gloo: the CPU tests and rehearsals with several ranks on one card)."""
    kind = "torch.distributed broadcast (RCCL under the nccl backend, gloo otherwise; lists packed on the host)"

    def __init__(self, dev, coll_device="cpu"):
        self.dev = dev; self.coll_device = coll_device

    def bcast(self, owner: int, my_slot: int, i_am_owner: bool):
        from . import lib as _l
        d = self.dev
        lists = (d.get_polys(_l.SLOT_LINES_INTRA, my_slot), d.get_taps(_l.TAPS_INTRA, my_slot)) if i_am_owner else None
        lines, taps = broadcast_layer(lists, owner, self.coll_device)
        if not i_am_owner:
            d.set_polys(_l.SLOT_LINES_INTRA, my_slot, lines)
            d.set_taps(_l.TAPS_INTRA, my_slot, taps)

    def close(self):
        pass


def make_comm(dev, rank: int, world: int, coll_device=None):
    """The exchange of a sharded step.  Default: the layer lists travel through torch.distributed broadcasts (RCCL over xGMI when the
    process group's backend is nccl and `coll_device` is this rank's GPU, gloo through host memory otherwise) -- the path every test
    and rehearsal has run.  ORIP_EXCHANGE=rccl selects the broadcast inside liborip.so (orip_bcast_layer, slot to slot, no host
    staging): it has only ever run with a one-rank communicator (no multi-GPU node was available to this build), so it is opt-in
    until a multi-GPU run has passed once."""
    import os
    import torch.distributed as dist
    if os.environ.get("ORIP_EXCHANGE", "").lower() == "rccl" and dist.get_backend() == "nccl":
        return RcclComm(dev, rank, world)
    return GlooComm(dev, coll_device or "cpu")


def comm_of(dev, rank: int, world: int, coll_device=None):
    """The communicator of `dev`, created on first use and kept: orip_comm_init refuses a second communicator on one context, so a
    sharded step must not build a new one every time it is called without `comm`."""
    cached = getattr(dev, "_comm", None)
    if cached is not None and cached[0] == (rank, world):
        return cached[1]
    if cached is not None:
        cached[1].close()
    comm = make_comm(dev, rank, world, coll_device)
    dev._comm = ((rank, world), comm)
    return comm


def run_path_sharded(dev, cfg, H: int, W: int, rank: int, world: int, coll_device=None, fetch_lines: bool = False, comm=None, timings: dict | None = None):
    """One step of stages 02 -> 12 with the image already resident on `dev` (orip.device.Device).  The op rows of every owned layer
    come back to the host; fetch_lines: so do the points of its lines (what ops.pkl holds, 12:206-208).
    world > 1: rank r ends with the lists and ops of its layers owned_layers(K, r, world) under local indices 0..len-1.

    world > 1: every rank pipelines its own layers (04 -> 08, one lane each).  Stage 10 is replicated and streamed: the
    layers are visited dark -> light; the owner of the next layer waits for its pipeline, broadcasts the layer's
    (lines_intra, taps_intra) and everybody cuts it against the shared raster.  So, as on one GPU, stage 10 of the early
    layers runs underneath the 04-08 work of the heavy ones, and a layer's stage 12 starts as soon as it has left stage 10.

    timings (world > 1): filled with this rank's host-side stage times of the step, in ms from its start -- `raster` (stage 02 returned),
    `front_ready` {global layer: its 04 -> 08 pipeline done}, `own_wait` (waiting for own layers in stage 10's visit order),
    `exchange` (inside the broadcasts: the wait for the other owners' layers plus the transfer), `cross` (stage 10 calls), `total`."""
    import time as _time
    from . import lib as _l
    from . import stages as S

    t_start = _time.perf_counter()
    ms_since = lambda: (_time.perf_counter() - t_start) * 1e3

    names = list(cfg.color_names)
    K = max(2, len(names))
    lnames = S.cluster_names(cfg)[:K]
    dev.contours_reserve(K if world == 1 else len(owned_layers(K, rank, world)))      # memo planes cleared under the k-means fit
    centers, _ = dev.kmeans_fit(S.subsample_indices(H * W), K)
    dev.extract_layers(centers, want_counts=False)
    order = sorted(range(K), key=lambda l: (S.darkness_rank10(lnames[l]), names.index(lnames[l])))
    R = S.r_insert12(cfg)

    def tail(i):
        ops = dev.plot_order(i, R)
        if fetch_lines:
            dev.get_polys_flat(_l.SLOT_LINES_CROSS, i)
        return ops

    if world == 1:
        S._detect_edges_resident(dev, cfg)
        res = S.run_layer_pipelines(dev, cfg, W, H, range(K), order, 12, tail)
        return sum(len(o) for o in res.values())
    if comm is None:
        comm = comm_of(dev, rank, world, coll_device)       # cached on the device: a second sharded step reuses it
    mine = owned_layers(K, rank, world)              # global layer ids; held under local indices 0..len(mine)-1
    T = {"rank": rank, "layers": list(mine), "raster": round(ms_since(), 2), "front_ready": {}, "own_wait": 0.0, "exchange": 0.0, "cross": 0.0}
    ready, errors = {}, []
    if mine:
        dev.keep_layers(mine)
        S._detect_edges_resident(dev, cfg)
        dev.contours_prepare()
        order_local = [mine.index(g) for g in order if g in mine]
        front = S.layer_front(dev, cfg, W, H, 8)

        def timed_front(i):
            try:
                front(i)
            finally:
                T["front_ready"][mine[i]] = round(ms_since(), 2)

        ready, errors = S._start_fronts(timed_front, range(len(mine)), order_local)
    pool = S._get_pool()
    dev.dedup_cross_begin(S.params10(cfg))
    # Every list of an owned layer lives under its LOCAL index i (CONTOURS ... LINES_CROSS, taps, ops) and all per-layer calls of
    # that layer run on lane i + 1.  (r01 addressed stage 10's output and stage 12 by the global id g: orip_plot_order(g) then ran
    # on the lane of local layer g -- a different layer whose 04->08 pipeline was still growing that lane's scratch buffers -- and
    # the GPU faulted at full size.  The library now also refuses a second call on a busy lane.)
    stage_slot = _l.MAX_LAYERS - 1                   # remote layers pass through this spare slot
    if len(mine) >= _l.MAX_LAYERS:
        raise ValueError(f"rank {rank} owns {len(mine)} layers: the spare slot {stage_slot} for remote layers would be one of them")
    tails = []
    for g in order:
        owner = g % world
        if owner == rank:
            i = mine.index(g)
            t = ms_since(); ready[i].wait(); T["own_wait"] += ms_since() - t
            failed = bool(errors)
            if failed:                                  # the other ranks are waiting in the collective: never skip it, send an empty layer
                dev.set_polys(_l.SLOT_LINES_INTRA, i, []); dev.set_taps(_l.TAPS_INTRA, i, [])
            t = ms_since(); comm.bcast(owner, i, True); T["exchange"] += ms_since() - t
            if failed:
                continue
            t = ms_since(); dev.dedup_cross_layer(i, src_layer=i, defer_reorder=True); T["cross"] += ms_since() - t      # tail(): plot_order reorders the lines first, on lane i + 1
            tails.append(pool.submit(tail, i))
        else:
            t = ms_since(); comm.bcast(owner, stage_slot, False); T["exchange"] += ms_since() - t
            t = ms_since(); dev.dedup_cross_layer(stage_slot, src_layer=stage_slot); T["cross"] += ms_since() - t
    n_ops = sum(len(f.result()) for f in tails)
    if timings is not None:
        for k in ("own_wait", "exchange", "cross"):
            T[k] = round(T[k], 2)
        T["total"] = round(ms_since(), 2)
        timings.clear(); timings.update(T)
    if errors:
        for e in ready.values():
            e.wait()
        raise errors[0]
    return n_ops
