"""Layer-sharded path (world_size 2 and 3 ranks sharing the one GPU of the test box, gloo collectives) against the
single-process path: every rank must end with the ops of its own layers identical to the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, cold=False):
    try:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
        import torch.distributed as dist
        from orip import lib as L, parallel as P, stages as S
        from orip.config import Config
        from orip.device import Device
        from orip.synth import synth_image, layer_names
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if cold:
            # full default canvas, an image large enough that every lane's scratch has to grow INSIDE the sharded step: the sharded run
            # comes first, on a fresh context (r01's fault needed exactly that: buffers growing under another layer's running pipeline)
            K, H, W = 8, 1024, 1024
            img = np.ascontiguousarray(synth_image(4096, 4096, K)[:H, :W])
            cfg = Config(); cfg.color_names = layer_names(K)
        else:
            K, H, W = 5, 384, 448
            img = synth_image(H, W, K, seed=7, sigma=4.0)
            cfg = Config(); cfg.color_names = layer_names(K); cfg.pixels_per_mm = 8
        dev = Device(0)

        def snapshot(layers):
            """{global layer id: artefacts}; `layers` lists the global ids in the order of the device's (local) layer indices"""
            R = S.r_insert12(cfg)
            return {g: (dev.get_polys_flat(L.SLOT_LINES_CROSS, i), dev.get_taps(L.TAPS_CROSS, i), dev.plot_order(i, R).copy()) for i, g in enumerate(layers)}

        mine = P.owned_layers(K, rank, world)
        if cold:
            dev.set_image(img)
            P.run_path_sharded(dev, cfg, H, W, rank, world, "cpu")
            got = snapshot(mine)
            dev.close(); dev = Device(0)                               # reference run on another fresh context
            dev.set_image(img)
            P.run_path_sharded(dev, cfg, H, W, 0, 1)
            want = snapshot(range(K))
        else:
            dev.set_image(img)
            P.run_path_sharded(dev, cfg, H, W, 0, 1)                   # the whole path on this process
            want = snapshot(range(K))
            dev.set_image(img)
            P.run_path_sharded(dev, cfg, H, W, rank, world, "cpu")
            dev.set_image(img)
            times = {}
            P.run_path_sharded(dev, cfg, H, W, rank, world, "cpu", timings=times)     # a second sharded step without `comm`: the communicator is reused, not rebuilt
            assert dev._comm is not None
            # the per-rank stage times bench.py prints for N > 1: this rank's layers, each front's finish time, and the parts of the stage-10 visit
            assert times["rank"] == rank and times["layers"] == mine and sorted(times["front_ready"]) == sorted(mine)
            assert 0 < times["raster"] <= times["total"] and all(times["raster"] <= t <= times["total"] for t in times["front_ready"].values())
            assert min(times["own_wait"], times["exchange"], times["cross"]) >= 0 and times["own_wait"] + times["exchange"] + times["cross"] <= times["total"]
            got = snapshot(mine)
        ok = True
        for g in mine:
            (oa, pa), ta, opa = want[g]; (ob, pb), tb, opb = got[g]
            ok &= np.array_equal(oa, ob) and np.array_equal(pa, pb) and ta == tb and np.array_equal(opa, opb)
        dev.close()
        dist.barrier(); dist.destroy_process_group()
        q.put((rank, bool(ok), len(mine)))
    except BaseException as e:   # noqa: BLE001 - reported to the parent
        q.put((rank, False, repr(e)))


def _run(world, cold):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cold)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(world):
    _run(world, cold=False)


def test_sharded_cold_start_default_canvas():
    """The r01 fault case in small: a fresh context whose first work is the sharded step, lane buffers growing while other layers run."""
    _run(2, cold=True)


def test_rccl_exchange_single_rank():
    """The RCCL entry points of liborip.so on the one GPU of the test box: communicator of size 1, broadcast of a layer from rank 0 leaves
    the lists as they were (more ranks on one card are refused by RCCL: the N > 1 exchange is covered by the gloo double and by the
    driver's multi-GPU bench)."""
    from orip import lib as L
    from orip.device import Device
    dev = Device(0)
    lines = [np.array([[1, 2], [3, 4], [5, 6]], np.int32).reshape(-1, 1, 2), np.array([[7, 8], [9, 10]], np.int32).reshape(-1, 1, 2)]
    taps = [(11, 12), (13, 14), (15, 16)]
    dev.set_polys(L.SLOT_LINES_INTRA, 2, lines); dev.set_taps(L.TAPS_INTRA, 2, taps)
    dev.comm_init(dev.comm_unique_id(), 0, 1)
    dev.bcast_layer(0, 2)
    got = dev.get_polys(L.SLOT_LINES_INTRA, 2)
    assert len(got) == 2 and all(np.array_equal(a, b) for a, b in zip(got, lines)) and dev.get_taps(L.TAPS_INTRA, 2) == taps
    dev.set_polys(L.SLOT_LINES_INTRA, 3, []); dev.set_taps(L.TAPS_INTRA, 3, [])
    dev.bcast_layer(0, 3)                                            # an empty layer is a valid broadcast
    assert dev.get_polys(L.SLOT_LINES_INTRA, 3) == [] and dev.get_taps(L.TAPS_INTRA, 3) == []
    with pytest.raises(Exception):
        dev.comm_init(dev.comm_unique_id(), 0, 1)                    # second init is refused
    dev.comm_destroy()
    dev.close()
