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


def _worker(rank, world, port, q):
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
        K, H, W = 5, 384, 448
        img = synth_image(H, W, K, seed=7, sigma=4.0)
        cfg = Config(); cfg.color_names = layer_names(K); cfg.pixels_per_mm = 8
        dev = Device(0)

        def snapshot(layers):
            """{global layer id: artefacts}; `layers` lists the global ids in the order of the device's (local) layer indices"""
            R = S.r_insert12(cfg)
            return {g: (dev.get_polys_flat(L.SLOT_LINES_CROSS, i), dev.get_taps(L.TAPS_CROSS, i), dev.plot_order(i, R).copy()) for i, g in enumerate(layers)}

        dev.set_image(img)
        P.run_path_sharded(dev, cfg, H, W, 0, 1)                       # the whole path on this process
        want = snapshot(range(K))
        dev.set_image(img)
        P.run_path_sharded(dev, cfg, H, W, rank, world, "cpu")
        mine = P.owned_layers(K, rank, world)
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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
