"""N>1 path on CPU: the all-gather(v) of per-layer (lines, taps) lists that precedes stage 10, over gloo with
world_size 2 and 3 (uneven ownership), checked against the single-process result.  No GPU involved."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make(K, seed=0):
    rng = np.random.default_rng(seed)
    out = {}
    for l in range(K):
        lines = [rng.integers(0, 9000, (int(rng.integers(2, 30)), 1, 2)).astype(np.int32) for _ in range(int(rng.integers(0, 12)))]
        taps = [(int(x), int(y)) for x, y in rng.integers(0, 9000, (int(rng.integers(0, 6)), 2))]
        out[l] = (lines, taps)
    return out


def _worker(rank, world, port, K, q):
    sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
    import torch.distributed as dist
    from orip import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _make(K)
    local = {l: full[l] for l in P.owned_layers(K, rank, world)}
    got = P.exchange_layer_lists(local, K, "cpu")
    ok = sorted(got) == list(range(K))
    for l in range(K):
        ok &= got[l][1] == full[l][1] and len(got[l][0]) == len(full[l][0])
        ok &= all(np.array_equal(a, b) for a, b in zip(got[l][0], full[l][0]))
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _worker_stream(rank, world, port, K, q):
    """The streamed form used by run_path_sharded: layers visited in a fixed order, the owner broadcasts each one."""
    sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
    import torch.distributed as dist
    from orip import parallel as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _make(K, seed=2)
    full[K // 2] = ([], [])                                   # a layer with nothing in it
    order = list(np.random.default_rng(5).permutation(K))     # the dark -> light order is a permutation of the layers
    ok = True
    for g in order:
        owner = int(g) % world
        got = P.broadcast_layer(full[g] if owner == rank else None, owner, "cpu")
        ok &= got[1] == full[g][1] and len(got[0]) == len(full[g][0])
        ok &= all(np.array_equal(a, b) for a, b in zip(got[0], full[g][0]))
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,K", [(2, 8), (3, 5)])
def test_broadcast_layer_stream_gloo(world, K):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker_stream, args=(r, world, port, K, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok in res), res


@pytest.mark.parametrize("world,K", [(2, 8), (3, 8), (2, 3)])
def test_exchange_layer_lists_gloo(world, K):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_pack_unpack_roundtrip_and_ownership():
    sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
    from orip import parallel as P
    K, world = 8, 3
    full = _make(K, seed=4)
    sizes = np.zeros((K, 3), np.int64); payloads = []
    for r in range(world):
        s, p = P.pack_layers({l: full[l] for l in P.owned_layers(K, r, world)}, K)
        sizes += s; payloads.append(p)
    got = P.unpack_layers(sizes, payloads, K, world)
    assert sorted(sum((P.owned_layers(K, r, world) for r in range(world)), [])) == list(range(K))
    for l in range(K):
        assert got[l][1] == full[l][1]
        assert all(np.array_equal(a, b) for a, b in zip(got[l][0], full[l][0]))
