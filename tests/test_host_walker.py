"""CPU unit test of the product's serial walker (csrc/walker.h compiled with g++ through tests/host/walk_harness.cpp):
the walk logic -- phases, guards, cycle fast-forward of the bounce tails -- against the reference's golden traces
and the oracle on random skeletons.  The GPU launch of the same header is covered by tests/test_gpu_raster.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from util import load, unflat, same_polys

HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return C.CDLL(os.path.join(HOST, "libwalk_harness.so"))


def _run(L, skel, factor=16):
    skel = np.ascontiguousarray(skel, np.uint8); H, W = skel.shape
    cap = 8_000_000
    off = np.zeros(400_000, np.int64); pts = np.zeros((cap, 2), np.int32); n = C.c_int64(); t = C.c_int64()
    rc = L.walk_harness(skel.ctypes.data_as(C.c_void_p), H, W, off.ctypes.data_as(C.c_void_p), len(off),
                        pts.ctypes.data_as(C.c_void_p), cap, C.byref(n), C.byref(t), int(factor))
    assert rc == 0, rc
    return [pts[off[i]:off[i + 1]].reshape(-1, 1, 2) for i in range(n.value)]


@pytest.mark.parametrize("t", range(4))
def test_walker_header_vs_reference_traces(harness, t):
    G = load("golden_pure.npz")
    want = [p for p in unflat(G, f"trace{t}") if len(p) >= 5]
    assert same_polys(_run(harness, G[f"thin{t}_out"]), want)


@pytest.mark.parametrize("seed", range(6))
def test_walker_header_vs_oracle_random(harness, seed):
    rng = np.random.default_rng(seed)
    img = (rng.random((70, 90)) < [0.08, 0.2, 0.35, 0.5, 0.12, 0.3][seed]).astype(np.uint8) * 255
    skel = O.thin_rot(img) if seed % 2 == 0 else img          # raw noise exercises junction-rich, unthinned components too
    want = [p for p in O.trace(skel) if len(p) >= 5]
    assert same_polys(_run(harness, skel), want)


def test_walker_memo_on_loops_with_many_leftovers(harness):
    """Thick, unthinned rings and grids: many leftover pixels per component, long shared bounce trajectories."""
    img = np.zeros((120, 160), np.uint8)
    for r, cx, cy in [(30, 45, 50), (22, 110, 60), (12, 80, 95)]:
        yy, xx = np.mgrid[0:120, 0:160]
        d = np.hypot(yy - cy, xx - cx)
        img[(d > r - 1.6) & (d < r + 1.6)] = 255
    img[20:100:9, 10:150] = 255; img[20:100, 10:150:11] = 255
    for skel in (img, O.thin_rot(img)):
        want = [p for p in O.trace(skel) if len(p) >= 5]
        assert same_polys(_run(harness, skel), want)
        assert same_polys(_run(harness, skel, factor=64), want)       # a different log layout must not change the result
