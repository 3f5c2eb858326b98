"""Pins the CPU restatement (oracle/) against golden vectors captured from the reference's OWN pure
numpy/Python functions (tests/golden/make_golden.py -> golden_pure.npz).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
from util import load, unflat, same_polys

G = load("golden_pure.npz")


@pytest.mark.parametrize("t", range(4))
def test_thinning_and_trace_04(t):
    skel = O.thin_rot(G[f"thin{t}_in"])
    assert np.array_equal(skel, G[f"thin{t}_out"])                       # 04:35-99
    assert same_polys(O.trace(skel), unflat(G, f"trace{t}"))             # 04:102-211 (order incl.)


@pytest.mark.parametrize("t", range(3))
def test_zhang_suen_fast_08(t):
    assert np.array_equal(O.zs_std(G[f"zs{t}_in"]), G[f"zs{t}_out"])     # 08:342-372


@pytest.mark.parametrize("i", range(4))
def test_scale_one_05(i):
    sx, sy, dx, dy = G[f"scale{i}_prm"]
    assert same_polys(O.scale(unflat(G, f"scale{i}_in"), sx, sy, dx, dy), unflat(G, f"scale{i}_out"))


@pytest.mark.parametrize("i", range(6))
def test_resample_arclen_08(i):
    p = G[f"resample{i}_in"]
    closed = len(p) > 2 and bool(np.all(p[0] == p[-1]))
    out, ps = O.resample_arclen(p, closed, float(G[f"resample{i}_step"]))
    assert ps == bool(G[f"resample{i}_pass"])
    assert np.array_equal(np.asarray(out, np.float64), G[f"resample{i}_out"])   # bit-exact float64


@pytest.mark.parametrize("i", range(4))
def test_split_on_long_jumps_08_and_10(i):
    p = G[f"jump{i}_in"]
    assert same_polys(O.split_jumps(p, 80.0, 8), unflat(G, f"jump{i}_out08"))
    assert same_polys(O.split_jumps(p, 80.0, 10), unflat(G, f"jump{i}_out10"))


@pytest.mark.parametrize("i", range(3))
def test_reorders_07_08_10(i):
    polys = unflat(G, f"reorder{i}_in")
    assert same_polys(O.reorder(polys, 0), unflat(G, f"reorder{i}_out08"))   # perimeter: numpy pairwise f32
    assert same_polys(O.reorder(polys, 1), unflat(G, f"reorder{i}_out10"))   # arcLength: cv2 stand-in
    assert same_polys(O.sort07(polys), unflat(G, f"reorder{i}_out07"))


def test_cluster_by_overlap_08():
    groups = O.cluster_by_overlap(G["cluster_in"])
    gid = np.zeros(len(G["cluster_in"]), np.int32)
    for k, grp in enumerate(groups):
        gid[grp] = k
    assert np.array_equal(gid, G["cluster_out"])


@pytest.mark.parametrize("i", range(3))
def test_bfs_and_best_path_08(i):
    comp = G[f"bfs{i}_img"]; a, b = (tuple(int(v) for v in r) for r in G[f"bfs{i}_ab"])
    assert np.array_equal(np.array(O.bfs_path(comp, a, b), np.int32).reshape(-1, 2), G[f"bfs{i}_path"])
    assert np.array_equal(np.array(O.component_best_path(comp, None, None, 4), np.int32).reshape(-1, 2), G[f"bfs{i}_best_none"])
    assert np.array_equal(np.array(O.component_best_path(comp, a, b, 4), np.int32).reshape(-1, 2), G[f"bfs{i}_best_ab"])


@pytest.mark.parametrize("i", range(3))
def test_cut_poly_against_mask_10(i):
    assert same_polys(O.cut_poly(G[f"cut{i}_in"], G["cut_mask"], 1.0), unflat(G, f"cut{i}_out"))


@pytest.mark.parametrize("i", range(3))
def test_build_ops_12(i):
    lines = unflat(G, f"ops{i}_lines"); taps = [tuple(int(v) for v in t) for t in G[f"ops{i}_taps"]]
    ops = O.build_ops12(lines, taps, 80.0)
    kinds = np.array([0 if o["type"] == "line" else 1 for o in ops], np.int32)
    assert np.array_equal(kinds, G[f"ops{i}_kinds"])
    got = [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops]
    assert same_polys(got, unflat(G, f"ops{i}_out"))


@pytest.mark.parametrize("i", range(3))
def test_virtual_draw_08(i):
    mask = np.zeros((500, 700), np.uint8)
    for j, p in enumerate(unflat(G, f"vdraw{i}_in")):
        segs = O.virtual_draw08(p, mask)
        assert same_polys(segs, unflat(G, f"vdraw{i}_out{j}")), (i, j)
    assert np.array_equal(np.packbits(mask > 0), G[f"vdraw{i}_mask"])


def test_lab8_to_bgr_round_trip():
    """a6 (02:58-61): the palette's approx_bgr.  Pinned to the oracle's FORWARD 8-bit Lab tables: converting the result back gives
    the same Lab triple within 2 LSB for in-gamut colours (greys, tinted greys, saturated primaries), exact for black and white."""
    rng = np.random.default_rng(3)
    cols = [np.array([v, v, v]) for v in range(0, 256, 5)] + [rng.integers(0, 256, 3) for _ in range(300)]
    for bgr in cols:
        lab = O.bgr2lab(np.asarray(bgr, np.uint8).reshape(1, 1, 3)).reshape(3)
        back = O.lab8_to_bgr(lab)
        lab2 = O.bgr2lab(np.asarray(back, np.uint8).reshape(1, 1, 3)).reshape(3)
        assert np.abs(lab2.astype(int) - lab.astype(int)).max() <= 2, (bgr, lab, back, lab2)
    assert O.lab8_to_bgr(O.bgr2lab(np.zeros((1, 1, 3), np.uint8)).reshape(3)) == (0, 0, 0)
    assert O.lab8_to_bgr(O.bgr2lab(np.full((1, 1, 3), 255, np.uint8)).reshape(3)) == (255, 255, 255)


def test_product_lab8_to_bgr_equals_oracle():
    import importlib.util, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "omnirevolve-image-processor_amd"))
    from orip.stages import lab8_to_bgr
    rng = np.random.default_rng(4)
    for _ in range(500):
        lab = rng.integers(0, 256, 3).astype(np.uint8)
        assert lab8_to_bgr(lab) == O.lab8_to_bgr(lab), lab
