"""Pins the CPU restatement (oracle/) against the reference's OWN stage drivers 04 -> 12, run in the build
container on small images with tests/golden/cv2_standin.py supplying the OpenCV primitives
(golden_e2e_*.npz: control flow pinned by the reference, cv2 primitives unpinned).  Each stage is fed the
reference's recorded input, so a mismatch is attributed to the stage that caused it.  CPU only."""
import json

import numpy as np
import pytest

from oracle import oracle as O
from util import load, unflat, same_polys, poly_multiset


def _cfg(g):
    return json.loads(bytes(g["cfg_json"]).decode())


def _taps(a):
    return [(int(x), int(y)) for x, y in a]


@pytest.fixture(scope="module", params=["a", "b"])
def G(request):
    return load(f"golden_e2e_{request.param}.npz")


def test_stage04_contours(G):
    cfg = _cfg(G); H, W = G["img"].shape[:2]
    for n in cfg["color_names"]:
        edges = np.unpackbits(G[f"edges_{n}"])[:H * W].reshape(H, W) * 255
        got = O.stage04(edges.astype(np.uint8))
        assert same_polys(got, unflat(G, f"contours_{n}")), n


def test_stage05_scale(G):
    cfg = _cfg(G); H, W = G["img"].shape[:2]
    for n in cfg["color_names"]:
        assert same_polys(O.stage05(unflat(G, f"contours_{n}"), W, H, cfg), unflat(G, f"scaled_{n}")), n


def test_stage07_sort(G):
    for n in _cfg(G)["color_names"]:
        assert same_polys(O.sort07(unflat(G, f"scaled_{n}")), unflat(G, f"sorted_{n}")), n


def test_stage08_intra(G):
    cfg = _cfg(G); prm = O.derived08(cfg)
    for n in cfg["color_names"]:
        lines, taps = O.stage08_layer(unflat(G, f"sorted_{n}"), prm)
        assert taps == _taps(G[f"taps_intra_{n}"]), n
        assert same_polys(lines, unflat(G, f"lines_intra_{n}")), n


def test_stage10_cross(G):
    cfg = _cfg(G)
    intra = {n: (unflat(G, f"lines_intra_{n}"), _taps(G[f"taps_intra_{n}"])) for n in cfg["color_names"]}
    out = O.stage10(intra, cfg)
    for n in cfg["color_names"]:
        assert out[n][1] == _taps(G[f"taps_cross_{n}"]), n
        assert same_polys(out[n][0], unflat(G, f"lines_cross_{n}")), n


def test_stage12_ops(G):
    cfg = _cfg(G)
    for n in cfg["color_names"]:
        ops = O.stage12(unflat(G, f"lines_cross_{n}"), _taps(G[f"taps_cross_{n}"]), cfg)
        kinds = np.array([0 if o["type"] == "line" else 1 for o in ops], np.int32)
        assert np.array_equal(kinds, G[f"ops_kinds_{n}"]), n
        got = [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops]
        assert same_polys(got, unflat(G, f"ops_{n}")), n


def test_chain_02_to_12_matches_reference_chain(G):
    """Whole oracle chain from the image: same final ops as the reference chain from the same edges."""
    cfg = _cfg(G)
    r = O.run_pipeline(G["img"], cfg)
    for n in cfg["color_names"]:
        got = [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in r["ops"][n]]
        assert same_polys(got, unflat(G, f"ops_{n}")), n
