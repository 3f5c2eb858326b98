"""GPU parity tests (through the C ABI) for the vector half of the path: stages 05, 07, 08, 10, 12.
Inputs and expected outputs come from the reference's own stage drivers (golden_e2e_*.npz) and from the oracle
on seeded random polylines.  Bit-exact: integer point lists, tap lists, op order."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from util import load, unflat, same_polys


@pytest.fixture(scope="module")
def dev():
    from orip.device import Device
    d = Device(0)
    yield d
    d.close()


def _cfgobj(d):
    from orip.config import Config
    c = Config()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def _taps(a):
    return [(int(x), int(y)) for x, y in a]


def _rand_polys(rng, n, lo=2, hi=40, span=8000, step=15, closed_p=0.3):
    out = []
    for _ in range(n):
        m = int(rng.integers(lo, hi))
        p = (np.cumsum(rng.integers(-step, step + 1, (m, 2)), axis=0) + rng.integers(200, span, 2)).astype(np.int32)
        if rng.random() < closed_p and m > 3:
            p[-1] = p[0]
        out.append(p.reshape(-1, 1, 2))
    return out


@pytest.fixture(scope="module", params=["a", "b"])
def G(request):
    return load(f"golden_e2e_{request.param}.npz")


def test_stage05_07_golden(dev, G):
    from orip import stages as S
    cfgd = json.loads(bytes(G["cfg_json"]).decode()); cfg = _cfgobj(cfgd); H, W = G["img"].shape[:2]
    for n in cfgd["color_names"]:
        scaled = S.scale_vectors(unflat(G, f"contours_{n}"), W, H, cfg, dev)
        assert same_polys(scaled, unflat(G, f"scaled_{n}")), n
        assert same_polys(S.sort_contours(unflat(G, f"scaled_{n}"), dev), unflat(G, f"sorted_{n}")), n


@pytest.mark.parametrize("seed", range(3))
def test_stage07_random_vs_oracle(dev, seed):
    from orip import stages as S
    rng = np.random.default_rng(seed)
    polys = _rand_polys(rng, [1, 37, 1500][seed])
    assert same_polys(S.sort_contours(polys, dev), O.sort07(polys))


@pytest.mark.parametrize("seed", range(2))
def test_stage07_grid_search_ties_and_long_jumps(dev, seed):
    """Thousands of contours on a coarse lattice, in a few dense clusters far apart: equal distances everywhere (index
    tie-break) and exhausted neighbourhoods (the grid search has to widen to the whole list)."""
    from orip import stages as S
    rng = np.random.default_rng(40 + seed)
    polys = []
    centres = rng.integers(500, 7500, (6, 2))
    for i in range([3000, 9000][seed]):
        c = centres[int(rng.integers(0, 6))] if rng.random() < 0.9 else rng.integers(0, 8000, 2)
        m = int(rng.integers(2, 6))
        p = (c + rng.integers(-6, 7, (m, 2)) * 25).astype(np.int32)
        if rng.random() < 0.3 and m > 3:
            p[-1] = p[0]
        polys.append(p.reshape(-1, 1, 2))
    assert same_polys(S.sort_contours(polys, dev), O.sort07(polys))


def test_stage10_golden(dev, G):
    from orip import stages as S
    cfgd = json.loads(bytes(G["cfg_json"]).decode()); cfg = _cfgobj(cfgd)
    intra = {n: (unflat(G, f"lines_intra_{n}"), _taps(G[f"taps_intra_{n}"])) for n in cfgd["color_names"]}
    out = S.dedup_cross(intra, cfg, dev)
    for n in cfgd["color_names"]:
        assert out[n][1] == _taps(G[f"taps_cross_{n}"]), n
        assert same_polys(out[n][0], unflat(G, f"lines_cross_{n}")), n


@pytest.mark.parametrize("paint", ["discs", "separable"])
def test_stage10_random_vs_oracle(dev, monkeypatch, paint):
    """both ways of painting a layer's lines into the forbidden raster (one disc per vertex / separable distance passes)"""
    from orip import stages as S
    if paint == "separable":
        monkeypatch.setenv("ORIP_PAINT_SEPARABLE", "1")
    else:
        monkeypatch.delenv("ORIP_PAINT_SEPARABLE", raising=False)
    rng = np.random.default_rng(5)
    cfgd = dict(O.DEFAULTS, pixels_per_mm=8, color_names=["layer_dark", "layer_mid", "x_extra", "layer_light"])   # canvas 1680x2376
    cfg = _cfgobj(cfgd)
    intra = {}
    for n in cfgd["color_names"]:
        lines = _rand_polys(rng, 60, lo=2, hi=25, span=1500, step=40, closed_p=0.0)
        taps = _taps(rng.integers(-30, 1700, (25, 2)))
        intra[n] = (lines, taps)
    want = O.stage10(intra, cfgd)
    got = S.dedup_cross(intra, cfg, dev)
    for n in cfgd["color_names"]:
        assert got[n][1] == want[n][1], n
        assert same_polys(got[n][0], want[n][0]), n


def test_stage12_golden(dev, G):
    from orip import stages as S
    cfgd = json.loads(bytes(G["cfg_json"]).decode()); cfg = _cfgobj(cfgd)
    for n in cfgd["color_names"]:
        ops = S.plot_order(unflat(G, f"lines_cross_{n}"), _taps(G[f"taps_cross_{n}"]), cfg, dev)
        kinds = np.array([0 if o["type"] == "line" else 1 for o in ops], np.int32)
        assert np.array_equal(kinds, G[f"ops_kinds_{n}"]), n
        got = [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops]
        assert same_polys(got, unflat(G, f"ops_{n}")), n


@pytest.mark.parametrize("case", [(0, 7), (9, 0), (300, 120), (1, 1), (60, 1500), (0, 900), (40, 2500, 40000)])
def test_stage12_random_vs_oracle(dev, case):
    """few / many lines and taps, taps only, lines only; the last case has coordinates beyond 2^14"""
    from orip import stages as S
    rng = np.random.default_rng(case[0] + case[1])
    span = case[2] if len(case) > 2 else 3000
    lines = _rand_polys(rng, case[0], lo=2, hi=12, span=span, step=25, closed_p=0.0)
    taps = _taps(rng.integers(0, span, (case[1], 2)))
    cfg = _cfgobj(dict(O.DEFAULTS))
    want = O.stage12(lines, taps, dict(O.DEFAULTS))
    got = S.plot_order(lines, taps, cfg, dev)
    assert [o["type"] for o in got] == [o["type"] for o in want]
    for a, b in zip(got, want):
        if a["type"] == "line":
            assert np.array_equal(a["points"], b["points"])
        else:
            assert (a["x"], a["y"]) == (b["x"], b["y"])


def test_stage08_golden(dev, G):
    from orip import stages as S
    cfgd = json.loads(bytes(G["cfg_json"]).decode()); cfg = _cfgobj(cfgd)
    for n in cfgd["color_names"]:
        lines, taps = S.dedup_layer(unflat(G, f"sorted_{n}"), cfg, dev)
        assert taps == _taps(G[f"taps_intra_{n}"]), n
        assert same_polys(lines, unflat(G, f"lines_intra_{n}")), n


@pytest.mark.parametrize("seed", range(3))
def test_stage08_random_vs_oracle(dev, seed):
    """Self-crossing random walks on a small canvas: exercises mask hits, hash hits, taps, clusters, anchors."""
    from orip import stages as S
    rng = np.random.default_rng(100 + seed)
    cfgd = dict(O.DEFAULTS, pixels_per_mm=[6, 8, 10][seed])
    cfg = _cfgobj(cfgd)
    W, H = O.canvas_size(cfgd)
    polys = []
    for _ in range([25, 60, 120][seed]):
        m = int(rng.integers(2, 80))
        p = (np.cumsum(rng.integers(-22, 23, (m, 2)), axis=0) + rng.integers(50, min(W, H) - 50, 2)).astype(np.int32)
        if rng.random() < 0.25 and m > 3:
            p[-1] = p[0]
        polys.append(p.reshape(-1, 1, 2))
    polys += [np.array([[5, 5], [9, 9], [14, 6]], np.int32).reshape(-1, 1, 2), np.array([[-40, 30], [60, 35], [90, -20]], np.int32).reshape(-1, 1, 2)]
    want_l, want_t = O.stage08_layer(polys, O.derived08(cfgd))
    got_l, got_t = S.dedup_layer(polys, cfg, dev)
    assert got_t == want_t
    assert same_polys(got_l, want_l), (len(got_l), len(want_l))


def test_stage08_component_size_classes(dev, monkeypatch):
    """Stage 08-B keeps small skeleton components in LDS, larger ones in a bigger LDS layout, the rest in global scratch:
    with the capacities forced down every class is exercised on the same input and must give the same lines."""
    from orip import stages as S
    rng = np.random.default_rng(321)
    cfgd = dict(O.DEFAULTS, pixels_per_mm=8)
    cfg = _cfgobj(cfgd)
    W, H = O.canvas_size(cfgd)
    polys = []
    for _ in range(80):
        m = int(rng.integers(2, 120))
        p = (np.cumsum(rng.integers(-22, 23, (m, 2)), axis=0) + rng.integers(50, min(W, H) - 50, 2)).astype(np.int32)
        polys.append(p.reshape(-1, 1, 2))
    want_l, want_t = O.stage08_layer(polys, O.derived08(cfgd))
    for caps in ["16,64", "8,8", None]:
        if caps:
            monkeypatch.setenv("ORIP_COMP_CAPS", caps)
        else:
            monkeypatch.delenv("ORIP_COMP_CAPS", raising=False)
        got_l, got_t = S.dedup_layer(polys, cfg, dev)
        assert got_t == want_t, caps
        assert same_polys(got_l, want_l), (caps, len(got_l), len(want_l))


def test_stage08_tail_simulation_both_forms(dev, monkeypatch):
    """The tail simulation runs in a parallel form (prefix sums + a margin test) with the sequential float64 recurrence as its
    fallback; forcing the fallback everywhere must give the same lines (and both must match the oracle)."""
    from orip import stages as S
    rng = np.random.default_rng(77)
    cfgd = dict(O.DEFAULTS, pixels_per_mm=10)
    cfg = _cfgobj(cfgd)
    W, H = O.canvas_size(cfgd)
    polys = []
    for _ in range(40):
        m = int(rng.integers(20, 400))
        p = (np.cumsum(rng.integers(-9, 10, (m, 2)), axis=0) + rng.integers(100, min(W, H) - 100, 2)).astype(np.int32)
        polys.append(np.concatenate([p, p[::-1], p]).reshape(-1, 1, 2))       # retraced paths: the tail rule decides what survives
    want_l, want_t = O.stage08_layer(polys, O.derived08(cfgd))
    for seq in [False, True]:
        if seq:
            monkeypatch.setenv("ORIP_TAIL_SEQ", "1")
            monkeypatch.setenv("ORIP_CAPS_TINY", "1")      # and the capsule table starts too small: growth path
            monkeypatch.setenv("ORIP_HASH_SORT", "1")      # and _PointHash.near through the sorted buckets instead of the direct comparison
        else:
            monkeypatch.delenv("ORIP_TAIL_SEQ", raising=False)
            monkeypatch.delenv("ORIP_HASH_SORT", raising=False)
        got_l, got_t = S.dedup_layer(polys, cfg, dev)
        assert got_t == want_t, seq
        assert same_polys(got_l, want_l), (seq, len(got_l), len(want_l))


def test_stage08_cumulative_lengths_long_polylines(dev, monkeypatch):
    """np.cumsum of float32 segment lengths (08:58) for polylines long enough for the wave kernels: the integer-scan form (binade by binade,
    real float adds only at binade crossings and round-half ties) and the serial chain of wave-shifted adds must both give the oracle's lines."""
    from orip import stages as S
    rng = np.random.default_rng(2024)
    cfgd = dict(O.DEFAULTS, pixels_per_mm=10)
    cfg = _cfgobj(cfgd)
    W, H = O.canvas_size(cfgd)
    polys = []
    for t in range(24):
        m = int(rng.integers(150, 6000))
        hi = (2, 3, 4, 12)[t % 4]                       # unit / diagonal steps, small mixed steps, and steps up to 17 px (many binades, many distinct lengths)
        d = rng.integers(-hi + 1, hi, (m, 2))
        p = np.cumsum(d, axis=0); p -= p.min(axis=0); p = p % np.array([W - 200, H - 200]) + 100      # wraps make a few long jumps as well
        polys.append(p.astype(np.int32).reshape(-1, 1, 2))
    base = np.cumsum(rng.integers(-1, 2, (300, 2)), axis=0) + np.array([W // 2, H // 2])
    polys.append(np.concatenate([base, base[::-1]] * 40).astype(np.int32).reshape(-1, 1, 2))         # a bounce tail: the same cycle again and again
    want_l, want_t = O.stage08_layer(polys, O.derived08(cfgd))
    for chain in [False, True]:
        if chain:
            monkeypatch.setenv("ORIP_CUM_CHAIN", "1")
        else:
            monkeypatch.delenv("ORIP_CUM_CHAIN", raising=False)
        got_l, got_t = S.dedup_layer(polys, cfg, dev)
        assert got_t == want_t, chain
        assert same_polys(got_l, want_l), (chain, len(got_l), len(want_l))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_full_chain_image_to_ops_matches_reference(dev, tag):
    """Resident path 02 -> 12 from the image: final ops identical to the reference chain's ops.pkl."""
    from orip import stages as S
    G = load(f"golden_e2e_{tag}.npz")
    cfgd = json.loads(bytes(G["cfg_json"]).decode()); cfg = _cfgobj(cfgd)
    ops = S.run_path(G["img"], cfg, dev)
    for n in cfgd["color_names"]:
        got = [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops[n]]
        assert same_polys(got, unflat(G, f"ops_{n}")), n


@pytest.mark.parametrize("case", [(256, 256, 4, 6), (384, 512, 8, 10)])
def test_full_chain_vs_oracle(dev, case):
    from orip import stages as S
    from orip.synth import synth_image, layer_names
    H, W, K, ppm = case
    img = synth_image(H, W, K, seed=77)
    cfgd = dict(O.DEFAULTS, color_names=layer_names(K), pixels_per_mm=ppm)
    cfg = _cfgobj(cfgd)
    want = O.run_pipeline(img, cfgd)
    ops = S.run_path(img, cfg, dev)
    for n in cfgd["color_names"]:
        assert len(ops[n]) == len(want["ops"][n]), n
        for a, b in zip(ops[n], want["ops"][n]):
            assert a["type"] == b["type"]
            if a["type"] == "line":
                assert np.array_equal(a["points"], b["points"]), n
            else:
                assert (a["x"], a["y"]) == (b["x"], b["y"]), n
    dg, tg = O.path_length(ops); dw, tw = O.path_length(want["ops"])
    assert abs(dg + tg - dw - tw) <= 1e-3 * (dw + tw)      # north_star: plotted path length within 1e-3 relative


# ---------------------------------------------------------------- edge cases (inputs in tests/edge_cases.py)
import edge_cases as E

_CFG08 = dict(O.DEFAULTS, pixels_per_mm=6)


@pytest.mark.parametrize("name", sorted(E.cases08(*O.canvas_size(_CFG08))))
def test_stage08_edge_cases(dev, name):
    """Empty / degenerate / duplicated / off-canvas inputs of the intra-layer dedup."""
    from orip import stages as S
    polys = E.cases08(*O.canvas_size(_CFG08))[name]
    want_l, want_t = O.stage08_layer(polys, O.derived08(_CFG08))
    got_l, got_t = S.dedup_layer(polys, _cfgobj(_CFG08), dev)
    assert got_t == want_t, name
    assert same_polys(got_l, want_l), (name, len(got_l), len(want_l))


@pytest.mark.parametrize("name", sorted(E.cases07()))
def test_stage07_edge_cases(dev, name):
    """One polyline, identical polylines (every distance ties), counts around the 64-polyline switch to the grid search."""
    from orip import stages as S
    polys = E.cases07()[name]
    assert same_polys(S.sort_contours(polys, dev), O.sort07(polys)), name


def test_stage10_edge_cases(dev):
    """A layer without lines, a layer with taps only, a layer repeating the darker layer's lines (all of it is cut away)."""
    from orip import stages as S
    cfgd = dict(O.DEFAULTS, pixels_per_mm=6, color_names=["layer_dark", "layer_mid", "x_extra", "layer_light"])
    rng = np.random.default_rng(12)
    base = _rand_polys(rng, 30, lo=2, hi=25, span=1100, step=40, closed_p=0.0)
    intra = {"layer_dark": (base, _taps(rng.integers(0, 1200, (10, 2)))),
             "layer_mid": ([], []),
             "x_extra": ([], _taps(rng.integers(0, 1200, (15, 2)))),
             "layer_light": ([p.copy() for p in base] + _rand_polys(rng, 5, lo=2, hi=25, span=1100, step=40, closed_p=0.0), [])}
    want = O.stage10(intra, cfgd)
    got = S.dedup_cross(intra, _cfgobj(cfgd), dev)
    for n in cfgd["color_names"]:
        assert got[n][1] == want[n][1], n
        assert same_polys(got[n][0], want[n][0]), n
