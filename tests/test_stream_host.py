"""Host logic of the plotter stream (orip/stream.py: speed plans, corner flags, byte assembly, colour remap) against bytes produced by the
REFERENCE's own 13_build_stream.py / stream helper (tests/golden/golden_stream.npz, make_golden_stream.py).  CPU only: the direction codes come
from the numpy test double (tests/stream_double.py), which the first test pins to the reference's bresenham_dir_codes."""
import json

import numpy as np
import pytest

from util import load
from stream_double import codes_numpy

G = load("golden_stream.npz")


def _st():
    from orip import stream as ST
    return ST


def test_double_matches_reference_bresenham():
    off, codes = codes_numpy(G["bres_segs"])
    assert np.array_equal(off, G["bres_off"]) and np.array_equal(codes, G["bres_codes"])


def _one_move(ST, move, plan, sc):
    P = ST._Plot(); P.move(*move, plan)
    off, codes = codes_numpy(np.array([move]))
    return np.frombuffer(ST.assemble(P, off, codes, sc), np.uint8)


def test_travel_ramps_match_reference():
    ST = _st(); sc = ST.StreamConfig()
    for i, m in enumerate(G["travel_moves"]):
        want = G["travel_bytes"][G["travel_off"][i]:G["travel_off"][i + 1]]
        got = _one_move(ST, tuple(int(v) for v in m), lambda n: ST.plan_travel(n, sc), sc)
        assert np.array_equal(got[:len(want)], want) and got[len(want)] == 0x3F, (i, m)


@pytest.mark.parametrize("profile", ["triangle", "scurve"])
def test_polylines_with_corners_match_reference(profile):
    ST = _st(); sc = ST.StreamConfig(profile=profile, div_start=25, corner_div=30, corner_window_steps=800)
    off, pts = G["poly_off"], G["poly_pts"]
    for i in range(len(off) - 1):
        pl = pts[off[i]:off[i + 1]].astype(np.int64)
        want = G[f"poly_{profile}_bytes"][G[f"poly_{profile}_off"][i]:G[f"poly_{profile}_off"][i + 1]]
        P = ST._Plot()
        sin, sout = ST.corner_flags(pl, sc.corner_deg)
        for j in range(len(pl) - 1):
            P.move(pl[j, 0], pl[j, 1], pl[j + 1, 0], pl[j + 1, 1], None if not (sin[j] or sout[j]) else (lambda n, a=bool(sin[j]), b=bool(sout[j]): ST.plan_segment(n, sc, a, b)))
        o, c = codes_numpy(np.asarray(P.moves, np.int64).reshape(-1, 4))
        got = np.frombuffer(ST.assemble(P, o, c, sc), np.uint8)
        assert np.array_equal(got[:len(want)], want) and got[len(want)] == 0x3F, (profile, i)


def _layers_from_e2e(tag):
    E = load(f"golden_e2e_{tag}.npz")
    cfg = json.loads(bytes(E["cfg_json"]).decode()); man = json.loads(bytes(E["manifest_json"]).decode())
    layers = []
    for L in man["layers"]:
        n = L.get("color_name", L.get("name"))
        kinds, off, pts = E[f"ops_kinds_{n}"], E[f"ops_{n}_off"], E[f"ops_{n}_pts"]
        ops = [{"type": "line", "points": pts[off[i]:off[i + 1]].astype(np.float32)} if k == 0 else {"type": "tap", "x": int(pts[off[i], 0]), "y": int(pts[off[i], 1])}
               for i, k in enumerate(kinds)]
        layers.append((str(n), int(L.get("color_index", 0)), ops))
    return cfg, layers


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("variant", ["", "_remap", "_env"])
def test_whole_stream_matches_reference(tag, variant, monkeypatch):
    ST = _st()
    from orip.config import Config, canvas_size_px
    cfgd, layers = _layers_from_e2e(tag)
    extra = json.loads(bytes(G[f"e2e_{tag}{variant}_cfg"]).decode())
    cfg = Config()
    for k, v in {**cfgd, **extra}.items():
        if k in Config.__dataclass_fields__:
            setattr(cfg, k, v)
    if variant == "_env":
        monkeypatch.setenv("STREAM_FORCE_COLOR_INDEX", "6")
    W, H = canvas_size_px(cfg)
    data, meta = ST.build_stream(layers, W, H, ST.stream_config_from_pipeline(cfg), codes_fn=codes_numpy, color_maps=ST.load_color_maps(cfg))
    want = bytes(G[f"e2e_{tag}{variant}_bin"])
    assert data == want
    wj = json.loads(bytes(G[f"e2e_{tag}{variant}_json"]).decode())
    assert meta["lines"] == wj["lines"] and meta["taps"] == wj["taps"] and meta["bytes"] == wj["bytes"] and wj["target_steps"] == {"width": W, "height": H}
    assert len(data) % 1024 == 0
