#!/usr/bin/env python3
"""tests/golden/make_golden_stream.py -- golden vectors for the plotter stream (13_build_stream.py + shared/omnirevolve_plotter_stream_creator_helper.py).

Runs ONLY in the build container (needs /root/reference).  Both reference files are pure Python (the stage imports cv2 only for a fall-back it
does not take: the stand-in of this directory is registered as `cv2`), so every byte recorded here is the reference's own output:
  * bres*      : bresenham_dir_codes on seeded random and degenerate segments
  * travel*    : bytes of travel_ramped for single moves (short / long / odd lengths), default helper Config
  * poly*      : bytes of emit_polyline for random polylines with sharp corners, short and long edges, repeated points, both ramp profiles
  * e2e_{a,b}* : plot_stream.bin / plot_stream.json of 13_build_stream.main() on the ops of golden_e2e_{a,b}.npz, plus a colour-remap variant
Nothing from the reference is copied: the fixture holds arrays only.   Usage: python tests/golden/make_golden_stream.py
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import json
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/image_processor"
sys.path.insert(0, HERE)
import cv2_standin  # noqa: E402

sys.modules["cv2"] = cv2_standin
sys.path.insert(0, REF)
sys.path.insert(0, "/root/reference/shared")
import omnirevolve_plotter_stream_creator_helper as RH  # noqa: E402


def load_ref(fname):
    spec = importlib.util.spec_from_file_location("ref_" + fname[:2], os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def main():
    rng = np.random.default_rng(13)
    g = {}
    # ---- bresenham_dir_codes
    segs = [(0, 0, 0, 0), (5, 5, 5, 9), (5, 9, 5, 5), (3, 3, 9, 3), (9, 3, 3, 3), (0, 0, 7, 7), (7, 7, 0, 0), (0, 7, 7, 0), (0, 0, 10, 5), (0, 0, 5, 10), (0, 0, 9, 4), (10, 4, 0, 0)]
    segs += [tuple(int(v) for v in rng.integers(0, 60, 4)) for _ in range(300)] + [tuple(int(v) for v in rng.integers(0, 12000, 4)) for _ in range(40)]
    g["bres_segs"] = np.array(segs, np.int32)
    codes = [np.array(RH.bresenham_dir_codes(*s), np.uint8) for s in segs]
    g["bres_off"] = np.concatenate([[0], np.cumsum([len(c) for c in codes])]).astype(np.int64)
    g["bres_codes"] = np.concatenate(codes)
    # ---- travel_ramped, single moves
    moves = [(0, 0, 1, 0), (0, 0, 2, 1), (0, 0, 3, 3), (10, 10, 10, 490), (0, 0, 479, 0), (0, 0, 480, 7), (0, 0, 481, 100), (5, 5, 2000, 900), (3000, 100, 20, 4000), (0, 0, 0, 7)]
    moves += [tuple(int(v) for v in rng.integers(0, 3000, 4)) for _ in range(12)]
    g["travel_moves"] = np.array(moves, np.int32)
    outs = []
    for m in moves:
        w = RH.StreamWriter(); RH.travel_ramped(w, *m, RH.Config()); outs.append(np.frombuffer(bytes(w.out), np.uint8))
    g["travel_off"] = np.concatenate([[0], np.cumsum([len(o) for o in outs])]).astype(np.int64); g["travel_bytes"] = np.concatenate(outs)
    # ---- emit_polyline
    polys = []
    for i in range(24):
        n = int(rng.integers(2, 30))
        scale = [4, 40, 400, 1500][i % 4]
        p = np.cumsum(rng.integers(-scale, scale + 1, (n, 2)), axis=0) + 4000
        if i % 5 == 0 and n > 4:
            p[3] = p[2]                                  # repeated point: a segment without steps between two corners
        if i % 3 == 0 and n > 5:
            p[4] = p[2]                                  # reversal: a 0-degree corner
        polys.append(np.clip(p, 0, 12000).astype(np.int64))
    polys.append(np.array([[0, 0], [2000, 0], [2000, 1], [0, 1]], np.int64))       # long edges around two sharp corners: full windows
    polys.append(np.array([[0, 0], [100, 0], [0, 0], [100, 0]], np.int64))
    offp = np.concatenate([[0], np.cumsum([len(p) for p in polys])]).astype(np.int64)
    g["poly_off"] = offp; g["poly_pts"] = np.concatenate(polys).astype(np.int32)
    for profile in ("triangle", "scurve"):
        outs = []
        for p in polys:
            w = RH.StreamWriter(); RH.emit_polyline(w, RH.Config(profile=profile, div_start=25, corner_div=30, corner_window_steps=800), [tuple(int(v) for v in q) for q in p])
            outs.append(np.frombuffer(bytes(w.out), np.uint8))
        g[f"poly_{profile}_off"] = np.concatenate([[0], np.cumsum([len(o) for o in outs])]).astype(np.int64)
        g[f"poly_{profile}_bytes"] = np.concatenate(outs) if outs else np.zeros(0, np.uint8)
    # ---- 13_build_stream.main() on the ops of the e2e fixtures
    m13 = load_ref("13_build_stream.py")
    for tag in ("a", "b"):
        G = np.load(os.path.join(HERE, f"golden_e2e_{tag}.npz"))
        cfg = json.loads(bytes(G["cfg_json"]).decode())
        man = json.loads(bytes(G["manifest_json"]).decode())
        for variant, extra, env in (("", {}, {}), ("_remap", {"stream_color_by_order": [3, 0, 1, 2], "stream_color_by_name": {"layer_mid": 5}}, {}),
                                    ("_env", {}, {"STREAM_FORCE_COLOR_INDEX": "6"})):
            with tempfile.TemporaryDirectory() as td:
                full = dict(cfg); full.update(extra); full["output_dir"] = td
                with open(os.path.join(td, "config.json"), "w") as f:
                    json.dump(full, f)
                for n in cfg["color_names"]:
                    os.makedirs(os.path.join(td, n), exist_ok=True)
                    kinds = G[f"ops_kinds_{n}"]; off = G[f"ops_{n}_off"]; pts = G[f"ops_{n}_pts"]
                    ops = []
                    for i, k in enumerate(kinds):
                        q = pts[off[i]:off[i + 1]]
                        ops.append({"type": "line", "points": q.astype(np.float32)} if k == 0 else {"type": "tap", "x": int(q[0, 0]), "y": int(q[0, 1])})
                    with open(os.path.join(td, n, "ops.pkl"), "wb") as f:
                        pickle.dump(ops, f)
                with open(os.path.join(td, "vector_manifest.json"), "w") as f:
                    json.dump(man, f)
                os.environ["CONFIG_PATH"] = os.path.join(td, "config.json")
                old = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        m13.main()
                finally:
                    for k, v in old.items():
                        if v is None: os.environ.pop(k, None)
                        else: os.environ[k] = v
                g[f"e2e_{tag}{variant}_bin"] = np.frombuffer(open(os.path.join(td, "plot_stream.bin"), "rb").read(), np.uint8)
                g[f"e2e_{tag}{variant}_json"] = np.frombuffer(open(os.path.join(td, "plot_stream.json"), "rb").read(), np.uint8)
                g[f"e2e_{tag}{variant}_cfg"] = np.frombuffer(json.dumps(extra).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "golden_stream.npz"), **g)
    print("golden_stream.npz:", len(g), "arrays;", {k: int(v.size) for k, v in g.items() if k.endswith("_bin")})


if __name__ == "__main__":
    main()
