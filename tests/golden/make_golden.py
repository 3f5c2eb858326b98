#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the golden fixtures in this directory.

Runs ONLY in the build container (needs /root/reference).  It loads the reference's own stage modules
(read-only, from /root/reference/image_processor) with tests/golden/cv2_standin.py registered as `cv2`
(OpenCV is not installed) and records inputs + outputs of
  (a) the reference's pure numpy/Python functions  -> golden_pure.npz   ("pinned by the reference")
  (b) the reference's stage drivers 04 -> 12 run on a small image, with the cv2 stand-in supplying the
      OpenCV primitives                              -> golden_e2e_*.npz ("control flow pinned, cv2 unpinned")
Nothing from the reference is copied: fixtures hold arrays only.
Usage:  python tests/golden/make_golden.py
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
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/image_processor"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))

import cv2_standin  # noqa: E402

sys.modules["cv2"] = cv2_standin
sys.path.insert(0, REF)  # for `from config import ...`


def load_ref(fname: str):
    spec = importlib.util.spec_from_file_location("ref_" + fname[:2], os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat(polys):
    off = np.zeros(len(polys) + 1, np.int64)
    arrs = [np.asarray(p).reshape(-1, 2) for p in polys]
    for i, a in enumerate(arrs):
        off[i + 1] = off[i] + len(a)
    pts = np.concatenate(arrs, 0).astype(np.int32) if arrs else np.zeros((0, 2), np.int32)
    return off, pts


def put(d, name, polys):
    d[name + "_off"], d[name + "_pts"] = flat(polys)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def random_blobs(rng, h, w, n, rmax):
    img = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        ry, rx = rng.integers(1, rmax), rng.integers(1, rmax)
        img[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
    return img


def make_pure():
    m04, m05, m07, m08, m10, m12 = (load_ref(f) for f in (
        "04_find_contours.py", "05_scale_vectors.py", "07_sort_contours.py", "08_dedup_layer_basic.py",
        "10_dedup_cross_basic.py", "12_optimize_plot_order.py"))
    rng = np.random.default_rng(7)
    g = {}
    # --- 04.thinning_zhangsuen / trace_centerlines on random blob outlines
    for t in range(4):
        h, w = [(24, 31), (40, 40), (33, 64), (64, 48)][t]
        blobs = random_blobs(rng, h, w, 3 + 2 * t, 4 + 3 * t)
        from scipy import ndimage as ndi
        edges = (blobs > 0) & ~ndi.binary_erosion(blobs > 0, np.ones((3, 3)), iterations=1 + t % 2)
        edges = (edges * 255).astype(np.uint8)
        if t == 3:
            edges[rng.random(edges.shape) < 0.04] = 255
        skel = quiet(m04.thinning_zhangsuen, edges.copy(), "g")
        paths = quiet(m04.trace_centerlines, skel, "g")
        g[f"thin{t}_in"] = edges; g[f"thin{t}_out"] = skel
        put(g, f"trace{t}", paths)
    # --- 08._zhang_suen_fast
    for t in range(3):
        img = random_blobs(rng, 30 + 5 * t, 37, 4, 7)
        g[f"zs{t}_in"] = img; g[f"zs{t}_out"] = m08._zhang_suen_fast(img.copy())
    # --- 05._scale_one
    polys = [rng.integers(0, 4096, (n, 1, 2)).astype(np.int32) for n in (1, 2, 7, 300)]
    for i, (sx, dx) in enumerate([(1.85546875, 400.0), (0.927734375, 400.0), (7600 / 1000, 400.0), (2.7050781, 123.0)]):
        put(g, f"scale{i}_in", polys); g[f"scale{i}_prm"] = np.array([sx, sx, dx, dx])
        put(g, f"scale{i}_out", m05._scale_one(polys, sx, sx, dx, dx))
    # --- 08._resample_arclen
    for i in range(6):
        n = [2, 3, 10, 50, 400, 5][i]
        p = np.cumsum(rng.integers(-9, 10, (n, 2)), axis=0).astype(np.float32) + 100
        if i == 5:
            p[-1] = p[0]
        step = [8.0, 6.0, 8.0, 6.0, 8.0, 8.0][i]
        out = m08._resample_arclen(p, step)
        g[f"resample{i}_in"] = p; g[f"resample{i}_step"] = np.array(step); g[f"resample{i}_out"] = np.asarray(out, np.float64)
        g[f"resample{i}_pass"] = np.array(out.dtype == np.float32)
    # --- split on long jumps (both variants)
    for i in range(4):
        n = [2, 6, 40, 200][i]
        p = np.cumsum(rng.integers(-60, 61, (n, 2)), axis=0).astype(np.int32) + 3000
        g[f"jump{i}_in"] = p
        put(g, f"jump{i}_out08", m08._split_on_long_jumps(p.reshape(-1, 1, 2), 80.0))
        put(g, f"jump{i}_out10", m10._split_on_long_jumps(p.reshape(-1, 1, 2), 80.0))
    # --- reorder (08 / 10 / 07)
    for i in range(3):
        polys = []
        for _ in range([1, 12, 80][i]):
            n = int(rng.integers(2, 30))
            p = (np.cumsum(rng.integers(-15, 16, (n, 2)), axis=0) + rng.integers(0, 8000, 2)).astype(np.int32)
            if rng.random() < 0.3 and n > 3:
                p[-1] = p[0]
            polys.append(p.reshape(-1, 1, 2))
        put(g, f"reorder{i}_in", polys)
        put(g, f"reorder{i}_out08", m08._reorder_only(polys))
        put(g, f"reorder{i}_out10", m10._reorder_for_travel(polys))
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "contours_scaled.pkl"), "wb") as f:
                pickle.dump(polys, f)
            quiet(m07.reorder_one_color, td)
            with open(os.path.join(td, "contours_sorted.pkl"), "rb") as f:
                put(g, f"reorder{i}_out07", pickle.load(f))
    # --- cluster by overlap
    bb = rng.integers(0, 500, (60, 2)); bb = np.concatenate([bb, bb + rng.integers(1, 60, (60, 2))], 1)
    g["cluster_in"] = bb.astype(np.int32)
    groups = m08._cluster_by_overlap([tuple(int(v) for v in b) for b in bb])
    gid = np.zeros(60, np.int32)
    for k, grp in enumerate(groups):
        gid[grp] = k
    g["cluster_out"] = gid
    # --- bfs / component best path on thinned blobs
    for i in range(3):
        img = m08._zhang_suen_fast(random_blobs(rng, 40, 50, 5, 9))
        n, lab = cv2_standin.connectedComponents((img > 0).astype(np.uint8))
        sizes = [(lab == c).sum() for c in range(1, n)]
        c = 1 + int(np.argmax(sizes))
        comp = ((lab == c) * 255).astype(np.uint8)
        ys, xs = np.nonzero(comp)
        a = (int(ys[0]), int(xs[0])); b = (int(ys[-1]), int(xs[-1]))
        g[f"bfs{i}_img"] = comp; g[f"bfs{i}_ab"] = np.array([a, b], np.int32)
        g[f"bfs{i}_path"] = np.array(m08._bfs_path((comp > 0).astype(np.uint8), a, b), np.int32).reshape(-1, 2)
        g[f"bfs{i}_best_none"] = np.array(m08._component_best_path(comp, None, None, 4), np.int32).reshape(-1, 2)
        g[f"bfs{i}_best_ab"] = np.array(m08._component_best_path(comp, a, b, 4), np.int32).reshape(-1, 2)
    # --- 10._cut_poly_against_mask
    forb = np.zeros((300, 400), np.uint8); forb[:, 150:180] = 255; forb[100:130, :] = 255
    for i in range(3):
        p = np.cumsum(rng.integers(-40, 41, (12, 2)), axis=0).astype(np.int32) + np.array([200, 150])
        g[f"cut{i}_in"] = p
        put(g, f"cut{i}_out", m10._cut_poly_against_mask(p.reshape(-1, 1, 2), forb, 1.0))
    g["cut_mask"] = forb
    # --- 12._build_ops_for_layer
    for i in range(3):
        lines = []
        for _ in range([0, 9, 60][i]):
            n = int(rng.integers(2, 12))
            lines.append((np.cumsum(rng.integers(-25, 26, (n, 2)), axis=0) + rng.integers(0, 3000, 2)).astype(np.int32).reshape(-1, 1, 2))
        taps = [(int(x), int(y)) for x, y in rng.integers(0, 3000, ([5, 7, 40][i], 2))]
        ops = m12._build_ops_for_layer(lines, taps, 80.0)
        put(g, f"ops{i}_lines", lines); g[f"ops{i}_taps"] = np.array(taps, np.int32).reshape(-1, 2)
        kinds = np.array([0 if o["type"] == "line" else 1 for o in ops], np.int32)
        put(g, f"ops{i}_out", [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops])
        g[f"ops{i}_kinds"] = kinds
    # --- 08 stage A on a small canvas (cv2.line = stand-in capsule)
    for i in range(3):
        Wc, Hc = 700, 500
        mask = np.zeros((Hc, Wc), np.uint8)
        polys = []
        for _ in range(4):
            n = int(rng.integers(5, 60))
            polys.append((np.cumsum(rng.integers(-30, 31, (n, 2)), axis=0) + rng.integers(100, 400, 2)).astype(np.int32).reshape(-1, 1, 2))
        put(g, f"vdraw{i}_in", polys)
        for j, p in enumerate(polys):
            segs = m08._virtual_draw_split_with_mask_and_tail(p, 8.0, 120.0, mask, 255, 18.0, 18.0, 36)
            put(g, f"vdraw{i}_out{j}", segs)
        g[f"vdraw{i}_mask"] = np.packbits(mask > 0)
    np.savez_compressed(os.path.join(HERE, "golden_pure.npz"), **g)
    print("golden_pure.npz:", len(g), "arrays")


def make_e2e(tag: str, H: int, W: int, K: int, cfg_over: dict, seed: int):
    """Reference stage drivers 04,05,07,08,10,12 on oracle-produced edges (02/03 are cv2-bound)."""
    from orip.synth import synth_image, layer_names
    from oracle import oracle as O

    names = layer_names(K)
    img = synth_image(H, W, K, seed=seed, sigma=max(2.0, H / 48.0))
    cfg = dict(color_names=names)
    cfg.update(cfg_over)
    r = O.run_pipeline(img, cfg, upto=3)
    g = {"img": img, "cfg_json": np.frombuffer(json.dumps(cfg).encode(), np.uint8)}
    with tempfile.TemporaryDirectory() as td:
        full = dict(cfg); full["output_dir"] = td
        with open(os.path.join(td, "config.json"), "w") as f:
            json.dump(full, f)
        os.environ["CONFIG_PATH"] = os.path.join(td, "config.json")
        cv2_standin.imwrite(os.path.join(td, "resized.png"), img)
        for n in names:
            os.makedirs(os.path.join(td, n), exist_ok=True)
            cv2_standin.imwrite(os.path.join(td, n, "edges.png"), r["edges"][n])
            g[f"edges_{n}"] = np.packbits(r["edges"][n] > 0)
        mods = {f[:2]: load_ref(f) for f in ("04_find_contours.py", "05_scale_vectors.py", "07_sort_contours.py",
                                              "08_dedup_layer_basic.py", "10_dedup_cross_basic.py", "12_optimize_plot_order.py")}
        from config import load_config
        quiet(lambda: mods["04"].vectorize_all(quiet(load_config)))
        quiet(mods["05"].main); quiet(mods["07"].main); quiet(mods["08"].main); quiet(mods["10"].main); quiet(mods["12"].main)
        for n in names:
            def pk(fn):
                p = os.path.join(td, n, fn)
                if not os.path.exists(p):
                    return []
                with open(p, "rb") as f:
                    return pickle.load(f)
            put(g, f"contours_{n}", pk("contours.pkl")); put(g, f"scaled_{n}", pk("contours_scaled.pkl"))
            put(g, f"sorted_{n}", pk("contours_sorted.pkl")); put(g, f"lines_intra_{n}", pk("lines_intra.pkl"))
            g[f"taps_intra_{n}"] = np.array(pk("taps_intra.pkl"), np.int32).reshape(-1, 2)
            put(g, f"lines_cross_{n}", pk("lines_cross.pkl")); g[f"taps_cross_{n}"] = np.array(pk("taps_cross.pkl"), np.int32).reshape(-1, 2)
            ops = pk("ops.pkl")
            g[f"ops_kinds_{n}"] = np.array([0 if o["type"] == "line" else 1 for o in ops], np.int32)
            put(g, f"ops_{n}", [o["points"] if o["type"] == "line" else np.array([[o["x"], o["y"]]]) for o in ops])
        with open(os.path.join(td, "vector_manifest.json")) as f:
            man = json.load(f)
        for L in man["layers"]:
            L["file"] = os.path.basename(os.path.dirname(L["file"])) + "/ops.pkl"
        g["manifest_json"] = np.frombuffer(json.dumps(man).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, f"golden_e2e_{tag}.npz"), **g)
    print(f"golden_e2e_{tag}.npz written:", {n: (len(g[f'contours_{n}_off']) - 1, len(g[f'lines_intra_{n}_off']) - 1,
                                                 len(g[f'taps_cross_{n}']), len(g[f'ops_kinds_{n}'])) for n in names})


if __name__ == "__main__":
    # Small canvases: the reference's stage 08-B allocates ROI-sized (h,w,2) arrays per BFS, so the default
    # 8400x11880 canvas takes hours in Python even for a 96-px image.
    which = sys.argv[1:] or ["pure", "a", "b"]
    if "pure" in which:
        make_pure()
    if "a" in which:
        make_e2e("a", 96, 96, 4, {"pixels_per_mm": 4}, seed=11)             # canvas 840x1188, scale ~7.9
    if "b" in which:
        make_e2e("b", 120, 88, 4, {"pixels_per_mm": 6, "edge_kernel_size": 5}, seed=12)   # canvas 1260x1782
