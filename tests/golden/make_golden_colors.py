#!/usr/bin/env python3
"""tests/golden/make_golden_colors.py -- golden vectors for process_colors.py, the pure-numpy parts.

Runs ONLY in the build container (needs /root/reference).  process_colors.py imports cv2 for imread / kmeans / imwrite; the stand-in of this
directory is registered as `cv2`, and only functions that never touch it are called, so every array recorded here is the reference's own output:
  * assign_*   : assign_labels (:69-77, int16 arithmetic that wraps for differences above 181) on seeded images and palettes, K = 2..16
  * sub_*      : the pixel subsample of kmeans_palette (:35-39, RandomState(seed).choice without replacement) for image sizes above / below the limit
  * names_*    : default_color_names (:80-82)
  * pal_*      : palette_from_json (:49-66) on both layouts it accepts (the second one ends in a NameError inside the reference, recorded as such)
The k-means itself (cv2.kmeans on RGB float32) is OpenCV: parity unpinned, as for stage 02.
Nothing from the reference is copied: the fixture holds arrays only.   Usage: python tests/golden/make_golden_colors.py
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/image_processor"
sys.path.insert(0, HERE)
import cv2_standin  # noqa: E402

sys.modules["cv2"] = cv2_standin
sys.path.insert(0, REF)


def load_ref(fname):
    spec = importlib.util.spec_from_file_location("ref_pc", os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod


def main():
    PC = load_ref("process_colors.py")
    rng = np.random.default_rng(41)
    g = {}
    cases = [(37, 53, 2), (64, 64, 4), (50, 81, 8), (33, 47, 16), (20, 20, 3)]
    for n, (h, w, k) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pal = rng.integers(0, 256, (k, 3), dtype=np.uint8)
        if n == 1:
            pal[:] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 255]]          # differences of 255: the int16 products wrap
            img[:8] = 0; img[8:16] = 255
        if n == 4:
            pal[2] = pal[0]                                                               # a duplicated colour: the first index wins
        g[f"assign_img_{n}"] = img; g[f"assign_pal_{n}"] = pal
        g[f"assign_lab_{n}"] = PC.assign_labels(img, pal)
    # the subsample of kmeans_palette: same statements on the same RandomState (the function itself goes on into cv2.kmeans)
    for n, (N, samples) in enumerate([(300 * 400, 200000), (700 * 900, 200000), (1000, 100)]):
        rs = np.random.RandomState(1)
        idx = rs.choice(N, size=samples, replace=False) if N > samples else np.arange(N)
        g[f"sub_N_{n}"] = np.array([N, samples], np.int64); g[f"sub_idx_{n}"] = np.asarray(idx, np.int64)
    g["names_6"] = np.array(PC.default_color_names(6)); g["names_2"] = np.array(PC.default_color_names(2))
    with tempfile.TemporaryDirectory() as td:
        a = {"recommended_colors": [{"name": "sky", "rgb": [10, 20, 200], "position": 2}, {"name": "ink", "rgb": [5, 5, 5], "position": 1}, {"rgb": [250, 240, 10]}]}
        b = {"palette": [{"rgb": [1, 2, 3]}, {"name": "x", "rgb": [200, 100, 50]}]}
        for tag, d in (("a", a), ("b", b)):
            p = os.path.join(td, tag + ".json")
            with open(p, "w") as f:
                json.dump(d, f)
            g[f"pal_json_{tag}"] = np.array(json.dumps(d))
            try:
                rgb, names = PC.palette_from_json(p)
                g[f"pal_rgb_{tag}"] = rgb; g[f"pal_names_{tag}"] = np.array(names)
            except Exception as e:       # the "palette" layout: the reference's comprehension at :63 names a variable of another scope
                g[f"pal_raises_{tag}"] = np.array(type(e).__name__)
    out = os.path.join(HERE, "golden_colors.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, len(g), "arrays")


if __name__ == "__main__":
    main()
