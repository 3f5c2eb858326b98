# 03_edge_detect.py -- drop-in: <layer>/mask.png -> <layer>/edges.png (+ edges_composite.png by-product).
# All layers go through one batched GPU call instead of the reference's process pool over layers (03:42-48).
import json
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def save_edges_composite(cfg, edges):
    names = list(cfg.color_names)
    first = next(iter(edges.values()))
    canvas = np.full(first.shape + (3,), 255, np.uint8)
    palette = {}
    pp = os.path.join(cfg.output_dir, "palette_by_name.json")
    if os.path.exists(pp):
        try:
            with open(pp, "r", encoding="utf-8") as f:
                palette = json.load(f)
        except Exception:
            palette = {}
    for i, name in enumerate(names):
        if name not in edges:
            continue
        bgr = palette[name]["bgr"] if name in palette and "bgr" in palette[name] else cfg.colors[i]   # IndexError if K > len(colors), as 03:85-91
        canvas[edges[name] > 0] = tuple(int(v) for v in bgr)
    out = os.path.join(cfg.output_dir, "edges_composite.png")
    _io.write_png(out, canvas)
    print(f"Edges composite saved: {out}")


def main():
    cfg = load_config()
    masks = {}
    for name in cfg.color_names:
        p = os.path.join(cfg.output_dir, name, "mask.png")
        if not _io.exists(p):
            raise FileNotFoundError(f"Mask not found: {p}")
        m = _io.read_gray(p)
        if m is None:
            raise ValueError(f"Failed to load mask image: {p}")
        masks[name] = m
    edges = S.detect_edges(masks, cfg)
    for name, e in edges.items():
        _io.write_png(os.path.join(cfg.output_dir, name, "edges.png"), e)
        print(f"Edges extracted: {name} | nz={int(np.count_nonzero(e))}")
    save_edges_composite(cfg, edges)


if __name__ == "__main__":
    main()
