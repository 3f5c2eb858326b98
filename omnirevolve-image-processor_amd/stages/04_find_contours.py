# 04_find_contours.py -- drop-in: <layer>/edges.png -> <layer>/contours.pkl (list of int32 (N,1,2), source px)
import os

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    edges = {}
    for name in cfg.color_names:
        p = os.path.join(cfg.output_dir, name, "edges.png")
        e = _io.read_gray(p)
        if e is None:
            raise FileNotFoundError(f"Edges not found: {p}")
        edges[name] = e
    contours = S.find_contours(edges, cfg)
    for name, paths in contours.items():
        out = os.path.join(cfg.output_dir, name, "contours.pkl")
        _io.save_pickle(out, _io.polys_out(paths))
        print(f"[{name}] Saved contours: {len(paths)} -> {out}", flush=True)


if __name__ == "__main__":
    main()
