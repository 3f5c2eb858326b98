# 12_optimize_plot_order.py -- drop-in: <layer>/lines_cross.pkl + taps_cross.pkl -> <layer>/ops.pkl + vector_manifest.json
import json
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config, canvas_size_px


def main():
    cfg = load_config()
    out = cfg.output_dir
    W, H = canvas_size_px(cfg)
    layers = []
    for name in cfg.color_names:
        layer_dir = os.path.join(out, name)
        os.makedirs(layer_dir, exist_ok=True)
        pL, pT = os.path.join(layer_dir, "lines_cross.pkl"), os.path.join(layer_dir, "taps_cross.pkl")
        if not _io.exists(pL) or not _io.exists(pT):
            raise SystemExit(f"Missing cross artifacts in {layer_dir}")
        lines = _io.load_pickle(pL)
        taps = []
        for it in _io.load_pickle(pT):
            a = np.asarray(it).reshape(-1)
            if a.size >= 2:
                taps.append((int(a[0]), int(a[1])))
        ops = S.plot_order(lines, taps, cfg)
        p_ops = os.path.join(layer_dir, "ops.pkl")
        _io.save_pickle(p_ops, ops)
        layers.append({"name": name, "color_name": name, "color_index": S.color_index12(name), "file": os.path.relpath(p_ops, out), "count_ops": len(ops)})
        nL = sum(1 for o in ops if o["type"] == "line")
        print(f"[plot-opt] {name}: ops={len(ops)} (lines={nL}, taps={len(ops) - nL})")
    manifest = {"image_size": [W, H], "layers": layers, "coords": "pixel_top_left"}
    with open(os.path.join(out, "vector_manifest.json"), "w", encoding="utf-8") as f:
        json.dump(manifest, f, ensure_ascii=False, indent=2)
    print(f"[plot-opt] manifest saved: {os.path.join(out, 'vector_manifest.json')}")


if __name__ == "__main__":
    main()
