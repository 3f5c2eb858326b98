# 02_color_extract.py -- drop-in for the reference stage of the same name (k-means mode): resized.png -> <layer>/mask.png
# + palette_by_name.json.  Compute: liborip.so on the GPU (Lab, k-means fit, assignment, 3x3 open/close).
import json
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    os.makedirs(cfg.output_dir, exist_ok=True)
    src = os.path.join(cfg.output_dir, "resized.png")
    img = _io.read_bgr(src)
    if img is None:
        raise RuntimeError(f"Cannot read resized image: {src}")
    if str(getattr(cfg, "extraction_mode", "kmeans")).lower() == "swatch":
        raise RuntimeError("swatch mode is unreachable through load_config (SURVEY A.0) and is not provided")
    masks, info = S.extract_colors(img, cfg)
    palette = {}
    for k, name in enumerate(S.cluster_names(cfg)[:len(masks)]):
        layer_dir = os.path.join(cfg.output_dir, name)
        os.makedirs(layer_dir, exist_ok=True)
        _io.write_png(os.path.join(layer_dir, "mask.png"), masks[name])
        nz = int(np.count_nonzero(masks[name]))
        lab = info["centers_lab"][k]
        palette[name] = {"mode": "kmeans", "cluster_index": int(k), "cluster_lab": [int(v) for v in lab],
                         "approx_bgr": list(S.lab8_to_bgr(lab.astype(np.uint8))), "pixels": int(info["counts"][k]), "mask_nonzero": nz}
        print(f"Extracted (kmeans): {name} | cluster={k} | L*={lab[0]:.1f} | pixels={int(info['counts'][k])} | nz={nz}")
    pal_path = os.path.join(cfg.output_dir, "palette_by_name.json")
    with open(pal_path, "w", encoding="utf-8") as f:
        json.dump(palette, f, ensure_ascii=False, indent=2)
    print(f"Palette saved: {pal_path}")
    print("Color extraction: done.")


if __name__ == "__main__":
    main()
