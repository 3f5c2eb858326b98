# 02_color_extract.py -- drop-in for the reference stage of the same name (k-means mode): resized.png -> <layer>/mask.png
# + palette_by_name.json.  Compute: liborip.so on the GPU (Lab, k-means fit, assignment, 3x3 open/close).
import json
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def _lab8_to_bgr(lab):
    """Approximate inverse of the 8-bit Lab encoding for the palette preview colour (a6 in SURVEY 8a; previews only)."""
    L = lab[0] * 100.0 / 255.0; a = lab[1] - 128.0; b = lab[2] - 128.0
    fy = (L + 16.0) / 116.0; fx = fy + a / 500.0; fz = fy - b / 200.0
    def finv(t): return t ** 3 if t ** 3 > 0.008856 else (t - 16.0 / 116.0) / 7.787
    X, Y, Z = 0.950456 * finv(fx), finv(fy), 1.088754 * finv(fz)
    rgb = np.array([3.240479 * X - 1.53715 * Y - 0.498535 * Z, -0.969256 * X + 1.875991 * Y + 0.041556 * Z, 0.055648 * X - 0.204043 * Y + 1.057311 * Z])
    rgb = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.clip(rgb, 0, None) ** (1 / 2.4) - 0.055)
    r, g, bb = (int(np.clip(round(v * 255.0), 0, 255)) for v in rgb)
    return bb, g, r


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
                         "approx_bgr": list(_lab8_to_bgr(lab.astype(np.uint8).astype(np.float64))), "pixels": int(info["counts"][k]), "mask_nonzero": nz}
        print(f"Extracted (kmeans): {name} | cluster={k} | L*={lab[0]:.1f} | pixels={int(info['counts'][k])} | nz={nz}")
    pal_path = os.path.join(cfg.output_dir, "palette_by_name.json")
    with open(pal_path, "w", encoding="utf-8") as f:
        json.dump(palette, f, ensure_ascii=False, indent=2)
    print(f"Palette saved: {pal_path}")
    print("Color extraction: done.")


if __name__ == "__main__":
    main()
