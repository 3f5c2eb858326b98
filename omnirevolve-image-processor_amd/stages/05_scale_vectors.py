# 05_scale_vectors.py -- drop-in: <layer>/contours.pkl -> <layer>/contours_scaled.pkl (canvas px)
import os

import stage_io as _io
from orip import stages as S
from orip.config import load_config, canvas_size_px, scale_factors


def main():
    cfg = load_config()
    base = _io.read_bgr(os.path.join(cfg.output_dir, "resized.png"))
    if base is None:
        raise RuntimeError("Missing resized.png (run step 1 first).")
    h, w = base.shape[:2]
    sx, sy, dx, dy = scale_factors(cfg, w, h)
    print(f"[scale] source={w}x{h}, target(full)={canvas_size_px(cfg)}, scale=({sx:.4f},{sy:.4f}), offset=({dx},{dy})")
    for name in cfg.color_names:
        cdir = os.path.join(cfg.output_dir, name)
        os.makedirs(cdir, exist_ok=True)
        src = os.path.join(cdir, "contours.pkl")
        if not _io.exists(src):
            print(f"[scale] {name}: missing {src}, skipping")
            continue
        contours = _io.load_pickle(src)
        scaled = S.scale_vectors(contours, w, h, cfg)
        _io.save_pickle(os.path.join(cdir, "contours_scaled.pkl"), _io.polys_out(scaled))
        print(f"[scale] {name}: contours={len(contours)}")


if __name__ == "__main__":
    main()
