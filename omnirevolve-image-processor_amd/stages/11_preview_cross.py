# 11_preview_cross.py -- drop-in: <layer>/lines_cross.pkl + taps_cross.pkl -> <layer>/preview_cross.png + preview_cross_composite.png on the full target
# canvas (11_preview_cross.py main): black lines and red tap discs per layer, the layers' palette colours in the composite (later layers on top).
# Visual QA only; rasterised on the GPU with the documented stand-in for cv2.LINE_AA (include/orip.h: orip_preview_cover -- parity unpinned).
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config, canvas_size_px


def main():
    cfg = load_config()
    outdir = cfg.output_dir
    size = canvas_size_px(cfg)
    pen_r = int(getattr(cfg, "pen_radius_px", max(1, int(round(getattr(cfg, "pixels_per_mm", 40) * 0.75)))))
    th = int(getattr(cfg, "preview_line_thickness_px", 1))
    aa = bool(getattr(cfg, "preview_antialiased", True))
    palette = S.preview_palette(cfg)
    composite = np.full((size[1], size[0], 3), 255, np.uint8)
    for name in cfg.color_names:
        layer_dir = os.path.join(outdir, name)
        os.makedirs(layer_dir, exist_ok=True)
        pL, pT = os.path.join(layer_dir, "lines_cross.pkl"), os.path.join(layer_dir, "taps_cross.pkl")
        for p in (pL, pT):
            if not _io.exists(p):
                raise RuntimeError(f"Missing required input: {p}")
        lines = _io.load_pickle(pL)
        if not isinstance(lines, list):
            raise RuntimeError(f"Invalid pickle format: {pL}")
        taps = []
        for it in _io.load_pickle(pT):
            a = np.asarray(it).reshape(-1)
            if a.size >= 2:
                taps.append((int(a[0]), int(a[1])))
        lay, col = S.preview_images(lines, taps, size, th, pen_r, aa, palette.get(name, (0, 0, 0)))
        out_l = os.path.join(layer_dir, "preview_cross.png")
        _io.write_png(out_l, lay)
        mask = (col != 255).any(axis=2)
        composite[mask] = col[mask]
        print(f"[preview_cross] {name}: lines={len(lines)}, taps={len(taps)} → {out_l}")
    out_c = os.path.join(outdir, "preview_cross_composite.png")
    _io.write_png(out_c, composite)
    print(f"[preview_cross] composite saved: {out_c}")


if __name__ == "__main__":
    main()
