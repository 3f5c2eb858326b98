# 06_preview_scaled.py -- drop-in: <layer>/contours_scaled.pkl (else contours_sorted.pkl, contours.pkl) -> <layer>/scaled_preview.png +
# scaled_preview_composite.png on the full target canvas (06_preview_scaled.py:91-136).  Visual QA only; the lines are rasterised on the GPU with the
# documented stand-in for cv2.LINE_AA (include/orip.h: orip_preview_cover -- parity unpinned).
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config, canvas_size_px, margins_px


def main():
    cfg = load_config()
    outdir = cfg.output_dir
    os.makedirs(outdir, exist_ok=True)
    thickness = int(getattr(cfg, "scaled_preview_thickness_px", 1))
    aa = bool(getattr(cfg, "scaled_preview_antialiased", True))
    size = canvas_size_px(cfg)
    ml, mr, mt, mb = margins_px(cfg)
    print(f"[scaled_preview] canvas(full)={size[0]}x{size[1]}, margins(l,r,t,b)=({ml},{mr},{mt},{mb}), inner={max(1, size[0] - ml - mr)}x{max(1, size[1] - mt - mb)}, offset=({ml},{mt})")
    palette = S.preview_palette(cfg)
    composite = np.full((size[1], size[0], 3), 255, np.uint8)
    total_polys = total_vertices = 0
    for name in cfg.color_names:
        layer_dir = os.path.join(outdir, name)
        os.makedirs(layer_dir, exist_ok=True)
        polys = []
        for fname in ("contours_scaled.pkl", "contours_sorted.pkl", "contours.pkl"):
            p = os.path.join(layer_dir, fname)
            if _io.exists(p):
                obj = _io.load_pickle(p)
                if isinstance(obj, list):
                    polys = obj
                    break
        vcount = sum(int(np.asarray(q).reshape(-1, 2).shape[0]) for q in polys) if polys else 0
        total_polys += len(polys); total_vertices += vcount
        lay, col = S.preview_images(polys, [], size, thickness, 0, aa, palette.get(name, (0, 0, 0)))
        out_layer = os.path.join(layer_dir, "scaled_preview.png")
        _io.write_png(out_layer, lay)
        mask = (col != 255).any(axis=2)
        composite[mask] = col[mask]
        print(f"[scaled_preview] {name}: contours={len(polys)}, vertices={vcount} → {out_layer}")
    out_comp = os.path.join(outdir, "scaled_preview_composite.png")
    _io.write_png(out_comp, composite)
    print(f"[scaled_preview] composite saved: {out_comp}")
    print(f"[scaled_preview] totals: contours={total_polys}, vertices={total_vertices}")


if __name__ == "__main__":
    main()
