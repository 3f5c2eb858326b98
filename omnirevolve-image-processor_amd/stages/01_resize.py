# 01_resize.py -- drop-in for the reference stage of the same name: input image -> resized.png (INTER_AREA shrink to max_dimension on the longest
# side, 01:7-23).  Compute: liborip.so on the GPU (csrc/raster01.hip).  The decode of the input file is Pillow's (the reference's is cv2.imread):
# 8-bit, three channels, BGR order in memory.
import os

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    cfg.ensure_output_dirs()
    img = _io.read_bgr(cfg.input_image)
    if img is None:
        raise ValueError(f"Failed to load image: {cfg.input_image}")
    h, w = img.shape[:2]
    out = S.resize_if_needed(img, cfg)
    if out is img:
        print(f"No resize required: {w}x{h}")
    else:
        print(f"Resizing: {w}x{h} -> {out.shape[1]}x{out.shape[0]}")
    out_path = os.path.join(cfg.output_dir, "resized.png")
    _io.write_png(out_path, out)
    print(f"Saved: {out_path}")


if __name__ == "__main__":
    main()
