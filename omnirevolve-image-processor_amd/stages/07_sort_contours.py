# 07_sort_contours.py -- drop-in: <layer>/contours_scaled.pkl (else contours.pkl) -> <layer>/contours_sorted.pkl
import os

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    for name in cfg.color_names:
        cdir = os.path.join(cfg.output_dir, name)
        os.makedirs(cdir, exist_ok=True)
        src = os.path.join(cdir, "contours_scaled.pkl")
        if not _io.exists(src):
            src = os.path.join(cdir, "contours.pkl")
        if not _io.exists(src):
            print(f"[sort] skip (missing): {src}")
            continue
        contours = _io.load_pickle(src)
        out = S.sort_contours(contours) if contours else []
        _io.save_pickle(os.path.join(cdir, "contours_sorted.pkl"), _io.polys_out(out))
        print(f"[sort] {name}: contours={len(out)}")


if __name__ == "__main__":
    main()
