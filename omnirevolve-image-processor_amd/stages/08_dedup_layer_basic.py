# 08_dedup_layer_basic.py -- drop-in: <layer>/contours_sorted.pkl -> <layer>/lines_intra.pkl + taps_intra.pkl
import os

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    for name in cfg.color_names:
        layer_dir = os.path.join(cfg.output_dir, name)
        src = os.path.join(layer_dir, "contours_sorted.pkl")
        if not _io.exists(src):
            raise RuntimeError(f"[intra] missing input: {src}. Run step 06 first.")
        polys = _io.load_pickle(src)
        if not isinstance(polys, list):
            raise RuntimeError(f"[intra] invalid pickle format: {src}")
        if not polys:
            print(f"[intra] {name}: empty input.")      # the reference writes nothing in this case (08:511-513)
            continue
        lines, taps = S.dedup_layer(polys, cfg)
        _io.save_pickle(os.path.join(layer_dir, "lines_intra.pkl"), _io.polys_out(lines))
        _io.save_pickle(os.path.join(layer_dir, "taps_intra.pkl"), [(int(x), int(y)) for x, y in taps])
        print(f"[intra] {name}: lines={len(lines)}, taps={len(taps)}")


if __name__ == "__main__":
    main()
