# 10_dedup_cross_basic.py -- drop-in: <layer>/lines_intra.pkl + taps_intra.pkl -> <layer>/lines_cross.pkl + taps_cross.pkl
import os

import numpy as np

import stage_io as _io
from orip import stages as S
from orip.config import load_config


def main():
    cfg = load_config()
    intra = {}
    for name in cfg.color_names:
        layer_dir = os.path.join(cfg.output_dir, name)
        os.makedirs(layer_dir, exist_ok=True)
        pL, pT = os.path.join(layer_dir, "lines_intra.pkl"), os.path.join(layer_dir, "taps_intra.pkl")
        lines, taps = [], []
        if _io.exists(pL):
            lines = _io.load_pickle(pL)
        else:
            print(f"[cross] WARNING: missing {pL}")
        if _io.exists(pT):
            for it in _io.load_pickle(pT):
                a = np.asarray(it).reshape(-1)
                if a.size >= 2:
                    taps.append((int(a[0]), int(a[1])))
        else:
            print(f"[cross] WARNING: missing {pT}")
        intra[name] = (lines, taps)
    out = S.dedup_cross(intra, cfg)
    for name, (lines, taps) in out.items():
        layer_dir = os.path.join(cfg.output_dir, name)
        _io.save_pickle(os.path.join(layer_dir, "lines_cross.pkl"), _io.polys_out(lines))
        _io.save_pickle(os.path.join(layer_dir, "taps_cross.pkl"), [(int(x), int(y)) for x, y in taps])
        print(f"[cross] {name}: lines {len(intra[name][0])}->{len(lines)}, taps {len(intra[name][1])}->{len(taps)}")


if __name__ == "__main__":
    main()
