#!/usr/bin/env python3
# process_colors.py -- GPU drop-in for the reference's standalone label-map tool: same command line (input, -o, -m adaptive|palette, -n,
# --palette, --edges-only) and the same files in the output directory (labels.png, labels.npy, palette.json, layer_<i>_<name>.png).
# Compute: liborip.so (RGB k-means palette: orip_kmeans_fit_rgb; nearest palette colour per pixel: orip_assign_palette).
import argparse
import json
import pathlib
import sys

import numpy as np

import stage_io as _io
from orip import colors as PC
from orip.device import Device


def _cli(argv):
    p = argparse.ArgumentParser(description="strict one-hot colour layers + label map (GPU)")
    p.add_argument("input")
    p.add_argument("-o", "--output", default="layers")
    p.add_argument("-m", "--mode", choices=("adaptive", "palette"), default="adaptive")
    p.add_argument("-n", "--colors", type=int, default=4)
    p.add_argument("--palette")
    p.add_argument("--edges-only", action="store_true")          # accepted and ignored, as in the reference
    return p.parse_args(argv)


def _palette(opts, dev):
    if opts.mode == "palette":
        if not opts.palette:
            raise ValueError("Mode 'palette' requires --palette JSON")
        rgb, names = PC.palette_from_json(opts.palette)
        if opts.colors and opts.colors != len(rgb):
            print(f"[WARN] --colors={opts.colors} ignored; palette has {len(rgb)} entries.")
        return rgb, names
    k = int(opts.colors) if opts.colors else 4
    return PC.kmeans_palette(dev, k), PC.default_color_names(k)


def run(opts) -> int:
    out = pathlib.Path(opts.output).absolute()
    out.mkdir(parents=True, exist_ok=True)
    bgr = _io.read_bgr(opts.input)
    if bgr is None:
        raise ValueError(f"Cannot load image: {opts.input}")
    print(f"[process_colors] {opts.input}: {bgr.shape[1]}x{bgr.shape[0]} -> {out}")
    dev = Device(0)
    try:
        dev.set_image(bgr)
        rgb, names = _palette(opts, dev)
        labels, counts = dev.assign_palette(rgb)
    finally:
        dev.close()
    name_of = lambda i: names[i] if i < len(names) else f"color_{i}"
    share = 100.0 / max(1, labels.size)
    for i in range(len(rgb)):
        print(f"  [{i}] {name_of(i):12s}  {tuple(int(v) for v in rgb[i])}  pixels={int(counts[i]):8d}  {share * int(counts[i]):5.1f}%")
    _io.write_png(str(out / "labels.png"), labels)
    np.save(str(out / "labels.npy"), labels)
    (out / "palette.json").write_text(json.dumps(PC.palette_dump(rgb, names), indent=2), encoding="utf-8")
    for i in range(len(rgb)):
        _io.write_png(str(out / f"layer_{i + 1}_{name_of(i)}.png"), np.where(labels == i, 255, 0).astype(np.uint8))
    print(f"[process_colors] {len(rgb)} layers, labels.png / labels.npy / palette.json written")
    return 0


if __name__ == "__main__":
    sys.exit(run(_cli(sys.argv[1:])))
