#!/usr/bin/env python3
# process_colors.py -- drop-in for the reference tool of the same name (v1.1.1 outputs): labels.png / labels.npy / palette.json /
# layer_<idx>_<name>.png from an image, adaptive (k-means) or palette mode.  Compute: liborip.so on the GPU (RGB k-means, nearest palette colour).
import argparse
import json
from pathlib import Path

import numpy as np

import stage_io as _io
from orip import colors as PC
from orip.device import Device


def main():
    ap = argparse.ArgumentParser(description="One-hot color layer generator with labels output")
    ap.add_argument("input", help="Input image")
    ap.add_argument("-o", "--output", default="layers", help="Output directory")
    ap.add_argument("-m", "--mode", choices=["adaptive", "palette"], default="adaptive", help="adaptive: KMeans; palette: load palette JSON")
    ap.add_argument("-n", "--colors", type=int, default=4, help="Number of colors for adaptive")
    ap.add_argument("--palette", help="Palette JSON (from analyze_colors.py) for mode=palette")
    ap.add_argument("--edges-only", action="store_true", help="Kept for pipeline compatibility (ignored)")
    args = ap.parse_args()
    out_dir = Path(args.output).absolute()
    out_dir.mkdir(parents=True, exist_ok=True)
    print("[process_colors] v1.1.1 (GPU)")
    print(f"[process_colors] Input: {args.input}")
    print(f"[process_colors] Output dir: {out_dir}")
    bgr = _io.read_bgr(args.input)
    if bgr is None:
        raise ValueError(f"Cannot load image: {args.input}")
    h, w = bgr.shape[:2]
    print(f"[process_colors] Size: {w}x{h}")
    dev = Device(0)
    dev.set_image(bgr)
    if args.mode == "palette":
        if not args.palette:
            raise ValueError("Mode 'palette' requires --palette JSON")
        palette_rgb, names = PC.palette_from_json(args.palette)
        K = len(palette_rgb)
        if args.colors and args.colors != K:
            print(f"[WARN] --colors={args.colors} ignored; palette has {K} entries.")
    else:
        K = int(args.colors) if args.colors else 4
        palette_rgb = PC.kmeans_palette(dev, K)
        names = PC.default_color_names(K)
    labels, counts = dev.assign_palette(palette_rgb)
    total = labels.size
    print("[process_colors] Class distribution:")
    for i in range(K):
        p = 100.0 * int(counts[i]) / total if total else 0.0
        nm = names[i] if i < len(names) else f"color_{i}"
        print(f"  [{i}] {nm:12s}  {tuple(int(v) for v in palette_rgb[i])}  pixels={int(counts[i]):8d}  {p:5.1f}%")
    _io.write_png(str(out_dir / "labels.png"), labels)
    np.save(str(out_dir / "labels.npy"), labels)
    with open(out_dir / "palette.json", "w", encoding="utf-8") as f:
        json.dump(PC.palette_dump(palette_rgb, names), f, indent=2)
    for i in range(K):
        nm = names[i] if i < len(names) else f"color_{i}"
        _io.write_png(str(out_dir / f"layer_{i+1}_{nm}.png"), (labels == i).astype(np.uint8) * 255)
    print(f"[process_colors] Done. {K} layer files written to: {out_dir}")
    if args.edges_only:
        print("[process_colors] NOTE: --edges-only is ignored here (kept for pipeline compatibility).")
    dev.close()


if __name__ == "__main__":
    main()
