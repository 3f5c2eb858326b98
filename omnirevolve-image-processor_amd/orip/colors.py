"""Host side of process_colors.py (the reference's standalone label-map tool): palette estimation and strict one-hot layers.
The two data-parallel steps run in liborip.so (orip_kmeans_fit_rgb, orip_assign_palette); what stays here is the reference's host logic:
the numpy subsample, the palette file layouts, names and the output files.  Citations: /root/reference/image_processor/process_colors.py."""
from __future__ import annotations

import json
from typing import List, Tuple

import numpy as np

from .device import Device


def subsample_indices(n_pixels: int, samples: int = 200000, seed: int = 1):
    """:35-39 -- RandomState(seed).choice(N, samples, replace=False) when the image has more pixels than `samples`, else every pixel (None)"""
    if n_pixels > samples:
        return np.random.RandomState(seed).choice(n_pixels, size=samples, replace=False).astype(np.int64)
    return None


def kmeans_palette(dev: Device, k: int, samples: int = 200000, seed: int = 1) -> np.ndarray:
    """:31-46 on the image resident in `dev` -- u8 [k,3] R, G, B (float centres truncated by astype(uint8), as the reference does)"""
    idx = subsample_indices(dev.H * dev.W, samples, seed)
    centers, _ = dev.kmeans_fit_rgb(idx, k, attempts=3, max_iter=30, eps=1.0)
    return centers.astype(np.uint8)


def default_color_names(k: int) -> List[str]:  # :80-82
    base = ["red", "green", "blue", "black"]
    return [base[i] if i < len(base) else f"color_{i}" for i in range(k)]


def palette_from_json(path: str) -> Tuple[np.ndarray, List[str]]:
    """:49-66.  "recommended_colors" (sorted by position) as the reference reads it.  The "palette" layout ends in a NameError inside the reference
    (:63 names the loop variable of another comprehension); here it loads, with the names the reference's expression spells out."""
    with open(path, "r", encoding="utf-8") as f:
        data = json.load(f)
    if "recommended_colors" in data:
        items = sorted(data["recommended_colors"], key=lambda it: it.get("position", 1e9))
        names = [str(it.get("name", f"color_{i}")) for i, it in enumerate(items)]
        return np.array([it["rgb"] for it in items], dtype=np.uint8), names
    if "palette" in data:
        items = data["palette"]
        return np.array([c["rgb"] for c in items], dtype=np.uint8), [str(c.get("name", f"color_{i}")) for i, c in enumerate(items)]
    raise ValueError(f"Unsupported palette JSON structure: {path}")


def palette_dump(palette_rgb: np.ndarray, names: List[str]) -> dict:  # :150-160
    K = len(palette_rgb)
    return {"colors": [{"index": int(i), "name": (names[i] if i < len(names) else f"color_{i}"), "rgb": [int(c) for c in palette_rgb[i].tolist()]} for i in range(K)]}
