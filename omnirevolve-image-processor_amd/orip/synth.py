"""Synthetic benchmark / test inputs: low-pass noise quantised into K equal-area
colour classes, grey palette with a small tint, +-3 per-channel noise.  Stored as BGR uint8 [H,W,3]
(the layout of `resized.png` as the reference's stage 02 reads it, 02_color_extract.py:70-71)."""
from __future__ import annotations

from typing import List

import numpy as np

DEFAULT_NAMES = ["layer_dark", "layer_mid", "layer_skin", "layer_light"]


def layer_names(K: int) -> List[str]:
    if K <= 4:
        return DEFAULT_NAMES[:K]
    return DEFAULT_NAMES + [f"layer_{i}" for i in range(4, K)]


def palette_bgr(K: int) -> np.ndarray:
    pal = np.zeros((K, 3), np.int32)
    for k in range(K):
        v = int(round(16 + 224 * k / max(1, K - 1)))
        pal[k] = (v, v, v)
        pal[k, k % 3] += 8
    return pal


def synth_image(H: int, W: int, K: int, seed: int = 20251121, sigma: float | None = None) -> np.ndarray:
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    f = gaussian_filter(rng.standard_normal((H, W), dtype=np.float32), sigma=(H / 128.0 if sigma is None else sigma), mode="reflect")
    qs = np.quantile(f, [i / K for i in range(1, K)])
    c = np.digitize(f, qs)
    img = palette_bgr(K)[c] + rng.integers(-3, 4, (H, W, 3))
    return np.clip(img, 0, 255).astype(np.uint8)
