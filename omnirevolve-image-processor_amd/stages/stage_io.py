"""File-level artefact I/O of the stage scripts: same names and types as the reference's chain:
PNG through Pillow (OpenCV is not a dependency), pickles of numpy int32 (N,1,2) lists / 2-tuples / op dicts."""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # the package directory that holds `orip`


def read_bgr(path: str):
    from PIL import Image
    if not os.path.exists(path):
        return None
    im = Image.open(path)
    if im.mode in ("L", "I;16", "1"):
        g = np.array(im.convert("L"))
        return np.repeat(g[:, :, None], 3, axis=2)
    return np.ascontiguousarray(np.array(im.convert("RGB"))[:, :, ::-1])


def read_gray(path: str):
    from PIL import Image
    if not os.path.exists(path):
        return None
    return np.array(Image.open(path).convert("L"))


def write_png(path: str, arr: np.ndarray):
    from PIL import Image
    a = np.asarray(arr)
    if a.ndim == 3:
        a = a[:, :, ::-1]
    Image.fromarray(a).save(path)


def load_pickle(path: str):
    with open(path, "rb") as f:
        return pickle.load(f)


def save_pickle(path: str, obj):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def polys_out(polys):
    return [np.asarray(p).reshape(-1, 1, 2).astype(np.int32) for p in polys]
