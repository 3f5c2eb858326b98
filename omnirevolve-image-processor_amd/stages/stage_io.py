"""File-level artefact I/O of the stage scripts: same names and types as the reference's chain:
PNG through Pillow (OpenCV is not a dependency), pickles of numpy int32 (N,1,2) lists / 2-tuples / op dicts.

Raw side channel: at GPU speed the PNG codec and pickle dominate the wall time of a stage script (02:71,156, 03:19,37,
04:216,227).  With ORIP_RAW_NPY=1 in the environment every raster is ALSO written as `<name>.png.npy` (the array as it is, BGR order for
colour images) and every polyline list as `<name>.pkl.npz` (off int64 [n+1], pts int32 [total,2]); readers take the raw file when it exists and is
at least as new as the reference-format file (or when that one is missing), so a chain of these scripts skips the codecs while the reference's
own stages still find the files they expect."""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # the package directory that holds `orip`


RAW = os.environ.get("ORIP_RAW_NPY", "0") not in ("", "0")


def _raw_ok(raw: str, ref: str) -> bool:
    return os.path.exists(raw) and (not os.path.exists(ref) or os.path.getmtime(raw) >= os.path.getmtime(ref))


def exists(path: str) -> bool:
    """an artefact is there in the reference's format or in the raw side channel"""
    return os.path.exists(path) or os.path.exists(path + ".npy") or os.path.exists(path + ".npz")


def read_bgr(path: str):
    from PIL import Image
    if _raw_ok(path + ".npy", path):
        a = np.load(path + ".npy")
        return np.repeat(a[:, :, None], 3, axis=2) if a.ndim == 2 else a
    if not os.path.exists(path):
        return None
    im = Image.open(path)
    if im.mode in ("L", "I;16", "1"):
        g = np.array(im.convert("L"))
        return np.repeat(g[:, :, None], 3, axis=2)
    return np.ascontiguousarray(np.array(im.convert("RGB"))[:, :, ::-1])


def read_gray(path: str):
    from PIL import Image
    if _raw_ok(path + ".npy", path):
        return np.load(path + ".npy")
    if not os.path.exists(path):
        return None
    return np.array(Image.open(path).convert("L"))


def write_png(path: str, arr: np.ndarray):
    from PIL import Image
    a = np.asarray(arr)
    Image.fromarray(a[:, :, ::-1] if a.ndim == 3 else a).save(path)
    if RAW:                                                  # after the PNG: the raw file must not be the older of the two
        np.save(path + ".npy", np.ascontiguousarray(a))


def _is_poly_list(obj) -> bool:
    return isinstance(obj, list) and all(isinstance(p, np.ndarray) and p.ndim == 3 and p.shape[1:] == (1, 2) for p in obj)


def load_pickle(path: str):
    if _raw_ok(path + ".npz", path):
        z = np.load(path + ".npz")
        off, pts = z["off"], z["pts"]
        return [pts[off[i]:off[i + 1]].reshape(-1, 1, 2) for i in range(len(off) - 1)]
    with open(path, "rb") as f:
        return pickle.load(f)


def save_pickle(path: str, obj):
    with open(path, "wb") as f:
        pickle.dump(obj, f)
    if RAW and _is_poly_list(obj):
        off = np.zeros(len(obj) + 1, np.int64)
        for i, p in enumerate(obj):
            off[i + 1] = off[i] + len(p)
        pts = np.concatenate([p.reshape(-1, 2) for p in obj], 0).astype(np.int32) if obj else np.zeros((0, 2), np.int32)
        np.savez(path + ".npz", off=off, pts=pts)


def polys_out(polys):
    return [np.asarray(p).reshape(-1, 1, 2).astype(np.int32) for p in polys]
