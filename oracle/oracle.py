"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding + stage-level composition of the CPU restatement (oracle/src/*.cpp) of the hot path of
omnirevolve-image-processor (stages 02 -> 12).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (omnirevolve-image-processor_amd/) never does.

Pinning status (DESIGN.md "Oracle"): every pure-numpy/Python function of the reference on the path is
pinned by golden vectors captured from the reference itself (tests/golden/make_golden.py); every step
that bottoms out in OpenCV (cv2 is not installed) is "parity unpinned" and follows SURVEY
Appendix B.  All file:line citations refer to /root/reference/image_processor/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborip_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, i32, f64, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_float
        sig = {
            "orc_pl_new": (vp, []), "orc_pl_free": (None, [vp]), "orc_pl_count": (i64, [vp]), "orc_pl_total": (i64, [vp]),
            "orc_pl_get": (None, [vp, vp, vp]), "orc_pl_set": (None, [vp, i64, vp, vp]),
            "orc_taps_new": (vp, []), "orc_taps_free": (None, [vp]), "orc_taps_count": (i64, [vp]),
            "orc_taps_get": (None, [vp, vp]), "orc_taps_set": (None, [vp, i64, vp]),
            "orc_lab_tables": (None, [vp, vp, vp]), "orc_bgr2lab": (None, [vp, i64, vp]),
            "orc_kmeans": (f64, [vp, i32, i32, i32, i32, f64, vp]), "orc_assign": (None, [vp, i64, vp, i32, vp]),
            "orc_morph_open_close": (None, [vp, i32, i32, i32, i32, i32, i32]), "orc_make_se": (None, [i32, i32, vp]),
            "orc_gaussian": (i32, [vp, vp, i32, i32, i32]), "orc_resize_area": (i32, [vp, i32, i32, i32, vp, i32, i32]), "orc_canny": (None, [vp, vp, i32, i32, i32, i32]),
            "orc_thin_rot": (i32, [vp, vp, i32, i32]), "orc_zs_std": (i32, [vp, vp, i32, i32, i32]),
            "orc_ccl8": (i32, [vp, vp, i32, i32]), "orc_trace": (None, [vp, i32, i32, vp]),
            "orc_stamp_capsule": (None, [vp, i32, i32, i32, i32, i32, i32, i32]),
            "orc_arc_length": (f64, [vp, i64, i32]), "orc_poly_perimeter": (f32, [vp, i64]), "orc_mec": (None, [vp, i64, vp]),
            "orc_scale": (None, [vp, f32, f32, f32, f32, vp]), "orc_sort07": (None, [vp, vp]), "orc_reorder": (None, [vp, vp, i32]),
            "orc_resample": (i64, [vp, i64, i32, f64, vp, i64, vp]), "orc_split_jumps": (None, [vp, i64, f64, i32, vp]),
            "orc_split_small_taps08": (None, [vp, vp, vp, vp]), "orc_virtual_draw08": (None, [vp, i64, vp, vp, vp]),
            "orc_cluster_by_overlap": (None, [vp, i32, vp]), "orc_bfs_path": (i64, [vp, i32, i32, i32, i32, i32, i32, vp, i64]),
            "orc_component_best_path": (i64, [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64]),
            "orc_post_skeleton_merge": (None, [vp, vp, vp]), "orc_stage08": (None, [vp, vp, vp, vp]),
            "orc_cut_poly": (None, [vp, i64, vp, i32, i32, f64, vp]), "orc_tiny_and_taps10": (None, [vp, vp, vp, vp]),
            "orc_stage10_layer": (None, [vp, vp, vp, vp, vp, vp]), "orc_build_ops12": (i64, [vp, vp, f64, vp, i64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ list marshalling
class PL:
    """Owned PolyList handle."""

    def __init__(self, polys: Sequence[np.ndarray] | None = None):
        self.h = lib().orc_pl_new()
        if polys is not None:
            self.set(polys)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pl_free(self.h)
            self.h = None

    def set(self, polys: Sequence[np.ndarray]):
        n = len(polys)
        off = np.zeros(n + 1, np.int64)
        flat = [np.asarray(p).reshape(-1, 2).astype(np.int32) for p in polys]
        for i, p in enumerate(flat):
            off[i + 1] = off[i] + len(p)
        pts = np.ascontiguousarray(np.concatenate(flat, 0) if n else np.zeros((0, 2), np.int32), dtype=np.int32)
        if pts.size == 0:
            pts = np.zeros((1, 2), np.int32)
        lib().orc_pl_set(self.h, n, _p(off), _p(pts))

    def get(self) -> List[np.ndarray]:
        n = lib().orc_pl_count(self.h)
        tot = lib().orc_pl_total(self.h)
        off = np.zeros(n + 1, np.int64)
        pts = np.zeros((max(tot, 1), 2), np.int32)
        lib().orc_pl_get(self.h, _p(off), _p(pts))
        return [pts[off[i]:off[i + 1]].reshape(-1, 1, 2).copy() for i in range(n)]


class TL:
    def __init__(self, taps: Sequence[Tuple[int, int]] | None = None):
        self.h = lib().orc_taps_new()
        if taps is not None:
            a = np.ascontiguousarray(np.asarray(list(taps), np.int32).reshape(-1, 2))
            if len(a):
                lib().orc_taps_set(self.h, len(a), _p(a))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_taps_free(self.h)
            self.h = None

    def get(self) -> List[Tuple[int, int]]:
        n = lib().orc_taps_count(self.h)
        a = np.zeros((max(n, 1), 2), np.int32)
        lib().orc_taps_get(self.h, _p(a))
        return [(int(a[i, 0]), int(a[i, 1])) for i in range(n)]


def _i32xy(poly) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(poly).reshape(-1, 2), dtype=np.int32)


# ------------------------------------------------------------------ raster primitives
def lab_tables():
    g = np.zeros(256, np.uint16); c = np.zeros(3072, np.uint16); k = np.zeros(9, np.int32)
    lib().orc_lab_tables(_p(g), _p(c), _p(k))
    return g, c, k


def bgr2lab(bgr: np.ndarray) -> np.ndarray:
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty_like(bgr)
    lib().orc_bgr2lab(_p(bgr), bgr.size // 3, _p(out))
    return out


def lab8_to_bgr(lab_u8) -> Tuple[int, int, int]:
    """_lab_to_bgr (02:58-61): cv2.cvtColor(1x1 Lab u8, COLOR_Lab2BGR).  PARITY UNPINNED (cv2 absent): restated as the
    float inverse of the CIE L*a*b* D65 / sRGB transform OpenCV documents for Lab2BGR (L = v0*100/255, a = v1-128, b = v2-128;
    fY = (L+16)/116, fX = fY + a/500, fZ = fY - b/200; t^3 above the 0.008856 knee, (t-16/116)/7.787 below; XYZ -> linear sRGB
    with the inverse of the matrix the forward path uses; sRGB gamma; round to nearest, saturate).  OpenCV's own 8-bit path is a
    fixed-point version of the same transform, so single-LSB differences are possible; the committed check is the round trip through
    the forward tables of this oracle (tests/test_oracle_golden_pure.py::test_lab8_to_bgr_round_trip)."""
    v = np.asarray(lab_u8, np.float64).reshape(3)
    L = v[0] * 100.0 / 255.0; a = v[1] - 128.0; b = v[2] - 128.0
    fy = (L + 16.0) / 116.0; fx = fy + a / 500.0; fz = fy - b / 200.0

    def finv(t):
        t3 = t * t * t
        return t3 if t3 > 0.008856 else (t - 16.0 / 116.0) / 7.787

    X, Y, Z = 0.950456 * finv(fx), finv(fy), 1.088754 * finv(fz)
    lin = (3.240479 * X - 1.53715 * Y - 0.498535 * Z, -0.969256 * X + 1.875991 * Y + 0.041556 * Z, 0.055648 * X - 0.204043 * Y + 1.057311 * Z)

    def gamma(c):
        return 12.92 * c if c <= 0.0031308 else 1.055 * (max(c, 0.0) ** (1.0 / 2.4)) - 0.055

    r, g, bb = (int(min(255, max(0, round(gamma(c) * 255.0)))) for c in lin)
    return bb, g, r


def kmeans(samples: np.ndarray, K: int, attempts=3, max_iter=40, eps=0.5) -> Tuple[np.ndarray, float]:
    s = np.ascontiguousarray(samples, np.float32).reshape(-1, 3)
    centers = np.zeros((K, 3), np.float32)
    comp = lib().orc_kmeans(_p(s), len(s), K, attempts, max_iter, eps, _p(centers))
    return centers, comp


def assign(lab: np.ndarray, centers: np.ndarray) -> np.ndarray:
    lab = np.ascontiguousarray(lab, np.uint8)
    c = np.ascontiguousarray(centers, np.float32)
    out = np.empty(lab.size // 3, np.int32)
    lib().orc_assign(_p(lab), lab.size // 3, _p(c), len(c), _p(out))
    return out.reshape(lab.shape[:-1])


def morph_open_close(mask, shape: int, k: int, open_iters=1, close_iters=1) -> np.ndarray:
    m = np.ascontiguousarray(mask, np.uint8).copy()
    lib().orc_morph_open_close(_p(m), m.shape[0], m.shape[1], shape, k, open_iters, close_iters)
    return m


def make_se(shape: int, k: int) -> np.ndarray:
    se = np.zeros((k, k), np.uint8)
    lib().orc_make_se(shape, k, _p(se))
    return se


def resize_area(img, new_w: int, new_h: int) -> np.ndarray:
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA) for shrinking (01:19); parity unpinned (OpenCV absent)"""
    a = np.ascontiguousarray(img, np.uint8)
    cn = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty((new_h, new_w) if a.ndim == 2 else (new_h, new_w, cn), np.uint8)
    if lib().orc_resize_area(_p(a), a.shape[0], a.shape[1], cn, _p(out), new_h, new_w) != 0:
        raise ValueError(f"resize_area: {a.shape[1]}x{a.shape[0]} -> {new_w}x{new_h} is not a shrink")
    return out


def resize_if_needed(img, max_dimension: int = 2000) -> np.ndarray:  # 01:7-23
    h, w = img.shape[:2]
    m = max(h, w)
    if m <= max_dimension:
        return img
    s = max_dimension / m
    return resize_area(img, int(w * s), int(h * s))


def assign_labels_rgb(img_rgb, palette_rgb) -> np.ndarray:
    """process_colors.py assign_labels (:69-77): int16 differences, int16 products (they wrap above |181|), int64 sums, first minimum"""
    px = np.ascontiguousarray(img_rgb, np.uint8).reshape(-1, 3).astype(np.int32)
    pal = np.ascontiguousarray(palette_rgb, np.uint8).reshape(-1, 3).astype(np.int32)
    best = None; lab = np.zeros(len(px), np.uint8)
    for k in range(len(pal)):
        d = px - pal[k]
        sq = ((d * d + 32768) & 0xffff) - 32768              # what an int16 multiplication leaves
        dist = sq.sum(axis=1, dtype=np.int64)
        if best is None:
            best = dist
        else:
            upd = dist < best
            lab[upd] = k; best = np.where(upd, dist, best)
    return lab.reshape(np.asarray(img_rgb).shape[:2])


def kmeans_palette_rgb(img_rgb, k: int, samples: int = 200000, seed: int = 1) -> np.ndarray:
    """process_colors.py kmeans_palette (:31-46); cv2.kmeans through this oracle's restatement: parity unpinned"""
    flat = np.ascontiguousarray(img_rgb, np.uint8).reshape(-1, 3)
    if len(flat) > samples:
        flat = flat[np.random.RandomState(seed).choice(len(flat), size=samples, replace=False)]
    centers, _ = kmeans(flat.astype(np.float32), k, attempts=3, max_iter=30, eps=1.0)
    return centers.astype(np.uint8)


def gaussian(img, k: int) -> np.ndarray:
    a = np.ascontiguousarray(img, np.uint8); out = np.empty_like(a)
    if lib().orc_gaussian(_p(a), _p(out), a.shape[0], a.shape[1], k) != 0:
        raise ValueError(f"GaussianBlur kernel size {k} not supported (3, 5, 7)")
    return out


def canny(img, low: int, high: int) -> np.ndarray:
    a = np.ascontiguousarray(img, np.uint8); out = np.empty_like(a)
    lib().orc_canny(_p(a), _p(out), a.shape[0], a.shape[1], int(math.floor(low)), int(math.floor(high)))
    return out


def thin_rot(edges) -> np.ndarray:
    a = np.ascontiguousarray(edges, np.uint8); out = np.empty_like(a)
    lib().orc_thin_rot(_p(a), _p(out), a.shape[0], a.shape[1])
    return out


def zs_std(img, max_iter=48) -> np.ndarray:
    a = np.ascontiguousarray(img, np.uint8); out = np.empty_like(a)
    if a.size:
        lib().orc_zs_std(_p(a), _p(out), a.shape[0], a.shape[1], max_iter)
    return out


def ccl8(fg) -> Tuple[int, np.ndarray]:
    a = np.ascontiguousarray((np.asarray(fg) > 0).astype(np.uint8)); lab = np.zeros(a.shape, np.int32)
    n = lib().orc_ccl8(_p(a), _p(lab), a.shape[0], a.shape[1])
    return n + 1, lab


def trace(skel) -> List[np.ndarray]:
    a = np.ascontiguousarray(skel, np.uint8); out = PL()
    lib().orc_trace(_p(a), a.shape[0], a.shape[1], out.h)
    return out.get()


# ------------------------------------------------------------------ vector primitives
def params08(**kw) -> np.ndarray:
    d = dict(tap_diam=60, tap_max_dim=25, min_keep=12, tap_max_per=160, tap_max_v=50, sample_step=8, tail_len_px=120,
             col_rad=18, grid_stride=18, max_jump=80, post_on=1, post_brush=16, post_step=6, post_eps=1.28,
             post_minlen=32, W=8400, H=11880, brush_forbid=36)
    d.update(kw)
    return np.array(list(d.values()), np.float64)


def params10(**kw) -> np.ndarray:
    d = dict(tap_diam=60, min_keep=12, tap_max_per=150, tap_max_v=50, max_jump=80, D_lines=120, D_taps=120, step_px=1.0,
             W=8400, H=11880)
    d.update(kw)
    return np.array(list(d.values()), np.float64)


def stamp_capsule(mask, x0, y0, x1, y1, r):
    assert mask.flags.c_contiguous and mask.dtype == np.uint8
    lib().orc_stamp_capsule(_p(mask), mask.shape[0], mask.shape[1], int(x0), int(y0), int(x1), int(y1), int(r))


def arc_length(poly, closed: bool) -> float:
    a = _i32xy(poly)
    return float(lib().orc_arc_length(_p(a), len(a), int(closed)))


def poly_perimeter(poly) -> float:
    a = _i32xy(poly)
    return float(lib().orc_poly_perimeter(_p(a), len(a)))


def min_enclosing_circle(pts_f32) -> Tuple[Tuple[float, float], float]:
    a = np.ascontiguousarray(np.asarray(pts_f32).reshape(-1, 2), np.float32); o = np.zeros(3, np.float32)
    lib().orc_mec(_p(a), len(a), _p(o))
    return (float(o[0]), float(o[1])), float(o[2])


def scale(polys, sx, sy, dx, dy) -> List[np.ndarray]:
    i, o = PL(polys), PL()
    lib().orc_scale(i.h, np.float32(sx), np.float32(sy), np.float32(dx), np.float32(dy), o.h)
    return o.get()


def sort07(polys) -> List[np.ndarray]:
    i, o = PL(polys), PL(); lib().orc_sort07(i.h, o.h); return o.get()


def reorder(polys, kind: int) -> List[np.ndarray]:
    i, o = PL(polys), PL(); lib().orc_reorder(i.h, o.h, kind); return o.get()


def resample_arclen(pts_f32, closed: bool, step: float):
    a = np.ascontiguousarray(np.asarray(pts_f32).reshape(-1, 2), np.float32)
    cap = 16
    while True:
        out = np.zeros((cap, 2), np.float64); ps = C.c_int(0)
        m = lib().orc_resample(_p(a), len(a), int(closed), float(step), _p(out), cap, C.byref(ps))
        if m <= cap:
            r = out[:m]
            return (r.astype(np.float32) if ps.value else r), bool(ps.value)
        cap = int(m)


def split_jumps(poly, max_jump: float, variant: int) -> List[np.ndarray]:
    a = _i32xy(poly); o = PL(); lib().orc_split_jumps(_p(a), len(a), float(max_jump), variant, o.h); return o.get()


def split_small_taps08(polys, prm=None):
    prm = params08() if prm is None else prm
    i, k, t = PL(polys), PL(), TL(); lib().orc_split_small_taps08(i.h, _p(prm), k.h, t.h); return k.get(), t.get()


def virtual_draw08(poly, mask, prm=None):
    prm = params08(W=mask.shape[1], H=mask.shape[0]) if prm is None else prm
    a = _i32xy(poly); o = PL()
    assert mask.flags.c_contiguous and mask.dtype == np.uint8
    lib().orc_virtual_draw08(_p(a), len(a), _p(prm), _p(mask), o.h)
    return o.get()


def cluster_by_overlap(bboxes) -> List[List[int]]:
    b = np.ascontiguousarray(np.asarray(bboxes, np.int32).reshape(-1, 4)); g = np.zeros(len(b), np.int32)
    if len(b) == 0:
        return []
    lib().orc_cluster_by_overlap(_p(b), len(b), _p(g))
    out: List[List[int]] = [[] for _ in range(int(g.max()) + 1)]
    for i, gi in enumerate(g):
        out[gi].append(i)
    return out


def bfs_path(img, start, goal) -> List[Tuple[int, int]]:
    a = np.ascontiguousarray((np.asarray(img) > 0).astype(np.uint8)); cap = a.size + 1
    out = np.zeros((cap, 2), np.int32)
    m = lib().orc_bfs_path(_p(a), a.shape[0], a.shape[1], start[0], start[1], goal[0], goal[1], _p(out), cap)
    return [(int(y), int(x)) for y, x in out[:m]]


def component_best_path(comp, a, b, min_len) -> List[Tuple[int, int]]:
    im = np.ascontiguousarray((np.asarray(comp) > 0).astype(np.uint8)); cap = im.size + 1
    out = np.zeros((cap, 2), np.int32)
    ha, hb = a is not None, b is not None
    ay, ax = a if ha else (0, 0); by, bx = b if hb else (0, 0)
    m = lib().orc_component_best_path(_p(im), im.shape[0], im.shape[1], int(ha), ay, ax, int(hb), by, bx, int(min_len), _p(out), cap)
    return [(int(y), int(x)) for y, x in out[:m]]


def post_skeleton_merge(lines, prm=None):
    prm = params08() if prm is None else prm
    i, o = PL(lines), PL(); lib().orc_post_skeleton_merge(i.h, _p(prm), o.h); return o.get()


def stage08_layer(sorted_contours, prm=None):
    prm = params08() if prm is None else prm
    i, l, t = PL(sorted_contours), PL(), TL(); lib().orc_stage08(i.h, _p(prm), l.h, t.h); return l.get(), t.get()


def cut_poly(poly, forb, step=1.0):
    a = _i32xy(poly); o = PL(); f = np.ascontiguousarray(forb, np.uint8)
    lib().orc_cut_poly(_p(a), len(a), _p(f), f.shape[0], f.shape[1], float(step), o.h); return o.get()


def tiny_and_taps10(polys, prm=None):
    prm = params10() if prm is None else prm
    i, k, t = PL(polys), PL(), TL(); lib().orc_tiny_and_taps10(i.h, _p(prm), k.h, t.h); return k.get(), t.get()


def stage10_layer(lines_in, taps_in, forbidden, prm=None):
    prm = params10(W=forbidden.shape[1], H=forbidden.shape[0]) if prm is None else prm
    assert forbidden.flags.c_contiguous and forbidden.dtype == np.uint8
    li, ti, lo, to = PL(lines_in), TL(taps_in), PL(), TL()
    lib().orc_stage10_layer(li.h, ti.h, _p(prm), _p(forbidden), lo.h, to.h)
    return lo.get(), to.get()


def build_ops12(lines, taps, R_insert=80.0) -> List[dict]:
    li, ti = PL(lines), TL(taps)
    cap = len(lines) + len(taps) + 1
    out = np.zeros((cap, 5), np.int32)
    m = lib().orc_build_ops12(li.h, ti.h, float(R_insert), _p(out), cap)
    ops = []
    for t, idx, flip, x, y in out[:m]:
        if t == 0:
            p = np.asarray(lines[idx]).reshape(-1, 2).astype(np.float32)
            ops.append({"type": "line", "points": p[::-1].copy() if flip else p})
        else:
            ops.append({"type": "tap", "x": int(x), "y": int(y)})
    return ops


# ------------------------------------------------------------------ stage-level composition
def preview_cover(polys, taps, W: int, H: int, thickness: int = 1, radius: int = 30, antialias: bool = True):
    """Coverage planes behind the previews 06 / 09 / 11 (06:76-88 _draw_layer, 09:71-88 _draw_lines / _draw_taps): (lines, taps), uint8 H x W,
    0 = untouched .. 255 = fully covered.  PARITY UNPINNED against cv2.polylines / cv2.circle with LINE_AA (the reference's tests hold nothing for
    them and cv2 is not importable here): what is drawn is the documented stand-in of csrc/vector_preview.hip -- coverage of a pixel centre by a
    segment clamp(thickness / 2 + 0.5 - d, 0, 1) with d the Euclidean distance to the segment (anti-aliasing off: d <= thickness / 2), by a tap
    clamp(radius + 0.5 - d, 0, 1); per pixel the largest coverage, stored as floor(255 a + 0.5).  float64 throughout, same order of operations as the
    kernel (this function is what the GPU tests compare with, bit for bit)."""
    lines = np.zeros((H, W), np.int32); discs = np.zeros((H, W), np.int32)
    half = np.float64(thickness) * 0.5
    def cov(a):
        return np.floor(np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.int32)
    for poly in polys:
        p = np.asarray(poly).reshape(-1, 2).astype(np.int64)
        r = int(np.ceil(half + 0.5))
        for k in range(len(p) - 1):
            x0, y0 = int(p[k, 0]), int(p[k, 1]); x1, y1 = int(p[k + 1, 0]), int(p[k + 1, 1])
            xa, xb = max(0, min(x0, x1) - r), min(W - 1, max(x0, x1) + r); ya, yb = max(0, min(y0, y1) - r), min(H - 1, max(y0, y1) + r)
            if xa > xb or ya > yb:
                continue
            ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1]
            vx, vy = np.float64(x1 - x0), np.float64(y1 - y0)
            wx, wy = xs.astype(np.float64) - np.float64(x0), ys.astype(np.float64) - np.float64(y0)
            L2 = vx * vx + vy * vy
            t = np.clip((wx * vx + wy * vy) / L2, 0.0, 1.0) if L2 > 0 else np.zeros_like(wx)
            dx, dy = wx - t * vx, wy - t * vy
            d = np.sqrt(dx * dx + dy * dy)
            A = cov((half + 0.5) - d) if antialias else np.where(d <= half, 255, 0).astype(np.int32)
            np.maximum(lines[ya:yb + 1, xa:xb + 1], A, out=lines[ya:yb + 1, xa:xb + 1])
    rad = np.float64(radius); r = int(np.ceil(rad + 0.5))
    for (cx, cy) in taps:
        cx, cy = int(cx), int(cy)
        xa, xb = max(0, cx - r), min(W - 1, cx + r); ya, yb = max(0, cy - r), min(H - 1, cy + r)
        if xa > xb or ya > yb:
            continue
        ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1]
        dx, dy = xs.astype(np.float64) - np.float64(cx), ys.astype(np.float64) - np.float64(cy)
        d = np.sqrt(dx * dx + dy * dy)
        A = cov((rad + 0.5) - d) if antialias else np.where(d <= rad, 255, 0).astype(np.int32)
        np.maximum(discs[ya:yb + 1, xa:xb + 1], A, out=discs[ya:yb + 1, xa:xb + 1])
    return lines.astype(np.uint8), discs.astype(np.uint8)


def preview_compose(img: np.ndarray, cover: np.ndarray, bgr) -> np.ndarray:
    """img (H, W, 3) uint8 with `bgr` laid over it at coverage `cover` (H, W) uint8: (img (255 - A) + colour A + 127) // 255 per channel."""
    A = cover.astype(np.int32)[:, :, None]
    col = np.asarray(bgr, np.int32).reshape(1, 1, 3)
    return ((img.astype(np.int32) * (255 - A) + col * A + 127) // 255).astype(np.uint8)


DEFAULTS = dict(
    color_names=["layer_dark", "layer_mid", "layer_skin", "layer_light"],
    edge_low_threshold=50, edge_high_threshold=150, edge_kernel_size=3, edge_morph_kernel=3,
    edge_morph_open_iters=1, edge_morph_close_iters=1,
    target_width_mm=210, target_height_mm=297, pixels_per_mm=40,
    margin_left_mm=10.0, margin_right_mm=10.0, margin_top_mm=10.0, margin_bottom_mm=10.0,
    pen_width_px=60, pen_radius_px=30, tap_max_perimeter=160.0, tap_max_dim=25, dedup_sample_step=8,
    ignore_tail_points_intra=120, collision_radius_intra_px=18.0, hash_stride_px=18.0, max_join_jump_px=80.0,
)


def _cfg(cfg: dict | None) -> dict:
    d = dict(DEFAULTS)
    if cfg:
        d.update(cfg)
    return d


def darkness_rank02(name: str) -> int:  # 02:17-23
    s = name.lower()
    if "dark" in s: return 0
    if "mid" in s: return 1
    if "skin" in s: return 2
    if "light" in s: return 3
    return 2


def subsample_indices(n: int, limit: int = 200_000):  # 02:39-44
    if n > limit:
        return np.random.default_rng(42).choice(n, size=limit, replace=False)
    return None


def stage02(bgr: np.ndarray, cfg=None, centers=None):
    """02_color_extract.py main(), k-means mode.  Returns (masks{name:u8[H,W]}, centers_sorted[K,3], labels[H,W])."""
    cfg = _cfg(cfg)
    names = list(cfg["color_names"])
    K = max(2, len(names))
    if bgr.ndim == 2:
        bgr = np.repeat(bgr[:, :, None], 3, axis=2)
    h, w = bgr.shape[:2]
    lab = bgr2lab(bgr)
    if centers is None:
        data = lab.reshape(-1, 3).astype(np.float32)
        idx = subsample_indices(len(data))
        sample = data[idx] if idx is not None else data
        centers, _ = kmeans(sample, K)
    centers = np.asarray(centers, np.float32)
    labels = assign(lab, centers)
    order = np.argsort(centers[:, 0], kind="stable")
    centers_sorted = centers[order]
    lut = np.zeros(len(order), np.int64); lut[order] = np.arange(len(order))
    labels = lut[labels]
    names_sorted = sorted(names, key=darkness_rank02)
    masks: Dict[str, np.ndarray] = {}
    for name, k in zip(names_sorted, range(K)):
        m = (labels == k).astype(np.uint8) * 255
        masks[name] = morph_open_close(m, 0, 3, 1, 1)
    return masks, centers_sorted, labels.astype(np.int32)


def ensure_odd(n: int) -> int:  # 03:9-11
    n = max(3, int(n))
    return n if n % 2 == 1 else n + 1


def stage03(mask: np.ndarray, cfg=None) -> np.ndarray:
    cfg = _cfg(cfg)
    k_m = max(1, int(cfg["edge_morph_kernel"]))
    m = morph_open_close(mask, 2, k_m, int(cfg["edge_morph_open_iters"]), int(cfg["edge_morph_close_iters"]))
    b = gaussian(m, ensure_odd(cfg["edge_kernel_size"]))
    return canny(b, cfg["edge_low_threshold"], cfg["edge_high_threshold"])


def stage04(edges: np.ndarray) -> List[np.ndarray]:
    paths = trace(thin_rot(edges))
    return [p for p in paths if len(p) >= 5]


def canvas_size(cfg) -> Tuple[int, int]:
    return int(round(cfg["target_width_mm"] * cfg["pixels_per_mm"])), int(round(cfg["target_height_mm"] * cfg["pixels_per_mm"]))


def scale_factors(w_src: int, h_src: int, cfg=None):
    cfg = _cfg(cfg)
    w_full, h_full = canvas_size(cfg)
    ppm = int(cfg["pixels_per_mm"])
    ml, mr, mt, mb = (max(0, int(round(float(cfg[k]) * ppm))) for k in ("margin_left_mm", "margin_right_mm", "margin_top_mm", "margin_bottom_mm"))
    inner_w = max(1, w_full - ml - mr); inner_h = max(1, h_full - mt - mb)
    s = min(inner_w / max(1e-6, w_src), inner_h / max(1e-6, h_src))
    return s, s, ml, mt


def stage05(contours, w_src, h_src, cfg=None):
    sx, sy, dx, dy = scale_factors(w_src, h_src, cfg)
    return scale(contours, sx, sy, dx, dy)


def derived08(cfg=None) -> np.ndarray:  # 08:484-509
    cfg = _cfg(cfg)
    pen_diam = float(cfg["pen_width_px"]); pen_radius = float(cfg["pen_radius_px"])
    W, H = canvas_size(cfg)
    col_rad = float(cfg["collision_radius_intra_px"])
    post_brush = 16
    return params08(tap_diam=pen_diam, tap_max_dim=float(cfg["tap_max_dim"]), min_keep=max(10.0, pen_radius * 0.4),
                    tap_max_per=float(cfg["tap_max_perimeter"]), tap_max_v=50, sample_step=float(cfg["dedup_sample_step"]),
                    tail_len_px=float(cfg["ignore_tail_points_intra"]), col_rad=col_rad, grid_stride=float(cfg["hash_stride_px"]),
                    max_jump=float(cfg["max_join_jump_px"]), post_on=1, post_brush=post_brush, post_step=6.0,
                    post_eps=max(1.0, 0.08 * post_brush), post_minlen=max(2 * post_brush, 12), W=W, H=H,
                    brush_forbid=max(1, int(round(2.0 * col_rad))))


def derived10(cfg=None) -> np.ndarray:  # 10:217-229
    cfg = _cfg(cfg)
    pen_diam = float(cfg["pen_width_px"]); W, H = canvas_size(cfg)
    return params10(tap_diam=pen_diam, min_keep=max(10.0, (pen_diam / 2.0) * 0.4), tap_max_per=2.5 * pen_diam, tap_max_v=50,
                    max_jump=float(cfg["max_join_jump_px"]), D_lines=pen_diam * 2.0, D_taps=pen_diam * 2.0, step_px=1.0, W=W, H=H)


def darkness_rank10(name: str) -> int:  # 10:206-208
    order = ["layer_dark", "layer_mid", "layer_skin", "layer_light"]
    return order.index(name) if name in order else 999


def stage10(intra: Dict[str, Tuple[list, list]], cfg=None):
    cfg = _cfg(cfg)
    prm = derived10(cfg); W, H = canvas_size(cfg)
    forbidden = np.zeros((H, W), np.uint8)
    names = sorted(list(cfg["color_names"]), key=darkness_rank10)
    out = {}
    for name in names:
        lines_in, taps_in = intra.get(name, ([], []))
        out[name] = stage10_layer(lines_in, taps_in, forbidden, prm)
    return out


def stage12(lines, taps, cfg=None):
    cfg = _cfg(cfg)
    return build_ops12(lines, taps, max(80.0, cfg["pen_width_px"]))


def run_pipeline(bgr: np.ndarray, cfg=None, centers=None, upto: int = 12, threads: int = 1):
    """Stages 02 -> `upto` on one image, returning every intermediate artefact.  threads > 1: the per-layer stages 03 -> 08 of
    different layers run in that many threads (the reference's only parallelism is its pool over layers in stage 03, 03:42-48; the
    C++ calls release the GIL); stages 02, 10 and 12 stay serial as in the reference."""
    cfg = _cfg(cfg)
    names = list(cfg["color_names"])
    h, w = bgr.shape[:2]
    res = {}
    masks, cs, labels = stage02(bgr, cfg, centers)
    res.update(masks=masks, centers=cs, labels=labels)
    if upto < 3: return res
    prm08 = derived08(cfg)

    def front(n):
        out = {"edges": stage03(masks[n], cfg)}
        if upto >= 4: out["contours"] = stage04(out["edges"])
        if upto >= 5: out["scaled"] = stage05(out["contours"], w, h, cfg)
        if upto >= 7: out["sorted"] = sort07(out["scaled"])
        if upto >= 8: out["intra"] = stage08_layer(out["sorted"], prm08)
        return out

    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as pool:
            fronts = dict(zip(names, pool.map(front, names)))
    else:
        fronts = {n: front(n) for n in names}
    for key in ("edges", "contours", "scaled", "sorted", "intra"):
        if key in fronts[names[0]]:
            res[key] = {n: fronts[n][key] for n in names}
    if upto < 10: return res
    res["cross"] = stage10(res["intra"], cfg)
    if upto < 12: return res
    res["ops"] = {n: stage12(res["cross"][n][0], res["cross"][n][1], cfg) for n in names}
    return res


def path_length(ops_by_layer: Dict[str, List[dict]]) -> Tuple[float, float]:
    """SURVEY 8(a) parity quantity (iii): (draw, travel) length in px, per 12:71-80 arithmetic."""
    draw = travel = 0.0
    for ops in ops_by_layer.values():
        pos = (0.0, 0.0)
        for o in ops:
            if o["type"] == "line":
                p = np.asarray(o["points"], np.float32)
                travel += math.hypot(pos[0] - float(p[0, 0]), pos[1] - float(p[0, 1]))
                d = p[1:] - p[:-1]
                draw += float(np.sum(np.hypot(d[:, 0], d[:, 1])))
                pos = (float(p[-1, 0]), float(p[-1, 1]))
            else:
                travel += math.hypot(pos[0] - o["x"], pos[1] - o["y"])
                pos = (float(o["x"]), float(o["y"]))
    return draw, travel
