"""In-process stage functions with the reference's per-stage inputs / outputs:

    extract_colors(bgr, cfg)            -> {name: mask u8[H,W]}, palette            (02_color_extract.py main)
    detect_edges(masks, cfg)            -> {name: edges u8[H,W]}                   (03_edge_detect.py process_color)
    find_contours(edges, cfg)           -> {name: [int32 (N,1,2)]}                 (04_find_contours.py vectorize_layer)
    scale_vectors / sort_contours / dedup_layer / dedup_cross / plot_order         (05, 07, 08, 10, 12)
    run_path(bgr, cfg)                  -> ops per layer, everything resident on the GPU between stages

Every function drives liborip.so through orip.device.Device; arrays cross the host boundary only where the
reference's stage API puts a file.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import os

import numpy as np

from . import lib as _l
from .config import Config, canvas_size_px, scale_factors
from .device import Device

_dev: Device | None = None


def device(device_id: int = 0) -> Device:
    global _dev
    if _dev is None:
        _dev = Device(device_id)
    return _dev


# ---------------------------------------------------------------- naming rules of the reference
def darkness_rank02(name: str) -> int:  # 02:17-23
    s = name.lower()
    if "dark" in s: return 0
    if "mid" in s: return 1
    if "skin" in s: return 2
    if "light" in s: return 3
    return 2


def darkness_rank10(name: str) -> int:  # 10:206-208
    order = ["layer_dark", "layer_mid", "layer_skin", "layer_light"]
    return order.index(name) if name in order else 999


def color_index12(name: str) -> int:  # 12:210-219
    if "dark" in name: return 3
    if "skin" in name: return 0
    if "mid" in name: return 1
    if "light" in name: return 2
    return 0


def cluster_names(cfg: Config) -> List[str]:
    """i-th darkest cluster <-> i-th name of sorted(names, key=_darkness_rank) (02:130-133); device layer l = cluster l."""
    return sorted(list(cfg.color_names), key=darkness_rank02)


def subsample_indices(n: int, limit: int = 200_000):  # 02:39-44
    if n > limit:
        return np.random.default_rng(42).choice(n, size=limit, replace=False)
    return None


def ensure_odd(n: int) -> int:  # 03:9-11
    n = max(3, int(n))
    return n if n % 2 == 1 else n + 1


def lab8_to_bgr(lab_u8) -> Tuple[int, int, int]:
    """_lab_to_bgr (02:58-61), the palette's approx_bgr: 8-bit Lab (L*255/100, a+128, b+128) -> BGR.  Host arithmetic on K triples
    (a by-product for the previews): CIE L*a*b* -> XYZ (D65 white 0.950456 / 1 / 1.088754) -> linear sRGB -> gamma.
    cv2 is not available to pin the last bit (parity unpinned); tests check it against the oracle's forward Lab tables."""
    L = float(lab_u8[0]) * 100.0 / 255.0
    fy = (L + 16.0) / 116.0
    f = (fy + (float(lab_u8[1]) - 128.0) / 500.0, fy, fy - (float(lab_u8[2]) - 128.0) / 200.0)
    xyz = []
    for t, white in zip(f, (0.950456, 1.0, 1.088754)):
        cube = t * t * t
        xyz.append(white * (cube if cube > 0.008856 else (t - 16.0 / 116.0) / 7.787))
    X, Y, Z = xyz
    out = []
    for m in ((0.055648, -0.204043, 1.057311), (-0.969256, 1.875991, 0.041556), (3.240479, -1.53715, -0.498535)):   # B, G, R rows
        c = m[0] * X + m[1] * Y + m[2] * Z
        c = 12.92 * c if c <= 0.0031308 else 1.055 * (max(c, 0.0) ** (1.0 / 2.4)) - 0.055
        out.append(int(min(255, max(0, round(c * 255.0)))))
    return out[0], out[1], out[2]


# ---------------------------------------------------------------- derived parameter blocks
def params08(cfg: Config) -> _l.Params08:  # 08:484-509
    pen_diam = float(cfg.pen_width_px); pen_radius = float(cfg.pen_radius_px)
    W, H = canvas_size_px(cfg)
    col_rad = float(cfg.collision_radius_intra_px)
    post_brush = 16
    return _l.Params08(tap_diam=pen_diam, tap_max_dim=float(cfg.tap_max_dim), min_keep=max(10.0, pen_radius * 0.4),
                       tap_max_per=float(cfg.tap_max_perimeter), tap_max_v=50, sample_step=float(cfg.dedup_sample_step),
                       tail_len_px=float(cfg.ignore_tail_points_intra), col_rad=col_rad, grid_stride=float(cfg.hash_stride_px),
                       max_jump=float(cfg.max_join_jump_px), post_on=1, post_brush=post_brush, post_step=6.0,
                       post_eps=max(1.0, 0.08 * post_brush), post_minlen=max(2 * post_brush, 12), W=W, H=H,
                       brush_forbid=max(1, int(round(2.0 * col_rad))))


def params10(cfg: Config) -> _l.Params10:  # 10:217-229
    pen_diam = float(cfg.pen_width_px); W, H = canvas_size_px(cfg)
    return _l.Params10(tap_diam=pen_diam, min_keep=max(10.0, (pen_diam / 2.0) * 0.4), tap_max_per=2.5 * pen_diam, tap_max_v=50,
                       max_jump=float(cfg.max_join_jump_px), D_lines=pen_diam * 2.0, D_taps=pen_diam * 2.0, step_px=1.0, W=W, H=H)


def r_insert12(cfg: Config) -> float:  # 12:197
    return float(max(80.0, cfg.pen_width_px))


# ---------------------------------------------------------------- stage functions (host arrays in / out)
def resized_size(h: int, w: int, max_dimension: int):
    """01:12-18 -- None when the longest side already fits, else (new_w, new_h) with Python's float scale and int() truncation"""
    m = max(h, w)
    if m <= max_dimension:
        return None
    scale = max_dimension / m
    return int(w * scale), int(h * scale)


def resize_if_needed(img: np.ndarray, cfg: Config, dev: Device | None = None, as_image: bool = False) -> np.ndarray:
    """01 resize_if_needed (:7-23): the image itself when it fits, else its INTER_AREA shrink; as_image also leaves it resident for stage 02"""
    size = resized_size(img.shape[0], img.shape[1], int(cfg.max_dimension))
    if size is None:
        return img
    return (dev or device()).resize_area(img, size[0], size[1], as_image=as_image)


def extract_colors(bgr: np.ndarray, cfg: Config, centers: np.ndarray | None = None, dev: Device | None = None):
    """02 main(), k-means mode.  Returns (masks {name: u8[H,W]}, info) with info = centres (dark->light), counts, labels."""
    d = dev or device()
    names = list(cfg.color_names)
    K = max(2, len(names))
    d.set_image(bgr)
    if centers is None:
        idx = subsample_indices(d.H * d.W)
        centers, _ = d.kmeans_fit(idx, K)
    cs, counts = d.extract_layers(np.asarray(centers, np.float32))
    masks = {n: d.get_mask(l) for l, n in enumerate(cluster_names(cfg)[:K])}
    return masks, {"centers_lab": cs, "counts": counts, "centers_fit": np.asarray(centers, np.float32)}


def detect_edges(masks: Dict[str, np.ndarray], cfg: Config, dev: Device | None = None) -> Dict[str, np.ndarray]:
    d = dev or device()
    names = list(masks.keys())
    d.set_masks(np.stack([masks[n] for n in names]))
    _detect_edges_resident(d, cfg)
    return {n: d.get_edges(l) for l, n in enumerate(names)}


def _detect_edges_resident(d: Device, cfg: Config):
    d.detect_edges(max(1, int(cfg.edge_morph_kernel)), int(cfg.edge_morph_open_iters), int(cfg.edge_morph_close_iters),
                   ensure_odd(cfg.edge_kernel_size), int(math.floor(cfg.edge_low_threshold)), int(math.floor(cfg.edge_high_threshold)))


def find_contours(edges: Dict[str, np.ndarray], cfg: Config | None = None, dev: Device | None = None) -> Dict[str, List[np.ndarray]]:
    d = dev or device()
    names = list(edges.keys())
    d.set_edges(np.stack([edges[n] for n in names]))
    d.find_contours()
    return {n: d.get_polys(_l.SLOT_CONTOURS, l) for l, n in enumerate(names)}


def scale_vectors(contours: Sequence[np.ndarray], w_src: int, h_src: int, cfg: Config, dev: Device | None = None, layer: int = 0):
    d = dev or device()
    d.set_polys(_l.SLOT_CONTOURS, layer, contours)
    sx, sy, dx, dy = scale_factors(cfg, w_src, h_src)
    d.scale_vectors(layer, sx, sy, dx, dy)
    return d.get_polys(_l.SLOT_SCALED, layer)


def sort_contours(contours: Sequence[np.ndarray], dev: Device | None = None, layer: int = 0):
    d = dev or device()
    d.set_polys(_l.SLOT_SCALED, layer, contours)
    d.sort_contours(layer)
    return d.get_polys(_l.SLOT_SORTED, layer)


def dedup_layer(contours_sorted: Sequence[np.ndarray], cfg: Config, dev: Device | None = None, layer: int = 0):
    d = dev or device()
    d.set_polys(_l.SLOT_SORTED, layer, contours_sorted)
    d.dedup_layer(layer, params08(cfg))
    return d.get_polys(_l.SLOT_LINES_INTRA, layer), d.get_taps(_l.TAPS_INTRA, layer)


def dedup_cross(intra: Dict[str, Tuple[Sequence[np.ndarray], Sequence[Tuple[int, int]]]], cfg: Config, dev: Device | None = None):
    d = dev or device()
    names = list(cfg.color_names)
    for l, n in enumerate(names):
        lines, taps = intra.get(n, ([], []))
        d.set_polys(_l.SLOT_LINES_INTRA, l, lines)
        d.set_taps(_l.TAPS_INTRA, l, taps)
    order = sorted(range(len(names)), key=lambda l: darkness_rank10(names[l]))
    d.dedup_cross(order, params10(cfg))
    return {n: (d.get_polys(_l.SLOT_LINES_CROSS, l), d.get_taps(_l.TAPS_CROSS, l)) for l, n in enumerate(names)}


# ---- previews 06 / 09 / 11 (visual QA; parity unpinned: include/orip.h orip_preview_cover says what is drawn instead of cv2.LINE_AA) ----
def preview_cover(polys: Sequence[np.ndarray], taps: Sequence[Tuple[int, int]], size: Tuple[int, int], thickness: int, radius: int, antialias: bool,
                  dev: Device | None = None, layer: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Coverage planes (lines, taps), uint8 (H, W), of one layer's polylines and taps on the full canvas `size` = (W, H)."""
    d = dev or device()
    d.set_polys(_l.SLOT_LINES_CROSS, layer, polys)
    d.set_taps(_l.TAPS_CROSS, layer, taps)
    return d.preview_cover(_l.SLOT_LINES_CROSS, layer, _l.TAPS_CROSS, size[0], size[1], max(1, int(thickness)), max(0, int(radius)), bool(antialias))


def preview_compose(img: np.ndarray, cover: np.ndarray, bgr) -> np.ndarray:
    """`bgr` laid over img (H, W, 3) uint8 at coverage `cover` (H, W) uint8: (img (255 - A) + colour A + 127) // 255 per channel."""
    A = cover.astype(np.int32)[:, :, None]
    col = np.asarray(bgr, np.int32).reshape(1, 1, 3)
    return ((img.astype(np.int32) * (255 - A) + col * A + 127) // 255).astype(np.uint8)


def preview_images(polys, taps, size, thickness: int, radius: int, antialias: bool, color, tap_color_layer=(0, 0, 255), dev: Device | None = None):
    """One layer of a preview stage: (layer image, colour image) as 09:103-118 builds them -- black lines and `tap_color_layer` discs on white for
    <layer>/preview_*.png, lines and discs in the layer's colour for the composite (06 has no taps: pass [])."""
    cov_l, cov_t = preview_cover(polys, taps, size, thickness, radius, antialias, dev)
    white = np.full((size[1], size[0], 3), 255, np.uint8)
    lay = preview_compose(preview_compose(white, cov_l, (0, 0, 0)), cov_t, tap_color_layer)
    col = preview_compose(preview_compose(white, cov_l, color), cov_t, color)
    return lay, col


def preview_palette(cfg: Config) -> Dict[str, Tuple[int, int, int]]:
    """_load_palette_by_name (06:46-62) / _palette (09:27-42): approx_bgr of palette_by_name.json, else cfg.colors[i]."""
    import json
    import os
    data = None
    path = os.path.join(cfg.output_dir, "palette_by_name.json")
    if os.path.exists(path):
        try:
            with open(path, "r", encoding="utf-8") as f:
                data = json.load(f)
        except Exception:
            data = None
    out = {}
    for i, n in enumerate(cfg.color_names):
        if data and n in data and "approx_bgr" in data[n]:
            b, g, r = data[n]["approx_bgr"]
        else:
            b, g, r = cfg.colors[i]
        out[n] = (int(b), int(g), int(r))
    return out


def ops_from_device(d: Device, layer: int, R: float) -> List[dict]:
    raw = d.plot_order(layer, R)
    lines = d.get_polys(_l.SLOT_LINES_CROSS, layer) if len(raw) else []
    ops = []
    for t, idx, flip, x, y in raw:
        if t == 0:
            p = np.asarray(lines[idx]).reshape(-1, 2).astype(np.float32)
            ops.append({"type": "line", "points": p[::-1].copy() if flip else p})
        else:
            ops.append({"type": "tap", "x": int(x), "y": int(y)})
    return ops


def plot_order(lines: Sequence[np.ndarray], taps: Sequence[Tuple[int, int]], cfg: Config, dev: Device | None = None, layer: int = 0) -> List[dict]:
    d = dev or device()
    d.set_polys(_l.SLOT_LINES_CROSS, layer, lines)
    d.set_taps(_l.TAPS_CROSS, layer, taps)
    return ops_from_device(d, layer, r_insert12(cfg))


# ---------------------------------------------------------------- layer concurrency
_pool = None


def for_each_layer(fn, layers):
    """Run fn(layer) for every layer from its own host thread.  The per-layer entry points of liborip.so (05, 07, 08, 12)
    use one HIP stream and one scratch set per layer and ctypes drops the GIL during the call, so the latency-bound
    kernels of different layers overlap on the GPU (the reference loops over layers serially: 05:114, 07:99, 08:561, 12:200)."""
    global _pool
    layers = list(layers)
    if len(layers) <= 1 or os.environ.get("ORIP_SERIAL_LAYERS"):      # the switch is a profiling aid: kernels of one layer at a time
        return [fn(l) for l in layers]
    return list(_get_pool().map(fn, layers))


def _get_pool():
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=_l.MAX_LAYERS + 2)
    return _pool


# ---------------------------------------------------------------- the resident end-to-end path
def run_path(bgr: np.ndarray, cfg: Config, dev: Device | None = None, centers: np.ndarray | None = None, upto: int = 12,
             fetch_ops: bool = True):
    """Stages 02 -> `upto` with every intermediate artefact resident on the GPU.  Device layer l carries the l-th darkest
    cluster, i.e. the name cluster_names(cfg)[l].  Returns {name: ops} (or None when upto < 12 / fetch_ops False)."""
    d = dev or device()
    names = list(cfg.color_names)
    K = max(2, len(names))
    lnames = cluster_names(cfg)[:K]
    H, W = bgr.shape[:2]
    d.set_image(bgr)
    if upto >= 5:
        d.contours_reserve(K)
    if centers is None:
        centers, _ = d.kmeans_fit(subsample_indices(H * W), K)
    d.extract_layers(np.asarray(centers, np.float32), want_counts=False)
    if upto < 3: return None
    _detect_edges_resident(d, cfg)
    if upto < 4: return None
    if upto < 5:
        d.find_contours()
        return None
    order = sorted(range(K), key=lambda l: (darkness_rank10(lnames[l]), names.index(lnames[l])))
    R = r_insert12(cfg)
    tail = None
    if upto >= 12:
        tail = (lambda l: ops_from_device(d, l, R)) if fetch_ops else (lambda l: d.plot_order(l, R))
    res = run_layer_pipelines(d, cfg, W, H, range(K), order if upto >= 10 else None, upto, tail)
    if upto < 12 or not fetch_ops:
        return None
    return {lnames[l]: res[l] for l in range(K)}


def run_layer_pipelines(d: Device, cfg: Config, W: int, H: int, layers, order, upto: int = 12, tail=None) -> dict:
    """Stages 04 -> 12 as one dependency-driven pipeline per layer (edges must be resident).  The reference runs stage after
    stage over all layers (pipeline.py:17-31); the only cross-layer dependency is stage 10, which visits the layers from dark
    to light and needs layer l's stage-08 output plus the raster painted by the layers before it (10:230-262).  So every
    layer walks 04 -> 05 -> 07 -> 08 on its own lane (host thread + HIP stream), the calling thread feeds the layers to
    stage 10 in `order` as they become ready, and `tail(layer)` (stage 12) runs as soon as the layer has left stage 10.
    order None: stop after stage min(upto, 8).  Returns {layer: tail result}."""
    layers = list(layers)
    d.contours_prepare()
    front = layer_front(d, cfg, W, H, upto)
    if order is None or os.environ.get("ORIP_SERIAL_LAYERS"):
        for_each_layer(front, layers)
        out = {}
        if order is not None:
            d.dedup_cross([l for l in order], params10(cfg))
            if tail is not None:
                out = dict(zip(layers, for_each_layer(tail, layers)))
        return out
    ready, errors = _start_fronts(front, layers, order)
    pool = _get_pool()
    d.dedup_cross_begin(params10(cfg))
    tails = {}
    for l in order:
        if l in ready:
            ready[l].wait()
        if errors:
            for e in ready.values(): e.wait()
            raise errors[0]
        d.dedup_cross_layer(l, defer_reorder=tail is not None and l in ready)      # the tail (stage 12) reorders on the layer's own lane
        if tail is not None and l in ready:
            tails[l] = pool.submit(tail, l)
    return {l: f.result() for l, f in tails.items()}


def _start_fronts(front, layers, order=None):
    """Submit front(layer) for every layer to the pool (layers that stage 10 visits first go first).  Returns
    ({layer: Event set when the layer is done or failed}, [exceptions])."""
    import threading
    pool = _get_pool()
    ready = {l: threading.Event() for l in layers}
    errors = []

    def guarded(l):
        try:
            front(l)
        except BaseException as e:      # surfaced on the calling thread
            errors.append(e)
        finally:
            ready[l].set()

    rank_of = (lambda l: order.index(l) if l in order else len(order)) if order else (lambda l: 0)
    for l in sorted(layers, key=rank_of):
        pool.submit(guarded, l)
    return ready, errors


def layer_front(d: Device, cfg: Config, W: int, H: int, upto: int = 8):
    """front(layer): stages 04 (walks of the layer) -> 05 -> 07 -> 08 on the layer's lane; needs d.contours_prepare() first."""
    sx, sy, dx, dy = scale_factors(cfg, W, H)
    p8 = params08(cfg)

    def front(l):
        d.layer_front(l, sx, sy, dx, dy, 8 if upto >= 8 else (7 if upto >= 7 else 5), p8)
    return front
