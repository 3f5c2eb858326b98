"""Device: one orip_ctx (one GPU, one HIP stream) with numpy-array marshalling around the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import lib as _l


class OripError(RuntimeError):
    pass


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Device:
    def __init__(self, device_id: int = 0):
        self.L = _l.load()
        h = C.c_void_p()
        rc = self.L.orip_create(int(device_id), C.byref(h))
        if rc != 0 or not h:
            raise OripError(f"orip_create(device={device_id}) failed with {rc}: no usable MI355X (there is no CPU fallback)")
        self.h = h
        self.H = self.W = self.K = 0

    def close(self):
        cached = getattr(self, "_comm", None)       # communicator of sharded steps (orip.parallel.comm_of)
        if cached is not None and getattr(self, "h", None):
            self._comm = None
            try:
                cached[1].close()
            except Exception:
                pass
        if getattr(self, "h", None):
            self.L.orip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int):
        if rc != 0:
            raise OripError((self.L.orip_last_error(self.h) or b"?").decode())

    def sync(self):
        self._ck(self.L.orip_sync(self.h))

    # ---- profiling hooks (bench.py roofline leg)
    def prof_enable(self, on: bool): self._ck(self.L.orip_prof_enable(self.h, int(on)))
    def prof_reset(self): self._ck(self.L.orip_prof_reset(self.h))

    def prof_get(self, kernel: str) -> Tuple[float, int]:
        ms, n = C.c_double(0), C.c_int64(0)
        self._ck(self.L.orip_prof_get(self.h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- stage 01
    def resize_area(self, img: np.ndarray, new_w: int, new_h: int, as_image: bool = False, fetch: bool = True):
        """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_AREA) for shrinking (01:19); as_image leaves the result as the stage-02 image"""
        img = np.ascontiguousarray(img, np.uint8)
        cn = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((new_h, new_w) if img.ndim == 2 else (new_h, new_w, cn), np.uint8) if fetch else None
        self._ck(self.L.orip_resize_area(self.h, _p(img), img.shape[0], img.shape[1], cn, new_h, new_w, _p(out) if fetch else None, int(as_image)))
        if as_image:
            self.H, self.W = new_h, new_w
        return out

    # ---- stage 02
    def set_image(self, bgr: np.ndarray):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if bgr.ndim == 2:  # _ensure_bgr (02:25-30)
            bgr = np.ascontiguousarray(np.repeat(bgr[:, :, None], 3, axis=2))
        assert bgr.ndim == 3 and bgr.shape[2] == 3
        self.H, self.W = bgr.shape[:2]
        self._img_ref = bgr  # keep alive until the async copy is consumed
        self._ck(self.L.orip_set_image(self.h, _p(bgr), self.H, self.W))
        self.sync()

    def lab_of(self, idx: np.ndarray | None = None) -> np.ndarray:
        if idx is None:
            out = np.empty((self.H, self.W, 3), np.uint8)
            self._ck(self.L.orip_lab_of(self.h, None, 0, _p(out)))
            return out
        idx = np.ascontiguousarray(idx, np.int64)
        out = np.empty((len(idx), 3), np.uint8)
        self._ck(self.L.orip_lab_of(self.h, _p(idx), len(idx), _p(out)))
        return out

    def kmeans_fit(self, sample_idx: np.ndarray | None, K: int, attempts=3, max_iter=40, eps=0.5) -> Tuple[np.ndarray, float]:
        centers = np.zeros((K, 3), np.float32)
        comp = C.c_double(0)
        if sample_idx is None:
            self._ck(self.L.orip_kmeans_fit(self.h, None, 0, K, attempts, max_iter, eps, _p(centers), C.byref(comp)))
        else:
            idx = np.ascontiguousarray(sample_idx, np.int64)
            self._ck(self.L.orip_kmeans_fit(self.h, _p(idx), len(idx), K, attempts, max_iter, eps, _p(centers), C.byref(comp)))
        return centers, comp.value

    # ---- process_colors.py
    def kmeans_fit_rgb(self, sample_idx: np.ndarray | None, K: int, attempts=3, max_iter=30, eps=1.0) -> Tuple[np.ndarray, float]:
        """kmeans_palette's cv2.kmeans (process_colors.py:41-45) over the R, G, B bytes of the sampled pixels; float centres in R, G, B order"""
        centers = np.zeros((K, 3), np.float32)
        comp = C.c_double(0)
        idx = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int64)
        self._ck(self.L.orip_kmeans_fit_rgb(self.h, _p(idx) if idx is not None else None, 0 if idx is None else len(idx), K, attempts, max_iter, eps, _p(centers), C.byref(comp)))
        return centers, comp.value

    def assign_palette(self, palette_rgb: np.ndarray, fetch: bool = True):
        """assign_labels (process_colors.py:69-77): (labels u8 [H,W] or None, pixels per label)"""
        pal = np.ascontiguousarray(palette_rgb, np.uint8).reshape(-1, 3)
        labels = np.empty((self.H, self.W), np.uint8) if fetch else None
        counts = np.zeros(len(pal), np.int64)
        self._ck(self.L.orip_assign_palette(self.h, _p(pal), len(pal), _p(labels) if fetch else None, _p(counts)))
        return labels, counts

    def extract_layers(self, centers: np.ndarray, open_iters=1, close_iters=1, want_counts=True):
        c = np.ascontiguousarray(centers, np.float32)
        K = len(c)
        cs = np.zeros((K, 3), np.float32)
        counts = np.zeros(K, np.int64)
        self._ck(self.L.orip_extract_layers(self.h, _p(c), K, open_iters, close_iters, _p(cs), _p(counts) if want_counts else None))
        self.K = K
        return cs, counts

    def get_labels(self) -> np.ndarray:
        out = np.empty((self.H, self.W), np.uint8)
        self._ck(self.L.orip_get_labels(self.h, _p(out)))
        return out

    def get_mask(self, layer: int) -> np.ndarray:
        out = np.empty((self.H, self.W), np.uint8)
        self._ck(self.L.orip_get_mask(self.h, layer, _p(out)))
        return out

    def set_masks(self, masks: np.ndarray):
        m = np.ascontiguousarray(masks, np.uint8)
        self.K, self.H, self.W = m.shape
        self._ck(self.L.orip_set_masks(self.h, _p(m), self.K, self.H, self.W))
        self.sync()

    def keep_layers(self, layers: Sequence[int]):
        a = np.ascontiguousarray(np.asarray(list(layers), np.int32))
        self._ck(self.L.orip_keep_layers(self.h, _p(a), len(a)))
        self.K = len(a)

    # ---- stage 03
    def detect_edges(self, morph_k=3, open_iters=1, close_iters=1, gauss_k=3, low=50, high=150):
        self._ck(self.L.orip_detect_edges(self.h, morph_k, open_iters, close_iters, gauss_k, int(low), int(high)))

    def get_edges(self, layer: int) -> np.ndarray:
        out = np.empty((self.H, self.W), np.uint8)
        self._ck(self.L.orip_get_edges(self.h, layer, _p(out)))
        return out

    def set_edges(self, edges: np.ndarray):
        e = np.ascontiguousarray(edges, np.uint8)
        self.K, self.H, self.W = e.shape
        self._ck(self.L.orip_set_edges(self.h, _p(e), self.K, self.H, self.W))
        self.sync()

    # ---- stage 04
    def find_contours(self):
        self._ck(self.L.orip_find_contours(self.h))

    def contours_reserve(self, K: int):
        """hint: K layers of the image just set will be traced (clears stage 04's memo planes while the card is idle)"""
        self._ck(self.L.orip_contours_reserve(self.h, int(K)))

    def contours_prepare(self):
        self._ck(self.L.orip_contours_prepare(self.h))

    def contours_layer(self, layer: int):
        self._ck(self.L.orip_contours_layer(self.h, layer))

    def get_skeleton(self, layer: int) -> np.ndarray:
        out = np.empty((self.H, self.W), np.uint8)
        self._ck(self.L.orip_get_skeleton(self.h, layer, _p(out)))
        return out

    # ---- slots
    def polys_size(self, slot: int, layer: int) -> Tuple[int, int]:
        n, t = C.c_int64(0), C.c_int64(0)
        self._ck(self.L.orip_polys_size(self.h, slot, layer, C.byref(n), C.byref(t)))
        return n.value, t.value

    def get_polys_flat(self, slot: int, layer: int) -> Tuple[np.ndarray, np.ndarray]:
        n, t = self.polys_size(slot, layer)
        off = np.zeros(n + 1, np.int64)
        pts = np.zeros((max(t, 1), 2), np.int32)
        self._ck(self.L.orip_get_polys(self.h, slot, layer, _p(off), _p(pts)))
        return off, pts[:t]

    def get_polys_offsets(self, slot: int, layer: int) -> np.ndarray:
        """off int64[n + 1] only: a walk-coded list is not expanded for this"""
        n, _ = self.polys_size(slot, layer)
        off = np.zeros(n + 1, np.int64)
        self._ck(self.L.orip_get_polys(self.h, slot, layer, _p(off), None))
        return off

    def get_polys(self, slot: int, layer: int) -> List[np.ndarray]:
        off, pts = self.get_polys_flat(slot, layer)
        return [pts[off[i]:off[i + 1]].reshape(-1, 1, 2) for i in range(len(off) - 1)]

    def set_polys(self, slot: int, layer: int, polys: Sequence[np.ndarray]):
        n = len(polys)
        flat = [np.asarray(p).reshape(-1, 2).astype(np.int32) for p in polys]
        off = np.zeros(n + 1, np.int64)
        for i, p in enumerate(flat):
            off[i + 1] = off[i] + len(p)
        pts = np.ascontiguousarray(np.concatenate(flat, 0) if n else np.zeros((1, 2), np.int32), np.int32)
        self._ck(self.L.orip_set_polys(self.h, slot, layer, n, _p(off), _p(pts)))

    def get_taps(self, which: int, layer: int) -> List[Tuple[int, int]]:
        n = C.c_int64(0)
        self._ck(self.L.orip_taps_size(self.h, which, layer, C.byref(n)))
        a = np.zeros((max(n.value, 1), 2), np.int32)
        self._ck(self.L.orip_get_taps(self.h, which, layer, _p(a)))
        return [(int(x), int(y)) for x, y in a[:n.value]]

    def set_taps(self, which: int, layer: int, taps: Sequence[Tuple[int, int]]):
        a = np.ascontiguousarray(np.asarray(list(taps), np.int32).reshape(-1, 2))
        self._ck(self.L.orip_set_taps(self.h, which, layer, len(a), _p(a) if len(a) else None))

    def set_layer_count(self, K: int):
        self._ck(self.L.orip_set_layer_count(self.h, K))
        self.K = K

    # ---- stages 05 .. 12
    def scale_vectors(self, layer: int, sx, sy, dx, dy):
        self._ck(self.L.orip_scale_vectors(self.h, layer, np.float32(sx), np.float32(sy), np.float32(dx), np.float32(dy)))

    def sort_contours(self, layer: int):
        self._ck(self.L.orip_sort_contours(self.h, layer))

    def dedup_layer(self, layer: int, prm: _l.Params08):
        self._ck(self.L.orip_dedup_layer(self.h, layer, C.byref(prm)))

    def layer_front(self, layer: int, sx, sy, dx, dy, upto: int, prm: _l.Params08 | None):
        """contours_layer -> scale_vectors [-> sort_contours [-> dedup_layer]] (upto 5 / 7 / 8) in one call on the layer's lane"""
        self._ck(self.L.orip_layer_front(self.h, layer, np.float32(sx), np.float32(sy), np.float32(dx), np.float32(dy), int(upto), C.byref(prm) if prm is not None else None))

    def dedup_cross(self, order: Sequence[int], prm: _l.Params10):
        o = np.ascontiguousarray(np.asarray(list(order), np.int32))
        self._ck(self.L.orip_dedup_cross(self.h, _p(o), len(o), C.byref(prm)))

    def dedup_cross_begin(self, prm: _l.Params10):
        self._ck(self.L.orip_dedup_cross_begin(self.h, C.byref(prm)))

    def dedup_cross_layer(self, layer: int, src_layer: int | None = None, defer_reorder: bool = False):
        """defer_reorder: leave the travel reorder of the layer's lines (10:253) to plot_order(layer) -- resident pipelines only"""
        src = layer if src_layer is None else src_layer
        if defer_reorder:
            self._ck(self.L.orip_dedup_cross_layer_deferred(self.h, src, layer))
        else:
            self._ck(self.L.orip_dedup_cross_layer_from(self.h, src, layer))

    # ---- 13_build_stream: direction codes of all moves of a plot
    def stream_codes(self, moves: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """moves int32 [n,4] (x0, y0, x1, y1) -> (off int64 [n+1], codes uint8 [total])"""
        m = np.ascontiguousarray(moves, np.int32).reshape(-1, 4)
        total = C.c_int64(0)
        self._ck(self.L.orip_stream_codes(self.h, _p(m) if len(m) else None, len(m), C.byref(total)))
        off = np.zeros(len(m) + 1, np.int64)
        codes = np.zeros(max(total.value, 1), np.uint8)
        self._ck(self.L.orip_stream_codes_fetch(self.h, _p(off), _p(codes)))
        return off, codes[:total.value]

    # ---- multi-GPU exchange (RCCL inside liborip.so)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * _l.COMM_ID_BYTES)()
        if self.L.orip_comm_unique_id(buf) != 0:
            raise OripError("orip_comm_unique_id failed (RCCL)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == _l.COMM_ID_BYTES
        buf = (C.c_uint8 * _l.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self.L.orip_comm_init(self.h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.L.orip_comm_destroy(self.h))

    def bcast_layer(self, root: int, my_slot: int):
        self._ck(self.L.orip_bcast_layer(self.h, int(root), int(my_slot)))

    def preview_cover(self, slot: int, layer: int, taps_which: int, W: int, H: int, thickness: int, radius: int, antialias: bool):
        """Coverage planes (lines, taps), uint8 (H, W) each, behind the previews 06 / 09 / 11 (include/orip.h: orip_preview_cover; parity unpinned)."""
        lines = np.zeros((H, W), np.uint8); taps = np.zeros((H, W), np.uint8)
        self._ck(self.L.orip_preview_cover(self.h, int(slot), int(layer), int(taps_which), int(W), int(H), int(thickness), int(radius), 1 if antialias else 0, _p(lines), _p(taps)))
        return lines, taps

    def plot_order(self, layer: int, R_insert: float) -> np.ndarray:
        n = C.c_int64(0)
        self._ck(self.L.orip_plot_order(self.h, layer, float(R_insert), C.byref(n)))
        ops = np.zeros((max(n.value, 1), 5), np.int32)
        if n.value:
            self._ck(self.L.orip_get_ops(self.h, layer, _p(ops)))
        return ops[:n.value]
