#!/usr/bin/env python3
"""tests/golden/make_fullsize_digests.py -- TEST INFRASTRUCTURE.  Run ONCE in the build container (CPU only, ~20 min):

    python tests/golden/make_fullsize_digests.py [--size 4096 --layers 8 --out tests/golden/c3_digests.json]

Runs the ORACLE (oracle/, the CPU restatement of the reference's stages 02 -> 12) on the bench image of BASELINE.json's
headline configuration -- synth_image(4096, 4096, 8), default A4 canvas 8400 x 11880 -- and writes one SHA-256 per artefact
and layer, so that the GPU path can be compared with the oracle at full size although the oracle is far too slow to run inside
a GPU test (tests/test_gpu_fullsize.py::test_c3_digests_vs_oracle hashes the same artefacts from the device).

Digest of a polyline list = sha256(off int64[n+1] bytes || pts int32[total,2] bytes): order of the list, order of the points and
every coordinate.  Digest of a tap list = sha256(int32[n,2] bytes).  Digest of ops = sha256(int32[n,5] rows (type, line index,
flip, x, y)), the form orip_get_ops returns (12:85-187).  Rasters: sha256 of the u8 plane.

Layers 04 -> 08 run in one process per layer (they are independent, 04:234 ... 08:561); stage 10 is sequential (10:236-267).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import math
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))

from oracle import oracle as O  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pl_flat(h):
    L = O.lib()
    n = L.orc_pl_count(h); tot = L.orc_pl_total(h)
    off = np.zeros(n + 1, np.int64); pts = np.zeros((max(tot, 1), 2), np.int32)
    L.orc_pl_get(h, _p(off), _p(pts))
    return off, pts[:tot]


def sha_polys(off, pts):
    h = hashlib.sha256(); h.update(np.ascontiguousarray(off, np.int64).tobytes()); h.update(np.ascontiguousarray(pts, np.int32).tobytes())
    return {"n": int(len(off) - 1), "points": int(len(pts)), "sha256": h.hexdigest()}


def sha_taps(taps):
    a = np.ascontiguousarray(np.asarray(list(taps), np.int32).reshape(-1, 2))
    return {"n": int(len(a)), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}


def sha_plane(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.uint8).tobytes()).hexdigest()


def layer_front(args):
    """stages 03 -> 08 of one layer; returns (name, digests, lines_intra (off, pts), taps_intra)"""
    name, mask, cfg, w, h = args
    t0 = time.time()
    L = O.lib()
    d = {"mask": sha_plane(mask)}
    edges = O.stage03(mask, cfg); d["edges"] = sha_plane(edges)
    skel = O.thin_rot(edges); d["skeleton"] = sha_plane(skel)
    raw = O.PL(); L.orc_trace(_p(skel), skel.shape[0], skel.shape[1], raw.h)
    off, pts = pl_flat(raw.h); del raw
    lens = np.diff(off); keep = lens >= 5                               # 04:224
    koff = np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.int64)
    kpts = pts[np.repeat(keep, lens)] if len(lens) else pts
    del pts
    d["contours"] = sha_polys(koff, kpts)
    cont = O.PL(); L.orc_pl_set(cont.h, len(koff) - 1, _p(koff), _p(np.ascontiguousarray(kpts if len(kpts) else np.zeros((1, 2), np.int32))))
    del kpts
    sx, sy, dx, dy = O.scale_factors(w, h, cfg)
    scaled = O.PL(); L.orc_scale(cont.h, np.float32(sx), np.float32(sy), np.float32(dx), np.float32(dy), scaled.h); del cont
    d["scaled"] = sha_polys(*pl_flat(scaled.h))
    srt = O.PL(); L.orc_sort07(scaled.h, srt.h); del scaled
    d["sorted"] = sha_polys(*pl_flat(srt.h))
    prm = O.derived08(cfg)
    lines, taps = O.PL(), O.TL(); L.orc_stage08(srt.h, _p(prm), lines.h, taps.h); del srt
    loff, lpts = pl_flat(lines.h); tp = taps.get()
    d["lines_intra"] = sha_polys(loff, lpts); d["taps_intra"] = sha_taps(tp)
    d["oracle_seconds_03_08"] = round(time.time() - t0, 1)
    print(f"[{name}] 03->08 done in {time.time() - t0:.0f} s: {d['contours']['n']} contours / {d['contours']['points']} points -> "
          f"{d['lines_intra']['n']} lines, {len(tp)} taps", flush=True)
    return name, d, (loff, lpts.copy()), tp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c3_digests.json"))
    a = ap.parse_args()
    from orip.synth import synth_image, layer_names
    H = W = a.size; K = a.layers
    t00 = time.time()
    img = synth_image(H, W, K)
    names = layer_names(K)
    cfg = O._cfg(dict(color_names=names))
    masks, centers, labels = O.stage02(img, cfg)
    out = {"config": {"H": H, "W": W, "K": K, "color_names": names, "image": "orip.synth.synth_image(H, W, K) (seed 20251121)",
                      "canvas": list(O.canvas_size(cfg))},
           "image_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
           "centers_lab_sorted": np.asarray(centers, np.float32).tolist(),
           "labels_sha256": sha_plane(labels.astype(np.uint8)), "layers": {}}
    print(f"stage 02 done in {time.time() - t00:.0f} s", flush=True)
    jobs = [(n, masks[n], cfg, W, H) for n in masks]          # masks is keyed dark -> light: device layer l = l-th entry
    out["device_layer_names"] = [n for n in masks]
    with mp.get_context("fork").Pool(min(a.procs, K)) as pool:
        res = pool.map(layer_front, jobs, chunksize=1)
    intra = {}
    for name, d, (loff, lpts), tp in res:
        out["layers"][name] = d
        intra[name] = ([lpts[loff[i]:loff[i + 1]].reshape(-1, 1, 2) for i in range(len(loff) - 1)], tp)
    t1 = time.time()
    cross = O.stage10(intra, cfg)
    ops_all = {}
    for name in names:
        lines, taps = cross[name]
        flat = [np.asarray(p).reshape(-1, 2).astype(np.int32) for p in lines]
        off = np.concatenate([[0], np.cumsum([len(p) for p in flat])]).astype(np.int64)
        pts = np.concatenate(flat, 0) if flat else np.zeros((0, 2), np.int32)
        d = out["layers"][name]
        d["lines_cross"] = sha_polys(off, pts); d["taps_cross"] = sha_taps(taps)
        li, ti = O.PL(lines), O.TL(taps)
        cap = len(lines) + len(taps) + 1
        raw = np.zeros((cap, 5), np.int32)
        m = O.lib().orc_build_ops12(li.h, ti.h, float(max(80.0, cfg["pen_width_px"])), _p(raw), cap)
        d["ops"] = {"n": int(m), "sha256": hashlib.sha256(np.ascontiguousarray(raw[:m]).tobytes()).hexdigest()}
        ops_all[name] = O.stage12(lines, taps, cfg)
    draw, travel = O.path_length(ops_all)
    out["path_length_px"] = {"draw": draw, "travel": travel}
    out["oracle_seconds_total"] = round(time.time() - t00, 1)
    print(f"stages 10/12 done in {time.time() - t1:.0f} s; draw {draw:.1f} px, travel {travel:.1f} px", flush=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out, flush=True)


if __name__ == "__main__":
    main()
