#!/usr/bin/env python3
"""tests/golden/make_c5_share_digests.py -- TEST INFRASTRUCTURE.  Run ONCE in the build container (CPU only):

    python tests/golden/make_c5_share_digests.py [--rank 0 --oracle-layers 0]

BASELINE config C5 is 8192 x 8192 with 16 colour layers on 8 GPUs; rank r owns the cluster layers r and r + 8.  This script runs
the ORACLE (oracle/, CPU restatement) on that image -- stage 02 on the whole image, stages 03 -> 08 for the layers named by
--oracle-layers -- and writes one SHA-256 per artefact, the form tests/golden/c3_digests.json uses, so that the GPU test of a rank's
share (tests/test_gpu_c5_share.py) has oracle results at full C5 size for them.  The heavy middle layers are NOT run: their contour
lists hold 10^9 .. 10^10 points in the reference's expanded form (bounce tails), beyond this container's 64 GiB, so
the GPU test checks them through properties only.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))

import make_fullsize_digests as M  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--oracle-layers", type=int, nargs="+", default=[0])
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c5_share_digests.json"))
    a = ap.parse_args()
    from orip.synth import synth_image, layer_names
    H = W = 8192; K = 16
    t0 = time.time()
    img = synth_image(H, W, K)
    names = layer_names(K)
    cfg = O._cfg(dict(color_names=names, max_dimension=8192))
    masks, centers, labels = O.stage02(img, cfg)
    dev_names = [n for n in masks]                      # dark -> light: device layer l = l-th entry
    out = {"config": {"H": H, "W": W, "K": K, "rank": a.rank, "world": 8, "owned_layers": [a.rank, a.rank + 8], "oracle_layers": a.oracle_layers,
                      "image": "orip.synth.synth_image(8192, 8192, 16) (seed 20251121)", "canvas": list(O.canvas_size(cfg))},
           "image_sha256": hashlib.sha256(img.tobytes()).hexdigest(), "centers_lab_sorted": np.asarray(centers, np.float32).tolist(),
           "labels_sha256": M.sha_plane(labels.astype(np.uint8)), "device_layer_names": dev_names, "layers": {}}
    print(f"stage 02 done in {time.time() - t0:.0f} s", flush=True)
    for l in a.oracle_layers:
        name, d, _, _ = M.layer_front((dev_names[l], masks[dev_names[l]], cfg, W, H))
        out["layers"][str(l)] = d
    out["oracle_seconds_total"] = round(time.time() - t0, 1)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out, flush=True)


if __name__ == "__main__":
    main()
