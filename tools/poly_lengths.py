"""Length distribution of the contour lists of the bench image (development aid).  usage: python tools/poly_lengths.py [SIZE] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
import numpy as np
from orip.config import Config
from orip.device import Device
from orip import stages as S, lib as L
from orip.synth import synth_image, layer_names
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img = synth_image(size, size, K)
cfg = Config(); cfg.color_names = layer_names(K)
d = Device(0)
S.run_path(img, cfg, d, fetch_ops=False)
for l in range(K):
    polys = d.get_polys(L.SLOT_CONTOURS, l)
    n = np.sort(np.array([len(p) for p in polys]))[::-1]
    print(f"layer {l}: {len(n)} polylines, {n.sum()} points, longest {n[:6].tolist()}, median {int(np.median(n))}, >100k: {(n > 100000).sum()}, >10k: {(n > 10000).sum()}, >128: {(n > 128).sum()}", flush=True)
    del polys
