"""One rank's share of BASELINE config C5 (8192 x 8192, 16 layers, 8 ranks: layers r and r + 8) on one GPU (development aid).
usage: python tools/c5_share.py [RANK]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
import numpy as np
from orip.config import Config
from orip.device import Device
from orip import stages as S, lib as L, parallel as P
from orip.synth import synth_image, layer_names
r = int(sys.argv[1]) if len(sys.argv) > 1 else 0
K, H, W = 16, 8192, 8192
t = time.time(); img = synth_image(H, W, K); print(f"synth {time.time() - t:.1f} s", flush=True)
cfg = Config(); cfg.color_names = layer_names(K); cfg.max_dimension = 8192
d = Device(0)
mine = P.owned_layers(K, r, 8)
for rep in range(2):
    d.set_image(img); d.sync(); t = time.perf_counter()
    d.contours_reserve(len(mine))
    centers, _ = d.kmeans_fit(S.subsample_indices(H * W), K)
    d.extract_layers(centers, want_counts=False)
    d.keep_layers(mine)
    S._detect_edges_resident(d, cfg)
    d.contours_prepare()
    front = S.layer_front(d, cfg, W, H, 8)
    S.for_each_layer(front, range(len(mine)))
    d.sync(); t1 = time.perf_counter()
    d.dedup_cross_begin(S.params10(cfg))
    for i in range(len(mine)):
        d.dedup_cross_layer(i, src_layer=i)
    ops = [d.plot_order(i, S.r_insert12(cfg)) for i in range(len(mine))]
    d.sync(); t2 = time.perf_counter()
    print(f"rep {rep}: layers {mine}: 02->08 {1e3 * (t1 - t):.1f} ms, 10+12 {1e3 * (t2 - t1):.1f} ms", flush=True)
    for i, g in enumerate(mine):
        print(f"   layer {g}: contours {d.polys_size(L.SLOT_CONTOURS, i)}, lines_intra {d.polys_size(L.SLOT_LINES_INTRA, i)}, lines_cross {d.polys_size(L.SLOT_LINES_CROSS, i)}, ops {len(ops[i])}", flush=True)
