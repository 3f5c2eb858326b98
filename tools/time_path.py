"""Per-stage wall timing of the resident GPU path (development aid).  usage: python tools/time_path.py SIZE [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
import numpy as np
from orip.config import Config, scale_factors
from orip.device import Device
from orip import stages as S, lib as L
from orip.synth import synth_image, layer_names

size = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time(); img = synth_image(size, size, K); print(f"synth {time.time()-t0:.1f}s", flush=True)
cfg = Config(); cfg.color_names = layer_names(K)
d = Device(0)
for rep in range(2):
    T = {}
    def lap(name, f):
        d.sync(); t = time.perf_counter(); r = f(); d.sync(); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    lap("set_image", lambda: d.set_image(img))
    idx = S.subsample_indices(size * size)
    centers = lap("kmeans", lambda: d.kmeans_fit(idx, K))[0]
    lap("extract", lambda: d.extract_layers(centers, want_counts=False))
    lap("edges", lambda: S._detect_edges_resident(d, cfg))
    lap("contours", lambda: d.find_contours())
    tot = [d.polys_size(L.SLOT_CONTOURS, l) for l in range(K)]
    sx, sy, dx, dy = scale_factors(cfg, size, size)
    lap("scale", lambda: S.for_each_layer(lambda l: d.scale_vectors(l, sx, sy, dx, dy), range(K)))
    lap("sort07", lambda: S.for_each_layer(lambda l: d.sort_contours(l), range(K)))
    p8 = S.params08(cfg)
    lap("dedup08", lambda: S.for_each_layer(lambda l: d.dedup_layer(l, p8), range(K)))
    lnames = S.cluster_names(cfg)[:K]
    order = sorted(range(K), key=lambda l: (S.darkness_rank10(lnames[l]), cfg.color_names.index(lnames[l])))
    lap("cross10", lambda: d.dedup_cross(order, S.params10(cfg)))
    lap("order12", lambda: S.for_each_layer(lambda l: d.plot_order(l, S.r_insert12(cfg)), range(K)))
    total = sum(v for k, v in T.items() if k != "set_image")
    print(f"rep {rep} size {size} K {K}: contours n/pts per layer {tot}")
    print("   " + "  ".join(f"{k}={v*1e3:.1f}ms" for k, v in T.items()) + f"  | total(02-12)={total*1e3:.1f}ms -> {size*size/1e6/total:.2f} Mpx/s", flush=True)
for rep in range(3):       # the pipelined schedule (what bench.py times): set_image excluded
    d.set_image(img); d.sync()
    from orip import parallel as P
    t = time.perf_counter(); n_ops = P.run_path_sharded(d, cfg, size, size, 0, 1); d.sync(); dt = time.perf_counter() - t
    print(f"pipelined rep {rep}: {dt*1e3:.1f} ms -> {size*size/1e6/dt:.2f} Mpx/s ({n_ops} ops)", flush=True)
