"""Degree statistics of the skeletons of the bench image (development aid): how long the forced stretches (pixels with exactly two skeleton
neighbours) between decision points are, overall and in the largest component of every layer.  usage: python tools/skel_stats.py [SIZE] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
import numpy as np
from scipy import ndimage
from orip.config import Config
from orip.device import Device
from orip import stages as S
from orip.synth import synth_image, layer_names
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img = synth_image(size, size, K)
cfg = Config(); cfg.color_names = layer_names(K)
d = Device(0)
S.run_path(img, cfg, d, fetch_ops=False)
for l in range(K):
    sk = d.get_skeleton(l) > 0
    deg = ndimage.convolve(sk.astype(np.uint8), np.ones((3, 3), np.uint8), mode="constant") - 1
    deg = np.where(sk, deg, 0)
    lab, n = ndimage.label(sk, structure=np.ones((3, 3)))
    sizes = np.bincount(lab.ravel())[1:]
    big = int(np.argmax(sizes)) + 1
    m = lab == big
    h_all = np.bincount(deg[sk], minlength=9); h_big = np.bincount(deg[m], minlength=9)
    # runs of the largest component: connected groups of its degree-2 pixels
    r_lab, r_n = ndimage.label(m & (deg == 2), structure=np.ones((3, 3)))
    r_sizes = np.bincount(r_lab.ravel())[1:]
    print(f"layer {l}: {int(sk.sum())} px, {n} components; largest {int(sizes.max())} px: degree histogram {h_big.tolist()}, "
          f"{r_n} degree-2 groups (mean {r_sizes.mean():.1f}, median {int(np.median(r_sizes))}, max {int(r_sizes.max())}); all components: {h_all.tolist()}", flush=True)
    tot2 = int(r_sizes.sum())
    print("   degree-2 pixels of the largest component in groups of >= 8 / 16 / 32 / 64 / 128 px: " +
          " / ".join(f"{100.0 * r_sizes[r_sizes >= t].sum() / max(tot2, 1):.0f}%" for t in (8, 16, 32, 64, 128)) + f" of {tot2}", flush=True)
    if os.environ.get("ORIP_SAVE_BIG") and sizes.max() > 20000:
        ys, xs = np.nonzero(m)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"bigcomp_l{l}.npz"), ys=ys.astype(np.int32), xs=xs.astype(np.int32), shape=np.array(sk.shape))
