"""Per-kernel HIP-event timing of the resident GPU path (development aid).  usage: python tools/prof_path.py SIZE [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
from orip.config import Config
from orip.device import Device
from orip import stages as S
from orip.synth import synth_image, layer_names
KERNELS = ["k_lab_gather", "k_kmeans_fit", "k_lab_assign", "k_morph_pass", "k_blur_sobel_nms", "k_ccl_init", "k_ccl_merge", "k_ccl_flatten", "k_hyst_mark",
           "k_hyst_out", "k_thin_sub", "k_skel_state", "k_compact_count", "k_compact_write", "radix_sort_pairs", "k_trace", "k_write_walks", "k_poly_features_long", "k_cumlen_long", "k_scale_pts", "k_greedy_nn", "k_split_small08", "k_cumlen", "k_samples", "k_tail_sim", "k_caps_insert",
           "k_caps_stamp", "sort_cells", "k_accept", "k_bbox_pairs", "k_stamp_groups", "k_zs_sub", "k_ccl2_merge", "k_comp_paths", "k_cut_slots",
           "k_row_hdist", "k_col_cover", "k_taps_sequential", "k_plot_order"]
size = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img = synth_image(size, size, K)
cfg = Config(); cfg.color_names = layer_names(K)
d = Device(0)
S.run_path(img, cfg, d, fetch_ops=False)        # warm-up (allocations)
d.prof_enable(True); d.prof_reset()
t = time.perf_counter(); S.run_path(img, cfg, d, fetch_ops=False); d.sync(); wall = time.perf_counter() - t
rows = [(k,) + d.prof_get(k) for k in KERNELS]
tot = sum(r[1] for r in rows)
print(f"size {size} K {K}: profiled wall {wall*1e3:.1f} ms, kernels {tot:.1f} ms")
for k, ms, n in sorted(rows, key=lambda r: -r[1]):
    if n: print(f"  {k:20s} {ms:10.2f} ms  {n:6d} launches  {ms/n*1e3:10.1f} us/launch")
