#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mpx/s end-to-end (colour -> order, stages 02 -> 12) on a
synthetic 4096 x 4096, 8-layer image (BASELINE.json; SURVEY 8(d) generator, seed 20251121).

  python bench.py --gpus N --steps K --warmup W          (N > 1 is launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the whole hot path over one image whose pixels are already resident in HBM (orip_set_image is
outside the timed region; the PCIe-inclusive figure is in DESIGN.md).  Nothing is cached between steps: every step
re-runs the k-means fit, all raster stages, the contour walk, both dedup stages and the plot ordering; the ops stay on
the GPU.  With N > 1 the single image is processed by all ranks together (colour-layer sharding), so the
scaling is "strong"; the timed region is bracketed by a barrier + device sync and the max over ranks is reported.

One JSON line on rank 0.  Extra objects:
  roofline     -- the kernel group with the largest device time in a profiled step: algorithmic bytes (DESIGN.md
                  "Algorithmic bytes") / mean duration measured with HIP events on the library's own stream
  kernel_groups-- the same figures for every raster kernel group, for reference
  cpu_baseline -- the CPU restatement (oracle/, "port") on a 512 x 512 crop of the same image, all 8 layers, stages 02->12
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # one hardware queue per layer lane (see orip/lib.py); before torch / HIP start
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def kernel_groups(H, W, K, n_points):
    """name -> (kernels, algorithmic bytes, per, note).  per == "launch": bytes of ONE launch of the (single) kernel; per == "step":
    bytes the whole group moves in one step (its kernels run once per layer).  DESIGN.md 'Algorithmic bytes'."""
    px = H * W
    return {
        "lab_assign": (["k_lab_assign"], 4 * px, "launch", "3 B BGR in + 1 B label out per pixel"),
        "morph_pass": (["k_morph_pass"], 2 * K * px, "launch", "1 B in + 1 B out per pixel per layer and pass (byte kernel: non-binary masks only)"),
        "morph_bits": (["k_morph_bits"], 2 * K * px // 8, "launch", "1 bit in + 1 bit out per pixel per layer and pass; the bit planes (2 MB per layer) stay in L2, "
                       "so this is cache traffic, not HBM traffic"),
        "blur_sobel_nms": (["k_blur_sobel_nms"], K * px + K * px // 4, "launch", "1 B mask in + 2 bits (candidate, strong planes) out per pixel per layer"),
        "thin_sub": (["k_thin_sub"], 2 * K * px, "launch", "1 B in + 1 B out per pixel per layer and sub-iteration (byte kernel, ORIP_THIN_BYTES only)"),
        "thin_bits": (["k_thin_bits"], 2 * K * px // 8, "launch", "1 bit in + 1 bit out per pixel per layer and sub-iteration; bit planes of 2 MB per layer: cache traffic"),
        "ccl_merge": (["k_ccl_merge"], 5 * K * px, "launch", "1 B image + 4 B parent per pixel per layer"),
        "stage04_write": (["k_write_walks"], 8 * n_points, "step", "8 B per emitted contour point (SURVEY 8d), one launch per layer"),
        "stage04_trace": (["k_trace", "k_write_walks"], K * px + 8 * n_points, "step",
                          "K B/px skeleton state read once + 8 B per emitted contour point (SURVEY 8d); k_trace is a serial dependent chain "
                          "per skeleton component (one wave each), so its time is latency, not bandwidth"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="image side (BASELINE: 4096)")
    ap.add_argument("--layers", type=int, default=8, help="colour layers (BASELINE: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None; coll_device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n_vis = torch.cuda.device_count()
        backend = os.environ.get("ORIP_DIST_BACKEND") or ("nccl" if n_vis >= world else "gloo")   # gloo: rehearsal of N ranks on fewer cards
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
        coll_device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        if n_vis and local_rank >= n_vis:
            local_rank = local_rank % n_vis

    import numpy as np
    from orip.config import Config
    from orip.device import Device
    from orip import lib as L
    from orip import parallel as P
    from orip.synth import synth_image, layer_names

    H = W = args.size; K = args.layers
    img = synth_image(H, W, K)
    cfg = Config(); cfg.color_names = layer_names(K)
    dev = Device(local_rank)
    dev.set_image(img)      # input resident in HBM before the timed region

    def barrier():
        dev.sync()
        if dist is not None:
            dist.barrier()
        dev.sync()

    def step():
        return P.run_path_sharded(dev, cfg, H, W, rank, world, coll_device)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    n_ops = 0
    for _ in range(args.steps):
        n_ops = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / max(1, args.steps) * 1e3
    value = (H * W / 1e6) * args.steps / elapsed

    # ---- roofline leg: one extra profiled step (HIP events around every launch on the library's stream); not timed above
    roofline = None; groups_out = {}
    if rank == 0:
        dev.prof_enable(True); dev.prof_reset()
    step()          # every rank takes part (the step holds collectives when N > 1); only rank 0 records events
    if rank == 0:
        dev.prof_enable(False)
        n_points = sum(dev.polys_size(L.SLOT_CONTOURS, l)[1] for l in range(K if world == 1 else len(P.owned_layers(K, rank, world))))
        Keff = K if world == 1 else len(P.owned_layers(K, rank, world))
        best = None
        for name, (kernels, nbytes, per, note) in kernel_groups(H, W, Keff, n_points).items():
            tot_ms = 0.0; launches = 0
            for k in kernels:
                ms, n = dev.prof_get(k); tot_ms += ms; launches = max(launches, n)
            if launches == 0:
                continue
            step_bytes = nbytes * launches if per == "launch" else nbytes
            gbs = step_bytes / (tot_ms * 1e-3) / 1e9          # bytes of the step / summed HIP-event durations of the group's launches
            entry = {"kernels": kernels, "launches": launches, "avg_ms": round(tot_ms / launches, 4), "total_ms": round(tot_ms, 3),
                     "algorithmic_bytes": int(step_bytes // launches), "achieved_GBs": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5), "note": note}
            groups_out[name] = entry
            if name != "stage04_write" and (best is None or tot_ms > groups_out[best]["total_ms"]):
                best = name
        if best:
            e = groups_out[best]
            # HBM-side bytes per launch from the committed PMC passes of this same command (tools/pmc_traffic.py; rocprofv3 cannot run
            # inside the timed process).  FETCH_SIZE is the raw counter: on gfx950 it shows half the bytes of wide coalesced reads and is
            # uncalibrated for the byte/dword-granular reads of the walk; WRITE_SIZE is exact.
            traffic = None; traffic_note = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
                if all(k in pm for k in e["kernels"]) and H == 4096 and K == 8 and world == 1:
                    traffic = int(sum(pm[k]["fetch_bytes_per_launch_raw"] + pm[k]["write_bytes_per_launch"] for k in e["kernels"]))
                    traffic_note = "profiles/r01_pmc_traffic.json: FETCH_SIZE (raw) + WRITE_SIZE per launch, separate rocprofv3 --pmc passes of this command"
            except (OSError, KeyError, ValueError):
                pass
            roofline = {"kernel": "+".join(e["kernels"]), "bound": "hbm", "achieved": e["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": e["frac"], "traffic": traffic, "traffic_note": traffic_note, "avg_ms": e["avg_ms"], "algorithmic_bytes": e["algorithmic_bytes"]}

    # ---- CPU baseline leg (rank 0, N = 1 only): the oracle as a "port", bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        crop = min(512, H)
        sub = np.ascontiguousarray(img[:crop, :crop])
        t = time.perf_counter()
        O.run_pipeline(sub, dict(O.DEFAULTS, color_names=layer_names(K)))
        dt = time.perf_counter() - t
        cpu = {"value": round(crop * crop / 1e6 / dt, 5), "unit": "Mpx/s", "cores": 1, "kind": "port",
               "sample": f"top-left {crop}x{crop} crop of the bench image ({crop * crop / (H * W):.4f} of the pixels), all {K} layers, stages 02->12, "
                         f"single thread, {dt:.1f} s"}

    if rank == 0:
        out = {
            "metric": "Mpx/s end-to-end (color->order)", "value": round(value, 4), "unit": "Mpx/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/i32 raster + f32/f64 geometry", "data": "synthetic",
            "config": {"workload": f"{W}x{H} BGR image, {K} colour layers, stages 02->12 (k-means fit, masks, edges, contours, scale, sort, "
                                   f"intra dedup, cross dedup, plot order), default A4 canvas 8400x11880", "parallelism": f"layer-sharded x{args.gpus}",
                       "ops_last_step_rank0": int(n_ops)},
            "roofline": roofline, "kernel_groups": groups_out, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    dev.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
