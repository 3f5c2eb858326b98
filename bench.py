#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: Mpx/s end-to-end (colour -> order, stages 02 -> 12) on a
synthetic 4096 x 4096, 8-layer image (BASELINE.json; SURVEY 8(d) generator, seed 20251121).

  python bench.py --gpus N --steps K --warmup W          (N > 1 is launched by torch.distributed.run, one rank per GPU)
  python bench.py --size 2048 --upto 3                    (BASELINE config C2: stages 02 + 03 only, raster kernels vs the HBM roofline)

A "step" is one pass of the whole hot path over one image whose pixels are already resident in HBM (orip_set_image is
outside the timed region).  Nothing is cached between steps: every step re-runs the k-means fit, all raster stages, the
contour walk, both dedup stages and the plot ordering, and ends with every op list AND its line points in host memory
("... to all ops lists in host memory"; r03: the fetch is inside the timed region of `value`).
With N > 1 the single image is processed by all ranks together (colour-layer sharding), so the scaling is
"strong"; the timed region is bracketed by a barrier + device sync and the max over ranks is reported.

One JSON line on rank 0.  Extra objects:
  inclusive    -- the same step with the upload of the image inside the timed region as well: SURVEY 8(d)'s "host pixels -> ops
                  lists in host memory" (PCIe-inclusive on both ends; `value` has the input resident, as the bench contract asks)
  c2           -- BASELINE config C2 in the same run: 2048 x 2048, 8 layers, stages 02 + 03 only, with the raster kernel groups'
                  measured GB/s against the HBM peak
  roofline     -- the kernel group with the largest device time in a profiled step: algorithmic bytes (DESIGN.md
                  "Algorithmic bytes") / mean duration measured with HIP events on the library's own stream
  kernel_groups-- the same figures for every raster kernel group
  cpu_baseline -- the CPU restatement (oracle/, "port") on the top-left 1024 x 1024 crop of the same image, all layers, stages
                  02 -> 12: single-threaded, and with min(K, nproc) layer threads (the reference's only parallelism, 03:42-48)
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # one hardware queue per layer lane (see orip/lib.py); before torch / HIP start
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")


def csrc_digest() -> str:
    """identifies the kernel sources a PMC file was collected on (tools/pmc_traffic.py stores the same digest)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "omnirevolve-image-processor_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_groups(H, W, K, n_points):
    """name -> (HIP-event labels, algorithmic bytes, per, note).  per == "launch": bytes of ONE launch of the (single) kernel;
    per == "step": bytes the whole group moves in one step (its kernels run once per layer).  DESIGN.md 'Algorithmic bytes'."""
    px = H * W
    return {
        "lab_assign": (["k_lab_assign"], 4 * px, "launch", "3 B BGR in + 1 B label out per pixel"),
        "morph_pass": (["k_morph_pass"], 2 * K * px, "launch", "1 B in + 1 B out per pixel per layer and pass (byte kernel: non-binary masks only)"),
        "morph_bits": (["k_morph_bits"], 2 * K * px // 8, "launch", "1 bit in + 1 bit out per pixel per layer and pass; the bit planes (2 MB per layer) stay in L2, "
                       "so this is cache traffic, not HBM traffic"),
        "blur_sobel_nms": (["k_blur_sobel_nms"], K * px // 8 + K * px // 4, "launch", "1 bit mask in + 2 bits (candidate, strong planes) out per pixel per layer"),
        "thin_bits": (["k_thin_bits"], 2 * K * px // 8, "launch", "1 bit in + 1 bit out per pixel per layer and sub-iteration; bit planes of 2 MB per layer: cache traffic"),
        "skel_state": (["k_skel_state"], K * px // 8 + 2 * K * px, "launch", "1 bit in + skeleton byte + state byte out per pixel per layer"),
        "stage04_write": (["k_write_walks"], 8 * n_points, "step", "8 B per emitted contour point (SURVEY 8d), one launch per layer; r03: the points are emitted in walk-coded "
                          "form (own points + tail records), so this launch (k_vown) only writes the distinct points -- the figure is the algorithmic one, not bytes moved"),
        "stage04_trace": (["k_trace", "k_write_walks"], K * px + 8 * n_points, "step",
                          "K B/px skeleton state read once + 8 B per emitted contour point (SURVEY 8d); k_trace is a serial dependent chain "
                          "per skeleton component (one wave each), so its time is latency, not bandwidth"),
    }


# HIP-event label -> kernel names as rocprofv3 reports them (for the PMC cross-check)
PMC_NAMES = {"k_trace": ["k_trace"], "k_write_walks": ["k_vown"], "k_lab_assign": ["k_lab_assign"], "k_blur_sobel_nms": ["k_blur_sobel_nms"],
             "k_morph_bits": ["k_morph_bits"], "k_thin_bits": ["k_thin_bits04"], "k_skel_state": ["k_bits_to_skel_state"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096, help="image side (BASELINE: 4096)")
    ap.add_argument("--layers", type=int, default=8, help="colour layers (BASELINE: 8)")
    ap.add_argument("--upto", type=int, default=12, choices=[3, 12], help="3: stages 02 + 03 only (BASELINE config C2, use with --size 2048)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="skip the BASELINE config C2 leg (counter passes: its 2048^2 launches would mix into the per-launch averages)")
    ap.add_argument("--in-flight", type=int, default=2, help="extra leg at N = 1: this many images in flight on the card, one context each (0 / 1: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(f"bench.py --gpus {args.gpus} does not start ranks itself: launch it with `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                 f"--master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...` (one rank per GPU)")
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None; coll_device = None; backend = None; n_vis = 1
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
    from orip import stages as S
    from orip.synth import synth_image, layer_names

    H = W = args.size; K = args.layers
    img = synth_image(H, W, K)
    cfg = Config(); cfg.color_names = layer_names(K)
    dev = Device(local_rank)
    comm = P.make_comm(dev, rank, world, coll_device) if world > 1 else None
    dev.set_image(img)      # input resident in HBM before the timed region

    def barrier():
        dev.sync()
        if dist is not None:
            dist.barrier()
        dev.sync()

    rank_times = {}             # N > 1: this rank's stage times of the LAST step (orip.parallel.run_path_sharded)

    def step(fetch=False):
        if args.upto == 3:
            centers, _ = dev.kmeans_fit(S.subsample_indices(H * W), K)
            dev.extract_layers(centers, want_counts=False)
            S._detect_edges_resident(dev, cfg)
            dev.sync()
            return 0
        return P.run_path_sharded(dev, cfg, H, W, rank, world, coll_device, fetch_lines=fetch, comm=comm, timings=rank_times)

    for _ in range(args.warmup):
        step(fetch=True)
    barrier()
    t0 = time.perf_counter()
    n_ops = 0
    for _ in range(args.steps):
        n_ops = step(fetch=True)      # ops AND line points of every layer end in host memory inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    per_rank = None
    if dist is not None and args.upto == 12:      # where each rank's time went in the last timed step (host clock, ms from the step's start)
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(rank_times))
        per_rank = gathered
    elapsed = max_over_ranks(elapsed)
    ms_per_step = elapsed / max(1, args.steps) * 1e3
    value = (H * W / 1e6) * args.steps / elapsed

    # ---- inclusive leg: host pixels -> every op list and its line points in host memory, same barrier bracket
    inclusive = None
    if args.upto == 12:
        n_inc = max(1, min(3, args.steps))
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_inc):
            dev.set_image(img)
            step(fetch=True)
        barrier()
        inc = max_over_ranks(time.perf_counter() - t1)
        inclusive = {"ms_per_step": round(inc / n_inc * 1e3, 2), "value": round((H * W / 1e6) * n_inc / inc, 4), "unit": "Mpx/s", "steps": n_inc,
                     "note": "upload of the 3 B/px image + the path + fetch of every layer's op rows and line points, per step (PCIe inclusive on both ends; "
                             "`value` differs only by the upload)"}

    # ---- roofline leg: one extra profiled step (HIP events around every launch on the library's stream); not timed above
    roofline = None; groups_out = {}
    if rank == 0:
        dev.prof_enable(True); dev.prof_reset()
    step()          # every rank takes part (the step holds collectives when N > 1); only rank 0 records events
    if rank == 0:
        dev.prof_enable(False)
        Keff = K if world == 1 else len(P.owned_layers(K, rank, world))
        n_points = sum(dev.polys_size(L.SLOT_CONTOURS, l)[1] for l in range(Keff)) if args.upto == 12 else 0
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
            # HBM-side bytes per launch from separate rocprofv3 --pmc passes of this command (tools/pmc_traffic.py; rocprofv3 cannot run
            # inside the timed process).  The file is only believed when it was collected on exactly these kernel sources.
            traffic = None; traffic_note = None; traffic_detail = None
            try:
                pmf = json.load(open(PMC_FILE))
                names = [n for k in e["kernels"] for n in PMC_NAMES.get(k, [k])]
                if pmf.get("csrc_digest") != csrc_digest():
                    traffic_note = f"{os.path.relpath(PMC_FILE, ROOT)} was collected on other kernel sources (digest {pmf.get('csrc_digest')} != {csrc_digest()}): refused"
                elif not all(n in pmf["kernels"] for n in names) or not (H == 4096 and K == 8 and world == 1 and args.upto == 12):
                    traffic_note = f"{os.path.relpath(PMC_FILE, ROOT)} does not cover this configuration / these kernels: refused"
                else:
                    raw_f = int(sum(pmf["kernels"][n]["fetch_bytes_per_launch_raw"] for n in names)); raw_w = int(sum(pmf["kernels"][n]["write_bytes_per_launch"] for n in names))
                    traffic = 2 * raw_f + raw_w
                    traffic_detail = {"fetch_size_raw": raw_f, "fetch_scale": 2, "write_size_raw": raw_w, "estimate": traffic,
                                      "calibration": "the factor 2 is calibrated for wide coalesced reads only (MI355X_MICROARCH.md, HBM section); the byte- and dword-granular "
                                                     "reads of k_trace are uncalibrated: the scaled figure is an ESTIMATE, the raw counters are what was measured"}
                    traffic_note = (f"{os.path.relpath(PMC_FILE, ROOT)}: 2 x FETCH_SIZE + WRITE_SIZE per launch (estimate, see traffic_detail), separate rocprofv3 --pmc passes of this command")
            except (OSError, KeyError, ValueError) as ex:
                traffic_note = f"no usable PMC file ({type(ex).__name__})"
            roofline = {"kernel": "+".join(e["kernels"]), "bound": "hbm", "achieved": e["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": e["frac"], "traffic": traffic, "traffic_detail": traffic_detail, "traffic_note": traffic_note, "avg_ms": e["avg_ms"], "algorithmic_bytes": e["algorithmic_bytes"]}

    # ---- pipelined leg (N = 1 only, reported next to `value`, never as `value`): M images in flight, one context and one host thread each.
    # A step alone leaves the card nearly idle while the walks of the heavy layers run; a second image fills that window.
    pipelined = None
    if world == 1 and args.upto == 12 and args.in_flight > 1:
        import threading
        try:
            devs = [dev] + [Device(local_rank) for _ in range(args.in_flight - 1)]
            for d in devs[1:]:
                d.set_image(img); P.run_path_sharded(d, cfg, H, W, 0, 1)      # warm-up: allocations
            for d in devs:
                d.sync()
            n_pl = max(2, args.steps)
            errs = []

            def work(d):
                try:
                    for _ in range(n_pl):
                        P.run_path_sharded(d, cfg, H, W, 0, 1)
                    d.sync()
                except BaseException as ex:       # reported below
                    errs.append(ex)
            t2 = time.perf_counter()
            th = [threading.Thread(target=work, args=(d,)) for d in devs]
            for x in th: x.start()
            for x in th: x.join()
            dtp = time.perf_counter() - t2
            for d in devs[1:]:
                d.close()
            if errs:
                raise errs[0]
            pipelined = {"in_flight": len(devs), "steps": n_pl * len(devs), "ms_per_step": round(dtp * 1e3 / (n_pl * len(devs)), 2),
                         "value": round((H * W / 1e6) * n_pl * len(devs) / dtp, 4), "unit": "Mpx/s",
                         "note": "throughput with several images in flight on one card (one context and host thread per image); `value` above is one image at a time"}
        except Exception as ex:
            pipelined = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- BASELINE config C2 in the same run (rank 0, N = 1, default size only): 2048^2 x 8, stages 02 + 03, raster kernel groups vs the HBM peak
    c2 = None
    if rank == 0 and world == 1 and args.upto == 12 and H == 4096 and not args.no_c2:
        try:
            H2 = W2 = 2048
            img2 = synth_image(H2, W2, K)
            dev.set_image(img2)

            def step2():
                centers, _ = dev.kmeans_fit(S.subsample_indices(H2 * W2), K)
                dev.extract_layers(centers, want_counts=False)
                S._detect_edges_resident(dev, cfg)
                dev.sync()
            step2()
            t3 = time.perf_counter()
            n2 = 5
            for _ in range(n2):
                step2()
            dt2 = (time.perf_counter() - t3) / n2
            dev.prof_enable(True); dev.prof_reset(); step2(); dev.prof_enable(False)
            g2 = {}
            for name, (kernels, nbytes, per, note) in kernel_groups(H2, W2, K, 0).items():
                tot_ms = 0.0; launches = 0
                for k in kernels:
                    ms, n = dev.prof_get(k); tot_ms += ms; launches = max(launches, n)
                if launches == 0 or name.startswith("stage04"):
                    continue
                gbs = nbytes * launches / (tot_ms * 1e-3) / 1e9
                g2[name] = {"launches": launches, "avg_ms": round(tot_ms / launches, 4), "algorithmic_bytes": int(nbytes), "achieved_GBs": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5)}
            kf_ms, kf_n = dev.prof_get("k_kmeans_fit")
            c2 = {"workload": f"{W2}x{H2} BGR image, {K} colour layers, stages 02+03 only (BASELINE config 2)", "ms_per_step": round(dt2 * 1e3, 3),
                  "value": round(H2 * W2 / 1e6 / dt2, 2), "unit": "Mpx/s", "steps": n2, "kmeans_fit_ms": round(kf_ms, 3) if kf_n else None, "kernel_groups": g2,
                  "peak_GBs": HBM_PEAK_GBS}
        except Exception as ex:
            c2 = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- CPU baseline legs (rank 0, N = 1 only): the oracle as a "port", bounded sample, single thread and layer threads
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        crop = min(1024, H)
        sub = np.ascontiguousarray(img[:crop, :crop])
        ocfg = dict(O.DEFAULTS, color_names=layer_names(K))
        nproc = os.cpu_count() or 1
        t = time.perf_counter(); O.run_pipeline(sub, ocfg, upto=args.upto); dt1 = time.perf_counter() - t
        nthr = max(1, min(K, nproc))
        t = time.perf_counter(); O.run_pipeline(sub, ocfg, upto=args.upto, threads=nthr); dtn = time.perf_counter() - t
        mpx = crop * crop / 1e6
        cpu = {"value": round(mpx / dt1, 5), "unit": "Mpx/s", "cores": 1, "kind": "port",
               "sample": f"top-left {crop}x{crop} crop of the bench image ({crop * crop / (H * W):.4f} of the pixels), all {K} layers, stages 02->{args.upto}, "
                         f"single thread, {dt1:.1f} s",
               "layer_parallel": {"value": round(mpx / dtn, 5), "unit": "Mpx/s", "cores": nthr, "nproc": nproc, "seconds": round(dtn, 1),
                                  "note": "stages 03-08 of different layers in min(K, nproc) threads, stages 02 / 10 / 12 serial (the reference only "
                                          "parallelises over layers, 03:42-48)"}}

    if rank == 0:
        stages = "stages 02->12 (k-means fit, masks, edges, contours, scale, sort, intra dedup, cross dedup, plot order), default A4 canvas 8400x11880" \
            if args.upto == 12 else "stages 02+03 only (k-means fit, label assignment, masks, morphology, blur, Canny): BASELINE config C2"
        out = {
            "metric": "Mpx/s end-to-end (color->order)", "value": round(value, 4), "unit": "Mpx/s",
            "n_gpus": min(world, max(1, n_vis)) if world > 1 else 1, "ranks": world, "backend": backend or "none",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/i32 raster + f32/f64 geometry", "data": "synthetic",
            "config": {"workload": f"{W}x{H} BGR image, {K} colour layers, {stages}", "parallelism": f"layer-sharded x{world}" if world > 1 else "one GPU, one pipeline per colour layer",
                       "ops_last_step_rank0": int(n_ops), "exchange": (comm.kind if comm is not None else "none")},
            "inclusive": inclusive, "pipelined": pipelined, "roofline": roofline, "kernel_groups": groups_out, "c2": c2, "cpu_baseline": cpu,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank          # host-side stage times of every rank in the last timed step (orip.parallel.run_path_sharded)
        if world > 1 and n_vis < world:
            out["note"] = f"{world} ranks shared {n_vis} visible GPU(s) (gloo rehearsal): NOT a multi-GPU scaling point"
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    dev.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
