"""Timeline of one run (ORIP_TRACE_RUN, default 1 = the first timed step; run bench.py with --in-flight 0 for traces: its runs are then warm-up, timed steps, the inclusive leg and last the roofline leg, which synchronises after every profiled kernel (with --in-flight 2 the pipelined leg and the C2 leg follow)) inside a rocprofv3 --kernel-trace CSV: dispatches after the last k_kmeans_fit launch that last
longer than MIN_MS, by start time (development aid).  usage: python tools/trace_timeline.py <dir-or-csv> [MIN_MS]"""
import csv, glob, os, re, sys
p = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
files = [p] if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]      # a directory that collected several runs (gpurun merges them): the newest one, never a mix
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
starts = [s for s, e, n, q, g in rows if "k_kmeans_fit" in n]          # one per run of the path; run 1 = the first timed step of bench.py --warmup 1
RUN = int(os.environ.get("ORIP_TRACE_RUN", "1")); t0 = starts[RUN]; t_end = starts[RUN + 1] if RUN + 1 < len(starts) else 1 << 62
sel = [r for r in rows if t0 <= r[0] < t_end]
end = max(e for s, e, n, q, g in sel)
print(f"run: {(end - t0)/1e6:.1f} ms, {len(sel)} dispatches, busy kernel time {sum(e-s for s,e,n,q,g in sel)/1e6:.1f} ms")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"void rocprim::.*::detail::", "rp::", n)
    return n[:48]
for s, e, n, q, g in sel:
    if (e - s) / 1e6 >= min_ms:
        print(f"{(s - t0)/1e6:9.2f} +{(e - s)/1e6:8.2f}  q{q:>3} grid {g:>10}  {short(n)}")
