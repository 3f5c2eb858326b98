"""The dispatches of ONE queue (default: the one holding the longest k_trace) in one run (ORIP_TRACE_RUN, default 1 = the first timed step; bench.py --in-flight 0: the runs are warm-up, timed steps, one inclusive leg, then the roofline leg, which synchronises after every profiled kernel; with --in-flight 2 the pipelined leg follows it) of a rocprofv3 --kernel-trace CSV, in order,
with the idle gap before each and a per-kernel total (development aid).  usage: python tools/chain_of_queue.py <dir-or-csv> [MIN_MS] [QUEUE|-] [FROM_MS]"""
import csv, glob, os, re, sys
from collections import defaultdict
p = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
files = [p] if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]      # a directory that collected several runs (gpurun merges them): the newest one, never a mix
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [s for s, e, n, q in rows if "k_kmeans_fit" in n]          # one per run of the path; run 1 = the first timed step of bench.py --warmup 1
RUN = int(os.environ.get("ORIP_TRACE_RUN", "1")); t0 = starts[RUN]; t_end = starts[RUN + 1] if RUN + 1 < len(starts) else 1 << 62
sel = [r for r in rows if t0 <= r[0] < t_end]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"void rocprim::.*::detail::", "rp::", n)
    return n[:56]
t_from = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
if len(sys.argv) > 3 and sys.argv[3] != "-":
    Q = sys.argv[3]
else:
    Q = max((e - s, q) for s, e, n, q in sel if "k_trace" in n)[1]
ch = [r for r in sel if r[3] == Q and (r[0] - t0) / 1e6 >= t_from]
print(f"queue {Q}: {len(ch)} dispatches, busy {sum(e - s for s, e, n, q in ch)/1e6:.1f} ms, span {(ch[0][0]-t0)/1e6:.1f} .. {(ch[-1][1]-t0)/1e6:.1f} ms")
tot = defaultdict(lambda: [0, 0.0]); prev = ch[0][0]; small = 0.0; nsmall = 0; gap_small = 0.0
for s, e, n, q in ch:
    d = (e - s) / 1e6; g = (s - prev) / 1e6
    tot[short(n)][0] += 1; tot[short(n)][1] += d
    if d >= min_ms or g >= min_ms:
        if nsmall:
            print(f"            ... {nsmall} short dispatches, {small:.2f} ms busy, {gap_small:.2f} ms idle")
            small = 0.0; nsmall = 0; gap_small = 0.0
        print(f"{(s - t0)/1e6:9.2f} +{d:7.2f}  (idle {g:6.2f})  {short(n)}")
    else:
        small += d; nsmall += 1; gap_small += max(g, 0.0)
    prev = max(prev, e)
print("\nper kernel on this queue:")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {d:8.2f} ms  {c:5d} x  {k}")
