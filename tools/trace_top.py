"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the longest single dispatches (development aid).
usage: python tools/trace_top.py <dir-or-csv> [N]"""
import csv, glob, os, sys
from collections import defaultdict
p = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
files = [p] if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]      # a directory that collected several runs (gpurun merges them): the newest one, never a mix
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
tot = defaultdict(lambda: [0, 0, 0])
for n, d, s, g, w in rows:
    t = tot[n]; t[0] += d; t[1] += 1; t[2] = max(t[2], d)
print(f"{len(rows)} dispatches, {sum(r[1] for r in rows)/1e6:.1f} ms of kernel time")
print("--- per kernel (total ms, launches, max ms)")
for n, (d, k, mx) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:N]:
    print(f"{d/1e6:10.3f} {k:7d} {mx/1e6:9.3f}  {n[:110]}")
print("--- longest dispatches (ms, grid, wg)")
for n, d, s, g, w in sorted(rows, key=lambda r: -r[1])[:N]:
    print(f"{d/1e6:10.3f} {g:>10} {w:>5}  {n[:110]}")
