"""Idle gaps of the critical queue in one run (ORIP_TRACE_RUN, default 1 = the first timed step; run bench.py with --in-flight 0 for traces: its runs are then warm-up, timed steps, the inclusive leg and last the roofline leg, which synchronises after every profiled kernel (with --in-flight 2 the pipelined leg and the C2 leg follow)) of a rocprofv3 --kernel-trace CSV: every gap of at least MIN_MS with the dispatches on either side
(development aid: a gap is a host round trip, a wait for another queue, or host work).  usage: python tools/chain_gaps.py <dir-or-csv> [MIN_MS]"""
import csv, glob, os, re, sys
p = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08
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
    return n[:40]
Q = max((e - s, q) for s, e, n, q in sel if "k_trace" in n)[1]
ch = [r for r in sel if r[3] == Q]
prev_end = ch[0][1]; prev_name = short(ch[0][2]); tot = 0.0; n = 0; small = 0.0
print(f"queue {Q}: span {(ch[0][0]-t0)/1e6:.1f} .. {(ch[-1][1]-t0)/1e6:.1f} ms, busy {sum(e - s for s, e, _, _ in ch)/1e6:.1f} ms")
for s, e, name, q in ch[1:]:
    g = (s - prev_end) / 1e6
    if g >= min_ms:
        print(f"{(prev_end - t0)/1e6:9.2f}  idle {g:6.2f} ms   after {prev_name:40s} before {short(name)}")
        tot += g; n += 1
    elif g > 0:
        small += g
    if e > prev_end:
        prev_end = e; prev_name = short(name)
print(f"{n} gaps >= {min_ms} ms: {tot:.2f} ms; smaller gaps: {small:.2f} ms")
