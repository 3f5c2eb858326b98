"""For the critical queue's idle gaps (see chain_gaps.py): which HIP API calls the host was in during each gap, from a rocprofv3
--kernel-trace --hip-trace CSV pair (development aid).  usage: python tools/host_gaps.py <dir> [MIN_MS]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]; min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ht = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)[0]
K = []
with open(kt) as fh:
    for r in csv.DictReader(fh):
        K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Thread_Id", "?"), r.get("Correlation_Id", "")))
K.sort()
starts = [s for s, e, n, q, t, c in K if "k_kmeans_fit" in n]          # one per run of the path; run 1 = the first timed step of bench.py --warmup 1
RUN = int(os.environ.get("ORIP_TRACE_RUN", "1")); t0 = starts[RUN]; t_end = starts[RUN + 1] if RUN + 1 < len(starts) else 1 << 62
sel = [r for r in K if t0 <= r[0] < t_end]
Q = max((e - s, q) for s, e, n, q, t, c in sel if "k_trace" in n)[1]
ch = [r for r in sel if r[3] == Q]
tid = ch[len(ch) // 2][4]
A = []
with open(ht) as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        A.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]))
A.sort()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"void rocprim::.*::detail::", "rp::", n)
    return n[:36]
print(f"queue {Q}, kernels launched by thread {tid}")
prev_end = ch[0][1]; prev_name = short(ch[0][2])
for s, e, name, q, t, c in ch[1:]:
    g = (s - prev_end) / 1e6
    if g >= min_ms:
        calls = defaultdict(float); cnt = defaultdict(int)
        for a0, a1, fn, th in A:
            if a1 <= prev_end or a0 >= s: continue
            if th != t: continue
            calls[fn] += (min(a1, s) - max(a0, prev_end)) / 1e6; cnt[fn] += 1
        top = sorted(calls.items(), key=lambda kv: -kv[1])[:4]
        print(f"{(prev_end - t0)/1e6:9.2f} idle {g:5.2f} ms after {prev_name:36s} before {short(name):28s} | " + ", ".join(f"{fn} x{cnt[fn]} {ms:.2f}" for fn, ms in top))
    if e > prev_end:
        prev_end = e; prev_name = short(name)
