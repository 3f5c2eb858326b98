#!/usr/bin/env python3
"""Where the host was while a layer's queue sat idle (development aid).

  rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 2 ...
  python tools/api_gaps.py DIR [min_gap_ms] [run_index]

Takes the queue with the largest busy time in one step (the critical heavy layer, as tools/chain_of_queue.py does), finds every gap
between consecutive dispatches of that queue longer than min_gap_ms (default 0.25), and prints the HIP runtime calls the thread that
feeds the queue was inside during the gap (name, start relative to the step, duration): a long hipStreamSynchronize / hipMemcpy there
is a host round trip, a hipStreamWaitEvent followed by nothing is a wait for another stream, an empty list is the host being late."""
import csv
import glob
import os
import sys
from collections import defaultdict


def newest(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    if not fs:
        sys.exit(f"no {pat} under {d}")
    return fs[-1]


def main():
    d = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    run = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("ORIP_TRACE_RUN", "-2"))
    K = list(csv.DictReader(open(newest(d, "*kernel_trace.csv"))))
    A = list(csv.DictReader(open(newest(d, "*hip_api_trace.csv"))))
    for r in K:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    K.sort(key=lambda r: r["s"])
    starts = [r["s"] for r in K if r["Kernel_Name"].startswith("k_kmeans_fit_mb")]      # one per step: the first kernel of stage 02
    if not starts:
        sys.exit("no k_kmeans_fit_mb dispatch: cannot cut the trace into steps")
    t0 = starts[run]
    later = [s for s in starts if s > t0]
    t1 = later[0] if later else K[-1]["e"] + 1
    step = [r for r in K if t0 <= r["s"] < t1]
    busy = defaultdict(int)
    for r in step:
        busy[r["Queue_Id"]] += r["e"] - r["s"]
    q = max(busy, key=busy.get)
    mine = [r for r in step if r["Queue_Id"] == q]
    tid = max(set(r["Thread_Id"] for r in mine), key=[r["Thread_Id"] for r in mine].count)
    api = [a for a in A if a["Thread_Id"] == tid]
    for a in api:
        a["s"] = int(a["Start_Timestamp"]); a["e"] = int(a["End_Timestamp"])
    ms = lambda t: (t - t0) / 1e6
    print(f"queue {q} (thread {tid}): {len(mine)} dispatches, step {ms(t1):.1f} ms")
    for a, b in zip(mine, mine[1:]):
        gap = (b["s"] - a["e"]) / 1e6
        if gap < min_gap:
            continue
        print(f"\n{ms(a['e']):8.2f} .. {ms(b['s']):8.2f}  idle {gap:5.2f} ms   after {a['Kernel_Name'][:40]}  before {b['Kernel_Name'][:40]}")
        for c in api:
            if c["e"] < a["e"] - 200000 or c["s"] > b["s"]:
                continue
            dur = (c["e"] - c["s"]) / 1e6
            if dur >= 0.02:
                print(f"            {ms(c['s']):8.2f} + {dur:5.2f}  {c['Function']}")


if __name__ == "__main__":
    main()
