"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only):
   python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
rocprofv3 reports both counters in units of 1024 B.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B read requests at
64 B, i.e. wide coalesced reads show HALF their bytes; WRITE_SIZE is exact for wide stores; byte-granular / scattered accesses
are uncalibrated.  The JSON keeps the raw sums and the per-launch averages so the correction stays visible."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_digest():
    """the kernel sources the counters were collected on (bench.py refuses the file when this no longer matches)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "omnirevolve-image-processor_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())
    return h.hexdigest()[:16]

def load(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != name: continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fs, fn = f.get(k, [0.0, 0]); ws, wn = w.get(k, [0.0, 0])
    n = max(fn, wn)
    if not n: continue
    out[k] = {"launches": n, "fetch_kb_total": round(fs, 1), "write_kb_total": round(ws, 1),
              "fetch_bytes_per_launch_raw": int(fs * 1024 / max(fn, 1)), "write_bytes_per_launch": int(ws * 1024 / max(wn, 1))}
json.dump({"csrc_digest": csrc_digest(), "unit_note": "raw counter x 1024 B; gfx950: FETCH_SIZE shows half the bytes of wide coalesced reads (MI355X_MICROARCH.md), other widths uncalibrated",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-c2 --in-flight 0 (warm-up, timed, inclusive and profiled step per pass)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k in ("k_trace", "k_vown", "k_lab_assign", "k_nms_bits3", "k_morph_bits", "k_thin_bits04", "k_bits_to_skel_state", "k_greedy_nn_fast"):
    if k in out: print(k, out[k])
