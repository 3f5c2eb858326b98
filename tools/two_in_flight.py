"""Throughput with M images in flight on one GPU (development aid): M contexts, one host thread each, every thread runs `steps` full passes
02 -> 12 over its own copy of the bench image.  usage: python tools/two_in_flight.py [M] [steps]"""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
from orip.config import Config
from orip.device import Device
from orip import parallel as P
from orip.synth import synth_image, layer_names

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H = W = 4096; K = 8
img = synth_image(H, W, K)
cfg = Config(); cfg.color_names = layer_names(K)
devs = [Device(0) for _ in range(M)]
for d in devs:
    d.set_image(img); P.run_path_sharded(d, cfg, H, W, 0, 1)          # warm-up: allocations
for d in devs: d.sync()
res = [0] * M
def work(i):
    for _ in range(steps):
        res[i] = P.run_path_sharded(devs[i], cfg, H, W, 0, 1)
    devs[i].sync()
t = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(M)]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t
print(f"{M} in flight: {M * steps} steps in {dt * 1e3:.1f} ms -> {dt * 1e3 / (M * steps):.1f} ms per step, {H * W / 1e6 * M * steps / dt:.1f} Mpx/s (ops {res})", flush=True)
