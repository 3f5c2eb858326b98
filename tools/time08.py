"""Phase times of stage 08 per layer with the layers run one after another (ORIP_SERIAL_LAYERS + ORIP_TIME08): what a phase costs when it has the
card to itself, to set against the concurrent timeline (development aid).  usage: python tools/time08.py [SIZE] [K]"""
import os, sys
os.environ["ORIP_SERIAL_LAYERS"] = "1"; os.environ["ORIP_TIME08"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
from orip.config import Config
from orip.device import Device
from orip import stages as S
from orip.synth import synth_image, layer_names
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img = synth_image(size, size, K)
cfg = Config(); cfg.color_names = layer_names(K)
d = Device(0)
S.run_path(img, cfg, d, fetch_ops=False)
print("---- second run ----", file=sys.stderr, flush=True)
S.run_path(img, cfg, d, fetch_ops=False); d.sync()
