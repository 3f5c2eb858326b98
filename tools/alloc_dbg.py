import os, sys
sys.path.insert(0, "omnirevolve-image-processor_amd"); sys.path.insert(0, ".")
from orip.config import Config
from orip.device import Device
from orip import parallel as P
from orip.synth import synth_image, layer_names
img = synth_image(4096, 4096, 8); cfg = Config(); cfg.color_names = layer_names(8)
d = Device(0)
for rep in range(4):
    print(f"=== step {rep}", file=sys.stderr, flush=True)
    d.set_image(img); P.run_path_sharded(d, cfg, 4096, 4096, 0, 1); d.sync()
