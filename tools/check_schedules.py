"""Development aid: the per-layer pipelines and the one-layer-at-a-time schedule must give byte-identical lines and ops, for several
layer counts (K = 16 uses every lane).  usage: python tools/check_schedules.py"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "omnirevolve-image-processor_amd")); sys.path.insert(0, ROOT)
import numpy as np
from orip.config import Config
from orip.device import Device
from orip import lib as L, parallel as P, stages as S
from orip.synth import synth_image, layer_names
for (H, W, K) in [(1024, 1024, 16), (768, 1280, 3), (1536, 1024, 2)]:
    img = synth_image(H, W, K, seed=5)
    cfg = Config(); cfg.color_names = layer_names(K)
    dev = Device(0)
    dig = []
    for serial in (False, True):
        if serial: os.environ["ORIP_SERIAL_LAYERS"] = "1"
        else: os.environ.pop("ORIP_SERIAL_LAYERS", None)
        dev.set_image(img)
        n = P.run_path_sharded(dev, cfg, H, W, 0, 1)
        h = hashlib.sha256()
        R = S.r_insert12(cfg)
        for g in range(max(2, K)):
            off, pts = dev.get_polys_flat(L.SLOT_LINES_CROSS, g); h.update(off.tobytes()); h.update(np.ascontiguousarray(pts).tobytes())
            h.update(np.ascontiguousarray(dev.plot_order(g, R)).tobytes())
        dig.append((n, h.hexdigest()[:16]))
    print(H, W, K, dig, "OK" if dig[0] == dig[1] else "MISMATCH", flush=True)
    dev.close()
