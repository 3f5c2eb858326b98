"""CPU-side checks of the boundary: liborip.so loads and exports every symbol include/orip.h declares (no compute
calls without a GPU), the ctypes table covers the header, the product fails loudly without a GPU, and the host-side
configuration / naming logic mirrors the reference's rules."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "orip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from orip import lib
    assert os.path.exists(lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"liborip.so lacks {s}"


def test_ctypes_table_matches_header():
    from orip import lib
    assert sorted(lib.SIGNATURES) == _header_symbols()
    lib.load()


def test_no_gpu_means_loud_failure_not_fallback():
    """In the build container there is no GPU: creating a Device must raise (there is no CPU path)."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from orip.device import Device, OripError\n"
            "try:\n    Device(0)\n    print('CREATED')\nexcept OripError as e:\n    print('RAISED', e)\n") % os.path.join(ROOT, "omnirevolve-image-processor_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
    import shutil
    has_gpu = os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)
    if not has_gpu:
        assert "RAISED" in out and "no CPU fallback" in out, out


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "omnirevolve-image-processor_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "orc_common" not in txt and "liborip_oracle" not in txt, os.path.join(dp, f)


def test_config_semantics_drop_unknown_keys(tmp_path):
    import json
    from orip.config import load_config, canvas_size_px, scale_factors, margins_px
    p = tmp_path / "config.json"
    p.write_text(json.dumps({"output_dir": "x", "cluster_k": 9, "target_width_px": 100, "pixels_per_mm": 40, "color_names": ["a", "b"]}))
    cfg = load_config(str(p))
    assert cfg.output_dir == "x" and not hasattr(cfg, "cluster_k") and getattr(cfg, "target_width_px", None) is None     # config.py:123-127
    assert canvas_size_px(cfg) == (8400, 11880) and margins_px(cfg) == (400, 400, 400, 400)
    sx, sy, dx, dy = scale_factors(cfg, 4096, 4096)
    assert abs(sx - 7600 / 4096) < 1e-12 and (dx, dy) == (400, 400)                                                    # 05:63-79, offset not centred
    assert load_config(str(tmp_path / "missing.json")).output_dir == "output"                                          # unreadable -> defaults


def test_layer_naming_rules():
    from orip import stages as S
    from orip.config import Config
    from orip.synth import layer_names
    cfg = Config(); cfg.color_names = layer_names(8)
    assert S.cluster_names(cfg) == ["layer_dark", "layer_mid", "layer_skin", "layer_4", "layer_5", "layer_6", "layer_7", "layer_light"]   # 02:17-23,130 stable sort
    names = list(cfg.color_names)
    assert sorted(names, key=S.darkness_rank10) == ["layer_dark", "layer_mid", "layer_skin", "layer_light", "layer_4", "layer_5", "layer_6", "layer_7"]  # 10:206-208
    assert [S.color_index12(n) for n in ["layer_dark", "layer_skin", "layer_mid", "layer_light", "layer_4"]] == [3, 0, 1, 2, 0]               # 12:210-219
    assert S.ensure_odd(4) == 5 and S.ensure_odd(1) == 3                                                                                       # 03:9-11
    p8 = S.params08(cfg); p10 = S.params10(cfg)
    assert (p8.min_keep, p8.tap_max_per, p8.brush_forbid, p8.post_eps, p8.post_minlen, p8.W, p8.H) == (12.0, 160.0, 36, 1.28, 32, 8400, 11880)  # App. A.3
    assert (p10.tap_max_per, p10.D_lines, p10.min_keep) == (150.0, 120.0, 12.0) and S.r_insert12(cfg) == 80.0                                  # App. A.4
