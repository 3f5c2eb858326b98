# 13_build_stream.py -- drop-in for the reference stage of the same name: vector_manifest.json + <layer>/ops.pkl -> plot_stream.bin
# + plot_stream.json.  The per-step work (direction codes of every move) runs on the GPU through liborip.so; see orip/stream.py.
import json
import os
from pathlib import Path

import stage_io as _io
from orip import stream as ST
from orip.config import canvas_size_px, load_config


def main():
    cfg = load_config()
    out = Path(cfg.output_dir)
    W, H = canvas_size_px(cfg)                       # _target_size_px (13:40-52): target_*_px are never fields -> the mm path
    sc = ST.stream_config_from_pipeline(cfg)
    man_path = out / "vector_manifest.json"
    if not man_path.exists():
        raise SystemExit(f"Missing manifest: {man_path}")
    man = json.loads(man_path.read_text(encoding="utf-8"))
    if man.get("coords") not in (None, "pixel_top_left"):
        raise SystemExit("Unsupported coordinates in manifest; expected pixel_top_left")
    ms = man.get("image_size")
    if not (isinstance(ms, (list, tuple)) and len(ms) == 2 and int(ms[0]) == W and int(ms[1]) == H):
        print(f"[stream] WARN: manifest size {ms} != target {W}x{H}")
    maps = ST.load_color_maps(cfg)
    print(f"[stream] color maps: force={maps[0]} by_name={maps[1]} by_order={maps[2]}")
    layers = []
    for ordinal, entry in enumerate(man.get("layers", [])):
        name = str(entry.get("color_name", entry.get("name", "unknown")))
        idx = int(entry.get("color_index", 0))
        pkl = out / entry["file"]
        if not pkl.exists():
            raise SystemExit(f"Missing layer file: {pkl}")
        ops = _io.load_pickle(str(pkl))
        print(f"[stream] layer#{ordinal + 1} '{name}': color {idx} -> {ST.resolve_color_index(name, idx, ordinal, *maps)} | ops={len(ops)}")
        layers.append((name, idx, ops))
    data, meta = ST.build_stream(layers, W, H, sc, color_maps=maps)
    dst = out / "plot_stream.bin"
    dst.write_bytes(data)
    (out / "plot_stream.json").write_text(json.dumps({"target_steps": {"width": W, "height": H}, "bytes": len(data), "lines": meta["lines"], "taps": meta["taps"]}, indent=2),
                                          encoding="utf-8")
    print("Stream saved:", str(dst))
    print("  Size:", len(data), "bytes")
    print("  Lines:", meta["lines"], "Taps:", meta["taps"])


if __name__ == "__main__":
    main()
