# pipeline.py -- runner with the reference's contract (pipeline.py:66-224) for the stages this repository provides:
# one subprocess per stage, CONFIG_PATH in the environment, stdout streamed, non-zero exit aborts.  A stage this repository
# does not provide (14 stream preview) is run from --ref-dir (the reference's own script) when given, otherwise skipped with a notice.
import argparse
import json
import os
import subprocess
import sys
from dataclasses import asdict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from orip.config import Config  # noqa: E402

STEPS = [("[1/14] Image resize…", "01_resize.py"), ("[2/14] RGBK color extraction…", "02_color_extract.py"), ("[3/14] Edge detection…", "03_edge_detect.py"),
         ("[4/14] Find contours…", "04_find_contours.py"), ("[5/14] Scale vectors…", "05_scale_vectors.py"), ("[6/14] Scaled vector preview…", "06_preview_scaled.py"),
         ("[7/14] Sort contours…", "07_sort_contours.py"), ("[8/14] Intra-layer dedup…", "08_dedup_layer_basic.py"), ("[9/14] Preview after intra-dedup…", "09_preview_intra.py"),
         ("[10/14] Cross-layer dedup…", "10_dedup_cross_basic.py"), ("[11/14] Final preview…", "11_preview_cross.py"), ("[12/14] Optimize plot order…", "12_optimize_plot_order.py"),
         ("[13/14] Build stream…", "13_build_stream.py"), ("[14/14] Preview stream…", "14_preview_stream.py")]


def run_step(title, module, env):
    print(f"\n{title}")
    proc = subprocess.Popen([sys.executable, module], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, bufsize=1)
    try:
        for line in proc.stdout:
            print(line, end="", flush=True)
        proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        raise
    if proc.returncode != 0:
        print(f"\nError in {module} (exit={proc.returncode})")
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser(description="Raster -> Vector pipeline (GPU hot path)")
    ap.add_argument("input_image")
    ap.add_argument("--output", required=True, dest="output_dir")
    ap.add_argument("--start-step", type=int, default=1)
    ap.add_argument("--end-step", type=int, default=len(STEPS))
    ap.add_argument("--pixels-per-mm", type=int, dest="pixels_per_mm")
    ap.add_argument("--target-width-mm", type=int, dest="target_width_mm")
    ap.add_argument("--target-height-mm", type=int, dest="target_height_mm")
    ap.add_argument("--colors", dest="colors_json")
    ap.add_argument("--ref-dir", default=None, help="directory of the reference's image_processor/ for the stages outside the hot path")
    a = ap.parse_args()
    os.makedirs(a.output_dir, exist_ok=True)
    dst = os.path.join(a.output_dir, "config.json")
    merged = asdict(Config())
    if os.path.exists(dst):
        try:
            with open(dst, "r", encoding="utf-8") as f:
                merged = json.load(f)
        except Exception:
            merged = {}
    over = {"input_image": a.input_image, "output_dir": a.output_dir, "pixels_per_mm": a.pixels_per_mm, "target_width_mm": a.target_width_mm, "target_height_mm": a.target_height_mm}
    if a.colors_json:
        try:
            over["colors"] = json.loads(a.colors_json)
        except Exception as e:
            print(f"Failed to parse --colors JSON: {e}", file=sys.stderr)
    merged.update({k: v for k, v in over.items() if v is not None})
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(merged, f, indent=2, ensure_ascii=False)
    print("Config saved to", dst)
    env = os.environ.copy(); env["CONFIG_PATH"] = dst; env["PYTHONUNBUFFERED"] = "1"
    s0 = max(1, min(a.start_step, len(STEPS))); s1 = max(1, min(a.end_step, len(STEPS)))
    if s0 > s1:
        s0, s1 = s1, s0
    for i in range(s0 - 1, s1):
        title, name = STEPS[i]
        mine = os.path.join(HERE, name)
        if os.path.exists(mine):
            run_step(title, mine, env)
        elif a.ref_dir and os.path.exists(os.path.join(a.ref_dir, name)):
            run_step(title, os.path.join(a.ref_dir, name), env)
        else:
            print(f"\n{title}\n  (stage {name} is outside the GPU hot path and no --ref-dir was given: skipped)")
    print("\nDone.")


if __name__ == "__main__":
    main()
