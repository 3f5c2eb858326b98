"""Config dataclass + load_config with the reference's semantics (config.py:9-132): same fields, same
defaults, JSON from CONFIG_PATH, unknown keys silently dropped (so the stage code's getattr(cfg, "x", d)
reads of non-fields always see the default -- SURVEY App. A.0)."""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

BGR = Tuple[int, int, int]


@dataclass
class Config:
    input_image: str = "input.png"
    output_dir: str = "output"
    n_cores: int = 12
    max_dimension: int = 2000
    color_names: List[str] = field(default_factory=lambda: ["layer_dark", "layer_mid", "layer_skin", "layer_light"])
    colors: List[BGR] = field(default_factory=lambda: [(0, 0, 0), (255, 0, 0), (0, 255, 0), (0, 0, 255)])
    color_tolerance: int = 30
    edge_low_threshold: int = 50
    edge_high_threshold: int = 150
    edge_kernel_size: int = 3
    edge_morph_kernel: int = 3
    edge_morph_open_iters: int = 1
    edge_morph_close_iters: int = 1
    smoothing_iterations: int = 2
    min_contour_area: float = 10.0
    epsilon_factor: float = 0.002
    dedup_max_passes: int = 10
    target_width_mm: int = 210
    target_height_mm: int = 297
    pixels_per_mm: int = 40
    margin_left_mm: float = 10.0
    margin_right_mm: float = 10.0
    margin_top_mm: float = 10.0
    margin_bottom_mm: float = 10.0
    pen_width_px: int = 60
    pen_radius_px: int = 30
    tap_max_area: float = 1200.0
    tap_max_perimeter: float = 160.0
    tap_max_dim: int = 25
    tap_merge_radius_px: int = 30
    thinning_min_segment_len: int = 5
    thinning_dt_margin: float = 0.0
    dedup_sample_step: int = 8
    dedup_overlap_threshold: float = 0.60
    dedup_draw_antialiased: bool = False
    ignore_tail_points_intra: int = 120
    collision_radius_intra_px: float = 18.0
    collision_radius_global_px: float = 21.0
    hash_stride_px: float = 18.0
    max_join_jump_px: float = 80.0
    simplify_enabled: bool = False
    stop_after_edges: bool = False
    stream_force_color_index: Optional[int] = None
    stream_color_by_name: Optional[Dict[str, int]] = None
    stream_color_by_order: Optional[List[int]] = None

    def ensure_output_dirs(self) -> None:
        os.makedirs(self.output_dir, exist_ok=True)
        for name in self.color_names:
            os.makedirs(os.path.join(self.output_dir, name), exist_ok=True)


def load_config(path: str | None = None) -> Config:
    p = path or os.environ.get("CONFIG_PATH")
    if not p:
        return Config()
    try:
        with open(p, "r", encoding="utf-8") as f:
            data = json.load(f)
    except Exception as e:  # same convention as the reference: warn and fall back to defaults
        print(f"[config] WARNING: failed to read JSON ({e}); using defaults.")
        return Config()
    known = {k: v for k, v in data.items() if k in Config.__dataclass_fields__}
    cfg = Config(**known)
    setattr(cfg, "_raw", data)
    setattr(cfg, "_path", p)
    print(f"[config] Loading config: {p} (exists=True)")
    return cfg


# ---- derived parameters ----
def canvas_size_px(cfg: Config) -> Tuple[int, int]:
    """_target_size_px (05:15-40, 08:103-113, 10:25-39, 12:38-52); target_*_px are never fields -> mm path."""
    return int(round(float(cfg.target_width_mm) * int(cfg.pixels_per_mm))), int(round(float(cfg.target_height_mm) * int(cfg.pixels_per_mm)))


def margins_px(cfg: Config) -> Tuple[int, int, int, int]:
    ppm = int(cfg.pixels_per_mm or 40)
    return tuple(max(0, int(round(float(v) * ppm))) for v in (cfg.margin_left_mm, cfg.margin_right_mm, cfg.margin_top_mm, cfg.margin_bottom_mm))


def scale_factors(cfg: Config, w_src: int, h_src: int) -> Tuple[float, float, int, int]:
    """_get_scale_factors_into_inner (05:63-79) with keep_aspect always True; returns (sx, sy, dx, dy)."""
    w_full, h_full = canvas_size_px(cfg)
    ml, mr, mt, mb = margins_px(cfg)
    inner_w = max(1, w_full - ml - mr)
    inner_h = max(1, h_full - mt - mb)
    s = min(inner_w / max(1e-6, w_src), inner_h / max(1e-6, h_src))
    return s, s, ml, mt
