"""Plotter byte stream from the ordered ops of every layer -- the step after the path (13_build_stream.py + the wire protocol of
shared/omnirevolve_plotter_stream_creator_helper.py).

Wire format (helper :7-15): step bytes 11 FFF SSS (two steps) / 10 SSS 000 (one); service bytes 0x40|div (speed), 0x01 pen up,
0x02 pen down, 0x03 tap, 0x08..0x0F colour, 0x3F end of stream; padded with zeros to a multiple of 1024 bytes (:170-175).

The reference emits the stream one segment at a time through a byte-appending writer.  Here a plot is compiled in three passes:
  1. every MOVE of the plot (pen-up travel, polyline segment) is collected in order, in plotter step space (numpy per polyline);
  2. the direction codes of all moves come from ONE launch of the HIP kernel behind orip_stream_codes (closed-form Bresenham, one thread
     per step) -- the only per-step work of the stage;
  3. the speed plan cuts every move into pieces (divider, step range) exactly as the helper's ramps do, and the bytes of all pieces are
     assembled with vectorised numpy (prefix sums of the byte counts; steps are paired per piece, as StreamWriter.add_steps pairs them).
Everything that decides a byte follows the reference line by line in meaning: rounding (Python round = half-to-even), clamping, the
Y flip, the "approach before colour select" rule, corner thresholds, ramp tables, the per-piece step pairing, the pen / tap sequence."""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

SPI_CHUNK_SIZE = 1024                      # helper :22
PEN_UP, PEN_DOWN, TAP, EOF_BYTE = 0x01, 0x02, 0x03, 0x3F


@dataclass
class StreamConfig:                        # helper Config (:100-128): same fields, same defaults
    steps_per_mm: float = 40.0
    invert_y: bool = True
    div_start: int = 28
    div_fast: int = 15
    profile: str = "triangle"
    corner_deg: float = 85.0
    corner_div: int = 28
    corner_window_steps: int = 300
    short_len_steps: int = 120
    short_div: int = 16
    travel_div_fast: int = 10
    travel_start_div: int = 28
    travel_window_steps: int = 240
    travel_quant_step: int = 4
    soft_tail_steps: int = 0
    soft_tail_div: int = 20


def stream_config_from_pipeline(cfg) -> StreamConfig:
    """13:55-68.  The draw_* / corner_* / travel_* keys are not Config fields, so load_config drops them and the getattr defaults rule."""
    return StreamConfig(steps_per_mm=float(getattr(cfg, "pixels_per_mm", 40.0)), invert_y=True,
                        div_start=int(getattr(cfg, "draw_div_start", 25)), div_fast=int(getattr(cfg, "draw_div_fast", 15)),
                        profile=str(getattr(cfg, "draw_profile", "triangle")), corner_deg=float(getattr(cfg, "corner_deg", 85.0)),
                        corner_div=int(getattr(cfg, "corner_div", 30)), corner_window_steps=int(getattr(cfg, "corner_window_steps", 800)),
                        travel_div_fast=int(getattr(cfg, "travel_div_fast", 10)))


# ------------------------------------------------------------------ speed plans: a move of N steps -> [(divider, count), ...]
def _even_parts(total: int, levels: int) -> List[int]:               # helper :70-74
    base, rem = divmod(total, levels)
    return [base + (1 if i < rem else 0) for i in range(levels)]


def _ramp_counts(profile: str, length: int, div_fast: int, div_slow: int) -> Dict[int, int]:     # helper :76-98, :211-216
    if length <= 0:
        return {}
    if div_slow < div_fast:
        raise ValueError("div_slow must be >= div_fast")
    out: Dict[int, int] = {}
    if profile == "triangle":
        for i, cnt in enumerate(_even_parts(length, div_slow - div_fast + 1)):
            if cnt > 0:
                out[div_slow - i] = out.get(div_slow - i, 0) + cnt
    elif profile == "scurve":
        span = div_slow - div_fast
        for i in range(length):
            t = (i + 0.5) / length
            div = max(div_fast, min(div_slow, round(div_slow - ((3 * t * t) - (2 * t * t * t)) * span)))
            out[div] = out.get(div, 0) + 1
    else:
        raise ValueError("profile must be 'triangle' or 'scurve'")
    return out


def _accel(n: int, profile: str, div_fast: int, start_div: int) -> List[Tuple[int, int]]:        # emit_steps_accel, helper :218-227
    if n <= 0:
        return []
    if start_div <= div_fast:
        return [(div_fast, n)]
    counts = _ramp_counts(profile, n, div_fast, start_div)
    return [(d, counts[d]) for d in range(start_div, div_fast - 1, -1) if counts.get(d, 0) > 0]


def _decel(n: int, profile: str, div_fast: int, end_div: int) -> List[Tuple[int, int]]:          # emit_steps_decel, helper :229-238
    if n <= 0:
        return []
    if end_div <= div_fast:
        return [(div_fast, n)]
    counts = _ramp_counts(profile, n, div_fast, end_div)
    return [(d, counts[d]) for d in range(div_fast, end_div + 1) if counts.get(d, 0) > 0]


def plan_segment(n: int, sc: StreamConfig, slow_in: bool, slow_out: bool) -> List[Tuple[int, int]]:
    """emit_segment_with_corner_profile (helper :251-292) as a list of (divider, count) pieces covering the n steps in order."""
    if n == 0:
        return []
    if not slow_in and not slow_out:
        return [(sc.short_div if n <= sc.short_len_steps else sc.div_fast, n)]
    entry = min(sc.corner_window_steps if slow_in else 0, n)
    exit_ = min(sc.corner_window_steps if slow_out else 0, max(0, n - entry))
    mid = max(0, n - entry - exit_)
    if entry + exit_ >= n:
        half = n // 2
        out = _accel(half, sc.profile, sc.div_fast, sc.corner_div if slow_in else sc.div_start) if half > 0 else []
        if n % 2 == 1:
            out.append((sc.div_fast, 1)); half += 1
        return out + _decel(n - half, sc.profile, sc.div_fast, sc.corner_div if slow_out else sc.div_start)
    out = _accel(entry, sc.profile, sc.div_fast, sc.corner_div)
    if mid > 0:
        out.append((sc.div_fast, mid))
    return out + _decel(exit_, sc.profile, sc.div_fast, sc.corner_div)


def _quantized_levels(div_slow: int, div_fast: int, step: int) -> List[int]:                      # helper :100-107 (:88-95 in the file)
    if div_slow < div_fast:
        div_slow, div_fast = div_fast, div_slow
    levels = list(range(div_slow, div_fast - 1, -step))
    if levels[-1] != div_fast:
        levels.append(div_fast)
    return levels


def plan_travel(n: int, sc: StreamConfig) -> List[Tuple[int, int]]:
    """travel_ramped (helper :340-380)."""
    if n == 0:
        return []
    win, div_fast, div_start = int(sc.travel_window_steps), int(sc.travel_div_fast), int(sc.travel_start_div)
    if div_start < div_fast:
        div_start = div_fast
    if n <= 2 * win:
        half = max(1, n // 2)
        out = _accel(half, sc.profile, div_fast, div_start)
        if n % 2 == 1:
            out.append((div_fast, 1 if half < n else 0)); half += 1     # n == 1: the helper still sets the speed, for a slice without steps
        return out + _decel(max(0, n - half), sc.profile, div_fast, div_start)
    down = _quantized_levels(div_start, div_fast, max(1, int(sc.travel_quant_step)))
    out = [(d, c) for d, c in zip(down, _even_parts(win, len(down))) if c > 0]
    if n - 2 * win > 0:
        out.append((div_fast, n - 2 * win))
    return out + [(d, c) for d, c in zip(reversed(down), _even_parts(win, len(down))) if c > 0]


# ------------------------------------------------------------------ geometry in step space
def to_steps(xy: np.ndarray, W: int, H: int) -> np.ndarray:
    """_to_steps (13:84-88) on an array of (x, y): round half-to-even, clamp to the sheet, Y flip."""
    p = np.rint(np.asarray(xy, np.float64).reshape(-1, 2))
    x = np.clip(p[:, 0], 0, W - 1).astype(np.int64)
    y = (H - 1) - np.clip(p[:, 1], 0, H - 1).astype(np.int64)
    return np.stack([x, y], 1)


def _angle(a, b, c) -> float:                                          # angle_degrees, helper :242-249 (Python floats, libm)
    v1x, v1y, v2x, v2y = a[0] - b[0], a[1] - b[1], c[0] - b[0], c[1] - b[1]
    n1, n2 = math.hypot(v1x, v1y), math.hypot(v2x, v2y)
    if n1 == 0 or n2 == 0:
        return 180.0
    return math.degrees(math.acos(max(-1.0, min(1.0, (v1x * v2x + v1y * v2y) / (n1 * n2)))))


def corner_flags(pl: np.ndarray, corner_deg: float) -> Tuple[np.ndarray, np.ndarray]:
    """slow_in / slow_out of every segment of a polyline in step space (emit_polyline, helper :300-312).  The interior angle at vertex j is
    computed vectorised; a vertex whose angle comes out within 1e-6 degrees of the threshold is decided again with the helper's scalar
    formula (math.hypot / acos / degrees), so the comparison is the reference's own arithmetic wherever it could matter."""
    n = len(pl)
    sharp = np.zeros(n, bool)                                           # sharp[j]: angle at vertex j (between j-1, j, j+1) below the threshold
    if n >= 3:
        p = pl.astype(np.float64)
        v1, v2 = p[:-2] - p[1:-1], p[2:] - p[1:-1]
        n1, n2 = np.hypot(v1[:, 0], v1[:, 1]), np.hypot(v2[:, 0], v2[:, 1])
        ok = (n1 > 0) & (n2 > 0)
        cosv = np.clip((v1[:, 0] * v2[:, 0] + v1[:, 1] * v2[:, 1]) / np.where(ok, n1 * n2, 1.0), -1.0, 1.0)
        ang = np.where(ok, np.degrees(np.arccos(cosv)), 180.0)
        sharp[1:-1] = ang < corner_deg
        for j in np.nonzero(np.abs(ang - corner_deg) < 1e-6)[0]:
            sharp[j + 1] = _angle(pl[j], pl[j + 1], pl[j + 2]) < corner_deg
    slow_in = sharp[:-1].copy(); slow_in[0] = False                     # segment i = (i, i+1): entry corner at vertex i (i > 0)
    slow_out = sharp[1:].copy(); slow_out[-1] = False                   # exit corner at vertex i + 1 (when a vertex i + 2 exists)
    return slow_in, slow_out


# ------------------------------------------------------------------ colour remap (13:92-160)
def _color_idx(x) -> int:
    try:
        return int(x) & 7
    except Exception:
        return 0


def load_color_maps(cfg):
    force = getattr(cfg, "stream_force_color_index", None)
    if force is not None:
        force = _color_idx(force)
    by_name = getattr(cfg, "stream_color_by_name", None)
    by_name = {str(k): _color_idx(v) for k, v in by_name.items()} if isinstance(by_name, dict) else None
    by_order = getattr(cfg, "stream_color_by_order", None)
    by_order = [_color_idx(v) for v in by_order] if isinstance(by_order, (list, tuple)) and len(by_order) > 0 else None
    env_force = os.environ.get("STREAM_FORCE_COLOR_INDEX")
    if env_force is not None:
        force = _color_idx(env_force)
    env_order = os.environ.get("STREAM_COLOR_ORDER")
    if env_order:
        by_order = [_color_idx(v) for v in env_order.split(",")]
    return force, by_name, by_order


def resolve_color_index(name: str, orig: int, ordinal: int, force, by_name, by_order) -> int:
    if force is not None:
        return force
    if by_name and name in by_name:
        return by_name[name]
    if by_order:
        return by_order[ordinal % len(by_order)]
    return _color_idx(orig)


# ------------------------------------------------------------------ the compiler
class _Plot:
    """Ordered items of a plot: service bytes and moves.  A move is recorded with the plan function that will cut it into speed pieces
    once its step count is known."""

    def __init__(self):
        self.kind: List[int] = []            # per item: >= 0 service byte, -1 a move
        self.moves: List[Tuple[int, int, int, int]] = []
        self.plans: List[Callable[[int], List[Tuple[int, int]]]] = []

    def svc(self, b: int):
        self.kind.append(b)

    def move(self, x0, y0, x1, y1, plan):
        self.kind.append(-1); self.moves.append((int(x0), int(y0), int(x1), int(y1))); self.plans.append(plan)


def _emit_layer(P: _Plot, ops: Sequence[dict], color_idx: int, W: int, H: int, sc: StreamConfig, cur: Tuple[int, int]) -> Tuple[int, int]:
    """13:179-227."""
    travel = lambda n: plan_travel(n, sc)        # noqa: E731
    if ops:
        first = ops[0]
        s = to_steps(np.array([[first["x"], first["y"]]]) if first["type"] == "tap" else np.asarray(first["points"]).reshape(-1, 2)[:1], W, H)[0]
        if cur != (s[0], s[1]):
            P.move(cur[0], cur[1], s[0], s[1], travel); cur = (int(s[0]), int(s[1]))
    if not (0 <= color_idx <= 7):
        raise ValueError("color index 0..7")
    P.svc(0x08 | (color_idx & 7))
    for op in ops:
        if op["type"] == "tap":
            t = to_steps(np.array([[op["x"], op["y"]]]), W, H)[0]
            if cur != (t[0], t[1]):
                P.svc(PEN_UP); P.move(cur[0], cur[1], t[0], t[1], travel); cur = (int(t[0]), int(t[1]))
            P.svc(TAP)
            continue
        pts = np.asarray(op["points"]).reshape(-1, 2)
        if len(pts) < 2:
            continue
        pl = to_steps(pts, W, H)
        if cur != (pl[0, 0], pl[0, 1]):
            P.svc(PEN_UP); P.move(cur[0], cur[1], pl[0, 0], pl[0, 1], travel)
        P.svc(PEN_DOWN)
        sin, sout = corner_flags(pl, sc.corner_deg)
        for i in range(len(pl) - 1):
            if sin[i] or sout[i]:
                P.move(pl[i, 0], pl[i, 1], pl[i + 1, 0], pl[i + 1, 1], (lambda n, a=bool(sin[i]), b=bool(sout[i]): plan_segment(n, sc, a, b)))
            else:
                P.move(pl[i, 0], pl[i, 1], pl[i + 1, 0], pl[i + 1, 1], None)          # the common case, planned vectorised in assemble()
        P.svc(PEN_UP)
        cur = (int(pl[-1, 0]), int(pl[-1, 1]))
    return cur


def assemble(P: _Plot, off: np.ndarray, codes: np.ndarray, sc: StreamConfig) -> bytes:
    """Bytes of the plot from its items, the step offsets of its moves and their direction codes (StreamWriter semantics, helper :130-175:
    a speed byte only when the divider changes, the steps of every piece paired on their own, end byte, padding)."""
    nmov = len(P.moves)
    counts = np.diff(off).astype(np.int64) if nmov else np.zeros(0, np.int64)
    # pieces of every move: (move index, divider, count); simple draw segments get their single piece without a Python call
    piece_mov: List[np.ndarray] = []; piece_div: List[np.ndarray] = []; piece_cnt: List[np.ndarray] = []
    simple = np.array([p is None for p in P.plans], bool) if nmov else np.zeros(0, bool)
    if nmov:
        idx = np.nonzero(simple & (counts > 0))[0]
        piece_mov.append(idx); piece_cnt.append(counts[idx])
        piece_div.append(np.where(counts[idx] <= sc.short_len_steps, sc.short_div, sc.div_fast).astype(np.int64))
        for m in np.nonzero(~simple)[0]:
            pcs = P.plans[m](int(counts[m]))
            if pcs:
                piece_mov.append(np.full(len(pcs), m, np.int64)); piece_div.append(np.array([d for d, _ in pcs], np.int64)); piece_cnt.append(np.array([c for _, c in pcs], np.int64))
    pm = np.concatenate(piece_mov) if piece_mov else np.zeros(0, np.int64)
    pd = np.concatenate(piece_div) if piece_div else np.zeros(0, np.int64)
    pc = np.concatenate(piece_cnt) if piece_cnt else np.zeros(0, np.int64)
    order = np.argsort(pm, kind="stable")                                 # pieces in move order, the pieces of one move in plan order
    pm, pd, pc = pm[order], pd[order], pc[order]
    # first step of every piece inside the code array
    within = np.zeros(len(pm), np.int64)
    if len(pm):
        first_of_move = np.r_[True, pm[1:] != pm[:-1]]
        csum = np.cumsum(pc) - pc
        within = csum - np.maximum.accumulate(np.where(first_of_move, csum, 0))
    pstart = off[pm] + within if len(pm) else np.zeros(0, np.int64)
    # speed bytes: helper set_speed appends only when the divider differs from the last one set (:139-143)
    spd = np.ones(len(pm), bool)
    if len(pm) > 1:
        spd[1:] = pd[1:] != pd[:-1]
    pbytes = spd.astype(np.int64) + (pc + 1) // 2
    # item order: every move contributes its pieces' bytes, a service item one byte
    kind = np.asarray(P.kind, np.int64)
    move_bytes = np.zeros(nmov, np.int64)
    np.add.at(move_bytes, pm, pbytes)
    item_bytes = np.ones(len(kind), np.int64)
    is_move = kind < 0
    item_bytes[is_move] = move_bytes
    item_pos = np.cumsum(item_bytes) - item_bytes
    total = int(item_bytes.sum())
    out = np.zeros(total + 1, np.uint8)
    out[item_pos[~is_move]] = kind[~is_move].astype(np.uint8)
    if len(pm):
        move_pos = item_pos[is_move]                                      # byte position of every move
        ppos = move_pos[pm] + (np.cumsum(pbytes) - pbytes) - np.maximum.accumulate(np.where(first_of_move, np.cumsum(pbytes) - pbytes, 0))
        sp = np.clip(pd, 0, 63)                                           # make_speed_byte (helper :46-51)
        out[ppos[spd]] = (0x40 | (sp[spd] & 0x3F)).astype(np.uint8)
        nb = (pc + 1) // 2                                                # step bytes per piece
        bpiece = np.repeat(np.arange(len(pm)), nb)
        j = np.arange(int(nb.sum())) - np.repeat(np.cumsum(nb) - nb, nb)  # index of the byte inside its piece
        a = codes[pstart[bpiece] + 2 * j].astype(np.int64) & 7
        has_b = 2 * j + 1 < pc[bpiece]
        b = np.where(has_b, codes[np.minimum(pstart[bpiece] + 2 * j + 1, max(len(codes) - 1, 0))].astype(np.int64) & 7, 0)
        out[(ppos + spd)[bpiece] + j] = np.where(has_b, 0x80 | 0x40 | (a << 3) | b, 0x80 | (a << 3)).astype(np.uint8)
    out[total] = EOF_BYTE
    pad = (-(total + 1)) % SPI_CHUNK_SIZE
    return out.tobytes() + b"\x00" * pad


def build_stream(layers: Sequence[Tuple[str, int, Sequence[dict]]], W: int, H: int, sc: StreamConfig, codes_fn: Optional[Callable] = None,
                 color_maps=(None, None, None)) -> Tuple[bytes, Dict[str, int]]:
    """13:231-281 for layers given as (colour name, manifest colour index, ops).  codes_fn(moves int32 [n,4]) -> (off, codes): the direction
    codes of all moves; None = the GPU (orip.device.Device.stream_codes, liborip.so) -- there is no CPU path in the product."""
    if codes_fn is None:
        from .stages import device
        codes_fn = device().stream_codes
    force, by_name, by_order = color_maps
    P = _Plot()
    P.svc(PEN_UP)
    cur = (0, 0)
    n_lines = n_taps = 0
    for ordinal, (name, orig_idx, ops) in enumerate(layers):
        cidx = resolve_color_index(name, orig_idx, ordinal, force, by_name, by_order)
        n_lines += sum(1 for o in ops if o["type"] == "line"); n_taps += sum(1 for o in ops if o["type"] == "tap")
        cur = _emit_layer(P, ops, cidx, W, H, sc, cur)
    moves = np.asarray(P.moves, np.int32).reshape(-1, 4)
    off, codes = codes_fn(moves)
    data = assemble(P, np.asarray(off, np.int64), np.asarray(codes, np.uint8), sc)
    return data, {"lines": n_lines, "taps": n_taps, "bytes": len(data)}
