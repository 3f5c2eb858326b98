"""BASELINE.json's configurations against the oracle, artefact by artefact, on the DEFAULT A4 canvas (8400 x 11880):

  C1  512 x 512, 4 layers, full path 02 -> 12            (oracle at test time)
      512 x 512, 8 layers, full path                     (the bench image's top-left crop: what bench.py's cpu_baseline times)
  C2  2048 x 2048, 8 layers, stages 02 + 03              (oracle at test time)
  C3  4096 x 4096, 8 layers, full path                   (tests/test_gpu_fullsize.py: SHA-256 digests the oracle left in tests/golden/c3_digests.json)
  K = 16 (C5's layer count) on a small image, full path
  a 2-D grey image (_ensure_bgr, 02:25-30)

Everything is bit-exact: labels, masks, edges, contour / line lists (order included), taps, ops; the plotted path length
(north_star tolerance 1e-3 relative) follows from that and is asserted as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from util import same_polys


@pytest.fixture(scope="module")
def dev():
    from orip.device import Device
    d = Device(0)
    yield d
    d.close()


def _cfgobj(d):
    from orip.config import Config
    c = Config()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def _compare_resident(dev, cfgd, want, upto=12):
    """every artefact left on the device by S.run_path against the oracle's run_pipeline result"""
    from orip import lib as L, stages as S
    cfg = _cfgobj(cfgd)
    lnames = S.cluster_names(cfg)
    assert np.array_equal(dev.get_labels(), want["labels"].astype(np.uint8))
    for l, n in enumerate(lnames):
        assert np.array_equal(dev.get_mask(l), want["masks"][n]), ("mask", n)
        if upto >= 3:
            assert np.array_equal(dev.get_edges(l), want["edges"][n]), ("edges", n)
        if upto < 12:
            continue
        assert same_polys(dev.get_polys(L.SLOT_CONTOURS, l), want["contours"][n]), ("contours", n)
        assert same_polys(dev.get_polys(L.SLOT_SCALED, l), want["scaled"][n]), ("scaled", n)
        assert same_polys(dev.get_polys(L.SLOT_SORTED, l), want["sorted"][n]), ("sorted", n)
        assert same_polys(dev.get_polys(L.SLOT_LINES_INTRA, l), want["intra"][n][0]), ("lines_intra", n)
        assert dev.get_taps(L.TAPS_INTRA, l) == want["intra"][n][1], ("taps_intra", n)
        assert same_polys(dev.get_polys(L.SLOT_LINES_CROSS, l), want["cross"][n][0]), ("lines_cross", n)
        assert dev.get_taps(L.TAPS_CROSS, l) == want["cross"][n][1], ("taps_cross", n)


def _compare_ops(ops, want_ops, names):
    for n in names:
        assert len(ops[n]) == len(want_ops[n]), n
        for a, b in zip(ops[n], want_ops[n]):
            assert a["type"] == b["type"]
            if a["type"] == "line":
                assert np.array_equal(a["points"], b["points"]), n
            else:
                assert (a["x"], a["y"]) == (b["x"], b["y"]), n
    dg, tg = O.path_length(ops); dw, tw = O.path_length(want_ops)
    assert abs(dg + tg - dw - tw) <= 1e-3 * (dw + tw)      # north_star: plotted path length within 1e-3 relative


@pytest.mark.parametrize("case", [(512, 512, 4, "C1"), (512, 512, 8, "bench crop")], ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}")
def test_full_path_default_canvas_vs_oracle(dev, case):
    """C1 and the 8-layer crop of the bench image: stages 02 -> 12 on the default canvas, every artefact."""
    from orip import stages as S
    from orip.synth import synth_image, layer_names
    H, W, K, _ = case
    img = synth_image(4096, 4096, K)[:H, :W] if K == 8 else synth_image(H, W, K)
    img = np.ascontiguousarray(img)
    cfgd = dict(O.DEFAULTS, color_names=layer_names(K))
    want = O.run_pipeline(img, cfgd)
    ops = S.run_path(img, _cfgobj(cfgd), dev)
    _compare_resident(dev, cfgd, want)
    _compare_ops(ops, want["ops"], cfgd["color_names"])


def test_c2_2048x2048x8_stages_02_03_vs_oracle(dev):
    from orip import stages as S
    from orip.synth import synth_image, layer_names
    H = W = 2048; K = 8
    img = synth_image(H, W, K)
    cfgd = dict(O.DEFAULTS, color_names=layer_names(K))
    want = O.run_pipeline(img, cfgd, upto=3)
    S.run_path(img, _cfgobj(cfgd), dev, upto=3)
    _compare_resident(dev, cfgd, want, upto=3)


def test_k16_full_path_vs_oracle(dev):
    """16 colour layers (BASELINE config 5's layer count, ORIP_MAX_LAYERS) through the whole path."""
    from orip import stages as S
    from orip.synth import synth_image, layer_names
    H, W, K = 320, 384, 16
    img = synth_image(H, W, K, seed=5, sigma=6.0)
    cfgd = dict(O.DEFAULTS, color_names=layer_names(K))
    want = O.run_pipeline(img, cfgd)
    ops = S.run_path(img, _cfgobj(cfgd), dev)
    _compare_resident(dev, cfgd, want)
    _compare_ops(ops, want["ops"], cfgd["color_names"])


def test_gray_2d_input_is_expanded_to_bgr(dev):
    """_ensure_bgr (02:25-30): a 2-D image is processed as the BGR image with three equal channels."""
    from orip import stages as S
    from orip.synth import synth_image, layer_names
    K = 4
    gray = np.ascontiguousarray(synth_image(200, 240, K, seed=11, sigma=5.0)[:, :, 1])
    cfgd = dict(O.DEFAULTS, color_names=layer_names(K), pixels_per_mm=8)
    want = O.run_pipeline(gray, cfgd)          # the oracle's stage02 expands 2-D input the same way
    want3 = O.run_pipeline(np.repeat(gray[:, :, None], 3, axis=2), cfgd, upto=2)
    assert np.array_equal(want["labels"], want3["labels"])
    ops = S.run_path(gray, _cfgobj(cfgd), dev)
    _compare_resident(dev, cfgd, want)
    _compare_ops(ops, want["ops"], cfgd["color_names"])
