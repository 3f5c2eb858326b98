"""GPU parity tests (call through the C ABI of liborip.so) for the raster half of the path:
stage 02 (Lab, k-means fit, assignment, masks), stage 03 (edges), stage 04 (skeleton, contours).
Bit-exact against the CPU restatement (oracle/) on the same seeded inputs and against the golden fixtures."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from util import load, unflat, same_polys


@pytest.fixture(scope="module")
def dev():
    from orip.device import Device
    d = Device(0)
    yield d
    d.close()


def _img(H, W, K, seed=5, sigma=None):
    from orip.synth import synth_image
    return synth_image(H, W, K, seed=seed, sigma=sigma)


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (256, 320)])
def test_lab_conversion_bit_exact(dev, shape):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    dev.set_image(img)
    assert np.array_equal(dev.lab_of(), O.bgr2lab(img))
    idx = rng.choice(shape[0] * shape[1], 500, replace=False)
    assert np.array_equal(dev.lab_of(idx), O.bgr2lab(img).reshape(-1, 3)[idx])


@pytest.mark.parametrize("case", [(128, 128, 4), (200, 150, 8), (300, 300, 2), (96, 96, 16)])
def test_kmeans_fit_matches_oracle(dev, case):
    H, W, K = case
    img = _img(H, W, K, seed=3 + K, sigma=3.0)
    dev.set_image(img)
    lab = O.bgr2lab(img).reshape(-1, 3).astype(np.float32)
    idx = O.subsample_indices(len(lab), limit=20000)
    sample = lab[idx] if idx is not None else lab
    want, comp_w = O.kmeans(sample, K)
    got, comp_g = dev.kmeans_fit(idx, K)
    assert np.array_equal(got, want), (got, want)
    assert abs(comp_g - comp_w) <= 1e-9 * max(1.0, abs(comp_w))


def test_kmeans_random_pixels_matches_oracle(dev):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (90, 110, 3), dtype=np.uint8)   # no cluster structure: exercises many Lloyd iterations
    dev.set_image(img)
    lab = O.bgr2lab(img).reshape(-1, 3).astype(np.float32)
    want, _ = O.kmeans(lab, 5)
    got, _ = dev.kmeans_fit(None, 5)
    assert np.array_equal(got, want)


def test_kmeans_both_kernels_agree(dev, monkeypatch):
    """The fit runs spread over groups of 64 workgroups with a device-wide barrier each; the single-workgroup version it replaced is kept
    behind ORIP_KMEANS_1WG in the variants build (`make -C csrc variants`, ORIP_LIB_VARIANTS=1).  Same centres and the same compactness (fixed
    reduction tree), also with a cluster that empties.  With the default library the switch reads as not set: both fits are the same kernel and
    the comparison with the oracle below is what the test holds."""
    rng = np.random.default_rng(11)
    for img, K in [(rng.integers(0, 256, (120, 150, 3), dtype=np.uint8), 7),
                   (np.repeat(rng.integers(0, 256, (3, 1, 3), dtype=np.uint8), 4000, axis=1).reshape(100, 120, 3), 6)]:
        dev.set_image(np.ascontiguousarray(img))
        monkeypatch.delenv("ORIP_KMEANS_1WG", raising=False)
        a, ca = dev.kmeans_fit(None, K)
        monkeypatch.setenv("ORIP_KMEANS_1WG", "1")
        b, cb = dev.kmeans_fit(None, K)
        monkeypatch.delenv("ORIP_KMEANS_1WG", raising=False)
        assert np.array_equal(a, b)
        assert ca == cb
        lab = O.bgr2lab(np.ascontiguousarray(img)).reshape(-1, 3).astype(np.float32)
        want, _ = O.kmeans(lab, K)
        assert np.array_equal(a, want)


@pytest.mark.parametrize("case", [(128, 128, 4, None), (255, 193, 8, 4.0), (512, 512, 8, None), (64, 1030, 3, 2.0)])
def test_stage02_labels_and_masks(dev, case):
    H, W, K, sigma = case
    from orip.synth import layer_names
    img = _img(H, W, K, sigma=sigma)
    cfg = dict(color_names=layer_names(K))
    lab = O.bgr2lab(img)
    centers, _ = O.kmeans(lab.reshape(-1, 3).astype(np.float32)[:: max(1, H * W // 5000)], max(2, K))
    masks_o, cs_o, labels_o = O.stage02(img, cfg, centers)
    dev.set_image(img)
    cs, counts = dev.extract_layers(centers)
    assert np.array_equal(cs, cs_o)
    assert np.array_equal(dev.get_labels(), labels_o.astype(np.uint8))            # layer assignment: bit exact
    assert np.array_equal(counts, np.bincount(labels_o.ravel(), minlength=len(centers)))
    names_sorted = sorted(cfg["color_names"], key=O.darkness_rank02)
    for l, name in enumerate(names_sorted):
        assert np.array_equal(dev.get_mask(l), masks_o[name]), name


@pytest.mark.parametrize("prm", [dict(), dict(edge_kernel_size=5, edge_low_threshold=22, edge_high_threshold=70),
                                 dict(edge_kernel_size=7, edge_morph_kernel=5), dict(edge_morph_open_iters=2, edge_morph_close_iters=0),
                                 dict(edge_morph_kernel=7, edge_morph_open_iters=1, edge_morph_close_iters=2)])
@pytest.mark.parametrize("shape", [(130, 170), (256, 256), (50, 40)])
def test_stage03_edges(dev, prm, shape):
    rng = np.random.default_rng(4)
    H, W = shape
    from scipy.ndimage import gaussian_filter
    K = 3
    masks = np.stack([(gaussian_filter(rng.standard_normal((H, W)), 3.0 + k) > 0.02 * k).astype(np.uint8) * 255 for k in range(K)])
    cfg = dict(O.DEFAULTS); cfg.update(prm)
    dev.set_masks(masks)
    dev.detect_edges(cfg["edge_morph_kernel"], cfg["edge_morph_open_iters"], cfg["edge_morph_close_iters"],
                     O.ensure_odd(cfg["edge_kernel_size"]), cfg["edge_low_threshold"], cfg["edge_high_threshold"])
    for k in range(K):
        assert np.array_equal(dev.get_edges(k), O.stage03(masks[k], cfg)), (prm, k)


def test_stage03_grayscale_masks_general_u8(dev):
    """masks with values other than {0,255} (a hand-edited mask.png): min/max morphology + Canny on grey levels."""
    rng = np.random.default_rng(6)
    from scipy.ndimage import gaussian_filter
    m = gaussian_filter(rng.random((120, 140)) * 255, 2.5)
    m = ((m - m.min()) / (m.max() - m.min()) * 255).astype(np.uint8)
    dev.set_masks(m[None])
    dev.detect_edges(3, 1, 1, 3, 20, 60)
    cfg = dict(O.DEFAULTS, edge_low_threshold=20, edge_high_threshold=60)
    assert np.array_equal(dev.get_edges(0), O.stage03(m, cfg))


@pytest.mark.parametrize("t", range(4))
def test_stage04_golden_pure(dev, t):
    G = load("golden_pure.npz")
    edges = G[f"thin{t}_in"]
    dev.set_edges(edges[None])
    dev.find_contours()
    assert np.array_equal(dev.get_skeleton(0), G[f"thin{t}_out"])                  # reference's thinning, bit exact
    want = [p for p in unflat(G, f"trace{t}") if len(p) >= 5]                       # 04:224 filter
    from orip.lib import SLOT_CONTOURS
    assert same_polys(dev.get_polys(SLOT_CONTOURS, 0), want)                        # reference's own trace output


@pytest.mark.parametrize("tag", ["a", "b"])
def test_stage04_golden_e2e(dev, tag):
    import json
    from orip.lib import SLOT_CONTOURS
    G = load(f"golden_e2e_{tag}.npz")
    cfg = json.loads(bytes(G["cfg_json"]).decode()); H, W = G["img"].shape[:2]
    names = cfg["color_names"]
    edges = np.stack([np.unpackbits(G[f"edges_{n}"])[:H * W].reshape(H, W) * 255 for n in names]).astype(np.uint8)
    dev.set_edges(edges)
    dev.find_contours()
    for l, n in enumerate(names):
        assert same_polys(dev.get_polys(SLOT_CONTOURS, l), unflat(G, f"contours_{n}")), n


@pytest.mark.parametrize("case", [(200, 260, 4, None), (512, 512, 8, None), (384, 640, 5, 6.0)])
def test_stage02_to_04_chain_vs_oracle(dev, case):
    """Resident chain image -> contours, all layers, bit-exact contour lists (order included) vs the oracle."""
    from orip.lib import SLOT_CONTOURS
    from orip.synth import layer_names
    H, W, K, sigma = case
    img = _img(H, W, K, seed=21, sigma=sigma)
    cfg = dict(color_names=layer_names(K))
    r = O.run_pipeline(img, cfg, upto=4)
    dev.set_image(img)
    idx = O.subsample_indices(H * W)
    centers, _ = dev.kmeans_fit(idx, max(2, K))
    dev.extract_layers(centers)
    dev.detect_edges()
    dev.find_contours()
    names_sorted = sorted(cfg["color_names"], key=O.darkness_rank02)
    for l, n in enumerate(names_sorted):
        assert np.array_equal(dev.get_edges(l), r["edges"][n]), n
        got = dev.get_polys(SLOT_CONTOURS, l)
        assert same_polys(got, r["contours"][n]), (n, len(got), len(r["contours"][n]))


def test_empty_and_degenerate_edges(dev):
    from orip.lib import SLOT_CONTOURS
    e = np.zeros((2, 40, 50), np.uint8)
    e[1, 10, 10] = 255                       # a single pixel: no path of >= 2 points
    dev.set_edges(e)
    dev.find_contours()
    assert dev.get_polys(SLOT_CONTOURS, 0) == [] and dev.get_polys(SLOT_CONTOURS, 1) == []
    e = np.zeros((1, 9, 20), np.uint8); e[0, 4, 2:18] = 255    # a straight line touching nothing
    dev.set_edges(e); dev.find_contours()
    assert same_polys(dev.get_polys(SLOT_CONTOURS, 0), O.stage04(e[0]))


ODD_SHAPES = [(1, 1), (1, 70), (70, 1), (2, 2), (3, 63), (5, 64), (5, 65), (17, 127), (33, 129), (64, 191), (9, 257)]


@pytest.mark.parametrize("shape", ODD_SHAPES)
def test_stage04_odd_shapes(dev, shape):
    """Widths around the 64-pixel word boundaries of the bit planes, one-pixel-wide and one-pixel-high edge maps, several
    densities (dense maps exercise thinning, sparse ones the isolated-pixel and short-path rules)."""
    from orip.lib import SLOT_CONTOURS
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    dens = [0.08, 0.35, 0.7, 1.0]
    e = np.stack([(rng.random((H, W)) < d).astype(np.uint8) * 255 for d in dens])
    dev.set_edges(e)
    dev.find_contours()
    for l in range(len(dens)):
        want = O.stage04(e[l])
        got = dev.get_polys(SLOT_CONTOURS, l)
        assert same_polys(got, want), (shape, dens[l], len(got), len(want))


@pytest.mark.parametrize("shape", ODD_SHAPES)
def test_stage03_odd_shapes(dev, shape):
    H, W = shape
    rng = np.random.default_rng(H * 77 + W)
    from scipy.ndimage import gaussian_filter
    masks = np.stack([(gaussian_filter(rng.standard_normal((H, W)), 1.5, mode="nearest") > t).astype(np.uint8) * 255 for t in (-0.1, 0.0, 0.1)])
    cfg = dict(O.DEFAULTS)
    dev.set_masks(masks)
    dev.detect_edges(cfg["edge_morph_kernel"], cfg["edge_morph_open_iters"], cfg["edge_morph_close_iters"],
                     O.ensure_odd(cfg["edge_kernel_size"]), cfg["edge_low_threshold"], cfg["edge_high_threshold"])
    for k in range(3):
        assert np.array_equal(dev.get_edges(k), O.stage03(masks[k], cfg)), (shape, k)
