"""BASELINE.json's full configuration (4096 x 4096, 8 layers).

* test_c3_digests_vs_oracle: every artefact of every layer (labels, masks, edges, skeleton, contours, scaled, sorted, lines / taps
  after 08 and after 10, ops) hashed on the device's output and compared with the SHA-256 digests the ORACLE produced for the same
  image (tests/golden/c3_digests.json, written once in the build container by tests/golden/make_fullsize_digests.py: the oracle
  needs ~20 minutes for this image, so it cannot run inside the test).  Bit-exact parity of the headline configuration.

Size-independent properties, no oracle involved:

* schedule independence: the per-layer pipelines (default), one-layer-at-a-time execution and the single-workgroup
  k-means must all produce byte-identical lines, taps and ops for every layer (checksums of every artefact);
* stage 04: every contour point is a skeleton pixel of its layer, consecutive points are 8-neighbours, >= 5 points (04:224);
* stage 05: every scaled point lies inside the margins of the canvas (05:63-96);
* stage 12: the ops of a layer are a permutation of its lines (each exactly once, possibly flipped) and taps (12:85-187).
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H = W = 4096
K = 8


@pytest.fixture(scope="module")
def setup():
    from orip.config import Config
    from orip.device import Device
    from orip.synth import synth_image, layer_names
    img = synth_image(H, W, K)
    cfg = Config(); cfg.color_names = layer_names(K)
    dev = Device(0)
    yield dev, cfg, img
    dev.close()


def _digest(dev, cfg):
    from orip import lib as L, stages as S
    R = S.r_insert12(cfg)
    out = {}
    for g in range(K):
        h = hashlib.sha256()
        off, pts = dev.get_polys_flat(L.SLOT_LINES_CROSS, g)
        h.update(off.tobytes()); h.update(np.ascontiguousarray(pts).tobytes())
        h.update(np.asarray(dev.get_taps(L.TAPS_CROSS, g), np.int32).tobytes())
        h.update(np.ascontiguousarray(dev.plot_order(g, R)).tobytes())
        out[g] = h.hexdigest()
    return out


def test_schedules_and_kernel_variants_agree(setup, monkeypatch):
    from orip import parallel as P
    dev, cfg, img = setup
    digests = []
    # (ORIP_KMEANS_1WG, ORIP_THIN_BYTES, ORIP_CUM_CHAIN select replaced kernels that only the variants build carries -- `make -C csrc variants`,
    # ORIP_LIB_VARIANTS=1 -- and read as not set with the default library; the other switches force paths the default library falls back to)
    for env in [{}, {"ORIP_SERIAL_LAYERS": "1"}, {"ORIP_KMEANS_1WG": "1", "ORIP_TAIL_SEQ": "1", "ORIP_NN_NOGRID": "1", "ORIP_PLOT_1WG": "1", "ORIP_TAPS_1WG": "1", "ORIP_MORPH_BYTES": "1", "ORIP_THIN_BYTES": "1", "ORIP_CCL_BYTES": "1", "ORIP_HASH_SORT": "1", "ORIP_NO_CHAINS": "1", "ORIP_NO_PREFETCH08": "1", "ORIP_CAPS_FULL": "1", "ORIP_CAPPREV_SCAN": "1", "ORIP_CUM_CHAIN": "1", "ORIP_NN_NOASM": "1", "ORIP_ARC_POINTS": "1", "ORIP_ZS_LAUNCHES": "1"}]:
        for k in ("ORIP_SERIAL_LAYERS", "ORIP_KMEANS_1WG", "ORIP_TAIL_SEQ", "ORIP_NN_NOGRID", "ORIP_PLOT_1WG", "ORIP_TAPS_1WG", "ORIP_MORPH_BYTES", "ORIP_THIN_BYTES", "ORIP_CCL_BYTES", "ORIP_HASH_SORT", "ORIP_NO_CHAINS", "ORIP_NO_PREFETCH08", "ORIP_CAPS_FULL", "ORIP_CAPPREV_SCAN", "ORIP_CUM_CHAIN", "ORIP_NN_NOASM", "ORIP_ARC_POINTS", "ORIP_ZS_LAUNCHES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev.set_image(img)
        P.run_path_sharded(dev, cfg, H, W, 0, 1)
        digests.append(_digest(dev, cfg))
    assert digests[0] == digests[1] == digests[2]


def test_contours_lie_on_the_skeleton_and_scaled_points_on_the_canvas(setup):
    from orip import lib as L, parallel as P
    from orip.config import canvas_size_px, margins_px
    dev, cfg, img = setup
    dev.set_image(img)
    P.run_path_sharded(dev, cfg, H, W, 0, 1)
    sizes = [dev.polys_size(L.SLOT_CONTOURS, l)[1] for l in range(K)]
    layer = int(np.argmin(sizes))                          # the layer with the fewest contour points (tens of MB on the host)
    off, pts = dev.get_polys_flat(L.SLOT_CONTOURS, layer)
    skel = dev.get_skeleton(layer)
    assert pts[:, 0].min() >= 0 and pts[:, 0].max() < W and pts[:, 1].min() >= 0 and pts[:, 1].max() < H
    assert skel[pts[:, 1], pts[:, 0]].all()                                  # walks only visit skeleton pixels (04:137-205)
    lens = np.diff(off)
    assert lens.min() >= 5                                                   # vectorize_layer keeps paths of >= 5 points (04:224)
    d = np.abs(np.diff(pts.astype(np.int64), axis=0))
    inner = np.ones(len(pts) - 1, bool); inner[off[1:-1] - 1] = False        # steps inside a path, not across two paths
    d_in = d[inner]
    # a path may end by re-appending its start (04:201-203): that closing step is < 1.5 px too, so every inner step is an 8-neighbour move
    assert d_in.max() <= 1
    cw, ch = canvas_size_px(cfg); ml, mr, mt, mb = margins_px(cfg)
    so, sp = dev.get_polys_flat(L.SLOT_SCALED, layer)
    assert np.array_equal(so, off)
    assert sp[:, 0].min() >= ml and sp[:, 0].max() <= cw - mr and sp[:, 1].min() >= mt and sp[:, 1].max() <= ch - mb


def test_ops_are_a_permutation_of_lines_and_taps(setup):
    from orip import lib as L, parallel as P, stages as S
    dev, cfg, img = setup
    dev.set_image(img)
    P.run_path_sharded(dev, cfg, H, W, 0, 1)
    R = S.r_insert12(cfg)
    for g in range(K):
        n_lines = dev.polys_size(L.SLOT_LINES_CROSS, g)[0]
        taps = dev.get_taps(L.TAPS_CROSS, g)
        ops = dev.plot_order(g, R)
        line_ops = ops[ops[:, 0] == 0]; tap_ops = ops[ops[:, 0] == 1]
        assert sorted(line_ops[:, 1].tolist()) == list(range(n_lines))
        assert sorted(map(tuple, tap_ops[:, 3:5].tolist())) == sorted(taps)
        assert np.array_equal(dev.plot_order(g, R), ops)                     # same input, same order


def _sha_polys(off, pts):
    h = hashlib.sha256(); h.update(np.ascontiguousarray(off, np.int64).tobytes()); h.update(np.ascontiguousarray(pts, np.int32).tobytes())
    return {"n": int(len(off) - 1), "points": int(len(pts)), "sha256": h.hexdigest()}


def _sha_taps(taps):
    a = np.ascontiguousarray(np.asarray(list(taps), np.int32).reshape(-1, 2))
    return {"n": int(len(a)), "sha256": hashlib.sha256(a.tobytes()).hexdigest()}


def test_c3_digests_vs_oracle(setup):
    """The headline configuration against the oracle: same image, same artefacts, same hash (see the module docstring)."""
    import json
    from orip import lib as L, parallel as P, stages as S
    dev, cfg, img = setup
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_digests.json")) as f:
        G = json.load(f)
    assert G["config"]["H"] == H and G["config"]["W"] == W and G["config"]["K"] == K
    assert hashlib.sha256(img.tobytes()).hexdigest() == G["image_sha256"]          # same synthetic image on this box
    dev.set_image(img)
    P.run_path_sharded(dev, cfg, H, W, 0, 1)
    assert hashlib.sha256(dev.get_labels().tobytes()).hexdigest() == G["labels_sha256"]
    names = S.cluster_names(cfg)
    assert names == G["device_layer_names"]
    R = S.r_insert12(cfg)
    ops_all = {}
    for l, n in enumerate(names):
        want = G["layers"][n]
        assert hashlib.sha256(dev.get_mask(l).tobytes()).hexdigest() == want["mask"], ("mask", n)
        assert hashlib.sha256(dev.get_edges(l).tobytes()).hexdigest() == want["edges"], ("edges", n)
        assert hashlib.sha256(dev.get_skeleton(l).tobytes()).hexdigest() == want["skeleton"], ("skeleton", n)
        for key, slot in (("contours", L.SLOT_CONTOURS), ("scaled", L.SLOT_SCALED), ("sorted", L.SLOT_SORTED), ("lines_intra", L.SLOT_LINES_INTRA), ("lines_cross", L.SLOT_LINES_CROSS)):
            off, pts = dev.get_polys_flat(slot, l)
            got = _sha_polys(off, pts); del off, pts
            assert got == want[key], (key, n, got, want[key])
        assert _sha_taps(dev.get_taps(L.TAPS_INTRA, l)) == want["taps_intra"], ("taps_intra", n)
        assert _sha_taps(dev.get_taps(L.TAPS_CROSS, l)) == want["taps_cross"], ("taps_cross", n)
        raw = dev.plot_order(l, R)
        assert {"n": int(len(raw)), "sha256": hashlib.sha256(np.ascontiguousarray(raw, np.int32).tobytes()).hexdigest()} == want["ops"], ("ops", n)
        ops_all[n] = S.ops_from_device(dev, l, R)
    # north_star parity quantity (iii): total plotted path length, computed as the oracle computes it (12:71-80)
    from oracle import oracle as O
    draw, travel = O.path_length(ops_all)
    assert abs(draw - G["path_length_px"]["draw"]) <= 1e-3 * G["path_length_px"]["draw"]
    assert abs(travel - G["path_length_px"]["travel"]) <= 1e-3 * G["path_length_px"]["travel"]
