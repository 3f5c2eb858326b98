"""process_colors.py on the GPU: orip_assign_palette against the reference's own labels (golden_colors.npz) and the oracle,
orip_kmeans_fit_rgb against the oracle's cv2.kmeans restatement (parity unpinned, as stage 02's), and the drop-in tool's files."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "golden_colors.npz"))


@pytest.fixture(scope="module")
def dev():
    from orip.device import Device
    d = Device(0)
    yield d
    d.close()


@pytest.mark.parametrize("n", range(5))
def test_assign_palette_matches_reference(dev, n):
    rgb = G[f"assign_img_{n}"]
    dev.set_image(np.ascontiguousarray(rgb[:, :, ::-1]))
    labels, counts = dev.assign_palette(G[f"assign_pal_{n}"])
    assert np.array_equal(labels, G[f"assign_lab_{n}"])
    assert counts.tolist() == np.bincount(labels.ravel(), minlength=len(counts)).tolist()
    assert np.array_equal(dev.get_labels(), labels)


@pytest.mark.parametrize("shape", [(301, 403), (1024, 1536), (7, 5), (1, 3)])
def test_assign_palette_vs_oracle_odd_sizes(dev, shape):
    rng = np.random.default_rng(shape[0])
    rgb = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    pal = rng.integers(0, 256, (9, 3), dtype=np.uint8)
    dev.set_image(np.ascontiguousarray(rgb[:, :, ::-1]))
    labels, _ = dev.assign_palette(pal)
    assert np.array_equal(labels, O.assign_labels_rgb(rgb, pal))


@pytest.mark.parametrize("case", [(300, 400, 4), (700, 900, 6)], ids=lambda c: "x".join(map(str, c)))
def test_kmeans_palette_vs_oracle(dev, case):
    """below and above the 200 000-pixel subsample limit"""
    from orip import colors as PC
    from orip.synth import synth_image
    h, w, k = case
    bgr = synth_image(h, w, k, seed=4, sigma=7.0)
    dev.set_image(bgr)
    got = PC.kmeans_palette(dev, k)
    want = O.kmeans_palette_rgb(bgr[:, :, ::-1], k)
    assert got.dtype == np.uint8 and np.array_equal(got, want)


def test_tool_writes_the_reference_file_set(tmp_path):
    from PIL import Image
    from orip.synth import synth_image
    bgr = synth_image(240, 320, 4, seed=8, sigma=6.0)
    src = tmp_path / "in.png"; Image.fromarray(bgr[:, :, ::-1]).save(src)
    out = tmp_path / "layers"
    tool = os.path.join(ROOT, "omnirevolve-image-processor_amd", "stages", "process_colors.py")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "omnirevolve-image-processor_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, tool, str(src), "-o", str(out), "-n", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pal = O.kmeans_palette_rgb(bgr[:, :, ::-1], 4)
    labels = O.assign_labels_rgb(bgr[:, :, ::-1], pal)
    assert np.array_equal(np.load(out / "labels.npy"), labels) and np.array_equal(np.array(Image.open(out / "labels.png")), labels)
    dump = json.loads((out / "palette.json").read_text())
    assert [c["rgb"] for c in dump["colors"]] == pal.tolist() and [c["name"] for c in dump["colors"]] == ["red", "green", "blue", "black"]
    for i, nm in enumerate(["red", "green", "blue", "black"]):
        assert np.array_equal(np.array(Image.open(out / f"layer_{i+1}_{nm}.png")), (labels == i).astype(np.uint8) * 255)
    # palette mode
    pj = tmp_path / "pal.json"
    pj.write_text(json.dumps({"recommended_colors": [{"name": "b", "rgb": [200, 30, 40], "position": 2}, {"name": "a", "rgb": [10, 10, 10], "position": 1}]}))
    out2 = tmp_path / "layers2"
    r = subprocess.run([sys.executable, tool, str(src), "-o", str(out2), "-m", "palette", "--palette", str(pj)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "--colors=4 ignored; palette has 2 entries" in r.stdout
    want = O.assign_labels_rgb(bgr[:, :, ::-1], np.array([[10, 10, 10], [200, 30, 40]], np.uint8))
    assert np.array_equal(np.load(out2 / "labels.npy"), want) and (out2 / "layer_1_a.png").exists() and (out2 / "layer_2_b.png").exists()
