"""process_colors.py: the oracle's restatement and the host module against the reference's own outputs
(tests/golden/golden_colors.npz, made by tests/golden/make_golden_colors.py from /root/reference) -- no GPU."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from orip import colors as PC

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_colors.npz"))


@pytest.mark.parametrize("n", range(5))
def test_oracle_assign_labels_matches_reference(n):
    got = O.assign_labels_rgb(G[f"assign_img_{n}"], G[f"assign_pal_{n}"])
    assert got.dtype == np.uint8 and np.array_equal(got, G[f"assign_lab_{n}"])


def test_int16_wrap_is_exercised_by_the_fixture():
    """case 1 holds black / white pixels against a palette with 255-differences: plain integer distances give other labels there"""
    img, pal = G["assign_img_1"].astype(np.int64), G["assign_pal_1"].astype(np.int64)
    plain = np.argmin(((img[:, :, None, :] - pal[None, None]) ** 2).sum(-1), axis=-1)
    assert not np.array_equal(plain, G["assign_lab_1"])


@pytest.mark.parametrize("n", range(3))
def test_subsample_matches_reference(n):
    N, samples = (int(v) for v in G[f"sub_N_{n}"])
    idx = PC.subsample_indices(N, samples, 1)
    if N > samples:
        assert np.array_equal(idx, G[f"sub_idx_{n}"])
    else:
        assert idx is None


def test_names_and_palette_files(tmp_path):
    assert PC.default_color_names(6) == list(G["names_6"]) and PC.default_color_names(2) == list(G["names_2"])
    p = tmp_path / "a.json"; p.write_text(str(G["pal_json_a"]))
    rgb, names = PC.palette_from_json(str(p))
    assert np.array_equal(rgb, G["pal_rgb_a"]) and names == list(G["pal_names_a"])
    # the "palette" layout: the reference ends in a NameError (recorded in the fixture); here it loads
    assert str(G["pal_raises_b"]) == "NameError"
    q = tmp_path / "b.json"; q.write_text(str(G["pal_json_b"]))
    rgb, names = PC.palette_from_json(str(q))
    assert rgb.tolist() == [[1, 2, 3], [200, 100, 50]] and names == ["color_0", "x"]
    r = tmp_path / "c.json"; r.write_text(json.dumps({"something": 1}))
    with pytest.raises(ValueError, match="Unsupported palette JSON"):
        PC.palette_from_json(str(r))
    d = PC.palette_dump(np.array([[1, 2, 3], [4, 5, 6]], np.uint8), ["a"])
    assert d == {"colors": [{"index": 0, "name": "a", "rgb": [1, 2, 3]}, {"index": 1, "name": "color_1", "rgb": [4, 5, 6]}]}
