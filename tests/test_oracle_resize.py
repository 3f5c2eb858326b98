"""oracle.resize_area -- the restatement of OpenCV's INTER_AREA shrink (01_resize.py:19).  PARITY UNPINNED: cv2 is absent and the reference holds no
resized image; these are the known answers any area average has to give, plus the documented arithmetic of the integer-ratio path."""
import numpy as np
import pytest

from oracle import oracle as O


def test_constant_image_stays_constant():
    for v in (0, 1, 77, 254, 255):
        a = np.full((301, 403, 3), v, np.uint8)
        assert np.all(O.resize_area(a, 133, 100) == v) and np.all(O.resize_area(a, 13, 7) == v)


def test_two_by_two_is_the_rounded_mean():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    s = a[0::2, 0::2].astype(int) + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2]
    assert np.array_equal(O.resize_area(a, 100, 60), ((s + 2) >> 2).astype(np.uint8))


def test_integer_ratio_is_float_scaled_cell_sum():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (90, 150), dtype=np.uint8)
    s = a.reshape(30, 3, 30, 5).astype(np.int64).sum(axis=(1, 3))
    want = np.rint((s.astype(np.float32) * np.float32(1.0 / 15)).astype(np.float32)).astype(np.uint8)      # rint: half to even, as cvRound
    assert np.array_equal(O.resize_area(a, 30, 30), want)


def test_general_ratio_within_one_level_of_the_exact_area_mean():
    """exact rational area average of every destination cell (float64): the float32 table arithmetic may round the other way at a tie, never more"""
    rng = np.random.default_rng(3)
    H, W, nh, nw = 37, 53, 10, 17
    a = rng.integers(0, 256, (H, W), dtype=np.uint8)
    got = O.resize_area(a, nw, nh).astype(np.float64)
    sx, sy = W / nw, H / nh

    def weights(n_src, n_dst, s):
        Wm = np.zeros((n_dst, n_src))
        for d in range(n_dst):
            lo, hi = d * s, min((d + 1) * s, n_src)
            for i in range(int(np.floor(lo)), int(np.ceil(hi))):
                Wm[d, i] = max(0.0, min(hi, i + 1) - max(lo, i))
            Wm[d] /= Wm[d].sum()
        return Wm
    exact = weights(H, nh, sy) @ a.astype(np.float64) @ weights(W, nw, sx).T
    assert np.max(np.abs(got - exact)) <= 0.5 + 1e-3


def test_rejects_enlarging():
    with pytest.raises(ValueError):
        O.resize_area(np.zeros((10, 10, 3), np.uint8), 11, 5)


def test_stage_rule():
    a = np.zeros((3000, 2500, 3), np.uint8)
    assert O.resize_if_needed(a).shape == (2000, 1666, 3) and O.resize_if_needed(a[:2000, :100]) is not None
    b = np.zeros((2000, 100, 3), np.uint8)
    assert O.resize_if_needed(b) is b
