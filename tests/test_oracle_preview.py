"""Previews 06 / 09 / 11: the coverage planes of the oracle (oracle.preview_cover).  PARITY UNPINNED against cv2.polylines / cv2.circle with LINE_AA
(the reference holds no fixture for them and cv2 is not importable): these tests pin the documented stand-in itself -- what an axis-aligned 1-px line,
a thick line, a disc and the composition rule must give -- and tests/test_gpu_preview.py holds the kernels to the oracle bit by bit."""
import numpy as np

from oracle import oracle as O


def test_axis_aligned_one_pixel_line_is_exactly_its_pixels():
    lines, taps = O.preview_cover([np.array([[3, 4], [12, 4]]), np.array([[5, 1], [5, 9]])], [], 20, 12, 1, 0, True)
    want = np.zeros((12, 20), np.uint8); want[4, 3:13] = 255; want[1:10, 5] = 255
    assert np.array_equal(lines, want) and not taps.any()


def test_diagonal_line_has_a_symmetric_fringe_and_thickness_widens_it():
    l1, _ = O.preview_cover([np.array([[2, 2], [12, 12]])], [], 16, 16, 1, 0, True)
    assert all(l1[i, i] == 255 for i in range(2, 13))
    assert np.array_equal(l1, l1.T) and 0 < l1[5, 6] < 255 and l1[5, 7] == 0            # d = 1 / sqrt 2 -> a = 0.29; d = sqrt 2 -> 0
    l3, _ = O.preview_cover([np.array([[2, 2], [12, 12]])], [], 16, 16, 3, 0, True)
    assert l3[5, 6] == 255 and 0 < l3[5, 7] < 255 and (l3 >= l1).all()
    hard, _ = O.preview_cover([np.array([[2, 2], [12, 12]])], [], 16, 16, 1, 0, False)
    assert set(np.unique(hard)) == {0, 255} and hard[5, 6] == 0


def test_disc_and_clipping():
    _, t = O.preview_cover([], [(4, 6), (-1, 0)], 10, 10, 1, 2, True)
    assert t[6, 4] == 255 and t[6, 5] == 255 and t[6, 6] == 128 and t[6, 7] == 0            # radius 2: d = 2 -> a = 0.5 -> 128, d = 3 -> 0
    assert t[4, 4] == 128 and t[3, 4] == 0 and np.array_equal(t[3:10, 1:8], t[3:10, 1:8].T)
    assert t[0, 0] == 255 and t[0, 1] == 128                                               # a centre off the canvas still covers what lies on it


def test_compose_is_integer_alpha_over_white():
    img = np.full((1, 3, 3), 255, np.uint8)
    out = O.preview_compose(img, np.array([[0, 128, 255]], np.uint8), (10, 20, 250))
    assert out[0, 0].tolist() == [255, 255, 255] and out[0, 2].tolist() == [10, 20, 250]
    assert out[0, 1].tolist() == [(255 * 127 + 10 * 128 + 127) // 255, (255 * 127 + 20 * 128 + 127) // 255, (255 * 127 + 250 * 128 + 127) // 255]
