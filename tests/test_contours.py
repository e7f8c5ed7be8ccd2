"""Contour tracer (td_find_contours, host code of libtreedet_hip.so) vs the pure-Python restatement and closed-form
cases. CPU only: the function touches no GPU."""
import numpy as np
import pytest

from oracle.contours_ref import find_contours as ref_contours
from treedetection_amd.contours import find_contours, xy


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def test_rectangle_gives_four_corners():
    m = np.zeros((10, 12), np.uint8)
    m[2:7, 3:9] = 1
    c = find_contours(m)
    assert len(c) == 1
    # cv2 order: start at the top-left pixel, then down the left side (counter-clockwise in image coordinates)
    assert c[0].tolist() == [[3, 2], [3, 6], [8, 6], [8, 2]]


def test_single_pixel_and_line():
    m = np.zeros((5, 5), np.uint8)
    m[2, 2] = 1
    assert [c.tolist() for c in find_contours(m)] == [[[2, 2]]]
    m = np.zeros((5, 7), np.uint8)
    m[2, 1:6] = 1
    assert [c.tolist() for c in find_contours(m)] == [[[1, 2], [5, 2]]]


def test_hole_is_child_and_order_is_most_recent_first():
    m = np.zeros((20, 30), np.uint8)
    m[1:8, 2:12] = 1          # blob A (found first) with a hole
    m[3:6, 5:9] = 0
    m[10:18, 15:28] = 1       # blob B (found later)
    c = find_contours(m)
    assert len(c) == 3
    assert c[0].tolist() == [[15, 10], [15, 17], [27, 17], [27, 10]]      # B first (most recently found sibling)
    assert c[1].tolist() == [[2, 1], [2, 7], [11, 7], [11, 1]]            # then A ...
    hole = c[2]                                                           # ... then A's hole, traced on A's pixels
    assert hole[:, 0].min() == 4 and hole[:, 0].max() == 9 and hole[:, 1].min() == 2 and hole[:, 1].max() == 6


def test_full_image_and_empty():
    assert find_contours(np.zeros((4, 4), np.uint8)) == []
    c = find_contours(np.ones((4, 6), np.uint8))
    assert [x.tolist() for x in c] == [[[0, 0], [0, 3], [5, 3], [5, 0]]]


@pytest.mark.parametrize("seed", range(12))
def test_matches_python_restatement_on_random_blobs(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(8, 40)), int(rng.integers(8, 40))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    m = np.zeros((h, w), bool)
    for _ in range(int(rng.integers(1, 5))):
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(2, 9)
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    m ^= rng.uniform(0, 1, (h, w)) < (0.08 if seed % 2 else 0.0)      # salt-and-pepper: holes, specks, diagonals
    assert same(find_contours(m), ref_contours(m))


def test_noise_image_matches_python_restatement():
    rng = np.random.default_rng(99)
    m = rng.uniform(0, 1, (25, 31)) < 0.5
    assert same(find_contours(m), ref_contours(m))


def test_xy_is_corner_based_affine():
    t = [0.2, 0.0, 412000.0, 0.0, -0.2, 5319000.0]
    x, y = xy(t, rows=[0, 10, 5], cols=[0, 0, 20])
    assert np.allclose(x, [412000.0, 412000.0, 412004.0]) and np.allclose(y, [5319000.0, 5318998.0, 5318999.0])
    assert x.dtype == np.float64
