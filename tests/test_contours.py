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


def _pixels_on(contour):
    """Every pixel of a CHAIN_APPROX_SIMPLE contour: consecutive vertices are joined by runs in one of the 8 directions."""
    pts = set()
    n = len(contour)
    for i in range(n):
        (x0, y0), (x1, y1) = contour[i], contour[(i + 1) % n]
        dx, dy = int(np.sign(x1 - x0)), int(np.sign(y1 - y0))
        steps = max(abs(int(x1 - x0)), abs(int(y1 - y0)))
        assert steps == 0 or (abs(int(x1 - x0)) in (0, steps) and abs(int(y1 - y0)) in (0, steps)), (contour[i], contour[(i + 1) % n])
        for s in range(steps + 1):
            pts.add((int(x0) + s * dx, int(y0) + s * dy))
    return pts


@pytest.mark.parametrize("seed", range(10))
def test_borders_and_counts_against_scipy_ndimage(seed):
    """Independent of both tracers (scipy.ndimage, not written by this build). Suzuki & Abe 1985, the algorithm behind
    cv2.findContours: with 8-connected foreground a border point is a 1-pixel with a 0-pixel among its FOUR neighbours (the
    frame counts as 0), every border point lies on a followed border, and RETR_TREE returns one contour per 8-connected
    foreground component plus one per hole (a 4-connected background component that does not reach the frame)."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(10, 60)), int(rng.integers(10, 60))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    m = np.zeros((h, w), bool)
    for _ in range(int(rng.integers(1, 7))):
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(2, 12)
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    if seed % 3:
        m ^= rng.uniform(0, 1, (h, w)) < 0.06
    cross = ndi.generate_binary_structure(2, 1)
    border = m & ~ndi.binary_erosion(m, structure=cross, border_value=0)
    n_fg = ndi.label(m, structure=np.ones((3, 3), int))[1]
    padded = np.pad(~m, 1, constant_values=True)                      # the frame joins every outside region
    n_holes = ndi.label(padded, structure=cross)[1] - 1
    for tracer in (find_contours, ref_contours):
        cs = tracer(m)
        assert len(cs) == n_fg + n_holes, (len(cs), n_fg, n_holes)
        got = set()
        for c in cs:
            got |= _pixels_on(np.asarray(c).reshape(-1, 2))
        want = {(int(x), int(y)) for y, x in zip(*np.nonzero(border))}
        assert got == want
