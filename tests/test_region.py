"""Outline predicates (td_region_relate: helpers.py:791-797 fuse_predictions, preprocessing.py:86-93 tile flags):
hand-derived cases for the DE-9IM meaning of intersects / within, and the C++ index path against the exact brute-force
oracle on random configurations."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import region_ref as O  # noqa: E402
from treedetection_amd.vector import Region, box_ring  # noqa: E402


def sq(x0, y0, x1, y1):
    return np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1], [x0, y0]], float)


FOREST = [
    [sq(0, 0, 10, 10), sq(4, 4, 6, 6)],          # A: square with a hole
    [sq(10, 0, 20, 10)],                         # B: shares the edge x = 10 with A
    [sq(30, 0, 40, 10)],                         # C: apart
    [sq(4.5, 4.5, 5.5, 5.5)],                    # D: an island inside A's hole
]

CASES = [
    ("inside A", sq(1, 1, 2, 2), True, True),
    ("outside everything", sq(22, 1, 28, 2), False, False),
    ("across the shared edge of A and B", sq(8, 1, 12, 2), True, True),
    ("across A's outer edge", sq(-1, 1, 1, 2), True, False),
    ("touching A from outside along an edge", sq(-2, 1, 0, 2), True, False),
    ("touching A from outside at a corner", sq(-2, -2, 0, 0), True, False),
    ("inside the hole, clear of the island", sq(4.1, 4.1, 4.4, 4.4), False, False),
    ("inside the hole, touching its rim", sq(4, 4.1, 4.3, 4.4), True, False),
    ("inside the island", sq(4.8, 4.8, 5.2, 5.2), True, True),
    ("covering the whole hole", sq(3, 3, 7, 7), True, False),
    ("covering island and part of the hole", sq(4.3, 4.3, 5.7, 5.7), True, False),
    ("A and B together, exactly", sq(0, 0, 20, 10), True, False),          # contains A's hole
    ("inside B up to its boundary", sq(10, 0, 20, 10), True, True),
    ("spanning the gap between B and C", sq(15, 1, 35, 2), True, False),
    ("containing C entirely", sq(29, -1, 41, 11), True, False),
    ("a sliver along the shared edge", sq(9.9, 2, 10.1, 8), True, True),
    ("vertex on A's boundary, rest inside", np.array([[0, 5], [2, 4], [2, 6], [0, 5]], float), True, True),
]


@pytest.mark.parametrize("name,query,want_i,want_w", CASES, ids=[c[0] for c in CASES])
def test_known_answers(name, query, want_i, want_w):
    assert O.relate(FOREST, query) == (want_i, want_w)
    i, w = Region(FOREST).relate([query])
    assert (bool(i[0]), bool(w[0])) == (want_i, want_w)


def test_batch_and_empty():
    r = Region(FOREST)
    qs = [c[1] for c in CASES]
    i, w = r.relate(qs)
    assert i.tolist() == [c[2] for c in CASES] and w.tolist() == [c[3] for c in CASES]
    i, w = Region([]).relate(qs)
    assert not i.any() and not w.any()
    assert Region(FOREST).relate([])[0].shape == (0,)
    assert box_ring(0, 1, 2, 3).tolist() == [[2, 1], [2, 3], [0, 3], [0, 1], [2, 1]]


def _blob(rng, cx, cy, r, n):
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = r * rng.uniform(0.6, 1.0, n)
    ring = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1)
    return np.concatenate([ring, ring[:1]])


@pytest.mark.parametrize("seed", range(4))
def test_matches_oracle_on_random_configurations(seed):
    """Star-shaped forest patches (some with holes, overlapping each other) on a 0.25-unit lattice — so vertices on
    edges, shared vertices and collinear overlaps do occur — against small query rings on the same lattice."""
    rng = np.random.default_rng(seed)
    forest = []
    for _ in range(8):
        cx, cy = rng.uniform(0, 40, 2)
        shell = np.round(_blob(rng, cx, cy, rng.uniform(4, 10), int(rng.integers(5, 12))) * 4) / 4
        poly = [shell]
        if rng.random() < 0.5:
            poly.append(np.round(_blob(rng, cx, cy, 1.5, 5) * 4) / 4)
        forest.append(poly)
    queries = []
    for _ in range(70):
        cx, cy = rng.uniform(-2, 42, 2)
        q = np.round(_blob(rng, cx, cy, rng.uniform(0.5, 3), int(rng.integers(3, 8))) * 4) / 4
        if abs(np.dot(q[:-1, 0], q[1:, 1]) - np.dot(q[1:, 0], q[:-1, 1])) > 0:
            queries.append(q)
    gi, gw = Region(forest).relate(queries)
    want = [O.relate(forest, q) for q in queries]
    assert gi.tolist() == [w[0] for w in want]
    assert gw.tolist() == [w[1] for w in want]
    assert 0 < sum(gi) < len(queries) and 0 < sum(gw) < sum(gi)          # all three outcomes occur
