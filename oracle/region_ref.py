"""TEST INFRASTRUCTURE ONLY — CPU restatement of the outline predicates of the reference
(`fuse_predictions`, TreeDetection/helpers.py:791-797: ``forest_shapes.intersects(forest_union)``,
``urban_shapes.within(forest_union)``; tile flags, preprocessing.py:86-93: ``candidates.intersects(bbox)``,
``unary_union.contains(bbox)``).

shapely/GEOS (absent here, un-pinned in the reference) evaluate these on an overlay of the outline. This oracle
restates their point-set meaning for a simple query ring Q against the union F of polygons with holes, in exact
rational arithmetic and by brute force over all edges:
  intersects: Q and F share a point — a vertex of Q in F, an edge of Q meeting an edge of F, or a ring of F inside Q;
  within: no point of Q outside F — every vertex of Q in F, every piece of Q's boundary between consecutive
  meetings with F's edges has its midpoint in F, and no hole of a polygon lies inside Q uncovered by the others.
**Parity unpinned** (no GEOS to run, no fixture in the reference); pinned instead by the hand-derived cases in
tests/test_region.py.
"""
from __future__ import annotations

from fractions import Fraction as Fr
from typing import List, Sequence, Tuple

Pt = Tuple[Fr, Fr]


def _fr(ring) -> List[Pt]:
    return [(Fr(float(x)), Fr(float(y))) for x, y in ring]


def _orient(a: Pt, b: Pt, p: Pt) -> int:
    d = (b[0] - a[0]) * (p[1] - b[1]) - (b[1] - a[1]) * (p[0] - b[0])
    return (d > 0) - (d < 0)


def _on_seg(a: Pt, b: Pt, p: Pt) -> bool:
    return _orient(a, b, p) == 0 and min(a[0], b[0]) <= p[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= p[1] <= max(a[1], b[1])


def _ring_state(ring: List[Pt], p: Pt) -> int:
    """0 outside, 1 inside, 2 on the ring."""
    inside = False
    for a, b in zip(ring[:-1], ring[1:]):
        if _on_seg(a, b, p):
            return 2
        if (a[1] <= p[1]) != (b[1] <= p[1]):
            o = _orient(a, b, p)
            if (a[1] <= p[1] and o > 0) or (a[1] > p[1] and o < 0):
                inside = not inside
    return 1 if inside else 0


def _in_polygon(poly: List[List[Pt]], p: Pt) -> int:
    """closed membership: 0 outside, 1 interior, 2 boundary."""
    s = _ring_state(poly[0], p)
    if s != 1:
        return s
    for hole in poly[1:]:
        h = _ring_state(hole, p)
        if h == 2:
            return 2
        if h == 1:
            return 0
    return 1


def _member(polys, p: Pt) -> bool:
    return any(_in_polygon(poly, p) for poly in polys)


def _meet_params(p1: Pt, p2: Pt, q1: Pt, q2: Pt) -> List[Fr]:
    a, b = _orient(p1, p2, q1), _orient(p1, p2, q2)
    c, d = _orient(q1, q2, p1), _orient(q1, q2, p2)
    if (a > 0 and b > 0) or (a < 0 and b < 0) or (c > 0 and d > 0) or (c < 0 and d < 0):
        return []
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    len2 = dx * dx + dy * dy

    def param(q):
        if len2 == 0:
            return Fr(0)
        return max(Fr(0), min(Fr(1), ((q[0] - p1[0]) * dx + (q[1] - p1[1]) * dy) / len2))

    if a == 0 and b == 0:
        if max(min(p1[0], p2[0]), min(q1[0], q2[0])) > min(max(p1[0], p2[0]), max(q1[0], q2[0])) or \
                max(min(p1[1], p2[1]), min(q1[1], q2[1])) > min(max(p1[1], p2[1]), max(q1[1], q2[1])):
            return []
        return sorted([param(q1), param(q2)])
    if a == 0:
        return [param(q1)] if _on_seg(p1, p2, q1) else []
    if b == 0:
        return [param(q2)] if _on_seg(p1, p2, q2) else []
    ex, ey = q2[0] - q1[0], q2[1] - q1[1]
    den = dx * ey - dy * ex
    if den == 0:
        return [Fr(0)]
    return [((q1[0] - p1[0]) * ey - (q1[1] - p1[1]) * ex) / den]


def relate(polygons: Sequence[Sequence], query) -> Tuple[bool, bool]:
    """→ (query.intersects(union), query.within(union))."""
    polys = [[_fr(r) for r in poly] for poly in polygons]
    Q = _fr(query)
    if not polys:
        return False, False
    intersects, within = False, True
    for v in Q[:-1]:
        if _member(polys, v):
            intersects = True
        else:
            within = False
    met = set()
    for a, b in zip(Q[:-1], Q[1:]):
        ts = [Fr(0), Fr(1)]
        for pi, poly in enumerate(polys):
            for ri, ring in enumerate(poly):
                for c, d in zip(ring[:-1], ring[1:]):
                    t = _meet_params(a, b, c, d)
                    if t:
                        intersects = True
                        met.add((pi, ri))
                        ts.extend(t)
        ts.sort()
        for t0, t1 in zip(ts[:-1], ts[1:]):
            if t1 > t0:
                tm = (t0 + t1) / 2
                if not _member(polys, (a[0] + tm * (b[0] - a[0]), a[1] + tm * (b[1] - a[1]))):
                    within = False
    for pi, poly in enumerate(polys):
        for ri, ring in enumerate(poly):
            if (pi, ri) in met or _ring_state(Q, ring[0]) == 0:
                continue
            intersects = True
            if ri > 0 and not any(_in_polygon(other, ring[0]) for pj, other in enumerate(polys) if pj != pi):
                within = False
    return intersects, intersects and within
