"""TEST INFRASTRUCTURE ONLY — CPU restatement of the geometry steps of the reference's stitching consumer.

``process_prediction_file_sync`` (TreeDetection/helpers.py:419-476) simplifies every crown polygon with
``GeoSeries.simplify(tolerance, preserve_topology=True)`` (464-465) and keeps those ``within`` the tile's shrunken
bounding box (``box_filter`` 305-319 / ``box_make`` 281-303, ``gpd.sjoin(..., "inner", "within")`` 469).

shapely/GEOS (the libraries that execute those calls) are third-party dependencies, absent from /root/reference and
from this image (no pinned version in the reference; shapely 2.x wheels bundle GEOS 3.11–3.13). The functions below
restate the published algorithms — GEOS ``TopologyPreservingSimplifier`` / ``TaggedLineStringSimplifier`` for a single
closed shell, and the DE-9IM meaning of *within* for a polygon against an axis-parallel box — with exact rational
arithmetic for the orientation predicate (GEOS uses a double-double evaluation; both give the true sign).
**Parity unpinned**: there is no reference-side fixture or runnable GEOS to check this restatement against.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import List, Sequence, Tuple

Pt = Tuple[float, float]


def orientation(p1: Pt, p2: Pt, q: Pt) -> int:
    d = (Fraction(p2[0]) - Fraction(p1[0])) * (Fraction(q[1]) - Fraction(p2[1])) - \
        (Fraction(p2[1]) - Fraction(p1[1])) * (Fraction(q[0]) - Fraction(p2[0]))
    return (d > 0) - (d < 0)


def _env_has(a: Pt, b: Pt, q: Pt) -> bool:
    return min(a[0], b[0]) <= q[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= q[1] <= max(a[1], b[1])


def interior_intersection(p1: Pt, p2: Pt, q1: Pt, q2: Pt) -> bool:
    """LineIntersector.computeIntersection + isInteriorIntersection."""
    if (min(q1[0], q2[0]) > max(p1[0], p2[0]) or max(q1[0], q2[0]) < min(p1[0], p2[0]) or
            min(q1[1], q2[1]) > max(p1[1], p2[1]) or max(q1[1], q2[1]) < min(p1[1], p2[1])):
        return False
    Pq1, Pq2 = orientation(p1, p2, q1), orientation(p1, p2, q2)
    if (Pq1 > 0 and Pq2 > 0) or (Pq1 < 0 and Pq2 < 0):
        return False
    Qp1, Qp2 = orientation(q1, q2, p1), orientation(q1, q2, p2)
    if (Qp1 > 0 and Qp2 > 0) or (Qp1 < 0 and Qp2 < 0):
        return False
    if Pq1 == 0 and Pq2 == 0 and Qp1 == 0 and Qp2 == 0:
        q1inP, q2inP = _env_has(p1, p2, q1), _env_has(p1, p2, q2)
        p1inQ, p2inQ = _env_has(q1, q2, p1), _env_has(q1, q2, p2)
        if q1inP and q2inP:
            pts = [q1, q2]
        elif p1inQ and p2inQ:
            pts = [p1, p2]
        elif q1inP and p1inQ:
            pts = [q1] if (q1 == p1 and not q2inP and not p2inQ) else [q1, p1]
        elif q1inP and p2inQ:
            pts = [q1] if (q1 == p2 and not q2inP and not p1inQ) else [q1, p2]
        elif q2inP and p1inQ:
            pts = [q2] if (q2 == p1 and not q1inP and not p2inQ) else [q2, p1]
        elif q2inP and p2inQ:
            pts = [q2] if (q2 == p2 and not q1inP and not p1inQ) else [q2, p2]
        else:
            return False
    elif 0 in (Pq1, Pq2, Qp1, Qp2):
        if p1 == q1 or p1 == q2:
            pts = [p1]
        elif p2 == q1 or p2 == q2:
            pts = [p2]
        elif Pq1 == 0:
            pts = [q1]
        elif Pq2 == 0:
            pts = [q2]
        elif Qp1 == 0:
            pts = [p1]
        else:
            pts = [p2]
    else:
        return True
    return any(ip not in (p1, p2) or ip not in (q1, q2) for ip in pts)


def point_segment_distance(p: Pt, a: Pt, b: Pt) -> float:
    if a == b:
        return math.hypot(p[0] - a[0], p[1] - a[1])
    len2 = (b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1])
    r = ((p[0] - a[0]) * (b[0] - a[0]) + (p[1] - a[1]) * (b[1] - a[1])) / len2
    if r <= 0.0:
        return math.hypot(p[0] - a[0], p[1] - a[1])
    if r >= 1.0:
        return math.hypot(p[0] - b[0], p[1] - b[1])
    s = ((a[1] - p[1]) * (b[0] - a[0]) - (a[0] - p[0]) * (b[1] - a[1])) / len2
    return abs(s) * math.sqrt(len2)


def simplify_ring(coords: Sequence[Pt], tolerance: float) -> List[Pt]:
    """TopologyPreservingSimplifier on one closed shell (minimum size 4)."""
    p = [(float(x), float(y)) for x, y in coords]
    n = len(p)
    if n < 2:
        return list(p)
    MIN = 4
    live = [True] * n
    flat: List[Tuple[int, int]] = []
    result: List[List[int]] = []

    def bad(si, sj, a, b, lo2=-1, hi2=-1, skip=()):
        for (i, j) in flat:
            if (i, j) in skip:
                continue
            if interior_intersection(p[i], p[j], a, b):
                return True
        for k in range(n - 1):
            if not live[k] or si <= k < sj or lo2 <= k < hi2:
                continue
            if interior_intersection(p[k], p[k + 1], a, b):
                return True
        return False

    def section(i, j, depth):
        depth += 1
        if i + 1 == j:
            result.append([i, j])
            return
        ok = True
        size = 0 if not result else len(result) + 1
        if size < MIN and depth + 1 < MIN:
            ok = False
        maxd, far = -1.0, i
        for k in range(i + 1, j):
            d = point_segment_distance(p[k], p[i], p[j])
            if d > maxd:
                maxd, far = d, k
        if maxd > tolerance:
            ok = False
        if ok and bad(i, j, p[i], p[j]):
            ok = False
        if ok:
            for k in range(i, j):
                live[k] = False
            flat.append((i, j))
            result.append([i, j])
            return
        section(i, far, depth)
        section(far, j, depth)

    section(0, n - 1, 0)
    if p[0] == p[-1] and len(result) + 1 > MIN:
        first, last = tuple(result[0]), tuple(result[-1])
        a, b, end = p[last[0]], p[first[1]], p[first[0]]
        if point_segment_distance(end, a, b) <= tolerance and \
                not bad(first[0], first[1], a, b, last[0], last[1], skip=(first, last)):
            result[0][0] = last[0]
            result.pop()
    out = [p[s[0]] for s in result]
    out.append(p[result[0][0]] if (p[0] == p[-1] and result[0][0] != 0) else p[result[-1][1]])
    return out


def polygon_within_box(coords: Sequence[Pt], box: Tuple[float, float, float, float]) -> bool:
    """shapely ``polygon.within(box)`` for a non-degenerate shell: no point of the polygon lies outside the closed
    box (all vertices inside — the box is convex) and the interiors meet (true as soon as the shell has area or a
    vertex strictly inside; a shell lying entirely on the box boundary is not *within*)."""
    minx, miny, maxx, maxy = box
    if not all(minx <= x <= maxx and miny <= y <= maxy for x, y in coords):
        return False
    return bool(any(minx < x < maxx and miny < y < maxy for x, y in coords) or abs(ring_area(coords)) > 0.0)


def ring_area(coords: Sequence[Pt]) -> float:
    s = 0.0
    for (x0, y0), (x1, y1) in zip(coords[:-1], coords[1:]):
        s += x0 * y1 - x1 * y0
    return 0.5 * s
