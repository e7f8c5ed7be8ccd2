"""Validity and repair of crown rings — the reference's ``geom.buffer(0) if not geom.is_valid else geom`` on every fused crown
(TreeDetection/helpers.py:815-821; shapely → GEOS, absent here).

``ring_is_valid``: OGC validity of one shell as GEOS' IsValidOp decides it (td_ring_is_valid, exact-sign predicates).

``buffer0``: what GEOS' BufferOp returns for a polygon shell at distance 0, restated from its published algorithm
(geos/operation/buffer: OffsetCurveSetBuilder → noding → PlanarGraph depth labelling → PolygonBuilder): the offset curve at
distance 0 is the ring itself, labelled with the interior on the side its orientation says (Orientation::isCCW — the turn at
the ring's highest vertex); the curve is noded; every face of the arrangement gets a depth = the number of times the curve
winds around it in the labelled sense; the result is the closure of the faces of depth >= 1, as polygons with holes (a
MultiPolygon when they only touch at points). Consequences: a spike (the ring running out and back over itself) vanishes; a
ring that touches itself at a vertex becomes two polygons (or a polygon with a hole); of a ring that CROSSES itself only the
lobes wound like the ring's orientation at its highest vertex survive — GEOS' well-known "buffer(0) drops half of a bow-tie".
Parity unpinned: there is no GEOS in this image; the cases in tests/test_validity.py are the documented behaviours above
(shapely manual, "object.buffer(0)" on the touching bow-tie → two triangles) and hand-derived arrangements.

Rare path (crowns are border-followed blobs, simplified with topology preserved; invalid ones come from one-pixel necks), so
plain Python on exact rational arithmetic where a floating-point sign would not be safe.
"""
from __future__ import annotations

import math
import struct
from fractions import Fraction
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib

Polygon = List[np.ndarray]          # [shell, hole, ...] closed rings


def ring_is_valid(ring: np.ndarray) -> bool:
    r = np.require(ring, dtype=np.float64, requirements=["C", "A"]).reshape(-1, 2)      # (rings cut out of a geometry blob start at odd addresses)
    st = _lib.load().td_ring_is_valid(r.ctypes.data, int(r.shape[0]))
    _lib.check(st, "td_ring_is_valid")
    return bool(st)


def _orient(a, b, c) -> int:
    """Sign of the turn a → b → c (+1 left), exact: floating point behind an error bound, rationals inside it."""
    d1 = (b[0] - a[0]) * (c[1] - a[1])
    d2 = (b[1] - a[1]) * (c[0] - a[0])
    det = d1 - d2
    bound = 4e-16 * (abs(d1) + abs(d2))
    if det > bound:
        return 1
    if det < -bound:
        return -1
    F = Fraction
    e = (F(b[0]) - F(a[0])) * (F(c[1]) - F(a[1])) - (F(b[1]) - F(a[1])) * (F(c[0]) - F(a[0]))
    return (e > 0) - (e < 0)


def _between(a, b, p) -> bool:
    """p (known collinear with a, b) lies on the closed segment."""
    return min(a[0], b[0]) <= p[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= p[1] <= max(a[1], b[1])


def _crossing(a, b, c, d):
    """The point where segments a-b and c-d properly cross, rounded once from the exact rational."""
    F = Fraction
    ax, ay, bx, by, cx, cy, dx, dy = (F(v) for v in (*a, *b, *c, *d))
    den = (bx - ax) * (dy - cy) - (by - ay) * (dx - cx)
    t = ((cx - ax) * (dy - cy) - (cy - ay) * (dx - cx)) / den
    return (float(ax + t * (bx - ax)), float(ay + t * (by - ay)))


def is_ccw(pts: Sequence[Tuple[float, float]]) -> bool:
    """Orientation::isCCW (JTS / GEOS): the turn at the highest vertex (the first one with the greatest y) between its nearest
    distinct neighbours; a flat or folded cap is decided by the neighbours' x order."""
    n = len(pts)
    hi = max(range(n), key=lambda i: (pts[i][1], -i))
    p = pts[hi]
    i = (hi - 1) % n
    while pts[i] == p and i != hi:
        i = (i - 1) % n
    j = (hi + 1) % n
    while pts[j] == p and j != hi:
        j = (j + 1) % n
    prev, nxt = pts[i], pts[j]
    if prev == p or nxt == p or prev == nxt:
        return False
    o = _orient(prev, p, nxt)
    if o == 0:
        return prev[0] > nxt[0]
    return o > 0


def _node(pts: List[Tuple[float, float]]) -> List[Tuple[Tuple[float, float], Tuple[float, float]]]:
    """The ring's segments split wherever another segment meets them → directed edges between nodes, in ring order."""
    n = len(pts)
    segs = [(pts[i], pts[(i + 1) % n]) for i in range(n)]
    splits: List[List[Tuple[float, float]]] = [[] for _ in range(n)]
    xs = np.array(pts)
    lo = np.minimum(xs, np.roll(xs, -1, axis=0))
    hi = np.maximum(xs, np.roll(xs, -1, axis=0))
    for i in range(n):
        a, b = segs[i]
        cand = np.nonzero((lo[:, 0] <= hi[i, 0]) & (hi[:, 0] >= lo[i, 0]) & (lo[:, 1] <= hi[i, 1]) & (hi[:, 1] >= lo[i, 1]))[0]
        for j in cand:
            j = int(j)
            if j <= i:
                continue
            c, d = segs[j]
            o1, o2 = _orient(a, b, c), _orient(a, b, d)
            if o1 * o2 > 0:
                continue
            o3, o4 = _orient(c, d, a), _orient(c, d, b)
            if o3 * o4 > 0:
                continue
            if o1 * o2 < 0 and o3 * o4 < 0:
                x = _crossing(a, b, c, d)
                splits[i].append(x)
                splits[j].append(x)
                continue
            # touching or collinear: end points that lie inside the other segment split it
            for p, o in ((c, o1), (d, o2)):
                if o == 0 and _between(a, b, p) and p != a and p != b:
                    splits[i].append(p)
            for p, o in ((a, o3), (b, o4)):
                if o == 0 and _between(c, d, p) and p != c and p != d:
                    splits[j].append(p)
    edges = []
    for i, (a, b) in enumerate(segs):
        if not splits[i]:
            edges.append((a, b))
            continue
        dx, dy = b[0] - a[0], b[1] - a[1]
        along = sorted(set(splits[i]), key=lambda p: (p[0] - a[0]) * dx + (p[1] - a[1]) * dy)
        chain = [a] + along + [b]
        edges.extend((u, v) for u, v in zip(chain[:-1], chain[1:]) if u != v)
    return edges


def _area2(cycle: Sequence[Tuple[float, float]]) -> float:
    """Twice the signed area, summed relative to the first vertex (map coordinates are ~1e6: products of raw values lose the area)."""
    ox, oy = cycle[0]
    s = 0.0
    for (x0, y0), (x1, y1) in zip(cycle, list(cycle[1:]) + [cycle[0]]):
        s += (x0 - ox) * (y1 - oy) - (x1 - ox) * (y0 - oy)
    return s


def _inside(p, cycle) -> bool:
    """Crossing parity of p against a closed cycle (p is never on it where this is called)."""
    c = False
    n = len(cycle)
    for k in range(n):
        a, b = cycle[k], cycle[(k + 1) % n]
        if (a[1] <= p[1]) != (b[1] <= p[1]):
            o = _orient(a, b, p)
            if (a[1] <= p[1] and o > 0) or (a[1] > p[1] and o < 0):
                c = not c
    return c


def _split_simple(cycle: List[Tuple[float, float]]) -> List[List[Tuple[float, float]]]:
    """A closed walk that passes some vertex more than once → the simple cycles it is made of."""
    out, stack, pos = [], [], {}
    for p in cycle:
        if p in pos:
            k = pos[p]
            loop = stack[k:]
            for q in loop:
                del pos[q]
            del stack[k:]
            if len(loop) >= 3:
                out.append(loop)
        pos[p] = len(stack)
        stack.append(p)
    if len(stack) >= 3:
        out.append(stack)
    return out


def buffer0(ring: np.ndarray) -> List[Polygon]:
    """GEOS' ``polygon.buffer(0)`` for a polygon of one shell → polygons [shell, hole, ...] (shells clockwise, holes
    counter-clockwise, as BufferOp emits them; [] when nothing of positive area is left)."""
    r = np.asarray(ring, dtype=np.float64).reshape(-1, 2)
    pts: List[Tuple[float, float]] = []
    for x, y in r:
        p = (float(x), float(y))
        if not (math.isfinite(p[0]) and math.isfinite(p[1])):
            return []
        if not pts or pts[-1] != p:
            pts.append(p)
    while len(pts) > 1 and pts[-1] == pts[0]:
        pts.pop()
    if len(pts) < 3:
        return []
    sign = 1 if is_ccw(pts) else -1                      # depth of a face = sign x its winding number
    edges = _node(pts)
    # net multiplicity per undirected edge: a stretch run once each way (a spike) cancels
    net: Dict[Tuple, int] = {}
    for u, v in edges:
        if (v, u) in net:
            net[(v, u)] -= 1
        else:
            net[(u, v)] = net.get((u, v), 0) + 1
    half: Dict[Tuple, int] = {}                          # half edge (u, v) → winding gained crossing it from its right to its left
    for (u, v), m in net.items():
        if m != 0:
            half[(u, v)] = m
            half[(v, u)] = -m
    if not half:
        return []
    out_edges: Dict[Tuple, List[Tuple]] = {}
    for (u, v) in half:
        out_edges.setdefault(u, []).append(v)

    def angle_sorted(u):
        # counter-clockwise order of the edges leaving u; exact comparison by half-plane + orientation
        def key(v):
            dx, dy = v[0] - u[0], v[1] - u[1]
            return (0 if (dy > 0 or (dy == 0 and dx > 0)) else 1)
        vs = out_edges[u]
        upper = [v for v in vs if key(v) == 0]
        lower = [v for v in vs if key(v) == 1]
        import functools
        cmp = functools.cmp_to_key(lambda a, b: -_orient(u, a, b))
        return sorted(upper, key=cmp) + sorted(lower, key=cmp)
    order = {u: angle_sorted(u) for u in out_edges}
    # face on the LEFT of half edge (u, v): at v continue with the edge that follows (v → u) clockwise
    nxt = {}
    for (u, v) in half:
        ring_v = order[v]
        k = ring_v.index(u)
        nxt[(u, v)] = (v, ring_v[(k - 1) % len(ring_v)])
    face_of, faces = {}, []
    for h in half:
        if h in face_of:
            continue
        cyc, e = [], h
        while e not in face_of:
            face_of[e] = len(faces)
            cyc.append(e)
            e = nxt[e]
        faces.append(cyc)
    area = [_area2([e[0] for e in cyc]) for cyc in faces]
    # connected components of the arrangement; in each, the cycles of negative area bound the component from outside
    wind = [None] * len(faces)
    comp_of, comps = {}, []
    for f0 in range(len(faces)):
        if f0 in comp_of:
            continue
        comp, todo = [], [f0]
        comp_of[f0] = len(comps)
        while todo:
            f = todo.pop()
            comp.append(f)
            for (u, v) in faces[f]:
                g = face_of[(v, u)]
                if g not in comp_of:
                    comp_of[g] = len(comps)
                    todo.append(g)
        comps.append(comp)
    # outermost first: a component nested in a face of another takes that face's winding as its outside value
    def outer(comp):
        return min(comp, key=lambda f: area[f])
    comps.sort(key=lambda comp: area[outer(comp)])        # most negative outer cycle = largest component first
    done_faces: List[int] = []
    for comp in comps:
        o = outer(comp)
        base = 0
        probe = faces[o][0][0]
        best = None
        for f in done_faces:                               # the smallest bounded face of an earlier component that holds this one
            if area[f] > 0 and _inside(probe, [e[0] for e in faces[f]]) and (best is None or area[f] < area[best]):
                best = f
        if best is not None:
            base = wind[best]
        wind[o] = base
        todo = [o]
        while todo:
            f = todo.pop()
            for (u, v) in faces[f]:
                g = face_of[(v, u)]                        # g lies on the right of (u, v): wind[f] = wind[g] + half[(u, v)]
                if wind[g] is None:
                    wind[g] = wind[f] - half[(u, v)]
                    todo.append(g)
        done_faces.extend(comp)
    # (the outer cycle of a component nested in a face carries that face's winding: its edges bound the face from inside)
    keep = [sign * wind[f] >= 1 for f in range(len(faces))]
    # boundary of the kept area: half edges with a kept face on the left and none on the right
    bnd = [(u, v) for f in range(len(faces)) if keep[f] for (u, v) in faces[f] if not keep[face_of[(v, u)]]]
    if not bnd:
        return []
    b_out: Dict[Tuple, List[Tuple]] = {}
    for u, v in bnd:
        b_out.setdefault(u, []).append(v)
    used, cycles = set(), []
    for start in bnd:
        if start in used:
            continue
        walk, e = [], start
        while e not in used:
            used.add(e)
            walk.append(e[0])
            u, v = e
            cands = b_out[v]
            if len(cands) == 1:
                e = (v, cands[0])
            else:                                          # stay on this face: the first boundary edge clockwise from (v → u)
                ring_v = order[v]
                k = ring_v.index(u)
                for step in range(1, len(ring_v) + 1):
                    w = ring_v[(k - step) % len(ring_v)]
                    if w in cands and (v, w) not in used:
                        e = (v, w)
                        break
                else:
                    break
        cycles.extend(_split_simple(walk))
    shells = [c for c in cycles if _area2(c) > 0]
    holes = [c for c in cycles if _area2(c) < 0]
    polys: List[Polygon] = []
    shells.sort(key=_area2)
    owner: List[List] = [[] for _ in shells]
    def hole_in(h, s):
        on = set(s)
        for q in h:                                        # (after noding, a hole meets its shell at common vertices only)
            if q not in on:
                return _inside(q, s)
        return _inside(((h[0][0] + h[1][0]) / 2, (h[0][1] + h[1][1]) / 2), s)
    for h in holes:
        for k, s in enumerate(shells):                     # the smallest shell that contains the hole
            if hole_in(h, s):
                owner[k].append(h)
                break

    def closed(c, clockwise):
        c = list(c)
        if (_area2(c) < 0) != clockwise:
            c.reverse()
        k = min(range(len(c)), key=lambda i: c[i])
        c = c[k:] + c[:k]
        return np.array(c + [c[0]], dtype=np.float64)
    for k, s in enumerate(shells):
        polys.append([closed(s, True)] + [closed(h, False) for h in owner[k]])
    polys.sort(key=lambda p: tuple(p[0][0]))
    return polys


def geometry_blob(polys: Sequence[Polygon], srs_id: int) -> bytes:
    """GeoPackage geometry blob of a Polygon (one) or MultiPolygon (several) — what geopandas writes for a repaired crown."""
    if not polys:                                          # POLYGON EMPTY: flags = little endian + empty, no envelope
        return struct.pack("<2sBBi", b"GP", 0, 0b00010001, int(srs_id)) + struct.pack("<BII", 1, 3, 0)
    allpts = np.concatenate([r for p in polys for r in p])
    head = struct.pack("<2sBBi4d", b"GP", 0, 0b00000011, int(srs_id), float(allpts[:, 0].min()), float(allpts[:, 0].max()),
                       float(allpts[:, 1].min()), float(allpts[:, 1].max()))

    def wkb_polygon(p):
        out = struct.pack("<BII", 1, 3, len(p))
        for r in p:
            r = np.ascontiguousarray(r, dtype="<f8")
            out += struct.pack("<I", r.shape[0]) + r.tobytes()
        return out
    if len(polys) == 1:
        return head + wkb_polygon(polys[0])
    return head + struct.pack("<BII", 1, 6, len(polys)) + b"".join(wkb_polygon(p) for p in polys)
