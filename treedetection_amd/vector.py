"""Polygon layers in and out of the outline steps: readers for the vector formats the reference opens with
``gpd.read_file`` (forest outline ``forrest_outline``: helpers.py:735, preprocessing.py:155; exclude files:
helpers.py:41) — GeoJSON, GeoPackage and ESRI Shapefile — and the ctypes wrapper of ``td_region_relate``.

A layer comes back as a list of polygons, each a list of closed rings ``[n,2] float64`` (shell first, then holes);
MultiPolygons are flattened into their parts (the predicates below work on the union anyway). No reprojection:
there is no PROJ here, so callers compare EPSG codes and refuse mixed layers instead of ``to_crs``.
"""
from __future__ import annotations

import json
import os
import re
import sqlite3
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

Polygon = List[np.ndarray]


# ---- WKB (ISO and EWKB flavours, both byte orders) ------------------------------------------------------
def _wkb_polygons(buf: bytes, off: int = 0) -> Tuple[List[Polygon], int]:
    order = "<" if buf[off] == 1 else ">"
    (gtype,) = struct.unpack_from(order + "I", buf, off + 1)
    off += 5
    has_z = bool(gtype & 0x80000000) or (gtype % 10000) // 1000 in (1, 3)
    has_m = bool(gtype & 0x40000000) or (gtype % 10000) // 1000 in (2, 3)
    if gtype & 0x20000000:          # EWKB SRID
        off += 4
    base = (gtype & 0x0fffffff) % 1000
    dims = 2 + has_z + has_m
    if base == 3:
        (nrings,) = struct.unpack_from(order + "I", buf, off)
        off += 4
        rings = []
        for _ in range(nrings):
            (npts,) = struct.unpack_from(order + "I", buf, off)
            off += 4
            pts = np.frombuffer(buf, dtype=order + "f8", count=npts * dims, offset=off).reshape(npts, dims)
            off += npts * dims * 8
            rings.append(np.ascontiguousarray(pts[:, :2], dtype=np.float64))
        return ([rings] if rings else []), off
    if base in (6, 7):              # MultiPolygon / GeometryCollection
        (n,) = struct.unpack_from(order + "I", buf, off)
        off += 4
        out: List[Polygon] = []
        for _ in range(n):
            part, off = _wkb_polygons(buf, off)
            out.extend(part)
        return out, off
    raise ValueError(f"WKB geometry type {gtype} is not a polygon type")


def _gpkg_geom(blob: bytes) -> List[Polygon]:
    if blob is None or len(blob) < 8 or blob[:2] != b"GP":
        return []
    flags = blob[3]
    if flags & 0x10:                # empty geometry
        return []
    env = {0: 0, 1: 32, 2: 48, 3: 48, 4: 64}[(flags >> 1) & 7]
    return _wkb_polygons(blob, 8 + env)[0]


def _read_gpkg(path: str) -> Tuple[List[Polygon], Optional[int], list]:
    con = sqlite3.connect(path)
    try:
        row = con.execute("SELECT table_name, srs_id FROM gpkg_contents WHERE data_type='features' LIMIT 1").fetchone()
        if row is None:
            return [], None, []
        table, srs_id = row
        col = con.execute("SELECT column_name FROM gpkg_geometry_columns WHERE table_name=?", (table,)).fetchone()[0]
        epsg = None
        r = con.execute("SELECT organization, organization_coordsys_id FROM gpkg_spatial_ref_sys WHERE srs_id=?", (srs_id,)).fetchone()
        if r and str(r[0]).upper() == "EPSG":
            epsg = int(r[1])
        polys, owner = [], []
        for fid, (blob,) in enumerate(con.execute(f'SELECT "{col}" FROM "{table}"')):
            parts = _gpkg_geom(blob)
            polys.extend(parts)
            owner.extend([fid] * len(parts))
        return polys, epsg, owner
    finally:
        con.close()


def _read_geojson(path: str) -> Tuple[List[Polygon], Optional[int], list]:
    with open(path) as f:
        gj = json.load(f)
    epsg = None
    name = (((gj.get("crs") or {}).get("properties") or {}).get("name")) or ""
    m = re.search(r"EPSG:+(\d+)", name.upper().replace("::", ":"))
    if m:
        epsg = int(m.group(1))      # a file without a crs member stays "unknown": it is used in the rasters' CRS as is
    polys, owner = [], []
    feats = gj.get("features", [gj] if gj.get("type") == "Feature" else [])
    for fid, feat in enumerate(feats):
        g = feat.get("geometry") or {}
        parts = []
        if g.get("type") == "Polygon":
            parts = [g["coordinates"]]
        elif g.get("type") == "MultiPolygon":
            parts = g["coordinates"]
        for part in parts:
            rings = [np.asarray(r, dtype=np.float64)[:, :2].copy() for r in part if len(r) >= 4]
            if rings:
                polys.append(rings)
                owner.append(fid)
    return polys, epsg, owner


def _ring_area2(r: np.ndarray) -> float:
    return float(np.dot(r[:-1, 0], r[1:, 1]) - np.dot(r[1:, 0], r[:-1, 1]))


def _point_in_ring(p, ring: np.ndarray) -> bool:
    x, y = ring[:-1, 0], ring[:-1, 1]
    x2, y2 = ring[1:, 0], ring[1:, 1]
    cond = (y <= p[1]) != (y2 <= p[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        xi = x + (p[1] - y) * (x2 - x) / (y2 - y)
    return bool(np.count_nonzero(cond & (xi > p[0])) % 2)


def _read_shp(path: str) -> Tuple[List[Polygon], Optional[int], list]:
    """ESRI Shapefile main file: Polygon / PolygonZ / PolygonM records (shape types 5, 15, 25). Rings wound clockwise
    are shells, counter-clockwise rings are holes of the shell of the same record that contains them."""
    with open(path, "rb") as f:
        buf = f.read()
    if struct.unpack_from(">i", buf, 0)[0] != 9994:
        raise ValueError(f"{path}: not a shapefile")
    polys, owner = [], []
    off, fid = 100, 0
    while off + 8 <= len(buf):
        _, clen = struct.unpack_from(">ii", buf, off)
        rec = off + 8
        off = rec + 2 * clen
        (stype,) = struct.unpack_from("<i", buf, rec)
        if stype in (5, 15, 25):
            nparts, npts = struct.unpack_from("<ii", buf, rec + 36)
            parts = list(struct.unpack_from(f"<{nparts}i", buf, rec + 44)) + [npts]
            pts = np.frombuffer(buf, dtype="<f8", count=2 * npts, offset=rec + 44 + 4 * nparts).reshape(npts, 2)
            shells, holes = [], []
            for a, b in zip(parts[:-1], parts[1:]):
                ring = np.array(pts[a:b], dtype=np.float64)
                if len(ring) < 4:
                    continue
                (shells if _ring_area2(ring) < 0 else holes).append(ring)
            if not shells and holes:            # writers that ignore the winding rule: treat the rings as shells
                shells, holes = holes, []
            rec_polys = [[s] for s in shells]
            for h in holes:
                for poly in rec_polys:
                    if _point_in_ring(h[0], poly[0]):
                        poly.append(h)
                        break
            polys.extend(rec_polys)
            owner.extend([fid] * len(rec_polys))
        fid += 1
    epsg = None
    prj = os.path.splitext(path)[0] + ".prj"
    if os.path.exists(prj):
        with open(prj) as f:
            m = re.findall(r'AUTHORITY\["EPSG","(\d+)"\]', f.read())
        if m:
            epsg = int(m[-1])
    return polys, epsg, owner


def read_polygon_layer(path: str, with_owner: bool = False):
    """→ (polygons, epsg or None) — or (polygons, epsg, feature index per polygon) with ``with_owner``."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".gpkg":
        polys, epsg, owner = _read_gpkg(path)
    elif ext in (".geojson", ".json"):
        polys, epsg, owner = _read_geojson(path)
    elif ext == ".shp":
        polys, epsg, owner = _read_shp(path)
    else:
        raise ValueError(f"{path}: unsupported vector format (GeoJSON, GeoPackage and Shapefile are read)")
    for poly in polys:
        for i, r in enumerate(poly):
            if not (r[0] == r[-1]).all():
                poly[i] = np.concatenate([r, r[:1]])
    return (polys, epsg, owner) if with_owner else (polys, epsg)


# ---- predicates ------------------------------------------------------------------------------------------
class Region:
    """Polygons with holes, packed once for ``td_region_relate``."""

    def __init__(self, polygons: Sequence[Polygon]):
        rings = [np.ascontiguousarray(r, dtype=np.float64).reshape(-1, 2) for poly in polygons for r in poly]
        self.ring_poly = np.asarray([i for i, poly in enumerate(polygons) for _ in poly], dtype=np.int32)
        self.ring_start = np.zeros(len(rings) + 1, dtype=np.int64)
        if rings:
            self.ring_start[1:] = np.cumsum([len(r) for r in rings])
        self.xy = np.concatenate(rings) if rings else np.zeros((0, 2))
        self.n_rings = len(rings)
        if rings:
            self.bounds = (float(self.xy[:, 0].min()), float(self.xy[:, 1].min()), float(self.xy[:, 0].max()), float(self.xy[:, 1].max()))
        else:
            self.bounds = None

    def relate(self, queries: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        """closed query rings → (intersects[n], within[n]) against the union of the polygons."""
        n = len(queries)
        flags = np.zeros(n, dtype=np.uint8)
        if n == 0 or self.n_rings == 0:
            return flags.astype(bool), flags.astype(bool)
        qs = [np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 2) for q in queries]
        q_start = np.zeros(n + 1, dtype=np.int64)
        q_start[1:] = np.cumsum([len(q) for q in qs])
        q_xy = np.concatenate(qs)
        st = _lib.load().td_region_relate(self.xy.ctypes.data, self.ring_start.ctypes.data, self.ring_poly.ctypes.data,
                                          self.n_rings, q_xy.ctypes.data, q_start.ctypes.data, n, flags.ctypes.data)
        _lib.check(st, "td_region_relate")
        return (flags & 1).astype(bool), (flags & 2).astype(bool)


def box_ring(minx, miny, maxx, maxy) -> np.ndarray:
    """shapely.geometry.box's ring (counter-clockwise from the lower right corner)."""
    return np.array([[maxx, miny], [maxx, maxy], [minx, maxy], [minx, miny], [maxx, miny]], dtype=np.float64)
