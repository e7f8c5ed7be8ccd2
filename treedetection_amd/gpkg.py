"""Minimal OGC GeoPackage (1.2) writer / reader for polygon feature layers, on the standard library's sqlite3.

The reference writes its per-image crown layers with ``GeoDataFrame.to_file(path, driver="GPKG")``
(TreeDetection/helpers.py:548-549) through geopandas → GDAL; neither is installed here. A GeoPackage is an SQLite
database with a fixed set of metadata tables and geometries stored as "GPKG binary" blobs (a small header with the
envelope and srs id, then ISO WKB), all specified by the OGC standard, so this module writes exactly that: application
id ``GPKG``, ``gpkg_spatial_ref_sys`` / ``gpkg_contents`` / ``gpkg_geometry_columns`` rows, one feature table named
after the file (geometry column ``geom``, primary key ``fid`` — the names GDAL uses) and the attribute columns the
reference's frame carries. No spatial-index extension is written (optional in the standard; readers scan instead).
"""
from __future__ import annotations

import os
import sqlite3
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_ETRS89 = ('GEOGCS["ETRS89",DATUM["European_Terrestrial_Reference_System_1989",SPHEROID["GRS 1980",6378137,298.257222101,'
           'AUTHORITY["EPSG","7019"]],AUTHORITY["EPSG","6258"]],PRIMEM["Greenwich",0,AUTHORITY["EPSG","8901"]],'
           'UNIT["degree",0.0174532925199433,AUTHORITY["EPSG","9122"]],AUTHORITY["EPSG","4258"]]')
_WGS84 = ('GEOGCS["WGS 84",DATUM["WGS_1984",SPHEROID["WGS 84",6378137,298.257223563,AUTHORITY["EPSG","7030"]],'
               'AUTHORITY["EPSG","6326"]],PRIMEM["Greenwich",0,AUTHORITY["EPSG","8901"]],'
               'UNIT["degree",0.0174532925199433,AUTHORITY["EPSG","9122"]],AUTHORITY["EPSG","4326"]]')


def _utm(name: str, geog: str, zone: int, south: bool, code: int) -> str:
    return (f'PROJCS["{name}",{geog},PROJECTION["Transverse_Mercator"],PARAMETER["latitude_of_origin",0],'
            f'PARAMETER["central_meridian",{zone * 6 - 183}],PARAMETER["scale_factor",0.9996],'
            f'PARAMETER["false_easting",500000],PARAMETER["false_northing",{10000000 if south else 0}],'
            f'UNIT["metre",1,AUTHORITY["EPSG","9001"]],AXIS["Easting",EAST],AXIS["Northing",NORTH],'
            f'AUTHORITY["EPSG","{code}"]]')


def srs_definition(epsg: int) -> Tuple[str, str]:
    """(srs_name, WKT) for the codes that can be written down from the EPSG numbering rule — WGS 84, the ETRS89 and
    WGS 84 UTM zones — else ("EPSG:<code>", "undefined"): readers then resolve the code through
    organization / organization_coordsys_id, which the row still carries."""
    if epsg == 4326:
        return "WGS 84", _WGS84
    if 25828 <= epsg <= 25838:
        z = epsg - 25800
        return f"ETRS89 / UTM zone {z}N", _utm(f"ETRS89 / UTM zone {z}N", _ETRS89, z, False, epsg)
    if 32601 <= epsg <= 32660:
        z = epsg - 32600
        return f"WGS 84 / UTM zone {z}N", _utm(f"WGS 84 / UTM zone {z}N", _WGS84, z, False, epsg)
    if 32701 <= epsg <= 32760:
        z = epsg - 32700
        return f"WGS 84 / UTM zone {z}S", _utm(f"WGS 84 / UTM zone {z}S", _WGS84, z, True, epsg)
    return f"EPSG:{epsg}", "undefined"


def polygon_blob(ring: np.ndarray, srs_id: int) -> bytes:
    """GeoPackage binary geometry of a polygon with one shell: 'GP', version 0, flags = little endian + XY envelope,
    srs id, envelope [minx, maxx, miny, maxy], then WKB (byte order 1, type 3, 1 ring, n points)."""
    r = np.ascontiguousarray(ring, dtype="<f8").reshape(-1, 2)
    head = struct.pack("<2sBBi4d", b"GP", 0, 0b00000011, int(srs_id), float(r[:, 0].min()), float(r[:, 0].max()),
                       float(r[:, 1].min()), float(r[:, 1].max()))
    return head + struct.pack("<BIII", 1, 3, 1, r.shape[0]) + r.tobytes()


def parse_polygon_blob(blob: bytes) -> Tuple[int, np.ndarray]:
    """Inverse of :func:`polygon_blob` (tests, downstream tools): → (srs_id, ring [n,2])."""
    magic, version, flags, srs_id = struct.unpack_from("<2sBBi", blob, 0)
    if magic != b"GP" or version != 0 or not (flags & 1):
        raise ValueError("not a little-endian GeoPackage geometry blob")
    env = {0: 0, 1: 32, 2: 48, 3: 48, 4: 64}[(flags >> 1) & 7]
    off = 8 + env
    order, gtype, nrings, npts = struct.unpack_from("<BIII", blob, off)
    if order != 1 or gtype != 3 or nrings != 1:
        raise ValueError("expected a single-shell WKB polygon")
    return srs_id, np.frombuffer(blob, dtype="<f8", count=2 * npts, offset=off + 13).reshape(npts, 2).copy()


_SCHEMA = """
CREATE TABLE gpkg_spatial_ref_sys (srs_name TEXT NOT NULL, srs_id INTEGER PRIMARY KEY, organization TEXT NOT NULL,
  organization_coordsys_id INTEGER NOT NULL, definition TEXT NOT NULL, description TEXT);
CREATE TABLE gpkg_contents (table_name TEXT NOT NULL PRIMARY KEY, data_type TEXT NOT NULL, identifier TEXT UNIQUE,
  description TEXT DEFAULT '', last_change DATETIME NOT NULL DEFAULT (strftime('%Y-%m-%dT%H:%M:%fZ','now')),
  min_x DOUBLE, min_y DOUBLE, max_x DOUBLE, max_y DOUBLE, srs_id INTEGER,
  CONSTRAINT fk_gc_r_srs_id FOREIGN KEY (srs_id) REFERENCES gpkg_spatial_ref_sys(srs_id));
CREATE TABLE gpkg_geometry_columns (table_name TEXT NOT NULL, column_name TEXT NOT NULL, geometry_type_name TEXT NOT NULL,
  srs_id INTEGER NOT NULL, z TINYINT NOT NULL, m TINYINT NOT NULL,
  CONSTRAINT pk_geom_cols PRIMARY KEY (table_name, column_name), CONSTRAINT uk_gc_table_name UNIQUE (table_name),
  CONSTRAINT fk_gc_tn FOREIGN KEY (table_name) REFERENCES gpkg_contents(table_name),
  CONSTRAINT fk_gc_srs FOREIGN KEY (srs_id) REFERENCES gpkg_spatial_ref_sys (srs_id));
"""


def write_polygons(path: str, rings: Sequence[np.ndarray], columns: Dict[str, Sequence], epsg: Optional[int],
                   layer: Optional[str] = None) -> None:
    """One polygon layer from closed [n,2] rings; see :func:`write_blobs` for ``columns`` / ``epsg``."""
    srs_id = int(epsg) if epsg else 4326
    if len(rings):
        lo = np.min([np.asarray(r).min(axis=0) for r in rings], axis=0)
        hi = np.max([np.asarray(r).max(axis=0) for r in rings], axis=0)
        extent = (float(lo[0]), float(lo[1]), float(hi[0]), float(hi[1]))
    else:
        extent = None
    write_blobs(path, (polygon_blob(r, srs_id) for r in rings), columns, epsg, extent, layer)


def write_blobs(path: str, blobs: Iterable, columns: Dict[str, Sequence], epsg: Optional[int],
                extent: Optional[Tuple[float, float, float, float]], layer: Optional[str] = None) -> None:
    """One polygon layer from ready GeoPackage geometry blobs (bytes / memoryview, e.g. from td_stitch_tile_json).
    ``columns``: name → per-feature values (float → DOUBLE, int → INTEGER, else TEXT); ``extent`` = (minx, miny, maxx,
    maxy) of the layer or None; ``epsg=None`` writes srs_id 4326 like the reference's empty frame (helpers.py:541-543)."""
    layer = layer or os.path.splitext(os.path.basename(path))[0]
    srs_id = int(epsg) if epsg else 4326
    final = path
    path = final + ".tmp"          # written under a temporary name, fsynced once, then renamed: a layer that EXISTS is complete
    for stale in (path, final):
        if os.path.exists(stale):
            os.remove(stale)
    con = sqlite3.connect(path)
    try:
        # a file written once from scratch: no rollback journal, no fsync per transaction; durability comes from the ONE fsync
        # before the rename below — stitching_recovery.yaml lists a folder only after its layer was written, and after an OS
        # crash the resume logic must not trust a layer whose pages never reached the disk
        con.execute("PRAGMA journal_mode = OFF")
        con.execute("PRAGMA synchronous = OFF")
        con.execute("PRAGMA application_id = 1196444487")      # 'GPKG'
        con.execute("PRAGMA user_version = 10200")
        con.executescript(_SCHEMA)
        rows = [("Undefined cartesian SRS", -1, "NONE", -1, "undefined", "undefined cartesian coordinate reference system"),
                ("Undefined geographic SRS", 0, "NONE", 0, "undefined", "undefined geographic coordinate reference system"),
                ("WGS 84", 4326, "EPSG", 4326, _WGS84, "longitude/latitude coordinates in decimal degrees on the WGS 84 spheroid")]
        if srs_id not in (-1, 0, 4326):
            name, wkt = srs_definition(srs_id)
            rows.append((name, srs_id, "EPSG", srs_id, wkt, None))
        con.executemany("INSERT INTO gpkg_spatial_ref_sys VALUES (?,?,?,?,?,?)", rows)

        def sql_type(values) -> str:
            v = next(iter(values), None)
            if isinstance(v, (bool, int, np.integer)):
                return "INTEGER"
            if isinstance(v, (float, np.floating)):
                return "DOUBLE"
            return "TEXT"

        names = list(columns)
        cols_sql = "".join(f', "{n}" {sql_type(columns[n])}' for n in names)
        con.execute(f'CREATE TABLE "{layer}" (fid INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL, geom POLYGON{cols_sql})')
        ext = tuple(extent) if extent else (None, None, None, None)
        con.execute("INSERT INTO gpkg_contents (table_name, data_type, identifier, description, min_x, min_y, max_x, max_y, srs_id) "
                    "VALUES (?, 'features', ?, '', ?, ?, ?, ?, ?)", (layer, layer, *ext, srs_id))
        con.execute("INSERT INTO gpkg_geometry_columns VALUES (?, 'geom', 'POLYGON', ?, 0, 0)", (layer, srs_id))
        ph = ",".join("?" * (1 + len(names)))
        quoted = ", ".join(f'"{n}"' for n in names)
        # columns as plain Python lists once (numpy scalars converted here, not per row), rows by zip: a stitched layer of a
        # noise-like image holds 500 000 features and the row generator was half of the writer's time
        cols = []
        for n in names:
            vals = columns[n]
            if isinstance(vals, np.ndarray):
                vals = vals.tolist()
            elif len(vals) and isinstance(vals[0], np.generic):
                vals = [_py(v) for v in vals]
            cols.append(vals)
        con.executemany(f'INSERT INTO "{layer}" (geom{", " + quoted if names else ""}) VALUES ({ph})', zip(blobs, *cols))
        con.commit()
    except BaseException:
        con.close()
        if os.path.exists(path):
            os.remove(path)
        raise
    con.close()
    fd = os.open(path, os.O_RDONLY)
    try:
        os.fsync(fd)
    finally:
        os.close(fd)
    os.replace(path, final)


def _py(v):
    return v.item() if isinstance(v, np.generic) else v


def read_polygons(path: str) -> Tuple[List[np.ndarray], Dict[str, list], int]:
    """→ (rings, attribute columns, srs_id) of the single feature layer of a file written by :func:`write_polygons`
    (or any GeoPackage whose layer holds single-shell little-endian polygons)."""
    con = sqlite3.connect(path)
    try:
        app = con.execute("PRAGMA application_id").fetchone()[0]
        if app != 1196444487:
            raise ValueError(f"{path}: application_id {app:#x} is not 'GPKG'")
        layer, srs_id = con.execute("SELECT table_name, srs_id FROM gpkg_contents WHERE data_type='features'").fetchone()
        gcol = con.execute("SELECT column_name FROM gpkg_geometry_columns WHERE table_name=?", (layer,)).fetchone()[0]
        info = con.execute(f'PRAGMA table_info("{layer}")').fetchall()
        names = [r[1] for r in info if r[1] not in (gcol, "fid")]
        cur = con.execute(f'SELECT "{gcol}"{"".join(", " + chr(34) + n + chr(34) for n in names)} FROM "{layer}" ORDER BY fid')
        rings, cols = [], {n: [] for n in names}
        for row in cur:
            rings.append(parse_polygon_blob(row[0])[1])
            for n, v in zip(names, row[1:]):
                cols[n].append(v)
        return rings, cols, int(srs_id)
    finally:
        con.close()


class Layer:
    """A polygon layer held as GeoPackage geometry blobs + attribute columns (what fusion and the exclude filter pass
    around: features are selected, never edited, so the blobs are written back untouched)."""

    def __init__(self, blobs: List[bytes], columns: Dict[str, list], srs_id: int):
        self.blobs, self.columns, self.srs_id = blobs, columns, int(srs_id)

    def __len__(self) -> int:
        return len(self.blobs)

    def blob(self, i: int) -> bytes:
        return self.blobs[i]

    def rings(self) -> List[np.ndarray]:
        """Shell of every feature (crowns are single-shell polygons)."""
        from .vector import _gpkg_geom
        out = []
        for b in self.blobs:
            parts = _gpkg_geom(b)
            out.append(parts[0][0] if parts else np.zeros((0, 2)))
        return out

    def envelopes(self) -> np.ndarray:
        """[n,4] minx, maxx, miny, maxy (from the blob header when it carries one, else from the shell)."""
        env = np.zeros((len(self.blobs), 4))
        for i, b in enumerate(self.blobs):
            if (b[3] >> 1) & 7:
                env[i] = np.frombuffer(b, dtype="<f8" if b[3] & 1 else ">f8", count=4, offset=8)
            else:
                r = self.rings()[i]
                env[i] = (r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max())
        return env


def read_layer(path: str) -> Layer:
    """The single feature layer of a GeoPackage (or the features of a GeoJSON file) as a :class:`Layer`."""
    if path.lower().endswith((".geojson", ".json")):
        import json
        from .vector import _read_geojson
        polys, epsg, owner = _read_geojson(path)
        with open(path) as f:
            feats = json.load(f).get("features", [])
        srs = epsg or 4326
        names = sorted({k for ft in feats for k in (ft.get("properties") or {})})
        blobs, cols = [], {n: [] for n in names}
        for poly, fid in zip(polys, owner):
            blobs.append(polygon_blob(poly[0], srs))
            for n in names:
                cols[n].append((feats[fid].get("properties") or {}).get(n))
        return Layer(blobs, cols, srs)
    con = sqlite3.connect(path)
    try:
        row = con.execute("SELECT table_name, srs_id FROM gpkg_contents WHERE data_type='features' LIMIT 1").fetchone()
        if row is None:
            return Layer([], {}, 4326)
        layer, srs_id = row
        gcol = con.execute("SELECT column_name FROM gpkg_geometry_columns WHERE table_name=?", (layer,)).fetchone()[0]
        info = con.execute(f'PRAGMA table_info("{layer}")').fetchall()
        pk = [r[1] for r in info if r[5]]
        names = [r[1] for r in info if r[1] != gcol and r[1] not in pk]
        sel = "".join(f', "{n}"' for n in names)
        blobs, cols = [], {n: [] for n in names}
        for row in con.execute(f'SELECT "{gcol}"{sel} FROM "{layer}" ORDER BY rowid'):
            if row[0] is None:
                continue
            blobs.append(bytes(row[0]))
            for n, v in zip(names, row[1:]):
                cols[n].append(v)
        return Layer(blobs, cols, int(srs_id))
    finally:
        con.close()
