"""Outline consumers: vector readers (GeoJSON / GeoPackage / Shapefile), fuse_predictions (helpers.py:703-834),
exclude_outlines (helpers.py:33-69) and the only_forest / only_urban tile flags (preprocessing.py:70-95)."""
import json
import os
import sqlite3
import struct
import sys

import numpy as np
import pytest
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from treedetection_amd import gpkg  # noqa: E402
from treedetection_amd.fusion import exclude_outlines, fuse_predictions  # noqa: E402
from treedetection_amd.geotiff import write_geotiff  # noqa: E402
from treedetection_amd.preprocessing import tile_data  # noqa: E402
from treedetection_amd.vector import read_polygon_layer  # noqa: E402


def sq(x0, y0, x1, y1, cw=False):
    r = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1], [x0, y0]], float)
    return r[::-1].copy() if cw else r


# forest: a block with a clearing (hole), and a second block sharing an edge with the first
FOREST = [[sq(100, 100, 200, 200), sq(140, 140, 160, 160)], [sq(200, 100, 260, 200)]]


def write_outline_geojson(path, epsg=25832):
    feats = [{"type": "Feature", "properties": {}, "geometry": {"type": "Polygon", "coordinates": [r.tolist() for r in FOREST[0]]}},
             {"type": "Feature", "properties": {}, "geometry": {"type": "MultiPolygon", "coordinates": [[r.tolist() for r in FOREST[1]]]}}]
    json.dump({"type": "FeatureCollection", "crs": {"type": "name", "properties": {"name": f"urn:ogc:def:crs:EPSG::{epsg}"}},
               "features": feats}, open(path, "w"))


def _wkb_polygon(rings, big_endian=False):
    o = ">" if big_endian else "<"
    out = struct.pack(o[0].replace(">", "B").replace("<", "B"), 0 if big_endian else 1) + struct.pack(o + "II", 3, len(rings))
    for r in rings:
        out += struct.pack(o + "I", len(r)) + np.asarray(r, o + "f8").tobytes()
    return out


def write_outline_gpkg(path):
    gpkg.write_polygons(path, [], {}, 25832, layer="forest")          # metadata tables + empty feature table
    con = sqlite3.connect(path)
    for i, poly in enumerate(FOREST):
        head = struct.pack("<2sBBi", b"GP", 0, 0b00000001, 25832)          # no envelope
        body = _wkb_polygon(poly, big_endian=(i == 1))                     # mixed WKB byte orders
        if i == 1:                                                         # wrap the second one in a MultiPolygon
            body = struct.pack(">BII", 0, 6, 1) + body
        con.execute('INSERT INTO "forest" (geom) VALUES (?)', (head + body,))
    con.commit()
    con.close()


def write_outline_shp(path):
    recs = []
    for poly in FOREST:
        rings = [poly[0][::-1]] + [h for h in poly[1:]]                    # shell clockwise, holes counter-clockwise
        pts = np.concatenate(rings)
        parts, k = [], 0
        for r in rings:
            parts.append(k)
            k += len(r)
        body = struct.pack("<i4d2i", 5, pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max(), len(rings), len(pts))
        body += struct.pack(f"<{len(parts)}i", *parts) + pts.astype("<f8").tobytes()
        recs.append(body)
    data = b"".join(struct.pack(">ii", i + 1, len(b) // 2) + b for i, b in enumerate(recs))
    header = struct.pack(">i5ii", 9994, 0, 0, 0, 0, 0, (100 + len(data)) // 2) + struct.pack("<ii4d4d", 1000, 5, 100, 100, 260, 200, 0, 0, 0, 0)
    open(path, "wb").write(header + data)
    open(os.path.splitext(path)[0] + ".prj", "w").write('PROJCS["ETRS89 / UTM zone 32N",GEOGCS["ETRS89",AUTHORITY["EPSG","4258"]],AUTHORITY["EPSG","25832"]]')


@pytest.mark.parametrize("writer,ext", [(write_outline_geojson, ".geojson"), (write_outline_gpkg, ".gpkg"), (write_outline_shp, ".shp")])
def test_outline_readers_agree(tmp_path, writer, ext):
    path = str(tmp_path / f"forest{ext}")
    writer(path)
    polys, epsg = read_polygon_layer(path)
    assert epsg == 25832 and len(polys) == 2 and [len(p) for p in polys] == [2, 1]
    for got, want in zip(polys, FOREST):
        for g, w in zip(got, want):
            assert abs(abs(np.dot(g[:-1, 0], g[1:, 1]) - np.dot(g[1:, 0], g[:-1, 1])) -
                       abs(np.dot(w[:-1, 0], w[1:, 1]) - np.dot(w[1:, 0], w[:-1, 1]))) < 1e-9
            assert {tuple(p) for p in g} == {tuple(p) for p in w} and (g[0] == g[-1]).all()
    with pytest.raises(ValueError, match="unsupported vector format"):
        read_polygon_layer(str(tmp_path / "x.kml"))


class Log:
    def __init__(self):
        self.msgs = []

    def __getattr__(self, name):
        return lambda m: self.msgs.append((name, m))


def _crown(x, y, s=4.0):
    return sq(x, y, x + s, y + s)


def test_fuse_predictions(tmp_path):
    urban, forest, out = tmp_path / "urban_geojson", tmp_path / "forrest_geojson", tmp_path / "geojson_predictions"
    os.makedirs(urban)
    os.makedirs(forest)
    outline = str(tmp_path / "forest.geojson")
    write_outline_geojson(outline)
    u = {"deep in the forest": (_crown(110, 110), False), "across the forest edge": (_crown(98, 150), True),
         "in the clearing": (_crown(148, 148), True), "over the clearing's rim": (_crown(138, 150), True),
         "across the shared border": (_crown(198, 150), False), "in town": (_crown(20, 20), True)}
    f = {"deep in the forest": (_crown(120, 120), True), "touching the edge from outside": (_crown(96, 120), True),
         "in the clearing": (_crown(150, 150, 2), False), "in town": (_crown(30, 30), False),
         "across the edge": (_crown(258, 110), True)}
    gpkg.write_polygons(str(urban / "img1.gpkg"), [v[0] for v in u.values()],
                        {"Confidence_score": [0.5 + 0.01 * i for i in range(len(u))], "filter_index_right": [0] * len(u)}, 25832)
    gpkg.write_polygons(str(forest / "img1.gpkg"), [v[0] for v in f.values()],
                        {"Confidence_score": [0.9 - 0.01 * i for i in range(len(f))], "filter_index_right": [0] * len(f)}, 25832)
    gpkg.write_polygons(str(urban / "img2.gpkg"), [], {}, None)                    # empty urban → forest passes through
    gpkg.write_polygons(str(forest / "img2.gpkg"), [_crown(30, 30)], {"Confidence_score": [0.7], "filter_index_right": [0]}, 25832)
    gpkg.write_polygons(str(urban / "img3.gpkg"), [_crown(30, 30)], {"Confidence_score": [0.6], "filter_index_right": [0]}, 25832)
    gpkg.write_polygons(str(urban / "img4.gpkg"), [_crown(110, 110)], {"Confidence_score": [0.6], "filter_index_right": [0]}, 25833)
    gpkg.write_polygons(str(forest / "img4.gpkg"), [_crown(110, 110)], {"Confidence_score": [0.6], "filter_index_right": [0]}, 25833)
    gpkg.write_polygons(str(urban / "img5.gpkg"), [_crown(110, 110)], {"Confidence_score": [0.6], "filter_index_right": [0]}, 31467)
    gpkg.write_polygons(str(forest / "img5.gpkg"), [_crown(110, 110)], {"Confidence_score": [0.6], "filter_index_right": [0]}, 31467)
    log = Log()
    fuse_predictions(str(urban), str(forest), outline, str(out), logger=log)
    rings, cols, srs = gpkg.read_polygons(str(out / "img1.gpkg"))
    want = [v[0] for v in f.values() if v[1]] + [v[0] for v in u.values() if v[1]]
    assert srs == 25832 and len(rings) == len(want) and all((a == b).all() for a, b in zip(rings, want))
    want_scores = [0.9 - 0.01 * i for i, v in enumerate(f.values()) if v[1]] + [0.5 + 0.01 * i for i, v in enumerate(u.values()) if v[1]]
    assert cols["Confidence_score"] == pytest.approx(want_scores) and set(cols) == {"Confidence_score", "filter_index_right"}
    assert len(gpkg.read_polygons(str(out / "img2.gpkg"))[0]) == 1               # forest-only image copied
    assert not os.path.exists(out / "img3.gpkg")                                   # no forest layer → skipped, logged
    # img4: crowns in EPSG:25833, outline in 25832 → the outline is reprojected (treedetection_amd.crs; reference helpers.py:785-790
    # to_crs) — in zone 33 it lies elsewhere, so the urban crown survives and the forest one does not
    r4, _, srs4 = gpkg.read_polygons(str(out / "img4.gpkg"))
    assert srs4 == 25833 and len(r4) == 1
    assert any(lvl == "warning" and "CRS mismatch" in m for lvl, m in log.msgs)
    assert not os.path.exists(out / "img5.gpkg")                                   # DHDN / Gauss-Krüger: not reprojected here → error, no output
    errs = [m for lvl, m in log.msgs if lvl == "error"]
    assert any("img3.gpkg" in m and "not found" in m for m in errs) and any("EPSG:31467" in m for m in errs)
    assert yaml.safe_load(open(out / "fusion_recovery.yaml")) == {"completed_files": ["img1", "img2", "img4"]}
    os.remove(out / "img1.gpkg")
    fuse_predictions(str(urban), str(forest), outline, str(out), logger=log)     # resume: img1 is not rebuilt
    assert not os.path.exists(out / "img1.gpkg")
    with pytest.raises(FileNotFoundError):
        fuse_predictions(str(urban), str(forest), str(tmp_path / "missing.shp"), str(out))


def test_exclude_outlines(tmp_path):
    pred = tmp_path / "geojson_predictions"
    os.makedirs(pred)
    outline = str(tmp_path / "lake.geojson")
    write_outline_geojson(outline)
    crowns = [_crown(110, 110), _crown(98, 150), _crown(20, 20)]
    gpkg.write_polygons(str(pred / "processed_a.gpkg"), crowns, {"Confidence_score": [0.1, 0.2, 0.3]}, 25832)
    gpkg.write_polygons(str(pred / "raw_b.gpkg"), crowns, {"Confidence_score": [0.1, 0.2, 0.3]}, 25832)
    exclude_outlines({"exclude_files": [outline, str(tmp_path / "nope.shp")], "output_directory": str(tmp_path)}, Log())
    rings, cols, _ = gpkg.read_polygons(str(pred / "processed_a.gpkg"))
    assert cols["Confidence_score"] == [0.2, 0.3] and len(rings) == 2             # the crown inside the outline is gone
    assert len(gpkg.read_polygons(str(pred / "raw_b.gpkg"))[0]) == 3              # only processed_* files are touched


def test_tile_flags_with_outline(tmp_path):
    img = np.zeros((3, 300, 300), np.uint8)
    tif = str(tmp_path / "t.tif")
    write_geotiff(tif, img, (1.0, 0, 0.0, 0, -1.0, 300.0), 25832)
    outline = str(tmp_path / "forest.geojson")
    write_outline_geojson(outline)
    tile_data([tif], str(tmp_path / "tiles"), buffer=5, tile_width=50, tile_height=50, forest_shapefile=outline)
    meta = json.load(open(tmp_path / "tiles" / "t.json"))
    flag = lambda x, y: (meta[f"t_{x}_{y}_50_5_25832"]["only_forest"], meta[f"t_{x}_{y}_50_5_25832"]["only_urban"])  # noqa: E731
    assert flag(0, 0) == (False, True)            # no outline polygon near
    assert flag(50, 50) == (False, True)          # candidates come from the UN-buffered tile [50,100]: none (reference rule)
    assert flag(100, 100) == (False, False)       # box [95,155] crosses the forest edge and the clearing
    assert flag(200, 100) == (False, False)       # box [195,255] x [95,155] pokes out below y = 100
    assert flag(150, 150) == (False, False)       # box [145,205]: inside A ∪ B but over the clearing's corner
    assert flag(250, 250) == (False, True)
    tile_data([tif], str(tmp_path / "tiles2"), buffer=0, tile_width=50, tile_height=50, forest_shapefile=outline)
    meta = json.load(open(tmp_path / "tiles2" / "t.json"))
    f0 = lambda x, y: (meta[f"t_{x}_{y}_50_0_25832"]["only_forest"], meta[f"t_{x}_{y}_50_0_25832"]["only_urban"])  # noqa: E731
    assert f0(200, 100) == (True, False)          # [200,250] x [100,150] lies inside block B
    assert f0(150, 100) == (False, False)         # [150,200] x [100,150] touches the clearing's corner region
    assert f0(100, 100) == (False, False)         # contains part of the clearing
    assert f0(50, 100) == (False, True)           # only touches the forest along x = 100 → envelopes do not overlap
