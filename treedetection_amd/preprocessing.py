"""Tile-metadata producer: the input format of the prediction stage (reference TreeDetection/preprocessing.py:33-224).

One JSON per image: ``{tile_id: {crs, transform, bounds, only_forest, only_urban}}``; no pixels are written — tiles
are cropped lazily at predict time. Tile ids, grid, buffered bounds and window transforms follow the reference
(preprocessing.py:57-65,99-123). Forest flags: without geopandas/shapely this package accepts the outline as GeoJSON
polygons and decides conservatively — ``only_urban`` when no polygon's bbox overlaps the tile (the reference's own
test, 69-96), ``only_forest`` only when ONE polygon contains the whole buffered tile; anything else stays mixed and is
predicted by both models, which is always correct.
"""
from __future__ import annotations

import json
import os
from concurrent.futures import ThreadPoolExecutor, as_completed
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import yaml

from .geotiff import GeoTiff


def _load_outline(path: Optional[str]):
    if not path:
        return None
    if not path.lower().endswith((".json", ".geojson")):
        raise ValueError(f"forest outline {path}: only GeoJSON is supported without geopandas")
    with open(path) as f:
        gj = json.load(f)
    polys = []
    for feat in gj.get("features", []):
        g = feat.get("geometry") or {}
        if g.get("type") == "Polygon":
            polys.append(np.asarray(g["coordinates"][0], dtype=np.float64))
        elif g.get("type") == "MultiPolygon":
            polys.extend(np.asarray(p[0], dtype=np.float64) for p in g["coordinates"])
    if not polys:
        raise ValueError(f"No valid geometries found in the forest shapefile {path}.")
    return polys


def _point_in_ring(px, py, ring) -> bool:
    x, y = ring[:, 0], ring[:, 1]
    x2, y2 = np.roll(x, -1), np.roll(y, -1)
    cond = ((y > py) != (y2 > py)) & (px < (x2 - x) * (py - y) / np.where(y2 == y, 1e-300, (y2 - y)) + x)
    return bool(np.count_nonzero(cond) % 2)


def _ring_crosses_box(ring, minx, miny, maxx, maxy) -> bool:
    x, y = ring[:, 0], ring[:, 1]
    inside = (x > minx) & (x < maxx) & (y > miny) & (y < maxy)
    return bool(inside.any())


def tile_single_file(data_path: str, out_dir: str, buffer: int = 0, tile_width: int = 50, tile_height: int = 50,
                     forest_polys=None, logger=None):
    if not os.path.exists(data_path) or not os.path.isfile(data_path):
        raise FileNotFoundError(f"File not found: {data_path}")
    os.makedirs(out_dir, exist_ok=True)
    data = GeoTiff(data_path)
    crs = data.epsg
    tilename = Path(data_path).stem
    left, bottom, right, top = data.bounds
    meta = {}
    for minx in np.arange(left, right, tile_width):
        for miny in np.arange(bottom, top, tile_height):
            tile_id = f"{tilename}_{int(minx)}_{int(miny)}_{int(tile_width)}_{int(buffer)}_{crs}"
            bounds = [float(minx - buffer), float(miny - buffer), float(minx + tile_width + buffer),
                      float(miny + tile_height + buffer)]
            only_forest, only_urban = False, False
            if forest_polys is not None:
                tb = (minx, miny, minx + tile_width, miny + tile_height)
                over = [p for p in forest_polys
                        if p[:, 0].max() > tb[0] and p[:, 0].min() < tb[2] and p[:, 1].max() > tb[1] and p[:, 1].min() < tb[3]]
                if not over:
                    only_urban = True
                else:
                    corners = [(bounds[0], bounds[1]), (bounds[2], bounds[1]), (bounds[2], bounds[3]), (bounds[0], bounds[3])]
                    for p in over:
                        if all(_point_in_ring(cx, cy, p) for cx, cy in corners) and not _ring_crosses_box(p, *bounds):
                            only_forest = True
                            break
            c0, r0, w, h = data.window_of_bounds(bounds)
            if w <= 0 or h <= 0:
                raise ValueError("Input shapes do not overlap raster, check geometry of incoming Tifs.")
            t = data.window_transform(c0, r0)
            meta[tile_id] = {"crs": crs, "transform": [t[0], t[1], t[2], t[3], t[4], t[5], 0.0, 0.0, 1.0],
                             "bounds": bounds, "only_forest": only_forest, "only_urban": only_urban}
    with open(Path(out_dir) / f"{tilename}.json", "w") as f:
        f.write(json.dumps(meta))


def load_recovery_data(file_list, buffer, tile_width, tile_height, logger, out_dir, recovery_file):
    """tiles/recovery.yaml keyed on (buffer, tile_w, tile_h) — reference preprocessing.py:226-278."""
    done = []
    if os.path.exists(recovery_file):
        with open(recovery_file) as f:
            st = yaml.safe_load(f) or {}
        if (st.get("buffer"), st.get("tile_width"), st.get("tile_height")) == (buffer, tile_width, tile_height):
            done = [p for p in st.get("processed_files", [])
                    if os.path.exists(os.path.join(out_dir, Path(p).stem + ".json"))]
    return [f for f in file_list if f not in done], done


def save_recovery_data(file_list, buffer, tile_width, tile_height, logger, recovered, processed, recovery_file):
    with open(recovery_file, "w") as f:
        yaml.safe_dump({"buffer": buffer, "tile_width": tile_width, "tile_height": tile_height,
                        "processed_files": sorted(set(list(recovered) + list(processed)))}, f, sort_keys=False)


def tile_data(file_list: Sequence[str], out_dir: str, buffer: int = 30, tile_width: int = 200, tile_height: int = 200,
              parallel: bool = False, max_workers: int = 4, forest_shapefile: str = None, logger=None):
    os.makedirs(out_dir, exist_ok=True)
    recovery = os.path.join(out_dir, "recovery.yaml")
    file_list, recovered = load_recovery_data(list(file_list), buffer, tile_width, tile_height, logger, out_dir, recovery)
    if not file_list:
        (logger.info if logger else print)("All files have already been processed. Exiting Tiling.")
        return
    polys = _load_outline(forest_shapefile)

    def one(p):
        try:
            tile_single_file(p, out_dir, buffer, tile_width, tile_height, polys, logger)
        except Exception as e:   # reference: log and continue (preprocessing.py:189-193)
            (logger.error if logger else print)(f"Error processing file: {e}")

    if parallel:
        with ThreadPoolExecutor(max_workers=max_workers or 4) as ex:
            for fut in as_completed([ex.submit(one, p) for p in file_list]):
                fut.result()
    else:
        for p in file_list:
            one(p)
    save_recovery_data(file_list, buffer, tile_width, tile_height, logger, recovered, file_list, recovery)
