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
from typing import Optional, Sequence

import numpy as np
import yaml

from .geotiff import GeoTiff


def _load_outline(path: Optional[str], target_epsg: Optional[int] = None):
    """Forest outline → list of polygons (each a list of closed rings, shell first) in the CRS of the rasters — reference
    preprocessing.py:153-163 (``gpd.read_file(forest_shapefile)`` then ``to_crs(crs of the first raster)``; GeoJSON, GeoPackage
    and Shapefile are read here, treedetection_amd.crs reprojects between geographic, UTM and Web Mercator codes)."""
    if not path:
        return None
    from .crs import to_crs
    from .vector import read_polygon_layer
    polys, epsg = read_polygon_layer(path)
    polys = to_crs(polys, epsg, target_epsg)
    polys = [p for p in polys if p and len(p[0]) >= 4]
    if not polys:
        raise ValueError(f"No valid geometries found in the forest shapefile {path}.")
    return polys


def _tile_flags(forest_polys, forest_boxes, minx, miny, tile_width, tile_height, bounds):
    """only_forest / only_urban of one tile (reference preprocessing.py:70-95): candidates = outline polygons whose
    envelope strictly overlaps the un-buffered tile; none of them intersects the buffered box → only_urban; the union
    of those that do contains the box → only_forest."""
    from .vector import Region, box_ring
    fb = forest_boxes
    cand = np.nonzero((fb[:, 2] > minx) & (fb[:, 0] < minx + tile_width) & (fb[:, 3] > miny) & (fb[:, 1] < miny + tile_height))[0]
    if cand.size == 0:
        return False, True
    inter, within = Region([forest_polys[i] for i in cand]).relate([box_ring(*bounds)])
    if not inter[0]:
        return False, True
    return bool(within[0]), False


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
    forest_boxes = None
    if forest_polys is not None:
        forest_boxes = np.array([[p[0][:, 0].min(), p[0][:, 1].min(), p[0][:, 0].max(), p[0][:, 1].max()] for p in forest_polys])
    for minx in np.arange(left, right, tile_width):
        for miny in np.arange(bottom, top, tile_height):
            tile_id = f"{tilename}_{int(minx)}_{int(miny)}_{int(tile_width)}_{int(buffer)}_{crs}"
            bounds = [float(minx - buffer), float(miny - buffer), float(minx + tile_width + buffer),
                      float(miny + tile_height + buffer)]
            only_forest, only_urban = False, False
            if forest_polys is not None:
                only_forest, only_urban = _tile_flags(forest_polys, forest_boxes, minx, miny, tile_width, tile_height, bounds)
            c0, r0, w, h = data.window_of_bounds(bounds)
            if w <= 0 or h <= 0:
                raise ValueError("Input shapes do not overlap raster, check geometry of incoming Tifs.")
            t = data.window_transform(c0, r0)
            meta[tile_id] = {"crs": crs, "transform": [t[0], t[1], t[2], t[3], t[4], t[5], 0.0, 0.0, 1.0],
                             "bounds": bounds, "only_forest": only_forest, "only_urban": only_urban}
    # written under a temporary name and renamed: a reader (another rank, a resumed run) never sees half a file
    final = Path(out_dir) / f"{tilename}.json"
    tmp = Path(out_dir) / f".{tilename}.json.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        f.write(json.dumps(meta))
    os.replace(tmp, final)


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
    target = None
    if forest_shapefile:
        with GeoTiff(file_list[0]) as first:          # the reference aligns the outline with the FIRST raster's CRS (preprocessing.py:157-158)
            target = first.epsg
    polys = _load_outline(forest_shapefile, target)

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
