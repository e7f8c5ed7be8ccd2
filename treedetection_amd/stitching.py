"""Consumer of the per-tile prediction files: ``Prediction_*.json`` → one vector file per image.

The reference's ``process_and_stitch_predictions`` (TreeDetection/helpers.py:556-600) builds GeoPackages through
geopandas/shapely (edge crop ``box_filter`` 305-319, simplify, CRS handling). Those libraries are not available here
and the step is outside this round's hot-path scope, so this module provides the same entry
point with the part that needs no geometry engine: it drops polygons whose centroid lies in the buffer band of their
tile (the de-duplication role of ``box_filter`` with ``shift``), keeps score and class, and writes one GeoJSON
FeatureCollection per image (CRS from the tile metadata) into ``output_path``.
"""
from __future__ import annotations

import json
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Optional


def _stitch_folder(tiles_path: str, folder: str, output_path: str, shift: float, logger=None) -> Optional[str]:
    stem = os.path.basename(folder)
    meta_path = os.path.join(tiles_path, stem + ".json")
    if not os.path.exists(meta_path):
        if logger:
            logger.debug(f"Missing JSON metadata for {folder}. Skipping.")
        return None
    with open(meta_path) as f:
        meta = json.load(f)
    feats, crs = [], None
    for name in sorted(os.listdir(folder)):
        if not (name.startswith("Prediction_") and name.endswith(".json")):
            continue
        tile_id = name[len("Prediction_"):-len(".json")]
        td = meta.get(tile_id)
        if td is None:
            continue
        crs = td.get("crs", crs)
        parts = tile_id.rsplit("_", 5)   # <stem>_<minx>_<miny>_<tile_width>_<buffer>_<epsg>
        buffer = float(parts[-2]) if len(parts) == 6 else 0.0
        minx, miny, maxx, maxy = td["bounds"][:4]
        inner = (minx + buffer - shift, miny + buffer - shift, maxx - buffer + shift, maxy - buffer + shift)
        with open(os.path.join(folder, name)) as f:
            for ev in json.load(f):
                ring = ev["polygon_coords"][0]
                cx = sum(p[0] for p in ring[:-1]) / max(len(ring) - 1, 1)
                cy = sum(p[1] for p in ring[:-1]) / max(len(ring) - 1, 1)
                if not (inner[0] <= cx <= inner[2] and inner[1] <= cy <= inner[3]):
                    continue
                feats.append({"type": "Feature",
                              "properties": {"Confidence_score": ev["score"], "category_id": ev["category_id"], "tile": tile_id},
                              "geometry": {"type": "Polygon", "coordinates": [ring]}})
    fc = {"type": "FeatureCollection", "features": feats}
    if crs:
        fc["crs"] = {"type": "name", "properties": {"name": f"urn:ogc:def:crs:EPSG::{crs}"}}
    out = os.path.join(output_path, stem + ".geojson")
    with open(out, "w") as f:
        json.dump(fc, f)
    return out


def process_and_stitch_predictions(tiles_path, pred_fold, output_path, max_workers=4, shift=1, simplify_tolerance=0.2,
                                   logger=None, verbose=False):
    """Same signature as the reference (helpers.py:556); see the module docstring for what is and is not done."""
    os.makedirs(output_path, exist_ok=True)
    folders = [os.path.join(pred_fold, d) for d in sorted(os.listdir(pred_fold)) if os.path.isdir(os.path.join(pred_fold, d))]
    with ThreadPoolExecutor(max_workers=max_workers or 4) as ex:
        return [r for r in ex.map(lambda fo: _stitch_folder(tiles_path, fo, output_path, float(shift), logger), folders) if r]
