"""Urban / forest fusion by the forest outline (reference TreeDetection/helpers.py:703-834 ``fuse_predictions``) and
the exclude-outline filter (helpers.py:33-69 ``exclude_outlines``).

Per stitched image: keep the forest model's crowns that *intersect* the forest, and the urban model's crowns that are
not *within* it; either layer alone passes through unchanged when the other is empty. The reference evaluates the two
predicates against ``unary_union`` of the outline polygons near the image (geopandas/shapely, absent here); this
module evaluates them against the polygons themselves with ``td_region_relate`` (libtreedet_hip.so, host code), which
gives the same answers without building the union. ``to_crs``: treedetection_amd.crs moves the OUTLINE into the crowns' CRS
(geographic / UTM / Web Mercator codes; any other pair is refused with an error instead of guessed).

Invalid geometries. The reference repairs an invalid OUTLINE with ``make_valid`` (helpers.py:740-751): GEOS rebuilds the area from
the noded linework, i.e. a point is inside when a ray from it crosses the polygon's rings an odd number of times — which is how
``td_region_relate`` decides membership in the first place (odd crossing count over ALL rings of a polygon), so a bow-tie or an
overlapping-hole outline answers as its repaired form would. Every fused CROWN that is not valid goes through ``buffer(0)``
(helpers.py:815-817): treedetection_amd.validity restates both (td_ring_is_valid; the depth-labelled arrangement of BufferOp) —
a crown with a one-pixel neck comes out as a MultiPolygon, a spike disappears.
"""
from __future__ import annotations

import os
import shutil
from typing import List

import numpy as np

from .gpkg import read_layer, write_blobs
from .recoveries import load_fusion_recovery, save_fusion_recovery
from .validity import buffer0, geometry_blob, ring_is_valid
from .vector import Region, read_polygon_layer


def _check_dir(path, what):
    if not os.path.exists(path) or not os.path.isdir(path):
        raise FileNotFoundError(f"{what} predictions path not found: {path}")


def fuse_predictions(urban_fold, forrest_fold, forrest_path, output_dir, logger=None):
    _check_dir(urban_fold, "Urban")
    _check_dir(forrest_fold, "Forest")
    if not os.path.exists(forrest_path) or not os.path.isfile(forrest_path):
        raise FileNotFoundError(f"Forest boundary path not found: {forrest_path}")
    if os.path.exists(output_dir) and not os.path.isdir(output_dir):
        os.unlink(output_dir)
    os.makedirs(output_dir, exist_ok=True)

    completed = load_fusion_recovery(output_dir, logger)
    files = sorted(f for f in os.listdir(urban_fold) if f.endswith(".geojson") or f.endswith(".gpkg"))
    todo = [f for f in files if os.path.splitext(f)[0] not in completed]
    if logger and files and not todo:
        logger.debug(f"All files have been completed. Skipping fusion for {len(files)} files.")
    elif logger and len(files) - len(todo) > 0:
        logger.debug(f"Skipping fusion for {len(files) - len(todo)} of {len(files)} files that have already been processed.")
    elif logger and not todo:
        logger.debug("No files to process. in fusion, returning.")

    forest_polys, forest_epsg = read_polygon_layer(forrest_path)
    forest_polys = [p for p in forest_polys if p and len(p[0]) >= 4]
    boxes = np.array([[p[0][:, 0].min(), p[0][:, 1].min(), p[0][:, 0].max(), p[0][:, 1].max()] for p in forest_polys]) \
        if forest_polys else np.zeros((0, 4))

    fused: List[str] = []
    forest_polys_in = {}          # the outline reprojected per crown CRS (normally none: same code, or a file without one)
    for name in todo:
        urban_path, forest_path_ = os.path.join(urban_fold, name), os.path.join(forrest_fold, name)
        if not os.path.exists(urban_path):
            if logger:
                logger.error(f"Urban GeoJSON for tile {name} at path {urban_path} not found. Skipping tile.")
            continue
        if not os.path.exists(forest_path_):
            if logger:
                logger.error(f"Forest GeoJSON for tile {name} at path {forest_path_} not found. Skipping tile.")
            continue
        out_path = os.path.join(output_dir, os.path.basename(name))
        try:
            urban, forest = read_layer(urban_path), read_layer(forest_path_)
            if len(urban) == 0:
                shutil.copyfile(forest_path_, out_path)
                if logger:
                    logger.debug(f"Only forest file saved to {out_path}")
                fused.append(out_path)
                continue
            if len(forest) == 0:
                shutil.copyfile(urban_path, out_path)
                if logger:
                    logger.debug(f"Only urban file saved to {out_path}")
                fused.append(out_path)
                continue
            if urban.srs_id != forest.srs_id:
                raise ValueError(f"CRS mismatch between the urban (EPSG:{urban.srs_id}) and the forest (EPSG:{forest.srs_id}) crowns of one image")
            if forest_epsg and forest_epsg != urban.srs_id and forest_polys_in.get(urban.srs_id) is None:
                # reference helpers.py:785-790 aligns the crowns with the outline's CRS; here the OUTLINE goes to the crowns' CRS
                # (vertex by vertex, as to_crs does) and the fused layer stays in the rasters' CRS
                from .crs import to_crs
                if logger:
                    logger.warning("CRS mismatch detected. Aligning the forest boundary with the crowns' CRS.")
                moved = to_crs(forest_polys, forest_epsg, urban.srs_id)
                forest_polys_in[urban.srs_id] = (moved, np.array([[p[0][:, 0].min(), p[0][:, 1].min(), p[0][:, 0].max(), p[0][:, 1].max()] for p in moved])
                                                 if moved else np.zeros((0, 4)))
            if forest_epsg and forest_epsg != urban.srs_id:
                forest_polys, boxes = forest_polys_in[urban.srs_id]
            ue, fe = urban.envelopes(), forest.envelopes()          # [n,4] minx, maxx, miny, maxy
            cb = (min(ue[:, 0].min(), fe[:, 0].min()), min(ue[:, 2].min(), fe[:, 2].min()),
                  max(ue[:, 1].max(), fe[:, 1].max()), max(ue[:, 3].max(), fe[:, 3].max()))
            # outline polygons whose envelope meets the image's combined bounds are the only ones that can matter
            # (the reference clips with `intersects(combined_bbox)` before the union)
            near = [forest_polys[i] for i in np.nonzero((boxes[:, 2] >= cb[0]) & (boxes[:, 0] <= cb[2]) &
                                                        (boxes[:, 3] >= cb[1]) & (boxes[:, 1] <= cb[3]))[0]] if len(boxes) else []
            if near:
                region = Region(near)
                keep_f = region.relate(forest.rings())[0]
                keep_u = ~region.relate(urban.rings())[1]
            else:
                keep_f = np.zeros(len(forest), bool)
                keep_u = np.ones(len(urban), bool)
            blobs = [forest.blob(i) for i in np.nonzero(keep_f)[0]] + [urban.blob(i) for i in np.nonzero(keep_u)[0]]
            env = np.concatenate([fe[keep_f], ue[keep_u]])
            # reference helpers.py:815-817: `geom.buffer(0) if not geom.is_valid else geom` on every fused crown
            f_rings, u_rings = forest.rings(), urban.rings()
            kept_rings = [f_rings[i] for i in np.nonzero(keep_f)[0]] + [u_rings[i] for i in np.nonzero(keep_u)[0]]
            repaired = 0
            for k, ring in enumerate(kept_rings):
                if len(ring) and not ring_is_valid(ring):
                    polys = buffer0(ring)
                    blobs[k] = geometry_blob(polys, forest.srs_id)
                    if polys:
                        pts = np.concatenate([r for p_ in polys for r in p_])
                        env[k] = (pts[:, 0].min(), pts[:, 0].max(), pts[:, 1].min(), pts[:, 1].max())
                    else:
                        env[k] = np.nan
                    repaired += 1
            if repaired and logger:
                logger.debug(f"{repaired} invalid crown geometries repaired (buffer(0)) in {name}")
            cols = {}
            for c in forest.columns:
                if c in urban.columns:
                    cols[c] = [forest.columns[c][i] for i in np.nonzero(keep_f)[0]] + [urban.columns[c][i] for i in np.nonzero(keep_u)[0]]
            env = env[~np.isnan(env[:, 0])]
            extent = (float(env[:, 0].min()), float(env[:, 2].min()), float(env[:, 1].max()), float(env[:, 3].max())) if len(env) else None
            write_blobs(out_path, blobs, cols, forest.srs_id, extent)
            if logger:
                logger.debug(f"Fused file saved to {out_path}")
            fused.append(out_path)
        except Exception as e:
            if logger:
                logger.error(f"Failed to process tile {name}: {e}")
    save_fusion_recovery(output_dir, list(completed) + fused, logger)


def exclude_outlines(config, logger=None):
    """Drops crowns that lie within the outlines of ``config['exclude_files']`` from every ``processed_*`` layer in
    ``<output_directory>/geojson_predictions`` (reference helpers.py:33-69)."""
    for outline in config.get("exclude_files", []) or []:
        try:
            polys, epsg = read_polygon_layer(outline)
        except Exception as e:
            (logger.error if logger else print)(f"Failed to read exclude file '{outline}': {e}")
            continue
        pred_dir = os.path.join(config["output_directory"], "geojson_predictions")
        for file in sorted(os.listdir(pred_dir)):
            if not (file.endswith(".geojson") or file.endswith(".gpkg")) or not file.startswith("processed_"):
                continue
            path = os.path.join(pred_dir, file)
            try:
                crowns = read_layer(path)
                if len(crowns) == 0:
                    continue
                outline = polys
                if epsg and crowns.srs_id != epsg:          # reference helpers.py:55: exclude_outline.to_crs(crowns.crs)
                    from .crs import to_crs
                    outline = to_crs(polys, epsg, crowns.srs_id)
                keep = ~Region(outline).relate(crowns.rings())[1]
                env = crowns.envelopes()[keep]
                extent = (float(env[:, 0].min()), float(env[:, 2].min()), float(env[:, 1].max()), float(env[:, 3].max())) if len(env) else None
                idx = np.nonzero(keep)[0]
                write_blobs(path, [crowns.blob(i) for i in idx], {c: [v[i] for i in idx] for c, v in crowns.columns.items()},
                            crowns.srs_id, extent)
            except Exception as e:
                (logger.error if logger else print)(f"Error processing file '{path}': {e}.")
