"""Crown post-processing (reference TreeDetection/postprocessing.py): confidence / area filters, box-IoU
de-duplication, per-crown height and NDVI statistics from the nDSM and RGBI rasters, height / NDVI / border / containment
selection, attributes (Area, TreeHeight, Centroid, Diameter, is_contained, num_contained) and the ``processed_*.gpkg``
layers — the only consumer of the nDSM side band.

What runs where: the per-crown raster statistics — in the reference a cupy test of every crown against EVERY pixel —
are one launch of ``td_crown_stats`` per raster (libtreedet_hip.so; a workgroup per crown over its circle's bounding
box, same membership arithmetic); the N x N box filters and the selection rules are small host numpy, written to follow
the reference line by line *including* its quirks, which are listed in DESIGN.md §7 and marked ``# ref:`` below.
GDAL's bilinear decimation of the rasters (``ndvi_scaling_factor`` 0.2 in the example config) is restated from its
published algorithm (:func:`resample_bilinear_gdal`, both directions). Not reproduced: fiona's
schema handling (the layer is written by
:mod:`treedetection_amd.gpkg`), and cupy's float32 reduction order for mean / variance / centroid (accumulated in
float64, rounded once). The reference holds no fixture for this stage: parity is unpinned (oracle/postprocess_ref.py).
"""
from __future__ import annotations

import json
import math
import os
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import yaml

from . import _lib
from .geotiff import GeoTiff
from .gpkg import polygon_blob, read_layer, write_blobs
from .stitching import simplify_ring

# config.py of the reference defaults none of these four (the stage raises AttributeError without them); its
# example/config.yml sets them and documents 0.2 / 1.0 as the scaling defaults. The NDVI thresholds have no documented
# default: neutral values (filter off) are used when the keys are absent.
DEFAULTS = {"height_scaling_factor": 1.0, "ndvi_scaling_factor": 0.2,
            "ndvi_mean_threshold": -1.0, "ndvi_var_threshold": float("inf")}


def _cfg(config, key):
    if key in config and config[key] is not None:
        return config[key]
    if key in DEFAULTS:
        return DEFAULTS[key]
    raise KeyError(f"config key '{key}' is required by the post-processing stage")


# ---- rasters ---------------------------------------------------------------------------------------------
def _geo_to_raster(t, x, y) -> Tuple[int, int]:
    """utilities.geo_to_raster: (int(row), int(col)), truncation toward zero."""
    if t[0] == 0 or t[4] == 0:
        raise ValueError(f"Affine transform scaling factors are zero: {t[0]}, {t[4]}")
    return int((y - t[5]) / t[4]), int((x - t[2]) / t[0])


def _window(transform, n_rows, n_cols, bounds) -> Tuple[int, int, int, int]:
    """The reference's subset of a raster for ``bounds`` (postprocessing.py:45-57; its row / col names are swapped
    twice and cancel): (row_lo, col_lo, row_hi, col_hi)."""
    minx, miny, maxx, maxy = bounds
    r_a, c_a = _geo_to_raster(transform, minx, miny)
    r_b, c_b = _geo_to_raster(transform, maxx, maxy)
    c_lo, c_hi = sorted([min(c_a, n_cols - 1), max(c_b, 0)])
    r_lo, r_hi = sorted([min(r_a, n_rows - 1), max(r_b, 0)])
    return r_lo, c_lo, r_hi, c_hi


def _decimation_weights(n_src: int, n_dst: int):
    """Taps of GDAL's convolution resampler for a bilinear (triangle) kernel when shrinking n_src → n_dst pixels
    (gcore/overview.cpp GDALResampleChunk_Convolution: kernel radius 1 stretched by the decimation ratio, weights
    1 - |d| with d in destination-pixel units, normalised over the taps that fall inside the raster)."""
    scale = n_dst / n_src
    # the same routine magnifies: its kernel is stretched only when shrinking (dfXScaleWeight = min(scale, 1)), so for n_dst > n_src
    # the radius stays ONE source pixel — plain bilinear interpolation between pixel centres, taps outside the raster dropped and
    # the rest renormalised (the border rows / columns replicate)
    sw = min(scale, 1.0)
    radius = 1.0 / sw
    rows = []
    for j in range(n_dst):
        centre = (j + 0.5) / scale
        start = max(int(math.floor(centre - radius + 0.5)), 0)
        stop = min(int(centre + radius + 0.5), n_src)
        idx = np.arange(start, stop)
        w = np.maximum(0.0, 1.0 - np.abs(sw * (idx - centre + 0.5)))
        if w.sum() <= 0.0:                       # a tap exactly one radius away on both sides: the nearest source pixel
            idx = np.array([min(max(int(centre), 0), n_src - 1)])
            w = np.ones(1)
        rows.append((idx, w / w.sum()))
    return rows


def resample_bilinear_gdal(arr: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``src.read(out_shape=(bands, out_h, out_w), resampling=Resampling.bilinear)`` (reference postprocessing.py:781-797):
    the separable triangle filter of GDAL's RasterIO — stretched by the ratio when the out_shape is SMALLER (scaling factor below
    1: decimation), one source pixel wide when it is LARGER (factor above 1: interpolation between pixel centres) — float32
    working type, integer rasters rounded half up and clamped. Same shape = plain copy. GDAL is not available here: restated
    from its published source, unpinned."""
    bands, h, w = arr.shape
    if (out_h, out_w) == (h, w):
        return arr
    if out_h < 1 or out_w < 1:
        raise ValueError(f"cannot resample a {h} x {w} raster to {out_h} x {out_w}")
    work = arr.astype(np.float32)
    tmp = np.empty((bands, h, out_w), np.float32)
    for j, (idx, wgt) in enumerate(_decimation_weights(w, out_w)):
        tmp[:, :, j] = work[:, :, idx] @ wgt.astype(np.float32)
    out = np.empty((bands, out_h, out_w), np.float32)
    for i, (idx, wgt) in enumerate(_decimation_weights(h, out_h)):
        out[:, i, :] = np.tensordot(wgt.astype(np.float32), tmp[:, idx, :], axes=([0], [1]))
    if np.issubdtype(arr.dtype, np.integer):
        info = np.iinfo(arr.dtype)
        return np.clip(np.floor(out + 0.5), info.min, info.max).astype(arr.dtype)
    return out.astype(arr.dtype)


def ndvi_from_rgbi(rgbi: np.ndarray) -> np.ndarray:
    """helpers.ndvi_array_from_rgbi (880-895): bands 0 (red) and 3 (NIR) / 255, (nir - red) / (nir + red + 1e-10)."""
    if rgbi.shape[0] < 4:
        raise ValueError(f"the RGBI raster has {rgbi.shape[0]} bands; NDVI needs the near-infrared band (index 3)")
    red = rgbi[0].astype(np.float64) / 255.0
    nir = rgbi[3].astype(np.float64) / 255.0
    return (nir - red) / (nir + red + 1e-10)


def crown_circles(rings: Sequence[np.ndarray]) -> np.ndarray:
    """[n,3] float32 (cx, cy, r): centre of the bounding box of the float32 vertex coordinates and the largest vertex
    distance from it (postprocessing.py:79-95)."""
    out = np.zeros((len(rings), 3), np.float32)
    for i, r in enumerate(rings):
        x, y = r[:, 0].astype(np.float32), r[:, 1].astype(np.float32)
        cx = (x.min() + x.max()) / np.float32(2)
        cy = (y.min() + y.max()) / np.float32(2)
        dx, dy = x - cx, y - cy
        out[i] = (cx, cy, np.sqrt(dx ** 2 + dy ** 2).max())
    return out


def crown_stats(raster: np.ndarray, transform, bounds, circles: np.ndarray, mode: int, radius_scale: float = 1.0,
                device: int = 0, clamp_shape: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """td_crown_stats over one raster → [n,3] (mode 0: max height, x, y) or [n,4] (mode 1: NDVI min, max, mean, var)."""
    n = circles.shape[0]
    cols_out = 3 if mode == 0 else 4
    if n == 0:
        return np.zeros((0, cols_out), np.float32)
    rows, cols = raster.shape
    cr, cc = clamp_shape if clamp_shape else (rows, cols)
    r_lo, c_lo, r_hi, c_hi = _window(transform, cr, cc, bounds)
    r_hi, c_hi = min(r_hi, rows - 1), min(c_hi, cols - 1)
    lib = _lib.load()
    import ctypes as C
    dev = torch.device("cuda", device)
    d_r = torch.from_numpy(np.ascontiguousarray(raster, dtype=np.float32)).to(dev)
    d_c = torch.from_numpy(np.ascontiguousarray(circles, dtype=np.float32)).to(dev)
    d_o = torch.empty((n, cols_out), dtype=torch.float32, device=dev)
    tr = (C.c_double * 6)(*[float(v) for v in transform[:6]])
    win = (C.c_int32 * 4)(r_lo, c_lo, r_hi, c_hi)
    _lib.check(lib.td_crown_stats(d_r.data_ptr(), rows, cols, tr, win, d_c.data_ptr(), n, mode, float(radius_scale),
                                  d_o.data_ptr(), _lib.stream_ptr()), "td_crown_stats")
    return d_o.cpu().numpy()


# ---- box filters (host numpy, the reference's dtypes) -------------------------------------------------------
def _box_iou(b: np.ndarray) -> np.ndarray:
    xA = np.maximum(b[:, 0][:, None], b[:, 0])
    yA = np.maximum(b[:, 1][:, None], b[:, 1])
    xB = np.minimum(b[:, 2][:, None], b[:, 2])
    yB = np.minimum(b[:, 3][:, None], b[:, 3])
    inter = np.maximum(0, xB - xA) * np.maximum(0, yB - yA)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area[:, None] + area - inter)


def filter_polygons_by_iou_and_area(bounds, areas, confidences, iou_threshold, area_threshold) -> List[int]:
    """postprocessing.py:349-406 → kept indices. ref: float32 boxes, float16 confidences / areas; ``area_threshold``
    is reused as the relative area-difference limit; removed members still vote in later groups."""
    n = len(areas)
    if n == 0:
        return []
    bb = np.array([[np.float32(v) for v in b] for b in bounds], dtype=np.float32).reshape(-1, 4)
    conf = np.array(confidences, dtype=np.float16)
    ar = np.array(areas, dtype=np.float16)
    iou = _box_iou(bb)
    with np.errstate(divide="ignore", invalid="ignore"):
        area_diff = np.abs(ar[:, None] - ar) / np.maximum(ar[:, None], ar)
    mask = (iou > iou_threshold) & (area_diff < area_threshold)
    removed = np.zeros(n, bool)
    for i in range(n):
        if removed[i]:
            continue
        connected = np.append(np.where(mask[i])[0], i)
        best = connected[int(np.argmax(conf[connected]))]
        for j in connected:
            if j != best:
                removed[j] = True
    return [i for i in range(n) if not removed[i]]


def containment(bounds, threshold):
    """postprocessing.py:408-476 on float32 boxes → (ratio[j], is_contained[j], num_contained[j])."""
    b = np.array(bounds, dtype=np.float32).reshape(-1, 4)
    n = b.shape[0]
    iw = np.maximum(0, np.minimum(b[:, 2][:, None], b[:, 2][None, :]) - np.maximum(b[:, 0][:, None], b[:, 0][None, :]))
    ih = np.maximum(0, np.minimum(b[:, 3][:, None], b[:, 3][None, :]) - np.maximum(b[:, 1][:, None], b[:, 1][None, :]))
    inner = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        ratios = (iw * ih) / inner[None, :]
    is_c = ratios >= threshold
    is_c[np.arange(n), np.arange(n)] = False
    num = is_c.sum(axis=1)
    return ([float(ratios[:, j].max()) for j in range(n)], [bool(is_c[:, j].any()) for j in range(n)], [int(v) for v in num])


def _near_border(pb, rb, eps) -> bool:
    """helpers.element_is_near_border; rb = (left, bottom, right, top)."""
    return pb[0] < rb[0] + eps or pb[2] > rb[2] - eps or pb[1] < rb[1] + eps or pb[3] > rb[3] - eps


def _round_ring(ring: np.ndarray) -> np.ndarray:
    """utilities.round_coordinates: round(coord * 1000) / 1000 with Python's round (half to even)."""
    return np.array([[round(float(c) * 1000) / 1000 for c in pt] for pt in ring], dtype=np.float64)


# ---- one layer -----------------------------------------------------------------------------------------
def process_layer(rings: List[np.ndarray], scores: Sequence[Optional[float]], config, height_path: str, rgbi_path: str,
                  device: int = 0) -> List[dict]:
    """process_geojson + process_features (postprocessing.py:722-808, 478-720) for the crowns of one image → the list
    of output features ``{"ring": [m,2], "properties": {...}}`` in the reference's order (duplicates included)."""
    conf_thr, iou_thr, area_thr = _cfg(config, "confidence_threshold"), _cfg(config, "iou_threshold"), _cfg(config, "area_threshold")
    h_scale, n_scale = float(_cfg(config, "height_scaling_factor")), float(_cfg(config, "ndvi_scaling_factor"))
    # 1-2: confidence filter, ids, areas of the simplify(2) polygons
    feats = []
    for ring, sc in zip(rings, scores):
        if sc is None or float(sc) < conf_thr:
            continue
        ring = np.asarray(ring, dtype=np.float64)
        simp = simplify_ring(ring, 2.0)
        area = abs(0.5 * float(np.dot(simp[:-1, 0], simp[1:, 1]) - np.dot(simp[1:, 0], simp[:-1, 1])))
        feats.append({"poly_id": str(len(feats)), "ring": ring, "score": sc, "area": area})
    if not feats:
        return []
    id_to_area = {f["poly_id"]: f["area"] for f in feats}
    feats = [f for f in feats if area_thr <= f["area"] <= 1000]
    # rasters, decimated by the scaling factors (postprocessing.py:780-797): transform scaled by src / out size
    hg = GeoTiff(height_path)
    hraw = hg.read()[:1]
    height = resample_bilinear_gdal(hraw, int(hg.height * h_scale), int(hg.width * h_scale))[0].astype(np.float32)
    h_t = (hg.transform[0] * (hg.width / height.shape[1]), hg.transform[1], hg.transform[2],
           hg.transform[3], hg.transform[4] * (hg.height / height.shape[0]), hg.transform[5])
    h_b = hg.bounds                                       # bounds: left, bottom, right, top
    rg = GeoTiff(rgbi_path)
    rraw = rg.read()
    rgbi = resample_bilinear_gdal(rraw, int(rg.height * n_scale), int(rg.width * n_scale))
    ndvi = ndvi_from_rgbi(rgbi).astype(np.float32)
    orig_t = rg.transform
    n_t = (orig_t[0] * (rg.width / ndvi.shape[1]), orig_t[1], orig_t[2], orig_t[3], orig_t[4] * (rg.height / ndvi.shape[0]), orig_t[5])
    n_b = rg.bounds
    hg.close()
    rg.close()
    # 3: box-IoU / area de-duplication
    def bounds_of(f):
        r = f["ring"]
        return (r[:, 0].min(), r[:, 1].min(), r[:, 0].max(), r[:, 1].max())
    kept = filter_polygons_by_iou_and_area([bounds_of(f) for f in feats], [id_to_area[f["poly_id"]] for f in feats],
                                           [f["score"] for f in feats], iou_thr, area_thr)
    features = [feats[i] for i in kept]
    if not features:
        return []
    # 4: statistics
    circles = crown_circles([f["ring"] for f in features])
    same_grid = all(abs(a - b) < 1e-5 for a, b in zip(h_t[:6], n_t[:6])) and all(abs(a - b) < 1e-3 for a, b in zip(h_b, n_b))
    if same_grid:     # get_metadata_within_polygon: NDVI in half the radius, both on the NDVI transform / bounds
        hs = crown_stats(height, n_t, n_b, circles, 0, 1.0, device, clamp_shape=height.shape)
        ns = crown_stats(ndvi, n_t, n_b, circles, 1, 0.5, device, clamp_shape=height.shape)
    else:
        hs = crown_stats(height, h_t, h_b, circles, 0, 1.0, device)
        ns = crown_stats(ndvi, n_t, n_b, circles, 1, 1.0, device)
    heights, mean_ndvi, var_ndvi = hs[:, 0], ns[:, 2], ns[:, 3]
    centroids = [(np.float32(f["ring"][:, 0].astype(np.float32).astype(np.float64).mean()),
                  np.float32(f["ring"][:, 1].astype(np.float32).astype(np.float64).mean())) for f in features]
    # preselection: image-border / overlap rules, height and NDVI thresholds
    height_thr = _cfg(config, "height_threshold")
    ndvi_mean_thr, ndvi_var_thr = _cfg(config, "ndvi_mean_threshold"), _cfg(config, "ndvi_var_threshold")
    preselected = []
    for i, f in enumerate(features):
        pb = bounds_of(f)
        if config.get("use_overlap", True):
            if _near_border(pb, n_b, 1.0):
                continue
            img_h, img_w = ndvi.shape
            sx, sy = abs(orig_t[0]), abs(orig_t[4])      # ref: the ORIGINAL image's pixel size, while img_h / img_w are
                                                         # the decimated NDVI raster's — kept as in the reference
            v_merged_h = ((config["tile_height"] + 2 * config["buffer"]) * config["overlapping_tiles_height"]) * sy
            h_merged_w = ((config["tile_width"] + 2 * config["buffer"]) * config["overlapping_tiles_width"]) * sx
            if not (img_h == v_merged_h or img_w == h_merged_w):
                right_b, left_b = n_b[2] - h_merged_w / 2.0, n_b[0] + h_merged_w / 2.0
                top_b, bottom_b = n_b[3] - v_merged_h / 2.0, n_b[1] + v_merged_h / 2.0
                if top_b < pb[1] or bottom_b > pb[3] or left_b > pb[2] or right_b < pb[0]:
                    continue
        if heights[i] < height_thr and heights[i] > -1.0:
            continue
        if (mean_ndvi[i] < ndvi_mean_thr or var_ndvi[i] > ndvi_var_thr) and mean_ndvi[i] > -1.0:
            continue
        preselected.append(f)
    # containment over ALL features of step 3
    ratios, is_cont, num_cont = containment([bounds_of(f) for f in features], _cfg(config, "containment_threshold"))
    info = {f["poly_id"]: {"is_contained": is_cont[j], "num_contained": num_cont[j], "containment_ratio": ratios[j]}
            for j, f in enumerate(features)}
    index_of = {f["poly_id"]: j for j, f in enumerate(features)}
    selected = []
    for i, f in enumerate(preselected):
        pid = f["poly_id"]
        cd = info.get(pid, {"is_contained": False, "num_contained": 0})
        if cd["num_contained"] >= 3:
            continue
        elif cd["num_contained"] == 2:
            # ref: nothing is appended on this branch — a crown containing exactly two others is dropped
            continue
        elif cd["num_contained"] == 1:
            # ref: "the other polygon" is the FIRST contained polygon of the whole list, not the one this crown contains
            other_id = [g["poly_id"] for g in features if info[g["poly_id"]]["is_contained"]][0]
            other = features[index_of[other_id]]
            if abs(mean_ndvi[index_of[pid]] - mean_ndvi[index_of[other_id]]) > 0.05:
                if var_ndvi[i] < var_ndvi[index_of[other_id]]:      # ref: var_ndvi[i] — position in the PREselected list
                    selected.append(f)
                else:
                    selected.append(other)
            elif id_to_area.get(pid, 0) > 0:                         # ref: the other area is looked up with an int key → 0
                selected.append(f)
        else:
            selected.append(f)
    out = []
    for f in selected:
        pid = f["poly_id"]
        j = index_of[pid]
        area = id_to_area.get(pid)
        cd = info.get(pid, {"is_contained": False, "num_contained": -1})
        props = {"Confidence_score": float(f["score"]), "poly_id": pid, "Area": float(area), "TreeHeight": float(heights[j]),
                 "Centroid": json.dumps({"x": float(centroids[j][0]), "y": float(centroids[j][1])}),
                 "Diameter": float(2 * (area / np.pi) ** 0.5), "is_contained": str(cd["is_contained"]),
                 "num_contained": int(cd["num_contained"])}
        out.append({"ring": _round_ring(f["ring"]), "properties": props})
    return out


COLUMNS = ("Confidence_score", "poly_id", "Area", "TreeHeight", "Centroid", "Diameter", "is_contained", "num_contained")


def process_single_file(file_path, processed_file_path, height_data_path, rgbi_data_path, device_id=0, config=None):
    """postprocessing.py:876-943: one stitched layer → ``processed_<name>.gpkg``; returns ``file_path`` or None on error
    (printed, like the reference)."""
    try:
        layer = read_layer(file_path)
        scores = layer.columns.get("Confidence_score", [None] * len(layer))
        feats = process_layer(layer.rings(), scores, config, height_data_path, rgbi_data_path,
                              int(str(device_id).replace("cuda:", "") or 0) if str(device_id) != "cpu" else 0)
        rings = [f["ring"] for f in feats]
        cols = {c: [f["properties"][c] for f in feats] for c in COLUMNS}
        if rings:
            xy = np.concatenate(rings)
            extent = (float(xy[:, 0].min()), float(xy[:, 1].min()), float(xy[:, 0].max()), float(xy[:, 1].max()))
        else:
            extent, cols = None, {}
        write_blobs(processed_file_path, (polygon_blob(r, layer.srs_id) for r in rings), cols, layer.srs_id, extent)
        return file_path
    except Exception as e:
        print(f"Error postprocessing file {file_path}: {e}")
        return None


# ---- resume file (postprocessing.py:827-874) -----------------------------------------------------------------
_PARAM_KEYS = ("tile_width", "tile_height", "buffer", "confidence_threshold", "containment_threshold", "height_threshold",
               "ndvi_mean_threshold", "ndvi_var_threshold", "iou_threshold", "confidence_threshold_stitching", "area_threshold")


def load_recovery_data_with_params(directory, config, logger=None):
    recovery_file = os.path.join(directory, "recovery.yaml")
    processed = set()
    params = {k: (_cfg(config, k) if k in DEFAULTS else config.get(k)) for k in _PARAM_KEYS}
    params = {k: (v if not (isinstance(v, float) and math.isinf(v)) else str(v)) for k, v in params.items()}
    if os.path.exists(recovery_file):
        try:
            with open(recovery_file) as f:
                data = yaml.safe_load(f)
            if data.get("parameters") == params:
                processed = set(data.get("processed_files", []))
                if logger:
                    logger.info(f"Loaded {len(processed)} previously processed files from recovery.")
            elif logger:
                logger.info("Parameter mismatch with recovery file. Resetting processed files.")
        except Exception as e:
            if logger:
                logger.warning(f"Failed to load recovery file: {e}")
    return params, processed


def save_recovery_data_with_params(directory, params, processed_files, logger=None):
    try:
        with open(os.path.join(directory, "recovery.yaml"), "w") as f:
            yaml.safe_dump({"parameters": params, "processed_files": sorted(p for p in processed_files if p)}, f, sort_keys=False)
        if logger:
            logger.info(f"Saved recovery file with {len(processed_files)} entries.")
    except Exception as e:
        if logger:
            logger.warning(f"Failed to save recovery file: {e}")


def process_files_in_directory(directory, height_directory, image_directory, parallel=True, filename_pattern=None, config=None):
    """postprocessing.py:945-1076: every stitched ``*.gpkg`` of ``directory`` that is not ``processed_*`` and not in the
    resume file → ``processed_<name>.gpkg``, with the image / height rasters found by the identifier regexes (plain
    names first, then the merged-strip patterns, searched recursively)."""
    logger = config.get("logger")
    params, processed = load_recovery_data_with_params(directory, config, logger)
    files = sorted(f for f in os.listdir(directory) if f.endswith(".gpkg") and not f.startswith("processed_"))
    files = [f for f in files if os.path.join(directory, f) not in processed]
    image_pattern, height_pattern = filename_pattern or (None, None)
    image_pattern = re.compile(image_pattern or "(\\d+)\\.tif")
    height_pattern = re.compile(height_pattern or "(\\d+)\\.tif")
    image_merged = re.compile(config.get("image_merged_regex", "FDOP20_(\\d+)_(\\d+)_(\\d+)_(\\d+)_(\\d+)\\.tif"))
    height_merged = re.compile(config.get("height_data_merged_regex", "FDOP20_(\\d+)_(\\d+)\\.tif"))

    def find(base_name, name_pattern, search_pattern, where):
        m = name_pattern.match(base_name + ".tif")
        if not m:
            return None
        want = "".join(m.groups())
        for root, _, names in os.walk(where):
            for n in sorted(names):
                sm = search_pattern.match(n)
                if sm and "".join(sm.groups()[:len(m.groups())]) == want:
                    return os.path.join(root, n)
        return None

    device = config.get("device", "0")
    for filename in files:
        base = os.path.splitext(filename)[0]
        hpath = find(base, image_pattern, height_pattern, height_directory)
        ipath = find(base, image_pattern, image_pattern, image_directory)
        if hpath is None or ipath is None:
            hpath = find(base, image_merged, height_merged, height_directory)
            ipath = find(base, image_merged, image_merged, image_directory)
        if hpath and ipath:
            res = process_single_file(os.path.join(directory, filename), os.path.join(directory, f"processed_{filename}"),
                                      hpath, ipath, device_id=device, config=config)
            if res is not None:
                processed.add(res)
        elif logger:
            logger.warning(f"Height data file not found for: {filename}, searched pattern for base name: {base}")
    save_recovery_data_with_params(directory, params, processed, logger)
