"""TEST INFRASTRUCTURE ONLY — CPU restatement of the crown post-processing math of the reference
(TreeDetection/postprocessing.py: ``get_height_within_polygon`` 25-115, ``get_ndvi_within_polygon`` 117-219,
``get_metadata_within_polygon`` 221-347, ``filter_polygons_by_iou_and_area`` 349-406,
``process_containment_features`` 408-476, the selection rules of ``process_features`` 572-670; helpers
``ndvi_array_from_rgbi`` helpers.py:880-895, ``utilities.geo_to_raster`` / ``raster_to_geo`` /
``is_point_in_polygon_batch`` / ``calculate_iou`` / ``get_centroids``).

The reference runs these on cupy against EVERY pixel of the raster per polygon; this file follows it literally in
numpy (same array types, same order of operations, brute force) so that the product path — which only visits each
polygon's bounding box, in a HIP kernel — can be compared with it on small rasters. cupy, rasterio, fiona and shapely
are absent here and the reference holds no fixture for this stage: **parity unpinned**. Known places where even a
literal restatement cannot be bit-identical to cupy: float32 ``mean`` / ``var`` / ``nanmean`` reductions (cupy's
accumulation order is its own); they are computed in float64 and rounded once here and in the product.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def geo_to_raster(transform, x, y) -> Tuple[int, int]:
    a, b, c, d, e, f = transform[:6]
    if a == 0 or e == 0:
        raise ValueError("Affine transform scaling factors are zero")
    return int((y - f) / e), int((x - c) / a)          # (row, col), truncated toward zero like int()


def ndvi_from_rgbi(rgbi: np.ndarray) -> np.ndarray:
    """helpers.ndvi_array_from_rgbi: (NIR - R) / (NIR + R + 1e-10) on bands 3 and 0 scaled by 1/255, float64."""
    red = rgbi[0].astype(np.float64) / 255.0
    nir = rgbi[3].astype(np.float64) / 255.0
    return (nir - red) / (nir + red + 1e-10)


def _subset(data, transform, n_rows, n_cols, bounds):
    """The reference's window arithmetic with its swapped names kept straight: rows clamp to n_rows - 1, cols to
    n_cols - 1; returns (subset, row0, col0)."""
    minx, miny, maxx, maxy = bounds
    r_a, c_a = geo_to_raster(transform, minx, miny)
    r_b, c_b = geo_to_raster(transform, maxx, maxy)
    c_lo, c_hi = sorted([min(c_a, n_cols - 1), max(c_b, 0)])
    r_lo, r_hi = sorted([min(r_a, n_rows - 1), max(r_b, 0)])
    return data[r_lo:r_hi + 1, c_lo:c_hi + 1], r_lo, c_lo


def _pixel_coords(sub_shape, transform, r_lo, c_lo):
    """raster_to_geo(transform, rows + min_row, cols + min_col) AS CALLED by the reference: the row offset is added to
    the column index and vice versa (both are 0 whenever the bounds are the raster's own)."""
    a, b, c, d, e, f = transform[:6]
    rows, cols = np.meshgrid(np.arange(sub_shape[0]), np.arange(sub_shape[1]), indexing="ij")
    rows, cols = rows.flatten(), cols.flatten()
    row_arg, col_arg = rows + c_lo, cols + r_lo            # the swap
    return a * col_arg + b * row_arg + c, d * col_arg + e * row_arg + f


def _circle(px32: np.ndarray, py32: np.ndarray):
    """centre of the bounding box and the largest vertex distance from it, all float32 like the cupy arrays."""
    ok = ~np.isnan(px32) & ~np.isnan(py32)
    vx, vy = px32[ok], py32[ok]
    cx = (vx.min() + vx.max()) / np.float32(2)
    cy = (vy.min() + vy.max()) / np.float32(2)
    dx, dy = vx - cx, vy - cy
    return cx, cy, np.sqrt(dx ** 2 + dy ** 2).max()


def heights_within(polys_x32, polys_y32, height, transform, bounds):
    """get_height_within_polygon: float64 pixel coordinates against the float32 circle → (max height, its x, y) per
    polygon, (-1, -1, -1) when the circle holds no pixel."""
    sub, r_lo, c_lo = _subset(height, transform, height.shape[0], height.shape[1], bounds)
    xs, ys = _pixel_coords(sub.shape, transform, r_lo, c_lo)
    flat = sub.flatten()
    out_h, out_xy = np.zeros(len(polys_x32), np.float32), np.zeros((len(polys_x32), 2), np.float32)
    for i, (px, py) in enumerate(zip(polys_x32, polys_y32)):
        cx, cy, rad = _circle(px, py)
        inside = (xs - cx) ** 2 + (ys - cy) ** 2 <= rad ** 2
        if not inside.any():
            out_h[i], out_xy[i] = -1, (-1, -1)
            continue
        k = int(np.argmax(flat[inside]))
        out_h[i] = flat[inside][k]
        out_xy[i] = (xs[inside][k], ys[inside][k])
    return out_h, out_xy


def ndvi_within(polys_x32, polys_y32, ndvi, transform, bounds, radius_scale=1.0):
    """get_ndvi_within_polygon (radius_scale 1) / the NDVI half of get_metadata_within_polygon (0.5): float32 pixel
    coordinates → (min, max, mean, var) per polygon, -1 when empty."""
    ndvi32 = ndvi.astype(np.float32)
    sub, r_lo, c_lo = _subset(ndvi32, transform, ndvi.shape[0], ndvi.shape[1], bounds)
    xs, ys = _pixel_coords(sub.shape, transform, r_lo, c_lo)
    xs, ys = xs.astype(np.float32), ys.astype(np.float32)
    flat = sub.flatten()
    res = np.zeros((4, len(polys_x32)), np.float32)
    for i, (px, py) in enumerate(zip(polys_x32, polys_y32)):
        cx, cy, rad = _circle(px, py)
        rad = rad * np.float32(radius_scale)
        inside = (xs - cx) ** 2 + (ys - cy) ** 2 <= rad ** 2
        v = flat[inside]
        if v.shape[0] == 0:
            res[:, i] = -1
        else:
            v64 = v.astype(np.float64)
            res[:, i] = (v.min(), v.max(), np.float32(v64.mean()), np.float32(v64.var()))
    return res[0], res[1], res[2], res[3]


def box_iou(b1: np.ndarray, b2: np.ndarray) -> np.ndarray:
    xA = np.maximum(b1[:, 0][:, None], b2[:, 0])
    yA = np.maximum(b1[:, 1][:, None], b2[:, 1])
    xB = np.minimum(b1[:, 2][:, None], b2[:, 2])
    yB = np.minimum(b1[:, 3][:, None], b2[:, 3])
    inter = np.maximum(0, xB - xA) * np.maximum(0, yB - yA)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (a1[:, None] + a2 - inter)


def filter_by_iou_and_area(bounds: Sequence[Sequence[float]], areas: Sequence[float], confidences: Sequence[float],
                           iou_threshold: float, area_threshold: float) -> List[int]:
    """filter_polygons_by_iou_and_area → indices kept. float32 boxes, float16 confidences and areas, groups visited in
    index order; a group's survivor is its first highest-confidence member (removed members still take part in later
    groups, exactly as in the reference)."""
    n = len(areas)
    bb = np.array([[np.float32(v) for v in b] for b in bounds], dtype=np.float32).reshape(-1, 4)
    conf = np.array(confidences, dtype=np.float16)
    ar = np.array(areas, dtype=np.float16)
    iou = box_iou(bb, bb)
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


def containment(bounds: Sequence[Sequence[float]], threshold: float):
    """process_containment_features on float32 boxes → per polygon j: (max containment ratio over all outers incl.
    itself, is j contained in another, how many others j contains)."""
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
    return [float(ratios[:, j].max()) for j in range(n)], [bool(is_c[:, j].any()) for j in range(n)], [int(v) for v in num]
