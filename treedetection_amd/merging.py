"""Seam strips between neighbouring images (reference TreeDetection/merging.py:10-118 ``merge_and_crop_images`` with
helpers.py ``retrieve_neighboring_image_filenames`` 984-1021, ``merge_images`` 1023-1051, ``crop_image`` 1053-1085).

With ``use_overlap`` the reference mosaics every image with its right and its bottom neighbour (rasterio.merge), cuts
a strip ``(tile + 2*buffer) * overlapping_tiles`` pixels wide out of the middle of the mosaic — the seam — writes it to
``<dir>/<merged_path>/`` and appends it to the work list, so crowns cut by an image border are predicted whole once.
Same steps here on :class:`treedetection_amd.geotiff.GeoTiff` (no rasterio): neighbours by origin arithmetic,
mosaic = rasterio.merge's "first" rule on the union extent at the first image's resolution (oracle/merging_ref.py),
centre crop, uncompressed GeoTIFF out; same file names, same ordering of the appended paths.
"""
from __future__ import annotations

import os
import warnings
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import numpy as np

from .config import Config
from .geotiff import GeoTiff, write_geotiff


def tif_geoinfo(filename) -> Tuple[tuple, Optional[int], int, int]:
    g = GeoTiff(filename)
    return g.transform, g.epsg, g.width, g.height


def retrieve_neighboring_image_filenames(filename, other_filenames, meta_info: Optional[Dict[str, tuple]] = None):
    """(left, right, up, down) neighbours of ``filename``: images whose origin lies exactly one image width / height
    away (eps 1e-3), as the reference computes it — offsets use the OTHER image's pixel size and THIS image's size."""
    transform, _, width, height = tif_geoinfo(filename)
    x, y = transform[2], transform[5]
    left = right = up = down = None
    eps = 1e-3
    for other in other_filenames:
        if other == filename:
            continue
        ot = meta_info[other] if meta_info is not None else tif_geoinfo(other)[0]
        oa, oc, of = ot[0], ot[2], ot[5]
        if abs(oc - (x - width * oa)) < eps and abs(of - y) < eps:
            left = other
        if abs(oc - (x + width * oa)) < eps and abs(of - y) < eps:
            right = other
        if abs(of - (y + height * oa)) < eps and abs(oc - x) < eps:
            up = other
        if abs(of - (y - height * oa)) < eps and abs(oc - x) < eps:
            down = other
    return left, right, up, down


def _fill_value(nodata: float, dt: np.dtype) -> float:
    """The value the mosaic starts from. rasterio.merge checks the nodata value against the range of the output dtype and,
    when it does not fit (GDAL_NODATA -9999 on a uint8 raster), warns and leaves the destination at zero.
    DELIBERATE DEVIATION for that case (read from rasterio's source, rasterio is not installed here to confirm):
    rasterio then still compares the destination with the ORIGINAL out-of-range value (``region == -9999`` on uint8 is False
    everywhere), so nothing is ever copied and its mosaic stays all zero — a reference result nobody can want. This port
    compares against the value the destination really holds (0): both images are copied by the "first" rule. For every
    nodata value that fits the dtype — the reference's own rasters — the two agree (tests/test_merging.py)."""
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        if np.isnan(nodata) or not (info.min <= nodata <= info.max) or float(nodata) != int(nodata):
            warnings.warn(f"nodata value {nodata} is beyond the valid range of {dt}: the mosaic starts from 0 (as rasterio.merge does)")
            return 0.0
    return float(nodata)


def _equals(a: np.ndarray, value: float) -> np.ndarray:
    if np.issubdtype(a.dtype, np.integer):
        return a == a.dtype.type(value)
    if np.isnan(value):
        return np.isnan(a)
    return np.isclose(a, value)


def merge_images(src1: GeoTiff, src2: GeoTiff):
    """rasterio.merge.merge([src1, src2], nodata=v), method "first", v = src1's nodata unless it is missing or absurd
    (then 0.0): union extent on src1's grid, filled with v; each image in turn is copied where the mosaic STILL HOLDS v
    AND the image's pixel is not one of ITS OWN nodata pixels (rasterio reads every source masked by the source's
    nodata: ``copyto(dest, new, where=dest_is_nodata & ~new_mask)``). A value rule on the destination side: where the
    first image holds v itself a later overlapping image shows through — for the exactly adjacent neighbours the
    reference merges the footprints never overlap. → (data [bands, rows, cols], transform)."""
    if src1.epsg != src2.epsg:
        raise ValueError("CRS of the two images do not match.")
    nodata = getattr(src1, "nodata", None)
    if nodata is None or abs(nodata) > 1e10:
        nodata = 0.0
    a, _, c1, _, e, f1 = src1.transform
    _, _, c2, _, _, f2 = src2.transform
    left, top = min(c1, c2), max(f1, f2)
    right = max(c1 + a * src1.width, c2 + src2.transform[0] * src2.width)
    bottom = min(f1 + e * src1.height, f2 + src2.transform[4] * src2.height)
    W, H = int(round((right - left) / a)), int(round((bottom - top) / e))
    dt = src1.dtype.newbyteorder("=")
    nodata = _fill_value(nodata, dt)
    out = np.full((src1.count, H, W), nodata, dtype=dt)
    for src in (src1, src2):
        col0, row0 = int(round((src.transform[2] - left) / a)), int(round((src.transform[5] - top) / e))
        data = src.read()
        h, w = min(src.height, H - row0), min(src.width, W - col0)
        bands = min(src.count, out.shape[0])
        view = out[:bands, row0:row0 + h, col0:col0 + w]
        new = data[:bands, :h, :w]
        free = _equals(view, nodata)
        own = getattr(src, "nodata", None)
        if own is not None:                      # the source's own nodata pixels stay out of the mosaic
            sdt = new.dtype
            fits = not np.issubdtype(sdt, np.integer) or (not np.isnan(own) and np.iinfo(sdt).min <= own <= np.iinfo(sdt).max)
            if fits:
                free &= ~_equals(new, own)
        np.copyto(view, new, where=free, casting="unsafe")
    return out, (a, 0.0, left, 0.0, e, top)


def crop_image(data: np.ndarray, transform, width: int, height: int):
    """Centre crop of ``width`` x ``height`` PIXELS (the reference passes metres here; kept). Raises when the window
    does not fit, which is where the reference's ``dest.write`` fails and the strip is skipped."""
    img_h, img_w = data.shape[1:]
    left = max(img_w // 2 - int(width) // 2, 0)
    top = max(img_h // 2 - int(height) // 2, 0)
    width, height = int(width), int(height)
    if left + width > img_w or top + height > img_h:
        raise ValueError(f"crop window {width}x{height} at ({left},{top}) exceeds the {img_w}x{img_h} mosaic")
    a, b, c, d, e, f = transform
    return data[:, top:top + height, left:left + width], (a, b, c + a * left + b * top, d, e, f + d * left + e * top)


def merge_and_crop_images(config, images_paths: List[str], height_paths: List[str]) -> None:
    """Appends the seam-strip files of the RGBI images to ``images_paths`` and those of the height rasters to
    ``height_paths`` (both lists are extended in place, like the reference)."""
    Config()._load_into_config(config)
    logger = config["logger"]
    merged_directory = config["merged_path"]

    def crop_single_image(paths, rgbi, meta_info, f):
        names = []
        _, right, _, down = retrieve_neighboring_image_filenames(f, paths, meta_info)
        result_directory = f"{os.path.dirname(f)}/{merged_directory}"
        os.makedirs(result_directory, exist_ok=True)
        stem = os.path.basename(f).replace(".tif", "")
        f_basename, f_name_end = stem.split("_")[0], stem.split("_")[-1]
        fx, fy = meta_info[f][2], meta_info[f][5]
        for other, horizontal in ((right, True), (down, False)):
            if other is None:
                continue
            ox, oy = meta_info[other][2], meta_info[other][5]
            try:
                first, second = GeoTiff(f), GeoTiff(other)
                merged, mt = merge_images(first, second)
                if rgbi:
                    name = f"{f_basename}_{round(fx)}_{round(fy)}_{round(ox)}_{round(oy)}_{f_name_end}.tif"
                else:
                    name = f"{f_basename}_{round(fx)}{round(fy)}{round(ox)}{round(oy)}_{f_name_end}.tif"
                if horizontal:
                    crop, ct = crop_image(merged, mt, (config["tile_width"] + 2 * config["buffer"]) * config["overlapping_tiles_width"],
                                          merged.shape[1])
                else:
                    crop, ct = crop_image(merged, mt, merged.shape[2],
                                          (config["tile_height"] + 2 * config["buffer"]) * config["overlapping_tiles_height"])
                write_geotiff(f"{result_directory}/{name}", np.ascontiguousarray(crop), ct, first.epsg or 0)
                names.append(f"{result_directory}/{name}")
            except Exception as e:
                logger.error(f"Error merging images {f} and {other}: {e}")
        return names

    def save_cropped_images(paths, rgbi=True):
        meta_info = {f: tif_geoinfo(f)[0] for f in paths}
        with ThreadPoolExecutor(max_workers=min(8, max(1, len(os.sched_getaffinity(0))))) as ex:
            results = list(ex.map(lambda f: crop_single_image(paths, rgbi, meta_info, f), list(paths)))
        return [n for sub in results for n in sub]

    try:
        cropped_images = save_cropped_images(images_paths, rgbi=True)
        cropped_heights = save_cropped_images(height_paths, rgbi=False)
        images_paths.extend(cropped_images)
        height_paths.extend(cropped_heights)
    except Exception as e:
        logger.error(f"Error merging and cropping images: {e}")
