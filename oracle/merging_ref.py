"""TEST INFRASTRUCTURE ONLY — CPU restatement of the seam-strip stage of the reference: ``merge_images``
(TreeDetection/helpers.py:1023-1051), ``crop_image`` (helpers.py:1053-1085), the neighbour search
(helpers.py:984-1021) and the strip geometry / file names of ``merge_and_crop_images`` (TreeDetection/merging.py:10-118).

``merge_images`` calls ``rasterio.merge.merge([src1, src2], nodata=v)`` (third-party, un-pinned in
installations.md:232, absent here), so its published algorithm (rasterio 1.3 ``merge``, method "first") is restated:
  * output bounds = union of the datasets' bounds, resolution = the first dataset's, output size =
    round((east - west) / xres) x round((north - south) / yres), transform = from_origin(west, north, xres, yres);
  * the destination starts filled with the nodata value (the reference passes 0.0 unless src1.nodata is a sane number);
  * every dataset in order: the part of it inside the destination is read at the destination's pixel grid (nearest
    neighbour; identity for equal grids) and copied where the DESTINATION STILL HOLDS THE NODATA VALUE and the source
    pixel is not masked ("first": ``copyto(dest, new, where=dest_is_nodata & ~new_mask)``) — a value rule, not a
    footprint rule: where the first image holds the nodata value itself, a later overlapping image shows through.
``crop_image`` = the centre window ``Window(max(w//2 - width//2, 0), max(h//2 - height//2, 0), width, height)`` read
from the in-memory mosaic, with ``window_transform``.
rasterio, GDAL and the example rasters are absent and the reference holds no fixture for this stage: **parity
unpinned**; the algorithm above is rasterio's documented behaviour, the call sites are the reference's.
Plain numpy on (array [bands, rows, cols], affine (a, b, c, d, e, f)) pairs — no file I/O, loops over pixels where that
is the clearest statement."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

Affine = Tuple[float, float, float, float, float, float]


def bounds_of(transform: Affine, width: int, height: int):
    a, _, c, _, e, f = transform
    return c, f + e * height, c + a * width, f          # west, south, east, north (north-up rasters: e < 0)


def _is_value(cur, value, integer: bool) -> bool:
    if isinstance(value, float) and np.isnan(value):
        return bool(np.isnan(cur))
    return bool(cur == value) if integer else bool(np.isclose(cur, value))


def merge_first(rasters: Sequence[Tuple[np.ndarray, Affine]], nodata: float = 0.0, own_nodata: Optional[Sequence[Optional[float]]] = None):
    """rasterio.merge.merge(datasets, nodata=nodata) for north-up rasters on a common pixel grid → (data, transform).
    ``own_nodata[i]`` = dataset i's own nodata value (its GDAL_NODATA tag) or None: rasterio reads every dataset
    ``masked=True``, so a source pixel that equals the SOURCE's nodata is masked and never copied; a ``nodata`` that
    does not fit the output dtype leaves the destination at zero (rasterio warns and skips the fill)."""
    d0, t0 = rasters[0]
    integer = np.issubdtype(d0.dtype, np.integer)
    if integer:
        info = np.iinfo(d0.dtype)
        if (isinstance(nodata, float) and np.isnan(nodata)) or not (info.min <= nodata <= info.max) or float(nodata) != int(nodata):
            nodata = 0.0
    if own_nodata is None:
        own_nodata = [None] * len(rasters)
    xres, yres = t0[0], -t0[4]
    bs = [bounds_of(t, d.shape[2], d.shape[1]) for d, t in rasters]
    west, south = min(b[0] for b in bs), min(b[1] for b in bs)
    east, north = max(b[2] for b in bs), max(b[3] for b in bs)
    W, H = int(round((east - west) / xres)), int(round((north - south) / yres))
    dest = np.full((d0.shape[0], H, W), nodata, dtype=d0.dtype)
    for (data, t), own in zip(rasters, own_nodata):
        if own is not None and np.issubdtype(data.dtype, np.integer) and not (
                not np.isnan(own) and np.iinfo(data.dtype).min <= own <= np.iinfo(data.dtype).max):
            own = None                       # a tag that cannot occur in the data masks nothing
        col0 = int(round((t[2] - west) / xres))
        row0 = int(round((north - t[5]) / yres))
        for r in range(data.shape[1]):
            rr = row0 + r
            if not 0 <= rr < H:
                continue
            for c in range(data.shape[2]):
                cc = col0 + c
                if not 0 <= cc < W:
                    continue
                for b in range(min(data.shape[0], dest.shape[0])):
                    if own is not None and _is_value(data[b, r, c], own, np.issubdtype(data.dtype, np.integer)):
                        continue             # masked in the source
                    if _is_value(dest[b, rr, cc], nodata, integer):
                        dest[b, rr, cc] = data[b, r, c]
    return dest, (xres, 0.0, west, 0.0, -yres, north)


def merge_images_ref(data1, t1, data2, t2, nodata1: Optional[float] = None, nodata2: Optional[float] = None):
    """helpers.merge_images: nodata = src1.nodata unless it is None or absurdly large, then 0.0; each source keeps its
    own nodata mask (``nodata1`` / ``nodata2`` = the files' GDAL_NODATA tags)."""
    nd = nodata1
    if nd is None or abs(nd) > 1e10:
        nd = 0.0
    return merge_first([(data1, t1), (data2, t2)], nd, own_nodata=[nodata1, nodata2])


def crop_center_ref(data: np.ndarray, transform: Affine, width: int, height: int):
    """helpers.crop_image on the mosaic. A window that leaves the raster cannot be written by the reference
    (dest.write receives fewer rows / columns than the profile announces): None."""
    img_h, img_w = data.shape[1:]
    left = max(img_w // 2 - int(width) // 2, 0)
    top = max(img_h // 2 - int(height) // 2, 0)
    if left + int(width) > img_w or top + int(height) > img_h:
        return None
    a, b, c, d, e, f = transform
    return data[:, top:top + int(height), left:left + int(width)].copy(), (a, b, c + a * left + b * top, d, e, f + d * left + e * top)


def neighbours_ref(i: int, metas: List[Tuple[Affine, int, int]]):
    """helpers.retrieve_neighboring_image_filenames on indices: → (left, right, up, down) or None each; the LAST match wins."""
    (t, width, height) = metas[i]
    x, y = t[2], t[5]
    out = [None, None, None, None]
    for j, (ot, _, _) in enumerate(metas):
        if j == i:
            continue
        if abs(ot[2] - (x - width * ot[0])) < 1e-3 and abs(ot[5] - y) < 1e-3:
            out[0] = j
        if abs(ot[2] - (x + width * ot[0])) < 1e-3 and abs(ot[5] - y) < 1e-3:
            out[1] = j
        if abs(ot[5] - (y + height * ot[0])) < 1e-3 and abs(ot[2] - x) < 1e-3:
            out[2] = j
        if abs(ot[5] - (y - height * ot[0])) < 1e-3 and abs(ot[2] - x) < 1e-3:
            out[3] = j
    return tuple(out)


def seam_strips_ref(names: List[str], rasters: List[Tuple[np.ndarray, Affine]], cfg: dict, rgbi: bool):
    """merging.merge_and_crop_images for one list of rasters → [(file name, data, transform)] in the reference's order
    (per image: right neighbour's strip, then the bottom neighbour's)."""
    metas = [(t, d.shape[2], d.shape[1]) for d, t in rasters]
    out = []
    for i, (name, (data, t)) in enumerate(zip(names, rasters)):
        _, right, _, down = neighbours_ref(i, metas)
        stem = name.replace(".tif", "")
        base, end = stem.split("_")[0], stem.split("_")[-1]
        fx, fy = t[2], t[5]
        for j, horizontal in ((right, True), (down, False)):
            if j is None:
                continue
            od, ot = rasters[j]
            ox, oy = ot[2], ot[5]
            merged, mt = merge_images_ref(data, t, od, ot)
            if rgbi:
                fn = f"{base}_{round(fx)}_{round(fy)}_{round(ox)}_{round(oy)}_{end}.tif"
            else:
                fn = f"{base}_{round(fx)}{round(fy)}{round(ox)}{round(oy)}_{end}.tif"
            if horizontal:
                res = crop_center_ref(merged, mt, (cfg["tile_width"] + 2 * cfg["buffer"]) * cfg["overlapping_tiles_width"], merged.shape[1])
            else:
                res = crop_center_ref(merged, mt, merged.shape[2], (cfg["tile_height"] + 2 * cfg["buffer"]) * cfg["overlapping_tiles_height"])
            if res is not None:
                out.append((fn, res[0], res[1]))
    return out
