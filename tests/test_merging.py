"""Seam strips between neighbouring images (reference merging.py:10-118, helpers.py:984-1085)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from treedetection_amd.geotiff import GeoTiff, write_geotiff  # noqa: E402
from treedetection_amd.merging import (crop_image, merge_and_crop_images, merge_images,  # noqa: E402
                                       retrieve_neighboring_image_filenames, tif_geoinfo)


class Log:
    def __init__(self):
        self.msgs = []

    def __getattr__(self, name):
        return lambda m: self.msgs.append((name, m))


def _grid(tmp_path, sub, bands, dtype, gsd):
    """2 x 2 images of 100 x 80 px; pixel value encodes (image, row, col)."""
    d = tmp_path / sub
    os.makedirs(d)
    paths, arrays = {}, {}
    for iy in range(2):
        for ix in range(2):
            k = iy * 2 + ix
            yy, xx = np.mgrid[0:80, 0:100]
            img = np.stack([(k * 50 + (yy + b) % 50) for b in range(bands)]).astype(dtype)
            img[0] = (xx % 200) + k
            name = str(d / f"{3241 + k}.tif")
            x0, y0 = 412000.0 + ix * 100 * gsd, 5318000.0 - iy * 80 * gsd
            write_geotiff(name, img, (gsd, 0, x0, 0, -gsd, y0), 25832)
            paths[(ix, iy)], arrays[(ix, iy)] = name, img
    return paths, arrays


def test_neighbours_and_mosaic(tmp_path):
    paths, arrays = _grid(tmp_path, "rgb", 4, np.uint8, 0.2)
    all_paths = sorted(paths.values())
    meta = {p: tif_geoinfo(p)[0] for p in all_paths}
    assert retrieve_neighboring_image_filenames(paths[(0, 0)], all_paths, meta) == (None, paths[(1, 0)], None, paths[(0, 1)])
    assert retrieve_neighboring_image_filenames(paths[(1, 1)], all_paths, meta) == (paths[(0, 1)], None, paths[(1, 0)], None)
    m, t = merge_images(GeoTiff(paths[(0, 0)]), GeoTiff(paths[(1, 0)]))
    assert m.shape == (4, 80, 200) and t == (0.2, 0.0, 412000.0, 0.0, -0.2, 5318000.0)
    assert np.array_equal(m[:, :, :100], arrays[(0, 0)]) and np.array_equal(m[:, :, 100:], arrays[(1, 0)])
    m, t = merge_images(GeoTiff(paths[(0, 1)]), GeoTiff(paths[(0, 0)]))       # order of arguments does not move the grid
    assert m.shape == (4, 160, 100) and t[5] == 5318000.0 and np.array_equal(m[:, :80], arrays[(0, 0)])
    c, ct = crop_image(m, t, 100, 30)
    assert c.shape == (4, 30, 100) and np.array_equal(c, m[:, 65:95]) and ct[5] == 5318000.0 - 0.2 * 65
    try:
        crop_image(m, t, 120, 30)
        assert False
    except ValueError:
        pass


def test_merge_and_crop_images(tmp_path):
    rgb, arrays = _grid(tmp_path, "rgb", 4, np.uint8, 0.2)
    ndsm, harrays = _grid(tmp_path, "ndsm", 1, np.float32, 0.2)
    images, heights = sorted(rgb.values()), sorted(ndsm.values())
    cfg = {"logger": Log(), "merged_path": "merged", "tile_width": 5, "tile_height": 4, "buffer": 1,
           "overlapping_tiles_width": 3, "overlapping_tiles_height": 2}
    n_img = len(images)
    merge_and_crop_images(cfg, images, heights)
    new = images[n_img:]
    assert len(new) == 4 and len(heights) == n_img + 4                      # 2 horizontal + 2 vertical seams
    d = str(tmp_path / "rgb" / "merged")
    want = [f"{d}/3241_412000_5318000_412020_5318000_3241.tif", f"{d}/3241_412000_5318000_412000_5317984_3241.tif",
            f"{d}/3242_412020_5318000_412020_5317984_3242.tif", f"{d}/3243_412000_5317984_412020_5317984_3243.tif"]
    assert new == want
    g = GeoTiff(want[0])                                                    # strip over the vertical seam of the top row
    assert (g.width, g.height, g.count, g.epsg) == (21, 80, 4, 25832)       # (5 + 2*1) * 3 px wide, full height
    full = np.concatenate([arrays[(0, 0)], arrays[(1, 0)]], axis=2)
    assert np.array_equal(g.read(), full[:, :, 90:111]) and g.transform[2] == 412000.0 + 0.2 * 90
    g = GeoTiff(want[1])                                                    # strip over the horizontal seam of the left column
    assert (g.width, g.height) == (100, 12)                                 # (4 + 2*1) * 2 px tall, full width
    full = np.concatenate([arrays[(0, 0)], arrays[(0, 1)]], axis=1)
    assert np.array_equal(g.read(), full[:, 74:86]) and g.transform[5] == 5318000.0 - 0.2 * 74
    hd = str(tmp_path / "ndsm" / "merged")
    assert heights[n_img] == f"{hd}/3241_41200053180004120205318000_3241.tif"      # height strips: no separators
    assert GeoTiff(heights[n_img]).read().dtype == np.float32
    assert not [m for lvl, m in cfg["logger"].msgs if lvl == "error"]
    # a strip wider than the mosaic cannot be cut: logged, skipped, the rest still comes out
    images2 = sorted(rgb.values())
    cfg["overlapping_tiles_width"] = 40
    merge_and_crop_images(cfg, images2, sorted(ndsm.values()))
    assert len(images2) == n_img + 2 and any("exceeds" in m for lvl, m in cfg["logger"].msgs if lvl == "error")


# ---- against the oracle (oracle/merging_ref.py: rasterio.merge "first" + the reference's centre crop, restated) -------------
from oracle.merging_ref import crop_center_ref, merge_images_ref, neighbours_ref, seam_strips_ref  # noqa: E402


def _rand_raster(rng, bands, h, w, dtype, zero_frac=0.0):
    if np.issubdtype(np.dtype(dtype), np.integer):
        a = rng.integers(1, 250, (bands, h, w)).astype(dtype)
    else:
        a = rng.uniform(0.5, 40.0, (bands, h, w)).astype(dtype)
    if zero_frac:
        a[:, rng.random((h, w)) < zero_frac] = 0
    return a


def test_merge_images_matches_the_oracle_adjacent_and_overlapping(tmp_path):
    """Adjacent neighbours (what the reference merges) and OVERLAPPING rasters with nodata-valued pixels in the first one
    (rasterio's "first" is a value rule: a later image shows through where the mosaic still holds the nodata value)."""
    rng = np.random.default_rng(7)
    gsd = 0.2
    cases = [
        # dtype, bands, (h1, w1), (h2, w2), offset of raster 2 in pixels (dx, dy), nodata tag of raster 1, zero fraction
        (np.uint8, 4, (40, 50), (40, 50), (50, 0), None, 0.0),        # right neighbour
        (np.uint8, 4, (40, 50), (40, 50), (0, 40), None, 0.0),        # bottom neighbour
        (np.float32, 1, (30, 30), (30, 30), (30, 0), None, 0.0),
        (np.uint8, 3, (40, 50), (40, 50), (35, 0), None, 0.3),        # 15-px overlap, zeros in both: second shows through
        (np.uint8, 3, (40, 50), (30, 60), (-20, 25), None, 0.3),      # partial overlap, second starts left of the first
        (np.float32, 1, (32, 32), (32, 32), (16, 16), -9999.0, 0.2),  # nodata tag on the first raster: -9999 is the hole value
        (np.float32, 1, (32, 32), (32, 32), (16, 0), 3.4e38, 0.2),    # absurd nodata → 0.0 (helpers.py:1037)
        (np.uint16, 2, (20, 25), (20, 25), (10, 5), None, 0.5),
    ]
    for k, (dtype, bands, s1, s2, (dx, dy), nd, zf) in enumerate(cases):
        a1 = _rand_raster(rng, bands, s1[0], s1[1], dtype, zf)
        a2 = _rand_raster(rng, bands, s2[0], s2[1], dtype, zf)
        if nd is not None and abs(nd) < 1e10:
            a1[:, rng.random(s1) < 0.2] = nd
        t1 = (gsd, 0.0, 412000.0, 0.0, -gsd, 5318000.0)
        t2 = (gsd, 0.0, 412000.0 + dx * gsd, 0.0, -gsd, 5318000.0 - dy * gsd)
        p1, p2 = str(tmp_path / f"a{k}.tif"), str(tmp_path / f"b{k}.tif")
        write_geotiff(p1, a1, t1, 25832, nodata=nd)
        write_geotiff(p2, a2, t2, 25832)
        g1 = GeoTiff(p1)
        assert g1.nodata == nd
        got, gt = merge_images(g1, GeoTiff(p2))
        ref, rt = merge_images_ref(a1, t1, a2, t2, nd)
        assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, ref.shape)
        assert np.array_equal(got, ref), k
        assert np.allclose(gt, rt, rtol=0, atol=1e-9), (k, gt, rt)
        for (cw, ch) in ((ref.shape[2], 12), (14, ref.shape[1]), (ref.shape[2] + 2, 4)):
            want = crop_center_ref(ref, rt, cw, ch)
            if want is None:
                with pytest.raises(ValueError):
                    crop_image(got, gt, cw, ch)
            else:
                c, ct = crop_image(got, gt, cw, ch)
                assert np.array_equal(c, want[0]) and np.allclose(ct, want[1], rtol=0, atol=1e-9)


def test_seam_strips_match_the_oracle_on_a_3x3_mosaic(tmp_path):
    """merge_and_crop_images end to end (files, names, order, georeferencing) vs the oracle's in-memory restatement."""
    rng = np.random.default_rng(11)
    gsd, W, H = 0.25, 64, 48
    for sub, bands, dtype, rgbi in (("rgb", 4, np.uint8, True), ("ndsm", 1, np.float32, False)):
        d = tmp_path / sub
        os.makedirs(d)
        names, rasters = [], []
        for iy in range(3):
            for ix in range(3):
                if (ix, iy) == (1, 1) and not rgbi:
                    continue                                  # a hole in the height mosaic: fewer neighbours there
                arr = _rand_raster(rng, bands, H, W, dtype, 0.1)
                t = (gsd, 0.0, 500000.0 + ix * W * gsd, 0.0, -gsd, 5400000.0 - iy * H * gsd)
                name = str(d / f"{7000 + iy * 3 + ix}_x.tif")
                write_geotiff(name, arr, t, 25832)
                names.append(name)
                rasters.append((arr, t))
        cfg = {"logger": Log(), "merged_path": "merged", "tile_width": 6, "tile_height": 5, "buffer": 1,
               "overlapping_tiles_width": 2, "overlapping_tiles_height": 3}
        imgs, hts = (list(names), []) if rgbi else ([], list(names))
        merge_and_crop_images(cfg, imgs, hts)
        new = (imgs if rgbi else hts)[len(names):]
        want = seam_strips_ref([os.path.basename(n) for n in names], rasters, cfg, rgbi)
        assert [os.path.basename(n) for n in new] == [w[0] for w in want]
        assert len(want) == (12 if rgbi else 8)
        for path, (fn, data, t) in zip(new, want):
            g = GeoTiff(path)
            assert np.array_equal(g.read(), data), fn
            assert np.allclose(g.transform, t, rtol=0, atol=1e-9), fn
        metas = [(t, a.shape[2], a.shape[1]) for a, t in rasters]
        meta_info = {n: m[0] for n, m in zip(names, metas)}
        for i, n in enumerate(names):
            got = retrieve_neighboring_image_filenames(n, names, meta_info)
            ref = neighbours_ref(i, metas)
            assert got == tuple(None if j is None else names[j] for j in ref)
        assert not [m for lvl, m in cfg["logger"].msgs if lvl == "error"]


def test_merge_images_masks_each_source_by_its_own_nodata(tmp_path):
    """rasterio reads every source ``masked=True``: pixels that equal the SOURCE's own GDAL_NODATA are never copied, also
    when the mosaic's fill value is another one (first raster untagged → 0; second tagged -9999): they stay at the fill
    value instead of showing up as -9999 in the seam strip. A tag that does not fit the dtype (-9999 on a
    uint8 raster) warns and fills with 0, as rasterio.merge does."""
    rng = np.random.default_rng(11)
    gsd = 0.2
    t1 = (gsd, 0.0, 412000.0, 0.0, -gsd, 5318000.0)
    for k, (dx, dy) in enumerate(((32, 0), (16, 8))):
        a1 = _rand_raster(rng, 1, 32, 32, np.float32, 0.2)          # untagged; zeros are its "holes" by the value rule
        a2 = _rand_raster(rng, 1, 32, 32, np.float32, 0.0)
        holes = rng.random((32, 32)) < 0.25
        a2[:, holes] = -9999.0
        t2 = (gsd, 0.0, 412000.0 + dx * gsd, 0.0, -gsd, 5318000.0 - dy * gsd)
        p1, p2 = str(tmp_path / f"m{k}a.tif"), str(tmp_path / f"m{k}b.tif")
        write_geotiff(p1, a1, t1, 25832)
        write_geotiff(p2, a2, t2, 25832, nodata=-9999.0)
        got, gt = merge_images(GeoTiff(p1), GeoTiff(p2))
        ref, rt = merge_images_ref(a1, t1, a2, t2, None, -9999.0)
        assert np.array_equal(got, ref) and np.allclose(gt, rt, rtol=0, atol=1e-9)
        assert not np.any(got == -9999.0), "a source's nodata pixels must not be pasted into the mosaic"
        assert np.any(got[:, dy:dy + 32, dx:dx + 32][:, holes] == 0.0)
    # integer raster whose tag cannot be represented: fill 0 + warning, no overflow
    b1 = _rand_raster(rng, 3, 16, 16, np.uint8, 0.0)
    b2 = _rand_raster(rng, 3, 16, 16, np.uint8, 0.0)
    t2 = (gsd, 0.0, 412000.0 + 16 * gsd, 0.0, -gsd, 5318000.0)
    p1, p2 = str(tmp_path / "u8a.tif"), str(tmp_path / "u8b.tif")
    write_geotiff(p1, b1, t1, 25832, nodata=-9999.0)
    write_geotiff(p2, b2, t2, 25832)
    with pytest.warns(UserWarning, match="beyond the valid range"):
        got, _ = merge_images(GeoTiff(p1), GeoTiff(p2))
    ref, _ = merge_images_ref(b1, t1, b2, t2, -9999.0, None)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)
