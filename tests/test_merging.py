"""Seam strips between neighbouring images (reference merging.py:10-118, helpers.py:984-1085)."""
import os
import sys

import numpy as np

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
