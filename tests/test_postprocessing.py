"""Crown post-processing (reference postprocessing.py): host-side filters against the literal numpy oracle; the raster
statistics kernel and the whole stage on the GPU against the brute-force (every crown x every pixel) oracle."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import postprocess_ref as O  # noqa: E402
from treedetection_amd import gpkg  # noqa: E402
from treedetection_amd import postprocessing as P  # noqa: E402
from treedetection_amd.geotiff import write_geotiff  # noqa: E402


def _boxes(rng, n, span=60.0):
    xy = rng.uniform(0, span, (n, 2))
    wh = rng.uniform(2, 14, (n, 2))
    return np.concatenate([xy, xy + wh], axis=1)


def test_box_filters_follow_the_reference_dtypes():
    rng = np.random.default_rng(0)
    for n in (1, 7, 40):
        b = _boxes(rng, n) + np.array([412000.0, 5318000.0, 412000.0, 5318000.0])
        areas = rng.uniform(0.5, 60, n)
        conf = np.round(rng.uniform(0.3, 1.0, n), 3)
        conf[: n // 3] = conf[0]                                   # ties: float16 makes them common
        for iou_t, area_t in ((0.5, 1.0), (0.1, 0.3)):
            assert P.filter_polygons_by_iou_and_area(b, areas, conf, iou_t, area_t) == O.filter_by_iou_and_area(b, areas, conf, iou_t, area_t)
        assert P.containment(b, 0.9) == O.containment(b, 0.9)
    # a pair with IoU > 0.5: the more confident one survives; float32 boxes at UTM magnitude lose the sub-metre offset
    b = np.array([[412000.0, 5318000.0, 412010.0, 5318010.0], [412000.2, 5318000.2, 412010.2, 5318010.2], [412100.0, 5318100.0, 412105.0, 5318105.0]])
    assert P.filter_polygons_by_iou_and_area(b, [100, 100, 25], [0.6, 0.9, 0.5], 0.5, 1) == [1, 2]
    ratio, is_c, num = P.containment([[0, 0, 10, 10], [2, 2, 4, 4], [20, 20, 30, 30]], 0.9)
    assert is_c == [False, True, False] and num == [1, 0, 0] and ratio[1] == 1.0
    assert P.filter_polygons_by_iou_and_area([], [], [], 0.5, 1) == []


def test_ndvi_and_helpers():
    rgbi = np.zeros((4, 2, 3), np.uint8)
    rgbi[0], rgbi[3] = 50, 150
    assert np.allclose(P.ndvi_from_rgbi(rgbi), (150 - 50) / (150 + 50 + 255e-10)) and (P.ndvi_from_rgbi(rgbi) == O.ndvi_from_rgbi(rgbi)).all()
    with pytest.raises(ValueError, match="near-infrared"):
        P.ndvi_from_rgbi(rgbi[:3])
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0)
    assert P._window(t, 500, 400, (412000.0, 5318000.0, 412080.0, 5318100.0)) == (0, 0, 499, 400)      # the raster's own bounds (the upper column is clipped by the slice, as in numpy)
    ring = np.array([[412000.1234, 5318000.0005], [412001.0, 5318000.0], [412001.0, 5318001.0015], [412000.1234, 5318000.0005]])
    assert P._round_ring(ring).tolist()[0] == [round(412000.1234 * 1000) / 1000, round(5318000.0005 * 1000) / 1000]
    c = P.crown_circles([ring])
    assert c.dtype == np.float32 and c.shape == (1, 3) and c[0, 2] > 0


def test_gdal_style_bilinear_decimation():
    """resample_bilinear_gdal: GDAL's triangle-filter decimation (read(out_shape=..., resampling=bilinear))."""
    const = np.full((2, 50, 35), 7, np.uint8)
    assert (P.resample_bilinear_gdal(const, 10, 7) == 7).all() and P.resample_bilinear_gdal(const, 50, 35) is const
    ramp = np.tile(np.arange(50, dtype=np.float32), (1, 20, 1))
    out = P.resample_bilinear_gdal(ramp, 4, 10)
    assert out.shape == (1, 4, 10) and np.allclose(out[0, :, 1:-1], [7, 12, 17, 22, 27, 32, 37, 42], atol=1e-4)   # block centres
    assert 2.0 < out[0, 0, 0] < 2.6 and 46.4 < out[0, 0, -1] < 47.0          # truncated kernels at the borders
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (4, 40, 60), dtype=np.uint8)
    got = P.resample_bilinear_gdal(img, 8, 12)
    # against a direct evaluation of the same separable triangle filter in float64
    def taps(n_src, n_dst):
        sc = n_dst / n_src
        m = np.zeros((n_dst, n_src))
        for j in range(n_dst):
            c = (j + 0.5) / sc
            for i in range(max(int(np.floor(c - 1 / sc + 0.5)), 0), min(int(c + 1 / sc + 0.5), n_src)):
                m[j, i] = max(0.0, 1 - abs(sc * (i - c + 0.5)))
            m[j] /= m[j].sum()
        return m
    want = np.einsum("ih,bhw,jw->bij", taps(40, 8), img.astype(np.float64), taps(60, 12))
    assert got.dtype == np.uint8 and np.abs(got.astype(np.float64) - np.floor(want + 0.5)).max() <= 1
    assert (got == np.floor(want + 0.5)).mean() > 0.99
    # magnification (scaling factor above 1): the kernel is NOT stretched — bilinear interpolation between pixel centres; a
    # constant stays constant, a linear ramp stays linear away from the borders, the borders replicate, 2x of [a, b] = a, (3a+b)/4, (a+3b)/4, b
    assert (P.resample_bilinear_gdal(const, 100, 70) == 7).all()
    up = P.resample_bilinear_gdal(ramp, 20, 100)
    assert up.shape == (1, 20, 100) and np.allclose(up[0, 3, 1:-1], (np.arange(1, 99) + 0.5) / 2 - 0.5, atol=1e-4)
    assert up[0, 0, 0] == 0.0 and up[0, 0, -1] == 49.0
    two = np.array([[[10.0, 30.0]]], np.float32)
    assert np.allclose(P.resample_bilinear_gdal(two, 1, 4), [[[10.0, 15.0, 25.0, 30.0]]])
    big = P.resample_bilinear_gdal(img, 80, 90)                     # 2x in rows, 1.5x in columns, integers rounded half up
    assert big.shape == (4, 80, 90) and big.dtype == np.uint8
    assert np.array_equal(big[:, 0, 0], img[:, 0, 0]) and np.array_equal(big[:, -1, -1], img[:, -1, -1])
    with pytest.raises(ValueError):
        P.resample_bilinear_gdal(img, 0, 60)


def _scene(tmp_path, rng, same_grid):
    """An RGBI image (0.2 m) and an nDSM (1 m, or the RGBI grid) with blob crowns of known height, plus crowns."""
    H = W = 300
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318060.0)
    rgbi = rng.integers(40, 120, (4, H, W), dtype=np.uint8)
    crowns, heights = [], []
    yy, xx = np.mgrid[0:H, 0:W]
    ndsm_fine = rng.uniform(0, 1.0, (H, W)).astype(np.float32)
    for k in range(14):
        cx, cy, r = rng.uniform(30, 270), rng.uniform(30, 270), rng.uniform(8, 25)
        inside = (xx - cx) ** 2 + (yy - cy) ** 2 < r ** 2
        rgbi[3][inside] = 220 if k % 4 else 60                 # every fourth "crown" is not green
        hgt = float(rng.uniform(1.0, 25.0))
        ndsm_fine[inside] = np.maximum(ndsm_fine[inside], hgt * (1 - ((xx - cx) ** 2 + (yy - cy) ** 2)[inside] / r ** 2 * 0.5))
        ang = np.linspace(0, 2 * np.pi, 24, endpoint=False)
        ring = np.stack([t[2] + t[0] * (cx + r * np.cos(ang)), t[5] + t[4] * (cy + r * np.sin(ang))], axis=1)
        crowns.append(np.concatenate([ring, ring[:1]]))
        heights.append(hgt)
    write_geotiff(str(tmp_path / "img" / "3241.tif"), rgbi, t, 25832)
    if same_grid:
        write_geotiff(str(tmp_path / "ndsm" / "3241.tif"), ndsm_fine[None], t, 25832)
        ndsm, nt = ndsm_fine, t
    else:
        ndsm = ndsm_fine.reshape(60, 5, 60, 5).max(axis=(1, 3))
        nt = (1.0, 0.0, 412000.0, 0.0, -1.0, 5318060.0)
        write_geotiff(str(tmp_path / "ndsm" / "3241.tif"), ndsm[None], nt, 25832)
    return rgbi, t, ndsm, nt, crowns, heights


@pytest.mark.gpu
@pytest.mark.parametrize("same_grid", [False, True])
def test_crown_stats_kernel_equals_brute_force(tmp_path, same_grid):
    os.makedirs(tmp_path / "img")
    os.makedirs(tmp_path / "ndsm")
    rng = np.random.default_rng(5 + same_grid)
    rgbi, t, ndsm, nt, crowns, _ = _scene(tmp_path, rng, same_grid)
    crowns.append(np.array([[413000.0, 5319000.0], [413001.0, 5319000.0], [413001.0, 5319001.0], [413000.0, 5319000.0]]))   # off the rasters
    ndvi = P.ndvi_from_rgbi(rgbi)
    px = [c[:, 0].astype(np.float32) for c in crowns]
    py = [c[:, 1].astype(np.float32) for c in crowns]
    circles = P.crown_circles(crowns)
    ib = (t[2], t[5] + t[4] * rgbi.shape[1], t[2] + t[0] * rgbi.shape[2], t[5])
    hb = (nt[2], nt[5] + nt[4] * ndsm.shape[0], nt[2] + nt[0] * ndsm.shape[1], nt[5])
    want_h, want_xy = O.heights_within(px, py, ndsm, nt + (0, 0, 1), hb)
    got = P.crown_stats(ndsm, nt, hb, circles, 0)
    assert np.array_equal(got[:, 0], want_h) and np.array_equal(got[:, 1:], want_xy)
    assert got[-1].tolist() == [-1, -1, -1] and (got[:-1, 0] > 0).all()
    for scale in (1.0, 0.5):
        w = O.ndvi_within(px, py, ndvi, t + (0, 0, 1), ib, scale)
        g = P.crown_stats(ndvi.astype(np.float32), t, ib, circles, 1, scale)
        assert np.array_equal(g[:, 0], w[0]) and np.array_equal(g[:, 1], w[1])                 # min / max: exact
        assert np.allclose(g[:, 2], w[2], rtol=0, atol=1e-7) and np.allclose(g[:, 3], w[3], rtol=0, atol=1e-7)
        assert g[-1].tolist() == [-1, -1, -1, -1]
    # a sub-window (bounds smaller than the raster): the reference's offset swap is part of the contract
    sub_b = (ib[0] + 10.0, ib[1] + 4.0, ib[2] - 6.0, ib[3] - 12.0)
    w = O.ndvi_within(px, py, ndvi, t + (0, 0, 1), sub_b, 1.0)
    g = P.crown_stats(ndvi.astype(np.float32), t, sub_b, circles, 1, 1.0)
    assert np.array_equal(g[:, :2].T, np.stack(w[:2])) and np.allclose(g[:, 2], w[2], atol=1e-7)


@pytest.mark.gpu
def test_postprocess_stage_end_to_end(tmp_path):
    """stitched layer + rasters → processed_*.gpkg: thresholds, de-duplication, attributes, resume file."""
    os.makedirs(tmp_path / "img")
    os.makedirs(tmp_path / "ndsm")
    pred = tmp_path / "out" / "geojson_predictions"
    os.makedirs(pred)
    rng = np.random.default_rng(9)
    rgbi, t, ndsm, nt, crowns, heights = _scene(tmp_path, rng, same_grid=False)
    scores = [0.95 - 0.03 * k for k in range(len(crowns))]
    crowns.append(crowns[0] + 0.05)                               # near-duplicate of crown 0, lower confidence → removed
    scores.append(0.4)
    scores[3] = 0.1                                               # under the confidence threshold
    gpkg.write_polygons(str(pred / "3241.gpkg"), crowns, {"Confidence_score": scores, "filter_index_right": [0] * len(crowns)}, 25832)

    class Log:
        def __getattr__(self, name):
            return lambda m: None

    config = {"logger": Log(), "output_directory": str(tmp_path / "out"), "height_data_path": str(tmp_path / "ndsm"),
              "image_directory": str(tmp_path / "img"), "confidence_threshold": 0.3, "iou_threshold": 0.5, "area_threshold": 1,
              "containment_threshold": 0.9, "height_threshold": 3, "ndvi_mean_threshold": 0.2, "ndvi_var_threshold": 0.5,
              "use_overlap": False, "tile_width": 50, "tile_height": 50, "buffer": 10, "overlapping_tiles_width": 3,
              "overlapping_tiles_height": 3, "device": "0", "parallel": False, "exclude_files": [],
              "confidence_threshold_stitching": 0.3, "timestamped_output_directory": False,
              "ndvi_scaling_factor": 1.0, "height_scaling_factor": 1.0}
    import treedetection_amd as T
    T.postprocess_files(config)
    rings, cols, srs = gpkg.read_polygons(str(pred / "processed_3241.gpkg"))
    assert srs == 25832 and set(cols) == set(P.COLUMNS) and os.path.exists(tmp_path / "out" / "3241.gpkg")
    kept_scores = cols["Confidence_score"]
    assert 0.4 not in kept_scores and all(s >= 0.3 for s in kept_scores)            # duplicate + low confidence gone
    assert all(h >= 3 or h == -1 for h in cols["TreeHeight"])                        # height threshold
    by_score = {round(s, 6): k for k, s in enumerate(scores)}
    for s, h, area, dia, cen in zip(kept_scores, cols["TreeHeight"], cols["Area"], cols["Diameter"], cols["Centroid"]):
        k = by_score[round(s, 6)]
        assert h >= 0.45 * heights[k]                  # at least the crown's own blob (a taller neighbour may reach into the circle)
        assert dia == pytest.approx(2 * (area / np.pi) ** 0.5) and set(json.loads(cen)) == {"x", "y"}
    # NDVI thresholds, checked with the brute-force oracle on the same crowns (separate grids → full radius)
    ib = (t[2], t[5] + t[4] * rgbi.shape[1], t[2] + t[0] * rgbi.shape[2], t[5])
    px = [c[:, 0].astype(np.float32) for c in crowns]
    py = [c[:, 1].astype(np.float32) for c in crowns]
    _, _, o_mean, o_var = O.ndvi_within(px, py, P.ndvi_from_rgbi(rgbi), t + (0, 0, 1), ib, 1.0)
    kept_k = {by_score[round(s, 6)] for s in kept_scores}
    assert all(o_mean[k] >= 0.2 and o_var[k] <= 0.5 for k in kept_k)
    assert any(o_mean[k] < 0.2 for k in range(len(crowns)) if scores[k] >= 0.3)     # the threshold did remove something
    assert all(abs(c * 1000 - round(c * 1000)) < 1e-6 for r in rings for c in r.ravel())   # coordinates rounded to mm
    import yaml
    rec = yaml.safe_load(open(pred / "recovery.yaml"))
    assert rec["processed_files"] == [str(pred / "3241.gpkg")] and rec["parameters"]["height_threshold"] == 3
    mtime = os.path.getmtime(pred / "processed_3241.gpkg")
    T.postprocess_files(config)                                                      # resume: nothing is redone
    assert os.path.getmtime(pred / "processed_3241.gpkg") == mtime


@pytest.mark.gpu
def test_containment_rules_as_in_the_reference(tmp_path):
    """process_features' selection (postprocessing.py:627-669), quirks included: a crown containing >= 3 others is
    dropped, one containing exactly two is dropped as well (that branch never appends), contained crowns themselves stay
    and carry is_contained = True."""
    os.makedirs(tmp_path / "img")
    os.makedirs(tmp_path / "ndsm")
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318080.0)
    rng = np.random.default_rng(1)
    rgbi = rng.integers(100, 110, (4, 400, 400), dtype=np.uint8)
    rgbi[3] = 200
    write_geotiff(str(tmp_path / "img" / "7.tif"), rgbi, t, 25832)
    write_geotiff(str(tmp_path / "ndsm" / "7.tif"), np.full((1, 80, 80), 12.0, np.float32), (1.0, 0, 412000.0, 0, -1.0, 5318080.0), 25832)

    def box(x0, y0, s):
        x0, y0 = 412000.0 + x0, 5318000.0 + y0
        return np.array([[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s], [x0, y0]])

    crowns = {"two_outer": box(5, 5, 20), "two_a": box(7, 7, 4), "two_b": box(15, 15, 4),
              "three_outer": box(40, 5, 24), "three_a": box(42, 7, 4), "three_b": box(50, 7, 4), "three_c": box(42, 20, 4),
              "alone": box(10, 45, 6)}
    names = list(crowns)
    scores = [0.9 - 0.01 * i for i in range(len(names))]

    class Log:
        def __getattr__(self, name):
            return lambda m: None

    config = {"logger": Log(), "confidence_threshold": 0.3, "iou_threshold": 0.5, "area_threshold": 1, "containment_threshold": 0.9,
              "height_threshold": 3, "use_overlap": False, "device": "0"}
    out = P.process_layer([crowns[n] for n in names], scores, config, str(tmp_path / "ndsm" / "7.tif"), str(tmp_path / "img" / "7.tif"))
    got = {names[[round(s, 6) for s in scores].index(round(f["properties"]["Confidence_score"], 6))]: f["properties"] for f in out}
    assert set(got) == {"two_a", "two_b", "three_a", "three_b", "three_c", "alone"}
    assert got["two_a"]["is_contained"] == "True" and got["alone"]["is_contained"] == "False" and got["alone"]["num_contained"] == 0
    assert all(p["TreeHeight"] == 12.0 for p in got.values()) and got["alone"]["Area"] == pytest.approx(36.0)
    # the config above has no scaling keys: the NDVI raster was decimated by the documented default 0.2 (400 → 80 px);
    # a factor above 1 magnifies it (GDAL's bilinear interpolation): the same crown survives with the same attributes
    up = P.process_layer([crowns["alone"]], [0.9], dict(config, ndvi_scaling_factor=2.0, height_scaling_factor=2.0), str(tmp_path / "ndsm" / "7.tif"),
                         str(tmp_path / "img" / "7.tif"))
    assert len(up) == 1 and up[0]["properties"]["TreeHeight"] == 12.0 and up[0]["properties"]["Area"] == pytest.approx(36.0)
