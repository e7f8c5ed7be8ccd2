"""Host-side mirror of the reference's API (config / tiling / recovery / exclusion / polygon epilogue) — CPU only."""
import json
import logging
import os
import re

import numpy as np
import pytest
import yaml

from treedetection_amd import recoveries
from treedetection_amd.config import get_config, setup_model_cfg
from treedetection_amd.geotiff import GeoTiff, write_geotiff
from treedetection_amd.prediction import Predictor, polygons_from_masks
from treedetection_amd.preprocessing import tile_single_file

HERE = os.path.dirname(os.path.abspath(__file__))


def test_geotiff_roundtrip_and_windows(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (4, 300, 500), dtype=np.uint8)
    p = str(tmp_path / "a.tif")
    write_geotiff(p, img, (0.2, 0, 412000.0, 0, -0.2, 5318100.0), 25832)
    g = GeoTiff(p)
    assert (g.width, g.height, g.count, g.epsg) == (500, 300, 4, 25832)
    assert g.bounds == (412000.0, 5318040.0, 412100.0, 5318100.0)
    assert np.array_equal(g.read(), img)
    # bbox on pixel boundaries: rasterio.mask(crop=True) returns exactly the covered window
    w = g.read_bounds([412010, 5318060, 412030, 5318090])
    assert w.shape == (4, 150, 100)
    assert np.array_equal(w, img[:, 50:200, 50:150])
    # bbox hanging over the raster edge is clipped (reference: 350 px edge tiles)
    w = g.read_bounds([411990, 5318080, 412010, 5318120])
    assert w.shape == (4, 100, 50) and np.array_equal(w, img[:, :100, :50])
    with pytest.raises(ValueError):
        g.read_bounds([0, 0, 10, 10])
    f = rng.standard_normal((1, 20, 30)).astype(np.float32)
    write_geotiff(str(tmp_path / "f.tif"), f, (1, 0, 0, 0, -1, 20), 25832)
    assert np.array_equal(GeoTiff(str(tmp_path / "f.tif")).read(), f)


def test_tile_metadata_matches_reference_format(tmp_path):
    img = np.zeros((4, 500, 500), dtype=np.uint8)     # 100 m x 100 m at 0.2 m
    p = str(tmp_path / "324125317.tif")
    write_geotiff(p, img, (0.2, 0, 412000.0, 0, -0.2, 5318100.0), 25832)
    tile_single_file(p, str(tmp_path / "tiles"), buffer=20, tile_width=50, tile_height=50)
    meta = json.load(open(tmp_path / "tiles" / "324125317.json"))
    assert len(meta) == 4
    # id scheme preprocessing.py:59; grid origin = raster bounds (left, bottom)
    assert list(meta)[0] == "324125317_412000_5318000_50_20_25832"
    t = meta["324125317_412050_5318050_50_20_25832"]
    assert t["bounds"] == [412030.0, 5318030.0, 412120.0, 5318120.0]
    assert t["crs"] == 25832 and t["only_forest"] is False and t["only_urban"] is False
    # window transform of the clipped window (col 150, row 0): 9-number affine like json.dumps(Affine)
    assert len(t["transform"]) == 9
    assert t["transform"][:6] == pytest.approx([0.2, 0.0, 412030.0, 0.0, -0.2, 5318100.0])
    t0 = meta["324125317_412000_5318000_50_20_25832"]
    assert t0["transform"][:6] == pytest.approx([0.2, 0.0, 412000.0, 0.0, -0.2, 5318070.0])


def test_forest_flags(tmp_path):
    img = np.zeros((3, 500, 500), dtype=np.uint8)
    p = str(tmp_path / "x.tif")
    write_geotiff(p, img, (0.2, 0, 0.0, 0, -0.2, 100.0), 25832)
    outline = {"type": "FeatureCollection", "features": [{"type": "Feature", "properties": {}, "geometry": {
        "type": "Polygon", "coordinates": [[[-100, -100], [49, -100], [49, 300], [-100, 300], [-100, -100]]]}}]}
    op = str(tmp_path / "forest.geojson")
    json.dump(outline, open(op, "w"))
    from treedetection_amd.preprocessing import _load_outline
    tile_single_file(p, str(tmp_path / "t"), buffer=5, tile_width=25, tile_height=25, forest_polys=_load_outline(op))
    meta = json.load(open(tmp_path / "t" / "x.json"))
    assert meta["x_0_0_25_5_25832"]["only_forest"] and not meta["x_0_0_25_5_25832"]["only_urban"]
    assert meta["x_75_75_25_5_25832"]["only_urban"] and not meta["x_75_75_25_5_25832"]["only_forest"]
    assert not meta["x_25_0_25_5_25832"]["only_forest"] and not meta["x_25_0_25_5_25832"]["only_urban"]   # straddles


def test_recovery_files_match_reference_fixture(tmp_path):
    """tests/golden/recovery_fixture.json was produced by the reference's own recoveries.py (generator committed)."""
    fx = json.load(open(os.path.join(HERE, "golden", "recovery_fixture.json")))
    os.chdir(tmp_path)
    os.makedirs("tiles")
    os.makedirs("out/a")
    os.makedirs("out/b")
    for stem, m in fx["metas"].items():
        json.dump(m, open(f"tiles/{stem}.json", "w"))
    for k in fx["metas"]["a"]:
        open(f"out/a/Prediction_{k}.json", "w").write("[]")
    open("out/b/Prediction_b_0_0_50_20_25832.json", "w").write("[]")
    recoveries.save_prediction_recovery_data("out", "tiles", "model.pth", {"img/b.tif"}, ["img/a.tif"])
    assert yaml.safe_load(open("out/prediction_recovery.yaml")) == yaml.safe_load(fx["yaml"])
    assert open("out/prediction_recovery.yaml").read() == fx["yaml"]          # byte-compatible
    log = logging.getLogger("t")
    for case in fx["cases"]:
        if "removed" in case:
            os.remove(f"out/a/Prediction_{case['removed']}.json")
        fl, done = recoveries.load_prediction_recovery_data("out", "tiles", case["model"], log, case["exclude"])
        assert fl == case["file_list"] and sorted(done) == case["processed"], case


def test_get_config_defaults_and_asserts(tmp_path):
    (tmp_path / "rgb").mkdir()
    (tmp_path / "ndsm").mkdir()
    (tmp_path / "m.npz").write_bytes(b"x")
    cfg_path = tmp_path / "config.yml"
    cfg_path.write_text(yaml.safe_dump({"image_directory": str(tmp_path / "rgb"), "height_data_path": str(tmp_path / "ndsm"),
                                        "combined_model": str(tmp_path / "m.npz"),
                                        "output_directory": str(tmp_path / "out"), "tile_width": 40}))
    config, obj = get_config(str(cfg_path))
    assert config["tile_width"] == 40 and config["tile_height"] == 50 and config["buffer"] == 20 and config["batch_size"] == 10
    assert config["merged_path"] == "merged" and config["simplify_tolerance"] == 0.2 and config["iou_threshold"] == 0.5
    assert config["continue"] == os.path.join(str(tmp_path / "out"), "continue.yml")
    assert config["device"] in ("cpu", "0") and obj.tile_width == 40
    bad = tmp_path / "bad.yml"
    bad.write_text(yaml.safe_dump({"image_directory": str(tmp_path / "rgb"), "height_data_path": str(tmp_path / "ndsm")}))
    with pytest.raises(AssertionError):
        get_config(str(bad))


def test_setup_model_cfg_mirrors_reference_values():
    cfg = setup_model_cfg(update_model="x.pth", device="cpu")
    assert cfg.MODEL.WEIGHTS == "x.pth" and cfg.MODEL.RESNETS.DEPTH == 101 and cfg.MODEL.DEVICE == "cpu"
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 1 and cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST == 0.3
    assert cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST == 0.5 and cfg.TEST.DETECTIONS_PER_IMAGE == 100
    assert setup_model_cfg("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml").MODEL.RESNETS.DEPTH == 50


def test_exclusion_and_tile_order(tmp_path):
    meta = {f"t{i}": {"bounds": [0, 0, 1, 1], "transform": [1, 0, 0, 0, -1, 0, 0, 0, 1], "crs": 1,
                      "only_forest": i % 2 == 0, "only_urban": i == 3} for i in range(5)}
    p = tmp_path / "img.json"
    p.write_text(json.dumps(meta))
    pr = Predictor.__new__(Predictor)        # host logic only: no engine, no GPU
    pr.exclude_vars = ["only_forest"]
    tiles = pr._load_tiles(str(p))
    assert [t["tile_id"] for t in tiles] == ["t1", "t3"]          # JSON key order, flagged tiles dropped
    assert "only_forest" not in tiles[0]
    pr.exclude_vars = []
    assert [t["tile_id"] for t in pr._load_tiles(str(p))] == [f"t{i}" for i in range(5)]


def test_predictor_refuses_cpu():
    with pytest.raises(RuntimeError):
        Predictor(setup_model_cfg(update_model="x.npz", device="cpu"), device_type="cpu")


def test_polygons_from_masks_schema_and_affine():
    masks = np.zeros((2, 40, 50), bool)
    masks[0, 10:20, 5:15] = True
    masks[1, 0:2, 0:1] = True         # 2 points only: dropped (contour.size < 8)
    regions = np.array([[4, 9, 16, 21], [0, 0, 3, 4]], np.int32)
    t = [0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0, 0, 0, 1]
    ev = polygons_from_masks(masks, regions, np.array([0.9, 0.8], np.float32), np.array([0, 0]), t, "a.tif")
    assert len(ev) == 1
    e = ev[0]
    assert set(e) == {"image_id", "category_id", "score", "polygon_coords"} and e["image_id"] == "a.tif"
    assert e["category_id"] == 0 and abs(e["score"] - 0.9) < 1e-6
    ring = e["polygon_coords"][0]
    assert ring[0] == ring[-1] and len(ring) == 5
    # pixel-corner affine (no +0.5): pixel (5,10) → x = 412000 + 0.2*5, y = 5318100 - 0.2*10
    assert ring[0] == pytest.approx([412001.0, 5318098.0])
    assert ring[2] == pytest.approx([412000 + 0.2 * 14, 5318100 - 0.2 * 19])
    json.dumps(ev)


def test_polygons_from_packed_equals_polygons_from_masks():
    """The production epilogue reads the engine's packed bit rows per paste region; it must produce exactly what the
    full-frame path (reference prediction.py:229-261) produces."""
    from treedetection_amd.prediction import polygons_from_packed
    rng = np.random.default_rng(2)
    h, w, n = 60, 90, 5
    masks = np.zeros((n, h, w), bool)
    regions = np.zeros((n, 4), np.int32)
    offsets = np.zeros(n, np.int64)
    words = []
    off = 0
    for d in range(n):
        x0, y0 = int(rng.integers(0, 40)), int(rng.integers(0, 30))
        x1, y1 = x0 + int(rng.integers(5, 45)), y0 + int(rng.integers(5, 28))
        sub = rng.uniform(0, 1, (y1 - y0, x1 - x0)) < 0.6
        masks[d, y0:y1, x0:x1] = sub
        regions[d] = (x0, y0, x1, y1)
        wpr = (x1 - x0 + 31) // 32
        padded = np.zeros((y1 - y0, wpr * 32), np.uint8)
        padded[:, : x1 - x0] = sub
        rows = np.packbits(padded, axis=1, bitorder="little").view(np.uint32)
        offsets[d] = off
        off += rows.size
        words.append(rows.ravel())
    bits = np.concatenate(words).view(np.int32)
    t = [0.2, 0.0, 10.0, 0.0, -0.2, 99.0, 0, 0, 1]
    scores = rng.uniform(0.3, 1, n).astype(np.float32)
    a = polygons_from_masks(masks, regions, scores, np.zeros(n, np.int64), t, "x.tif")
    b = polygons_from_packed(regions, offsets, bits, scores, np.zeros(n, np.int64), t, "x.tif")
    assert a == b and len(a) > n


def _packed_fixture(rng, n, h, w, noise):
    regions = np.zeros((n, 4), np.int32)
    offsets = np.zeros(n, np.int64)
    words, off = [], 0
    for d in range(n):
        x0, y0 = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
        ww, hh = int(rng.integers(1, min(120, w - x0))), int(rng.integers(1, min(120, h - y0)))
        yy, xx = np.mgrid[0:hh, 0:ww]
        sub = ((xx - ww / 2) ** 2 / (ww / 2) ** 2 + (yy - hh / 2) ** 2 / (hh / 2) ** 2) < 1
        sub ^= rng.random((hh, ww)) < noise
        wpr = (ww + 31) // 32
        padded = np.zeros((hh, wpr * 32), np.uint8)
        padded[:, :ww] = sub
        rows = np.packbits(padded, axis=1, bitorder="little").view(np.uint32)
        regions[d] = (x0, y0, x0 + ww, y0 + hh)
        offsets[d] = off
        off += rows.size
        words.append(rows.ravel())
    return regions, offsets, np.concatenate(words).view(np.int32)


@pytest.mark.parametrize("transform,image_id", [
    ((0.2, 0.0, 412000.0, 0.0, -0.2, 5319000.0), '/data/rgb/ö "x"\\a\t\x7f.tif'),      # the usual UTM case + escapes
    ((1e-7, 0.0, 1e-9, 0.0, -3e-8, 0.0), "a.tif"),                                       # small exponents, zeros
    ((1e15, 3.3, 1e22, 0.1, -1e17, -5e16), "b.tif"),                                     # repr switches to e+XX at 1e16
    ((0.30000000000000004, 0.1, -3.7, 0.01, -0.7, 1e16), "\U0001f600.tif"),              # 17-digit values, astral id
])
def test_tile_polygons_json_is_bytewise_json_dumps(transform, image_id):
    """td_tile_polygons_json (the production epilogue: packed masks → contours → affine → text) must write exactly
    the bytes ``json.dumps`` writes for the list the Python restatement of prediction.py:229-261 builds."""
    from treedetection_amd.contours import tile_polygons_json
    from treedetection_amd.prediction import polygons_from_packed
    rng = np.random.default_rng(7)
    n = 24
    regions, offsets, bits = _packed_fixture(rng, n, 300, 300, 0.02)
    regions[3] = (5, 5, 5, 9)                       # empty paste region: skipped
    scores = rng.random(n).astype(np.float32)
    classes = np.zeros(n, np.int32)
    want = json.dumps(polygons_from_packed(regions, offsets, bits, scores, classes, transform, image_id)).encode()
    got = tile_polygons_json(regions, offsets, bits, scores, classes, transform, image_id)
    assert got == want and len(json.loads(got)) > n


def test_tile_polygons_json_empty_and_errors():
    from treedetection_amd import _lib
    from treedetection_amd.contours import tile_polygons_json
    z = np.zeros
    assert tile_polygons_json(z((0, 4), np.int32), z(0, np.int64), z(1, np.int32), z(0, np.float32), z(0, np.int32),
                              (1, 0, 0, 0, -1, 0), "x.tif") == b"[]"
    with pytest.raises(_lib.TdError, match="outside"):       # rows beyond the buffer are refused, not read
        tile_polygons_json(np.array([[0, 0, 64, 64]], np.int32), np.array([10], np.int64), z(16, np.int32),
                           np.array([0.5], np.float32), z(1, np.int32), (1, 0, 0, 0, -1, 0), "x.tif")


def test_geotiff_window_into_staging_buffer(tmp_path):
    """The tile reader copies windows straight into (pinned) staging memory: same pixels as the allocating read."""
    from treedetection_amd.geotiff import GeoTiff, write_geotiff
    rng = np.random.default_rng(0)
    data = rng.integers(0, 255, (4, 70, 90), dtype=np.uint8)
    write_geotiff(str(tmp_path / "r.tif"), data, (0.5, 0, 100.0, 0, -0.5, 200.0), 25832)
    g = GeoTiff(str(tmp_path / "r.tif"))
    stage = np.full(5 + 40 * 30 * 4, 7, np.uint8)
    for bounds in [(105.0, 180.0, 120.0, 200.0), (90.0, 150.0, 110.2, 170.1)]:
        want = g.read_bounds_hwc(bounds)
        got = g.read_bounds_hwc(bounds, out=stage, out_off=5)
        assert got.base is not None and np.shares_memory(got, stage)
        assert got.shape == want.shape and (got == want).all() and (stage[:5] == 7).all()


def test_pth_checkpoint_round_trip(tmp_path):
    """The reference's only weight format: ``torch.save({"model": state_dict, "optimizer": ..., ...})``
    (detectron2 DetectionCheckpointer; TreeDetection/config.py:39, README.md:14). Extra non-weight buffers
    (pixel_mean / pixel_std, anchor cell_anchors), integer entries and half-precision tensors must not disturb it."""
    import torch
    from treedetection_amd.weights import infer_depth, load_checkpoint, make_synthetic_state_dict
    sd = make_synthetic_state_dict(50, seed=5, width_div=4)
    model = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    model["pixel_mean"] = torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    model["pixel_std"] = torch.ones(3, 1, 1)
    for i in range(5):
        model[f"proposal_generator.anchor_generator.cell_anchors.{i}"] = torch.zeros(3, 4)
    model["backbone.bottom_up.stem.conv1.norm.num_batches_tracked"] = torch.tensor(7)       # integer: dropped
    k16 = "roi_heads.box_head.fc2.bias"
    model[k16] = model[k16].half()                                                           # comes back as float32
    ckpt = {"model": model, "optimizer": {"state": {0: {"momentum_buffer": torch.zeros(3)}}, "param_groups": [{"lr": 0.01}]},
            "scheduler": {"last_epoch": 10}, "iteration": 1234}
    path = str(tmp_path / "model_combined.pth")
    torch.save(ckpt, path)
    got = load_checkpoint(path)
    assert infer_depth(got) == 50
    for k, v in sd.items():
        assert got[k].dtype == np.float32 and got[k].shape == v.shape, k
        if k == k16:
            assert np.array_equal(got[k], v.astype(np.float16).astype(np.float32))
        else:
            assert np.array_equal(got[k], v), k
    assert "pixel_mean" in got and "backbone.bottom_up.stem.conv1.norm.num_batches_tracked" not in got
    # a flat state dict (no "model" wrapper) loads the same way
    torch.save(model, str(tmp_path / "flat.pth"))
    flat = load_checkpoint(str(tmp_path / "flat.pth"))
    assert sorted(flat) == sorted(got)
    # a checkpoint that needs full unpickling (numpy scalar in the trainer state) still loads
    ckpt["best"] = np.float64(0.5)
    torch.save(ckpt, str(tmp_path / "np.pth"))
    assert sorted(load_checkpoint(str(tmp_path / "np.pth"))) == sorted(got)
    with pytest.raises(ValueError):
        torch.save([1, 2, 3], str(tmp_path / "bad.pth"))
        load_checkpoint(str(tmp_path / "bad.pth"))


def test_reference_ndsm_raster_reads_as_surveyed():
    """The one raster the reference ships (data/nDSM/324125317.tif): 1000 x 1000 float32, 1 m, EPSG:25832, origin
    (412000, 5318000) .. values 0 .. 50.9 m (SURVEY.md probe table). Skipped where /root/reference is absent."""
    path = "/root/reference/data/nDSM/324125317.tif"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this box")
    from treedetection_amd.geotiff import GeoTiff
    g = GeoTiff(path)
    try:
        assert (g.width, g.height, g.count) == (1000, 1000, 1) and g.dtype == np.float32 and g.epsg == 25832
        left, bottom, right, top = g.bounds
        assert (left, bottom, right, top) == (412000.0, 5317000.0, 413000.0, 5318000.0)      # origin = upper-left corner
        a = g.read_bounds([left, bottom, right, top])
        assert a.shape[-2:] == (1000, 1000)
        assert a.min() == 0.0 and abs(float(a.max()) - 50.9) < 0.05 and abs(float(a.mean()) - 5.7) < 0.1
        # a 50 m tile with 20 m buffer at the raster's corner is clipped to the extent, as rasterio.mask(crop=True) does
        w = g.read_bounds([left - 20, top - 70, left + 70, top + 20])
        assert w.shape[-2:] == (70, 70) and np.array_equal(w.reshape(70, 70), a.reshape(1000, 1000)[:70, :70])
    finally:
        g.close()
