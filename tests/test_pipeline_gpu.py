"""BASELINE config #1 (plumbing): small GeoTIFFs → preprocess_files → predict_tiles on the GPU → per-tile
Prediction_*.json, checked against the oracle run on the same tiles (resize, forward, paste, contours, affine)."""
import json
import os

import numpy as np
import pytest
import yaml

from oracle import ops_ref as R
from oracle.contours_ref import find_contours as ref_contours
from oracle.maskrcnn_ref import MaskRCNNOracle
from treedetection_amd.geotiff import GeoTiff, write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    import treedetection_amd as T
    root = tmp_path_factory.mktemp("cfg1")
    (root / "rgb").mkdir()
    (root / "ndsm").mkdir()
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    np.savez(root / "model_combined.npz", **sd)
    for k, name in enumerate(("324125317", "324125318")):
        rgb, ndsm = make_tile(100 + k, 600)
        rgbi = np.concatenate([rgb, rgb[..., :1]], axis=2).transpose(2, 0, 1)      # 4-band RGBI like the reference data
        t = (0.2, 0.0, 412000.0 + 120 * k, 0.0, -0.2, 5318120.0)
        write_geotiff(str(root / "rgb" / f"{name}.tif"), np.ascontiguousarray(rgbi), t, 25832)
        write_geotiff(str(root / "ndsm" / f"{name}.tif"), ndsm[::5, ::5].copy(), (1.0, 0, t[2], 0, -1.0, t[5]), 25832)
    cfg = {"image_directory": str(root / "rgb"), "height_data_path": str(root / "ndsm"),
           "combined_model": str(root / "model_combined.npz"), "output_directory": str(root / "output"),
           "tiles_path": str(root / "tiles"), "use_overlap": False, "tile_width": 50, "tile_height": 50, "buffer": 20,
           "batch_size": 4, "parallel": False, "num_workers": 2, "keep_intermediate": True, "device": "0"}
    (root / "config.yml").write_text(yaml.safe_dump(cfg))
    config, _ = T.get_config(str(root / "config.yml"))
    T.preprocess_files(config)
    T.predict_tiles(config)
    return root, config, sd


def test_outputs_exist_and_resume(workspace):
    root, config, _ = workspace
    for name in ("324125317", "324125318"):
        meta = json.load(open(root / "tiles" / f"{name}.json"))
        assert len(meta) == 9      # 120 m / 50 m → 3 x 3 tiles
        files = sorted(os.listdir(root / "output" / "predictions" / name))
        assert files == sorted(f"Prediction_{k}.json" for k in meta)
    rec = yaml.safe_load(open(root / "output" / "predictions" / "prediction_recovery.yaml"))
    assert rec["model_path"] == config["combined_model"] and len(rec["files"]) == 2
    assert os.path.exists(root / "output" / "geojson_predictions" / "324125317.gpkg")
    # second call: everything is recovered, nothing re-predicted
    import treedetection_amd as T
    before = os.path.getmtime(root / "output" / "predictions" / "324125317" / sorted(os.listdir(root / "output" / "predictions" / "324125317"))[0])
    T.predict_tiles(config)
    after = os.path.getmtime(root / "output" / "predictions" / "324125317" / sorted(os.listdir(root / "output" / "predictions" / "324125317"))[0])
    assert before == after


def test_prediction_json_matches_oracle_pipeline(workspace):
    root, config, sd = workspace
    name = "324125317"
    meta = json.load(open(root / "tiles" / f"{name}.json"))
    tif = str(root / "rgb" / f"{name}.tif")
    img = GeoTiff(tif)
    oracle = MaskRCNNOracle(sd)
    checked = 0
    for tile_id in list(meta)[:3] + list(meta)[-1:]:
        td = meta[tile_id]
        bands = img.read_bounds(td["bounds"])
        x, h, w = R.preprocess_tile_u8(bands)
        ref = oracle.forward([{"image": x, "height": h, "width": w}])[0]
        got = json.load(open(root / "output" / "predictions" / name / f"Prediction_{tile_id}.json"))
        exp = []
        for d in range(len(ref["scores"])):
            for c in ref_contours(ref["pred_masks"][d]):
                if c.size < 8:
                    continue
                cx, cy = c[:, 0].tolist(), c[:, 1].tolist()
                if (cx[0], cy[0]) != (cx[-1], cy[-1]):
                    cx.append(cx[0]); cy.append(cy[0])
                t = td["transform"]
                exp.append((float(ref["scores"][d]), [[t[0] * a + t[1] * b + t[2], t[3] * a + t[4] * b + t[5]] for a, b in zip(cx, cy)]))
        assert abs(len(got) - len(exp)) <= 1, (tile_id, len(got), len(exp))
        # entries whose score matches must carry near-identical rings (mask pixels may flip at 0.5 → allow few)
        same = 0
        for e in got:
            assert e["image_id"] == tif and e["category_id"] == 0
            for s, ring in exp:
                if abs(e["score"] - s) <= 1e-4 and len(ring) == len(e["polygon_coords"][0]):
                    if np.abs(np.asarray(ring) - np.asarray(e["polygon_coords"][0])).max() <= 0.2 + 1e-9:   # <= 1 px
                        same += 1
                        break
        assert same >= 0.9 * len(exp), (tile_id, same, len(exp))
        checked += len(exp)
    assert checked > 5


def test_two_model_flow_with_exclude_flags(tmp_path):
    """BASELINE config #3 shape: urban + forest models, forest outline → only_forest / only_urban flags; each model
    skips the tiles flagged for the other (reference detection.py:154-164, prediction.py:79-93)."""
    import treedetection_amd as T
    root = tmp_path
    (root / "rgb").mkdir()
    (root / "ndsm").mkdir()
    for seed, name in ((3, "urban"), (4, "forest")):
        np.savez(root / f"model_{name}.npz", **make_synthetic_state_dict(50, seed=seed, width_div=2))
    rgb, ndsm = make_tile(7, 500)
    t = (0.2, 0.0, 0.0, 0.0, -0.2, 100.0)       # 100 m x 100 m
    write_geotiff(str(root / "rgb" / "1.tif"), np.ascontiguousarray(rgb.transpose(2, 0, 1)), t, 25832)
    write_geotiff(str(root / "ndsm" / "1.tif"), ndsm[::5, ::5].copy(), (1.0, 0, 0, 0, -1.0, 100.0), 25832)
    outline = {"type": "FeatureCollection", "features": [{"type": "Feature", "properties": {}, "geometry": {
        "type": "Polygon", "coordinates": [[[-50, -50], [49, -50], [49, 150], [-50, 150], [-50, -50]]]}}]}
    (root / "forest.geojson").write_text(json.dumps(outline))
    cfg = {"image_directory": str(root / "rgb"), "height_data_path": str(root / "ndsm"),
           "urban_model": str(root / "model_urban.npz"), "forrest_model": str(root / "model_forest.npz"),
           "forrest_outline": str(root / "forest.geojson"), "output_directory": str(root / "output"),
           "tiles_path": str(root / "tiles"), "use_overlap": False, "tile_width": 25, "tile_height": 25, "buffer": 5,
           "batch_size": 3, "parallel": False, "num_workers": 2, "keep_intermediate": True, "device": "0"}
    (root / "config.yml").write_text(yaml.safe_dump(cfg))
    config, _ = T.get_config(str(root / "config.yml"))
    T.preprocess_files(config)
    meta = json.load(open(root / "tiles" / "1.json"))
    assert len(meta) == 16
    only_forest = {k for k, v in meta.items() if v["only_forest"]}
    only_urban = {k for k, v in meta.items() if v["only_urban"]}
    assert only_forest and only_urban and not (only_forest & only_urban)
    T.predict_tiles(config)
    urban = {f[len("Prediction_"):-5] for f in os.listdir(root / "output" / "urban_predictions" / "1")}
    forest = {f[len("Prediction_"):-5] for f in os.listdir(root / "output" / "forrest_predictions" / "1")}
    assert urban == set(meta) - only_forest          # urban model skips forest-only tiles
    assert forest == set(meta) - only_urban          # forest model skips urban-only tiles
    assert os.path.exists(root / "output" / "urban_geojson" / "1.gpkg")
    assert os.path.exists(root / "output" / "forrest_geojson" / "1.gpkg")
    # fusion by the outline (reference helpers.py:703-834): forest-model crowns that intersect the forest + urban-model
    # crowns that are not within it, one layer per image in geojson_predictions/
    from treedetection_amd import gpkg
    from treedetection_amd.vector import Region, read_polygon_layer
    fused, cols, srs = gpkg.read_polygons(str(root / "output" / "geojson_predictions" / "1.gpkg"))
    u_rings = gpkg.read_polygons(str(root / "output" / "urban_geojson" / "1.gpkg"))[0]
    f_rings = gpkg.read_polygons(str(root / "output" / "forrest_geojson" / "1.gpkg"))[0]
    region = Region(read_polygon_layer(str(root / "forest.geojson"))[0])
    n_f = int(region.relate(f_rings)[0].sum()) if f_rings else 0
    n_u = int((~region.relate(u_rings)[1]).sum()) if u_rings else 0
    assert srs == 25832 and len(fused) == (n_f + n_u if (u_rings and f_rings) else len(u_rings) + len(f_rings))
    assert os.path.exists(root / "output" / "geojson_predictions" / "fusion_recovery.yaml")


def test_sixteen_bit_tiles_take_the_float_path(tmp_path):
    """uint16 imagery (max(band 1) > 255): 255*x/65535, then detectron2's float resize (F.interpolate bilinear) —
    reference prediction.py:167-169 — and the engine's float32 CHW input. Compared with the oracle fed the same way."""
    import torch
    import torch.nn.functional as F
    import treedetection_amd as T
    from treedetection_amd.config import setup_model_cfg
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rgb, _ = make_tile(11, 300)
    img16 = (rgb.astype(np.uint16) * 257).transpose(2, 0, 1)
    p = str(tmp_path / "a.tif")
    write_geotiff(p, np.ascontiguousarray(img16), (0.2, 0, 0, 0, -0.2, 60.0), 25832)
    meta = {"a_0_0_60_0_25832": {"crs": 25832, "transform": [0.2, 0, 0, 0, -0.2, 60.0, 0, 0, 1], "bounds": [0, 0, 60, 60],
                                 "only_forest": False, "only_urban": False}}
    (tmp_path / "a.json").write_text(json.dumps(meta))
    cfg = setup_model_cfg(update_model="unused", device="0")
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.05      # the half-width fixture model scores low on this tile
    pred = T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "out"), state_dict=sd)
    got = pred(p, str(tmp_path / "a.json"))
    bgr = np.stack((img16[2], img16[1], img16[0])).astype(np.float64) * 255.0 / 65535.0
    x = F.interpolate(torch.from_numpy(bgr)[None], size=(800, 800), mode="bilinear", align_corners=False)[0].float().numpy()
    from oracle.maskrcnn_ref import Cfg
    ocfg = Cfg()
    ocfg.score_thresh = 0.05
    ref = MaskRCNNOracle(sd, ocfg).forward([{"image": x, "height": 300, "width": 300}])[0]
    n_ref = sum(1 for d in range(len(ref["scores"])) for c in ref_contours(ref["pred_masks"][d]) if c.size >= 8)
    assert len(ref["scores"]) > 3
    assert abs(len(got) - n_ref) <= 1
    ref_scores = sorted({round(float(s), 4) for s in ref["scores"]})
    got_scores = sorted({round(e["score"], 4) for e in got})
    assert len(set(ref_scores) & set(got_scores)) >= len(ref_scores) - 1


def test_process_files_end_to_end_with_overlap(tmp_path):
    """The reference's top-level entry with its default ``use_overlap``: two adjacent RGBI images + nDSM rasters →
    seam strip (merging) → tiles → predictions (plain images and the strip) → stitched layers → post-processed crowns
    copied to the output directory; intermediate folders removed unless ``keep_intermediate``."""
    import treedetection_amd as T
    from treedetection_amd import gpkg
    root = tmp_path
    (root / "rgb").mkdir()
    (root / "ndsm").mkdir()
    np.savez(root / "model.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    for k, name in enumerate(("3241", "3242")):
        rgb, ndsm = make_tile(200 + k, 400)
        rgbi = np.concatenate([rgb, 255 - rgb[..., :1] // 2], axis=2).transpose(2, 0, 1)
        t = (0.2, 0.0, 412000.0 + 80 * k, 0.0, -0.2, 5318080.0)                  # 80 m x 80 m, side by side
        write_geotiff(str(root / "rgb" / f"{name}.tif"), np.ascontiguousarray(rgbi), t, 25832)
        write_geotiff(str(root / "ndsm" / f"{name}.tif"), (ndsm[::5, ::5] + 5).copy(), (1.0, 0, t[2], 0, -1.0, t[5]), 25832)
    cfg = {"image_directory": str(root / "rgb"), "height_data_path": str(root / "ndsm"), "combined_model": str(root / "model.npz"),
           "output_directory": str(root / "output"), "tiles_path": str(root / "tiles"), "use_overlap": True,
           "overlapping_tiles_width": 2, "overlapping_tiles_height": 2, "tile_width": 40, "tile_height": 40, "buffer": 10,
           "batch_size": 4, "parallel": False, "num_workers": 2, "keep_intermediate": True, "device": "0", "height_threshold": 0}
    (root / "config.yml").write_text(yaml.safe_dump(cfg))
    config, _ = T.get_config(str(root / "config.yml"))
    T.process_files(config)
    strip = "3241_412000_5318080_412080_5318080_3241"
    assert os.path.exists(root / "rgb" / "merged" / f"{strip}.tif") and os.path.exists(root / "ndsm" / "merged")
    for name in ("3241", "3242", strip):
        assert os.path.exists(root / "tiles" / f"{name}.json")
        assert os.path.exists(root / "output" / "geojson_predictions" / f"{name}.gpkg")
    for name in ("3241", "3242"):                        # plain images find their rasters; the strip's names do not match
        rings, cols, srs = gpkg.read_polygons(str(root / "output" / f"{name}.gpkg"))       # the merged-file regex default
        assert srs == 25832 and set(cols) >= {"TreeHeight", "Area", "Diameter", "Centroid", "is_contained", "num_contained"}
        raw = gpkg.read_polygons(str(root / "output" / "geojson_predictions" / f"{name}.gpkg"))[0]
        assert len(rings) <= len(raw)
        assert all(h >= 5.0 or h == -1 for h in cols["TreeHeight"])                       # nDSM was offset by +5 m


def test_compressed_tiled_raster_gives_the_same_predictions(tmp_path):
    """The same image stored as one raw strip (memory-mapped windows) and as DEFLATE-compressed 64x64 tiles with the
    horizontal predictor (block decode + cache in the reader thread): byte-identical prediction files, for the plain
    and the pipelined launcher, with the borders followed on the host or on the GPU (``device_contours``)."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rgb, _ = make_tile(300, 500)
    rgbi = np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1))
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0)
    outs = {}
    # pipe: True = three engines in flight on their own HIP streams (the default schedule), "phases" = the phase pipeline,
    # False = one forward at a time
    for tag, kw, pipe, devc in (("raw", {}, True, False), ("deflate", {"tile": (64, 64), "compression": "deflate", "predictor": 2}, True, False),
                                ("deflate_plain", {"tile": (64, 64), "compression": "deflate", "predictor": 2}, False, False),
                                ("raw_phases", {}, "phases", False),
                                ("gpu_contours", {}, True, True), ("gpu_contours_phases", {}, "phases", True),
                                ("gpu_contours_plain", {}, False, True)):
        d = tmp_path / tag
        (d / "rgb").mkdir(parents=True)
        tif = str(d / "rgb" / "9.tif")
        write_geotiff(tif, rgbi, t, 25832, **kw)
        tile_single_file(tif, str(d / "tiles"), buffer=10, tile_width=40, tile_height=40)
        cfg = T.setup_model_cfg(update_model="x", device="0")
        with T.Predictor(cfg, device_type="0", max_batch_size=3, output_dir=str(d / "out"), state_dict=sd, pipeline=bool(pipe),
                         schedule="phases" if pipe == "phases" else "streams", device_contours=devc) as pred:
            res = pred(tif, str(d / "tiles" / "9.json"))
        files = sorted(os.listdir(d / "out" / "9"))
        outs[tag] = {f: open(d / "out" / "9" / f, "rb").read().replace(tif.encode(), b"IMG") for f in files}
        assert len(files) == 9 and sum(len(json.loads(v)) for v in outs[tag].values()) == len(res) > 5
    assert (outs["raw"] == outs["deflate"] == outs["deflate_plain"] == outs["raw_phases"] == outs["gpu_contours"]
            == outs["gpu_contours_phases"] == outs["gpu_contours_plain"])


def test_device_contours_auto_switches_between_batches_with_identical_files(tmp_path):
    """``device_contours: auto`` (round 6): the tracer is switched on while the epilogue workers — not the GPU — set the batch
    period (share of epilogue tasks that find their batch already finished when they start). Here the signal is driven by hand:
    one image with the tracer off, the switch thrown, the same image again — both modes ran, the files are byte-identical, and an
    image whose batches straddle the switch is identical too."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rgb, _ = make_tile(300, 500)
    rgbi = np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1))
    tif = str(tmp_path / "9.tif")
    write_geotiff(tif, rgbi, (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
    meta = str(tmp_path / "tiles" / "9.json")
    cfg = T.setup_model_cfg(update_model="x", device="0")

    def files(d):
        return {f: open(os.path.join(d, "9", f), "rb").read() for f in sorted(os.listdir(os.path.join(d, "9")))}

    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "a"), state_dict=sd, device_contours="auto") as pred:
        assert pred._contours_auto and pred.device_contours and not pred._contours_on
        pred._note_epilogue_start = lambda late: None                 # the signal is driven by hand below
        pred(tif, meta)
        host_only = list(pred.device_contour_batches)
        assert host_only[0] == 5 and host_only[1] == 0                # 9 tiles in batches of 2, all traced on the host
        first = files(str(tmp_path / "a"))
        pred._contours_on = True
        pred(tif, meta)
        assert pred.device_contour_batches == [5, 5]
        assert files(str(tmp_path / "a")) == first
        # the switch thrown in the middle of an image: batch 3 onwards on the device
        calls = {"n": 0}
        real = pred._contour_policy

        def flip():
            calls["n"] += 1
            pred._contours_on = calls["n"] > 2
            return real()
        pred._contour_policy = flip
        pred(tif, meta)
        assert pred.device_contour_batches == [7, 8]
        assert files(str(tmp_path / "a")) == first
    # the signal itself: late tasks switch it on, a long quiet stretch switches it off again
    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "b"), state_dict=sd, device_contours="auto", pipeline=False) as pred:
        for _ in range(31):
            pred._note_epilogue_start(True)
        assert not pred._contours_on
        for _ in range(40):
            pred._note_epilogue_start(True)
        assert pred._contours_on
        for _ in range(2500):
            pred._note_epilogue_start(False)
        assert not pred._contours_on
    with pytest.raises(ValueError):
        T.Predictor(cfg, device_type="0", output_dir=str(tmp_path / "c"), state_dict=sd, device_contours="sometimes")


def test_batch_epilogue_writes_the_same_files_as_a_task_per_tile(tmp_path, monkeypatch):
    """Where only the files are wanted (return_predictions=False: what predict_on_model runs) a batch's host epilogue is ONE task and ONE
    library call (td_batch_prediction_files); with return_predictions=True, or TD_BATCH_EPILOGUE=0, it is a task per tile
    (td_tile_prediction_file). Same files byte for byte — also for a batch that holds an unreadable tile (dropped by the reader)."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rgb, _ = make_tile(300, 500)
    rgbi = np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1))
    tif = str(tmp_path / "9.tif")
    write_geotiff(tif, rgbi, (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
    meta = str(tmp_path / "tiles" / "9.json")
    cfg = T.setup_model_cfg(update_model="x", device="0")
    outs = {}
    for tag, env, ret in (("batch", "1", False), ("per_tile", "0", False), ("with_predictions", "1", True)):
        monkeypatch.setenv("TD_BATCH_EPILOGUE", env)
        with T.Predictor(cfg, device_type="0", max_batch_size=4, output_dir=str(tmp_path / tag), state_dict=sd, return_predictions=ret) as pred:
            assert pred._batch_epilogue == (env == "1")
            res = pred(tif, meta)
            assert (len(res) > 5) == ret
        outs[tag] = {f: open(tmp_path / tag / "9" / f, "rb").read() for f in sorted(os.listdir(tmp_path / tag / "9"))}
        assert len(outs[tag]) == 9
    assert outs["batch"] == outs["per_tile"] == outs["with_predictions"]
    assert sum(len(json.loads(v)) for v in outs["batch"].values()) > 5


def test_predictor_fp16_engine_end_to_end(tmp_path):
    """config ``precision: fp16``: the same Predictor flow on the fp16 engine; per tile about the same crowns as fp32
    (engine-level tolerances: tests/test_engine_fp16_gpu.py)."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3)          # full width: the fp16 kernels need Cin % 64 == 0
    rgb, _ = make_tile(301, 500)
    tif = str(tmp_path / "5.tif")
    write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
    counts = {}
    for prec in ("fp32", "fp16"):
        cfg = T.setup_model_cfg(update_model="x", device="0")
        with T.Predictor(cfg, device_type="0", max_batch_size=4, output_dir=str(tmp_path / prec), state_dict=sd, precision=prec) as pred:
            pred(tif, str(tmp_path / "tiles" / "5.json"))
        counts[prec] = {f: len(json.load(open(tmp_path / prec / "5" / f))) for f in sorted(os.listdir(tmp_path / prec / "5"))}
    assert counts["fp32"].keys() == counts["fp16"].keys() and sum(counts["fp32"].values()) > 5
    a, b = sum(counts["fp32"].values()), sum(counts["fp16"].values())
    assert abs(a - b) <= max(3, 0.15 * a), (a, b)


def test_predictor_loads_pth_checkpoint(tmp_path):
    """The reference's weight format end to end: cfg.MODEL.WEIGHTS = a detectron2-style ``.pth``
    (``torch.save({"model": state_dict, "optimizer": ...})`` with the extra buffers detectron2 keeps) → load_checkpoint →
    Engine; the prediction files equal those of the same weights injected as a dict."""
    import torch
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    model = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    model["pixel_mean"] = torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    model["pixel_std"] = torch.ones(3, 1, 1)
    for i in range(5):
        model[f"proposal_generator.anchor_generator.cell_anchors.{i}"] = torch.zeros(3, 4)
    pth = str(tmp_path / "model_combined.pth")
    torch.save({"model": model, "optimizer": {"state": {}, "param_groups": []}, "scheduler": {}, "iteration": 99}, pth)
    rgb, _ = make_tile(302, 400)
    tif = str(tmp_path / "7.tif")
    write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0.0, 412000.0, 0.0, -0.2, 5318080.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
    outs = {}
    for tag, kw in (("pth", {}), ("dict", {"state_dict": sd})):
        cfg = T.setup_model_cfg(update_model=pth, device="0")
        assert cfg.MODEL.WEIGHTS == pth
        with T.Predictor(cfg, device_type="0", max_batch_size=4, output_dir=str(tmp_path / tag), **kw) as pred:
            pred(tif, str(tmp_path / "tiles" / "7.json"))
        outs[tag] = {f: open(tmp_path / tag / "7" / f, "rb").read() for f in sorted(os.listdir(tmp_path / tag / "7"))}
    assert len(outs["pth"]) == 4 and outs["pth"] == outs["dict"]
    assert sum(len(json.loads(v)) for v in outs["pth"].values()) > 3


def test_chained_images_equal_image_by_image_and_survive_a_bad_image(tmp_path):
    """Predictor.submit starts the next image while the previous one drains (what predict_on_model does in a single process):
    the tile files are byte-identical to calling the predictor image by image; an image that cannot be started (missing tile
    metadata) raises from submit, a tile task that fails raises from result() — in both cases the images around it are
    complete and the predictor keeps all its buffer slots."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0)
    tifs = []
    for k in range(3):
        rgb, _ = make_tile(310 + k, 500)
        tif = str(tmp_path / f"{k}.tif")
        write_geotiff(tif, np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1)), t, 25832)
        tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
        tifs.append(tif)
    tj = lambda tif: str(tmp_path / "tiles" / (os.path.basename(tif).replace(".tif", ".json")))      # noqa: E731
    cfg = T.setup_model_cfg(update_model="x", device="0")
    read = lambda d: {f"{sub}/{f}": open(d / sub / f, "rb").read() for sub in sorted(os.listdir(d)) for f in sorted(os.listdir(d / sub))}   # noqa: E731
    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "seq"), state_dict=sd) as pred:
        seq = [pred(tif, tj(tif)) for tif in tifs]
    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "chain"), state_dict=sd) as pred:
        nslots = pred._free.qsize()
        handles = [pred.submit(tif, tj(tif)) for tif in tifs]            # three images in flight
        chained = [h.result() for h in handles]
        assert handles[0].result() is chained[0]                        # result() may be asked twice
        assert pred._free.qsize() == nslots
        # a bad image between two good ones
        h0 = pred.submit(tifs[0], tj(tifs[0]))
        with pytest.raises(FileNotFoundError):
            pred.submit(tifs[1], str(tmp_path / "tiles" / "missing.json"))
        h2 = pred.submit(tifs[2], tj(tifs[2]))
        assert h0.result() == chained[0] and h2.result() == chained[2]
        # tile tasks that fail (their output folder vanishes under them): result() may raise, every slot comes back
        import shutil
        h1 = pred.submit(tifs[1], tj(tifs[1]))
        shutil.rmtree(tmp_path / "chain" / "1", ignore_errors=True)
        try:
            h1.result()
        except Exception:
            pass
        shutil.rmtree(tmp_path / "chain" / "1", ignore_errors=True)
        assert pred(tifs[1], tj(tifs[1])) == chained[1]
        assert pred._free.qsize() == nslots
    assert chained == seq and sum(len(c) for c in chained) > 10
    assert read(tmp_path / "seq") == read(tmp_path / "chain")


@pytest.mark.parametrize("mode", ["streams", "phases", "plain"])
def test_launch_failures_do_not_leak_buffer_slots(tmp_path, mode):
    """the buffer slots live in ONE free list for the Predictor's lifetime. A launch that fails after the launcher took
    (batch, slot) from the reader has no epilogue task to return the slot — so the launcher must. The engine's forward is made to
    fail for MORE images than there are slots (predict_on_model logs and walks on, reference detection.py:117-120); afterwards every
    slot is back, no stand-in slot stays in the list, and a good image gives the same files as on a fresh predictor."""
    import treedetection_amd as T
    from treedetection_amd.preprocessing import tile_single_file
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rgb, _ = make_tile(320, 500)
    tif = str(tmp_path / "0.tif")
    write_geotiff(tif, np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1)),
                  (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=10, tile_width=40, tile_height=40)
    tj = str(tmp_path / "tiles" / "0.json")
    cfg = T.setup_model_cfg(update_model="x", device="0")
    kw = dict(pipeline=mode != "plain", schedule=mode if mode != "plain" else "streams")
    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "good"), state_dict=sd, **kw) as pred:
        good = pred(tif, tj)
    assert len(good) > 3
    with T.Predictor(cfg, device_type="0", max_batch_size=2, output_dir=str(tmp_path / "out"), state_dict=sd, **kw) as pred:
        nslots = pred._free.qsize()
        calls = {"n": 0}

        def boom(*a, **k):
            calls["n"] += 1
            raise RuntimeError("injected launch failure")
        saved = [(e, e.forward_raw, e.forward_phase) for e in pred._engines]
        for e in pred._engines:
            e.forward_raw = boom
            e.forward_phase = boom
        for _ in range(nslots + 3):              # more failed images than slots
            with pytest.raises(RuntimeError, match="injected launch failure"):
                pred(tif, tj)
        assert calls["n"] >= nslots + 3
        for e, fr, fp in saved:
            e.forward_raw, e.forward_phase = fr, fp
        assert pred._free.qsize() == nslots
        assert not any(s.transient for s in list(pred._free.queue))
        assert pred(tif, tj) == good
        assert pred._free.qsize() == nslots
