"""BASELINE configs[2] at its stated shape: the two-model (urban + forest) path over 1000x1000-px tiles of a 4-band RGBI
raster (+ nDSM side file), two FULL-WIDTH R50-FPN weight sets, a forest outline that flags tiles both ways
(reference TreeDetection/detection.py:154-164: urban model with exclude_vars ["only_forest"], forest model with
["only_urban"]; prediction.py:79-93: a tile whose flag is set is skipped by that model).

Checked: which tiles each model visits; for one tile that ONLY the urban model and one that ONLY the forest model
predicts, the engine's detections against the oracle with that model's weights at the fp32 tolerances of
tests/test_fullsize_gpu.py (boxes <= 1e-2 px, scores <= 1e-4, mask probabilities <= 1e-3, pasted-mask IoU >= 0.995,
detections one to one) and the written Prediction_*.json against the oracle's contours (score <= 1e-4, rings <= 1 px)."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import ops_ref as R
from oracle.contours_ref import find_contours as ref_contours
from oracle.maskrcnn_ref import MaskRCNNOracle
from treedetection_amd.geotiff import GeoTiff, write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

pytestmark = pytest.mark.gpu

GSD = 0.2
COLS, ROWS = 3, 2          # 3 x 2 tiles of 200 m = 1000 px


@pytest.fixture(scope="module")
def two_model(tmp_path_factory):
    import treedetection_amd as T
    torch.set_num_threads(8)
    root = tmp_path_factory.mktemp("cfg2")
    (root / "rgb").mkdir()
    (root / "ndsm").mkdir()
    sds = {}
    for seed, name in ((0, "urban"), (2, "forest")):      # seeds whose random heads detect something on these tiles
        sds[name] = make_synthetic_state_dict(50, seed=seed)            # full width
        np.savez(root / f"model_{name}.npz", **sds[name])
    rgbi = np.zeros((4, ROWS * 1000, COLS * 1000), np.uint8)
    ndsm = np.zeros((ROWS * 1000, COLS * 1000), np.float32)
    for r in range(ROWS):
        for c in range(COLS):
            rgb, nd = make_tile(300 + r * COLS + c, 1000)
            rgbi[:3, r * 1000:(r + 1) * 1000, c * 1000:(c + 1) * 1000] = rgb.transpose(2, 0, 1)
            rgbi[3, r * 1000:(r + 1) * 1000, c * 1000:(c + 1) * 1000] = rgb[..., 1] // 2 + 60      # NIR band (unused by the model)
            ndsm[r * 1000:(r + 1) * 1000, c * 1000:(c + 1) * 1000] = nd
    x0, y_top = 412000.0, 5318000.0 + ROWS * 200.0
    t = (GSD, 0.0, x0, 0.0, -GSD, y_top)
    write_geotiff(str(root / "rgb" / "1.tif"), rgbi, t, 25832)
    write_geotiff(str(root / "ndsm" / "1.tif"), ndsm[::5, ::5].copy(), (1.0, 0.0, x0, 0.0, -1.0, y_top), 25832)
    # forest = everything left of x0 + 300 m: column 0 lies within it (only_forest), column 1 straddles its border
    # (both models), column 2 is outside (only_urban)
    outline = {"type": "FeatureCollection", "features": [{"type": "Feature", "properties": {}, "geometry": {
        "type": "Polygon", "coordinates": [[[x0 - 100, y_top - 600], [x0 + 300, y_top - 600], [x0 + 300, y_top + 100],
                                            [x0 - 100, y_top + 100], [x0 - 100, y_top - 600]]]}}]}
    (root / "forest.geojson").write_text(json.dumps(outline))
    cfg = {"image_directory": str(root / "rgb"), "height_data_path": str(root / "ndsm"),
           "urban_model": str(root / "model_urban.npz"), "forrest_model": str(root / "model_forest.npz"),
           "forrest_outline": str(root / "forest.geojson"), "output_directory": str(root / "output"),
           "tiles_path": str(root / "tiles"), "use_overlap": False, "tile_width": 200, "tile_height": 200, "buffer": 0,
           "batch_size": 4, "parallel": False, "num_workers": 2, "keep_intermediate": True, "device": "0"}
    (root / "config.yml").write_text(yaml.safe_dump(cfg))
    config, _ = T.get_config(str(root / "config.yml"))
    T.preprocess_files(config)
    T.predict_tiles(config)
    meta = json.load(open(root / "tiles" / "1.json"))
    return root, config, sds, meta


def test_each_model_visits_the_tiles_its_flags_leave(two_model):
    root, config, sds, meta = two_model
    assert len(meta) == COLS * ROWS
    only_forest = {k for k, v in meta.items() if v["only_forest"]}
    only_urban = {k for k, v in meta.items() if v["only_urban"]}
    assert len(only_forest) == ROWS and len(only_urban) == ROWS and not (only_forest & only_urban)
    for v in meta.values():         # 1000 x 1000-px windows
        b = v["bounds"]
        assert round((b[2] - b[0]) / GSD) == 1000 and round((b[3] - b[1]) / GSD) == 1000
    urban = {f[len("Prediction_"):-5] for f in os.listdir(root / "output" / "urban_predictions" / "1")}
    forest = {f[len("Prediction_"):-5] for f in os.listdir(root / "output" / "forrest_predictions" / "1")}
    assert urban == set(meta) - only_forest and forest == set(meta) - only_urban
    assert len(urban) + len(forest) == 2 * COLS * ROWS - 2 * ROWS          # tiles VISITED by the two models
    for d in ("urban_geojson", "forrest_geojson", "geojson_predictions"):
        assert os.path.exists(root / "output" / d / "1.gpkg")


@pytest.mark.parametrize("model,flag,folder", [("urban", "only_urban", "urban_predictions"), ("forest", "only_forest", "forrest_predictions")])
def test_one_tile_per_model_matches_the_oracle(two_model, model, flag, folder):
    from treedetection_amd.engine import Engine, INPUT_U8_HWC, unpack_outputs
    root, config, sds, meta = two_model
    tile_id = sorted(k for k, v in meta.items() if v[flag])[0]       # a tile only THIS model predicts
    td = meta[tile_id]
    tif = str(root / "rgb" / "1.tif")
    bands = GeoTiff(tif).read_bounds(td["bounds"])                   # [4, 1000, 1000] uint8, file band order
    assert bands.shape == (4, 1000, 1000)
    x, h, w = R.preprocess_tile_u8(bands)
    ref = MaskRCNNOracle(sds[model]).forward([{"image": x, "height": h, "width": w}])[0]
    assert len(ref["scores"]) >= 5
    # (1) the engine with this model's weights on the tile's bytes (the Predictor's device path: band pick + resize + forward + paste)
    eng = Engine(sds[model])
    tile = torch.from_numpy(np.ascontiguousarray(bands[:3].transpose(1, 2, 0))).cuda()
    batch, hv, ho = eng.preprocess_tiles_u8([tile])
    out = eng.alloc_outputs(1, 1000, 1000, paste=True)
    eng.forward_raw(batch, INPUT_U8_HWC, hv, ho, out)
    torch.cuda.synchronize()
    g = unpack_outputs(out, ho, True)[0]
    eng.close()
    assert len(g["scores"]) == len(ref["scores"])          # the oracle's detection set, exactly
    matched = 0
    for j in range(len(ref["scores"])):
        d = np.abs(g["pred_boxes"] - ref["pred_boxes"][j]).max(axis=1)
        k = int(np.argmin(d))
        if d[k] <= 1e-2 and abs(g["scores"][k] - ref["scores"][j]) <= 1e-4:
            a, b = g["pred_masks"][k], ref["pred_masks"][j]
            u = (a | b).sum()
            assert u == 0 or (a & b).sum() / u >= 0.995
            assert np.abs(g["mask_probs"][k] - ref["mask_probs"][j]).max() <= 1e-3
            matched += 1
    assert matched == len(ref["scores"]), (matched, len(ref["scores"]))
    # (2) the file predict_tiles wrote for this tile with this model
    got = json.load(open(root / "output" / folder / "1" / f"Prediction_{tile_id}.json"))
    exp = []
    t = td["transform"]
    for d in range(len(ref["scores"])):
        for c in ref_contours(ref["pred_masks"][d]):
            if c.size < 8:
                continue
            cx, cy = c[:, 0].tolist(), c[:, 1].tolist()
            if (cx[0], cy[0]) != (cx[-1], cy[-1]):
                cx.append(cx[0]); cy.append(cy[0])
            exp.append((float(ref["scores"][d]), [[t[0] * a + t[1] * b + t[2], t[3] * a + t[4] * b + t[5]] for a, b in zip(cx, cy)]))
    assert abs(len(got) - len(exp)) <= max(2, 0.02 * len(exp)), (len(got), len(exp))
    same = 0
    for e in got:
        assert e["image_id"] == tif and e["category_id"] == 0
        for s, ring in exp:
            if abs(e["score"] - s) <= 1e-4 and len(ring) == len(e["polygon_coords"][0]):
                if np.abs(np.asarray(ring) - np.asarray(e["polygon_coords"][0])).max() <= GSD + 1e-9:      # <= 1 px
                    same += 1
                    break
    assert same >= 0.9 * len(exp), (same, len(exp))
    # the OTHER model never wrote this tile
    other = "forrest_predictions" if folder == "urban_predictions" else "urban_predictions"
    assert not os.path.exists(root / "output" / other / "1" / f"Prediction_{tile_id}.json")


@pytest.fixture(scope="module")
def two_model_fp16(two_model):
    """The same two-model run through the fp16 engine (config key `precision: fp16`) into its own output folder — the flow
    bench.py times as `two_model_f16`."""
    import treedetection_amd as T
    root, config, sds, meta = two_model
    cfg = yaml.safe_load((root / "config.yml").read_text())
    cfg["precision"] = "fp16"
    cfg["output_directory"] = str(root / "output_fp16")
    (root / "config_fp16.yml").write_text(yaml.safe_dump(cfg))
    config16, _ = T.get_config(str(root / "config_fp16.yml"))
    T.predict_tiles(config16)
    return root, sds, meta


@pytest.mark.parametrize("model,flag,folder", [("urban", "only_urban", "urban_predictions"), ("forest", "only_forest", "forrest_predictions")])
def test_one_tile_per_model_matches_the_oracle_fp16(two_model_fp16, model, flag, folder):
    """Two-model flow x fp16 (reference detection.py:154-164 with the fp16 MFMA engine): per model, a tile only THAT model
    predicts — the fp16 engine with that model's weights against the fp32 oracle at the fp16 tolerances of
    tests/test_engine_fp16_gpu.py, and the file predict_tiles(precision=fp16) wrote for it: one entry per contour of the
    engine's own detections (same scores bit for bit), the other model never wrote it."""
    from tests.test_engine_fp16_gpu import check_fp16_detections
    from treedetection_amd.engine import Engine, INPUT_U8_HWC, unpack_outputs
    root, sds, meta = two_model_fp16
    tile_id = sorted(k for k, v in meta.items() if v[flag])[0]
    td = meta[tile_id]
    tif = str(root / "rgb" / "1.tif")
    bands = GeoTiff(tif).read_bounds(td["bounds"])
    x, h, w = R.preprocess_tile_u8(bands)
    ref = MaskRCNNOracle(sds[model]).forward([{"image": x, "height": h, "width": w}])[0]
    eng = Engine(sds[model], precision="fp16")
    tile = torch.from_numpy(np.ascontiguousarray(bands[:3].transpose(1, 2, 0))).cuda()
    batch, hv, ho = eng.preprocess_tiles_u8([tile])
    out = eng.alloc_outputs(1, 1000, 1000, paste=True)
    eng.forward_raw(batch, INPUT_U8_HWC, hv, ho, out)
    torch.cuda.synchronize()
    g = unpack_outputs(out, ho, True)[0]
    eng.close()
    check_fp16_detections([g], [ref], f"configs[2] {model} model, fp16")
    got = json.load(open(root / "output_fp16" / folder / "1" / f"Prediction_{tile_id}.json"))
    assert len(got) >= len(g["scores"]) * 0.5
    eng_scores = {float(s) for s in g["scores"]}
    assert {e["score"] for e in got} <= eng_scores               # every written entry belongs to a detection of the fp16 engine
    for e in got:
        assert e["image_id"] == tif and e["category_id"] == 0 and len(e["polygon_coords"][0]) >= 4
    other = "forrest_predictions" if folder == "urban_predictions" else "urban_predictions"
    assert not os.path.exists(root / "output_fp16" / other / "1" / f"Prediction_{tile_id}.json")
    # the fp16 run visits exactly the tiles the fp32 run visits
    for f in ("urban_predictions", "forrest_predictions"):
        assert set(os.listdir(root / "output_fp16" / f / "1")) == set(os.listdir(root / "output" / f / "1"))


@pytest.fixture(scope="module")
def two_model_trained(two_model):
    """The two-model flow once more with detectors whose box heads are TRAINED — committed fixtures (round 6:
    tests/golden/trained_heads_{urban,forest}.npz, made once on the CPU by tests/golden/make_trained_heads.py, hash-checked at load;
    nothing is trained on the GPU box) — each model on the two tiles of the column that ONLY it predicts (the raster is a 3 x 2 grid
    of generator tiles 300 … 305, so every tile's crowns are known) — through `precision: fp16` into its own output folder."""
    import treedetection_amd as T
    from tests.trained_heads import FIXTURES, load_trained_heads
    from treedetection_amd.weights import blob_mask_head
    root, config, sds, meta = two_model
    picks, trained = {}, {}
    for model, flag in (("urban", "only_urban"), ("forest", "only_forest")):
        tid = sorted(k for k, v in meta.items() if v[flag])[0]
        minx, miny = (int(p) for p in tid.split("_")[1:3])
        c, r = (minx - 412000) // 200, ROWS - 1 - (miny - 5318000) // 200
        picks[model] = (tid, 300 + r * COLS + c)
        column = [300 + rr * COLS + c for rr in range(ROWS)]                    # the tiles only this model sees
        assert tuple(column) == FIXTURES[model][2]                              # … are the tiles its fixture was trained on
        trained[model] = load_trained_heads(model, base=blob_mask_head(sds[model]))
        np.savez(root / f"model_{model}_trained.npz", **trained[model])
    cfg = yaml.safe_load((root / "config.yml").read_text())
    cfg.update(precision="fp16", output_directory=str(root / "output_trained_fp16"),
               urban_model=str(root / "model_urban_trained.npz"), forrest_model=str(root / "model_forest_trained.npz"))
    (root / "config_trained_fp16.yml").write_text(yaml.safe_dump(cfg))
    config16, _ = T.get_config(str(root / "config_trained_fp16.yml"))
    T.predict_tiles(config16)
    return root, trained, picks


@pytest.mark.parametrize("model,folder", [("urban", "urban_predictions"), ("forest", "forrest_predictions")])
def test_two_model_flow_fp16_set_rule_on_trained_box_heads(two_model_trained, model, folder):
    """The fp16 set rule on the two-model flow (reference detection.py:154-164) at full size: per model, on a tile only THAT model
    predicts, the fp16 engine against the fp32 oracle with the model's TRAINED box head (committed fixture) — every detection clear
    of the score cut pairs one-to-one at IoU >= 0.9 up to the stated share of near-tie exceptions, and the pairs meet the stated
    fp16 tolerances (tests/test_engine_fp16_gpu.FP16_TRAINED = BASELINE.md §3.4: the same bounds as the single-model fixtures) — and
    the file `predict_tiles(precision: fp16)` wrote for
    the tile carries exactly the fp16 engine's detections (scores bit for bit), the other model never wrote it."""
    from tests.test_engine_fp16_gpu import SCORE_THRESH, assert_fp16_trained_statement, match_detection_sets
    from treedetection_amd.engine import Engine, INPUT_U8_HWC, unpack_outputs
    root, trained, picks = two_model_trained
    tile_id, _ = picks[model]
    meta = json.load(open(root / "tiles" / "1.json"))
    tif = str(root / "rgb" / "1.tif")
    bands = GeoTiff(tif).read_bounds(meta[tile_id]["bounds"])
    x, h, w = R.preprocess_tile_u8(bands)
    ref = MaskRCNNOracle(trained[model]).forward([{"image": x, "height": h, "width": w}])[0]
    eng = Engine(trained[model], precision="fp16")
    tile = torch.from_numpy(np.ascontiguousarray(bands[:3].transpose(1, 2, 0))).cuda()
    batch, hv, ho = eng.preprocess_tiles_u8([tile])
    out = eng.alloc_outputs(1, 1000, 1000, paste=True)
    eng.forward_raw(batch, INPUT_U8_HWC, hv, ho, out)
    torch.cuda.synchronize()
    g = unpack_outputs(out, ho, True)[0]
    eng.close()
    band = 5e-3 * 4.0 * SCORE_THRESH * (1.0 - SCORE_THRESH) / 0.36
    strict, cluster, lost, extra = match_detection_sets(g, ref, band)
    far = lambda d, idx: d["scores"][idx] > SCORE_THRESH + band      # noqa: E731
    exceptions = sum(1 for i, j, _ in cluster if far(ref, i) or far(g, j)) + len(lost) + len(extra)
    print(f"\n[fp16 two-model, trained box head, {model}] tile {tile_id}: {len(ref['scores'])} oracle / {len(g['scores'])} engine detections, "
          f"{len(strict)} strict pairs, cluster pairs {[round(v, 2) for _, _, v in cluster]}, unpaired oracle {np.round(lost, 3).tolist()} "
          f"engine {np.round(extra, 3).tolist()}")
    assert 20 <= len(ref["scores"]) <= 60
    rows = [(abs(float(g["scores"][j]) - float(ref["scores"][i])), float(np.abs(g["pred_boxes"][j] - ref["pred_boxes"][i]).max()),
             float(np.abs(g["mask_probs"][j] - ref["mask_probs"][i]).max()), float(ref["scores"][i]), v) for i, j, v in strict]
    assert_fp16_trained_statement(model, rows, [exceptions], len(ref["scores"]))
    got = json.load(open(root / "output_trained_fp16" / folder / "1" / f"Prediction_{tile_id}.json"))
    eng_scores = {float(s) for s in g["scores"]}
    assert len(got) >= len(g["scores"]) and {e["score"] for e in got} <= eng_scores
    assert {e["score"] for e in got} == eng_scores                    # compact masks: every detection has a contour
    other = "forrest_predictions" if folder == "urban_predictions" else "urban_predictions"
    assert not os.path.exists(root / "output_trained_fp16" / other / "1" / f"Prediction_{tile_id}.json")
