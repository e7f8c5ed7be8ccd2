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
    assert os.path.exists(root / "output" / "geojson_predictions" / "324125317.geojson")
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
