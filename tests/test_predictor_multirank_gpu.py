"""N > 1 Predictor path on ONE GPU: 2 ranks (gloo rendezvous, both on cuda:0) shard the tiles of an image, gather the
detections to rank 0, and rank 0 writes every Prediction_*.json — the files must equal the single-process run's.
Also: ``process_files`` end to end under 2 ranks with ``keep_intermediate: false`` (rank 0 alone touches the shared
folders, nothing is removed under it), and BASELINE configs[3]'s shape — a 100 x 100 mosaic (10 000 tiles, reference
tile-id scheme) sharded over 2 ranks at a reduced tile size, every tile written exactly once."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import yaml

from treedetection_amd.geotiff import write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from treedetection_amd.config import setup_model_cfg
from treedetection_amd.prediction import Predictor
from treedetection_amd.weights import load_checkpoint
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
sd = load_checkpoint({model!r})
pred = Predictor(setup_model_cfg(update_model="x", device="0"), device_type="0", max_batch_size={batch}, output_dir={out!r},
                 state_dict=sd, return_predictions=False, **{kw!r})
pred({tif!r}, {meta!r})
pred.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""

PROCESS_FILES_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
import treedetection_amd as T
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
config, _ = T.get_config({cfg!r})
T.process_files(config)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(script, world, timeout=600, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    if world == 1:
        subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=timeout)
    else:
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       check=True, env=env, timeout=timeout)


def test_two_ranks_write_the_same_files_as_one(tmp_path):
    from treedetection_amd.preprocessing import tile_single_file
    np.savez(tmp_path / "m.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    rgb, _ = make_tile(100, 600)
    tif = str(tmp_path / "img.tif")
    write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0, 0, 0, -0.2, 120.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=20, tile_width=50, tile_height=50)
    meta = str(tmp_path / "tiles" / "img.json")
    outs = {}
    runs = {"one": (1, {"pipeline": False}),
            "two_plain": (2, {"pipeline": False}),
            "two_pipelined": (2, {"pipeline": True}),              # three engines per rank, gather on the side stream
            "two_local": (2, {"pipeline": True, "sharded_epilogue": "local"})}
    for name, (world, kw) in runs.items():
        out = str(tmp_path / f"out_{name}")
        script = tmp_path / f"w_{name}.py"
        script.write_text(WORKER.format(root=ROOT, model=str(tmp_path / "m.npz"), out=out, tif=tif, meta=meta, batch=3, kw=kw))
        _run(script, world)
        outs[name] = {f: open(os.path.join(out, "img", f), "rb").read() for f in sorted(os.listdir(os.path.join(out, "img")))}
    assert len(outs["one"]) == 9
    for name in runs:
        assert sorted(outs[name]) == sorted(outs["one"]), name
        for f in outs["one"]:
            assert outs[name][f] == outs["one"][f], (name, f)       # byte for byte
    assert sum(len(json.loads(v)) for v in outs["one"].values()) > 10


def test_process_files_two_ranks_cleanup(tmp_path):
    """``process_files`` under torch.distributed with the default ``keep_intermediate: false``: preprocess on rank 0
    only, predict sharded, stitch + post-process on rank 0, and only then (barrier) the intermediate folders go —
    the final layers must equal the single-process run's and nothing may be missing."""
    root = tmp_path
    np.savez(root / "model_combined.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    finals = {}
    for world in (1, 2):
        base = root / f"w{world}"
        (base / "rgb").mkdir(parents=True)
        (base / "ndsm").mkdir()
        for k, name in enumerate(("3241", "3242")):
            rgb, ndsm = make_tile(200 + k, 400)
            rgbi = np.concatenate([rgb, 255 - rgb[..., :1] // 2], axis=2).transpose(2, 0, 1)
            t = (0.2, 0.0, 412000.0 + 80 * k, 0.0, -0.2, 5318080.0)                  # 80 m x 80 m, side by side
            write_geotiff(str(base / "rgb" / f"{name}.tif"), np.ascontiguousarray(rgbi), t, 25832)
            write_geotiff(str(base / "ndsm" / f"{name}.tif"), (ndsm[::5, ::5] + 5).copy(), (1.0, 0, t[2], 0, -1.0, t[5]), 25832)
        cfg = {"image_directory": str(base / "rgb"), "height_data_path": str(base / "ndsm"),
               "combined_model": str(root / "model_combined.npz"), "output_directory": str(base / "output"),
               "tiles_path": str(base / "tiles"), "use_overlap": True, "overlapping_tiles_width": 2,
               "overlapping_tiles_height": 2, "tile_width": 40, "tile_height": 40, "buffer": 10,
               "batch_size": 4, "parallel": False, "num_workers": 2, "keep_intermediate": False, "device": "0",
               "height_threshold": 0}
        (base / "config.yml").write_text(yaml.safe_dump(cfg))
        script = base / "run.py"
        script.write_text(PROCESS_FILES_WORKER.format(root=ROOT, cfg=str(base / "config.yml")))
        _run(script, world)
        out = base / "output"
        assert not (base / "tiles").exists() and not (base / "rgb" / "merged").exists()   # intermediates are gone ...
        assert not (out / "predictions").exists() and not (out / "geojson_predictions").exists()
        layers = sorted(f for f in os.listdir(out) if f.endswith(".gpkg"))
        assert layers, os.listdir(out)                                # ... and the final layers are there
        finals[world] = {f: os.path.getsize(out / f) for f in layers}
    assert sorted(finals[1]) == sorted(finals[2])
    # same crowns either way: the GeoPackage payloads have the same size (feature bytes are deterministic)
    assert finals[1] == finals[2]


def test_config3_mosaic_10k_tiles_sharded(tmp_path):
    """BASELINE configs[3] on one GPU: a 100 x 100 mosaic = 10 000 tiles with the reference's tile-id scheme
    (TreeDetection/preprocessing.py:59), sharded i = r (mod W) over 2 ranks, padded rounds, gather to rank 0 — at a
    reduced tile size (50 x 50 px; every tile still goes through the 800 x 800 network input) with a half-width
    model, three engines per rank. Every tile must be written exactly once and carry valid JSON."""
    from treedetection_amd import distributed as D
    from treedetection_amd.preprocessing import tile_single_file
    np.savez(tmp_path / "m.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    rgb, _ = make_tile(7, 1000)
    mosaic = np.tile(rgb, (5, 5, 1))                                   # 5000 x 5000 px = 100 x 100 tiles of 50 px
    tif = str(tmp_path / "mosaic.tif")
    write_geotiff(tif, np.ascontiguousarray(mosaic.transpose(2, 0, 1)), (0.2, 0, 412000.0, 0, -0.2, 5319000.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=0, tile_width=10, tile_height=10)
    meta_path = str(tmp_path / "tiles" / "mosaic.json")
    meta = json.load(open(meta_path))
    assert len(meta) == 10000
    first = next(iter(meta))
    assert first == "mosaic_412000_5318000_10_0_25832"                 # {stem}_{minx}_{miny}_{tile_width}_{buffer}_{epsg}
    B = 16
    assert D.padded_rounds(10000, B, 2) == 313
    assert len(D.shard_indices(10000, 0, 2)) == len(D.shard_indices(10000, 1, 2)) == 5000
    out = str(tmp_path / "out")
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, model=str(tmp_path / "m.npz"), out=out, tif=tif, meta=meta_path, batch=B,
                                    kw={"pipeline": True}))
    _run(script, 2, timeout=1100)
    files = sorted(os.listdir(os.path.join(out, "mosaic")))
    assert files == sorted(f"Prediction_{k}.json" for k in meta)      # each of the 10 000 tiles exactly once
    nonempty = 0
    for f in files[::97]:
        entries = json.load(open(os.path.join(out, "mosaic", f)))
        for e in entries:
            assert e["image_id"] == tif and 0.3 < e["score"] <= 1.0 and len(e["polygon_coords"][0]) >= 4
        nonempty += bool(entries)
    assert nonempty > 0


def test_config3_tile_shard_and_rank0_gather_at_full_tile_size(tmp_path):
    """BASELINE configs[3]'s path — tiles of ONE mosaic sharded i = r (mod W), detections gathered to rank 0, rank 0 pastes,
    traces and writes every file — at the WORKLOAD's size: full-width R50-FPN, 1000 x 1000-px tiles, the
    400-tile raster of bench.py's e2e regions (20 x 20 tiles, 16 distinct generator tiles cycled), batch 8, three engines per
    rank; 2 gloo ranks on this one GPU against ONE process: every Prediction_*.json byte for byte. (The RCCL transport on 8
    GPUs is hardware-only; what runs here is everything around it: sharding, padded rounds, the fixed-shape gather, rank 0's
    paste of other ranks' batches.)"""
    import shutil
    from treedetection_amd import distributed as D
    from treedetection_amd.preprocessing import tile_single_file
    from treedetection_amd.weights import blob_mask_head
    np.savez(tmp_path / "m.npz", **blob_mask_head(make_synthetic_state_dict(50, seed=0)))          # full width, compact crowns
    S, side = 1000, 20
    distinct = [np.ascontiguousarray(make_tile(k, S)[0].transpose(2, 0, 1)) for k in range(16)]
    base = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    shm = os.path.join(base, f"td_cfg3_{os.getpid()}")
    os.makedirs(shm, exist_ok=True)
    try:
        tif = os.path.join(shm, "mosaic.tif")
        mosaic = np.empty((3, side * S, side * S), np.uint8)                  # 1.2 GB, like the bench's raster
        for r in range(side):
            for c in range(side):
                mosaic[:, r * S:(r + 1) * S, c * S:(c + 1) * S] = distinct[(r * side + c) % 16]
        write_geotiff(tif, mosaic, (0.2, 0, 412000.0, 0, -0.2, 5318000.0 + side * S * 0.2), 25832)
        del mosaic
        tile_single_file(tif, str(tmp_path / "tiles"), buffer=0, tile_width=200, tile_height=200)
        meta_path = str(tmp_path / "tiles" / "mosaic.json")
        meta = json.load(open(meta_path))
        assert len(meta) == 400
        assert D.padded_rounds(400, 8, 2) == 25 and len(D.shard_indices(400, 1, 2)) == 200
        outs = {}
        for name, world in (("one", 1), ("two", 2)):
            out = str(tmp_path / f"out_{name}")
            script = tmp_path / f"w_{name}.py"
            script.write_text(WORKER.format(root=ROOT, model=str(tmp_path / "m.npz"), out=out, tif=tif, meta=meta_path, batch=8,
                                            kw={"pipeline": True, "sharded_epilogue": "rank0"}))
            _run(script, world, timeout=900)
            outs[name] = {f: open(os.path.join(out, "mosaic", f), "rb").read() for f in sorted(os.listdir(os.path.join(out, "mosaic")))}
    finally:
        shutil.rmtree(shm, ignore_errors=True)
    assert sorted(outs["one"]) == sorted(f"Prediction_{k}.json" for k in meta)
    assert sorted(outs["two"]) == sorted(outs["one"])
    for f in outs["one"]:
        assert outs["two"][f] == outs["one"][f], f                           # byte for byte
    dets = sum(len(json.loads(v)) for v in outs["one"].values())
    assert dets > 400 * 5, dets                                               # ~ 20 crowns per tile on this fixture


PREDICT_TILES_WORKER = r"""
import logging, os, sys
sys.path.insert(0, {root!r})
import torch.distributed as dist
import treedetection_amd as T
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rank = dist.get_rank() if world > 1 else 0
log = logging.getLogger("td")
log.setLevel(logging.INFO)
log.addHandler(logging.FileHandler(os.path.join({out!r}, f"log_rank{{rank}}.txt")))
config = dict({cfg!r}, logger=log)
T.predict_tiles(config)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def _layer_rows(path):
    import sqlite3
    con = sqlite3.connect(path)
    try:
        table = con.execute("SELECT table_name FROM gpkg_contents").fetchone()[0]
        return con.execute(f'SELECT * FROM "{table}" ORDER BY fid').fetchall()
    finally:
        con.close()


def test_image_level_sharding_two_ranks_equals_one_process(tmp_path):
    """with at least as many images as ranks every rank owns WHOLE images (detection.assign_images), runs
    them through the chained single-process pipeline, writes their tile files and stitches them while it predicts the next
    one; no collective per image. A 6-image folder, one of them unreadable (logged by its owner, the walk goes on on every
    rank, reference detection.py:117-120): the 2-rank run's Prediction_*.json are byte-identical to the single process's
    and the stitched layers hold the same rows."""
    from treedetection_amd.preprocessing import tile_single_file
    np.savez(tmp_path / "model_combined.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    rgb_dir, tiles = tmp_path / "rgb", tmp_path / "tiles"
    rgb_dir.mkdir()
    names = ["3241", "3242", "3243", "3244", "3245", "3246"]
    for k, name in enumerate(names):
        tif = str(rgb_dir / f"{name}.tif")
        if name == "3244":
            continue
        rgb, _ = make_tile(300 + k, 400)
        write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0.0, 412000.0 + 80 * k, 0.0, -0.2, 5318080.0), 25832)
        tile_single_file(tif, str(tiles), buffer=10, tile_width=40, tile_height=40)
    (rgb_dir / "3244.tif").write_bytes(b"II*\x00 not a raster")                      # listed, tiled once, unreadable now
    (tiles / "3244.json").write_text((tiles / "3243.json").read_text().replace("3243", "3244"))
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"out{world}"
        out.mkdir()
        cfg = {"image_directory": str(rgb_dir), "merged_path": "merged", "tiles_path": str(tiles), "output_directory": str(out),
               "device": "0", "simplify_tolerance": 0.2, "num_workers": 2, "batch_size": 3, "precision": "fp32",
               "combined_model": str(tmp_path / "model_combined.npz")}
        script = tmp_path / f"pt{world}.py"
        script.write_text(PREDICT_TILES_WORKER.format(root=ROOT, out=str(out), cfg=cfg))
        _run(script, world)
        outs[world] = out
    logs = "".join((outs[2] / f"log_rank{r}.txt").read_text() for r in (0, 1))
    assert "sharding by image" in logs and "3244.tif" in logs and "Error processing" in logs
    # the unreadable image belongs to ONE rank; the other rank's log never mentions it
    assert sum("3244.tif" in (outs[2] / f"log_rank{r}.txt").read_text() for r in (0, 1)) == 1
    total = 0
    for name in names:
        d1, d2 = outs[1] / "predictions" / name, outs[2] / "predictions" / name
        if name == "3244":
            assert not d1.exists() or not os.listdir(d1)
            assert not d2.exists() or not os.listdir(d2)
        else:
            f1, f2 = sorted(os.listdir(d1)), sorted(os.listdir(d2))
            assert f1 == f2 and len(f1) == 4, (name, f1, f2)
            for f in f1:
                assert (d1 / f).read_bytes() == (d2 / f).read_bytes(), (name, f)          # byte for byte
                total += len(json.loads((d1 / f).read_bytes()))
        r1 = _layer_rows(str(outs[1] / "geojson_predictions" / f"{name}.gpkg"))
        r2 = _layer_rows(str(outs[2] / "geojson_predictions" / f"{name}.gpkg"))
        assert r1 == r2, name
    assert total > 20
    s1, s2 = (yaml.safe_load(open(o / "geojson_predictions" / "stitching_recovery.yaml")) for o in (outs[1], outs[2]))
    assert s1 == s2 and len(s1["completed_files"]) == 6


def test_a_tile_whose_crop_fails_is_dropped_in_every_mode(tmp_path):
    """Reference prediction.py:174-176: a tile whose crop raises is printed and DROPPED — no Prediction file, the image goes
    on. Single process, 2 ranks with the "rank0" epilogue (plain and pipelined: the failed tile travels as a black stand-in so
    that rank 0 can derive every rank's batch structure, with count = -1 in the gather) and 2 ranks with the "local" epilogue
    (the manifest lists it as dropped): the same files everywhere, the failed tile's file nowhere, nobody raises."""
    from treedetection_amd.preprocessing import tile_single_file
    np.savez(tmp_path / "m.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    rgb, _ = make_tile(100, 600)
    tif = str(tmp_path / "img.tif")
    write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0, 0, 0, -0.2, 120.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=20, tile_width=50, tile_height=50)
    meta = str(tmp_path / "tiles" / "img.json")
    ids = list(json.load(open(meta)))
    bad = ids[4]                                           # rank 0's tile under i = r (mod 2); also try one of rank 1's below
    outs = {}
    runs = {"one": (1, {"pipeline": True}, bad), "two_rank0": (2, {"pipeline": False, "sharded_epilogue": "rank0"}, bad),
            "two_rank0_pipelined": (2, {"pipeline": True, "sharded_epilogue": "rank0"}, bad),
            "two_local": (2, {"pipeline": True, "sharded_epilogue": "local"}, bad),
            "two_rank0_other_rank": (2, {"pipeline": True, "sharded_epilogue": "rank0"}, ids[3]), "one_other": (1, {"pipeline": True}, ids[3])}
    for name, (world, kw, fault) in runs.items():
        out = str(tmp_path / f"out_{name}")
        script = tmp_path / f"w_{name}.py"
        script.write_text(WORKER.format(root=ROOT, model=str(tmp_path / "m.npz"), out=out, tif=tif, meta=meta, batch=3, kw=kw))
        _run(script, world, extra_env={"TD_FAULT_TILE": fault})
        outs[name] = {f: open(os.path.join(out, "img", f), "rb").read() for f in sorted(os.listdir(os.path.join(out, "img")))}
    assert len(outs["one"]) == 8 and f"Prediction_{bad}.json" not in outs["one"]
    for name in ("two_rank0", "two_rank0_pipelined", "two_local"):
        assert outs[name] == outs["one"], name                 # same names, same bytes
    assert len(outs["one_other"]) == 8 and f"Prediction_{ids[3]}.json" not in outs["one_other"]
    assert outs["two_rank0_other_rank"] == outs["one_other"]
