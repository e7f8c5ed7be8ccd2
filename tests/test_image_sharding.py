"""Image-level sharding of ``predict_on_model`` (treedetection_amd/detection.py) on CPU: 2 gloo ranks, the GPU predictor
replaced by a stand-in that writes prediction files of the real schema. What is pinned here: every image is owned by
exactly one rank, an unreadable image is logged by its owner and the walk goes on everywhere (reference
detection.py:117-120), the owner stitches its images while it walks (layers + resume file equal a single-process run's),
and the number of collectives per ``predict_on_model`` is O(1) — the same for 6 and for 12 images."""
import json
import logging
import os
import socket
import sqlite3
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from treedetection_amd import detection  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeHandle:
    dropped = []

    def __init__(self, n):
        self.n = n

    def result(self):
        return []


class FakePredictor:
    """Writes, for every tile of an image, one crown (a square around the tile centre, well inside the edge filter's box) with a
    score derived from the tile id — the same bytes whichever rank owns the image. ``bad.tif`` cannot be opened."""
    calls = []

    def __init__(self, cfg, device_type="0", max_batch_size=5, output_dir="./output", exclude_vars=None, sharded_epilogue="rank0", **kw):
        self.output_dir, self.sharded_epilogue = output_dir, sharded_epilogue
        FakePredictor.calls.append(("init", sharded_epilogue))

    def submit(self, tifpath, tilepath, whole_image=False):
        assert whole_image or not (dist.is_initialized() and dist.get_world_size() > 1)
        if os.path.basename(tifpath).startswith("bad"):
            raise ValueError(f"cannot open {tifpath}")
        sub = os.path.join(self.output_dir, os.path.basename(tifpath).replace(".tif", ""))
        os.makedirs(sub, exist_ok=True)
        meta = json.load(open(tilepath))
        for tile_id, td in meta.items():
            x0, y0, x1, y1 = td["bounds"][:4]
            cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
            ring = [[cx - 3, cy - 3], [cx + 3, cy - 3], [cx + 3, cy + 3], [cx - 3, cy + 3], [cx - 3, cy - 3]]
            score = 0.31 + (sum(map(ord, tile_id)) % 60) / 100
            with open(os.path.join(sub, f"Prediction_{tile_id}.json"), "w") as f:
                json.dump([{"image_id": tifpath, "category_id": 0, "score": score, "polygon_coords": [ring]}], f)
        FakePredictor.calls.append(("submit", os.path.basename(tifpath)))
        return FakeHandle(len(meta))

    def __call__(self, tifpath, tilepath):
        raise AssertionError("image-level sharding must not enter the tile-sharded collective path")

    def close(self):
        pass


def _make_folder(root, names):
    from treedetection_amd.geotiff import write_geotiff
    from treedetection_amd.preprocessing import tile_single_file
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    rng = np.random.default_rng(1)
    for k, name in enumerate(names):
        tif = os.path.join(root, "rgb", f"{name}.tif")
        write_geotiff(tif, rng.integers(0, 255, (3, 200, 200), dtype=np.uint8), (0.2, 0, 1000.0 + 40 * k, 0, -0.2, 2040.0), 25832)
        tile_single_file(tif, os.path.join(root, "tiles"), buffer=5, tile_width=20, tile_height=20)


def _config(root, rank):
    log = logging.getLogger(f"shard-r{rank}")
    log.setLevel(logging.INFO)
    h = logging.FileHandler(os.path.join(root, f"log_rank{rank}.txt"))
    log.addHandler(h)
    return {"image_directory": os.path.join(root, "rgb"), "merged_path": "merged", "tiles_path": os.path.join(root, "tiles"),
            "output_directory": os.path.join(root, "out"), "device": "0", "simplify_tolerance": 0.2, "num_workers": 2,
            "batch_size": 4, "logger": log, "combined_model": os.path.join(root, "model.npz"), "sharded_epilogue": "local"}


def _count_collectives():
    counts = {}
    for name in ("broadcast_object_list", "gather_object", "all_reduce", "barrier", "gather", "all_gather", "broadcast"):
        orig = getattr(dist, name)

        def wrapped(*a, _orig=orig, _name=name, **k):
            counts[_name] = counts.get(_name, 0) + 1
            return _orig(*a, **k)
        setattr(dist, name, wrapped)
    return counts


def _worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        detection.Predictor = FakePredictor
        detection.setup_model_cfg = lambda **kw: None
        detection.D.bind_device = lambda configured: None            # no GPU in the CPU suite: the ranks have no device to bind
        detection.D.local_device = lambda configured: 0
        config = _config(root, rank)
        counts = _count_collectives() if world > 1 else {}
        detection.predict_tiles(config)
        q.put((rank, dict(counts), [c[1] for c in FakePredictor.calls if c[0] == "submit"], [c[1] for c in FakePredictor.calls if c[0] == "init"]))
        if world > 1:
            dist.barrier()
    finally:
        if world > 1:
            dist.destroy_process_group()


def _run(world, root):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return got


def _layer_rows(path):
    con = sqlite3.connect(path)
    try:
        table = con.execute("SELECT table_name FROM gpkg_contents").fetchone()[0]
        return con.execute(f'SELECT * FROM "{table}" ORDER BY fid').fetchall()      # (an empty layer has no attribute columns)
    finally:
        con.close()


def test_assign_images_is_round_robin_for_equal_images_and_balances_sizes(tmp_path):
    paths = []
    for k, size in enumerate([100, 100, 100, 100, 100, 100, 10, 10, 10]):
        p = tmp_path / f"{k}.tif"
        p.write_bytes(b"x" * size)
        paths.append(str(p))
    own = detection.assign_images(paths[:6], 2)
    assert own == [0, 1, 0, 1, 0, 1]
    own = detection.assign_images(paths, 4)
    load = [sum(os.path.getsize(p) for p, o in zip(paths, own) if o == r) for r in range(4)]
    assert max(load) - min(load) <= 100 and sorted(set(own)) == [0, 1, 2, 3]
    assert detection.assign_images(paths, 4) == own                                   # deterministic
    assert detection.resolve_shard_by({}, 1, "rank0", 5) == "single"
    assert detection.resolve_shard_by({}, 4, "local", 5) == "image" and detection.resolve_shard_by({}, 8, "local", 5) == "tile"
    assert detection.resolve_shard_by({}, 4, "rank0", 50) == "tile"
    assert detection.resolve_shard_by({"shard_by": "tile"}, 4, "local", 50) == "tile"


def test_two_ranks_own_whole_images_with_o1_collectives(tmp_path):
    results = {}
    for label, names, world in (("one", ["a1", "a2", "a3", "bad4", "a5", "a6"], 1), ("two", ["a1", "a2", "a3", "bad4", "a5", "a6"], 2),
                                ("two12", [f"b{k:02d}" for k in range(12)], 2)):
        root = str(tmp_path / label)
        _make_folder(root, names)
        open(os.path.join(root, "model.npz"), "wb").close()
        results[label] = (_run(world, root), root)
    got2, root2 = results["two"]
    got1, root1 = results["one"]
    # every image submitted exactly once, by its owner (equal rasters: round-robin over the sorted list); the unreadable one too
    subs = {r: s for r, _, s, _ in got2}
    assert sorted(subs[0] + subs[1]) == ["a1.tif", "a2.tif", "a3.tif", "a5.tif", "a6.tif"]
    assert subs[0] == ["a1.tif", "a3.tif", "a6.tif"] and subs[1] == ["a2.tif", "a5.tif"]         # bad4 was rank 1's: raised, logged, walk went on
    assert "Error processing" in open(os.path.join(root2, "log_rank1.txt")).read()
    assert "bad4.tif" in open(os.path.join(root2, "log_rank1.txt")).read()
    assert all(init == ["local"] for _, _, _, init in got2)
    # collectives per predict_on_model: O(1) — identical for 6 and for 12 images, and small
    c6 = {r: c for r, c, _, _ in got2}
    c12 = {r: c for r, c, _, _ in results["two12"][0]}
    assert c6 == c12, (c6, c12)
    assert c6[0] == {"broadcast_object_list": 1, "gather_object": 1, "all_reduce": 1, "barrier": 1}, c6[0]
    # same prediction files, same layers, same resume files as the single-process run
    for sub in ("a1", "a2", "a3", "a5", "a6"):
        f1 = sorted(os.listdir(os.path.join(root1, "out", "predictions", sub)))
        f2 = sorted(os.listdir(os.path.join(root2, "out", "predictions", sub)))
        assert f1 == f2 and len(f1) == 4
        for f in f1:
            assert open(os.path.join(root1, "out", "predictions", sub, f), "rb").read().replace(root1.encode(), b"") == \
                   open(os.path.join(root2, "out", "predictions", sub, f), "rb").read().replace(root2.encode(), b"")
    for name in ("a1", "a2", "a3", "bad4", "a5", "a6"):          # bad4: stitched afterwards by rank 0 (an empty layer), as the reference does
        l1, l2 = (os.path.join(r, "out", "geojson_predictions", f"{name}.gpkg") for r in (root1, root2))
        assert _layer_rows(l1) == _layer_rows(l2) and (len(_layer_rows(l1)) > 0) == (name != "bad4")
    s1, s2 = (yaml.safe_load(open(os.path.join(r, "out", "geojson_predictions", "stitching_recovery.yaml"))) for r in (root1, root2))
    assert s1 == s2 and len(s1["completed_files"]) == 6
    p1, p2 = (yaml.safe_load(open(os.path.join(r, "out", "predictions", "prediction_recovery.yaml"))) for r in (root1, root2))
    assert sorted(os.path.basename(k) for k in p1["files"]) == sorted(os.path.basename(k) for k in p2["files"])


def test_walk_images_stitches_as_it_goes_and_respects_the_resume_file(tmp_path):
    """detection.walk_images, single process: every finished image is handed to the eager stitcher (a failed one is not), the
    layers exist when the walk returns, and a folder the stitching resume file already lists is neither rebuilt by the eager
    stitcher nor by the pass afterwards (reference helpers.py:566-571 skips completed folders)."""
    from treedetection_amd.recoveries import save_stitching_recovery
    from treedetection_amd.stitching import process_and_stitch_predictions
    root = str(tmp_path)
    names = ["c1", "bad2", "c3"]
    _make_folder(root, names)
    cfg = _config(root, 0)
    pred = FakePredictor(None, output_dir=os.path.join(root, "out", "predictions"), sharded_epilogue="rank0")
    gp = os.path.join(root, "out", "gpkg")
    os.makedirs(gp)
    save_stitching_recovery(gp, ["c3.json"], None)                       # c3 counts as stitched already (no layer on disk)
    paths = [os.path.join(root, "rgb", f"{n}.tif") for n in names]
    rep = detection.walk_images(cfg, pred, paths, cfg["tiles_path"], pred.output_dir, chain=True, stitch_to=gp)
    assert [os.path.basename(p) for p in rep["done"]] == ["c1.tif", "c3.tif"] and [os.path.basename(p) for p in rep["failed"]] == ["bad2.tif"]
    assert rep["stitched"] == ["c1.json"]                                # bad2 failed, c3 was listed as complete
    assert os.path.exists(os.path.join(gp, "c1.gpkg")) and not os.path.exists(os.path.join(gp, "c3.gpkg"))
    assert len(_layer_rows(os.path.join(gp, "c1.gpkg"))) == 4
    # what predict_on_model does next: merge the report into the resume file, then the leftover pass handles bad2 only
    save_stitching_recovery(gp, ["c3", "c1.json"], None)
    process_and_stitch_predictions(cfg["tiles_path"], pred.output_dir, gp, max_workers=2, shift=1, simplify_tolerance=0.2, logger=cfg["logger"])
    assert os.path.exists(os.path.join(gp, "bad2.gpkg")) and _layer_rows(os.path.join(gp, "bad2.gpkg")) == []
    assert not os.path.exists(os.path.join(gp, "c3.gpkg"))
    assert yaml.safe_load(open(os.path.join(gp, "stitching_recovery.yaml")))["completed_files"] == ["bad2", "c1", "c3"]
    # without stitch_to nothing is stitched; chain=False goes through __call__ (which this stand-in refuses: tile-sharded path)
    rep2 = detection.walk_images(cfg, pred, paths[:1], cfg["tiles_path"], pred.output_dir, chain=True, stitch_to=None)
    assert rep2["stitched"] == [] and rep2["stitch_seconds"] == 0.0
    rep3 = detection.walk_images(cfg, pred, paths[:1], cfg["tiles_path"], pred.output_dir, chain=False)
    assert rep3["done"] == [] and len(rep3["failed"]) == 1               # logged, walk continued
    assert detection.resolve_shard_by({"shard_by": "image"}, 4, "local", 2) == "image"      # explicit: some ranks idle
    assert detection.resolve_shard_by({"shard_by": "image"}, 4, "rank0", 9) == "tile"       # needs the local epilogue
    with __import__("pytest").raises(ValueError):
        detection.resolve_shard_by({"shard_by": "rows"}, 2, "local", 9)
    assert detection.engine_batch_size({"precision": "fp16"}, 8) == 8 and detection.engine_batch_size({"precision": "fp16", "fp16_min_batch": 32}, 8) == 32
    assert detection.engine_batch_size({"precision": "fp32", "fp16_min_batch": 32}, 8) == 8
