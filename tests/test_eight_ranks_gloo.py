"""The 8-rank structure of BASELINE configs[3] / configs[4] rehearsed WITHOUT 8 GPUs: eight gloo ranks on
the CPU walk image folders of 7, 8 and 19 images with unequal tile counts through ``detection.predict_tiles`` (the GPU
predictor replaced by the stand-in of tests/test_image_sharding.py, which writes prediction files of the real schema).
Pinned here: ownership follows the TILE COUNTS (longest-processing-time first), every image is predicted exactly once, the
number of collectives per ``predict_on_model`` is the same constant for 8 and for 19 images (O(1)), fewer images than ranks
fall back to tile-level sharding, a rank whose walk breaks outside the per-image try still enters every collective (nobody
hangs, everybody raises, the resume file lists only what was really walked); tile-level sharding arithmetic for 10 001 tiles
at batch 32 over 8 ranks; device binding with LOCAL_WORLD_SIZE = 8. What stays hardware-only: RCCL itself on 8 GPUs."""
import json
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tests.test_image_sharding import FakePredictor, _config, _count_collectives, _free_port, _layer_rows  # noqa: E402
from treedetection_amd import detection  # noqa: E402
from treedetection_amd import distributed as D  # noqa: E402

WORLD = 8


def _make_uneven_folder(root, n_images):
    """Rasters of different sizes: 4, 6, 9, 12 or 16 tiles of 20 m (100 px) each; sorted names do NOT follow the sizes."""
    from treedetection_amd.geotiff import write_geotiff
    from treedetection_amd.preprocessing import tile_single_file
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    rng = np.random.default_rng(3)
    shapes = [(200, 200), (200, 300), (300, 300), (300, 400), (400, 400)]
    counts = {}
    for k in range(n_images):
        h, w = shapes[(k * 3 + k // 5) % len(shapes)]
        name = f"img{k:02d}"
        tif = os.path.join(root, "rgb", f"{name}.tif")
        write_geotiff(tif, rng.integers(0, 255, (3, h, w), dtype=np.uint8), (0.2, 0, 1000.0 + 100 * k, 0, -0.2, 2000.0 + 0.2 * h), 25832)
        tile_single_file(tif, os.path.join(root, "tiles"), buffer=5, tile_width=20, tile_height=20)
        counts[f"{name}.tif"] = len(json.load(open(os.path.join(root, "tiles", f"{name}.json"))))
    return counts


class BrokenWalkPredictor(FakePredictor):
    """The stand-in again; the test makes rank 2's whole walk fail OUTSIDE the per-image try (see _worker)."""


def _worker(rank, world, port, root, q, break_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        detection.Predictor = FakePredictor
        detection.setup_model_cfg = lambda **kw: None
        detection.D.bind_device = lambda configured: None            # no GPU in the CPU suite
        detection.D.local_device = lambda configured: 0
        if break_rank is not None and rank == break_rank:
            real = detection.walk_images

            def broken(*a, **k):
                raise OSError("stitching folder of this rank's mount is read-only")
            detection.walk_images = broken
        config = _config(root, rank)
        counts = _count_collectives()
        err = None
        try:
            detection.predict_tiles(config)
        except Exception as e:                                        # noqa: BLE001 — the test asserts on what was raised
            err = f"{type(e).__name__}: {e}"
        q.put((rank, dict(counts), [c[1] for c in FakePredictor.calls if c[0] == "submit"], err))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(root, break_rank=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, root, q, break_rank)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return got


def test_assign_images_follows_tile_counts_not_file_sizes(tmp_path):
    """DEFLATE / LZW rasters of equal geometry differ in bytes; the owner map must follow the tile count."""
    tiles = tmp_path / "tiles"
    tiles.mkdir()
    paths = []
    for k, (nbytes, ntiles) in enumerate([(10, 400), (5000, 400), (20, 400), (9000, 400), (7, 100), (8000, 100), (1, 100), (1, 100)]):
        p = tmp_path / f"{k}.tif"
        p.write_bytes(b"x" * nbytes)
        (tiles / f"{k}.json").write_text(json.dumps({f"t{i}": {} for i in range(ntiles)}))
        paths.append(str(p))
    assert detection.image_weights(paths, str(tiles)) == [400, 400, 400, 400, 100, 100, 100, 100]
    own = detection.assign_images(paths, 4, str(tiles))
    load = [sum(w for w, o in zip([400] * 4 + [100] * 4, own) if o == r) for r in range(4)]
    assert load == [500, 500, 500, 500], (own, load)                          # by file size rank 1 would own 5000 + 8000 bytes of "work"
    by_size = detection.assign_images(paths, 4)
    assert by_size != own
    (tiles / "3.json").unlink()                                               # one tile list missing → ONE unit for the whole list: file sizes
    assert detection.image_weights(paths, str(tiles)) == [os.path.getsize(p) for p in paths]


def test_eight_ranks_walk_8_and_19_images_with_the_same_collectives(tmp_path):
    results = {}
    for n in (8, 19):
        root = str(tmp_path / f"n{n}")
        counts = _make_uneven_folder(root, n)
        open(os.path.join(root, "model.npz"), "wb").close()
        results[n] = (_run(root), root, counts)
    for n, (got, root, counts) in results.items():
        assert all(err is None for *_, err in got), got
        subs = {r: s for r, _, s, _ in got}
        # every image exactly once, by the rank detection.assign_images names (tile counts, longest first)
        paths = sorted(os.path.join(root, "rgb", f) for f in counts)
        own = detection.assign_images(paths, WORLD, os.path.join(root, "tiles"))
        for r in range(WORLD):
            assert subs[r] == [os.path.basename(p) for p, o in zip(paths, own) if o == r], (n, r)
        assert sorted(f for s in subs.values() for f in s) == sorted(counts)
        load = [sum(counts[f] for f in subs[r]) for r in range(WORLD)]
        assert max(load) - min(load) <= max(counts.values()), (n, load)
        if n == 8:
            assert all(len(s) == 1 for s in subs.values())
        # every image's tile files and its stitched layer exist; the resume files list all of them
        for f, nt in counts.items():
            stem = f[:-4]
            assert len(os.listdir(os.path.join(root, "out", "predictions", stem))) == nt
            assert len(_layer_rows(os.path.join(root, "out", "geojson_predictions", f"{stem}.gpkg"))) > 0
        rec = yaml.safe_load(open(os.path.join(root, "out", "predictions", "prediction_recovery.yaml")))
        assert sorted(os.path.basename(k) for k in rec["files"]) == sorted(counts)
        st = yaml.safe_load(open(os.path.join(root, "out", "geojson_predictions", "stitching_recovery.yaml")))
        assert len(st["completed_files"]) == n
    # O(1) collectives per predict_on_model: the same constant for 8 and for 19 images, on every rank
    c8 = {r: c for r, c, _, _ in results[8][0]}
    c19 = {r: c for r, c, _, _ in results[19][0]}
    assert c8 == c19, (c8, c19)
    assert c8[0] == {"broadcast_object_list": 1, "gather_object": 1, "all_reduce": 1, "barrier": 1}, c8[0]
    assert all(c8[r] == c8[0] for r in range(WORLD))


def test_fewer_images_than_ranks_fall_back_to_tile_sharding():
    assert detection.resolve_shard_by({}, WORLD, "local", 7) == "tile"            # configs[3]: one mosaic, 8 ranks
    assert detection.resolve_shard_by({}, WORLD, "local", 8) == "image"
    assert detection.resolve_shard_by({"shard_by": "image"}, WORLD, "local", 7) == "image"      # explicit: one rank idles
    own = detection.assign_images([f"/nowhere/{k}.tif" for k in range(7)], WORLD)
    assert sorted(own) == list(range(7))                                          # one image each, rank 7 idle


def test_a_rank_whose_walk_breaks_still_enters_every_collective(tmp_path):
    """rank 2's walk raises outside the per-image try. Every rank must return (no hang in gather_object / all_reduce),
    every rank raises, and the resume file lists only the images that were really walked — the next run predicts the rest."""
    root = str(tmp_path / "broken")
    counts = _make_uneven_folder(root, 16)
    open(os.path.join(root, "model.npz"), "wb").close()
    got = _run(root, break_rank=2)
    errs = {r: e for r, _, _, e in got}
    assert all(e is not None for e in errs.values()), errs
    assert "read-only" in errs[2] and all("walk failed" in errs[r] or "read-only" in errs[r] for r in errs)
    paths = sorted(os.path.join(root, "rgb", f) for f in counts)
    own = detection.assign_images(paths, WORLD, os.path.join(root, "tiles"))
    lost = sorted(os.path.basename(p) for p, o in zip(paths, own) if o == 2)
    assert lost
    rec = yaml.safe_load(open(os.path.join(root, "out", "predictions", "prediction_recovery.yaml")))
    listed = sorted(os.path.basename(k) for k in rec["files"])
    assert listed == sorted(set(counts) - set(lost)), (listed, lost)
    c = {r: cc for r, cc, _, _ in got}
    assert all(c[r] == {"broadcast_object_list": 1, "gather_object": 1, "all_reduce": 1, "barrier": 1} for r in range(WORLD)), c


def test_tile_sharding_arithmetic_for_10001_tiles_at_batch_32():
    n, B = 10001, 32
    shards = [D.shard_indices(n, r, WORLD) for r in range(WORLD)]
    assert sorted(i for s in shards for i in s) == list(range(n))                  # every tile exactly once
    assert [len(s) for s in shards] == [1251] + [1250] * 7
    assert all(s == list(range(r, n, WORLD)) for r, s in enumerate(shards))        # i = r (mod W): neighbours go to different GPUs
    rounds = D.padded_rounds(n, B, WORLD)
    assert rounds == 40                                                            # ceil(1251 / 32); every rank runs all 40
    for r, s in enumerate(shards):
        full, tail = divmod(len(s), B)
        assert full == 39 and tail == (3 if r == 0 else 2)                         # the last round is a padded partial batch everywhere
    assert D.padded_rounds(10000, B, WORLD) == 40 and D.padded_rounds(10000, 8, WORLD) == 157
    assert D.padded_rounds(7, B, WORLD) == 1 and [len(D.shard_indices(7, r, WORLD)) for r in range(WORLD)] == [1] * 7 + [0]


class _FakeDist:
    def __init__(self, rank, world, backend):
        self._r, self._w, self._b = rank, world, backend

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def get_rank(self):
        return self._r

    def get_world_size(self):
        return self._w

    def get_backend(self):
        return self._b


@pytest.mark.parametrize("rank", range(WORLD))
def test_eight_local_ranks_bind_eight_gpus(monkeypatch, rank):
    import torch
    monkeypatch.setattr(D, "dist", _FakeDist(rank, WORLD, "nccl"))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    chosen = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: chosen.append(i))
    monkeypatch.setenv("LOCAL_RANK", str(rank))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert D.local_device("0") == rank
    assert D.bind_device("0") == rank and chosen == [rank]
    assert D.local_world() == 8


def test_nine_local_ranks_on_eight_gpus_are_refused(monkeypatch):
    import torch
    monkeypatch.setattr(D, "dist", _FakeDist(8, 9, "nccl"))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_RANK", "8")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "9")
    with pytest.raises(RuntimeError, match="one process per GPU"):
        D.local_device("0")
