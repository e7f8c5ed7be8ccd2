"""N > 1 path on CPU: 2 processes over gloo — tile sharding covers the list exactly once and the fixed-shape gather
delivers every rank's detections to rank 0 (the exchange step the bench and the Predictor run over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from treedetection_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_tiles, B, Dn = 11, 4, 5
        mine = D.shard_indices(n_tiles)
        rounds = D.padded_rounds(n_tiles, B)
        seen = []
        for r in range(rounds):
            idx = mine[r * B:(r + 1) * B]
            out = {"count": torch.zeros(B, dtype=torch.int32), "boxes": torch.zeros(B, Dn, 4),
                   "scores": torch.zeros(B, Dn), "mask_probs": torch.zeros(B, Dn, 28, 28)}
            for j, t in enumerate(idx):     # tile t yields (t % 3) detections, all stamped with t
                out["count"][j] = t % 3
                out["boxes"][j, : t % 3] = float(t)
                out["scores"][j, : t % 3] = float(t) / 100
                out["mask_probs"][j, : t % 3] = float(t)
            g = D.gather_detections(out, dst=0)
            metas = D.gather_objects([{"tile": t} for t in idx], dst=0)
            if rank == 0:
                assert len(g) == world and len(metas) == world
                for rk, (gd, ms) in enumerate(zip(g, metas)):
                    for j, m in enumerate(ms):
                        t = m["tile"]
                        assert int(gd["count"][j]) == t % 3
                        assert (gd["boxes"][j, : t % 3] == float(t)).all() and (gd["mask_probs"][j, : t % 3] == float(t)).all()
                        seen.append(t)
            else:
                assert g is None and metas is None
        if rank == 0:
            q.put(sorted(seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_two_ranks():
    assert D.shard_indices(7, 0, 2) == [0, 2, 4, 6] and D.shard_indices(7, 1, 2) == [1, 3, 5]
    assert D.padded_rounds(11, 4, 2) == 2 and D.padded_rounds(8, 4, 2) == 1 and D.padded_rounds(0, 4, 2) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert seen == list(range(11))      # every tile exactly once, all delivered to rank 0


def test_single_process_paths():
    out = {"count": torch.zeros(2, dtype=torch.int32), "boxes": torch.zeros(2, 3, 4), "scores": torch.zeros(2, 3),
           "mask_probs": torch.zeros(2, 3, 28, 28)}
    assert D.world() == 1 and D.rank() == 0
    assert D.gather_detections(out)[0]["boxes"] is out["boxes"]
    assert D.gather_objects([1]) == [[1]]


def _fs_worker(rank, world, port, root, q):
    """File-system stages of process_files under 2 ranks (CPU only): preprocess_files writes on rank 0 alone and
    hands every rank the same image list; the agreement helpers keep collective decisions paired."""
    import logging
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from treedetection_amd import detection
        config = {"image_directory": os.path.join(root, "rgb"), "height_data_path": os.path.join(root, "ndsm"),
                  "tiles_path": os.path.join(root, "tiles"), "output_directory": os.path.join(root, "out"),
                  "continue": os.path.join(root, "out", "continue.yml"), "use_overlap": False, "buffer": 10,
                  "tile_width": 40, "tile_height": 40, "parallel": False, "num_workers": 2, "merged_path": "merged",
                  "keep_intermediate": False, "logger": logging.getLogger(f"r{rank}")}
        images = detection.preprocess_files(config)
        # rank 0 has finished writing before any rank returns: the files are complete for everybody
        import json
        meta = json.load(open(os.path.join(root, "tiles", "7.json")))
        assert len(meta) == 4 and images == [os.path.join(root, "rgb", "7.tif")]
        assert D.all_ok(True) is True
        assert D.all_ok(rank != 1) is False             # one rank's failure is everybody's
        assert D.broadcast_object({"x": rank}) == {"x": 0}
        # "auto" epilogue: a folder every rank sees is proven shared (token file from rank 0); one that only rank 0 has is not
        assert D.output_is_shared(os.path.join(root, "out_shared")) is True
        assert not [f for f in os.listdir(os.path.join(root, "out_shared")) if f.startswith(".td_shared_")]
        assert D.output_is_shared(os.path.join(root, f"out_private_rank{rank}")) is False
        # a failure on the writing rank surfaces on every rank (nobody is left waiting in a collective)
        bad = dict(config, image_directory=os.path.join(root, "empty"))
        try:
            detection.preprocess_files(bad)
            raised = False
        except FileNotFoundError:
            raised = True
        assert raised
        # cleanup: one rank removes, after everybody is done; afterwards the folders are gone for all
        os.makedirs(config["output_directory"], exist_ok=True)
        D.barrier()
        if D.rank() == 0:
            detection.cleanup_files(config)
        D.barrier()
        assert not os.path.exists(config["tiles_path"])
        q.put((rank, len(meta)))
    finally:
        dist.destroy_process_group()


def test_filesystem_stages_two_ranks(tmp_path):
    import numpy as np
    from treedetection_amd.geotiff import write_geotiff
    root = str(tmp_path)
    for d in ("rgb", "ndsm", "empty"):
        os.makedirs(os.path.join(root, d))
    rng = np.random.default_rng(0)
    write_geotiff(os.path.join(root, "rgb", "7.tif"), rng.integers(0, 255, (3, 400, 400), dtype=np.uint8), (0.2, 0, 1000.0, 0, -0.2, 2080.0), 25832)
    write_geotiff(os.path.join(root, "ndsm", "7.tif"), rng.random((80, 80), dtype=np.float32), (1.0, 0, 1000.0, 0, -1.0, 2080.0), 25832)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got == [(0, 4), (1, 4)]


def test_local_device_mapping(monkeypatch):
    assert D.local_device("3") == 3          # single process: the configured device
