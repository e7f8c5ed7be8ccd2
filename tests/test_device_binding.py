"""Which GPU the control collectives of process_files / preprocess_files / predict_on_model run on under the nccl
backend: every rank must select ITS device (LOCAL_RANK, else rank % device_count) BEFORE its first
barrier / broadcast, and the collectives must name that device — RCCL aborts with "Duplicate GPU detected" when two
ranks of a node issue a collective from cuda:0. No GPU here: torch.distributed and torch.cuda are mocked and the test
records what each call would have used."""
import types

import pytest
import torch
import torch.distributed as dist

from treedetection_amd import distributed as D


class FakeNccl:
    """Stands in for an initialised nccl process group of `world` ranks; records the device of every collective."""

    def __init__(self, monkeypatch, rank, world, n_gpus, local_rank=None):
        self.current = 0
        self.calls = []
        monkeypatch.setattr(dist, "is_available", lambda: True)
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: world)
        monkeypatch.setattr(dist, "get_rank", lambda *a, **k: rank)
        monkeypatch.setattr(dist, "get_backend", lambda *a, **k: "nccl")
        monkeypatch.setattr(torch.cuda, "device_count", lambda: n_gpus)
        monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
        monkeypatch.setattr(torch.cuda, "set_device", self._set)
        monkeypatch.setattr(torch.cuda, "current_device", lambda: self.current)
        monkeypatch.setattr(dist, "barrier", self._barrier)
        monkeypatch.setattr(dist, "broadcast_object_list", self._bcast)
        if local_rank is None:
            monkeypatch.delenv("LOCAL_RANK", raising=False)
            monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
        else:
            monkeypatch.setenv("LOCAL_RANK", str(local_rank))
            monkeypatch.setenv("LOCAL_WORLD_SIZE", str(min(world, n_gpus)))

    def _set(self, idx):
        self.current = int(idx)
        self.calls.append(("set_device", int(idx)))

    def _barrier(self, group=None, async_op=False, device_ids=None):
        self.calls.append(("barrier", tuple(device_ids) if device_ids else None))

    def _bcast(self, box, src=0, group=None, device=None):
        self.calls.append(("broadcast", None if device is None else device.index))


@pytest.mark.parametrize("rank,local_rank,expect", [(0, 0, 0), (3, 3, 3), (5, None, 5), (11, None, 3), (9, 1, 1)])
def test_local_device_under_nccl(monkeypatch, rank, local_rank, expect):
    FakeNccl(monkeypatch, rank=rank, world=16, n_gpus=8, local_rank=local_rank)
    assert D.local_device("0") == expect          # the shared config.yml says "0" for everybody


def test_local_device_refuses_more_ranks_than_gpus(monkeypatch):
    FakeNccl(monkeypatch, rank=1, world=4, n_gpus=2, local_rank=1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    with pytest.raises(RuntimeError, match="one process per GPU"):
        D.local_device("0")


def test_control_collectives_run_on_the_ranks_own_gpu(monkeypatch):
    f = FakeNccl(monkeypatch, rank=5, world=8, n_gpus=8, local_rank=5)
    assert D.bind_device("0") == 5
    D.barrier()
    D.broadcast_object({"x": 1})
    assert D.collective_device() == torch.device("cuda", 5)
    assert f.calls == [("set_device", 5), ("barrier", (5,)), ("broadcast", 5)]


def test_setup_model_cfg_selects_the_local_gpu_when_sharded(monkeypatch):
    from treedetection_amd import config as C
    f = FakeNccl(monkeypatch, rank=2, world=4, n_gpus=4, local_rank=2)
    monkeypatch.setattr(C, "_cuda_available", lambda: True)
    cfg = C.setup_model_cfg(update_model="m.pth", device="0")
    assert cfg.MODEL.DEVICE_INDEX == 2 and ("set_device", 2) in f.calls and ("set_device", 0) not in f.calls


def test_process_files_binds_before_its_first_collective(monkeypatch, tmp_path):
    """process_files → preprocess_files: the device is selected before the broadcast of the tile list, and no
    collective of the stage runs on another device."""
    from treedetection_amd import detection as det
    f = FakeNccl(monkeypatch, rank=3, world=4, n_gpus=4, local_rank=3)
    (tmp_path / "img").mkdir()
    (tmp_path / "h").mkdir()
    monkeypatch.setattr(det, "predict_tiles", lambda cfg: None)
    monkeypatch.setattr(det, "postprocess_files", lambda cfg: None)
    monkeypatch.setattr(det, "cleanup_files", lambda cfg: None)
    monkeypatch.setattr(f, "_bcast", None)

    def bcast(box, src=0, group=None, device=None):
        f.calls.append(("broadcast", None if device is None else device.index))
        box[0] = (None, [])            # what rank 0 would have sent: no error, no images

    monkeypatch.setattr(dist, "broadcast_object_list", bcast)
    log = types.SimpleNamespace(info=lambda *a: None, debug=lambda *a: None, warning=lambda *a: None, error=lambda *a: None)
    config = {"logger": log, "device": "0", "image_directory": str(tmp_path / "img"), "height_data_path": str(tmp_path / "h"),
              "continue": str(tmp_path / "none.txt"), "use_overlap": False, "tiles_path": str(tmp_path / "tiles")}
    det.process_files(config)
    kinds = [c[0] for c in f.calls]
    first_collective = min(i for i, k in enumerate(kinds) if k in ("barrier", "broadcast"))
    assert "set_device" in kinds[:first_collective], f.calls
    assert all(c[1] in (3, (3,)) for c in f.calls), f.calls


def test_host_pools_are_sized_by_the_ranks_share_of_the_node(monkeypatch):
    """epilogue workers / window readers were sized per rank from the whole node. With 8 ranks on a node the plan is
    cores // 8 per rank (LOCAL_WORLD_SIZE; srun-style launches without it: the ranks that fit the node's GPUs)."""
    import os
    from treedetection_amd import prediction as P
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    FakeNccl(monkeypatch, rank=3, world=8, n_gpus=8, local_rank=3)
    assert D.local_world() == 8 and D.single_node()
    assert P.host_core_share() == 16
    assert max(2, min(16, P.host_core_share() - 2)) == 14 and max(2, min(8, P.host_core_share() // 2)) == 8
    # 16 ranks over two 8-GPU nodes, torchrun: 8 local ranks, not one node
    FakeNccl(monkeypatch, rank=11, world=16, n_gpus=8, local_rank=3)
    assert D.local_world() == 8 and not D.single_node() and P.host_core_share() == 16
    # srun / mpirun: no LOCAL_WORLD_SIZE → the ranks that fit the node's GPUs; "one node" is NOT assumed
    FakeNccl(monkeypatch, rank=11, world=16, n_gpus=8, local_rank=None)
    assert D.local_world() == 8 and not D.single_node()
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)))
    assert P.host_core_share() == 1                       # never zero
    # single process: everything
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    assert D.local_world() == 1 and P.host_core_share() == 8


def test_auto_epilogue_goes_local_only_where_rank0_can_read_the_files():
    """"auto" resolved to "local" from 4 ranks on whatever the node layout; on several nodes without a shared output
    folder rank 0 (which alone stitches) would never see the other nodes' tile files."""
    from treedetection_amd.prediction import resolve_sharded_epilogue
    assert resolve_sharded_epilogue(1) == "rank0" and resolve_sharded_epilogue(2) == "rank0"
    assert resolve_sharded_epilogue(8, "fp16", shared_output=True) == "local"
    assert resolve_sharded_epilogue(8, "fp16", shared_output=False) == "rank0"
    assert resolve_sharded_epilogue(16, "fp32", shared_output=False) == "rank0"
