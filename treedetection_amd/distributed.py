"""Multi-GPU layout of the prediction stage: tiles shard across ranks, detections gather to rank 0.

The reference is single-device (TreeDetection/config.py:45-53). Tiles are independent (each forward sees one tile;
overlap is handled by the buffer + later stitching, preprocessing.py:60-65), so rank r simply takes tiles
``i ≡ r (mod world)`` of the ordered tile list and every rank keeps a full weight replica. The only exchange step is
the gather of the per-tile detections (count, boxes, scores, 28x28 mask probabilities ≈ 3.2 KB per detection) to
rank 0, which pastes / traces / writes the ``Prediction_*.json`` files — ``torch.distributed.gather`` over RCCL
(backend "nccl") on GPUs, gloo in the CPU tests. No collective touches the forward itself.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

GATHER_KEYS = ("count", "boxes", "scores", "mask_probs")


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier() -> None:
    """All ranks meet here (no-op in a single process). Used around the file-system stages of ``process_files``:
    rank 0 alone writes / deletes the shared folders, the others wait."""
    if world() > 1:
        dist.barrier()


def all_ok(ok: bool) -> bool:
    """True iff ``ok`` holds on EVERY rank (one tiny all-reduce; the collective calls of the sharded predictor only
    stay paired if all ranks take the same decision about an image or a round)."""
    if world() == 1:
        return bool(ok)
    dev = "cpu"
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def broadcast_object(obj, src: int = 0):
    """Small picklable object from ``src`` to every rank."""
    if world() == 1:
        return obj
    box = [obj if rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def local_device(configured) -> int:
    """GPU index of THIS rank. A single process keeps ``config['device']`` (reference config.py:112-142). Under
    ``torch.distributed`` the shared config.yml cannot name one device per rank, so the index comes from LOCAL_RANK
    (set by ``torch.distributed.run``), else ``rank % device_count``; with the nccl backend two ranks of a node on
    one GPU abort inside RCCL ("Duplicate GPU detected"), so that case is refused here with a readable message.
    (gloo rehearsal runs may share a GPU.)"""
    import os
    if world() == 1:
        return int(configured)
    n = torch.cuda.device_count()
    if n < 1:
        raise RuntimeError("no GPU visible to this rank")
    lr = os.environ.get("LOCAL_RANK")
    idx = int(lr) if lr is not None else rank()
    if dist.get_backend() == "nccl":
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", world()))
        if lws > n:
            raise RuntimeError(f"{lws} ranks on this node but only {n} GPUs: one process per GPU (nccl backend)")
        return idx
    return idx % n


def shard_indices(n: int, r: Optional[int] = None, w: Optional[int] = None) -> List[int]:
    """Indices of the ordered tile list that rank r processes (round-robin keeps per-rank work even)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return list(range(r, n, w))


def padded_rounds(n: int, batch: int, w: Optional[int] = None) -> int:
    """Number of batches EVERY rank must run so that the collective calls line up (ranks with fewer tiles pad)."""
    w = world() if w is None else w
    per_rank = (n + w - 1) // w
    return (per_rank + batch - 1) // batch


def gather_detections(out: Dict[str, torch.Tensor], dst: int = 0) -> Optional[List[Dict[str, torch.Tensor]]]:
    """Gather the fixed-shape detection tensors of one batch from every rank to ``dst``.

    Returns on ``dst`` a list (one entry per rank) of dicts with the GATHER_KEYS tensors; None elsewhere.
    With a single process returns ``[out]``."""
    if world() == 1:
        return [{k: out[k] for k in GATHER_KEYS}]
    me = rank()
    res = None
    if me == dst:
        res = [dict() for _ in range(world())]
    host_backend = dist.get_backend() != "nccl"      # gloo (CPU tests, single-GPU rehearsal) moves host tensors
    for k in GATHER_KEYS:
        t = out[k].contiguous()
        if host_backend and t.is_cuda:
            t = t.cpu()
        lst = [torch.empty_like(t) for _ in range(world())] if me == dst else None
        dist.gather(t, lst, dst=dst)
        if me == dst:
            for i, g in enumerate(lst):
                res[i][k] = g
    return res


def gather_objects(obj, dst: int = 0):
    """Small picklable per-rank payloads (tile ids / sizes / transforms of the batch) to ``dst``."""
    if world() == 1:
        return [obj]
    lst = [None] * world() if rank() == dst else None
    dist.gather_object(obj, lst, dst=dst)
    return lst
