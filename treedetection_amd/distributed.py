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
