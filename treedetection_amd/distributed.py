"""Multi-GPU layout of the prediction stage: tiles shard across ranks, detections gather to rank 0.

The reference is single-device (TreeDetection/config.py:45-53). Tiles are independent (each forward sees one tile;
overlap is handled by the buffer + later stitching, preprocessing.py:60-65), so rank r simply takes tiles
``i ≡ r (mod world)`` of the ordered tile list and every rank keeps a full weight replica. The only exchange step is
the gather of the per-tile detections (count, boxes, scores, 28x28 mask probabilities: FIXED-SHAPE fp32 tensors,
[B,100,28,28] + boxes + scores + counts ≈ 2.5 MB per 8-tile batch and rank, whatever the detection count) to
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
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])     # never "the default GPU": every rank names its own
        else:
            dist.barrier()


def single_node() -> bool:
    """True when the launcher says every rank runs on THIS node (torch.distributed.run sets LOCAL_WORLD_SIZE); unknown
    (srun / mpirun give the global rank only) counts as False."""
    import os
    if world() == 1:
        return True
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    return lws is not None and int(lws) == world()


def local_world() -> int:
    """Ranks that share this node's host cores: LOCAL_WORLD_SIZE where the launcher sets it, else the ranks that fit the
    node's GPUs (srun / mpirun), 1 in a single process."""
    import os
    if world() == 1:
        return 1
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    if lws is not None:
        return max(1, int(lws))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return max(1, min(world(), n) if n else world())


def output_is_shared(folder: str) -> bool:
    """Proof that rank 0 — which alone stitches after a "local" run — can read what EVERY other rank writes into ``folder``:
    each rank drops its own token file ``<nonce>.<rank>`` (the nonce comes from rank 0), all meet, rank 0 looks for every
    token and broadcasts the verdict. Collective: all ranks call it, each on its own GPU under nccl (the caller binds the
    device first: :func:`bind_device`). The decision is only as good as this probe — a folder that is shared but slow to
    publish new entries (some NFS settings) reads as "not shared" and the run falls back to the "rank0" epilogue."""
    import os
    import uuid
    if world() == 1:
        return True
    me = rank()
    nonce = broadcast_object(f".td_shared_{uuid.uuid4().hex}" if me == 0 else None, 0)
    wrote = True
    try:
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, f"{nonce}.{me}"), "w") as f:
            f.write("x")
            f.flush()
            os.fsync(f.fileno())
    except OSError:
        wrote = False
    wrote = all_ok(wrote)          # an all-reduce: returns only after EVERY rank has written (or failed to)
    seen = False
    if me == 0 and wrote:
        try:
            names = set(os.listdir(folder))
        except OSError:
            names = set()
        seen = all(f"{nonce}.{r}" in names for r in range(world()))
    ok = bool(broadcast_object(seen if me == 0 else None, 0))
    try:
        os.remove(os.path.join(folder, f"{nonce}.{me}"))
    except OSError:
        pass
    barrier()                      # nobody goes on to list the folder before the tokens are gone
    return ok


def all_ok(ok: bool) -> bool:
    """True iff ``ok`` holds on EVERY rank (one tiny all-reduce; the collective calls of the sharded predictor only
    stay paired if all ranks take the same decision about an image or a round)."""
    if world() == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=collective_device())
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def broadcast_object(obj, src: int = 0):
    """Small picklable object from ``src`` to every rank."""
    if world() == 1:
        return obj
    box = [obj if rank() == src else None]
    if dist.get_backend() == "nccl":
        dist.broadcast_object_list(box, src=src, device=collective_device())
    else:
        dist.broadcast_object_list(box, src=src)
    return box[0]


def local_device(configured) -> int:
    """GPU index of THIS rank. A single process keeps ``config['device']`` (reference config.py:112-142). Under
    ``torch.distributed`` the shared config.yml cannot name one device per rank, so the index comes from LOCAL_RANK
    (set by ``torch.distributed.run``); launchers that do not set it (srun / mpirun, several nodes) give the GLOBAL
    rank only, which is out of range on every node but the first, so the fallback is ``rank % device_count``. With the
    nccl backend two ranks of a node on one GPU abort inside RCCL ("Duplicate GPU detected"), so that case is refused
    here with a readable message. (gloo rehearsal runs may share a GPU.)"""
    import os
    if world() == 1:
        return int(configured)
    n = torch.cuda.device_count()
    if n < 1:
        raise RuntimeError("no GPU visible to this rank")
    lr = os.environ.get("LOCAL_RANK")
    idx = int(lr) if lr is not None else rank() % n
    if dist.get_backend() == "nccl":
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", min(world(), n) if lr is None else world()))
        if lws > n or idx >= n:
            raise RuntimeError(f"{lws} ranks on this node (local rank {idx}) but only {n} GPUs: one process per GPU (nccl backend)")
        return idx
    return idx % n


_bound_device: Optional[int] = None


def bind_device(configured) -> Optional[int]:
    """Make THIS rank's GPU the current device BEFORE its first collective (world > 1 only; returns the index).

    With the nccl backend ``broadcast_object_list`` / ``barrier`` / ``all_reduce`` place their tensors on
    ``torch.cuda.current_device()``: if every rank still sits on cuda:0 when ``preprocess_files`` broadcasts, RCCL
    aborts with "Duplicate GPU detected" (or rank > 0 opens a context on GPU 0). ``process_files`` / ``predict_tiles``
    / ``preprocess_files`` call this first; the Engine constructor later selects the same index."""
    global _bound_device
    if world() == 1 or str(configured) == "cpu":
        return None
    idx = local_device(configured)
    if torch.cuda.is_available():
        torch.cuda.set_device(idx)
    _bound_device = idx
    return idx


def collective_device() -> torch.device:
    """Device the small control collectives (flags, pickled objects) put their tensors on: the rank's own GPU under
    nccl (RCCL moves device memory only), the host otherwise."""
    if world() > 1 and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


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
