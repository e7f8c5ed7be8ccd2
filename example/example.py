"""Runs the whole pipeline on the images of ``example/config.yml`` — the drop-in for the reference's ``example/example.py``
(TreeDetection: ``get_config`` → ``process_files``). One GPU:  ``python example/example.py``; one process per GPU of a node:
``python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 example/example.py``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import treedetection_amd as T  # noqa: E402

if __name__ == "__main__":
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                       # launched by torch.distributed.run: RCCL over xGMI, each rank binds its GPU (LOCAL_RANK)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # no collective runs while a rank walks its images (treedetection_amd/detection.py shards by whole image), so the closing
        # manifest gather must be allowed to wait for the slowest rank's walk: the 10-minute default would end a large job
        import datetime
        dist.init_process_group("nccl", timeout=datetime.timedelta(hours=6))
    config, _ = T.get_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config.yml"))
    # the three stages can also be called one by one, in this order: T.preprocess_files(config), T.predict_tiles(config),
    # T.postprocess_files(config)
    T.process_files(config)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
