"""N > 1 Predictor path on ONE GPU: 2 ranks (gloo rendezvous, both on cuda:0) shard the tiles of an image, gather the
detections to rank 0, and rank 0 writes every Prediction_*.json — the files must equal the single-process run's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

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
pred = Predictor(setup_model_cfg(update_model="x", device="0"), device_type="0", max_batch_size=3, output_dir={out!r}, state_dict=sd)
pred({tif!r}, {meta!r})
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


def test_two_ranks_write_the_same_files_as_one(tmp_path):
    from treedetection_amd.preprocessing import tile_single_file
    np.savez(tmp_path / "m.npz", **make_synthetic_state_dict(50, seed=3, width_div=2))
    rgb, _ = make_tile(100, 600)
    tif = str(tmp_path / "img.tif")
    write_geotiff(tif, np.ascontiguousarray(rgb.transpose(2, 0, 1)), (0.2, 0, 0, 0, -0.2, 120.0), 25832)
    tile_single_file(tif, str(tmp_path / "tiles"), buffer=20, tile_width=50, tile_height=50)
    meta = str(tmp_path / "tiles" / "img.json")
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"out{world}")
        script = tmp_path / f"w{world}.py"
        script.write_text(WORKER.format(root=ROOT, model=str(tmp_path / "m.npz"), out=out, tif=tif, meta=meta))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=300)
        else:
            subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                           check=True, env=env, timeout=300)
        outs[world] = {f: json.load(open(os.path.join(out, "img", f))) for f in sorted(os.listdir(os.path.join(out, "img")))}
    assert len(outs[1]) == 9 and sorted(outs[1]) == sorted(outs[2])
    total = 0
    for f in outs[1]:
        a, b = outs[1][f], outs[2][f]
        assert len(a) == len(b), f
        for ea, eb in zip(a, b):
            assert abs(ea["score"] - eb["score"]) <= 1e-6 and ea["polygon_coords"] == eb["polygon_coords"]
        total += len(a)
    assert total > 10
