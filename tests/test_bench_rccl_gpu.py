"""The RCCL leg of bench.py: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with the nccl backend
(= RCCL on ROCm), one rank per GPU — what the driver launches for SCALE_rNN.json. Needs two GPUs; on the one-GPU boxes of
this pool the test SKIPS (loudly: the reason names what was not exercised), and the gloo rehearsal of the same code path on
one GPU lives in tests/test_predictor_multirank_gpu.py. When it runs, the JSON line's `ranks` object is the evidence: world
size and backend as torch.distributed reports them, the device every rank bound (all-gathered), distinct GPUs."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, env_extra, port, detail=None):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--no-serial", "--no-r101", "--no-fp16-b32", "--no-fp16"]
    if detail:
        cmd += ["--detail", detail]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode != 0:       # the launcher's summary names only the first rank that died: keep every rank's stderr where a reader finds it
        log_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(log_dir):
            with open(os.path.join(log_dir, f"bench_launch_{nproc}ranks_{port}.stderr"), "w") as f:
                f.write(r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096      # the compact line is the LAST stdout line
    return json.loads(lines[0])


def test_bench_two_ranks_over_rccl():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"RCCL path NOT exercised: {n} GPU visible, bench.py --gpus 2 with backend nccl needs 2 (one rank per GPU)")
    line = _launch(2, {}, 29611)
    rk = line["ranks"]
    assert line["n_gpus"] == 2 and rk["world"] == 2 and rk["backend"] == "nccl"
    assert sorted(rk["devices"]) == [0, 1] and rk["distinct_gpus"] == 2
    assert rk["gather_bytes_per_step"] > 0 and line["value"] > 0 and line["scaling"] == "weak"


def test_bench_two_ranks_gloo_rehearsal_reports_what_the_collective_layer_saw(tmp_path):
    """The same launch with TD_BENCH_BACKEND=gloo and both ranks on this box's GPU: the N > 1 code path of bench.py (shard of the
    stream per rank, per-step gather to rank 0, barrier + max-over-ranks timing, the `ranks` object) on hardware that has one GPU."""
    detail = str(tmp_path / "detail.json")
    line = _launch(2, {"TD_BENCH_BACKEND": "gloo"}, 29612, detail)
    rk = line["ranks"]
    assert line["n_gpus"] == 2 and rk["world"] == 2 and rk["backend"] == "gloo" and len(rk["devices"]) == 2
    assert line["value"] > 0 and line["steps"] == 4 and line["timed_steps"] >= 4
    assert "roofline" in line and line["roofline"]["frac"] <= 1.0
    full = json.load(open(detail))                      # the per-rank records live in the detail file
    assert len(full["ranks"]["ranks"]) == 2 and {r["rank"] for r in full["ranks"]["ranks"]} == {0, 1}
