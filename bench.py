#!/usr/bin/env python3
"""Throughput bench of the predict_tiles hot path (BASELINE.json metric: tiles/sec on a fixed 1000x1000 RGB+nDSM
tile stream at 1/2/4/8 MI355X).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--batch` (8) synthetic 1000x1000 tiles, already resident in HBM as uint8 RGB (+ float32
nDSM side band), through the whole device path of the reference's model stage (TreeDetection/prediction.py:159-183):
band pick + Pillow-exact resize to 800x800 → Mask R-CNN R50-FPN forward (fp32 MFMA) → detections + 28x28 mask
probabilities + pasted bit masks in HBM. Tiles shard across ranks (weak scaling: every rank runs K steps of its own
batches); each step ends with the RCCL gather of the per-tile detections to rank 0 (the hand-off to stitching).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel family =
conv_igemm, MFMA-bound, timed live with HIP events on the forward's stream) and `cpu_baseline` (the torch-CPU oracle
on a bounded sample of the same stream, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# HIP streams and hardware queues (measured, round 2; tools/timeline.py on a rocprofv3 kernel trace, Queue_Id column): the
# software pipeline runs 1 main + 3 selection + 3 pre-stage streams, and ROCm maps streams round-robin onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), where two streams of one queue serialise. With 4 queues the main stream
# shares its queue with one selection stream and idles ~0.5 ms per fp16 step behind that stream's RoIAlign / NMS; with 8
# every stream has its own queue, the gaps vanish (24 us per step) — and the fp16 step gets SLOWER (5.3 -> 7.2 ms when all
# seven streams really run concurrently: the selection kernels' waves fragment the CUs' register files and the 256-register
# contraction blocks wait for whole SIMDs; 4.87 ms only when the host enqueues tick by tick). fp32 is indifferent
# (487 tiles/s either way). The default of 4 is kept; TD_BENCH_HW_QUEUES overrides it for experiments.
if "TD_BENCH_HW_QUEUES" in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["TD_BENCH_HW_QUEUES"]

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MATRIX_TFLOPS = 2500.0
_PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
PMC_TAG = next((t for t in ("r06", "r05", "r04", "r03", "r02") if os.path.exists(os.path.join(_PROFILES, f"{t}_pmc_conv_fp32.json"))), "r02")
HBM_ACHIEVABLE_TBS = 6.3            # MI355X_MICROARCH.md: measured streaming rate (8.0 TB/s spec)
MASK_HEAD_GFLOP_PER_DET = 1.028
SCHED = {"streams": "{n} engines, each a whole forward on its own HIP stream, batches round-robin (HBM-bound kernels and kernel tails of one "
                    "forward run under the MFMA-bound contractions of the others)",
         "phases": "3 batches in flight per GPU: contraction phases on a main HIP stream, selection phases (top-k/NMS/RoIAlign/paste) on one "
                   "side stream per batch in flight",
         "plain": "plain loop: one batch at a time ({n} stream(s))"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)     # 32 x 8 = the 256-tile stream of BASELINE config #2
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--depth", type=int, default=50, choices=(50, 101))
    ap.add_argument("--precision", default="fp32", choices=("fp32", "fp16"))
    ap.add_argument("--tile", type=int, default=1000)
    ap.add_argument("--stream-tiles", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=16, help="distinct seeds generated (cycled over the stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=16)   # ≈ 13 s of CPU work on 16 threads
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="every timed region lasts at least this long: the K-step block is repeated R times inside ONE timed region "
                         "(barrier + synchronize on both sides of the R x K steps) and ms_per_step = seconds / (R x K); 0 = exactly K steps")
    ap.add_argument("--detail", default=None, help="where the full (uncompacted) result goes; default bench_detail.json beside this script")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-fp16", action="store_true", help="skip the second timed region with the fp16 engine")
    ap.add_argument("--schedule", default="streams", choices=("streams", "phases", "plain"),
                    help="streams (default): --streams engines, each a whole forward on its own HIP stream, batches round-robin; "
                         "phases: the round-1/2 software pipeline (contraction phases on a main stream, selection phases on side "
                         "streams); plain: one batch at a time on one stream")
    ap.add_argument("--no-pipeline", action="store_true", help="same as --schedule plain (kept for the tools/ scripts)")
    ap.add_argument("--no-serial", action="store_true", help="skip the extra informational single-stream region")
    ap.add_argument("--no-r101", action="store_true", help="skip the extra timed region with the reference's own depth (R101-FPN)")
    ap.add_argument("--no-fp16-b32", action="store_true", help="skip the fp16 region at BASELINE configs[4]'s batch (32 per GPU)")
    ap.add_argument("--no-two-model", action="store_true", help="skip the two-model (urban + forest, exclude flags) region of BASELINE configs[2]")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end region: warm Predictor.__call__ over a synthetic GeoTIFF on tmpfs "
                    "(window reads → device → Prediction_*.json files)")
    ap.add_argument("--no-lzw", action="store_true", help="skip the LZW-raster region (device decode of compressed rasters, files to files)")
    ap.add_argument("--lzw-side", type=int, default=20, help="the LZW raster is side x side tiles of 450 x 450 pixels")
    ap.add_argument("--e2e-side", type=int, default=20, help="the e2e raster is side x side tiles of --tile pixels")
    ap.add_argument("--streams", type=int, default=0, help="engines / HIP streams the batches alternate over (default 3 for "
                    "--schedule streams, 1 for plain): the HBM-bound kernels and the kernel tails of one forward run under the "
                    "MFMA-bound contractions of the others")
    a = ap.parse_args()
    if a.no_pipeline:
        a.schedule = "plain"
    if a.streams <= 0:
        a.streams = 3 if a.schedule == "streams" else 1
    return a


T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Threads this process may really use: affinity, capped by the cgroup CPU quota and by 16 (the GPU box's share)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(sd, rgb_np, n_tiles, sd_r101=None):
    """The torch-CPU oracle (oracle/ = test infrastructure, used here only as the timed CPU baseline) on the first
    n_tiles of the same stream: Pillow-restated resize + forward + paste, all host cores. The object the driver reads is
    R50 / batch 1 (`self.model([one tile])`, reference prediction.py:183 with max_batch_size 1); `extra` holds the other three
    cells of SURVEY.md §8d's table — batch 8 (two batches of the same 16 tiles) and the reference's own depth R101 (8 tiles)
    at batch 1 and 8 — timed in the same run on the same host."""
    from oracle import ops_ref as R
    from oracle.maskrcnn_ref import MaskRCNNOracle

    cores = host_cores()
    torch.set_num_threads(cores)

    def timed(oracle, n, batch, tag):
        t0 = time.perf_counter()
        dets = 0
        for k in range(0, n, batch):
            inputs = []
            for i in range(k, min(k + batch, n)):
                img, h, w = R.preprocess_tile_u8(rgb_np[i % len(rgb_np)].transpose(2, 0, 1))
                inputs.append({"image": img, "height": h, "width": w})
            out = oracle.forward(inputs)
            dets += sum(len(o["scores"]) for o in out)
            log(f"cpu baseline {tag}: {min(k + batch, n)}/{n} tiles")
        return n / (time.perf_counter() - t0), time.perf_counter() - t0, dets

    oracle = MaskRCNNOracle(sd)
    img, h, w = R.preprocess_tile_u8(rgb_np[0].transpose(2, 0, 1))
    oracle.forward([{"image": img, "height": h, "width": w}])          # warm-up (thread pools, allocator)
    v1, dt, dets = timed(oracle, n_tiles, 1, "R50 batch 1")
    extra = {"r50_b1": v1}
    if n_tiles >= 8:
        extra["r50_b8"] = timed(oracle, n_tiles, 8, "R50 batch 8")[0]
    if sd_r101 is not None and n_tiles >= 8:
        o101 = MaskRCNNOracle(sd_r101)
        n101 = max(8, n_tiles // 2)
        extra["r101_b1"] = timed(o101, n101, 1, "R101 batch 1")[0]
        extra["r101_b8"] = timed(o101, n101, 8, "R101 batch 8")[0]
        extra["r101_tiles"] = n101
    return {"value": v1, "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"first {n_tiles} tiles of the same stream, R50, batch 1, torch {torch.__version__} CPU fp32 oracle port "
                      f"(resize+forward+paste), {dt:.1f} s, {dets} detections",
            "extra": extra}


def _r(x, n=4):
    """Round a float to n significant digits (the compact line carries scalars, not noise digits)."""
    if x is None or isinstance(x, (bool, int, str)):
        return x
    return float(f"{float(x):.{n}g}")


COMPACT_LIMIT = 4096        # the driver keeps a bounded tail of stdout: the LAST line must fit with room to spare


def compact_line(full):
    """The ONE JSON object bench.py prints as its last stdout line, built from the full result dict (which goes to
    bench_detail.json and stderr): the contract keys, the headline `roofline` and `cpu_baseline` objects, what the collective
    layer saw, and one scalar per extra timed region. No prose beyond one short string per object; < COMPACT_LIMIT bytes."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    c = {k: (_r(full[k], 6) if isinstance(full.get(k), float) else full.get(k)) for k in keep}
    cfg = full["config"]
    c["config"] = {k: cfg.get(k) for k in ("workload", "depth", "batch_per_gpu", "tile", "net_input", "parallelism", "schedule",
                                          "concurrent_forwards", "stream_tiles", "distinct_tiles", "detections_per_tile") if k in cfg}
    c["timed_steps"] = full.get("timed_steps")
    c["timed_seconds"] = _r(full.get("timed_seconds"))
    rf = full.get("roofline")
    if rf:
        o = {k: (_r(rf.get(k), 5) if isinstance(rf.get(k), float) else rf.get(k)) for k in
             ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch",
              "executed_flops_per_launch", "kernel", "launches_per_step", "method")}
        if o.get("traffic_source"):
            o["traffic_source"] = str(o["traffic_source"]).split(" ")[0]        # the file; the detail file keeps the sentence
        o["algorithmic_gflop_per_tile"] = {k: _r(v, 5) for k, v in (rf.get("algorithmic_gflop_per_tile") or {}).items()}
        o["avg_launch_us"] = _r(rf.get("span", {}).get("avg_launch_us"))
        o["effective_tflops"] = _r(rf.get("effective_tflops"))
        o["hbm_gbytes_per_step"] = _r(rf.get("hbm_gbytes_per_step"))
        ex = rf.get("exclusive")
        if ex:
            o["exclusive_frac"] = _r(ex["frac"])
            o["exclusive_sol_frac"] = _r(ex["sol_frac"])
            o["exclusive_ms_per_step"] = _r(ex["span_ms_per_step"])
        c["roofline"] = o
    cb = full.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample"][:200], "extra": {k: _r(v) for k, v in cb.get("extra", {}).items()}}
    rk = full.get("ranks")
    if rk:
        c["ranks"] = {k: rk.get(k) for k in ("world", "backend", "devices", "distinct_gpus", "gather_bytes_per_step")}

    def val(*path):
        o = full
        for k in path:
            if not isinstance(o, dict) or k not in o:
                return None
            o = o[k]
        return _r(o)

    scal = {"fp16": val("fp16", "value"), "fp16_frac": val("fp16", "roofline", "frac"),
            "fp16_exclusive_frac": val("fp16", "roofline", "exclusive", "frac"),
            "fp16_batch32": val("fp16_batch32", "value"), "fp16_batch32_frac": val("fp16_batch32", "roofline", "frac"),
            "r101_f32": val("r101", "f32", "value"), "r101_f32_frac": val("r101", "f32", "roofline", "frac"),
            "r101_f16": val("r101", "f16", "value"), "r101_f16_frac": val("r101", "f16", "roofline", "frac"),
            "single_stream": val("single_stream", "value"), "fp16_single_stream": val("fp16", "single_stream", "value"),
            "phase_pipeline": val("phase_pipeline", "value"),
            "two_model_f32": val("two_model", "f32", "value"), "two_model_f16": val("two_model", "f16", "value"),
            "e2e_f32": val("e2e", "f32", "value"), "e2e_f32_ratio": val("e2e", "f32", "ratio_to_model_stage"),
            "e2e_f16": val("e2e", "f16", "value"), "e2e_f16_ratio": val("e2e", "f16", "ratio_to_model_stage"),
            "e2e_chained_f32": val("e2e", "f32", "chained", "value"), "e2e_chained_f32_ratio": val("e2e", "f32", "chained", "ratio_to_model_stage"),
            "e2e_chained_f16": val("e2e", "f16", "chained", "value"), "e2e_chained_f16_ratio": val("e2e", "f16", "chained", "ratio_to_model_stage"),
            "e2e_model_f32": val("e2e", "f32", "model_stage_same_weights"), "e2e_model_f16": val("e2e", "f16", "model_stage_same_weights"),
            "e2e_crowns_model_f32": val("e2e_crowns", "f32", "model_stage_same_weights"), "e2e_crowns_model_f16": val("e2e_crowns", "f16", "model_stage_same_weights"),
            "e2e_crowns_f32": val("e2e_crowns", "f32", "value"), "e2e_crowns_f32_ratio": val("e2e_crowns", "f32", "ratio_to_model_stage"),
            "e2e_crowns_f16": val("e2e_crowns", "f16", "value"), "e2e_crowns_f16_ratio": val("e2e_crowns", "f16", "ratio_to_model_stage"),
            "e2e_crowns_chained_f32_ratio": val("e2e_crowns", "f32", "chained", "ratio_to_model_stage"),
            "e2e_crowns_chained_f16_ratio": val("e2e_crowns", "f16", "chained", "ratio_to_model_stage"),
            "predict_tiles_f32": val("predict_tiles", "f32", "value"), "predict_tiles_f16": val("predict_tiles", "f16", "value"),
            "predict_tiles_f32_ratio": val("predict_tiles", "f32", "ratio_to_model_stage"), "predict_tiles_f16_ratio": val("predict_tiles", "f16", "ratio_to_model_stage"),
            "predict_tiles_noise_f32": val("predict_tiles_noise", "f32", "value"), "predict_tiles_noise_f16": val("predict_tiles_noise", "f16", "value"),
            "e2e_contours_per_tile": val("e2e", "f32", "contours_per_tile"), "e2e_crowns_contours_per_tile": val("e2e_crowns", "f32", "contours_per_tile"),
            "e2e_json_kb_per_tile": _r((full.get("e2e", {}).get("f32", {}).get("json_bytes_per_tile") or 0) / 1e3) or None,
            "e2e_crowns_json_kb_per_tile": _r((full.get("e2e_crowns", {}).get("f32", {}).get("json_bytes_per_tile") or 0) / 1e3) or None,
            "e2e_lzw_f16": val("lzw", "f16", "device", "value"), "e2e_lzw_f16_ratio": val("lzw", "f16", "device", "ratio_to_model_stage"),
            "e2e_lzw_host_reader_f16": val("lzw", "f16", "host_reader", "value"),
            "e2e_lzw_same_raster_uncompressed_f16": val("lzw", "f16", "uncompressed", "value"),
            "lzw_decode_windows_450_per_s": val("lzw", "f16", "decode_windows_450x450x4_per_s"), "lzw_decode_gb_per_s": val("lzw", "f16", "decode_gbytes_per_s"),
            "lzw_kernel_gb_per_s": val("lzw", "f16", "kernel_gbytes_per_s"),
            "e2e_deflate_f16": val("deflate", "f16", "device", "value"), "e2e_deflate_f16_ratio": val("deflate", "f16", "device", "ratio_to_model_stage"),
            "deflate_decode_windows_450_per_s": val("deflate", "f16", "decode_windows_450x450x4_per_s"), "deflate_kernel_gb_per_s": val("deflate", "f16", "kernel_gbytes_per_s")}
    c["regions"] = {k: v for k, v in scal.items() if v is not None}
    c["regions_unit"] = "tiles/s (two_model: tile visits/s; *_frac: executed FLOPs / MFMA peak; *_ratio: e2e / model stage; predict_tiles_*: files to GeoPackage layers = predict + stitch, image-sharded at N > 1; *_per_tile: counts / kB)"
    c["detail"] = full.get("detail_file")
    return c


def emit(full, detail_path=None):
    """Full result → detail file (+ stderr); compact line → stdout (the last line, alone). Returns the compact string."""
    c = compact_line(full)
    s = json.dumps(c, separators=(",", ":"))
    assert len(s) < COMPACT_LIMIT, f"compact bench line is {len(s)} bytes (limit {COMPACT_LIMIT}): move something into the detail file"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config"):
        assert c.get(key) is not None, f"compact bench line lacks '{key}'"
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as exc:
            print(f"[bench] could not write {detail_path}: {exc}", file=sys.stderr)
    print("[bench] full result: " + json.dumps(full), file=sys.stderr, flush=True)
    print(s, flush=True)
    return s


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node "
                             f"{args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    import torch.distributed as dist

    # one process per GPU; TD_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    backend = os.environ.get("TD_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from treedetection_amd.engine import Engine, INPUT_U8_HWC
    from treedetection_amd.synth import make_stream
    from treedetection_amd.weights import make_synthetic_state_dict

    torch.set_num_threads(host_cores())
    log("generating weights")
    sd = make_synthetic_state_dict(args.depth, seed=0)
    log("generating tile stream")
    B, S = args.batch, args.tile
    # every rank gets its own shard of the stream: tile t of the global stream goes to rank t % world
    n_local = min(args.stream_tiles, B * (args.steps + args.warmup))
    rgb_np, ndsm_np = make_stream(n_local, S, distinct=args.distinct)
    if world > 1:   # different ranks see different tiles (shifted seeds would cost start-up; rotate instead)
        rgb_np = np.roll(rgb_np, rank, axis=0)
        ndsm_np = np.roll(ndsm_np, rank, axis=0)
    dev = torch.device("cuda", local_rank)
    rgb = torch.from_numpy(rgb_np).to(dev)            # [n,S,S,3] uint8, resident in HBM before the timed region
    ndsm = torch.from_numpy(ndsm_np).to(dev)          # side band: travels with the tile, not a network input
    gather_keys = ("boxes", "scores", "count", "mask_probs")

    nsteps = args.steps      # the extra regions below re-bind sd / B / nsteps before calling run_pipelined again

    repeats = {}       # region name -> R (how many K-step blocks its timed region held)

    def measure(enqueue, n, engs, profile, name):
        """The timed region: `enqueue(first, count)` enqueues steps [first, first + count). One K-step block is timed first
        (it doubles as extra warm-up and tells how long K steps take); if that is shorter than --min-seconds the region proper
        holds R blocks, R = ceil(min_seconds / block time) agreed over the ranks. Barrier + synchronize on both sides, one clock
        around all R x K steps. → (seconds per K steps = region seconds / R, host enqueue seconds per K steps, R)."""
        def region(first, count):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enqueue(first, count)
            t_enq = time.perf_counter() - t0          # host time to enqueue everything (launch-bound if close to dt)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - t0, t_enq

        R = 1
        first = 0
        if args.min_seconds > 0:
            d1, _ = region(0, n)
            first = n
            t1 = torch.tensor([d1], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            d1 = float(t1.item())
            R = max(1, int(np.ceil(args.min_seconds / max(d1, 1e-6))))
            if profile:
                for e in engs:
                    e.profile_read(reset=True)
                    e.profile_classes(reset=True)
        dtr, t_enq = region(first, n * R)
        repeats[name] = R
        return dtr / R, t_enq / R, R

    def collect_profile(engs, R=1):
        """Sum of the engines' per-category accumulators over the region, divided by its R blocks (so every figure downstream
        is per K steps); "_classes" = the speed-of-light accounting by kernel class."""
        prof = _collect_profile(engs)
        if R > 1:
            for k in prof:
                for f in prof[k]:
                    if isinstance(prof[k][f], dict):
                        for g in prof[k][f]:
                            prof[k][f][g] /= R
                    else:
                        prof[k][f] /= R
        return prof

    def _collect_profile(engs):
        prof = None
        for e in engs:
            p1 = e.profile_read(reset=True)
            p1["_classes"] = e.profile_classes(reset=True)
            e.profile_enable(False)
            if prof is None:
                prof = p1
            else:
                for k in prof:
                    for f in prof[k]:
                        if isinstance(prof[k][f], dict):
                            for g in prof[k][f]:
                                prof[k][f][g] += p1[k][f][g]
                        else:
                            prof[k][f] += p1[k][f]
        return prof

    def run(precision, ns, profile, name=None, weights=None):
        """Warm-up + timed region for one engine precision over `ns` engines / HIP streams → (seconds max over
        ranks, profile dict or None, detections). `weights`: another state dict of the same architecture (the compact-crown
        mask head of the e2e fixture) instead of the run's own."""
        log(f"creating engine ({precision}, {ns} stream(s))")
        engs = [Engine(weights if weights is not None else sd, device=local_rank, precision=precision) for _ in range(ns)]
        outs = [e.alloc_outputs(B, S, S, paste=True) for e in engs]
        streams = [torch.cuda.Stream() for _ in range(ns)] if ns > 1 else [torch.cuda.current_stream()]
        eng, out = engs[0], outs[0]
        gls = None
        if world > 1 and rank == 0:      # one set of receive buffers per engine: their gathers run on different streams
            gls = [{k: [torch.empty_like(out[k]) for _ in range(world)] for k in gather_keys} for _ in range(ns)]

        def step(i):
            e, o = engs[i % ns], outs[i % ns]
            gl = gls[i % ns] if gls is not None else None
            with torch.cuda.stream(streams[i % ns]):
                tiles = [rgb[(i * B + j) % n_local] for j in range(B)]
                batch, hw_valid, hw_out = e.preprocess_tiles_u8(tiles)
                e.forward_raw(batch, INPUT_U8_HWC, hw_valid, hw_out, o)
                if world > 1:
                    for k in gather_keys:   # RCCL gather of the per-tile detections to rank 0 (hand-off to stitching)
                        if backend == "nccl":
                            dist.gather(o[k], gl[k] if rank == 0 else None, dst=0)
                        else:   # rehearsal backend: gloo moves host tensors
                            h = o[k].cpu()
                            dist.gather(h, [torch.empty_like(h) for _ in range(world)] if rank == 0 else None, dst=0)

        log("warm-up (the first forward also measures the block-tile choice per layer)")
        for i in range(max(args.warmup, ns)):
            step(i)
            torch.cuda.synchronize()
            log(f"warm-up step {i + 1} done")
        if profile:
            for e in engs:
                e.profile_enable(profile)            # True / 2 (detail: an event pair per contraction launch)
                e.profile_read(reset=True)
                e.profile_classes(reset=True)

        def enqueue(first, count):
            for i in range(first, first + count):
                step(args.warmup + i)
        dt, t_enq, R = measure(enqueue, nsteps, engs, profile, name or precision)
        log(f"timed region done ({precision}): {dt * R:.3f} s for {R} x {nsteps} steps (host enqueue {t_enq * R:.3f} s)")
        prof = collect_profile(engs, R) if profile else None
        ndet = int(out["count"].sum().item())     # detections of one batch (for the mask-head FLOP estimate)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        for e in engs:
            e.close()
        return float(tmax.item()), prof, ndet

    def run_pipelined(precision, profile, name=None):
        """Three engines, one main HIP stream and one side stream per engine: every tick enqueues, on the main stream,
        the trunk of batch t, the mask-head convs of batch t-2 and the box-head FCs of batch t-1 (contractions back to
        back, never overlapping each other), and on each batch's own side stream the selection phase that follows
        (top-k / NMS / RoIAlign / detections / paste) — those low-occupancy kernels run underneath the next
        contractions, and because every selection phase was enqueued a whole trunk before the contraction that
        consumes it, the main stream does not stall on them. K batches take K + 2 ticks; the timed region covers all of
        them (fill and drain included)."""
        log(f"creating 3 engines ({precision}, software pipeline: one main stream, one side stream per engine)")
        engs = [Engine(sd, device=local_rank, precision=precision) for _ in range(3)]
        outs = [e.alloc_outputs(B, S, S, paste=True) for e in engs]
        main = torch.cuda.Stream()      # (a high-priority main stream was measured: fp32 -1.5 %, fp16 +-0)
        sides = [torch.cuda.Stream() for _ in range(3)]   # one per engine: a batch's selection phases only wait on that batch
        pres = [torch.cuda.Stream() for _ in range(3)]    # one per engine: resize + stem + pool of its NEXT batch (fp16 schedule)
        # (confining the selection / pre-stage streams to 32-128 CUs with hipExtStreamCreateWithCUMask was measured: fp16 1477 ->
        # 826-1036 tiles/s, fp32 487 -> 392 — the masked queues slow the unmasked main stream's contractions as well)
        layout = os.environ.get("TD_BENCH_STREAMS", "")
        if layout == "3":        # experiment: ONE selection stream and ONE pre-stage stream for all three engines
            sides = [sides[0]] * 3
            pres = [pres[0]] * 3
        elif layout == "2":      # experiment: selection and pre-stage work of all engines on one stream
            sides = [sides[0]] * 3
            pres = [sides[0]] * 3
        gl = None
        if world > 1 and rank == 0:
            gl = {k: [torch.empty_like(outs[0][k]) for _ in range(world)] for k in gather_keys}

        from treedetection_amd.engine import PHASE_STEM
        staged = set()
        # Resize + stem + pool ahead of time on the side stream: measured +5.6 % for the fp16 engine (its main stream
        # is short, the 0.4 ms count) and -2.5 % for fp32 (the VALU-heavy stem then competes with the fp32 MFMA convs
        # for the same CUs: conv time 18.1 -> 19.1 ms) — so only the fp16 schedule uses the pre-phase.
        prestage = precision == "fp16"

        def pre_stage(i):
            """Resize + stem + max-pool of batch i on its engine's side stream (VALU / HBM kernels: they run underneath
            the previous batch's contractions instead of occupying the main stream)."""
            # on a stream of its own: behind the engine's side stream it queued after the previous batch's whole selection
            # tail (mask predictor, paste) and the next trunk waited 0.8 ms per step for it (rocprof trace, round 2)
            e, o, side = engs[i % 3], outs[i % 3], pres[i % 3]
            with torch.cuda.stream(side):
                tiles = [rgb[(i * B + j) % n_local] for j in range(B)]
                batch, hw_valid, hw_out = e.preprocess_tiles_u8(tiles)
            e.forward_phase(PHASE_STEM, side, batch, INPUT_U8_HWC, hw_valid, hw_out, o)
            staged.add(i)

        def tick(t, first, last):
            # main-stream order per tick: trunk of the new batch first, then the mask convs of batch t-2 and the FCs of
            # batch t-1 — each of those waits on a selection phase that was enqueued a whole trunk earlier, so the main
            # stream never stalls on the side streams
            for age, (pm, ps) in ((0, (0, 1)), (2, (4, 5)), (1, (2, 3))):
                i = t - age
                if not first <= i < last:
                    continue
                e, o = engs[i % 3], outs[i % 3]
                side = sides[i % 3]
                if pm == 0 and prestage:
                    if i not in staged:
                        pre_stage(i)                 # pipeline fill: nothing ran ahead of an engine's first batch
                    e.forward_phase(0, main)         # continues at res2 once the pre-stage has finished
                elif pm == 0:
                    with torch.cuda.stream(main):
                        tiles = [rgb[(i * B + j) % n_local] for j in range(B)]
                        batch, hw_valid, hw_out = e.preprocess_tiles_u8(tiles)
                    e.forward_phase(0, main, batch, INPUT_U8_HWC, hw_valid, hw_out, o)
                else:
                    e.forward_phase(pm, main)
                e.forward_phase(ps, side)
                if ps == 5:
                    if world > 1:
                        with torch.cuda.stream(side):
                            for k in gather_keys:   # RCCL gather of the finished batch's detections to rank 0
                                if backend == "nccl":
                                    dist.gather(o[k], gl[k] if rank == 0 else None, dst=0)
                                else:
                                    h = o[k].cpu()
                                    dist.gather(h, [torch.empty_like(h) for _ in range(world)] if rank == 0 else None, dst=0)
                    if prestage and i + 3 < last:
                        pre_stage(i + 3)             # this engine's next batch: its buffers are free from here on

        def run_batches(first, last):
            for t in range(first, last + 2):
                tick(t, first, last)

        log("warm-up (each engine measures its block-tile choices on its first batch)")
        nw = max(args.warmup, 3)
        run_batches(0, nw)
        torch.cuda.synchronize()
        if profile:
            for e in engs:
                e.profile_enable(True)
                e.profile_read(reset=True)
                e.profile_classes(reset=True)
        dt, t_enq, R = measure(lambda first, count: run_batches(nw + first, nw + first + count), nsteps, engs, profile, name or f"{precision}_phases")
        log(f"timed region done ({precision}, pipelined): {dt * R:.3f} s for {R} x {nsteps} steps (host enqueue {t_enq * R:.3f} s)")
        prof = collect_profile(engs, R) if profile else None
        ndet = int(outs[0]["count"].sum().item())
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        for e in engs:
            e.close()
        return float(tmax.item()), prof, ndet

    def run_two_model(precision):
        """BASELINE configs[2]: the two-model path (reference detection.py:154-164 — the urban model over every tile that is not
        flagged only_forest, THEN the forest model over every tile that is not only_urban; prediction.py:79-93 drops a flagged
        tile before batching). Stream = the K steps' tiles with the flags a forest outline over the left part of a mosaic
        gives: of every three consecutive tiles one is forest-only, one mixed, one urban-only. Each model has its own engines
        (weights resident, three HIP streams as in the headline region); the timed region runs model after model, fill and
        drain included. → (seconds, tiles visited by urban, by forest)."""
        log(f"two-model region ({precision}): creating 2 x {args.streams} engines")
        sds = {"urban": sd, "forest": make_synthetic_state_dict(args.depth, seed=2)}
        ns = args.streams
        n_tiles = nsteps * B
        flags = [("only_forest", "mixed", "only_urban")[t % 3] for t in range(n_tiles)]
        visit = {"urban": [t for t in range(n_tiles) if flags[t] != "only_forest"],
                 "forest": [t for t in range(n_tiles) if flags[t] != "only_urban"]}
        engs = {m: [Engine(sds[m], device=local_rank, precision=precision) for _ in range(ns)] for m in sds}
        outs = {m: [e.alloc_outputs(B, S, S, paste=True) for e in engs[m]] for m in sds}
        streams = [torch.cuda.Stream() for _ in range(ns)]

        def model_pass(m, tiles_idx):
            for k in range(0, len(tiles_idx), B):
                idx = tiles_idx[k:k + B]
                j = (k // B) % ns
                with torch.cuda.stream(streams[j]):
                    tiles = [rgb[t % n_local] for t in idx]
                    batch, hw_valid, hw_out = engs[m][j].preprocess_tiles_u8(tiles)
                    o = outs[m][j]
                    engs[m][j].forward_raw(batch, INPUT_U8_HWC, hw_valid, hw_out, {kk: v[:len(idx)] for kk, v in o.items()})

        for m in sds:                                   # warm-up: the same passes once (tile choices of the full AND the tail batch shape)
            model_pass(m, visit[m])
        def both(first, count):
            for _ in range(count):
                for m in ("urban", "forest"):
                    model_pass(m, visit[m])
        dtm, _, R = measure(both, 1, [e for m in sds for e in engs[m]], False, f"two_model_{precision}")
        dets = {m: int(outs[m][0]["count"].sum().item()) for m in sds}
        for m in sds:
            for e in engs[m]:
                e.close()
        log(f"two-model region done: {dtm * R:.3f} s for {R} x (urban pass + forest pass)")
        return dtm, len(visit["urban"]), len(visit["forest"]), n_tiles, dets

    def make_fixture(side):
        """The e2e raster on tmpfs, built by rank 0 and seen by every rank of the node: side x side tiles of S x S pixels (default
        20 x 20: the 400 tiles the reference cuts from one 1 km² image, example/config.yml:26-28), 4-band RGBI uint8, tile
        metadata from the package's own tile producer. → {"root", "tif", "tjson", "ntiles", "tmpfs"}."""
        import tempfile
        from treedetection_amd.geotiff import write_geotiff
        from treedetection_amd.preprocessing import tile_data
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        fx = None
        if rank == 0:
            root = tempfile.mkdtemp(prefix="td_e2e_", dir=base)
            os.makedirs(f"{root}/rgb")
            img = np.zeros((4, side * S, side * S), np.uint8)
            for r in range(side):
                for c in range(side):
                    t = rgb_np[(r * side + c) % len(rgb_np)]
                    img[:3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t.transpose(2, 0, 1)
                    img[3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t[..., 1]
            gsd = 0.2
            tif = f"{root}/rgb/324125317.tif"
            write_geotiff(tif, img, (gsd, 0.0, 412000.0, 0.0, -gsd, 5318000.0 + side * S * gsd), 25832)
            del img
            tile_data([tif], f"{root}/tiles", buffer=0, tile_width=int(S * gsd), tile_height=int(S * gsd))
            tjson = f"{root}/tiles/324125317.json"
            fx = {"root": root, "tif": tif, "tjson": tjson, "ntiles": len(json.load(open(tjson))), "tmpfs": bool(base), "side": side}
        if world > 1:
            box = [fx]
            dist.broadcast_object_list(box, src=0, device=dev if backend == "nccl" else None)
            fx = box[0]
        return fx

    def run_predict_tiles(precision, fx, sd_w, tag, per_rank):
        """``predict_tiles`` as the reference defines it — predict AND stitch (detection.py:228-243) — files to GeoPackage layers,
        through the package's own walk (detection.walk_images: chained Predictor.submit, every finished image stitched on host
        threads while the GPU predicts the next one, then the leftover stitching pass + resume files exactly as
        predict_on_model / predict_tiles run them). ``per_rank`` images per rank = the fixture raster under per_rank x world
        names (hard links); at N > 1 the images are sharded whole over the ranks (detection.assign_images) with NO collective
        inside the walk: one manifest gather + one all-reduce after it, as predict_on_model does. Warm predictor (weights
        resident, tile choices measured, buffers allocated): one untimed image first. → dict (value = tiles/s, all ranks)."""
        import logging
        import shutil
        import treedetection_amd as T
        from treedetection_amd import detection as DT
        from treedetection_amd import distributed as TD
        from treedetection_amd.recoveries import load_stitching_recovery, save_stitching_recovery
        from treedetection_amd.stitching import process_and_stitch_predictions
        root, n_img = fx["root"], per_rank * world
        work = f"{root}/pt_{tag}_{precision}"
        tiles_pt, out_pred, out_gpkg = f"{work}/tiles", f"{work}/predictions", f"{work}/geojson_predictions"
        names = [str(324125400 + k) for k in range(n_img)]
        if rank == 0:
            os.makedirs(tiles_pt)
            os.makedirs(f"{work}/rgb")
            for nm in names:
                os.link(fx["tif"], f"{work}/rgb/{nm}.tif")
                os.link(fx["tjson"], f"{tiles_pt}/{nm}.json")
            os.makedirs(f"{work}/warm")
            for r in range(world):          # one warm-up image per rank (its own output folder: nobody deletes under a writer)
                os.link(fx["tif"], f"{work}/warm/{324125300 + r}.tif")
                os.link(fx["tjson"], f"{work}/warm/{324125300 + r}.json")
        if world > 1:
            dist.barrier()
        paths = [f"{work}/rgb/{nm}.tif" for nm in names]
        cfg = T.setup_model_cfg(update_model="synthetic", device=str(local_rank))
        Bp = DT.engine_batch_size({"precision": precision}, B)       # what predict_on_model would run (fp16_min_batch off by default: B)
        pred = T.Predictor(cfg, device_type=str(local_rank), max_batch_size=Bp, output_dir=out_pred, precision=precision,
                           state_dict=sd_w, return_predictions=False, sharded_epilogue="local")
        logger = logging.getLogger("td-bench")
        logger.setLevel(logging.ERROR)
        config = {"logger": logger, "simplify_tolerance": 0.2}
        try:
            pred.submit(f"{work}/warm/{324125300 + rank}.tif", f"{work}/warm/{324125300 + rank}.json", whole_image=True).result()   # warm-up image (not in the timed set)
            owner = DT.assign_images(paths, world)
            mine = [paths[i] for i in range(n_img) if owner[i] == rank]
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rep = DT.walk_images(config, pred, mine, tiles_pt, out_pred, chain=True, stitch_to=out_gpkg)
            t_walk = time.perf_counter() - t0
            reports = TD.gather_objects(rep) if world > 1 else [rep]
            ok = True
            if rank == 0:
                stitched = [f for r in reports for f in r["stitched"]]
                ok = sorted(p_ for r in reports for p_ in r["done"]) == sorted(paths)
                save_stitching_recovery(out_gpkg, sorted(load_stitching_recovery(out_gpkg, None)) + stitched, None)
                process_and_stitch_predictions(tiles_pt, out_pred, out_gpkg, max_workers=4, shift=1, simplify_tolerance=0.2, logger=logger)
            ok = TD.all_ok(ok)
            if world > 1:
                dist.barrier()
            dt_pt = time.perf_counter() - t0
            # the same walk once more WITHOUT stitching (same predictor, same images, files rewritten): what the stitching costs
            # the walk it runs beside
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            DT.walk_images(config, pred, mine, tiles_pt, out_pred, chain=True, stitch_to=None)
            if world > 1:
                dist.barrier()
            dt_plain = time.perf_counter() - t0
        finally:
            pred.close()
        tm = torch.tensor([dt_pt, t_walk, rep["stitch_seconds"], dt_plain], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        res = None
        if rank == 0:
            layers = [f for f in os.listdir(out_gpkg) if f.endswith(".gpkg")]
            nbytes = sum(os.path.getsize(f"{out_gpkg}/{f}") for f in layers)
            assert ok and len(layers) == n_img, f"predict_tiles region: {len(layers)} layers for {n_img} images (ok={ok})"
            files = sum(len(os.listdir(f"{out_pred}/{nm}")) for nm in names)
            res = {"value": n_img * fx["ntiles"] / float(tm[0]), "unit": "tiles/s", "images": n_img, "images_per_rank": per_rank,
                   "tiles_per_image": fx["ntiles"], "batch": Bp, "seconds": float(tm[0]), "walk_seconds_max": float(tm[1]),
                   "stitch_thread_seconds_max": float(tm[2]), "same_walk_without_stitching_seconds": float(tm[3]),
                   "predict_only_value": n_img * fx["ntiles"] / float(tm[3]), "prediction_files": files, "layers": len(layers), "layer_bytes": nbytes,
                   "sharding": "single process" if world == 1 else f"whole images over {world} ranks (detection.assign_images), no collective in the walk",
                   "note": "files to GeoPackage layers: window reads, H2D, resize, forward, paste, contours, Prediction_*.json, then per image "
                           "simplify + edge filter + <image>.gpkg on host threads while the next image predicts; + the resume files"}
            log(f"predict_tiles region ({tag}, {precision}): {n_img} images x {fx['ntiles']} tiles in {float(tm[0]):.3f} s "
                f"(walk {float(tm[1]):.3f} s, stitch threads {float(tm[2]):.3f} s)")
            shutil.rmtree(work, ignore_errors=True)
        if world > 1:
            dist.barrier()
        return res

    def run_e2e(precisions, fx, sd_e2e, tag):
        """predict_tiles' model stage end to end, files to files (reference prediction.py:47-77,197-265): a warm
        Predictor.__call__ over ONE synthetic GeoTIFF on tmpfs (``make_fixture``) — window reads into pinned memory, H2D, resize,
        forward, paste, D2H of the rows the paste wrote,
        contours → polygons → Prediction_<tile>.json written and counted. Per precision: first call = warm-up (weights, tile
        choices, buffers), then three timed calls (value = the fastest). `sd_e2e` = the weights: the seeded random set (noise-like
        masks: thousands of contours and ~1 MB of JSON per tile — a stress fixture for the host epilogue) or the same set with
        weights.blob_mask_head (compact crowns, tens of contours per tile: what a trained segmenter hands the epilogue)."""
        import shutil
        import treedetection_amd as T
        root, tif, tjson, ntiles, side, base = fx["root"], fx["tif"], fx["tjson"], fx["ntiles"], fx["side"], fx["tmpfs"]
        out = {}
        if True:
            cfg = T.setup_model_cfg(update_model="synthetic", device=str(local_rank))
            for precision in precisions:
                from treedetection_amd.detection import engine_batch_size
                Bp = engine_batch_size({"precision": precision}, B)    # what predict_on_model runs (fp16_min_batch off by default: B)
                pred = T.Predictor(cfg, device_type=str(local_rank), max_batch_size=Bp, output_dir=f"{root}/out_{precision}",
                                   precision=precision, state_dict=sd_e2e, return_predictions=False)
                pred(tif, tjson)                        # warm-up call
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    pred(tif, tjson)
                    times.append(time.perf_counter() - t0)
                stats = dict(pred.stats)
                # the stage as predict_on_model runs it over MANY images: the next image is submitted while the previous one
                # drains (Predictor.submit). Three images = the same raster under three names (hard links), one timed pass.
                chain = []
                for k in range(1, (3 if precision == "fp32" else 6) + 1):       # >= 1 s of chained work at either precision's rate
                    name = str(324125317 + k)
                    for src, dst in ((tif, f"{root}/rgb/{name}.tif"), (tjson, f"{root}/tiles/{name}.json")):
                        if not os.path.exists(dst):
                            os.link(src, dst)
                    chain.append((f"{root}/rgb/{name}.tif", f"{root}/tiles/{name}.json"))
                dt_chain = None
                for _ in range(2):                  # first pass creates the 1 200 files (as the single-image warm-up call did), second is timed
                    t0 = time.perf_counter()
                    pending = None
                    for pair in chain:
                        h = pred.submit(*pair)
                        if pending is not None:
                            pending.result()
                        pending = h
                    pending.result()
                    dt_chain = time.perf_counter() - t0
                chained_files = sum(len(os.listdir(f"{root}/out_{precision}/{os.path.basename(p_[0])[:-4]}")) for p_ in chain)
                pred.close()
                folder = f"{root}/out_{precision}/324125317"
                files = [f for f in os.listdir(folder) if f.startswith("Prediction_")]
                nbytes = sum(os.path.getsize(f"{folder}/{f}") for f in files)
                sample = sorted(files)[:: max(1, len(files) // 16)][:16]            # contours per tile from a sample of the files
                ncont = [len(json.load(open(f"{folder}/{f}"))) for f in sample]
                shutil.rmtree(f"{root}/out_{precision}", ignore_errors=True)
                dt_e = min(times)
                log(f"e2e region ({tag}, {precision}): {ntiles} tiles per call, calls {[round(t, 3) for t in times]} s")
                out[precision] = {"value": ntiles / dt_e, "unit": "tiles/s", "tiles_per_call": ntiles, "calls_s": times,
                                  "files_written": len(files), "prediction_bytes": nbytes, "batch": Bp,
                                  "json_bytes_per_tile": nbytes / max(len(files), 1), "contours_per_tile": float(np.mean(ncont)) if ncont else 0.0,
                                  "raster": f"{side * S}x{side * S}x4 uint8 GeoTIFF on {'tmpfs' if base else 'disk'}",
                                  "host_stage_seconds_last_call": stats,
                                  "chained": {"value": len(chain) * ntiles / dt_chain, "unit": "tiles/s", "images": len(chain), "seconds": dt_chain,
                                              "files_written": chained_files,
                                              "note": "images back to back as detection.predict_on_model walks them: image i+1 submitted while image i drains"}}
            return out

    def run_lzw(precision, sd_w, side, codec="lzw"):
        """SURVEY §8f-2's leftover: the raster as real orthophotos are stored — LZW, 256 x 256 tiles, predictor 2
        (GDAL: TILED=YES COMPRESS=LZW PREDICTOR=2) — cut into the REFERENCE's tiles (450 x 450 px, 400 per image) — files to files. The compressed blocks cross PCIe once and are decoded on the GPU
        (tiffdecode.hip, one wave per block), the tile windows are cut in HBM; the next image is decoded while the current one
        predicts (Predictor.prefetch). Reported: the decode alone (raster bytes/s, also as 450 x 450 x 4 windows/s, the reference's
        tile size), the chained files-to-files rate, and the same walk through the HOST reader's decode threads on a few tiles."""
        import shutil
        import tempfile
        import treedetection_amd as T
        from treedetection_amd.geotiff import GeoTiff, write_geotiff
        from treedetection_amd.preprocessing import tile_data
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        root = tempfile.mkdtemp(prefix="td_lzw_", dir=base)
        try:
            os.makedirs(f"{root}/rgb")
            TP = 450                                      # the reference's tile: 90 m x 90 m at 0.2 m (example/config.yml:26-28) → 800 x 800 network input
            px = side * TP
            nb = -(-px // S)
            img = np.zeros((4, nb * S, nb * S), np.uint8)
            for r in range(nb):
                for c in range(nb):
                    t = rgb_np[(r * nb + c) % len(rgb_np)]
                    img[:3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t.transpose(2, 0, 1)
                    img[3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t[..., 1]
            img = np.ascontiguousarray(img[:, :px, :px])
            tif = f"{root}/rgb/324125000.tif"
            t0 = time.perf_counter()
            write_geotiff(tif, img, (0.2, 0.0, 412000.0, 0.0, -0.2, 5318000.0 + px * 0.2), 25832, compression=codec, tile=(256, 256), predictor=2)
            t_enc = time.perf_counter() - t0
            raw_bytes, file_bytes = img.nbytes, os.path.getsize(tif)
            del img
            tile_data([tif], f"{root}/tiles", buffer=0, tile_width=int(TP * 0.2), tile_height=int(TP * 0.2))
            tjson = f"{root}/tiles/324125000.json"
            ntiles = len(json.load(open(tjson)))
            # (a) the decode alone: file → pinned memory → device → decoded raster in HBM
            from concurrent.futures import ThreadPoolExecutor
            g = GeoTiff(tif)
            times, ktimes = [], []
            pinned, rpool = [None], ThreadPoolExecutor(max_workers=8)
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                image, check = g.decode_to_device(f"cuda:{local_rank}", None, pinned, rpool)
                check()
                times.append(time.perf_counter() - t0)
                ktimes.append(check.kernel_ms * 1e-3)
                del image
            rpool.shutdown()
            del pinned
            t_dec, t_ker = min(times[1:]), min(ktimes[1:])
            cfg = T.setup_model_cfg(update_model="synthetic", device=str(local_rank))
            names = [str(324125001 + k) for k in range(5)]
            for nm in names:
                os.link(tif, f"{root}/rgb/{nm}.tif")
                os.link(tjson, f"{root}/tiles/{nm}.json")
            res = {"raster": f"{px}x{px}x4 uint8, {codec.upper()}, 256x256 tiles, predictor 2, on {'tmpfs' if base else 'disk'}; {ntiles} tiles of {TP}x{TP} px per image",
                   "kernel_seconds": t_ker, "kernel_gbytes_per_s": raw_bytes / t_ker / 1e9,
                   "raw_bytes": raw_bytes, "file_bytes": file_bytes, "compression_ratio": raw_bytes / file_bytes, "encode_seconds": t_enc,
                   "decode_seconds": t_dec, "decode_calls_s": times, "decode_gbytes_per_s": raw_bytes / t_dec / 1e9,
                   "decode_windows_450x450x4_per_s": raw_bytes / t_dec / (450 * 450 * 4), "tiles_per_image": ntiles}
            modes = [("device", "auto", names)]
            if codec == "lzw":
                # the same pixels stored uncompressed (one strip): what the 450-px walk does when no decode is in the way
                raw_tif = f"{root}/raw/324125000.tif"
                os.makedirs(f"{root}/raw")
                write_geotiff(raw_tif, GeoTiff(tif).read(), (0.2, 0.0, 412000.0, 0.0, -0.2, 5318000.0 + px * 0.2), 25832)
                for nm in names:
                    os.link(raw_tif, f"{root}/raw/{nm}.tif")
                modes += [("host_reader", False, names[:1]), ("uncompressed", "auto", names)]
            for mode, dd, imgs in modes:
                src_dir = "raw" if mode == "uncompressed" else "rgb"
                tif_w = f"{root}/raw/324125000.tif" if mode == "uncompressed" else tif
                pred = T.Predictor(cfg, device_type=str(local_rank), max_batch_size=B, output_dir=f"{root}/out_{mode}", precision=precision,
                                   state_dict=sd_w, return_predictions=False, device_decode=dd)
                try:
                    pred.prefetch(tif_w)
                    pred(tif_w, tjson)                 # warm-up image
                    t0 = time.perf_counter()
                    pending = None
                    pred.prefetch(f"{root}/{src_dir}/{imgs[0]}.tif")
                    for k, nm in enumerate(imgs):
                        if k + 1 < len(imgs):
                            pred.prefetch(f"{root}/{src_dir}/{imgs[k + 1]}.tif")
                        h = pred.submit(f"{root}/{src_dir}/{nm}.tif", f"{root}/tiles/{nm}.json")
                        if pending is not None:
                            pending.result()
                        pending = h
                    pending.result()
                    dt_l = time.perf_counter() - t0
                    files = sum(len(os.listdir(f"{root}/out_{mode}/{nm}")) for nm in imgs)
                    assert files == len(imgs) * ntiles, (mode, files)
                    res[mode] = {"value": len(imgs) * ntiles / dt_l, "unit": "tiles/s", "images": len(imgs), "seconds": dt_l,
                                 "decode": dict(pred.decode_stats)}
                finally:
                    pred.close()
                    shutil.rmtree(f"{root}/out_{mode}", ignore_errors=True)
            log(f"{codec} region ({precision}): decode {t_dec * 1e3:.1f} ms per {raw_bytes / 1e6:.0f} MB raster ({res['decode_windows_450x450x4_per_s']:.0f} windows of 450x450x4 per s), "
                f"files to files {res['device']['value']:.0f} tiles/s on the device decoder" +
                (f", {res['host_reader']['value']:.0f} through the host reader, {res['uncompressed']['value']:.0f} on the same raster stored uncompressed" if codec == "lzw" else ""))
            return res
        finally:
            shutil.rmtree(root, ignore_errors=True)

    if "TD_TUNE_CACHE" not in os.environ:     # engines of one run share their measured block-tile choices
        import tempfile
        os.environ["TD_TUNE_CACHE"] = os.path.join(tempfile.mkdtemp(prefix="td_tune_"), f"tiles_rank{rank}.txt")
    def go(precision, profile, name):
        if args.schedule == "phases":
            return run_pipelined(precision, profile, name)
        return run(precision, args.streams, profile, name)

    dt, prof, ndet = go(args.precision, not args.no_profile, "headline")
    extra = piped = None
    if args.precision == "fp32" and not args.no_fp16:
        extra = go("fp16", not args.no_profile, "fp16")
    if args.schedule != "plain" and not args.no_serial:
        # one forward at a time on one stream: the same kernels with nothing overlapping — the per-launch spans of THIS region
        # are the kernels' own durations (the `exclusive` roofline object)
        piped = run(args.precision, 1, not args.no_profile, "single_stream")
    # speed-of-light class table: a short plain-loop region with one HIP-event pair per contraction launch (it perturbs the
    # forward, so it is a region of its own and nothing else is quoted from it); fp16: its own plain loop + table
    detail = detail16 = piped16 = None
    detail_steps = max(1, min(4, args.steps))
    if not args.no_serial and not args.no_profile and world == 1:
        keep, keep_min = nsteps, args.min_seconds
        if extra is not None:
            piped16 = run("fp16", 1, True, "fp16_single_stream")
        nsteps, args.min_seconds = detail_steps, 0.0      # an event pair per launch perturbs the forward: a short region of its own
        detail = run(args.precision, 1, 2, "detail")
        if extra is not None:
            detail16 = run("fp16", 1, 2, "fp16_detail")
        nsteps, args.min_seconds = keep, keep_min
    phased = None
    if args.schedule == "streams" and not args.no_serial and world == 1:
        # informational: the phase pipeline — three batches in flight whose CONTRACTION kernels never overlap each other, so its
        # event spans are the kernels' own durations (the literal roofline of a schedule that still overlaps the selection work)
        phased = run_pipelined(args.precision, not args.no_profile, "phase_pipeline")
    r101 = b32 = None
    if args.precision == "fp32" and args.depth == 50 and args.schedule != "plain":
        if not args.no_r101 and world == 1:
            # the reference's own depth (TreeDetection/config.py:25 hard-codes R101-FPN): same stream, same schedule
            log("generating R101 weights")
            sd = make_synthetic_state_dict(101, seed=0)
            nsteps = -(-max(4, args.steps // 2) // args.streams) * args.streams      # whole rounds of the engines (these regions choose their own K)
            r101 = {"fp32": go("fp32", not args.no_profile, "r101_f32") + (nsteps,)}
            if not args.no_fp16:
                r101["fp16"] = go("fp16", not args.no_profile, "r101_f16") + (nsteps,)
            sd = make_synthetic_state_dict(args.depth, seed=0)
        if not args.no_fp16 and not args.no_fp16_b32:
            # BASELINE configs[4]: the fp16 MFMA path at batch 32 per GPU (same tiles, four times the rows per launch) — at every N:
            # configs[4] is quoted on 8 GPUs
            B, nsteps = 32, -(-max(4, args.steps // 4) // args.streams) * args.streams
            b32 = go("fp16", not args.no_profile, "fp16_batch32") + (nsteps,)
            B, nsteps = args.batch, args.steps
    two = e2e = e2e_c = None
    pt = pt_n = lzw = deflate = None
    if args.depth == 50 and args.schedule == "streams" and not args.no_e2e:
        import shutil
        from treedetection_amd.weights import blob_mask_head
        precs = [args.precision] + (["fp16"] if args.precision == "fp32" and not args.no_fp16 else [])
        fx = make_fixture(args.e2e_side)
        try:
            sd_c = blob_mask_head(sd, seed=0)
            per_rank = {"fp32": 5, "fp16": 6}        # images per rank in the predict_tiles regions: >= 1 s of work at either rate, the last image's stitching tail amortised
            if world == 1:
                if not args.no_two_model:
                    two = {args.precision: run_two_model(args.precision)}
                    if args.precision == "fp32" and not args.no_fp16:
                        two["fp16"] = run_two_model("fp16")
                # both fixtures of one precision back to back (an fp32 region that follows an fp16 one starts on a hotter, slower chip:
                # the crowns fp32 rate read 5 % low when it ran right after the fp16 noise region)
                e2e, e2e_c, pt, pt_n = {}, {}, {}, {}
                for pk in precs:
                    # each fixture's e2e rate is followed by ITS model stage (same stream, same schedule, its weights) in the same part of
                    # the run: late regions run on a hotter chip (5-9 % below the first region of the line), and the blob mask head changes
                    # what the mask-head contractions and the paste see — the e2e ratio is taken against this rate
                    e2e.update(run_e2e([pk], fx, sd, "noise-like masks"))
                    pt_n[pk] = run_predict_tiles(pk, fx, sd, "noise", per_rank[pk])
                    dtn, _, _ = run(pk, args.streams, False, name=f"e2e_model_{pk}")
                    e2e[pk]["model_stage_same_weights"] = pt_n[pk]["model_stage_same_weights"] = args.steps * B * world / dtn
                    e2e_c.update(run_e2e([pk], fx, sd_c, "compact crowns"))
                    pt[pk] = run_predict_tiles(pk, fx, sd_c, "crowns", per_rank[pk])
                    dtc, _, _ = run(pk, args.streams, False, name=f"crowns_model_{pk}", weights=sd_c)
                    e2e_c[pk]["model_stage_same_weights"] = pt[pk]["model_stage_same_weights"] = args.steps * B * world / dtc
                if not args.no_lzw:
                    pk = "fp16" if "fp16" in precs else precs[0]
                    lzw = {pk: run_lzw(pk, sd_c, args.lzw_side)}
                    lzw[pk]["model_stage_same_weights"] = e2e_c[pk]["model_stage_same_weights"]
                    deflate = {pk: run_lzw(pk, sd_c, args.lzw_side, codec="deflate")}
                    deflate[pk]["model_stage_same_weights"] = e2e_c[pk]["model_stage_same_weights"]
            else:
                # N > 1: predict_tiles files to GeoPackage layers with WHOLE IMAGES sharded over the ranks (the structure
                # detection.predict_on_model runs), compact-crown fixture
                pt = {}
                for pk in precs:
                    r = run_predict_tiles(pk, fx, sd_c, "crowns", per_rank[pk])
                    dtc, _, _ = run(pk, args.streams, False, name=f"crowns_model_{pk}", weights=sd_c)
                    if rank == 0:
                        r["model_stage_same_weights"] = args.steps * B * world / dtc
                        pt[pk] = r
        finally:
            if rank == 0:
                shutil.rmtree(fx["root"], ignore_errors=True)
    elif args.depth == 50 and world == 1 and args.schedule == "streams" and not args.no_two_model:
        two = {args.precision: run_two_model(args.precision)}
        if args.precision == "fp32" and not args.no_fp16:
            two["fp16"] = run_two_model("fp16")

    # what the collective layer saw (for the reader of an N > 1 line: did RCCL really run N ranks on N different GPUs?)
    props = torch.cuda.get_device_properties(local_rank)
    me_info = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
               "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
    ranks_info = [me_info]
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me_info)

    if rank == 0:
        tiles_total = args.steps * B * world
        # forwards whose contractions overlap (the phase pipeline keeps them back to back on one main stream: spans are the kernels' own)
        conc = 1 if args.schedule == "phases" else args.streams
        line = {
            "metric": "tiles/sec (1000x1000 RGB+nDSM) predict_tiles model stage",
            "value": tiles_total / dt,
            "unit": "tiles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: single model ResNet{args.depth}-FPN Mask R-CNN, "
                                   f"{n_local}-tile synthetic {S}x{S} RGB(+nDSM side band) stream per GPU ({min(args.distinct, n_local)} distinct "
                                   f"seeds cycled; re-walked R times for the >= 1 s region), batch={B} per GPU, "
                                   f"resize 800x800 + forward + paste on device, inputs resident in HBM",
                       "stream_tiles": n_local, "distinct_tiles": min(args.distinct, n_local), "detections_per_tile": ndet / B,
                       "depth": args.depth, "batch_per_gpu": B, "tile": S, "net_input": "3x800x800",
                       "parallelism": f"tile-shard x{world} (replicated weights, RCCL gather of detections to rank 0)",
                       "schedule": f"{args.schedule} x{args.streams}", "schedule_note": SCHED[args.schedule].format(n=args.streams),
                       "concurrent_forwards": conc, "detections_last_batch": ndet},
        }

        payload = B * (100 * 4 * 4 + 100 * 4 + 4 + 100 * 28 * 28 * 4)      # boxes, scores, count, mask_probs of one batch (fp32 / int32)
        line["ranks"] = {"world": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
                         "devices": [r["device"] for r in ranks_info], "distinct_gpus": len({(r["pci_bus_id"], r["uuid"], r["device"]) for r in ranks_info}),
                         "ranks": ranks_info, "gather_bytes_per_step": payload * (world - 1),
                         "gather": "torch.distributed.gather of count / boxes / scores / 28x28 probabilities to rank 0 after every batch, "
                                   "enqueued on the batch's own HIP stream (fixed-shape fp32 payload per rank and step: %d bytes)" % payload}

        def breakdown(profx, k):
            return {kk: v["ms"] / k for kk, v in profx.items() if kk not in ("executed", "_classes")}

        def class_table(profx, k, peak, detail=None, kd=1):
            """Per kernel class of the contraction family, per step: launches, executed GFLOP, GB the chosen algorithm moves,
            t_min = sum over launches of max(executed FLOPs / MFMA peak, bytes / achievable HBM rate), the resource that sets
            it; `ms` / `frac` = measured time (HIP event pair per launch, `detail` = a plain-loop region of kd steps run for
            this table) and t_min over it."""
            out = {}
            for name, c in profx["_classes"].items():
                if c["launches"] == 0:
                    continue
                t_m, t_h = c["exec_flops"] / (peak * 1e12), c["bytes"] / (HBM_ACHIEVABLE_TBS * 1e12)
                o = {"launches_per_step": c["launches"] / k, "executed_gflop": c["exec_flops"] / k / 1e9, "gbytes": c["bytes"] / k / 1e9,
                     "t_min_ms": c["tmin_ms"] / k, "bound": "mfma" if t_m >= t_h else "hbm"}
                if name == "mask_head":
                    o = {"launches_per_step": c["launches"] / k, "bound": "mfma", "note": "row count lives on the device: time only"}
                if detail is not None and detail["_classes"][name]["ms"] > 0:
                    o["ms"] = detail["_classes"][name]["ms"] / kd
                    if "t_min_ms" in o:
                        o["frac"] = o["t_min_ms"] / o["ms"]
                out[name] = o
            return out

        def roofline(profx, dtx, k, peak, pmc_name=None, detail=None, kd=1, ndet_tile=None, depth=None):
            """`roofline` object of one timed region of k steps (conv family = every MFMA contraction + the Winograd transform
            kernels of the layers on that path).

            achieved / frac: FLOPs the MFMA pipe really EXECUTED (a Winograd F(4x4,3x3) layer issues 1/4 of its direct-convolution
            multiplies, times its tile padding; F(2x2,3x3) 4/9) over the time — always <= peak. `effective_tflops` keeps the
            ALGORITHMIC rate (2 x MACs of the direct convolution), which may exceed the peak because of Winograd.
            With several forwards in flight the per-launch event spans of different streams overlap (their sum exceeds the wall
            time), so the time is the WALL time per step — a lower bound of the family's rate, since that wall time also holds
            every other kernel; `span` keeps the literal figure (FLOPs over the summed spans = what rocprofv3's per-kernel
            averages show). With one forward at a time both coincide.
            sol: the time-based speed of light — per launch t_min = max(executed FLOPs / MFMA peak, bytes moved by the chosen
            algorithm / achievable HBM rate); `frac_launches` = sum of t_min over the measured family time (one forward at a time:
            the `exclusive` object), `frac_chip` = max(sum FLOPs / peak, sum bytes / rate) over the wall time per step (under
            concurrency an HBM-bound kernel of one forward may hide under an MFMA-bound one of another, so only the chip-level
            bound is a bound there)."""
            cx = profx["conv_igemm"]
            ex = profx.get("executed", {"flops": cx["flops"], "launches": 0})
            exec_ratio = ex["flops"] / cx["flops"] if cx["flops"] > 0 else 1.0
            span_ms = cx["ms"] / k
            gflop = cx["flops"] / k / 1e9
            literal = cx["flops"] / (cx["ms"] * 1e-3) / 1e12 if cx["ms"] > 0 else 0.0
            wall = gflop / (1000.0 * dtx / k) if dtx > 0 else 0.0
            eff = wall if conc > 1 else literal
            ach = eff * exec_ratio
            traffic, traffic_src, hbm_gb = None, None, None
            if pmc_name and os.path.exists(os.path.join(ROOT, "profiles", pmc_name)):
                # HBM bytes per launch from the committed rocprofv3 --pmc passes of the plain-loop form of this command
                # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_summary2.py) — counters cannot be read in here
                with open(os.path.join(ROOT, "profiles", pmc_name)) as f:
                    pj = json.load(f)
                # per launch of THIS object's launch count (one per layer; a Winograd layer's transform kernels belong to it)
                traffic = pj["hbm_traffic_gb_per_step"] * 1e9 / max(cx["launches"] / k, 1.0)
                hbm_gb = pj["hbm_traffic_gb_per_step"]
                traffic_src = (f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; conv family: "
                               f"{pj['hbm_traffic_gb_per_step']:.2f} GB per step)")
            cls = profx["_classes"]
            static = [c for n, c in cls.items() if n != "mask_head"]
            sol_flops = sum(c["exec_flops"] for c in static) / k
            sol_bytes = sum(c["bytes"] for c in static) / k
            t_mfma, t_hbm = 1e3 * sol_flops / (peak * 1e12), 1e3 * sol_bytes / (HBM_ACHIEVABLE_TBS * 1e12)
            t_launch = sum(c["tmin_ms"] for c in static) / k
            wall_ms = 1000.0 * dtx / k
            o = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                 "effective_tflops": eff, "executed_over_algorithmic": exec_ratio,
                 "traffic": traffic, "traffic_unit": "bytes per launch (HBM, PMC)", "traffic_source": traffic_src,
                 "algorithmic_bytes_per_launch": cx["bytes"] / max(cx["launches"], 1),
                 "algorithmic_flops_per_launch": cx["flops"] / max(cx["launches"], 1),
                 "executed_flops_per_launch": ex["flops"] / max(cx["launches"], 1),
                 "kernel": "conv family (all MFMA contraction kernels + Winograd transforms)",
                 "kernel_detail": "conv_igemm_kernel / conv_pp8_kernel / conv_bd_kernel / bottleneck_tail_kernel / plane_gemm_kernel / wino43_fused_kernel "
                                  "(all trunk / FPN / RPN / box-head contractions) + the Winograd transform kernels of the layers that take that path",
                 "hbm_gbytes_per_step": hbm_gb,
                 "launches_per_step": cx["launches"] / k, "gflop_per_step": gflop, "executed_gflop_per_step": gflop * exec_ratio,
                 "algorithmic_gbytes_per_step": cx["bytes"] / k / 1e9,
                 "method": "wall" if conc > 1 else "span",
                 "method_note": ("wall: executed FLOPs per step / wall time per step (%d forwards overlap on %d HIP streams)" % (conc, conc))
                                if conc > 1 else "span: executed FLOPs / summed HIP-event spans of the family (nothing overlaps)",
                 "span": {"achieved": literal * exec_ratio, "frac": literal * exec_ratio / peak, "effective_tflops": literal,
                          "span_ms_per_step": span_ms,
                          "avg_launch_us": 1e3 * cx["ms"] / max(cx["launches"], 1), "concurrent_forwards": conc,
                          "note": "literal: FLOPs / summed HIP-event spans on each forward's own stream; spans of concurrent forwards "
                                  "overlap, so span_ms_per_step / concurrent_forwards (not span_ms_per_step) is what fits in ms_per_step"},
                 "sol": {"hbm_rate_tbs": HBM_ACHIEVABLE_TBS, "executed_gflop_per_step": sol_flops / 1e9, "moved_gbytes_per_step": sol_bytes / 1e9,
                         "t_mfma_ms": t_mfma, "t_hbm_ms": t_hbm, "t_min_launches_ms": t_launch,
                         "frac_chip": max(t_mfma, t_hbm) / wall_ms if wall_ms > 0 else 0.0,
                         "note": "t_min per launch = max(executed FLOPs / MFMA peak, bytes the chosen algorithm moves / achievable HBM rate); "
                                 "frac_chip = max(sum FLOPs / peak, sum bytes / rate) / wall time per step (the whole step, every kernel)"},
                 "classes": class_table(profx, k, peak, detail, kd)}
            o["note"] = ("achieved = FLOPs the MFMA pipe really issued (Winograd F(4x4,3x3) layers run 1/4 of their direct-convolution FLOPs "
                         "times the tile padding, F(2x2,3x3) layers 4/9) over the time; effective_tflops = algorithmic FLOPs (2 x MACs of "
                         "the direct convolution, SURVEY.md §8d) over the same time")
            o["winograd_layers_per_step"] = ex["launches"] / k
            # 261.9 (R50) / 356.6 (R101) GFLOP + 1.028 GFLOP per detection — at the stream's measured detections per
            # tile and at the 30 the survey quotes (the family's FLOPs above are the static layers: trunk, FPN, RPN, box head)
            base = 356.6 if (depth or args.depth) == 101 else 261.9
            o["algorithmic_gflop_per_tile"] = {"static": base, "at_30_detections": base + 30 * MASK_HEAD_GFLOP_PER_DET}
            if ndet_tile is not None:
                o["algorithmic_gflop_per_tile"]["detections_per_tile"] = ndet_tile
                o["algorithmic_gflop_per_tile"]["at_measured_detections"] = base + ndet_tile * MASK_HEAD_GFLOP_PER_DET
            for key in ("frac",):
                assert o[key] <= 1.0 + 1e-9, f"roofline.{key} = {o[key]} is not a fraction"
            assert o["sol"]["frac_chip"] <= 1.0 + 1e-9 and o["span"]["frac"] <= 1.0 + 1e-9, "speed-of-light fraction above 1"
            return o

        def exclusive(pp, k, peak):
            """The conv family one forward at a time (nothing else on the GPU): the kernel-quality figures."""
            cs = pp["conv_igemm"]
            exs = pp.get("executed", {"flops": cs["flops"]})
            r = exs["flops"] / cs["flops"] if cs["flops"] > 0 else 1.0
            lit = cs["flops"] / (cs["ms"] * 1e-3) / 1e12 if cs["ms"] > 0 else 0.0
            t_launch = sum(c["tmin_ms"] for n, c in pp["_classes"].items() if n != "mask_head") / k
            o = {"achieved": lit * r, "frac": lit * r / peak, "effective_tflops": lit, "avg_launch_us": 1e3 * cs["ms"] / max(cs["launches"], 1),
                 "span_ms_per_step": cs["ms"] / k, "t_min_launches_ms": t_launch,
                 "sol_frac": t_launch / (cs["ms"] / k) if cs["ms"] > 0 else 0.0,
                 "note": "the conv family with one forward at a time (the `single_stream` region of this run); sol_frac = sum over launches "
                         "of max(executed FLOPs / peak, moved bytes / HBM rate) over the family's measured time"}
            assert o["frac"] <= 1.0 + 1e-9 and o["sol_frac"] <= 1.0 + 1e-9, "exclusive roofline fraction above 1"
            return o

        peak_main = PEAK_F32_MATRIX_TFLOPS if args.precision == "fp32" else PEAK_F16_MATRIX_TFLOPS
        std = args.depth == 50 and B == 8
        kd = detail_steps
        if prof is not None:
            line["roofline"] = roofline(prof, dt, args.steps, peak_main,
                                        (f"{PMC_TAG}_pmc_conv_fp32.json" if args.precision == "fp32" else f"{PMC_TAG}_pmc_conv_fp16.json") if std else None,
                                        detail[1] if detail else None, kd, ndet_tile=ndet / B)
            line["breakdown_ms_per_step"] = breakdown(prof, args.steps)
        if extra is not None:
            dt16, prof16, ndet16 = extra
            o = {"value": tiles_total / dt16, "unit": "tiles/s", "ms_per_step": 1000.0 * dt16 / args.steps, "dtype": "f16",
                 "note": "same workload and schedule through the fp16 engine (fp16 storage, v_mfma_f32_32x32x16_f16, fp32 accumulate "
                         "and selection); parity tolerances in tests/test_engine_fp16_gpu.py",
                 "detections_last_batch": ndet16}
            if prof16 is not None:
                o["roofline"] = roofline(prof16, dt16, args.steps, PEAK_F16_MATRIX_TFLOPS, f"{PMC_TAG}_pmc_conv_fp16.json" if std else None,
                                         detail16[1] if detail16 else None, kd, ndet_tile=ndet16 / B)
                o["breakdown_ms_per_step"] = breakdown(prof16, args.steps)
                if piped16 is not None and piped16[1] is not None:
                    o["roofline"]["exclusive"] = exclusive(piped16[1], args.steps, PEAK_F16_MATRIX_TFLOPS)
                    o["single_stream"] = {"value": tiles_total / piped16[0], "unit": "tiles/s", "ms_per_step": 1000.0 * piped16[0] / args.steps,
                                          "breakdown_ms_per_step": breakdown(piped16[1], args.steps)}
            line["fp16"] = o

        def sub(res, batch, peak, depth):
            dtx, profx, ndetx, k = res
            o = {"value": k * batch * world / dtx, "unit": "tiles/s", "ms_per_step": 1000.0 * dtx / k, "steps": k,
                 "batch_per_gpu": batch, "depth": depth, "detections_last_batch": ndetx}
            if profx is not None:
                o["roofline"] = roofline(profx, dtx, k, peak, ndet_tile=ndetx / batch, depth=depth)
                o["breakdown_ms_per_step"] = breakdown(profx, k)
            return o
        if r101 is not None:
            line["r101"] = {"note": "the reference's own depth (config.py:25: mask_rcnn_R_101_FPN_3x), same stream and schedule",
                            "f32": sub(r101["fp32"], args.batch, PEAK_F32_MATRIX_TFLOPS, 101)}
            if "fp16" in r101:
                line["r101"]["f16"] = sub(r101["fp16"], args.batch, PEAK_F16_MATRIX_TFLOPS, 101)
        if b32 is not None:
            line["fp16_batch32"] = sub(b32, 32, PEAK_F16_MATRIX_TFLOPS, args.depth)
            line["fp16_batch32"]["note"] = "BASELINE configs[4]: fp16 MFMA conv path at batch 32 per GPU, R50-FPN, same tile stream"
        if piped is not None:
            line["single_stream"] = {"value": tiles_total / piped[0], "unit": "tiles/s", "ms_per_step": 1000.0 * piped[0] / args.steps,
                                     "note": "the same K steps, one forward at a time on one stream: nothing overlaps, so the event spans "
                                             "are the kernels' own durations"}
            if piped[1] is not None and prof is not None:
                line["single_stream"]["breakdown_ms_per_step"] = breakdown(piped[1], args.steps)
                # the kernel's own roofline (no other forward on the GPU): what the per-kernel rocprofv3 averages of the
                # plain-loop profile in profiles/ agree with
                line["roofline"]["exclusive"] = exclusive(piped[1], args.steps, peak_main)
        if phased is not None:
            o = {"value": tiles_total / phased[0], "unit": "tiles/s", "ms_per_step": 1000.0 * phased[0] / args.steps,
                 "note": "the same K steps through the phase pipeline (--schedule phases: contraction phases of three batches back to back on "
                         "a main stream, selection phases on side streams); its contraction spans do not overlap: literal roofline"}
            if phased[1] is not None:
                o["roofline"] = exclusive(phased[1], args.steps, peak_main)
                o["roofline"]["method"] = "span"
                o["roofline"]["note"] = "contraction spans of the phase pipeline do not overlap each other (selection kernels run underneath)"
                o["breakdown_ms_per_step"] = breakdown(phased[1], args.steps)
            line["phase_pipeline"] = o
        if two is not None:
            o = {"note": "BASELINE configs[2]: two-model path (urban model over the tiles not flagged only_forest, then the forest model over those "
                         "not flagged only_urban; one tile in three forest-only, one mixed, one urban-only), full-width R50-FPN x 2 weight sets, "
                         "same 1000x1000 tile stream and schedule; value = tiles VISITED (both models) per second; parity: "
                         "tests/test_config2_fullsize_gpu.py"}
            for pk, (dtm, nu, nf, nt, dets) in two.items():
                o["f32" if pk == "fp32" else "f16"] = {"value": (nu + nf) / dtm, "unit": "tile visits/s", "seconds": dtm, "stream_tiles": nt,
                                                       "visited_by_urban": nu, "visited_by_forest": nf,
                                                       "stream_tiles_per_s": nt / dtm, "detections_last_batch": dets}
            line["two_model"] = o
        for key, res, what in (("e2e", e2e, "the seeded random mask head: noise-like masks, a stress fixture for the host epilogue"),
                               ("e2e_crowns", e2e_c, "weights.blob_mask_head: compact crowns, what a trained segmenter hands the epilogue")):
            if res is None:
                continue
            o = {"note": "predict_tiles' model stage files to files: warm Predictor.__call__ over a synthetic GeoTIFF (window reads, H2D, resize, "
                         "forward, paste, D2H, contours, Prediction_*.json written); ratio = e2e rate / the model-stage rate of the same precision "
                         "(inputs resident in HBM, results left in HBM): the LARGER of this line's value for that precision "
                         "and `model_stage_same_weights`, the same timed region with the fixture's weights right after the e2e calls", "fixture": what}
            for pk, r in res.items():
                ref_rate = line["value"] if pk == args.precision else (line.get("fp16") or {}).get("value")
                # the larger of the line's model-stage rate (first region, cool chip) and the fixture's own rate measured right after it
                # (late regions drift by +-4 %): the ratio never flatters the pipeline; the fp16 e2e path runs batch 32, so its
                # model-stage reference is the batch-32 region when that is the larger one
                b32v = (line.get("fp16_batch32") or {}).get("value") if pk == "fp16" and r.get("batch") == 32 else None
                ref_rate = max(ref_rate or 0.0, r.get("model_stage_same_weights") or 0.0, b32v or 0.0) or None
                r["ratio_to_model_stage"] = r["value"] / ref_rate if ref_rate else None
                if "chained" in r:
                    r["chained"]["ratio_to_model_stage"] = r["chained"]["value"] / ref_rate if ref_rate else None
                o["f32" if pk == "fp32" else "f16"] = r
            line[key] = o
        for key, res, what in (("predict_tiles", pt, "weights.blob_mask_head: compact crowns"), ("predict_tiles_noise", pt_n, "seeded random mask head: noise-like masks")):
            if not res:
                continue
            o = {"note": "predict_tiles as the reference defines it (predict + stitch, detection.py:228-243), files to GeoPackage layers, warm "
                         "predictor; ratio = rate / model-stage rate of the same precision and weights (inputs resident in HBM)", "fixture": what}
            for pk, r in res.items():
                ref_rate = line["value"] if pk == args.precision else (line.get("fp16") or {}).get("value")
                b32v = (line.get("fp16_batch32") or {}).get("value") if pk == "fp16" and r.get("batch") == 32 else None
                ref_rate = max(ref_rate or 0.0, r.get("model_stage_same_weights") or 0.0, b32v or 0.0) or None
                r["ratio_to_model_stage"] = r["value"] / ref_rate if ref_rate else None
                o["f32" if pk == "fp32" else "f16"] = r
            line[key] = o
        if lzw:
            o = {"note": "files to files on an LZW-compressed raster (256 x 256 tiles, predictor 2): compressed blocks decoded on the GPU, one wave per "
                         "block, tile windows cut in HBM, next image decoded while the current one predicts; `host_reader` = the same raster through "
                         "the host decode threads (device_decode: false); ratio = rate / model-stage rate of the same precision and weights"}
            for pk, r in lzw.items():
                ref_rate = max((line["value"] if pk == args.precision else (line.get("fp16") or {}).get("value")) or 0.0, r.get("model_stage_same_weights") or 0.0) or None
                r["device"]["ratio_to_model_stage"] = r["device"]["value"] / ref_rate if ref_rate else None
                o["f32" if pk == "fp32" else "f16"] = r
            line["lzw"] = o
        if deflate:
            o = {"note": "the same region on a DEFLATE-compressed raster (zlib level 6, 256 x 256 tiles, predictor 2): tiff_inflate_blocks_kernel"}
            for pk, r in deflate.items():
                ref_rate = max((line["value"] if pk == args.precision else (line.get("fp16") or {}).get("value")) or 0.0, r.get("model_stage_same_weights") or 0.0) or None
                r["device"]["ratio_to_model_stage"] = r["device"]["value"] / ref_rate if ref_rate else None
                o["f32" if pk == "fp32" else "f16"] = r
            line["deflate"] = o
        if world == 1 and not args.no_cpu_baseline:
            sd101 = None
            if args.depth == 50 and not args.no_r101:
                log("generating R101 weights for the CPU baseline")
                sd101 = make_synthetic_state_dict(101, seed=0)
            line["cpu_baseline"] = cpu_baseline(sd, rgb_np, args.cpu_tiles, sd101)
        line["repeats"] = repeats
        line["timed_steps"] = args.steps * repeats.get("headline", 1)
        line["timed_seconds"] = dt * repeats.get("headline", 1)
        detail_path = args.detail or os.path.join(ROOT, "bench_detail.json")
        line["detail_file"] = os.path.basename(detail_path)
        emit(line, detail_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
