"""bench.py's last stdout line must fit the driver's bounded stdout tail (a 25 KB line was cut and the round
went unmeasured). compact_line() is exercised on a real full result — round 3's own 25 KB line, kept under profiles/ —
and on a worst case with every optional region present."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                 "algorithmic_flops_per_launch", "kernel", "launches_per_step")


def _full():
    with open(os.path.join(ROOT, "profiles", "r03_bench_n1.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000          # the line that did not fit
    return full


def test_compact_line_fits_and_keeps_the_contract_keys():
    full = _full()
    c = bench.compact_line(full)
    s = json.dumps(c, separators=(",", ":"))
    assert len(s) < bench.COMPACT_LIMIT, len(s)
    for k in REQUIRED:
        assert c.get(k) is not None or k == "vs_baseline", k
    for k in ROOFLINE_KEYS:
        assert k in c["roofline"], k
    assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert c["roofline"]["frac"] <= 1.0
    assert c["value"] == pytest.approx(full["value"], rel=1e-5)
    assert c["cpu_baseline"]["kind"] in ("port", "reference") and c["cpu_baseline"]["cores"] >= 1
    assert "workload" in c["config"] and "model" not in c["config"]
    # one scalar per extra region the full result holds
    reg = c["regions"]
    assert reg["fp16"] == pytest.approx(full["fp16"]["value"], rel=1e-3)
    assert reg["r101_f32"] == pytest.approx(full["r101"]["f32"]["value"], rel=1e-3)
    assert reg["r101_f16"] == pytest.approx(full["r101"]["f16"]["value"], rel=1e-3)
    assert reg["fp16_batch32"] == pytest.approx(full["fp16_batch32"]["value"], rel=1e-3)
    assert reg["two_model_f32"] == pytest.approx(full["two_model"]["f32"]["value"], rel=1e-3)
    assert reg["e2e_f16_ratio"] == pytest.approx(full["e2e"]["f16"]["ratio_to_model_stage"], rel=1e-3)
    assert all(isinstance(v, (int, float)) for v in reg.values())      # scalars only: no nested objects, no prose
    assert c["ranks"]["world"] == 1 and "ranks" not in c["ranks"]


def test_compact_line_worst_case_with_eight_ranks_and_every_region(tmp_path):
    full = _full()
    full["n_gpus"] = 8
    full["ranks"] = {"world": 8, "backend": "nccl", "devices": list(range(8)), "distinct_gpus": 8,
                     "ranks": [{"rank": r, "uuid": "x" * 40, "pci_bus_id": "0000:f5:00.0"} for r in range(8)],
                     "gather_bytes_per_step": 7 * 2521632, "gather": "y" * 400}
    full["cpu_baseline"]["extra"] = {"r50_b1": 1.157234, "r50_b8": 1.3123, "r101_b1": 0.81234, "r101_b8": 0.9, "r101_tiles": 8}
    full["cpu_baseline"]["sample"] = "z" * 1000
    full["e2e_crowns"] = json.loads(json.dumps(full["e2e"]))
    full["timed_steps"], full["timed_seconds"], full["detail_file"] = 160, 2.0123456, "bench_detail.json"
    # round 5: the line says what ran and carries the predict + stitch rate
    full["config"].update(stream_tiles=200, distinct_tiles=16, detections_per_tile=18.5)
    full["roofline"]["traffic_source"] = "profiles/r05_pmc_conv_fp32.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; conv family: 27.05 GB per step)"
    full["roofline"]["algorithmic_gflop_per_tile"] = {"static": 261.9, "at_30_detections": 292.74, "detections_per_tile": 18.5,
                                                      "at_measured_detections": 280.918}
    for key in ("predict_tiles", "predict_tiles_noise"):
        full[key] = {"note": "n" * 300, "f32": {"value": 612.3456, "ratio_to_model_stage": 0.93123, "images": 24},
                     "f16": {"value": 1912.3456, "ratio_to_model_stage": 0.90123, "images": 48}}
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        s = bench.emit(full, str(tmp_path / "bench_detail.json"))
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1 and lines[0] == s and len(s) < bench.COMPACT_LIMIT      # stdout = the compact line alone
    c = json.loads(s)
    assert c["ranks"] == {"world": 8, "backend": "nccl", "devices": list(range(8)), "distinct_gpus": 8,
                          "gather_bytes_per_step": 7 * 2521632}
    assert c["cpu_baseline"]["extra"]["r101_b8"] == 0.9 and len(c["cpu_baseline"]["sample"]) <= 200
    assert c["timed_steps"] == 160 and c["detail"] == "bench_detail.json"
    assert "e2e_crowns_f16_ratio" in c["regions"]
    assert c["config"]["stream_tiles"] == 200 and c["config"]["distinct_tiles"] == 16 and c["config"]["detections_per_tile"] == 18.5
    assert c["roofline"]["traffic_source"] == "profiles/r05_pmc_conv_fp32.json"
    assert c["roofline"]["algorithmic_gflop_per_tile"]["at_measured_detections"] == pytest.approx(280.92, rel=1e-4)
    assert c["roofline"]["algorithmic_gflop_per_tile"]["at_30_detections"] == pytest.approx(292.74, rel=1e-4)
    assert c["regions"]["predict_tiles_f32"] == pytest.approx(612.3, rel=1e-3) and c["regions"]["predict_tiles_f16_ratio"] == pytest.approx(0.9012, rel=1e-3)
    assert c["regions"]["predict_tiles_noise_f16"] == pytest.approx(1912, rel=1e-3)
    assert json.load(open(tmp_path / "bench_detail.json"))["roofline"]["sol"]          # everything else lives in the detail file
    assert "full result" in err.getvalue()


def test_emit_refuses_an_oversize_line():
    full = _full()
    full["config"]["workload"] = "w" * 5000
    with pytest.raises(AssertionError, match="compact bench line"):
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            bench.emit(full, None)
