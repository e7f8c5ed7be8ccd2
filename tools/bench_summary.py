"""One-line-per-region view of bench.py's FULL result: python tools/bench_summary.py bench_detail.json (the detail file
bench.py writes beside its compact stdout line), or the full JSON on stdin."""
import json
import sys

if len(sys.argv) > 1:
    j = json.load(open(sys.argv[1]))
else:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])


def show(tag, o):
    r = o.get("roofline") or {}
    b = {k: round(v, 2) for k, v in (o.get("breakdown_ms_per_step") or {}).items()}
    sp = r.get("span") or {}
    ex = r.get("exclusive") or {}
    extra = f" span {sp.get('frac', 0):.3f}" + (f" exclusive {ex['frac']:.3f}" if ex else "")
    print(f" {tag:14s} {o['value']:8.1f} tiles/s  {o['ms_per_step']:7.2f} ms/step  conv {r.get('achieved', 0):7.1f} TF/s (frac {r.get('frac', 0):.3f}{extra})  {b}")


show("fp32", j)
for k, tag in (("fp16", "fp16"), ("fp16_batch32", "fp16 b32"), ("single_stream", "plain loop"), ("phase_pipeline", "phase pipeline")):
    if k in j:
        show(tag, j[k])
if "r101" in j:
    for k, o in j["r101"].items():
        if isinstance(o, dict):
            show("r101 " + k, o)
if "cpu_baseline" in j:
    print(" cpu_baseline", j["cpu_baseline"]["value"], j["cpu_baseline"]["unit"], j["cpu_baseline"]["cores"], "cores")
