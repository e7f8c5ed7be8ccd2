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
for key in ("two_model", "e2e", "e2e_crowns", "predict_tiles", "predict_tiles_noise"):
    for p in ("f32", "f16"):
        o = (j.get(key) or {}).get(p)
        if not isinstance(o, dict):
            continue
        extra = ""
        if "chained" in o:
            extra += f"  chained {o['chained']['value']:.0f} ({o['chained'].get('ratio_to_model_stage') or 0:.3f})"
        if "same_walk_without_stitching_seconds" in o:
            extra += f"  walk {o['walk_seconds_max']:.3f} s, without stitching {o['same_walk_without_stitching_seconds']:.3f} s, stitch threads {o['stitch_thread_seconds_max']:.3f} s"
        ratio = o.get("ratio_to_model_stage")
        print(f" {key + ' ' + p:24s} {o['value']:8.1f} {o.get('unit', 'tiles/s')}" + (f"  ratio {ratio:.3f}" if ratio else "") + extra)
