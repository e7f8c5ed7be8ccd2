"""Pretty-print a bench.py JSON line from stdin: tools/bench_summary.py [label]"""
import json, sys
d = json.loads(sys.stdin.read())
lab = sys.argv[1] if len(sys.argv) > 1 else ""
r = d.get("roofline", {})
print(lab, "fp32" if d["dtype"] == "f32" else d["dtype"], round(d["value"], 1), "tiles/s  conv", round(r.get("achieved", 0), 1), "TF/s",
      {k: round(v, 2) for k, v in d.get("breakdown_ms_per_step", {}).items()})
if "fp16" in d:
    f = d["fp16"]
    print(lab, "fp16", round(f["value"], 1), "tiles/s  conv", round(f.get("roofline", {}).get("achieved", 0), 1), "TF/s",
          {k: round(v, 2) for k, v in f.get("breakdown_ms_per_step", {}).items()})
