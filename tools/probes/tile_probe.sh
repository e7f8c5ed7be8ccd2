# bash tools/probes/tile_probe.sh [fp16|fp32] [cfg,cfg,...] [shape,shape,...]: kernel us per (layer shape, block tile id)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tileprobe_${1:-fp16}
rm -rf $O && mkdir -p $O
export TILE_PROBE_PLAN=$O/plan.txt
timeout -k 10 500 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/probes/tile_probe.py "$@" > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
python3 - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" $O/plan.txt <<'PY' | tee $O/summary.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_igemm_kernel", "conv_pp8_kernel", "conv_bd_kernel", "plane_gemm_kernel", "conv_bs_kernel"))]
plan = [ln.split() for ln in open(sys.argv[2])]
i = 0
table = {}
for name, cfg, ok in plan:
    n = int(ok)
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + n])
    kn = rows[i]["Kernel_Name"] if n else ""
    i += n
    short = "pp8" if "conv_pp8" in kn else "bs" if "conv_bs" in kn else ("bd" if "conv_bd" in kn else ("plane" if "plane_gemm" in kn else "igemm"))
    table.setdefault(name, []).append((int(cfg), d[0] if d else float("nan"), short))
for name, lst in table.items():
    best = min(v for _, v, _ in lst)
    print(name + ": " + "  ".join(f"{c}:{v:.1f}{'*' if v == best else ''}({s})" for c, v, s in lst))
PY
