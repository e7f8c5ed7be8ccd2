# bash tools/probes/wino_fold_probe.sh [shape ...]: kernel durations of the three-launch F(4x4) form vs wino43_fused_kernel per layer shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wfold
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/tools/probes/wino_fold_probe.py "$@" > $O/probe.log 2>&1 || { tail -5 $O/probe.log; exit 1; }
python3 - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" <<'PY' | tee $O/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive launches: a layer run = input transform, then either (gemm, output) or fused
cur = None
out = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = None
    for k in ("wino43_input_kernel", "wino43_output_kernel", "wino43_fused_kernel", "conv_igemm_kernel", "plane_gemm_kernel", "conv_bd_kernel"):
        if k in n:
            key = k
    if key is None:
        continue
    g = (r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""))
    if key == "wino43_fused_kernel":
        import re
        m = re.search(r"ILi(\d+)ELi(\d+)E", n) or re.search(r"<(\d+), (\d+)>", n)
        key += f"<{m.group(1)},{m.group(2)}>" if m else ""
    out.setdefault((key, g), []).append(d)
for (k, g), v in out.items():
    v = sorted(v)
    print(f"{k:24s} grid {g[0]:>10s} wg {g[1]:>5s}  n={len(v):2d}  min {v[0]:8.1f} us  median {v[len(v)//2]:8.1f} us")
PY
