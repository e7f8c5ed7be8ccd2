# bash tools/probes/bs_probe.sh [fp16|fp32] [shape,...]: conv_bs_kernel, product build and ablation builds (DIAGS="0 1 2 ...")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}
O=$R/gpurun_out/bsprobe_$P
rm -rf $O && mkdir -p $O
for d in ${DIAGS:-0 1 2 4 8 3 7 15}; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t$d -o t --output-format csv -- python3 $R/tools/probes/bs_probe.py $P $d $2 > $O/p$d.log 2>&1 || { tail -5 $O/p$d.log; exit 1; }
  python3 - "$(find $O/t$d -name '*kernel_trace.csv' | head -1)" $d <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv_bs_kernel" in r["Kernel_Name"]]
print(f"TD_BS_DIAG={sys.argv[2]}: " + "  ".join(f"{min(d[i:i + 4]):.1f}" for i in range(0, len(d), 4)))
PY
done | tee $O/summary.txt
