# bash tools/probes/tail_probe.sh [fp16|fp32]: bottleneck_tail_kernel on the res2 shape — product build and the two ablation builds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}
O=$R/gpurun_out/tailprobe_$P
rm -rf $O && mkdir -p $O
for fd in ${DIAGS:-0:0 1:0 1:1 1:2 1:4 1:8 1:12}; do
  export TD_TAIL_FAST=${fd%%:*}; d=${fd##*:}
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t$fd -o t --output-format csv -- python3 $R/tools/probes/tail_probe.py $P $d > $O/p$fd.log 2>&1 || { tail -5 $O/p$fd.log; exit 1; }
  python3 - "$(find $O/t$fd -name '*kernel_trace.csv' | head -1)" $fd <<'PY'
import csv, sys
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "bottleneck_tail" in r["Kernel_Name"])
print(f"TD_TAIL_FAST:TD_TAIL_DIAG={sys.argv[2]}: n={len(d)} min {d[0]:.1f} us median {d[len(d)//2]:.1f} us")
PY
done | tee $O/summary.txt
