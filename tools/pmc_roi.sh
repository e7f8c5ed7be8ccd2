#!/bin/bash
# PMC passes over the plain-loop forward, summarised for the RoIAlign launches only (tools/roi_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export TD_TUNE_CACHE=/tmp/roi_tune.txt
P=${1:-fp16}
python3 $R/tools/roi_bench.py $P 2 > /dev/null 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcroi_$i -o p --output-format csv -- python3 $R/tools/roi_bench.py $P 2 > $R/gpurun_out/pmcroi_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmcroi_$i.log; echo FAILED set $i; continue; }
  python3 - $R/gpurun_out/pmcroi_$i/p_counter_collection.csv <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "roi_align" not in r["Kernel_Name"]:
        continue
    e = rows.setdefault(int(r["Dispatch_Id"]), {"grid": int(r["Grid_Size"]) if "Grid_Size" in r else 0, "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
for k, e in list(rows.items())[-2:]:
    print(k, e)
PY
done
