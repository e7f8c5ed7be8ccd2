# per-kernel PMC view of the fp16 plain loop: bash tools/pmc_fp16_kernels.sh [precision]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}
O=$R/gpurun_out/pmck_$P
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
python3 $R/bench.py --precision $P --steps 2 --warmup 2 --no-cpu-baseline --no-serial --no-fp16 --no-pipeline --no-profile --no-e2e --no-two-model > $O/warm.json 2> $O/warm.err || exit 1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --precision $P --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 --no-pipeline --no-serial --no-e2e --no-two-model > $O/p$i.log 2>&1 || { echo pass $i failed; tail -5 $O/p$i.log; }
done
python3 $R/tools/pmc_kernels.py $(find $O -name "*counter_collection.csv" | sort) > $O/kernels.txt 2>&1
tail -3 $O/kernels.txt
