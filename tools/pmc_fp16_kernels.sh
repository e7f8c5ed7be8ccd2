# per-kernel PMC view of the plain loop: bash tools/pmc_fp16_kernels.sh [precision]
# Seven counter passes (gfx950 slots per pass: 8 SQ, 4 TCC, 2 GRBM — MI355X_MICROARCH.md "rocprofv3 PMC slots"; round 3's 4th set
# mixed three TA counters with two SQ and one GRBM counter and rocprofv3 aborted with "error code 38: Request exceeds the
# capabilities of the hardware to collect", then sat in its signal handler until the box's silence timer killed the call;
# round 4 found that the three TA counters ALONE abort the same way, so every TA counter gets a pass of its own. A pass takes
# ~4 s; an aborted rocprofv3 hangs in its handler, hence the 120-s timeout per pass).
# Every pass runs under its own `timeout -k`, appends a line to progress.txt, and the loop STOPS at the first pass that fails.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}
O=$R/gpurun_out/pmck_$P
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
COMMON="--no-cpu-baseline --no-serial --no-fp16 --no-pipeline --no-profile --no-e2e --no-two-model --min-seconds 0"
echo "$(date +%T) warm-up run (fills the tile-choice cache)" | tee -a $O/progress.txt
timeout -k 10 400 python3 $R/bench.py --precision $P --steps 2 --warmup 2 $COMMON --detail $O/warm_detail.json > $O/warm.json 2> $O/warm.err || { echo warm-up failed; tail -5 $O/warm.err; exit 1; }
i=0
ok=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "$(date +%T) pass $i: $set" | tee -a $O/progress.txt
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --precision $P --steps 1 --warmup 1 $COMMON --detail $O/p$i.detail.json > $O/p$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "$(date +%T) pass $i FAILED (rc $rc): stopping; last lines of $O/p$i.log:" | tee -a $O/progress.txt
    tail -8 $O/p$i.log | tee -a $O/progress.txt
    ok=0
    break
  fi
  echo "$(date +%T) pass $i done" | tee -a $O/progress.txt
done
python3 $R/tools/pmc_kernels.py $(find $O -name "*counter_collection.csv" | sort) > $O/kernels.txt 2>&1
echo "passes completed: $(find $O -name '*counter_collection.csv' | wc -l) of 7" | tee -a $O/progress.txt
tail -3 $O/kernels.txt
[ $ok -eq 1 ]
