cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc16_$tag -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --precision fp16 --no-pipeline > $R/gpurun_out/pmc16_$tag.log 2>&1 || exit 1
  echo done $tag
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc16_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv gpurun_out/pmc16_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc16_WRITE_SIZE/p_counter_collection.csv 50 2 > gpurun_out/r01_pmc_conv_igemm_fp16.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r01_pmc_conv_igemm_fp16.json"))
print({k:v for k,v in d.items() if k!="layers"})
PY
