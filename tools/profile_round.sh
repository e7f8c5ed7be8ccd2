# Regenerates the judged artefacts of profiles/ on the GPU box (run as: bash tools/profile_round.sh).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-serial > $O/warm.json 2> $O/warm.err || exit 1      # fills the tile-choice cache
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-serial > $O/bench_profiled.json 2> $O/bench_profiled.err || exit 1
echo stats done
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$tag -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 --no-pipeline > $O/pmc_$tag.log 2>&1 || exit 1
  echo pmc $tag done
done
cd $R
python tools/pmc_summary.py $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv 50 4 > $O/r01_pmc_conv_igemm.json
cp $O/tune.txt $O/r01_tile_choices.txt
python3 bench.py > $O/r01_bench_n1.json 2> $O/bench_final.err
python tools/bench_summary.py < $O/r01_bench_n1.json
grep -c conv_igemm $O/stats/s_kernel_stats.csv
