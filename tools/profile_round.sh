# Regenerates the judged artefacts of profiles/ on the GPU box: bash tools/profile_round.sh [round tag, default r02]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-serial --no-r101 --no-fp16-b32 --no-e2e --no-two-model --detail $O/warm_detail.json > $O/warm.json 2> $O/warm.err || exit 1      # fills the tile-choice cache
python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-serial --no-r101 --no-fp16-b32 --no-e2e --no-two-model --no-pipeline --detail $O/warm2_detail.json > $O/warm2.json 2> $O/warm2.err || exit 1
# (1) the default command's schedule: three forwards overlap on three streams — per-kernel durations here are the spans under that concurrency
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-serial --no-r101 --no-fp16-b32 --no-e2e --no-two-model --detail $O/bench_profiled_detail.json > $O/bench_profiled.json 2> $O/bench_profiled.err || exit 1
echo stats done
# (2) the same steps one forward at a time (what roofline.exclusive and the single_stream region of the default command measure)
rocprofv3 --kernel-trace --stats -d $O/stats_plain -o s --output-format csv -- python3 $R/bench.py --schedule plain --steps 16 --warmup 3 --no-cpu-baseline --no-r101 --no-fp16-b32 --no-e2e --no-two-model --detail $O/bench_plain_profiled_detail.json > $O/bench_plain_profiled.json 2> $O/bench_plain_profiled.err || exit 1
echo plain stats done
for prec in fp32 fp16; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${prec}_$tag -o p --output-format csv -- python3 $R/bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp16 --no-pipeline --no-serial --no-e2e --no-two-model --min-seconds 0 --detail $O/pmc_${prec}_$tag.detail.json > $O/pmc_${prec}_$tag.log 2>&1 || exit 1
    echo pmc $prec $tag done
  done
done
cd $R
ALG32=$(python3 -c "import json;print(json.load(open('$O/warm_detail.json'))['roofline']['algorithmic_gbytes_per_step'])")
ALG16=$(python3 -c "print($ALG32/2)")
python3 tools/pmc_summary2.py $O/pmc_fp32_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv $O/pmc_fp32_FETCH_SIZE/p_counter_collection.csv $O/pmc_fp32_WRITE_SIZE/p_counter_collection.csv 2 $ALG32 > $O/${TAG}_pmc_conv_fp32.json
python3 tools/pmc_summary2.py $O/pmc_fp16_SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv $O/pmc_fp16_FETCH_SIZE/p_counter_collection.csv $O/pmc_fp16_WRITE_SIZE/p_counter_collection.csv 2 $ALG16 > $O/${TAG}_pmc_conv_fp16.json
cp $O/tune.txt $O/${TAG}_tile_choices.txt
cp $O/stats/s_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv 2>/dev/null || cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
cp $O/bench_profiled.json $O/${TAG}_bench_profiled_run.json
cp $O/bench_profiled_detail.json $O/${TAG}_bench_profiled_detail.json
cp $(find $O/stats_plain -name "*kernel_stats.csv" | head -1) $O/${TAG}_plain_kernel_stats.csv
cp $O/bench_plain_profiled.json $O/${TAG}_plain_profiled_run.json
cp $O/bench_plain_profiled_detail.json $O/${TAG}_plain_profiled_detail.json
python3 -c "import json;d=json.load(open('$O/${TAG}_pmc_conv_fp32.json'));print('fp32 conv family', d['conv_family'])"
python3 -c "import json;d=json.load(open('$O/${TAG}_pmc_conv_fp16.json'));print('fp16 conv family', d['conv_family'])"
# (round 6) the raster decode kernels: per-kernel durations of an LZW and a DEFLATE raster (256 x 256 tiles, predictor 2) decoded five times
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/decode_lzw -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=lzw side=20000 > $O/${TAG}_decode_lzw.json 2> $O/decode_lzw.err || exit 1
rocprofv3 --kernel-trace --stats -d $O/decode_deflate -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=deflate side=9000 > $O/${TAG}_decode_deflate.json 2> $O/decode_deflate.err || exit 1
cp $(find $O/decode_lzw -name "*kernel_stats.csv" | head -1) $O/${TAG}_decode_lzw_kernel_stats.csv
cp $(find $O/decode_deflate -name "*kernel_stats.csv" | head -1) $O/${TAG}_decode_deflate_kernel_stats.csv
echo decode stats done
