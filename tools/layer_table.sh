# per-layer conv table (rocprofv3 kernel trace of a plain-loop bench) for one precision: bash tools/layer_table.sh fp16 [depth]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}; D=${2:-50}
O=$R/gpurun_out/layers_$P$D
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
# experiments: TD_SEED_TUNE="prec cout cin kh*16+kw rows flags cfg[;...]" pins the block tile of one or more launch shapes
if [ -n "$TD_SEED_TUNE" ]; then echo "$TD_SEED_TUNE" | tr ";" "\n" > $O/tune.txt; fi
python3 $R/bench.py --precision $P --depth $D --steps 2 --warmup 2 --no-cpu-baseline --no-serial --no-fp16 --no-pipeline --no-profile --no-e2e --no-two-model --detail $O/warm_detail.json > $O/warm.json 2> $O/warm.err || exit 1
rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/bench.py --precision $P --depth $D --steps 3 --warmup 1 --no-cpu-baseline --no-serial --no-fp16 --no-pipeline --no-profile --no-e2e --no-two-model --min-seconds 0 --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err || exit 1
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_layers.py $F $D $P > $O/layers.txt
cp $O/tune.txt $O/tile_choices.txt
tail -80 $O/layers.txt
