#!/bin/bash
# kernel trace of the pipelined bench for one precision + main-stream timeline: bash tools/timeline.sh fp16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-fp16}
O=$R/gpurun_out/tl_$P
rm -rf $O && mkdir -p $O
export TD_TUNE_CACHE=$O/tune.txt
X=""; [ $P = fp32 ] && X="--no-fp16"
python3 $R/bench.py --precision $P --steps 3 --warmup 3 --no-cpu-baseline --no-serial --no-r101 --no-fp16-b32 --no-profile $X --detail $O/warm_detail.json > $O/warm.json 2> $O/warm.err || exit 1
rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/bench.py --precision $P --steps 12 --warmup 3 --no-cpu-baseline --no-serial --no-r101 --no-fp16-b32 --no-profile --min-seconds 0 --detail $O/bench_detail.json $X > $O/bench.json 2> $O/bench.err || exit 1
python3 $R/tools/bench_summary.py < $O/bench.json
python3 $R/tools/timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 8
