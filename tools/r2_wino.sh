cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_engine_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "not variant" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for slab in 0 1024 2048 8192 100000; do
  TD_WINO_SLAB=$slab python bench.py --steps 16 --no-cpu-baseline --no-r101 --no-fp16-b32 --no-fp16 --no-serial > $O/bench_$slab.json 2> $O/bench_$slab.err; echo "slab $slab"; python tools/bench_summary.py < $O/bench_$slab.json
done
