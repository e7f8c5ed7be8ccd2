cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "winograd" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -30 $O/tests.log | cut -c1-200
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/tests2.log 2>&1; echo "rc=$?" >> $O/tests2.log; tail -3 $O/tests2.log
grep -q "rc=0" $O/tests2.log || exit 1
for f in 1 0; do
  TD_WINO_FUSED=$f python bench.py --steps 16 --no-cpu-baseline --no-r101 --no-fp16-b32 --no-fp16 --no-serial > $O/bench_$f.json 2> $O/bench_$f.err; echo "fused $f"; python tools/bench_summary.py < $O/bench_$f.json
done
