# round-2 baseline on the GPU box: full GPU suite, fp16 diagnostics, bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -5 $O/gputests.log
python tools/fp16_diag.py 5 > $O/fp16_diag.log 2>&1; tail -40 $O/fp16_diag.log
python bench.py --steps 16 > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py < $O/bench.json
