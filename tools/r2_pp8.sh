cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pp8 or fp16_matches" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 800 python tools/conv_diag.py pp8 product unbalanced mfma16 > $O/where.log 2>&1; grep "prec=" $O/where.log
