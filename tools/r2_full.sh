cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -4 $O/gputests.log
