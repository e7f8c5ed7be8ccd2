#!/bin/bash
# RoIAlign variant sweep (plain loop, per-category ms per step)
export TD_TUNE_CACHE=/tmp/roi_tune.txt
for d in 1 2 3 4; do
  TD_ROI_DEPTH=$d timeout -k 10 200 python3 tools/probes/roi_bench.py fp16 8 2>/dev/null || exit 1
done
for d in 1 2 3 4 6; do
  TD_ROI_DEPTH=$d timeout -k 10 200 python3 tools/probes/roi_bench.py fp32 8 2>/dev/null || exit 1
done
