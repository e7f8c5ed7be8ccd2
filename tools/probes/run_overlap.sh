set -e
mkdir -p gpurun_out/r6_r
for env in "TD_DECODE_RING=large" "TD_DECODE_RING=small" "TD_DECODE_RING=small GPU_MAX_HW_QUEUES=8"; do
  env $env timeout -k 10 300 python tools/probes/decode_overlap_probe.py >> gpurun_out/r6_r/ovl.txt 2>gpurun_out/r6_r/err.txt || { tail -5 gpurun_out/r6_r/err.txt; exit 1; }
done
env TD_DECODE_RING=small timeout -k 10 300 python tools/probes/decode_overlap_probe.py priority=low >> gpurun_out/r6_r/ovl.txt 2>gpurun_out/r6_r/err.txt
env TD_DECODE_RING=small timeout -k 10 300 python tools/probes/decode_overlap_probe.py codec=deflate >> gpurun_out/r6_r/ovl.txt 2>gpurun_out/r6_r/err.txt
cat gpurun_out/r6_r/ovl.txt
