# decode / model overlap: decoder waves spread (a workgroup per block) or packed (8 / 12 per workgroup), 4 or 8 hardware queues, low-priority decode stream
set -e
mkdir -p gpurun_out/r6_z
run() {  # env string, probe args
  echo "# $1 $2" >> gpurun_out/r6_z/ovl.txt
  env $1 timeout -k 10 300 python tools/probes/decode_overlap_probe.py $2 >> gpurun_out/r6_z/ovl.txt 2>gpurun_out/r6_z/err.txt || { tail -5 gpurun_out/r6_z/err.txt; exit 1; }
}
run "TD_INFLATE_WPB=8 GPU_MAX_HW_QUEUES=8" "codec=deflate"
run "TD_INFLATE_WPB=12 GPU_MAX_HW_QUEUES=4" "codec=deflate priority=low"
run "TD_INFLATE_WPB=12 GPU_MAX_HW_QUEUES=8" "codec=deflate priority=low"
run "GPU_MAX_HW_QUEUES=8" "codec=lzw"
run "GPU_MAX_HW_QUEUES=4" "codec=lzw priority=low"
cat gpurun_out/r6_z/ovl.txt
