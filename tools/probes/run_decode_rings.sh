set -e
mkdir -p gpurun_out/r6_o
timeout -k 10 900 python -m pytest tests/test_tiff_decode_gpu.py -x -q > gpurun_out/r6_o/tests.log 2>&1 || { tail -30 gpurun_out/r6_o/tests.log; exit 1; }
tail -3 gpurun_out/r6_o/tests.log
for ring in large small; do
for args in "codec=lzw side=9000" "codec=deflate side=9000" "codec=lzw side=20000" "codec=deflate side=20000" "codec=lzw side=9000 strip=1" "codec=deflate side=9000 strip=1" "codec=deflate side=5000" "codec=lzw side=5000"; do
  echo "ring=$ring $args" >> gpurun_out/r6_o/dec.txt
  TD_DECODE_RING=$ring timeout -k 10 300 python tools/raster_decode_bench.py $args >> gpurun_out/r6_o/dec.txt 2>gpurun_out/r6_o/dec.err || { tail -5 gpurun_out/r6_o/dec.err; exit 1; }
done
done
cat gpurun_out/r6_o/dec.txt | cut -c1-400
