set -e
mkdir -p gpurun_out/r6_u
rm -f gpurun_out/r6_u/dec.txt
timeout -k 10 600 python -m pytest tests/test_tiff_decode_gpu.py -x -q > gpurun_out/r6_u/tests.log 2>&1 || { tail -30 gpurun_out/r6_u/tests.log; exit 1; }
tail -3 gpurun_out/r6_u/tests.log
for ring in small; do
for args in "codec=deflate side=9000" "codec=deflate side=9000 strip=1" "codec=deflate side=20000" "codec=deflate side=5000" "codec=deflate side=9000 data=noise"; do
  echo "## ring=$ring $args" >> gpurun_out/r6_u/dec.txt
  TD_DECODE_RING=$ring timeout -k 10 300 python tools/raster_decode_bench.py $args >> gpurun_out/r6_u/dec.txt 2>gpurun_out/r6_u/dec.err || { tail -5 gpurun_out/r6_u/dec.err; exit 1; }
done
done
cut -c1-400 gpurun_out/r6_u/dec.txt | grep -o "## .*\|\"blocks\": [0-9]*\|\"kernel_ms\": \[[^]]*\]\|\"kernel_gb_per_s\": [0-9.]*" | paste -sd' ' | sed 's/## /\n## /g'
