set -e
mkdir -p gpurun_out/r6_s
rm -f gpurun_out/r6_s/dec.txt
timeout -k 5 200 python tools/probes/lzw_chunk_debug.py | grep chunks
timeout -k 10 600 python -m pytest tests/test_tiff_decode_gpu.py -x -q > gpurun_out/r6_s/tests.log 2>&1 || { tail -30 gpurun_out/r6_s/tests.log; exit 1; }
tail -3 gpurun_out/r6_s/tests.log
for ring in large; do
for args in "codec=lzw side=9000" "codec=lzw side=20000" "codec=lzw side=9000 strip=1" "codec=lzw side=5000" "codec=lzw side=9000 data=noise" "codec=lzw side=5000 data=flat"; do
  echo "ring=$ring $args" >> gpurun_out/r6_s/dec.txt
  TD_DECODE_RING=$ring timeout -k 10 300 python tools/raster_decode_bench.py $args >> gpurun_out/r6_s/dec.txt 2>gpurun_out/r6_s/dec.err || { tail -5 gpurun_out/r6_s/dec.err; exit 1; }
done
done
cut -c1-400 gpurun_out/r6_s/dec.txt | grep -o "ring=.*\|\"blocks\": [0-9]*\|\"kernel_ms\": \[[^]]*\]\|\"kernel_gb_per_s\": [0-9.]*\|\"strings_through_memory\": [0-9]*" | paste -sd' ' | sed 's/ring=/\nring=/g'
