# after "blocks by ticket": decode tests, the decode table, the overlap probe
set -e
O=gpurun_out/r6_ag
rm -rf $O && mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_tiff_decode_gpu.py -x -q > $O/tiff.log 2>&1 || { tail -20 $O/tiff.log; exit 1; }
tail -1 $O/tiff.log
for args in "codec=lzw side=9000" "codec=lzw side=20000" "codec=lzw side=5000" "codec=lzw side=9000 strip=1" "codec=deflate side=9000" "codec=deflate side=20000" "codec=deflate side=5000" "codec=deflate side=9000 strip=1"; do
  echo "## $args" >> $O/dec.txt
  timeout -k 10 300 python tools/raster_decode_bench.py $args 2>$O/dec.err | tail -1 >> $O/dec.txt || { tail -5 $O/dec.err; exit 1; }
done
cut -c1-330 $O/dec.txt
for c in deflate lzw; do
  timeout -k 10 300 python tools/probes/decode_overlap_probe.py codec=$c priority=low 2>$O/ovl.err | tail -1 >> $O/ovl.txt || { tail -5 $O/ovl.err; exit 1; }
done
python - <<PY
import json
for l in open("$O/ovl.txt"):
    d=json.loads(l); print(d["codec"], "alone", d["decode_kernel_ms_alone"], "model", d["model_tiles_per_s_alone"], "while decoding", d["model_tiles_per_s_while_decoding"], "decode ms", sorted(d["decode_kernel_ms_while_model_runs"])[len(d["decode_kernel_ms_while_model_runs"])//2])
PY
