# round-6 re-measurement after the chunk decoder and the batch-level host calls: host cost table, decode table, decode kernel stats
set -e
O=gpurun_out/r6_t
rm -rf $O && mkdir -p $O
nproc > $O/nproc.txt
for a in "fp16" "fp16" "fp32" "fp16 raster=lzw" "fp16 raster=lzw" "fp16 device_raster=all"; do
  echo "## $a" >> $O/host.txt
  timeout -k 10 400 python tools/host_cost.py $a images=4 >> $O/host.txt 2>$O/host.err || { tail -5 $O/host.err; exit 1; }
done
echo host done
for args in "codec=lzw side=9000" "codec=lzw side=9000 data=noise" "codec=lzw side=9000 strip=1" "codec=lzw side=20000" "codec=lzw side=5000" "codec=lzw side=5000 tile=128" "codec=lzw side=5000 data=flat" "codec=deflate side=9000" "codec=deflate side=9000 strip=1" "codec=deflate side=20000" "codec=deflate side=5000" "codec=deflate side=9000 data=noise"; do
  echo "## $args" >> $O/dec.txt
  timeout -k 10 400 python tools/raster_decode_bench.py $args >> $O/dec.txt 2>$O/dec.err || { tail -5 $O/dec.err; exit 1; }
done
echo "## one by one: codec=lzw side=9000" >> $O/dec.txt
TD_LZW_ONE_BY_ONE=1 timeout -k 10 400 python tools/raster_decode_bench.py codec=lzw side=9000 >> $O/dec.txt 2>$O/dec.err
echo decode done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/decode_lzw -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=lzw side=20000 > $R/$O/r06_decode_lzw.json 2> $R/$O/decode_lzw.err || exit 1
rocprofv3 --kernel-trace --stats -d $R/$O/decode_deflate -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=deflate side=9000 > $R/$O/r06_decode_deflate.json 2> $R/$O/decode_deflate.err || exit 1
cp $(find $R/$O/decode_lzw -name "*kernel_stats.csv" | head -1) $R/$O/r06_decode_lzw_kernel_stats.csv
cp $(find $R/$O/decode_deflate -name "*kernel_stats.csv" | head -1) $R/$O/r06_decode_deflate_kernel_stats.csv
rm -rf $R/$O/decode_lzw $R/$O/decode_deflate
echo stats done
