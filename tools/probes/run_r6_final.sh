# final measurements of round 6 after the decoder waves were packed and the decode stream made low priority: the default bench line, the
# decode table, the decode kernels' rocprofv3 stats
set -e
O=gpurun_out/r6_final
rm -rf $O && mkdir -p $O
nproc > $O/nproc.txt
timeout -k 10 500 python3 bench.py --detail $O/r06_bench_n1_detail.json > $O/bench.log 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
grep "^{" $O/bench.log | tail -1 > $O/r06_bench_n1.json
echo bench done
for args in "codec=lzw side=9000" "codec=lzw side=20000" "codec=lzw side=5000" "codec=deflate side=9000" "codec=deflate side=20000" "codec=deflate side=5000"; do
  echo "## $args" >> $O/dec.txt
  timeout -k 10 300 python tools/raster_decode_bench.py $args 2>$O/dec.err | tail -1 >> $O/dec.txt || { tail -5 $O/dec.err; exit 1; }
done
echo decode done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/decode_lzw -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=lzw side=20000 > $R/$O/r06_decode_lzw.json 2> $R/$O/decode_lzw.err || exit 1
rocprofv3 --kernel-trace --stats -d $R/$O/decode_deflate -o s --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=deflate side=9000 > $R/$O/r06_decode_deflate.json 2> $R/$O/decode_deflate.err || exit 1
cp $(find $R/$O/decode_lzw -name "*kernel_stats.csv" | head -1) $R/$O/r06_decode_lzw_kernel_stats.csv
cp $(find $R/$O/decode_deflate -name "*kernel_stats.csv" | head -1) $R/$O/r06_decode_deflate_kernel_stats.csv
rm -rf $R/$O/decode_lzw $R/$O/decode_deflate
echo stats done
