# instruction mix of the DEFLATE kernel (rocprofv3 --pmc), 5000 x 5000 raster = 400 blocks, large ring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_v
rm -rf $O && mkdir -p $O
export TD_DECODE_RING=large
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR -d $O/pmc -o p --output-format csv -- python3 $R/tools/raster_decode_bench.py codec=deflate side=5000 > $O/out.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
for k, v in acc.items():
    print(k, dict(v))
PY
