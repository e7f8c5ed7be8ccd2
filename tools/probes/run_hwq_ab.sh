# files-to-files regions of bench.py with 4 (default) and 8 hardware queues, same box, alternating
set -e
mkdir -p gpurun_out/r6_ad
for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-r101 --no-fp16-b32 --no-two-model --no-cpu-baseline --no-serial --no-profile --no-lzw --detail gpurun_out/r6_ad/detail_$q.json > gpurun_out/r6_ad/q$q.log 2> gpurun_out/r6_ad/q$q.err || { tail -5 gpurun_out/r6_ad/q$q.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6_ad/q$q.log") if l.startswith("{")][-1])
r=d["regions"]
print("queues $q", d["value"], {k:r[k] for k in ("fp16","e2e_f32","e2e_f16","e2e_chained_f16","e2e_crowns_f16","e2e_crowns_f16_ratio","e2e_crowns_chained_f16_ratio","predict_tiles_f32","predict_tiles_f16","predict_tiles_f32_ratio","predict_tiles_f16_ratio") if k in r})
PY
done
