# e2e_lzw region of bench.py with the large / small decode rings (TD_DECODE_RING), same box
set -e
mkdir -p gpurun_out/r6_p
for ring in large small large small; do
  TD_DECODE_RING=$ring timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-r101 --no-fp16-b32 --no-two-model --no-cpu-baseline --no-serial --no-profile --detail gpurun_out/r6_p/detail_$ring.json > gpurun_out/r6_p/$ring.log 2> gpurun_out/r6_p/$ring.err || { tail -5 gpurun_out/r6_p/$ring.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6_p/$ring.log") if l.startswith("{")][-1])
r=d["regions"]
print("$ring", {k:r[k] for k in r if "lzw" in k}, r.get("fp16"))
PY
done
