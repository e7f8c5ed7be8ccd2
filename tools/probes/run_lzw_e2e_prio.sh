# e2e_lzw region of bench.py (and the decode stream at low priority), same box
set -e
mkdir -p gpurun_out/r6_q
run() {
  tag=$1; shift
  env "$@" timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-r101 --no-fp16-b32 --no-two-model --no-cpu-baseline --no-serial --no-profile --detail gpurun_out/r6_q/detail_$tag.json > gpurun_out/r6_q/$tag.log 2> gpurun_out/r6_q/$tag.err || { tail -5 gpurun_out/r6_q/$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6_q/$tag.log") if l.startswith("{")][-1])
r=d["regions"]
print("$tag", {k:r[k] for k in r if "lzw" in k}, r.get("fp16"), r.get("predict_tiles_f16_ratio"))
PY
}
run normal TD_X=1
run low TD_DECODE_PRIORITY=low
