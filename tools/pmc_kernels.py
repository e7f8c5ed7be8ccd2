"""Per-kernel view of rocprofv3 --pmc passes of a plain-loop bench: for every kernel of the LAST forward's conv family, the
counter values of one or more passes (CSV paths on argv) side by side, plus the ratios that say what a kernel waits for."""
import csv
import sys
from collections import defaultdict


def load(path):
    per = defaultdict(lambda: defaultdict(float))      # dispatch id -> counter -> value
    name = {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        name[d] = r["Kernel_Name"]
    return per, name


def short(n):
    for k in ("conv_pp8", "conv_bd", "conv_bs", "conv_sk", "bottleneck_tail", "plane_gemm", "wino_gemm", "wino43_input", "wino43_output", "wino_output"):
        if k in n:
            return k
    if "conv_igemm" in n:
        return "igemm" + n.split("conv_igemm_kernel")[1][:34]
    return n[:40]


def main(paths, last=140):
    merged = defaultdict(dict)
    names = {}
    order = None
    for p in paths:
        per, name = load(p)
        ids = sorted(per)
        fam = [d for d in ids if any(k in name[d] for k in ("conv_", "bottleneck", "plane_gemm", "wino"))][-last:]
        if order is None:
            order = fam
        for i, d in enumerate(fam):           # passes replay the same launch sequence: align by position
            merged[i].update(per[d])
            names[i] = name[d]
    cols = sorted({c for v in merged.values() for c in v})
    print("idx kernel " + " ".join(cols))
    for i in sorted(merged):
        v = merged[i]
        extra = []
        if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            w = v["SQ_WAVE_CYCLES"]
            extra.append("parked %.2f" % (v.get("SQ_WAIT_ANY", 0) / w))
            extra.append("issue_stall %.2f" % (v.get("SQ_WAIT_INST_ANY", 0) / w))
            extra.append("active %.2f" % (v.get("SQ_ACTIVE_INST_ANY", 0) / w))
        if "SQ_BUSY_CYCLES" in v and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            extra.append("mfma_busy/sq_busy %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_BUSY_CYCLES"]))
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            extra.append("l2_hit %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
        print(i, short(names[i]), " ".join("%s=%.4g" % (c, v.get(c, 0)) for c in cols), "|", " ".join(extra))


if __name__ == "__main__":
    main(sys.argv[1:])
