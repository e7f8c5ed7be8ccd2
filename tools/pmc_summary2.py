"""Summarise three rocprofv3 --pmc passes of a plain-loop bench.py run (SQ/GRBM, FETCH_SIZE, WRITE_SIZE — separate runs, as
MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes) per KERNEL FAMILY over the last `steps` forwards:
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reads exactly half of a wide coalesced stream's
bytes — the guide's correction; WRITE_SIZE is exact for 16-B stores), MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).

    python tools/pmc_summary2.py <sq_csv> <fetch_csv> <write_csv> <steps> <algorithmic GB per step> > profiles/rNN_pmc_*.json
"""
import collections
import csv
import json
import re
import sys


def family(name):
    if "conv_pp8" in name:
        return "conv_pp8_kernel"
    if "conv_bd" in name:
        return "conv_bd_kernel"
    if "conv_bs" in name:
        return "conv_bs_kernel"
    if "conv_sk" in name:
        return "conv_sk_kernel"
    if "bottleneck_tail" in name:
        return "bottleneck_tail_kernel"
    if "conv_igemm" in name:
        return "conv_igemm_kernel"
    if "plane_gemm" in name:
        return "plane_gemm_kernel"
    if "wino43_fused" in name:
        return "wino43_fused_kernel"
    for k in ("wino_gemm", "wino43_input", "wino43_output", "wino_input", "wino_output", "stem_conv", "stem_mfma", "maxpool", "roi_align", "rpn_topk",
              "rpn_keys", "nms_", "paste", "resize_"):
        if k in name:
            return k.rstrip("_") + "_kernel" if not k.endswith("kernel") else k
    return "other"


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"],
                                                 "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())


def last_steps(rows, steps):
    """Dispatches of the last `steps` forwards: a forward starts at its stem_conv launch."""
    starts = [i for i, r in enumerate(rows) if "stem_conv" in r["name"] or "stem_mfma" in r["name"]]
    return rows[starts[-steps]:] if len(starts) >= steps else rows


def main():
    sq, fs, ws = (load(p) for p in sys.argv[1:4])
    steps = int(sys.argv[4])
    alg_gb = float(sys.argv[5]) if len(sys.argv) > 5 else None
    sq, fs, ws = last_steps(sq, steps), last_steps(fs, steps), last_steps(ws, steps)
    fam = collections.OrderedDict()

    def acc(rows, keys):
        for r in rows:
            f = fam.setdefault(family(r["name"]), collections.Counter())
            for k in keys:
                f[k] += r.get(k, 0.0)

    for r in sq:
        f = fam.setdefault(family(r["name"]), collections.Counter())
        f["launches"] += 1
        f["us"] += r["t"]
    acc(sq, ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    acc(fs, ("FETCH_SIZE",))
    acc(ws, ("WRITE_SIZE",))
    out = {"what": f"last {steps} forwards of a plain-loop bench.py run, per kernel family", "steps": steps, "families": {}}
    conv = collections.Counter()
    for name, f in fam.items():
        hbm = (2.0 * f["FETCH_SIZE"] + f["WRITE_SIZE"]) * 1024
        simd = f["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        rec = {"launches_per_step": f["launches"] / steps, "ms_per_step": f["us"] / steps / 1e3,
               "avg_launch_us": f["us"] / max(f["launches"], 1), "hbm_gb_per_step": hbm / steps / 1e9,
               "hbm_bytes_per_launch": hbm / max(f["launches"], 1),
               "mfma_util": f["SQ_VALU_MFMA_BUSY_CYCLES"] / simd if simd else 0.0,
               "hbm_tb_per_s": hbm / (f["us"] * 1e-6) / 1e12 if f["us"] else 0.0}
        out["families"][name] = rec
        if name in ("conv_igemm_kernel", "conv_pp8_kernel", "conv_bd_kernel", "conv_bs_kernel", "conv_sk_kernel", "bottleneck_tail_kernel", "plane_gemm_kernel", "wino_gemm_kernel", "wino_input_kernel", "wino_output_kernel",
                    "wino43_input_kernel", "wino43_output_kernel", "wino43_fused_kernel"):
            for k in ("launches", "us", "SQ_VALU_MFMA_BUSY_CYCLES"):
                conv[k] += f[k]
            conv["hbm"] += hbm
            conv["simd"] += simd
    out["conv_family"] = {"members": "conv_igemm_kernel + conv_pp8_kernel + conv_bd_kernel + conv_bs_kernel + conv_sk_kernel + bottleneck_tail_kernel + plane_gemm_kernel + "
                                     "wino_gemm_kernel + wino_input_kernel + wino_output_kernel + wino43_input_kernel + wino43_output_kernel + wino43_fused_kernel",
                          "launches": conv["launches"] / steps, "ms_per_step": conv["us"] / steps / 1e3,
                          "hbm_traffic_gb_per_step": conv["hbm"] / steps / 1e9,
                          "hbm_bytes_per_launch": conv["hbm"] / max(conv["launches"], 1),
                          "mfma_util": conv["SQ_VALU_MFMA_BUSY_CYCLES"] / conv["simd"] if conv["simd"] else 0.0}
    if alg_gb:
        out["conv_family"]["algorithmic_gb_per_step"] = alg_gb
        out["conv_family"]["traffic_over_algorithmic"] = out["conv_family"]["hbm_traffic_gb_per_step"] / alg_gb
    # bench.py reads these two
    out["hbm_traffic_gb_per_step"] = out["conv_family"]["hbm_traffic_gb_per_step"]
    out["launches"] = out["conv_family"]["launches"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
