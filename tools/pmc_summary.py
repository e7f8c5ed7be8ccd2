"""Summarise three rocprofv3 --pmc passes of bench.py (SQ/GRBM, FETCH_SIZE, WRITE_SIZE — separate runs, as
MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes) for the conv_igemm launches of the LAST forward:
HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (gfx950: FETCH_SIZE reads exactly half of a wide coalesced
stream's bytes, the guide's correction), MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).

    python tools/pmc_summary.py <sq_csv> <fetch_csv> <write_csv> [depth] [bytes per element: 4 | 2] > profiles/rNN_pmc_conv_igemm.json
"""
import collections
import csv
import json
import sys

sys.path.insert(0, "tools")
from trace_layers import schedule  # noqa: E402


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"],
                                                 "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(d.values())


def main():
    sq, fs, ws = (load(p) for p in sys.argv[1:4])
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    es = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
    L = schedule(depth)
    n = len(L)
    pick = lambda rows: [r for r in rows if "conv_igemm" in r["name"]][-n:]
    tot = collections.Counter()
    layers = []
    for (name, M, N, K), a, f, w in zip(L, pick(sq), pick(fs), pick(ws)):
        if not M:
            continue   # mask-head launches: row count lives on the device
        taps = 9 if (K % 9 == 0 and not name.startswith("fc")) else 1
        alg = es * (M * K / taps + M * N + N * K + (M * N if name.endswith("conv3") or "lateral" in name and "5" not in name else 0))
        gui = a["GRBM_GUI_ACTIVE"] / 8.0
        rec = {"layer": name, "us": a["t"], "gflop": 2.0 * M * N * K / 1e9,
               "mfma_util": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0),
               "hbm_mb": (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / 1e6, "algorithmic_mb": alg / 1e6}
        layers.append(rec)
        tot["us"] += rec["us"]
        tot["gflop"] += rec["gflop"]
        tot["hbm_mb"] += rec["hbm_mb"]
        tot["alg_mb"] += rec["algorithmic_mb"]
        tot["mfma_busy"] += a["SQ_VALU_MFMA_BUSY_CYCLES"]
        tot["simd_cycles"] += gui * 1024.0
    out = {"what": "conv_igemm_%s launches of one forward (B=8, 800x800, R%d), static-M launches only" % ("f32" if es == 4 else "f16", depth),
           "launches": len(layers), "total_us": tot["us"], "gflop": tot["gflop"],
           "tflops_profiled": tot["gflop"] / tot["us"] * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else tot["gflop"] / tot["us"] / 1e3 * 1e3,
           "hbm_traffic_gb_per_step": tot["hbm_mb"] / 1e3, "algorithmic_gb_per_step": tot["alg_mb"] / 1e3,
           "mfma_util": tot["mfma_busy"] / tot["simd_cycles"], "layers": layers}
    out["tflops_profiled"] = tot["gflop"] / (tot["us"] * 1e-6) / 1e3
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
