"""Diagnostic: main-stream timeline of a pipelined bench run from a rocprofv3 --kernel-trace CSV. The main stream is the
one that carries the most conv launches; prints, over the last `steps` trunks, the wall time per step, the stream's
busy time, the idle gaps (count, total, the largest with the kernels either side) and what ran elsewhere meanwhile.

    python tools/timeline.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys


def short(n):
    for k in ("conv_pp8", "conv_igemm", "plane_gemm", "wino_gemm", "wino_output", "stem_conv", "maxpool", "roi_align", "rpn_topk", "nms_scan", "nms_mask",
              "paste_fill", "paste_plan", "mask_predict", "resize_v", "resize_h", "rpn_merge", "sort_boxes", "det_", "mask_scatter", "subsample", "copyBuffer"):
        if k in n:
            return k
    return n[:30]


def main(path, steps=8):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], short(r["Kernel_Name"]), int(r["Grid_Size_X"])))
    rows.sort()
    per = collections.Counter(r[2] for r in rows if r[3] in ("conv_igemm", "conv_pp8", "wino_gemm"))
    main_s = per.most_common(1)[0][0]
    m = [r for r in rows if r[2] == main_s]
    # a step starts at the first conv after a 'big M' marker: use the pp8 / largest-grid conv as the per-step anchor
    anchors = [i for i, r in enumerate(m) if r[3] in ("conv_pp8", "wino_gemm") and r[4] == max(x[4] for x in m if x[3] == r[3])]
    # two anchors per step (fpn_output2, rpn_conv p2): take every second
    anchors = anchors[::2]
    if len(anchors) < steps + 1:
        steps = len(anchors) - 1
    a0, a1 = anchors[-steps - 1], anchors[-1]
    seg = m[a0:a1]
    t0, t1 = seg[0][0], m[a1][0]
    wall = (t1 - t0) / 1e3
    busy = sum(e - s for s, e, *_ in seg) / 1e3
    gaps = []
    for x, y in zip(seg, seg[1:] + [m[a1]]):
        g = (y[0] - x[1]) / 1e3
        gaps.append((g, x[3], y[3]))
    tot_gap = sum(g for g, *_ in gaps if g > 0)
    print(f"main stream {main_s}: {steps} steps, wall {wall / steps:.1f} us/step, busy {busy / steps:.1f} us/step, gaps {tot_gap / steps:.1f} us/step over {len(gaps) / steps:.0f} launches/step")
    small = [g for g, *_ in gaps if 0 < g <= 5]
    print(f"  gaps <= 5 us: {len(small) / steps:.0f} per step, {sum(small) / steps:.1f} us/step (mean {sum(small) / max(len(small), 1):.2f} us)")
    big = sorted([g for g in gaps if g[0] > 5], reverse=True)
    print(f"  gaps > 5 us: {len(big) / steps:.1f} per step, {sum(g for g, *_ in big) / steps:.1f} us/step")
    agg = collections.Counter()
    for g, a, b in big:
        agg[(a, b)] += g
    for (a, b), g in agg.most_common(8):
        print(f"     {a:14s} -> {b:14s} {g / steps:8.1f} us/step")
    # per-kernel-family time on the main stream and elsewhere during the window
    fam = collections.Counter()
    oth = collections.Counter()
    for s, e, st, n, _ in rows:
        if s >= t0 and e <= t1:
            (fam if st == main_s else oth)[n] += (e - s) / 1e3
    print("  main stream per step:", {k: round(v / steps, 1) for k, v in fam.most_common(6)})
    print("  other streams per step:", {k: round(v / steps, 1) for k, v in oth.most_common(12)})


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
