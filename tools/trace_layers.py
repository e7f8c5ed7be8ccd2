"""Per-layer view of a rocprofv3 --kernel-trace CSV of bench.py: pairs the conv_igemm launches of the LAST forward
with the engine's layer schedule (R50/R101, B x Hp x Wp) and prints GFLOP, µs and TFLOP/s per launch."""
import csv
import sys


def schedule(depth=50, B=8, Hp=800, Wp=800, P=1000, fp32=False, fuse_tail=True, fused_heads=(), grouped=False):
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    mids, outs = (64, 128, 256, 512), (256, 512, 1024, 2048)
    L = []
    cin, h, w = 64, Hp // 4, Wp // 4
    for si, nb in enumerate(blocks):
        for bi in range(nb):
            s = 2 if (bi == 0 and si > 0) else 1
            ho, wo = h // s, w // s
            if bi == 0:
                L.append((f"res{si+2}.{bi}.shortcut", B * ho * wo, outs[si], cin))
            L.append((f"res{si+2}.{bi}.conv1", B * ho * wo, mids[si], cin))
            if (fuse_tail and si == 0) or (fuse_tail == 2 and not fp32 and si == 1):
                # bottleneck_tail_kernel: conv2 + conv3 in one launch (N, K of the 3x3; the 1x1's FLOPs ride in the 5th field)
                L.append((f"res{si+2}.{bi}.conv2+3", B * ho * wo, mids[si], mids[si] * 9, 2.0 * B * ho * wo * outs[si] * mids[si] / 1e9))
            else:
                L.append((f"res{si+2}.{bi}.conv2", B * ho * wo, mids[si], mids[si] * 9))
                L.append((f"res{si+2}.{bi}.conv3", B * ho * wo, outs[si], mids[si]))
            cin, h, w = outs[si], ho, wo
    hs = [Hp >> (l + 2) for l in range(4)]
    ws = [Wp >> (l + 2) for l in range(4)]
    for l in (3, 2, 1, 0):
        L.append((f"fpn_lateral{l+2}", B * hs[l] * ws[l], 256, outs[l]))
        if not grouped:
            L.append((f"fpn_output{l+2}", B * hs[l] * ws[l], 256, 2304))
    hs.append((hs[3] - 1) // 2 + 1)
    ws.append((ws[3] - 1) // 2 + 1)
    if grouped:       # conv_pp8_kernel<_Float16, true>: one grid over the levels (fp16 engine)
        L.append(("fpn_output p2-p5 (grouped)", B * sum(hs[l] * ws[l] for l in range(4)), 256, 2304))
        Mr = B * sum(hs[l] * ws[l] for l in range(5))
        L.append(("rpn_conv+head p2-p6 (grouped)", Mr, 256, 2304, 2.0 * Mr * 15 * 256 / 1e9))
    for l in range(5 if not grouped else 0):
        if l in fused_heads:      # the head contracted inside the 3x3 conv's launch (ConvArgs::head_w): its FLOPs ride in the 5th field
            L.append((f"rpn_conv+head p{l+2}", B * hs[l] * ws[l], 256, 2304, 2.0 * B * hs[l] * ws[l] * 15 * 256 / 1e9))
        else:
            L.append((f"rpn_conv p{l+2}", B * hs[l] * ws[l], 256, 2304))
            L.append((f"rpn_head p{l+2}", B * hs[l] * ws[l], 15, 256))
    L.append(("fc1", B * P, 1024, 12544))
    L.append(("fc2", B * P, 1024, 1024))
    L.append(("box_pred", B * P, 6, 1024))
    for i in range(4):
        L.append((f"mask_fcn{i+1} (dyn)", 0, 256, 2304))
    L.append(("mask_deconv (dyn)", 0, 1024, 256))
    return L


def launches_of(name, M, N, K, fp32, B=8, min43=12):
    """fp32 engine (engine.cpp run_conv): 3x3 stride-1 layers with >= 128 channels on both sides take a Winograd path —
    F(4x4,3x3) on maps of at least `min43` pixels a side: three launches (wino43_input_kernel, the batched conv_igemm launch,
    wino43_output_kernel); F(2x2,3x3) otherwise: two (wino_gemm_kernel + wino_output_kernel). → (launch count, label)"""
    three = ("conv2" in name or "fpn_output" in name or "rpn_conv" in name or "mask_fcn" in name)
    if not (fp32 and three and N >= 128 and K // 9 >= 128):
        return 1, None
    side = int(round((M / B) ** 0.5)) if M else 14          # M = 0: the mask head's 14 x 14 RoIs (device-side row count)
    if min43 > 0 and side >= min43:
        # round 4: the fold rule of engine.cpp (TD_WINO_FOLD, default 7): never the RPN layers (their head rides in the three-launch
        # form's output transform); 256 channels from 80 x 80, 128 channels from 80 x 80, the mask head → input transform + wino43_fused_kernel
        import os
        wf = int(os.environ.get("TD_WINO_FOLD", "39"))
        cin = K // 9
        rpn_fold = "rpn_conv p" in name and (wf & 1) and (wf & 32) and cin == 256 and side >= 160      # head as a launch of its own
        fold = rpn_fold or "rpn" not in name and (((wf & 1) and M and ((cin == 256 and side >= 160) or (cin == 128 and side >= 80))) or ((wf & 2) and not M) or
                                      ((wf & 4) and M and cin == 256 and side >= 80) or ((wf & 8) and M and side >= 40) or ((wf & 16) and M))
        if fold and cin in (128, 256, 512) and N % 64 == 0:
            return 2, "winograd F(4x4) folded"
        return 3, "winograd F(4x4)"
    return 2, "winograd F(2x2)"


def main(path, depth=50, fp32=False):
    import os
    min43 = int(os.environ.get("TD_WINO43_MIN", "12"))
    fam = ("conv_igemm", "conv_pp8", "plane_gemm", "wino_gemm", "wino_output", "wino_input", "wino43_input", "wino43_output", "wino43_fused", "bottleneck_tail", "conv_sk", "conv_bd", "conv_bs")
    # launch order = start order on the one stream of the plain loop (the CSV itself is not written in that order)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if any(f in r["Kernel_Name"] for f in fam)]
    fuse_tail = int(os.environ.get("TD_FUSE_TAIL", "1"))
    fused_heads = set()
    grouped = any("conv_pp8_kernel" in r["Kernel_Name"] and ("Lb1" in r["Kernel_Name"] or ", true>" in r["Kernel_Name"]) for r in rows[-40:])
    if not fp32 and not grouped:
        # which RPN levels ran with the head fused is the tuner's tile choice: read it off the trace, walking back from the
        # box head (fc1, fc2, box_pred, 4 mask convs, deconv = 8 launches): a level's last launch is either its head (an
        # fp16-in / fp32-out conv_igemm launch) or the fused 3x3 itself
        i = len(rows) - 8
        is_head = lambda kn: "conv_igemm" in kn and ("DF16_f" in kn or "_Float16, float" in kn)
        for l in (4, 3, 2, 1, 0):
            if is_head(rows[i - 1]["Kernel_Name"]):
                i -= 2
            else:
                fused_heads.add(l)
                i -= 1
    if fp32 and os.environ.get("TD_FUSE_HEAD", "1") != "0":
        # fp32 engine: a fixed rule — every RPN level that takes the F(4x4) path carries its head in the output transform
        hs = [800 >> (l + 2) for l in range(4)]
        hs.append((hs[3] - 1) // 2 + 1)
        wf = int(os.environ.get("TD_WINO_FOLD", "39"))
        fused_heads = {l for l in range(5) if min43 > 0 and hs[l] >= min43 and not ((wf & 1) and (wf & 32) and hs[l] >= 160)}
    L = schedule(depth, fp32=fp32, fuse_tail=fuse_tail, fused_heads=fused_heads, grouped=grouped)
    need = sum(launches_of(e[0], e[1], e[2], e[3], fp32, 8, min43)[0] for e in L)
    last = rows[-need:]
    tot_f = tot_t = 0.0
    i = 0
    for ent in L:
        name, M, N, K = ent[:4]
        k, label = launches_of(name, M, N, K, fp32, 8, min43)
        rs = last[i:i + k]
        i += k
        ts = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs]
        us = sum(ts)
        gf = 2.0 * M * N * K / 1e9 + (ent[4] if len(ent) > 4 else 0.0)
        kn = rs[0]["Kernel_Name"]
        if "conv2+3" in name:
            assert "bottleneck_tail" in kn, (name, kn)
        kern = label if label else ("pp8 grouped" if "grouped" in name else "pp8" if "conv_pp8" in kn else "bottleneck_tail" if "bottleneck_tail" in kn else
                                    "conv_sk" if "conv_sk" in kn else "conv_bs (filter-stationary)" if "conv_bs" in kn else ("conv_bd 64x256" if "Li2ELi2ELi4" in kn or "2, 2, 4" in kn else "conv_bd 128x256" if "Li4ELi2ELi4" in kn or "4, 2, 4" in kn else
                                                 "conv_bd 128x128" if "Li4ELi1ELi4" in kn or "4, 1, 4" in kn else "conv_bd 64x128") if "conv_bd" in kn else
                                    "plane_gemm" + kn.split("plane_gemm_kernel")[1].split("(")[0][:12] if "plane_gemm" in kn else
                                    kn.split("conv_igemm_")[1].split("(")[0][:28])
        if k == 3:
            assert "wino43_input" in kn and "wino43_output" in rs[2]["Kernel_Name"], (name, kn)
            kern += f" [{ts[0]:.0f}+{ts[1]:.0f}+{ts[2]:.0f}]"
        elif k == 2 and label and "folded" in label:
            assert "wino43_input" in kn and "wino43_fused" in rs[1]["Kernel_Name"], (name, kn, rs[1]["Kernel_Name"])
            kern += f" [{ts[0]:.0f}+{ts[1]:.0f}]"
        elif k == 2:
            assert "wino_gemm" in kn, (name, kn)
        print(f"{name:24s} M={M:7d} N={N:5d} K={K:6d} {kern:34s} {us:9.1f} us {gf:8.2f} GF {gf/us*1e3 if us else 0:7.1f} TF/s")
        if M:
            tot_f += gf
            tot_t += us
    print(f"static convs: {tot_f:.1f} algorithmic GF in {tot_t/1e3:.2f} ms = {tot_f/tot_t*1e3:.1f} TF/s")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 50, len(sys.argv) > 3 and sys.argv[3] == "fp32")
