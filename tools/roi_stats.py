"""Diagnostic: what RoIAlign really has to do on the bench workload. Runs one batch of the synthetic tile stream through
the engine, reads the proposals / detections and prints, per FPN level, the RoI count and the adaptive sampling grid
(gh x gw samples per bin, detectron2 sampling_ratio = 0) — the number of bilinear samples and 4-corner loads the box
(7x7) and mask (14x14) RoIAlign launches issue. Saves the boxes to gpurun_out/roi_stats.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from treedetection_amd.engine import Engine, INPUT_U8_HWC  # noqa: E402
from treedetection_amd.synth import make_stream  # noqa: E402
from treedetection_amd.weights import make_synthetic_state_dict  # noqa: E402


def level_of(b):
    s = np.sqrt(np.maximum((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), 0))
    lv = np.floor(4 + np.log2(s / 224 + 1e-8))
    return np.clip(lv, 2, 5).astype(int) - 2


def grids(b, lv, pooled):
    sc = 1.0 / (4 << lv)
    rw = (b[:, 2] - b[:, 0]) * sc
    rh = (b[:, 3] - b[:, 1]) * sc
    return np.maximum(np.ceil(rh / pooled), 0).astype(int), np.maximum(np.ceil(rw / pooled), 0).astype(int)


def report(name, b, pooled):
    lv = level_of(b)
    gh, gw = grids(b, lv, pooled)
    smp = gh * gw
    print(f"{name}: {len(b)} RoIs, pooled {pooled}: samples per bin mean {smp.mean():.2f} median {np.median(smp):.0f} p90 "
          f"{np.quantile(smp, 0.9):.0f} max {smp.max()}; total samples {(smp * pooled * pooled).sum() / 1e6:.2f} M "
          f"(x4 corner loads); region pixels if each were read once {((gh * pooled + 1) * (gw * pooled + 1)).sum() / 1e6:.2f} M")
    for l in range(4):
        m = lv == l
        if m.any():
            print(f"   p{l + 2}: {m.sum():5d} RoIs  gh mean {gh[m].mean():.2f} max {gh[m].max()}  gw mean {gw[m].mean():.2f} max {gw[m].max()}  "
                  f"samples/bin mean {smp[m].mean():.2f} max {smp[m].max()}")


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    B = 8
    sd = make_synthetic_state_dict(50, seed=0)
    rgb_np, _ = make_stream(B, 1000)
    rgb = torch.from_numpy(rgb_np).cuda()
    eng = Engine(sd, precision=prec)
    out = eng.alloc_outputs(B, 1000, 1000, paste=True)
    batch, hw_valid, hw_out = eng.preprocess_tiles_u8([rgb[j] for j in range(B)])
    eng.forward_raw(batch, INPUT_U8_HWC, hw_valid, hw_out, out)
    torch.cuda.synchronize()
    props = eng.tensor("proposals").cpu().numpy()
    pc = eng.tensor("proposal_count").cpu().numpy()
    dets = eng.tensor("det_boxes_net").cpu().numpy()
    dc = out["count"].cpu().numpy()
    print("proposals per image", pc.tolist(), "detections per image", dc.tolist())
    pb = np.concatenate([props[i, :pc[i]] for i in range(B)])
    db = np.concatenate([dets[i, :dc[i]] for i in range(B)])
    report("box head", pb, 7)
    report("mask head", db, 14)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "roi_stats.npz"), props=props, pc=pc, dets=dets, dc=dc)


if __name__ == "__main__":
    main()
