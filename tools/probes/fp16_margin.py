import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle.maskrcnn_ref import MaskRCNNOracle
from tests.test_engine_gpu import smooth_image
from tests.test_engine_fp16_gpu import iou
from treedetection_amd.weights import make_synthetic_state_dict
from treedetection_amd.engine import Engine
torch.set_num_threads(8)
for gc, gm in ((1, 1), (3, 4), (6, 8)):
    sd = make_synthetic_state_dict(50, seed=5)
    sd["roi_heads.box_predictor.cls_score.weight"] = sd["roi_heads.box_predictor.cls_score.weight"] * np.float32(gc)
    sd["roi_heads.mask_head.predictor.weight"] = sd["roi_heads.mask_head.predictor.weight"] * np.float32(gm)
    rng = np.random.default_rng(21)
    inputs = [{"image": smooth_image(rng, 256, 320), "height": 320, "width": 400},
              {"image": smooth_image(rng, 224, 256), "height": 224, "width": 256}]
    ref = MaskRCNNOracle(sd).forward(inputs)
    eng = Engine(sd, precision="fp16")
    got = eng(inputs)
    es, ious, nears, areas, diffs, perims = [], [], [], [], [], []
    for g, r in zip(got, ref):
        for i in range(len(r["scores"])):
            v = [iou(r["pred_boxes"][i], g["pred_boxes"][j]) for j in range(len(g["scores"]))]
            if not v: continue
            bj = int(np.argmax(v))
            if v[bj] < 0.9: continue
            es.append(abs(float(g["scores"][bj]) - float(r["scores"][i])))
            a, b = g["pred_masks"][bj], r["pred_masks"][i]
            u = (a | b).sum()
            ious.append((a & b).sum() / u if u else 1.0)
            areas.append(int(b.sum())); diffs.append(int((a ^ b).sum())); perims.append(int((b ^ np.roll(b, 1, 0)).sum() + (b ^ np.roll(b, 1, 1)).sum()))
            nears.append(float((np.abs(r["mask_probs"][i] - 0.5) <= 3e-2).mean()))
    es, ious = np.array(es), np.array(ious)
    print(f"gain cls x{gc} mask x{gm}: dets ref {[len(r['scores']) for r in ref]} got {[len(g['scores']) for g in got]} matched {len(es)}: "
          f"score err max {es.max():.4f} p90 {np.quantile(es,0.9):.4f}; mask IoU min {ious.min():.3f} median {np.median(ious):.3f}; near-cut frac median {np.median(nears):.3f}", flush=True)
    order = np.argsort(ious)[:8]
    print("   lowest IoU:", [(round(float(ious[i]), 3), areas[i], diffs[i], perims[i]) for i in order], "  (IoU, oracle area px, differing px, boundary px)")
    frac = np.array(diffs) / np.maximum(np.array(perims), 1)
    print(f"   differing pixels / boundary pixels: max {frac.max():.3f} median {np.median(frac):.3f}; IoU < 0.97: {int((ious < 0.97).sum())} of {len(ious)}, their max area {max([areas[i] for i in range(len(ious)) if ious[i] < 0.97] or [0])}")
    eng.close()
