"""fp16 engine vs fp32 oracle, stage by stage, on the fixture of tests/test_engine_fp16_gpu.py: where does the error
come from, and how is it distributed? (diagnostic; prints only)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.maskrcnn_ref import MaskRCNNOracle
from tests.test_engine_gpu import smooth_image, nchw
from tests.test_engine_fp16_gpu import iou
from treedetection_amd.engine import Engine
from treedetection_amd.weights import make_synthetic_state_dict
torch.set_num_threads(16)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sd = make_synthetic_state_dict(50, seed=seed)
rng = np.random.default_rng(21)
inputs = [{"image": smooth_image(rng, 256, 320), "height": 320, "width": 400},
          {"image": smooth_image(rng, 224, 256), "height": 224, "width": 256}]
ref, taps = MaskRCNNOracle(sd).forward(inputs, return_taps=True)
e32 = Engine(sd, precision="fp32")
g32 = e32(inputs)
eng = Engine(sd, precision="fp16")
got = eng(inputs)
print("stage: rel RMS err (fp16 vs oracle) | rel max err | mean/std of oracle")
for name, r in [(k, taps["res"][k]) for k in ("stem", "pool", "res2", "res3", "res4", "res5")] + [(k, taps["feats"][k]) for k in ("p2", "p3", "p4", "p5", "p6")]:
    r = r.numpy(); g = nchw(eng.tensor(name).float())
    d = g - r
    print(f"  {name:5s} rms {np.sqrt((d**2).mean())/r.std():.2e}  max {np.abs(d).max()/np.abs(r).max():.2e}  mean {r.mean():+.3f} std {r.std():.3f}")
for li in range(5):
    head = eng.tensor(f"rpn_head{li + 2}").cpu().numpy()
    lg = taps["rpn_logits"][li].numpy().reshape(head.shape[0], head.shape[1], head.shape[2], 3)
    d = head[..., :3] - lg
    print(f"  rpn logits p{li+2}: rms err {np.sqrt((d**2).mean()):.2e} (std {lg.std():.2f})")
bp = eng.tensor("box_pred").cpu().numpy()
print("box_pred shape", bp.shape)
for n, (g, r, f) in enumerate(zip(got, ref, g32)):
    print(f"image {n}: ref {len(r['scores'])} fp16 {len(g['scores'])} fp32 {len(f['scores'])}")
    rows = []
    for i in range(len(r["scores"])):
        v = [iou(r["pred_boxes"][i], g["pred_boxes"][j]) for j in range(len(g["scores"]))]
        j = int(np.argmax(v))
        a, b = g["pred_masks"][j], r["pred_masks"][i]
        u = (a | b).sum()
        pr = r["mask_probs"][i]
        near = (np.abs(pr - 0.5) <= 3e-2).mean()
        flips_ok = (np.abs(pr - 0.5)[(g["mask_probs"][j] >= 0.5) != (pr >= 0.5)] <= 3e-2).all()
        rows.append((v[j], abs(g["scores"][j] - r["scores"][i]), np.abs(g["pred_boxes"][j] - r["pred_boxes"][i]).max(),
                     np.abs(g["mask_probs"][j] - pr).max(), (a & b).sum() / max(u, 1), r["scores"][i], near, float(flips_ok)))
    rows = np.array(rows)
    print("  score err: max %.4f p90 %.4f median %.4f | box err max %.3f | maskprob err max %.4f median %.4f" % (
        rows[:, 1].max(), np.quantile(rows[:, 1], 0.9), np.median(rows[:, 1]), rows[:, 2].max(), rows[:, 3].max(), np.median(rows[:, 3])))
    print("  mask IoU: min %.4f p10 %.4f median %.4f | near-threshold pixel fraction: median %.3f max %.3f | all 28x28 flips within 3e-2 of 0.5: %s" % (
        rows[:, 4].min(), np.quantile(rows[:, 4], 0.1), np.median(rows[:, 4]), np.median(rows[:, 6]), rows[:, 6].max(), bool(rows[:, 7].all())))
    print("  scores (ref) quantiles:", np.quantile(rows[:, 5], [0, .25, .5, .75, 1]).round(3), " score err vs s(1-s):",
          np.round(np.corrcoef(rows[:, 1], rows[:, 5] * (1 - rows[:, 5]))[0, 1], 2))
    worst = rows[np.argsort(-rows[:, 1])[:5]]
    print("  worst score errs (err, score):", worst[:, [1, 5]].round(4).tolist())
