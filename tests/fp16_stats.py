"""Print fp16-vs-oracle error statistics for the fixture of tests/test_engine_fp16_gpu.py (to set honest tolerances)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.maskrcnn_ref import MaskRCNNOracle
from tests.test_engine_gpu import smooth_image
from tests.test_engine_fp16_gpu import iou
from treedetection_amd.engine import Engine
from treedetection_amd.weights import make_synthetic_state_dict
torch.set_num_threads(8)
sd = make_synthetic_state_dict(50, seed=5)
rng = np.random.default_rng(21)
inputs = [{"image": smooth_image(rng, 256, 320), "height": 320, "width": 400},
          {"image": smooth_image(rng, 224, 256), "height": 224, "width": 256}]
ref = MaskRCNNOracle(sd).forward(inputs)
got = Engine(sd, precision="fp16")(inputs)
for g, r in zip(got, ref):
    print("ref", len(r["scores"]), "got", len(g["scores"]))
    rows = []
    for i in range(len(r["scores"])):
        v = [iou(r["pred_boxes"][i], g["pred_boxes"][j]) for j in range(len(g["scores"]))]
        j = int(np.argmax(v))
        a, b = g["pred_masks"][j], r["pred_masks"][i]
        u = (a | b).sum()
        rows.append((v[j], abs(g["scores"][j] - r["scores"][i]), np.abs(g["pred_boxes"][j] - r["pred_boxes"][i]).max(),
                     np.abs(g["mask_probs"][j] - r["mask_probs"][i]).max(), (a & b).sum() / max(u, 1), r["scores"][i]))
    rows = np.array(rows)
    print(" boxIoU min %.4f | score err max %.5f | box err max %.4f px | maskprob err max %.4f | maskIoU min %.4f" %
          (rows[:, 0].min(), rows[:, 1].max(), rows[:, 2].max(), rows[:, 3].max(), rows[:, 4].min()))
    bad = rows[(rows[:, 0] < 0.9) | (rows[:, 1] > 5e-3)]
    print(" unmatched:", len(bad), bad[:, [0, 1, 5]].round(4).tolist())
    print(" maskIoU sorted:", np.sort(rows[:, 4])[:6].round(4), "maskprob err sorted:", np.sort(rows[:, 3])[-6:].round(4))
