"""fp16 engine vs the fp32 oracle on full-size tiles: which detections exist on one side only (and why: score near the cut,
NMS neighbours), and the error distribution of the matched ones. python tools/probes/fp16_set_diag.py [depth] [seed] [tile ids ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from oracle import ops_ref as R  # noqa: E402
from oracle.maskrcnn_ref import MaskRCNNOracle  # noqa: E402
from tests.test_engine_fp16_gpu import iou  # noqa: E402
from treedetection_amd.engine import Engine, INPUT_U8_HWC, unpack_outputs  # noqa: E402
from treedetection_amd.synth import make_tile  # noqa: E402
from treedetection_amd.weights import make_synthetic_state_dict  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tids = [int(a) for a in sys.argv[3:]] or [300, 303]
torch.set_num_threads(16)
sd = make_synthetic_state_dict(depth, seed=seed)
oracle = MaskRCNNOracle(sd)
for tid in tids:
    rgb, nd = make_tile(tid, 1000)
    x, h, w = R.preprocess_tile_u8(rgb.transpose(2, 0, 1))
    ref = oracle.forward([{"image": x, "height": h, "width": w}])[0]
    outs = {}
    for prec in ("fp32", "fp16"):
        eng = Engine(sd, precision=prec)
        tile = torch.from_numpy(rgb).cuda()
        batch, hv, ho = eng.preprocess_tiles_u8([tile])
        out = eng.alloc_outputs(1, 1000, 1000, paste=True)
        eng.forward_raw(batch, INPUT_U8_HWC, hv, ho, out)
        torch.cuda.synchronize()
        outs[prec] = unpack_outputs(out, ho, True)[0]
        eng.close()
    for prec in ("fp32", "fp16"):
        g = outs[prec]
        print(f"tile {tid} depth {depth} {prec}: {len(g['scores'])} detections, oracle {len(ref['scores'])}")
        used = set()
        rows = []
        for i in range(len(ref["scores"])):
            v = [iou(ref["pred_boxes"][i], g["pred_boxes"][j]) for j in range(len(g["scores"]))]
            bj = int(np.argmax(v))
            if v[bj] >= 0.9:
                used.add(bj)
                b = ref["pred_boxes"][i]
                rows.append((abs(float(g["scores"][bj]) - float(ref["scores"][i])), float(np.abs(g["pred_boxes"][bj] - b).max()),
                             float(max(b[2] - b[0], b[3] - b[1])), float(np.abs(g["mask_probs"][bj] - ref["mask_probs"][i]).max()), float(ref["scores"][i])))
            else:
                print("  oracle-only", i, round(float(ref["scores"][i]), 4), "best iou", round(v[bj], 3), "score there", round(float(g["scores"][bj]), 4))
        for j in range(len(g["scores"])):
            if j not in used:
                v = [iou(g["pred_boxes"][j], ref["pred_boxes"][i]) for i in range(len(ref["scores"]))]
                bi = int(np.argmax(v))
                print("  engine-only", j, round(float(g["scores"][j]), 4), "best iou with oracle", round(v[bi], 3), "oracle score", round(float(ref["scores"][bi]), 4))
        rows = np.array(rows)
        if len(rows):
            rel = rows[:, 1] / rows[:, 2]
            print(f"  matched {len(rows)}: score err median {np.median(rows[:, 0]):.2e} max {rows[:, 0].max():.2e} | box err px median {np.median(rows[:, 1]):.3f} "
                  f"max {rows[:, 1].max():.3f} | box err / box size median {np.median(rel):.2e} max {rel.max():.2e} | box sizes {rows[:, 2].min():.0f}..{rows[:, 2].max():.0f} | "
                  f"mask prob err median {np.median(rows[:, 3]):.2e} max {rows[:, 3].max():.2e}")
            k = int(np.argmax(rows[:, 1]))
            print(f"  worst box: err {rows[k, 1]:.3f} px on a {rows[k, 2]:.0f}-px box, score {rows[k, 4]:.3f}")
