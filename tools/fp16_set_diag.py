import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from oracle import ops_ref as R
from oracle.maskrcnn_ref import MaskRCNNOracle
from treedetection_amd.engine import Engine, INPUT_U8_HWC, unpack_outputs
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict
from tests.test_engine_fp16_gpu import iou
torch.set_num_threads(16)
sd = make_synthetic_state_dict(50, seed=2)
# tile of column 0 row 0 in the config2 fixture: make_tile(300 + r*3 + c)
for tid in (300, 303):
    rgb, nd = make_tile(tid, 1000)
    x, h, w = R.preprocess_tile_u8(rgb.transpose(2, 0, 1))
    ref = MaskRCNNOracle(sd).forward([{"image": x, "height": h, "width": w}])[0]
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
        print(tid, prec, "n", len(g["scores"]), "ref", len(ref["scores"]))
        used = set()
        for i in range(len(ref["scores"])):
            v = [iou(ref["pred_boxes"][i], g["pred_boxes"][j]) for j in range(len(g["scores"]))]
            bj = int(np.argmax(v))
            if v[bj] >= 0.9: used.add(bj)
            else: print("  ref-only", i, float(ref["scores"][i]), "best iou", v[bj], "score there", float(g["scores"][bj]))
        for j in range(len(g["scores"])):
            if j not in used:
                v = [iou(g["pred_boxes"][j], ref["pred_boxes"][i]) for i in range(len(ref["scores"]))]
                bi = int(np.argmax(v))
                # overlap with higher-scoring kept detections of g (NMS neighbours)
                nb = [(round(iou(g["pred_boxes"][j], g["pred_boxes"][k]), 3), round(float(g["scores"][k]), 3)) for k in range(len(g["scores"])) if k != j and iou(g["pred_boxes"][j], g["pred_boxes"][k]) > 0.3]
                print("  eng-only", j, float(g["scores"][j]), "best iou with ref", round(v[bi], 3), "ref score", float(ref["scores"][bi]), "neighbours", nb, "box", g["pred_boxes"][j])
