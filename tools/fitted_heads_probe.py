"""How the fitted / trained head fixtures (tests/trained_heads.py) behave under the fp16 engine: strict pairs (IoU >= 0.9),
duplicate-cluster pairs, unpaired detections, worst box / score error.
    python tools/fitted_heads_probe.py [depth]            # ridge-fitted output layers, four ridge strengths
    python tools/fitted_heads_probe.py [depth] train [weight seed] [tiles a,b] [steps s1,s2]     # + box head trained by gradient descent"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.maskrcnn_ref import MaskRCNNOracle
from tests.test_engine_fp16_gpu import match_detection_sets, SCORE_THRESH
from tests.trained_heads import fit_trained_like_heads, tile_inputs, train_box_head
from treedetection_amd.engine import Engine
from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
train = len(sys.argv) > 2 and sys.argv[2] == "train"
wseed = int(sys.argv[3]) if len(sys.argv) > 3 else 5                       # weight seed of the synthetic model
tiles = [int(t) for t in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1]
step_list = [int(t) for t in sys.argv[5].split(",")] if len(sys.argv) > 5 else [3000]
torch.set_num_threads(16)
base = blob_mask_head(make_synthetic_state_dict(depth, seed=wseed))
inputs = tile_inputs(tiles, 1000)
band = 5e-3 * 4.0 * SCORE_THRESH * (1.0 - SCORE_THRESH) / 0.36


def report(tag, sd):
    ref = MaskRCNNOracle(sd).forward(inputs, paste=False)
    for prec in ("fp32", "fp16"):
        eng = Engine(sd, precision=prec)
        got = eng(inputs, paste=False)
        eng.close()
        for n, (g, r) in enumerate(zip(got, ref)):
            strict, cluster, lost, extra = match_detection_sets(g, r, band)
            ebs = sorted([(float(np.abs(g["pred_boxes"][j] - r["pred_boxes"][i]).max()), float(r["pred_boxes"][i][2] - r["pred_boxes"][i][0]), round(v, 3))
                          for i, j, v in strict], reverse=True)
            eb = ebs[0][0] if ebs else 0.0
            if prec == "fp16":
                print("   largest box errors (px, box width, IoU):", [(round(a, 2), round(b), c) for a, b, c in ebs[:4]],
                      " pairs above 0.5 px:", sum(1 for a, _, _ in ebs if a > 0.5), "of", len(ebs))
            es = max([abs(float(g["scores"][j]) - float(r["scores"][i])) for i, j, _ in strict] or [0])
            print(f"R{depth} {tag} {prec} tile {tiles[n]}: oracle {len(r['scores'])} engine {len(g['scores'])} strict {len(strict)} "
                  f"cluster {[round(v, 2) for _, _, v in cluster]} oracle-only {np.round(lost, 3).tolist()} engine-only {np.round(extra, 3).tolist()} "
                  f"worst strict box {eb:.3f} px score {es:.2e} | |bbox_pred| {np.linalg.norm(sd['roi_heads.box_predictor.bbox_pred.weight']):.1f}", flush=True)


if train:
    import time
    rpn = fit_trained_like_heads(base, tiles)
    for steps, jitter in [(st, 48) for st in step_list]:
        t0 = time.time()
        sd = train_box_head(rpn, tiles, steps=steps, jitter_per_crown=jitter, verbose=True, predictor_init=base)
        print(f"trained in {time.time() - t0:.1f} s")
        report(f"trained steps {steps} jitter {jitter}", sd)
else:
    for lam_rpn, lam_box in ((1e-3, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1.0)):
        report(f"lam_rpn {lam_rpn:g} lam_box {lam_box:g}", fit_trained_like_heads(base, tiles, lam_rpn=lam_rpn, lam_box=lam_box))
