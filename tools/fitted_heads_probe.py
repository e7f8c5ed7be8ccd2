"""How the fitted-head fixture (tests/trained_heads.py) behaves under the fp16 engine for several ridge strengths: strict pairs
(IoU >= 0.9), duplicate-cluster pairs, unpaired detections, worst box / score error — the measurements behind the bounds of
tests/test_engine_fp16_gpu.py::test_fp16_detection_set_on_fitted_heads.   python tools/fitted_heads_probe.py [depth]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.maskrcnn_ref import MaskRCNNOracle
from tests.test_engine_fp16_gpu import match_detection_sets, SCORE_THRESH
from tests.trained_heads import fit_trained_like_heads, tile_inputs
from treedetection_amd.engine import Engine
from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.set_num_threads(16)
tiles = [0, 1]
base = blob_mask_head(make_synthetic_state_dict(depth, seed=5))
inputs = tile_inputs(tiles, 1000)
band = 5e-3 * 4.0 * SCORE_THRESH * (1.0 - SCORE_THRESH) / 0.36
for lam_rpn, lam_box in ((1e-3, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1.0)):
    sd = fit_trained_like_heads(base, tiles, lam_rpn=lam_rpn, lam_box=lam_box)
    ref = MaskRCNNOracle(sd).forward(inputs, paste=False)
    for prec in ("fp32", "fp16"):
        eng = Engine(sd, precision=prec)
        got = eng(inputs, paste=False)
        eng.close()
        for n, (g, r) in enumerate(zip(got, ref)):
            strict, cluster, lost, extra = match_detection_sets(g, r, band)
            eb = max([float(np.abs(g["pred_boxes"][j] - r["pred_boxes"][i]).max()) for i, j, _ in strict] or [0])
            es = max([abs(float(g["scores"][j]) - float(r["scores"][i])) for i, j, _ in strict] or [0])
            print(f"R{depth} lam_rpn {lam_rpn:g} lam_box {lam_box:g} {prec} tile {tiles[n]}: oracle {len(r['scores'])} engine {len(g['scores'])} strict {len(strict)} "
                  f"cluster {[round(v, 2) for _, _, v in cluster]} oracle-only {np.round(lost, 3).tolist()} engine-only {np.round(extra, 3).tolist()} "
                  f"worst strict box {eb:.3f} px score {es:.2e} | |bbox_pred| {np.linalg.norm(sd['roi_heads.box_predictor.bbox_pred.weight']):.1f}", flush=True)
