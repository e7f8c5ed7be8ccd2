"""Generates tests/golden/oracle_small.npz: the oracle's outputs on a small seeded fixture, frozen.

    python tests/golden/make_oracle_fixture.py

The reference holds no golden vectors for the forward and its arithmetic lives in detectron2 /
torchvision / cv2, none of which is installed here, so these vectors are NOT reference outputs: they freeze the
oracle's own restatement (oracle/*.py) at the state the round-2 GPU parity run was green against, so that a later
edit of the oracle cannot silently move the goalposts of every GPU test (tests/test_golden_oracle.py re-runs the
oracle against this file on the CPU; tests/test_golden_gpu.py compares the HIP engine with this file directly).

Contents (all arrays little-endian, float32 unless noted):
  forward of MaskRCNNOracle on 2 images (half-width R50-FPN, synthetic weights seed 3):
    img{n}_boxes [N,4], img{n}_scores [N], img{n}_mask_probs [<=24,28,28], img{n}_mask_bits (packbits of [N,h,w]),
    img{n}_proposals [<=1000,4] (first 64 kept in full + count), img{n}_topk_idx{l} int64 (per level, first 64),
    stage statistics tap_{name} = [mean, mean|x|, max|x|] of stem..res5, p2..p6 (float64)
  op-level known inputs → outputs:
    nms_* (random boxes incl. exact ties), roi_* (random feature map), paste_* (random 28x28 probabilities),
    resize_* (Pillow-exact 8-bit bilinear), contour_* (border following on a seeded mask)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def smooth_image(rng, h, w):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    img = np.zeros((3, h, w), dtype=np.float64)
    for c in range(3):
        for _ in range(6):
            fx, fy, ph = rng.uniform(0.005, 0.08), rng.uniform(0.005, 0.08), rng.uniform(0, 6.28)
            img[c] += 25 * np.cos(fx * xx + fy * yy + ph)
    for _ in range(25):
        cy, cx, s = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(5, 30)
        blob = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
        img += blob[None] * rng.uniform(-90, 90, (3, 1, 1))
    img += 115 + rng.normal(0, 6, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.float32)


def fixture_inputs():
    rng = np.random.default_rng(11)
    return [{"image": smooth_image(rng, 256, 320), "height": 200, "width": 250},
            {"image": smooth_image(rng, 224, 288), "height": 300, "width": 390}]


def op_inputs():
    """Seeded inputs of the op-level vectors (shared with the tests)."""
    rng = np.random.default_rng(2024)
    n = 300
    xy = rng.uniform(0, 200, (n, 2))
    wh = rng.uniform(4, 80, (n, 2))
    boxes = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    boxes[50:60] = boxes[40:50]                       # exact duplicates
    scores = rng.uniform(0, 1, n).astype(np.float32)
    scores[100:120] = scores[100]                     # exact score ties: stable order decides
    feat = rng.normal(0, 1, (8, 40, 52)).astype(np.float32)
    rois = np.array([[3.2, 4.1, 60.7, 33.3], [0, 0, 207.9, 159.9], [100.5, 80.25, 101.0, 81.0], [-5, -7, 30, 20],
                     [150, 100, 260, 200], [17.0, 9.0, 17.0, 9.0]], dtype=np.float32)
    probs = rng.uniform(0, 1, (5, 28, 28)).astype(np.float32)
    pboxes = np.array([[10.3, 12.8, 90.2, 70.1], [-4.0, -3.5, 40.0, 30.0], [100.0, 50.0, 159.6, 119.7],
                       [30.5, 30.5, 31.4, 31.2], [0, 0, 160, 120]], dtype=np.float32)
    tile = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)          # [H,W,C]
    yy, xx = np.meshgrid(np.arange(64), np.arange(80), indexing="ij")
    mask = (((yy - 30) ** 2 + (xx - 40) ** 2 < 400) & ~((yy - 28) ** 2 + (xx - 44) ** 2 < 60)) | (rng.uniform(0, 1, (64, 80)) > 0.93)
    return dict(boxes=boxes, scores=scores, feat=feat, rois=rois, probs=probs, pboxes=pboxes, tile=tile, mask=mask.astype(np.uint8))


def compute():
    import torch
    from oracle import ops_ref as R
    from oracle.contours_ref import find_contours
    from oracle.maskrcnn_ref import MaskRCNNOracle
    from treedetection_amd.weights import make_synthetic_state_dict

    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = {}
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    ref, taps = MaskRCNNOracle(sd).forward(fixture_inputs(), return_taps=True)
    for n, r in enumerate(ref):
        out[f"img{n}_boxes"] = r["pred_boxes"].astype(np.float32)
        out[f"img{n}_scores"] = r["scores"].astype(np.float32)
        out[f"img{n}_mask_probs"] = r["mask_probs"][:24].astype(np.float32)     # the 24 best detections
        out[f"img{n}_mask_shape"] = np.array(r["pred_masks"].shape, dtype=np.int64)
        out[f"img{n}_mask_bits"] = np.packbits(r["pred_masks"].astype(np.uint8).reshape(-1))
        props = taps["proposals"][n][0]
        out[f"img{n}_proposal_count"] = np.array([props.shape[0]], dtype=np.int64)
        out[f"img{n}_proposals"] = props[:64].astype(np.float32)
        for li, pl in enumerate(taps["rpn_taps"][n]["per_level"]):
            out[f"img{n}_topk_idx{li}"] = pl["topk_idx"][:64].astype(np.int64)
        out[f"img{n}_rpn_keep"] = np.asarray(taps["rpn_taps"][n]["keep"][:64], dtype=np.int64)
        out[f"img{n}_det_keep"] = np.asarray(taps["det_taps"][n]["keep"], dtype=np.int64)
    for group, names in (("res", ("stem", "pool", "res2", "res3", "res4", "res5")), ("feats", ("p2", "p3", "p4", "p5", "p6"))):
        for name in names:
            t = taps[group][name].double()
            out[f"tap_{name}"] = np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item()])
    op = op_inputs()
    out["nms_keep_05"] = R.nms(op["boxes"], op["scores"], 0.5).astype(np.int64)
    out["nms_keep_07"] = R.nms(op["boxes"], op["scores"], 0.7).astype(np.int64)
    out["nms_batched_07"] = R.batched_nms(op["boxes"], op["scores"], (np.arange(300) % 3).astype(np.int64), 0.7).astype(np.int64)
    out["roi_7"] = R.roi_align(op["feat"], op["rois"], 0.25, 7).astype(np.float32)
    out["roi_14"] = R.roi_align(op["feat"], op["rois"], 0.25, 14).astype(np.float32)
    out["roi_levels"] = R.level_assign(np.array([[0, 0, 10, 10], [0, 0, 111, 112], [0, 0, 112, 112], [0, 0, 224, 224],
                                                 [0, 0, 447, 448], [0, 0, 448, 448], [0, 0, 2000, 2000]], np.float32)).astype(np.int64)
    out["paste_bits"] = np.packbits(R.paste_masks(op["probs"], op["pboxes"], 120, 160, 0.5).astype(np.uint8).reshape(-1))
    out["resize_80x108"] = R.pil_resize_bilinear_u8(op["tile"], 80, 108)
    out["resize_160x216"] = R.pil_resize_bilinear_u8(op["tile"], 160, 216)
    out["resize_shape"] = np.array([R.resize_shortest_edge_shape(h, w) for h, w in ((1000, 1000), (450, 450), (350, 450), (300, 900), (97, 131))], np.int64)
    cs = find_contours(op["mask"])
    out["contour_count"] = np.array([len(cs)], np.int64)
    out["contour_sizes"] = np.array([c.shape[0] for c in cs], np.int64)
    out["contour_points"] = np.concatenate([c.reshape(-1, 2) for c in cs]).astype(np.int32)
    anchors = R.grid_anchors(3, 4, 16, 128.0)
    out["anchors_3x4_s16"] = anchors.astype(np.float32)
    dl = np.random.default_rng(5).normal(0, 1.5, (anchors.shape[0], 4)).astype(np.float32)
    dl[0, 2] = 9.0                                     # beyond the scale clamp log(1000/16)
    out["decode_rpn"] = R.apply_deltas(dl, anchors, (1.0, 1.0, 1.0, 1.0)).astype(np.float32)
    out["decode_box"] = R.apply_deltas(dl, anchors, (10.0, 10.0, 5.0, 5.0)).astype(np.float32)
    return out


if __name__ == "__main__":
    vec = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz")
    np.savez_compressed(path, **vec)
    print(f"wrote {path}: {len(vec)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
