"""Generator of tests/golden/trained_heads_<name>.npz — run ONCE in the build container, never on the GPU box.

    python tests/golden/make_trained_heads.py [name ...] [--rank 64] [--steps 3000] [--check]

For every fixture of tests/trained_heads.FIXTURES (depth, weight seed, generator tiles):
  1. the seeded synthetic state dict with the compact-blob mask head (treedetection_amd.weights);
  2. the RPN's two output layers and the box predictor fitted by ridge regression in float64 on the ORACLE's fp32 features of
     the tiles (tests/trained_heads.fit_trained_like_heads: closed form, deterministic);
  3. the box head trained by gradient descent ON THE CPU (torch autograd, float32, one thread pool, fixed seed,
     torch.use_deterministic_algorithms): fc1 stays FROZEN at its seeded value (so its 12 544 x 1 024 matrix is not part of
     the fixture and its activations are computed once), fc2 = seeded + U·V with rank ``--rank`` factors, fc2's bias,
     cls_score and bbox_pred in full; full-batch Adam with a cosine schedule on the fitted RPN's proposals plus jittered
     boxes around every crown, class labels by IoU with the crown (>= 0.5 foreground, < 0.4 background), smooth-L1 on the
     deltas onto the crown;
  4. the tensors that differ from the seeded state dict → ``trained_heads_<name>.npz`` (< 1 MB) and its SHA-256 →
     ``trained_heads.sha256``.
``--check`` prints, per fixture, what the ORACLE detects with the trained heads and how its detection set moves when every
weight is rounded to fp16 (a CPU proxy of the fp16 engine's noise; the real comparison is tests/test_engine_fp16_gpu.py).

Nothing of the reference is involved (there are no reference weights: SURVEY.md §8c); the oracle and the tile generator are
test infrastructure. The product has no backward pass.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.maskrcnn_ref import MaskRCNNOracle                                              # noqa: E402
from tests import trained_heads as TH                                                        # noqa: E402
from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict              # noqa: E402

FC1, FC2 = "roi_heads.box_head.fc1", "roi_heads.box_head.fc2"
CLS, BOX = "roi_heads.box_predictor.cls_score", "roi_heads.box_predictor.bbox_pred"
RPN = "proposal_generator.rpn_head."


def jittered_boxes(gt: np.ndarray, per_crown: int, rng, hw) -> np.ndarray:
    out = []
    for b in gt:
        w, h = b[2] - b[0], b[3] - b[1]
        for _ in range(per_crown):
            s = np.exp(rng.uniform(-0.45, 0.45, 2))
            dx, dy = rng.uniform(-0.3, 0.3, 2) * (w, h)
            cx, cy = (b[0] + b[2]) / 2 + dx, (b[1] + b[3]) / 2 + dy
            out.append([cx - s[0] * w / 2, cy - s[1] * h / 2, cx + s[0] * w / 2, cy + s[1] * h / 2])
    o = np.asarray(out, dtype=np.float64).reshape(-1, 4)
    o[:, 0::2] = np.clip(o[:, 0::2], 0, hw[1])
    o[:, 1::2] = np.clip(o[:, 1::2], 0, hw[0])
    return o[(o[:, 2] - o[:, 0] > 2) & (o[:, 3] - o[:, 1] > 2)]


def training_set(sd, tiles, size, jitter_per_crown, seed, verbose):
    """fc1's activations (fc1 is frozen: computed once), class labels, box targets and foreground weights of the training boxes."""
    inputs = TH.tile_inputs(tiles, size)
    oracle = MaskRCNNOracle(sd)
    rng = np.random.default_rng(seed)
    w1, b1 = torch.from_numpy(sd[FC1 + ".weight"]), torch.from_numpy(sd[FC1 + ".bias"])
    H, Ycls, Ybox, W = [], [], [], []
    with torch.no_grad():
        for k, inp in enumerate(inputs):
            x, sizes = oracle.batch_images([inp["image"]])
            feats = oracle.fpn(oracle.backbone(x))
            gt = TH.crown_boxes(tiles[k], size, sizes[0])
            logits, deltas = oracle.rpn_head(feats)
            feat_hw = [tuple(feats[f"p{l}"].shape[-2:]) for l in (2, 3, 4, 5, 6)]
            props, _ = oracle.rpn_proposals(logits, deltas, feat_hw, sizes)
            boxes = np.concatenate([props[0][0].astype(np.float64), jittered_boxes(gt, jitter_per_crown, rng, sizes[0])])
            pooled, _ = oracle.roi_pool(feats, [boxes.astype(np.float32)], 7)
            iou = TH.iou_matrix(boxes, gt)
            best, arg = iou.max(axis=1), iou.argmax(axis=1)
            fg, bg = best >= 0.5, best < 0.4
            keep = fg | bg
            y = np.zeros((len(boxes), 4))
            y[fg] = TH.box_deltas(boxes[fg], gt[arg[fg]], (10.0, 10.0, 5.0, 5.0))
            X = torch.from_numpy(np.ascontiguousarray(pooled[0][keep].reshape(int(keep.sum()), -1)))
            H.append(torch.relu(torch.nn.functional.linear(X, w1, b1)))
            Ycls.append(np.where(fg[keep], 0, 1))            # class 0 = the one foreground class, last = background
            Ybox.append(y[keep])
            W.append(fg[keep].astype(np.float32))
            if verbose:
                print(f"  tile {tiles[k]}: {len(gt)} crowns, {len(boxes)} boxes, {int(fg.sum())} on a crown, {int(bg.sum())} background", flush=True)
    return (torch.cat(H), torch.from_numpy(np.concatenate(Ycls)).long(), torch.from_numpy(np.concatenate(Ybox)).float(),
            torch.from_numpy(np.concatenate(W)))


def train_box_head_cpu(sd, base, tiles, size=1000, steps=3000, lr=2e-3, rank=64, jitter_per_crown=48, seed=0, verbose=True):
    """→ the fixture's tensors {state-dict key or low-rank factor: float32 array}. ``sd`` carries the fitted RPN output layers;
    the predictors start from the SEEDED ones of ``base`` (the ridge-fitted predictor has 20 x their norm and keeps amplifying
    fp16 feature noise through the whole training: round 5)."""
    h1, yc, yb, wf = training_set(sd, tiles, size, jitter_per_crown, seed, verbose)
    torch.manual_seed(seed)
    w2 = torch.from_numpy(base[FC2 + ".weight"])
    U = torch.zeros(w2.shape[0], rank, requires_grad=True)
    V = (torch.randn(rank, w2.shape[1]) / np.sqrt(w2.shape[1])).requires_grad_(True)
    P = {FC2 + ".bias": torch.tensor(base[FC2 + ".bias"], requires_grad=True)}
    for n in (CLS, BOX):
        for s in (".weight", ".bias"):
            P[n + s] = torch.tensor(base[n + s], requires_grad=True)
    params = [U, V] + list(P.values())
    opt = torch.optim.Adam(params, lr=lr, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps, eta_min=lr * 0.01)
    lin = torch.nn.functional.linear
    t0 = time.time()
    for it in range(steps):
        opt.zero_grad()
        h = torch.relu(lin(h1, w2, P[FC2 + ".bias"]) + (h1 @ V.t()) @ U.t())
        cls = lin(h, P[CLS + ".weight"], P[CLS + ".bias"])
        reg = lin(h, P[BOX + ".weight"], P[BOX + ".bias"])
        l_cls = torch.nn.functional.cross_entropy(cls, yc)
        l_box = (torch.nn.functional.smooth_l1_loss(reg, yb, beta=0.05, reduction="none").sum(dim=1) * wf).sum() / wf.sum().clamp(min=1)
        (l_cls + l_box).backward()
        opt.step()
        sched.step()
        if verbose and (it % 250 == 0 or it == steps - 1):
            print(f"  step {it}: class loss {float(l_cls):.4f}, box loss {float(l_box):.4f} ({time.time() - t0:.0f} s)", flush=True)
    out = {k: v.detach().numpy() for k, v in P.items()}
    out[FC2 + ".weight" + TH.LOWRANK_U] = U.detach().numpy()
    out[FC2 + ".weight" + TH.LOWRANK_V] = V.detach().numpy()
    for k in ("objectness_logits.weight", "objectness_logits.bias", "anchor_deltas.weight", "anchor_deltas.bias"):
        out[RPN + k] = sd[RPN + k]
    return TH.pack_heads(base, out)


def fp16_round(sd):
    return {k: (v.astype(np.float16).astype(np.float32) if v.dtype == np.float32 and v.ndim >= 2 else v) for k, v in sd.items()}


def check(name, sd, tiles):
    """CPU proxy of the fp16 comparison: the oracle on ``sd`` against the oracle on fp16-rounded weights."""
    from tests.test_engine_fp16_gpu import SCORE_THRESH, match_detection_sets
    band = 5e-3 * 4.0 * SCORE_THRESH * (1.0 - SCORE_THRESH) / 0.36
    inputs = TH.tile_inputs(tiles, 1000)
    ref = MaskRCNNOracle(sd).forward(inputs, paste=False)
    alt = MaskRCNNOracle(fp16_round(sd)).forward(inputs, paste=False)
    for n, (g, r) in enumerate(zip(alt, ref)):
        strict, cluster, lost, extra = match_detection_sets(g, r, band)
        es = max((abs(float(g["scores"][j]) - float(r["scores"][i])) for i, j, _ in strict), default=0.0)
        eb = max((float(np.abs(g["pred_boxes"][j] - r["pred_boxes"][i]).max()) for i, j, _ in strict), default=0.0)
        print(f"  [{name}] tile {tiles[n]}: {len(r['scores'])} detections (fp16-rounded weights: {len(g['scores'])}), {len(strict)} strict pairs, "
              f"cluster {[round(v, 2) for _, _, v in cluster]}, lost {np.round(lost, 3).tolist()}, extra {np.round(extra, 3).tolist()}, "
              f"worst strict pair: score {es:.1e}, box {eb:.2f} px; scores {np.round(np.sort(r['scores'])[[0, len(r['scores']) // 2, -1]], 3).tolist()}",
              flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=[])
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default=TH.GOLDEN)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    torch.use_deterministic_algorithms(True)
    names = a.names or list(TH.FIXTURES)
    for name in names:
        depth, seed, tiles = TH.FIXTURES[name]
        tiles = list(tiles)
        print(f"[{name}] R{depth}, weight seed {seed}, tiles {tiles}", flush=True)
        base = blob_mask_head(make_synthetic_state_dict(depth, seed=seed))
        fitted = TH.fit_trained_like_heads(base, tiles)
        tensors = train_box_head_cpu(fitted, base, tiles, steps=a.steps, rank=a.rank)
        path = os.path.join(a.out, f"trained_heads_{name}.npz")
        np.savez(path, **tensors)
        print(f"[{name}] wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB", flush=True)
        if a.check:
            check(name, TH.apply_heads(base, tensors), tiles)
    # the manifest covers every fixture file present, not only the ones just written
    lines = []
    for name in TH.FIXTURES:
        path = os.path.join(a.out, f"trained_heads_{name}.npz")
        if os.path.exists(path):
            lines.append(f"{TH._sha256(path)}  {os.path.basename(path)}\n")
    with open(os.path.join(a.out, "trained_heads.sha256"), "w") as f:
        f.writelines(lines)


if __name__ == "__main__":
    main()
