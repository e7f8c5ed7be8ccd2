"""GPU parity of the whole forward (td_engine_forward through the C ABI) against the torch-CPU oracle.

Two kinds of checks (SURVEY.md §8d tolerances):
  * end-to-end on the same seeded inputs: activations within a stated fp32 tolerance, detections matched one-to-one;
  * stage-wise on IDENTICAL inputs (the engine's own upstream tensors fed to the oracle): top-k indices, NMS keep
    order, proposals and detections bit-exact / index-exact.
"""
import numpy as np
import pytest
import torch

from oracle import ops_ref as R
from oracle.maskrcnn_ref import MaskRCNNOracle
from treedetection_amd.weights import make_synthetic_state_dict

pytestmark = pytest.mark.gpu

TOL_ACT = 2e-4      # |hip - oracle| <= TOL_ACT * max|oracle| for trunk activations (fp32, different sum order)
TOL_BOX = 1e-2      # px
TOL_SCORE = 1e-4
TOL_MASKP = 1e-3


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


@pytest.fixture(scope="module")
def setup():
    from treedetection_amd.engine import Engine
    torch.set_num_threads(8)
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    rng = np.random.default_rng(11)
    inputs = [{"image": smooth_image(rng, 256, 320), "height": 200, "width": 250},
              {"image": smooth_image(rng, 224, 288), "height": 300, "width": 390}]
    oracle = MaskRCNNOracle(sd)
    ref, taps = oracle.forward(inputs, return_taps=True)
    eng = Engine(sd)
    got = eng(inputs)
    return dict(sd=sd, inputs=inputs, oracle=oracle, ref=ref, taps=taps, eng=eng, got=got)


def nchw(t):
    return t.cpu().numpy().transpose(0, 3, 1, 2)


def test_trunk_activations(setup):
    eng, taps = setup["eng"], setup["taps"]
    for name in ("stem", "pool"):
        ref = taps["res"][name].numpy()
        got = nchw(eng.tensor(name))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= TOL_ACT * np.abs(ref).max(), name
    for name in ("res2", "res3", "res4", "res5"):
        ref = taps["res"][name].numpy()
        got = nchw(eng.tensor(name))
        assert np.abs(got - ref).max() <= TOL_ACT * np.abs(ref).max(), name
    for name in ("p2", "p3", "p4", "p5", "p6"):
        ref = taps["feats"][name].numpy()
        got = nchw(eng.tensor(name))
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= TOL_ACT * np.abs(ref).max(), name


def test_rpn_head(setup):
    eng, taps = setup["eng"], setup["taps"]
    for li in range(5):
        head = eng.tensor(f"rpn_head{li + 2}").cpu().numpy()      # [B,h,w,15]
        B = head.shape[0]
        logits = head[..., :3].reshape(B, -1)
        deltas = head[..., 3:].reshape(B, -1, 4)
        rl, rd = taps["rpn_logits"][li].numpy(), taps["rpn_deltas"][li].numpy()
        assert np.abs(logits - rl).max() <= TOL_ACT * max(1.0, np.abs(rl).max())
        assert np.abs(deltas - rd).max() <= TOL_ACT * max(1.0, np.abs(rd).max())


def test_rpn_topk_nms_bit_exact_on_identical_inputs(setup):
    """Feed the ENGINE's RPN head outputs to the oracle's proposal stage: top-k indices, per-level NMS keep lists and
    the merged proposal order must agree exactly; proposal coordinates to 1e-4 px (expf ulp)."""
    eng, oracle, taps = setup["eng"], setup["oracle"], setup["taps"]
    logits, deltas, feat_hw = [], [], []
    for li in range(5):
        head = eng.tensor(f"rpn_head{li + 2}").cpu()
        B, h, w, _ = head.shape
        logits.append(head[..., :3].reshape(B, -1).contiguous())
        deltas.append(head[..., 3:].reshape(B, -1, 4).contiguous())
        feat_hw.append((h, w))
    props, rtaps = oracle.rpn_proposals(logits, deltas, feat_hw, taps["sizes"])
    cand_idx = eng.tensor("rpn_cand_idx").cpu().numpy()
    cand_valid = eng.tensor("rpn_cand_valid").cpu().numpy()
    cand_boxes = eng.tensor("rpn_cand_boxes").cpu().numpy()
    keep = eng.tensor("rpn_keep").cpu().numpy()
    keep_count = eng.tensor("rpn_keep_count").cpu().numpy()
    gp = eng.tensor("proposals").cpu().numpy()
    gs = eng.tensor("proposal_scores").cpu().numpy()
    gc = eng.tensor("proposal_count").cpu().numpy()
    for n in range(len(props)):
        for li in range(5):
            ref_idx = rtaps[n]["per_level"][li]["topk_idx"]
            k = len(ref_idx)
            assert np.array_equal(cand_idx[n, li, :k], ref_idx), (n, li)
            assert (cand_idx[n, li, k:] == -1).all()
            assert np.abs(cand_boxes[n, li, :k][cand_valid[n, li, :k] > 0] -
                          R.clip_boxes(rtaps[n]["per_level"][li]["decoded"], *taps["sizes"][n])[cand_valid[n, li, :k] > 0]).max() < 1e-3
        rb, rs = props[n]
        assert gc[n] == len(rs)
        assert np.array_equal(gs[n, : gc[n]], rs), "proposal scores / order differ"
        assert np.abs(gp[n, : gc[n]] - rb).max() < 1e-3
    # per-level keep lists: recompute with the oracle on the ENGINE's candidate boxes (bit-identical inputs)
    cand_scores = eng.tensor("rpn_cand_scores").cpu().numpy()
    for n in range(keep.shape[0]):
        for li in range(5):
            v = cand_valid[n, li] > 0
            idx = np.nonzero(v)[0]
            ref_keep = idx[R.nms(cand_boxes[n, li][v], cand_scores[n, li][v], 0.7)]
            assert keep_count[n, li] == len(ref_keep)
            assert np.array_equal(keep[n, li, : len(ref_keep)], ref_keep), (n, li)


def test_proposals_end_to_end(setup):
    eng, taps = setup["eng"], setup["taps"]
    gp = eng.tensor("proposals").cpu().numpy()
    gc = eng.tensor("proposal_count").cpu().numpy()
    for n, (rb, rs) in enumerate(taps["proposals"]):
        assert abs(int(gc[n]) - len(rs)) <= 2
        m = min(int(gc[n]), len(rs))
        # allow a handful of order flips from 1e-6-level logit differences
        same = np.abs(gp[n, :m] - rb[:m]).max(axis=1) < TOL_BOX
        assert same.mean() > 0.98


def test_box_head_and_detections_on_identical_inputs(setup):
    eng, oracle, taps = setup["eng"], setup["oracle"], setup["taps"]
    props = eng.tensor("proposals").cpu().numpy()
    pc = eng.tensor("proposal_count").cpu().numpy()
    feats = {f"p{l}": torch.from_numpy(nchw(eng.tensor(f"p{l}"))) for l in (2, 3, 4, 5)}
    P = props.shape[1]
    pooled_g = eng.tensor("pooled7").cpu().numpy()       # [B*P,7,7,C]
    pred_g = eng.tensor("box_pred").cpu().numpy()
    got = setup["got"]
    for n in range(props.shape[0]):
        pb = props[n, : pc[n]]
        pooled_ref, _ = oracle.roi_pool({k: v[n:n + 1] for k, v in feats.items()}, [pb], 7)
        g = pooled_g[n * P: n * P + pc[n]].transpose(0, 3, 1, 2)
        assert np.abs(g - pooled_ref[0]).max() <= 1e-5 * max(1.0, np.abs(pooled_ref[0]).max())
        cls, reg = oracle.box_head(g)
        assert np.abs(pred_g[n * P: n * P + pc[n], :2] - cls).max() <= 2e-4 * max(1.0, np.abs(cls).max())
        assert np.abs(pred_g[n * P: n * P + pc[n], 2:] - reg).max() <= 2e-4 * max(1.0, np.abs(reg).max())
        # detections from the ENGINE's predictor outputs: identical inputs → same kept set and order
        b, s, t = oracle.detections(pred_g[n * P: n * P + pc[n], :2], pred_g[n * P: n * P + pc[n], 2:], pb, taps["sizes"][n])
        probs = np.zeros((len(s), 28, 28), np.float32)
        ob, os_, _, _ = oracle.postprocess(b, s, probs, taps["sizes"][n],
                                           (setup["inputs"][n]["height"], setup["inputs"][n]["width"]), paste=False)
        assert len(got[n]["scores"]) == len(os_)
        assert np.abs(got[n]["scores"] - os_).max() <= 1e-6
        assert np.abs(got[n]["pred_boxes"] - ob).max() <= 1e-3


def test_end_to_end_detections_and_masks(setup):
    got, ref = setup["got"], setup["ref"]
    for n in range(len(ref)):
        g, r = got[n], ref[n]
        assert len(r["scores"]) > 3, "test image yields too few detections to be meaningful"
        # one-to-one matching by box distance
        used, pairs = set(), []
        for i in range(len(r["scores"])):
            d = np.abs(g["pred_boxes"] - r["pred_boxes"][i]).max(axis=1) if len(g["scores"]) else np.array([])
            j = int(np.argmin(d)) if d.size else -1
            if j >= 0 and d[j] <= TOL_BOX and j not in used:
                used.add(j)
                pairs.append((i, j))
        assert len(pairs) == len(r["scores"]) == len(g["scores"])          # the oracle's detection set, exactly
        for i, j in pairs:
            assert abs(g["scores"][j] - r["scores"][i]) <= TOL_SCORE
            assert np.abs(g["mask_probs"][j] - r["mask_probs"][i]).max() <= TOL_MASKP
            a, b = g["pred_masks"][j], r["pred_masks"][i]
            inter, union = (a & b).sum(), (a | b).sum()
            assert union == 0 or inter / union >= 0.995
        assert g["pred_classes"].dtype == np.int64 and (g["pred_classes"] == 0).all()
        assert (np.diff(g["scores"]) <= 0).all()


def test_mask_head_on_identical_inputs(setup):
    eng, oracle = setup["eng"], setup["oracle"]
    got = setup["got"]
    feats = {f"p{l}": torch.from_numpy(nchw(eng.tensor(f"p{l}"))) for l in (2, 3, 4, 5)}
    dbn = eng.tensor("det_boxes_net").cpu().numpy()
    for n in range(len(got)):
        k = len(got[n]["scores"])
        pooled, _ = oracle.roi_pool({kk: v[n:n + 1] for kk, v in feats.items()}, [dbn[n, :k]], 14)
        probs = oracle.mask_head(pooled[0])
        assert np.abs(probs - got[n]["mask_probs"]).max() <= 2e-4
        ref_masks = R.paste_masks(got[n]["mask_probs"], got[n]["pred_boxes"], setup["inputs"][n]["height"],
                                  setup["inputs"][n]["width"], 0.5)
        assert np.array_equal(ref_masks, got[n]["pred_masks"])     # paste is bit-exact on identical inputs


def test_rpn_topk_with_massive_ties(setup):
    """All objectness logits equal (weights and bias zeroed): every level's top-k is decided purely by the tie rule
    (lower index first) — the ordered-compaction path of the top-k kernel. Compared with the oracle on the engine's
    own head outputs: indices must agree exactly."""
    from treedetection_amd.engine import Engine
    sd = dict(setup["sd"])
    k = "proposal_generator.rpn_head.objectness_logits"
    sd[k + ".weight"] = np.zeros_like(sd[k + ".weight"])
    sd[k + ".bias"] = np.zeros_like(sd[k + ".bias"])
    eng = Engine(sd)
    eng(setup["inputs"])
    oracle = MaskRCNNOracle(sd)
    logits, deltas, feat_hw = [], [], []
    for li in range(5):
        head = eng.tensor(f"rpn_head{li + 2}").cpu()
        B, h, w, _ = head.shape
        assert float(head[..., :3].abs().max()) == 0.0
        logits.append(head[..., :3].reshape(B, -1).contiguous())
        deltas.append(head[..., 3:].reshape(B, -1, 4).contiguous())
        feat_hw.append((h, w))
    props, rtaps = oracle.rpn_proposals(logits, deltas, feat_hw, setup["taps"]["sizes"])
    cand_idx = eng.tensor("rpn_cand_idx").cpu().numpy()
    gs = eng.tensor("proposal_scores").cpu().numpy()
    gc = eng.tensor("proposal_count").cpu().numpy()
    gp = eng.tensor("proposals").cpu().numpy()
    for n in range(len(props)):
        for li in range(5):
            ref_idx = rtaps[n]["per_level"][li]["topk_idx"]
            assert np.array_equal(cand_idx[n, li, : len(ref_idx)], ref_idx), (n, li)
            assert np.array_equal(ref_idx, np.arange(len(ref_idx)))      # lowest indices win the tie
        assert gc[n] == len(props[n][1])
        assert np.abs(gp[n, : gc[n]] - props[n][0]).max() < 1e-3


def test_resnet101_layout(setup):
    """The reference hard-codes R101 (config.py:25): blocks [3,4,23,3] are read from the checkpoint keys."""
    from treedetection_amd.engine import Engine
    from treedetection_amd.weights import infer_depth
    sd = make_synthetic_state_dict(101, seed=9, width_div=2)
    assert infer_depth(sd) == 101
    inputs = setup["inputs"][:1]
    ref, taps = MaskRCNNOracle(sd).forward(inputs, return_taps=True)
    eng = Engine(sd)
    got = eng(inputs)
    r4 = taps["res"]["res4"].numpy()
    assert np.abs(nchw(eng.tensor("res4")) - r4).max() <= TOL_ACT * np.abs(r4).max()
    assert len(got[0]["scores"]) == len(ref[0]["scores"])
    m = min(len(got[0]["scores"]), len(ref[0]["scores"]))
    assert m > 0 and np.abs(np.sort(got[0]["scores"])[-m:] - np.sort(ref[0]["scores"])[-m:]).max() <= 1e-3


def test_mask_head_beyond_the_fold_kernels_plane_limit(setup):
    """the mask head launches all B x detections_per_image reserved RoIs as ONE group; the folded F(4x4) kernel's
    32-bit plane offsets hold 36 * tiles * C * 4 < 4 GB (about 14 500 RoIs of 14 x 14 at this fixture's 128 channels, 7 279 at
    256). A configuration past that (16 x 1024 RoIs) must take the three-launch form instead of failing the forward — and give
    the detections of the default configuration (same boxes and scores bit for bit; mask probabilities within the fp32
    tolerance: the two Winograd forms associate their sums differently)."""
    from treedetection_amd.engine import Engine
    inputs = [setup["inputs"][k % 2] for k in range(16)]
    big = Engine(setup["sd"], detections_per_image=1024)
    try:
        got = big(inputs, paste=False)
    finally:
        big.close()
    ref = setup["eng"](inputs, paste=False)
    for g, r in zip(got, ref):
        assert len(r["scores"]) < 100                       # (else the two top-D limits would cut different sets)
        assert len(g["scores"]) == len(r["scores"]) > 0
        assert (g["pred_boxes"] == r["pred_boxes"]).all() and (g["scores"] == r["scores"]).all()
        assert np.abs(g["mask_probs"] - r["mask_probs"]).max() <= TOL_MASKP
