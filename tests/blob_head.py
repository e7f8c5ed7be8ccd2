"""A mask head whose OUTPUT is a compact blob (what a trained crown segmenter produces), for the fp16 mask-IoU statement.

The seeded random mask head of ``make_synthetic_state_dict`` yields noise-like masks (boundary pixels ~ 2 x area), on
which IoU measures the fixture rather than the engine. This construction keeps every kernel of the mask branch in play
(RoIAlign 14x14 of the real FPN features → four 3x3 convs + ReLU → 2x2 deconv + ReLU → 1x1 predictor → sigmoid → paste)
and only chooses the weights:
  * ``mask_fcn1``: channels 0..NF-1 = 3x3 box filter of the FPN channels 0..NF-1 (a smoothing filter, near identity);
    channel NF = constant 1 (bias only). All other output channels zero.
  * ``mask_fcn2..4``: 3x3 box filters channel by channel: the feature channels get smoother, and the constant channel —
    zero-padded at the RoI border by each conv — turns into a bump that peaks at the RoI centre and falls off towards
    its border (three box filters of the 14x14 indicator function).
  * ``deconv``: nearest 2x up-sampling of those channels (weight 1 on the diagonal for all four taps).
  * ``predictor``: logit = GAIN * (bump - LEVEL + sum_c sign_c * AMP * feature_c): the level set of a smooth function —
    a rounded blob around the RoI centre whose outline the (smoothed, real) features push in and out.
"""
import numpy as np

NF = 8          # feature channels that shape the outline
GAIN = 24.0     # logit slope: a trained head is confident away from the outline
LEVEL = 0.8     # bump level of the outline (bump: 1 at the centre, ~0.3 in the corners): the blob stays clear of the RoI border, so a
                # box edge that crosses an integer (the paste region moves by one pixel column) cannot flip a column of the mask
AMP = 0.06      # how much the features move the outline


def blob_mask_head(sd, seed=0):
    sd = dict(sd)
    rng = np.random.default_rng(seed)
    c = sd["roi_heads.mask_head.mask_fcn1.weight"].shape[0]
    box = np.full((3, 3), 1.0 / 9.0, np.float32)
    for i in range(1, 5):
        w = np.zeros((c, c, 3, 3), np.float32)
        b = np.zeros((c,), np.float32)
        for ch in range(NF):
            w[ch, ch] = box
        if i == 1:
            b[NF] = 1.0
        else:
            w[NF, NF] = box
        sd[f"roi_heads.mask_head.mask_fcn{i}.weight"] = w
        sd[f"roi_heads.mask_head.mask_fcn{i}.bias"] = b
    wd = np.zeros((c, c, 2, 2), np.float32)            # ConvTranspose2d weight [Cin, Cout, 2, 2]
    for ch in range(NF + 1):
        wd[ch, ch] = 1.0
    sd["roi_heads.mask_head.deconv.weight"] = wd
    sd["roi_heads.mask_head.deconv.bias"] = np.zeros((c,), np.float32)
    wp = np.zeros((1, c, 1, 1), np.float32)
    signs = rng.choice([-1.0, 1.0], NF).astype(np.float32)
    wp[0, :NF, 0, 0] = GAIN * AMP * signs
    wp[0, NF, 0, 0] = GAIN
    sd["roi_heads.mask_head.predictor.weight"] = wp
    sd["roi_heads.mask_head.predictor.bias"] = np.array([-GAIN * LEVEL], np.float32)
    return sd


def boundary_over_area(mask):
    """Boundary pixels (4-neighbour changes, counted like tests/test_engine_fp16_gpu.py) over the mask's area."""
    m = np.pad(mask.astype(bool), 1)
    boundary = int((m ^ np.roll(m, 1, 0)).sum() + (m ^ np.roll(m, 1, 1)).sum())
    return boundary / max(int(m.sum()), 1)
