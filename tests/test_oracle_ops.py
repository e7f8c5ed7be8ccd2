"""Known-answer tests that pin the oracle's op restatements (the reference has no fixtures of its
own for this path, so closed-form expectations stand in). CPU only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_ref as R


def test_nms_closed_form():
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], dtype=np.float32)
    # IoU(0,1) = 81/119 = 0.68; duplicates tie → lower index first
    assert list(R.nms(boxes, scores, 0.5)) == [0, 2]
    assert list(R.nms(boxes, scores, 0.7)) == [0, 1, 2]
    assert list(R.nms(boxes, scores, 1.0)) == [0, 3, 1, 2]
    # IoU exactly at the threshold is kept (strict >)
    two = np.array([[0, 0, 10, 10], [0, 0, 20, 10]], dtype=np.float32)
    assert list(R.nms(two, np.array([1.0, 0.5], np.float32), 0.5)) == [0, 1]
    assert list(R.nms(two, np.array([1.0, 0.5], np.float32), 0.49)) == [0]
    assert R.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)


def test_batched_nms_is_per_category_and_score_sorted():
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 10]], dtype=np.float32)
    scores = np.array([0.5, 0.9, 0.7], dtype=np.float32)
    assert list(R.batched_nms(boxes, scores, np.array([0, 1, 0]), 0.5)) == [1, 2]


def test_anchors_closed_form():
    a = R.cell_anchors(32)
    # ratio 0.5: w = sqrt(1024/0.5) = 45.25, h = 22.63; ratio 1: 32x32; ratio 2: 22.63 x 45.25
    assert np.allclose(a[1], [-16, -16, 16, 16])
    assert np.allclose(a[0], [-22.627417, -11.313708, 22.627417, 11.313708])
    assert np.allclose(a[2], [-11.313708, -22.627417, 11.313708, 22.627417])
    g = R.grid_anchors(2, 3, 4, 32)
    assert g.shape == (18, 4)
    assert np.allclose(g[3 * (1 * 3 + 2) + 1], [8 - 16, 4 - 16, 8 + 16, 4 + 16])   # (y=1, x=2, a=1)


def test_apply_deltas_closed_form():
    box = np.array([[10, 20, 30, 60]], dtype=np.float32)   # w 20 h 40 centre (20, 40)
    out = R.apply_deltas(np.array([[0.1, -0.2, math.log(2), 0.0]], np.float32), box)
    assert np.allclose(out, [[22 - 20, 32 - 20, 22 + 20, 32 + 20]], atol=1e-4)
    out = R.apply_deltas(np.array([[1.0, 1.0, 5.0, 5.0]], np.float32), box, (10, 10, 5, 5))
    # dx = .1 → cx 22; dy = .1 → cy 44; dw = 1 → w = 20e
    assert np.allclose(out, [[22 - 10 * math.e, 44 - 20 * math.e, 22 + 10 * math.e, 44 + 20 * math.e]], rtol=1e-5)
    # clamp: exp(log(1000/16)) = 62.5
    out = R.apply_deltas(np.array([[0, 0, 100.0, 100.0]], np.float32), box)
    assert np.allclose(out[0, 2] - out[0, 0], 20 * 62.5, rtol=1e-5)


def test_level_assign():
    def sq(s):
        return [0, 0, s, s]
    lv = R.level_assign(np.array([sq(10), sq(111), sq(112), sq(223.9), sq(224), sq(447), sq(448), sq(2000), sq(0)], np.float32))
    assert list(lv) == [0, 0, 1, 1, 2, 2, 3, 3, 0]


def test_roi_align_linear_ramp_and_scalar_vs_vectorised():
    H, W = 20, 30
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    feat = np.stack([xs, ys, 2 * xs - ys]).astype(np.float32)
    rois = np.array([[8, 8, 64, 48], [20.5, 12.25, 90.75, 70.5], [0, 0, 4, 4], [100, 60, 130, 90]], dtype=np.float32)
    slow = R.roi_align(feat, rois, 0.25, 7)
    fast = R.roi_align_fast(feat, rois, 0.25, 7)
    assert np.array_equal(slow, fast)
    for r in range(2):   # fully inside: value = f(bin centre)
        x0, y0 = rois[r, 0] * 0.25 - 0.5, rois[r, 1] * 0.25 - 0.5
        bw, bh = (rois[r, 2] - rois[r, 0]) * 0.25 / 7, (rois[r, 3] - rois[r, 1]) * 0.25 / 7
        cx = x0 + (np.arange(7) + 0.5) * bw
        cy = y0 + (np.arange(7) + 0.5) * bh
        assert np.allclose(slow[r, 0], np.broadcast_to(cx[None], (7, 7)), atol=1e-4)
        assert np.allclose(slow[r, 1], np.broadcast_to(cy[:, None], (7, 7)), atol=1e-4)


def test_roi_align_random_scalar_vs_vectorised():
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((8, 17, 23), dtype=np.float32)
    xy = rng.uniform(-20, 150, (40, 2))
    wh = rng.uniform(0, 120, (40, 2))
    rois = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    assert np.array_equal(R.roi_align(feat, rois, 0.125, 7), R.roi_align_fast(feat, rois, 0.125, 7))
    assert np.array_equal(R.roi_align(feat, rois[:10], 0.125, 14), R.roi_align_fast(feat, rois[:10], 0.125, 14))


def test_paste_matches_torch_grid_sample():
    """The oracle's paste against torch's own grid_sample, composed as detectron2's _do_paste_mask composes it."""
    rng = np.random.default_rng(5)
    h, w = 90, 130
    probs = rng.uniform(0, 1, (6, 28, 28)).astype(np.float32)
    boxes = np.array([[0, 0, w, h], [10.3, 20.7, 55.1, 61.9], [100, 50, 130, 90], [3, 3, 4.5, 80], [60, 10, 61, 11],
                      [-0.0, 0, 30, 30]], dtype=np.float32)
    for i in range(len(boxes)):
        vals, (x0, y0, x1, y1) = R.paste_mask_values(probs[i], boxes[i], h, w)
        b = boxes[i]
        img_y = (torch.arange(y0, y1, dtype=torch.float32) + 0.5 - b[1]) / (b[3] - b[1]) * 2 - 1
        img_x = (torch.arange(x0, x1, dtype=torch.float32) + 0.5 - b[0]) / (b[2] - b[0]) * 2 - 1
        gx = img_x[None, None, :].expand(1, len(img_y), len(img_x))
        gy = img_y[None, :, None].expand(1, len(img_y), len(img_x))
        ref = F.grid_sample(torch.from_numpy(probs[i])[None, None], torch.stack([gx, gy], dim=3), align_corners=False)[0, 0].numpy()
        assert np.abs(ref - vals).max() < 2e-6
        assert ((ref >= 0.5) != (vals >= 0.5)).sum() <= 1


def test_paste_region_and_constant_mask():
    assert R.paste_region(np.array([10.2, 20.9, 30.1, 40.0], np.float32), 100, 100) == (9, 19, 32, 41)
    assert R.paste_region(np.array([0, 0, 100, 100], np.float32), 100, 100) == (0, 0, 100, 100)
    m = R.paste_masks(np.ones((1, 28, 28), np.float32), np.array([[10, 20, 66, 76]], np.float32), 100, 100)
    assert m[0, 20:76, 10:66].all() and m[0].sum() == 56 * 56
