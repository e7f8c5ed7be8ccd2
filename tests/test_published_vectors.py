"""Known-answer vectors PUBLISHED by the third-party dependencies the path delegates to (the reference
itself holds no tests; its arithmetic lives in detectron2 v0.6 / torchvision, both absent here). These are the literal
expected values of those projects' own unit tests — data, restated from their test files, not code — and they pin the
oracle restatement (CPU) and the HIP ops behind the C ABI (GPU) to the dependency rather than to each other:

* detectron2 `tests/layers/test_roi_align.py::ROIAlignTest::test_forward_output` — 5x5 ramp, box (1, 1, 3, 3), 4x4 bins,
  `aligned=True` ("with 0.5 correction"), the variant `ROIPooler` uses (`detectron2/modeling/poolers.py`, ROIAlignV2).
* detectron2 `tests/modeling/test_anchor_generator.py::TestAnchorGenerator::test_default_anchor_generator` — sizes (32, 64),
  ratios (0.25, 1, 4), stride 4, a 1x2 feature map, offset 0.
* detectron2 `tests/structures/test_boxes.py::TestBoxIOU::test_pairwise_iou` — the IoU table of a unit box against six
  others; checked through the only place the path uses IoU, the strict `>` of NMS (torchvision `nms`).
* detectron2 `tests/modeling/test_box2box_transform.py::test_reconstruction` — apply_deltas(get_deltas(src, dst)) == dst for
  weights (5, 5, 10, 10); a property, run on seeded boxes.
"""
import numpy as np
import pytest

from oracle import ops_ref as R

ROI_RAMP = np.arange(25, dtype=np.float32).reshape(1, 5, 5)
ROI_BOX = np.array([[1, 1, 3, 3]], dtype=np.float32)
ROI_ALIGNED_EXPECTED = np.array([[4.5, 5.0, 5.5, 6.0],
                                 [7.0, 7.5, 8.0, 8.5],
                                 [9.5, 10.0, 10.5, 11.0],
                                 [12.0, 12.5, 13.0, 13.5]], dtype=np.float32)

ANCHORS_EXPECTED = np.array([[-32.0, -8.0, 32.0, 8.0],
                             [-16.0, -16.0, 16.0, 16.0],
                             [-8.0, -32.0, 8.0, 32.0],
                             [-64.0, -16.0, 64.0, 16.0],
                             [-32.0, -32.0, 32.0, 32.0],
                             [-16.0, -64.0, 16.0, 64.0],
                             [-28.0, -8.0, 36.0, 8.0],       # -28 = -32 + stride 4
                             [-12.0, -16.0, 20.0, 16.0],
                             [-4.0, -32.0, 12.0, 32.0],
                             [-60.0, -16.0, 68.0, 16.0],
                             [-28.0, -32.0, 36.0, 32.0],
                             [-12.0, -64.0, 20.0, 64.0]], dtype=np.float32)

IOU_BOX = np.array([0.0, 0.0, 1.0, 1.0], dtype=np.float32)
IOU_OTHERS = np.array([[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.5, 1.0], [0.0, 0.0, 1.0, 0.5], [0.0, 0.0, 0.5, 0.5],
                       [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 1.5, 1.5]], dtype=np.float32)
IOU_EXPECTED = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]


def _nms_pair_suppressed(nms_fn, other, thr):
    """The higher-scored unit box first, `other` second: True when NMS at `thr` drops `other`."""
    keep = nms_fn(np.stack([IOU_BOX, other]), np.array([0.9, 0.8], dtype=np.float32), thr)
    assert keep[0] == 0
    return len(keep) == 1


def _check_iou_table_through_nms(nms_fn):
    for other, iou in zip(IOU_OTHERS, IOU_EXPECTED):
        if iou < 1.0:
            assert not _nms_pair_suppressed(nms_fn, other, np.float32(iou + 1e-3)), iou
        assert _nms_pair_suppressed(nms_fn, other, np.float32(iou - 1e-3)), iou
    # the comparison is strict (torchvision: `iou > threshold`); 0.5 and 0.25 are exact in float32
    assert not _nms_pair_suppressed(nms_fn, IOU_OTHERS[1], np.float32(0.5))
    assert not _nms_pair_suppressed(nms_fn, IOU_OTHERS[3], np.float32(0.25))


def test_oracle_roi_align_detectron2_vector():
    for fn in (R.roi_align, R.roi_align_fast):
        got = np.asarray(fn(ROI_RAMP, ROI_BOX, 1.0, 4))
        assert got.shape == (1, 1, 4, 4)
        assert np.array_equal(got[0, 0], ROI_ALIGNED_EXPECTED)


def test_oracle_anchors_detectron2_vector():
    per_size = [R.cell_anchors(s, (0.25, 1.0, 4.0)) for s in (32.0, 64.0)]
    cell = np.concatenate(per_size, axis=0)                       # detectron2 order: sizes outer, ratios inner
    assert np.allclose(cell, ANCHORS_EXPECTED[:6], rtol=0, atol=1e-5)
    grid = np.concatenate([cell + np.array([4.0 * x, 0, 4.0 * x, 0], dtype=np.float32) for x in range(2)], axis=0)
    assert np.allclose(grid, ANCHORS_EXPECTED, rtol=0, atol=1e-5)
    g = R.grid_anchors(1, 2, 4, 32.0)                             # the path's ratios (0.5, 1, 2): x shift = stride, y shift 0
    assert np.allclose(g[3:] - g[:3], [4, 0, 4, 0])


def test_oracle_nms_iou_detectron2_vector():
    _check_iou_table_through_nms(R.nms)


def test_oracle_box2box_reconstruction():
    w = (5.0, 5.0, 10.0, 10.0)
    rng = np.random.default_rng(0)

    def boxes(n):
        xy = rng.uniform(0, 50, (n, 2))
        wh = rng.uniform(1, 50, (n, 2))
        return np.concatenate([xy, xy + wh], axis=1).astype(np.float32)

    src, dst = boxes(10), boxes(10)
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    dw, dh = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    dx, dy = dst[:, 0] + 0.5 * dw, dst[:, 1] + 0.5 * dh
    deltas = np.stack([w[0] * (dx - sx) / sw, w[1] * (dy - sy) / sh, w[2] * np.log(dw / sw), w[3] * np.log(dh / sh)], axis=1)
    rec = R.apply_deltas(deltas.astype(np.float32), src, w)
    assert np.allclose(rec, dst, rtol=0, atol=1e-3)


@pytest.mark.gpu
def test_hip_roi_align_detectron2_vector():
    import torch

    from tests.gpu_util import dev
    from treedetection_amd import _lib

    lib = _lib.load()
    C = 4                                                        # td_roi_align wants C % 4 == 0: the ramp in every channel
    feat = np.ascontiguousarray(np.repeat(ROI_RAMP, C, axis=0).transpose(1, 2, 0))
    out = torch.full((1, 4, 4, C), -1.0, dtype=torch.float32, device="cuda")
    f, r = dev(feat), dev(ROI_BOX)
    _lib.check(lib.td_roi_align(f.data_ptr(), 5, 5, C, r.data_ptr(), 1, 1.0, 4, out.data_ptr(), 0, _lib.stream_ptr()),
               "td_roi_align")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for c in range(C):
        assert np.array_equal(got[0, :, :, c], ROI_ALIGNED_EXPECTED), got[0, :, :, c]


@pytest.mark.gpu
def test_hip_nms_iou_detectron2_vector():
    from tests.test_ops_gpu import nms_hip

    _check_iou_table_through_nms(nms_hip)


def test_oracle_roi_align_bilinear_against_scipy_map_coordinates():
    """The bilinear sampler itself, against scipy.ndimage.map_coordinates(order=1) — code this build did not write. The
    sampling grid is the published one (torchvision `roi_align_kernel`: aligned offset 0.5, ceil(roi / pooled) samples per
    bin and axis, bin mean); boxes stay inside the map so that the kernel's border rules (tested elsewhere) do not enter."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(42)
    C, H, W, pooled, scale = 3, 23, 31, 7, 0.25
    feat = rng.standard_normal((C, H, W)).astype(np.float32)
    xy = rng.uniform(4, 50, (12, 2))
    wh = rng.uniform(3, 60, (12, 2))
    rois = np.concatenate([xy, np.minimum(xy + wh, [(W - 1) / scale, (H - 1) / scale])], axis=1).astype(np.float32)
    got = np.asarray(R.roi_align(feat, rois, scale, pooled))
    for r, (x1, y1, x2, y2) in enumerate(rois.astype(np.float64) * scale - 0.5):
        bw, bh = (x2 - x1) / pooled, (y2 - y1) / pooled
        gx, gy = int(np.ceil((x2 - x1) / pooled)), int(np.ceil((y2 - y1) / pooled))
        ys = y1 + (np.arange(pooled)[:, None] + (np.arange(gy)[None, :] + 0.5) / gy) * bh       # [pooled, gy]
        xs = x1 + (np.arange(pooled)[:, None] + (np.arange(gx)[None, :] + 0.5) / gx) * bw
        Y = np.broadcast_to(ys[:, None, :, None], (pooled, pooled, gy, gx))
        X = np.broadcast_to(xs[None, :, None, :], (pooled, pooled, gy, gx))
        assert Y.min() >= 0 and X.min() >= 0 and Y.max() <= H - 1 and X.max() <= W - 1
        for c in range(C):
            want = ndi.map_coordinates(feat[c].astype(np.float64), [Y.ravel(), X.ravel()], order=1).reshape(Y.shape).mean(axis=(2, 3))
            assert np.allclose(got[r, c], want, rtol=0, atol=5e-6), (r, c, np.abs(got[r, c] - want).max())
