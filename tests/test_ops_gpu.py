"""GPU parity of the op-level C-ABI entry points against the oracle (oracle/ops_ref.py) on seeded inputs.

NMS / resize: bit-exact (index order, bytes). RoIAlign / paste: same float32 operation order → exact or within
1 ulp-scale tolerance stated per test."""
import numpy as np
import pytest
import torch

from oracle import ops_ref as R
from tests.gpu_util import dev
from treedetection_amd import _lib

pytestmark = pytest.mark.gpu


def nms_hip(boxes, scores, thr):
    lib = _lib.load()
    n = boxes.shape[0]
    b = dev(boxes.astype(np.float32).reshape(-1, 4)) if n else torch.zeros((1, 4), device="cuda")
    s = dev(scores.astype(np.float32)) if n else torch.zeros((1,), device="cuda")
    keep = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.td_nms(b.data_ptr(), s.data_ptr(), n, thr, keep.data_ptr(), cnt.data_ptr(), _lib.stream_ptr()), "td_nms")
    torch.cuda.synchronize()
    c = int(cnt.item())
    return keep.cpu().numpy()[:c].astype(np.int64)


def random_boxes(rng, n, span=200.0, size=60.0):
    xy = rng.uniform(0, span, (n, 2)).astype(np.float32)
    wh = rng.uniform(1, size, (n, 2)).astype(np.float32)
    return np.concatenate([xy, xy + wh], axis=1).astype(np.float32)


@pytest.mark.parametrize("n,thr,seed", [(0, 0.5, 0), (1, 0.5, 1), (2, 0.5, 2), (63, 0.7, 3), (64, 0.5, 4), (65, 0.3, 5),
                                        (200, 0.5, 6), (1000, 0.7, 7), (1024, 0.5, 8), (777, 0.1, 9)])
def test_nms_index_order_bit_exact(n, thr, seed):
    rng = np.random.default_rng(seed)
    boxes = random_boxes(rng, n)
    scores = rng.standard_normal(n).astype(np.float32)
    got = nms_hip(boxes, scores, thr)
    ref = R.nms(boxes, scores, thr)
    assert np.array_equal(got, ref)


def test_nms_ties_and_threshold_edge():
    # duplicated scores (ties → lower index first) and IoU exactly at the threshold (strict >, so both stay)
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [5, 0, 15, 10], [0, 0, 10, 10], [100, 100, 110, 110],
                      [0, 0, 20, 10]], dtype=np.float32)
    scores = np.array([0.5, 0.9, 0.9, 0.9, 0.1, 0.9], dtype=np.float32)
    for thr in (1.0 / 3.0, 0.5, 0.3, 0.0):
        assert np.array_equal(nms_hip(boxes, scores, thr), R.nms(boxes, scores, thr)), thr
    # box 0 vs box 5: inter 100, union 200 → IoU exactly 0.5: not suppressed at thr 0.5
    two = np.array([[0, 0, 10, 10], [0, 0, 20, 10]], dtype=np.float32)
    assert list(nms_hip(two, np.array([1.0, 0.5], np.float32), 0.5)) == [0, 1]


def test_nms_dense_cluster_1000():
    rng = np.random.default_rng(42)
    base = random_boxes(rng, 40, span=300, size=80)
    boxes = (base[rng.integers(0, 40, 1000)] + rng.normal(0, 2.0, (1000, 4))).astype(np.float32)
    scores = np.round(rng.uniform(0, 1, 1000), 2).astype(np.float32)   # many exact ties
    for thr in (0.5, 0.7):
        assert np.array_equal(nms_hip(boxes, scores, thr), R.nms(boxes, scores, thr))


@pytest.mark.parametrize("n,thr,seed,dense", [(1025, 0.5, 11, False), (4507, 0.7, 12, False), (4507, 0.5, 13, True), (9000, 0.3, 14, True)])
def test_nms_beyond_1024_boxes_bit_exact(n, thr, seed, dense):
    """td_nms above one block's capacity (counting sort + bit matrix + LDS-resident scan, rpn.hip): 4 507 = the candidates of one
    image over the five RPN levels, the size torchvision's unbatched nms sees; index order bit-exact vs the
    oracle, ties (scores rounded to 2 decimals) included."""
    rng = np.random.default_rng(seed)
    if dense:
        base = random_boxes(rng, 300, span=800, size=90)
        boxes = (base[rng.integers(0, 300, n)] + rng.normal(0, 2.0, (n, 4))).astype(np.float32)
        scores = np.round(rng.uniform(0, 1, n), 2).astype(np.float32)
    else:
        boxes = random_boxes(rng, n, span=800.0)
        scores = rng.standard_normal(n).astype(np.float32)
    assert np.array_equal(nms_hip(boxes, scores, thr), R.nms(boxes, scores, thr))


@pytest.mark.parametrize("pooled,scale,C,H,W", [(7, 0.25, 64, 40, 48), (14, 0.125, 32, 25, 25), (7, 1.0 / 32, 128, 7, 9)])
def test_roi_align_matches_oracle(pooled, scale, C, H, W):
    lib = _lib.load()
    rng = np.random.default_rng(pooled * 100 + C)
    feat = rng.standard_normal((C, H, W), dtype=np.float32)
    img_w, img_h = W / scale, H / scale
    xy = rng.uniform(-0.1, 0.8, (60, 2)) * [img_w, img_h]
    wh = rng.uniform(0.01, 0.5, (60, 2)) * [img_w, img_h]
    rois = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    rois[0] = [0, 0, img_w, img_h]                    # whole image
    rois[1] = [img_w - 1, img_h - 1, img_w + 30, img_h + 30]   # hangs over the border
    rois[2] = [10, 10, 10, 10]                        # empty
    ref = R.roi_align_fast(feat, rois, scale, pooled)
    f = dev(np.ascontiguousarray(feat.transpose(1, 2, 0)))
    r = dev(rois)
    out = torch.empty((rois.shape[0], pooled, pooled, C), dtype=torch.float32, device="cuda")
    _lib.check(lib.td_roi_align(f.data_ptr(), H, W, C, r.data_ptr(), rois.shape[0], scale, pooled, out.data_ptr(), 0,
                                _lib.stream_ptr()), "td_roi_align")
    torch.cuda.synchronize()
    got = out.cpu().numpy().transpose(0, 3, 1, 2)
    assert np.allclose(got, ref, rtol=0, atol=2e-6 * max(1.0, np.abs(ref).max()))
    assert (got == ref).mean() > 0.99     # same op order: bit-identical almost everywhere


def test_roi_align_linear_ramp_known_answer():
    """Bilinear sampling of an affine feature map is exact: bin value = f(bin centre)."""
    lib = _lib.load()
    H, W, C = 30, 40, 4
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    feat = np.stack([xs, ys, 2 * xs + 3 * ys, np.ones_like(xs)], axis=0)
    rois = np.array([[8, 8, 64, 48], [20.5, 12.25, 90.75, 70.5]], dtype=np.float32)
    scale, pooled = 0.25, 7
    out = torch.empty((2, pooled, pooled, C), dtype=torch.float32, device="cuda")
    d_feat, d_rois = dev(np.ascontiguousarray(feat.transpose(1, 2, 0))), dev(rois)
    _lib.check(lib.td_roi_align(d_feat.data_ptr(), H, W, C, d_rois.data_ptr(), 2,
                                scale, pooled, out.data_ptr(), 0, _lib.stream_ptr()), "td_roi_align")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for r in range(2):
        x0, y0 = rois[r, 0] * scale - 0.5, rois[r, 1] * scale - 0.5
        bw, bh = (rois[r, 2] - rois[r, 0]) * scale / pooled, (rois[r, 3] - rois[r, 1]) * scale / pooled
        cx = x0 + (np.arange(pooled) + 0.5) * bw
        cy = y0 + (np.arange(pooled) + 0.5) * bh
        assert np.allclose(got[r, :, :, 0], np.broadcast_to(cx[None, :], (pooled, pooled)), atol=1e-4)
        assert np.allclose(got[r, :, :, 1], np.broadcast_to(cy[:, None], (pooled, pooled)), atol=1e-4)
        assert np.allclose(got[r, :, :, 3], 1.0, atol=1e-6)


def paste_hip(probs, boxes, h, w, thr=0.5):
    from treedetection_amd.engine import unpack_masks
    lib = _lib.load()
    n = probs.shape[0]
    words = n * ((w + 2 + 31) // 32) * h
    region = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    offset = torch.zeros((n,), dtype=torch.int64, device="cuda")
    bits = torch.zeros((words,), dtype=torch.int32, device="cuda")
    d_probs, d_boxes = dev(probs), dev(boxes)      # keep the device buffers alive across the call
    _lib.check(lib.td_paste_masks(d_probs.data_ptr(), d_boxes.data_ptr(), n, h, w, thr, region.data_ptr(),
                                  offset.data_ptr(), bits.data_ptr(), words, _lib.stream_ptr()), "td_paste_masks")
    torch.cuda.synchronize()
    return unpack_masks(region.cpu().numpy(), offset.cpu().numpy(), bits.cpu().numpy(), n, h, w), region.cpu().numpy()


def test_paste_masks_match_oracle():
    rng = np.random.default_rng(3)
    n, h, w = 24, 300, 420
    yy, xx = np.meshgrid(np.linspace(-1, 1, 28), np.linspace(-1, 1, 28), indexing="ij")
    probs = np.stack([1 / (1 + np.exp(-(rng.uniform(2, 8) * (rng.uniform(0.3, 1.0) - np.sqrt(xx ** 2 + yy ** 2))
                                        + rng.normal(0, 0.7, (28, 28))))) for _ in range(n)]).astype(np.float32)
    xy = rng.uniform(-5, 1, (n, 2)) * [-w / 1.2, -h / 1.2] * (rng.uniform(0, 1, (n, 2)) > 0.1)
    wh = rng.uniform(3, 150, (n, 2))
    boxes = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    boxes = R.clip_boxes(boxes, h, w)
    boxes[0] = [0, 0, w, h]
    boxes[1] = [w - 2.5, h - 3.25, w, h]
    boxes = boxes[(boxes[:, 2] > boxes[:, 0]) & (boxes[:, 3] > boxes[:, 1])]
    probs = probs[: boxes.shape[0]]
    got, region = paste_hip(probs, boxes, h, w)
    ref = R.paste_masks(probs, boxes, h, w, 0.5)
    for i in range(boxes.shape[0]):
        assert tuple(region[i]) == R.paste_region(boxes[i], h, w)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"{(got != ref).sum()} pixels differ"


def test_paste_constant_mask_known_answer():
    # a constant mask of 1.0 pastes to 1 inside the box (pixels whose centre is >= half a mask-pixel inside) and 0 outside
    probs = np.ones((1, 28, 28), dtype=np.float32)
    boxes = np.array([[10, 20, 66, 76]], dtype=np.float32)      # 56 px = 2 px per mask cell
    got, _ = paste_hip(probs, boxes, 100, 100)
    assert got[0, 20:76, 10:66].all()
    assert not got[0, :19].any() and not got[0, 77:].any() and not got[0, :, :9].any() and not got[0, :, 67:].any()


@pytest.mark.parametrize("h,w,c", [(100, 100, 3), (450, 450, 4), (350, 450, 4), (1000, 1000, 3), (97, 211, 5)])
def test_resize_tile_pillow_exact(h, w, c):
    lib = _lib.load()
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    oh, ow = R.resize_shortest_edge_shape(h, w)
    oh2, ow2 = (torch.zeros(1, dtype=torch.int32) for _ in range(2))
    import ctypes as C
    a, b = C.c_int(), C.c_int()
    lib.td_resize_shape(h, w, 800, 1333, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (oh, ow)
    ref = R.pil_resize_bilinear_u8(np.dstack((img[:, :, 2], img[:, :, 1], img[:, :, 0])), oh, ow)
    src = dev(img)
    pitch = (ow + 31) // 32 * 32
    dst = torch.zeros((oh, pitch, 3), dtype=torch.uint8, device="cuda")
    tmp = torch.empty((h * ow * 3,), dtype=torch.uint8, device="cuda")
    _lib.check(lib.td_resize_tile_u8(src.data_ptr(), h, w, c, dst.data_ptr(), oh, ow, pitch, tmp.data_ptr(),
                                     _lib.stream_ptr()), "td_resize_tile_u8")
    torch.cuda.synchronize()
    got = dst.cpu().numpy()[:, :ow, :]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("h,w,oh,ow", [(450, 450, 800, 800), (1000, 1000, 800, 800), (350, 450, 800, 1029), (97, 211, 613, 1333), (800, 800, 800, 800)])
def test_resize_bilinear_f64_matches_torch_interpolate(h, w, oh, ow):
    """td_resize_bilinear_f64 (the float branch of the reference's resize, prediction.py:167-169) against what the reference
    itself calls for non-uint8 tiles: torch.nn.functional.interpolate(bilinear, align_corners=False) on the CPU in float64,
    then .astype(float32). Same operations in the same order in float64; the only freedom is a fused multiply-add inside
    torch's vectorised loop, i.e. one float64 ulp before the final rounding: equal to 1 float32 ulp at most, bit-equal
    almost everywhere. Up- and down-scaling, 16-bit value range, padded destination."""
    import torch.nn.functional as F
    lib = _lib.load()
    rng = np.random.default_rng(h * 13 + w)
    src = (255.0 * rng.integers(0, 65536, (3, h, w)).astype(np.float64) / 65535.0)          # prediction.py:167
    ref = F.interpolate(torch.from_numpy(src)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].float().numpy()
    pitch, rows = (ow + 31) // 32 * 32, (oh + 31) // 32 * 32
    dst = torch.full((3, rows, pitch), -1.0, dtype=torch.float32, device="cuda")
    d = dev(src)
    _lib.check(lib.td_resize_bilinear_f64(d.data_ptr(), 3, h, w, dst.data_ptr(), oh, ow, pitch, rows * pitch, _lib.stream_ptr()),
               "td_resize_bilinear_f64")
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    got = out[:, :oh, :ow]
    assert (out[:, oh:, :] == -1.0).all() and (out[:, :, ow:] == -1.0).all()        # nothing written outside the image
    ulp = np.spacing(np.abs(ref).astype(np.float32))
    assert (np.abs(got - ref) <= ulp).all()
    assert (got == ref).mean() >= 0.9999


def test_paste_masks_batch_equals_per_image_paste():
    """td_paste_masks_batch (rank 0's paste of gathered detections: ragged image sizes, device-side counts, one
    asynchronous launch pair) against td_paste_masks image by image."""
    from treedetection_amd.engine import unpack_masks
    lib = _lib.load()
    rng = np.random.default_rng(11)
    B, Dn = 3, 16
    sizes = [(120, 200), (90, 64), (33, 150)]
    counts = np.array([16, 5, 0], np.int32)
    probs = rng.uniform(0, 1, (B, Dn, 28, 28)).astype(np.float32)
    boxes = np.zeros((B, Dn, 4), np.float32)
    for b, (h, w) in enumerate(sizes):
        xy = rng.uniform(-10, 0.8 * w, (Dn, 2))
        wh = rng.uniform(2, 80, (Dn, 2))
        boxes[b] = R.clip_boxes(np.concatenate([xy, xy + wh], axis=1).astype(np.float32), h, w)
    mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
    words = Dn * ((mw + 2 + 31) // 32) * mh
    region = torch.zeros((B, Dn, 4), dtype=torch.int32, device="cuda")
    offset = torch.zeros((B, Dn), dtype=torch.int64, device="cuda")
    bits = torch.zeros((B, words), dtype=torch.int32, device="cuda")
    d_probs, d_boxes, d_counts = dev(probs), dev(boxes), torch.from_numpy(counts).cuda()
    import ctypes as C
    hw = (C.c_int32 * (2 * B))(*[v for s in sizes for v in s])
    _lib.check(lib.td_paste_masks_batch(d_probs.data_ptr(), d_boxes.data_ptr(), d_counts.data_ptr(), hw, B, Dn, 0.5,
                                        region.data_ptr(), offset.data_ptr(), bits.data_ptr(), words, _lib.stream_ptr()),
               "td_paste_masks_batch")
    torch.cuda.synchronize()
    for b, (h, w) in enumerate(sizes):
        n = int(counts[b])
        got = unpack_masks(region[b].cpu().numpy(), offset[b].cpu().numpy(), bits[b].cpu().numpy(), n, h, w)
        if n == 0:
            assert got.shape == (0, h, w)
            continue
        keep = (boxes[b, :n, 2] > boxes[b, :n, 0]) & (boxes[b, :n, 3] > boxes[b, :n, 1])
        want, _ = paste_hip(probs[b, :n][keep], boxes[b, :n][keep], h, w)
        assert np.array_equal(got[keep], want)
    hw_bad = (C.c_int32 * (2 * B))(*[v for s in [(4000, 4000)] * B for v in s])
    with pytest.raises(_lib.TdError, match="mask words"):
        _lib.check(lib.td_paste_masks_batch(d_probs.data_ptr(), d_boxes.data_ptr(), d_counts.data_ptr(), hw_bad, B, Dn, 0.5,
                                            region.data_ptr(), offset.data_ptr(), bits.data_ptr(), words, _lib.stream_ptr()), "x")
