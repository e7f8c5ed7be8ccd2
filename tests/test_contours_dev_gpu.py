"""td_trace_contours_dev (border following on the GPU) against the ORACLE's restatement of cv2.findContours
(oracle/contours_ref.py — RETR_TREE order, CHAIN_APPROX_SIMPLE points), detection by detection; the product's host
tracer (td_find_contours) is checked against the same oracle on the same masks, so a restatement error shared by the
two product tracers cannot pass."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.contours_ref import find_contours  # noqa: E402  (the checker)
from treedetection_amd import _lib  # noqa: E402
from treedetection_amd.contours import find_contours as host_find_contours  # noqa: E402  (product, host tracer)

pytestmark = pytest.mark.gpu
CMAX = 256


def _pack(masks_per_image, Dn):
    """list (per image) of list of (x0, y0, bool mask) → region / offset / bits / counts arrays like td_detections."""
    B = len(masks_per_image)
    region = np.zeros((B, Dn, 4), np.int32)
    offset = np.zeros((B, Dn), np.int64)
    words = []
    for b, dets in enumerate(masks_per_image):
        cur, off = [], 0
        for d, (x0, y0, m) in enumerate(dets):
            h, w = m.shape
            wpr = (w + 31) // 32
            pad = np.zeros((h, wpr * 32), np.uint8)
            pad[:, :w] = m
            rows = np.packbits(pad, axis=1, bitorder="little").view(np.uint32).ravel()
            region[b, d] = (x0, y0, x0 + w, y0 + h)
            offset[b, d] = off
            off += rows.size
            cur.append(rows)
        words.append(np.concatenate(cur) if cur else np.zeros(1, np.uint32))
    wpi = max(len(x) for x in words) + 8
    bits = np.zeros((B, wpi), np.uint32)
    for b, x in enumerate(words):
        bits[b, : len(x)] = x
    counts = np.array([len(d) for d in masks_per_image], np.int32)
    return region, offset, bits, counts


def _trace(masks_per_image, Dn=12, pts_cap=60000):
    region, offset, bits, counts = _pack(masks_per_image, Dn)
    B = len(masks_per_image)
    dev = lambda a: torch.from_numpy(a).cuda()      # noqa: E731
    d_region, d_offset, d_bits, d_counts = dev(region), dev(offset), dev(bits.view(np.int32)), dev(counts)
    pts = torch.zeros((B, pts_cap, 2), dtype=torch.int16, device="cuda")
    img_pts = torch.zeros((B,), dtype=torch.int32, device="cuda")
    det_info = torch.zeros((B, Dn, 4), dtype=torch.int32, device="cuda")
    cont_info = torch.zeros((B, Dn, CMAX, 2), dtype=torch.int32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.td_trace_contours_dev(d_region.data_ptr(), d_offset.data_ptr(), d_bits.data_ptr(), bits.shape[1], d_counts.data_ptr(),
                                         B, Dn, pts.data_ptr(), pts_cap, img_pts.data_ptr(), det_info.data_ptr(), cont_info.data_ptr(),
                                         _lib.stream_ptr()), "td_trace_contours_dev")
    torch.cuda.synchronize()
    return pts.cpu().numpy(), img_pts.cpu().numpy(), det_info.cpu().numpy(), cont_info.cpu().numpy()


def _check(masks_per_image, **kw):
    pts, img_pts, det_info, cont_info = _trace(masks_per_image, **kw)
    traced = 0
    for b, dets in enumerate(masks_per_image):
        for d, (x0, y0, m) in enumerate(dets):
            status, nc, base, total = det_info[b, d]
            want = find_contours(m.astype(np.uint8))
            host = host_find_contours(m.astype(np.uint8))
            assert len(host) == len(want) and all(np.array_equal(a, b) for a, b in zip(host, want)), (b, d)
            if status != 0:
                continue
            traced += 1
            assert nc == len(want) and total == sum(len(c) for c in want), (b, d, nc, len(want))
            for k, c in enumerate(want):
                o, n = cont_info[b, d, k]
                got = pts[b, base + o: base + o + n].astype(np.int32)
                assert n == len(c) and np.array_equal(got, c + [x0, y0]), (b, d, k)
        assert img_pts[b] == sum(det_info[b, d, 3] for d in range(len(dets)) if det_info[b, d, 0] == 0) or (det_info[b, :, 0] == 4).any()
    return det_info, traced


def _blob(rng, h, w, holes=True, noise=0.0):
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(int(rng.integers(1, 4))):
        cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
        a, b = rng.uniform(0.15, 0.45) * w, rng.uniform(0.15, 0.45) * h
        m |= ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 < 1
    if holes:
        for _ in range(int(rng.integers(0, 3))):
            cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
            m &= ~(((xx - cx) / (0.08 * w + 1)) ** 2 + ((yy - cy) / (0.08 * h + 1)) ** 2 < 1)
    if noise:
        m ^= rng.random((h, w)) < noise
    return m


def test_known_small_cases():
    cases = [np.ones((1, 1), bool), np.ones((1, 5), bool), np.ones((4, 1), bool), np.ones((3, 3), bool), np.zeros((4, 4), bool),
             np.array([[1, 0, 1], [0, 1, 0], [1, 0, 1]], bool), np.array([[1, 1, 1], [1, 0, 1], [1, 1, 1]], bool),
             np.array([[1, 1, 1, 1, 1], [1, 0, 0, 0, 1], [1, 0, 1, 0, 1], [1, 0, 0, 0, 1], [1, 1, 1, 1, 1]], bool),
             np.array([[0, 1, 1, 0], [1, 1, 0, 1], [0, 1, 1, 1]], bool)]
    det_info, traced = _check([[(3 * i, 2 * i, m) for i, m in enumerate(cases)]], Dn=len(cases))
    assert traced == len(cases) and det_info[0, 4, 1] == 0            # the empty mask has no contour


@pytest.mark.parametrize("seed", range(4))
def test_random_crowns_with_holes_and_noise(seed):
    rng = np.random.default_rng(seed)
    images = []
    for _ in range(3):
        dets = []
        for _ in range(int(rng.integers(4, 12))):
            h, w = int(rng.integers(3, 150)), int(rng.integers(3, 150))
            dets.append((int(rng.integers(0, 800)), int(rng.integers(0, 800)), _blob(rng, h, w, noise=float(rng.choice([0, 0, 0.01])))))
        images.append(dets)
    det_info, traced = _check(images)
    assert traced >= sum(len(d) for d in images) - 6            # only the noisiest masks exceed TD_CONTOUR_MAX contours
    assert set(np.unique(det_info[:, :, 0])) <= {0, 2}


def test_limits_are_flagged_not_mis_traced():
    rng = np.random.default_rng(9)
    big = _blob(rng, 200, 200)                                        # 202 x 202 labels do not fit the on-chip image
    many = np.zeros((40, 120), bool)
    many[::2, ::2] = True                                             # 1200 single-pixel contours
    ok = _blob(rng, 60, 60)
    det_info, traced = _check([[(0, 0, big), (5, 5, many), (9, 9, ok)]], Dn=3)
    assert det_info[0, :, 0].tolist() == [1, 2, 0] and traced == 1
    det_info, _ = _check([[(0, 0, ok), (0, 0, ok)]], Dn=2, pts_cap=int(sum(len(c) for c in find_contours(ok.astype(np.uint8))) + 3))
    assert sorted(det_info[0, :, 0].tolist()) == [0, 4]               # the second block does not fit the point buffer


def test_json_from_device_contours_is_bytewise_the_host_path():
    """td_tile_polygons_json_dev on the traced points == td_tile_polygons_json on the packed rows, including tiles where
    some detections were left to the host tracer (then the mask rows must be supplied)."""
    from treedetection_amd.contours import tile_polygons_json, tile_polygons_json_dev
    rng = np.random.default_rng(3)
    many = np.zeros((30, 90), bool)
    many[::2, ::2] = True
    images = [[(int(rng.integers(0, 300)), int(rng.integers(0, 300)), _blob(rng, int(rng.integers(5, 120)), int(rng.integers(5, 120)))) for _ in range(9)],
              [(10, 20, _blob(rng, 50, 70)), (40, 40, many), (0, 0, _blob(rng, 180, 190))]]
    Dn = 9
    region, offset, bits, counts = _pack(images, Dn)
    pts, img_pts, det_info, cont_info = _trace(images, Dn=Dn)
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5319000.0)
    for b, dets in enumerate(images):
        n = len(dets)
        scores = rng.random(n).astype(np.float32)
        classes = np.zeros(n, np.int32)
        want = tile_polygons_json(region[b], offset[b], bits[b].view(np.int32), scores, classes, t, "img.tif")
        flagged = (det_info[b, :n, 0] != 0).any()
        got = tile_polygons_json_dev(pts[b], det_info[b], cont_info[b], region[b], offset[b], None, scores, classes, t, "img.tif")
        assert (got is None) == bool(flagged)
        got = tile_polygons_json_dev(pts[b], det_info[b], cont_info[b], region[b], offset[b], bits[b].view(np.int32), scores, classes, t, "img.tif")
        assert got == want and len(want) > 2
    assert (det_info[1, :3, 0] != 0).sum() == 2 and (det_info[0, :, 0] == 0).all()


def test_prediction_file_from_device_rows_is_bytewise_the_host_path(tmp_path):
    """td_tile_prediction_file (fetches only the words the records cover from the DEVICE buffer, traces, formats, writes the
    file) == td_tile_polygons_json on a host copy of the whole buffer; the words past the last region never leave the GPU
    (the pinned buffer keeps its sentinel there); an empty tile writes "[]"; records that overrun the buffer are refused."""
    from treedetection_amd.contours import tile_polygons_json, tile_prediction_file
    rng = np.random.default_rng(5)
    images = [[(int(rng.integers(0, 300)), int(rng.integers(0, 300)), _blob(rng, int(rng.integers(5, 120)), int(rng.integers(5, 120)))) for _ in range(9)],
              [(10, 20, _blob(rng, 50, 70)), (0, 0, _blob(rng, 180, 190))], []]
    Dn = 9
    region, offset, bits, counts = _pack(images, Dn)
    d_bits = torch.from_numpy(bits.view(np.int32)).cuda()
    pinned = torch.full(bits.shape, -1, dtype=torch.int32).pin_memory()
    host = pinned.numpy()
    t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5319000.0)
    stride = bits.shape[1] * 4
    for b, dets in enumerate(images):
        n = len(dets)
        scores = rng.random(n).astype(np.float32)
        classes = np.zeros(n, np.int32)
        want = tile_polygons_json(region[b], offset[b], bits[b].view(np.int32), scores, classes, t, "img.tif")
        path = str(tmp_path / f"Prediction_{b}.json")
        wrote = tile_prediction_file(0, region[b], offset[b], d_bits.data_ptr() + b * stride, host[b], scores, classes, t, "img.tif", path)
        got = open(path, "rb").read()
        assert got == want and wrote == len(want)
        used = 0 if n == 0 else int(offset[b, n - 1]) + ((region[b, n - 1, 2] - region[b, n - 1, 0] + 31) // 32) * int(region[b, n - 1, 3] - region[b, n - 1, 1])
        assert (host[b, used:] == -1).all() and (n == 0 or (host[b, :used].view(np.uint32) == bits[b, :used]).all())
    assert open(str(tmp_path / "Prediction_2.json"), "rb").read() == b"[]"
    bad = offset[1].copy()
    bad[1] = bits.shape[1]            # last region would start at the end of the buffer
    with pytest.raises(_lib.TdError):
        tile_prediction_file(0, region[1], bad, d_bits.data_ptr() + stride, host[1], np.ones(2, np.float32), np.zeros(2, np.int32), t,
                             "img.tif", str(tmp_path / "bad.json"))
