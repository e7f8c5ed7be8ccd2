"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the detection ops the reference's hot path invokes through detectron2 /
torchvision (absent from /root/reference and from this image: SURVEY.md §8c — "parity
unpinned": the reference holds no golden vectors for this path). Each function cites the
reference call site it stands behind and the SURVEY Appendix-A item that fixes its semantics.

All arithmetic is float32, one IEEE operation per step (numpy / torch elementwise ops do
not fuse multiply-add), so the HIP kernels can be compared bit-for-bit where they use the
same operation order.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np

F32 = np.float32
SCALE_CLAMP = F32(math.log(1000.0 / 16.0))


# ----------------------------------------------------------------------------------------------
# NMS  (reference: detectron2 batched_nms reached from prediction.py:183; Appendix A item 8)
# ----------------------------------------------------------------------------------------------
def stable_desc_order(scores: np.ndarray) -> np.ndarray:
    """Indices that sort ``scores`` descending; ties keep the lower index first."""
    return np.argsort(-scores.astype(np.float64), kind="stable")


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """Greedy NMS; returns kept indices in descending-score order (ties: lower index first).

    IoU = inter / (area_i + area_j - inter), suppress when IoU > thr (strict).
    """
    boxes = np.ascontiguousarray(boxes, dtype=F32).reshape(-1, 4)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = stable_desc_order(np.asarray(scores, dtype=F32))
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    thr32 = F32(thr)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(order[i])
        if i + 1 == n:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(F32(0), xx2 - xx1)
        h = np.maximum(F32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr32
    return np.asarray(keep, dtype=np.int64)


def batched_nms(boxes: np.ndarray, scores: np.ndarray, idxs: np.ndarray, thr: float) -> np.ndarray:
    """Per-category NMS on un-offset boxes; result sorted by descending score (stable)."""
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32)
    idxs = np.asarray(idxs)
    keep_mask = np.zeros(scores.shape[0], dtype=bool)
    for c in np.unique(idxs):
        sel = np.nonzero(idxs == c)[0]
        k = nms(boxes[sel], scores[sel], thr)
        keep_mask[sel[k]] = True
    kept = np.nonzero(keep_mask)[0]
    return kept[stable_desc_order(scores[kept])]


# ----------------------------------------------------------------------------------------------
# anchors / box decoding  (Appendix A items 6, 7, 11)
# ----------------------------------------------------------------------------------------------
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
FPN_STRIDES = (4, 8, 16, 32, 64)


def cell_anchors(size: float, ratios: Sequence[float] = ANCHOR_RATIOS) -> np.ndarray:
    out = []
    area = float(size) ** 2
    for r in ratios:
        w = math.sqrt(area / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return np.asarray(out, dtype=F32)


def grid_anchors(h: int, w: int, stride: int, size: float) -> np.ndarray:
    """[h*w*A, 4] anchors in (y, x, a) order, offset 0."""
    base = cell_anchors(size)
    sx = np.arange(0, w * stride, stride, dtype=F32)
    sy = np.arange(0, h * stride, stride, dtype=F32)
    yy, xx = np.meshgrid(sy, sx, indexing="ij")
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], axis=1)
    return (shifts[:, None, :] + base[None, :, :]).reshape(-1, 4).astype(F32)


def apply_deltas(deltas: np.ndarray, boxes: np.ndarray, weights=(1.0, 1.0, 1.0, 1.0)) -> np.ndarray:
    deltas = np.asarray(deltas, dtype=F32).reshape(-1, 4)
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + F32(0.5) * widths
    ctr_y = boxes[:, 1] + F32(0.5) * heights
    wx, wy, ww, wh = (F32(v) for v in weights)
    dx = deltas[:, 0] / wx
    dy = deltas[:, 1] / wy
    dw = np.minimum(deltas[:, 2] / ww, SCALE_CLAMP)
    dh = np.minimum(deltas[:, 3] / wh, SCALE_CLAMP)
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = np.exp(dw) * widths
    ph = np.exp(dh) * heights
    out = np.stack([pcx - F32(0.5) * pw, pcy - F32(0.5) * ph, pcx + F32(0.5) * pw, pcy + F32(0.5) * ph], axis=1)
    return out.astype(F32)


def clip_boxes(boxes: np.ndarray, h: float, w: float) -> np.ndarray:
    b = np.array(boxes, dtype=F32, copy=True).reshape(-1, 4)
    b[:, 0] = np.clip(b[:, 0], F32(0), F32(w))
    b[:, 1] = np.clip(b[:, 1], F32(0), F32(h))
    b[:, 2] = np.clip(b[:, 2], F32(0), F32(w))
    b[:, 3] = np.clip(b[:, 3], F32(0), F32(h))
    return b


def level_assign(boxes: np.ndarray, min_level=2, max_level=5, canon_size=224, canon_level=4) -> np.ndarray:
    """FPN level (0-based from min_level) of each box — Appendix A item 9."""
    b = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.sqrt(area)
        lv = np.floor(F32(canon_level) + np.log2(s / F32(canon_size) + F32(1e-8)))
    lv = np.clip(lv, min_level, max_level)
    return (lv.astype(np.int64) - min_level)


# ----------------------------------------------------------------------------------------------
# RoIAlign (aligned=True, sampling_ratio=0)  — Appendix A item 9
# ----------------------------------------------------------------------------------------------
def roi_align(feat: np.ndarray, rois: np.ndarray, spatial_scale: float, pooled: int) -> np.ndarray:
    """feat [C,H,W] float32, rois [R,4] (x1,y1,x2,y2) in image units → [R,C,pooled,pooled].

    Same operation order as the torchvision CPU kernel: per bin, sum over the adaptive sample
    grid of (w1*v1 + w2*v2 + w3*v3 + w4*v4), then divide by the sample count.
    """
    feat = np.asarray(feat, dtype=F32)
    C, H, W = feat.shape
    rois = np.asarray(rois, dtype=F32).reshape(-1, 4)
    R = rois.shape[0]
    out = np.zeros((R, C, pooled, pooled), dtype=F32)
    sc = F32(spatial_scale)
    half = F32(0.5)
    for r in range(R):
        sw = rois[r, 0] * sc - half
        sh = rois[r, 1] * sc - half
        ew = rois[r, 2] * sc - half
        eh = rois[r, 3] * sc - half
        rw = ew - sw
        rh = eh - sh
        bh = rh / F32(pooled)
        bw = rw / F32(pooled)
        gh = int(math.ceil(float(rh / F32(pooled))))
        gw = int(math.ceil(float(rw / F32(pooled))))
        count = F32(max(gh * gw, 1))
        for ph in range(pooled):
            for pw in range(pooled):
                acc = np.zeros(C, dtype=F32)
                for iy in range(gh):
                    y = sh + F32(ph) * bh + (F32(iy) + half) * bh / F32(gh)
                    for ix in range(gw):
                        x = sw + F32(pw) * bw + (F32(ix) + half) * bw / F32(gw)
                        if y < F32(-1.0) or y > F32(H) or x < F32(-1.0) or x > F32(W):
                            continue
                        yy = max(y, F32(0))
                        xx = max(x, F32(0))
                        yl = int(yy)
                        xl = int(xx)
                        if yl >= H - 1:
                            yh = yl = H - 1
                            yy = F32(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xh = xl = W - 1
                            xx = F32(xl)
                        else:
                            xh = xl + 1
                        ly = F32(yy - F32(yl))
                        lx = F32(xx - F32(xl))
                        hy = F32(1) - ly
                        hx = F32(1) - lx
                        w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
                        acc = acc + (w1 * feat[:, yl, xl] + w2 * feat[:, yl, xh]
                                     + w3 * feat[:, yh, xl] + w4 * feat[:, yh, xh])
                out[r, :, ph, pw] = acc / count
    return out


def roi_align_fast(feat, rois, spatial_scale: float, pooled: int):
    """Vectorised (torch) form of :func:`roi_align`: RoIs that share a sample grid (gh, gw) are evaluated
    together; per element the operation order is identical to the scalar version."""
    import torch

    feat = torch.as_tensor(feat, dtype=torch.float32)
    C, H, W = feat.shape
    rois = torch.as_tensor(np.asarray(rois, dtype=F32)).reshape(-1, 4)
    Rn = rois.shape[0]
    out = torch.zeros((Rn, C, pooled, pooled), dtype=torch.float32)
    if Rn == 0:
        return out.numpy()
    sc = torch.tensor(spatial_scale, dtype=torch.float32)
    sw = rois[:, 0] * sc - 0.5
    sh = rois[:, 1] * sc - 0.5
    ew = rois[:, 2] * sc - 0.5
    eh = rois[:, 3] * sc - 0.5
    rw = ew - sw
    rh = eh - sh
    bh = rh / pooled
    bw = rw / pooled
    gh = torch.clamp(torch.ceil(rh / pooled), min=0).to(torch.int64)
    gw = torch.clamp(torch.ceil(rw / pooled), min=0).to(torch.int64)
    flat = feat.reshape(C, H * W)
    p = torch.arange(pooled, dtype=torch.float32)
    keys = gh * 100000 + gw
    for key in torch.unique(keys).tolist():
        sel = torch.nonzero(keys == key).flatten()
        ghr, gwr = int(key // 100000), int(key % 100000)
        count = float(max(ghr * gwr, 1))
        G = sel.numel()
        acc = torch.zeros((C, G, pooled, pooled), dtype=torch.float32)
        s_sh, s_sw, s_bh, s_bw = sh[sel, None], sw[sel, None], bh[sel, None], bw[sel, None]
        for iy in range(ghr):
            y = s_sh + p[None, :] * s_bh + (torch.tensor(float(iy)) + 0.5) * s_bh / float(ghr)      # [G,P]
            for ix in range(gwr):
                x = s_sw + p[None, :] * s_bw + (torch.tensor(float(ix)) + 0.5) * s_bw / float(gwr)  # [G,P]
                Y = y[:, :, None].expand(G, pooled, pooled)
                X = x[:, None, :].expand(G, pooled, pooled)
                oob = (Y < -1.0) | (Y > H) | (X < -1.0) | (X > W)
                yy = torch.clamp(Y, min=0.0)
                xx = torch.clamp(X, min=0.0)
                yl = yy.to(torch.int64)
                xl = xx.to(torch.int64)
                ytop = yl >= H - 1
                xtop = xl >= W - 1
                yl = torch.where(ytop, torch.full_like(yl, H - 1), yl)
                xl = torch.where(xtop, torch.full_like(xl, W - 1), xl)
                yh = torch.where(ytop, yl, yl + 1)
                xh = torch.where(xtop, xl, xl + 1)
                yy = torch.where(ytop, yl.to(torch.float32), yy)
                xx = torch.where(xtop, xl.to(torch.float32), xx)
                ly = yy - yl.to(torch.float32)
                lx = xx - xl.to(torch.float32)
                hy = 1.0 - ly
                hx = 1.0 - lx
                w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
                yl = torch.where(oob, torch.zeros_like(yl), yl)
                yh = torch.where(oob, torch.zeros_like(yh), yh)
                xl = torch.where(oob, torch.zeros_like(xl), xl)
                xh = torch.where(oob, torch.zeros_like(xh), xh)
                shp = (C, G, pooled, pooled)
                v1 = flat[:, (yl * W + xl).reshape(-1)].reshape(shp)
                v2 = flat[:, (yl * W + xh).reshape(-1)].reshape(shp)
                v3 = flat[:, (yh * W + xl).reshape(-1)].reshape(shp)
                v4 = flat[:, (yh * W + xh).reshape(-1)].reshape(shp)
                val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
                val = torch.where(oob[None], torch.zeros_like(val), val)
                acc = acc + val
        out[sel] = (acc / count).permute(1, 0, 2, 3)
    return out.numpy()


# ----------------------------------------------------------------------------------------------
# paste_masks_in_image (CPU path: per mask, box region ±1 px) — Appendix A item 13
# ----------------------------------------------------------------------------------------------
def paste_region(box: np.ndarray, img_h: int, img_w: int) -> Tuple[int, int, int, int]:
    """Integer region (x0, y0, x1, y1) the CPU path evaluates for one box."""
    b = np.asarray(box, dtype=F32)
    x0 = int(max(np.floor(b[0]) - 1, 0))
    y0 = int(max(np.floor(b[1]) - 1, 0))
    x1 = int(min(np.ceil(b[2]) + 1, img_w))
    y1 = int(min(np.ceil(b[3]) + 1, img_h))
    return x0, y0, x1, y1


def paste_mask_values(mask: np.ndarray, box: np.ndarray, img_h: int, img_w: int,
                      region: Tuple[int, int, int, int] | None = None) -> Tuple[np.ndarray, Tuple[int, int, int, int]]:
    """Bilinear (grid_sample, align_corners=False, zero padding) resample of one M×M mask."""
    m = np.asarray(mask, dtype=F32)
    M = m.shape[0]
    b = np.asarray(box, dtype=F32)
    if region is None:
        region = paste_region(b, img_h, img_w)
    x0i, y0i, x1i, y1i = region
    ys = np.arange(y0i, y1i, dtype=F32) + F32(0.5)
    xs = np.arange(x0i, x1i, dtype=F32) + F32(0.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        gy = (ys - b[1]) / (b[3] - b[1]) * F32(2) - F32(1)
        gx = (xs - b[0]) / (b[2] - b[0]) * F32(2) - F32(1)
    iy = ((gy + F32(1)) * F32(M) - F32(1)) / F32(2)
    ix = ((gx + F32(1)) * F32(M) - F32(1)) / F32(2)
    out = np.zeros((ys.shape[0], xs.shape[0]), dtype=F32)
    if out.size == 0:
        return out, region
    iy0 = np.floor(iy)
    ix0 = np.floor(ix)
    # torch grid_sampler: west/north weight = (floor + 1) - coord, east/south weight = coord - floor
    wy1 = (iy - iy0).astype(F32)
    wx1 = (ix - ix0).astype(F32)
    wy0 = ((iy0 + F32(1)) - iy).astype(F32)
    wx0 = ((ix0 + F32(1)) - ix).astype(F32)
    pad = np.zeros((M + 2, M + 2), dtype=F32)
    pad[1:-1, 1:-1] = m

    def idx(v):
        # map floor index to the zero-padded array; anything outside [-1, M] reads 0
        vi = np.where(np.isfinite(v), v, -5).astype(np.int64)
        return np.clip(vi + 1, 0, M + 1), (vi >= -1) & (vi <= M)

    yi0, yok0 = idx(iy0)
    yi1, yok1 = idx(iy0 + 1)
    xi0, xok0 = idx(ix0)
    xi1, xok1 = idx(ix0 + 1)
    # torch grid_sample accumulates nw, ne, sw, se in this order
    nw = pad[yi0[:, None], xi0[None, :]] * (yok0[:, None] & xok0[None, :])
    ne = pad[yi0[:, None], xi1[None, :]] * (yok0[:, None] & xok1[None, :])
    sw = pad[yi1[:, None], xi0[None, :]] * (yok1[:, None] & xok0[None, :])
    se = pad[yi1[:, None], xi1[None, :]] * (yok1[:, None] & xok1[None, :])
    w_nw = (wy0[:, None] * wx0[None, :]).astype(F32)
    w_ne = (wy0[:, None] * wx1[None, :]).astype(F32)
    w_sw = (wy1[:, None] * wx0[None, :]).astype(F32)
    w_se = (wy1[:, None] * wx1[None, :]).astype(F32)
    out = nw * w_nw
    out = out + ne * w_ne
    out = out + sw * w_sw
    out = out + se * w_se
    out = np.where(np.isfinite(out), out, F32(0)).astype(F32)
    return out, region


def paste_masks(masks: np.ndarray, boxes: np.ndarray, img_h: int, img_w: int, thr: float = 0.5) -> np.ndarray:
    """[N,M,M] probabilities + [N,4] boxes → bool [N,img_h,img_w]."""
    masks = np.asarray(masks, dtype=F32)
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    N = masks.shape[0]
    out = np.zeros((N, img_h, img_w), dtype=bool)
    for i in range(N):
        vals, (x0, y0, x1, y1) = paste_mask_values(masks[i], boxes[i], img_h, img_w)
        if vals.size:
            out[i, y0:y1, x0:x1] = vals >= F32(thr)
    return out


# ----------------------------------------------------------------------------------------------
# ResizeShortestEdge + PIL bilinear for uint8 (reference prediction.py:169; Appendix A item 2)
# ----------------------------------------------------------------------------------------------
def resize_shortest_edge_shape(h: int, w: int, short: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    scale = short * 1.0 / min(h, w)
    if h < w:
        newh, neww = short, scale * w
    else:
        newh, neww = scale * h, short
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh *= s
        neww *= s
    return int(newh + 0.5), int(neww + 0.5)


PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow ``precompute_coeffs`` for the BILINEAR (triangle, support 1) filter, 8-bit path.

    Returns (xmin[out], int32 coefficient table [out, ksize], ksize). Restated from Pillow's
    published resampling algorithm (libImaging Resample.c); pinned against Pillow itself in
    tests/test_resize.py because Pillow IS installed here.
    """
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = xmin
        for x in range(ksize):
            v = k[x] * (1 << PIL_PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return bounds, kk, ksize


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H,W,C] → uint8 [out_h,out_w,C]; two fixed-point passes (horizontal, then vertical)."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, C = img.shape
    cur = img
    if out_w != W:
        xmin, kk, ks = pil_bilinear_coeffs(W, out_w)
        acc = np.full((H, out_w, C), 1 << (PIL_PRECISION_BITS - 1), dtype=np.int64)
        for t in range(ks):
            src = np.minimum(xmin + t, W - 1)
            acc += cur[:, src, :].astype(np.int64) * kk[None, :, t, None]
        cur = _clip8(acc)
    if out_h != H:
        ymin, kk, ks = pil_bilinear_coeffs(H, out_h)
        Wc = cur.shape[1]
        acc = np.full((out_h, Wc, C), 1 << (PIL_PRECISION_BITS - 1), dtype=np.int64)
        for t in range(ks):
            src = np.minimum(ymin + t, H - 1)
            acc += cur[src, :, :].astype(np.int64) * kk[:, None, t, None]
        cur = _clip8(acc)
    return cur


def preprocess_tile_u8(bands: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """Reference ``Predictor._process_tile`` (prediction.py:159-176) for an 8-bit tile.

    bands: uint8 [C>=3, h, w] as rasterio returns them → (float32 [3,H',W'] BGR 0..255, h, w).
    """
    b = np.asarray(bands)
    bgr = np.dstack((b[2], b[1], b[0]))
    h, w = bgr.shape[:2]
    nh, nw = resize_shortest_edge_shape(h, w)
    out = pil_resize_bilinear_u8(bgr, nh, nw)
    return out.astype(np.float32).transpose(2, 0, 1), h, w
