"""ORACLE — test infrastructure only (never imported by the product path).

torch-CPU fp32 restatement of the forward the reference runs at ``prediction.py:183``
(``self.model(batch_tensors)`` → detectron2 ``GeneralizedRCNN`` configured by
``config.py:25-66``: Mask R-CNN R50/R101-FPN, 1 class, score>0.3, NMS 0.5). detectron2 0.6 /
torchvision 0.20.1 are third-party, un-vendored and not installed here, so
this file restates their published inference algorithm.
**Parity unpinned**: the reference holds no golden vectors or tests for this path; the
restatement is cross-checked by closed-form known-answer tests (tests/test_oracle_*.py).

Every stage returns its intermediate tensors so the HIP kernels can be checked stage by
stage on identical inputs.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops_ref as R

PIXEL_MEAN = (103.530, 116.280, 123.675)  # BGR; std = 1
SIZE_DIVISIBILITY = 32
BN_EPS = 1e-5


class Cfg:
    """detectron2 defaults + the three overrides of reference config.py:35,59-61."""
    pre_nms_topk = 1000
    post_nms_topk = 1000
    rpn_nms_thresh = 0.7
    score_thresh = 0.3
    nms_thresh = 0.5
    detections_per_image = 100
    mask_thresh = 0.5
    rpn_weights = (1.0, 1.0, 1.0, 1.0)
    box_weights = (10.0, 10.0, 5.0, 5.0)


def _t(a) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a, dtype=np.float32))


class MaskRCNNOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray], cfg: Optional[Cfg] = None):
        self.sd = {k: _t(v) for k, v in state_dict.items()}
        self.cfg = cfg or Cfg()
        self.blocks = []
        for s in (2, 3, 4, 5):
            n = 0
            while f"backbone.bottom_up.res{s}.{n}.conv1.weight" in self.sd:
                n += 1
            self.blocks.append(n)

    # ---- layers -----------------------------------------------------------------------------
    def _conv_bn(self, x, name, stride=1, pad=0, relu=False):
        sd = self.sd
        y = F.conv2d(x, sd[name + ".weight"], None, stride=stride, padding=pad)
        scale = sd[name + ".norm.weight"] * (sd[name + ".norm.running_var"] + BN_EPS).rsqrt()
        bias = sd[name + ".norm.bias"] - sd[name + ".norm.running_mean"] * scale
        y = y * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
        return F.relu(y) if relu else y

    def _conv(self, x, name, stride=1, pad=0, relu=False):
        y = F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], stride=stride, padding=pad)
        return F.relu(y) if relu else y

    # ---- stages -----------------------------------------------------------------------------
    @staticmethod
    def batch_images(images: Sequence[np.ndarray]) -> Tuple[torch.Tensor, List[Tuple[int, int]]]:
        """(x - mean) / 1, zero-pad bottom/right to the batch max rounded up to 32."""
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
        hp = max(s[0] for s in sizes)
        wp = max(s[1] for s in sizes)
        hp = (hp + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
        wp = (wp + SIZE_DIVISIBILITY - 1) // SIZE_DIVISIBILITY * SIZE_DIVISIBILITY
        out = torch.zeros((len(images), 3, hp, wp), dtype=torch.float32)
        mean = torch.tensor(PIXEL_MEAN, dtype=torch.float32).reshape(3, 1, 1)
        for i, im in enumerate(images):
            t = _t(im)
            out[i, :, : t.shape[1], : t.shape[2]] = t - mean
        return out, sizes

    def backbone(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        taps: Dict[str, torch.Tensor] = {}
        x = self._conv_bn(x, "backbone.bottom_up.stem.conv1", stride=2, pad=3, relu=True)
        taps["stem"] = x
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        taps["pool"] = x
        for si, nblk in enumerate(self.blocks):
            for bi in range(nblk):
                p = f"backbone.bottom_up.res{si + 2}.{bi}"
                stride = 2 if (bi == 0 and si > 0) else 1
                if bi == 0:
                    sc = self._conv_bn(x, p + ".shortcut", stride=stride)
                else:
                    sc = x
                y = self._conv_bn(x, p + ".conv1", stride=stride, relu=True)
                y = self._conv_bn(y, p + ".conv2", pad=1, relu=True)
                y = self._conv_bn(y, p + ".conv3")
                x = F.relu(y + sc)
            taps[f"res{si + 2}"] = x
        return taps

    def fpn(self, res: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        prev = self._conv(res["res5"], "backbone.fpn_lateral5")
        out = {"p5": self._conv(prev, "backbone.fpn_output5", pad=1)}
        for lvl in (4, 3, 2):
            td = F.interpolate(prev, scale_factor=2.0, mode="nearest")
            lat = self._conv(res[f"res{lvl}"], f"backbone.fpn_lateral{lvl}")
            prev = lat + td
            out[f"p{lvl}"] = self._conv(prev, f"backbone.fpn_output{lvl}", pad=1)
        out["p6"] = F.max_pool2d(out["p5"], kernel_size=1, stride=2, padding=0)
        return out

    def rpn_head(self, feats: Dict[str, torch.Tensor]):
        """Per level: logits [B, H*W*A] and deltas [B, H*W*A, 4] in (y, x, a) order."""
        logits, deltas = [], []
        pfx = "proposal_generator.rpn_head."
        for lvl in (2, 3, 4, 5, 6):
            t = self._conv(feats[f"p{lvl}"], pfx + "conv", pad=1, relu=True)
            o = self._conv(t, pfx + "objectness_logits")
            d = self._conv(t, pfx + "anchor_deltas")
            B, A, H, W = o.shape
            logits.append(o.permute(0, 2, 3, 1).flatten(1))
            deltas.append(d.view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).flatten(1, -2))
        return logits, deltas

    def rpn_proposals(self, logits, deltas, feat_hw: List[Tuple[int, int]], image_sizes):
        """find_top_rpn_proposals (Appendix A item 7). Returns per image (boxes, logits) plus taps."""
        cfg = self.cfg
        B = logits[0].shape[0]
        results, taps = [], []
        for n in range(B):
            h_img, w_img = image_sizes[n]
            cand_boxes, cand_scores, cand_lvl, per_level = [], [], [], []
            for li, (lg, dl) in enumerate(zip(logits, deltas)):
                s = lg[n].numpy()
                k = min(cfg.pre_nms_topk, s.shape[0])
                order = R.stable_desc_order(s)[:k]
                fh, fw = feat_hw[li]
                anchors = R.grid_anchors(fh, fw, R.FPN_STRIDES[li], R.ANCHOR_SIZES[li])
                boxes = R.apply_deltas(dl[n].numpy()[order], anchors[order], cfg.rpn_weights)
                per_level.append({"topk_idx": order.astype(np.int64), "topk_scores": s[order], "decoded": boxes})
                cand_boxes.append(boxes)
                cand_scores.append(s[order])
                cand_lvl.append(np.full(k, li, dtype=np.int64))
            boxes = np.concatenate(cand_boxes)
            scores = np.concatenate(cand_scores)
            lvl = np.concatenate(cand_lvl)
            valid = np.isfinite(boxes).all(axis=1) & np.isfinite(scores)
            boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
            boxes = R.clip_boxes(boxes, h_img, w_img)
            ne = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
            boxes, scores, lvl = boxes[ne], scores[ne], lvl[ne]
            keep = R.batched_nms(boxes, scores, lvl, cfg.rpn_nms_thresh)[: cfg.post_nms_topk]
            results.append((boxes[keep], scores[keep]))
            taps.append({"per_level": per_level, "cand_boxes": boxes, "cand_scores": scores, "cand_lvl": lvl,
                         "keep": keep})
        return results, taps

    def roi_pool(self, feats: Dict[str, torch.Tensor], boxes_per_image: List[np.ndarray], pooled: int):
        """ROIPooler over p2..p5 (Appendix A item 9) → [sum R, C, pooled, pooled], level per roi."""
        C = feats["p2"].shape[1]
        outs, lvls = [], []
        for n, boxes in enumerate(boxes_per_image):
            lv = R.level_assign(boxes)
            o = np.zeros((boxes.shape[0], C, pooled, pooled), dtype=np.float32)
            for li in range(4):
                sel = np.nonzero(lv == li)[0]
                if sel.size == 0:
                    continue
                f = feats[f"p{li + 2}"][n].numpy()
                o[sel] = R.roi_align_fast(f, boxes[sel], 1.0 / R.FPN_STRIDES[li], pooled)
            outs.append(o)
            lvls.append(lv)
        return outs, lvls

    def box_head(self, pooled: np.ndarray):
        sd = self.sd
        x = _t(pooled).flatten(1)
        x = F.relu(F.linear(x, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
        x = F.relu(F.linear(x, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
        cls = F.linear(x, sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"])
        reg = F.linear(x, sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred.bias"])
        return cls.numpy(), reg.numpy()

    def detections(self, cls_logits: np.ndarray, deltas: np.ndarray, proposals: np.ndarray, image_size):
        """fast_rcnn_inference_single_image (Appendix A item 11), K = 1 foreground class."""
        cfg = self.cfg
        probs = F.softmax(_t(cls_logits), dim=-1).numpy()
        boxes = R.apply_deltas(deltas, proposals, cfg.box_weights)
        valid = np.isfinite(boxes).all(axis=1) & np.isfinite(probs).all(axis=1)
        boxes, probs = boxes[valid], probs[valid]
        scores = probs[:, 0]
        boxes = R.clip_boxes(boxes, image_size[0], image_size[1])
        sel = np.nonzero(scores > np.float32(cfg.score_thresh))[0]
        b, s = boxes[sel], scores[sel]
        keep = R.nms(b, s, cfg.nms_thresh)[: cfg.detections_per_image]
        return b[keep], s[keep], {"all_boxes": boxes, "all_scores": scores, "sel": sel, "keep": keep}

    def mask_head(self, pooled: np.ndarray) -> np.ndarray:
        x = _t(pooled)
        if x.shape[0] == 0:
            return np.zeros((0, 28, 28), dtype=np.float32)
        for i in range(1, 5):
            x = self._conv(x, f"roi_heads.mask_head.mask_fcn{i}", pad=1, relu=True)
        sd = self.sd
        x = F.relu(F.conv_transpose2d(x, sd["roi_heads.mask_head.deconv.weight"], sd["roi_heads.mask_head.deconv.bias"], stride=2))
        x = self._conv(x, "roi_heads.mask_head.predictor")
        return torch.sigmoid(x)[:, 0].numpy()

    @staticmethod
    def postprocess(boxes: np.ndarray, scores: np.ndarray, mask_probs: np.ndarray, image_size, out_hw,
                    mask_thresh: float = 0.5, paste: bool = True):
        """detector_postprocess (Appendix A item 13)."""
        oh, ow = out_hw
        sx = np.float32(ow / image_size[1])
        sy = np.float32(oh / image_size[0])
        b = np.array(boxes, dtype=np.float32, copy=True)
        b[:, 0::2] *= sx
        b[:, 1::2] *= sy
        b = R.clip_boxes(b, oh, ow)
        ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        b, s, m = b[ne], scores[ne], mask_probs[ne]
        masks = R.paste_masks(m, b, oh, ow, mask_thresh) if paste else None
        return b, s, m, masks

    # ---- whole forward ------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, batched_inputs: List[dict], paste: bool = True, return_taps: bool = False):
        """batched_inputs: list of {"image": float32 [3,H',W'] BGR 0..255, "height": h, "width": w}.

        Returns per image {"pred_boxes" [N,4], "scores" [N], "pred_classes" [N] int64,
        "pred_masks" bool [N,h,w], "mask_probs" [N,28,28]} — the ``Instances`` fields the
        reference consumes at prediction.py:217-221.
        """
        x, sizes = self.batch_images([bi["image"] for bi in batched_inputs])
        res = self.backbone(x)
        feats = self.fpn(res)
        logits, deltas = self.rpn_head(feats)
        feat_hw = [tuple(feats[f"p{l}"].shape[-2:]) for l in (2, 3, 4, 5, 6)]
        props, rpn_taps = self.rpn_proposals(logits, deltas, feat_hw, sizes)
        pooled7, lv7 = self.roi_pool(feats, [p[0] for p in props], 7)
        outs, taps_all = [], []
        det_boxes, det_scores, det_taps = [], [], []
        cls_all, reg_all = [], []
        for n in range(len(batched_inputs)):
            cls, reg = self.box_head(pooled7[n])
            cls_all.append(cls)
            reg_all.append(reg)
            b, s, t = self.detections(cls, reg, props[n][0], sizes[n])
            det_boxes.append(b)
            det_scores.append(s)
            det_taps.append(t)
        pooled14, lv14 = self.roi_pool(feats, det_boxes, 14)
        for n, bi in enumerate(batched_inputs):
            probs = self.mask_head(pooled14[n])
            oh = int(bi.get("height", sizes[n][0]))
            ow = int(bi.get("width", sizes[n][1]))
            b, s, m, masks = self.postprocess(det_boxes[n], det_scores[n], probs, sizes[n], (oh, ow),
                                              self.cfg.mask_thresh, paste)
            outs.append({"pred_boxes": b, "scores": s, "pred_classes": np.zeros(len(s), dtype=np.int64),
                         "pred_masks": masks, "mask_probs": m})
        if return_taps:
            taps = {"input": x, "sizes": sizes, "res": res, "feats": feats, "rpn_logits": logits,
                    "rpn_deltas": deltas, "proposals": props, "rpn_taps": rpn_taps, "pooled7": pooled7,
                    "lvl7": lv7, "cls": cls_all, "reg": reg_all, "det_boxes": det_boxes,
                    "det_scores": det_scores, "det_taps": det_taps, "pooled14": pooled14, "lvl14": lv14}
            return outs, taps
        return outs
