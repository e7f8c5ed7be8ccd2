"""Python handle of the HIP tile-inference engine (``td_engine`` in include/treedet.h).

This is the object that replaces ``self.model`` of the reference's ``Predictor``
(TreeDetection/prediction.py:182-183): it takes a batch of model-input tensors and returns the
``Instances`` fields the reference consumes (``pred_boxes``, ``scores``, ``pred_classes``,
``pred_masks``) — computed by libtreedet_hip.so on an MI355X. torch tensors are used only as
device-buffer containers (``data_ptr()``); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import Detections, ModelDesc, TensorDesc

MASK_SIDE = 28
INPUT_F32_CHW = 0
INPUT_U8_HWC = 1
PHASE_STEM = 6      # TD_PHASE_STEM
PRECISIONS = {"fp32": 0, "fp16": 1}


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class Engine:
    def __init__(self, state_dict: Dict[str, np.ndarray], device: int = 0, precision: str = "fp32",
                 score_thresh: float = 0.3, nms_thresh: float = 0.5, rpn_nms_thresh: float = 0.7,
                 pre_nms_topk: int = 1000, post_nms_topk: int = 1000, detections_per_image: int = 100,
                 mask_thresh: float = 0.5):
        if not torch.cuda.is_available():
            raise _lib.TdError("treedetection_amd.Engine needs a GPU (no CPU fallback; the oracle under oracle/ "
                               "is test infrastructure only)")
        self.lib = _lib.load()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        desc = ModelDesc()
        self.lib.td_model_desc_default(C.byref(desc))
        desc.precision = PRECISIONS[precision]
        desc.score_thresh = score_thresh
        desc.nms_thresh = nms_thresh
        desc.rpn_nms_thresh = rpn_nms_thresh
        desc.pre_nms_topk = pre_nms_topk
        desc.post_nms_topk = post_nms_topk
        desc.detections_per_image = detections_per_image
        desc.mask_thresh = mask_thresh
        self.desc = desc
        self.D = detections_per_image
        self.P = post_nms_topk
        handle = C.c_void_p()
        _lib.check(self.lib.td_engine_create(C.byref(desc), self.device, C.byref(handle)), "td_engine_create")
        self._h = handle
        self._load(state_dict)
        self._reserved = (0, 0, 0)

    # -- life cycle -------------------------------------------------------------------------------
    def _load(self, sd: Dict[str, np.ndarray]) -> None:
        keep = []   # host arrays must outlive the call
        descs = (TensorDesc * len(sd))()
        for i, (k, v) in enumerate(sd.items()):
            a = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(a)
            descs[i].name = k.encode()
            descs[i].data = a.ctypes.data
            descs[i].ndim = a.ndim
            for j, s in enumerate(a.shape):
                descs[i].shape[j] = s
        _lib.check(self.lib.td_engine_load_weights(self._h, descs, len(sd)), "td_engine_load_weights")

    def reserve(self, batch: int, hp: int, wp: int) -> None:
        b, h, w = self._reserved
        if batch <= b and hp <= h and wp <= w:
            return
        batch, hp, wp = max(batch, b), max(hp, h), max(wp, w)
        _lib.check(self.lib.td_engine_reserve(self._h, batch, hp, wp), "td_engine_reserve")
        self._reserved = (batch, hp, wp)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.td_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward ------------------------------------------------------------------------------------
    def alloc_outputs(self, B: int, max_out_h: int, max_out_w: int, paste: bool = True) -> Dict[str, torch.Tensor]:
        D = self.D
        dev = torch.device("cuda", self.device)
        out = {
            "boxes": torch.empty((B, D, 4), dtype=torch.float32, device=dev),
            "scores": torch.empty((B, D), dtype=torch.float32, device=dev),
            "classes": torch.empty((B, D), dtype=torch.int32, device=dev),
            "count": torch.zeros((B,), dtype=torch.int32, device=dev),
            "mask_probs": torch.empty((B, D, MASK_SIDE, MASK_SIDE), dtype=torch.float32, device=dev),
        }
        if paste:
            words = D * ((max_out_w + 2 + 31) // 32) * max_out_h
            out["mask_region"] = torch.empty((B, D, 4), dtype=torch.int32, device=dev)
            out["mask_offset"] = torch.empty((B, D), dtype=torch.int64, device=dev)
            out["mask_bits"] = torch.empty((B, words), dtype=torch.int32, device=dev)
        return out

    def forward_raw(self, images: torch.Tensor, input_format: int, hw_valid: Sequence[Sequence[int]],
                    hw_out: Sequence[Sequence[int]], out: Dict[str, torch.Tensor]) -> None:
        """Asynchronous launch on torch's current stream; ``out`` from :meth:`alloc_outputs`."""
        if input_format == INPUT_F32_CHW:
            B, c, Hp, Wp = images.shape
            assert c == 3 and images.dtype == torch.float32
        else:
            B, Hp, Wp, c = images.shape
            assert c == 3 and images.dtype == torch.uint8
        assert images.is_cuda and images.is_contiguous()
        self.reserve(B, Hp, Wp)
        hv = (C.c_int32 * (2 * B))(*[int(v) for p in hw_valid for v in p])
        ho = (C.c_int32 * (2 * B))(*[int(v) for p in hw_out for v in p])
        det = Detections()
        det.boxes = out["boxes"].data_ptr()
        det.scores = out["scores"].data_ptr()
        det.classes = out["classes"].data_ptr()
        det.count = out["count"].data_ptr()
        det.mask_probs = out["mask_probs"].data_ptr()
        if "mask_bits" in out:
            det.mask_region = out["mask_region"].data_ptr()
            det.mask_offset = out["mask_offset"].data_ptr()
            det.mask_bits = out["mask_bits"].data_ptr()
            det.mask_words_per_image = out["mask_bits"].shape[1]
        st = self.lib.td_engine_forward(self._h, images.data_ptr(), input_format, hv, ho, B, Hp, Wp,
                                        _lib.stream_ptr(), C.byref(det))
        _lib.check(st, "td_engine_forward")

    def forward_phase(self, phase: int, stream: torch.cuda.Stream, images: Optional[torch.Tensor] = None,
                      input_format: int = INPUT_U8_HWC, hw_valid=None, hw_out=None, out: Optional[Dict[str, torch.Tensor]] = None) -> None:
        """One of the six phases of the forward on `stream` (see td_engine_forward_phase). Phase 0 takes the batch;
        ``PHASE_STEM`` (6) is the optional pre-phase (stem + pool of the next batch) with the same arguments, after which
        phase 0 is called without a batch."""
        if phase in (0, PHASE_STEM) and images is not None:
            B, Hp, Wp = (images.shape[0], images.shape[2], images.shape[3]) if input_format == INPUT_F32_CHW else images.shape[:3]
            self.reserve(B, Hp, Wp)
            hv = (C.c_int32 * (2 * B))(*[int(v) for p in hw_valid for v in p])
            ho = (C.c_int32 * (2 * B))(*[int(v) for p in hw_out for v in p])
            det = Detections()
            for k in ("boxes", "scores", "classes", "count", "mask_probs"):
                setattr(det, k, out[k].data_ptr())
            if "mask_bits" in out:
                det.mask_region = out["mask_region"].data_ptr()
                det.mask_offset = out["mask_offset"].data_ptr()
                det.mask_bits = out["mask_bits"].data_ptr()
                det.mask_words_per_image = out["mask_bits"].shape[1]
            self._phase_keep = (images, out)      # keep the buffers alive until the batch has drained
            st = self.lib.td_engine_forward_phase(self._h, phase, images.data_ptr(), input_format, hv, ho, B, Hp, Wp,
                                                  int(stream.cuda_stream), C.byref(det))
        else:
            st = self.lib.td_engine_forward_phase(self._h, phase, None, 0, None, None, 0, 0, 0, int(stream.cuda_stream), None)
        _lib.check(st, f"td_engine_forward_phase({phase})")

    def tensor(self, name: str) -> torch.Tensor:
        """Copy of an internal activation of the last forward (stage-wise parity tests)."""
        ptr = C.c_void_p()
        dims = (C.c_int64 * 4)()
        elem = C.c_int()
        _lib.check(self.lib.td_engine_tensor(self._h, name.encode(), C.byref(ptr), dims, C.byref(elem)), "td_engine_tensor")
        shape = [int(d) for d in dims if d > 0]
        integer = name in ("rpn_cand_valid", "rpn_cand_idx", "rpn_keep", "rpn_keep_count", "proposal_count", "det_flags")
        dt = torch.int32 if integer else (torch.float32 if elem.value == 4 else torch.float16)
        t = torch.empty(shape, dtype=dt, device=torch.device("cuda", self.device))
        _lib.check(self.lib.td_engine_read_tensor(self._h, name.encode(), t.data_ptr(), t.numel() * t.element_size(),
                                                  _lib.stream_ptr()), "td_engine_read_tensor")
        torch.cuda.synchronize()
        return t

    # -- tile preprocessing (reference Predictor._process_tile, prediction.py:159-176) -----------------------
    def resize_shape(self, h: int, w: int) -> tuple:
        a, b = C.c_int(), C.c_int()
        self.lib.td_resize_shape(int(h), int(w), 800, 1333, C.byref(a), C.byref(b))
        return a.value, b.value

    def preprocess_tiles_u8(self, tiles: Sequence[torch.Tensor]):
        """uint8 CUDA tiles [h,w,C>=3] (band order as in the file: R,G,B[,I]) → (uint8 [B,Hp,Wp,3] BGR batch resized with
        Pillow-exact bilinear, hw_valid, hw_out). Everything stays in HBM."""
        shapes = [self.resize_shape(t.shape[0], t.shape[1]) for t in tiles]
        Hp = _round_up(max(s[0] for s in shapes), 32)
        Wp = _round_up(max(s[1] for s in shapes), 32)
        dev = torch.device("cuda", self.device)
        key = (len(tiles), Hp, Wp)
        cache = self.__dict__.setdefault("_pp_cache", {})
        if key not in cache:
            if len(cache) >= 8:
                cache.clear()
            cache[key] = torch.zeros((len(tiles), Hp, Wp, 3), dtype=torch.uint8, device=dev)
        batch = cache[key]
        need_tmp = sum(t.shape[0] * s[1] * 3 for t, s in zip(tiles, shapes))
        if getattr(self, "_pp_tmp", None) is None or self._pp_tmp.numel() < need_tmp:
            self._pp_tmp = torch.empty((need_tmp,), dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr()
        for t in tiles:
            assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.shape[2] >= 3
        if len({tuple(t.shape) for t in tiles}) == 1:       # the common case: one launch pair for the whole batch
            oh, ow = shapes[0]
            if oh < Hp or ow < Wp:
                batch.zero_()
            ptrs = (C.c_void_p * len(tiles))(*[t.data_ptr() for t in tiles])
            t0 = tiles[0]
            _lib.check(self.lib.td_resize_batch_u8(ptrs, len(tiles), t0.shape[0], t0.shape[1], t0.shape[2], batch.data_ptr(),
                                                   oh, ow, Wp, Hp * Wp * 3, self._pp_tmp.data_ptr(), st), "td_resize_batch_u8")
        else:
            for i, (t, (oh, ow)) in enumerate(zip(tiles, shapes)):
                if oh < Hp or ow < Wp:
                    batch[i].zero_()
                _lib.check(self.lib.td_resize_tile_u8(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], batch[i].data_ptr(),
                                                      oh, ow, Wp, self._pp_tmp.data_ptr(), st), "td_resize_tile_u8")
        hw_out = [(int(t.shape[0]), int(t.shape[1])) for t in tiles]
        return batch, shapes, hw_out

    def preprocess_tiles_f64(self, planes: Sequence[torch.Tensor]):
        """float64 CUDA tiles [3,h,w] (BGR already picked, 16-bit imagery already rescaled: reference prediction.py:166-167)
        → (float32 [B,3,Hp,Wp] batch, hw_valid): the reference's float resize branch — detectron2 hands non-uint8 images
        to F.interpolate(bilinear, align_corners=False) — on the device (td_resize_bilinear_f64), zero padding included."""
        shapes = [self.resize_shape(t.shape[1], t.shape[2]) for t in planes]
        Hp = _round_up(max(s[0] for s in shapes), 32)
        Wp = _round_up(max(s[1] for s in shapes), 32)
        dev = torch.device("cuda", self.device)
        x = torch.zeros((len(planes), 3, Hp, Wp), dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        for i, (t, (oh, ow)) in enumerate(zip(planes, shapes)):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.shape[0] == 3
            _lib.check(self.lib.td_resize_bilinear_f64(t.data_ptr(), 3, t.shape[1], t.shape[2], x[i].data_ptr(), oh, ow, Wp, Hp * Wp, st),
                       "td_resize_bilinear_f64")
        return x, shapes

    CONTOUR_MAX = 256         # TD_CONTOUR_MAX

    def alloc_contours(self, B: int, points_cap: int = 65536) -> Dict[str, torch.Tensor]:
        """Device buffers of td_trace_contours_dev for a batch of B images."""
        dev = torch.device("cuda", self.device)
        return {"points": torch.empty((B, points_cap, 2), dtype=torch.int16, device=dev),
                "image_points": torch.zeros((B,), dtype=torch.int32, device=dev),
                "det_info": torch.zeros((B, self.D, 4), dtype=torch.int32, device=dev),
                "contour_info": torch.empty((B, self.D, self.CONTOUR_MAX, 2), dtype=torch.int32, device=dev)}

    def trace_contours(self, out: Dict[str, torch.Tensor], cont: Dict[str, torch.Tensor], B: int, stream: Optional[torch.cuda.Stream] = None) -> None:
        """td_trace_contours_dev on the pasted masks of ``out`` (first B images) — asynchronous on ``stream`` (default:
        torch's current stream), to be enqueued after the forward's last phase."""
        s = int(stream.cuda_stream) if stream is not None else _lib.stream_ptr()
        _lib.check(self.lib.td_trace_contours_dev(out["mask_region"].data_ptr(), out["mask_offset"].data_ptr(), out["mask_bits"].data_ptr(),
                                                  int(out["mask_bits"].shape[1]), out["count"].data_ptr(), B, self.D,
                                                  cont["points"].data_ptr(), int(cont["points"].shape[1]), cont["image_points"].data_ptr(),
                                                  cont["det_info"].data_ptr(), cont["contour_info"].data_ptr(), s), "td_trace_contours_dev")

    def paste_masks_batch(self, probs: torch.Tensor, boxes: torch.Tensor, counts: torch.Tensor, hw: Sequence[Sequence[int]],
                          out: Dict[str, torch.Tensor], thresh: float = 0.5) -> None:
        """td_paste_masks_batch: the first ``len(hw)`` images of probs [B,D,28,28] / boxes [B,D,4] / counts [B] (CUDA
        tensors) → out["mask_region" | "mask_offset" | "mask_bits"] rows, asynchronously on torch's current stream."""
        n = len(hw)
        assert probs.is_cuda and boxes.is_cuda and counts.is_cuda and probs.is_contiguous() and boxes.is_contiguous()
        assert probs.dtype == torch.float32 and boxes.dtype == torch.float32 and counts.dtype == torch.int32
        ohw = (C.c_int32 * (2 * n))(*[int(v) for p in hw for v in p])
        _lib.check(self.lib.td_paste_masks_batch(probs.data_ptr(), boxes.data_ptr(), counts.data_ptr(), ohw, n, int(probs.shape[1]),
                                                 thresh, out["mask_region"].data_ptr(), out["mask_offset"].data_ptr(),
                                                 out["mask_bits"].data_ptr(), int(out["mask_bits"].shape[1]), _lib.stream_ptr()),
                   "td_paste_masks_batch")

    def paste_masks_packed(self, probs: torch.Tensor, boxes: torch.Tensor, h: int, w: int, thresh: float = 0.5):
        """td_paste_masks for n detections of one tile (CUDA tensors) → (region [n,4] int32, offset [n] int64, packed
        bit rows int32) on the host. Used by rank 0 for detections gathered from other ranks."""
        n = int(probs.shape[0])
        if n == 0:
            return np.zeros((0, 4), np.int32), np.zeros((0,), np.int64), np.zeros((1,), np.int32)
        dev = probs.device
        words = n * ((w + 2 + 31) // 32) * h
        region = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        offset = torch.zeros((n,), dtype=torch.int64, device=dev)
        bits = torch.zeros((words,), dtype=torch.int32, device=dev)
        probs = probs.contiguous().float()
        boxes = boxes.contiguous().float()
        _lib.check(self.lib.td_paste_masks(probs.data_ptr(), boxes.data_ptr(), n, h, w, thresh, region.data_ptr(),
                                           offset.data_ptr(), bits.data_ptr(), words, _lib.stream_ptr()), "td_paste_masks")
        torch.cuda.synchronize()
        return region.cpu().numpy(), offset.cpu().numpy(), bits.cpu().numpy()

    def paste_masks(self, probs: torch.Tensor, boxes: torch.Tensor, h: int, w: int, thresh: float = 0.5):
        """:meth:`paste_masks_packed` unpacked to (region [n,4], bool masks [n,h,w])."""
        n = int(probs.shape[0])
        if n == 0:
            return np.zeros((0, 4), np.int32), np.zeros((0, h, w), bool)
        rg, off, bits = self.paste_masks_packed(probs, boxes, h, w, thresh)
        return rg, unpack_masks(rg, off, bits, n, h, w)

    # -- device timing ------------------------------------------------------------------------------------
    PROF_NAMES = ("conv_igemm", "stem", "pool", "rpn_select", "roi_align", "detect", "mask_tail", "mask_convs", "executed")

    CLASS_NAMES = ("wino_contraction", "wino_transform", "conv1x1", "conv3x3_direct", "fc", "mask_head", "bottleneck_tail")

    def profile_enable(self, on=True) -> None:
        """on: False / True, or 2 = detail (an event pair per contraction launch for :meth:`profile_classes`)."""
        _lib.check(self.lib.td_engine_profile_enable(self._h, int(on)), "td_engine_profile_enable")

    def profile_classes(self, reset: bool = True) -> Dict[str, dict]:
        """Speed-of-light accounting of the contraction family by class (include/treedet.h td_engine_profile_classes)."""
        n = len(self.CLASS_NAMES)
        ms, fl, by, tm = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        la = (C.c_int64 * n)()
        _lib.check(self.lib.td_engine_profile_classes(self._h, ms, la, fl, by, tm, int(reset)), "td_engine_profile_classes")
        return {k: {"ms": ms[i], "launches": int(la[i]), "exec_flops": fl[i], "bytes": by[i], "tmin_ms": tm[i]}
                for i, k in enumerate(self.CLASS_NAMES)}

    def profile_read(self, reset: bool = True) -> Dict[str, dict]:
        n = len(self.PROF_NAMES)
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        la = (C.c_int64 * n)()
        _lib.check(self.lib.td_engine_profile_read(self._h, ms, la, fl, by, int(reset)), "td_engine_profile_read")
        return {k: {"ms": ms[i], "launches": int(la[i]), "flops": fl[i], "bytes": by[i]} for i, k in enumerate(self.PROF_NAMES)}

    # -- the reference's model seam ---------------------------------------------------------------------
    def __call__(self, batched_inputs: List[dict], paste: bool = True) -> List[dict]:
        """Same contract as ``self.model(batch_tensors)`` (prediction.py:182-183).

        batched_inputs: list of {"image": float32 [3,H',W'] BGR 0..255 (CPU or CUDA tensor / ndarray),
        "height": h, "width": w}. Returns per image {"pred_boxes" [N,4], "scores" [N], "pred_classes" [N],
        "pred_masks" bool [N,h,w], "mask_probs" [N,28,28]} as numpy arrays.
        """
        B = len(batched_inputs)
        sizes = [(int(b["image"].shape[1]), int(b["image"].shape[2])) for b in batched_inputs]
        Hp = _round_up(max(s[0] for s in sizes), 32)
        Wp = _round_up(max(s[1] for s in sizes), 32)
        dev = torch.device("cuda", self.device)
        x = torch.zeros((B, 3, Hp, Wp), dtype=torch.float32, device=dev)
        for i, b in enumerate(batched_inputs):
            t = torch.as_tensor(b["image"], dtype=torch.float32)
            x[i, :, : sizes[i][0], : sizes[i][1]] = t.to(dev)
        hw_out = [(int(b.get("height", s[0])), int(b.get("width", s[1]))) for b, s in zip(batched_inputs, sizes)]
        out = self.alloc_outputs(B, max(h for h, _ in hw_out), max(w for _, w in hw_out), paste)
        self.forward_raw(x, INPUT_F32_CHW, sizes, hw_out, out)
        torch.cuda.synchronize()
        return unpack_outputs(out, hw_out, paste)


def unpack_masks(region: np.ndarray, offset: np.ndarray, bits: np.ndarray, n: int, h: int, w: int) -> np.ndarray:
    """Packed bit rows of one image → bool [n, h, w] (what detectron2's paste_masks_in_image returns)."""
    masks = np.zeros((n, h, w), dtype=bool)
    bits = bits.view(np.uint32)
    for d in range(n):
        x0, y0, x1, y1 = (int(v) for v in region[d])
        if x1 <= x0 or y1 <= y0:
            continue
        wpr = (x1 - x0 + 31) // 32
        words = bits[int(offset[d]): int(offset[d]) + wpr * (y1 - y0)].reshape(y1 - y0, wpr)
        row_bits = np.unpackbits(words.view(np.uint8).reshape(y1 - y0, wpr * 4), axis=1, bitorder="little")
        masks[d, y0:y1, x0:x1] = row_bits[:, : x1 - x0].astype(bool)
    return masks


def unpack_outputs(out: Dict[str, torch.Tensor], hw_out, paste: bool = True) -> List[dict]:
    cnt = out["count"].cpu().numpy()
    boxes = out["boxes"].cpu().numpy()
    scores = out["scores"].cpu().numpy()
    classes = out["classes"].cpu().numpy()
    probs = out["mask_probs"].cpu().numpy()
    res = []
    if paste and "mask_bits" in out:
        region = out["mask_region"].cpu().numpy()
        offset = out["mask_offset"].cpu().numpy()
        bits = out["mask_bits"].cpu().numpy()
    for i, (h, w) in enumerate(hw_out):
        n = int(cnt[i])
        r = {"pred_boxes": boxes[i, :n].copy(), "scores": scores[i, :n].copy(),
             "pred_classes": classes[i, :n].astype(np.int64), "mask_probs": probs[i, :n].copy(), "pred_masks": None}
        if paste and "mask_bits" in out:
            r["pred_masks"] = unpack_masks(region[i], offset[i], bits[i], n, h, w)
            r["mask_region"] = region[i, :n].copy()
        res.append(r)
    return res
