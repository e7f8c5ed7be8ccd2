"""Mask → polygon epilogue of the reference's ``_process_and_save_single`` (prediction.py:229-249), host side.

``find_contours`` = cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE) computed by td_find_contours in
libtreedet_hip.so (pure host code in that library); ``xy`` = TreeDetection/utilities.py:182-207 (``xy_gpu``: affine
on pixel-corner integer coordinates, float64) without the cupy round trip.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib


def find_contours(mask: np.ndarray) -> List[np.ndarray]:
    """uint8/bool [h,w] → list of int32 [n,2] (x,y) contours in cv2 RETR_TREE order."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = m.shape
    lib = _lib.load()
    max_pts = max(64, 2 * h * w + 16)
    max_ct = max(16, (h * w) // 2 + 8)
    pts = np.empty((max_pts, 2), dtype=np.int32)
    starts = np.empty((max_ct + 1,), dtype=np.int32)
    n = lib.td_find_contours(m.ctypes.data, h, w, pts.ctypes.data, max_pts, starts.ctypes.data, max_ct)
    _lib.check(n, "td_find_contours")
    return [pts[starts[i]:starts[i + 1]].copy() for i in range(n)]


def xy(transform: Sequence[float], rows: Sequence[float], cols: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """x' = a*col + b*row + c ; y' = d*col + e*row + f   (utilities.py:203-204; no +0.5 pixel-centre offset)."""
    a, b, c, d, e, f = (float(v) for v in transform[:6])
    r = np.asarray(rows, dtype=np.float64)
    k = np.asarray(cols, dtype=np.float64)
    return a * k + b * r + c, d * k + e * r + f


def tile_polygons_json(regions: np.ndarray, offsets: np.ndarray, bits: np.ndarray, scores: np.ndarray,
                       classes: np.ndarray, transform: Sequence[float], image_id: str, _buf=None) -> bytes:
    """The text of one tile's ``Prediction_*.json`` (reference prediction.py:229-261) straight from the engine's
    packed masks — td_tile_polygons_json; the call releases the GIL, so tiles run in parallel on host threads."""
    lib = _lib.load()
    n = int(len(scores))
    regions = np.ascontiguousarray(regions[:n], dtype=np.int32)
    offsets = np.ascontiguousarray(offsets[:n], dtype=np.int64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    classes = np.ascontiguousarray(classes[:n], dtype=np.int32)
    words = np.ascontiguousarray(bits).view(np.uint32).reshape(-1)
    tr = (C.c_double * 6)(*[float(v) for v in transform[:6]])
    need = C.c_int64(0)
    cap = 1 << 20
    while True:
        buf = C.create_string_buffer(cap)
        st = lib.td_tile_polygons_json(regions.ctypes.data, offsets.ctypes.data, words.ctypes.data, words.size,
                                       scores.ctypes.data, classes.ctypes.data, n, tr, image_id.encode("utf-8", "surrogateescape"),
                                       buf, cap, C.byref(need))
        if st == _lib.ERR_CAPACITY:
            cap = int(need.value)
            continue
        _lib.check(st, "td_tile_polygons_json")
        return buf.raw[: need.value]


def tile_prediction_file(device: int, regions: np.ndarray, offsets: np.ndarray, bits_dev_ptr: int, rows_host: np.ndarray,
                         scores: np.ndarray, classes: np.ndarray, transform: Sequence[float], image_id: str, path: str) -> int:
    """One tile from the device to its ``Prediction_*.json`` (reference prediction.py:197-265): td_tile_prediction_file
    copies the words the paste wrote from ``bits_dev_ptr`` (device address of the image's bit rows) into the pinned
    ``rows_host``, traces, formats and writes ``path`` — all with the GIL released. → bytes written."""
    lib = _lib.load()
    n = int(len(scores))
    regions = np.ascontiguousarray(regions[:n], dtype=np.int32)
    offsets = np.ascontiguousarray(offsets[:n], dtype=np.int64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    classes = np.ascontiguousarray(classes[:n], dtype=np.int32)
    words = rows_host.view(np.uint32).reshape(-1)
    assert words.flags.c_contiguous
    tr = (C.c_double * 6)(*[float(v) for v in transform[:6]])
    wrote = C.c_int64(0)
    st = lib.td_tile_prediction_file(int(device), regions.ctypes.data, offsets.ctypes.data, int(bits_dev_ptr), words.ctypes.data,
                                     words.size, scores.ctypes.data, classes.ctypes.data, n, tr,
                                     image_id.encode("utf-8", "surrogateescape"), os.fsencode(path), C.byref(wrote))
    _lib.check(st, "td_tile_prediction_file")
    return int(wrote.value)


def batch_prediction_files(device: int, host: dict, n_tiles: int, bits_dev_ptr: int, transforms: np.ndarray, image_id: str,
                           paths: Sequence[str], threads: int = 4):
    """All tile files of one batch in ONE library call (td_batch_prediction_files): ``host`` = the slot's pinned numpy views of the
    engine outputs (count / mask_region / mask_offset / mask_bits / scores / classes, batch-major), ``bits_dev_ptr`` the device
    address of the batch's packed mask rows. → (status per tile, bytes per tile); raises on the first failing tile."""
    lib = _lib.load()
    D = int(host["scores"].shape[1])
    words = host["mask_bits"].view(np.uint32)
    stride = int(words.shape[1])
    status = np.zeros(n_tiles, np.int32)
    wrote = np.zeros(n_tiles, np.int64)
    arr = (C.c_char_p * n_tiles)(*[os.fsencode(p) for p in paths])
    tr = np.ascontiguousarray(transforms, dtype=np.float64).reshape(n_tiles, 6)
    st = lib.td_batch_prediction_files(int(device), int(n_tiles), D, host["mask_region"].ctypes.data, host["mask_offset"].ctypes.data,
                                       int(bits_dev_ptr), words.ctypes.data, stride, host["scores"].ctypes.data, host["classes"].ctypes.data,
                                       host["count"].ctypes.data, tr.ctypes.data, image_id.encode("utf-8", "surrogateescape"), arr, int(threads),
                                       status.ctypes.data, wrote.ctypes.data)
    _lib.check(st, "td_batch_prediction_files")
    return status, wrote


def tile_polygons_json_dev(points: np.ndarray, det_info: np.ndarray, contour_info: np.ndarray, regions: np.ndarray,
                           offsets: np.ndarray, bits, scores: np.ndarray, classes: np.ndarray, transform: Sequence[float],
                           image_id: str):
    """:func:`tile_polygons_json` from contours traced on the device (one image's slices of the td_trace_contours_dev
    outputs, on the host). ``bits`` may be None; returns None when a detection was left to the host tracer and the mask
    rows are needed — call again with them."""
    lib = _lib.load()
    n = int(len(scores))
    points = np.ascontiguousarray(points, dtype=np.int16)
    det_info = np.ascontiguousarray(det_info[:n], dtype=np.int32)
    contour_info = np.ascontiguousarray(contour_info[:n], dtype=np.int32)
    regions = np.ascontiguousarray(regions[:n], dtype=np.int32)
    offsets = np.ascontiguousarray(offsets[:n], dtype=np.int64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    classes = np.ascontiguousarray(classes[:n], dtype=np.int32)
    words = None if bits is None else np.ascontiguousarray(bits).view(np.uint32).reshape(-1)
    tr = (C.c_double * 6)(*[float(v) for v in transform[:6]])
    need = C.c_int64(0)
    cap = 1 << 20
    while True:
        buf = C.create_string_buffer(cap)
        st = lib.td_tile_polygons_json_dev(points.ctypes.data, points.shape[0], det_info.ctypes.data, contour_info.ctypes.data,
                                           regions.ctypes.data, offsets.ctypes.data, None if words is None else words.ctypes.data,
                                           0 if words is None else words.size, scores.ctypes.data, classes.ctypes.data, n, tr,
                                           image_id.encode("utf-8", "surrogateescape"), buf, cap, C.byref(need))
        if st == _lib.ERR_CAPACITY:
            cap = int(need.value)
            continue
        if st == _lib.ERR_STATE and words is None:
            return None
        _lib.check(st, "td_tile_polygons_json_dev")
        return buf.raw[: need.value]
