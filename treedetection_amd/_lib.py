"""ctypes binding of libtreedet_hip.so (C ABI declared in include/treedet.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C treedetection_amd/csrc``.
There is no CPU fallback: if the shared object is missing, or a call is made without a GPU, the
product path raises (SURVEY.md §8b — the oracle under ``oracle/`` is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtreedet_hip.so")

_lib: Optional[C.CDLL] = None
ERR_INVALID = -1    # TD_ERR_INVALID
ERR_CAPACITY = -4   # TD_ERR_CAPACITY
ERR_STATE = -5      # TD_ERR_STATE


class TdError(RuntimeError):
    """A libtreedet_hip call returned a negative status (message from td_last_error)."""


class ModelDesc(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("precision", C.c_int32), ("pre_nms_topk", C.c_int32),
                ("post_nms_topk", C.c_int32), ("detections_per_image", C.c_int32),
                ("rpn_nms_thresh", C.c_float), ("score_thresh", C.c_float), ("nms_thresh", C.c_float),
                ("mask_thresh", C.c_float)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class Detections(C.Structure):
    _fields_ = [("boxes", C.c_void_p), ("scores", C.c_void_p), ("classes", C.c_void_p), ("count", C.c_void_p),
                ("mask_probs", C.c_void_p), ("mask_region", C.c_void_p), ("mask_offset", C.c_void_p),
                ("mask_bits", C.c_void_p), ("mask_words_per_image", C.c_int64)]


# name -> (restype, argtypes); every symbol include/treedet.h declares
SIGNATURES = {
    "td_model_desc_default": (None, [C.POINTER(ModelDesc)]),
    "td_engine_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "td_engine_load_weights": (C.c_int, [C.c_void_p, C.POINTER(TensorDesc), C.c_size_t]),
    "td_engine_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "td_engine_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Detections)]),
    "td_engine_forward_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Detections)]),
    "td_engine_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int)]),
    "td_engine_read_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "td_engine_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "td_engine_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.c_int]),
    "td_engine_profile_classes": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "td_last_error": (C.c_char_p, []),
    "td_engine_destroy": (None, [C.c_void_p]),
    "td_resize_tile_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "td_resize_batch_u8": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "td_bottleneck_tail_nhwc": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]),
    "td_resize_bilinear_f64": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "td_resize_shape": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "td_conv2d_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
                       + [C.c_int] * 11 + [C.c_void_p]),
    "td_conv2d_head_nhwc": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 9 + [C.c_void_p]),
    "td_conv2d_winograd_head_nhwc": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    "td_conv2d_winograd_nhwc": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "td_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "td_roi_align": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int,
                               C.c_void_p, C.c_int, C.c_void_p]),
    "td_paste_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "td_paste_masks_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "td_crown_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p, C.c_int,
                                 C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "td_trace_contours_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "td_find_contours": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "td_batch_prediction_files": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_void_p]),
    "td_tiff_lzw_decode": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "td_tiff_packbits_decode": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "td_tiff_lzw_encode": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "td_tiff_inflate": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "td_tiff_inflate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "td_tiff_lzw_decode_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "td_tiff_blocks_to_image_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "td_read_windows": (C.c_int64, [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "td_tiff_unpredict": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "td_region_relate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "td_simplify_ring": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int]),
    "td_ring_is_valid": (C.c_int, [C.c_void_p, C.c_int]),
    "td_stitch_tile_json": (C.c_int, [C.c_char_p, C.c_int64, C.POINTER(C.c_double), C.c_double, C.c_int32, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "td_stitch_tile_files": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "td_tile_polygons_json_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_void_p, C.c_int64,
                                            C.POINTER(C.c_int64)]),
    "td_read_window": (C.c_int64, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "td_tile_prediction_file": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "td_tile_polygons_json": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_double), C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
}


def load() -> C.CDLL:
    """Load the shared object (after torch, so both share one HIP runtime) and bind signatures."""
    global _lib
    if _lib is not None:
        return _lib
    host_only = os.environ.get("TD_HOST_LIB")
    if host_only:
        # sanitizer runs of the CPU tests (make -C treedetection_amd/csrc asan-test): the host-side C++ built with
        # AddressSanitizer + UBSan and no device code. Only the symbols it exports are bound; anything that needs the
        # GPU library fails loudly on the missing attribute.
        lib = C.CDLL(host_only, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = lib
        return lib
    if not os.path.exists(LIB_PATH):
        raise TdError(f"{LIB_PATH} not found — build it first: python -c 'import __graft_entry__ as g; g.build()' "
                      f"(or make -C treedetection_amd/csrc). There is no CPU fallback.")
    import torch  # noqa: F401  (loads libamdhip64 first; our .so then binds to the same runtime)

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status < 0:
        msg = load().td_last_error().decode("utf-8", "replace")
        raise TdError(f"{what or 'libtreedet_hip'} failed ({status}): {msg}")


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the current device."""
    import torch

    return int(torch.cuda.current_stream().cuda_stream)


_low_priority_streams: dict = {}
_low_priority_lock = threading.Lock()


def low_priority_stream(device_index: int):
    """→ a torch stream wrapping a HIP stream with the LOWEST priority the device offers (torch itself only creates normal and
    higher). Background work that shares the chip with the forwards — the raster decode of the NEXT image — goes there: HIP gives
    every priority level its own hardware queues, so the long decode kernel never sits in front of a model stream's kernels in
    one queue (with the default four queues and five or more streams it did: tools/probes/decode_overlap_probe.py, 763 vs 1 732
    tiles/s while a DEFLATE raster decodes), and the dispatcher serves the forwards' workgroups first. The HIP entry points are
    taken from the runtime libtreedet_hip.so is bound to — the one torch loaded. ONE stream per device for the life of the
    process, never destroyed: torch's allocators remember the streams a block was used on and record events on them when the
    block is freed, long after a Predictor is closed (destroying the stream at close() ended in a segmentation fault there)."""
    import torch

    with _low_priority_lock:
        if device_index in _low_priority_streams:
            return _low_priority_streams[device_index]
        lib = load()
        least, greatest, handle = C.c_int(0), C.c_int(0), C.c_void_p()
        with torch.cuda.device(device_index):
            torch.cuda.current_stream()                                    # the device's context exists before the raw calls
            err = lib.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest))
            if err == 0:
                err = lib.hipStreamCreateWithPriority(C.byref(handle), C.c_uint(1), C.c_int(least.value))    # 1 = hipStreamNonBlocking
            if err != 0 or not handle.value:
                raise TdError(f"hipStreamCreateWithPriority(priority {least.value}) failed: hipError {err}")
            stream = torch.cuda.ExternalStream(handle.value, device=device_index)
        _low_priority_streams[device_index] = stream
        return stream
