"""Corrupt, truncated and adversarial inputs to the host-side C++ entry points (the code that parses bytes read from
files: TIFF codecs, prediction JSON, packed mask rows, rings). Every call must come back with a status — never read
or write out of bounds. Meant to run under the sanitizer build as well: ``make -C treedetection_amd/csrc asan-test``
(AddressSanitizer + UBSan abort the process on the first bad access), which tests/test_sanitizers.py drives."""
import ctypes as C
import json

import numpy as np
import pytest

from treedetection_amd import _lib


def _stitch(text: bytes, cap_b=1 << 16, cap_f=256):
    lib = _lib.load()
    box = (C.c_double * 4)(0.0, 0.0, 100.0, 100.0)
    blobs = np.empty(cap_b, np.uint8)
    offsets = np.empty(cap_f + 1, np.int64)
    scores = np.empty(cap_f, np.float64)
    nb, nf = C.c_int64(0), C.c_int(0)
    n = lib.td_stitch_tile_json(text, len(text), box, 0.2, 25832, blobs.ctypes.data, cap_b, offsets.ctypes.data,
                                scores.ctypes.data, cap_f, C.byref(nb), C.byref(nf))
    return n


GOOD = json.dumps([{"image_id": "a.tif", "category_id": 0, "score": 0.9,
                    "polygon_coords": [[[10.0, 10.0], [50.0, 10.0], [50.0, 50.0], [10.0, 50.0], [10.0, 10.0]]]}]).encode()


def test_stitch_json_truncations_and_garbage():
    assert _stitch(GOOD) == 1
    for cut in range(0, len(GOOD), 3):                    # every prefix: an error status or fewer features, never a crash
        assert _stitch(GOOD[:cut]) <= 1
    rng = np.random.default_rng(0)
    for _ in range(200):                                   # byte flips
        b = bytearray(GOOD)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        assert _stitch(bytes(b)) <= 1
    for text in (b"", b"[", b"]", b"[[[[[[[[[[[[[[[[", b'[{"polygon_coords": [[[1e999, -1e999], [NaN, 0]]]}]', b"\x00" * 64,
                 b'[{"score": 1, "polygon_coords": [[]]}]', b'[{"score": 1, "polygon_coords": [[[1,2]]]}]',
                 b'[{"score": "x", "polygon_coords": 7}]', b"[" + b'{"a":' * 500 + b"1" + b"}" * 500 + b"]"):
        assert _stitch(text) <= 1
    # tiny output capacities: the call reports what it needs instead of overrunning
    assert _stitch(GOOD, cap_b=8, cap_f=1) in (_lib.ERR_CAPACITY, 1)
    assert _stitch(GOOD, cap_b=1 << 16, cap_f=0) in (_lib.ERR_CAPACITY, 0, 1)


def test_tiff_codecs_on_random_bytes():
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 7, 64, 1000):
        for _ in range(60):
            src = rng.integers(0, 256, n, dtype=np.uint8)
            for cap in (0, 1, 5, 4096):
                dst = np.zeros(max(cap, 1), np.uint8)
                for fn in (lib.td_tiff_lzw_decode, lib.td_tiff_packbits_decode):
                    got = fn(src.ctypes.data, n, dst.ctypes.data, cap)
                    assert got <= cap
    # LZW stream that keeps defining codes until the table is full, then more
    stream = np.frombuffer(bytes([0x80]) + bytes(rng.integers(0, 256, 20000, dtype=np.uint8)), np.uint8)
    dst = np.zeros(1 << 16, np.uint8)
    assert lib.td_tiff_lzw_decode(stream.ctypes.data, stream.size, dst.ctypes.data, dst.size) <= dst.size
    # horizontal predictor on degenerate shapes
    for rows, cols, samples, bps in ((0, 0, 1, 1), (1, 1, 1, 1), (3, 1, 4, 2), (2, 5, 3, 4), (1, 7, 1, 8)):
        buf = np.zeros(max(rows * cols * samples * bps, 1), np.uint8)
        assert lib.td_tiff_unpredict(buf.ctypes.data, rows, cols, samples, bps) in (0, -1, -2, -3, -4, -5)


def _poly_json(regions, offsets, words, n):
    lib = _lib.load()
    regions = np.ascontiguousarray(regions, np.int32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    words = np.ascontiguousarray(words, np.uint32)
    scores = np.full(n, 0.5, np.float32)
    classes = np.zeros(n, np.int32)
    tr = (C.c_double * 6)(0.2, 0.0, 0.0, 0.0, -0.2, 100.0)
    need = C.c_int64(0)
    buf = C.create_string_buffer(1 << 20)
    return lib.td_tile_polygons_json(regions.ctypes.data, offsets.ctypes.data, words.ctypes.data, words.size, scores.ctypes.data,
                                     classes.ctypes.data, n, tr, b"img.tif", buf, 1 << 20, C.byref(need))


def test_packed_masks_with_inconsistent_offsets():
    """Regions / offsets that point outside the word buffer (a corrupted or mis-sized D2H copy) must be refused, not
    dereferenced."""
    words = np.full(64, 0xFFFFFFFF, np.uint32)
    assert _poly_json([[0, 0, 32, 32]], [0], words, 1) == 1                       # 32 rows x 1 word: fits, one square ring
    for region, off in (([0, 0, 32, 33], 32), ([0, 0, 64, 64], 0), ([0, 0, 32, 32], 40), ([0, 0, 32, 32], -8),
                        ([0, 0, 1 << 20, 1 << 20], 0), ([10, 10, 5, 5], 0), ([0, 0, 32, 32], 1 << 40),
                        ([-100, -100, 32, 32], 0), ([0, 0, 2147483647, 2], 0)):
        st = _poly_json([region], [off], words, 1)
        assert st <= 0, (region, off, st)                                        # refused (or an empty region: 0 entries)


def test_contours_degenerate_shapes_and_capacity():
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for h, w in ((1, 1), (1, 40), (40, 1), (2, 2), (17, 23)):
        for fill in (0.0, 0.5, 1.0):
            m = (rng.random((h, w)) < fill).astype(np.uint8) if 0 < fill < 1 else np.full((h, w), int(fill), np.uint8)
            for max_pts, max_ct in ((0, 0), (1, 1), (3, 1), (2 * h * w + 16, h * w)):
                pts = np.empty((max(max_pts, 1), 2), np.int32)
                starts = np.empty(max_ct + 1, np.int32)
                n = lib.td_find_contours(m.ctypes.data, h, w, pts.ctypes.data, max_pts, starts.ctypes.data, max_ct)
                assert n <= max_ct or n < 0


def test_simplify_ring_degenerate():
    lib = _lib.load()
    out = np.empty((64, 2), np.float64)
    for ring in ([[0, 0]], [[0, 0], [0, 0]], [[0, 0], [1, 1], [0, 0]], [[0, 0], [1e308, 1e308], [-1e308, 1e308], [0, 0]],
                 [[float("nan"), 0], [1, 1], [2, 0], [float("nan"), 0]], [[0, 0], [1, 0], [1, 1], [0, 1], [0, 0]] * 3):
        xy = np.ascontiguousarray(ring, np.float64)
        for cap in (0, 1, 3, 64):
            m = lib.td_simplify_ring(xy.ctypes.data, xy.shape[0], 0.5, out.ctypes.data, cap)
            assert m <= cap or m < 0


def test_region_relate_empty_and_degenerate():
    from treedetection_amd.vector import Region, box_ring
    r = Region([[np.array([[0, 0], [10, 0], [10, 10], [0, 10], [0, 0]], float)]])
    inter, within = r.relate([box_ring(1, 1, 2, 2), box_ring(20, 20, 30, 30), np.array([[5, 5], [5, 5], [5, 5], [5, 5]], float)])
    assert inter[0] and within[0] and not inter[1]
