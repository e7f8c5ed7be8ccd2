"""Minimal GeoTIFF reader / writer for the tile loader of the prediction stage.

The reference reads rasters through rasterio/GDAL (``rasterio.open`` + ``rasterio.mask.mask(img, shapes, crop=True)``,
TreeDetection/prediction.py:61,164); neither is installed here, and the hot path only needs windowed reads of
pixel-interleaved or planar, strip- or tile-organised, uncompressed rasters plus the three geo tags
(ModelPixelScale / ModelTiepoint / GeoKeyDirectory). Compressed files fall back to Pillow.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
          9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}
_DTYPES = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 8): np.int8, (2, 16): np.int16,
           (2, 32): np.int32, (3, 32): np.float32, (3, 64): np.float64}


class GeoTiff:
    """Read-only view of one raster. Arrays come back as [bands, rows, cols] like rasterio's ``read``."""

    def __init__(self, path: str):
        self.path = path
        self._pil = None
        with open(path, "rb") as f:
            head = f.read(16)
            if head[:2] == b"II":
                self._e = "<"
            elif head[:2] == b"MM":
                self._e = ">"
            else:
                raise ValueError(f"{path}: not a TIFF")
            magic = struct.unpack(self._e + "H", head[2:4])[0]
            self._big = magic == 43
            if magic not in (42, 43):
                raise ValueError(f"{path}: bad TIFF magic {magic}")
            if self._big:
                ifd = struct.unpack(self._e + "Q", head[8:16])[0]
            else:
                ifd = struct.unpack(self._e + "I", head[4:8])[0]
            self.tags = self._read_ifd(f, ifd)
        t = self.tags
        self.width = int(t[256][0])
        self.height = int(t[257][0])
        self.count = int(t.get(277, [1])[0])
        bits = int(t.get(258, [8])[0])
        fmt = int(t.get(339, [1])[0])
        if (fmt, bits) not in _DTYPES:
            raise ValueError(f"{path}: unsupported sample format {fmt}/{bits} bits")
        self.dtype = np.dtype(_DTYPES[(fmt, bits)]).newbyteorder(self._e)
        self.planar = int(t.get(284, [1])[0])
        self.compression = int(t.get(259, [1])[0])
        self._data: Optional[np.ndarray] = None
        # georeferencing
        if 33550 in t and 33922 in t:
            sx, sy = float(t[33550][0]), float(t[33550][1])
            i, j, _, x, y, _ = (float(v) for v in t[33922][:6])
            self.transform = (sx, 0.0, x - i * sx, 0.0, -sy, y + j * sy)
        elif 34264 in t:
            m = [float(v) for v in t[34264]]
            self.transform = (m[0], m[1], m[3], m[4], m[5], m[7])
        else:
            self.transform = (1.0, 0.0, 0.0, 0.0, -1.0, float(self.height))
        self.epsg = None
        if 34735 in t:
            keys = [int(v) for v in t[34735]]
            for k in range(4, len(keys) - 3, 4):
                if keys[k] in (3072, 2048) and keys[k + 1] == 0:
                    self.epsg = keys[k + 3]
                    if keys[k] == 3072:
                        break

    # -- tiff structure ---------------------------------------------------------------------------
    def _read_ifd(self, f, off) -> Dict[int, Sequence]:
        e = self._e
        f.seek(off)
        if self._big:
            n = struct.unpack(e + "Q", f.read(8))[0]
            ent, fmt, inl = 20, e + "HHQ", 8
        else:
            n = struct.unpack(e + "H", f.read(2))[0]
            ent, fmt, inl = 12, e + "HHI", 4
        raw = f.read(n * ent)
        tags = {}
        for i in range(n):
            rec = raw[i * ent:(i + 1) * ent]
            tag, typ, cnt = struct.unpack(fmt, rec[:ent - inl])
            if typ not in _TYPES:
                continue
            code, size = _TYPES[typ]
            total = size * cnt
            if total <= inl:
                buf = rec[ent - inl:ent - inl + total]
            else:
                ptr = struct.unpack(e + ("Q" if self._big else "I"), rec[ent - inl:])[0]
                pos = f.tell()
                f.seek(ptr)
                buf = f.read(total)
                f.seek(pos)
            if typ == 2:
                tags[tag] = [buf.rstrip(b"\0").decode("latin1")]
            elif typ in (5, 10):
                v = struct.unpack(e + str(cnt * 2) + code[0], buf)
                tags[tag] = [v[2 * k] / v[2 * k + 1] if v[2 * k + 1] else 0.0 for k in range(cnt)]
            else:
                tags[tag] = list(struct.unpack(e + str(cnt) + code, buf))
        return tags

    def _load(self) -> np.ndarray:
        """Whole raster as [bands, rows, cols] (memory-mapped when it is one contiguous uncompressed block)."""
        if self._data is not None:
            return self._data
        t = self.tags
        H, W, C = self.height, self.width, self.count
        if self.compression != 1:
            from PIL import Image
            Image.MAX_IMAGE_PIXELS = None
            arr = np.asarray(Image.open(self.path))
            arr = arr[None] if arr.ndim == 2 else arr.transpose(2, 0, 1)
            self._data = np.ascontiguousarray(arr)
            return self._data
        item = self.dtype.itemsize
        if 324 in t:   # tiled
            tw, th = int(t[322][0]), int(t[323][0])
            offs = t[324]
            nx, ny = (W + tw - 1) // tw, (H + th - 1) // th
            out = np.zeros((C, H, W), dtype=self.dtype.newbyteorder("="))
            mm = np.memmap(self.path, dtype=np.uint8, mode="r")
            planes = C if self.planar == 2 else 1
            for p in range(planes):
                for ty in range(ny):
                    for tx in range(nx):
                        o = int(offs[(p * ny + ty) * nx + tx])
                        if self.planar == 2:
                            blk = np.frombuffer(mm[o:o + tw * th * item], dtype=self.dtype).reshape(th, tw)
                            out[p, ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = blk[:H - ty * th, :W - tx * tw]
                        else:
                            blk = np.frombuffer(mm[o:o + tw * th * C * item], dtype=self.dtype).reshape(th, tw, C)
                            out[:, ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw] = blk[:H - ty * th, :W - tx * tw].transpose(2, 0, 1)
            self._data = out
            return out
        offs = [int(v) for v in t[273]]
        rps = int(t.get(278, [H])[0])
        if self.planar == 1:
            row_bytes = W * C * item
            contiguous = all(offs[i + 1] - offs[i] == rps * row_bytes for i in range(len(offs) - 1))
            if contiguous:
                mm = np.memmap(self.path, dtype=self.dtype, mode="r", offset=offs[0], shape=(H, W, C))
                self._data = mm.transpose(2, 0, 1)
                return self._data
            mm = np.memmap(self.path, dtype=np.uint8, mode="r")
            out = np.empty((H, W, C), dtype=self.dtype)
            for i, o in enumerate(offs):
                r0 = i * rps
                r1 = min(r0 + rps, H)
                out[r0:r1] = np.frombuffer(mm[o:o + (r1 - r0) * row_bytes], dtype=self.dtype).reshape(r1 - r0, W, C)
            self._data = out.transpose(2, 0, 1)
            return self._data
        mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        out = np.empty((C, H, W), dtype=self.dtype)
        spp = (H + rps - 1) // rps
        for c in range(C):
            for i in range(spp):
                o = offs[c * spp + i]
                r0 = i * rps
                r1 = min(r0 + rps, H)
                out[c, r0:r1] = np.frombuffer(mm[o:o + (r1 - r0) * W * item], dtype=self.dtype).reshape(r1 - r0, W)
        self._data = out
        return out

    # -- rasterio-like surface ------------------------------------------------------------------------
    @property
    def bounds(self) -> Tuple[float, float, float, float]:
        a, _, c, _, e, f = self.transform
        return (c, f + e * self.height, c + a * self.width, f)   # left, bottom, right, top

    def read(self) -> np.ndarray:
        return np.asarray(self._load())

    def window_of_bounds(self, bounds: Sequence[float]) -> Tuple[int, int, int, int]:
        """rasterio.features.geometry_window for a bbox: floor / ceil in pixel space, clipped to the raster.
        Returns (col_off, row_off, width, height); width/height may be 0 when the bbox misses the raster."""
        a, _, c, _, e, f = self.transform
        minx, miny, maxx, maxy = (float(v) for v in bounds[:4])
        cols = [(minx - c) / a, (maxx - c) / a]
        rows = [(maxy - f) / e, (miny - f) / e]
        c0, c1 = int(math.floor(min(cols))), int(math.ceil(max(cols)))
        r0, r1 = int(math.floor(min(rows))), int(math.ceil(max(rows)))
        c0c, r0c = max(c0, 0), max(r0, 0)
        c1c, r1c = min(c1, self.width), min(r1, self.height)
        return c0c, r0c, max(c1c - c0c, 0), max(r1c - r0c, 0)

    def window_transform(self, col_off: int, row_off: int) -> Tuple[float, ...]:
        a, b, c, d, e, f = self.transform
        return (a, b, c + a * col_off + b * row_off, d, e, f + d * col_off + e * row_off)

    def read_bounds_hwc(self, bounds: Sequence[float], out: Optional[np.ndarray] = None, out_off: int = 0) -> np.ndarray:
        """Same pixels as :meth:`read_bounds` but pixel-interleaved [rows, cols, bands] and contiguous — the layout
        the device resize consumes; for chunky files this is a plain row-slab copy of the memory map. With ``out``
        (a flat array of the raster's dtype, e.g. pinned staging memory) the window is written at ``out_off`` and the
        returned array is a view of it."""
        c0, r0, w, h = self.window_of_bounds(bounds)
        if w <= 0 or h <= 0:
            raise ValueError("Input shapes do not overlap raster.")
        data = self._load()
        if isinstance(data, np.ndarray) and data.ndim == 3 and data.strides[0] == data.dtype.itemsize:
            src = data.transpose(1, 2, 0)[r0:r0 + h, c0:c0 + w, :]    # already HWC in memory
        else:
            src = np.asarray(data[:, r0:r0 + h, c0:c0 + w]).transpose(1, 2, 0)
        if out is not None:
            hwc = out[out_off:out_off + h * w * self.count].reshape(h, w, self.count)
            np.copyto(hwc, src)
        else:
            hwc = np.ascontiguousarray(src)
        a, _, c, _, e, f = self.transform
        minx, miny, maxx, maxy = (float(v) for v in bounds[:4])
        xs = c + a * (np.arange(c0, c0 + w) + 0.5)
        ys = f + e * (np.arange(r0, r0 + h) + 0.5)
        okx = (xs >= minx) & (xs <= maxx)
        oky = (ys >= miny) & (ys <= maxy)
        if not okx.all():
            hwc[:, ~okx, :] = 0
        if not oky.all():
            hwc[~oky, :, :] = 0
        return hwc

    def read_bounds(self, bounds: Sequence[float]) -> np.ndarray:
        """``rasterio.mask.mask(img, [bbox], crop=True)[0]``: the window covering the bbox, all bands, pixels whose
        centre lies outside the bbox set to 0. Raises ValueError when the bbox does not overlap the raster."""
        c0, r0, w, h = self.window_of_bounds(bounds)
        if w <= 0 or h <= 0:
            raise ValueError("Input shapes do not overlap raster.")
        out = np.array(self._load()[:, r0:r0 + h, c0:c0 + w])
        a, _, c, _, e, f = self.transform
        minx, miny, maxx, maxy = (float(v) for v in bounds[:4])
        xs = c + a * (np.arange(c0, c0 + w) + 0.5)
        ys = f + e * (np.arange(r0, r0 + h) + 0.5)
        okx = (xs >= minx) & (xs <= maxx)
        oky = (ys >= miny) & (ys <= maxy)
        if not okx.all():
            out[:, :, ~okx] = 0
        if not oky.all():
            out[:, ~oky, :] = 0
        return out


def write_geotiff(path: str, data: np.ndarray, transform: Sequence[float], epsg: int = 25832) -> None:
    """Uncompressed, pixel-interleaved, single-strip classic TIFF with the GeoTIFF tags the reader understands.
    data: [bands, rows, cols] or [rows, cols]; uint8 / uint16 / float32."""
    arr = np.asarray(data)
    if arr.ndim == 2:
        arr = arr[None]
    C, H, W = arr.shape
    fmt = {np.dtype(np.uint8): (1, 8), np.dtype(np.uint16): (1, 16), np.dtype(np.float32): (3, 32)}[arr.dtype]
    pix = np.ascontiguousarray(arr.transpose(1, 2, 0)).tobytes()
    a, _, c, _, e, f = (float(v) for v in transform[:6])
    entries = []   # (tag, type, count, payload bytes)

    def add(tag, typ, values):
        code = {3: "H", 4: "I", 12: "d"}[typ]
        entries.append((tag, typ, len(values), struct.pack("<" + str(len(values)) + code, *values)))

    add(256, 4, [W])
    add(257, 4, [H])
    add(258, 3, [fmt[1]] * C)
    add(259, 3, [1])
    add(262, 3, [2 if C >= 3 else 1])
    add(273, 4, [0])   # patched below
    add(277, 3, [C])
    add(278, 4, [H])
    add(279, 4, [len(pix)])
    add(284, 3, [1])
    if C > 3:
        add(338, 3, [0] * (C - 3))
    add(339, 3, [fmt[0]] * C)
    add(33550, 12, [a, -e, 0.0])
    add(33922, 12, [0.0, 0.0, 0.0, c, f, 0.0])
    add(34735, 3, [1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, int(epsg)])
    entries.sort(key=lambda t: t[0])
    n = len(entries)
    ifd_off = 8
    extra_off = ifd_off + 2 + n * 12 + 4
    extra = b""
    recs = []
    for tag, typ, cnt, payload in entries:
        if len(payload) <= 4:
            recs.append([tag, typ, cnt, payload.ljust(4, b"\0"), None])
        else:
            recs.append([tag, typ, cnt, None, len(extra)])
            extra += payload + (b"\0" if len(payload) % 2 else b"")
    data_off = extra_off + len(extra)
    out = bytearray(struct.pack("<2sHI", b"II", 42, ifd_off))
    out += struct.pack("<H", n)
    for tag, typ, cnt, inline, eoff in recs:
        if tag == 273:
            inline = struct.pack("<I", data_off)
        val = inline if inline is not None else struct.pack("<I", extra_off + eoff)
        out += struct.pack("<HHI", tag, typ, cnt) + val
    out += struct.pack("<I", 0)
    out += extra
    out += pix
    with open(path, "wb") as fh:
        fh.write(bytes(out))
