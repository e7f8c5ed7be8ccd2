"""GeoTIFF reader / writer for the tile loader of the prediction stage.

The reference reads rasters through rasterio/GDAL (``rasterio.open`` + ``rasterio.mask.mask(img, shapes, crop=True)``,
TreeDetection/prediction.py:61,164); neither is installed here. The tile loader needs windowed reads of classic or
BigTIFF rasters — strips or tiles, pixel-interleaved or planar, 8/16/32-bit integer or float samples, uncompressed or
DEFLATE (zlib) / LZW / PackBits (td_tiff_*_decode in libtreedet_hip.so), horizontal-differencing predictor — plus the
three geo tags (ModelPixelScale / ModelTiepoint / GeoKeyDirectory). JPEG-in-TIFF (compression 7, 8-bit, one or three bands)
is windowed too: a block's abbreviated stream + the JPEGTables tag form one JPEG stream, decoded by Pillow's libjpeg block by
block (GDAL does the same through libtiff). Other codecs (old-style JPEG, floating-point predictor, ...): whole image through Pillow.
"""
from __future__ import annotations

import math
import os
import struct
import threading
import zlib
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
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
        # GDAL_NODATA (tag 42113, ASCII): what rasterio reports as ``src.nodata`` (helpers.merge_images reads it)
        self.nodata: Optional[float] = None
        if 42113 in t:
            try:
                self.nodata = float(str(t[42113][0]).strip())
            except (ValueError, IndexError):
                self.nodata = None
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

    # -- pixel access ---------------------------------------------------------------------------------
    # The raster is a grid of blocks (strips = full-width blocks of RowsPerStrip rows; tiles = TileWidth x TileLength),
    # chunky (one block holds all bands, pixel-interleaved) or planar (one grid per band). A window read touches only
    # the blocks it overlaps; decoded blocks of compressed files are kept in a small LRU cache because neighbouring
    # tile windows (buffer overlap) share them. Uncompressed chunky files whose strips are contiguous skip all of
    # this: the window is a slice of one memory map.
    _CACHE_BYTES = 256 << 20

    def _setup_blocks(self) -> None:
        if getattr(self, "_blocks_ready", False):
            return
        t = self.tags
        H, W = self.height, self.width
        if 324 in t:
            self._bw, self._bh = int(t[322][0]), int(t[323][0])
            self._offs, self._counts = [int(v) for v in t[324]], [int(v) for v in t[325]]
            self._strips = False
        else:
            self._bw, self._bh = W, min(int(t.get(278, [H])[0]), H)
            self._offs = [int(v) for v in t[273]]
            self._counts = [int(v) for v in t[279]] if 279 in t else None
            self._strips = True
        self._nx = (W + self._bw - 1) // self._bw
        self._ny = (H + self._bh - 1) // self._bh
        self._predictor = int(t.get(317, [1])[0])
        self._mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        self._cache: "OrderedDict" = OrderedDict()
        self._cache_bytes = 0
        self._lock = threading.Lock()
        self._pool = None
        self._flat = None
        self._fd = None
        item = self.dtype.itemsize
        if self.compression == 1 and self.planar == 1 and self._strips:
            row_bytes = W * self.count * item
            offs = self._offs
            if all(offs[i + 1] - offs[i] == self._bh * row_bytes for i in range(len(offs) - 1)):
                self._flat = np.memmap(self.path, dtype=self.dtype, mode="r", offset=offs[0], shape=(H, W, self.count))
                if self.dtype.byteorder in ("=", "|") and os.path.getsize(self.path) >= offs[0] + H * row_bytes:
                    # windows are read with pread (td_read_window): no page fault per window row, no munmap of a touched
                    # mapping at close; the map stays for whole-raster reads and foreign byte orders
                    self._fd, self._flat_off = os.open(self.path, os.O_RDONLY), offs[0]
        if self.compression == 7 and self._jpeg_blocks_ok():
            tb = bytes(bytearray(int(v) for v in t.get(347, [])))
            self._jpeg_tables = tb if len(tb) >= 4 and tb[:2] == b"\xff\xd8" and tb[-2:] == b"\xff\xd9" else b""
        elif self.compression not in (1, 5, 8, 32946, 32773) or self._predictor not in (1, 2):
            self._pil_fallback()
        self._blocks_ready = True

    def _jpeg_blocks_ok(self) -> bool:
        """New-style JPEG blocks this reader decodes one by one: 8-bit chunky samples, grey or three bands (RGB or YCbCr)."""
        return (self.dtype == np.uint8 and self.planar == 1 and self.count in (1, 3) and self._predictor == 1
                and int(self.tags.get(262, [2 if self.count == 3 else 1])[0]) in (1, 2, 6))

    def _decode_jpeg_block(self, raw: bytes, rows: int) -> np.ndarray:
        """One strip / tile of a compression-7 raster → uint8 [rows, bw, bands]. TIFF Technical Note 2: the block is an abbreviated JPEG
        stream (SOI, frame and scan headers, entropy-coded data, EOI); the quantisation / Huffman tables it leaves out are in the
        JPEGTables tag (SOI, DQT / DHT segments, EOI) — tables minus its EOI + block minus its SOI is a complete stream. The colour
        space is NOT in the stream (no JFIF header): libtiff tells libjpeg from PhotometricInterpretation; here an Adobe APP14 segment
        (transform 0 = as stored, 1 = YCbCr) says the same thing to Pillow's libjpeg."""
        import io
        from PIL import Image
        data = bytes(raw)
        if data[:2] != b"\xff\xd8":
            raise ValueError(f"{self.path}: a JPEG block does not start with SOI")
        head = b"\xff\xd8"
        if self.count == 3 and b"JFIF\0" not in data[:32]:
            ycc = int(self.tags.get(262, [2])[0]) == 6
            head += b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00" + (b"\x01" if ycc else b"\x00")
        stream = head + self._jpeg_tables[2:-2] + data[2:]
        with Image.open(io.BytesIO(stream)) as im:
            if im.format != "JPEG" or im.mode not in ("L", "RGB"):
                raise ValueError(f"{self.path}: a JPEG block decodes to mode {im.mode}")
            arr = np.asarray(im)
        arr = arr[:, :, None] if arr.ndim == 2 else arr
        if arr.shape[0] < rows or arr.shape[1] != self._bw or arr.shape[2] != self.count:
            raise ValueError(f"{self.path}: a JPEG block decodes to {arr.shape}, expected at least ({rows}, {self._bw}, {self.count})")
        return arr[:rows]

    def _pil_fallback(self) -> None:
        """Codecs this reader does not implement (JPEG, floating-point predictor, ...): whole image through Pillow."""
        from PIL import Image
        Image.MAX_IMAGE_PIXELS = None
        arr = np.asarray(Image.open(self.path))
        arr = arr[:, :, None] if arr.ndim == 2 else arr
        if arr.shape[2] != self.count:
            raise ValueError(f"{self.path}: compression {self.compression} / predictor {self._predictor} is not supported "
                             f"for {self.count}-band rasters")
        self._flat = np.ascontiguousarray(arr)

    def close(self) -> None:
        """Releases the decode threads, the block cache and the file mapping (the object can still be re-used: they are
        rebuilt on the next read)."""
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)
        self._pool = None
        self._data = None
        self._blocks_ready = False
        fd = getattr(self, "_fd", None)
        if fd is not None:
            self._fd = None
            os.close(fd)
        for name in ("_cache", "_mm", "_flat"):
            if hasattr(self, name):
                setattr(self, name, None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            pool = getattr(self, "_pool", None)
            if pool is not None:
                pool.shutdown(wait=False)
            fd = getattr(self, "_fd", None)
            if fd is not None:
                self._fd = None
                os.close(fd)
        except Exception:
            pass

    def _block_rows(self, by: int) -> int:
        return min(self._bh, self.height - by * self._bh) if self._strips else self._bh

    def _decode_block(self, plane: int, by: int, bx: int) -> np.ndarray:
        """→ [rows, bw, bands-in-block] in native byte order, predictor undone."""
        idx = (plane * self._ny + by) * self._nx + bx
        rows = self._block_rows(by)
        cb = self.count if self.planar == 1 else 1
        nbytes = rows * self._bw * cb * self.dtype.itemsize
        off = self._offs[idx]
        cnt = self._counts[idx] if self._counts is not None else nbytes
        raw = self._mm[off:off + cnt]
        comp = self.compression
        if comp == 7:
            return self._decode_jpeg_block(raw, rows)
        if comp == 1:
            buf = np.asarray(raw[:nbytes])
        elif comp in (8, 32946):
            buf = np.frombuffer(zlib.decompress(raw), dtype=np.uint8)
        else:
            from . import _lib
            lib = _lib.load()
            src = np.ascontiguousarray(raw)
            buf = np.empty(nbytes, dtype=np.uint8)
            fn = lib.td_tiff_lzw_decode if comp == 5 else lib.td_tiff_packbits_decode
            n = fn(src.ctypes.data, src.size, buf.ctypes.data, nbytes)
            _lib.check(n, "tiff block decode")
            buf = buf[:n]
        if buf.size < nbytes:
            raise ValueError(f"{self.path}: block ({plane},{by},{bx}) decodes to {buf.size} bytes, expected {nbytes}")
        blk = np.frombuffer(buf[:nbytes], dtype=self.dtype).reshape(rows, self._bw, cb)
        if self.dtype.byteorder not in ("=", "|"):     # file byte order differs from the host's
            blk = blk.astype(self.dtype.newbyteorder("="))
        if self._predictor == 2:                      # horizontal differencing per sample, modulo the sample width
            from . import _lib
            if not blk.flags.writeable or not blk.flags.c_contiguous:
                blk = np.array(blk)
            _lib.check(_lib.load().td_tiff_unpredict(blk.ctypes.data, rows, self._bw, cb, blk.dtype.itemsize), "td_tiff_unpredict")
        return blk

    def _get_block(self, key) -> np.ndarray:
        with self._lock:
            blk = self._cache.get(key)
            if blk is not None:
                self._cache.move_to_end(key)
                return blk
        blk = self._decode_block(*key)
        if self.compression != 1 or self._predictor != 1:
            with self._lock:
                if key not in self._cache:
                    self._cache[key] = blk
                    self._cache_bytes += blk.nbytes
                    while self._cache_bytes > self._CACHE_BYTES and len(self._cache) > 1:
                        _, old = self._cache.popitem(last=False)
                        self._cache_bytes -= old.nbytes
        return blk

    def _window_hwc(self, r0: int, c0: int, h: int, w: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Pixels [r0:r0+h, c0:c0+w] of every band as [h, w, bands] (into ``out`` when given)."""
        self._setup_blocks()
        if self._flat is not None and getattr(self, "_fd", None) is not None:
            if out is None:
                out = np.empty((h, w, self.count), dtype=self.dtype)
            if out.flags.c_contiguous and out.dtype == self.dtype:
                from . import _lib
                px = self.count * self.dtype.itemsize
                n = _lib.load().td_read_window(self._fd, self._flat_off + (r0 * self.width + c0) * px, self.width * px, w * px, h,
                                               out.ctypes.data)
                _lib.check(n, "td_read_window")
                return out
        if self._flat is not None:
            src = self._flat[r0:r0 + h, c0:c0 + w, :]
            if out is None:
                return np.ascontiguousarray(src)
            np.copyto(out, src)
            return out
        if out is None:
            out = np.empty((h, w, self.count), dtype=self.dtype.newbyteorder("="))
        by0, by1 = r0 // self._bh, (r0 + h - 1) // self._bh
        bx0, bx1 = c0 // self._bw, (c0 + w - 1) // self._bw
        planes = self.count if self.planar == 2 else 1
        keys = [(p, by, bx) for p in range(planes) for by in range(by0, by1 + 1) for bx in range(bx0, bx1 + 1)]
        if len(keys) > 1 and self.compression != 1:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=min(8, max(2, len(os.sched_getaffinity(0)))))
            # zlib / the C decoders release the GIL; a few blocks per task keep the executor overhead small
            per = max(1, (len(keys) + 15) // 16)
            chunks = [keys[i:i + per] for i in range(0, len(keys), per)]
            blocks = [b for part in self._pool.map(lambda ks: [self._get_block(k) for k in ks], chunks) for b in part]
        else:
            blocks = [self._get_block(k) for k in keys]
        for (p, by, bx), blk in zip(keys, blocks):
            br0, bc0 = by * self._bh, bx * self._bw
            rs, re = max(r0, br0), min(r0 + h, br0 + blk.shape[0])
            cs, ce = max(c0, bc0), min(c0 + w, bc0 + self._bw)
            piece = blk[rs - br0:re - br0, cs - bc0:ce - bc0, :]
            if self.planar == 2:
                out[rs - r0:re - r0, cs - c0:ce - c0, p] = piece[:, :, 0]
            else:
                out[rs - r0:re - r0, cs - c0:ce - c0, :] = piece
        return out

    # -- compressed raster → HBM (tiffdecode.hip) --------------------------------------------------------------------------
    def device_decodable(self) -> bool:
        """True when the raster's blocks can be decoded on the GPU: LZW or DEFLATE (zlib) strips or tiles of pixel-interleaved
        uint8 samples (<= 4 per pixel), predictor 1 or 2. Everything else keeps the host reader."""
        self._setup_blocks()
        return (self.compression in (5, 8, 32946) and self.planar == 1 and self.dtype == np.uint8 and 1 <= self.count <= 4
                and self._predictor in (1, 2) and self._counts is not None and self._pil is None
                and self._bw * self._bh * self.count < (1 << 31))

    def decode_to_device(self, device, stream=None, pinned=None, pool=None):
        """The whole raster decoded in HBM: the compressed blocks are read as they lie in the file (one pread of the span that
        holds them, into pinned memory), copied to the device once, decoded one wave per block (td_tiff_lzw_decode_dev /
        td_tiff_inflate_dev) and
        laid out as [height, width, bands] uint8 with predictor 2 undone (td_tiff_blocks_to_image_dev). → (image tensor,
        check) where ``check()`` waits for the kernels and raises ValueError when a block did not decode to its size (the
        caller then falls back to the host reader). Everything is enqueued on ``stream`` (default: the current one). ``pinned``:
        a one-element list holding a pinned uint8 tensor to read the file into (grown and put back when too small — pinning
        hundreds of MB per image costs as much as reading them); ``pool``: threads the file read is spread over."""
        import torch
        from . import _lib
        if not self.device_decodable():
            raise ValueError(f"{self.path}: not decodable on the device (compression {self.compression}, {self.dtype}, {self.count} bands)")
        lib = _lib.load()
        dev = torch.device(device)
        offs = np.asarray(self._offs, dtype=np.int64)
        cnts = np.asarray(self._counts, dtype=np.int64)
        nb = self._nx * self._ny
        assert len(offs) == nb == len(cnts)
        lo, hi = int(offs.min()), int((offs + cnts).max())
        span = hi - lo
        if pinned is not None and pinned and pinned[0] is not None and pinned[0].numel() >= span + 16:
            pin = pinned[0]
        else:
            pin = torch.empty((span + 16 + (span >> 3),), dtype=torch.uint8, pin_memory=True)
            if pinned is not None:
                pinned[:] = [pin]
        buf = pin.numpy()
        fd = os.open(self.path, os.O_RDONLY)

        def read_piece(pr):
            got = 0
            view = memoryview(buf)[pr[0]:pr[0] + pr[1]]
            while got < pr[1]:
                n = os.preadv(fd, [view[got:]], lo + pr[0] + got)
                if n <= 0:
                    raise ValueError(f"{self.path}: file ends inside its block data")
                got += n
        try:
            piece = 8 << 20
            parts = [(p, min(piece, span - p)) for p in range(0, span, piece)]
            if pool is not None and len(parts) > 1:
                list(pool.map(read_piece, parts))
            else:
                for pr in parts:
                    read_piece(pr)
        finally:
            os.close(fd)
        buf[span:] = 0
        block_cap = self._bw * self._bh * self.count
        expect = np.array([self._block_rows(by) * self._bw * self.count for by in range(self._ny) for _ in range(self._nx)], dtype=np.int64)
        ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
        with torch.cuda.device(dev), ctx:
            st = _lib.stream_ptr()
            comp = pin[:span + 16].to(dev, non_blocking=True)
            meta = torch.from_numpy(np.stack([offs - lo, cnts])).to(dev, non_blocking=True)
            blocks = torch.empty((nb, block_cap), dtype=torch.uint8, device=dev)
            decoded = torch.empty((nb,), dtype=torch.int64, device=dev)
            status = torch.empty((2 * nb + 1,), dtype=torch.int32, device=dev)      # [nb] status + scratch of the wide-table pass
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record()
            fn, fname = (lib.td_tiff_lzw_decode_dev, "td_tiff_lzw_decode_dev") if self.compression == 5 else (lib.td_tiff_inflate_dev, "td_tiff_inflate_dev")
            _lib.check(fn(comp.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(), nb, blocks.data_ptr(), block_cap,
                          decoded.data_ptr(), status.data_ptr(), st), fname)
            image = torch.empty((self.height, self.width, self.count), dtype=torch.uint8, device=dev)
            _lib.check(lib.td_tiff_blocks_to_image_dev(blocks.data_ptr(), block_cap, self._bw, self._bh, self._nx, self._ny, self.count,
                                                       self._predictor, image.data_ptr(), self.width, self.height, st), "td_tiff_blocks_to_image_dev")
            k1.record()
            done = torch.cuda.Event()
            done.record()
            dec_h = torch.empty((nb,), dtype=torch.int64, pin_memory=True)
            st_h = torch.empty((nb,), dtype=torch.int32, pin_memory=True)
            dec_h.copy_(decoded, non_blocking=True)
            st_h.copy_(status[:nb], non_blocking=True)
            copied = torch.cuda.Event(blocking=True)
            copied.record()
        keep = [pin, comp, meta, blocks, decoded, status]     # alive until check() has run: the kernels read them

        def check():
            copied.synchronize()
            check.kernel_ms = k0.elapsed_time(k1)          # the two decode launches + the scatter / predictor kernel
            keep.clear()
            produced = dec_h.numpy() & 0xffffffff
            check.slow_codes = int((dec_h.numpy() >> 32).sum())     # LZW: strings copied through memory (sources older than the LDS ring)
            bad = np.nonzero((st_h.numpy() != 0) | (produced != expect))[0]
            if bad.size:
                b = int(bad[0])
                raise ValueError(f"{self.path}: block {b} decodes to {int(produced[b])} bytes (status {int(st_h[b])}), expected {int(expect[b])}")
            return image
        check.event = done
        check.compressed_bytes = span
        return image, check

    def device_uploadable(self) -> bool:
        """True when the raster's pixels lie in the file as ONE dense [rows, cols, bands] uint8 array (uncompressed contiguous
        strips, pixel-interleaved): it can be copied to the GPU in large sequential pieces and its tile windows cut there."""
        self._setup_blocks()
        return self._flat is not None and getattr(self, "_fd", None) is not None and self.dtype == np.uint8 and self._pil is None

    def upload_to_device(self, device, stream=None, staging=None, pool=None, piece: int = 4 << 20):
        """The whole uncompressed raster in HBM: the file's pixel bytes are read in large sequential pieces (pread of ``piece``
        bytes, several side by side on ``pool``) into pinned staging buffers and copied to the device as they arrive — per tile
        one memcpy of its bytes out of the page cache, no system call per window row and no per-batch H2D of windows later.
        ``staging``: list of >= 2 pinned uint8 tensors to rotate through (created here when None). → (image tensor [rows,
        cols, bands] uint8, check) as :meth:`decode_to_device`."""
        import torch
        if not self.device_uploadable():
            raise ValueError(f"{self.path}: not one dense uint8 array in the file")
        dev = torch.device(device)
        total = self.height * self.width * self.count
        if staging is None:
            staging = [torch.empty((4 * piece,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        cap = min(t.numel() for t in staging)
        ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
        fd, base = self._fd, self._flat_off

        def read_into(view: memoryview, off: int) -> None:
            got = 0
            while got < len(view):
                n = os.preadv(fd, [view[got:]], base + off + got)
                if n <= 0:
                    raise ValueError(f"{self.path}: file ends inside its pixel data")
                got += n

        with torch.cuda.device(dev), ctx:
            image = torch.empty((self.height, self.width, self.count), dtype=torch.uint8, device=dev)
            flat = image.view(-1)
            events = [None] * len(staging)
            k = 0
            for o in range(0, total, cap):
                n = min(cap, total - o)
                buf = staging[k % len(staging)]
                if events[k % len(staging)] is not None:
                    events[k % len(staging)].synchronize()          # the copy that last used this staging buffer has finished
                mv = memoryview(buf.numpy())
                parts = [(p, min(piece, n - p)) for p in range(0, n, piece)]

                def one(pr, mv=mv, o=o):
                    read_into(mv[pr[0]:pr[0] + pr[1]], o + pr[0])
                    return pr
                # every piece goes to the device as soon as it (and the pieces before it) has been read: copies of a few MB keep
                # the DMA queue short for the small result copies of the running forwards (64-MB copies delayed them by a millisecond)
                for p0, pn in (pool.map(one, parts) if pool is not None and len(parts) > 1 else map(one, parts)):
                    flat[o + p0:o + p0 + pn].copy_(buf[p0:p0 + pn], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                events[k % len(staging)] = ev
                k += 1
            done = torch.cuda.Event(blocking=True)
            done.record()

        def check():
            done.synchronize()
            return image
        check.event = done
        check.compressed_bytes = total
        return image, check

    def _load(self) -> np.ndarray:
        """Whole raster as [bands, rows, cols]."""
        if self._data is None:
            self._data = self._window_hwc(0, 0, self.height, self.width).transpose(2, 0, 1)
        return self._data

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

    def _mask_outside(self, hwc: np.ndarray, bounds, c0: int, r0: int) -> None:
        """rasterio.mask semantics: pixels whose centre lies outside the bbox are set to 0."""
        h, w = hwc.shape[:2]
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

    def outside_mask(self, bounds, c0: int, r0: int, w: int, h: int):
        """The columns / rows of window (c0, r0, w, h) that ``_mask_outside`` keeps, as two boolean arrays — or None when it
        keeps everything (the usual case: tile bounds lie on pixel edges)."""
        a, _, c, _, e, f = self.transform
        minx, miny, maxx, maxy = (float(v) for v in bounds[:4])
        xs = c + a * (np.arange(c0, c0 + w) + 0.5)
        ys = f + e * (np.arange(r0, r0 + h) + 0.5)
        okx = (xs >= minx) & (xs <= maxx)
        oky = (ys >= miny) & (ys <= maxy)
        return None if okx.all() and oky.all() else (okx, oky)

    def read_bounds_hwc(self, bounds: Sequence[float], out: Optional[np.ndarray] = None, out_off: int = 0) -> np.ndarray:
        """``rasterio.mask.mask(img, [bbox], crop=True)[0]`` as pixel-interleaved [rows, cols, bands] — the layout the
        device resize consumes. With ``out`` (a flat array of the raster's dtype, e.g. pinned staging memory) the window
        is written at ``out_off`` and the returned array is a view of it. Raises ValueError when the bbox does not
        overlap the raster."""
        c0, r0, w, h = self.window_of_bounds(bounds)
        if w <= 0 or h <= 0:
            raise ValueError("Input shapes do not overlap raster.")
        dst = None if out is None else out[out_off:out_off + h * w * self.count].reshape(h, w, self.count)
        hwc = self._window_hwc(r0, c0, h, w, dst)
        self._mask_outside(hwc, bounds, c0, r0)
        return hwc

    def read_windows_flat(self, windows, out: np.ndarray, out_offs, threads: int = 8) -> bool:
        """The windows (col_off, row_off, width, height) of a whole batch of an uncompressed, pixel-interleaved raster in ONE library
        call (td_read_windows: pread per window row, row bands spread over ``threads`` C threads) into the flat array ``out`` at
        ``out_offs`` (elements). → False when the raster is not of that kind (the caller reads window by window)."""
        self._setup_blocks()
        if self._flat is None or getattr(self, "_fd", None) is None or out.dtype != self.dtype or not out.flags.c_contiguous:
            return False
        from . import _lib
        px = self.count * self.dtype.itemsize
        n = len(windows)
        foff = np.array([self._flat_off + (r0 * self.width + c0) * px for c0, r0, w, h in windows], dtype=np.int64)
        rbytes = np.array([w * px for c0, r0, w, h in windows], dtype=np.int64)
        rows = np.array([h for c0, r0, w, h in windows], dtype=np.int64)
        doff = np.asarray(out_offs, dtype=np.int64) * self.dtype.itemsize
        got = _lib.load().td_read_windows(self._fd, n, foff.ctypes.data, self.width * px, rbytes.ctypes.data, rows.ctypes.data, out.ctypes.data,
                                          doff.ctypes.data, int(threads))
        _lib.check(got, "td_read_windows")
        return True

    def read_bounds(self, bounds: Sequence[float]) -> np.ndarray:
        """Same window as [bands, rows, cols] (rasterio's axis order)."""
        return np.ascontiguousarray(self.read_bounds_hwc(bounds).transpose(2, 0, 1))


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _jpeg_block(blk: np.ndarray, quality: int = 90) -> bytes:
    """One strip / tile as a COMPLETE JPEG stream (its own tables: the JPEGTables tag is optional, TIFF Technical Note 2), YCbCr 4:2:0
    for three bands — Pillow's libjpeg encoder. Lossy: a test fixture for the windowed JPEG reader, not an archive format."""
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(blk[:, :, 0] if blk.shape[2] == 1 else blk).save(buf, "JPEG", quality=quality, subsampling=2 if blk.shape[2] == 3 else 0)
    return buf.getvalue()


def write_geotiff(path: str, data: np.ndarray, transform: Sequence[float], epsg: int = 25832, *,
                  tile: Optional[Tuple[int, int]] = None, rows_per_strip: Optional[int] = None,
                  compression: Optional[str] = None, predictor: int = 1, planar: bool = False,
                  nodata: Optional[float] = None) -> None:
    """Classic little-endian TIFF with the GeoTIFF tags the reader understands. data: [bands, rows, cols] or
    [rows, cols]; uint8 / uint16 / float32. Defaults: one uncompressed pixel-interleaved strip (what the tile reader
    maps without copying). Options: ``tile=(tile_rows, tile_cols)`` (multiples of 16) or ``rows_per_strip``,
    ``compression`` None / "deflate" / "lzw" (td_tiff_lzw_encode, blocks encoded on host threads) / "jpeg" (lossy; Pillow's encoder per
    block), ``predictor`` 1 / 2
    (horizontal differencing, integer samples), ``planar`` (one block grid per band)."""
    arr = np.asarray(data)
    if arr.ndim == 2:
        arr = arr[None]
    C, H, W = arr.shape
    fmt = {np.dtype(np.uint8): (1, 8), np.dtype(np.uint16): (1, 16), np.dtype(np.float32): (3, 32)}[arr.dtype]
    if compression not in (None, "deflate", "lzw", "jpeg"):
        raise ValueError("compression must be None, 'deflate', 'lzw' or 'jpeg'")
    if predictor not in (1, 2) or (predictor == 2 and fmt[0] != 1):
        raise ValueError("predictor 2 needs integer samples")
    hwc = np.ascontiguousarray(arr.transpose(1, 2, 0)).astype(arr.dtype.newbyteorder("<"), copy=False)
    if tile:
        bh, bw = int(tile[0]), int(tile[1])
        if bh % 16 or bw % 16:
            raise ValueError("tile sides must be multiples of 16")
    else:
        bh, bw = int(rows_per_strip or H), W
    if compression == "jpeg" and (planar or predictor != 1 or arr.dtype != np.uint8 or C not in (1, 3) or bh % 16):
        raise ValueError("jpeg: uint8 grey / three-band chunky rasters, no predictor, block heights in multiples of 16")
    ny, nx = (H + bh - 1) // bh, (W + bw - 1) // bw
    blocks = []
    for p in range(C if planar else 1):
        src = hwc[:, :, p:p + 1] if planar else hwc
        for by in range(ny):
            for bx in range(nx):
                rows = bh if tile else min(bh, H - by * bh)
                blk = np.zeros((rows, bw, src.shape[2]), dtype=hwc.dtype)
                piece = src[by * bh:by * bh + rows, bx * bw:(bx + 1) * bw]
                blk[:piece.shape[0], :piece.shape[1]] = piece
                if predictor == 2:
                    blk[:, 1:] = blk[:, 1:] - blk[:, :-1]          # modulo the sample width
                if compression == "jpeg":
                    blocks.append(_jpeg_block(blk))
                    continue
                raw = blk.tobytes()
                blocks.append(zlib.compress(raw, 6) if compression == "deflate" else raw)
    if compression == "lzw":
        from . import _lib
        lib = _lib.load()

        def enc(raw: bytes) -> bytes:
            src = np.frombuffer(raw, dtype=np.uint8)
            dst = np.empty(len(raw) * 3 // 2 + 64, dtype=np.uint8)          # 12-bit codes for single bytes at worst
            n = lib.td_tiff_lzw_encode(src.ctypes.data, src.size, dst.ctypes.data, dst.size)
            _lib.check(n, "td_tiff_lzw_encode")
            return dst[:n].tobytes()
        with ThreadPoolExecutor(max_workers=max(1, min(16, len(os.sched_getaffinity(0))))) as ex:
            blocks = list(ex.map(enc, blocks))
    a, _, c, _, e, f = (float(v) for v in transform[:6])
    entries = []   # (tag, type, count, payload bytes)

    def add(tag, typ, values):
        if typ == 2:                           # ASCII, NUL-terminated
            raw = values.encode("latin1") + b"\0"
            entries.append((tag, typ, len(raw), raw))
            return
        code = {3: "H", 4: "I", 12: "d"}[typ]
        entries.append((tag, typ, len(values), struct.pack("<" + str(len(values)) + code, *values)))

    add(256, 4, [W])
    add(257, 4, [H])
    add(258, 3, [fmt[1]] * C)
    add(259, 3, [{"deflate": 8, "lzw": 5, "jpeg": 7}.get(compression, 1)])
    add(262, 3, [(6 if compression == "jpeg" else 2) if C >= 3 else 1])
    add(277, 3, [C])
    add(284, 3, [2 if planar else 1])
    if predictor == 2:
        add(317, 3, [2])
    if tile:
        add(322, 4, [bw])
        add(323, 4, [bh])
        add(324, 4, [0] * len(blocks))   # offsets, patched below
        add(325, 4, [len(b) for b in blocks])
    else:
        add(273, 4, [0] * len(blocks))   # offsets, patched below
        add(278, 4, [bh])
        add(279, 4, [len(b) for b in blocks])
    if C > 3:
        add(338, 3, [0] * (C - 3))
    add(339, 3, [fmt[0]] * C)
    if compression == "jpeg" and C == 3:
        add(530, 3, [2, 2])                    # YCbCrSubSampling: 4:2:0, what _jpeg_block encodes
    add(33550, 12, [a, -e, 0.0])
    add(33922, 12, [0.0, 0.0, 0.0, c, f, 0.0])
    add(34735, 3, [1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, int(epsg)])
    if nodata is not None:
        add(42113, 2, repr(float(nodata)))
    entries.sort(key=lambda t: t[0])
    n = len(entries)
    ifd_off = 8
    extra_off = ifd_off + 2 + n * 12 + 4
    extra_len = sum(len(pl) + (len(pl) % 2) for _, _, _, pl in entries if len(pl) > 4)
    data_off = extra_off + extra_len
    offsets, pos = [], data_off
    for b in blocks:
        offsets.append(pos)
        pos += len(b) + (len(b) % 2)
    if pos >= 1 << 32:
        raise ValueError("raster too large for a classic TIFF")
    off_tag = 324 if tile else 273
    out = bytearray(struct.pack("<2sHI", b"II", 42, ifd_off))
    out += struct.pack("<H", n)
    extra = b""
    for tag, typ, cnt, payload in entries:
        if tag == off_tag:
            payload = struct.pack("<" + str(cnt) + "I", *offsets)
        if len(payload) <= 4:
            val = payload.ljust(4, b"\0")
        else:
            val = struct.pack("<I", extra_off + len(extra))
            extra += payload + (b"\0" if len(payload) % 2 else b"")
        out += struct.pack("<HHI", tag, typ, cnt) + val
    out += struct.pack("<I", 0)
    out += extra
    with open(path, "wb") as fh:
        fh.write(bytes(out))
        for b in blocks:
            fh.write(b)
            if len(b) % 2:
                fh.write(b"\0")
