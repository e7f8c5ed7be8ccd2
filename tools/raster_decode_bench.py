"""Decode rate of LZW / DEFLATE rasters on the GPU (tiffdecode.hip): python tools/raster_decode_bench.py [codec=lzw|deflate] [side=9000] [tile=256|strip=N]
[predictor=2] [data=tiles|noise|flat]
Raster side x side x 4 uint8 (default 9000: the 400 windows of 450 x 450 px the reference cuts from one image, twice over); prints per
call file → pinned → device → decoded raster in HBM, and the kernels alone (HIP events). Under rocprofv3 --kernel-trace --stats the
per-kernel durations land in profiles/r06_decode_kernel_stats.csv."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from treedetection_amd.geotiff import GeoTiff, write_geotiff      # noqa: E402
from treedetection_amd.synth import make_tile                      # noqa: E402

args = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
side, pred, data, codec = int(args.get("side", 9000)), int(args.get("predictor", 2)), args.get("data", "tiles"), args.get("codec", "lzw")
kw = {"rows_per_strip": int(args["strip"])} if "strip" in args else {"tile": (int(args.get("tile", 256)),) * 2}
base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
path = os.path.join(base, f"td_lzwbench_{os.getpid()}.tif")
rng = np.random.default_rng(0)
if data == "noise":
    img = rng.integers(0, 256, (4, side, side), dtype=np.uint8)
elif data == "flat":
    img = np.full((4, side, side), 77, np.uint8)
else:
    n = -(-side // 1000)
    tiles = [make_tile(i, 1000)[0] for i in range(min(16, n * n))]
    img = np.zeros((4, n * 1000, n * 1000), np.uint8)
    for r in range(n):
        for c in range(n):
            t = tiles[(r * n + c) % len(tiles)]
            img[:3, r * 1000:(r + 1) * 1000, c * 1000:(c + 1) * 1000] = t.transpose(2, 0, 1)
            img[3, r * 1000:(r + 1) * 1000, c * 1000:(c + 1) * 1000] = t[..., 1]
    img = np.ascontiguousarray(img[:, :side, :side])
try:
    t0 = time.perf_counter()
    write_geotiff(path, img, (0.2, 0, 412000.0, 0, -0.2, 5318000.0 + side * 0.2), 25832, compression=codec, predictor=pred, **kw)
    t_enc = time.perf_counter() - t0
    g = GeoTiff(path)
    g._setup_blocks()
    times, ktimes = [], []
    from concurrent.futures import ThreadPoolExecutor
    pinned, pool = [None], ThreadPoolExecutor(max_workers=8)
    for k in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        image, check = g.decode_to_device("cuda:0", None, pinned, pool)
        got = check()
        times.append(time.perf_counter() - t0)
        ktimes.append(check.kernel_ms)
        slow = getattr(check, "slow_codes", 0)
        if k == 0:
            assert np.array_equal(got.cpu().numpy().transpose(2, 0, 1), img), "decoded raster differs from what was written"
        del image, got
    raw = img.nbytes
    best = min(times[1:])
    print(json.dumps({"codec": codec, "raster": f"{side}x{side}x4", "layout": kw, "predictor": pred, "data": data, "blocks": g._nx * g._ny, "raw_mb": raw / 1e6,
                      "file_mb": os.path.getsize(path) / 1e6, "ratio": raw / os.path.getsize(path), "encode_s": round(t_enc, 2),
                      "decode_ms": [round(t * 1e3, 1) for t in times], "kernel_ms": [round(t, 1) for t in ktimes],
                      "kernel_gb_per_s": raw / (min(ktimes) * 1e-3) / 1e9, "strings_through_memory": slow, "decode_gb_per_s": raw / best / 1e9,
                      "windows_450x450x4_per_s": raw / best / (450 * 450 * 4)}))
finally:
    os.unlink(path)
