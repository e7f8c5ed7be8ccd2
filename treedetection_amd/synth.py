"""Seeded synthetic tile stream of SURVEY.md §8d (no network, no example RGB raster in the reference checkout).

Tile i: ``rng = np.random.default_rng(1234 + i)``; RGB uint8 [S,S,3] = clip(low-frequency field (6 random 2-D
cosines, amplitude 60) + 40 "crowns" (Gaussian blobs, sigma 8..40 px, green-dominant) + N(0,8) noise + 110);
nDSM float32 [S,S] = blob heights 3..30 m on a 0 m ground (the reference's nDSM raster statistics: min 0, p99 ≈ 26).
The nDSM plane travels with the tile as a side band (it is only consumed by post-processing,
TreeDetection/postprocessing.py:781-789); the network input is the 3 RGB bands (prediction.py:166).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

SEED0 = 1234


def make_tile(i: int, size: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(SEED0 + i)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float32), np.arange(size, dtype=np.float32), indexing="ij")
    field = np.zeros((size, size, 3), dtype=np.float32)
    for _ in range(6):
        fx, fy = rng.uniform(-0.02, 0.02, 2).astype(np.float32)
        ph = np.float32(rng.uniform(0, 2 * np.pi))
        amp = rng.uniform(-10, 10, 3).astype(np.float32)
        field += np.cos(fx * xx + fy * yy + ph)[..., None] * amp
    ndsm = np.zeros((size, size), dtype=np.float32)
    for _ in range(40):
        cy, cx = rng.uniform(0, size, 2)
        sg = rng.uniform(8, 40)
        r = int(3 * sg) + 1
        y0, y1 = max(int(cy) - r, 0), min(int(cy) + r + 1, size)
        x0, x1 = max(int(cx) - r, 0), min(int(cx) + r + 1, size)
        if y0 >= y1 or x0 >= x1:
            continue
        blob = np.exp(-(((yy[y0:y1, x0:x1] - np.float32(cy)) ** 2 + (xx[y0:y1, x0:x1] - np.float32(cx)) ** 2)
                        / np.float32(2 * sg * sg))).astype(np.float32)
        col = np.array([rng.uniform(-40, 10), rng.uniform(20, 70), rng.uniform(-50, 0)], dtype=np.float32)
        field[y0:y1, x0:x1] += blob[..., None] * col
        ndsm[y0:y1, x0:x1] = np.maximum(ndsm[y0:y1, x0:x1], blob * np.float32(rng.uniform(3, 30)))
    field += rng.normal(0, 8, field.shape).astype(np.float32) + np.float32(110)
    rgb = np.clip(np.rint(field), 0, 255).astype(np.uint8)
    return rgb, ndsm


def tile_crowns(i: int, size: int = 1000):
    """(cx, cy, sigma) of tile i's crowns in tile pixels, in generation order — the same random draws as :func:`make_tile`
    (test fixtures that need to know where the crowns are: tests/trained_heads.py)."""
    rng = np.random.default_rng(SEED0 + i)
    for _ in range(6):
        rng.uniform(-0.02, 0.02, 2)
        rng.uniform(0, 2 * np.pi)
        rng.uniform(-10, 10, 3)
    out = []
    for _ in range(40):
        cy, cx = rng.uniform(0, size, 2)
        sg = rng.uniform(8, 40)
        r = int(3 * sg) + 1
        if max(int(cy) - r, 0) >= min(int(cy) + r + 1, size) or max(int(cx) - r, 0) >= min(int(cx) + r + 1, size):
            continue
        rng.uniform(-40, 10), rng.uniform(20, 70), rng.uniform(-50, 0)
        rng.uniform(3, 30)
        out.append((float(cx), float(cy), float(sg)))
    return out


def make_stream(n: int, size: int = 1000, distinct: int | None = None):
    """n tiles (RGB uint8 [n,S,S,3], nDSM float32 [n,S,S]); only ``distinct`` different seeds are generated and
    cycled when given (start-up time of the bench)."""
    d = n if distinct is None else min(distinct, n)
    tiles = [make_tile(i, size) for i in range(d)]
    rgb = np.stack([tiles[i % d][0] for i in range(n)])
    ndsm = np.stack([tiles[i % d][1] for i in range(n)])
    return rgb, ndsm
