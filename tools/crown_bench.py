"""td_crown_stats on a realistic image: 5000 x 5000 NDVI raster (0.2 m), 1000 x 1000 nDSM (1 m), 20 000 crowns of
1.5–6 m radius; times the kernel (HIP events) and the brute-force restatement of the reference on a few crowns."""
import sys, time
sys.path.insert(0, ".")
import ctypes as C
import numpy as np, torch
from treedetection_amd import _lib
from treedetection_amd import postprocessing as P
from oracle import postprocess_ref as O

rng = np.random.default_rng(0)
n = 20000
t = (0.2, 0.0, 412000.0, 0.0, -0.2, 5319000.0)
nt = (1.0, 0.0, 412000.0, 0.0, -1.0, 5319000.0)
ndvi = rng.uniform(-0.2, 0.9, (5000, 5000)).astype(np.float32)
ndsm = rng.uniform(0, 30, (1000, 1000)).astype(np.float32)
cx, cy, r = rng.uniform(412010, 412990, n), rng.uniform(5318010, 5318990, n), rng.uniform(1.5, 6.0, n)
circles = np.stack([cx, cy, r], axis=1).astype(np.float32)
lib = _lib.load()
dev = torch.device("cuda", 0)
for name, raster, tr, mode in (("ndvi 5000x5000", ndvi, t, 1), ("ndsm 1000x1000", ndsm, nt, 0)):
    d_r = torch.from_numpy(raster).to(dev); d_c = torch.from_numpy(circles).to(dev)
    d_o = torch.empty((n, 4 if mode else 3), dtype=torch.float32, device=dev)
    trc = (C.c_double * 6)(*tr); win = (C.c_int32 * 4)(0, 0, raster.shape[0] - 1, raster.shape[1] - 1)
    call = lambda: _lib.check(lib.td_crown_stats(d_r.data_ptr(), raster.shape[0], raster.shape[1], trc, win, d_c.data_ptr(), n, mode, 1.0, d_o.data_ptr(), _lib.stream_ptr()), "x")
    call(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): call()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    px = float(np.sum((2 * r / abs(tr[0]) + 3) ** 2))
    print(f"{name}: {ms*1e3:.0f} us for {n} crowns ({px/1e6:.1f} M pixel tests, {px*4*(2 if mode else 1)/ms/1e6:.1f} GB/s of raster reads)")
# brute force on the CPU for 4 crowns of the NDVI raster (the reference's O(crowns x pixels) formulation)
k = 4
pxs = [np.array([c[0] - c[2], c[0] + c[2]], np.float32) for c in circles[:k]]
pys = [np.array([c[1], c[1]], np.float32) for c in circles[:k]]
t0 = time.time()
O.ndvi_within(pxs, pys, ndvi, t + (0, 0, 1), (412000.0, 5318000.0, 413000.0, 5319000.0), 1.0)
dt = time.time() - t0
print(f"brute-force numpy restatement: {dt/k*1e3:.0f} ms per crown -> {dt/k*n:.0f} s for {n} crowns on one host core")
