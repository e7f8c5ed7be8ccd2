"""Can the GPU read tile windows straight out of the page cache? mmap a raster on tmpfs, hipHostRegister the mapping and copy
windows with hipMemcpy2DAsync (no CPU copy at all): python tools/probes/hostregister_probe.py [MB]. Prints what the runtime says."""
import ctypes as C
import mmap
import os
import sys
import tempfile
import time

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W = 10000 * 4
H = mb * (1 << 20) // W
path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir(), f"td_hr_{os.getpid()}.bin")
a = np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)
a.tofile(path)
try:
    fd = os.open(path, os.O_RDONLY)
    for prot, flags, name in ((mmap.PROT_READ, mmap.MAP_SHARED, "MAP_SHARED read-only"), (mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE, "MAP_PRIVATE")):
        m = mmap.mmap(fd, H * W, flags=flags, prot=prot)
        buf = np.frombuffer(m, dtype=np.uint8)
        addr = buf.ctypes.data
        for fl, fname in ((0, "default"), (8, "hipHostRegisterReadOnly"), (2, "mapped")):
            t0 = time.perf_counter()
            st = hip.hipHostRegister(C.c_void_p(addr), C.c_size_t(H * W), C.c_uint(fl))
            dt = time.perf_counter() - t0
            print(f"{name}, flags {fname}: hipHostRegister -> {st} in {dt * 1e3:.1f} ms")
            if st == 0:
                dst = torch.empty((1000, 4000), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 50
                for k in range(n):
                    off = (k * 37 % (H - 1000)) * W + (k * 4000) % (W - 4000)
                    r = hip.hipMemcpy2DAsync(C.c_void_p(dst.data_ptr()), C.c_size_t(4000), C.c_void_p(addr + off), C.c_size_t(W), C.c_size_t(4000),
                                             C.c_size_t(1000), C.c_int(1), C.c_void_p(0))
                    assert r == 0, r
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                k = n - 1
                off = (k * 37 % (H - 1000)) * W + (k * 4000) % (W - 4000)
                ok = np.array_equal(dst.cpu().numpy(), a[off // W: off // W + 1000, off % W: off % W + 4000])
                print(f"  {n} windows of 1000 x 4000 bytes in {dt * 1e3:.1f} ms = {n * 4.0 / dt / 1e3:.1f} GB/s, last window correct: {ok}")
                print("  unregister ->", hip.hipHostUnregister(C.c_void_p(addr)))
                break
        del buf
        m.close()
finally:
    os.unlink(path)
