"""What does the residual epilogue cost? fp16 1x1 layers (FPN lateral 2: 256 -> 256 at M = 320 000; res3 conv3: 128 -> 512 at
M = 80 000) timed without a residual, with a same-size one and with the half-resolution one (nearest 2x upsample add), per tile id."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import torch
from treedetection_amd import _lib

def run(lib, B, H, W, Cin, Cout, res, cfg, n=20):
    x = torch.randn(B, H, W, Cin, device="cuda").relu().half()
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).half()
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.half)
    bias = torch.zeros(Cout, device="cuda")
    r = None
    if res == 1:
        r = torch.randn(B, H, W, Cout, device="cuda").half()
    elif res == 2:
        r = torch.randn(B, H // 2, W // 2, Cout, device="cuda").half()
    args = (x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), r.data_ptr() if r is not None else None, 1 if res == 2 else 0, y.data_ptr(),
            B, H, W, Cin, Cout, 1, 1, 1, 0, 0, 1 | ((cfg + 1) << 8), None)
    for _ in range(3):
        _lib.check(lib.td_conv2d_nhwc(*args))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        lib.td_conv2d_nhwc(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

lib = _lib.load()
for name, shp in (("fpn_lateral2", (8, 200, 200, 256, 256)), ("res3.conv3", (8, 100, 100, 128, 512)), ("res4.conv3", (8, 50, 50, 256, 1024))):
    for cfg in (0, 2, 15, 16, 24, 26, 27):
        t = [run(lib, *shp, res, cfg) for res in (0, 1, 2)]
        print(f"{name:14s} cfg {cfg:2d}: no residual {t[0]:7.1f} us   same-size {t[1]:7.1f} us   upsampled {t[2]:7.1f} us", flush=True)
