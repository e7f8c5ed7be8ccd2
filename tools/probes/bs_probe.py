"""Where does conv_bs_kernel (tile id 33) spend a row tile? Ablation builds of conv_bstat.hip (-DTD_BS_DIAG bit mask: 1 no stores,
2 no MFMAs, 4 no LDS transposition, 8 no activation loads — WRONG results, timing only) on the engine's shapes, under rocprofv3
(tools/probes/bs_probe.sh). python tools/probes/bs_probe.py fp16|fp32 <diag> [shape,...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.conv_diag import build  # noqa: E402

SHAPES = {  # name: (B, H, W, Cin, Cout, residual 0 / 1 same size / 2 half resolution)
    "fpn_lateral2": (8, 200, 200, 256, 256, 2),
    "lateral2_plain": (8, 200, 200, 256, 256, 0),
    "res3_conv3": (8, 100, 100, 128, 512, 1),
    "res2_shortcut": (8, 200, 200, 64, 256, 0),
}
prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else 0
diag = sys.argv[2] if len(sys.argv) > 2 else "0"
names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(SHAPES)
so = build(f"bs_{diag}", [f"-DTD_BS_DIAG={diag}"] if diag != "0" else [], src="conv_bstat")
lib = C.CDLL(so)
f = lib.td_conv2d_nhwc
f.restype = C.c_int
f.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p] + [C.c_int] * 11 + [C.c_void_p]
dt = torch.float16 if prec else torch.float32
for name in names:
    B, H, W, Cin, Cout, res = SHAPES[name]
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")).to(dt)
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).to(dt)
    bias = torch.zeros(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=dt)
    r = (torch.randn(B, H // 2, W // 2, Cout, device="cuda") if res == 2 else torch.randn(B, H, W, Cout, device="cuda")).to(dt) if res else None
    for _ in range(4):
        st = f(x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), r.data_ptr() if res else None, 1 if res == 2 else 0, y.data_ptr(), B, H, W, Cin, Cout,
               1, 1, 1, 0, 1, prec | (34 << 8), torch.cuda.current_stream().cuda_stream)
        assert st == 0, st
    torch.cuda.synchronize()
print("done", diag, flush=True)
