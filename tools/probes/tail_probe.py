"""Where does bottleneck_tail_kernel spend its time? Diagnostic builds of bottleneck.hip (-DTD_TAIL_DIAG=1: no 3x3 phase,
=2: nothing after the mid tile — WRONG results, timing only) on the res2 shape, under rocprofv3 (tools/probes/tail_probe.sh)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.conv_diag import build  # noqa: E402

prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else 0
diag = sys.argv[2] if len(sys.argv) > 2 else "0"
so = build(f"tail_{diag}", [f"-DTD_TAIL_DIAG={diag}"] if diag != "0" else [], src="bottleneck")
lib = C.CDLL(so)
f = lib.td_bottleneck_tail_nhwc
f.restype = C.c_int
f.argtypes = [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]
dt = torch.float16 if prec else torch.float32
B, H, W, mid, cout = 8, 200, 200, 64, 256
x = torch.relu(torch.randn(B, H, W, mid, device="cuda")).to(dt)
w2 = (torch.randn(mid, 3, 3, mid, device="cuda") / (9 * mid) ** 0.5).to(dt)
w3 = (torch.randn(cout, 1, 1, mid, device="cuda") / mid ** 0.5).to(dt)
s2, b2 = torch.ones(mid, device="cuda"), torch.zeros(mid, device="cuda")
s3, b3 = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
sc = torch.randn(B, H, W, cout, device="cuda").to(dt)
y = torch.empty(B, H, W, cout, device="cuda", dtype=dt)
for bm in (os.environ.get("TD_TAIL_BM", "64"),):
    for _ in range(5):
        st = f(x.data_ptr(), w2.data_ptr(), s2.data_ptr(), b2.data_ptr(), w3.data_ptr(), s3.data_ptr(), b3.data_ptr(), sc.data_ptr(), y.data_ptr(),
               B, H, W, mid, cout, prec, torch.cuda.current_stream().cuda_stream)
        assert st == 0, st
    torch.cuda.synchronize()
print("done", diag, flush=True)
