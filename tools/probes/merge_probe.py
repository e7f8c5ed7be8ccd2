"""Would one launch for a stage's first-block shortcut + conv1 (same input, same stride: filters concatenated) pay? fp16, per tile id:
the two layers apart against the merged channel count. Round 3: 20-25 us per 8-tile step over the four stages — not built."""
import sys
sys.path.insert(0, ".")
import torch
from treedetection_amd import _lib
def run(lib, B, H, W, Cin, Cout, stride, cfg, n=20):
    x = torch.randn(B, H, W, Cin, device="cuda").relu().half()
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") / Cin ** 0.5).half()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.half)
    bias = torch.zeros(Cout, device="cuda")
    args = (x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), None, 0, y.data_ptr(), B, H, W, Cin, Cout, 1, 1, stride, 0, 1, 1 | ((cfg + 1) << 8), None)
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
for name, (B, H, W, Cin, s, n1, n2) in (("res2.0", (8, 200, 200, 64, 1, 256, 64)), ("res3.0", (8, 200, 200, 256, 2, 512, 128)),
                                         ("res4.0", (8, 100, 100, 512, 2, 1024, 256)), ("res5.0", (8, 50, 50, 1024, 2, 2048, 512))):
    for cfg in (0, 1, 2, 3, 15):
        t = [run(lib, B, H, W, Cin, c, s, cfg) for c in (n1, n2, n1 + n2)]
        print(f"{name} cfg {cfg:2d}: shortcut {t[0]:6.1f} + conv1 {t[1]:6.1f} = {t[0]+t[1]:6.1f}   merged {t[2]:6.1f} us", flush=True)
