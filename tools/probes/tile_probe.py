"""Kernel duration of every block tile on the fp16 / fp32 engine's layer shapes (batch 8): run under rocprofv3 --kernel-trace
(tools/probes/tile_probe.sh); each (shape, tile id) launches td_conv2d_nhwc three times, the trace carries the kernel durations
(the entry point's host-side filter packing and allocations are outside them)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from treedetection_amd import _lib  # noqa: E402

SHAPES = {  # name: (B, H, W, Cin, Cout, k, stride, residual)
    "res3_conv2": (8, 100, 100, 128, 128, 3, 1, False),
    "res3_conv3": (8, 100, 100, 128, 512, 1, 1, True),
    "res4_conv1": (8, 50, 50, 1024, 256, 1, 1, False),
    "res4_conv2": (8, 50, 50, 256, 256, 3, 1, False),
    "res4_conv3": (8, 50, 50, 256, 1024, 1, 1, True),
    "res5_conv2": (8, 25, 25, 512, 512, 3, 1, False),
    "res5_conv3": (8, 25, 25, 512, 2048, 1, 1, True),
    "fc1": (8000, 1, 1, 12544, 1024, 1, 1, False),
    "fpn_lateral2": (8, 200, 200, 256, 256, 1, 1, False),
    "res2_conv1": (8, 200, 200, 256, 64, 1, 1, False),
    "res2_shortcut": (8, 200, 200, 64, 256, 1, 1, False),
    "res3_conv1": (8, 100, 100, 512, 128, 1, 1, False),
    "res3_shortcut": (8, 200, 200, 256, 512, 1, 2, False),
    "res4_shortcut": (8, 100, 100, 512, 1024, 1, 2, False),
    "res5_conv1": (8, 25, 25, 2048, 512, 1, 1, False),
    "fpn_lateral3": (8, 100, 100, 512, 256, 1, 1, False),
    "fc2": (8000, 1, 1, 1024, 1024, 1, 1, False),
    "rpn_head_p2": (8, 200, 200, 256, 15, 1, 1, False),
    "fpn_lateral2_up": (8, 200, 200, 256, 256, 1, 1, 2),        # as the engine runs it: + the half-resolution top-down map
    "fpn_lateral2_res": (8, 200, 200, 256, 256, 1, 1, True),
    "res2_shortcut": (8, 200, 200, 64, 256, 1, 1, False),
}


def main():
    prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else 0
    cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 10, 17, 23, 24, 25, 26, 27, 29, 30]
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(SHAPES)
    lib = _lib.load()
    dt = torch.float16 if prec else torch.float32
    plan = []
    for name in names:
        B, H, W, Cin, Cout, k, stride, res = SHAPES[name]
        x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")).to(dt)
        w = (torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5).to(dt)
        bias = torch.zeros(Cout, device="cuda")
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        y = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=dt)
        r = (torch.randn(B, Ho // 2, Wo // 2, Cout, device="cuda") if res == 2 else torch.randn(B, Ho, Wo, Cout, device="cuda")).to(dt) if res else None
        torch.cuda.synchronize()
        for cfg in cfgs:
            ok = 0
            for _ in range(3):
                st = lib.td_conv2d_nhwc(x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), r.data_ptr() if res else None, 1 if res == 2 else 0, y.data_ptr(), B, H, W, Cin, Cout,
                                        k, k, stride, k // 2, 1, prec | ((cfg + 1) << 8), _lib.stream_ptr())
                ok += st == 0
            torch.cuda.synchronize()
            plan.append((name, cfg, ok))
    with open(os.environ.get("TILE_PROBE_PLAN", "/tmp/tile_probe_plan.txt"), "w") as f:
        for name, cfg, ok in plan:
            f.write(f"{name} {cfg} {ok}\n")


if __name__ == "__main__":
    main()
