"""Times the three-launch F(4x4,3x3) form against wino43_fused_kernel on the engine's layer shapes (fp32, batch 8). Run under
rocprofv3 (tools/probes/wino_fold_probe.sh): every shape runs each form a few times through td_conv2d_winograd_nhwc, whose host-side
filter transform and allocations are outside the kernels; the kernel trace carries the durations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from treedetection_amd import _lib  # noqa: E402

SHAPES = [  # name, B, C, H, W, N
    ("fpn_output2", 8, 256, 200, 200, 256),
    ("fpn_output3", 8, 256, 100, 100, 256),
    ("res3_conv2", 8, 128, 100, 100, 128),
    ("res4_conv2", 8, 256, 50, 50, 256),
    ("res5_conv2", 8, 512, 25, 25, 512),
    ("mask_fcn", 240, 256, 14, 14, 256),
]


def main():
    diag = any(v not in ("", "0") for v in os.environ.get("WF_VARS", "0").split(","))
    lib = None if diag else _lib.load()      # (the diag library must be the ONLY one loaded: same symbols, first definition wins)
    if diag:
        # the ablation builds (TD_WF_VAR) live behind -DTD_WF_DIAG: build this one file into /tmp and bind the same entry point
        import ctypes as C
        from tools.conv_diag import build
        so = build("wf_diag", ["-DTD_WF_DIAG"], src="wino_fused")
        lib = C.CDLL(so)
        lib.td_conv2d_winograd_nhwc.restype = C.c_int
        lib.td_conv2d_winograd_nhwc.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]
    only = sys.argv[1:] or None
    rng = np.random.default_rng(0)
    for name, B, C, H, W, N in SHAPES:
        if only and name not in only:
            continue
        x = torch.from_numpy(np.maximum(rng.standard_normal((B, H, W, C), dtype=np.float32), 0)).cuda()
        w = torch.from_numpy(rng.standard_normal((N, 3, 3, C), dtype=np.float32) / np.float32(np.sqrt(9 * C))).cuda()
        b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
        y = torch.empty((B, H, W, N), dtype=torch.float32, device="cuda")
        variants = [("0", "0", "8")] + [("1", v, "8") for v in os.environ.get("WF_VARS", "0").split(",")]
        for fold, var, st in variants:
            os.environ["TD_WINO_TILE"], os.environ["TD_WINO_FOLD"], os.environ["TD_WF_VAR"], os.environ["TD_WF_STAGES"] = "4", fold, var, st
            for _ in range(4):
                st = lib.td_conv2d_winograd_nhwc(x.data_ptr(), w.data_ptr(), None, b.data_ptr(), y.data_ptr(), B, H, W, C, N, 1,
                                                 torch.cuda.current_stream().cuda_stream)
                assert st == 0, st
            torch.cuda.synchronize()
        print(name, "done", flush=True)


if __name__ == "__main__":
    main()
