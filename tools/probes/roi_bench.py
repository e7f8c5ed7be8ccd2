"""Diagnostic: per-category device time of a plain-loop forward (one batch at a time, nothing overlapping), for the
RoIAlign ring depth / blocks-per-RoI sweep:  TD_ROI_DEPTH=4 TD_ROI_PARTS=4 python tools/probes/roi_bench.py fp16 [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from treedetection_amd.engine import Engine, INPUT_U8_HWC  # noqa: E402
from treedetection_amd.synth import make_stream  # noqa: E402
from treedetection_amd.weights import make_synthetic_state_dict  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = 8
    sd = make_synthetic_state_dict(50, seed=0)
    rgb_np, _ = make_stream(2 * B, 1000)
    rgb = torch.from_numpy(rgb_np).cuda()
    eng = Engine(sd, precision=prec)
    out = eng.alloc_outputs(B, 1000, 1000, paste=True)

    def step(i):
        batch, hw_valid, hw_out = eng.preprocess_tiles_u8([rgb[(i * B + j) % (2 * B)] for j in range(B)])
        eng.forward_raw(batch, INPUT_U8_HWC, hw_valid, hw_out, out)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    p = eng.profile_read(reset=True)
    tag = f"depth={os.environ.get('TD_ROI_DEPTH', 'default')} parts={os.environ.get('TD_ROI_PARTS', 'default')}"
    print(f"{prec} {tag}: " + "  ".join(f"{k} {v['ms'] / steps:.3f}" for k, v in p.items() if v["ms"] > 0), flush=True)


if __name__ == "__main__":
    main()
