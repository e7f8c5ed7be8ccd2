"""Does the model stage itself cost more with the compact-crown mask head (weights.blob_mask_head) than with the seeded random
one? Plain loop, one engine, per-category device time (td_engine_profile_*)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from treedetection_amd.engine import Engine, INPUT_U8_HWC  # noqa: E402
from treedetection_amd.synth import make_stream  # noqa: E402
from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
sd = make_synthetic_state_dict(50, seed=0)
rgb_np, _ = make_stream(16, 1000, distinct=16)
rgb = torch.from_numpy(rgb_np).cuda()
for tag, w in (("noise", sd), ("crowns", blob_mask_head(sd, seed=0)), ("noise", sd), ("crowns", blob_mask_head(sd, seed=0))):
    eng = Engine(w, precision=prec)
    out = eng.alloc_outputs(8, 1000, 1000, paste=True)

    def step(i):
        tiles = [rgb[(i * 8 + j) % 16] for j in range(8)]
        batch, hv, ho = eng.preprocess_tiles_u8(tiles)
        eng.forward_raw(batch, INPUT_U8_HWC, hv, ho, out)
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = eng.profile_read(reset=True)
    print(tag, prec, f"{1e3 * dt / n:.3f} ms/step", "dets", int(out["count"].sum().item()), {k: round(v["ms"] / n, 3) for k, v in p.items() if isinstance(v, dict) and v.get("ms", 0) > 0.01}, flush=True)
    eng.close()
