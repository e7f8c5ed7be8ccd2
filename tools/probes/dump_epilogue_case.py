"""Saves the packed masks of a few synthetic tiles (regions, offsets, used bit rows, scores) for host-epilogue profiling
on a CPU-only machine: python tools/probes/dump_epilogue_case.py out.npz [n_tiles] [precision]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from treedetection_amd.engine import Engine, INPUT_U8_HWC
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

out, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
eng = Engine(make_synthetic_state_dict(50, seed=0), device=0, precision=precision)
tiles = [torch.from_numpy(make_tile(i, 1000)[0]).cuda() for i in range(n)]
images, hw_valid, hw_out = eng.preprocess_tiles_u8(tiles)
o = eng.alloc_outputs(n, 1000, 1000, paste=True)
eng.forward_raw(images, INPUT_U8_HWC, hw_valid, hw_out, o)
torch.cuda.synchronize()
h = {k: v.cpu().numpy() for k, v in o.items()}
save = {}
for i in range(n):
    c = int(h["count"][i])
    reg, off = h["mask_region"][i][:c], h["mask_offset"][i][:c]
    used = 0 if c == 0 else int(off[-1] + ((reg[-1, 2] - reg[-1, 0] + 31) // 32) * (reg[-1, 3] - reg[-1, 1]))
    save[f"region{i}"], save[f"offset{i}"] = reg, off
    save[f"bits{i}"], save[f"scores{i}"] = h["mask_bits"][i][:used], h["scores"][i][:c]
    print(i, c, used * 4 / 1e6, "MB")
np.savez_compressed(out, n=n, **save)
