"""Debug: decode single LZW streams on the GPU (chunk path) and report the first differing byte against the host decoder."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from treedetection_amd import _lib
from treedetection_amd.synth import make_tile
lib = _lib.load()
rng = np.random.default_rng(0)
rgb, _ = make_tile(3, 700)
cases = {"tile": np.ascontiguousarray(rgb[:128, :128]).tobytes(), "noise": rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(),
         "ramp": bytes(range(256)) * 100, "few": rng.integers(0, 3, 40000, dtype=np.uint8).tobytes(), "lit300": rng.integers(0, 256, 300, dtype=np.uint8).tobytes(),
         "lit1000": rng.integers(0, 256, 1000, dtype=np.uint8).tobytes(), "pairs": (bytes([1, 2, 3, 4, 5, 6, 7, 8]) * 40)}
for name, raw in cases.items():
    src = np.frombuffer(raw, np.uint8); enc = np.empty(len(raw) * 2 + 64, np.uint8)
    n = lib.td_tiff_lzw_encode(src.ctypes.data, src.size, enc.ctypes.data, enc.size)
    comp = torch.from_numpy(np.concatenate([enc[:n], np.zeros(16, np.uint8)])).cuda()
    off = torch.zeros(1, dtype=torch.int64, device="cuda"); nb = torch.tensor([n], dtype=torch.int64, device="cuda")
    cap = len(raw) + 64
    for one in ("1", "0"):
        os.environ["TD_LZW_ONE_BY_ONE"] = one
        out = torch.zeros((1, cap), dtype=torch.uint8, device="cuda"); dec = torch.zeros(1, dtype=torch.int64, device="cuda")
        st = torch.full((3,), -1, dtype=torch.int32, device="cuda")
        _lib.check(lib.td_tiff_lzw_decode_dev(comp.data_ptr(), off.data_ptr(), nb.data_ptr(), 1, out.data_ptr(), cap, dec.data_ptr(), st.data_ptr(), _lib.stream_ptr()), "dec")
        torch.cuda.synchronize()
        got = out[0, :len(raw)].cpu().numpy()
        d = np.nonzero(got != src)[0]
        print(name, "one_by_one" if one == "1" else "chunks", "status", st[:1].tolist(), "decoded", int(dec[0]) & 0xffffffff, "of", len(raw), "first diffs", d[:6].tolist(),
              "got", got[d[:6]].tolist() if len(d) else "", "want", src[d[:6]].tolist() if len(d) else "")
