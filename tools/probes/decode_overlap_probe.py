"""Does the raster decode kernel share the chip with the model stage? python tools/probes/decode_overlap_probe.py [codec=lzw] [side=9000]
Three fp16 engines on three streams run forwards (bench.py's schedule) while a fourth stream decodes the same compressed raster
again and again; prints the model rate alone, the decode time alone, and both while they overlap. TD_DECODE_RING=small|large."""
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from treedetection_amd.engine import Engine, INPUT_U8_HWC          # noqa: E402
from treedetection_amd.geotiff import GeoTiff, write_geotiff       # noqa: E402
from treedetection_amd.synth import make_tile                      # noqa: E402
from treedetection_amd.weights import make_synthetic_state_dict    # noqa: E402

args = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
side, codec = int(args.get("side", 9000)), args.get("codec", "lzw")
B, S = 8, 1000
dev = torch.device("cuda:0")
sd = make_synthetic_state_dict(depth=50, seed=0)
engs = [Engine(sd, device=0, precision="fp16") for _ in range(3)]
outs = [e.alloc_outputs(B, S, S, paste=True) for e in engs]
streams = [torch.cuda.Stream() for _ in range(3)]
rgb = [torch.from_numpy(make_tile(s, S)[0]).to(dev) for s in range(16)]


def step(i):
    e, o = engs[i % 3], outs[i % 3]
    with torch.cuda.stream(streams[i % 3]):
        tiles = [rgb[(i * B + j) % 16] for j in range(B)]
        batch, hv, ho = e.preprocess_tiles_u8(tiles)
        e.forward_raw(batch, INPUT_U8_HWC, hv, ho, o)


for i in range(6):
    step(i)
torch.cuda.synchronize()


def model_rate(seconds, stop=None):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and (stop is None or not stop.is_set()):
        for _ in range(6):
            step(n)
            n += 1
        for s in streams:          # keep the host at most a few batches ahead
            s.synchronize()
    torch.cuda.synchronize()
    return n * B / (time.perf_counter() - t0)


root = tempfile.mkdtemp(prefix="td_ovl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
nb = -(-side // S)
img = np.zeros((4, nb * S, nb * S), np.uint8)
for r in range(nb):
    for c in range(nb):
        t = make_tile((r * nb + c) % 16, S)[0]
        img[:3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t.transpose(2, 0, 1)
        img[3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t[..., 1]
img = np.ascontiguousarray(img[:, :side, :side])
tif = f"{root}/r.tif"
write_geotiff(tif, img, (0.2, 0, 412000.0, 0, -0.2, 5318000.0), 25832, compression=codec, tile=(256, 256), predictor=2)
del img
g = GeoTiff(tif)
prio = args.get("priority", "normal")
if prio == "low":
    from treedetection_amd import _lib
    dstream = _lib.low_priority_stream(0)
else:
    dstream = torch.cuda.Stream()
pinned = [None]


def decode_once():
    image, check = g.decode_to_device("cuda:0", dstream, pinned, None)
    check()
    return check.kernel_ms


decode_once()
alone = [decode_once() for _ in range(3)]
m_alone = model_rate(2.0)
stop = threading.Event()
during = []


def decoder():
    while not stop.is_set():
        during.append(decode_once())


th = threading.Thread(target=decoder)
th.start()
m_both = model_rate(3.0)
stop.set()
th.join()
m_after = model_rate(1.0)
print(json.dumps({"codec": codec, "side": side, "ring": os.environ.get("TD_DECODE_RING", "auto"), "priority": prio,
                  "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                  "decode_kernel_ms_alone": [round(x, 1) for x in alone], "model_tiles_per_s_alone": round(m_alone), "model_after": round(m_after),
                  "decode_kernel_ms_while_model_runs": [round(x, 1) for x in during], "model_tiles_per_s_while_decoding": round(m_both),
                  "decodes_in_3s": len(during)}))
for e in engs:
    e.close()
import shutil
shutil.rmtree(root, ignore_errors=True)
