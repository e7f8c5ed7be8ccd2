"""Does the matrix pipe run at its nominal clock under the fp16 contractions? Runs one conv launch shape in a loop for a few seconds
and samples `rocm-smi` (sclk, power) meanwhile: python tools/probes/clock_probe.py [tile id, default 17] [fp16|fp32]."""
import os, subprocess, sys, threading, time
sys.path.insert(0, ".")
import torch
from treedetection_amd import _lib

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
fp32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
dt_ = torch.float32 if fp32 else torch.float16
lib = _lib.load()
B, H, W, Cin, Cout = 8, 200, 200, 256, 256
x = torch.randn(B, H, W, Cin, device="cuda").relu().to(dt_)
w = (torch.randn(Cout, 3, 3, Cin, device="cuda") / (9 * Cin) ** 0.5).to(dt_)
y = torch.empty(B, H, W, Cout, device="cuda", dtype=dt_)
bias = torch.zeros(Cout, device="cuda")
args = (x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), None, 0, y.data_ptr(), B, H, W, Cin, Cout, 3, 3, 1, 1, 1, (0 if fp32 else 1) | ((cfg + 1) << 8), None)
stop = False
samples = []
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            s = [l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "power" in l]
            samples.append((time.time(), s))
        except Exception as e:
            samples.append((time.time(), [repr(e)]))
        time.sleep(0.25)
def idle_sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
    return [l.strip() for l in out.splitlines() if "sclk" in l or "ower" in l]
print("idle:", idle_sample(), flush=True)
th = threading.Thread(target=poll)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        lib.td_conv2d_nhwc(*args)
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
stop = True
th.join()
print(f"cfg {cfg}: {n} launches in {dt:.2f} s = {dt / n * 1e6:.1f} us each = {2.0 * B * H * W * Cout * Cin * 9 / (dt / n) / 1e12:.0f} TFLOP/s")
for t, s in samples:
    print(f"{t - t0:5.2f} s", s)
