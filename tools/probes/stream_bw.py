"""What an HBM-bound elementwise pass reaches on this GPU (reference point for the thin 1x1 conv layers)."""
import torch
n = 8 * 200 * 200 * 256
for dt in (torch.float32, torch.float16):
    x = torch.randn(n, device="cuda").to(dt); r = torch.randn(n, device="cuda").to(dt); y = torch.empty_like(x)
    es = x.element_size()
    for name, fn, nb in (("copy", lambda: y.copy_(x), 2), ("add", lambda: torch.add(x, r, out=y), 3), ("relu(add)", lambda: torch.relu_(torch.add(x, r, out=y)), 5)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"{str(dt):14s} {name:10s} {ms*1e3:8.1f} us  {nb*n*es/ms/1e9:6.2f} TB/s")
