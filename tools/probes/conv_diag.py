"""Diagnostic: where does a big 3x3 conv lose time? Times td_conv2d_nhwc for the fpn_output2 shape with three builds
of conv_igemm.hip — product, no global loads in the k-loop, no loads + no barrier — built into /tmp (never shipped)."""
import ctypes as C, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "treedetection_amd", "csrc")

def build(tag, defs, src="conv_igemm"):
    """One kernel file rebuilt with the diagnostic defines, linked with the product build's other objects (csrc/*.o)."""
    import glob
    out, obj = f"/tmp/libdiag_{tag}.so", f"/tmp/diag_{tag}.o"
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"] + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if src in ("conv_igemm", "bottleneck") else [])
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + defs + ["-c", os.path.join(CS, src + ".hip"), "-o", obj], check=True)
    others = [o for o in sorted(glob.glob(os.path.join(CS, "*.o"))) if os.path.basename(o) != src + ".o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + others + ["-o", out], check=True)
    return out

def bench(lib, prec, B, H, W, Cin, Cout, k, cfg_name, residual=False):
    es = 4 if prec == 0 else 2
    dt = torch.float32 if prec == 0 else torch.float16
    x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, k, k, Cin, device="cuda") / (Cin * k * k) ** 0.5).to(dt)
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=dt)
    bias = torch.zeros(Cout, device="cuda")
    f = lib.td_conv2d_nhwc
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p] + [C.c_int] * 11 + [C.c_void_p]
    res = torch.randn(B, H, W, Cout, device="cuda").to(dt) if residual else None
    args = (x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), res.data_ptr() if residual else None, 0, y.data_ptr(), B, H, W, Cin, Cout, k, k, 1, k // 2, 1, prec, None)
    for _ in range(3):
        assert f(*args) == 0
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 10
    for _ in range(n):
        f(*args)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    nbytes = es * (B * H * W * (Cin + Cout * (2 if residual else 1)) + Cout * Cin * k * k)
    print(f"{cfg_name:34s} prec={prec} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  {nbytes/ms/1e9:6.2f} TB/s", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tiles":
        # product build only, one forced block tile per process (TD_CONV_CFG is read once per process)
        lib = C.CDLL(os.path.join(ROOT, "treedetection_amd", "libtreedet_hip.so"))
        tag = "cfg" + os.environ.get("TD_CONV_CFG", "-1")
        for prec in (0, 1):
            bench(lib, prec, 8, 200, 200, 256, 256, 3, tag + " 3x3 256->256 M=320k")
            bench(lib, prec, 8, 100, 100, 128, 128, 3, tag + " 3x3 128->128 M=80k")
            bench(lib, prec, 8, 50, 50, 1024, 256, 1, tag + " 1x1 1024->256 M=20k")
            bench(lib, prec, 8000, 1, 1, 12544, 1024, 1, tag + " fc1 M=8000")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "thin":
        # the HBM-bound 1x1 layers of res2 / res3 / FPN (one or two k-steps)
        lib = C.CDLL(os.path.join(ROOT, "treedetection_amd", "libtreedet_hip.so"))
        tag = "cfg" + os.environ.get("TD_CONV_CFG", "-1")
        for prec in (0, 1):
            bench(lib, prec, 8, 200, 200, 64, 256, 1, tag + " res2.conv3 64->256 +res", residual=True)
            bench(lib, prec, 8, 200, 200, 256, 64, 1, tag + " res2.conv1 256->64")
            bench(lib, prec, 8, 200, 200, 64, 256, 1, tag + " res2.shortcut 64->256")
            bench(lib, prec, 8, 100, 100, 128, 512, 1, tag + " res3.conv3 128->512 +res", residual=True)
            bench(lib, prec, 8, 200, 200, 256, 256, 1, tag + " fpn_lateral2 256->256 +res", residual=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pp8":
        # where does conv_pp8_kernel (cfg 17) spend its time? builds without its in-loop DMA / fragment reads / barriers
        os.environ["TD_CONV_CFG"] = "17"
        variants = {"pp8 product": [], "pp8 no_dma": ["-DTD_DIAG_PP8_NO_DMA"],
                    "pp8 no_dma_no_reads": ["-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS"],
                    "pp8 no_reads": ["-DTD_DIAG_PP8_NO_READS"],
                    "pp8 mfma_only": ["-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS", "-DTD_DIAG_PP8_NO_BARRIER"],
                    "pp8 mfma_nobar_noepi": ["-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS", "-DTD_DIAG_PP8_NO_BARRIER", "-DTD_DIAG_PP8_NO_EPILOGUE"],
                    "pp8 nobar": ["-DTD_DIAG_PP8_NO_BARRIER"],
                    "pp8 no_epilogue": ["-DTD_DIAG_PP8_NO_EPILOGUE"],
                    "pp8 v1_dma_first": ["-DTD_PP8_V1"], "pp8 v2_dma_light_phases": ["-DTD_PP8_V2"],
                    "pp8 mfma16": ["-DTD_DIAG_MFMA16"],
                    "pp8 mfma16b_same_flops": ["-DTD_DIAG_MFMA16B"],
                    "pp8 mfma16b_loop_only": ["-DTD_DIAG_MFMA16B", "-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS", "-DTD_DIAG_PP8_NO_EPILOGUE"],
                    "pp8 mfma16_loop_only": ["-DTD_DIAG_MFMA16", "-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS", "-DTD_DIAG_PP8_NO_EPILOGUE"],
                    "pp8 loop_only": ["-DTD_DIAG_PP8_NO_DMA", "-DTD_DIAG_PP8_NO_READS", "-DTD_DIAG_PP8_NO_EPILOGUE"]}
        if len(sys.argv) > 2:
            variants = {k: v for k, v in variants.items() if any(a in k for a in sys.argv[2:])}
        for tag, defs in variants.items():
            lib = C.CDLL(build(tag.replace(" ", "_"), defs))
            bench(lib, 1, 8, 200, 200, 256, 256, 3, tag + " 3x3 256->256 M=320k")
            bench(lib, 1, 8, 100, 100, 512, 512, 1, tag + " 1x1 512->512 M=80k")
        sys.exit(0)
    libs = {"product": [], "no_loads": ["-DTD_DIAG_NO_LOADS"], "no_loads_no_barrier": ["-DTD_DIAG_NO_LOADS", "-DTD_DIAG_NO_BARRIER"]}
    for tag, defs in libs.items():
        lib = C.CDLL(build(tag, defs))
        for prec in (0, 1):
            bench(lib, prec, 8, 200, 200, 256, 256, 3, tag + " 3x3 256->256 M=320k")
            bench(lib, prec, 8, 100, 100, 128, 128, 3, tag + " 3x3 128->128 M=80k")
