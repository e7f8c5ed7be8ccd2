"""GPU parity: td_conv2d_nhwc (MFMA implicit GEMM) vs torch CPU F.conv2d on the same seeded inputs.

fp32 tolerance: the MFMA accumulates an exact f32 fmaf chain in a different order than the CPU
library, so |err| <= 2e-5 * sum|a*b| (stated here, checked per case)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import conv2d_hip

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, scale, bias, res(0 none,1 same,2 upsample), relu
    (1, 32, 8, 8, 16, 1, 1, 0, False, False, 0, False),
    (2, 64, 20, 24, 64, 3, 1, 1, True, True, 0, True),
    (2, 64, 33, 17, 256, 1, 1, 0, True, True, 1, True),
    (1, 256, 40, 40, 128, 1, 2, 0, True, True, 0, True),
    (1, 128, 25, 25, 128, 3, 1, 1, True, True, 0, True),
    (3, 96, 14, 14, 100, 3, 1, 1, False, True, 0, True),
    (1, 512, 16, 16, 256, 1, 1, 0, False, True, 2, False),
    (1, 256, 120, 120, 15, 1, 1, 0, False, True, 0, False),
    (1, 64, 200, 200, 64, 3, 1, 1, True, True, 0, True),
    (2, 1024, 13, 13, 2048, 1, 1, 0, True, True, 1, True),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch(case):
    B, Cin, H, W, Cout, k, stride, pad, use_scale, use_bias, res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.float32(np.sqrt(Cin * k * k))
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32) if use_scale else None
    bias = rng.standard_normal(Cout).astype(np.float32) if use_bias else None
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, stride=stride, padding=pad)
    if use_scale:
        ref = ref * torch.from_numpy(scale).reshape(1, -1, 1, 1)
    if use_bias:
        ref = ref + torch.from_numpy(bias).reshape(1, -1, 1, 1)
    r = None
    if res == 1:
        r = rng.standard_normal(tuple(ref.shape), dtype=np.float32)
        ref = ref + torch.from_numpy(r)
    elif res == 2:
        Ho, Wo = ref.shape[-2:]
        r = rng.standard_normal((B, Cout, Ho // 2, Wo // 2), dtype=np.float32)
        ref = ref + F.interpolate(torch.from_numpy(r), scale_factor=2.0, mode="nearest")
    if relu:
        ref = F.relu(ref)
    got = conv2d_hip(x, w, scale, bias, r, res_shift=1 if res == 2 else 0, stride=stride, pad=pad, relu=relu)
    ref = ref.numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err <= 5e-5 * max(1.0, np.abs(ref).max()), f"max abs err {err}"


FP16_CASES = [
    (1, 64, 16, 16, 64, 1, 1, 0, True, True, 0, True),
    (2, 64, 20, 24, 128, 3, 1, 1, True, True, 0, True),
    (2, 128, 33, 17, 256, 1, 1, 0, True, True, 1, True),
    (1, 256, 40, 40, 128, 1, 2, 0, True, True, 0, True),
    (1, 512, 16, 16, 256, 1, 1, 0, False, True, 2, False),
    (1, 256, 60, 60, 15, 1, 1, 0, False, True, 0, False),
    (2, 1024, 13, 13, 512, 1, 1, 0, True, True, 1, True),
    (1, 256, 50, 50, 256, 3, 1, 1, False, True, 0, True),
]


@pytest.mark.parametrize("case", FP16_CASES)
def test_conv_fp16_matches_torch(case):
    """fp16 storage + v_mfma_f32_32x32x16_f16 (fp32 accumulate): compare with an fp32 conv of the SAME fp16-rounded
    inputs; tolerance = one fp16 rounding of the output (2^-10 relative) plus accumulation-order noise."""
    B, Cin, H, W, Cout, k, stride, pad, use_scale, use_bias, res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32).astype(np.float16).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.float32(np.sqrt(Cin * k * k))).astype(np.float16).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32) if use_scale else None
    bias = rng.standard_normal(Cout).astype(np.float32) if use_bias else None
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, stride=stride, padding=pad)
    if use_scale:
        ref = ref * torch.from_numpy(scale).reshape(1, -1, 1, 1)
    if use_bias:
        ref = ref + torch.from_numpy(bias).reshape(1, -1, 1, 1)
    r = None
    if res == 1:
        r = rng.standard_normal(tuple(ref.shape), dtype=np.float32).astype(np.float16).astype(np.float32)
        ref = ref + torch.from_numpy(r)
    elif res == 2:
        Ho, Wo = ref.shape[-2:]
        r = rng.standard_normal((B, Cout, Ho // 2, Wo // 2), dtype=np.float32).astype(np.float16).astype(np.float32)
        ref = ref + F.interpolate(torch.from_numpy(r), scale_factor=2.0, mode="nearest")
    if relu:
        ref = F.relu(ref)
    got = conv2d_hip(x, w, scale, bias, r, res_shift=1 if res == 2 else 0, stride=stride, pad=pad, relu=relu, precision=1)
    ref = ref.numpy()
    err = np.abs(got - ref)
    assert (err <= 2e-3 * np.maximum(np.abs(ref), 1.0)).all(), f"max err {err.max()}"


PP8_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, res, relu   (fp16; 256 x 256 ping-pong tile = cfg 17)
    (1, 64, 16, 16, 64, 1, 1, 0, 0, True),            # one k-chunk, one block, N tail
    (1, 128, 20, 20, 256, 1, 1, 0, 1, True),          # two k-chunks
    (2, 192, 23, 19, 300, 1, 1, 0, 0, False),         # three k-chunks, M and N tails, 2 x 4 blocks
    (1, 256, 50, 50, 256, 3, 1, 1, 0, True),          # 36 k-chunks, zero padding, 10 M tiles
    (2, 64, 64, 64, 512, 3, 1, 1, 1, True),           # 9 k-chunks, 32 x 2 blocks
    (1, 256, 40, 40, 128, 1, 2, 0, 0, True),          # stride 2
    (1, 512, 16, 16, 256, 1, 1, 0, 2, False),         # nearest-2x upsampled residual
    (3, 1024, 13, 13, 2048, 1, 1, 0, 1, True),
]


@pytest.mark.parametrize("case", PP8_CASES)
def test_conv_pp8_equals_the_reference_tile_bit_for_bit(case):
    """conv_pp8_kernel (cfg 17) keeps the k order of conv_igemm_kernel, so on the same fp16 inputs its output must be
    IDENTICAL to the 128 x 128 tile's (cfg 0), which test_conv_fp16_matches_torch checks against torch."""
    B, Cin, H, W, Cout, k, stride, pad, res, relu = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, k, k), dtype=np.float32) / np.float32(np.sqrt(Cin * k * k))
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = None
    if res == 1:
        r = rng.standard_normal((B, Cout, Ho, Wo), dtype=np.float32)
    elif res == 2:
        r = rng.standard_normal((B, Cout, Ho // 2, Wo // 2), dtype=np.float32)
    kw = dict(scale=scale, bias=bias, residual_nchw=r, res_shift=1 if res == 2 else 0, stride=stride, pad=pad, relu=relu, precision=1)
    ref = conv2d_hip(x, w, tile_cfg=0, **kw)
    for _ in range(3):                                  # a racy schedule would not repeat
        got = conv2d_hip(x, w, tile_cfg=17, **kw)
        assert got.shape == ref.shape and np.array_equal(got, ref)
    assert np.abs(ref).max() > 0.5


@pytest.mark.parametrize("cfg", [4, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_conv_every_block_tile_variant(cfg):
    """The engine picks a block tile per layer by measurement; every variant must compute the same convolution.
    TD_CONV_CFG forces one variant for a whole process (diagnostic hook), so the cases above re-run in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TD_CONV_CFG=str(cfg))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_conv_gpu.py"), "-m", "gpu", "-q", "-x",
                        "-k", "test_conv_matches_torch or test_conv_fp16_matches_torch"], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"{len(CASES) + len(FP16_CASES)} passed" in r.stdout


WINO_CASES = [
    # B, Cin, H, W, Cout, scale, bias, relu
    (1, 32, 8, 8, 16, False, True, False),
    (2, 64, 20, 24, 64, True, True, True),
    (1, 128, 25, 25, 128, True, True, True),          # odd sizes: the last tile row / column is half outside
    (3, 96, 14, 14, 100, False, True, True),          # the mask head's 14 x 14 RoIs
    (1, 256, 50, 50, 256, False, True, False),
    (1, 64, 1, 7, 32, False, False, False),
    (2, 256, 13, 13, 256, False, True, True),
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_matches_torch(case):
    """Winograd F(2x2,3x3) path of the fp32 engine vs torch CPU F.conv2d: same tolerance as the direct kernel
    (|err| <= 5e-5 * max|ref|; the transforms add a few roundings per output)."""
    from treedetection_amd import _lib
    from tests.gpu_util import dev
    B, Cin, H, W, Cout, use_scale, use_bias, relu = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.float32(np.sqrt(Cin * 9))
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32) if use_scale else None
    bias = rng.standard_normal(Cout).astype(np.float32) if use_bias else None
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, stride=1, padding=1)
    if use_scale:
        ref = ref * torch.from_numpy(scale).reshape(1, -1, 1, 1)
    if use_bias:
        ref = ref + torch.from_numpy(bias).reshape(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    ref = ref.numpy()
    lib = _lib.load()
    xd, wd = dev(x.transpose(0, 2, 3, 1)), dev(w.transpose(0, 2, 3, 1))
    sd = dev(scale) if use_scale else None
    bd = dev(bias) if use_bias else None
    y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    _lib.check(lib.td_conv2d_winograd_nhwc(p(xd), p(wd), p(sd), p(bd), y.data_ptr(), B, H, W, Cin, Cout, int(relu), _lib.stream_ptr()),
               "td_conv2d_winograd_nhwc")
    got = y.cpu().numpy().transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err <= 5e-5 * max(1.0, np.abs(ref).max()), f"max abs err {err}"
    # and against the direct MFMA kernel on the same inputs
    direct = conv2d_hip(x, w, scale, bias, None, stride=1, pad=1, relu=relu)
    assert np.abs(got - direct).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    # the fused form (input transform inside the contraction's A staging, the default) and the three-kernel form
    # (wino_input_kernel + batched conv_igemm launch) associate identically: bit for bit
    import os
    os.environ["TD_WINO_FUSED"] = "0"
    try:
        y2 = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.td_conv2d_winograd_nhwc(p(xd), p(wd), p(sd), p(bd), y2.data_ptr(), B, H, W, Cin, Cout, int(relu), _lib.stream_ptr()),
                   "td_conv2d_winograd_nhwc")
    finally:
        del os.environ["TD_WINO_FUSED"]
    assert torch.equal(y, y2)


WINO43_CASES = WINO_CASES + [
    (1, 128, 100, 100, 128, True, True, True),        # res3 conv2 (whole tiles)
    (2, 256, 50, 50, 256, False, True, True),         # 50 = 12 whole tiles + a half one per row / column
    (1, 256, 200, 200, 256, False, True, False),      # FPN output 2 / RPN conv p2
    (1, 128, 47, 61, 160, True, False, True),         # 3 and 1 pixels into the last tile
]


@pytest.mark.parametrize("case", WINO43_CASES)
def test_winograd_f4x4_conv_matches_torch(case):
    """Winograd F(4x4,3x3) (the fp32 engine's path on the large maps) vs torch CPU F.conv2d. Stated tolerance: the same
    |err| <= 5e-5 * max|ref| as the direct kernel — measured about 1e-5 at 256 channels (the larger transform constants
    cost a decimal digit against F(2x2)); partial tiles at the right / bottom border included."""
    import os
    from treedetection_amd import _lib
    from tests.gpu_util import dev
    B, Cin, H, W, Cout, use_scale, use_bias, relu = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = np.maximum(rng.standard_normal((B, Cin, H, W), dtype=np.float32), -0.5)      # mostly positive, as post-ReLU maps are
    w = rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.float32(np.sqrt(Cin * 9))
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32) if use_scale else None
    bias = rng.standard_normal(Cout).astype(np.float32) if use_bias else None
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride=1, padding=1)
    if use_scale:
        ref = ref * torch.from_numpy(scale).double().reshape(1, -1, 1, 1)
    if use_bias:
        ref = ref + torch.from_numpy(bias).double().reshape(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    ref = ref.numpy()
    lib = _lib.load()
    xd, wd = dev(x.transpose(0, 2, 3, 1)), dev(w.transpose(0, 2, 3, 1))
    sd = dev(scale) if use_scale else None
    bd = dev(bias) if use_bias else None
    p = lambda t: t.data_ptr() if t is not None else None      # noqa: E731
    os.environ["TD_WINO_TILE"] = "4"
    try:
        ys = []
        for _ in range(2):
            y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.td_conv2d_winograd_nhwc(p(xd), p(wd), p(sd), p(bd), y.data_ptr(), B, H, W, Cin, Cout, int(relu), _lib.stream_ptr()),
                       "td_conv2d_winograd_nhwc")
            ys.append(y)
    finally:
        del os.environ["TD_WINO_TILE"]
    assert torch.equal(ys[0], ys[1])                            # deterministic
    # the persistent plane-contraction kernels (tile ids 18-20, plane_gemm_kernel) keep the k order: same bits
    for cfg in ("18", "19", "20", "0", "3"):
        os.environ["TD_WINO_TILE"], os.environ["TD_CONV_CFG"] = "4", cfg
        try:
            y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.td_conv2d_winograd_nhwc(p(xd), p(wd), p(sd), p(bd), y.data_ptr(), B, H, W, Cin, Cout, int(relu), _lib.stream_ptr()),
                       "td_conv2d_winograd_nhwc")
        finally:
            del os.environ["TD_WINO_TILE"], os.environ["TD_CONV_CFG"]
        assert torch.equal(y, ys[0]), f"tile_cfg {cfg} differs"
    got = ys[0].cpu().numpy().transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    print(f"\n[F(4x4)] {case}: max abs err {err:.3e} = {err / max(1.0, np.abs(ref).max()):.2e} of max|ref|")
    assert err <= 5e-5 * max(1.0, np.abs(ref).max()), f"max abs err {err}"
