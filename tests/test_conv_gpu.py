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


@pytest.mark.parametrize("cfg256", [17])
@pytest.mark.parametrize("case", PP8_CASES)
def test_conv_pp8_equals_the_reference_tile_bit_for_bit(case, cfg256):
    """conv_pp8_kernel (cfg 17: 8 waves, ping-pong phases) keeps the k order of conv_igemm_kernel, so on the same fp16 inputs its output must be IDENTICAL to the
    128 x 128 tile's (cfg 0), which test_conv_fp16_matches_torch checks against torch."""
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
        got = conv2d_hip(x, w, tile_cfg=cfg256, **kw)
        assert got.shape == ref.shape and np.array_equal(got, ref)
    assert np.abs(ref).max() > 0.5


@pytest.mark.parametrize("cfg", [4, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 31, 32])
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


WINO_FOLD_CASES = [
    # B, Cin, H, W, Cout, scale, bias, relu — Cin = 128, 256 or 512, Cout % 64 == 0 (wino43_fused_ok)
    (1, 128, 100, 100, 128, True, True, True),        # res3 conv2: 4 k-chunks per plane (the 4-stage pipeline), whole tiles
    (2, 256, 50, 50, 256, False, True, True),         # res4 conv2 / FPN output 4: a half tile per row / column
    (1, 256, 200, 200, 256, False, True, False),      # FPN output 2 / RPN conv p2: 2 500 tiles = 39 blocks + 4 tiles
    (1, 128, 47, 61, 192, True, False, True),         # 3 and 1 pixels into the last tile; three channel blocks
    (3, 256, 25, 25, 512, True, True, True),          # 147 tiles (a partial block of 19), eight channel blocks
    (5, 256, 14, 14, 256, False, True, True),         # mask-head RoIs: 16 tiles per image
    (1, 128, 9, 5, 64, False, False, False),          # 6 tiles: one block, mostly dead rows
    (2, 512, 25, 25, 512, True, False, True),         # res5 conv2 (round 5): 512 input channels = 16 chunks per plane, four ring passes
    (1, 512, 13, 19, 128, False, True, False),        # 512 channels, partial tiles on both axes
]


@pytest.mark.parametrize("case", WINO_FOLD_CASES)
def test_winograd_f4x4_folded_conv_matches_torch(case):
    """wino43_fused_kernel (wino_fused.hip): the 36 plane contractions AND the output transform in one launch — each plane's
    product is folded into the 4x4 outputs as its k loop ends, the M planes never reach memory. Against torch float64 with the
    SAME stated bound as every fp32 convolution here, |err| <= 5e-5 * max|ref| (the fold associates the transform sums plane by
    plane instead of column by column: measured at the level of the three-launch F(4x4) form); deterministic; within 2e-5 * max
    of the three-launch form; both pipeline depths (8 LDS stages at 256 input channels, 4 at 128)."""
    import os
    from treedetection_amd import _lib
    from tests.gpu_util import dev
    B, Cin, H, W, Cout, use_scale, use_bias, relu = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = np.maximum(rng.standard_normal((B, Cin, H, W), dtype=np.float32), -0.5)
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

    def run(fold, nwt=None):
        os.environ["TD_WINO_TILE"] = "4"
        os.environ["TD_WINO_FOLD"] = "1" if fold else "0"
        if nwt:
            os.environ["TD_WF_NWT"] = str(nwt)
        try:
            y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32, device="cuda")
            _lib.check(lib.td_conv2d_winograd_nhwc(p(xd), p(wd), p(sd), p(bd), y.data_ptr(), B, H, W, Cin, Cout, int(relu), _lib.stream_ptr()),
                       "td_conv2d_winograd_nhwc")
            return y
        finally:
            del os.environ["TD_WINO_TILE"], os.environ["TD_WINO_FOLD"]
            os.environ.pop("TD_WF_NWT", None)
    ys = [run(True) for _ in range(3)]
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])          # a racy pipeline would not repeat
    # round 5: the block geometries (64-tile blocks of 512 threads / 32-tile blocks of 256 threads on a 4-stage ring, whatever the
    # launcher picked by the launch's size) give the SAME BITS: every output sees the same chunks in the same order
    for nwt in (2, 4):
        if nwt == 4 and Cin == 512:
            continue                                   # 512 channels run on the 32-tile geometry only
        yv = run(True, nwt)
        assert torch.equal(yv, ys[0]), f"block geometry NWT = {nwt} differs from the launcher's choice"
    got = ys[0].cpu().numpy().transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    scale_ref = max(1.0, np.abs(ref).max())
    err = np.abs(got - ref).max()
    three = run(False).cpu().numpy().transpose(0, 3, 1, 2)
    err3 = np.abs(three - ref).max()
    print(f"\n[F(4x4) folded] {case}: max abs err {err:.3e} = {err / scale_ref:.2e} of max|ref| (three-launch form {err3 / scale_ref:.2e})")
    assert err <= 5e-5 * scale_ref, f"max abs err {err}"
    assert np.abs(got - three).max() <= 2e-5 * scale_ref


# ---- fused bottleneck tail (bottleneck.hip): conv2 3x3 + BN + ReLU → conv3 1x1 + BN + shortcut + ReLU in one launch ----
TAIL_CASES = [  # (precision, B, H, W, mid)
    (0, 2, 37, 41, 64),       # fp32 res2 shape class; M = 3034 is not a multiple of the 128-row tile
    (0, 1, 200, 200, 64),     # one res2 image at BASELINE size
    (1, 2, 37, 41, 64),       # fp16: the wave-private epilogue with the shortcut prefetch (two output pieces), ragged last block
    (1, 1, 200, 200, 64),     # fp16 res2 image at BASELINE size
    (1, 1, 100, 100, 128),    # fp16 res3 shape class (four output pieces: the rolling shortcut prefetch)
    (1, 3, 13, 9, 128),       # fewer rows than one tile per image, padding everywhere
]


@pytest.mark.parametrize("case", TAIL_CASES)
def test_bottleneck_tail_equals_the_two_launches_bit_for_bit(case):
    """td_bottleneck_tail_nhwc == td_conv2d_nhwc(3x3, BN, ReLU) followed by td_conv2d_nhwc(1x1, BN, + shortcut, ReLU), BIT FOR BIT
    (same k order in both contractions, the same single IEEE operations in both epilogues, the same fp16 rounding of the mid
    tensor in the fp16 engine) — and therefore within the conv tolerances of this file against torch (checked in fp64 below)."""
    import torch.nn.functional as F
    from treedetection_amd import _lib
    prec, B, H, W, mid = case
    cout = 4 * mid
    rng = np.random.default_rng(B * 1000 + H * 10 + mid + prec)
    x = rng.normal(0, 1, (B, mid, H, W)).astype(np.float32)
    x = np.maximum(x, 0)                                               # the block's first 1x1 ends in a ReLU
    w2 = (rng.normal(0, 1, (mid, mid, 3, 3)) * np.sqrt(2.0 / (9 * mid))).astype(np.float32)
    w3 = (rng.normal(0, 1, (cout, mid, 1, 1)) * np.sqrt(1.0 / mid)).astype(np.float32)
    s2, b2 = rng.uniform(0.8, 1.2, mid).astype(np.float32), rng.normal(0, 0.2, mid).astype(np.float32)
    s3, b3 = rng.uniform(0.8, 1.2, cout).astype(np.float32), rng.normal(0, 0.2, cout).astype(np.float32)
    sc = rng.normal(0, 1, (B, cout, H, W)).astype(np.float32)
    # two launches through the op-level entry (what the engine ran before the fused kernel)
    t2 = conv2d_hip(x, w2, scale=s2, bias=b2, pad=1, relu=True, precision=prec)
    two = conv2d_hip(t2, w3, scale=s3, bias=b3, residual_nchw=sc, relu=True, precision=prec)
    # fused
    lib = _lib.load()
    dt = torch.float32 if prec == 0 else torch.float16
    from tests.gpu_util import dev
    dx = dev(np.transpose(x, (0, 2, 3, 1)), dt)
    dw2 = dev(np.transpose(w2, (0, 2, 3, 1)), dt)
    dw3 = dev(w3.reshape(cout, mid), dt)
    dsc = dev(np.transpose(sc, (0, 2, 3, 1)), dt)
    ds2, db2, ds3, db3 = (dev(v, torch.float32) for v in (s2, b2, s3, b3))
    y = torch.full((B, H, W, cout), float("nan"), dtype=dt, device="cuda")
    _lib.check(lib.td_bottleneck_tail_nhwc(dx.data_ptr(), dw2.data_ptr(), ds2.data_ptr(), db2.data_ptr(), dw3.data_ptr(), ds3.data_ptr(),
                                           db3.data_ptr(), dsc.data_ptr(), y.data_ptr(), B, H, W, mid, cout, prec, _lib.stream_ptr()),
               "td_bottleneck_tail_nhwc")
    torch.cuda.synchronize()
    got = y.float().cpu().numpy().transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    assert np.array_equal(got, two), float(np.abs(got - two).max())
    # and against torch in float64 (fp16: inputs rounded as the kernel sees them)
    rd = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == 1 else (lambda a: a.astype(np.float64))
    t = F.conv2d(torch.from_numpy(rd(x)), torch.from_numpy(rd(w2)), padding=1) * torch.from_numpy(s2.astype(np.float64))[None, :, None, None] \
        + torch.from_numpy(b2.astype(np.float64))[None, :, None, None]
    t = torch.relu(t)
    if prec == 1:
        t = t.to(torch.float16).to(torch.float64)
    r = F.conv2d(t, torch.from_numpy(rd(w3))) * torch.from_numpy(s3.astype(np.float64))[None, :, None, None] \
        + torch.from_numpy(b3.astype(np.float64))[None, :, None, None] + torch.from_numpy(rd(sc))
    ref = torch.relu(r).numpy()
    tol = 5e-5 if prec == 0 else 4e-3
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), float(np.abs(got - ref).max() / np.abs(ref).max())


def test_engine_with_fused_tail_equals_engine_without_it():
    """The engine with the res2 (fp16: res2 + res3) tails fused (TD_FUSE_TAIL=2; the fp32 engine fuses res2 by default) against
    TD_FUSE_TAIL=0 on the same inputs: every output bit for bit, both precisions."""
    import os
    from tests.test_engine_gpu import smooth_image
    from treedetection_amd.engine import Engine
    from treedetection_amd.weights import make_synthetic_state_dict
    sd = make_synthetic_state_dict(50, seed=5)
    rng = np.random.default_rng(2)
    inputs = [{"image": smooth_image(rng, 256, 320), "height": 320, "width": 400},
              {"image": smooth_image(rng, 224, 256), "height": 224, "width": 256}]
    for prec in ("fp32", "fp16"):
        outs = []
        for env in ("2", "0"):
            os.environ["TD_FUSE_TAIL"] = env
            os.environ["TD_STREAMK"] = "0"       # at this small image size the stream-K rule would take the unfused 3x3s (another association)
            try:
                eng = Engine(sd, precision=prec)
            finally:
                os.environ.pop("TD_FUSE_TAIL", None)
                os.environ.pop("TD_STREAMK", None)
            got = eng(inputs)
            res3 = eng.tensor("res3").float().cpu().numpy()
            outs.append((got, res3))
            eng.close()
        assert np.array_equal(outs[0][1], outs[1][1]), prec
        for a, b in zip(outs[0][0], outs[1][0]):
            for k in ("pred_boxes", "scores", "mask_probs", "pred_masks"):
                assert np.array_equal(a[k], b[k]), (prec, k)


def test_retired_tile_ids_are_refused():
    """Stream-K (ids 21 / 22) and the 4-wave 256x256 tile (id 28) lost to the block tiles in round 3 and left the product library
    (round 6: deleted from the tree, the findings are in DESIGN.md): the entry point says so instead of silently running another tile."""
    x = np.zeros((1, 64, 8, 8), np.float32)
    w = np.zeros((64, 64, 1, 1), np.float32)
    for cfg in (21, 22, 28):
        with pytest.raises(Exception, match="experiment"):
            conv2d_hip(x, w, tile_cfg=cfg, precision=1)


# ---- filter-direct tiles (conv_bdirect.hip): tile_cfg 23 = 64 x 256, 24 = 64 x 128; A through LDS-DMA, filter fragments from a
# fragment-ordered copy of the bank straight into registers ----
@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("cfg", [23, 24, 25, 26, 27, 29, 30, 33])
@pytest.mark.parametrize("case", PP8_CASES + [(8, 256, 50, 50, 256, 3, 1, 1, 0, True), (8, 2048, 25, 25, 512, 1, 1, 0, 0, True),
                                              (2, 256, 13, 13, 15, 1, 1, 0, 0, False),
                                              (1, 64, 24, 24, 128, 1, 1, 0, 1, True),       # shortcut prefetch with a single fp16 k-step
                                              (2, 128, 9, 9, 20, 1, 1, 0, 1, False),        # shortcut, 20 channels (not a multiple of 8): the general epilogue
                                              (2, 256, 31, 17, 512, 1, 1, 0, 1, True),      # shortcut, ragged last row tile, four column tiles
                                              (1, 256, 16, 16, 256, 1, 1, 0, 2, False),     # filter-stationary tile (33): half-resolution shortcut
                                              (2, 128, 23, 19, 384, 1, 1, 0, 0, False),     # 33: no shortcut, a half-empty second column tile
                                              (8, 128, 100, 100, 512, 1, 1, 0, 1, True),    # 33: res3 conv3 at BASELINE size (10 row tiles per block)
                                              (1, 64, 9, 7, 256, 1, 1, 0, 1, True)])        # 33: one row tile in all (most blocks idle)
def test_conv_filter_direct_equals_the_reference_tile_bit_for_bit(case, cfg, prec):
    """conv_bd_kernel keeps the k order, the MFMA and the epilogue of conv_igemm_kernel: on the same inputs (fp16 and fp32) its
    output is IDENTICAL to the 128 x 128 tile's (cfg 0) — which is what lets the engine's tuner choose it per layer by measurement."""
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
    kw = dict(scale=scale, bias=bias, residual_nchw=r, res_shift=1 if res == 2 else 0, stride=stride, pad=pad, relu=relu, precision=prec)
    ref = conv2d_hip(x, w, tile_cfg=0, **kw)
    for _ in range(2):
        got = conv2d_hip(x, w, tile_cfg=cfg, **kw)
        assert got.shape == ref.shape and np.array_equal(got, ref), float(np.abs(got - ref).max())
    assert np.abs(ref).max() > 0.5


# ---- fused 1x1 head (ConvArgs::head_w): the RPN's 3x3 conv + ReLU and its 15-row objectness / delta head in one launch ----
@pytest.mark.parametrize("cfg", [9, 10, 12, 13, 17, 23, 27, 29])
@pytest.mark.parametrize("case", [(2, 256, 50, 50, 3, 1, 15), (1, 256, 200, 200, 3, 1, 15), (8, 256, 13, 13, 3, 1, 15), (1, 64, 37, 21, 1, 0, 32),
                                  (3, 128, 25, 25, 3, 1, 6)])
def test_conv_with_fused_head_equals_the_two_launches_bit_for_bit(case, cfg):
    """Every block tile that owns all 256 output channels can contract the head from its finished fp16 tile: the head's output must
    be IDENTICAL to conv (fp16 output, cfg 0) followed by the 1x1 head as a launch of its own (fp16 in, fp32 out) — same fp16
    rounding of the intermediate, same k order, same MFMA. Rows past M, head columns past head_n and the zero padding are covered
    by the odd shapes. A tile that cannot fuse (here: a single-stage one) is refused, not silently run unfused."""
    from tests.gpu_util import conv2d_head_hip
    from treedetection_amd import _lib
    B, Cin, H, W, k, pad, n = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = rng.standard_normal((256, Cin, k, k), dtype=np.float32) / np.float32(np.sqrt(Cin * k * k))
    bias = rng.standard_normal(256).astype(np.float32)
    hw = rng.standard_normal((n, 256, 1, 1), dtype=np.float32) / np.float32(16.0)
    hb = rng.standard_normal(n).astype(np.float32)
    t = conv2d_hip(x, w, bias=bias, pad=pad, relu=True, precision=1, tile_cfg=0)              # fp16 values as float32
    ref = conv2d_hip(t, hw, bias=hb, precision=1, tile_cfg=3, out_f32=True)
    for _ in range(2):
        got = conv2d_head_hip(x, w, bias, hw, hb, pad, cfg)
        assert got.shape == ref.shape and not np.isnan(got).any()
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    assert np.abs(ref).max() > 0.5
    with pytest.raises(_lib.TdError):
        conv2d_head_hip(x, w, bias, hw, hb, pad, 14)


def test_engine_with_fused_rpn_head_equals_engine_without_it():
    """TD_FUSE_HEAD=0 runs the RPN head as its own launch at every level; the default fuses it where the measured tile allows (fp16)
    / into the F(4x4) output transform (fp32). Same detections, boxes, scores and mask probabilities bit for bit, both engines."""
    import os
    from treedetection_amd.engine import Engine, INPUT_U8_HWC
    from treedetection_amd.synth import make_tile
    from treedetection_amd.weights import make_synthetic_state_dict
    sd = make_synthetic_state_dict(50, seed=0)
    tiles = [torch.from_numpy(make_tile(i, 600)[0]).cuda() for i in range(2)]
    for precision in ("fp16", "fp32"):
        _fused_head_engines_agree(sd, tiles, precision)


def _fused_head_engines_agree(sd, tiles, precision):
    import os
    from treedetection_amd.engine import Engine, INPUT_U8_HWC
    outs = []
    for flag in ("1", "0"):
        os.environ["TD_FUSE_HEAD"] = flag
        try:
            eng = Engine(sd, device=0, precision=precision)
            images, hw_valid, hw_out = eng.preprocess_tiles_u8(tiles)
            o = eng.alloc_outputs(2, 600, 600, paste=False)
            eng.forward_raw(images, INPUT_U8_HWC, hw_valid, hw_out, o)
            eng.forward_raw(images, INPUT_U8_HWC, hw_valid, hw_out, o)          # second pass: measured tile choices in use
            torch.cuda.synchronize()
            outs.append({k: v.cpu().numpy().copy() for k, v in o.items()})
            eng.close()
        finally:
            os.environ.pop("TD_FUSE_HEAD", None)
    a, b = outs
    assert a["count"].sum() > 0 and np.array_equal(a["count"], b["count"])
    for k in ("boxes", "scores", "mask_probs"):
        for i in range(2):
            n = int(a["count"][i])
            assert np.array_equal(a[k][i][:n], b[k][i][:n]), k


@pytest.mark.parametrize("case", [(2, 256, 50, 50, 15), (1, 128, 200, 200, 15), (8, 256, 13, 13, 15), (3, 64, 37, 21, 32), (1, 256, 25, 25, 6)])
def test_winograd_output_transform_with_fused_head_equals_the_two_launches_bit_for_bit(case):
    """fp32 engine: the RPN head rides in the F(4x4) output transform (wino43_output_head_kernel). Its output must be IDENTICAL to
    the F(4x4) layer written out followed by the 1x1 head as a conv_igemm launch — the head's k chunks, sub-steps and MFMA order
    are those of conv_igemm_kernel. Partial tiles (sizes not divisible by 4) and tile counts not divisible by 4 included."""
    import os
    from treedetection_amd import _lib
    from tests.gpu_util import dev
    B, Cin, H, W, n = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = np.maximum(rng.standard_normal((B, Cin, H, W), dtype=np.float32), -0.5)
    w = rng.standard_normal((256, Cin, 3, 3), dtype=np.float32) / np.float32(np.sqrt(Cin * 9))
    bias = rng.standard_normal(256).astype(np.float32)
    hw = rng.standard_normal((n, 256, 1, 1), dtype=np.float32) / np.float32(16.0)
    hb = rng.standard_normal(n).astype(np.float32)
    lib = _lib.load()
    xd, wd, bd = dev(x.transpose(0, 2, 3, 1)), dev(w.transpose(0, 2, 3, 1)), dev(bias)
    hwd, hbd = dev(hw.reshape(n, 256)), dev(hb)
    os.environ["TD_WINO_TILE"] = "4"
    try:
        y = torch.empty((B, H, W, 256), dtype=torch.float32, device="cuda")
        _lib.check(lib.td_conv2d_winograd_nhwc(xd.data_ptr(), wd.data_ptr(), None, bd.data_ptr(), y.data_ptr(), B, H, W, Cin, 256, 1, _lib.stream_ptr()),
                   "td_conv2d_winograd_nhwc")
    finally:
        del os.environ["TD_WINO_TILE"]
    ref = conv2d_hip(y.cpu().numpy().transpose(0, 3, 1, 2), hw, bias=hb, precision=0, tile_cfg=3)
    for _ in range(2):
        got = torch.full((B, H, W, n), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.td_conv2d_winograd_head_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), hwd.data_ptr(), hbd.data_ptr(), got.data_ptr(),
                                                    B, H, W, Cin, n, _lib.stream_ptr()), "td_conv2d_winograd_head_nhwc")
        torch.cuda.synchronize()
        g = got.cpu().numpy().transpose(0, 3, 1, 2)
        assert not np.isnan(g).any() and np.array_equal(g, ref), float(np.abs(g - ref).max())
    assert np.abs(ref).max() > 0.5


def test_engine_with_grouped_pyramid_launches_equals_engine_without_them():
    """TD_GROUP_LEVELS=0: the FPN output convs and the RPN conv + head run level by level; the default (fp16 engine) runs each family
    as ONE conv_pp8_kernel grid over the levels, laterals first. Every pyramid map, every RPN head output and the detections are
    bit-identical (batch of 2, odd tile size → maps whose row counts are not multiples of the 256-row tile)."""
    import os
    from treedetection_amd.engine import Engine, INPUT_U8_HWC
    from treedetection_amd.synth import make_tile
    from treedetection_amd.weights import make_synthetic_state_dict
    sd = make_synthetic_state_dict(50, seed=0)
    tiles = [torch.from_numpy(make_tile(i, 650)[0]).cuda() for i in range(2)]
    outs, named = [], []
    for flag in ("1", "0"):
        os.environ["TD_GROUP_LEVELS"] = flag
        try:
            eng = Engine(sd, device=0, precision="fp16")
            images, hw_valid, hw_out = eng.preprocess_tiles_u8(tiles)
            o = eng.alloc_outputs(2, 650, 650, paste=False)
            eng.forward_raw(images, INPUT_U8_HWC, hw_valid, hw_out, o)
            eng.forward_raw(images, INPUT_U8_HWC, hw_valid, hw_out, o)
            torch.cuda.synchronize()
            outs.append({k: v.cpu().numpy().copy() for k, v in o.items()})
            named.append({nm: eng.tensor(nm).float().cpu().numpy().copy() for nm in ["p2", "p3", "p4", "p5", "p6"] + [f"rpn_head{l}" for l in range(2, 7)]})
            eng.close()
        finally:
            os.environ.pop("TD_GROUP_LEVELS", None)
    for nm in named[0]:
        assert named[0][nm].shape == named[1][nm].shape and np.array_equal(named[0][nm], named[1][nm]), nm
    a, b = outs
    assert a["count"].sum() > 0 and np.array_equal(a["count"], b["count"])
    for k in ("boxes", "scores", "mask_probs"):
        for i in range(2):
            n = int(a["count"][i])
            assert np.array_equal(a[k][i][:n], b[k][i][:n]), k
