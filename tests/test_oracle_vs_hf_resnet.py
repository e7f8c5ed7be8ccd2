"""The oracle's ResNet bottom-up (`oracle/maskrcnn_ref.py:backbone`, restating detectron2 `modeling/backbone/resnet.py`:
BasicStem, BottleneckBlock with STRIDE_IN_1X1 = True, FrozenBN) against an INDEPENDENT third-party implementation that is
installed here: HuggingFace `transformers.ResNetModel` configured to the same topology (`downsample_in_bottleneck=True` puts
the stride on the first 1x1 as detectron2's MSRA models do; `downsample_in_first_stage=False` keeps res2 at stride 4). The
same synthetic weights go into both; stem / pool / res2..res5 maps must agree to float32 rounding. This pins stride placement,
padding, the BN fold, the max-pool and the shortcut rule of SURVEY.md §8 row a10 to code the build did not write."""
import numpy as np
import pytest
import torch

from oracle.maskrcnn_ref import MaskRCNNOracle
from treedetection_amd.weights import make_synthetic_state_dict

transformers = pytest.importorskip("transformers")


def _hf_resnet(sd, depths):
    from transformers import ResNetConfig, ResNetModel

    p = "backbone.bottom_up."
    stem = sd[p + "stem.conv1.weight"].shape[0]
    hidden = [sd[p + f"res{s}.0.conv3.weight"].shape[0] for s in (2, 3, 4, 5)]
    cfg = ResNetConfig(num_channels=3, embedding_size=stem, hidden_sizes=hidden, depths=list(depths), layer_type="bottleneck",
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=True)
    model = ResNetModel(cfg).eval()

    def conv_bn(dst, src):
        out = {dst + ".convolution.weight": sd[src + ".weight"]}
        for a, b in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"), ("running_var", "running_var")):
            out[dst + ".normalization." + a] = sd[src + ".norm." + b]
        return out

    new = conv_bn("embedder.embedder", p + "stem.conv1")
    for si, n in enumerate(depths):
        for bi in range(n):
            src, dst = p + f"res{si + 2}.{bi}", f"encoder.stages.{si}.layers.{bi}"
            if bi == 0:
                new.update(conv_bn(dst + ".shortcut", src + ".shortcut"))
            for ci in range(3):
                new.update(conv_bn(dst + f".layer.{ci}", src + f".conv{ci + 1}"))
    target = model.state_dict()
    loaded = {}
    for k, v in new.items():
        assert k in target and tuple(target[k].shape) == tuple(v.shape), (k, v.shape)
        loaded[k] = torch.from_numpy(np.ascontiguousarray(v))
    missing = [k for k in target if k not in loaded and not k.endswith("num_batches_tracked")]
    assert not missing, missing[:5]
    model.load_state_dict(loaded, strict=False)
    return model


@pytest.mark.parametrize("depth,depths,width_div,hw", [(50, (3, 4, 6, 3), 4, (96, 160)), (101, (3, 4, 23, 3), 8, (64, 96)),
                                                      (50, (3, 4, 6, 3), 8, (75, 131))])
def test_oracle_bottom_up_matches_transformers_resnet(depth, depths, width_div, hw):
    sd = make_synthetic_state_dict(depth, seed=11 + depth, width_div=width_div)
    oracle = MaskRCNNOracle(sd)
    assert tuple(oracle.blocks) == tuple(depths)
    rng = np.random.default_rng(depth + width_div)
    x = torch.from_numpy(rng.uniform(-120, 140, (2, 3) + hw).astype(np.float32))
    taps = oracle.backbone(x)
    with torch.no_grad():
        hs = _hf_resnet(sd, depths)(x, output_hidden_states=True).hidden_states
    assert len(hs) == 5
    for name, got in zip(("pool", "res2", "res3", "res4", "res5"), hs):
        ref = taps[name]
        assert tuple(ref.shape) == tuple(got.shape), (name, ref.shape, got.shape)
        scale = float(ref.abs().max())
        assert scale > 0
        err = float((ref - got).abs().max())
        assert err <= 2e-5 * scale, (name, err, scale)      # two float32 evaluation orders of the same BN (folded vs not)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol,width_div", [("fp32", 2e-4, 2), ("fp16", 2e-2, 1)])      # the fp16 kernels want channels in multiples of 64: full width
def test_hip_engine_bottom_up_matches_transformers_resnet(precision, tol, width_div):
    """The HIP engine's stem / res2..res5 maps (td_engine_tensor after td_engine_forward) against `transformers.ResNetModel` on the CPU,
    with NO oracle in between: detectron2's pre-processing written out here (BGR pixel mean 103.530 / 116.280 / 123.675 subtracted,
    std 1, zero padding bottom / right to a multiple of 32 — `config.py:25` loads the COCO Mask R-CNN yaml that sets them). fp32: the
    tolerance `test_engine_gpu.py` holds against the oracle; fp16: storage rounding through 16 bottleneck blocks."""
    from treedetection_amd.engine import Engine

    depths = (3, 4, 6, 3)
    sd = make_synthetic_state_dict(50, seed=61, width_div=width_div)
    rng = np.random.default_rng(8)
    imgs = [rng.uniform(0, 255, (3, 200, 296)).astype(np.float32).round(), rng.uniform(0, 255, (3, 224, 250)).astype(np.float32).round()]
    hp = (max(i.shape[1] for i in imgs) + 31) // 32 * 32
    wp = (max(i.shape[2] for i in imgs) + 31) // 32 * 32
    x = np.zeros((len(imgs), 3, hp, wp), np.float32)
    mean = np.array([103.530, 116.280, 123.675], np.float32).reshape(3, 1, 1)
    for k, im in enumerate(imgs):
        x[k, :, :im.shape[1], :im.shape[2]] = im - mean
    with torch.no_grad():
        hs = _hf_resnet(sd, depths)(torch.from_numpy(x), output_hidden_states=True).hidden_states
    eng = Engine(sd, precision=precision)
    try:
        eng([{"image": im, "height": im.shape[1], "width": im.shape[2]} for im in imgs])
        for name, ref in zip(("pool", "res2", "res3", "res4", "res5"), hs):
            got = eng.tensor(name).float().cpu().numpy().transpose(0, 3, 1, 2)
            ref = ref.numpy()
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            err = float(np.abs(got - ref).max())
            assert err <= tol * float(np.abs(ref).max()), (name, err, float(np.abs(ref).max()))
    finally:
        eng.close()
