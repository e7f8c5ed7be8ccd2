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
