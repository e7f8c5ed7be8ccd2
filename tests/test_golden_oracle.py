"""The oracle must still reproduce the frozen vectors of tests/golden/oracle_small.npz (generator:
tests/golden/make_oracle_fixture.py). An edit of oracle/*.py that changes any result fails here, on the CPU, before
it can move the target of the GPU parity tests. Integer / index / bit outputs are compared exactly; floating-point
outputs of the torch-CPU convolutions within 1e-4 relative (thread count and ISA change their summation order)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_oracle_fixture", os.path.join(HERE, "golden", "make_oracle_fixture.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def pair():
    gold = dict(np.load(os.path.join(HERE, "golden", "oracle_small.npz")))
    return gold, _gen().compute()


EXACT = ("nms_", "roi_levels", "paste_bits", "resize_", "contour_", "img0_topk_idx", "img1_topk_idx", "img0_rpn_keep",
         "img1_rpn_keep", "img0_det_keep", "img1_det_keep", "img0_proposal_count", "img1_proposal_count",
         "img0_mask_shape", "img1_mask_shape")


def test_same_arrays(pair):
    gold, now = pair
    assert sorted(gold) == sorted(now)
    for k in gold:
        assert gold[k].shape == now[k].shape and gold[k].dtype == now[k].dtype, k


def test_exact_vectors(pair):
    gold, now = pair
    n = 0
    for k in gold:
        if k.startswith(EXACT):
            assert np.array_equal(gold[k], now[k]), k
            n += 1
    assert n >= 20


def test_float_vectors(pair):
    gold, now = pair
    for k in gold:
        if k.startswith(EXACT) or k.endswith("mask_bits"):
            continue
        scale = max(1.0, float(np.abs(gold[k]).max()))
        assert np.abs(gold[k].astype(np.float64) - now[k].astype(np.float64)).max() <= 1e-4 * scale, k


def test_pasted_masks(pair):
    gold, now = pair
    for n in range(2):
        a = np.unpackbits(gold[f"img{n}_mask_bits"])
        b = np.unpackbits(now[f"img{n}_mask_bits"])
        assert a.size == b.size
        # a probability within float rounding of 0.5 may flip a pixel between machines; nothing more
        assert (a != b).sum() <= 1e-5 * a.size, n


def test_known_values_in_the_fixture(pair):
    """Spot values that follow from the published definitions (not from the oracle): they pin the FILE."""
    gold, _ = pair
    # ResizeShortestEdge(800, 1333): 1000x1000 → 800x800; 350x450 → 800x1029; 300x900 → 444x1333 (max-size cap)
    assert gold["resize_shape"].tolist() == [[800, 800], [800, 800], [800, 1029], [444, 1333], [800, 1080]]
    # level mapper floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [2,5] → index 0..3
    assert gold["roi_levels"].tolist() == [0, 0, 1, 2, 2, 3, 3]
    # anchors: size 128, ratios (.5,1,2), stride 16, offset 0 → first cell anchor of ratio 1 is (-64,-64,64,64)
    assert np.allclose(gold["anchors_3x4_s16"][1], [-64, -64, 64, 64])
    # dw = 9 > log(1000/16) is clamped: width = anchor width * 1000/16
    a = gold["anchors_3x4_s16"][0]
    w = gold["decode_rpn"][0, 2] - gold["decode_rpn"][0, 0]
    assert abs(w - (a[2] - a[0]) * 1000.0 / 16.0) <= 1e-2 * w
    # duplicates 50..59 of boxes 40..49 never survive NMS together
    keep = set(gold["nms_keep_05"].tolist())
    assert all(not (i in keep and i + 10 in keep) for i in range(40, 50))


def test_trained_head_fixtures_are_the_committed_data():
    """tests/golden/trained_heads_*.npz (round 6): made ONCE by tests/golden/make_trained_heads.py on the CPU; the GPU tests load them
    and train nothing. Here: every file matches the manifest's SHA-256, holds exactly the tensors the loader expects, and rebuilds
    a state dict that differs from the seeded one ONLY in the RPN output layers, fc2 and the two predictors (fc1 stays frozen)."""
    import os
    from tests import trained_heads as TH
    from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict
    man = TH.manifest()
    assert sorted(man) == sorted(f"trained_heads_{n}.npz" for n in TH.FIXTURES)
    for name, (depth, seed, tiles) in TH.FIXTURES.items():
        path = TH.fixture_path(name)
        assert TH._sha256(path) == man[os.path.basename(path)], name
        assert os.path.getsize(path) < 1 << 20
    base = blob_mask_head(make_synthetic_state_dict(50, seed=5))
    sd = TH.load_trained_heads("r50", base=base)
    changed = sorted(k for k in base if not np.array_equal(base[k], sd[k]))
    assert changed == sorted(["proposal_generator.rpn_head.objectness_logits.weight", "proposal_generator.rpn_head.objectness_logits.bias",
                              "proposal_generator.rpn_head.anchor_deltas.weight", "proposal_generator.rpn_head.anchor_deltas.bias",
                              "roi_heads.box_head.fc2.weight", "roi_heads.box_head.fc2.bias",
                              "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.cls_score.bias",
                              "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_predictor.bbox_pred.bias"]), changed
    assert sorted(sd) == sorted(base) and all(sd[k].dtype == base[k].dtype and sd[k].shape == base[k].shape for k in base)
    delta = sd["roi_heads.box_head.fc2.weight"].astype(np.float64) - base["roi_heads.box_head.fc2.weight"]
    assert np.linalg.matrix_rank(delta, tol=1e-4) <= 64                      # fc2 = seeded + a rank-64 delta
    # tampering is caught: a file that is not the manifest's is refused
    with pytest.raises(AssertionError, match="sha256"):
        real = TH.manifest
        try:
            TH.manifest = lambda: {k: "0" * 64 for k in real()}
            TH.load_trained_heads("r50", base=base)
        finally:
            TH.manifest = real
