"""Pin the oracle's Pillow restatement against Pillow itself (installed here): byte-exact."""
import numpy as np
import pytest
from PIL import Image

from oracle import ops_ref as R


@pytest.mark.parametrize("h,w,oh,ow", [(450, 450, 800, 800), (1000, 1000, 800, 800), (350, 450, 800, 1029),
                                       (97, 211, 800, 1333), (64, 64, 64, 64), (100, 50, 37, 211), (5, 7, 31, 3)])
def test_pil_restatement_is_byte_exact(h, w, oh, ow):
    rng = np.random.default_rng(h * 31 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    got = R.pil_resize_bilinear_u8(img, oh, ow)
    assert np.array_equal(got, ref)


def test_resize_shortest_edge_shapes():
    # detectron2 ResizeShortestEdge(800, 1333): known shapes
    assert R.resize_shortest_edge_shape(450, 450) == (800, 800)
    assert R.resize_shortest_edge_shape(1000, 1000) == (800, 800)
    assert R.resize_shortest_edge_shape(350, 450) == (800, 1029)
    assert R.resize_shortest_edge_shape(480, 640) == (800, 1067)
    assert R.resize_shortest_edge_shape(100, 400) == (333, 1333)
