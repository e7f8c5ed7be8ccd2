"""A mask head whose OUTPUT is a compact blob (what a trained crown segmenter produces), for the fp16 mask-IoU statement.

The constructor lives in the package (treedetection_amd/weights.py, where it is documented: bench.py's `e2e_crowns` region
uses it too); this module keeps the tests' import path."""
from treedetection_amd.weights import BLOB_AMP as AMP, BLOB_GAIN as GAIN, BLOB_LEVEL as LEVEL, BLOB_NF as NF  # noqa: F401
from treedetection_amd.weights import blob_mask_head, boundary_over_area  # noqa: F401
