"""treedetection_amd — MI355X-native drop-in for the prediction stage of Jonetz/TreeDetection.

Public surface mirrors the reference package (TreeDetection/__init__.py:3-9): process_files, preprocess_files,
predict_tiles, postprocess_files, predict_on_model, Predictor, get_config, setup_model_cfg. The model forward runs in
libtreedet_hip.so (hand-written HIP for gfx950) behind the C ABI of include/treedet.h; there is no CPU fallback.
"""
__version__ = "0.1.0"

from .config import Config, get_config, setup_model_cfg  # noqa: F401
from .detection import (cleanup_files, postprocess_files, predict_on_model, predict_tiles,  # noqa: F401
                        preprocess_files, process_files)
from .engine import Engine  # noqa: F401
from .prediction import Predictor  # noqa: F401
