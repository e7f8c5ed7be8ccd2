"""Configuration surface of the reference, kept as is (TreeDetection/config.py): the same ``config.yml`` keys and
defaults (``get_config``, config.py:144-238), the same device parsing (``set_device_configuration``, 112-142), the
``Config`` singleton (12-23) and ``setup_model_cfg`` (25-66). The latter no longer builds a detectron2 CfgNode
(detectron2 is not a dependency of this package) but returns a plain namespace with the same attribute paths the
reference reads or sets, so ``Predictor(cfg, ...)`` keeps its signature.

Optional extra keys (defaults preserve the reference's behaviour): ``precision`` ("fp32"), ``resnet_depth``
(101 — the reference hard-codes mask_rcnn_R_101_FPN_3x, config.py:25; a checkpoint's own depth wins); multi-rank runs:
``sharded_epilogue`` ("auto" | "rank0" | "local"), ``shard_by`` ("auto" | "image" | "tile": whole images per rank when there
are at least as many images as ranks, detection.resolve_shard_by), ``eager_stitch`` (true: every image is stitched as soon as
its tile files are complete, while the next one predicts), ``fp16_min_batch`` (0 = off: a larger batch for the fp16 engine only,
detection.engine_batch_size), ``device_contours`` ("auto" | true | false: mask borders followed on the GPU while the host epilogue,
not the GPU, sets the batch period; same files), ``device_decode`` ("auto" | false | "all": LZW rasters decoded on the GPU, tile windows cut
in HBM — "all": uncompressed rasters are kept in HBM too; same pixels).
"""
from __future__ import annotations

import logging
import os
import re
import warnings
from datetime import datetime
from types import SimpleNamespace

import yaml


class Config:
    """Process-wide mutable singleton mirroring the config dict as attributes (reference config.py:12-23)."""
    _instance = None

    def __new__(cls):
        if not cls._instance:
            cls._instance = super(Config, cls).__new__(cls)
            cls._instance.state = {}
        return cls._instance

    def _load_into_config(cls, config):
        for key, value in config.items():
            setattr(cls, key, value)


def _cuda_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _cuda_device_count() -> int:
    import torch
    return torch.cuda.device_count()


def setup_model_cfg(base_model="COCO-InstanceSegmentation/mask_rcnn_R_101_FPN_3x.yaml", update_model=None, device="cpu"):
    """Inference-only model configuration (reference config.py:25-66): Mask R-CNN R101-FPN (or whatever depth the
    base-model name carries), 1 class, score > 0.3, NMS 0.5; everything else detectron2's defaults (SURVEY.md
    Appendix A), which libtreedet_hip hard-wires."""
    m = re.search(r"R_(\d+)_FPN", base_model or "")
    depth = int(m.group(1)) if m else 101
    cfg = SimpleNamespace()
    cfg.MODEL = SimpleNamespace()
    cfg.MODEL.BASE = base_model
    cfg.MODEL.WEIGHTS = update_model if update_model else base_model
    cfg.MODEL.RESNETS = SimpleNamespace(DEPTH=depth)
    cfg.MODEL.PIXEL_MEAN = [103.530, 116.280, 123.675]
    cfg.MODEL.PIXEL_STD = [1.0, 1.0, 1.0]
    cfg.MODEL.RPN = SimpleNamespace(PRE_NMS_TOPK_TEST=1000, POST_NMS_TOPK_TEST=1000, NMS_THRESH=0.7)
    cfg.MODEL.ROI_HEADS = SimpleNamespace(NUM_CLASSES=1, SCORE_THRESH_TEST=0.3, NMS_THRESH_TEST=0.5)
    cfg.TEST = SimpleNamespace(DETECTIONS_PER_IMAGE=100)
    cfg.INPUT = SimpleNamespace(MIN_SIZE_TEST=800, MAX_SIZE_TEST=1333, FORMAT="BGR")
    if isinstance(device, int) or (isinstance(device, str) and device.isdigit()):
        gpu_id = int(device)
        if _cuda_available():
            import torch
            from . import distributed as D
            if D.world() > 1:
                # one process per GPU: config.yml names ONE device for all ranks, each rank takes its own
                # (LOCAL_RANK); selecting the configured index here would put every rank on the same GPU
                gpu_id = D.local_device(gpu_id)
            torch.cuda.set_device(gpu_id)
            cfg.MODEL.DEVICE = "cuda"
            cfg.MODEL.DEVICE_INDEX = gpu_id
        else:
            warnings.warn("CUDA not available, falling back to CPU.")
            cfg.MODEL.DEVICE = "cpu"
    elif device == "cuda" or (isinstance(device, str) and device.startswith("cuda:")):
        cfg.MODEL.DEVICE = "cuda"
        cfg.MODEL.DEVICE_INDEX = int(device.split(":")[1]) if ":" in device else 0
    else:
        cfg.MODEL.DEVICE = "cpu"
    cfg.CUDNN_BENCHMARK = True
    cfg.SOLVER = SimpleNamespace(AMP=SimpleNamespace(ENABLED=True))   # training-only switch in the reference
    return cfg


def load_config(config_path: str):
    with open(config_path, "r") as file:
        return yaml.safe_load(file)


def setup_logging(log_path: str, debug: bool):
    os.makedirs(log_path, exist_ok=True)
    log_file_path = os.path.join(log_path, f"logs_{datetime.now().strftime('%Y%m%d_%H%M%S')}.log")
    logging.basicConfig(filename=log_file_path, format="%(asctime)s - %(levelname)s - %(message)s",
                        level=logging.DEBUG if debug else logging.INFO, datefmt="%Y-%m-%d %H:%M:%S")
    logger = logging.getLogger(__name__)
    console_handler = logging.StreamHandler()
    console_handler.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
    logger.addHandler(console_handler)
    return logger


def set_device_configuration(config, raw_device):
    """config["device"] = GPU index as a string, or "cpu" (reference config.py:112-142)."""
    if _cuda_available():
        device_str = "0"
        if raw_device is not None:
            if isinstance(raw_device, int):
                device_str = raw_device
            elif isinstance(raw_device, str) and raw_device.startswith("cuda"):
                rest = raw_device.replace("cuda:", "")
                device_str = rest if rest.isdigit() else "0"
            elif isinstance(raw_device, str) and raw_device.isdigit():
                device_str = raw_device
            try:
                gpu_index = int(device_str)
            except (IndexError, ValueError):
                raise ValueError(f"Invalid CUDA device specification: {raw_device}")
            assert _cuda_device_count() > gpu_index, f"GPU index {gpu_index} is out of range."
        config["device"] = str(device_str)
    else:
        if isinstance(raw_device, str) and raw_device.startswith("cuda"):
            warnings.warn(f"CUDA device '{raw_device}' requested but CUDA is not available. Falling back to CPU.")
        config["device"] = "cpu"


def get_config(config_path: str):
    """YAML → (config dict with the reference's defaults, Config singleton) — reference config.py:144-238."""
    config = load_config(config_path)
    assert config.get("image_directory") and os.path.exists(config.get("image_directory")), \
        "Input path is missing from the configuration or path is incorrect."
    assert config.get("height_data_path") and os.path.exists(config.get("height_data_path")), \
        "nDOM path is missing from the configuration or path is incorrect."
    if not config.get("combined_model") or not os.path.exists(config.get("combined_model")):
        assert config.get("urban_model") and os.path.exists(config.get("urban_model")), \
            "Urban model path is missing from the configuration or path is incorrect."
        assert config.get("forrest_model") and os.path.exists(config.get("forrest_model")), \
            "Forrest model path is missing from the configuration."
        assert config.get("forrest_outline") and os.path.exists(config.get("forrest_outline")), \
            "Forrest outline path is missing from the configuration."
    defaults = {
        "output_directory": "./output", "tiles_path": "./tiles",
        "tile_width": 50, "tile_height": 50, "buffer": 20, "batch_size": 10,
        "use_overlap": True, "overlapping_tiles_width": 3, "overlapping_tiles_height": 3, "merged_path": "merged",
        "image_merged_regex": "FDOP20_(\\d+)_(\\d+)_(\\d+)_(\\d+)_(\\d+)\\.tif",
        "height_data_merged_regex": "FDOP20_(\\d+)_(\\d+)\\.tif",
        "iou_threshold": 0.5, "confidence_threshold_stitching": 0.3, "area_threshold": 1,
        "exclude_files": [], "confidence_threshold": 0.3, "containment_threshold": 0.9, "height_threshold": 3,
        "parallel": True, "num_workers": None, "verbose": False, "debug": False, "keep_intermediate": False,
        "timestamped_output_directory": False, "simplify_tolerance": 0.2, "building_shapes": None,
        # extensions of this package (defaults = the reference's behaviour)
        "precision": "fp32", "resnet_depth": 101, "sharded_epilogue": "auto", "shard_by": "auto", "eager_stitch": True,
        "fp16_min_batch": 0, "device_contours": "auto", "device_decode": "auto",
    }
    for k, v in defaults.items():
        config[k] = config.get(k, v)
    config["continue"] = config.get("continue", os.path.join(config["output_directory"], "continue.yml"))
    set_device_configuration(config, config.get("device", None))
    config["logger"] = setup_logging(os.path.join(config["output_directory"], "logs"), config["debug"])
    config_obj = Config()
    config_obj._load_into_config(config)
    return config, config_obj
