"""Resume bookkeeping of the prediction stage, file-format compatible with the reference
(TreeDetection/recoveries.py:5-108; called at detection.py:103-110,132).

``prediction_recovery.yaml`` = ``{model_path: str, files: {<tif path>: [<tile id>, ...]}}``. A file counts as done
when its output folder holds as many ``Prediction_*.json`` files as it has tile ids — all of them, or those left
after the exclude flags. Like the reference, ``load`` returns an EMPTY to-do list plus the set of finished files
(the caller filters with the set; reference recoveries.py:21,66 and detection.py:105-106).
"""
from __future__ import annotations

import json
import os

import yaml

RECOVERY_NAME = "prediction_recovery.yaml"


def _stem(path: str) -> str:
    return os.path.basename(path).replace(".tif", "")


def load_prediction_recovery_data(output_path, tiles_path, model_path, logger, exclude=None):
    recovery_file = os.path.join(output_path, RECOVERY_NAME)
    done = set()
    todo = []
    if not os.path.exists(recovery_file):
        return todo, done
    with open(recovery_file, "r") as f:
        state = yaml.safe_load(f) or {}
    if state.get("model_path") != model_path:
        logger.warning("Model path does not match the one stored in the recovery file. Skipping recovery.")
        return todo, done
    for tif, tile_ids in (state.get("files") or {}).items():
        meta = os.path.join(tiles_path, _stem(tif) + ".json")
        if not os.path.exists(meta):
            logger.debug(f"Missing JSON metadata for {tif}. Skipping.")
            continue
        n_out = len(os.listdir(os.path.join(output_path, _stem(tif))))
        if n_out == len(tile_ids):
            done.add(tif)
            continue
        with open(meta, "r") as jf:
            tiles = json.load(jf)
        if exclude:
            expected = [k for k, v in tiles.items() if not any(v.get(flag, False) for flag in exclude)]
        else:
            expected = list(tiles.keys())
        if n_out == len(expected):
            done.add(tif)
        else:
            logger.debug(f"Mismatch between output folder and JSON (after excludes) for {tif}.")
    return todo, done


def save_prediction_recovery_data(output_path, tiles_path, model_path, processed_files, file_list):
    state = {"model_path": model_path, "files": {}}
    try:
        for tif in list(file_list) + list(processed_files):
            meta = os.path.join(tiles_path, _stem(tif) + ".json")
            if os.path.exists(meta):
                with open(meta, "r") as jf:
                    state["files"][tif] = list(json.load(jf).keys())
        with open(os.path.join(output_path, RECOVERY_NAME), "w") as f:
            yaml.safe_dump(state, f, sort_keys=False)
    except Exception as e:   # the reference swallows and prints (recoveries.py:107-108)
        print(f"Failed to save prediction recovery file: {e}")


# ---- stitching resume file (reference recoveries.py:111-144) -----------------------------------------------
STITCHING_RECOVERY_NAME = "stitching_recovery.yaml"


def load_stitching_recovery(output_path, logger=None):
    """``stitching_recovery.yaml`` = ``{completed_files: [<image stem>, ...]}`` → set of stems (empty when the file is
    missing or unreadable; a read error is logged, not raised)."""
    recovery_file = os.path.join(output_path, STITCHING_RECOVERY_NAME)
    completed = set()
    if os.path.exists(recovery_file):
        try:
            with open(recovery_file) as f:
                data = yaml.safe_load(f)
            if data and "completed_files" in data:
                completed = set(os.path.basename(p) for p in data["completed_files"])
            if logger:
                logger.info(f"Loaded {len(completed)} completed files from recovery.")
        except Exception as e:
            if logger:
                logger.warning(f"Failed to load stitching recovery: {e}")
    return completed


def save_stitching_recovery(output_path, results, logger=None):
    recovery_file = os.path.join(output_path, STITCHING_RECOVERY_NAME)
    try:
        stems = sorted({os.path.splitext(os.path.basename(r))[0] for r in results if r is not None})
        with open(recovery_file, "w") as f:
            yaml.safe_dump({"completed_files": stems}, f, sort_keys=False)
        if logger:
            logger.info(f"Saved recovery with {len(stems)} files.")
    except Exception as e:
        if logger:
            logger.error(f"Failed to save stitching recovery: {e}")


# ---- fusion resume file (reference recoveries.py:251-284) --------------------------------------------------
FUSION_RECOVERY_NAME = "fusion_recovery.yaml"


def load_fusion_recovery(output_dir, logger=None):
    recovery_file = os.path.join(output_dir, FUSION_RECOVERY_NAME)
    completed = set()
    if os.path.exists(recovery_file):
        try:
            with open(recovery_file) as f:
                data = yaml.safe_load(f)
            if data and "completed_files" in data:
                completed = set(os.path.basename(p) for p in data["completed_files"])
            if logger:
                logger.info(f"Loaded {len(completed)} completed fusion files from recovery.")
        except Exception as e:
            if logger:
                logger.warning(f"Failed to load fusion recovery: {e}")
    return completed


def save_fusion_recovery(output_dir, results, logger=None):
    recovery_file = os.path.join(output_dir, FUSION_RECOVERY_NAME)
    try:
        stems = sorted({os.path.splitext(os.path.basename(r))[0] for r in results if r is not None})
        with open(recovery_file, "w") as f:
            yaml.safe_dump({"completed_files": stems}, f, sort_keys=False)
        if logger:
            logger.info(f"Saved fusion recovery with {len(stems)} files.")
    except Exception as e:
        if logger:
            logger.warning(f"Failed to save fusion recovery: {e}")
