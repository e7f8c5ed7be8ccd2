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


# ---- "which images are finished" lists of the stitching and the fusion stage ----------------------------------
# On disk (the format is the contract: reference recoveries.py:111-144 and 251-284 write the same one-key mapping, and
# tests/golden/recovery_fixture.json pins it against files the reference's own module wrote):
#     <stage>_recovery.yaml = {completed_files: [<image stem>, ...]}      stems sorted, no folder, no extension
# Both stages share one ledger class; the four module-level functions the rest of the package calls are its bound faces.
class _CompletedLedger:
    def __init__(self, file_name: str, label: str, failure_level: str):
        self.file_name, self.label, self.failure_level = file_name, label, failure_level

    @staticmethod
    def _say(logger, level: str, text: str) -> None:
        if logger is not None:
            getattr(logger, level)(text)

    def read(self, folder, logger=None) -> set:
        """→ the set of finished stems; a missing, empty or unreadable ledger is an empty set (a read error is logged, never raised)."""
        path = os.path.join(folder, self.file_name)
        if not os.path.exists(path):
            return set()
        try:
            with open(path) as f:
                entries = (yaml.safe_load(f) or {}).get("completed_files") or []
        except Exception as e:
            self._say(logger, "warning", f"Failed to load {self.label} recovery: {e}")
            return set()
        self._say(logger, "info", f"Loaded {len(entries)} completed {self.label} files from recovery.")
        return {os.path.basename(str(p)) for p in entries}

    def write(self, folder, results, logger=None) -> None:
        """``results``: paths or stems of finished images (None entries = failed ones, skipped). Errors are logged, never raised."""
        stems = sorted({os.path.splitext(os.path.basename(str(r)))[0] for r in results if r is not None})
        try:
            with open(os.path.join(folder, self.file_name), "w") as f:
                yaml.safe_dump({"completed_files": stems}, f, sort_keys=False)
        except Exception as e:
            self._say(logger, self.failure_level, f"Failed to save {self.label} recovery: {e}")
            return
        self._say(logger, "info", f"Saved {self.label} recovery with {len(stems)} files.")


STITCHING_RECOVERY_NAME = "stitching_recovery.yaml"
FUSION_RECOVERY_NAME = "fusion_recovery.yaml"
_stitching = _CompletedLedger(STITCHING_RECOVERY_NAME, "stitching", "error")
_fusion = _CompletedLedger(FUSION_RECOVERY_NAME, "fusion", "warning")
load_stitching_recovery, save_stitching_recovery = _stitching.read, _stitching.write
load_fusion_recovery, save_fusion_recovery = _fusion.read, _fusion.write
