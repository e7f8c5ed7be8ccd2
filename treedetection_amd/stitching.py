"""Consumer of the per-tile prediction files: ``Prediction_*.json`` → one GeoPackage per image
(reference TreeDetection/helpers.py: ``process_and_stitch_predictions`` 556-600, ``process_folder_sync`` 524-554,
``process_prediction_file_sync`` 419-476, ``box_filter`` 305-319 / ``box_make`` 281-303 / ``filename_geoinfo`` 265-279).

Same entry point, arguments, file layout and resume file as the reference; what it runs on is different. The reference
goes through geopandas/shapely/GDAL (not installed here): a GeoDataFrame per tile, ``simplify(preserve_topology=True)``,
``sjoin(..., "within")`` against the tile's shrunken box, ``to_file(driver="GPKG")``. Here each tile file is parsed
once by ``td_stitch_tile_json`` (libtreedet_hip.so, host code, GIL released): rings are simplified by the GEOS
algorithm restated in C++ (``td_simplify_ring``), the *within* test against an axis-parallel box is done on the
vertices, geometries come back already encoded, and the layer is written by :mod:`treedetection_amd.gpkg`. Per feature the output carries what the reference's frame carries: geometry,
``Confidence_score`` and ``filter_index_right`` (always 0 — the index of the single box row the join matched).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading
from concurrent.futures import ThreadPoolExecutor, as_completed
from pathlib import Path
from typing import List, Tuple

import numpy as np

from . import _lib
from .gpkg import write_blobs
from .recoveries import load_stitching_recovery, save_stitching_recovery


def filename_geoinfo(filename) -> Tuple[int, int, int, int, int]:
    """``<stem>_<minx>_<miny>_<width>_<buffer>_<crs>`` → the five integers (helpers.py:265-279)."""
    parts = os.path.basename(str(filename)).replace(".geojson", "").replace(".json", "").replace(".gpkg", "").split("_")
    minx, miny, width, buffer, crs = (int(p) for p in parts[-5:])
    return minx, miny, width, buffer, crs


def box_make(minx: int, miny: int, width: int, buffer: int, crs=None, shift: int = 0) -> Tuple[float, float, float, float]:
    """The tile's full extent (core + buffer) pulled in by ``shift`` on every side (helpers.py:281-303), as
    (minx, miny, maxx, maxy)."""
    return (minx - buffer + shift, miny - buffer + shift, minx + width + buffer - shift, miny + width + buffer - shift)


def box_filter(filename, shift: int = 0) -> Tuple[float, float, float, float]:
    minx, miny, width, buffer, crs = filename_geoinfo(filename)
    return box_make(minx, miny, width, buffer, crs, shift)


def simplify_ring(coords: np.ndarray, tolerance: float) -> np.ndarray:
    """``Polygon(coords).simplify(tolerance, preserve_topology=True)`` for a single shell → [m,2] (td_simplify_ring)."""
    xy = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 2)
    out = np.empty_like(xy)
    m = _lib.load().td_simplify_ring(xy.ctypes.data, xy.shape[0], float(tolerance), out.ctypes.data, out.shape[0])
    _lib.check(m, "td_simplify_ring")
    return out[:m]


def within_box(ring: np.ndarray, box: Tuple[float, float, float, float]) -> bool:
    """``polygon.within(box)``: no vertex outside the closed box (the box is convex) and the interiors meet."""
    minx, miny, maxx, maxy = box
    x, y = ring[:, 0], ring[:, 1]
    if x.min() < minx or x.max() > maxx or y.min() < miny or y.max() > maxy:
        return False
    if ((x > minx) & (x < maxx) & (y > miny) & (y < maxy)).any():
        return True
    return abs(float(np.dot(x[:-1], y[1:]) - np.dot(x[1:], y[:-1]))) > 0.0


class TileFeatures:
    """What one prediction file contributes to its image's layer: GeoPackage geometry blobs (one buffer + offsets),
    their scores, and the tile's EPSG code."""

    def __init__(self, blobs: bytes, offsets: np.ndarray, scores: np.ndarray, epsg):
        self.blobs, self.offsets, self.scores, self.epsg = blobs, offsets, scores, epsg

    def __len__(self) -> int:
        return int(self.scores.shape[0])

    def rings(self) -> List[np.ndarray]:
        from .gpkg import parse_polygon_blob
        return [parse_polygon_blob(self.blobs[self.offsets[i]:self.offsets[i + 1]])[1] for i in range(len(self))]

    def envelopes(self) -> np.ndarray:
        """[n, 4] (minx, maxx, miny, maxy) straight from the blob headers."""
        if not len(self):
            return np.zeros((0, 4))
        raw = np.frombuffer(self.blobs, dtype=np.uint8)
        idx = (self.offsets[:-1, None] + 8 + np.arange(32)[None, :]).reshape(-1)
        return raw[idx].view("<f8").reshape(-1, 4)


def _epsg_code(crs) -> int:
    return int(str(crs).upper().replace("EPSG:", ""))


def process_prediction_file_sync(file, tiles_path, tif_lookup, shift, simplify_tolerance, logger=None, metadata=None):
    """One tile file → :class:`TileFeatures` of the crowns that survive the edge filter, or None on any error (logged,
    like the reference's try/except around the whole file). ``metadata``: the image's tile JSON if already parsed.
    Parsing, simplification, the *within* test and the geometry encoding run in td_stitch_tile_json (GIL released)."""
    try:
        file = Path(file)
        tifpath = tif_lookup.get(file.stem.replace("Prediction_", ""))
        folder = file.parent.name
        if not tifpath:
            raise FileNotFoundError(f"No matching TIFF file for {file}")
        if metadata is None:
            metadata_path = Path(tiles_path) / tifpath.with_name(f"{folder}.json")
            if not os.path.exists(metadata_path):
                raise FileNotFoundError(f"No matching metadata for {tifpath} (metadata_path: {metadata_path})")
            with open(metadata_path) as f:
                metadata = json.load(f)
        if str(tifpath) not in metadata:
            raise FileNotFoundError(f"No matching metadata for {tifpath}")
        epsg = metadata[str(tifpath)]["crs"]
        with open(file, "rb") as f:
            text = f.read()
        box = (C.c_double * 4)(*[float(v) for v in box_filter(str(tifpath), shift)])
        lib = _lib.load()
        need_b, need_f = C.c_int64(0), C.c_int(0)
        cap_b, cap_f = max(len(text), 1024), max(len(text) // 64, 16)      # a blob is smaller than its JSON text
        while True:
            blobs = np.empty(cap_b, dtype=np.uint8)
            offsets = np.empty(cap_f + 1, dtype=np.int64)
            scores = np.empty(cap_f, dtype=np.float64)
            n = lib.td_stitch_tile_json(text, len(text), box, float(simplify_tolerance), _epsg_code(epsg), blobs.ctypes.data,
                                        cap_b, offsets.ctypes.data, scores.ctypes.data, cap_f, C.byref(need_b), C.byref(need_f))
            if n == _lib.ERR_CAPACITY:
                cap_b, cap_f = max(cap_b, int(need_b.value)), max(cap_f, int(need_f.value))
                continue
            _lib.check(n, "td_stitch_tile_json")
            return TileFeatures(blobs[: int(need_b.value)].tobytes(), offsets[: n + 1].copy(), scores[:n].copy(), epsg)
    except Exception as e:
        if logger:
            logger.warning(f"Error processing file {file}: {e}")
        return None


def stitch_tile_files(files, tif_lookup, metadata, shift, simplify_tolerance, threads=1, logger=None):
    """All tile files of one image through ONE library call (td_stitch_tile_files: read, parse, simplify, edge filter and
    encode on ``threads`` host threads, no Python per file) → :class:`TileFeatures` of the whole image in file order, or None
    when nothing survives. A file that cannot be matched to a tile, read or parsed is left out with a warning — what the
    reference's try / except around each file does (helpers.py:419-476)."""
    paths, boxes, srs, epsg_first = [], [], [], None
    for file in files:
        try:
            tifpath = tif_lookup.get(Path(file).stem.replace("Prediction_", ""))
            if not tifpath:
                raise FileNotFoundError(f"No matching TIFF file for {file}")
            if str(tifpath) not in metadata:
                raise FileNotFoundError(f"No matching metadata for {tifpath}")
            epsg = metadata[str(tifpath)]["crs"]
            box = box_filter(str(tifpath), shift)
            code = _epsg_code(epsg)
        except Exception as e:
            if logger:
                logger.warning(f"Error processing file {file}: {e}")
            continue
        paths.append(os.fsencode(str(file)))
        boxes.append(box)
        srs.append(code)
        epsg_first = epsg if epsg_first is None else epsg_first
    if not paths:
        return None
    n = len(paths)
    offs = np.zeros(n + 1, np.int64)
    np.cumsum([len(p) + 1 for p in paths], out=offs[1:])
    blob_paths = b"\0".join(paths) + b"\0"
    boxes_a = np.ascontiguousarray(boxes, dtype=np.float64)
    srs_a = np.ascontiguousarray(srs, dtype=np.int32)
    status = np.zeros(n, np.int32)
    total = 0
    for p in paths:
        try:
            total += os.path.getsize(p)
        except OSError:         # reported per file by the library call (status < 0)
            pass
    cap_b, cap_f = max(total, 1024), max(total // 64, 16)          # a blob is smaller than its JSON text
    lib = _lib.load()
    need_b, need_f = C.c_int64(0), C.c_int(0)
    while True:
        blobs = np.empty(cap_b, dtype=np.uint8)
        offsets = np.empty(cap_f + 1, dtype=np.int64)
        scores = np.empty(cap_f, dtype=np.float64)
        m = lib.td_stitch_tile_files(blob_paths, offs.ctypes.data, n, boxes_a.ctypes.data, float(simplify_tolerance), srs_a.ctypes.data,
                                     int(max(1, threads)), blobs.ctypes.data, cap_b, offsets.ctypes.data, scores.ctypes.data, cap_f,
                                     status.ctypes.data, C.byref(need_b), C.byref(need_f))
        if m == _lib.ERR_CAPACITY:
            cap_b, cap_f = max(cap_b, int(need_b.value)), max(cap_f, int(need_f.value))
            continue
        _lib.check(m, "td_stitch_tile_files")
        break
    if logger:
        for i in np.nonzero(status < 0)[0]:
            logger.warning(f"Error processing file {os.fsdecode(paths[i])}: status {int(status[i])} (unreadable or malformed prediction file)")
    if m == 0:
        return None
    return TileFeatures(blobs[: int(need_b.value)].tobytes(), offsets[: m + 1].copy(), scores[:m].copy(), epsg_first)


def process_folder_sync(folder, tiles_path, pred_fold, output_path, shift, simplify_tolerance, logger=None, threads=1):
    """All tile files of one image → ``<output_path>/<image>.gpkg`` (empty layer, EPSG:4326, when nothing survives).
    ``threads``: host threads the tile files are spread over inside the library call; the layer keeps the sorted-file order
    either way, so the bytes written do not depend on it."""
    try:
        image_meta_path = os.path.join(tiles_path, f"{folder}")
        folder = folder.replace(".json", "")
        with open(image_meta_path) as f:
            metadata = json.load(f)
        tif_lookup = {Path(t).stem: Path(t) for t in metadata}
        pred_files = sorted(Path(os.path.join(pred_fold, folder)).rglob("*.json"))
        feats = stitch_tile_files(pred_files, tif_lookup, metadata, shift, simplify_tolerance, threads, logger)
        output_file = os.path.join(output_path, f"{folder}.gpkg")
        if feats is None:
            if logger:
                logger.debug(f"No valid results for folder {folder}. Creating empty output.")
            write_blobs(output_file, [], {}, None, None)
        else:
            env = feats.envelopes()
            extent = (float(env[:, 0].min()), float(env[:, 2].min()), float(env[:, 1].max()), float(env[:, 3].max()))
            view = memoryview(feats.blobs)
            o = feats.offsets.tolist()
            blobs = (view[o[i]:o[i + 1]] for i in range(len(feats)))
            scores = feats.scores.tolist()
            write_blobs(output_file, blobs, {"Confidence_score": scores, "filter_index_right": [0] * len(scores)},
                        _epsg_code(feats.epsg), extent)
        return output_file
    except Exception as e:
        if logger:
            logger.error(f"Error processing folder {folder}: {e}")
        return None


def validate_paths(tiles_path, pred_fold, output_path) -> None:
    if not os.path.exists(tiles_path):
        raise FileNotFoundError(f"Tiles path not found: {tiles_path}")
    if not os.path.exists(pred_fold):
        raise FileNotFoundError(f"Predictions path not found: {pred_fold}")
    os.makedirs(output_path, exist_ok=True)


def process_and_stitch_predictions(tiles_path, pred_fold, output_path, max_workers=50, shift=1, simplify_tolerance=0.2,
                                   logger=None):
    """Reference helpers.py:556-600: every image JSON under ``tiles_path`` that the resume file does not list yet →
    one GeoPackage; the resume file is rewritten with everything attempted. Returns ``output_path``."""
    validate_paths(tiles_path, pred_fold, output_path)
    completed = load_stitching_recovery(output_path, logger)
    folders = [f for f in sorted(os.listdir(tiles_path)) if f.endswith(".json") and os.path.isfile(os.path.join(tiles_path, f))]
    todo = [f for f in folders if os.path.splitext(f)[0] not in completed]
    if logger and len(folders) - len(todo) > 0:
        logger.info(f"Skipping stiching {len(folders) - len(todo)} of {len(folders)} folders that have already been processed.")
    results = []
    cores = len(os.sched_getaffinity(0))
    workers = max(1, min(int(max_workers or 1), len(todo) or 1, cores))
    per_folder = max(1, min(8, cores // workers))       # few folders: the tile files of each are spread over the idle cores
    with ThreadPoolExecutor(max_workers=workers) as ex:
        futures = {ex.submit(process_folder_sync, f, tiles_path, pred_fold, output_path, shift, simplify_tolerance, logger, per_folder): f
                   for f in todo}
        total = len(todo)
        for i, fut in enumerate(as_completed(futures)):
            results.append(futures[fut])
            cur, prev = int(100 * (i + 1) / total), int(100 * i / total)
            if logger and ((cur // 5) != (prev // 5) or i == 0 or cur == 100):
                logger.info(f"Stitching file {i + 1}/{total} ({cur}%)")
    save_stitching_recovery(output_path, list(completed) + results, logger)
    return output_path


class EagerStitcher:
    """Stitches images as their prediction files become complete, while the predictor goes on with the next image
    (detection.predict_on_model): ``submit("<image>.json")`` queues ``process_folder_sync`` for one image, ``close()``
    waits and returns the folders whose layer was written. The reference stitches after ALL images are predicted
    (detection.py:170-195, 235-243) and on one process; the layers are the same files — ``process_and_stitch_predictions``
    afterwards finds them in the resume file and only handles what is left (images whose prediction failed). Folders the
    resume file already lists are skipped here as the reference would skip them."""

    def __init__(self, tiles_path, pred_fold, output_path, shift=1, simplify_tolerance=0.2, logger=None, workers=2):
        os.makedirs(output_path, exist_ok=True)
        self.args = (tiles_path, pred_fold, output_path, shift, simplify_tolerance, logger)
        self.completed_before = set(load_stitching_recovery(output_path, None))
        # two images may be stitched side by side (one finishing its GeoPackage while the next one parses tile files);
        # the tile files of an image are spread over ``workers`` threads inside td_stitch_tile_files (no Python per file: the
        # epilogue workers of the running prediction hold the GIL most of the time)
        self._images = ThreadPoolExecutor(max_workers=2, thread_name_prefix="td-stitch")
        self._threads = max(1, int(workers))
        self._futures = {}
        self.seconds = 0.0
        self._seconds_lock = threading.Lock()

    def submit(self, folder_json: str) -> None:
        name = os.path.splitext(folder_json)[0]
        if name in self.completed_before or folder_json in self._futures:
            return
        if not os.path.isfile(os.path.join(self.args[0], folder_json)):
            return
        self._futures[folder_json] = self._images.submit(self._one, folder_json)

    def _one(self, folder_json):
        import time
        t0 = time.perf_counter()
        tiles_path, pred_fold, output_path, shift, tol, logger = self.args
        out = process_folder_sync(folder_json, tiles_path, pred_fold, output_path, shift, tol, logger, threads=self._threads)
        with self._seconds_lock:
            self.seconds += time.perf_counter() - t0
        return out

    def close(self) -> List[str]:
        """→ folder names ("<image>.json", the resume file's spelling) whose layer exists now."""
        done = []
        for folder_json, fut in self._futures.items():
            try:
                if fut.result() is not None:
                    done.append(folder_json)
            except Exception:
                pass
        self._images.shutdown(wait=True)
        return done
