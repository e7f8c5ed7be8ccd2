"""Public API of the reference, kept: ``process_files`` / ``preprocess_files`` / ``predict_tiles`` /
``postprocess_files`` / ``predict_on_model`` (TreeDetection/detection.py:342,256,134,23,62).

``predict_tiles`` / ``predict_on_model`` are the hot path this package accelerates (model forward on the MI355X);
``preprocess_files`` produces the tile metadata; stitching writes one GeoPackage per image into the reference's
``*_geojson`` / ``geojson_predictions`` folders (treedetection_amd/stitching.py);
``postprocess_files`` filters the crowns with nDSM / NDVI statistics (treedetection_amd/postprocessing.py,
``td_crown_stats``).
"""
from __future__ import annotations

import datetime
import os
import re
import shutil
import time
from pathlib import Path

from . import distributed as D
from .config import Config, get_config, setup_model_cfg  # noqa: F401  (re-exported like the reference)
from .prediction import Predictor
from .preprocessing import tile_data
from .recoveries import load_prediction_recovery_data, save_prediction_recovery_data
from .fusion import exclude_outlines, fuse_predictions  # noqa: F401
from .merging import merge_and_crop_images
from .postprocessing import process_files_in_directory
from .stitching import process_and_stitch_predictions


def image_weights(paths, tiles_path=None):
    """What an image costs its owner: its TILE COUNT — the number of entries of ``<tiles_path>/<image>.json``, the list the
    predictor walks (reference prediction.py:127-157) — when every image's tile file can be read; else the raster's file size
    for all of them (one unit for the whole list). File size alone misleads on DEFLATE / LZW rasters, whose bytes follow the
    content and not the geometry: equal images would be split unevenly and the fast ranks would sit in the closing gather."""
    import json
    counts = []
    if tiles_path is not None:
        for p in paths:
            try:
                with open(os.path.join(tiles_path, os.path.basename(p).replace(".tif", ".json"))) as f:
                    counts.append(len(json.load(f)))
            except (OSError, ValueError):
                counts = None
                break
        if counts is not None:
            return counts
    sizes = []
    for p in paths:
        try:
            sizes.append(os.path.getsize(p))
        except OSError:
            sizes.append(0)
    return sizes


def assign_images(paths, world: int, tiles_path=None):
    """Image-level sharding: → owner rank of every image. Longest-processing-time first on ``image_weights`` (tile counts; the
    seam strips under ``merged/`` are a fraction of a full image), ties and equal weights fall back to round-robin in list
    order — with equal images this IS ``i mod world``. Deterministic in (paths, weights, world)."""
    sizes = image_weights(paths, tiles_path)
    order = sorted(range(len(paths)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    owner = [0] * len(paths)
    for n, i in enumerate(order):
        r = min(range(world), key=lambda k: (load[k], (k - n) % world))
        owner[i] = r
        load[r] += max(sizes[i], 1)
    return owner


def resolve_shard_by(config, world: int, epilogue: str, n_images: int) -> str:
    """How a multi-rank ``predict_on_model`` cuts its work (config key ``shard_by``: "auto" | "image" | "tile").

    "image": rank r predicts WHOLE images (``assign_images``) through the chained single-process pipeline, writes their tile
    files and stitches them; no collective per image (one list broadcast before the walk, one manifest gather + one all-reduce
    after it). Needs the "local" epilogue (every rank writes into the shared output folder). "tile": every image is sharded
    tile by tile over all ranks (``Predictor.__call__``: BASELINE configs[3]'s single 10k-tile mosaic, or the "rank0"
    epilogue). "auto" = "image" when the epilogue is "local" and there are at least as many images as ranks."""
    mode = str(config.get("shard_by", "auto"))
    if mode not in ("auto", "image", "tile"):
        raise ValueError(f"shard_by must be 'auto', 'image' or 'tile', got {mode!r}")
    if world == 1:
        return "single"
    if epilogue != "local":
        return "tile"
    if mode == "auto":
        return "image" if n_images >= world else "tile"
    return mode


def walk_images(config, predictor, paths, tiles_path, output_path, chain=True, stitch_to=None):
    """The walk of ``predict_on_model`` over the images THIS process is responsible for (reference detection.py:112-129), with
    no collective of its own. ``chain``: the images are this process's alone (single process, image-level sharding) — image
    i+1 is started (read, launched) while image i's last forwards and tile files finish (``Predictor.submit``); the per-image
    outcome (files written, or the error logged and the walk continuing, reference detection.py:117-120) is the same as
    calling the predictor image by image. Otherwise every image is one collective structure all ranks enter together
    (``Predictor.__call__``). ``stitch_to``: every image whose tile files are complete is stitched at once on host threads
    (stitching.EagerStitcher) while the GPU goes on. → {"done", "failed", "stitched", "stitch_seconds"}."""
    logger = config.get("logger", None)
    stitcher = None
    if stitch_to is not None:
        from .prediction import host_core_share
        from .stitching import EagerStitcher
        stitcher = EagerStitcher(tiles_path, output_path, stitch_to, shift=1, simplify_tolerance=config["simplify_tolerance"],
                                 logger=logger, workers=max(1, min(4, host_core_share() // 4)))
    total = len(paths)
    pending = None
    done, failed = [], []

    def finish(item):
        path, handle = item
        try:
            if handle is not None:
                handle.result()
            done.append(path)
            if stitcher is not None:
                stitcher.submit(os.path.basename(path).replace(".tif", ".json"))
        except Exception as e:
            failed.append(path)
            logger.error(f"Error processing {path}: {e}")

    try:
        if chain and paths and hasattr(predictor, "prefetch"):
            predictor.prefetch(paths[0])
        for n, fp in enumerate(paths):
            cur, prev = int(100 * (n + 1) / total), int(100 * n / total)
            if logger and ((cur // 5) != (prev // 5) or cur == 100 or n == 0):
                logger.info(f"Predicting file {n + 1}/{total} ({cur}%)")
            tile_json = os.path.join(tiles_path, os.path.basename(fp).replace(".tif", ".json"))
            handle = None
            try:
                if chain:
                    if n + 1 < total and hasattr(predictor, "prefetch"):
                        predictor.prefetch(paths[n + 1])      # an LZW raster is read and decoded on the GPU while this image predicts
                    handle = predictor.submit(fp, tile_json, whole_image=True)
                else:
                    predictor(fp, tile_json)
                    finish((fp, None))
            except Exception as e:
                failed.append(fp)
                logger.error(f"Error processing {fp}: {e}")
            if pending is not None:
                finish(pending)
            pending = (fp, handle) if handle is not None else None
        if pending is not None:
            finish(pending)
    finally:
        stitched = stitcher.close() if stitcher is not None else []
    return {"done": done, "failed": failed, "stitched": stitched, "stitch_seconds": stitcher.seconds if stitcher is not None else 0.0}


FP16_MIN_BATCH = 0


def engine_batch_size(config, batch_size: int) -> int:
    """Tiles per forward. ``batch_size`` is the reference's key ("1 per GB on GPU", default 10: example/config.yml:29) and is
    what both engines run by default. ``fp16_min_batch`` (default 0 = off) raises it for ``precision: fp16``: at the ENGINE
    level BASELINE configs[4]'s batch 32 is 5 % faster per tile than batch 8 (2 381 vs 2 266 tiles/s, inputs resident in HBM:
    four times the rows per launch fill the last wave of the small-map layers) and gives the same files bit for bit
    (tests/test_fullsize_gpu.py) — but files to files it is SLOWER on a 16-core host (round 5, same box: e2e 1 737 vs 2 122
    tiles/s, chained 1 797 vs 2 210): a 400-tile image is 13 batches of 32 over three engines, the reader, the H2D copy and
    the epilogue workers hand over 128 MB / 32 files at a time and overlap worse than with 8-tile batches. So the default
    stays the configured batch size; the key is for hosts with more cores per GPU."""
    if str(config.get("precision", "fp32")) != "fp16":
        return batch_size
    return max(int(batch_size), int(config.get("fp16_min_batch", FP16_MIN_BATCH)))


def predict_on_model(config, model_path, tiles_path, output_path, batch_size=10, exclude_vars=None, stitch_to=None):
    """Reference detection.py:62-132: build the predictor once, walk the images (+ ``merged/``), swallow and log per-image
    errors, write the resume file. ``stitch_to`` (not in the reference's signature; ``predict_tiles`` passes it): the folder
    of the stitched layers — every image whose tile files are complete is stitched right away on host threads while the GPU
    predicts the next one (stitching.EagerStitcher), by the rank that completed it."""
    logger = config.get("logger", None)
    for path, name in [(model_path, "Model file"), (tiles_path, "Tiles directory")]:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{name} not found: {path}")
        if name == "Tiles directory" and not os.path.isdir(path):
            raise NotADirectoryError(f"{name} is not a directory: {path}")
    os.makedirs(output_path, exist_ok=True)
    D.bind_device(config["device"])      # before the first collective of this stage (nccl picks the current device)
    W, me = D.world(), D.rank()
    # the image list first: how a multi-rank run cuts its work depends on it
    images_directory = Path(config["image_directory"])
    images_paths = sorted(str(f) for f in images_directory.glob("*.tif"))
    merged_directory = Path(f"{images_directory}/{config['merged_path']}")
    images_paths.extend(sorted(str(f) for f in merged_directory.glob("*.tif")))
    found = len(images_paths)
    processed_files = set()
    if images_paths:
        file_list, processed_files = load_prediction_recovery_data(output_path, tiles_path, model_path, logger, exclude_vars)
        if not file_list:
            images_paths = [f for f in images_paths if f not in processed_files]
    owner = None
    if W > 1:
        # ONE list for everybody (rank 0's view of the folders, of the resume file and of the images' tile counts): the ranks' walks
        # must agree on the images and — image-level sharding — on who owns which. The only collective before the walk.
        images_paths, processed_files, found, owner = D.broadcast_object((images_paths, processed_files, found, assign_images(images_paths, W, tiles_path)))
    if not images_paths:
        if logger and not found:
            logger.warning("No TIF files found for prediction.")
        elif logger:
            logger.info("All files have already been predicted. Exiting Prediction.")
        D.barrier()
        return
    epilogue = config.get("sharded_epilogue", "auto")
    if W > 1 and epilogue == "auto":
        # "local" (every rank pastes, traces and writes its own tiles) needs an output folder rank 0 can read; with at least
        # as many images as ranks it is also what lets a rank own whole images — no data-path collective at all. Otherwise
        # the rule of prediction.resolve_sharded_epilogue: "rank0" below 4 ranks.
        from .prediction import resolve_sharded_epilogue
        shared = D.single_node() or D.output_is_shared(output_path)
        by_image = str(config.get("shard_by", "auto")) in ("auto", "image") and len(images_paths) >= W
        epilogue = "local" if shared and by_image else resolve_sharded_epilogue(W, config.get("precision", "fp32"), shared)
    cfg = setup_model_cfg(update_model=model_path, device=config["device"])
    # one process per GPU under torch.distributed: the shared config names one device, each rank takes its own
    # (LOCAL_RANK), see distributed.local_device
    device = config["device"] if W == 1 else str(D.local_device(config["device"]))
    batch_size = engine_batch_size(config, batch_size)
    predictor = Predictor(cfg, device_type=device, max_batch_size=batch_size, output_dir=output_path,
                          exclude_vars=exclude_vars, precision=config.get("precision", "fp32"),
                          return_predictions=False,       # the files are the product; the list is unused here
                          pipeline=config.get("pipeline", True), device_contours=config.get("device_contours", "auto"),
                          device_decode=config.get("device_decode", "auto"),
                          sharded_epilogue=epilogue)
    try:
        shard_by = resolve_shard_by(config, W, predictor.sharded_epilogue, len(images_paths))
        if shard_by != "image":
            owner = None
        mine = [i for i in range(len(images_paths)) if owner is None or owner[i] == me]
        if logger and W > 1:
            logger.info(f"rank {me}/{W}: sharding by {shard_by}; {len(mine)} of {len(images_paths)} images to walk")
        # who stitches an image as soon as its tile files are complete: the rank that owns it (image-level sharding, single
        # process), else rank 0 (tile-level sharding: every image's files are complete when its collective call returns)
        stitch_here = stitch_to if (config.get("eager_stitch", True) and (shard_by in ("single", "image") or me == 0)) else None
        my_paths = [images_paths[i] for i in mine]
        walk_error = None
        try:
            report = walk_images(config, predictor, my_paths, tiles_path, output_path,
                                 chain=shard_by in ("single", "image"), stitch_to=stitch_here)
        except Exception as e:
            # outside the per-image try of the walk (the eager stitcher's folder on this rank's mount, say): under image-level
            # sharding the peers are on their way into the manifest gather — this rank must enter the SAME collectives in the
            # same order and raise afterwards, or they wait for it until the watchdog
            if not (W > 1 and shard_by == "image"):
                raise
            walk_error = e
            if logger:
                logger.error(f"rank {me}: the image walk failed: {e}")
            report = {"done": [], "failed": my_paths, "stitched": [], "stitch_seconds": 0.0, "error": repr(e)}
        # the ONLY collectives of the walk under image-level sharding: one manifest gather, one all-reduce
        reports = D.gather_objects(report) if (W > 1 and shard_by == "image") else [report]
        ok = walk_error is None
        if me == 0:
            all_done = [p for r in reports for p in r["done"]]
            all_failed = [p for r in reports for p in r["failed"]]
            all_stitched = [p for r in reports for p in r["stitched"]]
            handled = images_paths
            if shard_by == "image":
                seen = sorted(all_done + all_failed)
                covered = seen == sorted(images_paths)
                errors = [r["error"] for r in reports if r.get("error")]
                if not covered:
                    logger.error(f"image-level sharding: {len(images_paths)} images assigned, {len(seen)} reported "
                                 f"({len(set(images_paths) - set(seen))} missing, {len(seen) - len(set(seen))} twice)")
                if not covered or errors:
                    # the resume file lists only what some rank really walked (predicted, or failed and logged like the reference's
                    # log-and-continue): an image nobody reported, or one of a rank whose whole walk broke, is predicted by the next run
                    ok = False
                    broken = {p for r in reports if r.get("error") for p in r["failed"]}
                    handled = [p for p in images_paths if p in set(seen) and p not in broken]
            logger.info(f"Completed prediction for {len(handled)} of {len(images_paths)} images.")
            save_prediction_recovery_data(output_path, tiles_path, model_path, processed_files, handled)
            if stitch_to is not None and all_stitched:
                from .recoveries import load_stitching_recovery, save_stitching_recovery
                save_stitching_recovery(stitch_to, sorted(load_stitching_recovery(stitch_to, None)) + all_stitched, logger)
        if W > 1 and shard_by == "image" and not D.all_ok(ok):
            if walk_error is not None:
                raise walk_error
            raise RuntimeError("image-level sharding: a rank's walk failed or the ranks' manifests do not cover the image list exactly once")
    finally:
        predictor.close()
        D.barrier()      # every Prediction_*.json, every eagerly stitched layer and the resume files are written before anyone moves on


def _stitch(config, pred_dir, out_dir):
    if D.rank() != 0:
        return
    process_and_stitch_predictions(tiles_path=config["tiles_path"], pred_fold=pred_dir, output_path=out_dir,
                                   max_workers=config["num_workers"], shift=1,
                                   simplify_tolerance=config["simplify_tolerance"], logger=config["logger"])


def predict_tiles(config):
    """Reference detection.py:134-253: two-model flow (urban + forest, exclude flags) when all three of urban_model,
    forrest_model, forrest_outline exist, else the combined model, else FileNotFoundError."""
    Config()._load_into_config(config)
    D.bind_device(config.get("device", "cpu"))
    logger = config["logger"]
    out = config["output_directory"]
    two = all(config.get(k) and os.path.exists(config[k]) for k in ("urban_model", "forrest_model", "forrest_outline"))
    if two:
        logger.info("Urban, forrest models and forrest outline are available. Starting prediction...")
        t0 = time.time()
        predict_on_model(config, config["urban_model"], config["tiles_path"], os.path.join(out, "urban_predictions"),
                         batch_size=config["batch_size"], exclude_vars=["only_forest"], stitch_to=os.path.join(out, "urban_geojson"))
        t1 = time.time()
        predict_on_model(config, config["forrest_model"], config["tiles_path"], os.path.join(out, "forrest_predictions"),
                         batch_size=config["batch_size"], exclude_vars=["only_urban"], stitch_to=os.path.join(out, "forrest_geojson"))
        t2 = time.time()
        _stitch(config, os.path.join(out, "urban_predictions"), os.path.join(out, "urban_geojson"))
        _stitch(config, os.path.join(out, "forrest_predictions"), os.path.join(out, "forrest_geojson"))
        logger.info("Predictions have been processed and stitched. Begin fusing the predictions.")
        t3 = time.time()
        if D.rank() == 0:
            fuse_predictions(os.path.join(out, "urban_geojson"), os.path.join(out, "forrest_geojson"), config["forrest_outline"],
                             os.path.join(out, "geojson_predictions"), logger=logger)
        logger.info("Fusion based on forest outline has been completed.")
        logger.debug(f"fuse prediction took {time.time() - t3} seconds")
        logger.debug(f"predict on model for urban took {t1 - t0} seconds")
        logger.debug(f"predict on model for forrest took {t2 - t1} seconds")
    elif config.get("combined_model") and os.path.exists(config["combined_model"]):
        logger.info("Only Combined Model is given. Starting prediction...")
        t0 = time.time()
        predict_on_model(config, config["combined_model"], config["tiles_path"], os.path.join(out, "predictions"),
                         batch_size=config["batch_size"], stitch_to=os.path.join(out, "geojson_predictions"))
        t1 = time.time()
        _stitch(config, os.path.join(out, "predictions"), os.path.join(out, "geojson_predictions"))
        logger.debug(f"Prediction took {t1 - t0} seconds")
    else:
        raise FileNotFoundError("No model available for prediction. Either urban model or forrest model + outline or "
                                "combined model must be available.")


def preprocess_files(config):
    """Reference detection.py:256-339: collect the rasters, seam strips between neighbours (``use_overlap``), tile
    metadata for every image."""
    Config()._load_into_config(config)
    D.bind_device(config.get("device", "cpu"))      # the broadcast below is this stage's first collective
    logger = config["logger"]
    for key, what in (("image_directory", "Image"), ("height_data_path", "Height")):
        p = config[key]
        if not os.path.exists(p):
            raise FileNotFoundError(f"{what} directory not found: {p}")
        if not os.path.isdir(p):
            raise NotADirectoryError(f"{what} directory is not a directory: {p}")
    images = sorted(os.path.join(config["image_directory"], f) for f in os.listdir(config["image_directory"]) if f.endswith(".tif"))
    heights = sorted(os.path.join(config["height_data_path"], f) for f in os.listdir(config["height_data_path"]) if f.endswith(".tif"))
    if os.path.exists(config["continue"]):
        with open(config["continue"]) as f:
            done = f.read().splitlines()
        images = [f for f in images if f not in done]
    irx = re.compile(config.get("image_regex", "(\\d+)\\.tif"))
    hrx = re.compile(config.get("height_data_regex", "(\\d+)\\.tif"))
    images = [f for f in images if irx.search(os.path.basename(f))]
    ids = {"".join(irx.search(os.path.basename(f)).groups()): f for f in images}
    hids = {"".join(hrx.search(os.path.basename(f)).groups()) for f in heights if hrx.search(os.path.basename(f))}
    # The seam strips and the tile metadata are files in folders every rank shares: rank 0 alone writes them, the
    # others wait at the barrier and then see complete files (the reference is single-process, detection.py:313-337).
    err = None
    if D.rank() == 0:
        try:
            if config["use_overlap"]:
                logger.info("Using overlapping tiles for processing, do merging right now ...")
                merge_and_crop_images(config, images, heights)
            for ident, path in ids.items():
                if ident not in hids:
                    logger.warning(f"No corresponding height data found for image file {path}")
            if not images:
                raise FileNotFoundError(f"No image TIF-files matching the pattern found in the directory: "
                                        f"{config['image_directory']} or all files have already been processed.")
            logger.info(f"Found {len(images)} images for processing. Starting tiling...")
            tile_data(images, config["tiles_path"], config["buffer"], config["tile_width"], config["tile_height"],
                      parallel=config["parallel"], max_workers=config["num_workers"], logger=logger,
                      forest_shapefile=config.get("forrest_outline", None))
        except Exception as e:      # every rank must leave this stage the same way
            err = e
    err, images = D.broadcast_object((err, images))
    if err is not None:
        raise err
    return images


def postprocess_files(config):
    """Reference detection.py:23-60: exclude-outline filter, crown post-processing of every stitched layer
    (``processed_*.gpkg`` next to it), then the processed layers copied into ``output_directory`` under their plain
    names (and into a time-stamped sub-folder with ``timestamped_output_directory``)."""
    Config()._load_into_config(config)
    logger = config["logger"]
    if D.rank() != 0:       # one writer; the others wait in process_files' barrier before anything is cleaned up
        return
    logger.info("Postprocessing the predictions.")
    pattern = (config.get("image_regex", "(\\d+)\\.tif"), config.get("height_data_regex", "(\\d+)\\.tif"))
    logger.info("Excluding Outlines.")
    exclude_outlines(config, logger)
    pred_dir = os.path.join(config["output_directory"], "geojson_predictions")
    if not os.path.isdir(pred_dir):
        logger.warning("No stitched predictions to post-process.")
        return
    process_files_in_directory(pred_dir, config["height_data_path"], config["image_directory"], parallel=config.get("parallel", True),
                               filename_pattern=pattern, config=config)
    stamp = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
    for name in sorted(os.listdir(pred_dir)):
        if not (name.endswith(".geojson") or name.endswith(".gpkg")) or not name.startswith("processed_"):
            continue
        plain = name.replace("processed_", "")
        if config.get("timestamped_output_directory"):
            os.makedirs(os.path.join(config["output_directory"], stamp), exist_ok=True)
            shutil.copy(os.path.join(pred_dir, name), os.path.join(config["output_directory"], stamp, plain))
        shutil.copy(os.path.join(pred_dir, name), os.path.join(config["output_directory"], plain))


def cleanup_files(config):
    """Reference detection.py:375-399: unless ``keep_intermediate``, remove the tile metadata, the seam-strip folders
    next to the rasters, stray ``__`` files, and every sub-folder of the output directory except ``logs``."""
    if config.get("keep_intermediate", False):
        return
    for path in (config["tiles_path"], os.path.join(config["image_directory"], config["merged_path"]),
                 os.path.join(config["height_data_path"], config["merged_path"])):
        shutil.rmtree(path, ignore_errors=True)
    for key in ("image_directory", "height_data_path"):
        for name in os.listdir(config[key]):
            if "__" in name:
                os.remove(os.path.join(config[key], name))
    for folder in os.listdir(config["output_directory"]):
        p = os.path.join(config["output_directory"], folder)
        if os.path.isdir(p) and folder != "logs":
            shutil.rmtree(p, ignore_errors=True)


def process_files(config):
    """Reference detection.py:342-373."""
    logger = config["logger"]
    Config()._load_into_config(config)
    # world > 1: every rank selects ITS GPU (LOCAL_RANK) before any barrier / broadcast — under nccl those collectives
    # run on torch.cuda.current_device(), which is cuda:0 on every rank until someone sets it
    D.bind_device(config.get("device", "cpu"))
    t0 = time.time()
    preprocess_files(config)
    t1 = time.time()
    predict_tiles(config)
    t2 = time.time()
    postprocess_files(config)
    t3 = time.time()
    # rank 0 stitched and post-processed out of the intermediate folders while the others idled: only once it is
    # done may they be removed, and only one rank removes them (reference detection.py:366-368 is single-process)
    D.barrier()
    if D.rank() == 0:
        cleanup_files(config)
    D.barrier()
    logger.debug(f"preprocess step took {t1 - t0} seconds. ")
    logger.debug(f"predict step took {t2 - t1} seconds. ")
    logger.debug(f"postprocess step took {t3 - t2} seconds. ")
