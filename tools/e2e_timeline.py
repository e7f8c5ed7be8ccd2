"""Where a warm Predictor.__call__ spends its wall time (bench.py's e2e region, instrumented): per precision and raster size,
the call's marks (TD_E2E_TRACE) → fixed part (call start → first launch; last launch → return), steady-state batch period,
how long the reader waited for a slot and the launcher for the reader.

    python tools/e2e_timeline.py [fp16|fp32] [side ...]      # side x side tiles of 1000 px; default 12
    python tools/e2e_timeline.py [fp16|fp32] chain=3 [side]  # the chained walk of detection.walk_images over 3 images of side x side
                                                             # tiles (predict + eager stitching): launches of ALL images on one time axis —
                                                             # the largest gap between consecutive launches against the steady batch period
"""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, ".")
os.environ["TD_E2E_TRACE"] = "1"
import numpy as np
import treedetection_amd as T
from treedetection_amd.geotiff import write_geotiff
from treedetection_amd.preprocessing import tile_data
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

S, B = 1000, int(os.environ.get("TD_TL_BATCH", "8"))      # TD_TL_BATCH: tiles per forward (8 = the bench's batch)


def raster(root, side, tiles):
    os.makedirs(f"{root}/rgb", exist_ok=True)
    img = np.zeros((4, side * S, side * S), np.uint8)
    for r in range(side):
        for c in range(side):
            t = tiles[(r * side + c) % len(tiles)]
            img[:3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t.transpose(2, 0, 1)
            img[3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t[..., 1]
    tif = f"{root}/rgb/324125317.tif"
    write_geotiff(tif, img, (0.2, 0.0, 412000.0, 0.0, -0.2, 5318000.0 + side * S * 0.2), 25832)
    tile_data([tif], f"{root}/tiles", buffer=0, tile_width=int(S * 0.2), tile_height=int(S * 0.2))
    return tif, f"{root}/tiles/324125317.json"


def summarize(trace, ntiles, dt):
    t0 = trace[0][2]
    by = {}
    for what, k, t in trace:
        by.setdefault(what, []).append((k, t - t0))
    launches = [t for _, t in by["launch"]]
    done = [t for _, t in by["launch_done"]]
    period = (launches[-1] - launches[2]) / max(1, len(launches) - 3) if len(launches) > 3 else float("nan")
    reads = dict(by["read"]); reads_done = dict(by["read_done"]); slot_wait = dict(by["slot_wait"])
    wait_slot = sum(reads[k] - slot_wait[k] for k in reads)
    read_busy = sum(reads_done[k] - reads[k] for k in reads)
    epi = sorted(t for _, t in by["epi"]); epi_done = sorted(t for _, t in by["epi_done"])
    out = {"tiles": ntiles, "call_ms": dt * 1e3, "tiles_per_s": ntiles / dt,
           "tiles_loaded_ms": by["tiles_loaded"][0][1] * 1e3, "raster_open_ms": by["raster_open"][0][1] * 1e3,
           "first_read_done_ms": reads_done[0] * 1e3, "first_launch_ms": launches[0] * 1e3, "first_launch_done_ms": done[0] * 1e3,
           "first_epilogue_ms": epi[0] * 1e3, "last_launch_done_ms": done[-1] * 1e3, "last_epilogue_start_ms": epi[-1] * 1e3,
           "last_epilogue_done_ms": epi_done[-1] * 1e3, "call_done_ms": by["call_done"][0][1] * 1e3,
           "steady_batch_period_ms": period * 1e3, "reader_waited_for_slot_ms": wait_slot * 1e3, "reader_busy_ms": read_busy * 1e3,
           "launch_busy_ms": sum(d - l for l, d in zip(launches, done)) * 1e3}
    return {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}


def chained(precision, n_images, side, sd):
    """detection.walk_images over n_images copies (hard links) of one raster: where the launcher's time axis has gaps."""
    import logging
    from treedetection_amd import detection as DT
    os.environ["TD_E2E_TRACE"] = "keep"
    tiles = [make_tile(i, S)[0] for i in range(16)]
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix="td_e2e_", dir=base)
    try:
        tif, tjson = raster(root, side, tiles)
        n = len(json.load(open(tjson)))
        names = [str(324125400 + k) for k in range(n_images)]
        os.makedirs(f"{root}/w/rgb"), os.makedirs(f"{root}/w/tiles")
        for nm in names:
            os.link(tif, f"{root}/w/rgb/{nm}.tif")
            os.link(tjson, f"{root}/w/tiles/{nm}.json")
        cfg = T.setup_model_cfg(update_model="synthetic", device="0")
        pred = T.Predictor(cfg, device_type="0", max_batch_size=B, output_dir=f"{root}/w/pred", precision=precision, state_dict=sd,
                           return_predictions=False)
        pred(tif, tjson)                                   # warm-up
        log = logging.getLogger("tl")
        log.setLevel(logging.ERROR)
        best, passes = None, []
        for rep in range(3):                               # first pass creates the files; the last one is the one reported
            del pred._trace[:]
            shutil.rmtree(f"{root}/w/gpkg", ignore_errors=True)
            t0 = time.perf_counter()
            rep_ = DT.walk_images({"logger": log, "simplify_tolerance": 0.2}, pred, [f"{root}/w/rgb/{nm}.tif" for nm in names],
                                  f"{root}/w/tiles", f"{root}/w/pred", chain=True, stitch_to=f"{root}/w/gpkg")
            dt = time.perf_counter() - t0
            passes.append(round(dt, 4))
            best = (dt, list(pred._trace), rep_)
        pred.close()
        dt, trace, rep_ = best
        t0 = trace[0][2]
        launches = [t - t0 for what, k, t in trace if what == "launch"]
        per_image = (n + B - 1) // B
        gaps = np.diff(launches)
        steady = float(np.median(gaps))
        # gaps that straddle an image boundary: launch index per_image * i - 1 → per_image * i
        bound = [float(gaps[per_image * i - 1]) for i in range(1, n_images) if per_image * i - 1 < len(gaps)]
        calls = [t - t0 for what, k, t in trace if what == "call"]
        # GPU side: when each batch's results were seen by the first epilogue worker (its event had fired), in completion order.
        # The launcher runs up to nine batches ahead of the GPU, so gaps between LAUNCHES are waits for a free slot; gaps between
        # COMPLETIONS are what the GPU did: across an image boundary they should look like any other batch period.
        done = sorted(t - t0 for what, k, t in trace if what == "batch_done")
        dgap = np.diff(done)
        dbound = [float(dgap[per_image * i - 1]) for i in range(1, n_images) if per_image * i - 1 < len(dgap)]
        print(json.dumps({"precision": precision, "fixture": os.environ.get("E2E_FIXTURE", "noise"), "images": n_images, "tiles_per_image": n,
                          "walk_s": round(dt, 4), "passes_s (first creates the files)": passes, "tiles_per_s": round(n_images * n / dt, 1), "launches": len(launches),
                          "steady_batch_period_ms": round(steady * 1e3, 2), "largest_gap_ms": round(float(gaps.max()) * 1e3, 2),
                          "launch_gaps_across_image_boundaries_ms": [round(g * 1e3, 2) for g in bound],
                          "batch_completion_period_ms": {"median": round(float(np.median(dgap)) * 1e3, 2), "p90": round(float(np.quantile(dgap, 0.9)) * 1e3, 2),
                                                         "max": round(float(dgap.max()) * 1e3, 2)},
                          "completion_gaps_across_image_boundaries_ms": [round(g * 1e3, 2) for g in dbound],
                          "submit_calls_at_ms": [round(c * 1e3, 1) for c in calls], "last_launch_ms": round(launches[-1] * 1e3, 1),
                          "stitched": len(rep_["stitched"]), "stitch_thread_s": round(rep_["stitch_seconds"], 3)}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    chain = [a for a in sys.argv[2:] if a.startswith("chain=")]
    if chain:
        sd = make_synthetic_state_dict(50, seed=0)
        if os.environ.get("E2E_FIXTURE") == "crowns":
            from treedetection_amd.weights import blob_mask_head
            sd = blob_mask_head(sd, seed=0)
        rest = [int(a) for a in sys.argv[2:] if not a.startswith("chain=")]
        return chained(precision, int(chain[0].split("=")[1]), rest[0] if rest else 20, sd)
    sides = [int(a) for a in sys.argv[2:]] or [12]
    sd = make_synthetic_state_dict(50, seed=0)
    if os.environ.get("E2E_FIXTURE") == "crowns":        # the compact-crown mask head of bench.py's e2e_crowns region
        from treedetection_amd.weights import blob_mask_head
        sd = blob_mask_head(sd, seed=0)
    tiles = [make_tile(i, S)[0] for i in range(16)]
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    cfg = T.setup_model_cfg(update_model="synthetic", device="0")
    kw = {}
    if os.environ.get("TD_DEVICE_CONTOURS"):
        kw["device_contours"] = True
    for side in sides:
        root = tempfile.mkdtemp(prefix="td_e2e_", dir=base)
        try:
            tif, tjson = raster(root, side, tiles)
            n = len(json.load(open(tjson)))
            pred = T.Predictor(cfg, device_type="0", max_batch_size=B, output_dir=f"{root}/out", precision=precision, state_dict=sd,
                               return_predictions=False, **kw)
            pred(tif, tjson)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                pred(tif, tjson)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, list(pred._trace), dict(pred.stats))
            pred.close()
            print(json.dumps({"precision": precision, "fixture": os.environ.get("E2E_FIXTURE", "noise"), "side": side, **summarize(best[1], n, best[0]), "stats": {k: round(v, 4) for k, v in best[2].items()}}), flush=True)
        finally:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
