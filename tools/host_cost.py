"""What ONE GPU costs the host on the files-to-files path: CPU seconds per tile, split by stage.

    python tools/host_cost.py [fp16|fp32] [images=4] [side=20] [contours=host|dev|auto] [stitch=1|0] [batch=8] [raster=raw|lzw]
                              [device_raster=auto|all|false]

Fixture = bench.py's e2e raster (side x side tiles of 1000 x 1000 px, 4-band RGBI uint8 on tmpfs, 16 distinct generator tiles
cycled), compact-crown weights (weights.blob_mask_head: ~20 contours per tile). One warm-up image, then ``images`` images through
``detection.walk_images`` (chained ``Predictor.submit``, every finished image stitched while the next one predicts) exactly as
``predict_on_model`` runs them. Printed as one JSON object:
  * tiles/s of the walk, and of the same walk without stitching;
  * process CPU seconds per tile (getrusage user + sys over the timed walk; every thread of the process, C threads included);
  * per stage, wall and CPU milliseconds per tile: window reads (reader thread + window threads), launcher thread
    (H2D enqueue, resize, forward, result-copy enqueue), epilogue workers (wait for the batch, fetch mask rows or contour
    points, trace, format, write), stitching (thread seconds of EagerStitcher; its C threads are in the process total only);
  * cores_busy = process CPU seconds / wall seconds: the host cores one GPU keeps busy at this rate.
DESIGN.md §6's "host cores per GPU" table is made of these lines (profiles/r06_host_cost.txt)."""
from __future__ import annotations

import json
import logging
import os
import resource
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_seconds():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime, r.ru_stime


def main():
    os.environ["TD_HOST_STATS"] = "1"          # per-stage CPU accounting in the Predictor (off by default: it costs rate)
    import torch
    import treedetection_amd as T
    from treedetection_amd import detection as DT
    from treedetection_amd.geotiff import write_geotiff
    from treedetection_amd.preprocessing import tile_data
    from treedetection_amd.synth import make_tile
    from treedetection_amd.weights import blob_mask_head, make_synthetic_state_dict

    args = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    precision = next((a for a in sys.argv[1:] if a in ("fp16", "fp32")), "fp16")
    n_img, side, B = int(args.get("images", 4)), int(args.get("side", 20)), int(args.get("batch", 8))
    contours, stitch = args.get("contours", "host"), args.get("stitch", "1") != "0"
    raster, dd = args.get("raster", "raw"), {"auto": "auto", "all": "all", "false": False}[args.get("device_raster", "auto")]
    S = 1000
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix="td_hostcost_", dir=base)
    try:
        os.makedirs(f"{root}/rgb")
        tiles16 = [make_tile(i, S)[0] for i in range(16)]
        img = np.zeros((4, side * S, side * S), np.uint8)
        for r in range(side):
            for c in range(side):
                t = tiles16[(r * side + c) % 16]
                img[:3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t.transpose(2, 0, 1)
                img[3, r * S:(r + 1) * S, c * S:(c + 1) * S] = t[..., 1]
        tif0 = f"{root}/base.tif"
        kw = {"compression": "lzw", "tile": (256, 256), "predictor": 2} if raster == "lzw" else {}
        write_geotiff(tif0, img, (0.2, 0.0, 412000.0, 0.0, -0.2, 5318000.0 + side * S * 0.2), 25832, **kw)
        del img
        tile_data([tif0], f"{root}/tiles0", buffer=0, tile_width=200, tile_height=200)
        names = [str(324125400 + k) for k in range(n_img)] + ["warm"]
        os.makedirs(f"{root}/tiles")
        for nm in names:
            os.link(tif0, f"{root}/rgb/{nm}.tif")
            os.link(f"{root}/tiles0/base.json", f"{root}/tiles/{nm}.json")
        ntiles = len(json.load(open(f"{root}/tiles0/base.json")))
        sd = blob_mask_head(make_synthetic_state_dict(50, seed=0))
        cfg = T.setup_model_cfg(update_model="synthetic", device="0")
        dc = {"host": False, "dev": True, "auto": "auto"}[contours]
        pred = T.Predictor(cfg, device_type="0", max_batch_size=B, output_dir=f"{root}/pred", precision=precision, state_dict=sd,
                           return_predictions=False, device_contours=dc, device_decode=dd)
        logger = logging.getLogger("td-hostcost")
        logger.setLevel(logging.ERROR)
        config = {"logger": logger, "simplify_tolerance": 0.2}
        paths = [f"{root}/rgb/{nm}.tif" for nm in names[:-1]]
        pred.submit(f"{root}/rgb/warm.tif", f"{root}/tiles/warm.json", whole_image=True).result()
        out = {"precision": precision, "images": n_img, "tiles_per_image": ntiles, "batch": B, "contours": contours, "raster": raster,
               "device_raster": str(dd), "file_bytes": os.path.getsize(tif0),
               "host_cores": len(os.sched_getaffinity(0)), "epilogue_workers": pred._pool._max_workers}
        for label, st in (("walk", f"{root}/gpkg" if stitch else None), ("walk_again", f"{root}/gpkg2" if stitch else None)):
            pred.totals = dict.fromkeys(pred.stats, 0.0)
            pred._roll_stats()
            pred.totals = dict.fromkeys(pred.stats, 0.0)
            torch.cuda.synchronize()
            u0, s0 = cpu_seconds()
            t0 = time.perf_counter()
            rep = DT.walk_images(config, pred, paths, f"{root}/tiles", f"{root}/pred", chain=True, stitch_to=st)
            dt = time.perf_counter() - t0
            u1, s1 = cpu_seconds()
            assert len(rep["done"]) == n_img, rep
            tot = pred.host_totals()
            n = n_img * ntiles
            ms = lambda v: round(1e3 * v / n, 4)          # noqa: E731
            out[label] = {"tiles_per_s": round(n / dt, 1), "seconds": round(dt, 3),
                          "cpu_ms_per_tile": {"process_user": ms(u1 - u0), "process_sys": ms(s1 - s0), "process_total": ms(u1 - u0 + s1 - s0),
                                              "window_reads": ms(tot["read_cpu"]), "launcher": ms(tot["launch_cpu"]),
                                              "epilogue_workers": ms(tot["epilogue_cpu"])},
                          "wall_ms_per_tile": {"window_reads": ms(tot["read"]), "launcher": ms(tot["launch"]), "launcher_waiting_for_a_batch": ms(tot["launch_wait"]),
                                               "reader_waiting_for_a_slot": ms(tot["slot_wait"]), "epilogue_workers": ms(tot["epilogue"]),
                                               "epilogue_waiting_for_the_gpu": ms(tot["epilogue_wait"]), "stitch_threads": ms(rep["stitch_seconds"])},
                          "cores_busy": round((u1 - u0 + s1 - s0) / dt, 2),
                          "raster_upload": dict(pred.upload_stats), "raster_decode": dict(pred.decode_stats),
                          "device_contours_batches": getattr(pred, "device_contour_batches", None)}
        shutil.rmtree(f"{root}/pred", ignore_errors=True)
        pred.close()
        print(json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
