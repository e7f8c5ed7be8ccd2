"""Whole pipeline on a 2 x 2 mosaic of 1000 x 1000-px RGBI images (0.2 m) + nDSM rasters with the reference's default
layout (50 m tiles, 20 m buffer, 3-tile seam strips): preprocess → predict → stitch → post-process, stage times."""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, yaml
import treedetection_amd as T
from treedetection_amd.geotiff import write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict
from treedetection_amd import gpkg

def main(compress=False):
    root = tempfile.mkdtemp(prefix="e2e_pf_")
    os.makedirs(f"{root}/rgb"); os.makedirs(f"{root}/ndsm")
    np.savez(f"{root}/model.npz", **make_synthetic_state_dict(50, seed=0))
    kw = {"tile": (256, 256), "compression": "deflate", "predictor": 2} if compress else {}
    for iy in range(2):
        for ix in range(2):
            k = iy * 2 + ix
            rgb, nd = make_tile(10 + k, 1000)
            rgbi = np.ascontiguousarray(np.concatenate([rgb, 255 - rgb[..., :1] // 2], axis=2).transpose(2, 0, 1))
            x0, y0 = 412000.0 + 200 * ix, 5318400.0 - 200 * iy
            write_geotiff(f"{root}/rgb/{3240 + k}.tif", rgbi, (0.2, 0, x0, 0, -0.2, y0), 25832, **kw)
            write_geotiff(f"{root}/ndsm/{3240 + k}.tif", (nd[::5, ::5] + 4).copy(), (1.0, 0, x0, 0, -1.0, y0), 25832)
    cfg = {"image_directory": f"{root}/rgb", "height_data_path": f"{root}/ndsm", "combined_model": f"{root}/model.npz",
           "output_directory": f"{root}/output", "tiles_path": f"{root}/tiles", "batch_size": 16, "parallel": False,
           "num_workers": 8, "keep_intermediate": True, "device": "0", "ndvi_mean_threshold": 0.0, "ndvi_var_threshold": 1.0}
    open(f"{root}/config.yml", "w").write(yaml.safe_dump(cfg))
    config, _ = T.get_config(f"{root}/config.yml")
    t0 = time.time(); T.preprocess_files(config); t1 = time.time()
    T.predict_tiles(config); t2 = time.time()
    T.postprocess_files(config); t3 = time.time()
    ntiles = sum(len(json.load(open(f"{root}/tiles/{f}"))) for f in os.listdir(f"{root}/tiles") if f.endswith(".json"))
    strips = len(os.listdir(f"{root}/rgb/merged"))
    raw = sum(len(gpkg.read_polygons(f"{root}/output/geojson_predictions/{f}")[0]) for f in os.listdir(f"{root}/output/geojson_predictions")
              if f.endswith(".gpkg") and not f.startswith("processed_"))
    fin = {f: len(gpkg.read_polygons(f"{root}/output/{f}")[0]) for f in sorted(os.listdir(f"{root}/output")) if f.endswith(".gpkg")}
    print(f"{'deflate tiles' if compress else 'raw strips'}: 4 images + {strips} seam strips, {ntiles} tiles | preprocess {t1-t0:.2f}s | "
          f"predict+stitch {t2-t1:.2f}s ({ntiles/(t2-t1):.0f} tiles/s) | postprocess {t3-t2:.2f}s | crowns stitched {raw} -> final {fin}", flush=True)

if __name__ == "__main__":
    main(False)
    main(True)
