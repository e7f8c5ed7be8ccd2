"""End-to-end predict_on_model throughput (GeoTIFF read → device → JSON files) on one synthetic 1 km² image at
0.2 m GSD: 5000x5000 px, 20x20 tiles of 90 m (450 px, 350 at the raster edge) — the reference's default layout."""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, yaml
import treedetection_amd as T
from treedetection_amd.geotiff import write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

def main(size=5000, depth=50, batch=16):
    root = tempfile.mkdtemp(prefix="e2e_")
    os.makedirs(f"{root}/rgb"); os.makedirs(f"{root}/ndsm")
    np.savez(f"{root}/model.npz", **make_synthetic_state_dict(depth, seed=0))
    base, nd = make_tile(0, 1000)
    reps = size // 1000
    rgbi = np.concatenate([base, base[..., :1]], axis=2).transpose(2, 0, 1)
    img = np.tile(rgbi, (1, reps, reps))
    write_geotiff(f"{root}/rgb/324125317.tif", np.ascontiguousarray(img), (0.2, 0, 412000.0, 0, -0.2, 5318000.0 + size * 0.2), 25832)
    write_geotiff(f"{root}/ndsm/324125317.tif", nd, (1.0, 0, 412000.0, 0, -1.0, 5319000.0), 25832)
    cfg = {"image_directory": f"{root}/rgb", "height_data_path": f"{root}/ndsm", "combined_model": f"{root}/model.npz",
           "output_directory": f"{root}/output", "tiles_path": f"{root}/tiles", "use_overlap": False, "batch_size": batch,
           "parallel": False, "num_workers": 4, "keep_intermediate": True, "device": "0"}
    open(f"{root}/config.yml", "w").write(yaml.safe_dump(cfg))
    config, _ = T.get_config(f"{root}/config.yml")
    t0 = time.time(); T.preprocess_files(config); t1 = time.time()
    ntiles = len(json.load(open(f"{root}/tiles/324125317.json")))
    T.predict_tiles(config); t2 = time.time()
    nfiles = len(os.listdir(f"{root}/output/predictions/324125317"))
    print(f"tiles {ntiles} files {nfiles} | tiling {t1-t0:.2f}s | predict_tiles {t2-t1:.2f}s = {ntiles/(t2-t1):.1f} tiles/s end to end", flush=True)

if __name__ == "__main__":
    main()
