"""Stage times of Predictor.__call__ on the e2e_predict.py image (400 tiles of 450x450 px): second call on a warm
engine, so weight upload and per-layer tile tuning are out of the picture."""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np
import treedetection_amd as T
from treedetection_amd.geotiff import write_geotiff
from treedetection_amd.preprocessing import tile_data
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

def main(size=5000, depth=50, batch=16, precision="fp32", device_contours=False, schedule="streams"):
    root = tempfile.mkdtemp(prefix="e2e_")
    os.makedirs(f"{root}/rgb")
    base, _ = make_tile(0, 1000)
    rgbi = np.concatenate([base, base[..., :1]], axis=2).transpose(2, 0, 1)
    img = np.tile(rgbi, (1, size // 1000, size // 1000))
    tif = f"{root}/rgb/324125317.tif"
    write_geotiff(tif, np.ascontiguousarray(img), (0.2, 0, 412000.0, 0, -0.2, 5318000.0 + size * 0.2), 25832)
    tile_data([tif], f"{root}/tiles", buffer=20, tile_width=50, tile_height=50)
    ntiles = len(json.load(open(f"{root}/tiles/324125317.json")))
    cfg = T.setup_model_cfg(update_model="x", device="0")
    pred = T.Predictor(cfg, device_type="0", max_batch_size=batch, output_dir=f"{root}/out", precision=precision,
                       state_dict=make_synthetic_state_dict(depth, seed=0), return_predictions=False,
                       device_contours=device_contours, schedule=schedule)
    for rep in range(3):
        t0 = time.time()
        pred(tif, f"{root}/tiles/324125317.json")
        dt = time.time() - t0
        print(f"call {rep}: {ntiles} tiles in {dt:.2f}s = {ntiles/dt:.1f} tiles/s | " +
              " ".join(f"{k}={v:.2f}" for k, v in pred.stats.items()), flush=True)
    nb = sum(os.path.getsize(os.path.join(f"{root}/out/324125317", f)) for f in os.listdir(f"{root}/out/324125317"))
    print(f"prediction files: {nb/1e6:.1f} MB")
    pred.close()

if __name__ == "__main__":
    main(precision=sys.argv[1] if len(sys.argv) > 1 else "fp32", device_contours="gpu_contours" in sys.argv[2:],
         schedule="phases" if "phases" in sys.argv[2:] else "streams")
