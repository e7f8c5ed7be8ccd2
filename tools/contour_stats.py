import sys; sys.path.insert(0,'.')
import numpy as np, torch
from treedetection_amd.engine import Engine, INPUT_U8_HWC
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict
from treedetection_amd.contours import find_contours
from treedetection_amd.engine import unpack_masks
eng = Engine(make_synthetic_state_dict(50, seed=0))
base,_ = make_tile(0, 1000)
tiles = [torch.from_numpy(np.ascontiguousarray(base[y:y+450, x:x+450])).cuda() for (y,x) in [(0,0),(100,200),(300,300),(500,100),(50,500),(400,0),(250,250),(520,520)]]
x, hv, ho = eng.preprocess_tiles_u8(tiles)
out = eng.alloc_outputs(8, 450, 450, paste=True)
eng.forward_raw(x, INPUT_U8_HWC, hv, ho, out)
cont = eng.alloc_contours(8)
eng.trace_contours(out, cont, 8)
torch.cuda.synchronize()
di = cont["det_info"].cpu().numpy(); cnt = out["count"].cpu().numpy()
st = np.concatenate([di[b,:cnt[b],0] for b in range(8)])
print("detections", len(st), "status histogram", {int(k): int((st==k).sum()) for k in np.unique(st)})
nc = np.concatenate([di[b,:cnt[b],1] for b in range(8)]); print("contours per traced detection: median", np.median(nc[st==0]) if (st==0).any() else None, "max", nc.max())
# how many contours do the fallback ones have on the host?
rg = out["mask_region"].cpu().numpy(); off = out["mask_offset"].cpu().numpy(); bits = out["mask_bits"].cpu().numpy()
b=0; m = unpack_masks(rg[b], off[b], bits[b], int(cnt[b]), 450, 450)
ks=[len(find_contours(m[d, rg[b,d,1]:rg[b,d,3], rg[b,d,0]:rg[b,d,2]].astype(np.uint8))) for d in range(int(cnt[b]))]
print("host contour counts image 0:", sorted(ks)[-10:], "median", np.median(ks), "regions", [(int(r[2]-r[0]), int(r[3]-r[1])) for r in rg[b,:5]])
a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): eng.trace_contours(out, cont, 8)
bb.record(); torch.cuda.synchronize()
print(f"td_trace_contours_dev: {a.elapsed_time(bb)/10*1e3:.0f} us per batch of 8 tiles ({len(st)} detections, {int(nc.sum())} contours, {int(cont['image_points'].sum())} points)")
import time
t0=time.time()
for b in range(8):
    m = unpack_masks(rg[b], off[b], bits[b], int(cnt[b]), 450, 450)
t1=time.time()
from treedetection_amd.contours import tile_polygons_json
t=(0.2,0,412000.0,0,-0.2,5319000.0)
sc=out["scores"].cpu().numpy(); cl=out["classes"].cpu().numpy()
t0=time.time()
for b in range(8): tile_polygons_json(rg[b], off[b], bits[b], sc[b][:cnt[b]], cl[b], t, "x.tif")
print(f"host epilogue (trace + JSON), one thread: {(time.time()-t0)/8*1e3:.2f} ms per tile")
