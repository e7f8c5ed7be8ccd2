"""``Predictor`` — drop-in for the reference's tiled predictor (TreeDetection/prediction.py:18-269).

Same constructor and call signature, same tile order (JSON key order, 131), exclusion semantics (79-93), BGR band
pick and 16-bit rule (166-167), batch flush rule (69-75), output path / file naming (55, 201) and JSON schema
(254-261) — but the model forward runs in libtreedet_hip.so on an MI355X (``Engine``), the per-tile host work is
reduced to what the reference's output needs (no GeoDataFrame per tile, the metadata JSON is parsed once, no cupy
round trip per contour), and with ``torch.distributed`` initialised the tiles shard across ranks and the detections
gather to rank 0 (treedetection_amd/distributed.py).
"""
from __future__ import annotations

import json
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from . import distributed as D
from .contours import batch_prediction_files, find_contours, tile_polygons_json, tile_polygons_json_dev, tile_prediction_file, xy
from .engine import Engine, INPUT_F32_CHW, INPUT_U8_HWC
from .geotiff import GeoTiff
from .weights import load_checkpoint


TILE_TABLE_VERSION = 12      # bump when the tile ids of csrc/conv_igemm.hip:dispatch() change meaning, the candidate set grows or the
                             # tuner's timing method changes (12: ids 29-33, id 28 retired, cold-L2 timing + hysteresis)


_FAULT_TILE = os.environ.get("TD_FAULT_TILE", "")      # tests only: the tile id whose crop raises (reference prediction.py:174-176 drops such a tile)


def _tune_cache_path(device_index: int) -> str:
    try:
        name = torch.cuda.get_device_name(device_index).replace(" ", "_").replace("/", "_")
    except Exception:
        name = "gpu"
    base = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "treedetection_amd")
    try:
        os.makedirs(base, exist_ok=True)
        if not os.access(base, os.W_OK):
            raise OSError(base)
    except OSError:
        import tempfile
        base = tempfile.mkdtemp(prefix="td_tune_")
    return os.path.join(base, f"tile_choices_v{TILE_TABLE_VERSION}_{name}.txt")


class _Slot:
    """Buffers of one in-flight batch: pinned tile staging + its device copy, the engine's device outputs and their
    pinned host copies, and the event that marks the copies complete. A slot goes reader → launcher → epilogue
    workers → back to the free list."""

    def __init__(self):
        self.pin_in = self.dev_in = None
        self.dev_out = self.pin_out = self.host = None
        self.dev_cont = self.pin_cont = self.host_cont = None
        self.out_key = None
        self.event = torch.cuda.Event(blocking=True)
        self.pending = 0
        self.traced = False         # this batch's borders were followed on the device (Predictor._contour_policy)
        self.submitted = 0.0        # when the batch's epilogue tasks were queued
        self.lock = threading.Lock()
        self.transient = False      # stand-in for a round that has no real slot (reader failure, _drain): never enters a free list

    def staging(self, nbytes: int, device) -> np.ndarray:
        if self.pin_in is None or self.pin_in.numel() < nbytes:
            self.pin_in = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=True)
            self.dev_in = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return self.pin_in.numpy()

    def outputs(self, engine: Engine, B: int, h: int, w: int, device_contours: bool = False) -> Dict[str, torch.Tensor]:
        key = self.out_key
        if key is None or B > key[0] or h > key[1] or w > key[2]:
            key = (max(B, key[0]) if key else B, max(h, key[1]) if key else h, max(w, key[2]) if key else w)
            self.dev_out = engine.alloc_outputs(*key, paste=True)
            self.pin_out = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in self.dev_out.items()}
            self.host = {k: v.numpy() for k, v in self.pin_out.items()}
            self.out_key = key
            self.dev_cont = self.pin_cont = self.host_cont = None
        if device_contours and self.dev_cont is None:
            self.dev_cont = engine.alloc_contours(self.out_key[0])
            self.pin_cont = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in self.dev_cont.items()}
            self.host_cont = {k: v.numpy() for k, v in self.pin_cont.items()}
        return self.dev_out


    # -- sharded runs: send / receive / paste buffers ------------------------------------------------------------
    def gather_buffers(self, B: int, Dn: int, dev) -> Dict[str, torch.Tensor]:
        if getattr(self, "_send", None) is None or self._send["count"].shape[0] != B:
            self._send = {"count": torch.zeros((B,), dtype=torch.int32, device=dev),
                          "boxes": torch.zeros((B, Dn, 4), dtype=torch.float32, device=dev),
                          "scores": torch.zeros((B, Dn), dtype=torch.float32, device=dev),
                          "mask_probs": torch.zeros((B, Dn, 28, 28), dtype=torch.float32, device=dev)}
            self._classes = torch.zeros((B, Dn), dtype=torch.int32, device=dev)
        return self._send

    def classes_buffer(self, B: int, Dn: int, dev) -> torch.Tensor:
        self.gather_buffers(B, Dn, dev)
        return self._classes

    def recv_buffers(self, W: int, B: int, Dn: int, dev, host: bool):
        if getattr(self, "_recv", None) is None or len(self._recv) != W or self._recv[0]["count"].shape[0] != B:
            where = torch.device("cpu") if host else dev
            self._recv = [{"count": torch.zeros((B,), dtype=torch.int32, device=where),
                           "boxes": torch.zeros((B, Dn, 4), dtype=torch.float32, device=where),
                           "scores": torch.zeros((B, Dn), dtype=torch.float32, device=where),
                           "mask_probs": torch.zeros((B, Dn, 28, 28), dtype=torch.float32, device=where)} for _ in range(W)]
        return self._recv

    def paste(self, engine: Engine, src: int, g: Dict[str, torch.Tensor], hw, B: int, Dn: int, dev):
        """Pastes one rank's gathered batch (first len(hw) rows) on this device and queues the copies of what the
        host epilogue reads — packed masks, counts, scores — into pinned memory. → the pinned numpy views."""
        sets = self.__dict__.setdefault("_paste_sets", {})
        mh, mw = max(h for h, _ in hw), max(w for _, w in hw)
        cur = sets.get(src)
        if cur is None or cur["key"][0] < B or cur["key"][1] < mh or cur["key"][2] < mw:
            key = (B, max(mh, cur["key"][1]) if cur else mh, max(mw, cur["key"][2]) if cur else mw)
            words = Dn * ((key[2] + 2 + 31) // 32) * key[1]
            d = {"mask_region": torch.empty((B, Dn, 4), dtype=torch.int32, device=dev),
                 "mask_offset": torch.empty((B, Dn), dtype=torch.int64, device=dev),
                 "mask_bits": torch.empty((B, words), dtype=torch.int32, device=dev)}
            p = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in d.items()}
            p["count"] = torch.empty((B,), dtype=torch.int32, pin_memory=True)
            p["scores"] = torch.empty((B, Dn), dtype=torch.float32, pin_memory=True)
            cur = sets[src] = {"key": key, "dev": d, "pin": p, "np": {k: v.numpy() for k, v in p.items()}}
        n = len(hw)
        engine.paste_masks_batch(g["mask_probs"], g["boxes"], g["count"].clamp(min=0), hw, cur["dev"])     # (-1 = dropped tile)
        for k in ("mask_region", "mask_offset"):       # the bit rows: fetched per tile by the epilogue worker (used words only)
            cur["pin"][k][:n].copy_(cur["dev"][k][:n], non_blocking=True)
        cur["pin"]["count"][:n].copy_(g["count"][:n], non_blocking=True)
        cur["pin"]["scores"][:n].copy_(g["scores"][:n], non_blocking=True)
        bits = cur["dev"]["mask_bits"]
        return dict(cur["np"], bits_base=bits.data_ptr(), bits_stride=bits.shape[1] * bits.element_size())


    def bits_ptr(self, i: int) -> int:
        """Device address of image i's packed mask rows in this slot's outputs."""
        v = self.dev_out["mask_bits"]
        return v.data_ptr() + i * v.shape[1] * v.element_size()

    def copy_results(self, n: int, device_contours: bool) -> None:
        """Queues the D2H copies of the small per-detection records the host epilogue reads (current stream). The packed
        mask rows are NOT copied here: their buffer is sized for the worst case (12.8 MB per 1000 x 1000 tile, 102 MB per
        8-tile batch — what a Gen5 x16 link moves in the 4 ms a fp16 batch takes) and each epilogue worker fetches just the
        words its tile's records say the paste wrote (td_tile_prediction_file). With device contours the traced points and
        records travel instead (rows fetched only for detections the device tracer left to the host)."""
        skip = ("mask_bits", "mask_probs")
        for k, v in self.dev_out.items():
            if k not in skip:
                self.pin_out[k][:n].copy_(v[:n], non_blocking=True)
        if device_contours:
            for k, v in self.dev_cont.items():
                self.pin_cont[k][:n].copy_(v[:n], non_blocking=True)


# Who pastes / traces / writes the tile files of a sharded run, by arithmetic. "rank0": every rank's fixed-shape
# payload (2.5 MB per 8-tile batch) goes to rank 0, whose GPU pastes for everybody and whose PCIe link carries everybody's packed
# masks to its host (measured: ~0.83 MB of Prediction JSON and ~5 MB of packed mask rows per 1000x1000 tile on the synthetic
# stream). One Gen5 x16 link moves ~55 GB/s and one host's epilogue workers ~7-14 k tiles/s: 8 fp32 engines (8 x 630 tiles/s x
# 5 MB = 25 GB/s) still fit, 4 or more fp16 engines (4 x 2 000 x 5 MB = 40 GB/s plus rank 0's own paste launches for 8 000
# tiles/s) do not leave headroom — from 4 ranks on every rank finishes its own tiles ("local": same bytes in the files, tested).
SHARDED_LOCAL_FROM_WORLD = 4


def host_core_share() -> int:
    """Host cores this rank may plan with: the process's affinity mask divided by the ranks that share the node
    (LOCAL_WORLD_SIZE; distributed.local_world). Single process: all of them."""
    return max(1, len(os.sched_getaffinity(0)) // D.local_world())


def resolve_sharded_epilogue(world: int, precision: str = "fp32", shared_output: bool = True) -> str:
    """"auto" → who pastes / traces / writes the tile files. "local" (every rank writes the files of its own tiles) needs an
    output folder that rank 0 — which alone stitches afterwards — can read: ``shared_output`` = all ranks on one node
    (LOCAL_WORLD_SIZE == WORLD_SIZE) or the folder proven shared (distributed.output_is_shared); otherwise "rank0", whatever
    the world size — on several nodes without a shared folder "local" would silently drop every other node's tiles."""
    return "local" if world >= SHARDED_LOCAL_FROM_WORLD and shared_output else "rank0"


class Predictor:
    def __init__(self, cfg, device_type="cpu", max_batch_size=5, output_dir="./output", exclude_vars=None,
                 precision: str = "fp32", state_dict: Optional[Dict[str, np.ndarray]] = None,
                 return_predictions: bool = True, host_workers: Optional[int] = None, pipeline: bool = True,
                 device_contours: bool = False, sharded_epilogue: str = "rank0", schedule: str = "streams",
                 device_decode="auto"):
        """cfg from ``setup_model_cfg``; ``device_type`` = GPU index ("0", 0) as config["device"] carries it.
        ``state_dict`` lets tests and the bench inject weights instead of reading cfg.MODEL.WEIGHTS.
        ``return_predictions=False`` skips rebuilding the Python list ``__call__`` returns (the reference's own caller
        ignores it, detection.py:118); the per-tile files are written either way. ``pipeline`` (single-process runs):
        three engines keep three batches in flight — contraction phases back to back on a main stream, each batch's
        selection phases on its own side stream (td_engine_forward_phase; same results bit for bit as one plain
        forward per batch, tests/test_fullsize_gpu.py) — at the price of three weight / workspace replicas.
        ``schedule`` picks how the three engines overlap: "streams" (default) = every engine runs whole forwards on its
        own HIP stream, batches round-robin — the HBM-bound kernels (Winograd transforms, thin 1x1 layers, RoIAlign) and
        the kernel tails of one forward run under the MFMA-bound contractions of the others (bench: 615 vs 554 tiles/s
        fp32, 1868 vs 1550 fp16); "phases" = the phase pipeline described above."""
        self.cfg = cfg
        # LZW rasters are decoded on the GPU, whole, and their tile windows are cut in HBM (GeoTiff.decode_to_device; images this
        # process predicts alone: submit). "auto" / True: every LZW raster that qualifies; False: the host reader for everything
        if device_decode == "auto" and os.environ.get("TD_DEVICE_DECODE"):       # diagnostics: override the default
            device_decode = {"0": False, "false": False, "all": "all"}.get(os.environ["TD_DEVICE_DECODE"], "auto")
        if device_decode not in (True, False, "auto", "true", "false", "all"):
            raise ValueError(f"device_decode must be true, false, 'auto' or 'all', got {device_decode!r}")
        self.device_decode = device_decode in (True, "auto", "true", "all")
        # "all": uncompressed uint8 rasters are kept whole in HBM too (uploaded in 4-MB pieces). Measured both ways (round 6,
        # tools/host_cost.py, profiles/r06_host_cost.txt): on a 256-thread host with slow page-cache reads +10 % tiles/s and -30 % host CPU
        # per tile; on a 16-core host -20 % (the upload's reader threads compete with the epilogue workers, and an image that was not
        # prefetched waits for its whole raster before its first batch) — so the default keeps the per-window host reader for them
        self.device_upload = device_decode == "all"
        self._rasters: Dict[str, "object"] = {}
        self._raster_pool = None
        self._decode_stream = None
        self.decode_stats = {"images": 0, "seconds": 0.0, "compressed_bytes": 0, "decoded_bytes": 0}
        self.upload_stats = {"images": 0, "seconds": 0.0, "compressed_bytes": 0, "decoded_bytes": 0}
        self._upload_staging = self._upload_pool = None
        self._decode_pinned = [None]
        # uncompressed rasters up to this size are kept whole in HBM too (TD_DEVICE_RASTER_MAX_GB, default 24): beyond it the host
        # window reader serves them
        self.device_raster_max_bytes = int(float(os.environ.get("TD_DEVICE_RASTER_MAX_GB", "24")) * (1 << 30))
        if device_type == "cpu" or not torch.cuda.is_available():
            raise RuntimeError("treedetection_amd.Predictor runs on an MI355X only: the HIP path has no CPU fallback "
                               "(config['device'] resolved to 'cpu').")
        self.device_index = int(device_type)
        self.device = f"cuda:{self.device_index}"
        self.max_batch_size = max_batch_size
        self.output_dir = output_dir
        self.exclude_vars = exclude_vars or []
        self.return_predictions = return_predictions
        os.makedirs(self.output_dir, exist_ok=True)
        if sharded_epilogue not in ("rank0", "local", "auto"):
            raise ValueError(f"sharded_epilogue must be 'rank0', 'local' or 'auto', got {sharded_epilogue!r}")
        if D.world() > 1:
            # the collectives below (and every later one of this rank) run on torch.cuda.current_device() under nccl: a
            # Predictor built without a prior distributed.bind_device must not leave every rank on cuda:0
            torch.cuda.set_device(self.device_index)
        if sharded_epilogue == "auto":
            # (collective when world >= 4 and the launcher did not say that all ranks share a node: every rank builds its
            # Predictor with the same arguments, predict_on_model does)
            shared = D.world() < SHARDED_LOCAL_FROM_WORLD or D.single_node() or D.output_is_shared(self.output_dir)
            sharded_epilogue = resolve_sharded_epilogue(D.world(), precision, shared)
        # borders followed on the GPU (td_trace_contours_dev); rank 0's gathered-batch epilogue uses the host tracer
        # "auto" (round 6): the device tracer is switched on while the host epilogue — not the GPU — sets the batch period
        # (_contour_policy); the files are byte-identical either way, so the switch may happen between any two batches
        if device_contours not in (True, False, "auto", "true", "false"):
            raise ValueError(f"device_contours must be true, false or 'auto', got {device_contours!r}")
        can = D.world() == 1 or sharded_epilogue == "local"
        self._contours_auto = device_contours == "auto" and can
        self.device_contours = (device_contours in (True, "true") or self._contours_auto) and can     # buffers for the tracer exist
        self._contours_on = self.device_contours and not self._contours_auto                          # … and this batch uses it
        self._late = 0.0                      # running share of epilogue tasks that found their batch finished before they started
        self._late_n = 0
        self._on_since = 0
        self.device_contour_batches = [0, 0]  # batches traced on the host / on the device
        sd = state_dict if state_dict is not None else load_checkpoint(cfg.MODEL.WEIGHTS)
        rh = cfg.MODEL.ROI_HEADS
        eng_args = dict(device=self.device_index, precision=precision, score_thresh=rh.SCORE_THRESH_TEST,
                        nms_thresh=rh.NMS_THRESH_TEST, rpn_nms_thresh=cfg.MODEL.RPN.NMS_THRESH,
                        pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
                        detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE)
        self.pipeline = bool(pipeline)
        if schedule not in ("streams", "phases"):
            raise ValueError(f"schedule must be 'streams' or 'phases', got {schedule!r}")
        self.schedule = schedule
        self.sharded_epilogue = sharded_epilogue     # torch.distributed runs only: who pastes / traces / writes the tile files
        if "TD_TUNE_CACHE" not in os.environ:
            # Measured block-tile choices are shared between the engines of this process and kept for later runs (the
            # first forward of a fresh process otherwise spends ~1.5 s timing tile variants per engine). The file is
            # keyed by GPU name and the library's tile-table version; an unwritable cache directory just means no reuse.
            os.environ["TD_TUNE_CACHE"] = _tune_cache_path(self.device_index)
        self.engine = Engine(sd, **eng_args)
        self._engines = [self.engine] + ([Engine(sd, **eng_args) for _ in range(2)] if self.pipeline else [])
        # host threads are sized by THIS rank's share of the node: cores // ranks on the node (8 ranks on a 128-core host get 16
        # cores each, not 8 x 16 epilogue workers + 8 x 8 window readers on whatever the node has)
        workers = host_workers or int(os.environ.get("TD_HOST_WORKERS", "0")) or max(2, min(16, host_core_share() - 2))
        self._pool = ThreadPoolExecutor(max_workers=workers)
        # buffer slots (pinned staging + outputs) in rotation: three per engine keep the fp16 engines fed while the host epilogue
        # of earlier batches still reads its slots (e2e fp16: 6 slots 1 410 tiles/s, 9 slots 1 531, 12 slots 1 502; fp32 unchanged)
        self._slots = [_Slot() for _ in range(int(os.environ.get("TD_SLOTS", "0")) or (9 if self.pipeline else 3))]
        # single-process runs: ONE free list for the predictor's lifetime — a slot returns to it whichever image it served, so
        # the next image can be started (submit) while the last batches of the previous one are still in flight
        self._free = queue.Queue()
        for s in self._slots:
            self._free.put(s)
        self._stats_lock = threading.Lock()
        # per-stage CPU accounting (time.thread_time per tile and stage) is for tools/host_cost.py only: the files-to-files path is
        # bound by the Python work per tile at the fp16 rate (every thread's Python runs under one interpreter lock), and a few
        # extra calls per tile cost 10 - 30 % of the rate on the same box (round 6, gpurun_out/r6_j … r6_l)
        self._cpu_stats = bool(os.environ.get("TD_HOST_STATS"))
        # one epilogue task per BATCH where only the files are wanted (_submit_epilogue); TD_BATCH_EPILOGUE=0 = a task per tile
        self._batch_epilogue = os.environ.get("TD_BATCH_EPILOGUE", "1") != "0"
        self._epilogue_threads = max(1, min(4, workers // 3))
        # seconds spent per stage of the last __call__ (reader thread, launcher thread, sum over epilogue workers)
        # (*_cpu: CPU seconds of the threads that did the stage — time.thread_time — for the host-cores-per-GPU table of DESIGN.md §6)
        self.stats = {"read": 0.0, "launch": 0.0, "launch_wait": 0.0, "epilogue": 0.0, "epilogue_wait": 0.0, "slot_wait": 0.0,
                      "read_cpu": 0.0, "launch_cpu": 0.0, "epilogue_cpu": 0.0, "tiles": 0}
        # TD_E2E_TRACE=1: (what, batch / tile index, perf_counter) marks of the last __call__ (tools/e2e_timeline.py)
        self._trace = [] if os.environ.get("TD_E2E_TRACE") else None

    def _roll_stats(self) -> None:
        """A new image starts: its stage seconds start at zero; what the previous images spent stays in ``totals`` (a chained walk
        overlaps images, so late epilogue tasks of the previous image land in the new image's counters: only the totals are exact)."""
        with self._stats_lock:
            tot = self.__dict__.setdefault("totals", dict.fromkeys(self.stats, 0.0))
            for k, v in self.stats.items():
                tot[k] = tot.get(k, 0.0) + v
            self.stats = dict.fromkeys(self.stats, 0.0)

    def host_totals(self) -> Dict[str, float]:
        """Stage seconds (wall and CPU) summed over every image since the predictor was built (or since the caller cleared
        ``totals``), the running image included."""
        with self._stats_lock:
            tot = dict(self.__dict__.get("totals") or dict.fromkeys(self.stats, 0.0))
            for k, v in self.stats.items():
                tot[k] = tot.get(k, 0.0) + v
        return tot

    # -- device_contours: "auto" -------------------------------------------------------------------------------------
    # Signal: an epilogue task is LATE when it sat in the worker pool's queue for more than a millisecond (every worker was busy)
    # AND its batch had left the GPU by the time it started — the workers, not the GPU and not the reader, set the pace. (A
    # finished event alone says nothing: when the READER is the slow stage the GPU idles, every event is complete at once and
    # the workers are idle too — measured on a box with slow page-cache reads, where that signal switched the tracer on and cost
    # 20 %.) The share of late tasks (exponential mean over ~64 tasks) switches the device tracer on above 0.5 — from then on the
    # workers only format and write what the GPU traced — and off again once it stayed below 0.05 for 2 000 tasks (a re-probe).
    # Measured on the compact-crown fixture (tools/host_cost.py, profiles/r06_host_cost.txt): the device tracer is SLOWER there
    # than 16 host workers (one wave follows a crown's border serially); it is for hosts with few cores per GPU.
    def _note_epilogue_start(self, late: bool) -> None:
        if not self._contours_auto:
            return
        with self._stats_lock:
            self._late += ((1.0 if late else 0.0) - self._late) / 64.0
            self._late_n += 1
            if not self._contours_on and self._late_n >= 32 and self._late > 0.5:
                self._contours_on, self._on_since, self._late_n = True, 0, 0
            elif self._contours_on:
                self._on_since = self._on_since + 1 if self._late < 0.05 else 0
                if self._on_since >= 2000:
                    self._contours_on, self._late_n, self._late = False, 0, 0.0

    def _contour_policy(self) -> bool:
        """Whether the batch being launched is traced on the device."""
        on = bool(self._contours_on)
        self.device_contour_batches[1 if on else 0] += 1
        return on

    def _mark(self, what, k=-1) -> None:
        if self._trace is not None:
            self._trace.append((what, k, time.perf_counter()))

    def close(self) -> None:
        """Stops the host worker threads and releases the engine's device memory."""
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if getattr(self, "_read_pool", None) is not None:
            self._read_pool.shutdown(wait=True)
            self._read_pool = None
        if getattr(self, "_raster_pool", None) is not None:
            self._raster_pool.shutdown(wait=True)
            self._raster_pool = None
            self._rasters.clear()
        if getattr(self, "_upload_pool", None) is not None:
            self._upload_pool.shutdown(wait=True)
            self._upload_pool = self._upload_staging = None
        for eng in getattr(self, "_engines", []) or []:
            eng.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- compressed rasters decoded on the GPU -------------------------------------------------------------------
    def prefetch(self, tifpath) -> None:
        """Starts, on a thread of its own, what the NEXT image needs before its first batch can be cut: an LZW raster's compressed
        blocks read into pinned memory, copied to the device and decoded there (GeoTiff.decode_to_device), or an uncompressed
        raster's bytes copied to the device in large sequential pieces (GeoTiff.upload_to_device) — while the current image
        predicts. ``detection.walk_images`` calls it with the path after the one it submits. No-op for rasters the host reader
        serves (DEFLATE, PackBits, planar, 16-bit) and when ``device_decode`` is off."""
        if not self.device_decode or tifpath in self._rasters:
            return
        if self._raster_pool is None:
            self._raster_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="td-raster")
        self._rasters[tifpath] = self._raster_pool.submit(self._decode_raster, tifpath)

    def _decode_raster(self, tifpath):
        """→ the raster as a device tensor [rows, cols, bands] uint8, or None (not decodable on the device, or a block failed:
        the host reader then serves the image and reports what is wrong with it, if anything)."""
        img = None
        try:
            img = GeoTiff(tifpath)
            decode = img.device_decodable()
            if not decode and not (self.device_upload and img.device_uploadable()
                                   and img.height * img.width * img.count <= self.device_raster_max_bytes):
                return None
            if self._decode_stream is None:
                # lowest priority: its own hardware queue, and the forwards' workgroups are served first (_lib.low_priority_stream)
                self._decode_stream = _lib.low_priority_stream(self.device_index)
            t0 = time.perf_counter()
            c0 = time.thread_time()
            if self._upload_pool is None:
                self._upload_pool = ThreadPoolExecutor(max_workers=max(1, min(8, host_core_share() // 4)), thread_name_prefix="td-upload")
            if decode:
                image, check = img.decode_to_device(self.device, self._decode_stream, self._decode_pinned, self._upload_pool)
            else:
                # an uncompressed raster: its bytes go to HBM in large sequential pieces (GeoTiff.upload_to_device), the windows
                # are cut there — one memcpy per byte out of the page cache instead of a pread per window row + an H2D per batch
                if self._upload_staging is None:
                    self._upload_staging = [torch.empty((64 << 20,), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
                image, check = img.upload_to_device(self.device, self._decode_stream, self._upload_staging, self._upload_pool)
            check()
            with self._stats_lock:
                st = self.decode_stats if decode else self.upload_stats
                st["images"] += 1
                st["seconds"] += time.perf_counter() - t0
                st["thread_cpu"] = st.get("thread_cpu", 0.0) + time.thread_time() - c0
                st["compressed_bytes"] += check.compressed_bytes
                st["decoded_bytes"] += image.numel()
                st["kernel_ms"] = st.get("kernel_ms", 0.0) + getattr(check, "kernel_ms", 0.0)
            return image
        except Exception as e:
            print(f"device decode of {tifpath} failed ({e}): using the host reader")
            return None
        finally:
            if img is not None:
                img.close()

    def _device_raster(self, tifpath):
        if not self.device_decode:
            return None
        self.prefetch(tifpath)
        fut = self._rasters.pop(tifpath, None)
        return fut.result() if fut is not None else None

    # -- tile metadata ---------------------------------------------------------------------------------------
    def _filter_excluded_vars(self, tiles):
        kept = []
        for tile in tiles:
            if any(tile[flag] for flag in self.exclude_vars):
                continue
            kept.append({k: v for k, v in tile.items() if k not in self.exclude_vars})
        return kept

    def _load_tiles(self, tilepath):
        with open(tilepath) as f:
            meta = json.load(f)
        tiles = []
        for tile_id, td in meta.items():
            entry = {"bounds": td["bounds"][:4], "tile_id": tile_id, "json_name": tilepath, "meta": td}
            for var in self.exclude_vars:
                entry[var] = td.get(var, False)
            tiles.append(entry)
        if self.exclude_vars:
            tiles = self._filter_excluded_vars(tiles)
        return tiles

    # -- one tile: crop → BGR → (16-bit rescale) → device -------------------------------------------------------
    def _process_tile(self, tile, img: GeoTiff, staging: Optional[np.ndarray] = None, staging_off: int = 0):
        """Reference prediction.py:159-176. uint8 rasters: the window is copied (into the pinned ``staging`` buffer at
        ``staging_off`` when given) as it lies in the file, the BGR pick and resize happen on the device."""
        try:
            if _FAULT_TILE and tile["tile_id"] == _FAULT_TILE:      # fault injection (tests): this tile's crop fails
                raise OSError(f"injected read failure for tile {tile['tile_id']} (TD_FAULT_TILE)")
            if getattr(img, "_dev", None) is not None:            # the raster lies decoded in HBM: the window is cut there
                c0, r0, w, h = img.window_of_bounds(tile["bounds"])
                if w <= 0 or h <= 0:
                    raise ValueError("Input shapes do not overlap raster.")
                if img.count < 3:
                    raise ValueError(f"tile has {img.count} bands, need >= 3")
                info = {"orig_height": h, "orig_width": w, "height": h, "width": w,
                        "json_name": tile["json_name"], "tile_id": tile["tile_id"], "meta": tile["meta"]}
                return {"devwin": (r0, c0, h, w, img.outside_mask(tile["bounds"], c0, r0, w, h))}, info
            if staging is not None and img.dtype == np.uint8:
                hwc = img.read_bounds_hwc(tile["bounds"], out=staging, out_off=staging_off)
            else:
                hwc = img.read_bounds_hwc(tile["bounds"])              # [h, w, bands], band order of the file
            if hwc.shape[2] < 3:
                raise ValueError(f"tile has {hwc.shape[2]} bands, need >= 3")
            orig_h, orig_w = hwc.shape[:2]
            info = {"orig_height": orig_h, "orig_width": orig_w, "height": orig_h, "width": orig_w,
                    "json_name": tile["json_name"], "tile_id": tile["tile_id"], "meta": tile["meta"]}
            if hwc.dtype == np.uint8:                                   # max(band 1) <= 255 by construction
                if staging is not None:
                    return {"staged": (staging_off, hwc.shape)}, info
                return {"u8": torch.from_numpy(hwc)}, info                 # BGR pick happens on the device
            out_img = hwc.transpose(2, 0, 1)
            # non-uint8 rasters take detectron2's float resize path (F.interpolate bilinear); 16-bit imagery
            # (max(band 1) > 255) is first rescaled 255 * x / 65535 — reference prediction.py:167
            bgr = np.stack((out_img[2], out_img[1], out_img[0])).astype(np.float64)
            if np.max(out_img[1]) > 255:
                bgr = 255.0 * bgr / 65535.0
            return {"f": torch.from_numpy(bgr)}, info
        except Exception as e:
            print(f"Error processing tile {tile['json_name']}: {e}")
            return None, None

    def _to_model_input(self, batch, slot: Optional[_Slot] = None, engine: Optional[Engine] = None):
        """→ (images tensor on device, format, hw_valid, hw_out)."""
        eng = engine or self.engine
        if batch and all(("staged" in b["data"] and slot is not None) or "u8" in b["data"] or "devwin" in b["data"] for b in batch):
            staged = [b["data"]["staged"] for b in batch if "staged" in b["data"]]
            if staged:
                used = max(o + int(np.prod(shp)) for o, shp in staged)
                slot.dev_in[:used].copy_(slot.pin_in[:used], non_blocking=True)       # one H2D per batch, from pinned memory
            tiles = []
            for b in batch:
                d = b["data"]
                if "staged" in d:
                    o, shp = d["staged"]
                    tiles.append(slot.dev_in[o:o + int(np.prod(shp))].view(*shp))
                elif "devwin" in d:                                 # a window of the raster decoded in HBM
                    r0, c0, h, w, mask = d["devwin"]
                    dev = b["raster"]
                    dev.record_stream(torch.cuda.current_stream())
                    t = dev[r0:r0 + h, c0:c0 + w, :].contiguous()
                    if mask is not None:                            # rasterio.mask: pixels whose centre lies outside the bbox are 0
                        okx, oky = mask
                        t[:, torch.from_numpy(~okx).to(t.device), :] = 0
                        t[torch.from_numpy(~oky).to(t.device), :, :] = 0
                    tiles.append(t)
                else:
                    tiles.append(d["u8"].to(self.device, non_blocking=True))
            images, hw_valid, hw_out = eng.preprocess_tiles_u8(tiles)
            return images, INPUT_U8_HWC, hw_valid, hw_out
        planes = []
        for b in batch:
            if "u8" in b["data"]:
                t = b["data"]["u8"].to(self.device).permute(2, 0, 1)[[2, 1, 0]].double()
            else:
                t = b["data"]["f"].to(self.device)
            planes.append(t.contiguous())
        # the reference's float resize branch (prediction.py:167-169) as a HIP kernel: td_resize_bilinear_f64
        x, shapes = eng.preprocess_tiles_f64(planes)
        return x, INPUT_F32_CHW, shapes, [(b["orig_height"], b["orig_width"]) for b in batch]

    # -- single process: reader thread → launcher (this thread) → epilogue workers --------------------------------
    def _read_batch(self, tiles, indices, img: GeoTiff, slot: _Slot, keep_failed: bool = False, dropped: Optional[list] = None):
        """Crops the tiles of one batch into the slot's pinned staging buffer (uint8 rasters) — reader thread. A tile whose
        crop fails is dropped (reference prediction.py:174-176; its index is appended to ``dropped``). ``keep_failed``
        (tile-sharded "rank0" runs, where rank 0 derives every rank's batch structure from the raster's geometry alone): the
        tile stays in the batch as a black stand-in marked ``failed`` — the gather then carries count = -1 for it and rank 0
        writes no file, so the outcome is the reference's there too."""
        staging = None
        raster = getattr(img, "_dev", None)
        if img.dtype == np.uint8 and raster is None:
            need = 0
            for idx in indices:
                _, _, w, h = img.window_of_bounds(tiles[idx]["bounds"])
                need += max(w, 0) * max(h, 0) * img.count
            staging = slot.staging(max(need, 1), self.device)
        # offsets of the tiles in the staging buffer follow from the window sizes alone, so the windows can be copied side by
        # side: an uncompressed raster is read by several threads at once (pread per window row, GIL released) — one thread
        # alone moves ~2.5 GB/s of 4 KB rows, three fp16 engines consume 8 GB/s of 1000x1000x4-byte windows. Compressed
        # rasters keep the sequential walk: their blocks are decoded by the reader's own thread pool already.
        offs, o = [], 0
        for idx in indices:
            offs.append(o)
            if staging is not None:
                _, _, w, h = img.window_of_bounds(tiles[idx]["bounds"])
                o += max(w, 0) * max(h, 0) * img.count
        img._setup_blocks()
        results = None
        if staging is not None and getattr(img, "_flat", None) is not None and not _FAULT_TILE:
            # an uncompressed raster: the batch's windows in ONE library call (td_read_windows: pread per window row on C threads) — no
            # Python per tile on a thread pool (the files-to-files path is bound by the interpreter lock at the fp16 rate). One thread
            # alone moves ~2.5 GB/s of 4-KB rows; three fp16 engines consume 9 GB/s of 1000 x 1000 x 4-byte windows.
            wins = [img.window_of_bounds(tiles[idx]["bounds"]) for idx in indices]
            good = [k for k, (c0, r0, w, h) in enumerate(wins) if w > 0 and h > 0 and img.count >= 3]
            try:
                nthreads = int(os.environ.get("TD_READ_THREADS", "0")) or max(2, min(8, host_core_share() // 2))
                if good and img.read_windows_flat([wins[k] for k in good], staging, [offs[k] for k in good], nthreads):
                    results = [(None, None)] * len(indices)
                    for k in good:
                        c0, r0, w, h = wins[k]
                        t = tiles[indices[k]]
                        m = img.outside_mask(t["bounds"], c0, r0, w, h)
                        if m is not None:
                            img._mask_outside(staging[offs[k]:offs[k] + h * w * img.count].reshape(h, w, img.count), t["bounds"], c0, r0)
                        results[k] = ({"staged": (offs[k], (h, w, img.count))},
                                      {"orig_height": h, "orig_width": w, "height": h, "width": w, "json_name": t["json_name"],
                                       "tile_id": t["tile_id"], "meta": t["meta"]})
                    for k in range(len(indices)):
                        if results[k][0] is None:
                            print(f"Error processing tile {tiles[indices[k]]['json_name']}: Input shapes do not overlap raster.")
            except Exception:
                results = None          # a read error somewhere in the batch: window by window below finds (and drops) the tile it belongs to
        if results is not None:
            pass
        elif staging is not None and getattr(img, "_flat", None) is not None and len(indices) > 1:
            if getattr(self, "_read_pool", None) is None:
                # windows are copied row by row out of the page cache (td_read_window: ~3 us per 4 KB row on tmpfs), eight of them
                # side by side: e2e fp16, 400 tiles — 2 threads 1 758 tiles/s (reader-bound), 4: 1 851, 8: 1 905
                nthreads = int(os.environ.get("TD_READ_THREADS", "0")) or max(2, min(8, host_core_share() // 2))
                self._read_pool = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="td-window")

            def one(k):
                c0 = time.thread_time()
                r = self._process_tile(tiles[indices[k]], img, staging, offs[k])
                with self._stats_lock:
                    self.stats["read_cpu"] += time.thread_time() - c0
                return r
            if self._cpu_stats:
                results = list(self._read_pool.map(one, range(len(indices))))
            else:
                results = list(self._read_pool.map(lambda k: self._process_tile(tiles[indices[k]], img, staging, offs[k]), range(len(indices))))
        else:
            results = [self._process_tile(tiles[idx], img, staging, offs[k]) for k, idx in enumerate(indices)]
        batch = []
        for k, idx in enumerate(indices):
            data, info = results[k]
            off = offs[k]
            if data is None and keep_failed:
                _, _, w, h = img.window_of_bounds(tiles[idx]["bounds"])
                black = np.zeros((h, w, 3), np.uint8)
                data = {"u8": torch.from_numpy(black)}
                if staging is not None:
                    staging[off:off + black.size] = 0
                    data = {"staged": (off, black.shape)}
                info = {"orig_height": h, "orig_width": w, "height": h, "width": w, "json_name": tiles[idx]["json_name"],
                        "tile_id": tiles[idx]["tile_id"], "meta": tiles[idx]["meta"], "failed": True}
            if data is None:
                if dropped is not None:
                    dropped.append(idx)
                continue
            entry = {"data": data, **info}
            if "devwin" in data:
                entry["raster"] = raster
            batch.append(entry)
        return batch

    def _launch_batch(self, batch, slot: _Slot, pred_subdir, tifpath):
        """Forward + asynchronous copy of the packed results to pinned memory; the per-tile host epilogue is queued on
        the worker pool and waits on the slot's event, so this thread goes straight on to the next batch."""
        try:
            images, fmt, hw_valid, hw_out = self._to_model_input(batch, slot)
            dev_out = slot.outputs(self.engine, len(batch), max(h for h, _ in hw_out), max(w for _, w in hw_out), self.device_contours)
            view = {k: v[: len(batch)] for k, v in dev_out.items()}     # leading-dim slices stay contiguous
            self.engine.forward_raw(images, fmt, hw_valid, hw_out, view)
            slot.traced = self._contour_policy()
            if slot.traced:
                self.engine.trace_contours(view, slot.dev_cont, len(batch))
            slot.copy_results(len(batch), slot.traced)
            slot.event.record()
        except BaseException:
            # no epilogue task exists yet that would return the slot: the launcher does (the free list lives as long as the
            # Predictor; predict_on_model logs the image's error and walks on, so a slot lost here would be lost for good)
            self._give_back(slot, self._free)
            raise
        return self._submit_epilogue(batch, slot, pred_subdir, tifpath)

    def _submit_epilogue(self, batch, slot: _Slot, pred_subdir, tifpath) -> list:
        """Queues the host epilogue of a launched batch → its futures. Where only the files are wanted (``return_predictions=False``:
        what predict_on_model runs) and the host traces the borders, the whole batch is ONE task and ONE library call
        (td_batch_prediction_files, its tiles spread over a few C threads): the files-to-files path is bound by the Python work per
        tile at the fp16 rate — every thread's Python runs under one interpreter lock — and a task per tile costs eight times the
        Python of a task per batch. Otherwise one task per tile (:meth:`_process_and_save_single`)."""
        slot.submitted = time.perf_counter()
        if self._batch_epilogue and not self.return_predictions and not slot.traced:
            slot.pending = 1
            return [self._pool.submit(self._save_batch, batch, slot, pred_subdir, tifpath, self._free)]
        slot.pending = len(batch)
        return [self._pool.submit(self._process_and_save_single, b, i, slot, pred_subdir, tifpath, self._free) for i, b in enumerate(batch)]

    def _save_batch(self, batch, slot: _Slot, pred_subdir, tifpath, free: "queue.Queue"):
        try:
            t0 = time.perf_counter()
            c0 = time.thread_time() if self._cpu_stats else 0.0
            slot.event.synchronize()
            t1 = time.perf_counter()
            self._mark("epi", 0)
            self._mark("batch_done", getattr(slot, "mark_id", -1))
            n = len(batch)
            paths = [os.path.join(pred_subdir, f"Prediction_{os.path.basename(b['tile_id'])}.json") for b in batch]
            tr = np.array([b["meta"]["transform"][:6] for b in batch], dtype=np.float64)
            batch_prediction_files(self.device_index, slot.host, n, slot.bits_ptr(0), tr, tifpath, paths, threads=self._epilogue_threads)
            self._mark("epi_done", n - 1)
            with self._stats_lock:
                self.stats["epilogue_wait"] += t1 - t0
                self.stats["epilogue"] += time.perf_counter() - t1
                if self._cpu_stats:
                    self.stats["epilogue_cpu"] += time.thread_time() - c0      # (this thread only: the call's C threads are in the process total)
            return []
        finally:
            with slot.lock:
                slot.pending -= 1
                if slot.pending == 0:
                    self._give_back(slot, free)

    def _process_and_save_single(self, b, i, slot: _Slot, pred_subdir, tifpath, free: "queue.Queue"):
        """Reference prediction.py:198-266 for one tile: polygons of its instance masks → Prediction_<tile>.json."""
        try:
            t0 = time.perf_counter()
            c0 = time.thread_time() if self._cpu_stats else 0.0
            if self._contours_auto:
                self._note_epilogue_start(t0 - slot.submitted > 1e-3 and slot.event.query())
            slot.event.synchronize()
            t1 = time.perf_counter()
            self._mark("epi", i)
            if i == 0:
                self._mark("batch_done", getattr(slot, "mark_id", -1))      # the batch's results have left the GPU (first worker to see it)
            host = slot.host
            output_file = os.path.join(pred_subdir, f"Prediction_{os.path.basename(b['tile_id'])}.json")
            n = int(host["count"][i])
            if slot.traced:
                hc = slot.host_cont
                args = (hc["points"][i], hc["det_info"][i], hc["contour_info"][i], host["mask_region"][i], host["mask_offset"][i])
                tail = (host["scores"][i][:n], host["classes"][i], b["meta"]["transform"], tifpath)
                text = tile_polygons_json_dev(*args, None, *tail)
                if text is None:
                    # a detection too large / too fragmented for the device tracer (its region exceeds the 56-KB label image): this
                    # tile goes through the host path — only the words the paste wrote are fetched, not the worst-case buffer
                    # (12.8 MB per 1000 x 1000 tile: on the compact-crown fixture most tiles hold one such crown)
                    tile_prediction_file(self.device_index, host["mask_region"][i], host["mask_offset"][i], slot.bits_ptr(i),
                                         host["mask_bits"][i], host["scores"][i][:n], host["classes"][i], b["meta"]["transform"],
                                         tifpath, output_file)
            else:
                # rows fetched (only the words written), traced, formatted and written by one call without the GIL
                text = None
                tile_prediction_file(self.device_index, host["mask_region"][i], host["mask_offset"][i], slot.bits_ptr(i),
                                     host["mask_bits"][i], host["scores"][i][:n], host["classes"][i], b["meta"]["transform"],
                                     tifpath, output_file)
            if text is not None:
                with open(output_file, "wb") as f:
                    f.write(text)
            res = []
            if self.return_predictions:
                if text is None:
                    with open(output_file, "rb") as f:
                        text = f.read()
                res = json.loads(text)
            self._mark("epi_done", i)
            with self._stats_lock:
                self.stats["epilogue_wait"] += t1 - t0
                self.stats["epilogue"] += time.perf_counter() - t1
                if self._cpu_stats:
                    self.stats["epilogue_cpu"] += time.thread_time() - c0
            return res
        finally:
            with slot.lock:
                slot.pending -= 1
                if slot.pending == 0:
                    self._give_back(slot, free)

    def _finish_local(self, item, pred_subdir, tifpath, futures, stream=None) -> None:
        """A batch's last phase is enqueued: packed results → pinned memory (on ``stream``), epilogue tasks queued."""
        batch, slot, eng = item["batch"], item["slot"], item["eng"]
        ctx = torch.cuda.stream(stream) if stream is not None else _null_ctx()
        try:
            with ctx:
                slot.traced = self._contour_policy()
                if slot.traced:
                    eng.trace_contours({k: v[: len(batch)] for k, v in slot.dev_out.items()}, slot.dev_cont, len(batch))
                slot.copy_results(len(batch), slot.traced)
                slot.event.record()
        except BaseException:
            self._give_back(slot, self._free)       # no epilogue task will: see _launch_batch
            raise
        futures.extend(self._submit_epilogue(batch, slot, pred_subdir, tifpath))

    def _launch_pipelined(self, ready, prepare, finish, total_rounds: Optional[int] = None) -> None:
        """Launcher loop of a pipelined run. Tick t enqueues, on the main stream, the trunk of batch t, the mask-head
        convs of batch t-2 and the box-head FCs of batch t-1, and on each batch's own side stream the selection phase
        that follows; after a batch's last phase ``finish(item)`` runs (single process: copy the packed results to
        pinned memory on the side stream and queue the epilogue tasks; sharded: the gather to rank 0).
        ``prepare(item, eng)`` → the engine-output views of a new batch. Batches arrive in order from ``ready``;
        ``finish`` is called once per batch, in that same order, ALSO for empty or failed batches (``item["failed"]``)
        — the sharded run pairs one collective with every round on every rank. With ``total_rounds`` the loop ends
        after that many batches (no end marker is expected)."""
        if self.schedule == "streams":
            return self._launch_streams(ready, prepare, finish, total_rounds)
        if getattr(self, "_main", None) is None:
            self._main = torch.cuda.Stream()
            for slot in self._slots:
                slot.side = torch.cuda.Stream()
        main = self._main
        window = []                 # batches in flight, oldest first
        done = False
        launched = 0
        while not done or window:
            if not done:
                t0 = time.perf_counter()
                batch, slot = ready.get()
                self.stats["launch_wait"] += time.perf_counter() - t0
                if isinstance(batch, BaseException) and total_rounds is None:
                    raise batch
                if batch is None:
                    done = True
                elif not batch and total_rounds is None:
                    self._free.put(slot)
                    continue
                else:
                    # engines rotate over the batches: an engine is free for a new batch as soon as its previous
                    # batch's last phase is ENQUEUED (phase 0 waits on that batch's events on the device); the slot —
                    # pinned buffers the host epilogue reads — stays busy longer, hence more slots than engines
                    item = {"batch": batch, "slot": slot, "phase": 0, "eng": self._engines[launched % len(self._engines)],
                            "failed": None, "round": launched}
                    if isinstance(batch, BaseException):
                        item["failed"], item["batch"] = batch, []
                    window.append(item)
                    launched += 1
                    if total_rounds is not None and launched == total_rounds:
                        done = True
            t0 = time.perf_counter()
            # the newest batch's trunk leads the tick, then the mask convs of the oldest, then the FCs of the middle one:
            # every contraction then finds the selection phase it waits on enqueued a whole trunk earlier
            for item in sorted(window, key=lambda it: (0, 2, 1)[it["phase"] // 2]):
                batch, slot, phase, eng = item["batch"], item["slot"], item["phase"], item["eng"]
                if batch and item["failed"] is None:
                    try:
                        if phase == 0:
                            with torch.cuda.stream(main):      # allocations (and their fills) are ordered with the kernels
                                images, fmt, hw_valid, hw_out = self._to_model_input(batch, slot, eng)
                                view = prepare(item, eng, hw_out)
                            eng.forward_phase(0, main, images, fmt, hw_valid, hw_out, view)
                        else:
                            eng.forward_phase(phase, main)
                        eng.forward_phase(phase + 1, slot.side)
                    except Exception as e:
                        if total_rounds is None:
                            # single process: the image fails as a whole. Every batch still in the window holds a slot that
                            # no epilogue task will return (finish() never ran for it) — hand them back before raising
                            for it in window:
                                self._give_back(it["slot"], self._free)
                            raise
                        item["failed"] = e       # sharded: the round still takes part in its gather, with no detections
                item["phase"] = phase + 2
            # batches finish in arrival order (a failed or empty one may be "ready" early: it waits for its elders)
            while window and window[0]["phase"] >= 6:
                item = window.pop(0)
                try:
                    finish(item)
                except BaseException:
                    # finish() hands back its OWN slot when it fails (_finish_local); the batches still in the window hold
                    # slots no epilogue task will ever return — single process: give them back before the image fails
                    if total_rounds is None:
                        for it in window:
                            self._give_back(it["slot"], self._free)
                        del window[:]
                    raise
            self.stats["launch"] += time.perf_counter() - t0

    def _launch_streams(self, ready, prepare, finish, total_rounds: Optional[int] = None) -> None:
        """Launcher loop of the "streams" schedule: batch k runs as ONE whole forward on the HIP stream of engine
        k mod 3 (its input copies, the forward and ``finish`` — result copies or the gather — all on that stream, so an
        engine's batches follow each other in stream order and nothing else needs ordering). Same contract as
        :meth:`_launch_pipelined`: ``finish`` once per batch in arrival order, also for empty / failed batches."""
        if getattr(self, "_eng_streams", None) is None:
            self._eng_streams = [torch.cuda.Stream() for _ in self._engines]
        launched = 0
        while total_rounds is None or launched < total_rounds:
            t0 = time.perf_counter()
            batch, slot = ready.get()
            self.stats["launch_wait"] += time.perf_counter() - t0
            if isinstance(batch, BaseException) and total_rounds is None:
                raise batch
            if batch is None:
                break
            if not batch and total_rounds is None:
                self._free.put(slot)
                continue
            k = launched % len(self._engines)
            eng, stream = self._engines[k], self._eng_streams[k]
            item = {"batch": batch, "slot": slot, "phase": 6, "eng": eng, "failed": None, "round": launched}
            if isinstance(batch, BaseException):
                item["failed"], item["batch"] = batch, []
            launched += 1
            slot.mark_id = launched - 1
            self._mark("launch", launched - 1)
            t0 = time.perf_counter()
            c0 = time.thread_time() if self._cpu_stats else 0.0
            slot.side = stream               # where finish() enqueues this batch's copies / gather
            if item["batch"] and item["failed"] is None:
                try:
                    with torch.cuda.stream(stream):      # allocations (and their fills) are ordered with the kernels
                        images, fmt, hw_valid, hw_out = self._to_model_input(item["batch"], slot, eng)
                        view = prepare(item, eng, hw_out)
                        eng.forward_raw(images, fmt, hw_valid, hw_out, view)
                except Exception as e:
                    if total_rounds is None:
                        self._give_back(slot, self._free)       # taken from `ready`, never handed to an epilogue task
                        raise
                    item["failed"] = e       # sharded: the round still takes part in its gather, with no detections
            finish(item)
            self.stats["launch"] += time.perf_counter() - t0
            if self._cpu_stats:
                self.stats["launch_cpu"] += time.thread_time() - c0
            self._mark("launch_done", launched - 1)

    @staticmethod
    def _transient_slot() -> _Slot:
        slot = _Slot()
        slot.transient = True
        return slot

    def _drain(self, reader, ready, futures, stop) -> None:
        """Leaves no work of this call behind (every exit path of ``__call__`` runs it): the reader thread is told to
        stop and unblocked, and every epilogue task that was submitted is waited for — their slots' pinned buffers
        must not be refilled by the next image while a worker of this one still reads them, and a late worker must
        not hand a slot to the next call's free list a second time."""
        stop.set()
        def put_back(item):
            if isinstance(item, tuple) and len(item) == 2 and isinstance(item[1], _Slot):
                self._give_back(item[1], self._free)        # a batch that was read but never launched
        while reader.is_alive():
            try:
                put_back(ready.get(timeout=0.05))
            except queue.Empty:
                pass
            # the reader may also be waiting for a free slot: hand it one it will not use (never a real slot: those
            # may still be in flight, and a second copy in the list would be handed out twice)
            if self._free.empty():
                self._free.put(self._transient_slot())
        reader.join()
        while True:
            try:
                put_back(ready.get_nowait())
            except queue.Empty:
                break
        # a stand-in the reader did not consume must not stay in the lifetime free list: the next image would use it as a
        # real slot (a full pinned + device buffer set for one use; under schedule="phases" it has no side stream)
        kept = []
        while True:
            try:
                sl = self._free.get_nowait()
            except queue.Empty:
                break
            if not sl.transient:
                kept.append(sl)
        for sl in kept:
            self._free.put(sl)
        for f in futures:
            try:
                f.result()
            except BaseException:
                pass
        try:
            torch.cuda.synchronize(self.device_index)
        except Exception:
            pass

    def _run_single(self, tiles, img: GeoTiff, pred_subdir, tifpath):
        return self._start_single(tiles, img, pred_subdir, tifpath).result()

    def _start_single(self, tiles, img: GeoTiff, pred_subdir, tifpath) -> "_PendingImage":
        """Reads and launches every batch of one image; returns when the last batch is ENQUEUED (the raster is no longer
        needed then). What is still running — the last forwards on the GPU, the epilogue workers of the last batches — is
        waited for by ``.result()`` of the returned handle; meanwhile the next image may be started: its batches queue up
        behind these on the engines' streams and its reader takes slots as the epilogue workers return them."""
        B = self.max_batch_size
        rounds = [list(range(r * B, min((r + 1) * B, len(tiles)))) for r in range((len(tiles) + B - 1) // B)]
        self._roll_stats()
        ready: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()
        dropped: List[int] = []

        def reader():
            try:
                for k, indices in enumerate(rounds):
                    self._mark("slot_wait", k)
                    t0 = time.perf_counter()
                    slot = self._free.get()
                    self.stats["slot_wait"] += time.perf_counter() - t0
                    if stop.is_set():
                        self._give_back(slot, self._free)
                        return
                    self._mark("read", k)
                    t0 = time.perf_counter()
                    c0 = time.thread_time() if self._cpu_stats else 0.0
                    try:
                        batch = self._read_batch(tiles, indices, img, slot, dropped=dropped)
                    except BaseException:
                        self._give_back(slot, self._free)
                        raise
                    self.stats["read"] += time.perf_counter() - t0
                    if self._cpu_stats:
                        with self._stats_lock:
                            self.stats["read_cpu"] += time.thread_time() - c0
                            self.stats["tiles"] += len(batch)
                    self._mark("read_done", k)
                    ready.put((batch, slot))
                ready.put((None, None))
            except BaseException as e:      # surfaces in the launcher thread
                ready.put((e, None))

        t = threading.Thread(target=reader, name="td-tile-reader", daemon=True)
        t.start()
        futures: list = []
        try:
            while not self.pipeline:
                t0 = time.perf_counter()
                batch, slot = ready.get()
                self.stats["launch_wait"] += time.perf_counter() - t0
                if isinstance(batch, BaseException):
                    raise batch
                if batch is None:
                    break
                if not batch:
                    self._free.put(slot)
                    continue
                t0 = time.perf_counter()
                futures.extend(self._launch_batch(batch, slot, pred_subdir, tifpath))
                self.stats["launch"] += time.perf_counter() - t0
            if self.pipeline:
                def prepare(item, eng, hw_out):
                    slot, n = item["slot"], len(item["batch"])
                    dev_out = slot.outputs(eng, n, max(h for h, _ in hw_out), max(w for _, w in hw_out), self.device_contours)
                    return {k: v[:n] for k, v in dev_out.items()}      # leading-dim slices stay contiguous
                self._launch_pipelined(ready, prepare,
                                       lambda item: self._finish_local(item, pred_subdir, tifpath, futures, item["slot"].side))
            t.join()
            return _PendingImage(self, futures, dropped)
        except BaseException:
            self._drain(t, ready, futures, stop)
            raise

    # -- multi-GPU: tiles shard over ranks, detections gather to rank 0 ------------------------------------------
    def _tile_ok(self, tile, img: GeoTiff) -> bool:
        """Whether a tile yields a model input at all (reference: a failing crop is skipped, prediction.py:174-176).
        Decided from the raster's geometry alone, so every rank reaches the same verdict for every tile."""
        _, _, w, h = img.window_of_bounds(tile["bounds"])
        return w > 0 and h > 0 and img.count >= 3

    def _run_sharded(self, tiles, img: GeoTiff, pred_subdir, tifpath):
        """Rank r predicts tiles r, r+W, r+2W, ... in batches (the plain loop or, with ``pipeline``, three engines with
        three batches in flight, exactly as a single process runs them); after a batch's last phase its fixed-shape
        detection tensors (count, boxes, scores, 28x28 probabilities) go to rank 0 with ``torch.distributed.gather``
        (RCCL on GPUs: enqueued on the batch's side stream and ordered on the device, nobody waits on the host).
        Rank 0 pastes each rank's batch with one td_paste_masks_batch launch, copies the packed masks to pinned memory
        and its worker threads write the ``Prediction_*.json`` files — the same epilogue as the single-process path.
        Tiles that cannot be cropped are left out by a rule every rank evaluates alike (:meth:`_tile_ok`), so rank 0
        knows each batch's tiles without any metadata exchange.

        Every rank issues exactly ``rounds`` gathers per image whatever happens to it: a reader or launch failure
        turns that round into an empty batch (count = 0), the error is kept, and after the last round all ranks agree
        (one all-reduce) whether the image failed — then every rank raises, so the caller's log-and-continue
        (detection.py:117-120) moves all of them to the next image together."""
        import torch.distributed as dist
        B, W, me = self.max_batch_size, D.world(), D.rank()
        Dn = self.engine.D
        dev = torch.device(self.device)
        host_backend = dist.get_backend() != "nccl"          # gloo (tests / one-GPU rehearsal) moves host tensors
        ok = [self._tile_ok(t, img) for t in tiles]
        per_rank = max(len(D.shard_indices(len(tiles), r, W)) for r in range(W))
        rounds = (per_rank + B - 1) // B
        # batches are cut from the shard BEFORE dropping bad tiles, so the round structure is the same on every rank
        def round_tiles(r, k):
            return [i for i in D.shard_indices(len(tiles), r, W)[k * B:(k + 1) * B] if ok[i]]
        self.stats = dict.fromkeys(self.stats, 0.0)
        free = self._free = queue.Queue()       # this image's free list: tasks capture it, never look it up later
        for s in self._slots:
            free.put(s)
        ready: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()
        errors: List[BaseException] = []

        def reader():
            k = 0
            try:
                for k in range(rounds):
                    slot = self._free.get()
                    if stop.is_set():
                        return
                    t0 = time.perf_counter()
                    try:
                        batch = self._read_batch(tiles, round_tiles(me, k), img, slot, keep_failed=True)
                    except Exception as e:          # this round runs empty; the image is reported failed at the end
                        batch = e
                    self.stats["read"] += time.perf_counter() - t0
                    ready.put((batch, slot))
            except BaseException as e:              # cannot get slots any more: fail the remaining rounds, keep the count
                for _ in range(k, rounds):
                    # a stand-in slot per failed round, NOT one of self._slots: those may be in flight for other
                    # batches, and finish() zeroes / gathers into / pastes from the slot it is given
                    ready.put((e, self._transient_slot()))

        th = threading.Thread(target=reader, name="td-tile-reader", daemon=True)
        th.start()
        futures, predictions = [], []

        def prepare(item, eng, hw_out):
            slot, n = item["slot"], len(item["batch"])
            send = slot.gather_buffers(B, Dn, dev)          # count / boxes / scores / mask_probs padded to B rows
            view = {key: send[key][:n] for key in ("count", "boxes", "scores", "mask_probs")}
            view["classes"] = slot.classes_buffer(B, Dn, dev)[:n]
            return view

        def finish(item, stream=None):
            """Gather of round item["round"] (+ on rank 0: paste, copies, epilogue tasks) on ``stream``."""
            slot, k = item["slot"], item["round"]
            n = 0 if item["failed"] is not None else len(item["batch"])
            if item["failed"] is not None:
                errors.append(item["failed"])
            ctx = torch.cuda.stream(stream) if stream is not None else _null_ctx()
            with ctx:
                send = slot.gather_buffers(B, Dn, dev)
                if n < B:
                    send["count"][n:].zero_()
                for j in range(n):
                    if item["batch"][j].get("failed"):          # a black stand-in for a tile whose crop failed: no file for it
                        send["count"][j:j + 1].fill_(-1)
                recv = slot.recv_buffers(W, B, Dn, dev, host_backend) if me == 0 else None
                for key in D.GATHER_KEYS:
                    t = send[key].cpu() if host_backend else send[key]
                    dist.gather(t, [g[key] for g in recv] if me == 0 else None, dst=0)
                if me != 0:
                    # the slot (pinned staging included) may be refilled once this batch's copies and sends have run
                    slot.event.record()
                    futures.append(self._pool.submit(self._release_when_done, slot, free))
                    return
                jobs = []
                try:
                    for src in range(W):
                        idx = round_tiles(src, k)
                        if not idx:
                            continue
                        hw = [img.window_of_bounds(tiles[i]["bounds"]) for i in idx]
                        hw = [(h, w) for _, _, w, h in hw]
                        g = recv[src] if not host_backend else {key: v.to(dev, non_blocking=True) for key, v in recv[src].items()}
                        pin = slot.paste(self.engine, src, g, hw, B, Dn, dev)
                        jobs.append((src, idx, pin))
                except Exception as e:      # rank 0 keeps receiving the later rounds; the image is reported failed
                    errors.append(e)
                    jobs = []
                slot.event.record()
            slot.pending = sum(len(idx) for _, idx, _ in jobs)
            if slot.pending == 0:
                futures.append(self._pool.submit(self._release_when_done, slot, free))
            for src, idx, pin in jobs:
                for j, ti in enumerate(idx):
                    futures.append(self._pool.submit(self._save_gathered, slot, pin, j, tiles[ti], pred_subdir, tifpath, free))

        try:
            if self.pipeline:
                self._launch_pipelined(ready, prepare, lambda item: finish(item, item["slot"].side), total_rounds=rounds)
            else:
                for k in range(rounds):
                    t0 = time.perf_counter()
                    batch, slot = ready.get()
                    self.stats["launch_wait"] += time.perf_counter() - t0
                    t0 = time.perf_counter()
                    item = {"batch": batch, "slot": slot, "eng": self.engine, "failed": None, "round": k}
                    if isinstance(batch, BaseException):
                        item["failed"], item["batch"] = batch, []
                    if item["batch"]:
                        try:
                            images, fmt, hw_valid, hw_out = self._to_model_input(item["batch"], slot)
                            self.engine.forward_raw(images, fmt, hw_valid, hw_out, prepare(item, self.engine, hw_out))
                        except Exception as e:
                            item["failed"] = e
                    finish(item)
                    self.stats["launch"] += time.perf_counter() - t0
            th.join()
            for f in futures:
                try:
                    predictions.extend(f.result())
                except Exception as e:
                    errors.append(e)
        except BaseException as e:          # not expected (every per-round failure is caught above): leave nothing behind
            errors.append(e)
            self._drain(th, ready, futures, stop)
        if not D.all_ok(not errors):
            if errors:
                raise errors[0]
            raise RuntimeError(f"another rank failed while predicting {tifpath}")
        return predictions

    def _run_local_shard(self, tiles, img: GeoTiff, pred_subdir, tifpath):
        """``sharded_epilogue="local"``: rank r predicts tiles r, r+W, ... with the single-process pipeline INCLUDING
        paste, contours and the ``Prediction_*.json`` files of its own tiles (all ranks of a node share the output
        folder), so no host, PCIe link or GPU carries another rank's masks; what rank 0 collects is a manifest —
        (tile index, detections) per tile — and it checks that every tile of the image was written exactly once.
        For nodes where rank 0's link cannot take every rank's packed masks (8 fp16 engines produce them faster than
        one PCIe link moves them)."""
        me, W = D.rank(), D.world()
        mine = D.shard_indices(len(tiles), me, W)
        err, preds, dropped = None, [], []
        try:
            handle = self._start_single([tiles[i] for i in mine], img, pred_subdir, tifpath)
            dropped = [mine[k] for k in handle.dropped]        # crops that failed: no file, as in the reference (174-176)
            preds = handle.result()
        except Exception as e:
            err = e
        written = []
        for i in mine:
            fn = os.path.join(pred_subdir, f"Prediction_{os.path.basename(tiles[i]['tile_id'])}.json")
            if os.path.exists(fn):
                written.append(i)
        manifests = D.gather_objects((written, dropped))
        if me == 0 and err is None:
            seen = sorted(i for m, _ in manifests for i in m)
            gone = set(i for _, d in manifests for i in d)
            expect = [i for i in range(len(tiles)) if self._tile_ok(tiles[i], img) and i not in gone]
            if len(seen) != len(set(seen)) or not set(expect) <= set(seen):
                err = RuntimeError(f"sharded prediction of {tifpath}: {len(set(expect) - set(seen))} tiles missing, "
                                   f"{len(seen) - len(set(seen))} written twice")
        if not D.all_ok(err is None):
            raise err if err is not None else RuntimeError(f"another rank failed while predicting {tifpath}")
        return preds

    @staticmethod
    def _give_back(slot: _Slot, free: "queue.Queue") -> None:
        """Slot → the free list of the call that owns it (``free`` is captured when the task is created: a task that
        outlives its image must not feed the NEXT call's fresh list a second copy of the slot)."""
        if not slot.transient:
            free.put(slot)

    def _release_when_done(self, slot: _Slot, free: "queue.Queue"):
        slot.event.synchronize()
        self._give_back(slot, free)
        return []

    def _save_gathered(self, slot: _Slot, pin, j, tile, pred_subdir, tifpath, free: "queue.Queue"):
        try:
            slot.event.synchronize()
            n = int(pin["count"][j])
            if n < 0:           # the owning rank could not crop this tile: dropped, as the reference drops it (prediction.py:174-176)
                return []
            path = os.path.join(pred_subdir, f"Prediction_{os.path.basename(tile['tile_id'])}.json")
            tile_prediction_file(self.device_index, pin["mask_region"][j], pin["mask_offset"][j],
                                 pin["bits_base"] + j * pin["bits_stride"], pin["mask_bits"][j], pin["scores"][j][:n],
                                 np.zeros(n, np.int32), tile["meta"]["transform"], tifpath, path)
            if not self.return_predictions:
                return []
            with open(path, "rb") as f:
                return json.loads(f.read())
        finally:
            with slot.lock:
                slot.pending -= 1
                if slot.pending == 0:
                    self._give_back(slot, free)

    def submit(self, tifpath, tilepath, whole_image: bool = False) -> "_PendingImage":
        """Starts an image THIS process predicts alone and returns once its last batch is enqueued; ``.result()`` of the handle
        waits for its tile files (and returns the predictions list). Between the two the caller may submit the next image, so
        one image's drain (its last forwards and epilogues, ≈ 20 ms) overlaps the next image's fill — what
        ``detection.predict_on_model`` does. ``predictor(tif, tiles)`` = ``predictor.submit(tif, tiles).result()`` in a single
        process. Under ``torch.distributed`` an image is by default one collective structure all ranks enter together
        (``__call__``); ``whole_image=True`` says this rank owns the image whole (image-level sharding, detection.py): no
        collective is issued, every tile is read, predicted, traced and written here, and a tile whose crop fails is dropped
        as the reference drops it (prediction.py:174-176)."""
        if D.world() != 1 and not whole_image:
            raise RuntimeError("Predictor.submit under torch.distributed needs whole_image=True (this rank owns the image); "
                               "tile-sharded images go through __call__ (one collective structure per image)")
        pred_subdir = os.path.join(self.output_dir, os.path.basename(tifpath).replace(".tif", "").replace(".json", ""))
        os.makedirs(pred_subdir, exist_ok=True)
        if self._trace is not None and os.environ.get("TD_E2E_TRACE") != "keep":      # "keep": one trace over a chained walk of images
            del self._trace[:]
        self._mark("call")
        tiles = self._load_tiles(tilepath)
        self._mark("tiles_loaded")
        raster = self._device_raster(tifpath)          # an LZW raster decoded in HBM (prefetched while the previous image ran), or None
        img = GeoTiff(tifpath)
        img._dev = raster
        self._mark("raster_open")
        try:
            pending = self._start_single(tiles, img, pred_subdir, tifpath)
            pending.keep = raster                      # alive until the image's last forward has read its windows
            return pending
        finally:
            img.close()

    def __call__(self, tifpath, tilepath):
        if D.world() == 1:
            try:
                return self.submit(tifpath, tilepath).result()
            finally:
                self._mark("call_done")
        pred_subdir = os.path.join(self.output_dir, os.path.basename(tifpath).replace(".tif", "").replace(".json", ""))
        os.makedirs(pred_subdir, exist_ok=True)
        # sharded: every rank must enter (or skip) the image together — the per-round gathers only pair up if all
        # ranks read the same tile list from the same raster
        tiles, img, err = None, None, None
        try:
            tiles = self._load_tiles(tilepath)
            img = GeoTiff(tifpath)
        except Exception as e:
            err = e
        try:
            if not D.all_ok(err is None):
                raise err if err is not None else RuntimeError(f"another rank could not open {tifpath} / {tilepath}")
            if self.sharded_epilogue == "local":
                return self._run_local_shard(tiles, img, pred_subdir, tifpath)
            return self._run_sharded(tiles, img, pred_subdir, tifpath)
        finally:
            if img is not None:
                img.close()


class _PendingImage:
    """An image whose batches are all enqueued (Predictor.submit): ``result()`` waits for its epilogue tasks — every
    ``Prediction_*.json`` of the image is on disk when it returns — and gives the predictions list (empty with
    ``return_predictions=False``). The first failed tile task is re-raised after ALL tasks have finished, so no worker of this
    image is still reading its slot when the caller moves on."""

    def __init__(self, predictor: "Predictor", futures, dropped=None):
        self._predictor, self._futures, self._done = predictor, futures, None
        self.dropped = dropped if dropped is not None else []      # indices (into the image's tile list) whose crop failed
        self.keep = None                                           # the image's device-resident raster, if it has one

    def result(self):
        if self._done is None:
            preds, err = [], None
            for f in self._futures:
                try:
                    preds.extend(f.result())
                except BaseException as e:      # keep waiting for the others: their slots must be back before anyone re-uses them
                    err = err or e
            self._done = (preds, err)
            self._futures = None
            self.keep = None
        preds, err = self._done
        if err is not None:
            raise err
        return preds


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _ring_entries(sub: np.ndarray, x0: int, y0: int, score: float, cls: int, transform, tifpath, out: List[dict]) -> None:
    for contour in find_contours(sub):
        if contour.size < 8:
            continue
        cx = (contour[:, 0] + x0).tolist()
        cy = (contour[:, 1] + y0).tolist()
        if (cx[0], cy[0]) != (cx[-1], cy[-1]):
            cx.append(cx[0])
            cy.append(cy[0])
        gx, gy = xy(transform, rows=cy, cols=cx)
        out.append({"image_id": tifpath, "category_id": cls, "score": score,
                    "polygon_coords": [[[float(a), float(b)] for a, b in zip(gx, gy)]]})


def polygons_from_packed(regions, offsets, bits, scores, classes, transform, tifpath) -> List[dict]:
    """Same result as :func:`polygons_from_masks`, straight from the engine's packed bit rows: each detection's
    paste region is unpacked on its own (no full-frame [n,h,w] array; the mask is zero outside its region)."""
    words = np.asarray(bits).view(np.uint32)
    out: List[dict] = []
    for d in range(len(scores)):
        x0, y0, x1, y1 = (int(v) for v in regions[d])
        if x1 <= x0 or y1 <= y0:
            continue
        wpr = (x1 - x0 + 31) // 32
        o = int(offsets[d])
        rows = words[o:o + wpr * (y1 - y0)].reshape(y1 - y0, wpr)
        sub = np.unpackbits(rows.view(np.uint8).reshape(y1 - y0, wpr * 4), axis=1, bitorder="little")[:, : x1 - x0]
        _ring_entries(sub, x0, y0, float(scores[d]), int(classes[d]), transform, tifpath, out)
    return out


def polygons_from_masks(masks: np.ndarray, regions: np.ndarray, scores, classes, transform, tifpath) -> List[dict]:
    """Reference prediction.py:229-261 for one tile: every contour with >= 4 points (``contour.size >= 8``) of every
    instance mask becomes one entry; the ring is closed; pixel-corner coordinates go through the tile's affine."""
    evaluations = []
    for d in range(masks.shape[0]):
        x0, y0, x1, y1 = (int(v) for v in regions[d])
        if x1 <= x0 or y1 <= y0:
            continue
        sub = masks[d, y0:y1, x0:x1]        # the mask is zero outside its paste region
        for contour in find_contours(sub):
            if contour.size < 8:
                continue
            cx = (contour[:, 0] + x0).tolist()
            cy = (contour[:, 1] + y0).tolist()
            if (cx[0], cy[0]) != (cx[-1], cy[-1]):
                cx.append(cx[0])
                cy.append(cy[0])
            gx, gy = xy(transform, rows=cy, cols=cx)
            evaluations.append({"image_id": tifpath, "category_id": int(classes[d]), "score": float(scores[d]),
                                "polygon_coords": [[[float(a), float(b)] for a, b in zip(gx, gy)]]})
    return evaluations
