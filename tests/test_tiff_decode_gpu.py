"""LZW rasters decoded on the GPU (tiffdecode.hip; SURVEY.md §8f-2 — the reference reads every tile window through rasterio /
GDAL / libtiff on the host, prediction.py:61,164): the decoded raster in HBM must equal the host reader's pixels byte for
byte, for every block layout the writer and libtiff produce; a corrupt block is reported and the Predictor falls back to the
host reader; prediction files are byte-identical with the device decoder on or off."""
import json
import os

import numpy as np
import pytest
import torch

from treedetection_amd import _lib
from treedetection_amd.geotiff import GeoTiff, write_geotiff
from treedetection_amd.synth import make_tile
from treedetection_amd.weights import make_synthetic_state_dict

pytestmark = pytest.mark.gpu
T = (0.2, 0.0, 412000.0, 0.0, -0.2, 5318100.0)


def _raster(bands, h, w, seed=0):
    rgb, _ = make_tile(seed, max(h, w))
    img = np.concatenate([rgb, rgb[..., 1:2]], axis=2)[:h, :w, :bands].transpose(2, 0, 1).copy()
    img[:, h // 5: h // 2, w // 8: w // 2] = 7                      # a flat area: long strings, KwKwK codes
    img[:, -40:, -90:] = np.arange(90, dtype=np.uint8)              # ramps: constant differences under predictor 2
    return img


@pytest.fixture(params=["large", "small"])
def ring(request, monkeypatch):
    """Both LDS footprints of the block decoders (tiffdecode.hip: the ring of recent output — 16 KB / 4 KB for LZW, 32 KB / 8 KB for
    DEFLATE; the library takes the small one by itself once a raster has more blocks than fit the chip in one round)."""
    monkeypatch.setenv("TD_DECODE_RING", request.param)
    return request.param


@pytest.mark.parametrize("codec", ["lzw", "deflate"])
@pytest.mark.parametrize("bands", [3, 4, 1])
@pytest.mark.parametrize("kw", [{"tile": (128, 128)}, {"tile": (64, 256), "predictor": 2}, {"rows_per_strip": 7}, {"rows_per_strip": 1, "predictor": 2},
                                {"rows_per_strip": 64, "predictor": 2}, {}])
def test_device_decode_equals_the_host_reader(tmp_path, kw, bands, codec, ring):
    img = _raster(bands, 517, 683, seed=bands)
    path = str(tmp_path / "r.tif")
    write_geotiff(path, img, T, 25832, compression=codec, **kw)
    g = GeoTiff(path)
    assert g.device_decodable()
    image, check = g.decode_to_device("cuda:0")
    got = check().cpu().numpy()
    assert got.shape == (517, 683, bands)
    assert np.array_equal(got.transpose(2, 0, 1), img)
    assert np.array_equal(got.transpose(2, 0, 1), GeoTiff(path).read())
    assert check.compressed_bytes < img.nbytes


def test_device_decode_of_a_file_written_by_libtiff(tmp_path):
    """libtiff's own encoder (through Pillow): strips of its choosing, its ClearCode / width-change timing."""
    from PIL import Image
    img = _raster(3, 700, 900, seed=5)
    path = str(tmp_path / "pil.tif")
    Image.fromarray(img.transpose(1, 2, 0)).save(path, compression="tiff_lzw")
    g = GeoTiff(path)
    assert g.compression == 5 and g.device_decodable()
    image, check = g.decode_to_device("cuda:0")
    assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), img)


def test_deflate_streams_of_every_block_type_on_the_device(tmp_path, ring):
    """zlib streams as libtiff / GDAL write them at any setting: stored blocks (level 0, and what zlib emits for noise), fixed Huffman
    codes (tiny strips), dynamic codes (levels 1 / 6 / 9), run-length and Huffman-only strategies, matches at the far end of the 32-KB
    window, a file written by libtiff (through Pillow: 'tiff_adobe_deflate'). The device decoder (inflate_core.h, one wave per block)
    against zlib's own output, byte for byte; the same source runs on the host as td_tiff_inflate (tests/test_geotiff_formats.py)."""
    import struct
    import zlib
    from PIL import Image
    lib = _lib.load()
    rng = np.random.default_rng(7)
    img = _raster(4, 300, 1100, seed=2)
    raws = [img.transpose(1, 2, 0).tobytes(), bytes(200000), rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(), b"x",
            (b"abcdefghijklmnopqrstuvwxyz0123456789" * 3000), rng.integers(0, 3, 250000, dtype=np.uint8).tobytes()]
    # a match that reaches back exactly 32 768 bytes
    far = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    raws.append(far + far[:300] + far[5:400])
    streams, expect = [], []
    for raw in raws:
        for level, strategy in ((0, 0), (1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY)):
            c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
            streams.append(c.compress(raw) + c.flush())
            expect.append(raw)
    cap = max(len(r) for r in raws) + 64
    offs, blob = [], bytearray()
    for st in streams:
        blob += b"\0" * ((-len(blob)) % 3)                      # odd alignments on purpose
        offs.append(len(blob))
        blob += st
    blob += b"\0" * 16
    comp = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
    d_off = torch.tensor(offs, dtype=torch.int64, device="cuda")
    d_n = torch.tensor([len(st) for st in streams], dtype=torch.int64, device="cuda")
    out = torch.zeros((len(streams), cap), dtype=torch.uint8, device="cuda")
    dec = torch.zeros((len(streams),), dtype=torch.int64, device="cuda")
    status = torch.full((len(streams),), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.td_tiff_inflate_dev(comp.data_ptr(), d_off.data_ptr(), d_n.data_ptr(), len(streams), out.data_ptr(), cap, dec.data_ptr(),
                                       status.data_ptr(), _lib.stream_ptr()), "td_tiff_inflate_dev")
    torch.cuda.synchronize()
    assert (status == 0).all(), status.cpu().tolist()
    got = out.cpu().numpy()
    for k, raw in enumerate(expect):
        assert int(dec[k]) == len(raw), (k, int(dec[k]), len(raw))
        assert got[k, :len(raw)].tobytes() == raw, k
    # corrupt streams are reported per block, the others still decode
    bad = bytearray(blob)
    bad[offs[2] + 40] ^= 0x5a
    comp2 = torch.from_numpy(np.frombuffer(bytes(bad), dtype=np.uint8).copy()).cuda()
    _lib.check(lib.td_tiff_inflate_dev(comp2.data_ptr(), d_off.data_ptr(), d_n.data_ptr(), len(streams), out.data_ptr(), cap, dec.data_ptr(),
                                       status.data_ptr(), _lib.stream_ptr()), "td_tiff_inflate_dev")
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert (st[:2] == 0).all() and (st[3:] == 0).all() and (st[2] != 0 or int(dec[2]) != len(expect[2]) or out[2, :len(expect[2])].cpu().numpy().tobytes() != expect[2])
    # a file libtiff wrote
    path = str(tmp_path / "pil_deflate.tif")
    rgb = _raster(3, 700, 900, seed=5)
    Image.fromarray(rgb.transpose(1, 2, 0)).save(path, compression="tiff_adobe_deflate")
    g = GeoTiff(path)
    assert g.compression in (8, 32946) and g.device_decodable()
    image, check = g.decode_to_device("cuda:0")
    assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), rgb)


def test_large_blocks_with_many_table_clears_and_incompressible_data(tmp_path, ring):
    """One strip = the whole raster (4 MB decoded: hundreds of table clears in one stream), noise (mostly literal codes) and a raster of
    one value (strings up to thousands of bytes, copied 64 bytes per step)."""
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (4, 1000, 1000), dtype=np.uint8)
    flat = np.full((3, 900, 1100), 201, np.uint8)
    # a pattern that repeats every 20 000 bytes: every string / match points further back than the small rings hold
    period = rng.integers(0, 256, 20000, dtype=np.uint8)
    far = np.resize(period, 3 * 600 * 700).reshape(600, 700, 3).transpose(2, 0, 1).copy()
    for name, img, kw in (("noise", noise, {}), ("flat", flat, {}), ("flat_tiles", flat, {"tile": (512, 512), "predictor": 2}), ("far", far, {})):
        for codec in ("lzw", "deflate"):
            path = str(tmp_path / f"{name}_{codec}.tif")
            write_geotiff(path, img, T, 25832, compression=codec, **kw)
            image, check = GeoTiff(path).decode_to_device("cuda:0")
            assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), img), (name, codec)


def test_rasters_with_more_blocks_than_one_round_take_the_small_ring_by_themselves(tmp_path, monkeypatch):
    """No override: 1 700 one-row strips (beyond 256 CUs x 4 DEFLATE waves → the library picks the small DEFLATE ring by itself)."""
    monkeypatch.delenv("TD_DECODE_RING", raising=False)
    img = _raster(4, 1700, 640, seed=9)
    for codec in ("lzw", "deflate"):
        path = str(tmp_path / f"many_{codec}.tif")
        write_geotiff(path, img, T, 25832, compression=codec, rows_per_strip=1, predictor=2)
        image, check = GeoTiff(path).decode_to_device("cuda:0")
        assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), img), codec


def test_more_blocks_than_decoder_waves_go_by_ticket(tmp_path, monkeypatch):
    """3 300 one-row strips: more blocks than the chip holds decoder waves (256 CUs x 12 DEFLATE / x 6 LZW waves, one workgroup per
    CU) — every wave fetches several blocks from the launch's counter (tiffdecode.hip: take_block); two decodes in a row reuse
    the waves' LDS and take fresh counters."""
    monkeypatch.delenv("TD_DECODE_RING", raising=False)
    img = _raster(4, 3300, 96, seed=4)
    for codec in ("lzw", "deflate"):
        path = str(tmp_path / f"ticket_{codec}.tif")
        write_geotiff(path, img, T, 25832, compression=codec, rows_per_strip=1, predictor=2)
        for _ in range(2):
            image, check = GeoTiff(path).decode_to_device("cuda:0")
            assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), img), codec


def test_decode_on_the_low_priority_stream_beside_other_work(tmp_path):
    """The Predictor decodes the NEXT image on a stream of the lowest HIP priority (_lib.low_priority_stream: its own hardware queue,
    one per device for the life of the process) while forwards run on other streams: same bytes, and the stream is what it says."""
    import ctypes as C
    low = _lib.low_priority_stream(0)
    assert _lib.low_priority_stream(0) is low
    lib = _lib.load()
    least, greatest, prio = C.c_int(0), C.c_int(0), C.c_int(123)
    assert lib.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) == 0
    assert lib.hipStreamGetPriority(C.c_void_p(low.cuda_stream), C.byref(prio)) == 0
    assert prio.value == least.value and least.value >= greatest.value
    img = _raster(4, 1500, 1100, seed=6)
    paths = {}
    for codec in ("lzw", "deflate"):
        paths[codec] = str(tmp_path / f"low_{codec}.tif")
        write_geotiff(paths[codec], img, T, 25832, compression=codec, tile=(128, 128), predictor=2)
    a = torch.randn((4096, 4096), device="cuda")
    pinned = [None]
    for codec in ("lzw", "deflate"):
        busy = torch.tanh(a * 1.0001) + a              # other work on the current stream while the decode runs
        image, check = GeoTiff(paths[codec]).decode_to_device("cuda:0", low, pinned)
        busy = torch.tanh(busy) * a
        assert np.array_equal(check().cpu().numpy().transpose(2, 0, 1), img), codec
    torch.cuda.synchronize()
    assert bool(torch.isfinite(busy).any())


def test_a_corrupt_block_is_reported_and_the_predictor_falls_back(tmp_path, capsys):
    import treedetection_amd as TD
    from treedetection_amd.preprocessing import tile_single_file
    img = _raster(3, 500, 500, seed=9)
    good, bad = str(tmp_path / "good.tif"), str(tmp_path / "bad.tif")
    write_geotiff(good, img, T, 25832, compression="lzw", tile=(128, 128))
    g = GeoTiff(good)
    g._setup_blocks()
    raw = bytearray(open(good, "rb").read())
    off = g._offs[5]
    raw[off + 2: off + 6] = b"\xff\xff\xff\xff"                     # codes beyond the table in block 5
    open(bad, "wb").write(bytes(raw))
    image, check = GeoTiff(bad).decode_to_device("cuda:0")
    with pytest.raises(ValueError, match="block 5"):
        check()
    # the Predictor logs it and serves the image through the host reader — whose decoder rejects the same block: the tiles that
    # need it are dropped (reference prediction.py:174-176), the others are predicted
    tile_single_file(bad, str(tmp_path / "tiles"), buffer=0, tile_width=25, tile_height=25)
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    cfg = TD.setup_model_cfg(update_model="x", device="0")
    with TD.Predictor(cfg, device_type="0", max_batch_size=4, output_dir=str(tmp_path / "out"), state_dict=sd) as pred:
        pred(bad, str(tmp_path / "tiles" / "bad.json"))
        assert pred.decode_stats["images"] == 0
    assert "using the host reader" in capsys.readouterr().out
    files = os.listdir(tmp_path / "out" / "bad")
    assert 0 < len(files) < 16


def test_prediction_files_are_identical_with_the_device_decoder_on_or_off(tmp_path):
    """An LZW raster (4 bands, tiles with predictor 2) through the Predictor: windows cut in HBM (device_decode auto) against the
    host reader (false), plus the same pixels stored uncompressed (uploaded whole to HBM, or read window by window on the host) —
    byte-identical Prediction_*.json; bounds that do not lie on
    pixel edges go through rasterio.mask's centre rule on the device as on the host."""
    import treedetection_amd as TD
    from treedetection_amd.preprocessing import tile_single_file
    rgb, _ = make_tile(300, 500)
    rgbi = np.ascontiguousarray(np.concatenate([rgb, rgb[..., 1:2]], axis=2).transpose(2, 0, 1))
    sd = make_synthetic_state_dict(50, seed=3, width_div=2)
    cfg = TD.setup_model_cfg(update_model="x", device="0")
    outs = {}
    for tag, kw, dd in (("raw", {}, "all"), ("raw_host", {}, "auto"), ("raw_strips", {"rows_per_strip": 16}, "all"), ("lzw_dev", {"compression": "lzw", "tile": (128, 128), "predictor": 2}, "auto"),
                        ("lzw_host", {"compression": "lzw", "tile": (128, 128), "predictor": 2}, False),
                        ("lzw_strips_dev", {"compression": "lzw", "rows_per_strip": 3}, True),
                        ("deflate_dev", {"compression": "deflate", "tile": (128, 128), "predictor": 2}, "auto")):
        d = tmp_path / tag
        (d / "rgb").mkdir(parents=True)
        tif = str(d / "rgb" / "9.tif")
        write_geotiff(tif, rgbi, T, 25832, **kw)
        tile_single_file(tif, str(d / "tiles"), buffer=10, tile_width=40, tile_height=40)
        meta = json.load(open(d / "tiles" / "9.json"))
        # one tile whose bounds are off the pixel grid: half a pixel inwards on every side
        k0 = next(iter(meta))
        b = meta[k0]["bounds"]
        meta[k0]["bounds"] = [b[0] + 0.1, b[1] + 0.1, b[2] - 0.1, b[3] - 0.1] + list(b[4:])
        json.dump(meta, open(d / "tiles" / "9.json", "w"))
        with TD.Predictor(cfg, device_type="0", max_batch_size=3, output_dir=str(d / "out"), state_dict=sd, device_decode=dd) as pred:
            for _ in range(2):                                      # twice: the second image is prefetched by the first's walk
                pred.prefetch(tif)
                pred(tif, str(d / "tiles" / "9.json"))
            assert pred.decode_stats["images"] == (2 if tag in ("lzw_dev", "lzw_strips_dev", "deflate_dev") else 0), (tag, pred.decode_stats)
            # device_decode "all": an uncompressed raster is kept whole in HBM as well (uploaded in 4-MB pieces); "auto" leaves it to the host reader
            assert pred.upload_stats["images"] == (2 if tag in ("raw", "raw_strips") else 0), (tag, pred.upload_stats)
        files = sorted(os.listdir(d / "out" / "9"))
        outs[tag] = {f: open(d / "out" / "9" / f, "rb").read().replace(tif.encode(), b"IMG") for f in files}
        assert len(files) == 9
    assert outs["raw"] == outs["raw_host"] == outs["raw_strips"] == outs["lzw_dev"] == outs["lzw_host"] == outs["lzw_strips_dev"] == outs["deflate_dev"]
    assert sum(len(json.loads(v)) for v in outs["raw"].values()) > 5


def test_streams_longer_than_their_block_are_reported_and_stay_inside_it(ring):
    """A block whose stream decodes to more bytes than the block holds (a corrupt byte count, a wrong tile size) ends with status 2; nothing
    is written or read past the block's capacity: the bytes behind it keep their pattern (LZW and DEFLATE, flat / noisy / repeating data)."""
    import zlib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    raws = [bytes(60000), rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(), (bytes(range(251)) * 300)[:70000], rng.integers(0, 4, 90000, dtype=np.uint8).tobytes()]
    cap = 5000
    for codec in ("lzw", "deflate"):
        streams = []
        for raw in raws:
            if codec == "deflate":
                streams.append(zlib.compress(raw, 6))
            else:
                src = np.frombuffer(raw, np.uint8)
                enc = np.empty(len(raw) * 2 + 64, np.uint8)
                n = lib.td_tiff_lzw_encode(src.ctypes.data, src.size, enc.ctypes.data, enc.size)
                streams.append(enc[:n].tobytes())
        offs, blob = [], bytearray()
        for st in streams:
            offs.append(len(blob))
            blob += st
        blob += b"\0" * 16
        comp = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).cuda()
        d_off = torch.tensor(offs, dtype=torch.int64, device="cuda")
        d_n = torch.tensor([len(st) for st in streams], dtype=torch.int64, device="cuda")
        out = torch.full((len(streams) + 1, cap), 0xA5, dtype=torch.uint8, device="cuda")          # one block of guard bytes behind the last
        dec = torch.zeros((len(streams),), dtype=torch.int64, device="cuda")
        status = torch.full((2 * len(streams) + 1,), -1, dtype=torch.int32, device="cuda")
        fn = lib.td_tiff_lzw_decode_dev if codec == "lzw" else lib.td_tiff_inflate_dev
        _lib.check(fn(comp.data_ptr(), d_off.data_ptr(), d_n.data_ptr(), len(streams), out.data_ptr(), cap, dec.data_ptr(), status.data_ptr(),
                      _lib.stream_ptr()), codec)
        torch.cuda.synchronize()
        assert status[:len(streams)].tolist() == [2] * len(streams), (codec, status.tolist())
        got = out.cpu().numpy()
        for k, raw in enumerate(raws):
            assert (int(dec[k]) & 0xffffffff) == len(raw), (codec, k)
            assert got[k].tobytes() == raw[:cap], (codec, k)                                    # the part that fits is right
        assert (got[-1] == 0xA5).all(), codec
