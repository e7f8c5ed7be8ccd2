// Strip / tile decompressors for the GeoTIFF tile reader (the reference reads rasters through rasterio → GDAL →
// libtiff: TreeDetection/prediction.py:61 `rasterio.open`, 164 `rasterio.mask.mask`; none of them is installed here).
// Real orthophoto mosaics are almost always stored compressed; DEFLATE goes through Python's zlib, the two codecs
// below cover what zlib cannot. Both follow the TIFF 6.0 specification (sections 13 "LZW" and 9 "PackBits").
// Pure host code; called from the reader's decode threads with the GIL released.
#include "common.h"

#include <cstring>

// TIFF LZW: MSB-first variable-width codes (9..12 bits), ClearCode 256, EndOfInformation 257, first free code 258,
// code width grows one code EARLY (when the next free code reaches 2^width - 1).
extern "C" int64_t td_tiff_lzw_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    if (!src || !dst || n < 0 || cap < 0) {
        td_set_error("td_tiff_lzw_decode: bad argument");
        return TD_ERR_INVALID;
    }
    constexpr int CLEAR = 256, EOI = 257, FIRST = 258, MAXC = 4096;
    static thread_local uint16_t prefix[MAXC];
    static thread_local uint8_t suffix[MAXC], first[MAXC];
    static thread_local uint16_t length[MAXC];
    for (int i = 0; i < 256; ++i) {
        prefix[i] = 0;
        suffix[i] = first[i] = (uint8_t)i;
        length[i] = 1;
    }
    int nbits = 9, next = FIRST, old = -1;
    uint64_t acc = 0;
    int have = 0;
    int64_t ip = 0, op = 0;
    for (;;) {
        while (have < nbits && ip < n) {
            acc = (acc << 8) | src[ip++];
            have += 8;
        }
        if (have < nbits) break;                       // ran out of input without an EOI: accept what was decoded
        const int code = (int)((acc >> (have - nbits)) & ((1u << nbits) - 1));
        have -= nbits;
        if (code == EOI) break;
        if (code == CLEAR) {
            nbits = 9;
            next = FIRST;
            old = -1;
            continue;
        }
        if (old < 0) {                                 // first code after a clear: a literal
            if (code > 255) {
                td_set_error("td_tiff_lzw_decode: corrupt stream (code %d after clear)", code);
                return TD_ERR_INVALID;
            }
            if (op < cap) dst[op] = (uint8_t)code;
            ++op;
            old = code;
            continue;
        }
        if (code > next || (code == next && next >= MAXC)) {
            td_set_error("td_tiff_lzw_decode: corrupt stream (code %d, table size %d)", code, next);
            return TD_ERR_INVALID;
        }
        if (next < MAXC) {                             // new entry = string(old) + first char of string(code | old)
            prefix[next] = (uint16_t)old;
            first[next] = first[old];
            length[next] = (uint16_t)(length[old] + 1);
            suffix[next] = code < next ? first[code] : first[old];
            ++next;
            if (next > (1 << nbits) - 2 && nbits < 12) ++nbits;
        }
        // write string(code) back to front
        const int len = length[code];
        if (op + len <= cap) {
            uint8_t* w = dst + op + len;
            int c = code;
            for (int k = 0; k < len; ++k) {
                *--w = suffix[c];
                c = prefix[c];
            }
        }
        op += len;
        old = code;
    }
    if (op > cap) {
        td_set_error("td_tiff_lzw_decode: %lld bytes decoded, capacity %lld", (long long)op, (long long)cap);
        return TD_ERR_CAPACITY;
    }
    return op;
}

// TIFF LZW encoder (the writer's side of the codec above: test rasters and the bench's LZW fixture; libtiff's bit layout —
// MSB-first codes, ClearCode first, the width grows one code early, a ClearCode when the table is full, EOI last). The string
// table is an open-addressing hash of (prefix code, next byte) → code. Returns the compressed size or TD_ERR_CAPACITY.
extern "C" int64_t td_tiff_lzw_encode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    if ((!src && n > 0) || !dst || n < 0 || cap < 0) {
        td_set_error("td_tiff_lzw_encode: bad argument");
        return TD_ERR_INVALID;
    }
    constexpr int CLEAR = 256, EOI = 257, FIRST = 258, MAXC = 4096, HSIZE = 1 << 14;
    static thread_local int32_t hkey[HSIZE];
    static thread_local uint16_t hval[HSIZE];
    uint64_t acc = 0;
    int have = 0, nbits = 9, next = FIRST;
    int64_t op = 0;
    bool full = false;
    auto put = [&](int code) {
        acc = (acc << nbits) | (uint64_t)code;
        have += nbits;
        while (have >= 8) {
            if (op < cap) dst[op] = (uint8_t)(acc >> (have - 8));
            else full = true;
            ++op;
            have -= 8;
        }
    };
    auto reset = [&] {
        for (int i = 0; i < HSIZE; ++i) hkey[i] = -1;
        nbits = 9;
        next = FIRST;
    };
    reset();
    put(CLEAR);
    if (n > 0) {
        int cur = src[0];
        for (int64_t i = 1; i < n; ++i) {
            const int c = src[i];
            const int32_t key = (cur << 8) | c;
            uint32_t h = ((uint32_t)key * 2654435761u) >> 18;
            int found = -1;
            while (hkey[h] != -1) {
                if (hkey[h] == key) {
                    found = hval[h];
                    break;
                }
                h = (h + 1) & (HSIZE - 1);
            }
            if (found >= 0) {
                cur = found;
                continue;
            }
            put(cur);
            hkey[h] = key;
            hval[h] = (uint16_t)next;
            ++next;
            // the decoder adds its entry one code later than the encoder and widens at 2^w - 2 entries of ITS table: the
            // encoder therefore widens when its own next free code reaches 2^w - 1 (libtiff's CODE_MAX test)
            if (next > (1 << nbits) - 1 && nbits < 12) ++nbits;
            if (next >= MAXC - 1) {                       // table full (libtiff clears at 4094 entries)
                put(CLEAR);
                reset();
            }
            cur = c;
        }
        put(cur);
        // the decoder has added one more entry for the last code: it may have widened
        if (next + 1 > (1 << nbits) - 1 && nbits < 12) ++nbits;
    }
    put(EOI);
    if (have > 0) {
        if (op < cap) dst[op] = (uint8_t)(acc << (8 - have));
        else full = true;
        ++op;
    }
    if (full || op > cap) {
        td_set_error("td_tiff_lzw_encode: %lld bytes needed, capacity %lld", (long long)op, (long long)cap);
        return TD_ERR_CAPACITY;
    }
    return op;
}

// DEFLATE in a zlib wrapper (TIFF compression 8 / 32946): the decoder the GPU runs (inflate_core.h, one wave per block), instantiated
// for ONE lane — the same source on the host, so its parity with zlib is tested without a GPU (tests/test_geotiff_formats.py). The
// tile reader itself keeps Python's zlib for DEFLATE blocks (it also checks the Adler-32 trailer, which this decoder skips).
#include "inflate_core.h"
extern "C" int64_t td_tiff_inflate(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    if (!src || !dst || n < 0 || cap < 0 || cap >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 28)) {
        td_set_error("td_tiff_inflate: bad argument");
        return TD_ERR_INVALID;
    }
    static thread_local InflateScratchT<8192> scratch;       // (the ring the device uses on large rasters: its far-match path runs here too)
    // the core reads whole dwords from the 4-byte-aligned address below src up to the dword that holds the stream's last byte: copy
    // the stream into a padded, aligned buffer so that a caller's tight buffer is never over-read
    std::vector<uint32_t> padded((size_t)(n + 8) / 4 + 2, 0u);
    std::memcpy(padded.data(), src, (size_t)n);
    const InflateResult r = inflate_block<1>(scratch, reinterpret_cast<const uint8_t*>(padded.data()), n, dst, (uint32_t)cap, 0);
    if (r.status == 1) {
        td_set_error("td_tiff_inflate: corrupt stream");
        return TD_ERR_INVALID;
    }
    if (r.status == 2) {
        td_set_error("td_tiff_inflate: %lld bytes decoded, capacity %lld", (long long)r.produced, (long long)cap);
        return TD_ERR_CAPACITY;
    }
    return (int64_t)r.produced;
}

// PackBits (TIFF 6.0 section 9): header byte h: 0..127 → copy h+1 literal bytes; -127..-1 → repeat the next byte
// 1-h times; -128 → no operation.
extern "C" int64_t td_tiff_packbits_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    if (!src || !dst || n < 0 || cap < 0) {
        td_set_error("td_tiff_packbits_decode: bad argument");
        return TD_ERR_INVALID;
    }
    int64_t ip = 0, op = 0;
    while (ip < n) {
        const int h = (int8_t)src[ip++];
        if (h >= 0) {
            const int64_t cnt = h + 1;
            if (ip + cnt > n) {
                td_set_error("td_tiff_packbits_decode: truncated literal run");
                return TD_ERR_INVALID;
            }
            if (op + cnt <= cap) std::memcpy(dst + op, src + ip, (size_t)cnt);
            ip += cnt;
            op += cnt;
        } else if (h != -128) {
            const int64_t cnt = 1 - h;
            if (ip >= n) {
                td_set_error("td_tiff_packbits_decode: truncated repeat run");
                return TD_ERR_INVALID;
            }
            if (op + cnt <= cap) std::memset(dst + op, src[ip], (size_t)cnt);
            ++ip;
            op += cnt;
        }
    }
    if (op > cap) {
        td_set_error("td_tiff_packbits_decode: %lld bytes decoded, capacity %lld", (long long)op, (long long)cap);
        return TD_ERR_CAPACITY;
    }
    return op;
}

// Predictor 2 (TIFF 6.0 section 14, horizontal differencing): each sample was stored as the difference to the same
// sample of the pixel on its left, modulo the sample width; undo it in place, row by row.
extern "C" int td_tiff_unpredict(void* data, int64_t rows, int64_t cols, int samples, int bytes_per_sample) {
    if (!data || rows < 0 || cols < 0 || samples < 1 || (bytes_per_sample != 1 && bytes_per_sample != 2 && bytes_per_sample != 4)) {
        td_set_error("td_tiff_unpredict: bad argument");
        return TD_ERR_INVALID;
    }
    const int64_t row_elems = cols * samples;
    for (int64_t r = 0; r < rows; ++r) {
        if (bytes_per_sample == 1) {
            uint8_t* p = static_cast<uint8_t*>(data) + r * row_elems;
            for (int64_t i = samples; i < row_elems; ++i) p[i] = (uint8_t)(p[i] + p[i - samples]);
        } else if (bytes_per_sample == 2) {
            uint16_t* p = static_cast<uint16_t*>(data) + r * row_elems;
            for (int64_t i = samples; i < row_elems; ++i) p[i] = (uint16_t)(p[i] + p[i - samples]);
        } else {
            uint32_t* p = static_cast<uint32_t*>(data) + r * row_elems;
            for (int64_t i = samples; i < row_elems; ++i) p[i] = p[i] + p[i - samples];
        }
    }
    return TD_OK;
}

// Window of an uncompressed, pixel-interleaved raster whose strips lie back to back (include/treedet.h): `rows` pieces of
// `row_bytes` bytes, `row_stride` apart in the file, read with pread into a dense destination. A memory map of the same
// file pays a page fault per window row (a 1000-px row of a 12 000-px raster touches a page of its own: ~1.2 us against
// ~0.1 us for the 4 KB copy) and its munmap tears all those entries down again at close; pread copies straight from the
// page cache. Returns the bytes read, TD_ERR_INVALID when the file ends early or a read fails.
#include <cerrno>
#include <unistd.h>

extern "C" int64_t td_read_window(int fd, int64_t file_off, int64_t row_stride, int64_t row_bytes, int64_t rows, uint8_t* dst) {
    if (fd < 0 || file_off < 0 || row_bytes < 0 || rows < 0 || row_stride < row_bytes || (rows > 0 && row_bytes > 0 && !dst)) {
        td_set_error("td_read_window: bad argument");
        return TD_ERR_INVALID;
    }
    for (int64_t r = 0; r < rows; ++r) {
        uint8_t* d = dst + r * row_bytes;
        int64_t got = 0;
        while (got < row_bytes) {
            const ssize_t n = pread(fd, d + got, (size_t)(row_bytes - got), (off_t)(file_off + r * row_stride + got));
            if (n < 0 && errno == EINTR) continue;
            if (n <= 0) {
                td_set_error("td_read_window: row %lld: %s", (long long)r, n == 0 ? "file ends inside the window" : strerror(errno));
                return TD_ERR_INVALID;
            }
            got += n;
        }
    }
    return rows * row_bytes;
}

// The windows of a whole batch in one call (round 6): window i = rows[i] pieces of row_bytes[i] bytes, row_stride apart from file_off[i],
// into dst + dst_off[i] — spread over `threads` threads (each window is cut into row bands so that a batch of few, tall windows
// still fills them). The files-to-files path is bound by the Python work per tile at the fp16 rate; one call per batch replaces
// eight Python tasks on a thread pool. Returns the bytes read or TD_ERR_INVALID.
#include <atomic>
#include <thread>
extern "C" int64_t td_read_windows(int fd, int n, const int64_t* file_off, int64_t row_stride, const int64_t* row_bytes, const int64_t* rows,
                                   uint8_t* dst, const int64_t* dst_off, int threads) {
    if (fd < 0 || n < 0 || (n > 0 && (!file_off || !row_bytes || !rows || !dst || !dst_off)) || row_stride < 0) {
        td_set_error("td_read_windows: bad argument");
        return TD_ERR_INVALID;
    }
    struct Band { int w; int64_t r0, r1; };
    std::vector<Band> bands;
    const int64_t band_rows = 128;
    for (int i = 0; i < n; ++i) {
        if (file_off[i] < 0 || row_bytes[i] < 0 || rows[i] < 0 || row_stride < row_bytes[i] || dst_off[i] < 0) {
            td_set_error("td_read_windows: window %d is malformed", i);
            return TD_ERR_INVALID;
        }
        for (int64_t r = 0; r < rows[i]; r += band_rows) bands.push_back({i, r, r + band_rows < rows[i] ? r + band_rows : rows[i]});
    }
    std::atomic<size_t> next{0};
    std::atomic<int64_t> total{0};
    std::atomic<int> failed{0};
    auto work = [&] {
        for (size_t k = next.fetch_add(1); k < bands.size(); k = next.fetch_add(1)) {
            const Band& b = bands[k];
            const int64_t got = td_read_window(fd, file_off[b.w] + b.r0 * row_stride, row_stride, row_bytes[b.w], b.r1 - b.r0,
                                               dst + dst_off[b.w] + b.r0 * row_bytes[b.w]);
            if (got < 0) failed.store(1);
            else total.fetch_add(got);
        }
    };
    const int nt = threads < 1 ? 1 : (threads > (int)bands.size() ? (bands.empty() ? 1 : (int)bands.size()) : threads);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed.load()) return TD_ERR_INVALID;          // td_read_window has set the message
    return total.load();
}
