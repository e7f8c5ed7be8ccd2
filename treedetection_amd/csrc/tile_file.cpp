// One tile's epilogue from device rows to the file (include/treedet.h: td_tile_prediction_file; reference
// TreeDetection/prediction.py:197-265 — `_process_and_save_single` moves every mask to the host, traces it and writes the
// tile's JSON). The engine leaves each image's pasted masks as packed bit rows in a buffer sized for the worst case
// (100 detections x the whole tile = 12.8 MB at 1000 x 1000); what the paste actually wrote is known from the image's
// region / offset records, so this host thread copies only those words over PCIe before tracing — a fixed-size copy of
// the whole buffer (102 MB per 8-tile batch) was exactly what one Gen5 x16 link moves in the 4 ms a fp16 batch takes.
#include "common.h"

#include <atomic>
#include <cstdio>
#include <mutex>
#include <thread>

namespace {

constexpr int kCopyStreams = 4;
constexpr int kMaxDevices = 16;

struct CopyStreams {
    std::once_flag once;
    hipStream_t s[kCopyStreams] = {};
    hipError_t err = hipSuccess;
    std::atomic<unsigned> next{0};
};
CopyStreams g_copy[kMaxDevices];

}  // namespace

extern "C" int td_tile_prediction_file(int device, const int32_t* mask_region, const int64_t* mask_offset,
                                       const uint32_t* mask_bits_dev, uint32_t* rows_host, int64_t mask_words, const float* scores,
                                       const int32_t* classes, int n, const double* transform, const char* image_id,
                                       const char* path, int64_t* bytes_written) {
    if (n < 0 || device < 0 || device >= kMaxDevices || !transform || !image_id || !path || !bytes_written ||
        (n > 0 && (!mask_region || !mask_offset || !mask_bits_dev || !rows_host || !scores)) || mask_words < 0) {
        td_set_error("td_tile_prediction_file: bad argument");
        return TD_ERR_INVALID;
    }
    // the words the paste wrote: regions are laid out one after the other (paste_plan_kernel), empty ones take none
    int64_t used = 0;
    for (int k = 0; k < n; ++k) {
        const int64_t w = (int64_t)mask_region[4 * k + 2] - mask_region[4 * k], h = (int64_t)mask_region[4 * k + 3] - mask_region[4 * k + 1];
        if (w <= 0 || h <= 0) continue;
        if (w > (1 << 20) || h > (1 << 20) || mask_offset[k] < 0) {
            td_set_error("td_tile_prediction_file: detection %d has an impossible paste record", k);
            return TD_ERR_INVALID;
        }
        const int64_t end = mask_offset[k] + ((w + 31) / 32) * h;
        if (end > used) used = end;
    }
    if (used > mask_words) {
        td_set_error("td_tile_prediction_file: the records cover %lld words, the buffer holds %lld", (long long)used,
                     (long long)mask_words);
        return TD_ERR_INVALID;
    }
    if (used > 0) {
        TD_HIP_CHECK(hipSetDevice(device));
        CopyStreams& cs = g_copy[device];
        std::call_once(cs.once, [&] {
            for (int i = 0; i < kCopyStreams && cs.err == hipSuccess; ++i)
                cs.err = hipStreamCreateWithFlags(&cs.s[i], hipStreamNonBlocking);
        });
        TD_HIP_CHECK(cs.err);
        hipStream_t st = cs.s[cs.next.fetch_add(1, std::memory_order_relaxed) % kCopyStreams];
        TD_HIP_CHECK(hipMemcpyAsync(rows_host, mask_bits_dev, (size_t)used * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        // (an event-query loop with 50-us naps instead of this call was measured in round 6: the epilogue workers' CPU time fell, but every
        // image took 5 batch periods longer — queries do not push the runtime's queued copies out the way a synchronisation does)
        TD_HIP_CHECK(hipStreamSynchronize(st));
    }
    std::string text;
    const int entries = td_polygons_json_text(mask_region, mask_offset, rows_host, used, scores, classes, n, transform, image_id, text,
                                              "td_tile_prediction_file");
    if (entries < 0) return entries;
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        td_set_error("td_tile_prediction_file: cannot open %s for writing", path);
        return TD_ERR_INVALID;
    }
    const size_t wrote = std::fwrite(text.data(), 1, text.size(), f);
    const int closed = std::fclose(f);
    if (wrote != text.size() || closed != 0) {
        td_set_error("td_tile_prediction_file: short write to %s", path);
        return TD_ERR_INVALID;
    }
    *bytes_written = (int64_t)text.size();
    return entries;
}

// A whole batch's tile files in ONE call (round 6): the files-to-files path is bound, at the fp16 rate, by the Python work per tile —
// every worker thread's Python runs under one interpreter lock — so the per-tile call above costs the rate more than its C part does.
// Tile i of the batch: records at mask_region + i * D * 4, mask_offset + i * D, scores / classes + i * D, count[i] detections,
// device rows at mask_bits_dev + i * bits_stride_words, pinned host rows at rows_host + i * bits_stride_words, transform at
// transforms + 6 * i, file paths[i]. status[i] receives the entry count or the negative status of tile i (a tile with count < 0 is
// skipped: status 0, no file); the tiles are spread over `threads` threads. Returns TD_OK, or the first failing tile's status.
extern "C" int td_batch_prediction_files(int device, int n_tiles, int dets_per_image, const int32_t* mask_region, const int64_t* mask_offset,
                                         const uint32_t* mask_bits_dev, uint32_t* rows_host, int64_t bits_stride_words, const float* scores,
                                         const int32_t* classes, const int32_t* counts, const double* transforms, const char* image_id,
                                         const char* const* paths, int threads, int32_t* status, int64_t* bytes_written) {
    if (n_tiles < 0 || dets_per_image < 1 || !counts || !transforms || !image_id || !paths || !status || !bytes_written ||
        (n_tiles > 0 && (!mask_region || !mask_offset || !mask_bits_dev || !rows_host || !scores || !classes)) || bits_stride_words < 0) {
        td_set_error("td_batch_prediction_files: bad argument");
        return TD_ERR_INVALID;
    }
    std::atomic<int> next{0};
    auto work = [&] {
        for (int i = next.fetch_add(1); i < n_tiles; i = next.fetch_add(1)) {
            bytes_written[i] = 0;
            if (counts[i] < 0) {
                status[i] = 0;
                continue;
            }
            const size_t D = (size_t)dets_per_image;
            status[i] = td_tile_prediction_file(device, mask_region + i * D * 4, mask_offset + i * D, mask_bits_dev + (size_t)i * bits_stride_words,
                                                rows_host + (size_t)i * bits_stride_words, bits_stride_words, scores + i * D, classes + i * D,
                                                counts[i] > dets_per_image ? dets_per_image : counts[i], transforms + 6 * (size_t)i, image_id, paths[i],
                                                bytes_written + i);
        }
    };
    const int nt = threads < 1 ? 1 : (threads > n_tiles ? (n_tiles > 0 ? n_tiles : 1) : threads);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    for (int i = 0; i < n_tiles; ++i)
        if (status[i] < 0) return status[i];
    return TD_OK;
}
