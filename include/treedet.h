/*
 * treedet.h — C ABI of libtreedet_hip.so, the MI355X (gfx950) tile-inference engine that
 * stands behind the reference's model seam:
 *
 *     batch_predictions = self.model(batch_tensors)      TreeDetection/prediction.py:182-183
 *
 * (a Python call into detectron2's GeneralizedRCNN configured by TreeDetection/config.py:25-66;
 * the reference itself has no FFI — this header is the FFI a maintainer would bind instead, see
 * INTEGRATION.md). Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns td_status: 0 = ok, < 0 = error; the message is td_last_error().
 *   - "dev" pointers are device (HBM) addresses on the engine's GPU; "host" pointers are host.
 *   - activations are NHWC float32 (or float16 when precision = 1) inside the engine.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - the engine is NOT thread-safe; one engine per device and stream.
 */
#ifndef TREEDET_H
#define TREEDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int td_status;
typedef struct td_engine td_engine;

enum {
    TD_OK = 0,
    TD_ERR_INVALID = -1,   /* bad argument / shape */
    TD_ERR_HIP = -2,       /* a HIP runtime call failed */
    TD_ERR_WEIGHTS = -3,   /* missing or mis-shaped tensor in load_weights */
    TD_ERR_CAPACITY = -4,  /* forward() exceeds what reserve() sized */
    TD_ERR_STATE = -5      /* call order (forward before load/reserve) */
};

enum { TD_PRECISION_FP32 = 0, TD_PRECISION_FP16 = 1 };
enum { TD_INPUT_F32_CHW = 0,  /* float32 [B,3,Hp,Wp], BGR 0..255 — what prediction.py:170 builds   */
       TD_INPUT_U8_HWC = 1 }; /* uint8   [B,Hp,Wp,3],  BGR        — the same values before astype() */

/* Model hyper-parameters. Defaults (td_model_desc_default) = detectron2 defaults + the overrides
 * of TreeDetection/config.py:35,59-61 (NUM_CLASSES=1, SCORE_THRESH_TEST=0.3, NMS_THRESH_TEST=0.5). */
typedef struct td_model_desc {
    int32_t num_classes;           /* 1 */
    int32_t precision;             /* TD_PRECISION_* */
    int32_t pre_nms_topk;          /* 1000 per FPN level */
    int32_t post_nms_topk;         /* 1000 per image */
    int32_t detections_per_image;  /* 100 */
    float rpn_nms_thresh;          /* 0.7 */
    float score_thresh;            /* 0.3 (strict >) */
    float nms_thresh;              /* 0.5 */
    float mask_thresh;             /* 0.5 (>=) */
} td_model_desc;

/* One named fp32 host tensor; names are detectron2 state-dict keys
 * (e.g. "backbone.bottom_up.res2.0.conv1.norm.running_var"). */
typedef struct td_tensor_desc {
    const char* name;
    const float* data;   /* host, contiguous, torch layout ([Cout,Cin,kh,kw] / [out,in]) */
    int32_t ndim;
    int64_t shape[4];
} td_tensor_desc;

#define TD_MASK_SIDE 28

/* Caller-allocated device output of one forward(). D = detections_per_image.
 * Fields may be NULL to skip that output (mask_bits == NULL skips the paste). */
typedef struct td_detections {
    float* boxes;          /* dev [B, D, 4]  x1,y1,x2,y2 in OUTPUT (pre-resize tile) pixels, score-descending */
    float* scores;         /* dev [B, D] */
    int32_t* classes;      /* dev [B, D] */
    int32_t* count;        /* dev [B]    number of valid rows */
    float* mask_probs;     /* dev [B, D, 28, 28] sigmoid probabilities */
    int32_t* mask_region;  /* dev [B, D, 4] x0,y0,x1,y1: the integer region the CPU paste evaluates */
    int64_t* mask_offset;  /* dev [B, D]   offset (in 32-bit words) of detection's bit rows from mask_bits of image b */
    uint32_t* mask_bits;   /* dev [B, mask_words_per_image]; row r of a region starts at offset + r*ceil((x1-x0)/32);
                              bit (x-x0)&31 of word (x-x0)>>5 = pasted mask >= mask_thresh */
    int64_t mask_words_per_image;
} td_detections;

/* ---- engine life cycle ------------------------------------------------------------------- */
void td_model_desc_default(td_model_desc* desc);
td_status td_engine_create(const td_model_desc* desc, int device, td_engine** out);
td_status td_engine_load_weights(td_engine* e, const td_tensor_desc* tensors, size_t n);
/* Size the workspace for forward() calls of up to max_batch images of padded size max_hp x max_wp
 * (multiples of 32). May be called again to grow. */
td_status td_engine_reserve(td_engine* e, int max_batch, int max_hp, int max_wp);
/* images: dev, format per `input_format`, B images padded to Hp x Wp (multiples of 32);
 * hw_valid: host int32 [B,2] un-padded (H', W') of each image (detectron2 image_sizes);
 * hw_out:   host int32 [B,2] output (height, width) = the tile's pre-resize size (prediction.py:168,182).
 * Asynchronous on `stream`; results are complete once the stream has drained. */
td_status td_engine_forward(td_engine* e, const void* images, int input_format, const int32_t* hw_valid,
                            const int32_t* hw_out, int B, int Hp, int Wp, void* stream, td_detections* out);
/* The same forward cut into six phases for cross-batch software pipelining (three engines, a main stream and one
 * side stream per engine):
 *   0 trunk (stem .. RPN heads)   1 RPN top-k / NMS / merge + RoIAlign 7x7   2 box-head FCs + predictors
 *   3 detections + RoIAlign 14x14 4 mask-head convolutions                   5 mask predictor, scatter, paste
 * Even phases are dense contractions, odd phases low-occupancy selection work: enqueue the even phases of successive
 * batches back to back on a main stream and each batch's odd phases on its own side stream, and the selection work of
 * one batch overlaps the contractions of the next (bench.py: per tick trunk(t), mask convs(t-2), FCs(t-1)). Phase 0 takes the arguments of td_engine_forward and stores them; phases 1-5
 * ignore everything but `e`, `phase` and `stream`. Each phase waits (hipStreamWaitEvent) for the previous phase of the
 * same batch; phase 0 waits for phases 0-4 of the engine's previous batch (the trunk rewrites what they read, the
 * Winograd workspace of the fp32 mask-head convolutions included) and phase 3 for the previous batch's phase 5 (whose
 * predictor / paste read the mask buffers and row count that phases 3-4 rewrite), so any stream assignment is correct.
 * Optional pre-phase TD_PHASE_STEM (6): stem convolution + max-pool of the NEXT batch (VALU / HBM work, no matrix
 * cores) with the arguments of phase 0; the following phase 0 (images may be NULL) then starts at res2. Run it on a side
 * stream while the previous batch's contractions hold the main stream. */
#define TD_PHASE_STEM 6
td_status td_engine_forward_phase(td_engine* e, int phase, const void* images, int input_format, const int32_t* hw_valid,
                                  const int32_t* hw_out, int B, int Hp, int Wp, void* stream, td_detections* out);
/* Expose an internal activation of the last forward() for stage-wise parity tests: names
 * "stem","pool","res2".."res5","p2".."p6","rpn_logits","rpn_deltas","proposals","proposal_scores",
 * "proposal_count","pooled7","cls_logits","box_deltas","det_boxes_net","pooled14","mask_logits".
 * dims is filled with up to 4 extents (0-padded); *elem_size with the element size in bytes. */
td_status td_engine_tensor(td_engine* e, const char* name, void** dev_ptr, int64_t dims[4], int* elem_size);
/* Copy that activation into a caller-owned device buffer of `bytes` bytes (asynchronous on `stream`). */
td_status td_engine_read_tensor(td_engine* e, const char* name, void* dst_dev, int64_t bytes, void* stream);
/* Per-category device timing of forward(): when enabled, every kernel launch is bracketed by HIP events on the
 * forward's stream. Categories: 0 conv_igemm (all MFMA contractions), 1 stem, 2 pool/subsample, 3 rpn_select
 * (top-k, decode, NMS, merge), 4 roi_align, 5 detect (decode/sort/NMS/finalize), 6 mask_tail (predictor, scatter,
 * paste), 7 mask-head contractions (row count lives on the device: flops/bytes are left 0 for the caller to fill),
 * 8 "executed": no time of its own — flops[8] = multiply-adds x 2 the category-0 launches really issued (the Winograd
 * path of the fp32 engine runs 4/9 of a 3x3 layer's algorithmic FLOPs; its two transform kernels are timed inside
 * category 0, the conv family), launches[8] = layers that took the Winograd path. td_engine_profile_read synchronises the recorded events and ACCUMULATES since the last reset:
 * ms[c] device milliseconds, launches[c], flops[c] algorithmic FLOPs (2*M*N*K, conv only), bytes[c] algorithmic
 * HBM bytes (inputs read once + outputs written once). Arrays hold TD_PROF_CATEGORIES entries. */
#define TD_PROF_CATEGORIES 9
/* enable: 0 off, 1 one event pair per category group, 2 "detail": additionally one event pair per launch of the
 * contraction family for td_engine_profile_classes (perturbs the forward: use it on a region of its own). */
td_status td_engine_profile_enable(td_engine* e, int enable);
td_status td_engine_profile_read(td_engine* e, double* ms, int64_t* launches, double* flops, double* bytes, int reset);
/* Speed-of-light accounting of the contraction family by class. Per class since the last reset: launches; the FLOPs the
 * launches really EXECUTE (a Winograd layer's plane products, not the direct convolution's multiplies); the bytes the
 * chosen algorithm moves through HBM when every tensor is read / written once (V / M planes of the Winograd layers
 * included); tmin_ms = sum over launches of max(executed FLOPs / MFMA peak of the engine's precision, bytes /
 * achievable HBM rate) with the constants below (MI355X_MICROARCH.md); ms = measured time (detail mode only, else 0).
 * Class 5 (mask head) has a device-side row count: only its time is reported. Arrays hold TD_PROF_CLASSES entries. */
#define TD_PROF_CLASSES 7
#define TD_CLS_WINO_GEMM 0   /* Winograd plane contractions (fp32 engine) */
#define TD_CLS_WINO_XFORM 1  /* Winograd input / output transform kernels */
#define TD_CLS_CONV1X1 2     /* 1x1 convolutions: bottleneck conv1 / conv3 / shortcut, FPN laterals, RPN heads */
#define TD_CLS_CONV3X3 3     /* direct 3x3 convolutions (fp16 engine: all of them; fp32: the 64-channel res2 layers) */
#define TD_CLS_FC 4          /* box head: fc1, fc2, predictors */
#define TD_CLS_MASK_HEAD 5   /* mask-head contractions and transforms (rows = live detections, known on the device only) */
#define TD_CLS_TAIL 6        /* fused bottleneck tail: 3x3 (mid -> mid) + 1x1 (mid -> 4 mid) + shortcut in one launch (res2; fp16: res3 too) */
#define TD_PEAK_F32_MFMA_TFLOPS 157.3   /* v_mfma_f32_32x32x2_f32, dense */
#define TD_PEAK_F16_MFMA_TFLOPS 2500.0  /* v_mfma_f32_32x32x16_f16, dense */
#define TD_HBM_ACHIEVABLE_TBS 6.3       /* measured streaming rate (8.0 TB/s spec) */
td_status td_engine_profile_classes(td_engine* e, double* ms, int64_t* launches, double* exec_flops, double* bytes, double* tmin_ms, int reset);
const char* td_last_error(void);
void td_engine_destroy(td_engine* e);

/* ---- tile preprocessing (reference Predictor._process_tile, prediction.py:159-176) ---------- */
/* Pillow-exact 8-bit bilinear resize (two fixed-point passes) of one tile, fused with the
 * band pick (2,1,0)=BGR of prediction.py:166: src dev uint8 [h, w, C] (pixel-interleaved, C >= 3),
 * dst dev uint8 [dst_pitch_rows.., 3] written at dst[(y*dst_pitch_px + x)*3 + c] for y<out_h, x<out_w. */
td_status td_resize_tile_u8(const uint8_t* src, int h, int w, int c, uint8_t* dst, int out_h, int out_w,
                            int dst_pitch_px, void* tmp_dev /* >= h*out_w*3 bytes */, void* stream);
/* The same for n tiles of one common size in two launches: src_tiles = HOST array of n device pointers; tile i is
 * written to dst + i*dst_image_stride_bytes; tmp_dev >= n*h*out_w*3 bytes. */
td_status td_resize_batch_u8(const uint8_t* const* src_tiles, int n, int h, int w, int c, uint8_t* dst, int out_h,
                             int out_w, int dst_pitch_px, int64_t dst_image_stride_bytes, void* tmp_dev, void* stream);
/* The float branch of the same step (prediction.py:167-169): a tile that is not uint8 (16-bit imagery rescaled by
 * 255 * x / 65535, float rasters) goes through detectron2's ResizeTransform → torch.nn.functional.interpolate(mode="bilinear",
 * align_corners=False) and is then cast to float32. src dev float64 [c, h, w] (planar, BGR already picked); dst dev float32,
 * written at dst[ch * dst_plane_stride + y * dst_pitch_px + x] for y < out_h, x < out_w (the caller zero-fills the padding).
 * Arithmetic in float64 in torch's own order (half-pixel centres, source index clamped at 0, border neighbours clamped),
 * rounded to float32 once. */
td_status td_resize_bilinear_f64(const double* src, int c, int h, int w, float* dst, int out_h, int out_w, int dst_pitch_px,
                                 int64_t dst_plane_stride, void* stream);
/* ResizeShortestEdge(800, 1333) output shape for an h x w tile (Appendix A item 2). */
void td_resize_shape(int h, int w, int short_edge, int max_size, int* out_h, int* out_w);

/* ---- op-level entry points (parity tests; each is the kernel the engine itself launches) ---- */
/* Fused tail of a bottleneck block (detectron2 BottleneckBlock: conv2 3x3 + FrozenBN + ReLU, conv3 1x1 + FrozenBN, + shortcut,
 * ReLU) in one launch: x [B,H,W,mid], w2 [mid,3,3,mid], w3 [cout,mid], scale / bias per layer (NULL = 1 / 0),
 * shortcut and y [B,H,W,cout]; cout = 4 mid; mid = 64 (fp32, fp16) or 128 (fp16). precision: TD_PRECISION_*.
 * Bit-identical to td_conv2d_nhwc(3x3, relu) followed by td_conv2d_nhwc(1x1, residual, relu). */
td_status td_bottleneck_tail_nhwc(const void* x, const void* w2, const float* scale2, const float* bias2, const void* w3,
                                  const float* scale3, const float* bias3, const void* shortcut, void* y, int B, int H, int W,
                                  int mid, int cout, int precision, void* stream);
/* NHWC convolution: y = act(conv(x, w) * scale + bias [+ residual]).
 * x [B,H,W,Cin], w [Cout,KH,KW,Cin], scale/bias [Cout] (NULL = 1 / 0), residual [B,Ho>>rs,Wo>>rs,Cout]
 * (rs = res_shift: 1 = nearest-2x upsampled add, the FPN top-down path), y [B,Ho,Wo,Cout].
 * Cin must be a multiple of 32. precision selects float32 or float16 tensors (weights follow); bits 8..15 of
 * `precision` may carry (block-tile id + 1) to force one kernel variant (parity tests of every variant; 0 = the
 * library chooses); bit 16 with float16 tensors: y is float32 (how the engine runs the RPN / box-predictor heads). */
td_status td_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* bias,
                         const void* residual, int res_shift, void* y, int B, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int relu, int precision,
                         void* stream);
/* A 256-channel float16 convolution (stride 1, + bias, ReLU) whose output feeds ONLY a 1x1 head (reference: detectron2's
 * StandardRPNHead behind prediction.py:183 — conv3x3 + ReLU, then the objectness / anchor-delta 1x1 layers, SURVEY.md
 * Appendix A item 5): the head is contracted from the finished tile inside the same launch and only
 * head_y [B*Ho*Wo][head_n] (float32) = conv_out . head_w^T + head_b is written. head_w [head_n <= 32][256] float16.
 * bits 8..15 of `precision` must carry (tile id + 1) of a block tile that owns all 256 output channels and stages them as one float16 tile
 * (9, 10, 12, 13, 17, 23, 27). Bit-identical to td_conv2d_nhwc(relu) followed by a 1x1 td_conv2d_nhwc with float32 output. */
td_status td_conv2d_head_nhwc(const void* x, const void* w, const float* bias, const void* head_w, const float* head_b,
                              float* head_y, int B, int H, int W, int Cin, int KH, int KW, int pad, int head_n, int precision,
                              void* stream);
/* The same 3x3 / stride 1 / pad 1 convolution (float32) through the Winograd F(2x2,3x3) path the fp32 engine uses where it
 * measures faster (input transform, 16 batched plane contractions on the MFMA kernel, output transform): x [B,H,W,Cin],
 * w [Cout,3,3,Cin], y [B,H,W,Cout] = act(conv * scale + bias), all DEVICE pointers; Cin % 32 == 0, Cout % 4 == 0.
 * Synchronous (allocates its own scratch): parity tests only. */
td_status td_conv2d_winograd_nhwc(const float* x, const float* w, const float* scale, const float* bias, float* y, int B,
                                  int H, int W, int Cin, int Cout, int relu, void* stream);
/* F(4x4,3x3) form of td_conv2d_head_nhwc for float32 tensors (how the fp32 engine runs the RPN, reference StandardRPNHead behind
 * prediction.py:183): 3x3 / stride 1 / pad 1 conv to 256 channels + bias + ReLU, whose output feeds only the 1x1 head
 * head_w [head_n <= 32][256]; the head is contracted inside the output transform and only head_y [B*H*W][head_n] is written.
 * Bit-identical to td_conv2d_winograd_nhwc (TD_WINO_TILE=4) followed by a 1x1 td_conv2d_nhwc. */
td_status td_conv2d_winograd_head_nhwc(const float* x, const float* w, const float* bias, const float* head_w, const float* head_b,
                                       float* head_y, int B, int H, int W, int Cin, int head_n, void* stream);
/* Greedy NMS of n boxes (dev [n,4], scores dev [n]); keep_idx dev int32 [n] receives the kept
 * indices in descending-score order (ties: lower index first), *keep_count (dev) their number. */
td_status td_nms(const float* boxes, const float* scores, int n, float iou_thresh, int32_t* keep_idx,
                 int32_t* keep_count, void* stream);
/* RoIAlign (aligned, adaptive sampling) over one NHWC level: feat [H,W,C], rois dev [R,4],
 * out [R,pooled,pooled,C]. */
td_status td_roi_align(const void* feat, int H, int W, int C, const float* rois, int R, float spatial_scale,
                       int pooled, void* out, int precision, void* stream);
/* Paste N 28x28 probability masks into out_h x out_w at `boxes` (output units): fills
 * mask_region/mask_offset/mask_bits exactly like forward() does for one image. */
td_status td_paste_masks(const float* mask_probs, const float* boxes, int n, int out_h, int out_w,
                         float thresh, int32_t* mask_region, int64_t* mask_offset, uint32_t* mask_bits,
                         int64_t mask_words_cap, void* stream);

/* Same for a batch of images, asynchronously on `stream` (no host synchronisation): mask_probs [batch][D][28*28],
 * boxes [batch][D][4], counts [batch] — all DEVICE pointers; out_hw = host array of (height, width) per image.
 * Outputs as in td_detections: mask_region [batch][D][4], mask_offset [batch][D], mask_bits
 * [batch][mask_words_per_image]. This is how rank 0 pastes the detections gathered from the other ranks. */
td_status td_paste_masks_batch(const float* mask_probs, const float* boxes, const int32_t* counts, const int32_t* out_hw,
                               int batch, int dets_per_image, float thresh, int32_t* mask_region, int64_t* mask_offset,
                               uint32_t* mask_bits, int64_t mask_words_per_image, void* stream);

/* Border following ON THE DEVICE for the packed masks of a batch (same contours, order and points as
 * td_find_contours on each detection's region; prediction.py:232-236). Inputs: the mask_region / mask_offset /
 * mask_bits / count arrays of td_detections (device). Outputs (device, caller-allocated):
 *   points        int16 [batch][points_cap][2]   (x, y) in tile pixels
 *   image_points  int32 [batch]                  points allocated per image (may exceed points_cap: see status 4)
 *   det_info      int32 [batch][D][4]            status, contour count, first point, total points of the detection
 *   contour_info  int32 [batch][D][TD_CONTOUR_MAX][2]   per contour in RETR_TREE order: (point offset inside the
 *                                                detection's block, number of points)
 * status: 0 traced; 1 region larger than the on-chip label image; 2 more than TD_CONTOUR_MAX contours; 4 point
 * buffer full — such detections are left to td_find_contours on the host. Asynchronous on `stream`. */
#define TD_CONTOUR_MAX 256
td_status td_trace_contours_dev(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits,
                                int64_t mask_words_per_image, const int32_t* counts, int batch, int dets_per_image,
                                int16_t* points, int points_cap, int32_t* image_points, int32_t* det_info,
                                int32_t* contour_info, void* stream);

/* ---- host-side epilogue (reference prediction.py:232-245, utilities.py:182-207) ------------- */
/* Border following on a binary image (row-major uint8, non-zero = foreground) equivalent to
 * cv2.findContours(RETR_TREE / RETR_LIST ordering, CHAIN_APPROX_SIMPLE): writes contour points as
 * int32 (x,y) pairs into `points` (capacity max_points pairs) and contour start offsets into
 * `starts` (capacity max_contours+1). Returns the number of contours, or < 0 on overflow/error. */
int td_find_contours(const uint8_t* img, int h, int w, int32_t* points, int max_points, int32_t* starts,
                     int max_contours);
/* Whole host epilogue of one tile (reference prediction.py:229-261, Predictor._process_and_save_single):
 * the n detections' packed masks (mask_region / mask_offset / mask_bits of td_detections for ONE image, copied
 * to host memory; mask_words = capacity of mask_bits in 32-bit words) → border following → contours with >= 4
 * points, ring closed → pixel-corner coordinates through `transform` (rasterio order a,b,c,d,e,f; float64) →
 * the JSON array the reference writes to Prediction_<tile>.json, one entry per contour:
 *   {"image_id": <image_id>, "category_id": <class>, "score": <score>, "polygon_coords": [[[x, y], ...]]}
 * byte-identical to Python's json.dumps of that list. Writes the text (no terminator) to buf if it fits in
 * cap bytes; *needed always receives its length. Returns the number of entries, TD_ERR_CAPACITY when cap is
 * too small (call again with *needed bytes), or another negative status. Thread-safe; host code only. */
int td_tile_polygons_json(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits,
                          int64_t mask_words, const float* scores, const int32_t* classes, int n,
                          const double* transform, const char* image_id, char* buf, int64_t cap, int64_t* needed);

/* The same epilogue for a tile whose packed rows are still on the device (reference prediction.py:197-265 end to end for
 * one tile: the masks leave the GPU, become polygons and are written to Prediction_<tile>.json): mask_region / mask_offset /
 * scores / classes are the host copies of the image's small records, mask_bits_dev the DEVICE pointer of its bit rows
 * (td_detections.mask_bits + b * mask_words_per_image) and rows_host a pinned host buffer of mask_words words. Only the
 * words the paste wrote (mask_offset[n-1] + size of the last region; 0.1-5 MB against a 12.8 MB capacity at 1000 x 1000)
 * cross PCIe: copied on one of the library's own copy streams of `device`, waited for by this thread alone. Then
 * td_tile_polygons_json's text is written to `path` (created / truncated). *bytes_written receives the file size.
 * Returns the number of entries or a negative status. Thread-safe; the caller has made sure the forward that produced
 * the rows has completed (the host copies it passes were read after that event). */
int td_tile_prediction_file(int device, const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits_dev,
                            uint32_t* rows_host, int64_t mask_words, const float* scores, const int32_t* classes, int n,
                            const double* transform, const char* image_id, const char* path, int64_t* bytes_written);

/* The tile files of a whole batch in ONE call: tile i reads its records at mask_region + i * D * 4, mask_offset + i * D, scores /
 * classes + i * D (D = dets_per_image), counts[i] detections (count < 0: the tile is skipped, as a dropped tile is), its device rows
 * at mask_bits_dev + i * bits_stride_words, uses the pinned rows_host + i * bits_stride_words, the affine at transforms + 6 * i and
 * writes paths[i]; status[i] / bytes_written[i] receive td_tile_prediction_file's result per tile. The tiles are spread over
 * `threads` host threads. Returns TD_OK or the first failing tile's status (the other tiles are still written). */
int td_batch_prediction_files(int device, int n_tiles, int dets_per_image, const int32_t* mask_region, const int64_t* mask_offset,
                              const uint32_t* mask_bits_dev, uint32_t* rows_host, int64_t bits_stride_words, const float* scores,
                              const int32_t* classes, const int32_t* counts, const double* transforms, const char* image_id,
                              const char* const* paths, int threads, int32_t* status, int64_t* bytes_written);

/* ---- raster input (reference prediction.py:61,164: rasterio.open / rasterio.mask.mask → GDAL → libtiff) ---- */
/* Decompress one TIFF strip or tile: LZW (compression 5) and PackBits (32773) per TIFF 6.0; DEFLATE strips go
 * through zlib on the host side. Return the number of bytes written to dst (capacity cap), or a negative status
 * (TD_ERR_CAPACITY when the stream decodes to more than cap bytes, TD_ERR_INVALID for a corrupt stream). */
int64_t td_tiff_lzw_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);
int64_t td_tiff_packbits_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);
/* The writer's side of td_tiff_lzw_decode (test rasters, the bench's LZW fixture): one strip or tile → a TIFF 6.0 LZW stream
 * (ClearCode first, EOI last) that libtiff / GDAL / td_tiff_lzw_decode read back. Returns the compressed size or a negative status. */
int64_t td_tiff_lzw_encode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);
/* ---- compressed rasters decoded on the GPU (tiffdecode.hip). The reference decodes every tile window on the
 * host (prediction.py:164 → GDAL → libtiff); here the compressed blocks of a raster cross PCIe once and are decoded by one wave
 * each. comp: DEVICE copy of the file's compressed bytes (padded by >= 8 bytes); block_off / block_nbytes: DEVICE int64
 * [nblocks], position and size of every LZW strip / tile in comp; blocks_out: DEVICE [nblocks][block_cap] bytes, block b decoded
 * at b * block_cap (block_cap = bytes of a full block); decoded: DEVICE int64 [nblocks], status: DEVICE int32 [2 * nblocks + 1] (the
 * first nblocks entries are the blocks' status, the rest is scratch for the second pass) — bytes produced (low 32 bits; the high 32
 * bits count the strings that were copied through memory instead of the LDS ring: a diagnostic), and 0 = ok,
 * 1 = corrupt stream, 2 = more than block_cap bytes (the rules of td_tiff_lzw_decode). Asynchronous on `stream`, whose device must be
 * the calling thread's current one. The decoder waves share workgroups (6 - 12 each, one workgroup per CU at most) and take their blocks
 * from a per-launch counter, so a raster that decodes while forwards run on other streams leaves whole CUs to them. */
td_status td_tiff_lzw_decode_dev(const uint8_t* comp, const int64_t* block_off, const int64_t* block_nbytes, int nblocks,
                                 uint8_t* blocks_out, int64_t block_cap, int64_t* decoded, int32_t* status, void* stream);
/* The same for DEFLATE blocks (TIFF compression 8 / 32946: zlib streams; inflate_core.h): status int32 [nblocks], decoded int64
 * [nblocks]; the Adler-32 trailer is not checked. */
td_status td_tiff_inflate_dev(const uint8_t* comp, const int64_t* block_off, const int64_t* block_nbytes, int nblocks,
                              uint8_t* blocks_out, int64_t block_cap, int64_t* decoded, int32_t* status, void* stream);
/* The decoder td_tiff_inflate_dev runs, instantiated for one lane on the host (parity tests against zlib without a GPU): one zlib
 * stream → dst; returns the bytes produced or a negative status (TD_ERR_INVALID corrupt stream, TD_ERR_CAPACITY). */
int64_t td_tiff_inflate(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap);
/* Decoded blocks (strips: block_w = width; tiles: full padded tiles, row-major grid blocks_across x blocks_down) → the raster
 * image [height][width][spp] uint8 (DEVICE), predictor 2 undone per block row on the way (td_tiff_unpredict's arithmetic).
 * spp <= 4. Asynchronous on `stream`. */
td_status td_tiff_blocks_to_image_dev(const uint8_t* blocks, int64_t block_cap, int block_w, int block_h, int blocks_across,
                                      int blocks_down, int spp, int predictor, uint8_t* image, int width, int height, void* stream);
/* Window of an uncompressed pixel-interleaved raster with contiguous strips (the tile windows of reference
 * prediction.py:164, rasterio.mask.mask(..., crop=True)): `rows` pieces of `row_bytes` bytes lying `row_stride` bytes apart
 * from `file_off` on, read with pread(2) into the dense buffer dst (e.g. pinned staging memory). Returns the bytes read or
 * TD_ERR_INVALID (bad argument, read error, file shorter than the window). Thread-safe (no file position is used). */
int64_t td_read_window(int fd, int64_t file_off, int64_t row_stride, int64_t row_bytes, int64_t rows, uint8_t* dst);
/* The windows of a whole batch in one call: window i = rows[i] pieces of row_bytes[i] bytes lying row_stride apart from file_off[i],
 * read into dst + dst_off[i]; the windows are cut into row bands and spread over `threads` host threads. Returns the bytes read or
 * TD_ERR_INVALID. */
int64_t td_read_windows(int fd, int n, const int64_t* file_off, int64_t row_stride, const int64_t* row_bytes, const int64_t* rows,
                        uint8_t* dst, const int64_t* dst_off, int threads);
/* Undo TIFF predictor 2 (horizontal differencing) in place on one decoded block of rows x cols pixels with
 * `samples` interleaved samples of 1, 2 or 4 bytes (host byte order). */
int td_tiff_unpredict(void* data, int64_t rows, int64_t cols, int samples, int bytes_per_sample);

/* The same file text from contours traced on the device: points / det_info / contour_info are ONE image's slices of the
 * td_trace_contours_dev outputs, copied to the host. Detections the device left to the host (status != 0) are traced
 * from mask_bits as above; when such a detection exists and mask_bits is NULL the call returns TD_ERR_STATE and the
 * caller repeats it with the image's mask rows. Byte-identical to td_tile_polygons_json on the same masks. */
int td_tile_polygons_json_dev(const int16_t* points, int64_t points_cap, const int32_t* det_info, const int32_t* contour_info,
                              const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits,
                              int64_t mask_words, const float* scores, const int32_t* classes, int n,
                              const double* transform, const char* image_id, char* buf, int64_t cap, int64_t* needed);

/* ---- stitching consumer of the prediction files (reference helpers.py:419-476) -------------- */
/* `geometry.simplify(tolerance, preserve_topology=True)` (helpers.py:464-465) for one closed shell ring:
 * topology-preserving Douglas-Peucker as GEOS' TopologyPreservingSimplifier performs it. xy = n (x,y) pairs,
 * first == last for a ring; writes the kept vertices (a subsequence, closed again) to out_xy (capacity out_cap
 * pairs) and returns their number (>= 4 for a ring of >= 4 points), or a negative status. Host code only. */
int td_simplify_ring(const double* xy, int n, double tolerance, double* out_xy, int out_cap);

/* OGC validity of one closed polygon shell (n points, first == last) as GEOS decides it: 1 valid, 0 invalid (self-touching,
 * self-crossing, spikes, fewer than three distinct points, non-finite coordinates), < 0 error. Replaces `geom.is_valid` on the
 * fused crowns, reference TreeDetection/helpers.py:816,819. Host code. */
int td_ring_is_valid(const double* xy, int n);
/* One tile's prediction file → the features it contributes to the image's layer (helpers.py:436-470,
 * process_prediction_file_sync): parse the JSON array the predictor wrote (`json` / `len`, UTF-8 text), take
 * "score" and "polygon_coords" of every entry (coordinates flattened and paired like reshape(-1, 2); fewer than 4
 * points or a missing key fail the whole file, as the reference's exception path does), simplify the ring
 * (tolerance > 0), keep it when it lies `within` box = (minx, miny, maxx, maxy), and encode it as a GeoPackage
 * geometry blob (header with envelope + `srs_id`, little-endian WKB polygon). Feature i = blobs[blob_offsets[i] ..
 * blob_offsets[i+1]) with scores[i]. Returns the feature count; TD_ERR_CAPACITY (needed_bytes / needed_features
 * say how much) when a buffer is too small; TD_ERR_INVALID for malformed input. Host code only, thread-safe. */
int td_stitch_tile_json(const char* json, int64_t len, const double* box, double tolerance, int32_t srs_id,
                        uint8_t* blobs, int64_t blob_cap, int64_t* blob_offsets, double* scores, int max_features,
                        int64_t* needed_bytes, int* needed_features);

/* The same for a whole image in ONE call (reference helpers.py:524-554 process_folder_sync: every tile file of the image's
 * prediction folder): file i = the NUL-terminated path at paths + path_offsets[i] with its own box (boxes[4 i ..]) and
 * srs_ids[i]; the files are read, parsed, simplified, filtered and encoded on `threads` host threads and the features come
 * back concatenated in FILE order (feature numbering as in td_stitch_tile_json, blob_offsets has total + 1 entries).
 * file_status[i] = features of file i, or the negative status that file failed with — such a file is left out, as the
 * reference's per-file try / except leaves it out (helpers.py:419-476). Returns the total feature count; TD_ERR_CAPACITY
 * (needed_bytes / needed_features) when a buffer is too small. Host code only. */
int td_stitch_tile_files(const char* paths, const int64_t* path_offsets, int n_files, const double* boxes, double tolerance,
                         const int32_t* srs_ids, int threads, uint8_t* blobs, int64_t blob_cap, int64_t* blob_offsets,
                         double* scores, int max_features, int32_t* file_status, int64_t* needed_bytes, int* needed_features);

/* ---- outline predicates (reference helpers.py:703-834 fuse_predictions; preprocessing.py:70-95 tile flags) ---- */
/* Relates query rings to a region given as polygons with holes (the outline file's geometries, NOT unioned):
 * ring r = ring_xy[ring_start[r] .. ring_start[r+1]) (x,y pairs, closed), ring_poly[r] = polygon it belongs to,
 * rings of one polygon consecutive with the shell first. For each closed query ring q:
 *   flags[q] bit 0 = query.intersects(union of the polygons), bit 1 = query.within(union of the polygons)
 * (shapely/GEOS semantics for a simple query ring; exact-sign arithmetic). Host code only. */
int td_region_relate(const double* ring_xy, const int64_t* ring_start, const int32_t* ring_poly, int n_rings,
                     const double* query_xy, const int64_t* query_start, int n_queries, uint8_t* flags);

/* ---- crown post-processing (reference postprocessing.py:25-347) ----------------------------- */
/* Raster statistics inside each crown's bounding circle. raster: device float32 [rows][cols]; transform: host
 * (a,b,c,d,e,f) of the raster; window: host (row_lo, col_lo, row_hi, col_hi) = the reference's subset of the raster
 * (its whole extent when the bounds passed are the raster's own); circles: device float32 [n][3] (centre x, y of
 * the crown's bounding box and the largest vertex distance from it, computed from the float32 vertex arrays).
 * mode 0 (get_height_within_polygon): out [n][3] = max value, x, y of its first occurrence (float64 membership test);
 * mode 1 (get_ndvi_within_polygon / NDVI half of get_metadata_within_polygon with radius_scale 0.5): out [n][4] =
 * min, max, mean, population variance (float32 membership test). -1 everywhere for a circle without pixels.
 * Asynchronous on `stream`. */
td_status td_crown_stats(const float* raster, int rows, int cols, const double* transform, const int32_t* window,
                         const float* circles, int n, int mode, float radius_scale, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TREEDET_H */
