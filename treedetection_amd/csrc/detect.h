// Shared pieces of the detection kernels (rpn.hip, roi.hip): level tables, box decoding, IoU test.
#pragma once
#include "common.h"

constexpr int RPN_LEVELS = 5;        // p2..p6
constexpr int RPN_A = 3;             // anchors per position (ratios 0.5, 1, 2)
constexpr int RPN_HEAD_C = 15;       // fused head: 3 objectness logits + 12 anchor deltas per position
constexpr int RPN_CAND = 1024;       // per-(image, level) candidate slots (>= pre_nms_topk)
constexpr int RPN_CAND_POW2 = 1024;
constexpr int ROI_LEVELS = 4;        // p2..p5

struct RpnLevels {
    const float* head[RPN_LEVELS];   // NHWC [B, h, w, 15] fused head output of each level
    int h[RPN_LEVELS], w[RPN_LEVELS], stride[RPN_LEVELS];
    int anchor_off[RPN_LEVELS];      // prefix of h*w*A
    int total_anchors;
    float base[RPN_LEVELS][RPN_A][4];  // cell anchors (x1,y1,x2,y2), computed in double on the host
};

struct FeatLevels {
    const void* feat[ROI_LEVELS];    // NHWC [B, h, w, C]
    int h[ROI_LEVELS], w[ROI_LEVELS];
    float scale[ROI_LEVELS];         // 1/stride
    int C;
};

#define TD_SCALE_CLAMP 4.135166556742356f   // log(1000/16)

// Box2BoxTransform.apply_deltas — one IEEE op per step, float32.
__device__ __forceinline__ void decode_box(float ax1, float ay1, float ax2, float ay2, float d0, float d1, float d2,
                                           float d3, float wx, float wy, float ww, float wh, float& x1, float& y1,
                                           float& x2, float& y2) {
    const float widths = __fsub_rn(ax2, ax1), heights = __fsub_rn(ay2, ay1);
    const float cx = __fadd_rn(ax1, __fmul_rn(0.5f, widths));
    const float cy = __fadd_rn(ay1, __fmul_rn(0.5f, heights));
    const float dx = __fdiv_rn(d0, wx), dy = __fdiv_rn(d1, wy);
    const float dw = fminf(__fdiv_rn(d2, ww), TD_SCALE_CLAMP), dh = fminf(__fdiv_rn(d3, wh), TD_SCALE_CLAMP);
    const float pcx = __fadd_rn(__fmul_rn(dx, widths), cx);
    const float pcy = __fadd_rn(__fmul_rn(dy, heights), cy);
    const float pw = __fmul_rn(expf(dw), widths);
    const float ph = __fmul_rn(expf(dh), heights);
    x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw));
    y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
    x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw));
    y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
}

// torchvision nms: inter / (area_a + area_b - inter) > thr
__device__ __forceinline__ bool iou_gt(const float4& a, float area_a, const float4& b, float thr) {
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
    return __fdiv_rn(inter, uni) > thr;
}

// ---- launchers (rpn.hip) ------------------------------------------------------------------------------
// key_ws: rpn_topk_ws_elems(B, total_anchors) uint32 (dense keys + the per-(image, level) coarse histograms)
size_t rpn_topk_ws_elems(int B, int total_anchors);
td_status rpn_topk_decode_launch(const RpnLevels& lv, const ImgSizes& valid, int B, int topk, uint32_t* key_ws,
                                 float* cand_boxes, float* cand_scores, int* cand_valid, int* cand_idx,
                                 hipStream_t stream);
td_status nms_launch(const float* sorted_boxes, const int* counts, const int* valid, int items, int stride_items,
                     float thr, unsigned long long* mask_ws, int* keep_idx, int* keep_count, int max_keep,
                     hipStream_t stream);
// n > 1024 boxes of ONE item (td_nms): counting sort + bit matrix + LDS-resident scan; mask_ws: n * ceil(n / 64) words
td_status nms_big_launch(const float* boxes, const float* scores, int n, float thr, float* sboxes, float* sscores, int* sidx,
                         int* scount, unsigned long long* mask_ws, int* keep_pos, int* keep_count, hipStream_t stream);
td_status rpn_merge_launch(const float* cand_boxes, const float* cand_scores, const int* keep_idx, const int* keep_count,
                           int B, int post_topk, float* props, float* prop_scores, int* prop_count, int prop_stride,
                           hipStream_t stream);
td_status sort_boxes_launch(const float* boxes, const float* scores, const int* flags, const int* counts, int items,
                            int stride_items, float* sboxes, float* sscores, int* sidx, int* scount, hipStream_t stream);
td_status gather_keep_launch(const int* sidx, const int* keep_pos, const int* keep_count, int n, int* out_idx,
                             hipStream_t stream);

// ---- launchers (roi.hip) ------------------------------------------------------------------------------
// RoIAlign over FPN levels. rois [items][roi_stride][4]; counts[items]; row of (item, r):
//   compact == 0: item*roi_stride + r        compact == 1: prefix(counts)[item] + r
td_status roi_align_launch(const FeatLevels& fl, const float* rois, const int* counts, int items, int roi_stride,
                           int pooled, int compact, void* out, int* total_rows, int precision, hipStream_t stream);
td_status roi_align_single_launch(const void* feat, int H, int W, int C, const float* rois, int R, float scale,
                                  int pooled, void* out, int precision, hipStream_t stream);
// softmax + decode (10,10,5,5) + clip + score filter per proposal
td_status det_decode_launch(const float* cls_reg /*[rows][cr_stride]: 2 logits + 4 deltas*/, int cr_stride,
                            const float* props, const int* prop_count, const ImgSizes& valid, int B, int prop_stride,
                            float score_thresh, float* boxes, float* scores, int* flags, hipStream_t stream);
// kept detections → output-space boxes (scale, clip, drop empty) + compact per image
td_status det_finalize_launch(const float* sboxes, const float* sscores, const int* keep_pos, const int* keep_count,
                              const ImgSizes& valid, const ImgSizes& outsz, int B, int stride_items, int max_det,
                              float* det_boxes_net, float* out_boxes, float* out_scores, int* out_classes,
                              int* out_count, hipStream_t stream);
// 1x1 conv to one channel + sigmoid over [rows*784][C] → probs
td_status mask_predict_launch(const void* x, const float* w, float bias, int C, int rows_max, const int* rows_dyn,
                              int rows_mul, float* logits_out, float* probs_out, int precision, hipStream_t stream);
// scatter compact mask probs [total][784] to per-image slots [B][D][784]
td_status mask_scatter_launch(const float* compact, const int* counts, int B, int D, float* out, hipStream_t stream);
td_status paste_masks_launch(const float* probs /*[B][D][784]*/, const float* boxes /*[B][D][4]*/, const int* counts,
                             const ImgSizes& outsz, int B, int D, float thresh, int* region, long long* offset,
                             uint32_t* bits, long long words_per_image, hipStream_t stream);
