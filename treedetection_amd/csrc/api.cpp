// extern "C" surface of libtreedet_hip.so (include/treedet.h): error channel + op-level entry points.
// (The engine entry points live in engine.cpp, the host contour tracer in contours.cpp.)
#include "common.h"
#include "detect.h"
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

// (td_set_error / td_last_error: error.cpp — shared with the host-only sanitizer build)

namespace {
// scratch for the op-level NMS / paste entry points (tests only; the engine owns its own workspace)
td_status scratch(void** p, size_t bytes) {
    TD_HIP_CHECK(hipMalloc(p, bytes ? bytes : 16));
    return TD_OK;
}
}  // namespace

extern "C" {

void td_model_desc_default(td_model_desc* d) {
    d->num_classes = 1;
    d->precision = TD_PRECISION_FP32;
    d->pre_nms_topk = 1000;
    d->post_nms_topk = 1000;
    d->detections_per_image = 100;
    d->rpn_nms_thresh = 0.7f;
    d->score_thresh = 0.3f;
    d->nms_thresh = 0.5f;
    d->mask_thresh = 0.5f;
}

static td_status conv2d_api(ConvArgs& a, int precision, void* stream);

td_status td_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* bias,
                         const void* residual, int res_shift, void* y, int B, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int relu, int precision,
                         void* stream) {
    TD_REQUIRE(x && w && y, "td_conv2d_nhwc: null pointer");
    TD_REQUIRE(stride >= 1 && KH >= 1 && KW >= 1 && B >= 1, "td_conv2d_nhwc: bad geometry");
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.res_shift = res_shift; a.relu = relu; a.out_mode = 0;
    a.M = B * a.Ho * a.Wo; a.m_dyn = nullptr; a.m_mul = 1;
    a.out_f32 = (precision & 0x10000) && (precision & 0xff) == TD_PRECISION_FP16 ? 1 : 0;      // float16 tensors, float32 y (the heads)
    a.tile_cfg = ((precision >> 8) & 0xff) - 1;          // tests: force one block-tile variant (0 = the library chooses)
    return conv2d_api(a, precision, stream);
}

td_status td_conv2d_head_nhwc(const void* x, const void* w, const float* bias, const void* head_w, const float* head_b,
                              float* head_y, int B, int H, int W, int Cin, int KH, int KW, int pad, int head_n, int precision,
                              void* stream) {
    TD_REQUIRE(x && w && head_w && head_y, "td_conv2d_head_nhwc: null pointer");
    TD_REQUIRE((precision & 0xff) == TD_PRECISION_FP16, "td_conv2d_head_nhwc: float16 tensors only");
    TD_REQUIRE(KH >= 1 && KW >= 1 && B >= 1 && head_n >= 1 && head_n <= 32, "td_conv2d_head_nhwc: bad geometry");
    ConvArgs a{};
    a.x = x; a.w = w; a.bias = bias;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = 256; a.KH = KH; a.KW = KW;
    a.stride = 1; a.pad = pad;
    a.Ho = H + 2 * pad - KH + 1;
    a.Wo = W + 2 * pad - KW + 1;
    a.relu = 1; a.M = B * a.Ho * a.Wo; a.m_mul = 1;
    a.tile_cfg = ((precision >> 8) & 0xff) - 1;
    TD_REQUIRE(conv_head_capable(a.tile_cfg, TD_PRECISION_FP16), "td_conv2d_head_nhwc: tile id %d does not own 256 output channels per block", a.tile_cfg);
    a.head_w = head_w; a.head_b = head_b; a.head_y = head_y; a.head_n = head_n;
    // the layer's own output is never written by a fused launch; `y` still has to be a valid pointer for the argument checks
    a.y = head_y;
    return conv2d_api(a, precision, stream);
}

static td_status conv2d_api(ConvArgs& a, int precision, void* stream) {
    const int Cin = a.Cin, Cout = a.Cout, KH = a.KH, KW = a.KW;
    const void* w = a.w;
    if (conv_cfg_is_bd(a.tile_cfg) && Cin % ((precision & 0xff) == TD_PRECISION_FP16 ? 64 : 32) == 0) {
        // tests: the filter-direct tiles (conv_bdirect.hip) need the filters in fragment order: packed here from the caller's
        // [Cout][KH][KW][Cin] bank (the engine packs once at load time)
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int es = (precision & 0xff) == TD_PRECISION_FP16 ? 2 : 4;
        const size_t n = (size_t)Cout * KH * KW * Cin;
        std::vector<unsigned char> bits(n * es), packed;
        TD_HIP_CHECK(hipStreamSynchronize(s));
        TD_HIP_CHECK(hipMemcpy(bits.data(), w, n * es, hipMemcpyDeviceToHost));
        conv_bd_pack(bits.data(), es, Cout, KH, KW, Cin, packed);
        void* wf = nullptr;
        td_status st = scratch(&wf, packed.size());
        if (st < 0) return st;
        hipError_t herr = hipMemcpy(wf, packed.data(), packed.size(), hipMemcpyHostToDevice);
        a.w_frag = wf;
        if (herr == hipSuccess) st = conv2d_launch(a, precision & 0xff, s);
        hipError_t herr2 = hipStreamSynchronize(s);
        (void)hipFree(wf);
        if (st < 0) return st;
        TD_HIP_CHECK(herr);
        TD_HIP_CHECK(herr2);
        return TD_OK;
    }
    if (a.tile_cfg == 21 || a.tile_cfg == 22 || a.tile_cfg == 28) {
        td_set_error("conv2d: tile_cfg %d (stream-K / 4-wave 256x256) is an experiment that lost to the block tiles; it lives in "
                     "csrc/experimental/ and is not part of the product library", a.tile_cfg);
        return TD_ERR_INVALID;
    }
    return conv2d_launch(a, precision & 0xff, static_cast<hipStream_t>(stream));
}

static td_status wino_api(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int H, int W, int Cin,
                          int Cout, int relu, void* stream, const float* head_w, const float* head_b, float* head_y, int head_n);

td_status td_conv2d_winograd_nhwc(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int H,
                                  int W, int Cin, int Cout, int relu, void* stream) {
    TD_REQUIRE(y, "td_conv2d_winograd_nhwc: bad arguments");
    return wino_api(x, w, scale, bias, y, B, H, W, Cin, Cout, relu, stream, nullptr, nullptr, nullptr, 0);
}

td_status td_conv2d_winograd_head_nhwc(const float* x, const float* w, const float* bias, const float* head_w, const float* head_b,
                                       float* head_y, int B, int H, int W, int Cin, int head_n, void* stream) {
    TD_REQUIRE(head_w && head_y && head_n >= 1 && head_n <= 32, "td_conv2d_winograd_head_nhwc: bad arguments");
    return wino_api(x, w, nullptr, bias, nullptr, B, H, W, Cin, 256, 1, stream, head_w, head_b, head_y, head_n);
}

static td_status wino_api(const float* x, const float* w, const float* scale, const float* bias, float* y, int B, int H, int W, int Cin,
                          int Cout, int relu, void* stream, const float* head_w, const float* head_b, float* head_y, int head_n) {
    TD_REQUIRE(x && w && B >= 1 && H >= 1 && W >= 1, "td_conv2d_winograd_nhwc: bad arguments");
    TD_REQUIRE(Cin % 32 == 0 && Cout % 4 == 0, "td_conv2d_winograd_nhwc: Cin must be a multiple of 32, Cout of 4");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const char* tenv = getenv("TD_WINO_TILE");            // tests: 4 = the F(4x4,3x3) form the engine uses on the large maps
    const bool f43 = head_w || (tenv && atoi(tenv) == 4);  // the head rides in the F(4x4) output transform only
    const int P = f43 ? 36 : 16;                          // transform planes
    const size_t T = f43 ? (size_t)B * ((H + 3) / 4) * ((W + 3) / 4) : (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    std::vector<float> wh((size_t)Cout * 9 * Cin), uh((size_t)P * Cout * Cin);
    TD_HIP_CHECK(hipMemcpy(wh.data(), w, wh.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (f43) wino43_filter_transform(wh.data(), Cout, Cin, uh.data());
    else wino_filter_transform(wh.data(), Cout, Cin, uh.data());
    void *U = nullptr, *V = nullptr, *Mb = nullptr;
    td_status st;
    if ((st = scratch(&U, uh.size() * 4)) < 0 || (st = scratch(&V, P * T * Cin * 4)) < 0 || (st = scratch(&Mb, P * T * Cout * 4)) < 0) {
        (void)hipFree(U); (void)hipFree(V); (void)hipFree(Mb);
        return st;
    }
    TD_HIP_CHECK(hipMemcpy(U, uh.data(), uh.size() * 4, hipMemcpyHostToDevice));
    const char* folde = getenv("TD_WINO_FOLD");           // tests: F(4x4) contraction + output transform in one launch (wino_fused.hip)
    const bool fold = f43 && !head_w && folde && atoi(folde) != 0;
    if (fold) TD_REQUIRE(wino43_fused_ok(B, H, W, Cin, Cout), "td_conv2d_winograd_nhwc: TD_WINO_FOLD needs Cin %% 128 == 0, Cout %% 64 == 0");
    const char* fenv = getenv("TD_WINO_FUSED");           // tests compare the two forms of the contraction (bit-identical)
    const bool fused = !f43 && (!fenv || atoi(fenv) != 0);
    ConvArgs a{};
    a.w = U; a.y = Mb;
    a.Cin = Cin; a.Cout = Cout; a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
    a.M = (int)T; a.m_mul = 1; a.tile_cfg = -1;
    a.w_bs = (long long)Cout * Cin; a.y_bs = (long long)T * Cout;
    if (fused) {
        a.x = x; a.B = B; a.H = H; a.W = W;
        st = wino_gemm_launch(a, s);
    } else {
        st = f43 ? wino43_input_launch(x, B, H, W, Cin, static_cast<float*>(V), nullptr, s)
                 : wino_input_launch(x, B, H, W, Cin, static_cast<float*>(V), nullptr, 1, 0, (int)T, s);
        if (st == TD_OK && fold) {
            st = wino43_fused_launch(static_cast<float*>(V), static_cast<float*>(U), B, H, W, Cin, Cout, scale, bias, relu, y, nullptr, s);
        } else if (st == TD_OK) {
            a.x = V; a.B = 1; a.H = 1; a.W = (int)T; a.Ho = 1; a.Wo = (int)T;
            a.batch_count = P; a.x_bs = (long long)T * Cin;
            st = conv2d_launch(a, TD_PRECISION_FP32, s);
        }
    }
    if (fold) {
    } else if (st == TD_OK && head_w)
        st = wino43_output_head_launch(static_cast<float*>(Mb), B, H, W, Cout, scale, bias, relu, head_w, head_b, head_y, head_n, s);
    else if (st == TD_OK)
        st = f43 ? wino43_output_launch(static_cast<float*>(Mb), B, H, W, Cout, scale, bias, relu, y, nullptr, s)
                 : wino_output_launch(static_cast<float*>(Mb), B, H, W, Cout, scale, bias, relu, y, nullptr, 1, 0, (int)T, s);
    hipError_t herr = hipStreamSynchronize(s);
    (void)hipFree(U); (void)hipFree(V); (void)hipFree(Mb);
    if (st < 0) return st;
    TD_HIP_CHECK(herr);
    return TD_OK;
}

td_status td_resize_tile_u8(const uint8_t* src, int h, int w, int c, uint8_t* dst, int out_h, int out_w,
                            int dst_pitch_px, void* tmp_dev, void* stream) {
    return resize_tile_u8_launch(src, h, w, c, dst, out_h, out_w, dst_pitch_px, tmp_dev, static_cast<hipStream_t>(stream));
}

td_status td_resize_batch_u8(const uint8_t* const* src_tiles, int n, int h, int w, int c, uint8_t* dst, int out_h,
                             int out_w, int dst_pitch_px, int64_t dst_image_stride_bytes, void* tmp_dev, void* stream) {
    return resize_batch_u8_launch(src_tiles, n, h, w, c, dst, out_h, out_w, dst_pitch_px, (size_t)dst_image_stride_bytes,
                                  tmp_dev, static_cast<hipStream_t>(stream));
}

td_status td_bottleneck_tail_nhwc(const void* x, const void* w2, const float* scale2, const float* bias2, const void* w3,
                                  const float* scale3, const float* bias3, const void* shortcut, void* y, int B, int H, int W,
                                  int mid, int cout, int precision, void* stream) {
    TD_REQUIRE(x && w2 && w3 && shortcut && y && B > 0 && H > 0 && W > 0, "td_bottleneck_tail_nhwc: null argument or empty tensor");
    TailArgs a{};
    a.x = x; a.w2 = w2; a.scale2 = scale2; a.bias2 = bias2; a.w3 = w3; a.scale3 = scale3; a.bias3 = bias3; a.res = shortcut; a.y = y;
    a.B = B; a.H = H; a.W = W; a.MID = mid; a.COUT = cout; a.M = B * H * W;
    return bottleneck_tail_launch(a, precision, static_cast<hipStream_t>(stream));
}

td_status td_resize_bilinear_f64(const double* src, int c, int h, int w, float* dst, int out_h, int out_w, int dst_pitch_px,
                                 int64_t dst_plane_stride, void* stream) {
    return resize_bilinear_f64_launch(src, c, h, w, dst, out_h, out_w, dst_pitch_px, (long long)dst_plane_stride, static_cast<hipStream_t>(stream));
}

void td_resize_shape(int h, int w, int short_edge, int max_size, int* out_h, int* out_w) {
    // detectron2 ResizeShortestEdge.get_output_shape (python doubles; int(x + 0.5)) — Appendix A item 2
    const double scale = (double)short_edge / (double)(h < w ? h : w);
    double newh, neww;
    if (h < w) { newh = short_edge; neww = scale * w; }
    else { newh = scale * h; neww = short_edge; }
    const double mx = newh > neww ? newh : neww;
    if (mx > max_size) {
        const double s2 = (double)max_size / mx;
        newh *= s2;
        neww *= s2;
    }
    *out_h = (int)(newh + 0.5);
    *out_w = (int)(neww + 0.5);
}

td_status td_nms(const float* boxes, const float* scores, int n, float iou_thresh, int32_t* keep_idx,
                 int32_t* keep_count, void* stream) {
    TD_REQUIRE(boxes && scores && keep_idx && keep_count, "td_nms: null pointer");
    TD_REQUIRE(n >= 0 && n <= 32768, "td_nms: n=%d must be in [0, 32768]", n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        TD_HIP_CHECK(hipMemsetAsync(keep_count, 0, sizeof(int32_t), s));
        return TD_OK;
    }
    void *sb = nullptr, *ss = nullptr, *si = nullptr, *sc = nullptr, *mk = nullptr, *kp = nullptr;
    td_status st;
    if ((st = scratch(&sb, sizeof(float) * 4 * n)) < 0) return st;
    if ((st = scratch(&ss, sizeof(float) * n)) < 0) return st;
    if ((st = scratch(&si, sizeof(int) * n)) < 0) return st;
    if ((st = scratch(&sc, sizeof(int))) < 0) return st;
    if ((st = scratch(&mk, sizeof(unsigned long long) * n * td_cdiv(n, 64))) < 0) return st;
    if ((st = scratch(&kp, sizeof(int) * n)) < 0) return st;
    if (n > 1024) {      // beyond one block's LDS sort and 16-wave scan (the engine's own item sizes are bounded at td_engine_create)
        st = nms_big_launch(boxes, scores, n, iou_thresh, (float*)sb, (float*)ss, (int*)si, (int*)sc, (unsigned long long*)mk, (int*)kp, keep_count, s);
    } else {
        st = sort_boxes_launch(boxes, scores, nullptr, nullptr, 1, n, (float*)sb, (float*)ss, (int*)si, (int*)sc, s);
        if (st >= 0) st = nms_launch((float*)sb, (int*)sc, nullptr, 1, n, iou_thresh, (unsigned long long*)mk, (int*)kp, keep_count, n, s);
    }
    if (st >= 0) st = gather_keep_launch((int*)si, (int*)kp, keep_count, n, keep_idx, s);
    hipError_t herr = hipStreamSynchronize(s);
    (void)hipFree(sb); (void)hipFree(ss); (void)hipFree(si); (void)hipFree(sc); (void)hipFree(mk); (void)hipFree(kp);
    if (st < 0) return st;
    TD_HIP_CHECK(herr);
    return TD_OK;
}

td_status td_roi_align(const void* feat, int H, int W, int C, const float* rois, int R, float spatial_scale,
                       int pooled, void* out, int precision, void* stream) {
    TD_REQUIRE(feat && rois && out, "td_roi_align: null pointer");
    TD_REQUIRE(H >= 1 && W >= 1 && pooled >= 1 && R >= 0, "td_roi_align: bad geometry");
    if (R == 0) return TD_OK;
    return roi_align_single_launch(feat, H, W, C, rois, R, spatial_scale, pooled, out, precision, static_cast<hipStream_t>(stream));
}

td_status td_paste_masks(const float* mask_probs, const float* boxes, int n, int out_h, int out_w, float thresh,
                         int32_t* mask_region, int64_t* mask_offset, uint32_t* mask_bits, int64_t mask_words_cap,
                         void* stream) {
    TD_REQUIRE(mask_probs && boxes && mask_region && mask_offset && mask_bits, "td_paste_masks: null pointer");
    TD_REQUIRE(n >= 0 && n <= 1024 && out_h >= 1 && out_w >= 1, "td_paste_masks: bad shape");
    if (n == 0) return TD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* cnt = nullptr;
    td_status st = scratch(&cnt, sizeof(int));
    if (st < 0) return st;
    TD_HIP_CHECK(hipMemcpy(cnt, &n, sizeof(int), hipMemcpyHostToDevice));
    ImgSizes outsz{};
    outsz.h[0] = out_h;
    outsz.w[0] = out_w;
    st = paste_masks_launch(mask_probs, boxes, (int*)cnt, outsz, 1, n, thresh, mask_region,
                            reinterpret_cast<long long*>(mask_offset), mask_bits, mask_words_cap, s);
    hipError_t herr = hipStreamSynchronize(s);
    (void)hipFree(cnt);
    if (st < 0) return st;
    TD_HIP_CHECK(herr);
    return TD_OK;
}

td_status td_paste_masks_batch(const float* mask_probs, const float* boxes, const int32_t* counts, const int32_t* out_hw,
                               int batch, int dets_per_image, float thresh, int32_t* mask_region, int64_t* mask_offset,
                               uint32_t* mask_bits, int64_t mask_words_per_image, void* stream) {
    TD_REQUIRE(mask_probs && boxes && counts && out_hw && mask_region && mask_offset && mask_bits,
               "td_paste_masks_batch: null pointer");
    TD_REQUIRE(batch >= 1 && batch <= TD_MAX_BATCH && dets_per_image >= 1 && dets_per_image <= 1024,
               "td_paste_masks_batch: batch %d (max %d) / detections per image %d (max 1024)", batch, TD_MAX_BATCH, dets_per_image);
    ImgSizes outsz{};
    for (int i = 0; i < batch; ++i) {
        TD_REQUIRE(out_hw[2 * i] >= 1 && out_hw[2 * i + 1] >= 1, "td_paste_masks_batch: bad output size of image %d", i);
        outsz.h[i] = out_hw[2 * i];
        outsz.w[i] = out_hw[2 * i + 1];
        const int64_t need = (int64_t)dets_per_image * ((outsz.w[i] + 2 + 31) / 32) * outsz.h[i];
        TD_REQUIRE(mask_words_per_image >= need, "td_paste_masks_batch: image %d needs %lld mask words, capacity %lld", i,
                   (long long)need, (long long)mask_words_per_image);
    }
    return paste_masks_launch(mask_probs, boxes, counts, outsz, batch, dets_per_image, thresh, mask_region,
                              reinterpret_cast<long long*>(mask_offset), mask_bits, mask_words_per_image,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
