// extern "C" surface of libtreedet_hip.so (include/treedet.h): error channel + op-level entry points.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cmath>

static thread_local char g_err[1024] = "";

void td_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* td_last_error(void) { return g_err; }

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

td_status td_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* bias,
                         const void* residual, int res_shift, void* y, int B, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int relu, int precision,
                         void* stream) {
    TD_REQUIRE(x && w && y, "td_conv2d_nhwc: null pointer");
    TD_REQUIRE(stride >= 1 && KH >= 1 && KW >= 1, "td_conv2d_nhwc: bad geometry");
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.bias = bias; a.res = residual; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.stride = stride; a.pad = pad;
    a.Ho = (H + 2 * pad - KH) / stride + 1;
    a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.res_shift = res_shift; a.relu = relu; a.out_mode = 0;
    a.M = B * a.Ho * a.Wo; a.m_dyn = nullptr; a.m_mul = 1;
    return conv2d_launch(a, precision, static_cast<hipStream_t>(stream));
}


// ---- not built yet (replaced as the engine grows) ---------------------------------------------
#define TD_STUB(...) { td_set_error("%s: not built yet", __func__); return TD_ERR_STATE; }
td_status td_engine_create(const td_model_desc*, int, td_engine**) TD_STUB()
td_status td_engine_load_weights(td_engine*, const td_tensor_desc*, size_t) TD_STUB()
td_status td_engine_reserve(td_engine*, int, int, int) TD_STUB()
td_status td_engine_forward(td_engine*, const void*, int, const int32_t*, const int32_t*, int, int, int, void*, td_detections*) TD_STUB()
td_status td_engine_tensor(td_engine*, const char*, void**, int64_t*, int*) TD_STUB()
void td_engine_destroy(td_engine*) {}
td_status td_resize_tile_u8(const uint8_t*, int, int, int, uint8_t*, int, int, int, void*, void*) TD_STUB()
void td_resize_shape(int, int, int, int, int*, int*) {}
td_status td_nms(const float*, const float*, int, float, int32_t*, int32_t*, void*) TD_STUB()
td_status td_roi_align(const void*, int, int, int, const float*, int, float, int, void*, int, void*) TD_STUB()
td_status td_paste_masks(const float*, const float*, int, int, int, float, int32_t*, int64_t*, uint32_t*, int64_t, void*) TD_STUB()
int td_find_contours(const uint8_t*, int, int, int32_t*, int, int32_t*, int) TD_STUB()

}  // extern "C"
