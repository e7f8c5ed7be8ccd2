// Shared declarations of libtreedet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstddef>
#include "../../include/treedet.h"
#include <string>
#include <vector>

void td_set_error(const char* fmt, ...);

// epilogue.cpp: the text of one tile's prediction file from its packed rows on the host; returns the entry count or < 0
int td_polygons_json_text(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits, int64_t mask_words,
                          const float* scores, const int32_t* classes, int n, const double* transform, const char* image_id,
                          std::string& out, const char* who);

#define TD_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            td_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TD_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

#define TD_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            td_set_error(__VA_ARGS__);   \
            return TD_ERR_INVALID;       \
        }                                \
    } while (0)

#define TD_KERNEL_CHECK() TD_HIP_CHECK(hipGetLastError())

static inline int td_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- convolution (conv_igemm.hip) -------------------------------------------------------------
struct ConvArgs {
    const void* x;        // NHWC [B,H,W,Cin]
    const void* w;        // [Cout][KH][KW][Cin]
    const float* scale;   // [Cout] or nullptr
    const float* bias;    // [Cout] or nullptr
    const void* res;      // residual or nullptr
    void* y;              // NHWC [B,Ho,Wo,Cout] (out_mode 0) / [B,2Ho,2Wo,Cout/4] (out_mode 1)
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int res_shift;        // 0: residual same size; 1: residual at half resolution (nearest 2x upsample)
    int relu;
    int out_mode;         // 0 plain, 1 deconv2x2 pixel shuffle: n = (dy*2+dx)*Cq + co, Cq = Cout/4
    int M;                // B*Ho*Wo
    const int* m_dyn;     // optional device scalar: effective rows = min(M, *m_dyn * m_mul - m_off)
    int m_mul;
    int m_off;            // rows of the dynamic count that belong to earlier launches (Winograd slabs); 0 otherwise
    int out_f32;          // fp16 path only: write y as float32 (RPN / box-predictor heads feed the fp32 selection kernels)
    // batched launch (blockIdx.y = 0 .. batch_count-1): per-batch element offsets of x / w / y. Used by the Winograd path,
    // whose 16 transform planes are 16 independent 1x1 contractions (winograd.hip). 0 / 1 = a plain launch.
    int batch_count;
    long long x_bs, w_bs, y_bs;
    const void* w_frag;   // the same filters in MFMA fragment order (conv_bdirect.hip: tile ids 23 - 27), or nullptr
    // only read by csrc/experimental/conv_streamk.hip (not in the product library): partial-tile slots / ticket counters
    float* sk_ws;
    int* sk_cnt;
    int tile_cfg;         // -1 = heuristic; 0..3 = block tile 128x128, 128x64, 64x128, 64x64 with 2 LDS stages,
                          // 4..7 = the same tiles with 3 stages (engine autotunes)
    // fused 1x1 head (fp16 engine, block tiles that own all 256 output channels: conv_head_capable): the finished fp16
    // tile — this layer's output — is contracted with head_w [head_n <= 32][Cout = 256] straight from its LDS staging and
    // only head_y [M][head_n] (fp32, + head_b) is written; y is NOT written. The RPN's 3x3 conv + its 15-row head.
    const void* head_w;
    const float* head_b;
    float* head_y;
    int head_n;
    // grouped launch over pyramid levels (conv_pp8_grouped_launch; fp16, 3x3 / stride 1 / pad 1, 256 -> 256): ONE grid whose tiles
    // walk nlev independent problems of the same layer shape on different maps — the FPN's four output convs (own filters per
    // level) or the RPN conv + head on p2..p6 (shared filters). Per level: input, filters, bias, output (or head output), map size.
    int nlev;
    struct Level {
        const void* x;
        const void* w;
        const float* bias;
        void* y;
        float* head_y;
        int H, W, M, tile0;       // M = B*H*W rows; tile0 = index of the level's first 256-row tile in the grid
    } lev[5];
    int ntiles;                   // tiles of all levels
};
// tile ids whose block owns 256 output channels at once (fp16): the head fusion above applies
static inline bool conv_head_capable(int cfg, int precision) {
    // (the single-stage 256-wide tiles 14 / 16 stage their output as fp32 wave-rows, not as one fp16 tile: not capable)
    return precision == TD_PRECISION_FP16 && (cfg == 9 || cfg == 10 || cfg == 12 || cfg == 13 || cfg == 17 || cfg == 23 || cfg == 27 || cfg == 29);
}
// tile_cfg ids (conv_igemm.hip:dispatch): 0..3 4-wave tiles 128x128 / 128x64 / 64x128 / 64x64 (2 LDS stages), 4..7 the same
// with 3 stages (measured no better: not tuned over), 8 = 256x128 / 9 = 128x256 (8 waves), 10 = 256x256 (16 waves),
// 11..13 = 256x256 with larger per-wave tiles, 14..16 = single-LDS-stage 256x256 / 128x128 / 128x256 (thin 1x1 layers),
// 17 = conv_pp8_kernel: 256x256, 8 waves, ping-pong phases, DMA 1.5 k-chunks ahead (fp16 only),
// 18..20 = plane_gemm_kernel: persistent 64x128 / 128x128 / 64x64 tile walk for the fp32 Winograd plane contractions,
// 21 / 22 / 28 = retired ids (stream-K and the 4-wave 256x256 tile: measured slower in round 3, sources kept under csrc/experimental/, not built
// into the product; conv2d_launch refuses them), 23 / 24 / 25 / 26 / 27 = conv_bd_kernel: 64x256 / 64x128 / 64x128 with two k-chunks per barrier / 64x128 with three k-steps of loads in flight / 64x256 with two, filter fragments
// straight from a fragment-ordered copy of the filters into registers (conv_bdirect.hip)
// 29 / 30 = conv_bd_kernel 128x256 / 128x128, three k-steps of loads in flight: taller tiles, half the filter re-reads (round 4)
// 31 / 32 = conv_igemm_kernel 256x32 / 128x32 (4 x 1 waves): layers with at most 32 output channels (round 4)
// 33 = conv_bs_kernel (conv_bstat.hip, round 4): filter-stationary 1x1 for the thin-K layers (<= 4 k-chunks), one block per CU
#define TD_CONV_TILE_CFG_MAX 33
static inline bool conv_cfg_is_bd(int cfg) { return (cfg >= 23 && cfg <= 27) || cfg == 29 || cfg == 30 || cfg == 33; }      // tiles that read the fragment-ordered filter copy
// Tried in this order — from the tile that moves the fewest bytes per FLOP to the one that moves the most — and a later candidate
// replaces the best so far only when it is more than TD_TUNE_HYST percent faster (default 2): among tiles that tie within the
// measurement noise the one with the larger footprint wins, which keeps the choice (and with it the HBM / L2 traffic the PMC
// passes report) from flipping between runs — fc1 was seen on the 128 x 128 tile in one run and on a 64 x 128 tile (+2 GB of filter
// re-reads per step, same time) in the next.
// 15 / 16 only for <= 4 k-steps, 17 only for fp16, 18-20 only for plane contractions, 23-27 / 29 / 30 / 33 only with packed filters, 31 / 32 only for <= 32 output channels, 33 only where conv_bs_ok
static const int TD_CONV_TUNE_CANDIDATES[] = {33, 10, 17, 29, 16, 0, 15, 30, 23, 27, 1, 2, 24, 25, 26, 31, 3, 32, 18, 19, 20};
td_status conv2d_launch(const ConvArgs& a, int precision, hipStream_t stream);
// a.nlev levels (x / w / bias / y / head_y / H / W filled in; M, tile0, ntiles are computed here) in one conv_pp8_kernel grid;
// everything else (B, Cin, Cout = 256, KH = KW = 3, relu, head_w / head_b / head_n) from the common fields. Bit-identical to one
// conv2d_launch per level on any tile.
td_status conv_pp8_grouped_launch(ConvArgs a, hipStream_t stream);
// filter-direct form (conv_bdirect.hip)
void conv_bd_pack(const void* w_ohwi, int elem_bytes, int cout, int kh, int kw, int cin, std::vector<unsigned char>& out);
bool conv_bd_ok(const ConvArgs& a, int precision);
td_status conv_bd_launch(const ConvArgs& a, int precision, int variant, hipStream_t stream);
// filter-stationary form (conv_bstat.hip, tile id 33)
bool conv_bs_ok(const ConvArgs& a, int precision);
td_status conv_bs_launch(const ConvArgs& a, int precision, hipStream_t stream);
bool conv_plane_ok(const ConvArgs& a, int precision);       // tile ids 18-20 apply to this launch
td_status wino_gemm_launch(const ConvArgs& a, hipStream_t stream);     // Winograd plane contractions, input transform fused (fp32)

// ---- fused bottleneck tail (bottleneck.hip): 3x3 (mid -> mid) + BN + ReLU, then 1x1 (mid -> 4 mid) + BN + shortcut + ReLU ----
struct TailArgs {
    const void* x;        // NHWC [B,H,W,MID]: output of the block's first 1x1
    const void* w2;       // [MID][3][3][MID]
    const float* scale2;  // [MID] FrozenBN fold of conv2 (nullptr = 1 / 0)
    const float* bias2;
    const void* w3;       // [COUT][MID]
    const float* scale3;  // [COUT]
    const float* bias3;
    const void* res;      // NHWC [B,H,W,COUT]: the block's shortcut (same size)
    void* y;              // NHWC [B,H,W,COUT]
    int B, H, W, MID, COUT, M;
};
bool bottleneck_tail_ok(int precision, int mid, int cout);          // shapes the fused kernel is built for
td_status bottleneck_tail_launch(const TailArgs& a, int precision, hipStream_t stream);

// ---- Winograd F(2x2,3x3) transforms (winograd.hip; fp32 engine) ---------------------------------
// tiles [t0, t0 + Ts) of the layer ("slab"): V / Mb hold 16 planes of [Ts][C]
td_status wino_input_launch(const float* x, int B, int H, int W, int C, float* V, const int* m_dyn, int m_mul, long long t0, int Ts,
                            hipStream_t s);
td_status wino_output_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                             float* y, const int* m_dyn, int m_mul, long long t0, int Ts, hipStream_t s);
void wino_filter_transform(const float* w_ohwi, int N, int C, float* U);     // host: U [16][N][C]
// F(4x4,3x3): whole layers only; V / Mb hold 36 planes of [T][C], T = B * ceil(H/4) * ceil(W/4)
// (m_dyn: optional device-side image count <= B, the mask head's live RoIs)
td_status wino43_input_launch(const float* x, int B, int H, int W, int C, float* V, const int* m_dyn, hipStream_t s);
// F(4x4,3x3) contraction + output transform in ONE launch (wino_fused.hip): V [36][T][C] x U [36][N][C] → y, the M planes never
// reach memory. Another association of the transform sums than wino43_output_launch: a fixed rule picks the layers (engine.cpp).
bool wino43_fused_ok(int B, int H, int W, int C, int N);
td_status wino43_fused_launch(const float* V, const float* U, int B, int H, int W, int C, int N, const float* scale, const float* bias,
                              int relu, float* y, const int* m_dyn, hipStream_t s);
td_status wino43_output_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                               float* y, const int* m_dyn, hipStream_t s);
// the same for a 256-channel layer whose only consumer is a 1x1 head [head_n <= 32][256]: head_y [B*H*W][head_n] = y . head_w^T + head_b,
// y itself is not written (bit-identical to wino43_output_launch followed by the head as a conv2d_launch of its own)
td_status wino43_output_head_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                                    const float* head_w, const float* head_b, float* head_y, int head_n, hipStream_t s);
void wino43_filter_transform(const float* w_ohwi, int N, int C, float* U);   // host: U [36][N][C]

// ---- stem / pooling / resize (stem.hip) ---------------------------------------------------------
struct ImgSizes {           // per-image valid sizes, passed by value (B <= TD_MAX_BATCH)
    int h[64];
    int w[64];
};
#define TD_MAX_BATCH 64
// w16 / bias16 (optional, from stem_mfma_prepare): the fp16 engine's uint8 inputs with 64 channels take the MFMA stem
td_status stem_launch(const void* images, int input_format, const ImgSizes& valid, int B, int Hp, int Wp,
                      const float* w_kc /*[147][cout]*/, const float* scale, const float* bias, void* y,
                      int cout, int precision, hipStream_t stream, const void* w16 = nullptr, const float* bias16 = nullptr);
void stem_mfma_prepare(const float* w_kc, const float* scale, const float* bias, int cout, std::vector<unsigned short>& w16_bits,
                       std::vector<float>& bias16);       // w16_bits: IEEE half bit patterns (this header is also compiled by g++)
td_status maxpool3x3s2_launch(const void* x, void* y, int B, int H, int W, int C, int precision, hipStream_t stream);
td_status subsample2_launch(const void* x, void* y, int B, int H, int W, int C, int precision, hipStream_t stream);
td_status resize_batch_u8_launch(const uint8_t* const* srcs, int n, int h, int w, int c, uint8_t* dst, int out_h,
                                 int out_w, int dst_pitch_px, size_t dst_img_bytes, void* tmp, hipStream_t stream);
td_status resize_bilinear_f64_launch(const double* src, int c, int h, int w, float* dst, int out_h, int out_w, int dst_pitch,
                                     long long dst_plane, hipStream_t stream);
td_status resize_tile_u8_launch(const uint8_t* src, int h, int w, int c, uint8_t* dst, int out_h, int out_w,
                                int dst_pitch_px, void* tmp, hipStream_t stream);
