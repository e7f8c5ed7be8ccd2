// Input side of the forward: Pillow-exact tile resize, normalise + 7x7/s2 stem conv + FrozenBN + ReLU,
// 3x3/s2 max-pool, and the stride-2 subsample that makes p6.
//
// Reference: Predictor._process_tile (TreeDetection/prediction.py:159-176: bands (2,1,0) = BGR, PIL
// bilinear via ResizeShortestEdge for uint8 tiles) and detectron2's preprocess_image + BasicStem
// (SURVEY.md Appendix A items 1-3). These are HBM-bound byte / float streams: coalesced 16-B
// accesses, LDS-staged input patches, no MFMA (Cin = 3 is not GEMM-shaped).
#include "common.h"
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int PIL_PRECISION_BITS = 32 - 8 - 2;

// ---- Pillow resample coefficients (libImaging Resample.c precompute_coeffs, bilinear filter) ----
struct CoeffTable {
    int ksize = 0;
    int* d_bounds = nullptr;   // [out] xmin
    int* d_kk = nullptr;       // [out][ksize] fixed-point weights
};

static std::mutex g_coeff_mu;
static std::map<std::pair<long long, int>, CoeffTable> g_coeff_cache;   // ((in<<32|out), device) -> table

static void host_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign(out_size, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) k[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        bounds[xx] = xmin;
        for (int x = 0; x < ksize; ++x) {
            const double v = k[x] * (double)(1 << PIL_PRECISION_BITS);
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
    }
}

static td_status get_coeffs(int in_size, int out_size, CoeffTable& out) {
    int device = 0;
    TD_HIP_CHECK(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_coeff_mu);
    const auto key = std::make_pair(((long long)in_size << 32) | (unsigned)out_size, device);
    auto it = g_coeff_cache.find(key);
    if (it != g_coeff_cache.end()) {
        out = it->second;
        return TD_OK;
    }
    std::vector<int> bounds, kk;
    CoeffTable t;
    host_coeffs(in_size, out_size, bounds, kk, t.ksize);
    TD_HIP_CHECK(hipMalloc(&t.d_bounds, bounds.size() * sizeof(int)));
    TD_HIP_CHECK(hipMalloc(&t.d_kk, kk.size() * sizeof(int)));
    TD_HIP_CHECK(hipMemcpy(t.d_bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice));
    TD_HIP_CHECK(hipMemcpy(t.d_kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice));
    g_coeff_cache[key] = t;
    out = t;
    return TD_OK;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct TilePtrs {             // device pointers of the tiles of one batch, passed by value
    const uint8_t* p[TD_MAX_BATCH];
};

// horizontal pass + band pick: tmp[y][x][j] = sum_t src[y][xmin+t][2-j] * kk[x][t]      (blockIdx.z = tile)
// The taps of an output pixel are ksize CONTIGUOUS source pixels: for 3-band tiles and ksize <= 6 their bytes are fetched
// as five aligned dwords (20 bytes cover the 18 + 3 of any alignment... up to ksize 5: 15 + 3) and picked apart with
// shifts — a third of the load instructions of the byte-wise loop (these two passes moved 40 MB in 0.2 ms: instruction-
// bound, one byte per lane and load). Same integer arithmetic, same bytes.
__global__ void resize_h_u8(TilePtrs srcs, int h, int w, int c, uint8_t* __restrict__ tmp_all,
                            int out_w, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= out_w) return;
    const uint8_t* __restrict__ src = srcs.p[blockIdx.z];
    uint8_t* __restrict__ tmp = tmp_all + (size_t)blockIdx.z * h * out_w * 3;
    const int xmin = bounds[x];
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint8_t* row = src + (size_t)y * w * c;
    const size_t first = (size_t)(row - src) + (size_t)xmin * 3;                 // byte offset of the first tap inside the tile
    const size_t tile_bytes = (size_t)h * w * 3;
    if (c == 3 && ksize <= 5 && xmin + ksize <= w && (reinterpret_cast<size_t>(src) & 3) == 0 && (first & ~(size_t)3) + 20 <= tile_bytes) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + (first & ~(size_t)3));
        uint32_t d[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) d[i] = q[i];
        const unsigned sh = (unsigned)(first & 3);
        uint32_t e[4];                                          // the 16 bytes from the first tap byte on (v_alignbyte: a funnel shift by whole bytes)
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
        auto byte_at = [&](int i) -> int { return (int)((e[i >> 2] >> ((i & 3) * 8)) & 0xffu); };      // i: compile-time after unrolling
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (t < ksize) {
                const int k = kk[x * ksize + t];
                s0 += byte_at(3 * t + 2) * k;
                s1 += byte_at(3 * t + 1) * k;
                s2 += byte_at(3 * t + 0) * k;
            }
        }
    } else {
        for (int t = 0; t < ksize; ++t) {
            const int k = kk[x * ksize + t];
            int xs = xmin + t;
            xs = xs < w ? xs : w - 1;      // weights beyond xmax are 0
            const uint8_t* p = row + (size_t)xs * c;
            s0 += p[2] * k;
            s1 += p[1] * k;
            s2 += p[0] * k;
        }
    }
    uint8_t* o = tmp + ((size_t)y * out_w + x) * 3;
    o[0] = clip8(s0);
    o[1] = clip8(s1);
    o[2] = clip8(s2);
}

// vertical pass over the 3-channel intermediate      (blockIdx.z = tile; images dst_img_bytes apart)
// Four byte columns per thread (one aligned dword per tap row, one dword store) where the row pitches allow it.
__global__ void resize_v_u8(const uint8_t* __restrict__ tmp_all, int h, int row_bytes, uint8_t* __restrict__ dst_all,
                            int out_h, int dst_pitch_bytes, size_t dst_img_bytes, const int* __restrict__ bounds,
                            const int* __restrict__ kk, int ksize, int vec4) {
    const int y = blockIdx.y;
    const uint8_t* __restrict__ tmp = tmp_all + (size_t)blockIdx.z * h * row_bytes;
    uint8_t* __restrict__ dst = dst_all + (size_t)blockIdx.z * dst_img_bytes;
    const int ymin = bounds[y];
    if (vec4) {
        const int xb = (blockIdx.x * blockDim.x + threadIdx.x) * 4;     // first of four byte columns
        if (xb >= row_bytes) return;
        int s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = 1 << (PIL_PRECISION_BITS - 1);
        for (int t = 0; t < ksize; ++t) {
            int ys = ymin + t;
            ys = ys < h ? ys : h - 1;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(tmp + (size_t)ys * row_bytes + xb);
            const int k = kk[y * ksize + t];
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += (int)((v >> (8 * j)) & 0xffu) * k;
        }
        const uint32_t o = (uint32_t)clip8(s[0]) | ((uint32_t)clip8(s[1]) << 8) | ((uint32_t)clip8(s[2]) << 16) | ((uint32_t)clip8(s[3]) << 24);
        *reinterpret_cast<uint32_t*>(dst + (size_t)y * dst_pitch_bytes + xb) = o;
        return;
    }
    const int xb = blockIdx.x * blockDim.x + threadIdx.x;   // byte column (x*3 + j)
    if (xb >= row_bytes) return;
    int s = 1 << (PIL_PRECISION_BITS - 1);
    for (int t = 0; t < ksize; ++t) {
        int ys = ymin + t;
        ys = ys < h ? ys : h - 1;
        s += tmp[(size_t)ys * row_bytes + xb] * kk[y * ksize + t];
    }
    dst[(size_t)y * dst_pitch_bytes + xb] = clip8(s);
}

// ---- stem: (x - mean) → conv 7x7 / s2 / p3 (3 → 64) → *scale + bias → ReLU -------------------------
// One block = 16x16 output pixels; the 37x37x3 normalised input patch sits in LDS; each thread owns
// one pixel and all 64 output channels (weights are wave-uniform → scalar loads).
constexpr int ST = 16;                 // output tile side
constexpr int SP = 2 * ST + 5;         // input patch side (37)
constexpr int SPW = SP * 3 + 1;        // patch row stride in floats (padded)

template <int FORMAT, int COUT, typename TO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const void* __restrict__ images, ImgSizes valid, int Hp, int Wp,
                                                        const float* __restrict__ w_kc, const float* __restrict__ scale,
                                                        const float* __restrict__ bias, TO* __restrict__ y) {
    __shared__ float patch[SP * SPW];
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    const int Ho = Hp >> 1, Wo = Wp >> 1;
    const int vh = valid.h[b], vw = valid.w[b];
    const float mean[3] = {103.530f, 116.280f, 123.675f};
    for (int i = threadIdx.x; i < SP * SP * 3; i += 256) {
        const int py = i / (SP * 3);
        const int r = i - py * (SP * 3);
        const int px = r / 3, c = r - px * 3;
        const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + px;
        float v = 0.f;
        if (iy >= 0 && iy < vh && ix >= 0 && ix < vw) {
            if (FORMAT == TD_INPUT_U8_HWC)
                v = (float)static_cast<const uint8_t*>(images)[((size_t)(b * Hp + iy) * Wp + ix) * 3 + c];
            else
                v = static_cast<const float*>(images)[((size_t)(b * 3 + c) * Hp + iy) * Wp + ix];
            v = __fsub_rn(v, mean[c]);
        }
        patch[py * SPW + r] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x & (ST - 1), ty = threadIdx.x >> 4;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
        const float* prow = &patch[(2 * ty + ky) * SPW + 2 * tx * 3];
        for (int kx = 0; kx < 7; ++kx) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = prow[kx * 3 + c];
                const float* wk = w_kc + (size_t)((ky * 7 + kx) * 3 + c) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = __fmaf_rn(a, wk[co], acc[co]);
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < Ho && ox < Wo) {
        TO* o = y + ((size_t)(b * Ho + oy) * Wo + ox) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; co += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float u = __fadd_rn(__fmul_rn(acc[co + q], scale[co + q]), bias[co + q]);
                t[q] = u > 0.f ? u : 0.f;
            }
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(o + co) = make_float4(t[0], t[1], t[2], t[3]);
                *reinterpret_cast<float4*>(o + co + 4) = make_float4(t[4], t[5], t[6], t[7]);
            } else {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                h8 h;
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = (_Float16)t[q];
                *reinterpret_cast<h8*>(o + co) = h;
            }
        }
    }
}

// ---- fp16 engine, uint8 input: the stem on the matrix cores ---------------------------------------------------------
// The VALU kernel above spends 0.27 ms per 8-tile batch on 24 GFLOP (5-6 % of the fp16 step). Here the taps are the k
// axis of a contraction with A = the RAW pixels (0..255 are exact in fp16) and Wt = the fp16-rounded filters; the mean
// subtraction moves into the bias (conv(x - mean) = conv(x) - sum_k w_k mean_c(k), folded on the host with the SAME
// rounded filters), and a tap in the zero padding — (x - mean) = 0 — becomes the value mean_c in fp16 (103.5 / 116.25 /
// 123.6875 for 103.53 / 116.28 / 123.675: an error of <= 0.03 |w| on the 3-pixel border ring only).
// k is laid out so that NO im2col image is needed: k' = ky * 32 + kx * 4 + c with kx in 0..7 and c in 0..3 (kx = 7, c = 3
// and ky = 7 are dummies with zero filters): 256 = four 128-B chunks of halves. The input patch sits in LDS as fp16 BGR0
// pixels (8 B each), so the 16-B MFMA fragment of output pixel (ty, tx) for (ky, two neighbouring kx) is simply the two
// patch pixels (2 ty + ky, 2 tx + kx), (.., + 1): one ds_read_b128 straight from the patch — the first version of this
// kernel built a 48-KB A image per 128 pixels and lost to the VALU kernel (0.34 vs 0.27 ms).
// A block is resident and walks 8 x 16-pixel tiles (the 32-KB filter image is loaded once per block); 4 waves as 2 x 2
// run 2 x 1 MFMA tiles (v_mfma_f32_32x32x16_f16) over the 16 k-steps; scale / bias' / ReLU / the fp16 rounding finish in
// registers and the tile leaves through LDS as whole 128-B pixel rows.
typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sh8 __attribute__((ext_vector_type(8)));
typedef _Float16 sh4 __attribute__((ext_vector_type(4)));
constexpr int SM_TY = 8, SM_TX = 16;                       // output tile
constexpr int SM_PH = 2 * SM_TY + 6, SM_PW = 2 * SM_TX + 6;   // 22 x 38 patch pixels (one dummy row / column for ky = 7 / kx = 7)
constexpr int SM_K = 256, SM_CH = 4;                       // padded k, chunks of 64 halves

__global__ __launch_bounds__(256) void stem_mfma_kernel(const uint8_t* __restrict__ images, ImgSizes valid, int B, int Hp, int Wp,
                                                        const _Float16* __restrict__ w16 /*[64][256]*/,
                                                        const float* __restrict__ scale, const float* __restrict__ bias16,
                                                        _Float16* __restrict__ y) {
    constexpr int W_BYTES = SM_CH * 64 * 128, P_BYTES = SM_PH * SM_PW * 8, O_STRIDE = 72;
    __shared__ __attribute__((aligned(16))) char lds[W_BYTES + P_BYTES + 128 * O_STRIDE * 2];
    char* Ws = lds;                                        // [4][64 rows][128 B], XOR-swizzled pieces
    char* Ps = lds + W_BYTES;                              // [22][38] pixels x {B, G, R, 0} halves
    _Float16* Os = reinterpret_cast<_Float16*>(lds + W_BYTES + P_BYTES);   // [128 pixels][64 + 8] halves
    const int Ho = Hp >> 1, Wo = Wp >> 1;
    const int tiles_x = (Wo + SM_TX - 1) / SM_TX, tiles_y = (Ho + SM_TY - 1) / SM_TY;
    const int total = tiles_x * tiles_y * B;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;               // 2 x 2 waves: pixels 64 wm .., channels 32 wn ..

    // filter image, once per block: 4 chunks x 64 rows x 8 pieces of 16 B, swizzled like the conv kernels' B rows
    {
        constexpr int W_IT = SM_CH * 64 * 8 / 256;
        sf32x4 wv[W_IT];
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int i = tid + 256 * j;
            wv[j] = *reinterpret_cast<const sf32x4*>(w16 + (size_t)((i >> 3) & 63) * SM_K + (i >> 9) * 64 + (i & 7) * 8);
        }
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int i = tid + 256 * j;
            const int pc = i & 7, n = (i >> 3) & 63, c = i >> 9;
            *reinterpret_cast<sf32x4*>(Ws + (c * 64 + n) * 128 + ((pc ^ ((n >> 1) & 7)) * 16)) = wv[j];
        }
    }
    const int n_col = wn * 32 + (lane & 31);
    const float sc = scale[n_col], bi = bias16[n_col];
    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    const _Float16 m0 = (_Float16)103.530f, m1 = (_Float16)116.280f, m2 = (_Float16)123.675f;
    // this lane's two output pixels (MFMA rows lane & 31 of the wave's two 32-row tiles) inside the tile
    int pix_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + (lane & 31);
        pix_off[i] = ((2 * (m >> 4)) * SM_PW + 2 * (m & 15)) * 8;
    }

    // patch of tile t: 22 x 38 pixels, 3 bytes each → fp16 BGR0 in registers; outside the image: fp16(mean)
    constexpr int P_IT = (SM_PH * SM_PW + 255) / 256;
    sh4 pv[P_IT];
    auto load_patch = [&](int t) {
        const int b = t / (tiles_x * tiles_y), r0 = t - b * (tiles_x * tiles_y);
        const int iy_base = 2 * (r0 / tiles_x) * SM_TY - 3, ix_base = 2 * (r0 - (r0 / tiles_x) * tiles_x) * SM_TX - 3;
        const int vh = valid.h[b], vw = valid.w[b];
#pragma unroll
        for (int j = 0; j < P_IT; ++j) {
            const int i = tid + 256 * j;
            const int py = i / SM_PW, px = i - py * SM_PW;
            const int iy = iy_base + py, ix = ix_base + px;
            sh4 v = {m0, m1, m2, (_Float16)0.f};
            if (i < SM_PH * SM_PW && iy >= 0 && iy < vh && ix >= 0 && ix < vw) {
                const uint8_t* q = images + ((size_t)(b * Hp + iy) * Wp + ix) * 3;
                v[0] = (_Float16)(float)q[0];
                v[1] = (_Float16)(float)q[1];
                v[2] = (_Float16)(float)q[2];
            }
            pv[j] = v;
        }
    };
    load_patch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y), r0 = t - b * (tiles_x * tiles_y);
        const int oy0 = (r0 / tiles_x) * SM_TY, ox0 = (r0 - (r0 / tiles_x) * tiles_x) * SM_TX;
        __syncthreads();                                   // the previous tile's fragment reads and output copies are done
#pragma unroll
        for (int j = 0; j < P_IT; ++j) {
            const int i = tid + 256 * j;
            if (i < SM_PH * SM_PW) *reinterpret_cast<sh4*>(Ps + i * 8) = pv[j];
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) load_patch(t + gridDim.x);      // the next tile's bytes travel under this tile's MFMAs and stores

        sf32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
        for (int c = 0; c < SM_CH; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // k' = 64 c + 16 kk + 8 hi .. + 8  =  filter row ky = 2 c + (kk >> 1), pixels kx = 2 (2 (kk & 1) + hi), + 1
                const sf32x4 fb = *reinterpret_cast<const sf32x4*>(Ws + (c * 64 + wn * 32 + (lane & 31)) * 128 + ((((unsigned)(2 * kk) + hi) ^ swz) * 16));
                const int tap = ((2 * c + (kk >> 1)) * SM_PW + 2 * (2 * (kk & 1) + (int)hi)) * 8;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const sf32x4 fa = *reinterpret_cast<const sf32x4*>(Ps + pix_off[i] + tap);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sh8, fa), __builtin_bit_cast(sh8, fb), acc[i], 0, 0, 0);
                }
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const float u = __fadd_rn(__fmul_rn(acc[i][q], sc), bi);
                Os[row * O_STRIDE + n_col] = (_Float16)(u > 0.f ? u : 0.f);
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const int m = i >> 3, pc = i & 7;
            const int oy = oy0 + (m >> 4), ox = ox0 + (m & 15);
            if (oy < Ho && ox < Wo)
                *reinterpret_cast<sf32x4*>(y + ((size_t)(b * Ho + oy) * Wo + ox) * 64 + pc * 8) = *reinterpret_cast<const sf32x4*>(Os + m * O_STRIDE + pc * 8);
        }
    }
}

// ---- max-pool 3x3 / s2 / p1 over NHWC (16 bytes of channels per thread: 4 floats or 8 halves) -------------------------
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};
template <>
struct Vec16<_Float16> {
    typedef _Float16 type __attribute__((ext_vector_type(8)));
    static constexpr int N = 8;
};

template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho,
                                    int Wo) {
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int cv = C / N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo * cv;
    if (idx >= total) return;
    const int c = (int)(idx % cv);
    size_t p = idx / cv;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    V m;
#pragma unroll
    for (int q = 0; q < N; ++q) m[q] = (T)(-65504.f);     // every window holds at least one real pixel
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = 2 * oy + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = 2 * ox + dx;
            if (ix < 0 || ix >= W) continue;
            const V v = *reinterpret_cast<const V*>(x + ((size_t)(b * H + iy) * W + ix) * C + c * N);
#pragma unroll
            for (int q = 0; q < N; ++q) m[q] = v[q] > m[q] ? v[q] : m[q];
        }
    }
    *reinterpret_cast<V*>(y + ((size_t)(b * Ho + oy) * Wo + ox) * C + c * N) = m;
}

// ---- p6 = max_pool2d(p5, kernel 1, stride 2) = p5[::2, ::2] ------------------------------------------
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho,
                                  int Wo) {
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    const int cv = C / N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * Ho * Wo * cv;
    if (idx >= total) return;
    const int c = (int)(idx % cv);
    size_t p = idx / cv;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    *reinterpret_cast<V*>(y + ((size_t)(b * Ho + oy) * Wo + ox) * C + c * N) =
        *reinterpret_cast<const V*>(x + ((size_t)(b * H + 2 * oy) * W + 2 * ox) * C + c * N);
}

}  // namespace

td_status resize_batch_u8_launch(const uint8_t* const* srcs, int n, int h, int w, int c, uint8_t* dst, int out_h,
                                 int out_w, int dst_pitch_px, size_t dst_img_bytes, void* tmp, hipStream_t stream) {
    TD_REQUIRE(srcs && dst && tmp && n >= 1 && n <= TD_MAX_BATCH, "resize: bad batch");
    TD_REQUIRE(h > 0 && w > 0 && c >= 3 && out_h > 0 && out_w > 0 && dst_pitch_px >= out_w, "resize: bad geometry");
    CoeffTable th, tv;
    td_status st = get_coeffs(w, out_w, th);
    if (st < 0) return st;
    st = get_coeffs(h, out_h, tv);
    if (st < 0) return st;
    TilePtrs tp{};
    for (int i = 0; i < n; ++i) {
        TD_REQUIRE(srcs[i], "resize: null tile pointer");
        tp.p[i] = srcs[i];
    }
    hipLaunchKernelGGL(resize_h_u8, dim3(td_cdiv(out_w, 256), h, n), dim3(256), 0, stream, tp, h, w, c,
                       static_cast<uint8_t*>(tmp), out_w, th.d_bounds, th.d_kk, th.ksize);
    TD_KERNEL_CHECK();
    // four byte columns per thread when every row of the intermediate and of the destination starts on a dword
    const int vec4 = ((out_w * 3) % 4 == 0 && (dst_pitch_px * 3) % 4 == 0 && dst_img_bytes % 4 == 0 && ((size_t)h * out_w * 3) % 4 == 0 &&
                      (reinterpret_cast<size_t>(tmp) & 3) == 0 && (reinterpret_cast<size_t>(dst) & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL(resize_v_u8, dim3(td_cdiv(vec4 ? td_cdiv(out_w * 3, 4) : out_w * 3, 256), out_h, n), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(tmp), h, out_w * 3, dst, out_h, dst_pitch_px * 3, dst_img_bytes,
                       tv.d_bounds, tv.d_kk, tv.ksize, vec4);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status resize_tile_u8_launch(const uint8_t* src, int h, int w, int c, uint8_t* dst, int out_h, int out_w,
                                int dst_pitch_px, void* tmp, hipStream_t stream) {
    return resize_batch_u8_launch(&src, 1, h, w, c, dst, out_h, out_w, dst_pitch_px, 0, tmp, stream);
}

// ---- float resize of the non-uint8 rasters (reference prediction.py:167-169) -----------------------------------------
// 16-bit / float tiles do not take Pillow's 8-bit filter: detectron2's ResizeTransform.apply_image hands every other dtype to
// torch.nn.functional.interpolate(mode="bilinear", align_corners=False). Restated here operation for operation in the
// input's precision (float64: the reference's 255 * x / 65535 makes the tile a float64 array): scale = in / out,
// src = max(scale * (dst + 0.5) - 0.5, 0), i1 = (int)src, lambda1 = src - i1, lambda0 = 1 - lambda1, neighbour index
// clamped at the border, value = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11) — one IEEE operation per step
// (-ffp-contract=off), then ONE rounding to float32 (the reference's .astype("float32")). HBM-bound: 8 B in, 4 B out.
__global__ __launch_bounds__(256) void resize_bilinear_f64_kernel(const double* __restrict__ src, int C, int h, int w, float* __restrict__ dst,
                                                                   int out_h, int out_w, int dst_pitch, long long dst_plane) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (ox >= out_w || oy >= out_h) return;
    const double sh = (double)h / (double)out_h, sw = (double)w / (double)out_w;
    double fy = sh * ((double)oy + 0.5) - 0.5;
    double fx = sw * ((double)ox + 0.5) - 0.5;
    fy = fy < 0.0 ? 0.0 : fy;
    fx = fx < 0.0 ? 0.0 : fx;
    const int y1 = (int)fy, x1 = (int)fx;
    const int yp = y1 < h - 1 ? 1 : 0, xp = x1 < w - 1 ? 1 : 0;
    const double h1 = fy - (double)y1, h0 = 1.0 - h1;
    const double w1 = fx - (double)x1, w0 = 1.0 - w1;
    for (int c = 0; c < C; ++c) {
        const double* p = src + ((size_t)c * h + y1) * w + x1;
        const double v00 = p[0], v01 = p[xp], v10 = p[(size_t)yp * w], v11 = p[(size_t)yp * w + xp];
        const double top = __dadd_rn(__dmul_rn(w0, v00), __dmul_rn(w1, v01));
        const double bot = __dadd_rn(__dmul_rn(w0, v10), __dmul_rn(w1, v11));
        dst[(size_t)c * dst_plane + (size_t)oy * dst_pitch + ox] = (float)__dadd_rn(__dmul_rn(h0, top), __dmul_rn(h1, bot));
    }
}

td_status resize_bilinear_f64_launch(const double* src, int c, int h, int w, float* dst, int out_h, int out_w, int dst_pitch,
                                     long long dst_plane, hipStream_t stream) {
    TD_REQUIRE(src && dst && c >= 1 && h >= 1 && w >= 1 && out_h >= 1 && out_w >= 1 && dst_pitch >= out_w && dst_plane >= (long long)out_h * dst_pitch,
               "resize_bilinear_f64: bad geometry (%d x %d x %d -> %d x %d, pitch %d)", c, h, w, out_h, out_w, dst_pitch);
    hipLaunchKernelGGL(resize_bilinear_f64_kernel, dim3(td_cdiv(out_w, 256), out_h), dim3(256), 0, stream, src, c, h, w, dst, out_h, out_w,
                       dst_pitch, dst_plane);
    TD_KERNEL_CHECK();
    return TD_OK;
}

// host: the fp16 filter image [cout][192] (k = (ky, kx, c), zero-padded) of stem_mfma_kernel and the bias with the mean
// subtraction folded in: bias' = bias - scale * sum_k w16[n][k] * mean_c(k), from the ROUNDED filters (float64, rounded once)
void stem_mfma_prepare(const float* w_kc /*[147][cout]*/, const float* scale, const float* bias, int cout,
                       std::vector<unsigned short>& w16, std::vector<float>& bias16) {
    static const double mean[3] = {103.530, 116.280, 123.675};
    w16.assign((size_t)cout * SM_K, (unsigned short)0);      // k' = ky * 32 + kx * 4 + c; kx = 7, c = 3, ky = 7 stay zero
    bias16.assign(cout, 0.f);
    for (int n = 0; n < cout; ++n) {
        double corr = 0.0;
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx)
                for (int c = 0; c < 3; ++c) {
                    const _Float16 h = (_Float16)w_kc[(size_t)((ky * 7 + kx) * 3 + c) * cout + n];
                    w16[(size_t)n * SM_K + ky * 32 + kx * 4 + c] = __builtin_bit_cast(unsigned short, h);
                    corr += (double)(float)h * mean[c];
                }
        bias16[n] = (float)((double)bias[n] - (double)scale[n] * corr);
    }
}

td_status stem_launch(const void* images, int input_format, const ImgSizes& valid, int B, int Hp, int Wp,
                      const float* w_kc, const float* scale, const float* bias, void* y, int cout, int precision,
                      hipStream_t stream, const void* w16, const float* bias16) {
    TD_REQUIRE(Hp % 2 == 0 && Wp % 2 == 0 && B <= TD_MAX_BATCH, "stem: bad geometry");
    if (w16 && bias16 && precision == TD_PRECISION_FP16 && input_format == TD_INPUT_U8_HWC && cout == 64) {
        const int total = td_cdiv(Wp / 2, SM_TX) * td_cdiv(Hp / 2, SM_TY) * B;
        const int cap = 256 * 3;                               // resident blocks (57 KB of LDS each)
        hipLaunchKernelGGL(stem_mfma_kernel, dim3(total < cap ? total : cap), dim3(256), 0, stream, static_cast<const uint8_t*>(images),
                           valid, B, Hp, Wp, static_cast<const _Float16*>(w16), scale, bias16, static_cast<_Float16*>(y));
        TD_KERNEL_CHECK();
        return TD_OK;
    }
    const dim3 grid(td_cdiv(Wp / 2, ST), td_cdiv(Hp / 2, ST), B);
    const bool u8 = input_format == TD_INPUT_U8_HWC, h = precision == TD_PRECISION_FP16;
#define TD_STEM_CASE(F, C, TO)                                                                                       \
    hipLaunchKernelGGL((stem_conv_kernel<F, C, TO>), grid, dim3(256), 0, stream, images, valid, Hp, Wp, w_kc, scale, \
                       bias, static_cast<TO*>(y))
#define TD_STEM_FMT(C)                                                             \
    do {                                                                           \
        if (u8 && h) TD_STEM_CASE(TD_INPUT_U8_HWC, C, _Float16);                   \
        else if (u8) TD_STEM_CASE(TD_INPUT_U8_HWC, C, float);                      \
        else if (h) TD_STEM_CASE(TD_INPUT_F32_CHW, C, _Float16);                   \
        else TD_STEM_CASE(TD_INPUT_F32_CHW, C, float);                             \
    } while (0)
    if (cout == 64) TD_STEM_FMT(64);
    else if (cout == 32) TD_STEM_FMT(32);
    else {
        td_set_error("stem: %d output channels not built (32 or 64)", cout);
        return TD_ERR_INVALID;
    }
#undef TD_STEM_FMT
#undef TD_STEM_CASE
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status maxpool3x3s2_launch(const void* x, void* y, int B, int H, int W, int C, int precision, hipStream_t stream) {
    TD_REQUIRE(C % 8 == 0, "maxpool: C must be a multiple of 8");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (precision == TD_PRECISION_FP16) {
        const size_t total = (size_t)B * Ho * Wo * (C / 8);
        hipLaunchKernelGGL((maxpool3x3s2_kernel<_Float16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const _Float16*>(x), static_cast<_Float16*>(y), B, H, W, C, Ho, Wo);
    } else {
        const size_t total = (size_t)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL((maxpool3x3s2_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const float*>(x), static_cast<float*>(y), B, H, W, C, Ho, Wo);
    }
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status subsample2_launch(const void* x, void* y, int B, int H, int W, int C, int precision, hipStream_t stream) {
    TD_REQUIRE(C % 8 == 0, "subsample: C must be a multiple of 8");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (precision == TD_PRECISION_FP16) {
        const size_t total = (size_t)B * Ho * Wo * (C / 8);
        hipLaunchKernelGGL((subsample2_kernel<_Float16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const _Float16*>(x), static_cast<_Float16*>(y), B, H, W, C, Ho, Wo);
    } else {
        const size_t total = (size_t)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL((subsample2_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                           static_cast<const float*>(x), static_cast<float*>(y), B, H, W, C, Ho, Wo);
    }
    TD_KERNEL_CHECK();
    return TD_OK;
}
