// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the fp32 engine (FPN outputs, RPN conv, bottleneck conv2,
// mask-head convs — 67 % of the forward's multiplies; TreeDetection/prediction.py:183 → detectron2 Conv2d 3x3).
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs
// The element-wise products summed over the input channels are 16 independent contractions
//   M_xi[t][n] = sum_c V_xi[t][c] * U_xi[n][c]        xi = 0..15, t = output tile (b, y/2, x/2)
// with 4/9 of the direct convolution's multiplies; they run as ONE batched launch of conv_igemm_kernel (blockIdx.y =
// xi, 1x1 "convolution" over the plane V_xi), so the MFMA path, its tiles and its tuner are reused. V and M are four
// times the size of the layer's input / output: the engine therefore walks a layer in SLABS of a few thousand tiles
// (engine.cpp run_wino) whose V and M planes together stay inside the 256 MiB Infinity Cache — the transforms then hand
// over on-die and only the layer's own input and output cross HBM. This file holds the two transforms:
//   wino_input_kernel   x [B,H,W,C]      → V [16][T][C]      (32 add/sub per patch and channel)
//   wino_output_kernel  M [16][T][N]     → y [B,H,W,N] = act((A^T M A) * scale + bias)   (24 add/sub per tile and channel)
// Standard matrices (Lavin & Gray): B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
// A^T = [1 1 1 0; 0 1 -1 -1]. The filter transform U = G g G^T is done once on the host (engine.cpp, float64 → float32).
// Numerics: same products in a different association than the direct kernel; |error| ~ 1e-6 relative, inside the fp32
// parity tolerances (tests/test_conv_gpu.py, tests/test_engine_gpu.py state them).
#include "common.h"

namespace {

// one thread = one tile x 4 channels
// Tiles [t0, t0 + Ts) of the layer form one SLAB: its 16 planes are [16][Ts][C], indexed by the tile's position in the slab.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                         float* __restrict__ V, const int* __restrict__ m_dyn, int m_mul,
                                                         long long t0, int Ts) {
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    int nimg = B;
    if (m_dyn) {
        const int n = (int)(((long long)*m_dyn * m_mul) / ((long long)H * W));
        nimg = n < B ? n : B;
    }
    const int c4n = C >> 2;
    const long long T = Ts;                                           // plane stride: tiles of one slab
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long tl = idx / c4n;                                   // tile inside the slab
    const int c = (int)(idx - tl * c4n) * 4;
    const long long t = t0 + tl;
    if (tl >= Ts || t >= (long long)nimg * TH * TW) return;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int b = (int)(t / ((long long)TW * TH));
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = x0 + j;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                d[i][j] = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
            else
                d[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#define TD_SUB(a, b) make_float4(__fsub_rn(a.x, b.x), __fsub_rn(a.y, b.y), __fsub_rn(a.z, b.z), __fsub_rn(a.w, b.w))
#define TD_ADD(a, b) make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w))
    float4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // B^T d: combine rows
        r[0][j] = TD_SUB(d[0][j], d[2][j]);
        r[1][j] = TD_ADD(d[1][j], d[2][j]);
        r[2][j] = TD_SUB(d[2][j], d[1][j]);
        r[3][j] = TD_SUB(d[1][j], d[3][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {            // (B^T d) B: combine columns; plane xi = 4 i + j
        const float4 v0 = TD_SUB(r[i][0], r[i][2]);
        const float4 v1 = TD_ADD(r[i][1], r[i][2]);
        const float4 v2 = TD_SUB(r[i][2], r[i][1]);
        const float4 v3 = TD_SUB(r[i][1], r[i][3]);
        float* p = V + ((size_t)(4 * i) * T + tl) * C + c;
        *reinterpret_cast<float4*>(p) = v0;
        *reinterpret_cast<float4*>(p + (size_t)T * C) = v1;
        *reinterpret_cast<float4*>(p + (size_t)2 * T * C) = v2;
        *reinterpret_cast<float4*>(p + (size_t)3 * T * C) = v3;
    }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, int B, int H, int W, int N,
                                                          const float* __restrict__ scale, const float* __restrict__ bias,
                                                          int relu, float* __restrict__ y, const int* __restrict__ m_dyn,
                                                          int m_mul, long long t0, int Ts) {
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    int nimg = B;
    if (m_dyn) {
        const int n = (int)(((long long)*m_dyn * m_mul) / ((long long)H * W));
        nimg = n < B ? n : B;
    }
    const int c4n = N >> 2;
    const long long T = Ts;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long tl = idx / c4n;
    const int c = (int)(idx - tl * c4n) * 4;
    const long long t = t0 + tl;
    if (tl >= Ts || t >= (long long)nimg * TH * TW) return;
    const int tx = (int)(t % TW);
    const int ty = (int)((t / TW) % TH);
    const int b = (int)(t / ((long long)TW * TH));
    float4 m[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const float4*>(Mb + ((size_t)(4 * i + j) * T + tl) * N + c);
    float4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // A^T M
        s[0][j] = TD_ADD(TD_ADD(m[0][j], m[1][j]), m[2][j]);
        s[1][j] = TD_SUB(TD_SUB(m[1][j], m[2][j]), m[3][j]);
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
    if (bias) bi = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int yy = 2 * ty + i;
        if (yy >= H) continue;
        const float4 o0 = TD_ADD(TD_ADD(s[i][0], s[i][1]), s[i][2]);       // (A^T M) A
        const float4 o1 = TD_SUB(TD_SUB(s[i][1], s[i][2]), s[i][3]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int xx = 2 * tx + j;
            if (xx >= W) continue;
            float4 v = j ? o1 : o0;
            if (scale) v = make_float4(__fmul_rn(v.x, sc.x), __fmul_rn(v.y, sc.y), __fmul_rn(v.z, sc.z), __fmul_rn(v.w, sc.w));
            if (bias) v = TD_ADD(v, bi);
            if (relu) v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
            *reinterpret_cast<float4*>(y + (((size_t)b * H + yy) * W + xx) * N + c) = v;
        }
    }
#undef TD_SUB
#undef TD_ADD
}


// ---- F(4x4, 3x3): 6x6 input patches, 36 planes, 4x4 outputs per tile ----------------------------------------------------
// 36 / 16 = 2.25 multiplies per output instead of F(2x2)'s 4 and the direct kernel's 9: on the big maps, where the
// contraction is the cost, the fp32 MFMA work drops another 1.78x. Matrices (Lavin & Gray, points 0, +-1, +-2, inf):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// A V element combines up to 16 pixels, so the input transform cannot ride in the contraction's A staging as F(2x2)'s does
// (wino_gemm_kernel): three launches — x → V [36][T][C] (2.25x the input), 36 batched plane contractions through
// conv_igemm_kernel, M [36][T][N] → y. Numerics: the larger transform constants cost about one decimal digit
// (|error| ~ 1e-5 of max|y| at 256 channels against 6e-7 for F(2x2), measured against float64) — inside the fp32 parity
// tolerances; which layers take it is a fixed rule in engine.cpp (never a timing decision).
#define TD_SUB(a, b) make_float4(__fsub_rn(a.x, b.x), __fsub_rn(a.y, b.y), __fsub_rn(a.z, b.z), __fsub_rn(a.w, b.w))
#define TD_ADD(a, b) make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w))
#define TD_MUL(k, a) make_float4(__fmul_rn(k, a.x), __fmul_rn(k, a.y), __fmul_rn(k, a.z), __fmul_rn(k, a.w))

// 1-D input transform t = B^T d of six values (fixed association: part of the layer's defined rounding)
__device__ __forceinline__ void wino43_bt(const float4 (&d)[6], float4 (&t)[6]) {
    const float4 a = TD_SUB(d[4], TD_MUL(4.f, d[2]));       // d4 - 4 d2
    const float4 b = TD_SUB(d[3], TD_MUL(4.f, d[1]));       // d3 - 4 d1
    const float4 c = TD_SUB(d[4], d[2]);                    // d4 - d2
    const float4 e = TD_MUL(2.f, TD_SUB(d[3], d[1]));       // 2 (d3 - d1)
    t[0] = TD_ADD(TD_SUB(TD_MUL(4.f, d[0]), TD_MUL(5.f, d[2])), d[4]);
    t[1] = TD_ADD(a, b);
    t[2] = TD_SUB(a, b);
    t[3] = TD_ADD(c, e);
    t[4] = TD_SUB(c, e);
    t[5] = TD_ADD(TD_SUB(TD_MUL(4.f, d[1]), TD_MUL(5.f, d[3])), d[5]);
}

// one thread = one tile x 4 channels: x [B,H,W,C] → V [36][T][C], T = B * ceil(H/4) * ceil(W/4)
// (m_dyn: device-side image count — the mask head's live RoIs; planes keep the stride T of the full batch)
__global__ __launch_bounds__(256) void wino43_input_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                           float* __restrict__ V, long long T, const int* __restrict__ m_dyn) {
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int c4n = C >> 2;
    // 32-bit index arithmetic (the launcher checks T * C / 4 < 2^31): 64-bit divisions are ~5x the instructions
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned t = idx / (unsigned)c4n;
    const int c = (int)(idx - t * (unsigned)c4n) * 4;
    long long live = T;
    if (m_dyn) {
        const long long n = (long long)*m_dyn * TH * TW;
        live = n < T ? n : T;
    }
    if ((long long)t >= live) return;
    const unsigned tyx = t % (unsigned)(TW * TH);
    const int b = (int)(t / (unsigned)(TW * TH));
    const int ty = (int)(tyx / (unsigned)TW);
    const int tx = (int)(tyx - (unsigned)ty * (unsigned)TW);
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    float4 d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int yy = y0 + i;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = x0 + j;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                d[i][j] = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yy) * W + xx) * C + c);
            else
                d[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {            // B^T d: combine rows, column by column (in place)
        float4 col[6], r[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = d[i][j];
        wino43_bt(col, r);
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i][j] = r[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {            // (B^T d) B: combine columns; plane xi = 6 i + j
        float4 v[6];
        wino43_bt(d[i], v);
        float* p = V + ((size_t)(6 * i) * T + t) * C + c;
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(p + (size_t)j * T * C) = v[j];
    }
}

// 1-D output transform s = A^T m of six values
__device__ __forceinline__ void wino43_at(const float4 (&m)[6], float4 (&s)[4]) {
    const float4 p12 = TD_ADD(m[1], m[2]), m12 = TD_SUB(m[1], m[2]);
    const float4 p34 = TD_ADD(m[3], m[4]), m34 = TD_SUB(m[3], m[4]);
    s[0] = TD_ADD(TD_ADD(m[0], p12), p34);
    s[1] = TD_ADD(m12, TD_MUL(2.f, m34));
    s[2] = TD_ADD(p12, TD_MUL(4.f, p34));
    s[3] = TD_ADD(TD_ADD(m12, TD_MUL(8.f, m34)), m[5]);
}

// one thread = one tile x 4 channels: M [36][T][N] → y [B,H,W,N] = act((A^T M A) * scale + bias); partial tiles at the
// bottom / right border store only the pixels inside the map
__global__ __launch_bounds__(256) void wino43_output_kernel(const float* __restrict__ Mb, int B, int H, int W, int N,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            int relu, float* __restrict__ y, long long T,
                                                            const int* __restrict__ m_dyn) {
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int c4n = N >> 2;
    // 32-bit index arithmetic (the launcher checks T * C / 4 < 2^31): 64-bit divisions are ~5x the instructions
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned t = idx / (unsigned)c4n;
    const int c = (int)(idx - t * (unsigned)c4n) * 4;
    long long live = T;
    if (m_dyn) {
        const long long n = (long long)*m_dyn * TH * TW;
        live = n < T ? n : T;
    }
    if ((long long)t >= live) return;
    const unsigned tyx = t % (unsigned)(TW * TH);
    const int b = (int)(t / (unsigned)(TW * TH));
    const int ty = (int)(tyx / (unsigned)TW);
    const int tx = (int)(tyx - (unsigned)ty * (unsigned)TW);
    float4 s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {            // A^T M, column by column
        float4 col[6], r[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const float4*>(Mb + ((size_t)(6 * i + j) * T + t) * N + c);
        wino43_at(col, r);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = r[i];
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
    if (bias) bi = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = 4 * ty + i;
        float4 o[4];
        wino43_at(s[i], o);                  // (A^T M) A
        if (yy >= H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = 4 * tx + j;
            if (xx >= W) continue;
            float4 v = o[j];
            if (scale) v = make_float4(__fmul_rn(v.x, sc.x), __fmul_rn(v.y, sc.y), __fmul_rn(v.z, sc.z), __fmul_rn(v.w, sc.w));
            if (bias) v = TD_ADD(v, bi);
            if (relu) v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
            *reinterpret_cast<float4*>(y + (((size_t)b * H + yy) * W + xx) * N + c) = v;
        }
    }
}
// The same output transform for a 256-channel layer whose ONLY consumer is a 1x1 head (the RPN's 3x3 conv and its 15-row
// objectness / delta head): a block = 4 tiles x all 256 channels = 64 pixels, so the finished pixels go to LDS instead of
// memory and two waves contract them with the head filters on the matrix cores — chunk by chunk of 32 channels, k-sub-steps
// and MFMA order exactly as conv_igemm_kernel runs a 1x1 layer (bit-identical to the separate head launch on the stored
// tensor). Only head_y [B*H*W][hn] is written: the layer's own output (327 MB per 8 tiles at p2) never exists.
__global__ __launch_bounds__(256) void wino43_output_head_kernel(const float* __restrict__ Mb, int B, int H, int W,
                                                                 const float* __restrict__ scale, const float* __restrict__ bias,
                                                                 int relu, long long T, const float* __restrict__ head_w,
                                                                 const float* __restrict__ head_b, float* __restrict__ head_y, int hn) {
    // The 64 x 256 tile passes through LDS in two halves of 128 channels (lanes 0-31 of a wave own channels 0-127, lanes 32-63
    // the rest): 33 KB instead of 66 KB per block, four blocks per CU instead of two — the kernel is an HBM stream and needs
    // the waves in flight. The head's accumulator chain runs over the halves in order, so k stays ascending.
    constexpr int N = 256, NH = 128, RS = NH + 4;      // row stride: 4 banks of skew per pixel row (conflict-free 16-B fragment reads)
    __shared__ __attribute__((aligned(16))) float tile[64 * RS];
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v16f __attribute__((ext_vector_type(16)));
    const int TH = (H + 3) >> 2, TW = (W + 3) >> 2;
    const int lt = threadIdx.x >> 6, c = (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + lt;
    const bool live = t < T;
    float4 s[4][6];
    if (live) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 col[6], r[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) col[i] = *reinterpret_cast<const float4*>(Mb + ((size_t)(6 * i + j) * T + t) * N + c);
            wino43_at(col, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i][j] = r[i];
        }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) sc = *reinterpret_cast<const float4*>(scale + c);
    if (bias) bi = *reinterpret_cast<const float4*>(bias + c);
    float4 px[4][4];                                       // this thread's 16 finished pixels x 4 channels
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 o[4];
        if (live) wino43_at(s[i], o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);          // dead tiles / pixels past the map: rows nobody stores
            if (live) {
                v = o[j];
                if (scale) v = make_float4(__fmul_rn(v.x, sc.x), __fmul_rn(v.y, sc.y), __fmul_rn(v.z, sc.z), __fmul_rn(v.w, sc.w));
                if (bias) v = TD_ADD(v, bi);
                if (relu) v = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
            }
            px[i][j] = v;
        }
    }
    const int col = lane & 31, hi4 = (lane >> 5) * 4;
    const bool lcol = col < hn;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = &tile[((wave & 1) * 32 + col) * RS + hi4];
    const float* brow = head_w + (size_t)(lcol ? col : 0) * N + hi4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                          // the first half's fragment reads are done
        if ((c >= NH) == (half == 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&tile[(lt * 16 + i * 4 + j) * RS + (c - half * NH)]) = px[i][j];
        }
        __syncthreads();
        if (wave < 2) {                                     // two 32-pixel row tiles, one wave each (k stays in one accumulator chain)
#pragma unroll 2
            for (int ch = 0; ch < NH / 32; ++ch) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const v4f fa = *reinterpret_cast<const v4f*>(arow + ch * 32 + 8 * kk);
                    v4f fb = {0.f, 0.f, 0.f, 0.f};
                    if (lcol) fb = *reinterpret_cast<const v4f*>(brow + half * NH + ch * 32 + 8 * kk);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
                }
            }
        }
    }
    if (wave >= 2 || !lcol) return;
    const float hb = head_b ? head_b[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long long tt = (long long)blockIdx.x * 4 + (row >> 4);
        if (tt >= T) continue;
        const unsigned tyx = (unsigned)(tt % (unsigned)(TW * TH));
        const int b = (int)(tt / (unsigned)(TW * TH));
        const int ty = (int)(tyx / (unsigned)TW), tx = (int)(tyx - (unsigned)ty * (unsigned)TW);
        const int yy = 4 * ty + ((row >> 2) & 3), xx = 4 * tx + (row & 3);
        if (yy >= H || xx >= W) continue;
        head_y[(((size_t)b * H + yy) * W + xx) * hn + col] = head_b ? __fadd_rn(acc[r], hb) : acc[r];
    }
}
#undef TD_SUB
#undef TD_ADD
#undef TD_MUL

}  // namespace

td_status wino_input_launch(const float* x, int B, int H, int W, int C, float* V, const int* m_dyn, int m_mul, long long t0,
                            int Ts, hipStream_t s) {
    TD_REQUIRE(x && V && B >= 1 && H >= 1 && W >= 1 && C >= 4 && (C & 3) == 0 && Ts >= 1 && t0 >= 0, "winograd input transform: bad arguments");
    const long long threads = (long long)Ts * (C / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, B, H, W, C, V, m_dyn, m_mul, t0, Ts);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status wino_output_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                             float* y, const int* m_dyn, int m_mul, long long t0, int Ts, hipStream_t s) {
    TD_REQUIRE(Mb && y && B >= 1 && H >= 1 && W >= 1 && N >= 4 && (N & 3) == 0 && Ts >= 1 && t0 >= 0, "winograd output transform: bad arguments");
    const long long threads = (long long)Ts * (N / 4);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, Mb, B, H, W, N, scale, bias,
                       relu, y, m_dyn, m_mul, t0, Ts);
    TD_KERNEL_CHECK();
    return TD_OK;
}

// U[xi][n][c] = (G g G^T)[xi] of the OHWI filter bank w [N][3][3][C] (float64 arithmetic, rounded once)
void wino_filter_transform(const float* w, int N, int C, float* U) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            double g[3][3], t[4][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) g[a][b] = w[(((size_t)n * 3 + a) * 3 + b) * C + c];
            for (int i = 0; i < 4; ++i)
                for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[((size_t)(4 * i + j) * N + n) * C + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}

// ---- F(4x4, 3x3) ---------------------------------------------------------------------------------------------------------
td_status wino43_input_launch(const float* x, int B, int H, int W, int C, float* V, const int* m_dyn, hipStream_t s) {
    TD_REQUIRE(x && V && B >= 1 && H >= 1 && W >= 1 && C >= 4 && (C & 3) == 0, "winograd F(4x4) input transform: bad arguments");
    const long long T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    const long long threads = T * (C / 4);
    TD_REQUIRE(threads < (1ll << 31), "winograd F(4x4) input transform: grid too large");
    hipLaunchKernelGGL(wino43_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, B, H, W, C, V, T, m_dyn);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status wino43_output_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                               float* y, const int* m_dyn, hipStream_t s) {
    TD_REQUIRE(Mb && y && B >= 1 && H >= 1 && W >= 1 && N >= 4 && (N & 3) == 0, "winograd F(4x4) output transform: bad arguments");
    const long long T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    const long long threads = T * (N / 4);
    TD_REQUIRE(threads < (1ll << 31), "winograd F(4x4) output transform: grid too large");
    hipLaunchKernelGGL(wino43_output_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, Mb, B, H, W, N, scale, bias,
                       relu, y, T, m_dyn);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status wino43_output_head_launch(const float* Mb, int B, int H, int W, int N, const float* scale, const float* bias, int relu,
                                    const float* head_w, const float* head_b, float* head_y, int head_n, hipStream_t s) {
    TD_REQUIRE(Mb && head_w && head_y && B >= 1 && H >= 1 && W >= 1, "winograd F(4x4) output transform + head: bad arguments");
    TD_REQUIRE(N == 256 && head_n >= 1 && head_n <= 32, "winograd F(4x4) output transform + head: 256 channels, at most 32 head rows (got %d, %d)", N, head_n);
    const long long T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    TD_REQUIRE((T + 3) / 4 < (1ll << 31), "winograd F(4x4) output transform + head: grid too large");
    hipLaunchKernelGGL(wino43_output_head_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, Mb, B, H, W, scale, bias, relu, T,
                       head_w, head_b, head_y, head_n);
    TD_KERNEL_CHECK();
    return TD_OK;
}

// U[xi][n][c] = (G g G^T)[xi], xi = 6 i + j, of the OHWI filter bank w [N][3][3][C] (float64 arithmetic, rounded once)
void wino43_filter_transform(const float* w, int N, int C, float* U) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            double g[3][3], t[6][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) g[a][b] = w[(((size_t)n * 3 + a) * 3 + b) * C + c];
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    U[((size_t)(6 * i + j) * N + n) * C + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
}
