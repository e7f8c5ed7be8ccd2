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
