// Implicit-GEMM NHWC convolution on the gfx950 matrix cores.
//
//   y[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + bias[n] (+ residual[m][n]) )
//   m = (b, oy, ox)   n = output channel   k = (ky, kx, c)  — c fastest, so one k-chunk of 32 is a
//   contiguous 128-B run of one input pixel (coalesced HBM reads, no im2col buffer).
//
// This one kernel family carries every contraction of the reference's forward
// (TreeDetection/prediction.py:183 → detectron2 GeneralizedRCNN):
// bottleneck 1x1 / 3x3 (stride lives in the 1x1, STRIDE_IN_1X1), FPN lateral (+ nearest-2x
// upsampled top-down add in the epilogue) and output convs, RPN conv + heads, the box-head FCs
// (1x1 "conv" over R rows), the mask-head 3x3s and the 2x2/s2 deconv (out_mode 1: four 1x1
// GEMMs with a pixel-shuffle store).
//
// fp32 path: v_mfma_f32_32x32x2_f32 — exact f32 fmaf chains (MI355X_MICROARCH.md §Matrix cores),
// 128x128x32 block tile, 4 waves (2x2), 64x64 per wave, LDS rows padded to 36 floats so every
// ds_read_b128 lane group hits 16 distinct 16-B slots. Global→register→LDS double buffering with
// one barrier per k-step.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;        // k-chunk (floats)
constexpr int LDS_STRIDE = 36;  // padded row stride in floats (144 B = 9 x 16 B)

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (id % 8) get a
// contiguous run of tiles, so the A rows / weight panels they share stay in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_f32(const ConvArgs a) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int AROWS = BM / 32, BROWS = BN / 32;  // rows staged per thread
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_STRIDE];
    float* As = lds;                          // [2][BM][36]
    float* Bs = lds + 2 * BM * LDS_STRIDE;    // [2][BN][36]

    int M = a.M;
    if (a.m_dyn) {
        const int md = *a.m_dyn * a.m_mul;
        M = md < M ? md : M;
    }
    // the grid is sized for a.M; with a device-side row count only the first `nblk` blocks have work, and the
    // XCD remap runs over THAT count so the live tiles still spread over all 8 XCDs
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int nblk = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nblk) return;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ld_c = tid & 7;    // 16-B chunk inside the 128-B k-run
    const int ld_r = tid >> 3;   // 0..31

    const float* __restrict__ X = static_cast<const float*>(a.x);
    const float* __restrict__ Wt = static_cast<const float*>(a.w);
    const int K = a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / BK;
    const int ntaps = a.KH * a.KW;
    const int nit = ntaps * cchunks;

    // per-thread A rows: pixel index of tap (0,0) and its (iy, ix)
    int a_pix[AROWS], a_iy[AROWS], a_ix[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + ld_r + 32 * i;
        if (m < M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            a_iy[i] = oy * a.stride - a.pad;
            a_ix[i] = ox * a.stride - a.pad;
            a_pix[i] = (b * a.H + a_iy[i]) * a.W + a_ix[i];
        } else {
            a_iy[i] = -(1 << 28);   // fails every bounds test
            a_ix[i] = -(1 << 28);
            a_pix[i] = 0;
        }
    }
    size_t b_off[BROWS];
    bool b_ok[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + ld_r + 32 * i;
        b_ok[i] = n < a.Cout;
        b_off[i] = (size_t)(b_ok[i] ? n : 0) * K + ld_c * 4;
    }

    f32x4 ra[AROWS], rb[BROWS];
    auto load_global = [&](int it) {
        // k order: channel chunk OUTER, filter tap INNER — the 9 taps of a 3x3 re-read the same few input rows of one
        // 128-B channel slice back to back, so the halo re-reads hit L1/L2 instead of going back to HBM
        const int cc = it / ntaps;
        const int tap = it - cc * ntaps;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        const int coff = cc * BK + ld_c * 4;
        const size_t woff = (size_t)tap * a.Cin + cc * BK;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            if (ok) {
                const size_t off = (size_t)(a_pix[i] + ky * a.W + kx) * a.Cin + coff;
                ra[i] = *reinterpret_cast<const f32x4*>(X + off);
            } else {
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            if (b_ok[i]) rb[i] = *reinterpret_cast<const f32x4*>(Wt + b_off[i] + woff);
            else rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + ld_r + 32 * i) * LDS_STRIDE + ld_c * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + ld_r + 32 * i) * LDS_STRIDE + ld_c * 4]) = rb[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_global(0);
    store_lds(0);
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    int cur = 0;
    for (int it = 0; it < nit; ++it) {
        if (it + 1 < nit) load_global(it + 1);
        const float* Ab = &As[(cur * BM + wm * 32 * MT + frag_row) * LDS_STRIDE + frag_k];
        const float* Bb = &Bs[(cur * BN + wn * 32 * NT + frag_row) * LDS_STRIDE + frag_k];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < nit) store_lds(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: accumulators → LDS tile → 16-B-per-lane rows: scale/bias (+residual) (+ReLU), one IEEE op per
    // step (no fma contraction). Staging through LDS turns the MFMA layout (32 lanes x 4 B per row) into whole
    // 512-B / 256-B row segments, so residual loads and stores are dwordx4 and fully coalesced (HBM-bound 1x1 layers).
    constexpr int CS = BN + 4;                         // padded row stride of the staged tile (floats)
    static_assert(BM * CS <= 2 * (BM + BN) * LDS_STRIDE, "epilogue tile must fit in the k-loop LDS");
    float* Cs = lds;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 * MT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = wn * 32 * NT + j * 32 + (lane & 31);
                Cs[row * CS + col] = acc[i][j][r];
            }
    __syncthreads();

    float* __restrict__ Y = static_cast<float*>(a.y);
    const float* __restrict__ Rs = static_cast<const float*>(a.res);
    const int Cq = a.out_mode == 1 ? a.Cout >> 2 : a.Cout;
    constexpr int CHUNKS = BN / 4;                     // float4 chunks per tile row
    constexpr int ROWS_PER_PASS = 256 / CHUNKS;
    const int c4 = tid % CHUNKS;
    const int n = n0 + c4 * 4;
    const bool vec = (a.Cout & 3) == 0 && n + 3 < a.Cout;   // aligned, whole chunk in range
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n + e < a.Cout) {
            const int co = a.out_mode == 1 ? (n + e) % Cq : n + e;
            if (a.scale) sc[e] = a.scale[co];
            if (a.bias) bi[e] = a.bias[co];
        }
    }
    const int hw = a.Ho * a.Wo;
    for (int row = tid / CHUNKS; row < BM; row += ROWS_PER_PASS) {
        const int m = m0 + row;
        if (m >= M || n >= a.Cout) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CS + c4 * 4]);
        size_t yoff, roff = 0;
        if (a.out_mode == 0) {
            yoff = (size_t)m * a.Cout + n;
            roff = yoff;
            if (Rs && a.res_shift) {
                const int b = m / hw;
                const int rem = m - b * hw;
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                roff = ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n;
            }
        } else {
            const int q = n / Cq, co = n - q * Cq;   // q = dy*2+dx; a chunk never straddles q (Cq % 4 == 0)
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            yoff = ((size_t)(b * 2 * a.Ho + 2 * oy + (q >> 1)) * (2 * a.Wo) + 2 * ox + (q & 1)) * Cq + co;
        }
        f32x4 rs = {0.f, 0.f, 0.f, 0.f};
        if (Rs) {
            if (vec) rs = *reinterpret_cast<const f32x4*>(Rs + roff);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a.Cout) rs[e] = Rs[roff + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (a.scale) t = __fmul_rn(t, sc[e]);
            if (a.bias) t = __fadd_rn(t, bi[e]);
            if (Rs) t = __fadd_rn(t, rs[e]);
            if (a.relu) t = t > 0.f ? t : 0.f;
            v[e] = t;
        }
        if (vec) *reinterpret_cast<f32x4*>(Y + yoff) = v;
        else
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < a.Cout) Y[yoff + e] = v[e];
    }
}

template <int MT, int NT>
td_status launch_f32(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    const int tiles = td_cdiv(a.M, BM) * td_cdiv(a.Cout, BN);
    hipLaunchKernelGGL((conv_igemm_f32<MT, NT>), dim3(tiles), dim3(256), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

}  // namespace

td_status conv2d_launch(const ConvArgs& a, int precision, hipStream_t stream) {
    TD_REQUIRE(a.Cin % BK == 0, "conv2d: Cin=%d must be a multiple of %d", a.Cin, BK);
    TD_REQUIRE(a.M > 0 && a.Cout > 0, "conv2d: empty problem (M=%d, Cout=%d)", a.M, a.Cout);
    TD_REQUIRE((size_t)a.B * a.H * a.W * (size_t)a.Cin < (1ull << 31), "conv2d: input too large for 32-bit pixel math");
    TD_REQUIRE(a.out_mode == 0 || (a.Cout % 4 == 0 && !a.res), "conv2d: bad deconv configuration");
    if (precision != TD_PRECISION_FP32) {
        td_set_error("conv2d: precision %d not built", precision);
        return TD_ERR_INVALID;
    }
    int cfg = a.tile_cfg;
    if (cfg < 0) {
        // heuristic (the engine replaces it by a measured choice per layer shape): wide N for wide layers, 64-row
        // tiles when there is less than one 128-row tile per CU
        const bool small_m = a.M <= 64 * 256;
        cfg = a.Cout <= 64 ? (small_m ? 3 : 1) : (small_m ? 2 : 0);
    }
    switch (cfg) {
        case 0: return launch_f32<2, 2>(a, stream);
        case 1: return launch_f32<2, 1>(a, stream);
        case 2: return launch_f32<1, 2>(a, stream);
        case 3: return launch_f32<1, 1>(a, stream);
        default: td_set_error("conv2d: bad tile_cfg %d", cfg); return TD_ERR_INVALID;
    }
}
