// wino43_fused_kernel: the 36 plane contractions of a Winograd F(4x4,3x3) layer AND its output transform in one launch
// (fp32 engine; reference op: detectron2 Conv2d 3x3 / stride 1 inside GeneralizedRCNN.forward, TreeDetection/prediction.py:183).
//
// Three-launch form (winograd.hip + conv_igemm.hip): x → V [36][T][C] → 36 batched contractions → M [36][T][N] → y. The M planes
// are 2.25x the layer's output, written once and read once: ~11 GB of the fp32 step's 31.8 GB of HBM traffic and 0.95 ms of
// output-transform kernels that run at the HBM roof doing nothing else (DESIGN r3 §10.1).
// Here a block OWNS all 36 planes of its 64 tiles x 64 output channels. Y = A^T M A is linear in M, so each plane's product is
// folded into the 4x4 outputs the moment its k loop ends and M never exists in memory:
//     plane xi = 6 r + c finished:   S[j]   += A^T[j][c] * M_xi          (row r of A^T M ... A, four values j)
//     c == 5 (row r complete):       Y[i][j] += A^T[i][r] * S[j]         (sixteen outputs), S = 0
// 324 FMAs per (tile, channel) over the whole layer against 36 * 2 C MFMA FLOPs — 3.5 % at C = 256, VALU work that issues
// beside the SIMD partner's MFMAs.
//
// Geometry (gfx950): 512 threads = 8 waves = 4 tile quarters x 2 channel halves; a wave owns 16 tiles x 32 channels as two
// v_mfma_f32_16x16x4_f32 blocks computed as D = U V^T — MFMA rows = output channels, columns = tiles — so a lane ends up
// with FOUR CONSECUTIVE CHANNELS of one tile per block: the epilogue stores 16-B channel runs straight from registers (no LDS
// staging round). Registers per lane: Y 16 x 8 = 128, S 4 x 8 = 32, M 8, fragments 24 → two waves per SIMD, one block per CU.
// Data path = conv_igemm_kernel's: a k-chunk is 32 floats = one 128-B row; V rows and U rows go global → LDS by LDS-DMA
// (buffer_load ... lds, 16 B per lane, OOB rows → zeros) into the XOR-swizzled lane-linear image (piece ^ ((row >> 1) & 7)),
// fragments leave it as conflict-free ds_read_b128 (a 16-B fragment feeds 4 MFMAs). ONE continuous stream of 36 * C/32
// chunk-steps per block: AHEAD chunks of DMA in flight across the plane boundaries, counted s_waitcnt vmcnt, one raw s_barrier
// per chunk-step; fragment reads run half a chunk ahead of their MFMAs in a second register set.
// Numerics: same products as the three-launch form, the transform sums associated plane by plane (FMA with the exact
// constants 2, 4, 8) instead of column by column: |error| vs float64 stays at the F(4x4) level (tests/test_conv_gpu.py keeps
// the 5e-5 * max|y| bound), every engine-level fp32 tolerance unchanged. Which layers take this form is a FIXED rule on the
// layer shape (engine.cpp), never a timing decision: it rounds differently from the three-launch form.
#include "common.h"
#include "conv_tiles.h"
#include <utility>

namespace {

struct WinoFusedArgs {
    const float* V;        // [36][T][C]
    const float* U;        // [36][N][C]
    const float* scale;    // [N] or null
    const float* bias;     // [N] or null
    float* y;              // [B][H][W][N]
    const int* m_dyn;      // device-side image count (mask head) or null
    int B, H, W, C, N, relu;
    long long T;           // tiles of the full batch = plane stride in rows
};

constexpr int WF_BT = 64, WF_BN = 64;            // tiles x output channels per block
constexpr int WF_STAGE = (WF_BT + WF_BN) * CHUNK_BYTES;      // 16 KB: V rows then U rows

// A^T of F(4x4,3x3), [i][component]: the fold's coefficients, looked up per plane (wave-uniform scalar loads)
__constant__ float WF_AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -2.f, 0.f}, {0.f, 1.f, 1.f, 4.f, 4.f, 0.f}, {0.f, 1.f, -1.f, 8.f, -8.f, 1.f}};

template <class F, int... I>
__device__ __forceinline__ void wf_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wf_static_for(F&& f) {
    wf_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int NS>
__global__ __launch_bounds__(512, 2) void wino43_fused_kernel(const WinoFusedArgs a) {
    constexpr int AHEAD = NS - 1;                     // chunks of DMA in flight; NS LDS stages (see the WAR argument at the barrier)
    static_assert((NS == 4 || NS == 8) && NS * WF_STAGE <= 160 * 1024, "pipeline depth");
    __shared__ __attribute__((aligned(16))) char lds[NS * WF_STAGE];

    const int TH = (a.H + 3) >> 2, TW = (a.W + 3) >> 2;
    long long live = a.T;
    if (a.m_dyn) {
        const long long n = (long long)*a.m_dyn * TH * TW;
        live = n < live ? n : live;
    }
    const int tb_n = (int)((live + WF_BT - 1) / WF_BT), nb_n = a.N / WF_BN;
    const int nblk = tb_n * nb_n;
    if ((int)blockIdx.x >= nblk) return;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int tb = pid / nb_n, nb = pid - tb * nb_n;          // the channel blocks of one tile range run side by side on one XCD: V rows are fetched once
    const long long t0 = (long long)tb * WF_BT;
    const int n0 = nb * WF_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave & 3, wc = wave >> 2;
    const int KC = a.C >> 5;                          // k-chunks of 32 floats per plane (a multiple of NS: launcher)

    // ---- LDS-DMA source offsets (bytes into V / U; rows past the live tiles read beyond num_records → zeros) ----
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned row_bytes = (unsigned)a.C * 4u;
    const unsigned planeV = (unsigned)((unsigned long long)a.T * row_bytes), planeU = (unsigned)a.N * row_bytes;
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V), 0, (int)(36u * planeV), 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, (int)(36u * planeU), 0x00020000);
    const int ld_c = tid & 7, ld_r = tid >> 3;       // 16-B piece, row 0..63 (a wave instruction = 8 rows x 128 B)
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16u;
    const bool v_ok = t0 + ld_r < live;
    const unsigned v_off = (unsigned)(t0 + ld_r) * row_bytes + src_piece;
    const unsigned u_off = (unsigned)(n0 + ld_r) * row_bytes + src_piece;
    typedef __attribute__((address_space(3))) void lds_void;
    char* const dstV = lds + wave * 8 * CHUNK_BYTES;
    char* const dstU = lds + WF_BT * CHUNK_BYTES + wave * 8 * CHUNK_BYTES;
    int ld_kc = 0;
    unsigned ld_v = 0, ld_u = 0;                      // plane * plane bytes + chunk * 128 of the next chunk to issue
    auto issue = [&](auto st_c) {
        constexpr int ST = decltype(st_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(dstV + ST * WF_STAGE), 16, v_ok ? v_off + ld_v : OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(dstU + ST * WF_STAGE), 16, u_off + ld_u, 0, 0, 0);
        ld_v += CHUNK_BYTES;
        ld_u += CHUNK_BYTES;
        if (++ld_kc == KC) {                          // next plane
            ld_kc = 0;
            ld_v += planeV - row_bytes;
            ld_u += planeU - row_bytes;
        }
    };

    // ---- fragment addresses: lane (r = lane & 15, q = lane >> 4) reads the 16-B piece 4 kk + q of its row ----
    const unsigned r16 = lane & 15, q = lane >> 4, swz = (r16 >> 1) & 7;
    const unsigned pc0 = ((0u + q) ^ swz) * 16u, pc1 = ((4u + q) ^ swz) * 16u;
    const unsigned fv = (unsigned)(wt * 16 + r16) * CHUNK_BYTES;                               // V rows (MFMA B operand: tiles)
    const unsigned fu = (unsigned)(WF_BT + wc * 32 + r16) * CHUNK_BYTES;                       // U rows (MFMA A operand: channels)
    struct Frag { f32x4 u0, u1, v; };
    auto read_frag = [&](Frag& f, auto st_c, unsigned pc) {
        const char* sb = lds + decltype(st_c)::value * WF_STAGE;
        f.u0 = *reinterpret_cast<const f32x4*>(sb + fu + pc);
        f.u1 = *reinterpret_cast<const f32x4*>(sb + fu + 16 * CHUNK_BYTES + pc);
        f.v = *reinterpret_cast<const f32x4*>(sb + fv + pc);
    };
    f32x4 acc[2];
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.u0[e], f.v[e], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.u1[e], f.v[e], acc[1], 0, 0, 0);
        }
    };

    f32x4 Y[16][2], S[4][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[i][0] = Y[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) S[j][0] = S[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: chunks 0 .. AHEAD-1 in flight (stages 0 .. AHEAD-1); chunk 0 landed and its first half-fragment read ----
    wf_static_for<AHEAD>([&](auto i_c) { issue(i_c); });          // KC % NS == 0 (launcher): the stream is longer than the pipeline
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AHEAD - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    read_frag(f0, std::integral_constant<int, 0>{}, pc0);

    // Step s (chunk s of the flat stream, stage s % NS): on entry chunk s has landed for every wave and f0 holds its first
    // half (read during step s - 1). The DMA of chunk s + AHEAD goes into the stage chunk s - 1 occupied: every wave finished
    // its reads of chunk s - 1 before the barrier that ended step s - 1 (lgkmcnt(0) sits in front of each barrier), and this
    // wave is past that barrier. Before the barrier that ends step s every wave waits until its own DMAs of chunk s + 1 have
    // landed (vmcnt: all but the chunks issued after it — AHEAD - 1 of them in the steady state, two DMAs each), so after the
    // barrier chunk s + 1 is complete for all. The chunk loop is unrolled NS times: every LDS address is a base register + an
    // immediate. TAIL = the stream's last NS steps: only the first of them still issues (the last chunk), the others wait for
    // all but the NS - 2 - ST chunks issued after chunk s + 1. The very last step re-reads stage 0, which nobody needs (in
    // bounds, harmless): every step has the same shape and hipcc counts its own lgkmcnt waits.
    auto body = [&](auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        wf_static_for<NS>([&](auto st_c) {
            constexpr int ST = decltype(st_c)::value, NEXT = (ST + 1) % NS;
            constexpr bool ISSUE = !TAIL || ST == 0;
            constexpr int NEWER = ISSUE ? AHEAD - 1 : (NS - 2 - ST > 0 ? NS - 2 - ST : 0);
            if constexpr (ISSUE) issue(std::integral_constant<int, (ST + AHEAD) % NS>{});
            read_frag(f1, st_c, pc1);
            mma(f0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NEWER) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            read_frag(f0, std::integral_constant<int, NEXT>{}, pc0);
            mma(f1);
        });
    };
    const int nb_k = KC / NS;
    for (int p = 0; p < 36; ++p) {
        for (int kb = 0; kb + 1 < nb_k; ++kb) body(std::false_type{});
        if (p < 35) body(std::false_type{});
        else body(std::true_type{});
        // plane p = 6 xr + xc complete: S[j] += A^T[j][xc] * M;  xc == 5: Y[i][j] += A^T[i][xr] * S[j], S = 0
        // (runtime coefficients: fma(1, m, s) = s + m and fma(0, m, s) = s exactly, so this IS the sparse sum)
        const int xr = p / 6, xc = p - 6 * xr;
        const float c0 = WF_AT[0][xc], c1 = WF_AT[1][xc], c2 = WF_AT[2][xc], c3 = WF_AT[3][xc];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m = acc[b][e];
                S[0][b][e] = __fmaf_rn(c0, m, S[0][b][e]);
                S[1][b][e] = __fmaf_rn(c1, m, S[1][b][e]);
                S[2][b][e] = __fmaf_rn(c2, m, S[2][b][e]);
                S[3][b][e] = __fmaf_rn(c3, m, S[3][b][e]);
            }
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xc == 5) {
            const float r0 = WF_AT[0][xr], r1 = WF_AT[1][xr], r2 = WF_AT[2][xr], r3 = WF_AT[3][xr];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = S[j][b][e];
                        S[j][b][e] = 0.f;
                        Y[0 + j][b][e] = __fmaf_rn(r0, v, Y[0 + j][b][e]);
                        Y[4 + j][b][e] = __fmaf_rn(r1, v, Y[4 + j][b][e]);
                        Y[8 + j][b][e] = __fmaf_rn(r2, v, Y[8 + j][b][e]);
                        Y[12 + j][b][e] = __fmaf_rn(r3, v, Y[12 + j][b][e]);
                    }
        }
    }

    // ---- epilogue: lane = tile (t0 + wt*16 + r16), channels n0 + wc*32 + blk*16 + 4 q .. +3; 16 pixels x 2 blocks of 16-B stores.
    // Branch-free: a pixel outside the map (or a dead tile) stores to an offset beyond num_records, which the buffer unit drops —
    // with exec-masked branches hipcc put a vmcnt(0) in front of every store and the 32 stores of a lane ran one after the other.
    const long long t = t0 + wt * 16 + (int)r16;
    const bool tlive = t < live;
    const unsigned tt = tlive ? (unsigned)t : 0u;
    const unsigned tyx = tt % (unsigned)(TW * TH);
    const unsigned b = tt / (unsigned)(TW * TH);
    const unsigned ty = tyx / (unsigned)TW, tx = tyx - ty * (unsigned)TW;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.B * a.H * a.W * a.N * 4u), 0x00020000);
    f32x4 sc[2], bi[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int n = n0 + wc * 32 + blk * 16 + 4 * (int)q;
        sc[blk] = a.scale ? *reinterpret_cast<const f32x4*>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        bi[blk] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned pix0 = (b * (unsigned)a.H + 4u * ty) * (unsigned)a.W + 4u * tx;
    const unsigned nbytes = (unsigned)a.N * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool rok = tlive && (int)(4u * ty) + i < a.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rok && (int)(4u * tx) + j < a.W;
            const unsigned pbase = (pix0 + (unsigned)(i * a.W + j)) * nbytes;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f32x4 v = Y[4 * i + j][blk];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = v[e];
                    if (a.scale) o = __fmul_rn(o, sc[blk][e]);
                    if (a.bias) o = __fadd_rn(o, bi[blk][e]);
                    if (a.relu) o = o > 0.f ? o : 0.f;
                    v[e] = o;
                }
                const unsigned off = ok ? pbase + (unsigned)(n0 + wc * 32 + blk * 16 + 4 * (int)q) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yrsrc, off, 0, 0);
            }
        }
    }
}

}  // namespace

bool wino43_fused_ok(int B, int H, int W, int C, int N) {
    const long long T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    return C >= 128 && C % 128 == 0 && N >= WF_BN && N % WF_BN == 0 && 36ull * (unsigned long long)T * C * 4 < 0xfffffff0ull - (1u << 20) &&
           36ull * (unsigned long long)N * C * 4 < 0xfffffff0ull - (1u << 20) && (unsigned long long)B * H * W * N * 4 < 0xfffffff0ull - (1u << 20);
}

// V [36][T][C] (wino43_input_launch) x U [36][N][C] → y [B,H,W,N] = act((A^T (sum_c U .* V) A) * scale + bias)
td_status wino43_fused_launch(const float* V, const float* U, int B, int H, int W, int C, int N, const float* scale, const float* bias,
                              int relu, float* y, const int* m_dyn, hipStream_t s) {
    TD_REQUIRE(V && U && y && B >= 1 && H >= 1 && W >= 1, "winograd F(4x4) fused contraction: bad arguments");
    TD_REQUIRE(wino43_fused_ok(B, H, W, C, N), "winograd F(4x4) fused contraction: needs C %% 128 == 0, N %% 64 == 0 and planes below 4 GB (C %d, N %d)", C, N);
    WinoFusedArgs a{};
    a.V = V; a.U = U; a.scale = scale; a.bias = bias; a.y = y; a.m_dyn = m_dyn;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.relu = relu;
    a.T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    const long long blocks = ((a.T + WF_BT - 1) / WF_BT) * (N / WF_BN);
    TD_REQUIRE(blocks < (1ll << 31), "winograd F(4x4) fused contraction: grid too large");
    // LDS stages: 8 (seven chunks of DMA in flight, 128 KB) where the plane's chunk count allows the 8-fold unrolled loop, else 4
    static const int stages = getenv("TD_WF_STAGES") ? atoi(getenv("TD_WF_STAGES")) : 8;
    if (stages >= 8 && (C / 32) % 8 == 0) hipLaunchKernelGGL(wino43_fused_kernel<8>, dim3((unsigned)blocks), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(wino43_fused_kernel<4>, dim3((unsigned)blocks), dim3(512), 0, s, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}
