// Device pieces shared by the contraction kernels (conv_igemm.hip, bottleneck.hip): MFMA element traits, the XCD-aware block
// remap and the LDS-staged epilogue. Included by .hip files only.
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CHUNK_BYTES = 128;     // one k-chunk of one row = one LDS row (8 pieces of 16 B, XOR-swizzled)

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (id % 8) get a contiguous run of tiles,
// so the A rows / weight panels they share stay in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int PER_CHUNK = 32;
    static __device__ __forceinline__ void mma(const f32x4& fa, const f32x4& fb, f32x16& acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(float v) { return v; }
};
template <>
struct Elem<_Float16> {
    static constexpr int PER_CHUNK = 64;
    static __device__ __forceinline__ void mma(const f32x4& fa, const f32x4& fb, f32x16& acc) {
#if defined(TD_DIAG_MFMA16B)    // timing experiment only (tools/conv_diag.py): TWO 16x16x32 = the FLOPs and issue cycles of one 32x32x16 (wrong sums)
        f32x4 q[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            q[k] = f32x4{acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
            q[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), q[k], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * k + e] = q[k][e];
        }
#elif defined(TD_DIAG_MFMA16)     // timing experiment only (tools/conv_diag.py): four 16x16x32 = TWICE the FLOPs of one 32x32x16
        f32x4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = f32x4{acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
            q[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), q[k], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * k + e] = q[k][e];
        }
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), acc, 0, 0, 0);
#endif
    }
    static __device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
};

// ---- epilogue ---------------------------------------------------------------------------------------------------------
// y = act(acc * scale + bias (+ residual)), one IEEE op per step (no fma contraction), identical in both paths below.
// The MFMA layout (a lane = one column, 16 scattered rows of a 32 x 32 tile) is turned into whole row segments through
// LDS, so residual loads and stores are 16-B accesses of contiguous channels and fully coalesced.
//  * fp16 output without residual (every 3x3, the stride / first 1x1 of a block, FC layers): scale / bias / ReLU and
//    the fp16 rounding happen in REGISTERS, neighbouring lanes swap one value (DPP quad_perm) so that each lane owns a
//    packed channel pair, and the whole block tile is staged as fp16 in ONE round: half the ds_write instructions and
//    LDS bytes of the fp32 staging, no second barrier round, 16-B (8-channel) stores. The epilogue of the 256 x 256
//    fp16 tile cost 20 us of a 95-us tile before (store-issue bound at 8 B per lane).
//  * otherwise (residual add, fp32 output, odd channel counts): fp32 tile staged RWM wave-rows at a time; fp16 outputs
//    leave as 8 channels (16 B) per lane with 16-B residual loads issued U rows ahead.
// `lds` must hold the staged tile (conv_epilogue_lds_bytes) and every wave must be past its last LDS read of the k-loop
// (the caller's final barrier).
template <typename TO, int MT, int NT, int WM, int WN, int RWM, bool WIDE_OK = true>
constexpr int conv_epilogue_lds_bytes() {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int f32 = RWM * 32 * MT * (BN + 4) * 4;
    constexpr int f16 = sizeof(TO) == 2 && WIDE_OK ? BM * (BN + 8) * 2 : 0;
    return f32 > f16 ? f32 : f16;
}

template <typename T, typename TO, int MT, int NT, int WM, int WN, int RWM, bool WIDE_OK = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], char* lds, int M, int m0, int n0,
                                              int tid, int lane, int wm, int wn) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int CS = BN + 4;                        // padded row stride (floats) of the staged fp32 tile
    constexpr int WROW = 32 * MT;                     // rows of one wave-row
    TO* __restrict__ Y = static_cast<TO*>(a.y);
    const T* __restrict__ Rs = static_cast<const T*>(a.res);
    const int hw = a.Ho * a.Wo;

    // The 8-channel-per-lane general path costs ~30 more live registers than the 4-channel one: worth it on the 8- and
    // 16-wave tiles that own a CU anyway, not on the 4-wave tiles whose thin layers live on three or four co-resident
    // blocks per CU. The register-finished fp16 staging is cheap in registers as long as the accumulators live in VGPRs
    // (the Makefile builds this file with -mllvm -amdgpu-mfma-vgpr-form: with AGPR accumulators hipcc copies all of
    // them to VGPRs ahead of this code, +64 registers).
    constexpr bool WIDE = sizeof(TO) == 2 && THREADS >= 512 && WIDE_OK;      // 8 channels per lane in the general path
    constexpr bool E16 = sizeof(TO) == 2 && WIDE_OK;                         // register-finished fp16 staging
    if constexpr (E16) {
        if (!Rs && a.out_mode == 0 && (a.Cout & 7) == 0) {
            // ---- fp16 fast path: finish in registers, stage packed fp16, one round ----
            constexpr int HS = BN + 8;                // row stride in halves: 16-B aligned rows, 4-dword skew between rows
            _Float16* Hs = reinterpret_cast<_Float16*>(lds);
            const int odd = lane & 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = wn * 32 * NT + j * 32 + (lane & 31);
                const int nn = n0 + col;
                float sc = 1.f, bi = 0.f;
                if (nn < a.Cout) {
                    if (a.scale) sc = a.scale[nn];
                    if (a.bias) bi = a.bias[nn];
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float t0 = acc[i][j][r], t1 = acc[i][j][r + 1];     // rows R and R + 1 of this lane's column
                        if (a.scale) { t0 = __fmul_rn(t0, sc); t1 = __fmul_rn(t1, sc); }
                        if (a.bias) { t0 = __fadd_rn(t0, bi); t1 = __fadd_rn(t1, bi); }
                        if (a.relu) { t0 = t0 > 0.f ? t0 : 0.f; t1 = t1 > 0.f ? t1 : 0.f; }
                        // even lanes keep row R (and get the right neighbour's R value), odd lanes keep row R + 1
                        const float give = odd ? t0 : t1;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, give), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        f16x2 pk;
                        pk[0] = (_Float16)(odd ? got : t0);
                        pk[1] = (_Float16)(odd ? t1 : got);
                        const int row = wm * WROW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) + odd;
                        *reinterpret_cast<f16x2*>(&Hs[row * HS + (col & ~1)]) = pk;
                        // keep the pairs sequential: hoisting all conversions ahead of the stores costs ~35 live registers
                        // (and with them a block per CU on the small tiles)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();
            if constexpr (BN == 256 && sizeof(T) == 2) {
                if (a.head_w) {
                    // ---- fused 1x1 head: rows of the staged tile x head_w^T, k = the tile's 256 channels in ascending order with
                    // the MFMA of every other fp16 tile (bit-identical to a separate launch over the stored tile) ----
                    constexpr int NWAVE = THREADS / 64, RT = BM / 32;
                    const int wv = tid >> 6;
                    const int col = lane & 31, kh8 = (lane >> 5) * 8;
                    const _Float16* __restrict__ Wh = static_cast<const _Float16*>(a.head_w);
                    const bool live = col < a.head_n;
                    const float hb = live && a.head_b ? a.head_b[col] : 0.f;
                    f32x4 fbh[16];
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) {
                        f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        fbh[kk] = live ? *reinterpret_cast<const f32x4*>(Wh + (size_t)col * BN + kk * 16 + kh8) : z;
                    }
                    for (int rt = wv; rt < RT; rt += NWAVE) {
                        f32x16 hacc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) {
                            const f32x4 fah = *reinterpret_cast<const f32x4*>(&Hs[(rt * 32 + col) * HS + kk * 16 + kh8]);
                            Elem<T>::mma(fah, fbh[kk], hacc);
                        }
                        if (live) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int m = m0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                if (m < M) a.head_y[(size_t)m * a.head_n + col] = a.head_b ? __fadd_rn(hacc[r], hb) : hacc[r];
                            }
                        }
                    }
                    return;
                }
            }
            constexpr int PIECES = BN / 8;            // 16-B pieces per tile row
            constexpr int ROWS_PER_PASS = THREADS / PIECES;
            static_assert(BM % ROWS_PER_PASS == 0, "fp16 epilogue rows must split evenly over the threads");
            const int pc = tid % PIECES, pr = tid / PIECES;
            const int n = n0 + pc * 8;
            if (n < a.Cout) {                         // Cout % 8 == 0: a piece is in range as a whole
#pragma unroll 4
                for (int it = 0; it < BM / ROWS_PER_PASS; ++it) {
                    const int row = pr + it * ROWS_PER_PASS;
                    const int m = m0 + row;
                    if (m >= M) break;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Hs[row * HS + pc * 8]);
                    *reinterpret_cast<f32x4*>(Y + (size_t)m * a.Cout + n) = v;
                }
            }
            return;
        }
    }

    // ---- general path: fp32 tile, RWM wave-rows per round ----
    float* Cs = reinterpret_cast<float*>(lds);
    const int Cq = a.out_mode == 1 ? a.Cout >> 2 : a.Cout;
    constexpr int CPL = WIDE ? 8 : 4;                  // channels per lane
    constexpr int CHUNKS = BN / CPL;                   // pieces per tile row
    constexpr int ROWS_PER_PASS = THREADS / CHUNKS;
    const int c4 = tid % CHUNKS;
    const int n = n0 + c4 * CPL;
    const bool vec = (a.Cout % CPL) == 0 && n + CPL - 1 < a.Cout;   // aligned, whole piece in range
    float sc[CPL], bi[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        sc[e] = 1.f;
        bi[e] = 0.f;
        if (n + e < a.Cout) {
            const int co = a.out_mode == 1 ? (n + e) % Cq : n + e;
            if (a.scale) sc[e] = a.scale[co];
            if (a.bias) bi[e] = a.bias[co];
        }
    }
    for (int q = 0; q < WM; q += RWM) {
        if (q > 0) __syncthreads();                    // the previous round's readers are done with Cs
        if (wm >= q && wm < q + RWM) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm - q) * WROW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int col = wn * 32 * NT + j * 32 + (lane & 31);
                        Cs[row * CS + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        // Each thread finishes ITERS rows of this round. The residual reads are the only loads left in the kernel and
        // nothing hides their latency (the accumulators are already in LDS, the waves have no other work), so they are
        // issued U rows ahead: U x 16 B per lane in flight instead of 16 B (the thin 1x1 layers with a shortcut add
        // ran at 3.3 TB/s of HBM before, latency-bound right here).
        constexpr int ITERS = (RWM * WROW) / ROWS_PER_PASS;
        static_assert((RWM * WROW) % ROWS_PER_PASS == 0 && ITERS >= 1, "epilogue rows must split evenly over the threads");
        constexpr int UMAX = CPL == 8 ? 4 : 8;         // 64 B per lane in flight either way
        constexpr int U = ITERS < UMAX ? ITERS : UMAX;
        static_assert(ITERS % U == 0, "epilogue batch must divide the rows per thread");
        typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
        typedef typename std::conditional<sizeof(T) == 4, f32x4, typename std::conditional<CPL == 8, f16x8v, f16x4>::type>::type ResVec;
        auto res_offset = [&](int m) -> size_t {
            if (!a.res_shift) return (size_t)m * a.Cout + n;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            return ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n;
        };
        for (int it0 = 0; it0 < ITERS; it0 += U) {
            ResVec rbuf[U];
            if (Rs && vec) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int m = m0 + q * WROW + tid / CHUNKS + (it0 + u) * ROWS_PER_PASS;
                    if (m < M) rbuf[u] = *reinterpret_cast<const ResVec*>(Rs + res_offset(m));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = tid / CHUNKS + (it0 + u) * ROWS_PER_PASS;
                const int m = m0 + q * WROW + row;
                if (m >= M || n >= a.Cout) continue;
                float v[CPL];
#pragma unroll
                for (int g = 0; g < CPL / 4; ++g) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&Cs[row * CS + c4 * CPL + 4 * g]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = t[e];
                }
                size_t yoff;
                if (a.out_mode == 0) {
                    yoff = (size_t)m * a.Cout + n;
                } else {
                    const int qq = n / Cq, co = n - qq * Cq;   // qq = dy*2+dx; a piece never straddles it (Cq % CPL == 0)
                    const int b = m / hw;
                    const int rem = m - b * hw;
                    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                    yoff = ((size_t)(b * 2 * a.Ho + 2 * oy + (qq >> 1)) * (2 * a.Wo) + 2 * ox + (qq & 1)) * Cq + co;
                }
                float rs[CPL];
#pragma unroll
                for (int e = 0; e < CPL; ++e) rs[e] = 0.f;
                if (Rs) {
                    if (vec) {
#pragma unroll
                        for (int e = 0; e < CPL; ++e) rs[e] = (float)rbuf[u][e];
                    } else {
                        const size_t roff = res_offset(m);
#pragma unroll
                        for (int e = 0; e < CPL; ++e)
                            if (n + e < a.Cout) rs[e] = Elem<T>::to_f32(Rs[roff + e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    float t = v[e];
                    if (a.scale) t = __fmul_rn(t, sc[e]);
                    if (a.bias) t = __fadd_rn(t, bi[e]);
                    if (Rs) t = __fadd_rn(t, rs[e]);
                    if (a.relu) t = t > 0.f ? t : 0.f;
                    v[e] = t;
                }
#if defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 8)      // timing builds only (tools/tail_probe.py): no output stores
                if (m0 != 0x7fffff00) continue;
#endif
                if (vec) {
                    if constexpr (sizeof(TO) == 4) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(Y + yoff) = o;
                    } else if constexpr (CPL == 8) {
                        f16x8v h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
                        *reinterpret_cast<f16x8v*>(Y + yoff) = h;
                    } else {
                        f16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
                        *reinterpret_cast<f16x4*>(Y + yoff) = h;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < CPL; ++e)
                        if (n + e < a.Cout) Y[yoff + e] = (TO)v[e];
                }
            }
        }
    }
}

}  // namespace
