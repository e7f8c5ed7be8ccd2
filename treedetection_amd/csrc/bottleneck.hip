// bottleneck_tail_kernel: conv2 (3x3, mid -> mid) + FrozenBN + ReLU and conv3 (1x1, mid -> 4 mid) + FrozenBN + shortcut add +
// ReLU of a ResNet bottleneck block in ONE launch (detectron2 BottleneckBlock behind TreeDetection/prediction.py:183; SURVEY.md
// Appendix A item 3). The mid tensor between the two convolutions never exists in HBM.
//
// Why: in res2 / res3 the 1x1 "expand" layer is HBM-bound on its own (K = 64 / 128: one to four k-steps of MFMA work per
// tile against a residual read and an output write of 4x the input) and the 3x3 before it writes a tensor that is read back
// once. Fused, a block of 128 output pixels
//   1. runs the 3x3 as conv_igemm_kernel does (LDS-DMA staging of 128-B k-chunks into the XOR-swizzled image, channel chunk
//      outer / filter tap inner, two LDS stages, raw s_barrier + counted vmcnt) into 128 x mid accumulators,
//   2. applies scale / bias / ReLU (and the fp16 rounding of the fp16 engine) in registers and writes the tile into LDS in
//      exactly the image the DMA would have produced from a [row][mid] tensor — it is the A operand of the 1x1,
//   3. streams the 1x1 filters through LDS 128 output channels at a time and contracts (k = mid), then finishes each
//      128 x 128 piece with conv_epilogue (scale, bias, + shortcut, ReLU; whole coalesced row segments).
// Every sum keeps the k order of the two separate launches and every epilogue operation is the same single IEEE operation,
// so the result is BIT-IDENTICAL to conv2d_launch(3x3) followed by conv2d_launch(1x1 + residual) (tests/test_conv_gpu.py) —
// all engine-level parity statements carry over unchanged.
// HBM traffic per block of a stage: t1 (read, halo through L2) + shortcut (read) + y (write) instead of those plus one write
// and one read of the mid tensor; one launch instead of two.
#include "common.h"
#include "conv_tiles.h"
#include <cstdlib>
#include <utility>

namespace {

constexpr int cmax(int x, int y) { return x > y ? x : y; }

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {            // f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>)
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// FAST (fp16, 64-row blocks): the wave-private wide epilogue below instead of conv_epilogue
template <typename T, int MID, bool OVERLAP, int MT, bool FAST = false>
struct TailGeom {
    static constexpr int ES = sizeof(T);
    static constexpr int KE = Elem<T>::PER_CHUNK;            // elements per 128-B k-chunk
    static constexpr int KC = MID / KE;                      // channel chunks of the mid tensor
    static constexpr int BM = 64 * MT, THREADS = 256, LDROWS = 32;   // 2 x 2 waves, a wave owns 32 MT rows
    static constexpr int NT1 = MID / 64;                     // phase 1: wave tile 64 x (32 NT1), block tile 128 x MID
    static constexpr int STAGE1 = 2 * (BM + MID) * CHUNK_BYTES;
    static constexpr int W3_BYTES = KC * 128 * CHUNK_BYTES;  // the 1x1 filters of one 128-channel output piece: [KC][128][128 B]
    static_assert(!FAST || MT == 1, "the wave-private epilogue is written for 64-row blocks");
    static constexpr int EPI_BYTES = FAST ? 4 * 16 * 64 * 4 : conv_epilogue_lds_bytes<T, MT, 2, 2, 2, 1, false>();    // FAST: 16 rows x 64 floats per wave
    // OVERLAP / fp16 FAST: the next piece's filters arrive while this piece is finished; fp32 FAST stages on top of the filters it
    // has just read (with both regions the block would cost the third block per CU) and restages them behind a barrier
    static constexpr int EPI_OFF = OVERLAP || (FAST && ES == 2) ? W3_BYTES : 0;
    // region 0 = the phase-1 stages, later the filters of a piece and the epilogue's staging tile (side by side or aliased)
    static constexpr int R0 = cmax(STAGE1, cmax(W3_BYTES, EPI_OFF + EPI_BYTES));
    static constexpr int T2_BYTES = KC * BM * CHUNK_BYTES;   // the mid tile as the 1x1's A image: [KC][128][128 B]
    static constexpr int LDS_BYTES = R0 + T2_BYTES;
    static constexpr int BPC = 160 * 1024 / LDS_BYTES >= 4 ? 4 : 160 * 1024 / LDS_BYTES;      // blocks per CU the LDS footprint allows
};

// The body is a __device__ function (the __global__ entry below only owns the LDS array): with the staging lambdas called
// straight from a __global__ template, hipcc (ROCm 7.2) silently dropped the kernel's HOST stub from the object file.
template <typename T, int MID, bool OVERLAP, int MT, bool FAST>
__device__ __forceinline__ void bottleneck_tail_body(const TailArgs& a, char* lds) {
    typedef TailGeom<T, MID, OVERLAP, MT, FAST> G;
    constexpr int ES = G::ES, KE = G::KE, KC = G::KC, BM = G::BM, LDROWS = G::LDROWS, NT1 = G::NT1;
    constexpr int AROWS = BM / LDROWS, BROWS1 = MID / LDROWS, BROWS3 = 128 / LDROWS;
    char* As = lds;                                   // phase 1: [2][BM][128 B]
    char* Bs = lds + 2 * BM * CHUNK_BYTES;            //          [2][MID][128 B]
    char* W3s = lds;                                  // phase 3: [KC][128][128 B] (the phase-1 stages are dead by then)
    char* Epi = lds + G::EPI_OFF;
    char* T2s = lds + G::R0;                          // [KC][BM][128 B]

    const int M = a.M;
    const int nblk = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= nblk) return;
    const int m0 = xcd_remap(blockIdx.x, nblk) * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ld_c = tid & 7, ld_r = tid >> 3;

    // ---- phase 1: the 3x3 (stride 1, pad 1) as in conv_igemm_kernel ------------------------------------------------------
    const int K2 = 9 * MID;
    const int nit = 9 * KC;
    const unsigned pix_bytes = (unsigned)MID * ES;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.H * a.W * pix_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w2), 0, (int)((size_t)MID * K2 * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w3), 0, (int)((size_t)a.COUT * MID * ES), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;

    unsigned a_off[AROWS], a_ok[AROWS];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + ld_r + LDROWS * i;
        a_off[i] = 0;
        a_ok[i] = 0;
        if (m < M) {
            const int hw = a.H * a.W;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / a.W;
            const int ox = rem - oy * a.W;
            const int iy0 = oy - 1, ix0 = ox - 1;
            a_off[i] = (unsigned)((b * a.H + iy0) * a.W + ix0) * pix_bytes + src_piece;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W)
                        a_ok[i] |= 1u << (ky * 3 + kx);
        }
    }
    unsigned b_off[BROWS1];
#pragma unroll
    for (int i = 0; i < BROWS1; ++i) b_off[i] = (unsigned)(ld_r + LDROWS * i) * (unsigned)K2 * ES + src_piece;

    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;
    int ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_cc = 0;
    auto stage = [&](int buf) {
        const unsigned xs = (unsigned)(ld_ky * a.W + ld_kx) * pix_bytes + (unsigned)ld_cc * CHUNK_BYTES;
        const unsigned ws = (unsigned)ld_tap * pix_bytes + (unsigned)ld_cc * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const unsigned off = ((a_ok[i] >> ld_tap) & 1u) ? a_off[i] + xs : OOB;
            char* dst = As + ((unsigned)buf * BM + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BROWS1; ++i) {
            char* dst = Bs + ((unsigned)buf * MID + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, (lds_void*)dst, 16, b_off[i] + ws, 0, 0, 0);
        }
        if (++ld_kx == 3) {
            ld_kx = 0;
            ++ld_ky;
        }
        if (++ld_tap == 9) {
            ld_tap = 0;
            ld_ky = 0;
            ld_kx = 0;
            ++ld_cc;
        }
    };

    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;

    // FrozenBN constants in the accumulator layout (a lane owns one channel per 32-column block), loaded ahead of every
    // counted wait below: a later load would force a vmcnt(0) on the shortcut prefetch of the FAST epilogue
    float sc2[NT1], bi2[NT1];
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
        const int n = wn * 32 * NT1 + j * 32 + (lane & 31);
        sc2[j] = a.scale2 ? a.scale2[n] : 1.f;
        bi2[j] = a.bias2 ? a.bias2[n] : 0.f;
    }
    constexpr int NP = MID / 32;                       // 128-channel output pieces (COUT = 4 MID)
    [[maybe_unused]] float sc3[NP][2], bi3[NP][2];
    if constexpr (FAST) {
#pragma unroll
        for (int nc = 0; nc < NP; ++nc)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nc * 128 + wn * 64 + j * 32 + (lane & 31);
                sc3[nc][j] = a.scale3 ? a.scale3[n] : 1.f;
                bi3[nc][j] = a.bias3 ? a.bias3[n] : 0.f;
            }
    }

    f32x16 acc2[MT][NT1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

#if defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 1)      // timing builds only (tools/tail_probe.py): no phase 1 (the 3x3)
    if (false)
#endif
    {
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }
#if defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 1)
    for (int it = 0; it < 0; ++it) {
#else
    for (int it = 0; it < nit; ++it) {
#endif
        const int cur = it & 1;
        if (it + 1 < nit) stage(cur ^ 1);
        const char* Ab = &As[(cur * BM + wm * 32 * MT) * CHUNK_BYTES];
        const char* Bb = &Bs[(cur * MID + wn * 32 * NT1) * CHUNK_BYTES];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[MT], fb[NT1];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int j = 0; j < NT1; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT1; ++j) Elem<T>::mma(fa[i], fb[j], acc2[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next k-step has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's fragment reads of buf[cur] are done
        __builtin_amdgcn_s_barrier();
    }

    // ---- the 1x1's filters of output piece `nc` (128 channels x MID): KC pieces of [128 rows][128 B] -------------------------
    auto stage_w3 = [&](int nc) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int i = 0; i < BROWS3; ++i) {
                const unsigned n = (unsigned)nc * 128u + (unsigned)(ld_r + LDROWS * i);
                const unsigned off = n < (unsigned)a.COUT ? n * pix_bytes + (unsigned)kc * CHUNK_BYTES + src_piece : OOB;
                char* dst = W3s + ((unsigned)kc * 128 + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
    };
    stage_w3(0);                                       // the phase-1 stages are free (barrier above); lands under phase 2

    // FAST: the shortcut rows this lane finishes (8 channels = 16 B of 4 rows per 128-channel piece) are requested here, two
    // pieces ahead of their use — behind the filters in the queue, so that the counted wait below does not include them.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NDMA3 = KC * BROWS3;                 // DMA instructions of one stage_w3 per wave
    const int er = lane >> 3, ec = (lane & 7) * 8;     // FAST epilogue: row within a pass of 8, first of the lane's 8 channels
    [[maybe_unused]] u32x4 rb[2][4];
    [[maybe_unused]] unsigned eoff[4];
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rrsrc, yrsrc;
    [[maybe_unused]] auto load_res = [&](int nc, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            rb[slot][k] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, eoff[k] == OOB ? OOB : eoff[k] + (unsigned)nc * 256u, 0, 0);
    };
    // fp32 FAST: one register set of 8 rows x 4 channels per 128-channel piece, requested one piece ahead (the fp32 MFMAs of a
    // piece take ~2 us: cover enough), with two sets the kernel would leave the 170-register budget of three blocks per CU
    const int er4 = lane >> 4, ec4 = (lane & 15) * 4;
    [[maybe_unused]] u32x4 rbf[8];
    [[maybe_unused]] unsigned eoff8[8];
    [[maybe_unused]] auto load_res4 = [&](int nc) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            rbf[k] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, eoff8[k] == OOB ? OOB : eoff8[k] + (unsigned)nc * 512u, 0, 0);
    };
    if constexpr (FAST && sizeof(T) == 4) {
        const int bytes = (int)((size_t)M * a.COUT * ES);
        rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.y), 0, bytes, 0x00020000);
        yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = m0 + wm * 32 + 4 * k + er4;
            eoff8[k] = m < M ? ((unsigned)m * (unsigned)a.COUT + (unsigned)(wn * 64 + ec4)) * ES : OOB;
        }
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
        if (a.res) load_res4(0);
#endif
    }
    if constexpr (FAST && sizeof(T) == 2) {
        const int bytes = (int)((size_t)M * a.COUT * ES);
        rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.res ? a.res : a.y), 0, bytes, 0x00020000);
        yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + wm * 32 + 8 * k + er;
            eoff[k] = m < M ? ((unsigned)m * (unsigned)a.COUT + (unsigned)(wn * 64 + ec)) * ES : OOB;
        }
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
        if (a.res) {
            load_res(0, 0);
            load_res(1, 1);
        }
#endif
    }

    // ---- phase 2: mid tile = ReLU(acc2 * scale2 + bias2) → LDS, in the image the DMA builds from a [row][MID] tensor --------
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
        const int n = wn * 32 * NT1 + j * 32 + (lane & 31);            // mid channel of this lane's accumulator column
        const float sc = sc2[j], bi = bi2[j];
        const unsigned c = (unsigned)n / KE, e = (unsigned)n % KE;
        const unsigned piece = (e * ES) >> 4, inb = (e * ES) & 15;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float t = acc2[i][j][q];
                    if (a.scale2) t = __fmul_rn(t, sc);
                    if (a.bias2) t = __fadd_rn(t, bi);
                    t = t > 0.f ? t : 0.f;
                    const unsigned row = (unsigned)(wm * 32 * MT + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5));
                    *reinterpret_cast<float*>(T2s + (c * BM + row) * CHUNK_BYTES + ((piece ^ ((row >> 1) & 7)) << 4) + inb) = t;
                }
            } else {
                // fp16: neighbouring lanes hold neighbouring channels of the same rows — swap one value (DPP quad_perm) so that
                // an even lane owns rows R of the channel pair and an odd lane rows R + 1: 4-B stores of packed pairs
                const int odd = lane & 1;
                const unsigned inb2 = ((e & ~1u) * ES) & 15;
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    float t0 = acc2[i][j][q], t1 = acc2[i][j][q + 1];
                    if (a.scale2) { t0 = __fmul_rn(t0, sc); t1 = __fmul_rn(t1, sc); }
                    if (a.bias2) { t0 = __fadd_rn(t0, bi); t1 = __fadd_rn(t1, bi); }
                    t0 = t0 > 0.f ? t0 : 0.f;
                    t1 = t1 > 0.f ? t1 : 0.f;
                    const float give = odd ? t0 : t1;
                    const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                        0, __builtin_bit_cast(int, give), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false));
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    f16x2 pk;
                    pk[0] = (_Float16)(odd ? got : t0);
                    pk[1] = (_Float16)(odd ? t1 : got);
                    const unsigned row = (unsigned)(wm * 32 * MT + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5) + odd);
                    *reinterpret_cast<f16x2*>(T2s + (c * BM + row) * CHUNK_BYTES + ((piece ^ ((row >> 1) & 7)) << 4) + inb2) = pk;
                }
            }
        }
    }
    if constexpr (FAST) {
        if (a.res) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");        // the filters; the 8 shortcut loads stay in flight
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // mid tile visible to every wave, filters of piece 0 landed

#if defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 2)      // timing builds only: stop after phase 2 (no 1x1, no epilogue)
    if (tid == 0 && m0 == 0x7fffff00) static_cast<T*>(a.y)[0] = *reinterpret_cast<const T*>(T2s);
    return;
#endif
    // ---- phase 3: the 1x1, 128 output channels per piece -------------------------------------------------------------------------
    if constexpr (FAST && sizeof(T) == 4) {
        // fp32 form of the wave-private epilogue below: 4 channels (16 B) of 8 rows per lane and piece; the staging tiles lie on
        // the filters the piece has just read (block barrier before and after), the next piece's filters and shortcut rows are
        // requested behind it and awaited BEFORE this piece's stores are issued — so the counted wait never has a store to sit out.
        float* Ew = reinterpret_cast<float*>(Epi) + (wave * 16 * 64);
        auto piece4 = [&](auto nc_c) __attribute__((always_inline)) {
            constexpr int nc = decltype(nc_c)::value;
            f32x16 acc3[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const char* Ab = T2s + (kc * BM + wm * 32) * CHUNK_BYTES;
                const char* Bb = W3s + (kc * 128 + wn * 64) * CHUNK_BYTES;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(Ab + frag_off[kk]);
                    f32x4 fb[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) Elem<T>::mma(fa, fb[j], acc3[j]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // every wave has read this piece's filters: the region is the staging tiles' now
            f32x4 out[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lr = e + 8 * g + 4 * (lane >> 5);
                        const int flip = (((lr >> 2) & 1) << 5) ^ ((lr & 1) << 2);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float t = acc3[j][4 * (2 * h + g) + e];
                            if (a.scale3) t = __fmul_rn(t, sc3[nc][j]);
                            if (a.bias3) t = __fadd_rn(t, bi3[nc][j]);
                            Ew[lr * 64 + ((j * 32 + (lane & 31)) ^ flip)] = t;
                        }
                    }
                __builtin_amdgcn_wave_barrier();       // staged in one lane layout, read back in another (same wave): pin the order
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int lr = 4 * p + er4;
                    const int flip = (((lr >> 2) & 1) << 5) ^ ((lr & 1) << 2);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Ew[lr * 64 + (ec4 ^ flip)]);
                    const f32x4 rs = __builtin_bit_cast(f32x4, rbf[4 * h + p]);
                    f32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = v[q];
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
                        if (a.res) t = __fadd_rn(t, rs[q]);
#endif
                        o[q] = t > 0.f ? t : 0.f;
                    }
                    out[4 * h + p] = o;
                }
                __builtin_amdgcn_wave_barrier();       // the next half's staging writes stay behind these reads
            }
            if constexpr (nc + 1 < NP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // every wave is done with its staging tile: the next filters may land on it
                stage_w3(nc + 1);
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
                if (a.res) {
                    load_res4(nc + 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the filters; the 8 shortcut loads behind them stay in flight
                } else
#endif
                {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 8))
#pragma unroll
            for (int k = 0; k < 8; ++k)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out[k]), yrsrc, eoff8[k] == OOB ? OOB : eoff8[k] + (unsigned)nc * 512u, 0, 0);
#else
            if (m0 == 0x7fffff00) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, out[0] + out[7]), yrsrc, 0, 0, 0);
#endif
        };
        static_for<NP>(piece4);
        return;
    }
    if constexpr (FAST && sizeof(T) == 2) {
        // Wave-private wide epilogue. A wave owns 32 rows x 64 channels of a piece; it transposes them 16 rows at a time through
        // its OWN 4 KB of LDS (no block barrier; scale and bias applied on the way in), finishes 8 channels of a row per lane in
        // fp32 with the same single IEEE operations as conv_epilogue (+ shortcut, ReLU, one rounding to fp16) and stores 16 B per
        // lane: 128-B row segments per 8 lanes. The shortcut was requested before phase 2 (and two pieces ahead after that), so that no load
        // is issued and awaited inside the epilogue; the only block barriers left are the two around the filter restaging.
        float* Ew = reinterpret_cast<float*>(Epi) + (wave * 16 * 64);
        auto piece = [&](auto nc_c) __attribute__((always_inline)) {
            constexpr int nc = decltype(nc_c)::value;
            f32x16 acc3[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const char* Ab = T2s + (kc * BM + wm * 32) * CHUNK_BYTES;
                const char* Bb = W3s + (kc * 128 + wn * 64) * CHUNK_BYTES;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(Ab + frag_off[kk]);
                    f32x4 fb[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) Elem<T>::mma(fa, fb[j], acc3[j]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // every wave has read this piece's filters
            if constexpr (nc + 1 < NP) stage_w3(nc + 1);
            u32x4 out[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // rows 16 h .. 16 h + 15 of the wave's tile: accumulator registers 8 h .. 8 h + 7 of both column blocks. Bank
                // picture: a row is 64 floats = all 64 banks; rows 4-7 / 12-15 (the upper half-wave's) are flipped by 32
                // columns and odd rows by one float4, so that both the 4-B writes and the 16-B reads are conflict-free.
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lr = e + 8 * g + 4 * (lane >> 5);
                        const int flip = (((lr >> 2) & 1) << 5) ^ ((lr & 1) << 2);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float t = acc3[j][4 * (2 * h + g) + e];        // scale and bias here: one channel per lane and block
                            if (a.scale3) t = __fmul_rn(t, sc3[nc][j]);
                            if (a.bias3) t = __fadd_rn(t, bi3[nc][j]);
                            Ew[lr * 64 + ((j * 32 + (lane & 31)) ^ flip)] = t;
                        }
                    }
                __builtin_amdgcn_wave_barrier();       // staged in one lane layout, read back in another (same wave): pin the order
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int lr = 8 * p + er;
                    const int flip = (((lr >> 2) & 1) << 5) ^ ((lr & 1) << 2);
                    f32x4 v[2];
                    v[0] = *reinterpret_cast<const f32x4*>(&Ew[lr * 64 + (ec ^ flip)]);
                    v[1] = *reinterpret_cast<const f32x4*>(&Ew[lr * 64 + ((ec + 4) ^ flip)]);
                    if constexpr (nc == 0) {
                        // queue: [shortcut piece 0][shortcut piece 1][filters of piece 1]: piece 0's rows have landed when at most
                        // the later ones are outstanding (loads return in order); later pieces were awaited by the vmcnt(0) below
                        if (h == 0 && p == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + (NP > 1 ? NDMA3 : 0)) : "memory");
                    }
                    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
                    const f16x8v rs = __builtin_bit_cast(f16x8v, rb[nc & 1][2 * h + p]);
                    f16x8v o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float t = v[q >> 2][q & 3];
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
                        if (a.res) t = __fadd_rn(t, (float)rs[q]);
#endif
                        t = t > 0.f ? t : 0.f;
                        o[q] = (_Float16)t;
                    }
                    out[2 * h + p] = __builtin_bit_cast(u32x4, o);
                }
                __builtin_amdgcn_wave_barrier();       // the next half's staging writes stay behind these reads
            }
            if constexpr (nc + 1 < NP) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next piece's filters (and its shortcut rows)
                __builtin_amdgcn_s_barrier();
            }
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 8))
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_buffer_store_b128(out[k], yrsrc, eoff[k] == OOB ? OOB : eoff[k] + (unsigned)nc * 256u, 0, 0);
#else
            if (m0 == 0x7fffff00) __builtin_amdgcn_raw_buffer_store_b128(out[0] + out[1] + out[2] + out[3], yrsrc, 0, 0, 0);
#endif
            if constexpr (nc + 2 < NP) {
#if !(defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4))
                if (a.res) load_res(nc + 2, nc & 1);
#endif
            }
        };
        static_for<NP>(piece);
        return;
    }
    ConvArgs e{};
    e.y = a.y; e.res = a.res; e.scale = a.scale3; e.bias = a.bias3;
#if defined(TD_TAIL_DIAG) && (TD_TAIL_DIAG & 4)      // timing builds only: no shortcut read
    e.res = nullptr;
#endif
    e.Cout = a.COUT; e.relu = 1; e.out_mode = 0; e.res_shift = 0; e.Ho = a.H; e.Wo = a.W;
    const int npieces = (a.COUT + 127) / 128;
    for (int nc = 0; nc < npieces; ++nc) {
        f32x16 acc3[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const char* Ab = T2s + (kc * BM + wm * 32 * MT) * CHUNK_BYTES;
            const char* Bb = W3s + (kc * 128 + wn * 64) * CHUNK_BYTES;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 fa[MT], fb[2];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Elem<T>::mma(fa[i], fb[j], acc3[i][j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave has read this piece's filters (and the staging tile is free)
        const bool more = nc + 1 < npieces;
        if (OVERLAP && more) stage_w3(nc + 1);         // arrives while this piece is finished below
        conv_epilogue<T, T, MT, 2, 2, 2, 1, false>(e, acc3, Epi, M, m0, nc * 128, tid, lane, wm, wn);
        if (more) {
            if (!OVERLAP) {
                __syncthreads();                       // the staging tile (aliased with the filters) has been read
                stage_w3(nc + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// BPC = blocks per CU the LDS footprint allows (__launch_bounds__' second argument is waves per SIMD = BPC for 4-wave blocks)
template <typename T, int MID, bool OVERLAP, int MT, bool FAST, int BPC>
__global__ __launch_bounds__(256, BPC)
void bottleneck_tail_kernel(const TailArgs a) {
    static_assert(BPC >= 1 && BPC * TailGeom<T, MID, OVERLAP, MT, FAST>::LDS_BYTES <= 160 * 1024, "LDS footprint does not allow that many blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[TailGeom<T, MID, OVERLAP, MT, FAST>::LDS_BYTES];
    bottleneck_tail_body<T, MID, OVERLAP, MT, FAST>(a, lds);
}

template <typename T, int MID, bool OVERLAP, int MT, bool FAST = false>
td_status launch_tail(const TailArgs& a, hipStream_t stream) {
    typedef TailGeom<T, MID, OVERLAP, MT, FAST> G;
    const int tiles = td_cdiv(a.M, G::BM);
    hipLaunchKernelGGL((bottleneck_tail_kernel<T, MID, OVERLAP, MT, FAST, G::BPC>), dim3(tiles), dim3(256), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

}  // namespace

bool bottleneck_tail_ok(int precision, int mid, int cout) {
    if (cout != 4 * mid) return false;
    if (precision == TD_PRECISION_FP32) return mid == 64;            // larger fp32 3x3 layers take the Winograd path
    return mid == 64 || mid == 128;
}

td_status bottleneck_tail_launch(const TailArgs& a, int precision, hipStream_t stream) {
    TD_REQUIRE(bottleneck_tail_ok(precision, a.MID, a.COUT), "bottleneck tail: unsupported shape (precision %d, mid %d, out %d)", precision, a.MID, a.COUT);
    TD_REQUIRE(a.M == a.B * a.H * a.W && a.M > 0, "bottleneck tail: M must be B*H*W");
    const size_t es = precision == TD_PRECISION_FP16 ? 2 : 4;
    TD_REQUIRE((size_t)a.M * a.MID * es < 0xfffffff0ull - (1u << 20), "bottleneck tail: input tensor must stay below 4 GB (32-bit buffer offsets)");
    // rows per block: 64 (three or four blocks per CU: more independent latency chains in flight) or 128 (half the filter
    // re-reads); TD_TAIL_BM picks for experiments, the default is what measured faster per shape (profiles/)
    static const int forced = getenv("TD_TAIL_BM") ? atoi(getenv("TD_TAIL_BM")) : 0;
    const int bm = forced == 64 || forced == 128 ? forced : 64;
    static const int fast_env32 = getenv("TD_TAIL_FAST") ? atoi(getenv("TD_TAIL_FAST")) : 1;
    const bool fast32 = fast_env32 && (size_t)a.M * a.COUT * es < 0xfffffff0ull - (1u << 20);
    if (precision == TD_PRECISION_FP32) {
        if (bm == 64) return fast32 ? launch_tail<float, 64, false, 1, true>(a, stream) : launch_tail<float, 64, false, 1>(a, stream);
        return launch_tail<float, 64, false, 2>(a, stream);
    }
    // fp16, 64 rows: the wave-private wide epilogue (32-bit buffer offsets into the shortcut / output: below 4 GB)
    static const int fast_env = getenv("TD_TAIL_FAST") ? atoi(getenv("TD_TAIL_FAST")) : 1;
    const bool fast = fast_env && (size_t)a.M * a.COUT * es < 0xfffffff0ull - (1u << 20);
    if (a.MID == 64) {
        if (bm == 64) return fast ? launch_tail<_Float16, 64, false, 1, true>(a, stream) : launch_tail<_Float16, 64, false, 1>(a, stream);
        return launch_tail<_Float16, 64, true, 2>(a, stream);
    }
    if (bm == 64) return fast ? launch_tail<_Float16, 128, false, 1, true>(a, stream) : launch_tail<_Float16, 128, false, 1>(a, stream);
    return launch_tail<_Float16, 128, true, 2>(a, stream);
}
