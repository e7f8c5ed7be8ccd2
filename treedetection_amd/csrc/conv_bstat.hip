// conv_bs_kernel (tile id 33): FILTER-STATIONARY 1x1 convolution for the thin-K layers (K = Cin <= 256 fp16 / 128 fp32: the
// bottleneck "expand" 1x1s of res3 / res4 with their shortcut add, the projection shortcuts, FPN lateral 2).
//
// Why: these layers have 1 - 4 k-steps. Per-kernel PMC and the layer tables say what bounds them: not HBM (1.3 - 1.9x above their
// HBM floor) and not the MFMA pipe, but the bytes a CU can pull through its vector-memory path (~58 GB/s per CU measured, L2 hits
// included) times the latency of a block's load → contract → shortcut → store chain. A 64 x 128 block of conv_bd_kernel moves
// 32 KB of activations, 64 KB of FILTERS, 16 KB of shortcut and 16 KB of output (K = 256, fp16): half of its traffic is the filter
// bank, re-read by every 64 rows. Here a block keeps its filters in REGISTERS for its whole life and streams row tiles past them:
//   * one block per CU, 8 consumer waves x 32 output channels = 256 channels; a wave loads its 32 x K filter slab once
//     (fragment order, conv_bd_pack: 16 NIT registers) and contracts every row tile of the block against it;
//   * 2 producer waves do nothing but LDS-DMA the activation rows of the block's NEXT row tiles into a ring of D + 1 stages
//     (XOR-swizzled image of conv_igemm_kernel), D tiles ahead. Their vmcnt queues hold only those DMAs, so the counted wait that
//     releases a tile is exact — and the consumers' queues hold only their own shortcut loads and output stores: the loads of the
//     NEXT tile's shortcut rows are issued before this tile's MFMAs (two register sets), nothing in a consumer ever waits for an
//     activation load (a single queue would return the shortcut rows behind D tiles of DMA);
//   * one s_barrier per row tile: the producers arrive when the tile has landed, the consumers when they have finished the
//     previous one (its ring stage is then free for the producers' next issue);
//   * the epilogue is wave-private: a wave transposes its 32 MT x 32 accumulator tile 16 rows at a time through its own 2.5 KB of
//     LDS (scale and bias applied on the way in, one channel per lane), adds the shortcut, applies ReLU, rounds once and stores
//     16 B per lane. No block barrier, no LDS shared with other waves.
// CU-side bytes per 64 rows x 256 channels (K = 256, fp16): 32 KB + 32 KB + 32 KB instead of 2 x 128 KB.
// Same k order (channel chunk ascending, four 16-B pieces per chunk), same MFMA, the same single IEEE operations in the epilogue
// as conv_epilogue: BIT-IDENTICAL to every other tile (tests/test_conv_gpu.py), so the tuner may choose it by measurement.
#include "common.h"
#include "conv_tiles.h"
#include <type_traits>

namespace {

template <typename T, int MT, int NIT>
struct BsGeom {
    static constexpr int ES = sizeof(T);
    static constexpr int BM = 32 * MT;
    static constexpr int CONS = 8, PROD = 2, THREADS = 64 * (CONS + PROD);
    static constexpr int BN = 32 * CONS;
    static constexpr int STAGE = NIT * BM * CHUNK_BYTES;              // one row tile: [NIT][BM][128 B]
    static constexpr int IPT = NIT * BM / 8 / PROD;                   // DMA instructions (8 rows each) per tile and producer wave
    static constexpr int EWF = 16 * 40;                               // floats of a wave's staging tile: 16 rows, stride 40
    static constexpr int RING_MAX = (160 * 1024 - CONS * EWF * 4) / STAGE;
    static constexpr int D = RING_MAX - 1 < 6 ? RING_MAX - 1 : 6;     // row tiles in flight
    static constexpr int S = D + 1;
    static constexpr int LDS_BYTES = S * STAGE + CONS * EWF * 4;
    static constexpr int CPL = 16 / ES;                               // output channels per lane (16 B)
    static constexpr int LPR = 32 / CPL;                              // lanes per output row of a wave (32 channels)
    static constexpr int RPP = 64 / LPR;                              // rows per pass
    static constexpr int PASSES = 16 / RPP;                           // passes per 16-row quarter
    static constexpr int NQ = BM / 16;
    static constexpr int NRES = NQ * PASSES;                          // 16-B shortcut loads / output stores per lane and tile
    static_assert(D >= 1 && (D - 1) * IPT <= 63 && IPT >= 1, "producer vmcnt immediates");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <typename T, int MT, int NIT>
__global__ __launch_bounds__(640, 3) void conv_bs_kernel(const ConvArgs a, const int G) {
    typedef BsGeom<T, MT, NIT> Gm;
    constexpr int ES = Gm::ES, BM = Gm::BM, S = Gm::S, D = Gm::D, IPT = Gm::IPT, NRES = Gm::NRES, CPL = Gm::CPL;
    __shared__ __attribute__((aligned(16))) char lds[Gm::LDS_BYTES];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0xfffffff0u;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.M;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (a.Cout + Gm::BN - 1) / Gm::BN;
    // block → (column tile, row group). blockIdx round-robins over the 8 XCDs; an XCD owns the CONTIGUOUS range of row tiles
    // xcd_remap gives it in every other conv kernel (the layer before wrote those rows through this XCD's L2, the layer after reads
    // them through it), its G / 8 row groups interleave inside that range, and the tiles_n blocks of a group sit on the same XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % tiles_n, gl = slot / tiles_n, tstep = G >> 3;
    const int xq = tiles_m >> 3, xr = tiles_m & 7;
    const int xcnt = xq + (xcd < xr ? 1 : 0), xstart = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
    const int nk = gl < xcnt ? (xcnt - 1 - gl) / tstep + 1 : 0;       // row tiles t0, t0 + tstep, ... of this block
    if (nk == 0) return;
    const int t0 = xstart + gl;
    const int n0 = tn * Gm::BN;

    if (wave >= Gm::CONS) {
        // ---- producers: activation rows → ring, D tiles ahead ----
        const int p = wave - Gm::CONS;
        const unsigned pix_bytes = (unsigned)a.Cin * ES;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)((size_t)M * pix_bytes), 0x00020000);
        const int lr = lane >> 3, lc = lane & 7;
        auto issue_tile = [&](int k) __attribute__((always_inline)) {
            const int m0 = (t0 + k * tstep) * BM;
            char* base = lds + (k % S) * Gm::STAGE;
#pragma unroll
            for (int c = 0; c < NIT; ++c)
#pragma unroll
                for (int j = 0; j < BM / 8 / Gm::PROD; ++j) {
                    const int r0 = 8 * (j * Gm::PROD + p);           // first of the 8 rows of this instruction
                    const int row = r0 + lr, m = m0 + row;
                    const unsigned piece = (unsigned)(lc ^ ((row >> 1) & 7));
                    const unsigned off = m < M ? (unsigned)m * pix_bytes + (unsigned)c * CHUNK_BYTES + piece * 16u : OOB;
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 8))       // timing builds (tools/bs_probe.py): 8 = no activation loads
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(base + (c * BM + r0) * CHUNK_BYTES), 16, off, 0, 0, 0);
#else
                    if (off == 0x12345u) lds[0] = 1;
#endif
                }
        };
        for (int k = 0; k < D && k < nk; ++k) issue_tile(k);
        for (int k = 0; k < nk; ++k) {
            const int out = nk - k < D ? nk - k : D;                 // tiles outstanding: k .. k + out - 1
            // tile k has landed when at most the younger tiles' instructions are outstanding (a queue of DMAs only: in order)
            switch (out - 1) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT <= 63 ? 2 * IPT : 0) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPT <= 63 ? 3 * IPT : 0) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * IPT <= 63 ? 4 * IPT : 0) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * IPT <= 63 ? 5 * IPT : 0) : "memory"); break;
            }
            __builtin_amdgcn_s_barrier();                            // consumers: tile k is in LDS; they are done with tile k - 1
            if (k + D < nk) issue_tile(k + D);                       // into the stage tile k - 1 occupied
        }
        return;
    }

    // ---- consumers ----
    const int w = wave;
    const int hi = lane >> 5;
    const int col = n0 + 32 * w + (lane & 31);                        // this lane's output channel in the accumulator layout
    // the wave's filter slab: 32 channels x K, fragment order [tile32][k-chunk][kk][lane][16 B]
    const int ntiles32 = (a.Cout + 31) / 32;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w_frag), 0, (int)((size_t)ntiles32 * NIT * 4096), 0x00020000);
    f32x4 fb[NIT][4];
    {
        const int t32 = n0 / 32 + w;
#pragma unroll
        for (int c = 0; c < NIT; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned off = t32 < ntiles32 ? ((unsigned)(t32 * NIT + c) * 4096u + (unsigned)kk * 1024u + (unsigned)lane * 16u) : OOB;
                fb[c][kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
            }
    }
    float sc = 1.f, bi = 0.f;
    if (col < a.Cout) {
        if (a.scale) sc = a.scale[col];
        if (a.bias) bi = a.bias[col];
    }
    const unsigned swz = (lane >> 1) & 7;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;

    // output layout of the epilogue: a lane finishes CPL channels (16 B) of NRES rows of the wave's BM x 32 tile
    const int er = lane / Gm::LPR, ec = (lane % Gm::LPR) * CPL;
    const int n_out = n0 + 32 * w + ec;
    const bool col_ok = n_out < a.Cout;                                // Cout % 8 == 0: the whole 16-B piece is in range
    const bool has_res = a.res != nullptr;
    const size_t res_bytes = (a.res_shift ? (size_t)a.B * (a.Ho >> 1) * (a.Wo >> 1) : (size_t)M) * a.Cout * ES;
    const unsigned long long rbase = (unsigned long long)(has_res ? a.res : a.y);
    i32x4 rdesc;
    rdesc[0] = __builtin_amdgcn_readfirstlane((int)(rbase & 0xffffffffu));
    rdesc[1] = __builtin_amdgcn_readfirstlane((int)((rbase >> 32) & 0xffffu));
    rdesc[2] = __builtin_amdgcn_readfirstlane((int)res_bytes);
    rdesc[3] = 0x00020000;
    const int hw = a.Ho * a.Wo;
    // byte offset of (row RPP q + er of a tile, this lane's channels) from the tile's first row: the output and the same-size
    // shortcut are addressed as this + m0 Cout ES, rows past M fall off the end of the buffer (loads return 0, stores are dropped),
    // lanes past Cout carry the out-of-range offset — no per-row compare, no 64-bit address arithmetic
    unsigned row_off[NRES];
#pragma unroll
    for (int q = 0; q < NRES; ++q) row_off[q] = col_ok ? ((unsigned)(Gm::RPP * q + er) * (unsigned)a.Cout + (unsigned)n_out) * ES : OOB;
    const bool walk = a.Wo >= BM && a.Ho >= 2;                         // half-resolution shortcut: (image, y, x) of a row by at most one wrap per axis
    f32x4 rb[2][NRES];
    auto issue_res = [&](auto set_c, int k) {
        constexpr int SET = decltype(set_c)::value;
        auto& rb_ = rb;                                               // (asm operands alone do not capture in a generic lambda)
        const i32x4& rdesc_ = rdesc;
        const int m0 = __builtin_amdgcn_readfirstlane((t0 + k * tstep) * BM);
        const unsigned tile_off = (unsigned)m0 * (unsigned)a.Cout * ES;
        int b0 = 0, oy0 = 0, ox0 = 0;
        if (a.res_shift) {
            b0 = m0 / hw;
            const int rem = m0 - b0 * hw;
            oy0 = rem / a.Wo;
            ox0 = rem - oy0 * a.Wo;
        }
#pragma unroll
        for (int q = 0; q < NRES; ++q) {
            unsigned off = row_off[q] == OOB ? OOB : row_off[q] + tile_off;
            if (a.res_shift) {
                const int r = Gm::RPP * q + er, m = m0 + r;           // rows RPP q + er (quarter q / PASSES, pass q % PASSES)
                int b = b0, oy = oy0, ox = ox0 + r;
                if (walk) {
                    if (ox >= a.Wo) { ox -= a.Wo; ++oy; }
                    if (oy >= a.Ho) { oy -= a.Ho; ++b; }
                } else {
                    b = m / hw;
                    const int rem = m - b * hw;
                    oy = rem / a.Wo;
                    ox = rem - oy * a.Wo;
                }
                off = m < M && col_ok ? (unsigned)(((b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n_out) * ES : OOB;
            }
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb_[SET][q]) : "v"(off), "s"(rdesc_) : "memory");
        }
    };
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((size_t)M * a.Cout * ES), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    float* Ew = reinterpret_cast<float*>(lds + S * Gm::STAGE) + w * Gm::EWF;

    auto tile = [&](auto set_c, int k) {
        constexpr int SET = decltype(set_c)::value;
        auto& rb_ = rb;
        __builtin_amdgcn_s_barrier();                                  // tile k has landed
        if (has_res && k + 1 < nk) issue_res(std::integral_constant<int, SET ^ 1>{}, k + 1);
        f32x16 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const char* Ab = lds + (k % S) * Gm::STAGE;
#pragma unroll
        for (int c = 0; c < NIT; ++c)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 fa[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + (c * BM + i * 32) * CHUNK_BYTES + frag_off[kk]);
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 2))       // 2 = no MFMAs
#pragma unroll
                for (int i = 0; i < MT; ++i) Elem<T>::mma(fa[i], fb[c][kk], acc[i]);
#else
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][kk] += fa[i][0] * fb[c][kk][0];
#endif
            }
        // the shortcut rows of THIS tile were requested one tile ago; younger in this wave's queue: the next tile's rows only
        // (stores are older or not issued yet: an allowance that counts younger LOADS only is safe whatever order stores retire in)
        if (has_res) {
            if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < NRES; ++q) asm volatile("" : "+v"(rb_[SET][q]));
        }
        const unsigned tile_off = (unsigned)__builtin_amdgcn_readfirstlane((t0 + k * tstep) * BM) * (unsigned)a.Cout * ES;
#pragma unroll
        for (int qq = 0; qq < Gm::NQ; ++qq) {
            // rows 16 qq .. 16 qq + 15 of the wave's tile = accumulator registers 8 (qq & 1) .. + 7 of row block qq >> 1
            const int i = qq >> 1, hq = qq & 1;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][4 * (2 * hq + gg) + e];
                    if (a.scale) t = __fmul_rn(t, sc);
                    if (a.bias) t = __fadd_rn(t, bi);
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 4))       // 4 = no transposition through LDS (wrong layout)
                    Ew[(e + 8 * gg + 4 * hi) * 40 + (lane & 31)] = t;
#else
                    acc[i][4 * (2 * hq + gg) + e] = t;
#endif
                }
            // wave-private staging: written in the accumulator lane layout, read back transposed by OTHER lanes of the same wave.
            // DS operations of a wave retire in order; the wave barrier (no instruction) keeps the compiler from moving the
            // reads above the writes, or the next quarter's writes above these reads
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < Gm::PASSES; ++p) {
                const int lrow = Gm::RPP * p + er;
                float v[CPL];
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 4))
#pragma unroll
                for (int h = 0; h < CPL / 4; ++h) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&Ew[lrow * 40 + ec + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * h + e] = t[e];
                }
#else
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] = acc[i][(8 * hq + 4 * p + e) & 15];
#endif
                const int q = qq * Gm::PASSES + p;
                const unsigned yoff = row_off[q] == OOB ? OOB : row_off[q] + tile_off;
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e];
                        if (has_res) t = __fadd_rn(t, rb[SET][q][e]);
                        if (a.relu) t = t > 0.f ? t : 0.f;
                        v[e] = t;
                    }
                    const f32x4 o = {v[0], v[1], v[2], v[3]};
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 1))       // 1 = no output stores
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yoff, 0, 0);
#else
                    if (yoff == 0x12345u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yoff, 0, 0);
#endif
                } else {
                    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
                    const f16x8v rs = __builtin_bit_cast(f16x8v, rb[SET][q]);
                    f16x8v o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = v[e];
                        if (has_res) t = __fadd_rn(t, (float)rs[e]);
                        if (a.relu) t = t > 0.f ? t : 0.f;
                        o[e] = (_Float16)t;
                    }
#if !(defined(TD_BS_DIAG) && (TD_BS_DIAG & 1))       // 1 = no output stores
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yoff, 0, 0);
#else
                    if (yoff == 0x12345u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yoff, 0, 0);
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();       // the next quarter's staging writes stay behind these reads
        }
    };
    if (has_res) issue_res(std::integral_constant<int, 0>{}, 0);
    for (int k = 0; k < nk; k += 2) {
        tile(std::integral_constant<int, 0>{}, k);
        if (k + 1 < nk) tile(std::integral_constant<int, 1>{}, k + 1);
    }
}

template <typename T, int MT, int NIT>
td_status launch_bs(const ConvArgs& a, hipStream_t stream) {
    typedef BsGeom<T, MT, NIT> Gm;
    const int tiles_n = td_cdiv(a.Cout, Gm::BN);
    // one block per CU: G row groups per column tile, a multiple of 8 (the XCD count) — 256 CUs on MI355X
    static const int num_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 64) n = 256;
        return n;
    }();
    int G = (num_cu / tiles_n) & ~7;
    if (G < 8) G = 8;
    hipLaunchKernelGGL((conv_bs_kernel<T, MT, NIT>), dim3(tiles_n * G), dim3(Gm::THREADS), 0, stream, a, G);
    TD_KERNEL_CHECK();
    return TD_OK;
}

}  // namespace

bool conv_bs_ok(const ConvArgs& a, int precision) {
    if (precision != TD_PRECISION_FP16 && precision != TD_PRECISION_FP32) return false;
    const int es = precision == TD_PRECISION_FP16 ? 2 : 4, ke = 128 / es;
    if (!a.w_frag || a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.out_mode != 0 || a.batch_count > 1 || a.m_dyn || a.head_w || a.out_f32)
        return false;
    if (a.Cin % ke != 0 || a.Cout < 128 || a.Cout % 8 != 0 || a.Cout > 8 * 256) return false;
    const int nit = a.Cin / ke;
    if (!(nit == 1 || nit == 2 || nit == 4)) return false;
    const size_t lim = 0xfffffff0ull - (1u << 20);
    if ((size_t)a.M * a.Cin * es >= lim || (size_t)a.M * a.Cout * es >= lim) return false;
    return a.M > 0 && a.M == a.B * a.Ho * a.Wo && a.H == a.Ho && a.W == a.Wo;
}

td_status conv_bs_launch(const ConvArgs& a, int precision, hipStream_t stream) {
    TD_REQUIRE(conv_bs_ok(a, precision), "filter-stationary convolution: unsupported launch (1x1, stride 1, Cin <= 4 k-chunks, packed filters)");
    const int nit = a.Cin / (precision == TD_PRECISION_FP16 ? 64 : 32);
    if (precision == TD_PRECISION_FP16) {
        if (nit == 1) return launch_bs<_Float16, 2, 1>(a, stream);
        if (nit == 2) return launch_bs<_Float16, 2, 2>(a, stream);
        return launch_bs<_Float16, 2, 4>(a, stream);
    }
    if (nit == 1) return launch_bs<float, 2, 1>(a, stream);
    if (nit == 2) return launch_bs<float, 2, 2>(a, stream);
    return launch_bs<float, 1, 4>(a, stream);        // 32-row tiles: with 64 the two shortcut register sets (64 VGPRs) spill
}
