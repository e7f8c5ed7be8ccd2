// conv_bd_kernel: implicit-GEMM convolution whose FILTER fragments bypass LDS (tile ids 23 - 27, 29, 30; fp16 and, since the deep load
// pipelines, fp32 as well: the data path is counted in bytes — a k-chunk is 128 B, a fragment 16 B — and Elem<T> picks the MFMA).
//
// Why: the fp16 engine's mid-size layers (res4 / res5, FPN and RPN at p4 - p6, the 1x1 layers of res3) are bound by the
// LDS-DMA fill of their block tiles: a CU's `buffer_load ... lds` path moves ~60-70 GB/s (MI355X_MICROARCH.md "ldsdma-fill"),
// a k-step of the measured 64 x 128 tile stages 24 KB of which 16 KB are filter rows, and the MFMAs of that k-step take a
// quarter of the time the staging does. Plain vector loads from L2 run at about twice that rate per CU (L2 34.5 TB/s over
// 256 CUs), on a path that the LDS-DMA does not fill. So here
//   * the activations (A: BM rows x 128 B per k-step, shared by all waves of the block) keep conv_igemm_kernel's path — LDS-DMA
//     into the XOR-swizzled image, two stages, raw s_barrier;
//   * the filters (B) are read by each wave for ITSELF, straight into the registers the MFMA takes them from: the engine keeps
//     a second copy of every fp16 filter bank in FRAGMENT ORDER — [32-column tile][k-step][kk][lane][8 halves], exactly the
//     16 bytes lane `l` of a wave feeds to v_mfma_f32_32x32x16_f16 for that (tile, k-step, kk) — so one wave instruction is one
//     fully contiguous 1-KB read (no fragment-shaped gather: that is what costs TA cycles), issued one k-step ahead;
//   * the waves of a block tile N only (1 x WN waves, each 32 MT rows x 32 NT columns): no two waves want the same filter
//     fragment, so nothing is fetched twice, and the A tile — the only operand in LDS — is read by every wave.
// Per k-step a 64 x 256 block stages 8 KB through LDS-DMA and 32 KB through vector loads for 2 x 64 x 256 x 64 FLOP: half the
// L2 bytes per FLOP of the 64 x 128 tile, a sixth of its LDS-DMA bytes.
// Same k order (channel chunk outer, filter tap inner), same MFMA, same epilogue (conv_epilogue) as every other block tile:
// BIT-IDENTICAL results (tests/test_conv_gpu.py), so the engine's tuner may pick it per layer shape by measurement.
#include "common.h"
#include "conv_tiles.h"
#include <type_traits>

namespace {

// KS = k-chunks (128-B rows) per barrier interval: 1, or 2 ("super-steps": half the barriers and exposed round trips per MFMA —
// the per-kernel PMC view shows the waves of these tiles parked at s_waitcnt / s_barrier for 50-70 % of their cycles)
// DEEP = k-steps of loads kept in flight beyond the one being consumed (0: the simple form — everything of step it + 1 issued at
// the start of step it and waited for, vmcnt(0), at its end). The per-kernel PMC view and the layer table agree on what bounds
// these tiles: ONE k-step round trip — DMA issue → L2 → LDS → block-wide barrier — takes ~0.65 us even on an idle CU (res5-sized
// layers with one block per CU: 36 k-steps in 24 us) against 0.12 us of MFMAs, and with one step of prefetch nothing covers it.
// With DEEP = 3 the loads of step it + 3 are issued before step it is consumed: DEEP + 2 LDS stages of 8 KB (A only — that is
// what the filter-direct layout buys: the stages are small), DEEP + 1 register sets for the filter fragments, ONE barrier per
// k-step (the stage a DMA overwrites was consumed two barriers ago), and COUNTED s_waitcnt vmcnt: the filter loads are inline
// asm so that hipcc, which waits vmcnt(0) for any VGPR load it knows of next to an LDS-DMA, does not drain the pipeline.
// RES = the shortcut-prefetch form of the deep pipeline (its own instantiation: the extra live registers and the wider waits cost
// the layers without a shortcut 5-20 % when compiled into the same kernel)
template <typename T, typename TO, int MT, int NT, int WN, int KS, int DEEP, bool RES = false>
__device__ __forceinline__ void conv_bd_body(const ConvArgs& a, char* lds) {
    constexpr int THREADS = 64 * WN;
    constexpr int BM = 32 * MT, BN = 32 * NT * WN;
    constexpr int LDROWS = THREADS / 8;
    constexpr int AROWS = BM / LDROWS;
    static_assert(BM % LDROWS == 0 && AROWS >= 1, "tile / thread-count mismatch");
    constexpr int ES = sizeof(T), KE = Elem<T>::PER_CHUNK;      // a k-chunk is 128 B either way: 64 halves or 32 floats
    char* As = lds;                                   // [2 stages][KS chunks][BM][128 B]

    int M = a.M;
    if (a.m_dyn) {
        int md = *a.m_dyn * a.m_mul - a.m_off;
        md = md < 0 ? 0 : md;
        M = md < M ? md : M;
    }
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int nblk = tiles_m * tiles_n;
    if ((int)blockIdx.x >= nblk) return;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;       // wave = wn (WM = 1)
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const int cchunks = a.Cin / KE;
    const int ntaps = a.KH * a.KW;
    const int nit = ntaps * cchunks;
    const unsigned pix_bytes = (unsigned)a.Cin * ES;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.H * a.W * pix_bytes), 0x00020000);
    const int ntiles32 = (a.Cout + 31) / 32;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w_frag), 0, (int)((size_t)ntiles32 * nit * 4096), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;

    unsigned a_off[AROWS], a_ok[AROWS];
    const bool plain_rows = ntaps == 1 && a.stride == 1 && a.pad == 0;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + ld_r + LDROWS * i;
        a_off[i] = 0;
        a_ok[i] = 0;
        if (m < M && plain_rows) {
            a_off[i] = (unsigned)m * pix_bytes + src_piece;
            a_ok[i] = 1u;
        } else if (m < M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            a_off[i] = (unsigned)((b * a.H + iy0) * a.W + ix0) * pix_bytes + src_piece;
            for (int ky = 0; ky < a.KH; ++ky)
                for (int kx = 0; kx < a.KW; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)a.H && (unsigned)(ix0 + kx) < (unsigned)a.W)
                        a_ok[i] |= 1u << (ky * a.KW + kx);
        }
    }
    // this wave's filter fragments: 32-column tiles n0/32 + wave*NT + j; a tile past Cout reads beyond num_records → zeros
    unsigned w_off[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nt = n0 / 32 + wave * NT + j;
        w_off[j] = nt < ntiles32 ? (unsigned)nt * (unsigned)nit * 4096u + (unsigned)lane * 16u : OOB;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;
    int ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_cc = 0;
    auto stage_a = [&](int buf) {                     // buf = stage * KS + chunk
        const unsigned xs = (unsigned)(ld_ky * a.W + ld_kx) * pix_bytes + (unsigned)ld_cc * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const unsigned off = ((a_ok[i] >> ld_tap) & 1u) ? a_off[i] + xs : OOB;
            char* dst = As + ((unsigned)buf * BM + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
        if (++ld_kx == a.KW) {
            ld_kx = 0;
            ++ld_ky;
        }
        if (++ld_tap == ntaps) {
            ld_tap = 0;
            ld_ky = 0;
            ld_kx = 0;
            ++ld_cc;
        }
    };
    auto load_b = [&](f32x4 (&fb)[NT][4], int it) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned off = w_off[j] == OOB ? OOB : w_off[j] + (unsigned)it * 4096u + (unsigned)kk * 1024u;
                fb[j][kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0));
            }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;
    auto compute = [&](int buf, const f32x4 (&fb)[NT][4]) {
        const char* Ab = &As[buf * BM * CHUNK_BYTES];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) Elem<T>::mma(fa[i], fb[j][kk], acc[i][j]);
        }
    };

    if constexpr (DEEP > 0) {
        static_assert(KS == 1, "deep pipeline: one k-chunk per step");
        constexpr int NS = DEEP + 2;                  // LDS stages
        constexpr int NR = DEEP + 1;                  // register sets
        constexpr int OPS = NT * 4 + AROWS;           // vector-memory operations a wave issues per k-step
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        // the filter bank's buffer descriptor as four SGPRs for the inline-asm loads (base, stride 0, num_records, raw dword access)
        const unsigned long long wbase = (unsigned long long)a.w_frag;
        i32x4 wdesc;
        wdesc[0] = __builtin_amdgcn_readfirstlane((int)(wbase & 0xffffffffu));
        wdesc[1] = __builtin_amdgcn_readfirstlane((int)((wbase >> 32) & 0xffffu));
        wdesc[2] = __builtin_amdgcn_readfirstlane((int)((size_t)ntiles32 * nit * 4096));
        wdesc[3] = 0x00020000;
        f32x4 fb[NR][NT][4];
        auto issue = [&](auto set_c, int it) {        // loads of k-step `it`: filter fragments → register set, A rows → LDS stage it % NS
            constexpr int S = decltype(set_c)::value;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned off = w_off[j] == OOB ? OOB : w_off[j] + (unsigned)it * 4096u + (unsigned)kk * 1024u;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(fb[S][j][kk]) : "v"(off), "s"(wdesc) : "memory");
                }
            stage_a(it % NS);
        };
        // Shortcut prefetch (fp16 64 x 128 tiles): the residual rows a thread will finish in the epilogue
        // — 16 B of NRES rows — are requested right behind the prologue's k-steps, so that they travel under the first DEEP
        // k-steps instead of being issued and awaited inside the epilogue (these layers have 2 - 8 k-steps: the epilogue's load
        // round trip was a third of a block's life). In the in-order queue they are younger than k-steps 0 .. DEEP - 1 only:
        // the counted waits of those steps allow NRES more outstanding operations, later steps retire them on the way.
        // fp16 only: measured on the fp32 engine the same layers (MFMA-bound there, 168 registers with the 8 extra rows) lose 1-2 %.
        constexpr bool RESPRE_OK = RES;
        static_assert(!RES || (sizeof(T) == 2 && sizeof(TO) == 2 && MT == 2 && NT == 1 && WN == 4), "shortcut prefetch: fp16 64 x 128 tiles");
        constexpr int CPL = 16 / (int)sizeof(TO), CHUNKS = BN / CPL, RPP = THREADS / CHUNKS, NRES = RESPRE_OK ? BM / RPP : 1;
        const size_t res_bytes = (a.res_shift ? (size_t)a.B * (a.Ho >> 1) * (a.Wo >> 1) : (size_t)a.M) * a.Cout * sizeof(T);
        constexpr bool respre = RES;                  // conv_bd_launch checked: shortcut present, Cout % 8 == 0, shortcut below 4 GB
        [[maybe_unused]] f32x4 rb[NRES];
        [[maybe_unused]] const int ec = (tid % CHUNKS) * CPL, er = tid / CHUNKS;
        [[maybe_unused]] auto issue_res = [&]() {
            const unsigned long long rbase = (unsigned long long)a.res;
            i32x4 rdesc;
            rdesc[0] = __builtin_amdgcn_readfirstlane((int)(rbase & 0xffffffffu));
            rdesc[1] = __builtin_amdgcn_readfirstlane((int)((rbase >> 32) & 0xffffu));
            rdesc[2] = __builtin_amdgcn_readfirstlane((int)res_bytes);
            rdesc[3] = 0x00020000;
            const int hw = a.Ho * a.Wo;
#pragma unroll
            for (int k = 0; k < NRES; ++k) {
                const int m = m0 + er + RPP * k, n = n0 + ec;
                unsigned off = OOB;
                if (m < M && n < a.Cout) {
                    size_t e = (size_t)m * a.Cout + n;
                    if (a.res_shift) {
                        const int b = m / hw;
                        const int rem = m - b * hw;
                        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                        e = ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n;
                    }
                    off = (unsigned)(e * sizeof(T));
                }
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rb[k]) : "v"(off), "s"(rdesc) : "memory");
            }
        };
        auto wait_for = [&](int younger, bool res_out) {      // all but the `younger` newest k-steps' operations (and the shortcut rows behind them) have completed
            if (RESPRE_OK && res_out) {
                if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * OPS + NRES) : "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS + NRES) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS + NRES) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
                return;
            }
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * OPS) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        static_assert(DEEP <= 3 && 3 * OPS + NRES < 64, "vmcnt immediates");
        auto step = [&](auto set_c, int it) {
            constexpr int S = decltype(set_c)::value;
            if (it + DEEP < nit) issue(std::integral_constant<int, (S + DEEP) % NR>{}, it + DEEP);
            const int younger = nit - 1 - it < DEEP ? nit - 1 - it : DEEP;
            wait_for(younger, respre && it < DEEP);   // the shortcut rows sit behind the prologue's k-steps 0 .. DEEP - 1: younger than those only
            __builtin_amdgcn_s_barrier();             // every wave's A rows of step `it` have landed; stage (it - 1) % NS is free
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(fb[S][j][kk]));      // the fragments exist from here on
            compute(it % NS, fb[S]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // prologue: steps 0 .. DEEP - 1 in flight
        if (0 < nit) issue(std::integral_constant<int, 0>{}, 0);
        if constexpr (DEEP >= 2) { if (1 < nit) issue(std::integral_constant<int, 1 % NR>{}, 1); }
        if constexpr (DEEP >= 3) { if (2 < nit) issue(std::integral_constant<int, 2 % NR>{}, 2); }
        if constexpr (RESPRE_OK) { if (respre) issue_res(); }      // ONE issue site, right behind the prologue: in flight under the first DEEP k-steps
        for (int it = 0; it < nit; it += NR) {
            step(std::integral_constant<int, 0>{}, it);
            if constexpr (NR > 1) { if (it + 1 < nit) step(std::integral_constant<int, 1 % NR>{}, it + 1); }
            if constexpr (NR > 2) { if (it + 2 < nit) step(std::integral_constant<int, 2 % NR>{}, it + 2); }
            if constexpr (NR > 3) { if (it + 3 < nit) step(std::integral_constant<int, 3 % NR>{}, it + 3); }
        }
        __builtin_amdgcn_s_barrier();                 // all fragment reads are done: LDS is the epilogue's
        if constexpr (RESPRE_OK) {
            if (respre) {
                // conv_epilogue's general path with the loads already done: scale and bias in the accumulator layout (one channel
                // per lane), the fp32 tile through LDS once, then 16 B of NRES rows per thread: + shortcut, ReLU, one rounding —
                // the same single IEEE operations in the same order, bit-identical to conv_epilogue
                constexpr int CS = BN + 4;
                float* Cs = reinterpret_cast<float*>(lds);
                const int col = wave * 32 + (lane & 31);
                float sc = 1.f, bi = 0.f;
                if (n0 + col < a.Cout) {
                    if (a.scale) sc = a.scale[n0 + col];
                    if (a.bias) bi = a.bias[n0 + col];
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = acc[i][0][r];
                        if (a.scale) t = __fmul_rn(t, sc);
                        if (a.bias) t = __fadd_rn(t, bi);
                        Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CS + col] = t;
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NRES; ++k) asm volatile("" : "+v"(rb[k]));
                __syncthreads();
                TO* __restrict__ Y = static_cast<TO*>(a.y);
                const int n = n0 + ec;
#pragma unroll
                for (int k = 0; k < NRES; ++k) {
                    const int row = er + RPP * k, m = m0 + row;
                    if (m >= M || n >= a.Cout) continue;
                    float v[CPL];
#pragma unroll
                    for (int g = 0; g < CPL / 4; ++g) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(&Cs[row * CS + ec + 4 * g]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * g + e] = t[e];
                    }
                    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
                    const f16x8v rs = __builtin_bit_cast(f16x8v, rb[k]);
                    f16x8v h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = __fadd_rn(v[e], (float)rs[e]);
                        if (a.relu) t = t > 0.f ? t : 0.f;
                        h[e] = (_Float16)t;
                    }
                    *reinterpret_cast<f16x8v*>(Y + (size_t)m * a.Cout + n) = h;
                }
                return;
            }
        }
        conv_epilogue<T, TO, MT, NT, 1, WN, 1>(a, acc, lds, M, m0, n0, tid, lane, 0, wave);
        return;
    }
    // two register sets for the filter fragments (the loop is unrolled by two so that they never have to be copied): set P
    // holds super-step `s`, set Q is being filled for `s + 1` while P is consumed
    f32x4 fbP[KS][NT][4], fbQ[KS][NT][4];
    const int nsuper = (nit + KS - 1) / KS;
    auto fill = [&](int stage, f32x4 (&fb)[KS][NT][4], int s) {       // DMA + filter loads of super-step s
#pragma unroll
        for (int c = 0; c < KS; ++c)
            if (s * KS + c < nit) {
                stage_a(stage * KS + c);
                load_b(fb[c], s * KS + c);
            }
    };
    auto run = [&](int stage, const f32x4 (&fb)[KS][NT][4], int s) {
#pragma unroll
        for (int c = 0; c < KS; ++c)
            if (s * KS + c < nit) compute(stage * KS + c, fb[c]);
    };
    fill(0, fbP, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nsuper; s += 2) {
        if (s + 1 < nsuper) fill(1, fbQ, s + 1);
        run(0, fbP, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 1 >= nsuper) break;
        if (s + 2 < nsuper) fill(0, fbP, s + 2);
        run(1, fbQ, s + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    conv_epilogue<T, TO, MT, NT, 1, WN, 1>(a, acc, lds, M, m0, n0, tid, lane, 0, wave);
}

template <typename T, typename TO, int MT, int NT, int WN, int BPC, int KS, int DEEP, bool RES>
__global__ __launch_bounds__(64 * WN, (BPC * WN + 3) / 4)
void conv_bd_kernel(const ConvArgs a) {
    constexpr int BM = 32 * MT;
    constexpr int STAGE_BYTES = (DEEP > 0 ? DEEP + 2 : 2 * KS) * BM * CHUNK_BYTES;
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<TO, MT, NT, 1, WN, 1>();
    constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    static_assert(BPC * LDS_BYTES <= 160 * 1024, "LDS footprint does not allow that many blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    conv_bd_body<T, TO, MT, NT, WN, KS, DEEP, RES>(a, lds);
}

template <typename T, typename TO, int MT, int NT, int WN, int BPC, int KS, int DEEP = 0, bool RES = false>
td_status launch_bd(const ConvArgs& a, hipStream_t stream) {
    const int tiles = td_cdiv(a.M, 32 * MT) * td_cdiv(a.Cout, 32 * NT * WN);
    hipLaunchKernelGGL((conv_bd_kernel<T, TO, MT, NT, WN, BPC, KS, DEEP, RES>), dim3(tiles), dim3(64 * WN), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

}  // namespace

// Filter bank [Cout][KH][KW][Cin] (elements of `es` bytes: 2 = fp16 bit patterns, 4 = fp32) → fragment order
// [ceil(Cout/32)][nit][4][64 lanes][16 B], nit = KH*KW*(Cin/KE) k-steps of KE = 128 / es elements in the kernels' order (channel chunk
// outer, filter tap inner). Lane l = (r = l & 31, h = l >> 5) of (tile t, step it, kk) holds the 16-B piece 2 kk + h of row
// W[t*32 + r][tap][cc*KE .. +KE] — what the LDS fragment read hands that lane; columns past Cout are zero.
void conv_bd_pack(const void* w_ohwi, int es, int cout, int kh, int kw, int cin, std::vector<unsigned char>& out) {
    const int ke = 128 / es;
    const int ntaps = kh * kw, cchunks = cin / ke, nit = ntaps * cchunks, nt32 = (cout + 31) / 32;
    out.assign((size_t)nt32 * nit * 4096, (unsigned char)0);
    const unsigned char* w = static_cast<const unsigned char*>(w_ohwi);
    for (int t = 0; t < nt32; ++t)
        for (int cc = 0; cc < cchunks; ++cc)
            for (int tap = 0; tap < ntaps; ++tap) {
                const int it = cc * ntaps + tap;
                for (int kk = 0; kk < 4; ++kk)
                    for (int l = 0; l < 64; ++l) {
                        const int n = t * 32 + (l & 31), h = l >> 5;
                        if (n >= cout) continue;
                        const unsigned char* src = w + (((size_t)n * ntaps + tap) * cin + (size_t)cc * ke) * es + (size_t)(2 * kk + h) * 16;
                        unsigned char* dst = out.data() + ((((size_t)t * nit + it) * 4 + kk) * 64 + l) * 16;
                        for (int j = 0; j < 16; ++j) dst[j] = src[j];
                    }
            }
}

bool conv_bd_ok(const ConvArgs& a, int precision) {
    const int ke = precision == TD_PRECISION_FP16 ? 64 : 32;
    return (precision == TD_PRECISION_FP16 || precision == TD_PRECISION_FP32) && a.w_frag && a.out_mode == 0 && a.batch_count <= 1 &&
           a.Cin % ke == 0 && a.KH * a.KW <= 32 && !(precision == TD_PRECISION_FP32 && a.out_f32) &&
           (size_t)((a.Cout + 31) / 32) * (size_t)(a.KH * a.KW * (a.Cin / ke)) * 4096 < 0xfffffff0ull - (1u << 20);
}

namespace {
template <typename T, typename TO>
td_status bd_variant(const ConvArgs& a, int variant, hipStream_t stream) {
    switch (variant) {
        case 6: return launch_bd<T, TO, 4, 1, 4, 2, 1, 3>(a, stream);   // 128 x 128 (4 waves of 128 x 32), three k-steps in flight: half the filter re-reads of the 64-row tiles
        case 5: return launch_bd<T, TO, 4, 2, 4, 1, 1, 3>(a, stream);   // 128 x 256 (4 waves of 128 x 64, one per SIMD: ~300 registers), three k-steps in flight: long-K layers with few row tiles (fc1)
        case 3:                                                         // 64 x 128, three k-steps of loads in flight (five 8-KB LDS stages, four filter register sets)
            if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
                const size_t res_bytes = (a.res_shift ? (size_t)a.B * (a.Ho >> 1) * (a.Wo >> 1) : (size_t)a.M) * a.Cout * sizeof(T);
                if (a.res && (a.Cout & 7) == 0 && !a.out_f32 && res_bytes < 0xfffffff0ull - (1u << 20))
                    return launch_bd<T, TO, 2, 1, 4, 3, 1, 3, true>(a, stream);       // + the shortcut rows prefetched behind the last k-step
            }
            return launch_bd<T, TO, 2, 1, 4, 3, 1, 3>(a, stream);
        case 4: return launch_bd<T, TO, 2, 2, 4, 2, 1, 2>(a, stream);   // 64 x 256, two k-steps in flight
        case 2: return launch_bd<T, TO, 2, 1, 4, 3, 2>(a, stream);      // 64 x 128, two k-chunks per barrier interval
        case 1: return launch_bd<T, TO, 2, 1, 4, 4, 1>(a, stream);      // 64 x 128 (4 waves of 64 x 32: narrow layers, more blocks)
        default: return launch_bd<T, TO, 2, 2, 4, 2, 1>(a, stream);     // 64 x 256 (4 waves of 64 x 64)
    }
}
}  // namespace

td_status conv_bd_launch(const ConvArgs& a, int precision, int variant, hipStream_t stream) {
    TD_REQUIRE(conv_bd_ok(a, precision), "filter-direct convolution: unsupported launch (packed filters, plain output only)");
    if (precision == TD_PRECISION_FP32) return bd_variant<float, float>(a, variant, stream);
    if (a.out_f32) return bd_variant<_Float16, float>(a, variant, stream);
    return bd_variant<_Float16, _Float16>(a, variant, stream);
}
