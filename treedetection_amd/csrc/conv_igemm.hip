// Implicit-GEMM NHWC convolution on the gfx950 matrix cores.
//
//   y[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + bias[n] (+ residual[m][n]) )
//   m = (b, oy, ox)   n = output channel   k = (c-chunk, ky, kx, c-in-chunk)
//   One k-chunk is a contiguous 128-B run of one input pixel (32 floats / 64 halves): coalesced HBM reads, no
//   im2col buffer; the chunk loop is OUTER and the filter-tap loop INNER, so the 9 taps of a 3x3 re-read the same
//   few rows of one channel slice back to back (halo re-reads hit L1/L2; measured HBM traffic 1.27x algorithmic).
//
// This one kernel family carries every contraction of the reference's forward
// (TreeDetection/prediction.py:183 → detectron2 GeneralizedRCNN): bottleneck 1x1 / 3x3
// (stride lives in the 1x1, STRIDE_IN_1X1), FPN lateral (+ nearest-2x upsampled top-down add in the epilogue) and
// output convs, RPN conv + heads, the box-head FCs (1x1 "conv" over R rows), the mask-head 3x3s and the 2x2/s2
// deconv (out_mode 1: four 1x1 GEMMs with a pixel-shuffle store).
//
// Two element types share the structure (everything is addressed in bytes; a k-chunk is 128 B either way):
//   float    — v_mfma_f32_32x32x2_f32: exact f32 fmaf chains; one 16-B fragment read feeds 4 MFMAs (k-pairs {e,e+4})
//   _Float16 — v_mfma_f32_32x32x16_f16: f32 accumulate; one 16-B fragment read (8 halves) feeds 1 MFMA
// Block tile 128x128 (also 128x64, 64x128, 64x64; the engine measures which is fastest per layer), 4 waves as 2x2,
// LDS rows padded to 144 B so every ds_read_b128 lane group hits 16 distinct 16-B slots, global→register→LDS double
// buffering with one barrier per k-step, bijective XCD-aware block remap over the live tile count, and an epilogue
// staged through LDS so residual loads / stores are whole coalesced row segments.
#include "common.h"
#include <cstdlib>
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f32(_Float16 v) { return (float)v; }
};

// T = element type of x / w / residual; TO = element type of y (float outputs are kept for the RPN / box heads)
// WM x WN waves per block (each wave owns a (32*MT) x (32*NT) output sub-tile): 2x2 = the 4-wave tiles above;
// 4x2 / 2x4 / 4x4 = 256x128 / 128x256 / 256x256 block tiles whose larger operand reuse cuts the DMA instructions and
// the L2 traffic per flop (what bounds the fp16 path).
template <typename T, typename TO, int MT, int NT, int NSTAGE, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int LDROWS = THREADS / 8;               // rows staged per pass (8 lanes x 16 B per row)
    constexpr int AROWS = BM / LDROWS, BROWS = BN / LDROWS;   // rows staged per thread
    static_assert(BM % LDROWS == 0 && BN % LDROWS == 0 && LDROWS % 16 == 0, "tile / thread-count mismatch");
    constexpr int ES = sizeof(T);
    constexpr int KE = Elem<T>::PER_CHUNK;            // elements per k-chunk
    constexpr int CS = BN + 4;                        // padded row stride (floats) of the epilogue's staged tile
    constexpr int WROW = 32 * MT;                     // rows of one wave-row
    constexpr int STAGE_BYTES = NSTAGE * (BM + BN) * CHUNK_BYTES;
    // the epilogue stages RWM wave-rows at a time through LDS (all of them when the k-loop's LDS is large enough)
    constexpr int FIT = STAGE_BYTES / (WROW * CS * 4);
    constexpr int RWM = FIT >= WM ? WM : (FIT >= 2 && WM % 2 == 0 ? 2 : 1);
    constexpr int EPI_BYTES = RWM * WROW * CS * 4;
    __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    char* As = lds;                                   // [NSTAGE][BM][128 B]   piece c of row r sits at slot c ^ ((r>>1)&7)
    char* Bs = lds + NSTAGE * BM * CHUNK_BYTES;       // [NSTAGE][BN][128 B]

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
    const int wm = wave / WN, wn = wave % WN;
    const int ld_c = tid & 7;    // 16-B piece inside the 128-B k-chunk
    const int ld_r = tid >> 3;   // 0..31

    // ---- global → LDS staging by LDS-DMA (buffer_load ... lds) ---------------------------------------------------
    // One wave instruction moves 64 x 16 B = 8 rows x 128 B straight into LDS (no VGPR round trip, no ds_write:
    // the write path of ds_write_b128, ~80 B/clk/CU, was the fp16 bottleneck). The LDS image is lane-linear, so the
    // bank-conflict fix is an XOR swizzle applied on the SOURCE piece (lane (r, c) fetches piece c ^ ((r>>1)&7)) and
    // again on the fragment reads. Every lane keeps ONE 32-bit byte offset per staged row (tap (0,0)); the per-step
    // part of the address (filter tap, channel chunk) is wave-uniform. Rows / taps in the zero padding (or past M /
    // Cout) get an offset beyond num_records: the range check makes the DMA write zeros (tools/lds_dma_probe.hip).
    const int K = a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / KE;
    const int ntaps = a.KH * a.KW;
    const int nit = ntaps * cchunks;
    const unsigned pix_bytes = (unsigned)a.Cin * ES;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.H * a.W * pix_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w), 0, (int)((size_t)a.Cout * K * ES), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;

    unsigned a_off[AROWS], a_ok[AROWS];      // byte offset of tap (0,0); bit t set = tap t reads a real pixel
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + ld_r + LDROWS * i;
        a_off[i] = 0;
        a_ok[i] = 0;
        if (m < M) {
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
    unsigned b_off[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + ld_r + LDROWS * i;
        b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)K * ES + src_piece : OOB;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;   // this wave stages rows 8w..8w+7 of each 32
    int ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_cc = 0;     // scalar walk over (chunk outer, tap inner)
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
        for (int i = 0; i < BROWS; ++i) {
            const unsigned off = b_off[i] == OOB ? OOB : b_off[i] + ws;
            char* dst = Bs + ((unsigned)buf * BN + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
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

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software pipeline: NSTAGE LDS buffers, NSTAGE-1 k-steps of DMA in flight --------------------------------
    // Each wave issues LOADS = AROWS + BROWS DMA instructions per k-step. Before the barrier that closes k-step `it`,
    // `s_waitcnt vmcnt((NSTAGE-2)*LOADS)` retires this wave's pieces of step it+1 and leaves the younger steps in
    // flight; the barrier then (a) publishes step it+1's buffer to every wave and (b) frees buffer it % NSTAGE, which
    // the DMA of step it+NSTAGE overwrites right after it. A RAW s_barrier is used on purpose: __syncthreads() would
    // drain vmcnt to 0 and serialise the pipeline (cdna_hip_programming.md §5 "Pipelining across barriers").
    constexpr int LOADS = AROWS + BROWS;
    constexpr int INFLIGHT = (NSTAGE - 2) * LOADS;
    auto wait_stage = [&]() {
        if constexpr (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (INFLIGHT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (INFLIGHT == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (INFLIGHT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else static_assert(INFLIGHT < 0, "add the vmcnt immediate for this configuration");
    };
    // fragment reads: row = lane & 31, piece = 2*kk + (lane >> 5), swizzled with the row's key (lane >> 1) & 7
    const unsigned frag_row = lane & 31;
    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = frag_row * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;
    // one k-step: MT + NT fragment reads per 16-B piece pair, MT x NT MFMA groups
    auto compute = [&](int buf) {
        const char* Ab = &As[(buf * BM + wm * 32 * MT) * CHUNK_BYTES];
        const char* Bb = &Bs[(buf * BN + wn * 32 * NT) * CHUNK_BYTES];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) Elem<T>::mma(fa[i], fb[j], acc[i][j]);
        }
    };
    if constexpr (NSTAGE == 1) {
        // One LDS buffer: load, wait, compute, release — nothing overlaps inside the block, but the LDS footprint is
        // half, so a second block shares the CU and ITS loads / stores overlap this one's MFMAs. For the thin 1x1
        // layers (Cin 64..128: one to four k-steps), where a prefetch pipeline has nothing to run ahead of.
        for (int it = 0; it < nit; ++it) {
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
#pragma unroll
        for (int p = 0; p < NSTAGE - 1; ++p)
            if (p < nit) stage(p);
        // prologue: step 0 must have landed; with fewer than NSTAGE-1 steps issued simply drain
        if (nit >= NSTAGE - 1) wait_stage();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0, nxt = NSTAGE - 1;       // buffer of step it / buffer the DMA of step it+NSTAGE-1 goes to
        for (int it = 0; it < nit; ++it) {
#if !defined(TD_DIAG_NO_LOADS)     // diagnostic builds only (tools/conv_diag.py): never defined in the product
            if (it + NSTAGE - 1 < nit) stage(nxt);
#endif
            compute(cur);
#if !defined(TD_DIAG_NO_BARRIER)
            // retire step it+1 (younger DMA stays in flight while steps remain to be issued; the tail drains fully)
            if (it + NSTAGE - 1 < nit) wait_stage();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of buf[cur] are done
            __builtin_amdgcn_s_barrier();
#endif
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        }
    }

    // ---- epilogue: accumulators → LDS tile → 4 channels per lane: scale/bias (+residual) (+ReLU), one IEEE op per
    // step (no fma contraction). Staging through LDS turns the MFMA layout (32 lanes x 4 B per row) into whole row
    // segments, so residual loads and stores are vector accesses and fully coalesced (HBM-bound 1x1 layers).
    float* Cs = reinterpret_cast<float*>(lds);
    TO* __restrict__ Y = static_cast<TO*>(a.y);
    const T* __restrict__ Rs = static_cast<const T*>(a.res);
    const int Cq = a.out_mode == 1 ? a.Cout >> 2 : a.Cout;
    constexpr int CHUNKS = BN / 4;                     // 4-channel pieces per tile row
    constexpr int ROWS_PER_PASS = THREADS / CHUNKS;
    const int c4 = tid % CHUNKS;
    const int n = n0 + c4 * 4;
    const bool vec = (a.Cout & 3) == 0 && n + 3 < a.Cout;   // aligned, whole piece in range
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
        static_assert((RWM * WROW) % ROWS_PER_PASS == 0, "epilogue rows must split evenly over the threads");
        constexpr int U = ITERS < 8 ? ITERS : 8;
        static_assert(ITERS % U == 0, "epilogue batch must divide the rows per thread");
        typedef typename std::conditional<sizeof(T) == 4, f32x4, f16x4>::type ResVec;
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
                f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CS + c4 * 4]);
                size_t yoff;
                if (a.out_mode == 0) {
                    yoff = (size_t)m * a.Cout + n;
                } else {
                    const int qq = n / Cq, co = n - qq * Cq;   // qq = dy*2+dx; a piece never straddles it (Cq % 4 == 0)
                    const int b = m / hw;
                    const int rem = m - b * hw;
                    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                    yoff = ((size_t)(b * 2 * a.Ho + 2 * oy + (qq >> 1)) * (2 * a.Wo) + 2 * ox + (qq & 1)) * Cq + co;
                }
                float rs[4] = {0.f, 0.f, 0.f, 0.f};
                if (Rs) {
                    if (vec) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) rs[e] = (float)rbuf[u][e];
                    } else {
                        const size_t roff = res_offset(m);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < a.Cout) rs[e] = Elem<T>::to_f32(Rs[roff + e]);
                    }
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
                if (vec) {
                    if constexpr (sizeof(TO) == 4) {
                        *reinterpret_cast<f32x4*>(Y + yoff) = v;
                    } else {
                        f16x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
                        *reinterpret_cast<f16x4*>(Y + yoff) = h;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.Cout) Y[yoff + e] = (TO)v[e];
                }
            }
        }
    }
}

template <typename T, typename TO, int MT, int NT, int NSTAGE, int WM = 2, int WN = 2>
td_status launch(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const int tiles = td_cdiv(a.M, BM) * td_cdiv(a.Cout, BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, TO, MT, NT, NSTAGE, WM, WN>), dim3(tiles), dim3(64 * WM * WN), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

template <typename T, typename TO>
td_status dispatch(const ConvArgs& a, int cfg, hipStream_t stream) {
    switch (cfg) {      // 0-3: two LDS stages (one k-step of DMA in flight); 4-7: three stages (two in flight)
        case 0: return launch<T, TO, 2, 2, 2>(a, stream);
        case 1: return launch<T, TO, 2, 1, 2>(a, stream);
        case 2: return launch<T, TO, 1, 2, 2>(a, stream);
        case 3: return launch<T, TO, 1, 1, 2>(a, stream);
        case 4: return launch<T, TO, 2, 2, 3>(a, stream);
        case 5: return launch<T, TO, 2, 1, 3>(a, stream);
        case 6: return launch<T, TO, 1, 2, 3>(a, stream);
        case 7: return launch<T, TO, 1, 1, 3>(a, stream);
        case 8: return launch<T, TO, 2, 2, 2, 4, 2>(a, stream);     // 256 x 128, 8 waves
        case 9: return launch<T, TO, 2, 2, 2, 2, 4>(a, stream);     // 128 x 256, 8 waves
        case 10: return launch<T, TO, 2, 2, 2, 4, 4>(a, stream);    // 256 x 256, 16 waves
        // 256 x 256 with bigger per-wave register tiles: fewer LDS fragment bytes per MFMA (the fp16 limiter)
        case 11: return launch<T, TO, 4, 4, 2, 2, 2>(a, stream);    // 4 waves of 128 x 128 (256 accumulator registers)
        case 12: return launch<T, TO, 4, 2, 2, 2, 4>(a, stream);    // 8 waves of 128 x 64
        case 13: return launch<T, TO, 2, 4, 2, 4, 2>(a, stream);    // 8 waves of 64 x 128
        // single LDS stage (half the LDS: two blocks per CU overlap each other) for layers with 1-2 k-steps
        case 14: return launch<T, TO, 2, 2, 1, 4, 4>(a, stream);    // 256 x 256
        case 15: return launch<T, TO, 2, 2, 1, 2, 2>(a, stream);    // 128 x 128
        case 16: return launch<T, TO, 2, 2, 1, 2, 4>(a, stream);    // 128 x 256
        default: td_set_error("conv2d: bad tile_cfg %d", cfg); return TD_ERR_INVALID;
    }
}

}  // namespace

td_status conv2d_launch(const ConvArgs& a, int precision, hipStream_t stream) {
    const int ke = precision == TD_PRECISION_FP16 ? 64 : 32;
    TD_REQUIRE(precision == TD_PRECISION_FP32 || precision == TD_PRECISION_FP16, "conv2d: bad precision %d", precision);
    TD_REQUIRE(a.Cin % ke == 0, "conv2d: Cin=%d must be a multiple of %d", a.Cin, ke);
    TD_REQUIRE(a.M > 0 && a.Cout > 0, "conv2d: empty problem (M=%d, Cout=%d)", a.M, a.Cout);
    const size_t es = precision == TD_PRECISION_FP16 ? 2 : 4;
    TD_REQUIRE((size_t)a.B * a.H * a.W * (size_t)a.Cin * es < 0xfffffff0ull - (1u << 20), "conv2d: input tensor must stay below 4 GB (32-bit buffer offsets)");
    TD_REQUIRE((size_t)a.Cout * a.KH * a.KW * a.Cin * es < 0xfffffff0ull - (1u << 20), "conv2d: weight tensor must stay below 4 GB");
    TD_REQUIRE(a.KH * a.KW <= 32, "conv2d: at most 32 filter taps (got %dx%d)", a.KH, a.KW);
    TD_REQUIRE(a.out_mode == 0 || (a.Cout % 16 == 0 && !a.res), "conv2d: bad deconv configuration");
    int cfg = a.tile_cfg;
    if (cfg < 0) {
        static const char* forced = getenv("TD_CONV_CFG");     // diagnostics only (tools/conv_diag.py)
        if (forced) cfg = atoi(forced);
    }
    if (cfg < 0) {
        // heuristic (the engine replaces it by a measured choice per layer shape): wide N for wide layers, 64-row
        // tiles when there is less than one 128-row tile per CU
        const bool small_m = a.M <= 64 * 256;
        cfg = a.Cout <= 64 ? (small_m ? 3 : 1) : (small_m ? 2 : 0);
    }
    if (precision == TD_PRECISION_FP32) return dispatch<float, float>(a, cfg, stream);
    if (a.out_f32) return dispatch<_Float16, float>(a, cfg, stream);
    return dispatch<_Float16, _Float16>(a, cfg, stream);
}
