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
// Block tiles 64x64 .. 256x256 with 4 / 8 / 16 waves (dispatch() lists them; the engine measures which is fastest per
// layer shape), global → LDS by LDS-DMA (buffer_load ... lds, no VGPR round trip) into a lane-linear image whose bank
// conflicts are removed by an XOR swizzle on the SOURCE piece and again on the ds_read_b128 fragment reads, two LDS
// stages with one raw s_barrier per k-step (single-stage variants for the thin 1x1 layers), bijective XCD-aware block
// remap over the live tile count, an epilogue staged through LDS (conv_epilogue), batched launches (blockIdx.y) for the
// Winograd plane contractions, and conv_pp8_kernel: the fp16 256x256 ping-pong schedule.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#include "conv_tiles.h"

namespace {

// T = element type of x / w / residual; TO = element type of y (float outputs are kept for the RPN / box heads)
// WM x WN waves per block (each wave owns a (32*MT) x (32*NT) output sub-tile): 2x2 = the 4-wave tiles above;
// 4x2 / 2x4 / 4x4 = 256x128 / 128x256 / 256x256 block tiles whose larger operand reuse cuts the DMA instructions and
// the L2 traffic per flop (what bounds the fp16 path).
template <typename T, typename TO, int MT, int NT, int NSTAGE, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, NSTAGE == 1 && MT * NT <= 4 ? 4 : 1)     // single-stage tiles: >= 4 waves per SIMD (<= 128 registers)
void conv_igemm_kernel(const ConvArgs a_in) {
    ConvArgs a = a_in;
    if (a.batch_count > 1) {                         // batched contraction: plane blockIdx.y of x / w / y
        const long long bz = blockIdx.y;
        a.x = static_cast<const T*>(a.x) + bz * a.x_bs;
        a.w = static_cast<const T*>(a.w) + bz * a.w_bs;
        a.y = static_cast<TO*>(a.y) + bz * a.y_bs;
    }
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
    constexpr bool WIDE_OK = NSTAGE > 1;      // the single-stage tiles trade everything for blocks per CU
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<TO, MT, NT, WM, WN, RWM, WIDE_OK>();
    __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    char* As = lds;                                   // [NSTAGE][BM][128 B]   piece c of row r sits at slot c ^ ((r>>1)&7)
    char* Bs = lds + NSTAGE * BM * CHUNK_BYTES;       // [NSTAGE][BN][128 B]

    int M = a.M;
    if (a.m_dyn) {
        int md = *a.m_dyn * a.m_mul - a.m_off;
        md = md < 0 ? 0 : md;
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
    // plain contractions (1x1, stride 1, no padding: most launches, and every Winograd plane) address row m directly — the
    // (image, y, x) decomposition costs two integer divisions per staged row, a visible part of a block that runs only a
    // handful of k-steps
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

    conv_epilogue<T, TO, MT, NT, WM, WN, RWM, WIDE_OK>(a, acc, lds, M, m0, n0, tid, lane, wm, wn);
}

// ---- plane_gemm_kernel: PERSISTENT tile walk for the fp32 contractions with plain rows (Winograd planes, 1x1 / stride-1 layers) ----
// A plane contraction M_xi[t][n] = sum_c V_xi[t][c] U_xi[n][c] has K = Cin = 128 .. 512: four to sixteen k-steps per
// block tile. In conv_igemm_kernel such a block lives ≈ 28 us of which ≈ 7 are not MFMA work — block launch, address
// set-up, the first loads' HBM latency (V is streamed once, never cached), the LDS-staged epilogue — and with three
// blocks per CU the matrix pipe ends at 0.73 (FPN output 2: 94 executed GFLOP in 824 us). Here a block stays resident
// and walks tiles t = i G + b (G = grid, XCD-remapped so that one XCD's blocks take neighbouring tiles); its k-steps
// form ONE continuous stream across tile boundaries — the DMA of the next tile's first k-step is issued under the
// last MFMAs of the current tile, exactly like any other k-step — and a finished tile leaves straight from the
// accumulator registers (a register of a 32 x 32 MFMA tile is 32 consecutive floats of one row per half-wave: whole
// 128-B lines; planes carry no scale / bias / residual), so the staging LDS is never borrowed by an epilogue. Same k
// order → bit-identical to every other block tile (tests/test_conv_gpu.py).
template <int MT, int NT>
__global__ __launch_bounds__(256) void plane_gemm_kernel(const ConvArgs a) {
    typedef float T;
    constexpr int WN = 2, THREADS = 256;
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int LDROWS = THREADS / 8;               // 32 rows per DMA pass
    constexpr int AROWS = BM / LDROWS, BROWS = BN / LDROWS;
    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * CHUNK_BYTES];
    char* As = lds;                                   // [2][BM][128 B], XOR-swizzled pieces as in conv_igemm_kernel
    char* Bs = lds + 2 * BM * CHUNK_BYTES;            // [2][BN][128 B]

    int M = a.M;
    if (a.m_dyn) {
        int md = *a.m_dyn * a.m_mul - a.m_off;
        md = md < 0 ? 0 : md;
        M = md < M ? md : M;
    }
    const int planes = a.batch_count > 1 ? a.batch_count : 1;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int per_plane = tiles_m * tiles_n;
    const int total = per_plane * planes;
    const int G = gridDim.x;
    const int me = xcd_remap(blockIdx.x, G);
    if (me >= total) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const int cchunks = a.Cin / 32;
    const unsigned pix_bytes = (unsigned)a.Cin * 4;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;
    const unsigned x_records = (unsigned)((size_t)a.M * pix_bytes), w_records = (unsigned)((size_t)a.Cout * pix_bytes);
    constexpr unsigned OOB = 0xfffffff0u;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;

    // ---- loader cursor: runs one k-step ahead of the MFMAs, across tile boundaries ----
    int l_t = me, l_cc = 0;
    bool l_more = true;
    unsigned a_off[AROWS], b_off[BROWS];
    __amdgpu_buffer_rsrc_t xrsrc, wrsrc;
    auto load_setup = [&](int t) {
        const int plane = t / per_plane, r = t - plane * per_plane;
        const int tm = r / tiles_n, tn = r - tm * tiles_n;
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(static_cast<const T*>(a.x) + (long long)plane * a.x_bs), 0, (int)x_records, 0x00020000);
        wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(static_cast<const T*>(a.w) + (long long)plane * a.w_bs), 0, (int)w_records, 0x00020000);
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int m = tm * BM + ld_r + LDROWS * i;
            a_off[i] = m < M ? (unsigned)m * pix_bytes + src_piece : OOB;
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const int n = tn * BN + ld_r + LDROWS * i;
            b_off[i] = n < a.Cout ? (unsigned)n * pix_bytes + src_piece : OOB;
        }
    };
    auto stage = [&](int buf) {
        const unsigned cs = (unsigned)l_cc * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const unsigned off = a_off[i] == OOB ? OOB : a_off[i] + cs;
            char* dst = As + ((unsigned)buf * BM + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const unsigned off = b_off[i] == OOB ? OOB : b_off[i] + cs;
            char* dst = Bs + ((unsigned)buf * BN + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
        if (++l_cc == cchunks) {                      // the next k-step belongs to the next tile of this block
            l_cc = 0;
            l_t += G;
            if (l_t < total) load_setup(l_t);
            else l_more = false;
        }
    };

    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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

    // ---- compute cursor ----
    int c_t = me, c_cc = 0, cur = 0;
    load_setup(l_t);
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    while (true) {
        if (l_more) stage(cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next k-step has landed (and the previous tile's stores are out)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
        if (++c_cc == cchunks) {
            // tile c_t is complete: store it from the accumulator layout and start the next one
            const int plane = c_t / per_plane, r = c_t - plane * per_plane;
            const int tm = r / tiles_n, tn = r - tm * tiles_n;
            float* __restrict__ Y = static_cast<float*>(a.y) + (long long)plane * a.y_bs;
            const float* __restrict__ Rs = static_cast<const float*>(a.res);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = tn * BN + wn * 32 * NT + j * 32 + (lane & 31);
                const bool n_ok = n < a.Cout;
                // y = act(acc * scale + bias (+ residual)): the ops and their order of conv_epilogue, applied in the accumulator
                // layout (a lane owns one output channel: its scale / bias are two scalars per 32-column tile)
                const float sc = a.scale && n_ok ? a.scale[n] : 1.f;
                const float bi = a.bias && n_ok ? a.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float rs[16];
                    if (Rs) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int m = tm * BM + wm * 32 * MT + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                            rs[q] = m < M && n_ok ? Rs[(size_t)m * a.Cout + n] : 0.f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int m = tm * BM + wm * 32 * MT + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                        float t = acc[i][j][q];
                        if (a.scale) t = __fmul_rn(t, sc);
                        if (a.bias) t = __fadd_rn(t, bi);
                        if (Rs) t = __fadd_rn(t, rs[q]);
                        if (a.relu) t = t > 0.f ? t : 0.f;
                        if (m < M && n_ok) Y[(size_t)m * a.Cout + n] = t;
                        acc[i][j][q] = 0.f;
                    }
                }
            }
            c_cc = 0;
            c_t += G;
            if (c_t >= total) break;
        }
    }
}

// ---- wino_gemm_kernel: the 16 plane contractions of Winograd F(2x2,3x3) with the INPUT TRANSFORM FUSED into the A
// staging (fp32 engine; winograd.hip holds the algebra and the output transform) ------------------------------------------
//   M_xi[t][n] = sum_c V_xi[t][c] * U_xi[n][c],   V_xi[t][c] = (B^T d B)[xi] of tile t's 4x4 input patch
// Every V element is a signed sum of FOUR input pixels (B^T has two non-zeros per row): V = (p[a1][b1] (+-) p[a2][b1])
// (+-) (p[a1][b2] (+-) p[a2][b2]), in exactly the association wino_input_kernel uses — so instead of writing V (4x the
// layer input) to HBM and reading it back, the A tile of a k-step is built on the way into LDS: four 16-B loads per row
// piece (zero padding = out-of-range buffer offsets), three packed add/sub, one ds_write_b128 into the same XOR-swizzled
// image the LDS-DMA path produces. The fp32 MFMA loop has the slack for it (a k-step is 2048 MFMA cycles per wave; the
// staging is 8 loads + 24 VALU + 2 ds_writes per thread). Register double buffering: the loads of step it+1 are issued
// before the MFMAs of step it and combined after them. The weights U_xi still arrive by LDS-DMA. Block ids run plane-
// fastest, so the 16 planes of one tile range execute together and share the patch rows in L2 (the XCD remap keeps
// them on one XCD). Tile 128 (tiles) x 128 (channels), 8 waves as 4 x 2, two LDS stages, 2 blocks per CU.
// Results are bit-identical to wino_input_kernel + the batched conv_igemm launch (tests/test_conv_gpu.py).
__global__ __launch_bounds__(512, 2) void wino_gemm_kernel(const ConvArgs a) {
    typedef float T;
    constexpr int MT = 1, NT = 2, WM = 4, WN = 2;
    constexpr int BM = 128, BN = 128, THREADS = 512;
    constexpr int LDROWS = THREADS / 8;               // 64 rows per staging pass
    constexpr int AROWS = BM / LDROWS, BROWS = BN / LDROWS;
    constexpr int STAGE_BYTES = 2 * (BM + BN) * CHUNK_BYTES;
    constexpr int RWM = 2;
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<float, MT, NT, WM, WN, RWM>();
    __shared__ __attribute__((aligned(16))) char lds[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];
    char* As = lds;
    char* Bs = lds + 2 * BM * CHUNK_BYTES;

    int M = a.M;                                      // tiles of this slab (a.m_off = first tile of the slab)
    if (a.m_dyn) {
        int md = *a.m_dyn * a.m_mul - a.m_off;
        md = md < 0 ? 0 : md;
        M = md < M ? md : M;
    }
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int nblk = tiles_m * tiles_n * 16;
    if ((int)blockIdx.x >= nblk) return;
    const int pid = xcd_remap(blockIdx.x, nblk);
    const int plane = pid & 15;
    const int rest = pid >> 4;
    const int tm = rest / tiles_n, tn = rest - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    // B^T rows: i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3   (first operand, second operand, subtract?)
    const int pi = plane >> 2, pj = plane & 3;
    const int ra1 = pi == 0 ? 0 : (pi == 2 ? 2 : 1), ra2 = pi == 0 ? 2 : (pi == 1 ? 2 : (pi == 2 ? 1 : 3));
    const int cb1 = pj == 0 ? 0 : (pj == 2 ? 2 : 1), cb2 = pj == 0 ? 2 : (pj == 1 ? 2 : (pj == 2 ? 1 : 3));
    const bool row_sub = pi != 1, col_sub = pj != 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const int cchunks = a.Cin / 32;
    const unsigned pix_bytes = (unsigned)a.Cin * 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.H * a.W * pix_bytes), 0x00020000);
    const float* wplane = static_cast<const float*>(a.w) + (size_t)plane * a.w_bs;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(wplane), 0, (int)((size_t)a.Cout * a.Cin * 4), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;
    const int TH = (a.H + 1) >> 1, TW = (a.W + 1) >> 1;

    unsigned p_off[AROWS][4];                 // the four pixels of this plane, per staged row (OOB = zero padding / past M)
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + ld_r + LDROWS * i;
#pragma unroll
        for (int q = 0; q < 4; ++q) p_off[i][q] = OOB;
        if (m < M) {
            const long long t = (long long)a.m_off + m;
            const int tx = (int)(t % TW);
            const int ty = (int)((t / TW) % TH);
            const int b = (int)(t / ((long long)TW * TH));
            const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
            const int ys[2] = {y0 + ra1, y0 + ra2}, xs[2] = {x0 + cb1, x0 + cb2};
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // q = 2 * column + row:  (a1,b1) (a2,b1) (a1,b2) (a2,b2)
                const int yy = ys[q & 1], xx = xs[q >> 1];
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                    p_off[i][q] = (unsigned)((b * a.H + yy) * a.W + xx) * pix_bytes + src_piece;
            }
        }
    }
    unsigned b_off[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + ld_r + LDROWS * i;
        b_off[i] = n < a.Cout ? (unsigned)n * (unsigned)a.Cin * 4 + src_piece : OOB;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;

    f32x4 px[AROWS][4];                       // the k-step in flight: 4 pixel pieces per staged row
    auto load_a = [&](int cc) {
        const unsigned xs = (unsigned)cc * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < AROWS; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned off = p_off[i][q] == OOB ? OOB : p_off[i][q] + xs;
                px[i][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
            }
    };
    auto stage_b = [&](int buf, int cc) {
        const unsigned ws = (unsigned)cc * CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const unsigned off = b_off[i] == OOB ? OOB : b_off[i] + ws;
            char* dst = Bs + ((unsigned)buf * BN + (unsigned)LDROWS * i + wave_rows) * CHUNK_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
    };
    auto combine_store = [&](int buf) {       // V piece = (p0 rs p1) cs (p2 rs p3), IEEE add / sub in wino_input_kernel's order
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c0 = row_sub ? __fsub_rn(px[i][0][e], px[i][1][e]) : __fadd_rn(px[i][0][e], px[i][1][e]);
                const float c1 = row_sub ? __fsub_rn(px[i][2][e], px[i][3][e]) : __fadd_rn(px[i][2][e], px[i][3][e]);
                v[e] = col_sub ? __fsub_rn(c0, c1) : __fadd_rn(c0, c1);
            }
            *reinterpret_cast<f32x4*>(As + ((unsigned)buf * BM + (unsigned)LDROWS * i + (unsigned)ld_r) * CHUNK_BYTES + ld_c * 16) = v;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    const unsigned frag_row = lane & 31;
    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = frag_row * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;
    auto compute = [&](int buf) {
        const char* Ab = &As[(buf * BM + wm * 32 * MT) * CHUNK_BYTES];
        const char* Bb = &Bs[(buf * BN + wn * 32 * NT) * CHUNK_BYTES];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(Ab + frag_off[kk]);
            f32x4 fb[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * CHUNK_BYTES + frag_off[kk]);
#pragma unroll
            for (int j = 0; j < NT; ++j) Elem<T>::mma(fa, fb[j], acc[0][j]);
        }
    };

    load_a(0);
    stage_b(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    combine_store(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < cchunks; ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        const bool more = it + 1 < cchunks;
        if (more) {
            load_a(it + 1);
            stage_b(nxt, it + 1);
        }
        compute(cur);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            combine_store(nxt);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#if !defined(TD_WINO_EPI_DIRECT)   // staged epilogue; the direct register stores below measured no faster (475-477 vs 478-485 tiles/s)
    ConvArgs e = a;                           // plain store of the plane's accumulators: M_xi [tiles][Cout]
    e.y = static_cast<float*>(a.y) + (size_t)plane * a.y_bs;
    e.scale = nullptr; e.bias = nullptr; e.res = nullptr; e.relu = 0; e.out_mode = 0; e.res_shift = 0;
    e.Ho = 1; e.Wo = M > 0 ? M : 1;
    conv_epilogue<float, float, MT, NT, WM, WN, RWM>(e, acc, lds, M, m0, n0, tid, lane, wm, wn);
#else
    // M_xi [tiles][Cout] gets the raw accumulators: a register of a 32 x 32 tile is 32 consecutive floats of one row per
    // half-wave = one whole 128-B line (rows are Cout * 4 B apart, columns start at multiples of 32), so the plane is
    // stored straight from registers — no LDS round trip, no barriers
    float* __restrict__ Y = static_cast<float*>(a.y) + (size_t)plane * a.y_bs;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * 32 * NT + j * 32 + (lane & 31);
        if (n >= a.Cout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) Y[(size_t)m * a.Cout + n] = acc[0][j][r];
        }
    }
#endif
}

// ---- conv_pp8_kernel: 256 x 256 block tile, 8 waves, ping-pong schedule (fp16 only) ---------------------------------
// The k-loop of conv_igemm_kernel ends every 128-B k-chunk with a block-wide barrier that all 16 waves reach with
// empty pipelines: at the fp16 MFMA rate the refill (DMA issue, first fragment reads) is a quarter of the step. This
// kernel keeps the same data path (LDS-DMA staging of 128-B rows, XOR-swizzled image, v_mfma_f32_32x32x16_f16, same
// k order → bit-identical sums) and changes the schedule (cdna_hip_programming.md §5 "256² 8-phase template"):
//   * 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 outputs = 4 x 2 MFMA tiles (128 accumulator registers), so an A
//     fragment feeds 2 MFMAs and a B fragment 4 (24 ds_read_b128 per 32 MFMAs; the 16-wave tile needs 32).
//   * a k-chunk is computed in 4 PHASES, one output quadrant (64 x 32 = 2 MFMA tiles x 4 k-sub-steps = 8 MFMAs) each:
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0). A phase = load segment (issue this phase's fragment reads and 2 DMA
//     instructions, counted vmcnt) | s_barrier | compute segment (lgkmcnt(0), 8 MFMAs) | s_barrier.
//   * the two wave-rows run ONE BARRIER APART (wave-row 1 takes an extra s_barrier first): the SIMD partners (wave w
//     and w + 4) are always in opposite segments, so the matrix pipe of every SIMD is fed by one wave while the other
//     issues its LDS reads and DMA — nobody meets the barrier with an empty pipeline.
//   * the quadrant order frees LDS rows early (A0 and B0 rows are dead after phase 0, B1 after phase 1, A1 after
//     phase 2), so with only two 64-KB stages the DMA runs SIX phases (1.5 k-chunks) ahead of the MFMAs: the DMA
//     "pair" m (2 instructions per wave = 128 rows: pair 0 = A0 rows, 1 = B0, 2 = B1, 3 = A1 of chunk m / 4) is issued
//     in load segment m - 6 and waited for (s_waitcnt vmcnt(8): four younger pairs stay in flight) in load segment
//     m - 2 — one barrier before the first wave reads it (RAW: counted vmcnt, then a barrier, then the read; WAR: a
//     slot is re-filled at least two phases after its last read). vmcnt never drains to 0 inside the loop.
// Rows / taps in the zero padding read beyond num_records and land as zeros, as in conv_igemm_kernel.
template <typename TO, bool GROUPED = false>
__global__ __launch_bounds__(512) void conv_pp8_kernel(const ConvArgs a) {
    typedef _Float16 T;
    constexpr int MT = 4, NT = 2, WM = 2, WN = 4;
    constexpr int BM = 256, BN = 256;
    constexpr int STAGE = (BM + BN) * CHUNK_BYTES;            // 64 KB: A rows then B rows
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<TO, MT, NT, WM, WN, 1>();   // fp32: one wave-row per round; fp16: whole tile
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES];

    int M = a.M;
    if (a.m_dyn) {
        int md = *a.m_dyn * a.m_mul - a.m_off;
        md = md < 0 ? 0 : md;
        M = md < M ? md : M;
    }
    // the map this block works on: the launch's own, or (GROUPED) the pyramid level its tile index falls into
    const void* xp = a.x;
    const void* wp = a.w;
    int mapH = a.H, mapW = a.W, mapHo = a.Ho, mapWo = a.Wo;
    int lvl = 0, m0, n0;
    if constexpr (GROUPED) {
        if ((int)blockIdx.x >= a.ntiles) return;
        const int pid = xcd_remap(blockIdx.x, a.ntiles);
        while (lvl + 1 < a.nlev && pid >= a.lev[lvl + 1].tile0) ++lvl;
        lvl = __builtin_amdgcn_readfirstlane(lvl);
        xp = a.lev[lvl].x;
        wp = a.lev[lvl].w;
        mapH = mapHo = a.lev[lvl].H;
        mapW = mapWo = a.lev[lvl].W;
        M = a.lev[lvl].M;
        m0 = (pid - a.lev[lvl].tile0) * BM;
        n0 = 0;
    } else {
        const int tiles_n = (a.Cout + BN - 1) / BN;
        const int tiles_m = (M + BM - 1) / BM;
        const int nblk = tiles_m * tiles_n;
        if ((int)blockIdx.x >= nblk) return;
        const int pid = xcd_remap(blockIdx.x, nblk);
        const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
        m0 = tm * BM;
        n0 = tn * BN;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                  // wave-row = ping-pong group

    const int K = a.KH * a.KW * a.Cin;
    const int cchunks = a.Cin / 64;
    const int ntaps = a.KH * a.KW;
    const int nchunks = ntaps * cchunks;
    const unsigned pix_bytes = (unsigned)a.Cin * 2;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(xp), 0, (int)((size_t)a.B * mapH * mapW * pix_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(wp), 0, (int)((size_t)a.Cout * K * 2), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- staged rows of this thread: 2 per pair (lane = 8 rows x 8 pieces of one DMA instruction) -----------------
    // pair 0 (A0): tile rows i*128 + wave*8 + r          pair 3 (A1): the same + 64
    // pair 1 (B0): tile rows (2i + wave/4)*64 + (wave%4)*8 + r   pair 2 (B1): the same + 32      (i = 0, 1; r = lane/8)
    const int ld_c = lane & 7, ld_r = lane >> 3;
    unsigned a_off[4], a_ok[4];              // [pair 0: i=0,1 | pair 3: i=0,1]
    unsigned a_lds[4];                       // LDS byte offset (inside a stage) of the instruction's first row
    unsigned b_off[4], b_lds[4];             // [pair 1: i=0,1 | pair 2: i=0,1]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = k & 1, hi = k >> 1;
        const int row0 = i * 128 + hi * 64 + wave * 8;        // first row of the instruction (multiple of 8)
        const int row = row0 + ld_r;
        a_lds[k] = (unsigned)row0 * CHUNK_BYTES;
        const unsigned piece = (unsigned)(ld_c ^ ((row >> 1) & 7)) * 16;
        const int m = m0 + row;
        a_off[k] = 0;
        a_ok[k] = 0;
        if (m < M) {
            const int hw = mapHo * mapWo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / mapWo;
            const int ox = rem - oy * mapWo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            a_off[k] = (unsigned)((b * mapH + iy0) * mapW + ix0) * pix_bytes + piece;
            for (int ky = 0; ky < a.KH; ++ky)
                for (int kx = 0; kx < a.KW; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)mapH && (unsigned)(ix0 + kx) < (unsigned)mapW)
                        a_ok[k] |= 1u << (ky * a.KW + kx);
        }
        const int brow0 = (2 * i + (wave >> 2)) * 64 + hi * 32 + (wave & 3) * 8;
        const int brow = brow0 + ld_r;
        b_lds[k] = (unsigned)(BM + brow0) * CHUNK_BYTES;
        const unsigned bpiece = (unsigned)(ld_c ^ ((brow >> 1) & 7)) * 16;
        const int n = n0 + brow;
        b_off[k] = n < a.Cout ? (unsigned)n * (unsigned)K * 2 + bpiece : OOB;
    }

    typedef __attribute__((address_space(3))) void lds_void;
    // DMA cursor: pair index dm (0 .. 4*nchunks-1), its chunk's filter tap / channel chunk (scalar walk, chunk outer)
    int dm = 0, d_tap = 0, d_ky = 0, d_kx = 0, d_cc = 0;
    const int dm_end = 4 * nchunks;
    // pr = dm & 3 is passed as a compile-time constant (every call site knows it): the per-thread row tables are then
    // indexed statically and stay in registers
    auto issue_pair = [&](auto pr_c) {
        constexpr int pr = decltype(pr_c)::value;
        if (dm >= dm_end) return;
#if defined(TD_DIAG_PP8_NO_DMA)            // diagnostic builds only (tools/conv_diag.py pp8): prologue DMA only
        if (dm >= 6) { ++dm; return; }
#endif
        char* st = lds + ((dm >> 2) & 1) * STAGE;
        if constexpr (pr == 0 || pr == 3) {
            const unsigned xs = (unsigned)(d_ky * mapW + d_kx) * pix_bytes + (unsigned)d_cc * CHUNK_BYTES;
            constexpr int k0 = pr == 0 ? 0 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = ((a_ok[k0 + i] >> d_tap) & 1u) ? a_off[k0 + i] + xs : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(st + a_lds[k0 + i]), 16, off, 0, 0, 0);
            }
        } else {
            const unsigned ws = (unsigned)d_tap * pix_bytes + (unsigned)d_cc * CHUNK_BYTES;
            constexpr int k0 = pr == 1 ? 0 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned off = b_off[k0 + i] == OOB ? OOB : b_off[k0 + i] + ws;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(st + b_lds[k0 + i]), 16, off, 0, 0, 0);
            }
        }
        ++dm;
        if constexpr (pr == 3) {             // next chunk: tap inner, channel chunk outer
            if (++d_kx == a.KW) {
                d_kx = 0;
                ++d_ky;
            }
            if (++d_tap == ntaps) {
                d_tap = 0;
                d_ky = 0;
                d_kx = 0;
                ++d_cc;
            }
        }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    typedef std::integral_constant<int, 2> P2;
    typedef std::integral_constant<int, 3> P3;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row = lane & 31, piece = 2*kk + (lane >> 5), swizzled with the row's key (lane >> 1) & 7
    const unsigned swz = (lane >> 1) & 7, hi5 = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi5) ^ swz) * 16;
    const unsigned a_base = (unsigned)(wm * 128) * CHUNK_BYTES;
    const unsigned b_base = (unsigned)(BM + wn * 64) * CHUNK_BYTES;
    auto rd = [&](const char* st, unsigned base, int tile, int kk) -> f32x4 {
#if defined(TD_DIAG_PP8_NO_READS)
        f32x4 z = {1.f, 2.f, 3.f, (float)kk};
        asm volatile("" : "+v"(z));
        return z;
#else
        return *reinterpret_cast<const f32x4*>(st + base + tile * 32 * CHUNK_BYTES + frag_off[kk]);
#endif
    };
#if defined(TD_DIAG_PP8_NO_BARRIER)
#define TD_PP8_BARRIER() do {} while (0)
#else
#define TD_PP8_BARRIER() __builtin_amdgcn_s_barrier()
#endif

    // ---- prologue: pairs 0..5 in flight, pairs 0 and 1 (A0, B0 of chunk 0) landed ------------------------------------
    issue_pair(P0{}); issue_pair(P1{}); issue_pair(P2{}); issue_pair(P3{}); issue_pair(P0{}); issue_pair(P1{});
    if (dm >= 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // four younger pairs stay in flight
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");               // a single chunk: pairs 2, 3 are the younger ones
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();         // wave-row 1 runs one barrier behind wave-row 0

    f32x4 fa[2][4], fb0[4], fb1[4];
    // The compute segment. hipcc treats MFMAs as pure register arithmetic and would sink them past the closing barrier
    // into the next load segment (or hoist them above the wait): the empty asm statements make the fragments
    // "produced" after the lgkmcnt wait and the two accumulators "consumed" before the barrier, which pins the
    // cluster between the two s_barriers without hiding the MFMAs from the scheduler's hazard handling.
#define TD_PIN4(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])
#define TD_COMPUTE(FB, AI0, AI1, NJ)                                                                         \
    do {                                                                                                     \
        TD_PP8_BARRIER();                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("" : TD_PIN4(fa[0]), TD_PIN4(fa[1]), TD_PIN4(FB));                                      \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                   \
            Elem<T>::mma(fa[0][kk], FB[kk], acc[AI0][NJ]);                                                   \
            Elem<T>::mma(fa[1][kk], FB[kk], acc[AI1][NJ]);                                                   \
        }                                                                                                    \
        asm volatile("" : "+v"(acc[AI0][NJ]), "+v"(acc[AI1][NJ]));                                           \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        TD_PP8_BARRIER();                                                                                    \
    } while (0)
    // mode 0: steady state (every phase issues its pair; vmcnt(8) = the pair issued four phases ago has landed);
    // mode 1: second-to-last chunk (pairs remain for phases 0 and 1 only); mode 2: last chunk (nothing left to issue)
    auto run_chunk = [&](int c, auto mode_c) {
        constexpr int mode = decltype(mode_c)::value;
        const char* st = lds + (c & 1) * STAGE;
#if defined(TD_PP8_V2)
        // experiment: DMA only in the light phases — [P2,P3] of chunk c+1 in phase 1, [P0,P1] of chunk c+2 in phase 3
        // ---- phase 0 ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb0[kk] = rd(st, b_base, 0, kk);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = rd(st, a_base, 0, kk);
            fa[1][kk] = rd(st, a_base, 1, kk);
        }
        if constexpr (mode <= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TD_COMPUTE(fb0, 0, 1, 0);
        // ---- phase 1 ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb1[kk] = rd(st, b_base, 1, kk);
        if constexpr (mode <= 1) { issue_pair(P2{}); issue_pair(P3{}); }
        TD_COMPUTE(fb1, 0, 1, 1);
        // ---- phase 2 ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = rd(st, a_base, 2, kk);
            fa[1][kk] = rd(st, a_base, 3, kk);
        }
        TD_COMPUTE(fb1, 2, 3, 1);
        // ---- phase 3 ----
        if constexpr (mode == 0) { issue_pair(P0{}); issue_pair(P1{}); }
        if constexpr (mode == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (mode == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        TD_COMPUTE(fb0, 2, 3, 0);
#else
        // ---- phase 0: quadrant (A0, B0) ----
#if defined(TD_PP8_V1)
        if constexpr (mode <= 1) issue_pair(P2{});
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb0[kk] = rd(st, b_base, 0, kk);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = rd(st, a_base, 0, kk);
            fa[1][kk] = rd(st, a_base, 1, kk);
        }
#if !defined(TD_PP8_V1)
        if constexpr (mode <= 1) issue_pair(P2{});
#endif
        if constexpr (mode <= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        TD_COMPUTE(fb0, 0, 1, 0);
        // ---- phase 1: quadrant (A0, B1) ----
#if defined(TD_PP8_V1)
        if constexpr (mode <= 1) issue_pair(P3{});
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb1[kk] = rd(st, b_base, 1, kk);
#if !defined(TD_PP8_V1)
        if constexpr (mode <= 1) issue_pair(P3{});
#endif
        if constexpr (mode <= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TD_COMPUTE(fb1, 0, 1, 1);
        // ---- phase 2: quadrant (A1, B1) ----
#if defined(TD_PP8_V1)
        if constexpr (mode == 0) issue_pair(P0{});
#endif
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = rd(st, a_base, 2, kk);
            fa[1][kk] = rd(st, a_base, 3, kk);
        }
#if !defined(TD_PP8_V1)
        if constexpr (mode == 0) issue_pair(P0{});
#endif
        if constexpr (mode == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (mode == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        TD_COMPUTE(fb1, 2, 3, 1);
        // ---- phase 3: quadrant (A1, B0): no new fragments ----
        if constexpr (mode == 0) issue_pair(P1{});
        if constexpr (mode == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (mode == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        TD_COMPUTE(fb0, 2, 3, 0);
#endif
    };
    int c = 0;
    for (; c + 2 < nchunks; ++c) run_chunk(c, std::integral_constant<int, 0>{});
    if (nchunks >= 2) run_chunk(c++, std::integral_constant<int, 1>{});
    run_chunk(c, std::integral_constant<int, 2>{});
#undef TD_COMPUTE
#undef TD_PIN4
#undef TD_PP8_BARRIER
    if (wm == 0) __builtin_amdgcn_s_barrier();         // the barrier wave-row 1 took in the prologue
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is past its last fragment read: LDS is the epilogue's
#if defined(TD_DIAG_PP8_NO_EPILOGUE)                   // diagnostic builds only: keep the accumulators live, store one value
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) keep += acc[i][j][0] + acc[i][j][15];
    if (keep == 12345.678f) static_cast<TO*>(a.y)[tid] = (TO)keep;
    return;
#endif
    if constexpr (GROUPED) {
        ConvArgs ea = a;                                // this level's bias / output / head output
        ea.bias = a.lev[lvl].bias;
        ea.y = a.lev[lvl].y;
        ea.head_y = a.lev[lvl].head_y;
        ea.H = ea.Ho = mapH;
        ea.W = ea.Wo = mapW;
        conv_epilogue<T, TO, MT, NT, WM, WN, 1>(ea, acc, lds, M, m0, n0, tid, lane, wm, wn);
    } else {
        conv_epilogue<T, TO, MT, NT, WM, WN, 1>(a, acc, lds, M, m0, n0, tid, lane, wm, wn);
    }
}

template <typename T, typename TO, int MT, int NT, int NSTAGE, int WM = 2, int WN = 2>
td_status launch(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const int tiles = td_cdiv(a.M, BM) * td_cdiv(a.Cout, BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, TO, MT, NT, NSTAGE, WM, WN>), dim3(tiles, a.batch_count > 1 ? a.batch_count : 1),
                       dim3(64 * WM * WN), 0, stream, a);
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
        // 32-column tiles for the thin heads (RPN objectness + deltas: 15 rows, box predictor: 6): a 64-column tile spends three
        // quarters of its MFMAs on padding — the RPN head at p2 (M = 320 000, K = 256) was MFMA-bound on it (round 4)
        case 31: return launch<T, TO, 2, 1, 2, 4, 1>(a, stream);    // 256 x 32, 4 waves of 64 x 32
        case 32: return launch<T, TO, 1, 1, 2, 4, 1>(a, stream);    // 128 x 32, 4 waves of 32 x 32
        case 17:                                                     // 256 x 256, 8 waves, ping-pong phases (fp16 only)
            if constexpr (std::is_same<T, _Float16>::value) {
                const int tiles = td_cdiv(a.M, 256) * td_cdiv(a.Cout, 256);
                hipLaunchKernelGGL((conv_pp8_kernel<TO>), dim3(tiles), dim3(512), 0, stream, a);
                TD_KERNEL_CHECK();
                return TD_OK;
            } else {
                td_set_error("conv2d: tile_cfg 17 is an fp16 kernel");
                return TD_ERR_INVALID;
            }
        case 18:                                                     // persistent plane contractions (fp32 Winograd planes)
        case 19:
        case 20:
            if constexpr (std::is_same<T, float>::value && std::is_same<TO, float>::value) {
                const int planes = a.batch_count > 1 ? a.batch_count : 1;
                const int bm = cfg == 19 ? 128 : 64, bn = cfg == 20 ? 64 : 128;
                const long long total = (long long)td_cdiv(a.M, bm) * td_cdiv(a.Cout, bn) * planes;
                const int per_cu = (160 * 1024) / (2 * (bm + bn) * CHUNK_BYTES);       // resident blocks per CU by LDS
                const long long cap = 256ll * (per_cu > 0 ? per_cu : 1);
                const unsigned grid = (unsigned)(total < cap ? total : cap);
                if (cfg == 18) hipLaunchKernelGGL((plane_gemm_kernel<1, 2>), dim3(grid), dim3(256), 0, stream, a);
                else if (cfg == 19) hipLaunchKernelGGL((plane_gemm_kernel<2, 2>), dim3(grid), dim3(256), 0, stream, a);
                else hipLaunchKernelGGL((plane_gemm_kernel<1, 1>), dim3(grid), dim3(256), 0, stream, a);
                TD_KERNEL_CHECK();
                return TD_OK;
            } else {
                td_set_error("conv2d: tile_cfg 18-20 are fp32 kernels");
                return TD_ERR_INVALID;
            }
        default: td_set_error("conv2d: bad tile_cfg %d", cfg); return TD_ERR_INVALID;
    }
}

}  // namespace

// Winograd plane contractions with the input transform fused (fp32): x = layer input [B,H,W,Cin], w = U [16][Cout][Cin]
// (w_bs = Cout*Cin), y = M planes [16][M][Cout] (y_bs = M*Cout), M = tiles of this slab starting at tile m_off.
td_status wino_gemm_launch(const ConvArgs& a, hipStream_t stream) {
    TD_REQUIRE(a.Cin % 32 == 0 && a.Cout % 4 == 0 && a.M > 0, "winograd contraction: Cin %% 32, Cout %% 4, M > 0 required");
    TD_REQUIRE((size_t)a.B * a.H * a.W * (size_t)a.Cin * 4 < 0xfffffff0ull - (1u << 20), "winograd contraction: input tensor must stay below 4 GB");
    const long long nblk = (long long)td_cdiv(a.M, 128) * td_cdiv(a.Cout, 128) * 16;
    TD_REQUIRE(nblk < (1ll << 31), "winograd contraction: grid too large");
    hipLaunchKernelGGL(wino_gemm_kernel, dim3((unsigned)nblk), dim3(512), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

// tile ids 18-20 (plane_gemm_kernel) take the fp32 contractions with plain rows: the Winograd planes and the 1x1 / stride-1
// layers (scale, bias, same-size residual and ReLU are applied in the accumulator layout)
bool conv_plane_ok(const ConvArgs& a, int precision) {
    return precision == TD_PRECISION_FP32 && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.out_mode == 0 &&
           a.res_shift == 0 && !a.out_f32 && a.m_off == 0 && a.Cin >= 32 && a.Cin % 32 == 0 && a.M > 0 && a.Cout > 0 &&
           (a.batch_count <= 1 || !a.res) &&
           (size_t)a.M * a.Cin * 4 < 0xfffffff0ull - (1u << 20);
}

td_status conv_pp8_grouped_launch(ConvArgs a, hipStream_t stream) {
    TD_REQUIRE(a.nlev >= 1 && a.nlev <= 5, "grouped conv: 1 to 5 levels (got %d)", a.nlev);
    TD_REQUIRE(a.Cout == 256 && a.Cin % 64 == 0 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.res && a.out_mode == 0 && !a.m_dyn &&
                   a.batch_count <= 1 && !a.scale && !a.out_f32,
               "grouped conv: 3x3 / stride 1 / pad 1 layers to 256 channels with a bias only");
    TD_REQUIRE(!a.head_w || (a.head_n >= 1 && a.head_n <= 32), "grouped conv: at most 32 head rows");
    int tiles = 0;
    for (int l = 0; l < a.nlev; ++l) {
        ConvArgs::Level& L = a.lev[l];
        TD_REQUIRE(L.x && L.w && L.H >= 1 && L.W >= 1 && (a.head_w ? L.head_y != nullptr : L.y != nullptr), "grouped conv: level %d is incomplete", l);
        TD_REQUIRE((size_t)a.B * L.H * L.W * (size_t)a.Cin * 2 < 0xfffffff0ull - (1u << 20), "grouped conv: level %d input must stay below 4 GB", l);
        L.M = a.B * L.H * L.W;
        L.tile0 = tiles;
        tiles += td_cdiv(L.M, 256);
    }
    a.ntiles = tiles;
    a.H = a.Ho = a.lev[0].H;          // unused by the grouped kernel; kept coherent for error messages
    a.W = a.Wo = a.lev[0].W;
    a.M = a.lev[0].M;
    a.x = a.lev[0].x;
    a.w = a.lev[0].w;
    hipLaunchKernelGGL((conv_pp8_kernel<_Float16, true>), dim3(tiles), dim3(512), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status conv2d_launch(const ConvArgs& a, int precision, hipStream_t stream) {
    const int ke = precision == TD_PRECISION_FP16 ? 64 : 32;
    TD_REQUIRE(precision == TD_PRECISION_FP32 || precision == TD_PRECISION_FP16, "conv2d: bad precision %d", precision);
    TD_REQUIRE(a.Cin % ke == 0, "conv2d: Cin=%d must be a multiple of %d", a.Cin, ke);
    TD_REQUIRE(a.M > 0 && a.Cout > 0, "conv2d: empty problem (M=%d, Cout=%d)", a.M, a.Cout);
    const size_t es = precision == TD_PRECISION_FP16 ? 2 : 4;
    TD_REQUIRE((size_t)a.B * a.H * a.W * (size_t)a.Cin * es < 0xfffffff0ull - (1u << 20), "conv2d: input tensor must stay below 4 GB (32-bit buffer offsets)");
    TD_REQUIRE((size_t)a.Cout * a.KH * a.KW * a.Cin * es < 0xfffffff0ull - (1u << 20), "conv2d: weight tensor must stay below 4 GB");
    TD_REQUIRE(a.KH * a.KW <= 32, "conv2d: at most 32 filter taps (got %dx%d)", a.KH, a.KW);
    TD_REQUIRE(a.out_mode == 0 || (a.Cout % 32 == 0 && !a.res), "conv2d: bad deconv configuration");
    int cfg = a.tile_cfg;
    if (cfg < 0) {
        const char* forced = getenv("TD_CONV_CFG");            // diagnostics / tests only (read per call: tests switch it)
        if (forced) cfg = atoi(forced);
    }
    const bool no_fp16_tile = precision != TD_PRECISION_FP16 || a.out_mode != 0 || a.batch_count > 1;
    if (cfg == 17 && no_fp16_tile) cfg = -1;                       // fp16-only variant
    if (cfg == 28 && (no_fp16_tile || a.m_dyn)) cfg = -1;          // fp16-only, static row counts only
    if (cfg >= 18 && cfg <= 20 && !conv_plane_ok(a, precision)) cfg = -1;                                       // plane contractions only
    if (cfg == 33 && !conv_bs_ok(a, precision)) cfg = -1;                                                       // thin 1x1 layers with packed filters only
    if (conv_cfg_is_bd(cfg) && !conv_bd_ok(a, precision)) cfg = -1;                                             // needs the fragment-ordered filters
    // a fused head (ConvArgs::head_w) exists only in the tiles that stage all 256 output channels as one fp16 tile: any other
    // resolution of the tile id would silently write y and leave head_y untouched
    TD_REQUIRE(!a.head_w || (conv_head_capable(cfg, precision) && a.Cout == 256 && !a.res && a.out_mode == 0 && a.head_y && a.head_n >= 1 && a.head_n <= 32),
               "conv2d: a fused head needs a 256-channel fp16 layer on a tile that owns all its channels (tile id %d)", cfg);
    if (cfg == 33) return conv_bs_launch(a, precision, stream);
    if (conv_cfg_is_bd(cfg)) return conv_bd_launch(a, precision, cfg <= 27 ? cfg - 23 : cfg - 24, stream);
    TD_REQUIRE(a.batch_count <= 1 || (a.KH == 1 && a.KW == 1 && !a.res), "conv2d: batched launches are 1x1 contractions");
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
