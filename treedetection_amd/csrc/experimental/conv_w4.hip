// conv_w4_kernel lives in a file of its own because of ONE compiler flag: conv_igemm.hip is built with
// -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs: the register-finished epilogues of the 4- to 16-wave tiles want them
// there), and with 256 accumulator registers per lane that form makes hipcc shuffle whole accumulators through a 16-register
// window around every MFMA (11 000 lines of v_accvgpr_mov for this kernel: 564 us against conv_pp8_kernel's 465 on the big
// 3x3). Built without the flag the accumulators sit in the 256 AGPRs and the 256 VGPRs hold everything else — the split
// this kernel's register budget was drawn for.
#include "experimental.h"
#include "../conv_tiles.h"

namespace {

// ---- conv_w4_kernel: 256 x 256 block tile, FOUR waves of 128 x 128, software-pipelined inside the wave (fp16 only) ----------
// conv_pp8_kernel reads 192 KB of fragments and takes 8 barriers per 64-deep chunk and CU; its MFMA loop alone runs at 58 % of
// peak and fragment reads + DMA cost another 29 % together (the LDS is the shared resource). Here a wave owns
// 128 x 128 outputs = 4 x 4 MFMA tiles (256 accumulator registers; one wave per SIMD, 512 registers each):
//   * 32 ds_read_b128 per 64 MFMAs (0.5 per MFMA against 0.75): 128 KB of fragment reads per chunk and CU;
//   * ONE barrier per chunk: the k-sub-steps of a chunk are pipelined in registers — the fragments of sub-step kk + 1 are
//     read while the 16 MFMAs of sub-step kk issue (two fragment sets), the first sub-step of the next chunk while the last
//     one of this chunk computes;
//   * DMA one chunk ahead: the 16 instructions of chunk c + 2 are issued right after the barrier that ends chunk c's reads,
//     into the stage those reads just left; vmcnt(0) before the next barrier waits for data issued a whole chunk earlier.
// Same LDS image, k order and MFMA as every other fp16 tile → bit-identical sums.
template <typename TO, bool GROUPED = false>
__global__ __launch_bounds__(256) void conv_w4_kernel(const ConvArgs a) {
    typedef _Float16 T;
    constexpr int MT = 4, NT = 4, WM = 2, WN = 2;
    constexpr int BM = 256, BN = 256;
    constexpr int STAGE = (BM + BN) * CHUNK_BYTES;            // 64 KB: A rows then B rows
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<TO, MT, NT, WM, WN, 1>();
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES];

    int M = a.M;
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
    const int wm = wave >> 1, wn = wave & 1;

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

    // ---- staged rows of this thread: DMA instruction g = 4 j + wave (j = 0..15) covers stage rows 8 g .. 8 g + 7 (lane = row x
    // piece); j < 8: A rows, j >= 8: B rows ------------------------------------------------------------------------------------
    const int ld_c = lane & 7, ld_r = lane >> 3;
    unsigned a_off[8], a_ok[8], b_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (4 * j + wave) * 8 + ld_r;            // 0 .. 255
        const unsigned piece = (unsigned)(ld_c ^ ((row >> 1) & 7)) * 16;
        const int m = m0 + row;
        a_off[j] = 0;
        a_ok[j] = 0;
        if (m < M) {
            const int hw = mapHo * mapWo;
            const int b = m / hw;
            const int rem = m - b * hw;
            const int oy = rem / mapWo;
            const int ox = rem - oy * mapWo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            a_off[j] = (unsigned)((b * mapH + iy0) * mapW + ix0) * pix_bytes + piece;
            for (int ky = 0; ky < a.KH; ++ky)
                for (int kx = 0; kx < a.KW; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)mapH && (unsigned)(ix0 + kx) < (unsigned)mapW)
                        a_ok[j] |= 1u << (ky * a.KW + kx);
        }
        const int n = n0 + row;                               // the B row of instruction j + 8 has the same index inside its half
        b_off[j] = n < a.Cout ? (unsigned)n * (unsigned)K * 2 + piece : OOB;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    int d_tap = 0, d_ky = 0, d_kx = 0, d_cc = 0, d_chunk = 0;      // DMA cursor (chunk d_chunk): filter tap inner, channel chunk outer
    auto issue_chunk = [&]() {
        if (d_chunk >= nchunks) return;
#if defined(TD_DIAG_W4_NO_DMA)             // diagnostic builds only (tools/conv_diag.py w4diag): prologue DMA only
        if (d_chunk >= 2) { ++d_chunk; return; }
#endif
        char* st = lds + (d_chunk & 1) * STAGE;
        const unsigned xs = (unsigned)(d_ky * mapW + d_kx) * pix_bytes + (unsigned)d_cc * CHUNK_BYTES;
        const unsigned ws = (unsigned)d_tap * pix_bytes + (unsigned)d_cc * CHUNK_BYTES;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = ((a_ok[j] >> d_tap) & 1u) ? a_off[j] + xs : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(st + (unsigned)((4 * j + wave) * 8) * CHUNK_BYTES), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = b_off[j] == OOB ? OOB : b_off[j] + ws;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(st + (unsigned)(BM + (4 * j + wave) * 8) * CHUNK_BYTES), 16, off, 0, 0, 0);
        }
        ++d_chunk;
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
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row = lane & 31, piece = 2 kk + (lane >> 5), swizzled with the row's key (lane >> 1) & 7
    const unsigned swz = (lane >> 1) & 7, hi5 = lane >> 5;
    const unsigned a_base = (unsigned)(wm * 128 + (lane & 31)) * CHUNK_BYTES;
    const unsigned b_base = (unsigned)(BM + wn * 128 + (lane & 31)) * CHUNK_BYTES;
    f32x4 fa[2][MT], fb[2][NT];
    auto read_frags = [&](const char* st, int kk, int set) {
#if defined(TD_DIAG_W4_NO_READS)
        if (kk >= 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) { f32x4 z = {1.f, 2.f, 3.f, (float)kk}; asm volatile("" : "+v"(z)); fa[set][i] = z; fb[set][i] = z; }
            return;
        }
#endif
        const unsigned po = (((unsigned)(2 * kk) + hi5) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(st + a_base + i * 32 * CHUNK_BYTES + po);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(st + b_base + j * 32 * CHUNK_BYTES + po);
    };
    auto mma8 = [&](int set, int half) {               // rows 2 half, 2 half + 1 of the wave's 4 x 4 MFMA tiles
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) Elem<T>::mma(fa[set][i], fb[set][j], acc[i][j]);
    };
    auto mma16 = [&](int set) {
        mma8(set, 0);
        mma8(set, 1);
    };
#define TD_W4_PIN(set)                                                                                                   \
    asm volatile("" : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fb[set][0]), "+v"(fb[set][1]), \
                 "+v"(fb[set][2]), "+v"(fb[set][3]))

    // ---- prologue: chunks 0 and 1 in flight; chunk 0 landed; its first fragments in registers ---------------------------------
    issue_chunk();
    issue_chunk();
    if (nchunks >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(lds, 0, 0);
    for (int c = 0; c < nchunks; ++c) {
        const char* st = lds + (c & 1) * STAGE;
        // sub-steps 0 .. 2: fragments of kk + 1 are read while the MFMAs of kk issue
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            read_frags(st, kk + 1, (kk + 1) & 1);
            mma16(kk & 1);
            // issue order of the sub-step: one MFMA, one fragment read, eight times, then eight MFMAs — a burst of 8 ds_read_b128 per wave (32 KB per
            // CU) fills the LDS request queue and the wave's MFMAs wait behind its own reads (ablation: reads cost 100+ us of 440)
#pragma unroll
            for (int q = 0; q < 8; ++q) {                               // the reads ride on the first eight MFMAs: landed a quarter
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // of a sub-step before the next one wants them
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        // sub-step 3: the stage is read out (set 1 holds kk = 3). Next chunk's data must have landed in every wave's share, and
        // the stage just read may be refilled: wait own DMA, barrier, refill, first fragments of the next chunk, then compute.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TD_W4_PIN(1);
        // the first half of the MFMAs runs under the wait for the other waves, the second half under the latency of the next
        // chunk's first fragment reads and the DMA issue
        mma8(1, 0);
        if (c + 1 < nchunks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read_frags(lds + ((c + 1) & 1) * STAGE, 0, 0);
            issue_chunk();                                               // chunk c + 2 into the stage of chunk c
        }
        asm volatile("" : "+v"(fa[1][2]), "+v"(fa[1][3]));
        mma8(1, 1);
    }
#undef TD_W4_PIN
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is past its last fragment read: LDS is the epilogue's
#if defined(TD_DIAG_W4_NO_EPILOGUE)
    {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) keep += acc[i][j][0] + acc[i][j][15];
        if (keep == 12345.678f) static_cast<TO*>(a.y)[tid] = (TO)keep;
        return;
    }
#endif
    if constexpr (GROUPED) {
        ConvArgs ea = a;
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

}  // namespace

td_status conv_w4_launch(const ConvArgs& a, bool out_f32, hipStream_t stream) {
    TD_REQUIRE(a.Cin % 64 == 0 && a.out_mode == 0 && a.batch_count <= 1 && !a.m_dyn, "conv_w4: fp16 layers with static row counts only");
    const int tiles = td_cdiv(a.M, 256) * td_cdiv(a.Cout, 256);
    if (out_f32) hipLaunchKernelGGL((conv_w4_kernel<float>), dim3(tiles), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv_w4_kernel<_Float16>), dim3(tiles), dim3(256), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}
