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
//
// Round 5 — the same kernel for grids that cannot fill the chip and for 512 channels. NWT = tile-waves per block: 4 (the geometry
// above: 64 tiles, 512 threads) or 2 (32 tiles x 64 channels, 256 threads = 2 tile halves x 2 channel halves; per WAVE nothing
// changes — 16 tiles x 32 channels, the same fragments, MFMAs, fold and stores — a block just has half the rows: a layer of
// 1 352 tiles x 256 channels (res4 conv2, FPN output 4 at batch 8) is 172 blocks instead of 88 on 256 CUs). KREP = ring passes
// per plane: a plane of C / 32 chunks no longer has to equal the NS LDS stages — C = 256 on a 4-stage ring (48 KB: two 256-thread
// blocks per CU) is two passes, C = 512 four; the fold slices ride in the first pass only. Every output element sees the same
// chunks in the same order through the same instruction in every variant: NWT / NS / KREP are bit-identical to each other
// (tests/test_conv_gpu.py), so the launcher may pick them by the launch's size.
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

constexpr int WF_BN = 64;                        // output channels per block; tiles per block = 16 NWT

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

// VAR (TD_WF_VAR) = timing diagnostics with WRONG results: bit 3 = no DMA inside the loop, bit 4 = no MFMAs, bit 5 = no fragment
// reads, bit 6 = no barriers
template <int NS, int KREP, int NWT, int VAR>
__global__ __launch_bounds__(NWT * 128, 2) void wino43_fused_kernel(const WinoFusedArgs a) {
    constexpr int AHEAD = NS - 1;                     // chunks of DMA in flight; NS LDS stages (see the WAR argument at the barrier)
    constexpr int WF_BT = 16 * NWT, THREADS = 128 * NWT;
    constexpr int DROWS = THREADS / 8;                // rows one block-wide DMA instruction moves (= WF_BT)
    constexpr int NU = WF_BN / DROWS;                 // DMA instructions per chunk for the 64 U rows: 1 (NWT = 4) or 2 (NWT = 2)
    constexpr int NDMA = 1 + NU;                      // per wave and chunk: what the counted waits count in
    constexpr int WF_STAGE = (WF_BT + WF_BN) * CHUNK_BYTES;      // V rows then U rows: 16 KB (NWT = 4) / 12 KB (NWT = 2)
    static_assert((NWT == 2 || NWT == 4) && (NS == 4 || NS == 8) && KREP >= 1 && NS * WF_STAGE <= 160 * 1024 && DROWS == WF_BT, "geometry");
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
    const int wt = wave & (NWT - 1), wc = wave / NWT;
    constexpr int KC = NS * KREP;                     // k-chunks of 32 floats per plane: C = 32 NS KREP (launcher)

    // ---- LDS-DMA source offsets (bytes into V / U; rows past the live tiles read beyond num_records → zeros) ----
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned row_bytes = (unsigned)a.C * 4u;
    const unsigned planeV = (unsigned)((unsigned long long)a.T * row_bytes), planeU = (unsigned)a.N * row_bytes;
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V), 0, (int)(36u * planeV), 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.U), 0, (int)(36u * planeU), 0x00020000);
    const int ld_c = tid & 7, ld_r = tid >> 3;       // 16-B piece, row 0..DROWS-1 (a wave instruction = 8 rows x 128 B)
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16u;
    const bool v_ok = t0 + ld_r < live;
    const unsigned v_off = (unsigned)(t0 + ld_r) * row_bytes + src_piece;
    const unsigned u_off = (unsigned)(n0 + ld_r) * row_bytes + src_piece;       // (+ DROWS rows for the second U instruction: same swizzle, DROWS % 16 == 0)
    typedef __attribute__((address_space(3))) void lds_void;
    char* const dstV = lds + wave * 8 * CHUNK_BYTES;
    char* const dstU = lds + WF_BT * CHUNK_BYTES + wave * 8 * CHUNK_BYTES;
    int ld_kc = 0;
    unsigned ld_v = 0, ld_u = 0;                      // plane * plane bytes + chunk * 128 of the next chunk to issue
    auto issue = [&](auto st_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(st_c)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(dstV + ST * WF_STAGE), 16, v_ok ? v_off + ld_v : OOB, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(dstU + ST * WF_STAGE), 16, u_off + ld_u, 0, 0, 0);
        if constexpr (NU == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void*)(dstU + ST * WF_STAGE + DROWS * CHUNK_BYTES), 16, u_off + (unsigned)DROWS * row_bytes + ld_u, 0, 0, 0);
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
    auto read_frag = [&](Frag& f, auto st_c, unsigned pc) __attribute__((always_inline)) {
        if constexpr (VAR & 32) {
            asm volatile("" : "+v"(f.u0), "+v"(f.u1), "+v"(f.v));
            return;
        }
        const char* sb = lds + decltype(st_c)::value * WF_STAGE;
        f.u0 = *reinterpret_cast<const f32x4*>(sb + fu + pc);
        f.u1 = *reinterpret_cast<const f32x4*>(sb + fu + 16 * CHUNK_BYTES + pc);
        f.v = *reinterpret_cast<const f32x4*>(sb + fv + pc);
    };
    // Two sets of plane accumulators: while the MFMAs of plane p run into set p & 1, the finished product of plane p - 1 (the
    // other set) is folded into S — and, after the sixth plane of a row, S into Y — by VALU slices placed in the gaps between
    // the MFMA pairs of plane p's first two chunk-steps (an MFMA holds the SIMD's vector issue for a quarter of its cycles; with
    // the fold after the k loop both waves of a SIMD folded at the same time and the matrix pipe idled: 6 % of the kernel).
    f32x4 acc[2][2];
    auto mma = [&](const Frag& f, auto set_c, auto&& filler, auto slot0_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value, SLOT0 = decltype(slot0_c)::value;
        if constexpr (VAR & 16) {
            asm volatile("" ::"v"(f.u0), "v"(f.u1), "v"(f.v));
            wf_static_for<4>([&](auto e_c) __attribute__((always_inline)) { filler(std::integral_constant<int, SLOT0 + decltype(e_c)::value>{}); });
            return;
        }
        wf_static_for<4>([&](auto e_c) __attribute__((always_inline)) {
            constexpr int e = decltype(e_c)::value;
            acc[SET][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.u0[e], f.v[e], acc[SET][0], 0, 0, 0);
            acc[SET][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.u1[e], f.v[e], acc[SET][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);        // the two accumulator chains stay in alternation (dependent latency 40 > issue 32)
            filler(std::integral_constant<int, SLOT0 + e>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    f32x4 Y[16][2], S[4][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[i][0] = Y[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) S[j][0] = S[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fold slices. A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] (rows i, columns = plane component).
    // S slice `sl` of 8: element (b, e) = (sl >> 2, sl & 3) of the finished plane XC: S[j] += A^T[j][XC] * m, zero terms skipped,
    // +-1 as add / sub, 2 4 8 as one fma (exact constants): 1 - 4 VALU; the accumulator element is cleared for its next plane.
    auto fold_s = [&](auto xc_c, auto set_c, auto sl_c) __attribute__((always_inline)) {
        constexpr int XC = decltype(xc_c)::value, SET = decltype(set_c)::value, sl = decltype(sl_c)::value, b = sl >> 2, e = sl & 3;
        const float m = acc[SET][b][e];
        acc[SET][b][e] = 0.f;
        if constexpr (XC <= 4) S[0][b][e] = __fadd_rn(S[0][b][e], m);
        if constexpr (XC == 1) {
            S[1][b][e] = __fadd_rn(S[1][b][e], m);
            S[2][b][e] = __fadd_rn(S[2][b][e], m);
            S[3][b][e] = __fadd_rn(S[3][b][e], m);
        } else if constexpr (XC == 2) {
            S[1][b][e] = __fsub_rn(S[1][b][e], m);
            S[2][b][e] = __fadd_rn(S[2][b][e], m);
            S[3][b][e] = __fsub_rn(S[3][b][e], m);
        } else if constexpr (XC == 3) {
            S[1][b][e] = __fmaf_rn(2.f, m, S[1][b][e]);
            S[2][b][e] = __fmaf_rn(4.f, m, S[2][b][e]);
            S[3][b][e] = __fmaf_rn(8.f, m, S[3][b][e]);
        } else if constexpr (XC == 4) {
            S[1][b][e] = __fmaf_rn(-2.f, m, S[1][b][e]);
            S[2][b][e] = __fmaf_rn(4.f, m, S[2][b][e]);
            S[3][b][e] = __fmaf_rn(-8.f, m, S[3][b][e]);
        } else if constexpr (XC == 5) {
            S[3][b][e] = __fadd_rn(S[3][b][e], m);
        }
    };
    // Y slice `sl` of 8: element (b, e) of the finished row: Y[i][j] += A^T[i][row] * S[j] (runtime row: r0..r3 — fma(1, v, y) = y + v
    // and fma(0, v, y) = y exactly, so this IS the sparse sum), S[j] = 0: 16 fma
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    auto fold_y = [&](auto sl_c) __attribute__((always_inline)) {
        constexpr int sl = decltype(sl_c)::value, b = sl >> 2, e = sl & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = S[j][b][e];
            S[j][b][e] = 0.f;
            Y[0 + j][b][e] = __fmaf_rn(r0, v, Y[0 + j][b][e]);
            Y[4 + j][b][e] = __fmaf_rn(r1, v, Y[4 + j][b][e]);
            Y[8 + j][b][e] = __fmaf_rn(r2, v, Y[8 + j][b][e]);
            Y[12 + j][b][e] = __fmaf_rn(r3, v, Y[12 + j][b][e]);
        }
    };

    // ---- prologue: chunks 0 .. AHEAD-1 in flight (stages 0 .. AHEAD-1); chunk 0 landed and its first half-fragment read ----
    wf_static_for<AHEAD>([&](auto i_c) __attribute__((always_inline)) { issue(i_c); });          // KC % NS == 0 (launcher): the stream is longer than the pipeline
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (AHEAD - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    if constexpr (VAR & 32) {
        f0.u0 = f0.u1 = f0.v = f32x4{1.f, 2.f, 3.f, 4.f};
        f1 = f0;
    }
    read_frag(f0, std::integral_constant<int, 0>{}, pc0);

    // Step s (chunk s of the flat stream, stage s % NS): on entry chunk s has landed for every wave and f0 holds its first
    // half (read during step s - 1). The DMA of chunk s + AHEAD goes into the stage chunk s - 1 occupied: every wave finished
    // its reads of chunk s - 1 before the barrier that ended step s - 1 (lgkmcnt(0) sits in front of each barrier), and this
    // wave is past that barrier. Before the barrier that ends step s every wave waits until its own DMAs of chunk s + 1 have
    // landed (vmcnt: all but the AHEAD - 1 chunks issued after it, two DMAs each), so after the barrier chunk s + 1 is complete
    // for all. A plane is exactly NS chunks (C = 32 NS: launcher) and its loop is unrolled: every LDS address is a base register
    // + an immediate. EVERY step issues a chunk: past the end of the stream the offsets lie beyond num_records and the DMA
    // writes zeros into stages nobody reads again — no tail variant of the loop; the wave drains its DMAs before it ends.
    // XC = this plane's component (its accumulators: set XC & 1). Its first two steps carry the fold of the plane before it
    // (component (XC + 5) % 6, set (XC + 1) & 1): step 0 the S slices, step 1 — when that plane closed a row — the row's Y slices.
    // FOLD = the plane's first ring pass (it carries the fold slices of the plane before it); the KREP - 1 further passes of a
    // plane longer than the ring run the same steps with empty fillers.
    auto ring_pass = [&](auto xc_c, auto fold_c) __attribute__((always_inline)) {
        constexpr int XC = decltype(xc_c)::value, SET = XC & 1, PXC = (XC + 5) % 6;
        constexpr bool FOLD = decltype(fold_c)::value;
        wf_static_for<NS>([&](auto st_c) __attribute__((always_inline)) {
            constexpr int ST = decltype(st_c)::value, NEXT = (ST + 1) % NS;
            auto filler = [&](auto slot_c) __attribute__((always_inline)) {
                if constexpr (FOLD && ST == 0) fold_s(std::integral_constant<int, PXC>{}, std::integral_constant<int, 1 - SET>{}, slot_c);
                if constexpr (FOLD && ST == 1 && PXC == 5) fold_y(slot_c);
            };
            if constexpr (!(VAR & 8)) issue(std::integral_constant<int, (ST + AHEAD) % NS>{});
            read_frag(f1, st_c, pc1);
            mma(f0, std::integral_constant<int, SET>{}, filler, std::integral_constant<int, 0>{});
            if constexpr (!(VAR & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (AHEAD - 1)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(VAR & 64)) __builtin_amdgcn_s_barrier();
            read_frag(f0, std::integral_constant<int, NEXT>{}, pc0);
            mma(f1, std::integral_constant<int, SET>{}, filler, std::integral_constant<int, 4>{});
        });
    };
    auto plane = [&](auto xc_c) __attribute__((always_inline)) {
        ring_pass(xc_c, std::true_type{});
        if constexpr (KREP > 1) {
#pragma unroll 1
            for (int rep = 1; rep < KREP; ++rep) ring_pass(xc_c, std::false_type{});
        }
    };
    for (int xr = 0; xr < 6; ++xr) {
        // coefficients of the row that closed before this one (none before row 0: S is zero and so are they)
        r0 = xr ? WF_AT[0][xr - 1] : 0.f; r1 = xr ? WF_AT[1][xr - 1] : 0.f; r2 = xr ? WF_AT[2][xr - 1] : 0.f; r3 = xr ? WF_AT[3][xr - 1] : 0.f;
        wf_static_for<6>(plane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero-fill DMAs past the stream's end: LDS is released only after they landed
    // the last plane (component 5, set 1) and the last row
    wf_static_for<8>([&](auto sl_c) __attribute__((always_inline)) { fold_s(std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{}, sl_c); });
    r0 = WF_AT[0][5]; r1 = WF_AT[1][5]; r2 = WF_AT[2][5]; r3 = WF_AT[3][5];
    wf_static_for<8>([&](auto sl_c) __attribute__((always_inline)) { fold_y(sl_c); });

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
    return (C == 128 || C == 256 || C == 512) && N >= WF_BN && N % WF_BN == 0 && 36ull * (unsigned long long)T * C * 4 < 0xfffffff0ull - (1u << 20) &&
           36ull * (unsigned long long)N * C * 4 < 0xfffffff0ull - (1u << 20) && (unsigned long long)B * H * W * N * 4 < 0xfffffff0ull - (1u << 20);
}

// V [36][T][C] (wino43_input_launch) x U [36][N][C] → y [B,H,W,N] = act((A^T (sum_c U .* V) A) * scale + bias)
td_status wino43_fused_launch(const float* V, const float* U, int B, int H, int W, int C, int N, const float* scale, const float* bias,
                              int relu, float* y, const int* m_dyn, hipStream_t s) {
    TD_REQUIRE(V && U && y && B >= 1 && H >= 1 && W >= 1, "winograd F(4x4) fused contraction: bad arguments");
    TD_REQUIRE(wino43_fused_ok(B, H, W, C, N), "winograd F(4x4) fused contraction: needs C = 128, 256 or 512, N %% 64 == 0 and planes below 4 GB (C %d, N %d)", C, N);
    WinoFusedArgs a{};
    a.V = V; a.U = U; a.scale = scale; a.bias = bias; a.y = y; a.m_dyn = m_dyn;
    a.B = B; a.H = H; a.W = W; a.C = C; a.N = N; a.relu = relu;
    a.T = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    // Block geometry by the launch's size (every variant gives the same bits): 64-tile blocks where they fill the chip — at
    // least one block per CU —, else 32-tile blocks (res4 / res5 conv2, FPN output 4 / 5 at batch 8: 88 - 104 blocks of 64 tiles
    // on 256 CUs). C = 512 always takes them (a 4-stage ring, four passes per plane). TD_WF_NWT = 2 / 4 forces one (tests, probes).
    static const int num_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 64) n = 256;
        return n;
    }();
    const long long blocks64 = ((a.T + 63) / 64) * (N / WF_BN);
    int nwt = (C == 512 || (blocks64 < num_cu && !m_dyn)) ? 2 : 4;
    if (const char* f = getenv("TD_WF_NWT")) {
        const int v = atoi(f);
        if ((v == 2 || v == 4) && !(v == 4 && C == 512)) nwt = v;
    }
    const long long blocks = ((a.T + 16 * nwt - 1) / (16 * nwt)) * (N / WF_BN);
    TD_REQUIRE(blocks < (1ll << 31), "winograd F(4x4) fused contraction: grid too large");
#define TD_WF_LAUNCH(NSV, KREPV, NWTV, VARV) hipLaunchKernelGGL((wino43_fused_kernel<NSV, KREPV, NWTV, VARV>), dim3((unsigned)blocks), dim3(128 * NWTV), 0, s, a)
#if defined(TD_WF_DIAG)     // timing builds only (tools/wino_fold_probe.py builds this file with -DTD_WF_DIAG into /tmp; never shipped): TD_WF_VAR picks an ablation
    const int var = getenv("TD_WF_VAR") ? atoi(getenv("TD_WF_VAR")) : 0;
    if (var && C == 256 && nwt == 4) {
        switch (var) {
            case 8: TD_WF_LAUNCH(8, 1, 4, 8); break;        // wrong results: no DMA in the loop
            case 16: TD_WF_LAUNCH(8, 1, 4, 16); break;      // no MFMAs
            case 32: TD_WF_LAUNCH(8, 1, 4, 32); break;      // no fragment reads
            case 40: TD_WF_LAUNCH(8, 1, 4, 40); break;      // MFMAs + barriers (+ fold) only
            case 104: TD_WF_LAUNCH(8, 1, 4, 104); break;    // MFMAs (+ fold) only
            default: TD_WF_LAUNCH(8, 1, 4, 0); break;
        }
        TD_KERNEL_CHECK();
        return TD_OK;
    }
#endif
    // a plane = C / 32 chunk-steps = NS LDS stages x KREP ring passes. 64-tile blocks: 8 stages at C = 256 (seven chunks of DMA in
    // flight, 128 KB, one block per CU), 4 at C = 128; 32-tile blocks: 4 stages of 12 KB (two blocks per CU)
    if (nwt == 4) {
        if (C == 256) TD_WF_LAUNCH(8, 1, 4, 0);
        else TD_WF_LAUNCH(4, 1, 4, 0);
    } else {
        if (C == 512) TD_WF_LAUNCH(4, 4, 2, 0);
        else if (C == 256) TD_WF_LAUNCH(4, 2, 2, 0);
        else TD_WF_LAUNCH(4, 1, 2, 0);
    }
#undef TD_WF_LAUNCH
    TD_KERNEL_CHECK();
    return TD_OK;
}
