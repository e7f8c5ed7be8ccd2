// conv_sk_kernel: stream-K implicit-GEMM convolution for the fp16 engine's small-map layers (res4 / res5, FPN and RPN at
// p4 - p6: M = B * Ho * Wo of 1 352 ... 20 000 rows at batch 8).
//
// Why: at the fp16 MFMA rate a CU computes a 128-B k-step faster than the LDS-DMA path can stage it (~60-70 GB/s per CU,
// MI355X_MICROARCH.md "ldsdma-fill"), so these layers are bound by the bytes staged per FLOP — (BM + BN) / (BM * BN) — and
// want LARGE block tiles; but M * N of such a layer holds only 40 ... 630 tiles of 128 x 128, which leaves 256 CUs with zero,
// one or two tiles each: the measured block-tile choice of conv_igemm_kernel for them is 64 x 64 ... 64 x 128 (twice the
// staged bytes per FLOP) because that is what fills the chip. Stream-K removes the trade: the (tile, k-step) space of the
// layer — tiles * nit units — is cut into G equal contiguous ranges, one per resident block, whatever the tile count. A
// block walks its range tile by tile with conv_igemm_kernel's data path (LDS-DMA of 128-B k-chunks into the XOR-swizzled
// image, two LDS stages, raw s_barrier, v_mfma_f32_32x32x16_f16; channel chunk outer, filter tap inner). A range that
// covers a whole tile ends in the usual epilogue. A range that starts or ends inside a tile leaves that tile's fp32
// partial sums in a workspace slot and takes a ticket on the tile's counter; whoever draws the LAST ticket adds the
// tile's partials IN SEGMENT ORDER (its own from registers) and runs the epilogue. Nobody ever waits for another block
// (no spin, no residency requirement: safe with three forwards in flight on three streams); visibility follows the
// guide's split-K recipe in its write-through form (sc1 slab stores → every wave's vmcnt(0) → workgroup barrier → lane 0's
// relaxed agent fetch_add; the last arriver reads the slabs with sc1 loads). The first form — plain stores + agent-scope
// release / acquire — cost 20-40 us per launch (each release writes back the XCD's whole L2).
// At most two partial tiles per block: <= 2 G slots of BM * BN floats are ever written per launch.
//
// MEASURED (round 3): slower than the plain block tiles on every layer of the R50 trunk it was built for except res5 conv2
// (profiles/r03_streamk_layers.txt: 1 588 us against 1 218 us summed over the layers the rule takes) — a block's partial-tile
// publish (128 KB of fp32), the last arriver's slab reads and the per-segment prologue (10-15 us per launch at one or two
// blocks per CU) exceed what the even k-step distribution saves. Kept as an opt-in path (TD_STREAMK=1) with
// its parity tests; the engine does not use it by default.
//
// Numerics: a tile's k range is summed in up to a few pieces instead of one chain — deterministic (the piece order is the
// segment order, the split points depend on M, N, K and the constant G only), but NOT the association of the other block
// tiles, and the split points move with the batch size. That is why only the fp16 engine uses it, by a FIXED RULE on the
// layer shape (engine.cpp), never by timing: fp16 activations carry 1e-3 of rounding noise per layer anyway
// (tests/test_engine_fp16_gpu.py), while the fp32 engine keeps its bit-for-bit batch invariance.
#include "experimental.h"
#include "../conv_tiles.h"

namespace {

// logical block (unit-range owner) → first unit of its range: floor(b * U / G)
__device__ __forceinline__ long long sk_first_unit(int b, long long U, int G) { return (long long)b * U / G; }
// the block whose range holds unit u: the largest b with floor(b * U / G) <= u
__device__ __forceinline__ int sk_owner(long long u, long long U, int G) { return (int)(((u + 1) * G - 1) / U); }

template <typename T, typename TO, int MT, int NT, int WM, int WN>
__device__ __forceinline__ void conv_sk_body(const ConvArgs& a, char* lds) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int LDROWS = THREADS / 8;
    constexpr int AROWS = BM / LDROWS, BROWS = BN / LDROWS;
    static_assert(BM % LDROWS == 0 && BN % LDROWS == 0, "tile / thread-count mismatch");
    constexpr int ES = sizeof(T);
    constexpr int KE = Elem<T>::PER_CHUNK;
    constexpr int STAGE_BYTES = 2 * (BM + BN) * CHUNK_BYTES;
    constexpr int WROW = 32 * MT, CS = BN + 4;
    constexpr int FIT = STAGE_BYTES / (WROW * CS * 4);
    constexpr int RWM = FIT >= WM ? WM : (FIT >= 2 && WM % 2 == 0 ? 2 : 1);
    char* As = lds;
    char* Bs = lds + 2 * BM * CHUNK_BYTES;

    const int M = a.M;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tiles_m = (M + BM - 1) / BM;
    const int ntaps = a.KH * a.KW;
    const int nit = ntaps * (a.Cin / KE);
    const long long U = (long long)tiles_m * tiles_n * nit;
    const int G = gridDim.x;                           // <= U (launcher)
    const int vb = xcd_remap(blockIdx.x, G);           // neighbouring ranges share an XCD (the A rows / filter panels they share stay in its L2)
    long long u = sk_first_unit(vb, U, G);
    const long long u_end = sk_first_unit(vb + 1, U, G);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ld_c = tid & 7, ld_r = tid >> 3;
    const int K = ntaps * a.Cin;
    const unsigned pix_bytes = (unsigned)a.Cin * ES;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.B * a.H * a.W * pix_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.w), 0, (int)((size_t)a.Cout * K * ES), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const unsigned src_piece = (unsigned)(ld_c ^ ((ld_r >> 1) & 7)) * 16;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned wave_rows = (unsigned)__builtin_amdgcn_readfirstlane(wave) * 8u;
    const unsigned swz = (lane >> 1) & 7, hi = lane >> 5;
    unsigned frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = (unsigned)(lane & 31) * CHUNK_BYTES + (((unsigned)(2 * kk) + hi) ^ swz) * 16;
    const bool plain_rows = ntaps == 1 && a.stride == 1 && a.pad == 0;

    while (u < u_end) {
        const int tile = (int)(u / nit);
        const int k0 = (int)(u - (long long)tile * nit);
        const long long left = u_end - u;
        const int k1 = left < (long long)(nit - k0) ? k0 + (int)left : nit;
        u += k1 - k0;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;

        unsigned a_off[AROWS], a_ok[AROWS];
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
        // scalar walk over (chunk outer, tap inner), starting at k-step k0 of the tile
        int ld_cc = k0 / ntaps;
        int ld_tap = k0 - ld_cc * ntaps;
        int ld_ky = ld_tap / a.KW;
        int ld_kx = ld_tap - ld_ky * a.KW;
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

        const int nsteps = k1 - k0;
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < nsteps; ++it) {
            const int cur = it & 1;
            if (it + 1 < nsteps) stage(cur ^ 1);
            const char* Ab = &As[(cur * BM + wm * 32 * MT) * CHUNK_BYTES];
            const char* Bb = &Bs[(cur * BN + wn * 32 * NT) * CHUNK_BYTES];
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        bool finish = true;                            // this block runs the tile's epilogue
        if (k0 != 0 || k1 != nit) {
            // ---- partial tile: publish the fp32 sums, take a ticket; the last arriver of the tile reduces and finishes ----
            // Write-through (sc1) 16-B slab stores and sc1 slab loads, no agent-scope fences: an agent release is a write-back
            // of the XCD's whole L2 (every other block's freshly written slab included) and cost 20-40 us per launch here
            // (MI355X_MICROARCH.md "publish-large": tens of KB per workgroup → write-through wins). Hand-off = the guide's
            // measured row "ONE lane of each storing workgroup adds to ONE unsharded counter; the workgroup whose add came
            // last, told by the value its add returned": every storing wave drains its stores (vmcnt(0)), the workgroup
            // barrier, then lane 0's relaxed agent-scope fetch_add; the last arriver's other waves load after a workgroup
            // barrier the adding wave joins; EVERY load of slab bytes is an sc1 load to registers.
            constexpr int TILE_BYTES = BM * BN * 4;
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            char* wsb = reinterpret_cast<char*>(a.sk_ws);
            const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(
                wsb + (size_t)(2 * vb + (k0 == 0 ? 1 : 0)) * TILE_BYTES, 0, TILE_BYTES, 0x00020000);
            // slot layout [wave][i][j][g][lane][4 floats]: a wave instruction moves 1 KB of consecutive bytes
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), mine,
                                                               ((((wave * MT + i) * NT + j) * 4 + g) * 64 + lane) * 16, 0, 16 /* sc1 */);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its own stores
            __syncthreads();
            const long long t_first = (long long)tile * nit;
            const int b_first = sk_owner(t_first, U, G), b_last = sk_owner(t_first + nit - 1, U, G);
            const int nseg = b_last - b_first + 1;
            int* flag = reinterpret_cast<int*>(lds);              // the ONE LDS array (a second __shared__ object de-pipelines the DMA waits)
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(&a.sk_cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = old;                                       // (the add has returned: its value is used)
            }
            __syncthreads();
            const int ticket = *flag;
            finish = ticket == nseg - 1;
            __syncthreads();                                       // everyone has read the flag word (the LDS stages are refilled next)
            if (finish) {
                // every piece comes from its slab, this block's own included (it was stored before the ticket was drawn): one
                // accumulator array stays live instead of two, and the order of the additions is the segment order by construction
                for (int s = 0; s < nseg; ++s) {
                    const __amdgpu_buffer_rsrc_t theirs = __builtin_amdgcn_make_buffer_rsrc(
                        wsb + (size_t)(2 * (b_first + s) + (s == 0 ? 1 : 0)) * TILE_BYTES, 0, TILE_BYTES, 0x00020000);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            f32x4 v[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                v[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                           theirs, ((((wave * MT + i) * NT + j) * 4 + g) * 64 + lane) * 16, 0, 16 /* sc1 */));
#pragma unroll
                            for (int g = 0; g < 4; ++g)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    acc[i][j][4 * g + q] = s == 0 ? v[g][q] : __fadd_rn(acc[i][j][4 * g + q], v[g][q]);
                        }
                }
                if (tid == 0) __hip_atomic_store(&a.sk_cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
        }
        if (finish) {
            conv_epilogue<T, TO, MT, NT, WM, WN, RWM>(a, acc, lds, M, m0, n0, tid, lane, wm, wn);
            __syncthreads();                           // the staging tile is read: the next segment may refill the LDS stages
        }
    }
}

template <typename T, typename TO, int MT, int NT, int WM, int WN, int BPC>
__global__ __launch_bounds__(64 * WM * WN, (BPC * WM * WN) / 4)
void conv_sk_kernel(const ConvArgs a) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int STAGE_BYTES = 2 * (BM + BN) * CHUNK_BYTES;
    constexpr int WROW = 32 * MT, CS = BN + 4;
    constexpr int FIT = STAGE_BYTES / (WROW * CS * 4);
    constexpr int RWM = FIT >= WM ? WM : (FIT >= 2 && WM % 2 == 0 ? 2 : 1);
    constexpr int EPI_BYTES = conv_epilogue_lds_bytes<TO, MT, NT, WM, WN, RWM>();
    constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
    static_assert(BPC * LDS_BYTES <= 160 * 1024, "LDS footprint does not allow that many blocks per CU");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    conv_sk_body<T, TO, MT, NT, WM, WN>(a, lds);
}

template <typename T, typename TO, int MT, int NT, int WM, int WN, int BPC>
td_status launch_sk(const ConvArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    const int ke = sizeof(T) == 2 ? 64 : 32;
    const long long U = (long long)td_cdiv(a.M, BM) * td_cdiv(a.Cout, BN) * (a.KH * a.KW * (a.Cin / ke));
    // at least 8 k-steps per block: with fewer, a tile is cut into dozens of pieces and its last arriver reads them all
    const long long cap = conv_sk_grid(BM, BN), by_work = U / 8 > 0 ? U / 8 : 1;
    const unsigned G = (unsigned)(by_work < cap ? by_work : cap);
    hipLaunchKernelGGL((conv_sk_kernel<T, TO, MT, NT, WM, WN, BPC>), dim3(G), dim3(64 * WM * WN), 0, stream, a);
    TD_KERNEL_CHECK();
    return TD_OK;
}

}  // namespace

// resident blocks the launch is cut into: a CONSTANT of the tile shape (the split points, hence the rounding, depend on it)
int conv_sk_grid(int bm, int bn) { return bm * bn > 128 * 128 ? 256 : 512; }
size_t conv_sk_workspace_floats(void) { return (size_t)2 * 512 * 128 * 128; }      // 2 slots per block, either tile shape (2 * 256 * 256 * 128 is the same)
int conv_sk_max_tiles(void) { return 1 << 16; }

td_status conv_sk_launch(const ConvArgs& a, int precision, int variant, hipStream_t stream) {
    TD_REQUIRE(precision == TD_PRECISION_FP16, "stream-K convolution: fp16 engine only");
    TD_REQUIRE(a.sk_ws && a.sk_cnt, "stream-K convolution: workspace missing");
    TD_REQUIRE(a.Cin % 64 == 0 && a.M > 0 && a.Cout > 0 && a.out_mode == 0 && a.batch_count <= 1 && !a.m_dyn && a.KH * a.KW <= 32,
               "stream-K convolution: unsupported launch shape");
    TD_REQUIRE((size_t)a.B * a.H * a.W * (size_t)a.Cin * 2 < 0xfffffff0ull - (1u << 20), "stream-K convolution: input tensor must stay below 4 GB");
    TD_REQUIRE((size_t)a.Cout * a.KH * a.KW * a.Cin * 2 < 0xfffffff0ull - (1u << 20), "stream-K convolution: weight tensor must stay below 4 GB");
    const int bm = variant == 1 ? 256 : 128;
    TD_REQUIRE((long long)td_cdiv(a.M, bm) * td_cdiv(a.Cout, 128) <= conv_sk_max_tiles(), "stream-K convolution: too many tiles for the ticket array");
    if (variant == 1) {
        if (a.out_f32) return launch_sk<_Float16, float, 2, 2, 4, 2, 1>(a, stream);
        return launch_sk<_Float16, _Float16, 2, 2, 4, 2, 1>(a, stream);
    }
    if (a.out_f32) return launch_sk<_Float16, float, 2, 2, 2, 2, 2>(a, stream);
    return launch_sk<_Float16, _Float16, 2, 2, 2, 2, 2>(a, stream);
}
