// TIFF LZW blocks decoded on the GPU (the tile producer's raster input). The reference reads every tile
// window through rasterio → GDAL → libtiff on the host (TreeDetection/prediction.py:61 `rasterio.open`, :164
// `rasterio.mask.mask`); real orthophoto mosaics are stored LZW- or DEFLATE-compressed, and host threads decode a few hundred
// 450 x 450 x 4 windows per second where one MI355X consumes thousands. Here a raster's compressed strips / tiles cross PCIe
// once as they lie in the file, every block is decoded by ONE wave (an LZW stream is sequential; the parallelism is the
// 10^3 - 10^4 blocks of an image), and the decoded raster stays in HBM, where the tile windows are cut.
//
// Decoder (same stream format and the same accept / reject rules as the host decoder td_tiff_lzw_decode, tiffcodec.cpp:
// MSB-first 9..12-bit codes, ClearCode 256, EOI 257, width grows one code early). A string of the table is a WINDOW OF THE
// OUTPUT: entry k was defined when the code after string(old) arrived, so its bytes are out[pos(old) .. pos(old) + len(old)]
// — no prefix / suffix chains, and len(k) = start(k + 1) - start(k) + 1, so the table is ONE array of output positions in
// LDS. Two things make the stream parallel inside a block: between two ClearCodes a code's WIDTH depends only on its index
// (the table grows by one entry per code), so 64 lanes cut 64 codes out of the stream at once; and a string whose source
// lies before the chunk's own output depends on nothing the chunk writes, so the lanes copy those all at once. What stays
// sequential is a few per cent of the codes (strings that begin in bytes the same chunk writes). profiles/r06_decode.txt:
// 0.65 us per code for the code-by-code loop, 12 x less with the chunks (324 MB raster: 109.5 -> 9.1 ms).
#include "common.h"
#include "inflate_core.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <cstring>

namespace {

constexpr int LZW_CLEAR = 256, LZW_EOI = 257, LZW_FIRST = 258, LZW_MAX = 4096;

// Which block a decoder wave takes next: a counter per launch (one of kTickets slots, zeroed on the launch's stream). The waves of a
// workgroup share its LDS for the whole launch, so a raster of more blocks than the chip holds waves is NOT left to the dispatcher
// (a workgroup's slot is only refilled when its slowest wave is done: 20000 x 20000 px took 28 ms instead of 22 that way): one
// workgroup per CU at most, every wave fetching blocks until the counter passes the last one.
constexpr int kTickets = 64;
__device__ int g_block_ticket[kTickets];

__device__ __forceinline__ int take_block(int* ticket, int lane) {
    int v = 0;
    if (lane == 0) v = atomicAdd(ticket, 1);
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// status: 0 ok, 1 corrupt stream, 2 more bytes than the block holds, 3 left to the wide-table pass
//
// Where a string's bytes come from: every source lies in the output of the current table epoch (since the last ClearCode),
// and on imagery an epoch is a few KB (3 800 codes of 1 - 3 bytes) — so the wave keeps the last RING bytes of its output in
// LDS and copies from there: LDS operations of one wave execute in order, a copy can follow the write it depends on without
// any wait, and a code costs two dependent LDS reads (table, ring) instead of a round trip to L2. Sources older than the ring
// (long runs of flat pixels: strings of thousands of bytes) are read from the block's output in memory, behind a wait for
// this wave's stores. The code-by-code loop (the first code after a ClearCode, the ClearCode itself, the last table entry, EOI)
// is bound by the instruction latency of ONE wave walking a sequential stream (~150 dependent instructions per code). The
// table holds positions relative to the epoch's start as uint16 (8 KB; an epoch whose output outgrows 16 bits — flat areas,
// zero-padded edge tiles — flags its block for the second launch, the same kernel with a uint32 table), literals are strings
// of a 256-byte identity table so that every code of that loop takes the same copy path.
// LZW_RING bytes of recent output are kept in LDS: 16 KB (25 KB per wave: six waves per CU); a 4-KB variant (13 KB: twelve waves
// per CU) exists for measurements and to exercise the through-memory paths in the tests.
// this wave's LDS writes before its later LDS reads: the fences of __syncthreads() without the s_barrier (waves of one workgroup
// decode different blocks, each in its own loops)
#define TD_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                           __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
template <typename TableT, bool SECOND, int LZW_RING, bool FAST, int WPB>
__global__ __launch_bounds__(64 * WPB) void tiff_lzw_blocks_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ block_off,
                                                             const int64_t* __restrict__ block_nbytes, uint8_t* __restrict__ out,
                                                             int64_t block_cap, int64_t* __restrict__ decoded,
                                                             int32_t* __restrict__ status, int nblocks, int* __restrict__ wide_list,
                                                             int* __restrict__ ticket) {
    // WPB waves share a workgroup, each with its own tables and its own block (why: tiff_inflate_blocks_kernel below); nothing in
    // here meets a workgroup barrier — the waves run their own loops
    struct Lds {
        TableT t_start[LZW_MAX + 4];                       // start of entry k relative to the epoch's start; len(k) = t[k + 1] - t[k] + 1
        uint32_t inbuf[128];                               // two chunks of 64 big-endian dwords of the compressed stream
        uint8_t ring_lit[LZW_RING + 256];
    };
    __shared__ __attribute__((aligned(8))) Lds lds[WPB];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    TableT* const t_start = lds[wave].t_start;
    uint32_t* const inbuf = lds[wave].inbuf;
    uint8_t* const ring_lit = lds[wave].ring_lit;
    constexpr int LZW_LIT = LZW_RING;                      // the identity table follows the ring: ring_lit[LZW_LIT + c] = c
    const int lane = (int)threadIdx.x & 63;
    constexpr uint32_t REL_MAX = sizeof(TableT) == 2 ? 65535u - 4096u : 0xffffffffu;
    // first launch: blocks by ticket. Second launch (wide table): a few resident waves walk the list of blocks the first one
    // gave up on (wide_list[0] = their number) — nothing to do on imagery, so its cost must be that of an empty kernel
    for (int item = SECOND ? (int)blockIdx.x * WPB + wave : take_block(ticket, lane); item < (SECOND ? wide_list[0] : nblocks);
         item = SECOND ? item + (int)gridDim.x * WPB : take_block(ticket, lane)) {
    const int b = SECOND ? wide_list[1 + item] : item;
    const int64_t n = block_nbytes[b];
    uint8_t* dst = out + (int64_t)b * block_cap;
    const uint32_t cap = (uint32_t)block_cap;
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(comp + block_off[b]);
    const uint32_t skip = (uint32_t)(a0 & 3);
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(a0 - skip);
    const uint32_t ndw = (uint32_t)((n + skip + 3) >> 2); // dwords that hold the stream (the buffer is padded: reading the last one is safe)
    const uint32_t end_bit = (uint32_t)(n + skip) * 8u;    // (blocks are < 512 MB: bit positions fit 32 bits)
    TD_WAVE_SYNC();                                       // (second launch: the previous item's LDS reads are done)
    for (int c = lane; c < 256; c += 64) ring_lit[LZW_LIT + c] = (uint8_t)c;
    for (int c = lane; c < LZW_FIRST + 1; c += 64) t_start[c] = 0;      // literals: len = t[c + 1] - t[c] + 1 = 1
    // the first two chunks of the stream; afterwards chunk k + 1 is loaded when the reader enters chunk k
    inbuf[lane] = (uint32_t)lane < ndw ? bswap32(src32[lane]) : 0u;
    inbuf[64 + lane] = (uint32_t)(64 + lane) < ndw ? bswap32(src32[64 + lane]) : 0u;
    uint32_t loaded = 128;                                 // dwords [loaded - 128, loaded) are in inbuf
    uint32_t bitpos = skip * 8u;                           // next unread bit, counted from the aligned base, MSB first
    int nbits = 9, next = LZW_FIRST;
    uint32_t op = 0, old_pos = 0, old_len = 0, safe = 0, epoch = 0;      // old_len == 0: no previous code (start, or right after a ClearCode)
    int err = 0;
    uint32_t slow = 0;                                      // codes that went through memory instead of the ring (diagnostic: decoded[b] >> 32)
    TD_WAVE_SYNC();
    uint32_t hold = 0;                                     // codes left to the one-by-one path before the next chunk attempt
    for (;;) {
        if (FAST && hold == 0 && old_len != 0 && next < LZW_MAX - 1 && op - epoch <= REL_MAX) {
            // ---- 64 codes at once -------------------------------------------------------------------------------------------
            // Between two ClearCodes the WIDTH of a code depends only on how many codes came before it (the table grows by one
            // entry per code), so the 64 lanes cut the next 64 codes out of the stream at once; a string's length is the
            // distance of two table positions (or one more than the length of an earlier code of this chunk), its start a
            // prefix sum over the lanes — the table arithmetic of 64 codes costs what one code costs in the loop below. Then the
            // bytes, in three steps: (A) literals: every lane stores its own; (B) strings whose source ends before this chunk's
            // output begins (the table holds thousands of older entries: on imagery nearly all of them): every lane copies
            // its own string out of the ring, one byte per step; (C) what is left — strings that begin in bytes this chunk
            // writes, or beyond the ring — one after the other in code order, everything they need already in registers.
            // The chunk ends early at a ClearCode / EOI / the end of the input / a code the table cannot hold yet / the last
            // table entry; those codes take the loop below, one by one.
            while (((bitpos + 64u * 12u) >> 5) + 1u >= loaded) {
                const uint32_t idx = loaded + lane;
                inbuf[idx & 127] = idx < ndw ? bswap32(src32[idx]) : 0u;
                loaded += 64;
                TD_WAVE_SYNC();
                safe = op;
            }
            const uint32_t c0 = (uint32_t)next - 257u;     // this chunk's first code is the c0-th since the ClearCode (c0 >= 1)
            const uint32_t idx = c0 + lane;
            auto bits_before = [](uint32_t i) {            // bits of the codes 0 .. i - 1 of a table epoch: 9 each, one more from the 254th, 766th, 1790th on
                return 9u * i + (i > 254u ? i - 254u : 0u) + (i > 766u ? i - 766u : 0u) + (i > 1790u ? i - 1790u : 0u);
            };
            const uint32_t my_bits = 9u + (idx >= 254u) + (idx >= 766u) + (idx >= 1790u);
            const uint32_t off = bitpos + bits_before(idx) - bits_before(c0);
            const uint32_t dwl = off >> 5;
            const uint64_t twol = ((uint64_t)inbuf[dwl & 127] << 32) | inbuf[(dwl + 1) & 127];
            const uint32_t code = (uint32_t)((twol >> (64 - (off & 31) - my_bits)) & ((1u << my_bits) - 1u));
            const bool stop = (code - (uint32_t)LZW_CLEAR) < 2u || off + my_bits > end_bit || code > 257u + idx || 257u + idx >= (uint32_t)LZW_MAX - 1u;
            const uint64_t stops = __ballot(stop);
            uint32_t n = stops ? (uint32_t)__builtin_ctzll(stops) : 64u;
            const bool lit = code < 256u;
            const uint32_t e = code - 258u;                // the table entry: string(e-th code) + first byte of the (e + 1)-th
            const bool inchunk = !lit && e >= c0 && (uint32_t)lane < n;
            uint32_t L = 0, src = 0;
            if (lane == 0) t_start[LZW_FIRST + c0] = (TableT)(op - epoch);       // where the chunk's first string will begin: the end of entry c0 - 1
            if ((uint32_t)lane < n && !inchunk) {
                const uint32_t t0 = t_start[code], t1 = t_start[code + 1];      // (literals: both 0)
                L = t1 - t0 + 1u;
                src = epoch + t0;
            }
            // lengths that hang on a code of this chunk: L = L(that code) + 1. Every round settles at least the lowest open lane
            // (it points at a lower one); runs of one value take many rounds, imagery one or two. (Lane reads stay outside
            // branches: inside one they return 0 for the lanes that did not take it.)
            while (__ballot(inchunk && L == 0)) {
                const uint32_t Lm = (uint32_t)__shfl((int)L, (int)((e - c0) & 63u));
                if (inchunk && L == 0 && Lm != 0) L = Lm + 1u;
            }
            uint32_t incl = (uint32_t)lane < n ? L : 0u;   // inclusive prefix sum over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += up;
            }
            const uint32_t pos = op + incl - L;            // where this lane's string begins
            const uint32_t pos_m = (uint32_t)__shfl((int)pos, (int)((e - c0) & 63u));
            if (inchunk) src = pos_m;
            // the chunk's output stays well inside the ring (step B reads old bytes while younger lanes write ahead), and inside the table's range
            const uint64_t hards = __ballot((uint32_t)lane < n && (incl > (uint32_t)LZW_RING / 2u || pos + L - epoch > REL_MAX));
            if (hards) n = min(n, (uint32_t)__builtin_ctzll(hards));
            if (n == 0) {
                hold = 1;
            } else {
                const uint32_t last_pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)(n - 1));        // (lane reads: the wave's state stays in scalar registers)
                const uint32_t last_len = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)(n - 1));
                const uint32_t op_end = last_pos + last_len;
                const bool mine = (uint32_t)lane < n;
                if (mine) t_start[LZW_FIRST + idx] = (TableT)(pos - epoch);
                // (A) literals
                if (mine && lit) {
                    ring_lit[pos & (LZW_RING - 1)] = (uint8_t)code;
                    if (pos < cap) dst[pos] = (uint8_t)code;
                }
                // (B) sources that end before this chunk's first byte and are still in the ring when its last byte is written
                const bool early = mine && !lit && src + L <= op && op_end - src <= (uint32_t)LZW_RING;
                for (uint32_t t = 0;; ++t) {
                    const bool act = early && t < L;
                    if (!__ballot(act)) break;
                    if (act) {
                        const uint8_t v = ring_lit[(src + t) & (LZW_RING - 1)];
                        ring_lit[(pos + t) & (LZW_RING - 1)] = v;
                        if (pos + t < cap) dst[pos + t] = v;
                    }
                }
                // (C) the rest, in code order
                uint64_t late = __ballot(mine && !lit && !early);
                while (late) {
                    const int k = __builtin_ctzll(late);
                    late &= late - 1;
                    const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)src, k), lk = (uint32_t)__builtin_amdgcn_readlane((int)L, k);
                    const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pos, k);
                    if (lk <= 64u && op_end - sk <= (uint32_t)LZW_RING) {       // (steps A and B have already written up to op_end: the source must have survived that)
                        if ((uint32_t)lane < lk) {
                            // (a string that runs into its own first byte — the code names the entry being defined — repeats that byte)
                            const uint32_t from = (sk + lane == pk) ? sk : sk + lane;
                            const uint8_t v = ring_lit[from & (LZW_RING - 1)];
                            ring_lit[(pk + lane) & (LZW_RING - 1)] = v;
                            if (pk + lane < cap) dst[pk + lane] = v;
                        }
                    } else {                               // long strings and sources that have left the ring: through the block's output in memory
                        ++slow;
                        if (sk + lk > safe) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            safe = pk;                     // every byte before this string has been stored (steps A, B and the earlier strings of C)
                        }
                        for (uint32_t k0 = 0; k0 < lk; k0 += 64) {
                            const uint32_t kk = k0 + lane;
                            if (kk < lk) {
                                const uint32_t from = (sk + kk == pk) ? sk : sk + kk;
                                // (a source past the block's capacity exists only in a stream that decodes to too many bytes: status 2 either way)
                                const uint8_t v = from < cap ? __hip_atomic_load(dst + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
                                ring_lit[(pk + kk) & (LZW_RING - 1)] = v;
                                if (pk + kk < cap) dst[pk + kk] = v;
                            }
                        }
                    }
                }
                bitpos += bits_before(c0 + n) - bits_before(c0);
                next += (int)n;
                nbits = 9 + (next >= 511) + (next >= 1023) + (next >= 2047);
                old_pos = last_pos;
                old_len = last_len;
                op = op_end;
                TD_WAVE_SYNC();
                continue;
            }
        }
        if (hold) --hold;
        if (bitpos + (uint32_t)nbits > end_bit) break;     // ran out of input without an EOI: accept what was decoded (host decoder's rule)
        const uint32_t dw = bitpos >> 5;
        if (__builtin_expect(dw + 1 >= loaded, 0)) {       // the reader enters the last loaded chunk: bring in the next one over the older
            const uint32_t idx = loaded + lane;
            inbuf[idx & 127] = idx < ndw ? bswap32(src32[idx]) : 0u;
            loaded += 64;
            TD_WAVE_SYNC();                               // (one wave: orders the LDS write before the reads below; every store of this wave has completed too)
            safe = op;
        }
        const uint64_t two = ((uint64_t)inbuf[dw & 127] << 32) | inbuf[(dw + 1) & 127];
        const int code = (int)((two >> (64 - (bitpos & 31) - nbits)) & ((1u << nbits) - 1u));
        bitpos += nbits;
        // the rare cases first, behind ONE test: EOI / ClearCode, the literal after a clear, a code beyond the table
        if (__builtin_expect((unsigned)(code - LZW_CLEAR) < 2u || old_len == 0 || code > next || (code == next && next >= LZW_MAX), 0)) {
            if (code == LZW_EOI) break;
            if (code == LZW_CLEAR) {
                nbits = 9;
                next = LZW_FIRST;
                old_len = 0;
                continue;
            }
            if (old_len != 0 || code > 255) {              // beyond the table, or not a literal right after a clear
                err = 1;
                break;
            }
            if (lane == 0) {
                ring_lit[op & (LZW_RING - 1)] = (uint8_t)code;
                if (op < cap) dst[op] = (uint8_t)code;
            }
            epoch = op;                                    // table positions are relative to the first byte of the epoch
            old_pos = op;
            old_len = 1;
            op += 1;
            continue;
        }
        // string(code) = out[s_start .. s_start + s_len): a window of the output — for a literal a window of the identity table
        const bool kwkwk = code == next;
        uint32_t t0, t1;
        if (sizeof(TableT) == 2) {                         // t[code] and t[code + 1] in ONE LDS round trip: the two dwords around them
            const uint32_t* t32 = reinterpret_cast<const uint32_t*>(t_start);
            const uint64_t both = ((uint64_t)t32[(code >> 1) + 1] << 32) | t32[code >> 1];
            const uint32_t pair = (uint32_t)(both >> ((code & 1) * 16));
            t0 = pair & 0xffffu;
            t1 = pair >> 16;
        } else {
            t0 = t_start[code];
            t1 = t_start[code + 1];
        }
        const uint32_t s_start = kwkwk ? old_pos : epoch + t0;
        const uint32_t s_len = kwkwk ? old_len + 1 : t1 - t0 + 1;     // (a literal's entries are all 0: length 1)
        if (next < LZW_MAX) {                              // new entry = string(old) + first byte of string(code): out[old_pos .. op]
            if (op - epoch > REL_MAX) {                    // (uint16 table only) this epoch's output no longer fits: the wide pass decodes the block
                err = 3;
                break;
            }
            if (lane == 0) {
                t_start[next] = (TableT)(old_pos - epoch);
                t_start[next + 1] = (TableT)(op - epoch);  // provisional: becomes the next entry's start (its old_pos is this op)
            }
            ++next;
            nbits += (next > (1 << nbits) - 2 && nbits < 12) ? 1 : 0;
        }
        if (__builtin_expect(s_len <= 64 && (code < 256 || op - s_start + 64 <= (uint32_t)LZW_RING), 1)) {
            // one step: read (ring or identity table), then write ring + memory; LDS operations of a wave execute in order
            if ((uint32_t)lane < s_len) {
                const uint32_t sk = (kwkwk && (uint32_t)lane == s_len - 1) ? 0u : (uint32_t)lane;
                const uint32_t a = code < 256 ? (uint32_t)(LZW_LIT + code) : ((s_start + sk) & (LZW_RING - 1));
                const uint8_t v = ring_lit[a];
                ring_lit[(op + lane) & (LZW_RING - 1)] = v;
                if (op + lane < cap) dst[op + lane] = v;
            }
        } else {
            // long strings and sources that have left the ring: through the block's output in memory
            ++slow;
            if (s_start + s_len > safe) {                  // the source may still be on its way to L2: wait for this wave's stores
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                safe = op;
            }
            for (uint32_t k0 = 0; k0 < s_len; k0 += 64) {
                const uint32_t k = k0 + lane;
                if (k < s_len) {
                    const uint32_t sk = (kwkwk && k == s_len - 1) ? 0u : k;
                    // agent-scope load: served by L2, never by a stale L1 line of bytes this wave stored earlier
                    const uint8_t v = s_start + sk < cap ? __hip_atomic_load(dst + s_start + sk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
                    ring_lit[(op + k) & (LZW_RING - 1)] = v;
                    if (op + k < cap) dst[op + k] = v;
                }
            }
        }
        old_pos = op;
        old_len = s_len;
        op += s_len;
    }
    if (lane == 0) {
        decoded[b] = (int64_t)op | ((int64_t)slow << 32);
        status[b] = err == 3 && !SECOND ? 3 : (err ? 1 : (op > cap ? 2 : 0));
        if (!SECOND && err == 3) wide_list[1 + atomicAdd(wide_list, 1)] = b;
    }
    }
}

// DEFLATE blocks (TIFF compression 8 / 32946: zlib streams), one wave per block: inflate_core.h. LDS: a ring of the output +
// tables — the whole 32-KB window (37 KB per wave) while the blocks fit the chip in one round, 8 KB (12 KB per wave) beyond; literals
// and matches go to memory as byte stores. WPB waves (= blocks of the raster) share a workgroup, so that the long-lived decoder
// waves sit TOGETHER on few CUs (twelve per CU with the small ring) instead of a few on every CU: a decoder wave holds 104
// vector registers and 12 KB of LDS for the whole 60 ms, and one such wave per SIMD is enough to keep the model's large tiles
// (256 registers x 2 waves per SIMD, > 100 KB of LDS) off that CU — spread one per workgroup they stalled the forward of the
// image that predicts meanwhile on every CU of the chip.
template <int RING, int WPB>
__global__ __launch_bounds__(64 * WPB) void tiff_inflate_blocks_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ block_off,
                                                                       const int64_t* __restrict__ block_nbytes, uint8_t* __restrict__ out,
                                                                       int64_t block_cap, int64_t* __restrict__ decoded,
                                                                       int32_t* __restrict__ status, int nblocks, int* __restrict__ ticket) {
    __shared__ InflateScratchT<RING> S[WPB];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;      // the wave's number as a scalar: its LDS base stays one
    for (int b = take_block(ticket, lane); b < nblocks; b = take_block(ticket, lane)) {      // whole waves leave; nothing in here meets a workgroup barrier
        const InflateResult r = inflate_block<64>(S[wave], comp + block_off[b], block_nbytes[b], out + (int64_t)b * block_cap, (uint32_t)block_cap, lane);
        if (lane == 0) {
            decoded[b] = (int64_t)r.produced;
            status[b] = r.status;
        }
        TD_INF_SYNC();                                     // the next block's set-up writes follow this block's last LDS reads
    }
}

// Decoded blocks → the raster [height][width][spp] (uint8, pixel-interleaved), undoing predictor 2 (TIFF 6.0 section 14:
// each sample is the difference to the same sample of the pixel on its left, within the row of its block) on the way. One
// workgroup per (image row, block column): a block row of bw pixels is a prefix sum per sample — per-thread runs, a scan of
// the 256 run totals in LDS, then the stores. HBM-bound: one read and one write of the raster.
constexpr int SC_THREADS = 256, SC_MAX_SPP = 4;
template <int spp>
__global__ __launch_bounds__(SC_THREADS) void tiff_blocks_to_image_kernel(const uint8_t* __restrict__ blocks, int64_t block_cap, int bw, int bh,
                                                                          int blocks_across, int predictor,
                                                                          uint8_t* __restrict__ image, int width, int height) {
    __shared__ uint8_t tot[SC_THREADS][SC_MAX_SPP];
    const int y = blockIdx.x, bx = blockIdx.y, tid = threadIdx.x;
    const int by = y / bh;
    const uint8_t* src = blocks + ((int64_t)by * blocks_across + bx) * block_cap + (int64_t)(y - by * bh) * bw * spp;
    const int x0 = bx * bw;
    const int valid = min(bw, width - x0);                 // pixels of this block row that lie inside the raster
    uint8_t* dst = image + ((int64_t)y * width + x0) * spp;
    const int per = (bw + SC_THREADS - 1) / SC_THREADS;    // pixels per thread (contiguous run)
    const int p0 = tid * per, p1 = min(p0 + per, bw);
    if (predictor != 2) {
        for (int p = p0; p < min(p1, valid); ++p)
            for (int c = 0; c < spp; ++c) dst[p * spp + c] = src[p * spp + c];
        return;
    }
    uint8_t run[spp];
#pragma unroll
    for (int c = 0; c < spp; ++c) run[c] = 0;
    for (int p = p0; p < p1; ++p)
        for (int c = 0; c < spp; ++c) run[c] = (uint8_t)(run[c] + src[p * spp + c]);
    for (int c = 0; c < spp; ++c) tot[tid][c] = run[c];
    __syncthreads();
    // inclusive scan of the run totals (Hillis-Steele over 256 entries, all samples at once)
    for (int off = 1; off < SC_THREADS; off <<= 1) {
        uint8_t add[spp];
        for (int c = 0; c < spp; ++c) add[c] = tid >= off ? tot[tid - off][c] : 0;
        __syncthreads();
        for (int c = 0; c < spp; ++c) tot[tid][c] = (uint8_t)(tot[tid][c] + add[c]);
        __syncthreads();
    }
    uint8_t base[spp];
    for (int c = 0; c < spp; ++c) base[c] = tid > 0 ? tot[tid - 1][c] : 0;
    for (int p = p0; p < p1; ++p)
        for (int c = 0; c < spp; ++c) {
            base[c] = (uint8_t)(base[c] + src[p * spp + c]);
            if (p < valid) dst[p * spp + c] = base[c];
        }
}

// The same for four samples per pixel (RGBI, the reference's rasters): one WAVE per block row, a pixel is one dword, the four
// byte-wise running sums ride in it (carry-less packed add), 64 pixels per step: a scan across the lanes by lane reads, no LDS,
// no barrier. 4 rows per workgroup.
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b) {
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__global__ __launch_bounds__(256) void tiff_blocks_to_image_rgbi_kernel(const uint8_t* __restrict__ blocks, int64_t block_cap, int bw, int bh,
                                                                        int blocks_across, int predictor, uint8_t* __restrict__ image, int width,
                                                                        int height) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), bx = blockIdx.y;
    if (y >= height) return;
    const int by = y / bh;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(blocks + ((int64_t)by * blocks_across + bx) * block_cap) + (int64_t)(y - by * bh) * bw;
    const int x0 = bx * bw;
    const int valid = min(bw, width - x0);
    uint32_t* dst = reinterpret_cast<uint32_t*>(image) + (int64_t)y * width + x0;
    uint32_t carry = 0;
    for (int p0 = 0; p0 < valid; p0 += 64) {
        const int p = p0 + lane;
        uint32_t v = p < bw ? src[p] : 0u;
        if (predictor == 2) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)v, d);
                if (lane >= d) v = add_bytes(v, up);
            }
            v = add_bytes(v, carry);
            carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        }
        if (p < valid) dst[p] = v;
    }
}

struct Tickets {
    std::once_flag once;
    int* base = nullptr;
    int cus = 0;
    hipError_t err = hipSuccess;
};
Tickets g_tickets[16];
std::atomic<unsigned> g_ticket_next{0};

// → a zeroed block counter for one launch on `s` (a slot is reused 64 launches later: long after its launch has ended) and the
// number of CUs (= the most workgroups worth launching: one fills a CU's LDS)
td_status next_ticket(hipStream_t s, int** ticket, int* cus) {
    int dev = 0;
    TD_HIP_CHECK(hipGetDevice(&dev));
    TD_REQUIRE(dev >= 0 && dev < 16, "raster decode: device %d", dev);
    Tickets& t = g_tickets[dev];
    std::call_once(t.once, [&] {
        t.err = hipGetSymbolAddress(reinterpret_cast<void**>(&t.base), HIP_SYMBOL(g_block_ticket));
        if (t.err == hipSuccess) t.err = hipDeviceGetAttribute(&t.cus, hipDeviceAttributeMultiprocessorCount, dev);
    });
    TD_HIP_CHECK(t.err);
    *ticket = t.base + g_ticket_next.fetch_add(1, std::memory_order_relaxed) % kTickets;
    *cus = t.cus > 0 ? t.cus : 256;
    TD_HIP_CHECK(hipMemsetAsync(*ticket, 0, sizeof(int), s));
    return TD_OK;
}

// Which ring. DEFLATE: the small one as soon as the blocks no longer fit the chip in one round with the large one (256 CUs x
// four waves). TD_DECODE_RING = small | large overrides (tests, measurements: tools/raster_decode_bench.py).
int ring_override() {
    const char* e = getenv("TD_DECODE_RING");
    return !e ? -1 : (!strcmp(e, "small") ? 1 : (!strcmp(e, "large") ? 0 : -1));
}
int lzw_ring_choice(int) {              // LZW: the large ring always — with 64 codes per step a block is short, and every string the
    const int o = ring_override();      // small ring has lost costs a round trip to L2 (20000 x 20000 px: 26.7 ms against 38.4)
    return o >= 0 ? o : 0;
}
int inflate_ring_choice(int nblocks) {
    const int o = ring_override();
    return o >= 0 ? o : nblocks > 256 * 4;
}

}  // namespace

extern "C" td_status td_tiff_lzw_decode_dev(const uint8_t* comp, const int64_t* block_off, const int64_t* block_nbytes, int nblocks,
                                            uint8_t* blocks_out, int64_t block_cap, int64_t* decoded, int32_t* status, void* stream) {
    TD_REQUIRE(comp && block_off && block_nbytes && blocks_out && decoded && status, "td_tiff_lzw_decode_dev: null pointer");
    TD_REQUIRE(nblocks >= 0 && block_cap >= 1 && block_cap < ((int64_t)1 << 31), "td_tiff_lzw_decode_dev: %d blocks of %lld bytes", nblocks,
               (long long)block_cap);
    if (nblocks == 0) return TD_OK;
    // narrow table first (six waves per CU); blocks whose epochs outgrow it (flat rasters) are listed in wide_list and decoded again
    // by a few resident waves with the wide table — an empty loop on imagery
    hipStream_t s = static_cast<hipStream_t>(stream);
    int* wide_list = reinterpret_cast<int*>(status) + nblocks;      // status has room for 2 * nblocks + 1 ints (include/treedet.h)
    TD_HIP_CHECK(hipMemsetAsync(wide_list, 0, sizeof(int), s));
    const int small_ring = lzw_ring_choice(nblocks);
    const char* one = getenv("TD_LZW_ONE_BY_ONE");          // measurements: the code-by-code loop alone (tools/raster_decode_bench.py)
    const bool chunks = !(one && one[0] == '1');
    // waves per workgroup: as many as 160 KB of LDS hold (13 / 25 / 33 KB per wave); one workgroup per CU at most, blocks by ticket
    int* ticket = nullptr;
    int cus = 0;
    const td_status tst = next_ticket(s, &ticket, &cus);
    if (tst < 0) return tst;
#define TD_LZW(RING, FAST, WPB) hipLaunchKernelGGL((tiff_lzw_blocks_kernel<uint16_t, false, RING, FAST, WPB>), \
                                                   dim3((nblocks + WPB - 1) / WPB < cus ? (nblocks + WPB - 1) / WPB : cus), dim3(64 * WPB), 0, s, comp, \
                                                   block_off, block_nbytes, blocks_out, block_cap, decoded, status, nblocks, wide_list, ticket)
    if (small_ring && chunks) TD_LZW(4096, true, 12);
    else if (small_ring) TD_LZW(4096, false, 12);
    else if (chunks) TD_LZW(16384, true, 6);
    else TD_LZW(16384, false, 6);
#undef TD_LZW
    TD_KERNEL_CHECK();
    const int second_groups = (nblocks + 3) / 4 < 256 ? (nblocks + 3) / 4 : 256;
    hipLaunchKernelGGL((tiff_lzw_blocks_kernel<uint32_t, true, 16384, true, 4>), dim3(second_groups), dim3(64 * 4), 0, s, comp, block_off,
                       block_nbytes, blocks_out, block_cap, decoded, status, nblocks, wide_list, ticket);
    TD_KERNEL_CHECK();
    return TD_OK;
}

extern "C" td_status td_tiff_inflate_dev(const uint8_t* comp, const int64_t* block_off, const int64_t* block_nbytes, int nblocks,
                                         uint8_t* blocks_out, int64_t block_cap, int64_t* decoded, int32_t* status, void* stream) {
    TD_REQUIRE(comp && block_off && block_nbytes && blocks_out && decoded && status, "td_tiff_inflate_dev: null pointer");
    TD_REQUIRE(nblocks >= 0 && block_cap >= 1 && block_cap < ((int64_t)1 << 31), "td_tiff_inflate_dev: %d blocks of %lld bytes", nblocks,
               (long long)block_cap);
    if (nblocks == 0) return TD_OK;
    constexpr int WPB_SMALL = 12, WPB_WINDOW = 4;          // 148 KB of LDS per workgroup either way
    static const char* wpb_env = getenv("TD_INFLATE_WPB");  // measurements: 1 = a workgroup per block, 8 (tools/probes/decode_overlap_probe.py)
    const int wpb = wpb_env ? atoi(wpb_env) : 0;
    int* ticket = nullptr;
    int cus = 0;
    const td_status tst = next_ticket(static_cast<hipStream_t>(stream), &ticket, &cus);
    if (tst < 0) return tst;
    // one workgroup per CU at most (WPB = 1: as many single waves as CUs hold: twelve each)
#define TD_INFLATE(RING, WPB) hipLaunchKernelGGL((tiff_inflate_blocks_kernel<RING, WPB>), \
                                                 dim3((nblocks + WPB - 1) / WPB < cus * (WPB == 1 ? 12 : 1) ? (nblocks + WPB - 1) / WPB : cus * (WPB == 1 ? 12 : 1)), \
                                                 dim3(64 * WPB), 0, static_cast<hipStream_t>(stream), comp, block_off, block_nbytes, blocks_out, block_cap, \
                                                 decoded, status, nblocks, ticket)
    if (inflate_ring_choice(nblocks)) {
        if (wpb == 1) TD_INFLATE(8192, 1);
        else if (wpb == 8) TD_INFLATE(8192, 8);
        else TD_INFLATE(8192, WPB_SMALL);
    } else {
        if (wpb == 1) TD_INFLATE(INF_WINDOW, 1);
        else TD_INFLATE(INF_WINDOW, WPB_WINDOW);
    }
#undef TD_INFLATE
    TD_KERNEL_CHECK();
    return TD_OK;
}

extern "C" td_status td_tiff_blocks_to_image_dev(const uint8_t* blocks, int64_t block_cap, int block_w, int block_h, int blocks_across,
                                                 int blocks_down, int spp, int predictor, uint8_t* image, int width, int height,
                                                 void* stream) {
    TD_REQUIRE(blocks && image, "td_tiff_blocks_to_image_dev: null pointer");
    TD_REQUIRE(block_w >= 1 && block_h >= 1 && blocks_across >= 1 && blocks_down >= 1 && width >= 1 && height >= 1,
               "td_tiff_blocks_to_image_dev: bad geometry");
    TD_REQUIRE(spp >= 1 && spp <= SC_MAX_SPP && (predictor == 1 || predictor == 2), "td_tiff_blocks_to_image_dev: %d samples per pixel, predictor %d",
               spp, predictor);
    TD_REQUIRE((int64_t)block_w * block_h * spp <= block_cap, "td_tiff_blocks_to_image_dev: a %d x %d x %d block does not fit %lld bytes", block_w,
               block_h, spp, (long long)block_cap);
    TD_REQUIRE((int64_t)blocks_across * block_w >= width && (int64_t)blocks_down * block_h >= height && (int64_t)(blocks_across - 1) * block_w < width &&
               (int64_t)(blocks_down - 1) * block_h < height, "td_tiff_blocks_to_image_dev: %d x %d blocks of %d x %d do not tile a %d x %d raster",
               blocks_across, blocks_down, block_w, block_h, width, height);
    TD_REQUIRE(blocks_across <= 65535, "td_tiff_blocks_to_image_dev: too many block columns");
    const dim3 grid(height, blocks_across);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (spp == 4 && block_cap % 4 == 0 && reinterpret_cast<uintptr_t>(blocks) % 4 == 0 && reinterpret_cast<uintptr_t>(image) % 4 == 0) {
        hipLaunchKernelGGL(tiff_blocks_to_image_rgbi_kernel, dim3((height + 3) / 4, blocks_across), dim3(256), 0, s, blocks, block_cap, block_w, block_h,
                           blocks_across, predictor, image, width, height);
        TD_KERNEL_CHECK();
        return TD_OK;
    }
#define TD_SCATTER(N) hipLaunchKernelGGL(tiff_blocks_to_image_kernel<N>, grid, dim3(SC_THREADS), 0, s, blocks, block_cap, block_w, block_h, \
                                         blocks_across, predictor, image, width, height)
    switch (spp) {
        case 1: TD_SCATTER(1); break;
        case 2: TD_SCATTER(2); break;
        case 3: TD_SCATTER(3); break;
        default: TD_SCATTER(4); break;
    }
#undef TD_SCATTER
    TD_KERNEL_CHECK();
    return TD_OK;
}
