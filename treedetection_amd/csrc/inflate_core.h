// DEFLATE (RFC 1951) inside a zlib wrapper (RFC 1950) — the stream of TIFF compression 8 / 32946 strips and tiles — decoded by
// ONE 64-lane wave per block on the GPU (tiffdecode.hip) and, from the same source, by one host thread (tiffcodec.cpp:
// td_tiff_inflate, NL = 1): the reference reads such rasters through rasterio → GDAL → zlib on the host, one tile window at
// a time (TreeDetection/prediction.py:61,164). Symbols are decoded by every lane alike (a DEFLATE stream is sequential);
// matches are copied 64 bytes per step out of a ring of the last RING bytes of the output in LDS (RING = 32 KB = DEFLATE's
// whole window: no copy ever reads memory; RING = 8 KB: 13 KB per wave, twelve waves per CU instead of four, and the rare match
// that reaches further back reads the block's output in memory behind a wait for this wave's stores — on imagery matches
// point at the neighbouring pixels or the row above); the Huffman tables of a block are built by lane 0 (a few thousand
// operations per 16 - 64 KB of output).
// Tables: a 10-bit lookup for literal / length codes and an 8-bit one for distance codes (entry = length << 9 | symbol, 0 =
// longer code), canonical count / symbol arrays for the codes that are longer (decoded bit by bit, as zlib's `puff` does).
#pragma once
#include <cstdint>

#ifdef __HIP_DEVICE_COMPILE__
// orders this WAVE's LDS writes before its later LDS reads (the fences of __syncthreads without its s_barrier): a block is decoded by
// one wave, and several such waves share a workgroup (tiffdecode.hip), each inside its own loops — a workgroup barrier would hang
#define TD_INF_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
// this wave's stores have reached L2; an agent-scope load is served there, never by a stale L1 line
#define TD_INF_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TD_INF_LOAD_OUT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// Values read from LDS at a wave-uniform address ARE uniform; telling the compiler so keeps the decoder's whole state (bit
// reservoir, positions, table entries) in scalar registers and its control flow in scalar branches — left alone it carried the
// state in vector registers, ran the loops under exec masks and fetched the length / distance base tables with per-lane loads.
#define TD_INF_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define TD_INF_SYNC() ((void)0)
#define TD_INF_STORES_DONE() ((void)0)
#define TD_INF_LOAD_OUT(p) (*(p))
#define TD_INF_UNIFORM(x) ((uint32_t)(x))
#endif
#ifdef __HIPCC__
#define TD_INF_HD __host__ __device__ inline
// the decoder's loop belongs INSIDE each kernel that runs it (several instantiate it: called out of line it cost 13 % and 70 registers)
#define TD_INF_INLINE __attribute__((always_inline))
#else
#define TD_INF_HD static inline
#define TD_INF_INLINE
#endif

constexpr int INF_WINDOW = 32768;        // DEFLATE's window
constexpr int INF_LIT_FAST = 10, INF_DIST_FAST = 8;

template <int RING>
struct InflateScratchT {                 // LDS on the device (RING + 4.6 KB), a plain struct on the host
    static_assert(RING >= 1024 && (RING & (RING - 1)) == 0 && RING <= INF_WINDOW, "ring size");
    uint8_t ring[RING];
    uint32_t inbuf[128];                 // two chunks of 64 little-endian dwords of the stream
    uint16_t lit_fast[1 << INF_LIT_FAST];
    uint16_t dist_fast[1 << INF_DIST_FAST];
    uint16_t lit_sym[288], dist_sym[32]; // symbols in canonical order
    uint16_t lit_count[16], dist_count[16];
    uint8_t lens[384];                   // code lengths of the block being set up (19 of the code-length code, then 32 .. 348: literal / length + distance)
};
using InflateScratch = InflateScratchT<INF_WINDOW>;

// status: 0 ok, 1 corrupt stream, 2 more bytes than the block holds
struct InflateResult {
    uint32_t produced;
    int status;
};

namespace inflate_detail {

struct Reader {
    const uint32_t* src32;               // aligned base of the stream
    uint32_t ndw, loaded, rd;            // dwords in the stream / loaded into inbuf so far / consumed into acc
    uint64_t acc;
    int have;
    uint32_t end_bit;                    // first bit past the stream (from the aligned base)
};

template <int NL, typename Scratch>
TD_INF_HD void load_chunk(Scratch& S, Reader& r, int lane) {
    for (int k = lane; k < 64; k += NL) {
        const uint32_t idx = r.loaded + k;
        S.inbuf[idx & 127] = idx < r.ndw ? r.src32[idx] : 0u;
    }
    r.loaded += 64;
    TD_INF_SYNC();
}

// make at least n <= 32 bits available in acc (zeros past the end of the stream: the callers check `overrun`)
template <int NL, typename Scratch>
TD_INF_HD void ensure(Scratch& S, Reader& r, int n, int lane) {
    if (r.have < n) {
        if (r.rd >= r.loaded) load_chunk<NL>(S, r, lane);
        r.acc |= (uint64_t)TD_INF_UNIFORM(S.inbuf[r.rd & 127]) << r.have;
        ++r.rd;
        r.have += 32;
    }
}
TD_INF_HD uint32_t take(Reader& r, int n) {
    const uint32_t v = (uint32_t)(r.acc & ((1ull << n) - 1ull));
    r.acc >>= n;
    r.have -= n;
    return v;
}
TD_INF_HD uint32_t bitpos(const Reader& r) { return r.rd * 32u - (uint32_t)r.have; }

// restart the reader at an absolute bit position (after a stored block)
template <int NL, typename Scratch>
TD_INF_HD void seek(Scratch& S, Reader& r, uint32_t bit, int lane) {
    TD_INF_SYNC();
    r.rd = bit >> 5;
    r.loaded = r.rd & ~63u;
    load_chunk<NL>(S, r, lane);
    load_chunk<NL>(S, r, lane);
    r.acc = 0;
    r.have = 0;
    ensure<NL>(S, r, 1, lane);
    take(r, (int)(bit & 31));
}

TD_INF_HD uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t o = 0;
    for (int i = 0; i < n; ++i) {
        o = (o << 1) | (v & 1u);
        v >>= 1;
    }
    return o;
}

// canonical Huffman tables from code lengths lens[0..n): counts per length, symbols in canonical order, the fast lookup.
// Returns false for an over-subscribed set (an incomplete one is allowed only for a single distance code, as zlib allows).
TD_INF_HD bool build(const uint8_t* lens, int n, uint16_t* count, uint16_t* sym, uint16_t* fast, int fast_bits) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (lens[s]) sym[offs[lens[s]]++] = (uint16_t)s;
    for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
    uint32_t code = 0;
    int index = 0;
    for (int l = 1; l <= fast_bits; ++l) {
        for (int k = 0; k < count[l]; ++k) {
            const uint32_t rev = reverse_bits(code, l);
            const uint16_t e = (uint16_t)((l << 9) | sym[index]);
            for (uint32_t f = rev; f < (1u << fast_bits); f += 1u << l) fast[f] = e;
            ++code;
            ++index;
        }
        code <<= 1;
    }
    return true;
}

// one symbol: the fast table, else bit by bit over the canonical arrays (codes longer than the table's index)
template <int NL, typename Scratch>
TD_INF_HD int decode_sym(Scratch& S, Reader& r, const uint16_t* fast, int fast_bits, const uint16_t* count, const uint16_t* sym, int lane) {
    ensure<NL>(S, r, 15, lane);
    const uint32_t e = TD_INF_UNIFORM(fast[r.acc & ((1u << fast_bits) - 1u)]);
    if (e) {
        take(r, e >> 9);
        return e & 511;
    }
    int code = 0, first = 0, index = 0;
    uint64_t bits = r.acc;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int c = (int)TD_INF_UNIFORM(count[l]);
        if (code - c < first) {
            take(r, l);
            return (int)TD_INF_UNIFORM(sym[index + (code - first)]);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

}  // namespace inflate_detail

// src: the zlib stream (n bytes); dst: the block's output (cap bytes); every lane of the wave calls this with its lane id
// (host: NL = 1, lane = 0). The result is the same on every lane.
template <int NL, int RING>
TD_INF_HD TD_INF_INLINE InflateResult inflate_block(InflateScratchT<RING>& S, const uint8_t* src, int64_t n, uint8_t* dst, uint32_t cap, int lane) {
    using namespace inflate_detail;
    // length / distance codes: base value | extra bits << 16 (dwords: a scalar load on the device; byte tables would be per-lane loads)
    constexpr uint32_t LEN_CODE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11 | 1 << 16, 13 | 1 << 16, 15 | 1 << 16, 17 | 1 << 16, 19 | 2 << 16, 23 | 2 << 16, 27 | 2 << 16,
                                       31 | 2 << 16, 35 | 3 << 16, 43 | 3 << 16, 51 | 3 << 16, 59 | 3 << 16, 67 | 4 << 16, 83 | 4 << 16, 99 | 4 << 16,
                                       115 | 4 << 16, 131 | 5 << 16, 163 | 5 << 16, 195 | 5 << 16, 227 | 5 << 16, 258};
    constexpr uint32_t DIST_CODE[30] = {1, 2, 3, 4, 5 | 1 << 16, 7 | 1 << 16, 9 | 2 << 16, 13 | 2 << 16, 17 | 3 << 16, 25 | 3 << 16, 33 | 4 << 16, 49 | 4 << 16,
                                        65 | 5 << 16, 97 | 5 << 16, 129 | 6 << 16, 193 | 6 << 16, 257 | 7 << 16, 385 | 7 << 16, 513 | 8 << 16, 769 | 8 << 16,
                                        1025 | 9 << 16, 1537 | 9 << 16, 2049 | 10 << 16, 3073 | 10 << 16, 4097 | 11 << 16, 6145 | 11 << 16, 8193 | 12 << 16,
                                        12289 | 12 << 16, 16385 | 13 << 16, 24577 | 13 << 16};
    constexpr uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    InflateResult res{0, 1};
    if (n < 6) return res;                                  // header + at least an empty block + trailer
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(src);
    const uint32_t skip = (uint32_t)(a0 & 3);
    Reader r{};
    r.src32 = reinterpret_cast<const uint32_t*>(a0 - skip);
    r.ndw = (uint32_t)((n + skip + 3) >> 2);
    r.end_bit = (uint32_t)(n + skip) * 8u;
    seek<NL>(S, r, skip * 8u, lane);
    ensure<NL>(S, r, 16, lane);
    const uint32_t cmf = take(r, 8), flg = take(r, 8);
    if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) return res;      // not deflate / bad check / preset dictionary
    uint32_t op = 0;
    for (;;) {
        ensure<NL>(S, r, 3, lane);
        const uint32_t last = take(r, 1), type = take(r, 2);
        if (type == 3) return res;
        if (type == 0) {                                    // stored: LEN, ~LEN, bytes — copied straight from the stream in memory
            take(r, r.have & 7);
            ensure<NL>(S, r, 32, lane);
            const uint32_t len = take(r, 16), nlen = take(r, 16);
            if ((len ^ nlen) != 0xffffu) return res;
            const uint32_t byte0 = bitpos(r) >> 3;          // from the aligned base
            if ((uint64_t)byte0 + len > (uint64_t)n + skip) return res;
            const uint8_t* from = reinterpret_cast<const uint8_t*>(r.src32) + byte0;
            for (uint32_t k0 = 0; k0 < len; k0 += NL) {
                const uint32_t k = k0 + lane;
                if (k < len) {
                    const uint8_t v = from[k];
                    S.ring[(op + k) & (RING - 1)] = v;
                    if (op + k < cap) dst[op + k] = v;
                }
            }
            op += len;
            seek<NL>(S, r, (byte0 + len) * 8u, lane);
        } else {
            TD_INF_SYNC();
            bool ok = true;
            if (type == 1) {                                // fixed codes
                if (lane == 0) {
                    for (int s = 0; s < 288; ++s) S.lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
                    ok = build(S.lens, 288, S.lit_count, S.lit_sym, S.lit_fast, INF_LIT_FAST);
                    for (int s = 0; s < 30; ++s) S.lens[s] = 5;
                    ok = ok && build(S.lens, 30, S.dist_count, S.dist_sym, S.dist_fast, INF_DIST_FAST);
                }
            } else {                                        // dynamic codes: the code-length code first, then the two tables
                ensure<NL>(S, r, 14, lane);
                const int hlit = (int)take(r, 5) + 257, hdist = (int)take(r, 5) + 1, hclen = (int)take(r, 4) + 4;
                if (hlit > 286 || hdist > 30) return res;
                uint8_t cl[19];
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                for (int i = 0; i < hclen; ++i) {
                    ensure<NL>(S, r, 3, lane);
                    cl[ORDER[i]] = (uint8_t)take(r, 3);
                }
                TD_INF_SYNC();
                if (lane == 0) {
                    for (int i = 0; i < 19; ++i) S.lens[i] = cl[i];
                    ok = build(S.lens, 19, S.lit_count, S.lit_sym, S.lit_fast, 7);       // (the code-length code uses the literal table's arrays for a moment)
                }
                TD_INF_SYNC();
#ifdef __HIP_DEVICE_COMPILE__
                ok = TD_INF_UNIFORM(__shfl((int)ok, 0)) != 0;
#endif
                if (!ok) return res;
                // the hlit + hdist code lengths, run-length coded; every lane decodes them (uniform), lane 0 keeps them
                uint8_t prev = 0;
                int idx = 0;
                uint8_t* L = S.lens + 32;                   // (past the 19 entries still in use by nothing: the tables above are built)
                while (idx < hlit + hdist) {
                    const int s = decode_sym<NL>(S, r, S.lit_fast, 7, S.lit_count, S.lit_sym, lane);
                    if (s < 0 || bitpos(r) > r.end_bit) return res;
                    int rep = 1;
                    uint8_t v = (uint8_t)s;
                    if (s == 16) {
                        if (idx == 0) return res;
                        ensure<NL>(S, r, 2, lane);
                        rep = 3 + (int)take(r, 2);
                        v = prev;
                    } else if (s == 17) {
                        ensure<NL>(S, r, 3, lane);
                        rep = 3 + (int)take(r, 3);
                        v = 0;
                    } else if (s == 18) {
                        ensure<NL>(S, r, 7, lane);
                        rep = 11 + (int)take(r, 7);
                        v = 0;
                    }
                    if (idx + rep > hlit + hdist) return res;
                    if (lane == 0)
                        for (int k = 0; k < rep; ++k) L[idx + k] = v;
                    idx += rep;
                    prev = v;
                }
                TD_INF_SYNC();
                if (lane == 0) {
                    ok = L[256] != 0;                        // no end-of-block code
                    ok = ok && build(L, hlit, S.lit_count, S.lit_sym, S.lit_fast, INF_LIT_FAST);
                    ok = ok && build(L + hlit, hdist, S.dist_count, S.dist_sym, S.dist_fast, INF_DIST_FAST);
                }
            }
            TD_INF_SYNC();
#ifdef __HIP_DEVICE_COMPILE__
            ok = TD_INF_UNIFORM(__shfl((int)ok, 0)) != 0;
#endif
            if (!ok) return res;
            for (;;) {                                      // the block's symbols
                uint32_t len = 0, dist = 0;                 // the match this step ends with (len == 0: none)
#ifdef __HIP_DEVICE_COMPILE__
                if (NL == 64) {
                    r.have = (int)TD_INF_UNIFORM(r.have);
                    r.acc = ((uint64_t)TD_INF_UNIFORM(r.acc >> 32) << 32) | TD_INF_UNIFORM((uint32_t)r.acc);
                    r.rd = TD_INF_UNIFORM(r.rd);
                    r.loaded = TD_INF_UNIFORM(r.loaded);
                    op = TD_INF_UNIFORM(op);
                    // Several symbols per step: every lane looks up the literal / length code AND the distance code that WOULD start
                    // at its bit of the 64-bit reservoir (two LDS accesses for all 64 candidates), then the wave follows the chain
                    // code → next code through those answers with lane reads — a few scalar instructions per symbol instead of a
                    // table round trip each. The lanes on the chain store their literals together; a length code that ends the
                    // chain takes its extra bits, its distance code (the other lookup, at the bit where it starts) and that one's
                    // extra bits out of the same reservoir when they are all there. The chain also ends at the end-of-block code,
                    // at a code longer than a table's index, or where the reservoir runs out; the one-symbol path below takes over.
                    ensure<NL>(S, r, 33, lane);             // 33 .. 64 bits in the reservoir
                    const int avail = r.have - INF_LIT_FAST;                 // a code may start at bits 0 .. avail: its index bits are all there
                    const uint64_t here = r.acc >> lane;
                    const int el = (int)S.lit_fast[(uint32_t)here & ((1u << INF_LIT_FAST) - 1u)];
                    const int dl = (int)S.dist_fast[(uint32_t)here & ((1u << INF_DIST_FAST) - 1u)];
                    int cur = 0, cnt = 0, rank = -1;
                    while (cur <= avail) {
                        const int ec = __builtin_amdgcn_readlane(el, cur);
                        if ((ec >> 9) == 0 || (ec & 511) >= 256) break;
                        if (lane == cur) rank = cnt;
                        cur += ec >> 9;
                        ++cnt;
                    }
                    int used = cur;
                    if (cur <= avail) {
                        const int ec = __builtin_amdgcn_readlane(el, cur);
                        const int sy = ec & 511;
                        if ((ec >> 9) != 0 && sy > 256 && sy <= 285) {
                            int p = cur + (ec >> 9);
                            // base and extra bits of a length code by arithmetic (RFC 1951 3.2.5: groups of four codes per extra bit) — a
                            // table in memory is a scalar load of ~200 cycles on the chain's critical path
                            const int c = sy - 257;
                            const int xb = c < 8 || c == 28 ? 0 : (c >> 2) - 1;
                            const uint32_t lbase = c < 8 ? 3u + (uint32_t)c : (c == 28 ? 258u : 3u + ((4u + ((uint32_t)c & 3u)) << xb));
                            if (p + xb + INF_DIST_FAST <= r.have) {          // the length's extra bits and the distance code's index bits
                                const uint32_t l0 = lbase + ((uint32_t)(r.acc >> p) & ((1u << xb) - 1u));
                                p += xb;
                                const int dc = __builtin_amdgcn_readlane(dl, p);
                                const int ds = dc & 511;
                                if ((dc >> 9) != 0 && ds <= 29) {
                                    p += dc >> 9;
                                    const int db = ds < 4 ? 0 : (ds >> 1) - 1;                // (pairs of codes per extra bit)
                                    const uint32_t dbase = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << db);
                                    if (p + db <= r.have) {
                                        dist = dbase + (db ? (uint32_t)(r.acc >> p) & ((1u << db) - 1u) : 0u);      // (p <= 63 whenever db > 0)
                                        len = l0;
                                        used = p + db;
                                    }
                                }
                            }
                        }
                    }
                    if (cnt || len) {
                        if (rank >= 0) {
                            S.ring[(op + rank) & (RING - 1)] = (uint8_t)el;
                            if (op + rank < cap) dst[op + rank] = (uint8_t)el;
                        }
                        r.acc = used < 64 ? r.acc >> used : 0;                // (used <= have)
                        r.have -= used;
                        op += (uint32_t)cnt;
                        if (bitpos(r) > r.end_bit) return res;
                        if (!len) continue;
                    }
                }
#endif
                if (!len) {                                 // one symbol, any code length
                    const int s = decode_sym<NL>(S, r, S.lit_fast, INF_LIT_FAST, S.lit_count, S.lit_sym, lane);
                    if (s < 0 || bitpos(r) > r.end_bit) return res;
                    if (s < 256) {
                        if (lane == 0) {
                            S.ring[op & (RING - 1)] = (uint8_t)s;
                            if (op < cap) dst[op] = (uint8_t)s;
                        }
                        ++op;
                        continue;
                    }
                    if (s == 256) break;
                    if (s > 285) return res;
                    ensure<NL>(S, r, 5, lane);
                    const uint32_t lc = TD_INF_UNIFORM(LEN_CODE[s - 257]);
                    len = (lc & 0xffffu) + take(r, (int)(lc >> 16));
                    const int ds = decode_sym<NL>(S, r, S.dist_fast, INF_DIST_FAST, S.dist_count, S.dist_sym, lane);
                    if (ds < 0 || ds > 29) return res;
                    ensure<NL>(S, r, 13, lane);
                    const uint32_t dcode = TD_INF_UNIFORM(DIST_CODE[ds]);
                    dist = (dcode & 0xffffu) + take(r, (int)(dcode >> 16));
                }
                if (dist > op || bitpos(r) > r.end_bit) return res;
                // out[op + k] = out[op - dist + (k mod dist)]: every source byte was written before this match began. Nearly every
                // match is shorter than its distance (k mod dist = k): the division runs only for the overlapping ones (runs).
                const bool wraps = dist < len;
                if (RING == INF_WINDOW || dist + len + 64 <= (uint32_t)RING) {       // the sources outlive this match's own writes to the ring
                    for (uint32_t k0 = 0; k0 < len; k0 += NL) {
                        const uint32_t k = k0 + lane;
                        if (k < len) {
                            const uint8_t v = S.ring[(op - dist + (wraps ? k % dist : k)) & (RING - 1)];
                            S.ring[(op + k) & (RING - 1)] = v;
                            if (op + k < cap) dst[op + k] = v;
                        }
                    }
                } else {                                    // sources that have left the ring: the block's output in memory
                    TD_INF_STORES_DONE();
                    for (uint32_t k0 = 0; k0 < len; k0 += NL) {
                        const uint32_t k = k0 + lane;
                        if (k < len) {
                            const uint32_t from = op - dist + (wraps ? k % dist : k);
                            const uint8_t v = from < cap ? TD_INF_LOAD_OUT(dst + from) : (uint8_t)0;     // (past cap: the block fails with status 2 anyway)
                            S.ring[(op + k) & (RING - 1)] = v;
                            if (op + k < cap) dst[op + k] = v;
                        }
                    }
                }
                op += len;
            }
        }
        if (last) break;
    }
    res.produced = op;
    res.status = op > cap ? 2 : 0;
    return res;
}
