// Proposal generation: per-level top-k over the RPN logits, anchor decoding, clipping, per-level NMS and
// the cross-level merge — detectron2 find_top_rpn_proposals, reached from
// TreeDetection/prediction.py:183. Also the generic sort + greedy-NMS kernels shared with the box head
// (fast_rcnn_inference, Appendix A item 11) and exported as td_nms.
//
// Ordering rules (bit-exact with oracle/ops_ref.py): scores descending, ties by lower index first;
// IoU = inter / (area_i + area_j - inter) in float32, one IEEE operation per step, suppress when IoU > thr.
// These are latency-bound integer / compare kernels: wave64 ballots and LDS-resident bit masks, no MFMA.
#include "common.h"
#include "detect.h"

namespace {

__device__ __forceinline__ uint32_t float_to_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float → larger key
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Bitonic sort of n (power of two) u64 in LDS, DESCENDING, by all threads of the block.
__device__ void bitonic_sort_desc(unsigned long long* s, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s[i], b = s[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- dense keys + coarse histogram (all images, all levels, every CU) -----------------------------------------------
// The fused head tensor interleaves 3 logits with 12 deltas per position (60-B rows): gathering the logits is a strided
// read of the whole tensor, and p2 alone holds 120 000 anchors per image — as the first pass of the one-block-per-
// (image, level) top-k kernel it was a third of that kernel's 0.3-0.5 ms. Here every CU takes part: key = order-
// preserving integer image of the logit, written densely, and a 4096-bin histogram of the keys' top 12 bits per
// (image, level) that lets the top-k kernel cut each level down to a few thousand candidates in ONE pass.
constexpr int HIST_BITS = 12, HIST_BINS = 1 << HIST_BITS;
constexpr int KEYS_PER_BLOCK = 4096;       // 256 threads x 16 anchors
__global__ __launch_bounds__(256) void rpn_keys_hist_kernel(RpnLevels lv, uint32_t* __restrict__ key_ws,
                                                            unsigned int* __restrict__ hist) {
    // the logits of a level crowd into a few dozen of the 4096 bins: counted in an LDS histogram per block and level first
    // (a block's anchors span at most two levels), only the non-empty bins reach the global one (one atomic per bin and
    // block — per-key global atomics on the same few addresses took 0.4 ms)
    __shared__ unsigned int lh[2][HIST_BINS];
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * KEYS_PER_BLOCK;
    int level0 = 0;
#pragma unroll
    for (int l = 1; l < RPN_LEVELS; ++l) level0 += i0 >= lv.anchor_off[l] ? 1 : 0;
    for (int t = threadIdx.x; t < 2 * HIST_BINS; t += blockDim.x) (&lh[0][0])[t] = 0u;
    __syncthreads();
    bool second = false;
    for (int i = i0 + threadIdx.x; i < i0 + KEYS_PER_BLOCK && i < lv.total_anchors; i += blockDim.x) {
        int level = 0;
#pragma unroll
        for (int l = 1; l < RPN_LEVELS; ++l) level += i >= lv.anchor_off[l] ? 1 : 0;
        const int j = i - lv.anchor_off[level];
        const int pos = j / RPN_A, a = j - pos * RPN_A;
        const float* __restrict__ head = lv.head[level] + (size_t)b * lv.h[level] * lv.w[level] * RPN_HEAD_C;
        const uint32_t key = float_to_key(head[(size_t)pos * RPN_HEAD_C + a]);
        key_ws[(size_t)b * lv.total_anchors + i] = key;
        const int slot = level - level0;                       // 0, or 1.. when the block crosses into the next level(s)
        if (slot <= 1) {
            atomicAdd(&lh[slot][key >> (32 - HIST_BITS)], 1u);
            second |= slot == 1;
        } else {                                               // a third level inside one block (tiny top levels): straight to global
            atomicAdd(&hist[((size_t)b * RPN_LEVELS + level) * HIST_BINS + (key >> (32 - HIST_BITS))], 1u);
        }
    }
    const bool any_second = __syncthreads_or(second);
    for (int t = threadIdx.x; t < HIST_BINS; t += blockDim.x) {
        const unsigned int c0 = lh[0][t];
        if (c0) atomicAdd(&hist[((size_t)b * RPN_LEVELS + level0) * HIST_BINS + t], c0);
        if (any_second) {
            const unsigned int c1 = lh[1][t];
            if (c1) atomicAdd(&hist[((size_t)b * RPN_LEVELS + level0 + 1) * HIST_BINS + t], c1);
        }
    }
}

// ---- per (image, level) top-k + decode ---------------------------------------------------------------
// One block (1024 threads) per (image, level), after rpn_keys_hist_kernel. Fast path: the coarse histogram names the bin
// that holds the k-th largest key; the keys of that bin and above (<= TOPK_FAST of them) are gathered in one pass over
// the dense keys and sorted in LDS as (key, ~index) pairs — the first k are the top-k in (key desc, index asc) order, ties
// at the cut included. Otherwise (a bin so crowded that the candidates do not fit: near-constant logits) the general
// path: radix-select the k-th largest key (4 x 8-bit passes, LDS histogram), gather the winners (ties at the threshold:
// lowest indices first), bitonic-sort them by (key desc, index asc). Then decode / clip the boxes. Same result either way.
constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_FAST = 4096;

__global__ __launch_bounds__(TOPK_THREADS) void rpn_topk_decode_kernel(RpnLevels lv, ImgSizes valid, int B, int topk,
                                                                       const uint32_t* __restrict__ key_ws,
                                                                       const unsigned int* __restrict__ hist12,
                                                                       float* __restrict__ cand_boxes,
                                                                       float* __restrict__ cand_scores,
                                                                       int* __restrict__ cand_valid,
                                                                       int* __restrict__ cand_idx) {
    const int level = blockIdx.x, b = blockIdx.y;
    const int H = lv.h[level], W = lv.w[level];
    const int n = H * W * RPN_A;
    const int k = topk < n ? topk : n;
    const float* __restrict__ head = lv.head[level] + (size_t)b * H * W * RPN_HEAD_C;
    const uint32_t* __restrict__ keys = key_ws + ((size_t)b * lv.total_anchors + lv.anchor_off[level]);

    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sel[TOPK_FAST];
    __shared__ unsigned int s_prefix, s_need, s_cnt_gt, s_cnt_eq, s_eq_total;
    __shared__ unsigned int wave_cnt[2][TOPK_THREADS / 64];
    __shared__ unsigned int scan[TOPK_THREADS];
    __shared__ int s_tbin;
    __shared__ unsigned int s_cand;

    const int tid = threadIdx.x;
    // ---- fast path: threshold bin of the coarse histogram ----
    {
        const unsigned int* __restrict__ h = hist12 + ((size_t)b * RPN_LEVELS + level) * HIST_BINS;
        constexpr int PER = HIST_BINS / TOPK_THREADS;          // bins per thread; thread 0 owns the HIGHEST bins
        unsigned int c[PER], mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            c[q] = h[HIST_BINS - 1 - (tid * PER + q)];
            mine += c[q];
        }
        scan[tid] = mine;
        if (tid == 0) {
            s_tbin = -1;
            s_cand = 0;
        }
        __syncthreads();
        for (int off = 1; off < TOPK_THREADS; off <<= 1) {     // inclusive scan over the threads (highest bins first)
            const unsigned int v = tid >= off ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        unsigned int above = scan[tid] - mine;                 // keys in bins above this thread's
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (above < (unsigned)k && above + c[q] >= (unsigned)k) {      // the k-th largest key lies in this bin (one bin only)
                s_tbin = HIST_BINS - 1 - (tid * PER + q);
                s_cand = above + c[q];
            }
            above += c[q];
        }
        __syncthreads();
    }
    if (k > 0 && s_tbin >= 0 && s_cand <= (unsigned)TOPK_FAST) {
        const uint32_t tb = (uint32_t)s_tbin;
        const unsigned int ncand = s_cand;
        __syncthreads();
        if (tid == 0) s_cnt_gt = 0;
        __syncthreads();
        for (int i = tid; i < n; i += TOPK_THREADS) {
            const uint32_t key = keys[i];
            if ((key >> (32 - HIST_BITS)) >= tb) {
                const unsigned int pos = atomicAdd(&s_cnt_gt, 1u);
                sel[pos] = ((unsigned long long)key << 32) | (0xffffffffu - (uint32_t)i);
            }
        }
        int P = 1024;
        while (P < (int)ncand) P <<= 1;
        for (int i = (int)ncand + tid; i < P; i += TOPK_THREADS) sel[i] = 0ull;      // padding sorts last
        __syncthreads();
        bitonic_sort_desc(sel, P);
    } else {
    if (tid == 0) {
        s_prefix = 0;
        s_need = k;
    }
    __syncthreads();
    // radix select: after the 4 passes s_prefix is the k-th largest key, s_need the number of elements
    // equal to it that still belong to the top-k
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned int prefix = s_prefix;
        const unsigned int himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (int i = tid; i < n; i += TOPK_THREADS) {
            const uint32_t key = keys[i];
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int need = s_need, bin = 255;
            for (;; --bin) {
                const unsigned int c = hist[bin];
                if (c >= need || bin == 0) break;
                need -= c;
            }
            s_prefix = prefix | (bin << shift);
            s_need = need;
            s_eq_total = hist[bin];      // after the last pass: how many keys equal the threshold exactly
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;
    const unsigned int need_eq = s_need;
    // compaction: every key > thr, plus need_eq keys == thr. When ALL keys equal to the threshold belong to the
    // top-k (the normal case: float logits rarely tie exactly at the cut) the slot order is irrelevant — the bitonic
    // sort below orders by (key, index) — so one barrier-free pass with an LDS counter does it. Only a tie that
    // straddles the cut needs the ordered pass (lowest indices first).
    if (tid == 0) {
        s_cnt_gt = 0;
        s_cnt_eq = 0;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if (s_eq_total == need_eq) {
        for (int i = tid; i < n; i += TOPK_THREADS) {
            const uint32_t key = keys[i];
            if (key >= thr) {
                const unsigned int pos = atomicAdd(&s_cnt_gt, 1u);
                sel[pos] = ((unsigned long long)key << 32) | (0xffffffffu - (uint32_t)i);
            }
        }
        __syncthreads();
    } else {
        for (int base = 0; base < n; base += TOPK_THREADS) {
            const int i = base + tid;
            uint32_t key = 0;
            bool gt = false, eq = false;
            if (i < n) {
                key = keys[i];
                gt = key > thr;
                eq = key == thr;
            }
            const unsigned long long mg = __ballot(gt), me = __ballot(eq);
            if (lane == 0) {
                wave_cnt[0][wave] = __popcll(mg);
                wave_cnt[1][wave] = __popcll(me);
            }
            __syncthreads();
            unsigned int off_g = s_cnt_gt, off_e = s_cnt_eq;
            for (int w = 0; w < wave; ++w) {
                off_g += wave_cnt[0][w];
                off_e += wave_cnt[1][w];
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            off_g += __popcll(mg & below);
            off_e += __popcll(me & below);
            // slots: [0, k - need_eq) hold the > thr keys, [k - need_eq, k) the == thr keys
            if (gt) sel[off_g] = ((unsigned long long)key << 32) | (0xffffffffu - (uint32_t)i);
            if (eq && off_e < need_eq) sel[k - need_eq + off_e] = ((unsigned long long)key << 32) | (0xffffffffu - (uint32_t)i);
            __syncthreads();
            if (tid == 0) {
                unsigned int tg = 0, te = 0;
                for (int w = 0; w < TOPK_THREADS / 64; ++w) {
                    tg += wave_cnt[0][w];
                    te += wave_cnt[1][w];
                }
                s_cnt_gt += tg;
                s_cnt_eq += te;
            }
            __syncthreads();
        }
    }
    for (int i = k + tid; i < 1024; i += TOPK_THREADS) sel[i] = 0ull;   // padding sorts last
    __syncthreads();
    bitonic_sort_desc(sel, 1024);
    }   // general path

    // decode + clip (apply_deltas, weights 1,1,1,1; Boxes.clip; nonempty(0); isfinite)
    const size_t obase = ((size_t)b * RPN_LEVELS + level) * RPN_CAND;
    if (tid < RPN_CAND) {
        int ok = 0;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, score = 0.f;
        int idx = -1;
        if (tid < k) {
            const unsigned long long e = sel[tid];
            idx = (int)(0xffffffffu - (uint32_t)(e & 0xffffffffull));
            score = key_to_float((uint32_t)(e >> 32));
            const int pos = idx / RPN_A, a = idx - pos * RPN_A;
            const int py = pos / W, px = pos - py * W;
            const float sx = (float)(px * lv.stride[level]), sy = (float)(py * lv.stride[level]);
            const float ax1 = __fadd_rn(sx, lv.base[level][a][0]), ay1 = __fadd_rn(sy, lv.base[level][a][1]);
            const float ax2 = __fadd_rn(sx, lv.base[level][a][2]), ay2 = __fadd_rn(sy, lv.base[level][a][3]);
            const float* d = head + (size_t)pos * RPN_HEAD_C + RPN_A + a * 4;
            decode_box(ax1, ay1, ax2, ay2, d[0], d[1], d[2], d[3], 1.f, 1.f, 1.f, 1.f, x1, y1, x2, y2);
            const bool fin = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(score);
            const float ih = (float)valid.h[b], iw = (float)valid.w[b];
            x1 = fminf(fmaxf(x1, 0.f), iw);
            y1 = fminf(fmaxf(y1, 0.f), ih);
            x2 = fminf(fmaxf(x2, 0.f), iw);
            y2 = fminf(fmaxf(y2, 0.f), ih);
            ok = fin && (__fsub_rn(x2, x1) > 0.f) && (__fsub_rn(y2, y1) > 0.f);
        }
        float4 bx = make_float4(x1, y1, x2, y2);
        *reinterpret_cast<float4*>(cand_boxes + (obase + tid) * 4) = bx;
        cand_scores[obase + tid] = score;
        cand_valid[obase + tid] = ok;
        cand_idx[obase + tid] = idx;
    }
}

// ---- suppression bit matrix --------------------------------------------------------------------------
// grid (col block, row block, item); 64 threads; thread = one row box, loops over the 64 column boxes.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                      int stride_items, float thr, unsigned long long* __restrict__ mask,
                                                      int words) {
    const int cb = blockIdx.x, rb = blockIdx.y, item = blockIdx.z;
    const int n = counts ? counts[item] : stride_items;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
    __shared__ float4 cbox[64];
    const float4* bx = reinterpret_cast<const float4*>(boxes) + (size_t)item * stride_items;
    const int j0 = cb * 64;
    if (j0 + (int)threadIdx.x < n) cbox[threadIdx.x] = bx[j0 + threadIdx.x];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 a = bx[i];
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    unsigned long long bits = 0ull;
    const int jn = (n - j0) < 64 ? (n - j0) : 64;
    for (int jj = (cb == rb ? (int)threadIdx.x + 1 : 0); jj < jn; ++jj) {
        const float4 c = cbox[jj];
        if (iou_gt(a, area_a, c, thr)) bits |= 1ull << jj;
    }
    mask[((size_t)item * stride_items + i) * words + cb] = bits;
}

// ---- greedy scan ----------------------------------------------------------------------------------------
// One block of 16 waves per item. Wave w owns word w of the "removed" bitmap. Chunk c (boxes 64c..64c+63):
// wave c resolves the 64 in-chunk decisions serially from the diagonal words, publishes the keep bits, then
// every later wave ORs in the rows of the kept boxes.
__global__ __launch_bounds__(1024) void nms_scan_kernel(const int* __restrict__ counts, int stride_items,
                                                        const int* __restrict__ valid,
                                                        const unsigned long long* __restrict__ mask, int words,
                                                        int* __restrict__ keep_idx, int* __restrict__ keep_count,
                                                        int max_keep) {
    const int item = blockIdx.x;
    const int n = counts ? counts[item] : stride_items;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned long long s_keep[16];
    __shared__ int s_base;
    const unsigned long long* M = mask + (size_t)item * stride_items * words;
    const int* vd = valid ? valid + (size_t)item * stride_items : nullptr;
    unsigned long long removed = 0ull;   // word `wave` of the removed set (wave-uniform)
    {
        // boxes flagged invalid (non-finite / empty / beyond n) start out removed
        const int i = wave * 64 + lane;
        const bool dead = (i >= n) || (vd && !vd[i]);
        removed = __ballot(dead);
    }
    if (threadIdx.x == 0) s_base = 0;
    const int chunks = (n + 63) >> 6;
    for (int c = 0; c < chunks; ++c) {
        if (wave == c) {
            const int i = c * 64 + lane;
            unsigned long long diag = (i < n) ? M[(size_t)i * words + c] : 0ull;
            unsigned long long rem = removed, kept = 0ull;
            for (int j = 0; j < 64; ++j) {
                const unsigned long long dj = __shfl(diag, j);
                if (!((rem >> j) & 1ull)) {
                    kept |= 1ull << j;
                    rem |= dj;
                }
            }
            if (lane == 0) s_keep[c] = kept;
        }
        __syncthreads();
        const unsigned long long kept = s_keep[c];
        if (wave > c && wave < chunks) {
            const int i = c * 64 + lane;
            unsigned long long v = 0ull;
            if (((kept >> lane) & 1ull) && i < n) v = M[(size_t)i * words + wave];
            for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
            removed |= v;
        }
        // emit kept indices of this chunk in order
        if (wave == 0) {
            const int base = s_base;
            const bool k = (kept >> lane) & 1ull;
            const int pos = base + __popcll(kept & ((1ull << lane) - 1ull));
            if (k && pos < max_keep) keep_idx[(size_t)item * max_keep + pos] = c * 64 + lane;
            if (lane == 0) s_base = base + __popcll(kept);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int total = s_base;
        keep_count[item] = total < max_keep ? total : max_keep;
    }
}

// ---- cross-level merge: kept candidates of the 5 levels → top post_nms_topk by (score desc, concat index asc) -----
__global__ __launch_bounds__(1024) void rpn_merge_kernel(const float* __restrict__ cand_boxes,
                                                         const float* __restrict__ cand_scores,
                                                         const int* __restrict__ keep_idx,
                                                         const int* __restrict__ keep_count, int post_topk,
                                                         float* __restrict__ props, float* __restrict__ prop_scores,
                                                         int* __restrict__ prop_count, int prop_stride) {
    // Every level's keep list is already ordered (score desc, candidate index asc), and the composite
    // (score key, concat index) is unique, so the merged position of an element is its own rank plus, for each other
    // level, the number of that level's elements that sort before it — four binary searches in LDS, no sort.
    const int b = blockIdx.x;
    __shared__ unsigned long long s[RPN_LEVELS][RPN_CAND_POW2];
    __shared__ int s_cnt[RPN_LEVELS];
    if (threadIdx.x < RPN_LEVELS) s_cnt[threadIdx.x] = keep_count[b * RPN_LEVELS + threadIdx.x];
    __syncthreads();
    for (int i = threadIdx.x; i < RPN_LEVELS * RPN_CAND_POW2; i += blockDim.x) {
        const int level = i >> 10, r = i & 1023;
        unsigned long long e = 0ull;
        if (r < s_cnt[level]) {
            const int item = b * RPN_LEVELS + level;
            const int ci = keep_idx[(size_t)item * RPN_CAND + r];
            const float sc = cand_scores[(size_t)item * RPN_CAND + ci];
            // concat index: level-major, then rank inside the level's (filtered, score-sorted) list
            e = ((unsigned long long)float_to_key(sc) << 32) | (0xffffffffu - (uint32_t)(level * RPN_CAND_POW2 + ci));
        }
        s[level][r] = e;
    }
    __syncthreads();
    int total = 0;
    for (int l = 0; l < RPN_LEVELS; ++l) total += s_cnt[l];
    const int cnt = total < post_topk ? total : post_topk;
    for (int i = threadIdx.x; i < RPN_LEVELS * RPN_CAND_POW2; i += blockDim.x) {
        const int level = i >> 10, r = i & 1023;
        if (r >= s_cnt[level]) continue;
        const unsigned long long e = s[level][r];
        int rank = r;
        for (int l = 0; l < RPN_LEVELS; ++l) {
            if (l == level) continue;
            int lo = 0, hi = s_cnt[l];          // first position whose composite is < e (list is descending)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s[l][mid] > e) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < cnt) {
            const uint32_t ci_all = 0xffffffffu - (uint32_t)(e & 0xffffffffull);
            const int ci = ci_all & 1023;
            const size_t src = ((size_t)(b * RPN_LEVELS + level)) * RPN_CAND + ci;
            *reinterpret_cast<float4*>(props + ((size_t)b * prop_stride + rank) * 4) =
                *reinterpret_cast<const float4*>(cand_boxes + src * 4);
            prop_scores[(size_t)b * prop_stride + rank] = cand_scores[src];
        }
    }
    for (int i = cnt + threadIdx.x; i < prop_stride; i += blockDim.x) {
        *reinterpret_cast<float4*>(props + ((size_t)b * prop_stride + i) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        prop_scores[(size_t)b * prop_stride + i] = 0.f;
    }
    if (threadIdx.x == 0) prop_count[b] = cnt;
}

// ---- generic: sort n <= 1024 boxes per item by (score desc, index asc); gather sorted boxes ---------------------
__global__ __launch_bounds__(1024) void sort_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const int* __restrict__ flags, const int* __restrict__ counts,
                                                          int stride_items, float* __restrict__ sboxes,
                                                          float* __restrict__ sscores, int* __restrict__ sidx,
                                                          int* __restrict__ scount) {
    const int item = blockIdx.x;
    const int n = counts ? counts[item] : stride_items;
    __shared__ unsigned long long s[1024];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int i = threadIdx.x;
    unsigned long long e = 0ull;
    bool on = false;
    if (i < n && i < 1024) {
        on = flags ? flags[(size_t)item * stride_items + i] != 0 : true;
        if (on) e = ((unsigned long long)float_to_key(scores[(size_t)item * stride_items + i]) << 32) | (0xffffffffu - (uint32_t)i);
    }
    s[i] = e;
    if (on) atomicAdd(&s_n, 1);
    __syncthreads();
    bitonic_sort_desc(s, 1024);
    const int m = s_n;
    if (i < stride_items) {
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        int src = -1;
        if (i < m) {
            src = (int)(0xffffffffu - (uint32_t)(s[i] & 0xffffffffull));
            bx = *reinterpret_cast<const float4*>(boxes + ((size_t)item * stride_items + src) * 4);
            sc = scores[(size_t)item * stride_items + src];
        }
        *reinterpret_cast<float4*>(sboxes + ((size_t)item * stride_items + i) * 4) = bx;
        sscores[(size_t)item * stride_items + i] = sc;
        sidx[(size_t)item * stride_items + i] = src;
    }
    if (i == 0) scount[item] = m;
}

// one item, any number of 64-box chunks: row block = blockIdx.y + rb0
__global__ __launch_bounds__(64) void nms_mask_rows_kernel(const float* __restrict__ boxes, int n, float thr,
                                                           unsigned long long* __restrict__ mask, int words, int rb0) {
    const int cb = blockIdx.x, rb = blockIdx.y + rb0;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
    __shared__ float4 cbox[64];
    const float4* bx = reinterpret_cast<const float4*>(boxes);
    const int j0 = cb * 64;
    if (j0 + (int)threadIdx.x < n) cbox[threadIdx.x] = bx[j0 + threadIdx.x];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 a = bx[i];
    const float area_a = __fmul_rn(__fsub_rn(a.z, a.x), __fsub_rn(a.w, a.y));
    unsigned long long bits = 0ull;
    const int jn = (n - j0) < 64 ? (n - j0) : 64;
    for (int jj = (cb == rb ? (int)threadIdx.x + 1 : 0); jj < jn; ++jj) {
        const float4 c = cbox[jj];
        if (iou_gt(a, area_a, c, thr)) bits |= 1ull << jj;
    }
    mask[(size_t)i * words + cb] = bits;
}

// ---- n > 1024 boxes per item (td_nms only: the engine's item sizes are bounded at td_engine_create) ---------------------
// Sort by counting: rank(i) = #{j : key_j > key_i, or key_j == key_i and j < i} — the same total order as the bitonic
// path (score descending, lower index first), O(n^2) compares staged through LDS, every CU takes part.
__global__ __launch_bounds__(256) void rank_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n,
                                                        float* __restrict__ sboxes, float* __restrict__ sscores,
                                                        int* __restrict__ sidx, int* __restrict__ scount) {
    __shared__ uint32_t tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ki = i < n ? float_to_key(scores[i]) : 0u;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        tile[threadIdx.x] = j < n ? float_to_key(scores[j]) : 0u;
        __syncthreads();
        const int jn = (n - j0) < 256 ? (n - j0) : 256;
        for (int jj = 0; jj < jn; ++jj) {
            const uint32_t kj = tile[jj];
            rank += (kj > ki || (kj == ki && j0 + jj < i)) ? 1 : 0;
        }
        __syncthreads();
    }
    if (i < n) {
        *reinterpret_cast<float4*>(sboxes + (size_t)rank * 4) = *reinterpret_cast<const float4*>(boxes + (size_t)i * 4);
        sscores[rank] = scores[i];
        sidx[rank] = i;
    }
    if (i == 0) scount[0] = n;
}

// Greedy scan for any number of 64-box chunks: the removed bitmap lives in LDS (one word per chunk), wave 0 resolves chunk
// c from the diagonal word exactly as nms_scan_kernel does, then the 16 waves share the words c+1 .. of the kept rows.
constexpr int NMS_BIG_MAX = 32768;
__global__ __launch_bounds__(1024) void nms_scan_big_kernel(int n, const unsigned long long* __restrict__ mask, int words,
                                                            int* __restrict__ keep_idx, int* __restrict__ keep_count) {
    __shared__ unsigned long long removed[NMS_BIG_MAX / 64];
    __shared__ unsigned long long s_kept;
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = threadIdx.x; w < words; w += 1024) {
        const int left = n - w * 64;                         // boxes beyond n start out removed
        removed[w] = left >= 64 ? 0ull : (left <= 0 ? ~0ull : (~0ull << left));
    }
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c = 0; c < words; ++c) {
        if (wave == 0) {
            const int i = c * 64 + lane;
            const unsigned long long diag = (i < n) ? mask[(size_t)i * words + c] : 0ull;
            unsigned long long rem = removed[c], kept = 0ull;
            for (int j = 0; j < 64; ++j) {
                const unsigned long long dj = __shfl(diag, j);
                if (!((rem >> j) & 1ull)) {
                    kept |= 1ull << j;
                    rem |= dj;
                }
            }
            const int base = s_base;
            const int pos = base + __popcll(kept & ((1ull << lane) - 1ull));
            if ((kept >> lane) & 1ull) keep_idx[pos] = i;
            if (lane == 0) {
                s_kept = kept;
                s_base = base + __popcll(kept);
            }
        }
        __syncthreads();
        const unsigned long long kept = s_kept;
        const int i = c * 64 + lane;
        const bool mine = ((kept >> lane) & 1ull) && i < n;
        for (int w = c + 1 + wave; w < words; w += 16) {
            unsigned long long v = mine ? mask[(size_t)i * words + w] : 0ull;
            for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
            if (lane == 0) removed[w] |= v;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) keep_count[0] = s_base;
}

// map kept positions (in the sorted order) back to original indices
__global__ void gather_keep_kernel(const int* __restrict__ sidx, const int* __restrict__ keep_pos,
                                   const int* __restrict__ keep_count, int n, int* __restrict__ out_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out_idx[i] = i < keep_count[0] ? sidx[keep_pos[i]] : -1;
}

}  // namespace

size_t rpn_topk_ws_elems(int B, int total_anchors) { return (size_t)B * ((size_t)total_anchors + (size_t)RPN_LEVELS * HIST_BINS); }

td_status rpn_topk_decode_launch(const RpnLevels& lv, const ImgSizes& valid, int B, int topk, uint32_t* key_ws,
                                 float* cand_boxes, float* cand_scores, int* cand_valid, int* cand_idx,
                                 hipStream_t stream) {
    TD_REQUIRE(topk >= 1 && topk <= RPN_CAND, "rpn: pre_nms_topk=%d must be in [1, %d]", topk, RPN_CAND);
    // key_ws holds B * total_anchors keys followed by the B * RPN_LEVELS coarse histograms (rpn_topk_ws_elems)
    unsigned int* hist = key_ws + (size_t)B * lv.total_anchors;
    TD_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)B * RPN_LEVELS * HIST_BINS * sizeof(unsigned int), stream));
    hipLaunchKernelGGL(rpn_keys_hist_kernel, dim3(td_cdiv(lv.total_anchors, KEYS_PER_BLOCK), B), dim3(256), 0, stream, lv, key_ws, hist);
    TD_KERNEL_CHECK();
    hipLaunchKernelGGL(rpn_topk_decode_kernel, dim3(RPN_LEVELS, B), dim3(TOPK_THREADS), 0, stream, lv, valid, B, topk,
                       key_ws, hist, cand_boxes, cand_scores, cand_valid, cand_idx);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status nms_launch(const float* sorted_boxes, const int* counts, const int* valid, int items, int stride_items,
                     float thr, unsigned long long* mask_ws, int* keep_idx, int* keep_count, int max_keep,
                     hipStream_t stream) {
    TD_REQUIRE(stride_items >= 1 && stride_items <= 1024, "nms: at most 1024 boxes per item (got %d)", stride_items);
    const int words = td_cdiv(stride_items, 64);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, items), dim3(64), 0, stream, sorted_boxes, counts,
                       stride_items, thr, mask_ws, words);
    TD_KERNEL_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(items), dim3(1024), 0, stream, counts, stride_items, valid, mask_ws, words,
                       keep_idx, keep_count, max_keep);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status nms_big_launch(const float* boxes, const float* scores, int n, float thr, float* sboxes, float* sscores, int* sidx,
                         int* scount, unsigned long long* mask_ws, int* keep_pos, int* keep_count, hipStream_t stream) {
    TD_REQUIRE(n >= 1 && n <= NMS_BIG_MAX, "nms: at most %d boxes (got %d)", NMS_BIG_MAX, n);
    const int words = td_cdiv(n, 64);
    hipLaunchKernelGGL(rank_sort_kernel, dim3(td_cdiv(n, 256)), dim3(256), 0, stream, boxes, scores, n, sboxes, sscores, sidx, scount);
    TD_KERNEL_CHECK();
    // nms_mask_kernel only writes the words cb >= rb of a row: the scan reads exactly those (diagonal and later words)
    for (int z0 = 0; z0 < words; z0 += 512) {      // grid.y slices (hip grid dims are generous; slices keep launches short)
        const int zn = words - z0 < 512 ? words - z0 : 512;
        hipLaunchKernelGGL(nms_mask_rows_kernel, dim3(words, zn), dim3(64), 0, stream, sboxes, n, thr, mask_ws, words, z0);
        TD_KERNEL_CHECK();
    }
    hipLaunchKernelGGL(nms_scan_big_kernel, dim3(1), dim3(1024), 0, stream, n, mask_ws, words, keep_pos, keep_count);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status rpn_merge_launch(const float* cand_boxes, const float* cand_scores, const int* keep_idx, const int* keep_count,
                           int B, int post_topk, float* props, float* prop_scores, int* prop_count, int prop_stride,
                           hipStream_t stream) {
    hipLaunchKernelGGL(rpn_merge_kernel, dim3(B), dim3(1024), 0, stream, cand_boxes, cand_scores, keep_idx, keep_count,
                       post_topk, props, prop_scores, prop_count, prop_stride);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status sort_boxes_launch(const float* boxes, const float* scores, const int* flags, const int* counts, int items,
                            int stride_items, float* sboxes, float* sscores, int* sidx, int* scount,
                            hipStream_t stream) {
    TD_REQUIRE(stride_items >= 1 && stride_items <= 1024, "sort: at most 1024 boxes per item (got %d)", stride_items);
    hipLaunchKernelGGL(sort_boxes_kernel, dim3(items), dim3(1024), 0, stream, boxes, scores, flags, counts, stride_items,
                       sboxes, sscores, sidx, scount);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status gather_keep_launch(const int* sidx, const int* keep_pos, const int* keep_count, int n, int* out_idx,
                             hipStream_t stream) {
    hipLaunchKernelGGL(gather_keep_kernel, dim3(td_cdiv(n, 256)), dim3(256), 0, stream, sidx, keep_pos, keep_count, n,
                       out_idx);
    TD_KERNEL_CHECK();
    return TD_OK;
}
