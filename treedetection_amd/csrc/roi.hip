// RoI side of the forward: FPN level assignment + RoIAlign (aligned, adaptive sampling), detection decoding,
// output-space finalisation, the mask predictor (1x1 → sigmoid) and the box-region mask paste.
// detectron2 semantics: SURVEY.md Appendix A items 9, 11, 12, 13 (reached from TreeDetection/prediction.py:183);
// operation order mirrors oracle/ops_ref.py so results agree to float32 rounding.
// HBM-gather bound (RoIAlign reads whole 1-KB channel rows per bilinear corner) or trivially small: no MFMA.
#include "common.h"
#include "detect.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const _Float16* p) {
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(_Float16* p, const float4& v) {
    h4 h;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    *reinterpret_cast<h4*>(p) = h;
}

__device__ __forceinline__ int fpn_level(float x1, float y1, float x2, float y2) {
    const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    const float s = sqrtf(area);
    float lv = floorf(__fadd_rn(4.f, log2f(__fadd_rn(__fdiv_rn(s, 224.f), 1e-8f))));
    lv = fminf(fmaxf(lv, 2.f), 5.f);
    if (!(lv >= 2.f)) lv = 2.f;   // NaN area → lowest level (detectron2 casts NaN to int64 min, then clamps)
    return (int)lv - 2;
}

// raw 4-channel piece as it sits in memory (converted to float when it is consumed, not when it is loaded)
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<_Float16> { typedef h4 type; };
__device__ __forceinline__ float4 widen4(const float4& v) { return v; }
__device__ __forceinline__ float4 widen4(const h4& v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }

// ---- RoI geometry shared by the two RoIAlign kernels ---------------------------------------------------------------
// Everything a block needs about its RoI: output row, FPN level, sampling grid and the float32 quantities of the
// oracle's coordinate arithmetic (oracle/ops_ref.py roi_align: aligned = True, sampling_ratio = 0), in its operation order.
struct RoiGeo {
    size_t row;                 // output row of this RoI
    int H, W;                   // map of its FPN level
    int gh, gw, ghw;            // sampling grid per bin
    float sh, sw, bh, bw;       // start and bin size in map pixels
    float count;                // divisor of a bin's sum
    size_t feat_off;            // element offset of the image's map inside the level tensor
    int lvl;
};

// → false when this block has no RoI (r beyond the item's count). Also block (0, 0)'s thread 0 publishes the compact row total.
__device__ __forceinline__ bool roi_setup(const FeatLevels& fl, const float* __restrict__ rois, const int* __restrict__ counts,
                                          int items, int roi_stride, int pooled, int compact, int* __restrict__ total_rows,
                                          int single_level, int item, int r, RoiGeo& g) {
    const int cnt = counts ? counts[item] : roi_stride;
    int prefix = 0;
    if (compact) {
        for (int i = 0; i < item; ++i) prefix += counts[i];
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && total_rows) {
            int t = 0;
            for (int i = 0; i < items; ++i) t += counts[i];
            *total_rows = t;
        }
    }
    if (r >= cnt) return false;
    g.row = compact ? (size_t)(prefix + r) : (size_t)item * roi_stride + r;
    const float4 bx = *reinterpret_cast<const float4*>(rois + ((size_t)item * roi_stride + r) * 4);
    g.lvl = single_level ? 0 : fpn_level(bx.x, bx.y, bx.z, bx.w);
    g.H = fl.h[g.lvl];
    g.W = fl.w[g.lvl];
    const float sc = fl.scale[g.lvl];
    g.feat_off = single_level ? 0 : (size_t)item * g.H * g.W * fl.C;
    g.sw = __fsub_rn(__fmul_rn(bx.x, sc), 0.5f);
    g.sh = __fsub_rn(__fmul_rn(bx.y, sc), 0.5f);
    const float ew = __fsub_rn(__fmul_rn(bx.z, sc), 0.5f), eh = __fsub_rn(__fmul_rn(bx.w, sc), 0.5f);
    const float rw = __fsub_rn(ew, g.sw), rh = __fsub_rn(eh, g.sh);
    g.bh = __fdiv_rn(rh, (float)pooled);
    g.bw = __fdiv_rn(rw, (float)pooled);
    int gh = (int)ceilf(__fdiv_rn(rh, (float)pooled)), gw = (int)ceilf(__fdiv_rn(rw, (float)pooled));
    g.gh = gh > 0 ? gh : 0;
    g.gw = gw > 0 ? gw : 0;
    g.ghw = g.gh * g.gw;
    g.count = (float)(g.ghw > 1 ? g.ghw : 1);
    return true;
}

// sample i of bin p along one axis: → false when it lies outside the map; else the two taps and their weights
__device__ __forceinline__ bool roi_sample(float start, float bin, int p, int i, int g, int size, int& lo, int& hi, float& l, float& h) {
    const float c = __fadd_rn(__fadd_rn(start, __fmul_rn((float)p, bin)), __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), bin), (float)g));
    if (c < -1.f || c > (float)size) return false;
    float cc = c <= 0.f ? 0.f : c;
    lo = (int)cc;
    if (lo >= size - 1) { hi = lo = size - 1; cc = (float)lo; } else hi = lo + 1;
    l = __fsub_rn(cc, (float)lo);
    h = __fsub_rn(1.f, l);
    return true;
}

// One block (256 threads = 4 waves) per (RoI, part); wave g = part * 4 + wave walks bins g, g + 4 * parts, ...; lane =
// channel quad. A bin is the mean of gh x gw bilinear samples = 4 corner loads each; summed one sample after the other
// (the oracle's order: iy outer, ix inner, ((w1 v1 + w2 v2) + w3 v3) + w4 v4 per sample) the loop is a chain of
// dependent rounds "table lookup → 4 loads → wait → add", ≈ 2 µs each next to the trunk's contractions: 66 rounds per
// wave for the box head (12 bins x 5.4 samples), 113 for the mask head when one block owned all 196 bins. The sums keep
// that order, the LOADS do not wait for it: the wave's samples form one flat sequence across its bins, and a ring of D
// samples (4 D raw corner pieces in registers) is kept in flight — sample s + D is issued right after sample s is
// consumed, so the compiler's counted vmcnt leaves 4 (D - 1) loads outstanding at every add. Same arithmetic, same
// order → bit-identical outputs (tests/test_ops_gpu.py, tests/test_engine_gpu.py).
template <typename T, int D>
__global__ __launch_bounds__(256) void roi_align_kernel(FeatLevels fl, const float* __restrict__ rois,
                                                        const int* __restrict__ counts, int items, int roi_stride,
                                                        int pooled, int compact, T* __restrict__ out,
                                                        int* __restrict__ total_rows, int single_level, int parts) {
    const int item = blockIdx.y, r = blockIdx.x / parts, part = blockIdx.x - r * parts;
    RoiGeo geo;
    if (!roi_setup(fl, rois, counts, items, roi_stride, pooled, compact, total_rows, single_level, item, r, geo)) return;
    const size_t row = geo.row;
    const int H = geo.H, W = geo.W, C = fl.C, gh = geo.gh, gw = geo.gw, ghw = geo.ghw;
    const float sh = geo.sh, sw = geo.sw, bh = geo.bh, bw = geo.bw, count = geo.count;
    const T* __restrict__ feat = static_cast<const T*>(fl.feat[geo.lvl]) + geo.feat_off;

    // Sample coordinates are the same for every channel lane: tabulate the (lo, hi, weights, in-range) tuple of every
    // y-sample and x-sample of this RoI ONCE in LDS (same float32 operation order as the oracle), so the bin loop
    // below is loads + multiply-adds only (the per-sample divisions made this kernel VALU-bound).
    constexpr int TAB = 14 * 24;     // pooled * grid entries per axis held in LDS; larger grids take the direct path
    __shared__ int t_lo[2][TAB], t_hi[2][TAB];
    __shared__ float t_l[2][TAB], t_h[2][TAB];
    const bool tab = pooled * gh <= TAB && pooled * gw <= TAB;
    if (tab) {
        for (int t = threadIdx.x; t < pooled * gh + pooled * gw; t += blockDim.x) {
            const int ax = t < pooled * gh ? 0 : 1;
            const int u = ax ? t - pooled * gh : t;
            const int g = ax ? gw : gh;
            const int p = u / g, i = u - p * g;
            int lo = 0, hi = 0;
            float l = 0.f, h = 0.f;
            const bool ok = ax ? roi_sample(sw, bw, p, i, g, W, lo, hi, l, h) : roi_sample(sh, bh, p, i, g, H, lo, hi, l, h);
            t_lo[ax][u] = ok ? lo : -1;
            t_hi[ax][u] = hi;
            t_l[ax][u] = l;
            t_h[ax][u] = h;
        }
        __syncthreads();
    }

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbins = pooled * pooled;
    const int g0 = part * 4 + wave, gstride = 4 * parts;
#define TD_RA(f) acc.f = __fadd_rn(acc.f, __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1.f), __fmul_rn(w2, v2.f)), __fmul_rn(w3, v3.f)), __fmul_rn(w4, v4.f)))
    if (tab && ghw > 0) {
        typedef typename Raw4<T>::type R4;
        const int nb_w = g0 < nbins ? (nbins - 1 - g0) / gstride + 1 : 0;     // bins of this wave
        const int total = nb_w * ghw;                                         // its samples, bin after bin
        for (int cb = 0; cb < C; cb += 256) {
            const int c0 = cb + lane * 4;
            const bool act = c0 < C;
            // issue cursor (sample s + D) and consume cursor (sample s): both walk bin → iy → ix
            int i_bin = g0, i_iy = 0, i_ix = 0, i_ph = g0 / pooled, i_pw = g0 - (g0 / pooled) * pooled;
            int c_bin = g0, c_n = 0;
            R4 v[D][4];
            float wgt[D][4];
            int ok[D];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c0l = act ? c0 : 0;                 // idle lanes (C < 256) load a valid address and drop the value
            const int nb_all = nbins;
            auto issue = [&](int d) {
                // past the wave's last sample the cursor keeps walking (at most D - 1 dummy samples: clamped bin, never consumed)
                const int ph = i_bin < nb_all ? i_ph : 0, pw = i_bin < nb_all ? i_pw : 0;
                const int yi = ph * gh + i_iy, xi = pw * gw + i_ix;
                // every lane reads the same table entry: the indices go to scalar registers (scalar address arithmetic), the
                // weights stay per lane
                const int yl = __builtin_amdgcn_readfirstlane(t_lo[0][yi]), xl = __builtin_amdgcn_readfirstlane(t_lo[1][xi]);
                const int yh = __builtin_amdgcn_readfirstlane(t_hi[0][yi]), xh = __builtin_amdgcn_readfirstlane(t_hi[1][xi]);
                ok[d] = yl >= 0 && xl >= 0;
                const float ly = t_l[0][yi], hy = t_h[0][yi], lx = t_l[1][xi], hx = t_h[1][xi];
                wgt[d][0] = __fmul_rn(hy, hx);
                wgt[d][1] = __fmul_rn(hy, lx);
                wgt[d][2] = __fmul_rn(ly, hx);
                wgt[d][3] = __fmul_rn(ly, lx);
                // the loads are UNCONDITIONAL (a sample outside the map reads pixel (0, 0) and is dropped at the add): with a
                // branch around them the compiler cannot count what is in flight and drains vmcnt to 0 at every add
                const int syl = ok[d] ? yl : 0, syh = ok[d] ? yh : 0, sxl = ok[d] ? xl : 0, sxh = ok[d] ? xh : 0;
                v[d][0] = *reinterpret_cast<const R4*>(feat + ((size_t)syl * W + sxl) * C + c0l);
                v[d][1] = *reinterpret_cast<const R4*>(feat + ((size_t)syl * W + sxh) * C + c0l);
                v[d][2] = *reinterpret_cast<const R4*>(feat + ((size_t)syh * W + sxl) * C + c0l);
                v[d][3] = *reinterpret_cast<const R4*>(feat + ((size_t)syh * W + sxh) * C + c0l);
                if (++i_ix == gw) {
                    i_ix = 0;
                    if (++i_iy == gh) {
                        i_iy = 0;
                        i_bin += gstride;
                        i_ph = i_bin / pooled;
                        i_pw = i_bin - i_ph * pooled;
                    }
                }
            };
            auto consume = [&](int d) {
                if (ok[d]) {
                    const float w1 = wgt[d][0], w2 = wgt[d][1], w3 = wgt[d][2], w4 = wgt[d][3];
                    const float4 v1 = widen4(v[d][0]), v2 = widen4(v[d][1]), v3 = widen4(v[d][2]), v4 = widen4(v[d][3]);
                    TD_RA(x); TD_RA(y); TD_RA(z); TD_RA(w);
                }
                if (++c_n == ghw) {
                    if (act) {
                        acc.x = __fdiv_rn(acc.x, count);
                        acc.y = __fdiv_rn(acc.y, count);
                        acc.z = __fdiv_rn(acc.z, count);
                        acc.w = __fdiv_rn(acc.w, count);
                        store4(out + (row * nbins + c_bin) * C + c0, acc);
                    }
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    c_n = 0;
                    c_bin += gstride;
                }
            };
            if (total > 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) issue(d);
                for (int s0 = 0; s0 < total; s0 += D) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if (s0 + d < total) consume(d);
                        issue(d);
                    }
                }
            }
        }
#undef TD_RA
        return;
    }
    // degenerate RoIs (no samples: every bin is 0) and sampling grids too large for the table: one sample at a time
    for (int bin = g0; bin < nbins; bin += gstride) {
        const int ph = bin / pooled, pw = bin - ph * pooled;
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int iy = 0; iy < gh; ++iy) {
                int yl, yh;
                float ly, hy;
                if (!roi_sample(sh, bh, ph, iy, gh, H, yl, yh, ly, hy)) continue;
                for (int ix = 0; ix < gw; ++ix) {
                    int xl, xh;
                    float lx, hx;
                    if (!roi_sample(sw, bw, pw, ix, gw, W, xl, xh, lx, hx)) continue;
                    const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                    const float4 v1 = load4(feat + ((size_t)yl * W + xl) * C + c0);
                    const float4 v2 = load4(feat + ((size_t)yl * W + xh) * C + c0);
                    const float4 v3 = load4(feat + ((size_t)yh * W + xl) * C + c0);
                    const float4 v4 = load4(feat + ((size_t)yh * W + xh) * C + c0);
#define TD_RA(f) acc.f = __fadd_rn(acc.f, __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1.f), __fmul_rn(w2, v2.f)), __fmul_rn(w3, v3.f)), __fmul_rn(w4, v4.f)))
                    TD_RA(x); TD_RA(y); TD_RA(z); TD_RA(w);
#undef TD_RA
                }
            }
            acc.x = __fdiv_rn(acc.x, count);
            acc.y = __fdiv_rn(acc.y, count);
            acc.z = __fdiv_rn(acc.z, count);
            acc.w = __fdiv_rn(acc.w, count);
            store4(out + (row * nbins + bin) * C + c0, acc);
        }
    }
}

// fp16 features with C <= 256: a channel row is 512 B, so with 4 channels (8 B) per lane a wave-load moves half of what
// the texture path handles per instruction (PMC, box head: 8.4 M loads = 140 M TCP accesses, TA busy 0.53, 137
// instructions per sample of which 65 scalar). Here a lane owns 8 channels (16-B loads) and the two halves of a wave work
// on two NEIGHBOURING BINS of the RoI at once (bin b on lanes 0-31, bin b + 1 on lanes 32-63): both bins have the same
// gh x gw sample grid, so the halves run in lockstep through (iy, ix) and differ only in their table entries — half the
// load instructions, half the per-sample bookkeeping. Tables are packed {lo, hi, l, h} (one ds_read_b128 per axis);
// addresses are 32-bit element offsets from the level's base. Per channel the arithmetic and its order are those of
// roi_align_kernel → bit-identical outputs.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int D>
__global__ __launch_bounds__(256) void roi_align_h8_kernel(FeatLevels fl, const float* __restrict__ rois,
                                                           const int* __restrict__ counts, int items, int roi_stride,
                                                           int pooled, int compact, _Float16* __restrict__ out,
                                                           int* __restrict__ total_rows, int single_level, int parts) {
    const int item = blockIdx.y, r = blockIdx.x / parts, part = blockIdx.x - r * parts;
    RoiGeo geo;
    if (!roi_setup(fl, rois, counts, items, roi_stride, pooled, compact, total_rows, single_level, item, r, geo)) return;
    const size_t row = geo.row;
    const int H = geo.H, W = geo.W, C = fl.C, gh = geo.gh, gw = geo.gw, ghw = geo.ghw;
    const float sh = geo.sh, sw = geo.sw, bh = geo.bh, bw = geo.bw, count = geo.count;
    const _Float16* __restrict__ feat = static_cast<const _Float16*>(fl.feat[geo.lvl]) + geo.feat_off;
    const int nbins = pooled * pooled;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hl = lane >> 5, c0 = (lane & 31) * 8;
    const bool act = c0 < C;

    constexpr int TAB = 14 * 24;
    __shared__ int4 tq[2][TAB];      // {lo (-1: sample outside), hi, bits(l), bits(h)} per y-sample / x-sample
    const bool tab = pooled * gh <= TAB && pooled * gw <= TAB;
    if (!tab || ghw == 0) {
        // no samples (every bin is 0) or a sampling grid too large for the table: one sample at a time, 8 channels per lane,
        // the two halves of a wave on two bins
        for (int bin = (part * 4 + wave) * 2 + hl; bin < nbins; bin += 8 * parts) {
            const int ph = bin / pooled, pw = bin - ph * pooled;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                int yl, yh;
                float ly, hy;
                if (!roi_sample(sh, bh, ph, iy, gh, H, yl, yh, ly, hy)) continue;
                for (int ix = 0; ix < gw; ++ix) {
                    int xl, xh;
                    float lx, hx;
                    if (!roi_sample(sw, bw, pw, ix, gw, W, xl, xh, lx, hx)) continue;
                    if (!act) continue;
                    const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                    const h8 v1 = *reinterpret_cast<const h8*>(feat + ((size_t)yl * W + xl) * C + c0);
                    const h8 v2 = *reinterpret_cast<const h8*>(feat + ((size_t)yl * W + xh) * C + c0);
                    const h8 v3 = *reinterpret_cast<const h8*>(feat + ((size_t)yh * W + xl) * C + c0);
                    const h8 v4 = *reinterpret_cast<const h8*>(feat + ((size_t)yh * W + xh) * C + c0);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[e] = __fadd_rn(acc[e], __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, (float)v1[e]), __fmul_rn(w2, (float)v2[e])),
                                                                         __fmul_rn(w3, (float)v3[e])), __fmul_rn(w4, (float)v4[e])));
                }
            }
            if (act) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)__fdiv_rn(acc[e], count);
                *reinterpret_cast<h8*>(out + (row * nbins + bin) * C + c0) = o;
            }
        }
        return;
    }
    for (int t = threadIdx.x; t < pooled * gh + pooled * gw; t += blockDim.x) {
        const int ax = t < pooled * gh ? 0 : 1;
        const int u = ax ? t - pooled * gh : t;
        const int g = ax ? gw : gh;
        const int p = u / g, i = u - p * g;
        int lo = 0, hi = 0;
        float l = 0.f, h = 0.f;
        const bool ok = ax ? roi_sample(sw, bw, p, i, g, W, lo, hi, l, h) : roi_sample(sh, bh, p, i, g, H, lo, hi, l, h);
        tq[ax][u] = make_int4(ok ? lo : -1, hi, __float_as_int(l), __float_as_int(h));
    }
    __syncthreads();

    // bins of the two halves: b0 + hl, b0 = 2 (4 part + wave), advancing by 8 parts; a half past the last bin idles
    const int b_first = (part * 4 + wave) * 2, bstride = 8 * parts;
    const int nb0 = b_first < nbins ? (nbins - 1 - b_first) / bstride + 1 : 0;     // rounds of this wave (half 0 has the most)
    const int total = nb0 * ghw;
    int i_b0 = b_first, i_iy = 0, i_ix = 0;         // issue cursor (uniform; the bin of a lane is i_b0 + hl)
    int c_b0 = b_first, c_n = 0;                    // consume cursor
    h8 v[D][4];
    float wgt[D][4];
    bool okl[D];
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const unsigned uW = (unsigned)W, uC = (unsigned)C;
    const unsigned c0l = act ? (unsigned)c0 : 0u;     // idle lanes (C < 256) load a valid address and drop the value
    auto issue = [&](int d) {
        // past the wave's last sample the cursor keeps walking (at most D - 1 dummy samples: clamped bin, never consumed)
        const int bin = i_b0 + hl;
        const bool live = bin < nbins && act;
        const int binc = bin < nbins ? bin : nbins - 1;
        const int ph = binc / pooled, pw = binc - ph * pooled;
        const int4 ey = tq[0][ph * gh + i_iy], ex = tq[1][pw * gw + i_ix];
        okl[d] = live && ey.x >= 0 && ex.x >= 0;
        const float ly = __int_as_float(ey.z), hy = __int_as_float(ey.w), lx = __int_as_float(ex.z), hx = __int_as_float(ex.w);
        wgt[d][0] = __fmul_rn(hy, hx);
        wgt[d][1] = __fmul_rn(hy, lx);
        wgt[d][2] = __fmul_rn(ly, hx);
        wgt[d][3] = __fmul_rn(ly, lx);
        // UNCONDITIONAL loads (a sample outside the map reads pixel (0, 0) and is dropped at the add): with a branch around
        // them the compiler cannot count what is in flight and drains vmcnt to 0 at every add
        const bool in = ey.x >= 0 && ex.x >= 0;
        const unsigned yl = in ? (unsigned)ey.x : 0u, yh = in ? (unsigned)ey.y : 0u, xl = in ? (unsigned)ex.x : 0u, xh = in ? (unsigned)ex.y : 0u;
        v[d][0] = *reinterpret_cast<const h8*>(feat + ((yl * uW + xl) * uC + c0l));
        v[d][1] = *reinterpret_cast<const h8*>(feat + ((yl * uW + xh) * uC + c0l));
        v[d][2] = *reinterpret_cast<const h8*>(feat + ((yh * uW + xl) * uC + c0l));
        v[d][3] = *reinterpret_cast<const h8*>(feat + ((yh * uW + xh) * uC + c0l));
        if (++i_ix == gw) {
            i_ix = 0;
            if (++i_iy == gh) {
                i_iy = 0;
                i_b0 += bstride;
            }
        }
    };
    auto consume = [&](int d) {
        if (okl[d]) {
            const float w1 = wgt[d][0], w2 = wgt[d][1], w3 = wgt[d][2], w4 = wgt[d][3];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[e] = __fadd_rn(acc[e], __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, (float)v[d][0][e]), __fmul_rn(w2, (float)v[d][1][e])),
                                                                 __fmul_rn(w3, (float)v[d][2][e])), __fmul_rn(w4, (float)v[d][3][e])));
        }
        if (++c_n == ghw) {
            const int bin = c_b0 + hl;
            if (act && bin < nbins) {
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)__fdiv_rn(acc[e], count);
                *reinterpret_cast<h8*>(out + (row * nbins + bin) * C + c0) = o;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            c_n = 0;
            c_b0 += bstride;
        }
    };
    if (total > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d);
        for (int s0 = 0; s0 < total; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (s0 + d < total) consume(d);
                issue(d);
            }
        }
    }
}

// softmax (2 logits, foreground first) + apply_deltas (10,10,5,5) + clip + score > thresh
__global__ void det_decode_kernel(const float* __restrict__ cls_reg, int cr_stride, const float* __restrict__ props,
                                  const int* __restrict__ prop_count, ImgSizes valid, int B, int prop_stride,
                                  float score_thresh, float* __restrict__ boxes, float* __restrict__ scores,
                                  int* __restrict__ flags) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * prop_stride) return;
    const int b = idx / prop_stride, r = idx - b * prop_stride;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, score = 0.f;
    int ok = 0;
    if (r < prop_count[b]) {
        const float* cr = cls_reg + (size_t)idx * cr_stride;
        const float l0 = cr[0], l1 = cr[1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(__fsub_rn(l0, m)), e1 = expf(__fsub_rn(l1, m));
        score = __fdiv_rn(e0, __fadd_rn(e0, e1));
        const float4 p = *reinterpret_cast<const float4*>(props + (size_t)idx * 4);
        decode_box(p.x, p.y, p.z, p.w, cr[2], cr[3], cr[4], cr[5], 10.f, 10.f, 5.f, 5.f, x1, y1, x2, y2);
        const bool fin = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(score) &&
                         isfinite(__fsub_rn(1.f, score));
        const float ih = (float)valid.h[b], iw = (float)valid.w[b];
        x1 = fminf(fmaxf(x1, 0.f), iw);
        y1 = fminf(fmaxf(y1, 0.f), ih);
        x2 = fminf(fmaxf(x2, 0.f), iw);
        y2 = fminf(fmaxf(y2, 0.f), ih);
        ok = fin && score > score_thresh;
    }
    *reinterpret_cast<float4*>(boxes + (size_t)idx * 4) = make_float4(x1, y1, x2, y2);
    scores[idx] = score;
    flags[idx] = ok;
}

// detector_postprocess for the kept detections of one image (block per image, serial over <= max_det rows)
__global__ void det_finalize_kernel(const float* __restrict__ sboxes, const float* __restrict__ sscores,
                                    const int* __restrict__ keep_pos, const int* __restrict__ keep_count,
                                    ImgSizes valid, ImgSizes outsz, int stride_items, int max_det,
                                    float* __restrict__ det_boxes_net, float* __restrict__ out_boxes,
                                    float* __restrict__ out_scores, int* __restrict__ out_classes,
                                    int* __restrict__ out_count) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int n = keep_count[b];
    n = n < max_det ? n : max_det;
    const float sx = (float)((double)outsz.w[b] / (double)valid.w[b]);
    const float sy = (float)((double)outsz.h[b] / (double)valid.h[b]);
    const float ow = (float)outsz.w[b], oh = (float)outsz.h[b];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const int pos = keep_pos[(size_t)b * max_det + i];
        const float4 bx = *reinterpret_cast<const float4*>(sboxes + ((size_t)b * stride_items + pos) * 4);
        float x1 = fminf(fmaxf(__fmul_rn(bx.x, sx), 0.f), ow), y1 = fminf(fmaxf(__fmul_rn(bx.y, sy), 0.f), oh);
        float x2 = fminf(fmaxf(__fmul_rn(bx.z, sx), 0.f), ow), y2 = fminf(fmaxf(__fmul_rn(bx.w, sy), 0.f), oh);
        if (__fsub_rn(x2, x1) > 0.f && __fsub_rn(y2, y1) > 0.f) {
            const size_t o = (size_t)b * max_det + m;
            *reinterpret_cast<float4*>(det_boxes_net + o * 4) = bx;
            *reinterpret_cast<float4*>(out_boxes + o * 4) = make_float4(x1, y1, x2, y2);
            out_scores[o] = sscores[(size_t)b * stride_items + pos];
            out_classes[o] = 0;
            ++m;
        }
    }
    for (int i = m; i < max_det; ++i) {
        const size_t o = (size_t)b * max_det + i;
        *reinterpret_cast<float4*>(det_boxes_net + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(out_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        out_scores[o] = 0.f;
        out_classes[o] = 0;
    }
    out_count[b] = m;
}

// mask predictor: one wave per pixel, dot over C channels, + bias, sigmoid
template <typename T>
__global__ __launch_bounds__(256) void mask_predict_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           float bias, int C, int rows_max, const int* __restrict__ rows_dyn,
                                                           int rows_mul, float* __restrict__ logits,
                                                           float* __restrict__ probs) {
    int rows = rows_max;
    if (rows_dyn) {
        const int rd = *rows_dyn * rows_mul;
        rows = rd < rows ? rd : rows;
    }
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = load4(x + (size_t)row * C + c);
        const float4 k = *reinterpret_cast<const float4*>(w + c);
        acc = __fmaf_rn(v.x, k.x, acc);
        acc = __fmaf_rn(v.y, k.y, acc);
        acc = __fmaf_rn(v.z, k.z, acc);
        acc = __fmaf_rn(v.w, k.w, acc);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        const float l = __fadd_rn(acc, bias);
        if (logits) logits[row] = l;
        probs[row] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-l)));
    }
}

__global__ void mask_scatter_kernel(const float* __restrict__ compact, const int* __restrict__ counts, int B, int D,
                                    float* __restrict__ out) {
    const int b = blockIdx.y, d = blockIdx.x;
    int prefix = 0;
    for (int i = 0; i < b; ++i) prefix += counts[i];
    float* o = out + ((size_t)b * D + d) * (TD_MASK_SIDE * TD_MASK_SIDE);
    const bool live = d < counts[b];
    const float* s = compact + (size_t)(prefix + d) * (TD_MASK_SIDE * TD_MASK_SIDE);
    for (int i = threadIdx.x; i < TD_MASK_SIDE * TD_MASK_SIDE; i += blockDim.x) o[i] = live ? s[i] : 0.f;
}

// paste step 1: integer regions + word offsets per image (CPU-path region: floor(x0)-1 .. ceil(x1)+1, clamped)
__global__ void paste_plan_kernel(const float* __restrict__ boxes, const int* __restrict__ counts, ImgSizes outsz, int D,
                                  int* __restrict__ region, long long* __restrict__ offset, long long words_cap) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int n = counts[b];
    long long off = 0;
    for (int d = 0; d < D; ++d) {
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        long long o = 0;
        if (d < n) {
            const float4 bx = *reinterpret_cast<const float4*>(boxes + ((size_t)b * D + d) * 4);
            x0 = (int)fmaxf(__fsub_rn(floorf(bx.x), 1.f), 0.f);
            y0 = (int)fmaxf(__fsub_rn(floorf(bx.y), 1.f), 0.f);
            x1 = (int)fminf(__fadd_rn(ceilf(bx.z), 1.f), (float)outsz.w[b]);
            y1 = (int)fminf(__fadd_rn(ceilf(bx.w), 1.f), (float)outsz.h[b]);
            if (x1 < x0) x1 = x0;
            if (y1 < y0) y1 = y0;
            const long long sz = (long long)((x1 - x0 + 31) >> 5) * (y1 - y0);
            if (off + sz > words_cap) { x1 = x0; y1 = y0; }   // out of room: empty region (caller sized the buffer)
            else { o = off; off += sz; }
        }
        int* rg = region + ((size_t)b * D + d) * 4;
        rg[0] = x0; rg[1] = y0; rg[2] = x1; rg[3] = y1;
        offset[(size_t)b * D + d] = o;
    }
}

// paste step 2: one block per detection; thread per 32-pixel word
__global__ __launch_bounds__(256) void paste_fill_kernel(const float* __restrict__ probs, const float* __restrict__ boxes,
                                                         const int* __restrict__ counts, int D, float thresh,
                                                         const int* __restrict__ region,
                                                         const long long* __restrict__ offset, uint32_t* __restrict__ bits,
                                                         long long words_per_image) {
    const int b = blockIdx.y, d = blockIdx.x;
    if (d >= counts[b]) return;
    constexpr int MS = TD_MASK_SIDE;
    __shared__ float m[MS * MS];
    const float* src = probs + ((size_t)b * D + d) * (MS * MS);
    for (int i = threadIdx.x; i < MS * MS; i += blockDim.x) m[i] = src[i];
    __syncthreads();
    const int* rg = region + ((size_t)b * D + d) * 4;
    const int x0 = rg[0], y0 = rg[1], x1 = rg[2], y1 = rg[3];
    const int wpr = (x1 - x0 + 31) >> 5, rows = y1 - y0;
    if (wpr <= 0 || rows <= 0) return;
    const float4 bx = *reinterpret_cast<const float4*>(boxes + ((size_t)b * D + d) * 4);
    const float bw = __fsub_rn(bx.z, bx.x), bh = __fsub_rn(bx.w, bx.y);
    uint32_t* out = bits + (size_t)b * words_per_image + offset[(size_t)b * D + d];
    for (int wi = threadIdx.x; wi < wpr * rows; wi += blockDim.x) {
        const int ry = wi / wpr, wx = wi - ry * wpr;
        const float py = __fadd_rn((float)(y0 + ry), 0.5f);
        const float gy = __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(py, bx.y), bh), 2.f), 1.f);
        const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)MS), 1.f), 2.f);
        const float iy0f = floorf(iy);
        const float wn = __fsub_rn(__fadd_rn(iy0f, 1.f), iy), ws = __fsub_rn(iy, iy0f);
        const int iy0 = (int)iy0f, iy1 = iy0 + 1;
        const bool yok0 = iy0 >= 0 && iy0 < MS, yok1 = iy1 >= 0 && iy1 < MS;
        uint32_t word = 0u;
        for (int bit = 0; bit < 32; ++bit) {
            const int x = x0 + wx * 32 + bit;
            if (x >= x1) break;
            const float px = __fadd_rn((float)x, 0.5f);
            const float gx = __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(px, bx.x), bw), 2.f), 1.f);
            const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)MS), 1.f), 2.f);
            const float ix0f = floorf(ix);
            const float ww = __fsub_rn(__fadd_rn(ix0f, 1.f), ix), we = __fsub_rn(ix, ix0f);
            float v = 0.f;
            if (isfinite(ix) && isfinite(iy)) {
                const int ix0 = (int)ix0f, ix1 = ix0 + 1;
                const bool xok0 = ix0 >= 0 && ix0 < MS, xok1 = ix1 >= 0 && ix1 < MS;
                const float nw = (yok0 && xok0) ? m[iy0 * MS + ix0] : 0.f;
                const float ne = (yok0 && xok1) ? m[iy0 * MS + ix1] : 0.f;
                const float sw = (yok1 && xok0) ? m[iy1 * MS + ix0] : 0.f;
                const float se = (yok1 && xok1) ? m[iy1 * MS + ix1] : 0.f;
                v = __fmul_rn(nw, __fmul_rn(wn, ww));
                v = __fadd_rn(v, __fmul_rn(ne, __fmul_rn(wn, we)));
                v = __fadd_rn(v, __fmul_rn(sw, __fmul_rn(ws, ww)));
                v = __fadd_rn(v, __fmul_rn(se, __fmul_rn(ws, we)));
            }
            if (v >= thresh) word |= 1u << bit;
        }
        out[(size_t)ry * wpr + wx] = word;
    }
}

}  // namespace

// ring depth of roi_align_kernel (samples in flight per wave); TD_ROI_DEPTH overrides it for diagnostics (tools/roi_bench.py)
static bool roi_h8_ok(int C) {
    static const char* env = getenv("TD_ROI_H8");      // diagnostics: 0 = the 4-channel-per-lane kernel for fp16 too
    if (env && atoi(env) == 0) return false;
    return C <= 256 && C % 8 == 0;
}

template <typename T>
static void roi_align_dispatch(dim3 grid, hipStream_t stream, int depth, const FeatLevels& fl, const float* rois, const int* counts,
                               int items, int roi_stride, int pooled, int compact, T* out, int* total_rows, int single_level, int parts) {
    if constexpr (std::is_same<T, _Float16>::value) {
        if (roi_h8_ok(fl.C)) {
#define TD_ROI_LAUNCH8(DD) hipLaunchKernelGGL((roi_align_h8_kernel<DD>), grid, dim3(256), 0, stream, fl, rois, counts, items, \
                                              roi_stride, pooled, compact, out, total_rows, single_level, parts)
            switch (depth) {
                case 1: TD_ROI_LAUNCH8(1); break;
                case 3: TD_ROI_LAUNCH8(3); break;
                case 4: TD_ROI_LAUNCH8(4); break;
                default: TD_ROI_LAUNCH8(2); break;
            }
#undef TD_ROI_LAUNCH8
            return;
        }
    }
#define TD_ROI_LAUNCH(DD) hipLaunchKernelGGL((roi_align_kernel<T, DD>), grid, dim3(256), 0, stream, fl, rois, counts, items, \
                                             roi_stride, pooled, compact, out, total_rows, single_level, parts)
    switch (depth) {
        case 1: TD_ROI_LAUNCH(1); break;
        case 2: TD_ROI_LAUNCH(2); break;
        case 3: TD_ROI_LAUNCH(3); break;
        case 6: TD_ROI_LAUNCH(6); break;
        case 8: TD_ROI_LAUNCH(8); break;
        default: TD_ROI_LAUNCH(4); break;
    }
#undef TD_ROI_LAUNCH
}

static int roi_depth(int precision) {
    static const char* env = getenv("TD_ROI_DEPTH");
    if (env) return atoi(env);
    return precision == TD_PRECISION_FP16 ? 2 : 4;        // measured (tools/roi_sweep.sh): fp16 0.417 / 0.351 / 0.353 / 0.362 ms at depth 1 / 2 / 3 / 4, fp32 0.659 / 0.581 / 0.575 / 0.545 / 0.550 at 1 / 2 / 3 / 4 / 6
}

// blocks per RoI: the 14 x 14 mask-head launch has few RoIs (the detections) of 196 bins each
static int roi_parts(int pooled) {
    static const char* env = getenv("TD_ROI_PARTS");
    if (env) return atoi(env) > 0 ? atoi(env) : 1;
    return pooled * pooled >= 128 ? 4 : 1;
}

td_status roi_align_launch(const FeatLevels& fl, const float* rois, const int* counts, int items, int roi_stride,
                           int pooled, int compact, void* out, int* total_rows, int precision, hipStream_t stream) {
    TD_REQUIRE(fl.C % 4 == 0, "roi_align: C must be a multiple of 4");
    const int parts = roi_parts(pooled);
    const dim3 grid(roi_stride * parts, items);
    if (precision == TD_PRECISION_FP16)
        roi_align_dispatch<_Float16>(grid, stream, roi_depth(precision), fl, rois, counts, items, roi_stride, pooled, compact,
                                     static_cast<_Float16*>(out), total_rows, 0, parts);
    else
        roi_align_dispatch<float>(grid, stream, roi_depth(precision), fl, rois, counts, items, roi_stride, pooled, compact,
                                  static_cast<float*>(out), total_rows, 0, parts);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status roi_align_single_launch(const void* feat, int H, int W, int C, const float* rois, int R, float scale,
                                  int pooled, void* out, int precision, hipStream_t stream) {
    TD_REQUIRE(C % 4 == 0 && R >= 1, "roi_align: bad shape");
    FeatLevels fl{};
    fl.feat[0] = feat; fl.h[0] = H; fl.w[0] = W; fl.scale[0] = scale; fl.C = C;
    const int parts = roi_parts(pooled);
    const dim3 grid(R * parts, 1);
    if (precision == TD_PRECISION_FP16)
        roi_align_dispatch<_Float16>(grid, stream, roi_depth(precision), fl, rois, nullptr, 1, R, pooled, 0,
                                     static_cast<_Float16*>(out), nullptr, 1, parts);
    else
        roi_align_dispatch<float>(grid, stream, roi_depth(precision), fl, rois, nullptr, 1, R, pooled, 0,
                                  static_cast<float*>(out), nullptr, 1, parts);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status det_decode_launch(const float* cls_reg, int cr_stride, const float* props, const int* prop_count,
                            const ImgSizes& valid, int B, int prop_stride, float score_thresh, float* boxes,
                            float* scores, int* flags, hipStream_t stream) {
    hipLaunchKernelGGL(det_decode_kernel, dim3(td_cdiv(B * prop_stride, 256)), dim3(256), 0, stream, cls_reg, cr_stride,
                       props, prop_count, valid, B, prop_stride, score_thresh, boxes, scores, flags);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status det_finalize_launch(const float* sboxes, const float* sscores, const int* keep_pos, const int* keep_count,
                              const ImgSizes& valid, const ImgSizes& outsz, int B, int stride_items, int max_det,
                              float* det_boxes_net, float* out_boxes, float* out_scores, int* out_classes,
                              int* out_count, hipStream_t stream) {
    hipLaunchKernelGGL(det_finalize_kernel, dim3(B), dim3(64), 0, stream, sboxes, sscores, keep_pos, keep_count, valid,
                       outsz, stride_items, max_det, det_boxes_net, out_boxes, out_scores, out_classes, out_count);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status mask_predict_launch(const void* x, const float* w, float bias, int C, int rows_max, const int* rows_dyn,
                              int rows_mul, float* logits_out, float* probs_out, int precision, hipStream_t stream) {
    TD_REQUIRE(C % 4 == 0, "mask_predict: C must be a multiple of 4");
    if (rows_max <= 0) return TD_OK;
    if (precision == TD_PRECISION_FP16)
        hipLaunchKernelGGL((mask_predict_kernel<_Float16>), dim3(td_cdiv(rows_max, 4)), dim3(256), 0, stream,
                           static_cast<const _Float16*>(x), w, bias, C, rows_max, rows_dyn, rows_mul, logits_out, probs_out);
    else
        hipLaunchKernelGGL((mask_predict_kernel<float>), dim3(td_cdiv(rows_max, 4)), dim3(256), 0, stream,
                           static_cast<const float*>(x), w, bias, C, rows_max, rows_dyn, rows_mul, logits_out, probs_out);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status mask_scatter_launch(const float* compact, const int* counts, int B, int D, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(mask_scatter_kernel, dim3(D, B), dim3(256), 0, stream, compact, counts, B, D, out);
    TD_KERNEL_CHECK();
    return TD_OK;
}

td_status paste_masks_launch(const float* probs, const float* boxes, const int* counts, const ImgSizes& outsz, int B,
                             int D, float thresh, int* region, long long* offset, uint32_t* bits,
                             long long words_per_image, hipStream_t stream) {
    hipLaunchKernelGGL(paste_plan_kernel, dim3(B), dim3(64), 0, stream, boxes, counts, outsz, D, region, offset,
                       words_per_image);
    TD_KERNEL_CHECK();
    hipLaunchKernelGGL(paste_fill_kernel, dim3(D, B), dim3(256), 0, stream, probs, boxes, counts, D, thresh, region,
                       offset, bits, words_per_image);
    TD_KERNEL_CHECK();
    return TD_OK;
}
