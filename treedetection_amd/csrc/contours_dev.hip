// Border following on the device for the packed instance masks of a batch — the GPU counterpart of contours.cpp
// (reference TreeDetection/prediction.py:232-236: cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE) per instance;
// SURVEY.md §8f rank 1 "contour tracing on GPU").
//
// One wave per detection. The detection's paste region is unpacked into a padded int16 label image in LDS; the raster
// scan of Suzuki & Abe runs 64 pixels per step (a ballot finds the next pixel that can start a border, everything
// else only moves LNBD), lane 0 follows each border exactly like the host tracer (same neighbour search, same
// marks), and once the contour list of the detection is known the lanes re-trace the contours in parallel — one lane
// per contour, on the same label image, where only "non-zero" matters — and write the CHAIN_APPROX_SIMPLE points into
// a block of the image's point buffer obtained with one atomicAdd. Contours are reported in RETR_TREE order.
// Detections whose region does not fit the LDS image, with more than TD_CONTOUR_MAX contours, or whose points do not
// fit the buffer are flagged (status != 0) and left to the host tracer. Every loop is bounded by the region size.
#include "common.h"

namespace {

constexpr int CMAX = 256;                // contours per detection handled on the device (TD_CONTOUR_MAX)
constexpr int LABEL_SMALL = 6 * 1024;    // int16 label images in LDS: 12 KB for crowns up to ~75 x 75 px (most of them),
constexpr int LABEL_LARGE = 28 * 1024;   // 56 KB for regions up to (h+2)*(w+2) <= 28672; two launches, one per size class

struct TraceArgs {
    const int32_t* region;      // [B][D][4]
    const long long* offset;    // [B][D]
    const uint32_t* bits;       // [B][words]
    const int32_t* counts;      // [B]
    int D;
    long long words_per_image;
    int16_t* pts;               // [B][pts_cap][2]  (x, y) in tile pixels
    int pts_cap;
    int32_t* img_pts;           // [B] points allocated so far (zeroed by the launcher)
    int32_t* det_info;          // [B][D][4]  status, contour count, point base, total points
    int32_t* cont_info;         // [B][D][CMAX][2]  in RETR_TREE order: (point offset inside the detection's block, points)
};

// One border: the 8-neighbour following of OpenCV's icvFetchContour as restated in contours.cpp. MARK writes the
// Suzuki labels (scan pass), EMIT writes the CHAIN_APPROX_SIMPLE points (re-trace pass). Returns the point count, or -1
// when the step guard trips (cannot happen on a consistent image; keeps the loop provably finite).
template <bool MARK, bool EMIT>
__device__ int follow(int16_t* F, int step, const int* deltas, int i0, int x, int y, bool hole, int nbd, int guard_max,
                      int16_t* out, int x_off, int y_off) {
    const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int px = x, py = y, n = 0;
    int s_end = hole ? 0 : 4, s = s_end, i1;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (F[i1] == 0 && s != s_end);
    if (s == s_end) {                    // single pixel
        if (MARK) F[i0] = (int16_t)-nbd;
        if (EMIT) { out[0] = (int16_t)(px + x_off); out[1] = (int16_t)(py + y_off); }
        return 1;
    }
    int i3 = i0, prev_s = s ^ 4;
    for (int guard = 0; guard < guard_max; ++guard) {
        s_end = s;
        int i4 = i3;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (F[i4] != 0) break;
        }
        s &= 7;
        if (MARK) {
            if ((unsigned)(s - 1) < (unsigned)s_end) F[i3] = (int16_t)-nbd;      // east neighbour examined as 0
            else if (F[i3] == 1) F[i3] = (int16_t)nbd;
        }
        if (s != prev_s) {
            if (EMIT) { out[2 * n] = (int16_t)(px + x_off); out[2 * n + 1] = (int16_t)(py + y_off); }
            ++n;
            prev_s = s;
        }
        px += dx[s];
        py += dy[s];
        if (i4 == i0 && i3 == i1) return n;
        i3 = i4;
        s = (s + 4) & 7;
    }
    return -1;
}

template <int LABEL_ELEMS, int LABEL_MIN>
__global__ __launch_bounds__(64) void contour_trace_kernel(const TraceArgs A) {
    const int d = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    int32_t* info = A.det_info + ((size_t)b * A.D + d) * 4;
    if (d >= A.counts[b]) {
        if (lane == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }
        return;
    }
    const int32_t* rg = A.region + ((size_t)b * A.D + d) * 4;
    const int x0 = rg[0], y0 = rg[1], w = rg[2] - rg[0], h = rg[3] - rg[1];
    if (w <= 0 || h <= 0) {
        if (lane == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }
        return;
    }
    const int step = w + 2;
    const long long need = (long long)step * (h + 2);
    if (need <= LABEL_MIN) return;                   // the other size class handles (and reports) this detection
    if (need > LABEL_ELEMS) {
        if (LABEL_ELEMS == LABEL_LARGE && lane == 0) { info[0] = 1; info[1] = 0; info[2] = 0; info[3] = 0; }   // too large for LDS
        return;
    }
    __shared__ int16_t F[LABEL_ELEMS];
    __shared__ int16_t s_parent[CMAX + 2];
    __shared__ uint8_t s_hole[CMAX + 2];
    __shared__ int s_start[CMAX + 2];       // index into F of each contour's start pixel
    __shared__ int s_npts[CMAX + 2], s_ptoff[CMAX + 2];
    __shared__ int s_order[CMAX];
    __shared__ int s_misc[4];               // [0] contour count, [1] status, [2] point base
    const int total = step * (h + 2);
    for (int i = lane; i < total; i += 64) F[i] = 0;
    __syncthreads();
    const uint32_t* rows = A.bits + (size_t)b * A.words_per_image + A.offset[(size_t)b * A.D + d];
    const int wpr = (w + 31) / 32;
    for (int i = lane; i < w * h; i += 64) {
        const int yy = i / w, xx = i - yy * w;
        F[(yy + 1) * step + xx + 1] = (int16_t)((rows[(size_t)yy * wpr + (xx >> 5)] >> (xx & 31)) & 1u);
    }
    __syncthreads();
    int deltas[16];
    {
        const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
        const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
#pragma unroll
        for (int s = 0; s < 16; ++s) deltas[s] = dy[s & 7] * step + dx[s & 7];
    }
    const int guard_max = 8 * total + 16;
    if (lane == 0) {
        s_parent[1] = 0;
        s_hole[1] = 1;          // the frame
        s_misc[0] = 0;
        s_misc[1] = 0;
    }
    __syncthreads();
    // ---- raster scan: 64 pixels per step; only pixels that can start a border are handled one by one ------------
    int nbd = 1;                // wave-uniform copy kept in lane 0 and re-broadcast
    int status = 0;
    for (int y = 1; y <= h && status == 0; ++y) {
        int lnbd = 1;
        int x = 1;
        while (x <= w && status == 0) {
            const int xi = x + lane;
            int f = 0, left = 0, right = 0;
            if (xi <= w) {
                f = F[y * step + xi];
                left = F[y * step + xi - 1];
                right = F[y * step + xi + 1];
            }
            const bool cand = f != 0 && ((f == 1 && left == 0) || (f >= 1 && right == 0));
            const bool mark = f != 0 && f != 1;
            const unsigned long long cmask = __ballot(cand), mmask = __ballot(mark);
            if (cmask == 0ull) {
                if (mmask != 0ull) {
                    const int src = 63 - __clzll((long long)mmask);
                    const int v = __shfl(f, src);
                    lnbd = v < 0 ? -v : v;
                }
                x += 64;
                continue;
            }
            const int c = __ffsll((long long)cmask) - 1;
            const unsigned long long below = mmask & ((1ull << c) - 1ull);
            if (below != 0ull) {
                const int src = 63 - __clzll((long long)below);
                const int v = __shfl(f, src);
                lnbd = v < 0 ? -v : v;
            }
            const int xc = x + c;
            if (lane == 0) {
                const int i0 = y * step + xc;
                const int fv = F[i0];
                bool outer = false, hole = false;
                if (fv == 1 && F[i0 - 1] == 0) outer = true;
                else if (fv >= 1 && F[i0 + 1] == 0) hole = true;
                if (outer || hole) {
                    if (hole && fv > 1) lnbd = fv;
                    if (nbd - 1 >= CMAX) {
                        status = 2;                       // more contours than the device handles
                    } else {
                        ++nbd;
                        const bool ln_hole = s_hole[lnbd] != 0;
                        int parent = (ln_hole == hole) ? s_parent[lnbd] : lnbd;
                        if (parent == 0) parent = 1;
                        s_parent[nbd] = (int16_t)parent;
                        s_hole[nbd] = hole ? 1 : 0;
                        s_start[nbd] = i0;
                        const int n = follow<true, false>(F, step, deltas, i0, xc - 1, y - 1, hole, nbd, guard_max, nullptr, 0, 0);
                        if (n < 0) status = 3;
                        s_npts[nbd] = n;
                    }
                }
                const int fa = F[i0];
                if (fa != 1) lnbd = fa < 0 ? -fa : fa;
            }
            lnbd = __shfl(lnbd, 0);
            nbd = __shfl(nbd, 0);
            status = __shfl(status, 0);
            x = xc + 1;
        }
    }
    // ---- RETR_TREE order (pre-order, most recent sibling first), point offsets, allocation ----------------------
    if (lane == 0) {
        const int nc = nbd - 1;
        int total_pts = 0;
        if (status == 0) {
            int stack[CMAX], sp = 0, k = 0;
            for (int c = 2; c <= nbd; ++c)
                if (s_parent[c] == 1) stack[sp++] = c;
            while (sp > 0) {
                const int n = stack[--sp];
                s_order[k++] = n;
                for (int c = n + 1; c <= nbd; ++c)
                    if (s_parent[c] == n) stack[sp++] = c;
            }
            for (int i = 0; i < nc; ++i) {
                const int n = s_order[i];
                s_ptoff[n] = total_pts;
                total_pts += s_npts[n];
            }
        }
        int base = 0;
        if (status == 0 && total_pts > 0) {
            base = atomicAdd(A.img_pts + b, total_pts);
            if (base + total_pts > A.pts_cap) status = 4;       // the image's point buffer is full
        }
        s_misc[0] = nc;
        s_misc[1] = status;
        s_misc[2] = base;
        info[0] = status;
        info[1] = status == 0 ? nc : 0;
        info[2] = base;
        info[3] = status == 0 ? total_pts : 0;
        if (status == 0) {
            int32_t* ci = A.cont_info + ((size_t)b * A.D + d) * CMAX * 2;
            for (int i = 0; i < nc; ++i) {
                ci[2 * i] = s_ptoff[s_order[i]];
                ci[2 * i + 1] = s_npts[s_order[i]];
            }
        }
    }
    __syncthreads();
    if (s_misc[1] != 0) return;
    // ---- re-trace: one lane per contour writes its points -----------------------------------------------------------
    const int nc = s_misc[0];
    for (int c = 2 + lane; c <= nc + 1; c += 64) {
        const int i0 = s_start[c];
        const int yy = i0 / step, xx = i0 - yy * step;
        int16_t* out = A.pts + ((size_t)b * A.pts_cap + s_misc[2] + s_ptoff[c]) * 2;
        (void)follow<false, true>(F, step, deltas, i0, xx - 1, yy - 1, s_hole[c] != 0, c, guard_max, out, x0, y0);
    }
}

}  // namespace

extern "C" td_status td_trace_contours_dev(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits,
                                           int64_t mask_words_per_image, const int32_t* counts, int batch, int dets_per_image,
                                           int16_t* points, int points_cap, int32_t* image_points, int32_t* det_info,
                                           int32_t* contour_info, void* stream) {
    TD_REQUIRE(mask_region && mask_offset && mask_bits && counts && points && image_points && det_info && contour_info,
               "td_trace_contours_dev: null pointer");
    TD_REQUIRE(batch >= 1 && batch <= TD_MAX_BATCH && dets_per_image >= 1 && dets_per_image <= 1024 && points_cap >= 1,
               "td_trace_contours_dev: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TD_HIP_CHECK(hipMemsetAsync(image_points, 0, sizeof(int32_t) * (size_t)batch, s));
    TraceArgs A{};
    A.region = mask_region;
    A.offset = reinterpret_cast<const long long*>(mask_offset);
    A.bits = mask_bits;
    A.counts = counts;
    A.D = dets_per_image;
    A.words_per_image = mask_words_per_image;
    A.pts = points;
    A.pts_cap = points_cap;
    A.img_pts = image_points;
    A.det_info = det_info;
    A.cont_info = contour_info;
    hipLaunchKernelGGL((contour_trace_kernel<LABEL_SMALL, 0>), dim3(dets_per_image, batch), dim3(64), 0, s, A);
    TD_KERNEL_CHECK();
    hipLaunchKernelGGL((contour_trace_kernel<LABEL_LARGE, LABEL_SMALL>), dim3(dets_per_image, batch), dim3(64), 0, s, A);
    TD_KERNEL_CHECK();
    return TD_OK;
}
