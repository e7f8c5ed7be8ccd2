// Per-crown raster statistics of the post-processing stage
// (reference TreeDetection/postprocessing.py: get_height_within_polygon 25-115, get_ndvi_within_polygon 117-219,
// get_metadata_within_polygon 221-347; utilities.is_point_in_polygon_batch 78-98).
//
// The reference tests EVERY pixel of the raster against every crown's bounding circle on cupy
// (O(crowns x pixels)). Here one workgroup per crown visits only the pixels of the circle's bounding box; the
// membership test and the values it selects are the reference's, operation for operation:
//   pixel position  x = (a*col' + b*row') + c,  y = (d*col' + e*row') + f  in float64 (col', row' = the indices the
//                   reference passes to raster_to_geo — it adds the row offset of its window to the column and vice
//                   versa; both offsets are 0 when the window is the whole raster);
//   height mode     (x - cx)^2 + (y - cy)^2 <= r^2 in float64 with cx, cy, r^2 the float32 values widened;
//                   result = max value, position of its FIRST occurrence in row-major order (argmax), as float32;
//   NDVI mode       x, y rounded to float32 first, then the same test entirely in float32;
//                   result = min, max, mean, population variance (mean / variance accumulated in float64 and rounded
//                   once — cupy's own float32 reduction order is not reproducible).
// A crown whose circle holds no pixel gets -1 in every field. HBM-bound gather; no matrix work.
#include "common.h"

namespace {

struct CrownArgs {
    const float* raster;
    int rows, cols;                 // raster shape
    int r_lo, c_lo, sub_rows, sub_cols;   // the reference's window (subset) inside the raster
    double a, b, c, d, e, f;
    const float* circles;           // [n][3] cx, cy, r (float32, from the float32 vertex arrays)
    int n;
    int mode;                       // 0 height, 1 NDVI
    float radius_scale;
    float* out;                     // mode 0: [n][3] max, x, y    mode 1: [n][4] min, max, mean, var
};

__device__ __forceinline__ bool inside_f64(const CrownArgs& A, int row_sub, int col_sub, float cx, float cy, float r2,
                                           double& x, double& y) {
    const double col_arg = (double)(col_sub + A.r_lo), row_arg = (double)(row_sub + A.c_lo);   // the reference's swap
    x = (A.a * col_arg + A.b * row_arg) + A.c;
    y = (A.d * col_arg + A.e * row_arg) + A.f;
    const double dx = x - (double)cx, dy = y - (double)cy;
    return dx * dx + dy * dy <= (double)r2;
}

__device__ __forceinline__ bool inside_f32(const CrownArgs& A, int row_sub, int col_sub, float cx, float cy, float r2) {
    const double col_arg = (double)(col_sub + A.r_lo), row_arg = (double)(row_sub + A.c_lo);
    const float x = (float)((A.a * col_arg + A.b * row_arg) + A.c);
    const float y = (float)((A.d * col_arg + A.e * row_arg) + A.f);
    const float dx = __fsub_rn(x, cx), dy = __fsub_rn(y, cy);
    return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) <= r2;
}

__global__ __launch_bounds__(256) void crown_stats_kernel(const CrownArgs A) {
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    const float cx = A.circles[3 * k], cy = A.circles[3 * k + 1];
    const float r = __fmul_rn(A.circles[3 * k + 2], A.radius_scale);
    const float r2 = __fmul_rn(r, r);
    // bounding box of the circle in subset indices (whole subset when the transform is rotated)
    int c0 = 0, c1 = A.sub_cols - 1, r0 = 0, r1 = A.sub_rows - 1;
    if (A.b == 0.0 && A.d == 0.0 && A.a != 0.0 && A.e != 0.0) {
        const double m = (double)r * 1.0001 + 1e-3;
        double u0 = ((double)cx - m - A.c) / A.a, u1 = ((double)cx + m - A.c) / A.a;     // col' range
        double v0 = ((double)cy - m - A.f) / A.e, v1 = ((double)cy + m - A.f) / A.e;     // row' range
        if (u0 > u1) { const double t = u0; u0 = u1; u1 = t; }
        if (v0 > v1) { const double t = v0; v0 = v1; v1 = t; }
        const double lo_c = floor(u0) - 1.0 - A.r_lo, hi_c = ceil(u1) + 1.0 - A.r_lo;
        const double lo_r = floor(v0) - 1.0 - A.c_lo, hi_r = ceil(v1) + 1.0 - A.c_lo;
        c0 = lo_c > 0.0 ? (lo_c > 2e9 ? A.sub_cols : (int)lo_c) : 0;
        r0 = lo_r > 0.0 ? (lo_r > 2e9 ? A.sub_rows : (int)lo_r) : 0;
        c1 = hi_c < (double)(A.sub_cols - 1) ? (hi_c < -1.0 ? -1 : (int)hi_c) : A.sub_cols - 1;
        r1 = hi_r < (double)(A.sub_rows - 1) ? (hi_r < -1.0 ? -1 : (int)hi_r) : A.sub_rows - 1;
    }
    const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
    const long long total = (bw > 0 && bh > 0) ? (long long)bw * bh : 0;

    __shared__ float s_max[256], s_min[256];
    __shared__ long long s_idx[256];
    __shared__ double s_sum[256];
    __shared__ long long s_cnt[256];
    float vmax = -INFINITY, vmin = INFINITY;
    long long imax = -1, cnt = 0;
    double sum = 0.0;
    for (long long p = tid; p < total; p += 256) {
        const int rs = r0 + (int)(p / bw), cs = c0 + (int)(p % bw);
        double x, y;
        const bool in = A.mode == 0 ? inside_f64(A, rs, cs, cx, cy, r2, x, y) : inside_f32(A, rs, cs, cx, cy, r2);
        if (!in) continue;
        const float v = A.raster[(size_t)(rs + A.r_lo) * A.cols + (cs + A.c_lo)];
        const long long lin = (long long)rs * A.sub_cols + cs;       // position in the flattened subset
        if (imax < 0 || v > vmax || (v == vmax && lin < imax)) {
            vmax = v;
            imax = lin;
        }
        vmin = v < vmin ? v : vmin;
        sum += (double)v;
        ++cnt;
    }
    s_max[tid] = vmax; s_min[tid] = vmin; s_idx[tid] = imax; s_sum[tid] = sum; s_cnt[tid] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const long long oi = s_idx[tid + s];
            if (oi >= 0) {
                const float ov = s_max[tid + s];
                if (s_idx[tid] < 0 || ov > s_max[tid] || (ov == s_max[tid] && oi < s_idx[tid])) {
                    s_max[tid] = ov;
                    s_idx[tid] = oi;
                }
            }
            s_min[tid] = s_min[tid + s] < s_min[tid] ? s_min[tid + s] : s_min[tid];
            s_sum[tid] += s_sum[tid + s];
            s_cnt[tid] += s_cnt[tid + s];
        }
        __syncthreads();
    }
    const long long N = s_cnt[0];
    if (A.mode == 0) {
        if (tid == 0) {
            float* o = A.out + 3 * (size_t)k;
            if (N == 0) {
                o[0] = -1.f; o[1] = -1.f; o[2] = -1.f;
            } else {
                const int rs = (int)(s_idx[0] / A.sub_cols), cs = (int)(s_idx[0] % A.sub_cols);
                double x, y;
                (void)inside_f64(A, rs, cs, cx, cy, r2, x, y);
                o[0] = s_max[0]; o[1] = (float)x; o[2] = (float)y;
            }
        }
        return;
    }
    if (N == 0) {
        if (tid == 0) {
            float* o = A.out + 4 * (size_t)k;
            o[0] = o[1] = o[2] = o[3] = -1.f;
        }
        return;
    }
    const double mean = s_sum[0] / (double)N;
    const float fmin_ = s_min[0], fmax_ = s_max[0];
    __syncthreads();
    double sq = 0.0;                                   // second pass: population variance about the float64 mean
    for (long long p = tid; p < total; p += 256) {
        const int rs = r0 + (int)(p / bw), cs = c0 + (int)(p % bw);
        if (!inside_f32(A, rs, cs, cx, cy, r2)) continue;
        const double dv = (double)A.raster[(size_t)(rs + A.r_lo) * A.cols + (cs + A.c_lo)] - mean;
        sq += dv * dv;
    }
    s_sum[tid] = sq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_sum[tid] += s_sum[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        float* o = A.out + 4 * (size_t)k;
        o[0] = fmin_; o[1] = fmax_; o[2] = (float)mean; o[3] = (float)(s_sum[0] / (double)N);
    }
}

}  // namespace

extern "C" td_status td_crown_stats(const float* raster, int rows, int cols, const double* transform, const int32_t* window,
                                    const float* circles, int n, int mode, float radius_scale, float* out, void* stream) {
    TD_REQUIRE(raster && transform && window && out && (n == 0 || circles), "td_crown_stats: null pointer");
    TD_REQUIRE(rows >= 1 && cols >= 1 && n >= 0 && (mode == 0 || mode == 1), "td_crown_stats: bad shape / mode");
    const int r_lo = window[0], c_lo = window[1], r_hi = window[2], c_hi = window[3];
    TD_REQUIRE(r_lo >= 0 && c_lo >= 0 && r_hi < rows && c_hi < cols, "td_crown_stats: window [%d..%d] x [%d..%d] outside the %d x %d raster",
               r_lo, r_hi, c_lo, c_hi, rows, cols);
    if (n == 0) return TD_OK;
    CrownArgs A{};
    A.raster = raster;
    A.rows = rows;
    A.cols = cols;
    A.r_lo = r_lo;
    A.c_lo = c_lo;
    A.sub_rows = r_hi >= r_lo ? r_hi - r_lo + 1 : 0;
    A.sub_cols = c_hi >= c_lo ? c_hi - c_lo + 1 : 0;
    A.a = transform[0]; A.b = transform[1]; A.c = transform[2];
    A.d = transform[3]; A.e = transform[4]; A.f = transform[5];
    A.circles = circles;
    A.n = n;
    A.mode = mode;
    A.radius_scale = radius_scale;
    A.out = out;
    hipLaunchKernelGGL(crown_stats_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    TD_KERNEL_CHECK();
    return TD_OK;
}
