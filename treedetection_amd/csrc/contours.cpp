// Host-side border following for the mask → polygon epilogue of the reference
// (TreeDetection/prediction.py:232-234: cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)).
//
// OpenCV is not vendored in the reference and not installed here, so this restates its published algorithm:
// Suzuki & Abe border following (raster scan; an outer border starts at a 0→1 transition, a hole border at the
// last foreground pixel before a 1→0 transition whose east neighbour was not yet examined as a 0-pixel), the
// 8-neighbour tracing loop of OpenCV's icvFetchContour (clockwise search for the first neighbour, counter-clockwise
// search while following, right-edge pixels marked negative) and CHAIN_APPROX_SIMPLE (a point is emitted only where
// the step direction changes). Contours come back in RETR_TREE order: depth-first, siblings most-recently-found
// first (OpenCV prepends each new contour to its parent's child list). Labels are int32, so there is no 7-bit
// label wrap. Pure host code: this runs on the CPU thread pool next to the GPU forward.
#include "common.h"

#include <cstring>
#include <vector>

namespace {

struct Node {
    int parent = 0;        // index into nodes (0 = frame)
    bool hole = false;
    int first = 0, count = 0;   // slice of the point pool
    std::vector<int> children;
};

// Traces every border of the padded label image F ((h+2) x (w+2), 0/1 on entry) and appends the contours in
// RETR_TREE order to `pts` (x,y pairs) / `starts` (point index of each contour + end sentinel).
void trace_padded(std::vector<int32_t>& F, int h, int w, std::vector<int32_t>& pts, std::vector<int32_t>& starts) {
    const int step = w + 2;
    // direction s: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards)
    const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int deltas[16];
    for (int s = 0; s < 16; ++s) deltas[s] = dy[s & 7] * step + dx[s & 7];

    std::vector<Node> nodes(2);     // [0] unused, [1] = frame (label 1)
    nodes[1].parent = 0;
    nodes[1].hole = true;
    std::vector<int32_t> pool;
    pool.reserve(4096);
    int nbd = 1;
    for (int y = 1; y <= h; ++y) {
        int lnbd = 1;
        for (int x = 1; x <= w; ++x) {
            int32_t* p0 = &F[(size_t)y * step + x];
            const int32_t f = *p0;
            if (f == 0) continue;
            bool outer = false, hole = false;
            if (f == 1 && p0[-1] == 0) outer = true;
            else if (f >= 1 && p0[1] == 0) hole = true;
            if (outer || hole) {
                if (hole && f > 1) lnbd = f;
                ++nbd;
                Node nd;
                nd.hole = hole;
                // Suzuki's parent table: same kind as LNBD's border → sibling (LNBD's parent), else child of LNBD
                const Node& ln = nodes[lnbd];
                nd.parent = (ln.hole == hole) ? ln.parent : lnbd;
                if (nd.parent == 0) nd.parent = 1;
                nd.first = (int)pool.size() / 2;
                // ---- follow the border (OpenCV icvFetchContour, CHAIN_APPROX_SIMPLE) ----
                int32_t* i0 = p0;
                int px = x - 1, py = y - 1;      // un-padded coordinates of the current point
                int s_end = hole ? 0 : 4, s = s_end;
                int32_t* i1 = nullptr;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) {      // single pixel
                    *i0 = -nbd;
                    pool.push_back(px);
                    pool.push_back(py);
                } else {
                    int32_t* i3 = i0;
                    int prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        int32_t* i4 = nullptr;
                        while (s < 15) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = -nbd;      // east neighbour examined as 0
                        else if (*i3 == 1) *i3 = nbd;
                        if (s != prev_s) {
                            pool.push_back(px);
                            pool.push_back(py);
                            prev_s = s;
                        }
                        px += dx[s];
                        py += dy[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                nd.count = (int)pool.size() / 2 - nd.first;
                nodes.push_back(nd);
                nodes[nd.parent].children.push_back((int)nodes.size() - 1);
            }
            const int32_t fv = *p0;
            if (fv != 1) lnbd = fv < 0 ? -fv : fv;
        }
    }
    // ---- RETR_TREE order: pre-order DFS, children most-recently-found first ----
    std::vector<int> stack;
    for (int c : nodes[1].children) stack.push_back(c);   // popped from the back = most recent first
    pts.clear();
    starts.clear();
    while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        starts.push_back((int32_t)(pts.size() / 2));
        pts.insert(pts.end(), pool.begin() + 2 * (size_t)nodes[n].first,
                   pool.begin() + 2 * (size_t)(nodes[n].first + nodes[n].count));
        for (int c : nodes[n].children) stack.push_back(c);
    }
    starts.push_back((int32_t)(pts.size() / 2));
}

}  // namespace

void td_trace_contours_u8(const uint8_t* img, int h, int w, std::vector<int32_t>& pts, std::vector<int32_t>& starts) {
    const int step = w + 2;
    std::vector<int32_t> F((size_t)(h + 2) * step, 0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) F[(size_t)(y + 1) * step + x + 1] = img[(size_t)y * w + x] ? 1 : 0;
    trace_padded(F, h, w, pts, starts);
}

void td_trace_contours_bits(const uint32_t* rows, int words_per_row, int h, int w, std::vector<int32_t>& pts,
                            std::vector<int32_t>& starts) {
    const int step = w + 2;
    // the label image of the calling thread, grown as needed and reused: a fresh vector per detection is a fresh
    // mapping for every region above 128 KB (page faults, kernel zeroing, munmap) — a third of this function's time
    static thread_local std::vector<int32_t> F;
    const size_t need = (size_t)(h + 2) * step;
    if (F.size() < need) F.resize(need);
    std::memset(F.data(), 0, (size_t)step * sizeof(int32_t));                              // top frame row
    std::memset(F.data() + (size_t)(h + 1) * step, 0, (size_t)step * sizeof(int32_t));     // bottom frame row
    for (int y = 0; y < h; ++y) {
        const uint32_t* r = rows + (size_t)y * words_per_row;
        int32_t* f = &F[(size_t)(y + 1) * step + 1];
        f[-1] = 0;
        f[w] = 0;
        for (int x0 = 0; x0 < w; x0 += 32) {
            const uint32_t word = r[x0 >> 5];
            const int nb = w - x0 < 32 ? w - x0 : 32;
            if (word == 0u) {
                std::memset(f + x0, 0, (size_t)nb * sizeof(int32_t));
            } else {
                for (int b = 0; b < nb; ++b) f[x0 + b] = (int32_t)((word >> b) & 1u);
            }
        }
    }
    trace_padded(F, h, w, pts, starts);
}

extern "C" int td_find_contours(const uint8_t* img, int h, int w, int32_t* points, int max_points, int32_t* starts,
                                int max_contours) {
    if (!img || !points || !starts || h < 1 || w < 1 || max_points < 1 || max_contours < 1) {
        td_set_error("td_find_contours: bad argument");
        return TD_ERR_INVALID;
    }
    std::vector<int32_t> pts, st;
    td_trace_contours_u8(img, h, w, pts, st);
    const int total = (int)st.size() - 1;
    if (total > max_contours) {
        td_set_error("td_find_contours: %d contours exceed capacity %d", total, max_contours);
        return TD_ERR_CAPACITY;
    }
    if ((int)(pts.size() / 2) > max_points) {
        td_set_error("td_find_contours: %d points exceed capacity %d", (int)(pts.size() / 2), max_points);
        return TD_ERR_CAPACITY;
    }
    if (!pts.empty()) std::memcpy(points, pts.data(), pts.size() * sizeof(int32_t));   // an empty vector's data() may be null
    if (!st.empty()) std::memcpy(starts, st.data(), st.size() * sizeof(int32_t));
    return total;
}
