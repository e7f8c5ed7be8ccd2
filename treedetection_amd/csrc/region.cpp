// Polygon-vs-region predicates for the outline steps of the reference:
//   * fuse_predictions (TreeDetection/helpers.py:703-834): forest crowns that `intersects(forest_union)`, urban crowns
//     that are not `within(forest_union)`, forest_union = unary_union of the outline polygons near the image;
//   * the only_forest / only_urban tile flags (preprocessing.py:70-95): `candidates.intersects(bbox)` and
//     `unary_union.contains(bbox)`.
// shapely/GEOS evaluate those through an overlay of the whole outline. Here the region stays what it is on disk — a
// set of polygons with holes — and every query ring is related to it directly:
//   membership(x) = x lies in the closed point set of some polygon (inside its shell and in none of its holes, or on
//   one of its rings), decided by exact-sign crossing counts per polygon;
//   intersects(Q) = a vertex of Q is a member, or an edge of Q meets a region edge, or a region ring lies inside Q;
//   within(Q)     = every vertex of Q is a member, every piece of Q's boundary between two consecutive meetings with
//   region edges has a member midpoint, and no region ring lies loose inside Q bounding uncovered ground.
// That is the DE-9IM meaning of the two predicates for a simple query ring against the union of the polygons; it
// differs from an overlay only for uncovered gaps that lie strictly inside a query without touching its boundary
// and are bounded by several polygons at once (the hole of a single polygon is handled).
// A y-binned edge index keeps each crossing count local. Pure host code.
#include "common.h"
#include "geom_predicates.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace tdgeom;

struct Edge {
    Pt a, b;
    int poly, ring;
};

struct Region {
    std::vector<Edge> edges;
    std::vector<std::vector<int>> bins;       // edge ids per y-bin
    std::vector<std::vector<int>> ring_bins;  // ring ids by the y-bin of the ring's first vertex
    std::vector<Pt> ring_first;
    std::vector<int> ring_poly_;
    std::vector<char> ring_is_hole;
    double y0 = 0, inv_h = 0;
    int nb = 1;

    int bin_of(double y) const {
        const double f = (y - y0) * inv_h;
        if (!(f > 0)) return 0;
        return f >= nb ? nb - 1 : (int)f;
    }

    // Crossing state of point p against every polygon near it: *on = polygon ids with p on one of their rings,
    // *in = polygon ids with an odd crossing count (horizontal ray to +x, half-open rule on y).
    void classify(const Pt& p, std::vector<int>& on, std::vector<int>& in, std::vector<int>& scratch) const {
        on.clear();
        in.clear();
        scratch.clear();
        if (edges.empty()) return;
        if (p.y < ymin || p.y > ymax) return;
        for (int id : bins[bin_of(p.y)]) {
            const Edge& e = edges[id];
            const double lo = std::fmin(e.a.y, e.b.y), hi = std::fmax(e.a.y, e.b.y);
            if (p.y < lo || p.y > hi) continue;
            if (std::fmax(e.a.x, e.b.x) < p.x) continue;
            const int o = orientation(e.a.x, e.a.y, e.b.x, e.b.y, p.x, p.y);
            if (o == 0 && env_has(e.a, e.b, p)) {
                on.push_back(e.poly);
                continue;
            }
            if ((e.a.y <= p.y) != (e.b.y <= p.y)) {          // half-open: counts an edge once at a shared vertex
                const bool up = e.a.y <= p.y;                // a below-or-level, b above
                if ((up && o > 0) || (!up && o < 0)) scratch.push_back(e.poly);
            }
        }
        std::sort(scratch.begin(), scratch.end());
        for (size_t i = 0; i < scratch.size();) {
            size_t j = i;
            while (j < scratch.size() && scratch[j] == scratch[i]) ++j;
            if ((j - i) & 1) in.push_back(scratch[i]);
            i = j;
        }
        std::sort(on.begin(), on.end());
        on.erase(std::unique(on.begin(), on.end()), on.end());
    }
    double ymin = 0, ymax = 0;
};

// does segment (p1,p2) meet segment (q1,q2) at all (touching counts)?
bool segments_meet(const Pt& p1, const Pt& p2, const Pt& q1, const Pt& q2) {
    if (!env_overlap(p1, p2, q1, q2)) return false;
    const int a = orientation(p1.x, p1.y, p2.x, p2.y, q1.x, q1.y);
    const int b = orientation(p1.x, p1.y, p2.x, p2.y, q2.x, q2.y);
    if ((a > 0 && b > 0) || (a < 0 && b < 0)) return false;
    const int c = orientation(q1.x, q1.y, q2.x, q2.y, p1.x, p1.y);
    const int d = orientation(q1.x, q1.y, q2.x, q2.y, p2.x, p2.y);
    if ((c > 0 && d > 0) || (c < 0 && d < 0)) return false;
    return true;     // collinear pairs reach here only with overlapping envelopes, i.e. they share a point
}

// parameters t in [0,1] along (p1,p2) where it meets (q1,q2): none, one point, or the two ends of a collinear overlap
int meet_params(const Pt& p1, const Pt& p2, const Pt& q1, const Pt& q2, double t[2]) {
    if (!segments_meet(p1, p2, q1, q2)) return 0;
    const double dx = p2.x - p1.x, dy = p2.y - p1.y;
    const double len2 = dx * dx + dy * dy;
    auto param = [&](const Pt& q) {
        double v = len2 > 0 ? ((q.x - p1.x) * dx + (q.y - p1.y) * dy) / len2 : 0.0;
        return v < 0 ? 0.0 : (v > 1 ? 1.0 : v);
    };
    const int a = orientation(p1.x, p1.y, p2.x, p2.y, q1.x, q1.y);
    const int b = orientation(p1.x, p1.y, p2.x, p2.y, q2.x, q2.y);
    if (a == 0 && b == 0) {                    // collinear overlap
        double t0 = param(q1), t1 = param(q2);
        if (t0 > t1) std::swap(t0, t1);
        t[0] = t0;
        t[1] = t1;
        return 2;
    }
    if (a == 0) { t[0] = param(q1); return 1; }
    if (b == 0) { t[0] = param(q2); return 1; }
    // proper or end-of-p touch: solve with the cross products
    const double ex = q2.x - q1.x, ey = q2.y - q1.y;
    const double den = dx * ey - dy * ex;
    double v = den != 0 ? ((q1.x - p1.x) * ey - (q1.y - p1.y) * ex) / den : 0.0;
    t[0] = v < 0 ? 0.0 : (v > 1 ? 1.0 : v);
    return 1;
}

// closed point-in-ring for the (small) query ring: 0 outside, 1 inside, 2 on the boundary
int point_in_query(const Pt* q, int n, const Pt& p) {
    bool in = false;
    for (int i = 0; i + 1 < n; ++i) {
        const Pt &a = q[i], &b = q[i + 1];
        const int o = orientation(a.x, a.y, b.x, b.y, p.x, p.y);
        if (o == 0 && env_has(a, b, p)) return 2;
        if ((a.y <= p.y) != (b.y <= p.y)) {
            const bool up = a.y <= p.y;
            if ((up && o > 0) || (!up && o < 0)) in = !in;
        }
    }
    return in ? 1 : 0;
}

}  // namespace

extern "C" int td_region_relate(const double* ring_xy, const int64_t* ring_start, const int32_t* ring_poly, int n_rings,
                                const double* query_xy, const int64_t* query_start, int n_queries, uint8_t* flags) {
    if (n_rings < 0 || n_queries < 0 || (n_rings > 0 && (!ring_xy || !ring_start || !ring_poly)) ||
        (n_queries > 0 && (!query_xy || !query_start || !flags))) {
        td_set_error("td_region_relate: bad argument");
        return TD_ERR_INVALID;
    }
    // ---- index the region ------------------------------------------------------------------------------
    Region R;
    const Pt* rp = reinterpret_cast<const Pt*>(ring_xy);
    double ymin = INFINITY, ymax = -INFINITY;
    int64_t n_edges = 0;
    for (int r = 0; r < n_rings; ++r) {
        const int64_t s = ring_start[r], e = ring_start[r + 1];
        if (e - s < 4 || !(rp[s] == rp[e - 1])) {
            td_set_error("td_region_relate: region ring %d is not a closed ring of >= 4 points", r);
            return TD_ERR_INVALID;
        }
        n_edges += e - s - 1;
        for (int64_t i = s; i < e; ++i) {
            ymin = std::fmin(ymin, rp[i].y);
            ymax = std::fmax(ymax, rp[i].y);
        }
    }
    R.ymin = ymin;
    R.ymax = ymax;
    R.nb = (int)std::min<int64_t>(8192, std::max<int64_t>(1, n_edges / 4));
    R.y0 = ymin;
    R.inv_h = (ymax > ymin) ? R.nb / (ymax - ymin) : 0.0;
    R.bins.assign(R.nb, {});
    R.ring_bins.assign(R.nb, {});
    R.edges.reserve((size_t)n_edges);
    R.ring_first.resize(n_rings);
    R.ring_poly_.resize(n_rings);
    R.ring_is_hole.resize(n_rings);
    for (int r = 0; r < n_rings; ++r) {
        const int64_t s = ring_start[r], e = ring_start[r + 1];
        R.ring_first[r] = rp[s];
        R.ring_poly_[r] = ring_poly[r];
        R.ring_is_hole[r] = r > 0 && ring_poly[r - 1] == ring_poly[r];
        R.ring_bins[R.bin_of(rp[s].y)].push_back(r);
        for (int64_t i = s; i + 1 < e; ++i) {
            const int id = (int)R.edges.size();
            R.edges.push_back({rp[i], rp[i + 1], ring_poly[r], r});
            const int b0 = R.bin_of(std::fmin(rp[i].y, rp[i + 1].y)), b1 = R.bin_of(std::fmax(rp[i].y, rp[i + 1].y));
            for (int b = b0; b <= b1; ++b) R.bins[b].push_back(id);
        }
    }
    // ---- relate each query -------------------------------------------------------------------------------
    const Pt* qp = reinterpret_cast<const Pt*>(query_xy);
    std::vector<int> on, in, scratch, cand;
    std::vector<double> ts;
    std::vector<char> ring_met((size_t)n_rings, 0);
    std::vector<int> met_list;
    auto member = [&](const Pt& p) {
        R.classify(p, on, in, scratch);
        return !on.empty() || !in.empty();
    };
    for (int q = 0; q < n_queries; ++q) {
        const int64_t s = query_start[q], e = query_start[q + 1];
        const int n = (int)(e - s);
        flags[q] = 0;
        if (n < 4 || !(qp[s] == qp[e - 1])) {
            td_set_error("td_region_relate: query %d is not a closed ring of >= 4 points", q);
            return TD_ERR_INVALID;
        }
        if (R.edges.empty()) continue;
        const Pt* Q = qp + s;
        double qx0 = INFINITY, qx1 = -INFINITY, qy0 = INFINITY, qy1 = -INFINITY;
        for (int i = 0; i < n; ++i) {
            qx0 = std::fmin(qx0, Q[i].x);
            qx1 = std::fmax(qx1, Q[i].x);
            qy0 = std::fmin(qy0, Q[i].y);
            qy1 = std::fmax(qy1, Q[i].y);
        }
        bool intersects = false, within = true;
        if (qy1 < R.ymin || qy0 > R.ymax) continue;               // disjoint in y: neither predicate holds
        // candidate region edges: those in the query's y-bins whose envelope overlaps the query's
        cand.clear();
        const int b0 = R.bin_of(qy0), b1 = R.bin_of(qy1);
        for (int b = b0; b <= b1; ++b)
            for (int id : R.bins[b]) {
                const Edge& ed = R.edges[id];
                if (std::fmax(ed.a.x, ed.b.x) < qx0 || std::fmin(ed.a.x, ed.b.x) > qx1) continue;
                if (std::fmax(ed.a.y, ed.b.y) < qy0 || std::fmin(ed.a.y, ed.b.y) > qy1) continue;
                cand.push_back(id);
            }
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        // (1) vertices
        for (int i = 0; i + 1 < n; ++i) {
            if (member(Q[i])) intersects = true;
            else within = false;
            if (intersects && !within) break;
        }
        // (2) edges: meetings with region edges, then the pieces between them
        met_list.clear();
        if (!(intersects && !within)) {
            for (int i = 0; i + 1 < n; ++i) {
                ts.clear();
                for (int id : cand) {
                    const Edge& ed = R.edges[id];
                    double t[2];
                    const int k = meet_params(Q[i], Q[i + 1], ed.a, ed.b, t);
                    if (k) {
                        intersects = true;
                        if (!ring_met[ed.ring]) {
                            ring_met[ed.ring] = 1;
                            met_list.push_back(ed.ring);
                        }
                    }
                    for (int j = 0; j < k; ++j) ts.push_back(t[j]);
                }
                if (!within) continue;
                ts.push_back(0.0);
                ts.push_back(1.0);
                std::sort(ts.begin(), ts.end());
                for (size_t j = 0; j + 1 < ts.size(); ++j) {
                    if (!(ts[j + 1] > ts[j])) continue;
                    const double tm = 0.5 * (ts[j] + ts[j + 1]);
                    const Pt m{Q[i].x + tm * (Q[i + 1].x - Q[i].x), Q[i].y + tm * (Q[i + 1].y - Q[i].y)};
                    if (!member(m)) {
                        within = false;
                        break;
                    }
                }
            }
        }
        // (3) region rings lying loose inside the query (no meeting with its boundary): a polygon inside the query
        //     makes them intersect; an uncovered hole inside the query breaks `within`
        if (!intersects || within) {
            for (int b = b0; b <= b1 && (!intersects || within); ++b)
                for (int r : R.ring_bins[b]) {
                    if (ring_met[r]) continue;
                    const Pt& v = R.ring_first[r];
                    if (v.x < qx0 || v.x > qx1 || v.y < qy0 || v.y > qy1) continue;
                    if (point_in_query(Q, n, v) == 0) continue;
                    intersects = true;
                    if (within && R.ring_is_hole[r]) {
                        // the hole is open ground unless another polygon covers it: its first vertex must be a
                        // member through a polygon other than its own
                        R.classify(v, on, in, scratch);
                        bool covered = false;
                        for (int pid : in) covered |= pid != R.ring_poly_[r];
                        for (int pid : on) covered |= pid != R.ring_poly_[r];
                        if (!covered) within = false;
                    }
                }
        }
        for (int r : met_list) ring_met[r] = 0;
        flags[q] = (uint8_t)((intersects ? 1 : 0) | ((intersects && within) ? 2 : 0));
    }
    return TD_OK;
}
