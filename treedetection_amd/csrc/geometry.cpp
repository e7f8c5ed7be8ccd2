// Ring simplification for the stitching consumer of the per-tile prediction files
// (reference TreeDetection/helpers.py:464-465: `gdf["geometry"].simplify(tol, preserve_topology=True)`).
//
// shapely hands that call to GEOS' TopologyPreservingSimplifier; GEOS is a third-party dependency that is neither
// vendored in the reference nor installed here, so this restates its published algorithm (TaggedLineStringSimplifier,
// GEOS 3.11 / JTS 1.19) for the one geometry kind the pipeline produces — a polygon that is a single closed shell:
//   * Douglas-Peucker recursion over sections [i, j] of the ring; a section is flattened to the segment (p_i, p_j)
//     only if (a) every interior point is within `tolerance` of it, (b) the ring keeps at least 4 points in the
//     worst case, and (c) the new segment has no interior intersection with any segment already written to the
//     output or any input segment outside the section (segments of flattened sections leave the input set);
//   * otherwise the section is split at its furthest point;
//   * finally the ring's start/end vertex is dropped too when the segment joining its neighbours passes (a)-(c).
// Orientation tests use a floating-point filter backed by double-double arithmetic, as GEOS does
// (CGAlgorithmsDD::orientationIndex), so collinear staircase points behave the same way.
// Pure host code.
#include "common.h"
#include "geom_predicates.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdio>
#include <thread>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

using namespace tdgeom;

// LineIntersector::isInteriorIntersection of segments (p1,p2) and (q1,q2): some intersection point is not an
// endpoint of one of the two segments.
bool interior_intersection(const Pt& p1, const Pt& p2, const Pt& q1, const Pt& q2) {
    if (!env_overlap(p1, p2, q1, q2)) return false;
    const int Pq1 = orientation(p1.x, p1.y, p2.x, p2.y, q1.x, q1.y);
    const int Pq2 = orientation(p1.x, p1.y, p2.x, p2.y, q2.x, q2.y);
    if ((Pq1 > 0 && Pq2 > 0) || (Pq1 < 0 && Pq2 < 0)) return false;
    const int Qp1 = orientation(q1.x, q1.y, q2.x, q2.y, p1.x, p1.y);
    const int Qp2 = orientation(q1.x, q1.y, q2.x, q2.y, p2.x, p2.y);
    if ((Qp1 > 0 && Qp2 > 0) || (Qp1 < 0 && Qp2 < 0)) return false;
    Pt ip[2];
    int n = 0;
    if (Pq1 == 0 && Pq2 == 0 && Qp1 == 0 && Qp2 == 0) {       // collinear: overlap of the two envelopes
        const bool q1inP = env_has(p1, p2, q1), q2inP = env_has(p1, p2, q2);
        const bool p1inQ = env_has(q1, q2, p1), p2inQ = env_has(q1, q2, p2);
        if (q1inP && q2inP) { ip[0] = q1; ip[1] = q2; n = 2; }
        else if (p1inQ && p2inQ) { ip[0] = p1; ip[1] = p2; n = 2; }
        else if (q1inP && p1inQ) { ip[0] = q1; ip[1] = p1; n = (q1 == p1 && !q2inP && !p2inQ) ? 1 : 2; }
        else if (q1inP && p2inQ) { ip[0] = q1; ip[1] = p2; n = (q1 == p2 && !q2inP && !p1inQ) ? 1 : 2; }
        else if (q2inP && p1inQ) { ip[0] = q2; ip[1] = p1; n = (q2 == p1 && !q1inP && !p2inQ) ? 1 : 2; }
        else if (q2inP && p2inQ) { ip[0] = q2; ip[1] = p2; n = (q2 == p2 && !q1inP && !p1inQ) ? 1 : 2; }
        else return false;
    } else if (Pq1 == 0 || Pq2 == 0 || Qp1 == 0 || Qp2 == 0) {   // touches at an endpoint
        if (p1 == q1 || p1 == q2) ip[0] = p1;
        else if (p2 == q1 || p2 == q2) ip[0] = p2;
        else if (Pq1 == 0) ip[0] = q1;
        else if (Pq2 == 0) ip[0] = q2;
        else if (Qp1 == 0) ip[0] = p1;
        else ip[0] = p2;
        n = 1;
    } else {
        return true;                                              // proper crossing
    }
    for (int i = 0; i < n; ++i) {
        if (!(ip[i] == p1 || ip[i] == p2)) return true;
        if (!(ip[i] == q1 || ip[i] == q2)) return true;
    }
    return false;
}

// Distance::pointToSegment
double point_segment_distance(const Pt& p, const Pt& a, const Pt& b) {
    if (a.x == b.x && a.y == b.y) return std::hypot(p.x - a.x, p.y - a.y);
    const double len2 = (b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y);
    const double r = ((p.x - a.x) * (b.x - a.x) + (p.y - a.y) * (b.y - a.y)) / len2;
    if (r <= 0.0) return std::hypot(p.x - a.x, p.y - a.y);
    if (r >= 1.0) return std::hypot(p.x - b.x, p.y - b.y);
    const double s = ((a.y - p.y) * (b.x - a.x) - (a.x - p.x) * (b.y - a.y)) / len2;
    return std::fabs(s) * std::sqrt(len2);
}

struct OutSeg {
    int i, j;      // input vertex indices of its end points (it replaces input segments i .. j-1)
};

struct Simplifier {
    const Pt* p;
    int n;
    double tol;
    static constexpr int MIN_SIZE = 4;             // LinearRing
    std::vector<char> in_live;                     // input segment k = (p[k], p[k+1]) still in the input set
    std::vector<OutSeg> flat;                      // flattened segments (the "output index")
    std::vector<OutSeg> result;                    // every result segment, in ring order

    int result_size() const { return result.empty() ? 0 : (int)result.size() + 1; }

    bool bad_intersection(int si, int sj, const Pt& a, const Pt& b, int skip_lo2 = -1, int skip_hi2 = -1) const {
        for (const OutSeg& s : flat)
            if (interior_intersection(p[s.i], p[s.j], a, b)) return true;
        for (int k = 0; k + 1 < n; ++k) {
            if (!in_live[k]) continue;
            if (k >= si && k < sj) continue;                         // inside the section being replaced
            if (k >= skip_lo2 && k < skip_hi2) continue;
            if (interior_intersection(p[k], p[k + 1], a, b)) return true;
        }
        return false;
    }

    void section(int i, int j, int depth) {
        depth += 1;
        if (i + 1 == j) {
            result.push_back({i, j});                                // stays in the input set
            return;
        }
        bool ok = true;
        if (result_size() < MIN_SIZE && depth + 1 < MIN_SIZE) ok = false;
        double maxd = -1.0;
        int far = i;
        for (int k = i + 1; k < j; ++k) {
            const double d = point_segment_distance(p[k], p[i], p[j]);
            if (d > maxd) {
                maxd = d;
                far = k;
            }
        }
        if (maxd > tol) ok = false;
        if (ok && bad_intersection(i, j, p[i], p[j])) ok = false;
        if (ok) {
            for (int k = i; k < j; ++k) in_live[k] = 0;
            flat.push_back({i, j});
            result.push_back({i, j});
            return;
        }
        section(i, far, depth);
        section(far, j, depth);
    }

    void ring_endpoint() {
        if (result_size() <= MIN_SIZE) return;
        const OutSeg first = result.front(), last = result.back();
        const Pt &a = p[last.i], &b = p[first.j], &end = p[first.i];
        if (point_segment_distance(end, a, b) > tol) return;
        // the two result segments being merged (and what they replaced) do not count as obstacles
        std::vector<OutSeg> keep;
        for (const OutSeg& s : flat)
            if (!((s.i == first.i && s.j == first.j) || (s.i == last.i && s.j == last.j))) keep.push_back(s);
        flat.swap(keep);
        const bool bad = bad_intersection(first.i, first.j, a, b, last.i, last.j);
        flat.swap(keep);
        if (bad) return;
        result.front().i = last.i;        // first segment now starts at the last segment's start
        result.pop_back();
    }
};

}  // namespace

// OGC validity of a polygon shell as GEOS' IsValidOp decides it for one ring (reference helpers.py:816 `geom.is_valid` on every
// fused crown): finite coordinates, closed, at least four points of which three are distinct, and no two segments meeting
// anywhere but at the shared end point of consecutive ones — a ring that touches itself at a vertex, crosses itself, or runs
// back over itself (a spike) is invalid. Repeated consecutive points are allowed. Exact-sign orientation tests; O(n^2) pairs
// behind an envelope test (crowns have tens of vertices).
extern "C" int td_ring_is_valid(const double* xy, int n) {
    using namespace tdgeom;
    if (!xy || n < 0) {
        td_set_error("td_ring_is_valid: bad argument");
        return TD_ERR_INVALID;
    }
    if (n < 4) return 0;
    for (int i = 0; i < 2 * n; ++i)
        if (!std::isfinite(xy[i])) return 0;
    if (xy[0] != xy[2 * (n - 1)] || xy[1] != xy[2 * (n - 1) + 1]) return 0;
    std::vector<Pt> p;                                   // the ring without its closing point and without repeated points
    for (int i = 0; i + 1 < n; ++i) {
        const Pt q{xy[2 * i], xy[2 * i + 1]};
        if (p.empty() || !(p.back() == q)) p.push_back(q);
    }
    while (p.size() > 1 && p.back() == p.front()) p.pop_back();
    const int m = (int)p.size();
    if (m < 3) return 0;
    auto at = [&](int i) -> const Pt& { return p[(i % m + m) % m]; };
    for (int i = 0; i < m; ++i) {
        const Pt &a = at(i), &b = at(i + 1);
        // consecutive segments share b: they may not overlap (c back on the line a - b, towards a)
        const Pt& c = at(i + 2);
        if (orientation(a.x, a.y, b.x, b.y, c.x, c.y) == 0) {
            const double dot = (a.x - b.x) * (c.x - b.x) + (a.y - b.y) * (c.y - b.y);
            if (dot > 0) return 0;
        }
        for (int j = i + 2; j < m; ++j) {
            if (i == 0 && j == m - 1) continue;          // the closing segment is consecutive to the first
            const Pt &q1 = at(j), &q2 = at(j + 1);
            if (!env_overlap(a, b, q1, q2)) continue;
            const int o1 = orientation(a.x, a.y, b.x, b.y, q1.x, q1.y), o2 = orientation(a.x, a.y, b.x, b.y, q2.x, q2.y);
            if (o1 * o2 > 0) continue;
            const int o3 = orientation(q1.x, q1.y, q2.x, q2.y, a.x, a.y), o4 = orientation(q1.x, q1.y, q2.x, q2.y, b.x, b.y);
            if (o3 * o4 > 0) continue;
            return 0;                                    // they meet: crossing, touching, or collinear overlap (envelopes overlap)
        }
    }
    return 1;
}

extern "C" int td_simplify_ring(const double* xy, int n, double tolerance, double* out_xy, int out_cap) {
    if (!xy || !out_xy || n < 0 || out_cap < 0 || !(tolerance >= 0.0)) {
        td_set_error("td_simplify_ring: bad argument");
        return TD_ERR_INVALID;
    }
    auto copy_through = [&]() -> int {
        if (n > out_cap) {
            td_set_error("td_simplify_ring: %d points exceed capacity %d", n, out_cap);
            return TD_ERR_CAPACITY;
        }
        for (int i = 0; i < 2 * n; ++i) out_xy[i] = xy[i];
        return n;
    };
    const Pt* p = reinterpret_cast<const Pt*>(xy);
    if (n < 2) return copy_through();
    Simplifier s;
    s.p = p;
    s.n = n;
    s.tol = tolerance;
    s.in_live.assign((size_t)n, 1);
    s.section(0, n - 1, 0);
    const bool closed = p[0] == p[n - 1];
    if (closed) s.ring_endpoint();
    const int m = (int)s.result.size() + 1;
    if (m > out_cap) {
        td_set_error("td_simplify_ring: %d points exceed capacity %d", m, out_cap);
        return TD_ERR_CAPACITY;
    }
    int k = 0;
    for (const OutSeg& sg : s.result) {
        out_xy[2 * k] = p[sg.i].x;
        out_xy[2 * k + 1] = p[sg.i].y;
        ++k;
    }
    const int lastj = s.result.back().j;
    if (closed && s.result.front().i != 0) {        // the ring endpoint was dropped: close on the new first vertex
        out_xy[2 * k] = p[s.result.front().i].x;
        out_xy[2 * k + 1] = p[s.result.front().i].y;
    } else {
        out_xy[2 * k] = p[lastj].x;
        out_xy[2 * k + 1] = p[lastj].y;
    }
    return m;
}

// ---- one prediction file → GeoPackage geometry blobs ---------------------------------------------------
// (reference helpers.py:436-470: json.load, Polygon(coords), simplify, sjoin "within" the tile's shrunken box)
namespace {

struct JsonCursor {
    const char* p;
    const char* end;
    const char* err = nullptr;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool fail(const char* what) {
        if (!err) err = what;
        return false;
    }
    bool lit(const char* s) {
        const size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    // string → raw bytes between the quotes (escapes kept; only compared against plain ASCII keys)
    bool str(const char*& s, size_t& n) {
        if (p >= end || *p != '"') return fail("expected a string");
        s = ++p;
        while (p < end && *p != '"') p += (*p == '\\' && p + 1 < end) ? 2 : 1;
        if (p >= end) return fail("unterminated string");
        n = (size_t)(p - s);
        ++p;
        return true;
    }
    bool num(double& v) {
        if (lit("NaN")) { v = NAN; return true; }
        if (lit("Infinity")) { v = INFINITY; return true; }
        if (lit("-Infinity")) { v = -INFINITY; return true; }
        auto r = std::from_chars(p, end, v);
        if (r.ec != std::errc() || r.ptr == p) return fail("expected a number");
        p = r.ptr;
        return true;
    }
    bool skip() {      // any value
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '"') {
            const char* s;
            size_t n;
            return str(s, n);
        }
        if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            const bool obj = *p == '{';
            ++p;
            ws();
            if (p < end && *p == close) { ++p; return true; }
            for (;;) {
                if (obj) {
                    const char* s;
                    size_t n;
                    ws();
                    if (!str(s, n)) return false;
                    ws();
                    if (p >= end || *p != ':') return fail("expected ':'");
                    ++p;
                }
                if (!skip()) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == close) { ++p; return true; }
                return fail("expected ',' or a closing bracket");
            }
        }
        if (lit("true") || lit("false") || lit("null")) return true;
        double v;
        return num(v);
    }
    // nested arrays of numbers, flattened in reading order (what np.array(...).reshape(-1, 2) sees)
    bool numbers(std::vector<double>& out) {
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p != '[') {
            double v;
            if (!num(v)) return false;
            out.push_back(v);
            return true;
        }
        ++p;
        ws();
        if (p < end && *p == ']') { ++p; return true; }
        for (;;) {
            if (!numbers(out)) return false;
            ws();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == ']') { ++p; return true; }
            return fail("expected ',' or ']'");
        }
    }
};

inline void put(std::vector<uint8_t>& b, const void* src, size_t n) {
    const uint8_t* s = (const uint8_t*)src;
    b.insert(b.end(), s, s + n);
}

}  // namespace

extern "C" int td_simplify_ring(const double* xy, int n, double tolerance, double* out_xy, int out_cap);

namespace {

// One prediction file's text → geometry blobs / offsets / scores appended to the vectors (the body of td_stitch_tile_json).
// `what` names the entry point in error messages.
int stitch_text(const char* what_fn, const char* json, int64_t len, const double* box, double tolerance, int32_t srs_id,
                std::vector<uint8_t>& out, std::vector<int64_t>& offs, std::vector<double>& sc) {
    JsonCursor c{json, json + len};
    std::vector<double> coords, simp;
    if (offs.empty()) offs.push_back(0);
    auto bad = [&](const char* what) {
        td_set_error("%s: %s at byte %lld", what_fn, what, (long long)(c.p - json));
        return TD_ERR_INVALID;
    };
    c.ws();
    if (c.p >= c.end || *c.p != '[') return bad("expected a JSON array of predictions");
    ++c.p;
    c.ws();
    bool first = true;
    while (!(c.p < c.end && *c.p == ']' && first)) {
        first = false;
        c.ws();
        if (c.p >= c.end || *c.p != '{') return bad("expected a prediction object");
        ++c.p;
        double score = 0.0;
        bool have_score = false, have_ring = false;
        coords.clear();
        c.ws();
        if (c.p < c.end && *c.p == '}') {
            ++c.p;
        } else {
            for (;;) {
                const char* k;
                size_t kn;
                c.ws();
                if (!c.str(k, kn)) return bad(c.err);
                c.ws();
                if (c.p >= c.end || *c.p != ':') return bad("expected ':'");
                ++c.p;
                c.ws();
                if (kn == 5 && std::memcmp(k, "score", 5) == 0) {
                    if (!c.num(score)) return bad(c.err);
                    have_score = true;
                } else if (kn == 14 && std::memcmp(k, "polygon_coords", 14) == 0) {
                    coords.clear();
                    if (!c.numbers(coords)) return bad(c.err);
                    have_ring = true;
                } else if (!c.skip()) {
                    return bad(c.err);
                }
                c.ws();
                if (c.p < c.end && *c.p == ',') { ++c.p; continue; }
                if (c.p < c.end && *c.p == '}') { ++c.p; break; }
                return bad("expected ',' or '}'");
            }
        }
        if (!have_ring) return bad("entry without polygon_coords (RLE segmentations are not produced by this pipeline)");
        if (!have_score) return bad("entry without score");
        if (coords.size() % 2) return bad("odd number of coordinates");
        int n = (int)(coords.size() / 2);
        if (n < 4) return bad("A linearring requires at least 4 coordinates");
        const double* ring = coords.data();
        if (tolerance > 0.0) {
            simp.resize(coords.size());
            n = td_simplify_ring(coords.data(), n, tolerance, simp.data(), n);
            if (n < 0) return n;
            ring = simp.data();
        }
        // polygon.within(box): no vertex outside the closed box, and the interiors meet
        double minx = ring[0], maxx = ring[0], miny = ring[1], maxy = ring[1], area2 = 0.0;
        bool strictly = false;
        for (int i = 0; i < n; ++i) {
            const double x = ring[2 * i], y = ring[2 * i + 1];
            minx = std::fmin(minx, x);
            maxx = std::fmax(maxx, x);
            miny = std::fmin(miny, y);
            maxy = std::fmax(maxy, y);
            strictly |= x > box[0] && x < box[2] && y > box[1] && y < box[3];
            if (i + 1 < n) area2 += x * ring[2 * i + 3] - ring[2 * i + 2] * y;
        }
        const bool inside = minx >= box[0] && maxx <= box[2] && miny >= box[1] && maxy <= box[3];
        if (inside && (strictly || area2 != 0.0)) {
            // GeoPackage binary: 'GP', version 0, flags (little endian | xy envelope), srs id, envelope; then WKB polygon
            const uint8_t head[4] = {'G', 'P', 0, 0x03};
            put(out, head, 4);
            put(out, &srs_id, 4);
            const double env[4] = {minx, maxx, miny, maxy};
            put(out, env, 32);
            const uint8_t order = 1;
            const uint32_t wkb[3] = {3u, 1u, (uint32_t)n};
            put(out, &order, 1);
            put(out, wkb, 12);
            put(out, ring, (size_t)n * 16);
            offs.push_back((int64_t)out.size());
            sc.push_back(score);
        }
        c.ws();
        if (c.p < c.end && *c.p == ',') { ++c.p; continue; }
        if (c.p < c.end && *c.p == ']') break;
        return bad("expected ',' or ']'");
    }
    ++c.p;
    c.ws();
    if (c.p != c.end) return bad("trailing data");
    return TD_OK;
}

}  // namespace

extern "C" int td_stitch_tile_json(const char* json, int64_t len, const double* box, double tolerance, int32_t srs_id,
                                   uint8_t* blobs, int64_t blob_cap, int64_t* blob_offsets, double* scores, int max_features,
                                   int64_t* needed_bytes, int* needed_features) {
    if (!json || len < 0 || !box || !needed_bytes || !needed_features || std::isnan(tolerance)) {
        td_set_error("td_stitch_tile_json: bad argument");
        return TD_ERR_INVALID;
    }
    std::vector<uint8_t> out;
    std::vector<int64_t> offs{0};
    std::vector<double> sc;
    const int st = stitch_text("td_stitch_tile_json", json, len, box, tolerance, srs_id, out, offs, sc);
    if (st < 0) return st;
    *needed_bytes = (int64_t)out.size();
    *needed_features = (int)sc.size();
    if ((int64_t)out.size() > blob_cap || (int)sc.size() > max_features) {
        td_set_error("td_stitch_tile_json: %lld bytes / %d features needed, capacity %lld / %d", (long long)out.size(),
                     (int)sc.size(), (long long)blob_cap, max_features);
        return TD_ERR_CAPACITY;
    }
    if (!out.empty()) std::memcpy(blobs, out.data(), out.size());
    std::memcpy(blob_offsets, offs.data(), offs.size() * sizeof(int64_t));
    if (!sc.empty()) std::memcpy(scores, sc.data(), sc.size() * sizeof(double));
    return (int)sc.size();
}


// Whole image in one call: every tile file read, parsed, simplified, edge-filtered and encoded on `threads` host threads;
// the features come back concatenated in FILE order (the order process_folder_sync gives the layer).
extern "C" int td_stitch_tile_files(const char* paths, const int64_t* path_offsets, int n_files, const double* boxes, double tolerance,
                                    const int32_t* srs_ids, int threads, uint8_t* blobs, int64_t blob_cap, int64_t* blob_offsets,
                                    double* scores, int max_features, int32_t* file_status, int64_t* needed_bytes, int* needed_features) {
    if (!paths || !path_offsets || n_files < 0 || !boxes || !srs_ids || !file_status || !needed_bytes || !needed_features || std::isnan(tolerance) ||
        (n_files > 0 && (!blob_offsets || blob_cap < 0 || max_features < 0))) {
        td_set_error("td_stitch_tile_files: bad argument");
        return TD_ERR_INVALID;
    }
    struct Part {
        std::vector<uint8_t> out;
        std::vector<int64_t> offs;
        std::vector<double> sc;
    };
    std::vector<Part> parts;
    try {
        parts.resize((size_t)n_files);
    } catch (...) {
        td_set_error("td_stitch_tile_files: out of memory for %d files", n_files);
        return TD_ERR_INVALID;
    }
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<char> text;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_files) return;
            Part& pt = parts[(size_t)i];
            pt.offs.assign(1, 0);
            const char* path = paths + path_offsets[i];
            FILE* f = std::fopen(path, "rb");
            if (!f) { file_status[i] = TD_ERR_INVALID; continue; }
            text.clear();
            char buf[1 << 16];
            size_t got;
            bool rerr = false;
            try {
                while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.insert(text.end(), buf, buf + got);
            } catch (...) {
                rerr = true;
            }
            rerr = rerr || std::ferror(f) != 0;
            std::fclose(f);
            if (rerr) { file_status[i] = TD_ERR_INVALID; continue; }
            int st;
            try {       // no exception may leave a worker thread (std::terminate) or cross the C ABI: an allocation failure fails this file
                st = stitch_text("td_stitch_tile_files", text.data(), (int64_t)text.size(), boxes + 4 * (size_t)i, tolerance, srs_ids[i],
                                 pt.out, pt.offs, pt.sc);
            } catch (...) {
                st = TD_ERR_INVALID;
            }
            if (st < 0) {           // the file is left out (the reference's try / except around each tile file, helpers.py:419-476)
                pt.out.clear();
                pt.offs.assign(1, 0);
                pt.sc.clear();
                file_status[i] = st;
            } else {
                file_status[i] = (int32_t)pt.sc.size();
            }
        }
    };
    const int nt = std::max(1, std::min(threads, std::max(n_files, 1)));
    {
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        } catch (...) {       // no more threads to be had: the calling thread (and those that did start) do all the files
        }
        work();
        for (auto& t : pool) t.join();
    }
    int64_t bytes = 0, feats = 0;
    for (const Part& pt : parts) {
        bytes += (int64_t)pt.out.size();
        feats += (int64_t)pt.sc.size();
    }
    *needed_bytes = bytes;
    *needed_features = (int)feats;
    if (bytes > blob_cap || feats > max_features) {
        td_set_error("td_stitch_tile_files: %lld bytes / %lld features needed, capacity %lld / %d", (long long)bytes, (long long)feats,
                     (long long)blob_cap, max_features);
        return TD_ERR_CAPACITY;
    }
    int64_t bo = 0, fo = 0;
    if (n_files > 0 || blob_offsets) blob_offsets[0] = 0;
    for (const Part& pt : parts) {
        if (!pt.out.empty()) std::memcpy(blobs + bo, pt.out.data(), pt.out.size());
        for (size_t k = 0; k < pt.sc.size(); ++k) {
            blob_offsets[fo + (int64_t)k + 1] = bo + pt.offs[k + 1];
            scores[fo + (int64_t)k] = pt.sc[k];
        }
        bo += (int64_t)pt.out.size();
        fo += (int64_t)pt.sc.size();
    }
    return (int)feats;
}
