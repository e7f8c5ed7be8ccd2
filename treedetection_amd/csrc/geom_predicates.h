// Exact-sign planar predicates shared by the geometry code (geometry.cpp, region.cpp): orientation of three points
// with a floating-point filter backed by double-double arithmetic — the scheme GEOS uses
// (CGAlgorithmsDD::orientationIndex), so collinear / touching configurations are classified the same way.
#pragma once
#include <cmath>

namespace tdgeom {

struct DD {
    double hi, lo;
};
inline DD two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
inline DD two_prod(double a, double b) {
    const double p = a * b;
    return {p, std::fma(a, b, -p)};
}
inline DD dd_add(DD a, DD b) {
    DD s = two_sum(a.hi, b.hi);
    DD t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = two_sum(s.hi, s.lo);     // renormalise
    s.lo += t.lo;
    return two_sum(s.hi, s.lo);
}
inline DD dd_neg(DD a) { return {-a.hi, -a.lo}; }
inline DD dd_mul(DD a, DD b) {
    DD p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return two_sum(p.hi, p.lo);
}
inline DD dd_diff(double a, double b) { return two_sum(a, -b); }   // exact a - b

// sign of the orientation of q relative to the directed line p1 -> p2 (+1 left, -1 right, 0 collinear)
inline int orientation(double p1x, double p1y, double p2x, double p2y, double qx, double qy) {
    // fast filter: the double determinant decides unless it is within its rounding error bound
    const double detleft = (p1x - qx) * (p2y - qy);
    const double detright = (p1y - qy) * (p2x - qx);
    const double det = detleft - detright;
    const auto sign = [](double v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); };
    double detsum;
    if (detleft > 0.0) {
        if (detright <= 0.0) return sign(det);
        detsum = detleft + detright;
    } else if (detleft < 0.0) {
        if (detright >= 0.0) return sign(det);
        detsum = -detleft - detright;
    } else {
        return sign(det);
    }
    const double errbound = 1e-15 * detsum;
    if (det >= errbound || -det >= errbound) return sign(det);
    // double-double evaluation of (p2 - p1) x (q - p2)
    const DD dx1 = dd_diff(p2x, p1x), dy1 = dd_diff(p2y, p1y);
    const DD dx2 = dd_diff(qx, p2x), dy2 = dd_diff(qy, p2y);
    const DD d = dd_add(dd_mul(dx1, dy2), dd_neg(dd_mul(dy1, dx2)));
    if (d.hi > 0 || (d.hi == 0 && d.lo > 0)) return 1;
    if (d.hi < 0 || (d.hi == 0 && d.lo < 0)) return -1;
    return 0;
}

struct Pt {
    double x, y;
    bool operator==(const Pt& o) const { return x == o.x && y == o.y; }
};

inline bool env_has(const Pt& a, const Pt& b, const Pt& q) {   // q inside the envelope of segment (a, b)
    return q.x >= std::fmin(a.x, b.x) && q.x <= std::fmax(a.x, b.x) && q.y >= std::fmin(a.y, b.y) && q.y <= std::fmax(a.y, b.y);
}
inline bool env_overlap(const Pt& p1, const Pt& p2, const Pt& q1, const Pt& q2) {
    return !(std::fmin(q1.x, q2.x) > std::fmax(p1.x, p2.x) || std::fmax(q1.x, q2.x) < std::fmin(p1.x, p2.x) ||
             std::fmin(q1.y, q2.y) > std::fmax(p1.y, p2.y) || std::fmax(q1.y, q2.y) < std::fmin(p1.y, p2.y));
}


}  // namespace tdgeom
