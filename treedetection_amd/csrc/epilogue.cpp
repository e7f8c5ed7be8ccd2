// Host epilogue of one tile: packed instance masks → polygons → the text of the per-tile prediction file
// (reference TreeDetection/prediction.py:229-261, `_process_and_save_single`; affine of utilities.py:182-207).
//
// Per detection: unpack the paste region from the engine's bit rows, follow every border (contours.cpp), keep
// contours with >= 4 points (`contour.size >= 8`), close the ring, map the integer pixel-corner coordinates through
// the tile's affine in float64 (x' = a*col + b*row + c, y' = d*col + e*row + f, evaluated left to right like the
// numpy expression) and emit {"image_id", "category_id", "score", "polygon_coords": [[[x, y], ...]]}. The text is
// byte-for-byte what `json.dumps(evaluations)` produces (default separators, ensure_ascii, float repr = shortest
// round-trip digits laid out by CPython's rule), so a file written from it cannot be told from the reference's.
// Pure host code; the Python host calls it from worker threads with the GIL released.
#include "common.h"

#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

void td_trace_contours_bits(const uint32_t* rows, int words_per_row, int h, int w, std::vector<int32_t>& pts,
                            std::vector<int32_t>& starts);

namespace {

// repr(float) of CPython (float_repr_style 'short'): shortest digits that round-trip; fixed notation when
// -4 < decpt <= 16, otherwise d[.ddd]e±XX with at least two exponent digits.
void append_double(std::string& out, double v) {
    if (std::isnan(v)) {
        out += "NaN";
        return;
    }
    if (std::isinf(v)) {
        out += v < 0 ? "-Infinity" : "Infinity";
        return;
    }
    if (v == 0.0) {
        out += std::signbit(v) ? "-0.0" : "0.0";
        return;
    }
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof(sci) - 1, v, std::chars_format::scientific);
    *r.ptr = 0;
    const char* p = sci;
    if (*p == '-') {
        out += '-';
        ++p;
    }
    char digits[24];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    const int e10 = std::atoi(p + 1);          // "e+05" / "e-07"
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    const int decpt = e10 + 1;
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            out += "0.";
            out.append((size_t)-decpt, '0');
            out.append(digits, nd);
        } else if (decpt < nd) {
            out.append(digits, decpt);
            out += '.';
            out.append(digits + decpt, nd - decpt);
        } else {
            out.append(digits, nd);
            out.append((size_t)(decpt - nd), '0');
            out += ".0";
        }
    } else {
        out += digits[0];
        if (nd > 1) {
            out += '.';
            out.append(digits + 1, nd - 1);
        }
        const int e = decpt - 1;
        out += 'e';
        out += e < 0 ? '-' : '+';
        char eb[16];
        std::snprintf(eb, sizeof(eb), "%02d", e < 0 ? -e : e);
        out += eb;
    }
}

// json.dumps(str) with ensure_ascii=True: UTF-8 in, \uXXXX (surrogate pairs above the BMP) out.
void append_json_string(std::string& out, const char* s) {
    out += '"';
    const unsigned char* p = (const unsigned char*)s;
    char buf[16];
    while (*p) {
        uint32_t cp;
        int len;
        if (*p < 0x80) { cp = *p; len = 1; }
        else if ((*p >> 5) == 6 && (p[1] & 0xc0) == 0x80) { cp = ((*p & 0x1f) << 6) | (p[1] & 0x3f); len = 2; }
        else if ((*p >> 4) == 14 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80) {
            cp = ((*p & 0x0f) << 12) | ((p[1] & 0x3f) << 6) | (p[2] & 0x3f); len = 3;
        } else if ((*p >> 3) == 30 && (p[1] & 0xc0) == 0x80 && (p[2] & 0xc0) == 0x80 && (p[3] & 0xc0) == 0x80) {
            cp = ((*p & 0x07) << 18) | ((p[1] & 0x3f) << 12) | ((p[2] & 0x3f) << 6) | (p[3] & 0x3f); len = 4;
        } else { cp = 0xdc00 | *p; len = 1; }      // undecodable byte: what os.fsdecode's surrogateescape yields
        p += len;
        switch (cp) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (cp >= 0x20 && cp < 0x7f) out += (char)cp;
                else if (cp < 0x10000) {
                    std::snprintf(buf, sizeof(buf), "\\u%04x", cp);
                    out += buf;
                } else {
                    const uint32_t v = cp - 0x10000;
                    std::snprintf(buf, sizeof(buf), "\\u%04x\\u%04x", 0xd800 | (v >> 10), 0xdc00 | (v & 0x3ff));
                    out += buf;
                }
        }
    }
    out += '"';
}

}  // namespace

namespace {

// Where the contours of one tile come from: the host tracer on the packed bit rows, or — for detections the device
// tracer handled (status 0) — the points td_trace_contours_dev wrote (already in tile coordinates).
struct DevContours {
    const int16_t* points;       // this image's [points_cap][2]
    int64_t points_cap;
    const int32_t* det_info;     // [n][4] status, contour count, first point, total points
    const int32_t* contour_info; // [n][TD_CONTOUR_MAX][2] (offset, count) in RETR_TREE order
};

int polygons_json_text(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits, int64_t mask_words,
                       const DevContours* dev, const float* scores, const int32_t* classes, int n, const double* transform,
                       const char* image_id, std::string& out, const char* who) {
    const double a = transform[0], b = transform[1], c = transform[2], d = transform[3], e = transform[4], f = transform[5];
    // North-up rasters (b == d == 0, the usual case): x depends on the column only and y on the row only, so each
    // distinct column / row is converted to text once per tile and copied from then on (number formatting is
    // otherwise ~80 % of this function).
    const bool axis_aligned = b == 0.0 && d == 0.0;
    struct Cached { char txt[27]; uint8_t len; };
    std::vector<Cached> xcache, ycache;
    auto cached = [&](std::vector<Cached>& cache, int idx, double value, std::string& out) {
        if (idx < 0 || idx >= (1 << 20)) {
            append_double(out, value);
            return;
        }
        if ((size_t)idx >= cache.size()) cache.resize((size_t)idx + 64, Cached{{0}, 0});
        Cached& cd = cache[(size_t)idx];
        if (cd.len == 0) {
            std::string t;
            append_double(t, value);
            if (t.size() > sizeof(cd.txt)) {
                out += t;
                return;
            }
            std::memcpy(cd.txt, t.data(), t.size());
            cd.len = (uint8_t)t.size();
        }
        out.append(cd.txt, cd.len);
    };
    std::string id;
    append_json_string(id, image_id);
    out.clear();
    out.reserve(1 << 16);
    out += '[';
    int entries = 0;
    // one entry of the file: contour points (tile pixel corners) -> ring closed -> affine -> text
    auto emit = [&](int k, int np, auto&& point) {
        if (np < 4) return;
        if (entries++) out += ", ";
        out += "{\"image_id\": ";
        out += id;
        out += ", \"category_id\": ";
        out += std::to_string(classes ? classes[k] : 0);
        out += ", \"score\": ";
        append_double(out, (double)scores[k]);
        out += ", \"polygon_coords\": [[";
        int fx, fy, lx, ly;
        point(0, fx, fy);
        point(np - 1, lx, ly);
        const int total = np + ((fx == lx && fy == ly) ? 0 : 1);
        for (int i = 0; i < total; ++i) {
            int ic, ir;
            point(i < np ? i : 0, ic, ir);
            const double col = (double)ic, row = (double)ir;
            const double gx = (a * col + b * row) + c;
            const double gy = (d * col + e * row) + f;
            if (i) out += ", ";
            out += '[';
            if (axis_aligned) cached(xcache, ic, gx, out);
            else append_double(out, gx);
            out += ", ";
            if (axis_aligned) cached(ycache, ir, gy, out);
            else append_double(out, gy);
            out += ']';
        }
        out += "]]}";
    };
    std::vector<int32_t> pts, starts;
    for (int k = 0; k < n; ++k) {
        const int x0 = mask_region[4 * k], y0 = mask_region[4 * k + 1], x1 = mask_region[4 * k + 2], y1 = mask_region[4 * k + 3];
        if (x1 <= x0 || y1 <= y0) continue;
        if ((int64_t)x1 - x0 > (1 << 20) || (int64_t)y1 - y0 > (1 << 20)) {     // no raster tile is that large: corrupt record
            td_set_error("%s: detection %d has an impossible paste region [%d,%d,%d,%d]", who, k, x0, y0, x1, y1);
            return TD_ERR_INVALID;
        }
        if (dev && dev->det_info[4 * k] == 0) {                    // traced on the device
            const int nc = dev->det_info[4 * k + 1];
            const int64_t base = dev->det_info[4 * k + 2];
            if (nc < 0 || nc > TD_CONTOUR_MAX || base < 0 || base + dev->det_info[4 * k + 3] > dev->points_cap) {
                td_set_error("%s: detection %d has an inconsistent device contour record", who, k);
                return TD_ERR_INVALID;
            }
            for (int ci = 0; ci < nc; ++ci) {
                const int32_t* rec = dev->contour_info + ((size_t)k * TD_CONTOUR_MAX + ci) * 2;
                if (rec[0] < 0 || rec[1] < 0 || base + (int64_t)rec[0] + rec[1] > dev->points_cap) {
                    td_set_error("%s: detection %d contour %d points outside the point buffer", who, k, ci);
                    return TD_ERR_INVALID;
                }
                const int16_t* p = dev->points + 2 * (base + rec[0]);
                emit(k, rec[1], [&](int i, int& x, int& y) { x = p[2 * i]; y = p[2 * i + 1]; });
            }
            continue;
        }
        if (!mask_bits) {
            td_set_error("%s: detection %d was left to the host tracer (device status %d) but no mask bits were passed", who, k,
                         dev ? dev->det_info[4 * k] : -1);
            return TD_ERR_STATE;
        }
        const int w = x1 - x0, h = y1 - y0, wpr = (w + 31) / 32;
        const int64_t o = mask_offset[k];
        if (o < 0 || o + (int64_t)wpr * h > mask_words) {
            td_set_error("%s: detection %d rows [%lld, +%lld) outside the %lld-word mask buffer", who, k, (long long)o,
                         (long long)wpr * h, (long long)mask_words);
            return TD_ERR_INVALID;
        }
        td_trace_contours_bits(mask_bits + o, wpr, h, w, pts, starts);
        const int nc = (int)starts.size() - 1;
        for (int ci = 0; ci < nc; ++ci) {
            const int p0 = starts[ci];
            emit(k, starts[ci + 1] - p0, [&](int i, int& x, int& y) { x = pts[2 * (p0 + i)] + x0; y = pts[2 * (p0 + i) + 1] + y0; });
        }
    }
    out += ']';
    return entries;
}

int polygons_json_core(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits, int64_t mask_words,
                       const DevContours* dev, const float* scores, const int32_t* classes, int n, const double* transform,
                       const char* image_id, char* buf, int64_t cap, int64_t* needed, const char* who) {
    std::string out;
    const int entries = polygons_json_text(mask_region, mask_offset, mask_bits, mask_words, dev, scores, classes, n, transform,
                                           image_id, out, who);
    if (entries < 0) return entries;
    *needed = (int64_t)out.size();
    if ((int64_t)out.size() > cap) {
        td_set_error("%s: %lld bytes needed, capacity %lld", who, (long long)out.size(), (long long)cap);
        return TD_ERR_CAPACITY;
    }
    std::memcpy(buf, out.data(), out.size());
    return entries;
}

}  // namespace

// The text of one tile's prediction file from its packed rows on the host (common.h; td_tile_prediction_file in api.cpp
// fetches the rows from the device first and writes the text to the file itself).
int td_polygons_json_text(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits, int64_t mask_words,
                          const float* scores, const int32_t* classes, int n, const double* transform, const char* image_id,
                          std::string& out, const char* who) {
    return polygons_json_text(mask_region, mask_offset, mask_bits, mask_words, nullptr, scores, classes, n, transform, image_id, out,
                              who);
}

extern "C" int td_tile_polygons_json(const int32_t* mask_region, const int64_t* mask_offset, const uint32_t* mask_bits,
                                     int64_t mask_words, const float* scores, const int32_t* classes, int n,
                                     const double* transform, const char* image_id, char* buf, int64_t cap,
                                     int64_t* needed) {
    if (n < 0 || !transform || !image_id || !needed || (n > 0 && (!mask_region || !mask_offset || !mask_bits || !scores)) ||
        (cap > 0 && !buf)) {
        td_set_error("td_tile_polygons_json: bad argument");
        return TD_ERR_INVALID;
    }
    return polygons_json_core(mask_region, mask_offset, mask_bits, mask_words, nullptr, scores, classes, n, transform, image_id, buf,
                              cap, needed, "td_tile_polygons_json");
}

extern "C" int td_tile_polygons_json_dev(const int16_t* points, int64_t points_cap, const int32_t* det_info,
                                         const int32_t* contour_info, const int32_t* mask_region, const int64_t* mask_offset,
                                         const uint32_t* mask_bits, int64_t mask_words, const float* scores,
                                         const int32_t* classes, int n, const double* transform, const char* image_id,
                                         char* buf, int64_t cap, int64_t* needed) {
    if (n < 0 || !transform || !image_id || !needed || (cap > 0 && !buf) ||
        (n > 0 && (!points || !det_info || !contour_info || !mask_region || !scores || points_cap < 0))) {
        td_set_error("td_tile_polygons_json_dev: bad argument");
        return TD_ERR_INVALID;
    }
    const DevContours dev{points, points_cap, det_info, contour_info};
    return polygons_json_core(mask_region, mask_offset, mask_bits, mask_words, &dev, scores, classes, n, transform, image_id, buf, cap,
                              needed, "td_tile_polygons_json_dev");
}
