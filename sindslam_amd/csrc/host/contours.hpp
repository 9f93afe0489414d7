// Host stage: border following and contour drawing on bit images (product-side counterpart of cv::findContours
// CHAIN_APPROX_NONE with RETR_EXTERNAL / RETR_CCOMP membership, cv::contourArea, cv::arcLength, cv::drawContours
// FILLED / thickness 2, cv::floodFill FLOODFILL_MASK_ONLY) as called by reference DynaDetect.cc:605, 617, 675-715,
// 1579-1605 and PEAC/AHCPlaneFitter.hpp:398-399.  Serial, tiny, order-defined -> host (DESIGN.md "host stages").
// Border following works on a 0-padded signed-char copy of the mask's bounding box only.
#pragma once
#include <deque>
#include "bitimg.hpp"

namespace sind {

struct PtI { int x, y; };
typedef std::vector<PtI> Contour;

// Suzuki-Abe border following (OpenCV contours.cpp icvFetchContour semantics: 8-connectivity, start pixel order of the
// raster scan, pixel marks so that every border is traced once).  external_only: outermost borders only.
inline void find_contours(const BitImg& src, std::vector<Contour>& out, bool external_only, const Rect* roi = nullptr) {
    out.clear();
    Rect bb = roi ? *roi : src.bbox();
    if (bb.empty()) return;
    bb.x0 = std::max(bb.x0, 0); bb.y0 = std::max(bb.y0, 0); bb.x1 = std::min(bb.x1, src.w - 1); bb.y1 = std::min(bb.y1, src.h - 1);
    const int bw = bb.x1 - bb.x0 + 1, bh = bb.y1 - bb.y0 + 1, w = bw + 2, h = bh + 2;
    std::vector<signed char> img((size_t)w * h, 0);
    for (int y = 0; y < bh; y++) {                                     // expand the set bits of the bounding box rows
        const uint64_t* r = src.row(bb.y0 + y); signed char* o = &img[(size_t)(y + 1) * w + 1 - bb.x0];
        for (int k = bb.x0 >> 6; k <= bb.x1 >> 6; k++) { uint64_t m = r[k]; while (m) { const int x = (k << 6) + __builtin_ctzll(m); m &= m - 1; if (x >= bb.x0 && x <= bb.x1) o[x] = 1; } }
    }
    const int d8[8] = {1, -w + 1, -w, -w - 1, -1, w - 1, w, w + 1};
    int deltas[16]; for (int i = 0; i < 16; i++) deltas[i] = d8[i & 7];
    static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    const int ox = bb.x0 - 1, oy = bb.y0 - 1;
    int nbd = 2;
    for (int y = 1; y < h - 1; y++) {
        int prev = 0, lnbd_x = 0;
        signed char* rowp = &img[(size_t)y * w];
        for (int x = 1; x < w; x++) {
            int p = rowp[x];
            if (p == prev) continue;
            bool is_hole = false, start = true;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) start = false;
                else { if (prev & -2) lnbd_x = x - 1; is_hole = true; }
            }
            if (start && external_only && (is_hole || rowp[lnbd_x] > 0)) start = false;
            if (start) {
                lnbd_x = x - (is_hole ? 1 : 0);
                int px = x - (is_hole ? 1 : 0), py = y;
                signed char* i0 = rowp + px; signed char *i1, *i3, *i4 = nullptr;
                const int mark = external_only ? 2 : nbd;
                out.emplace_back(); Contour& c = out.back();
                int s, s_end; s_end = s = is_hole ? 0 : 4;
                do { s = (s - 1) & 7; i1 = i0 + deltas[s]; } while (*i1 == 0 && s != s_end);
                if (s == s_end) { *i0 = (signed char)(mark | -128); c.push_back({px + ox, py + oy}); }
                else {
                    i3 = i0;
                    for (;;) {
                        s_end = s; s = std::min(s, 15);
                        while (s < 15) { i4 = i3 + deltas[++s]; if (*i4 != 0) break; }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(mark | -128);
                        else if (*i3 == 1) *i3 = (signed char)mark;
                        c.push_back({px + ox, py + oy});
                        px += cdx[s]; py += cdy[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4; s = (s + 4) & 7;
                    }
                }
                if (!external_only) { nbd++; if (nbd > 127) nbd = 2; }
                p = is_hole ? 0 : rowp[x];
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
}

inline double contour_area(const Contour& c) {
    if (c.empty()) return 0;
    double a = 0; PtI prev = c.back();
    for (const PtI& p : c) { a += (double)prev.x * p.y - (double)prev.y * p.x; prev = p; }
    return std::fabs(a * 0.5);
}
inline double arc_length_closed(const Contour& c) {
    if (c.size() <= 1) return 0;
    double per = 0; PtI prev = c.back();
    for (const PtI& p : c) { const float dx = (float)p.x - (float)prev.x, dy = (float)p.y - (float)prev.y; per += std::sqrt(dx * dx + dy * dy); prev = p; }
    return per;
}
inline Rect contour_bbox(const Contour& c) { Rect r{1 << 30, 1 << 30, -1, -1}; for (const PtI& p : c) { r.x0 = std::min(r.x0, p.x); r.x1 = std::max(r.x1, p.x); r.y0 = std::min(r.y0, p.y); r.y1 = std::max(r.y1, p.y); } return r; }

// thickness-2 polyline of a closed unit-step chain = chain pixels dilated by the 3x3 cross
inline void draw_thick2(BitImg& img, const Contour& c) {
    for (const PtI& p : c) { img.set_safe(p.x, p.y); img.set_safe(p.x - 1, p.y); img.set_safe(p.x + 1, p.y); img.set_safe(p.x, p.y - 1); img.set_safe(p.x, p.y + 1); }
}
// FILLED drawing of one or several contours sharing one edge table: boundary + even-odd interior per scanline
inline void draw_filled(BitImg& img, const std::vector<const Contour*>& cs) {
    int y0 = 1 << 30, y1 = -1;
    for (const Contour* c : cs) for (const PtI& p : *c) { y0 = std::min(y0, p.y); y1 = std::max(y1, p.y); }
    if (y1 < y0) return;
    std::vector<std::vector<int>> cross(y1 - y0 + 1);
    for (const Contour* cp : cs) {
        const Contour& c = *cp; const size_t n = c.size();
        for (size_t i = 0; i < n; i++) {
            const PtI& p = c[i]; const PtI& q = c[(i + 1) % n];
            img.set_safe(p.x, p.y);
            if (p.y == q.y) continue;
            const PtI& top = p.y < q.y ? p : q;
            cross[top.y - y0].push_back(top.x);
        }
    }
    for (int y = y0; y <= y1; y++) {
        if (y < 0 || y >= img.h) continue;
        std::vector<int>& xs = cross[y - y0];
        if (xs.size() < 2) continue;
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2) {
            const int a = std::max(xs[k], 0), b = std::min(xs[k + 1], img.w - 1);
            if (b < a) continue;
            uint64_t* r = img.row(y);
            const int ka = a >> 6, kb = b >> 6;
            for (int q = ka; q <= kb; q++) {
                uint64_t m = ~0ull;
                if (q == ka) m &= ~0ull << (a & 63);
                if (q == kb) m &= (b & 63) == 63 ? ~0ull : ((1ull << ((b & 63) + 1)) - 1);
                r[q] |= m;
            }
        }
    }
}
inline void draw_filled(BitImg& img, const Contour& c) { std::vector<const Contour*> one(1, &c); draw_filled(img, one); }

// floodFill(FLOODFILL_MASK_ONLY, 8-connected) on a two-valued image: `same` holds the pixels whose value equals the
// seed's value, `blocked` the non-zero mask pixels; newly filled pixels are added to `filled` and to `blocked`.
inline int flood_fill(const BitImg& same, BitImg& blocked, BitImg& filled, PtI seed) {
    if (seed.x < 0 || seed.y < 0 || seed.x >= same.w || seed.y >= same.h) return 0;
    if (blocked.get(seed.x, seed.y)) return 0;
    std::deque<PtI> q; q.push_back(seed); blocked.set(seed.x, seed.y); filled.set(seed.x, seed.y);
    int area = 0;
    while (!q.empty()) {
        const PtI p = q.front(); q.pop_front(); area++;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            const int x = p.x + dx, y = p.y + dy;
            if (x < 0 || y < 0 || x >= same.w || y >= same.h) continue;
            if (blocked.get(x, y) || !same.get(x, y)) continue;
            blocked.set(x, y); filled.set(x, y); q.push_back({x, y});
        }
    }
    return area;
}

}  // namespace sind
