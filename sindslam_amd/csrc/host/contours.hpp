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
    // byte copy of the box, widened to whole 64-pixel words (8 bits -> 8 bytes per step instead of one store per set pixel) plus a
    // one-pixel empty frame; pixels of the words that lie outside the box columns are masked off
    const int kx0 = bb.x0 >> 6, kx1 = bb.x1 >> 6, xa = kx0 << 6;
    const int bw = (kx1 - kx0 + 1) << 6, bh = bb.y1 - bb.y0 + 1, w = bw + 2, h = bh + 2;
    std::vector<signed char> img((size_t)w * h, 0);
    const uint64_t first = ~0ull << (bb.x0 & 63), last = (bb.x1 & 63) == 63 ? ~0ull : ((1ull << ((bb.x1 & 63) + 1)) - 1);
    for (int y = 0; y < bh; y++) {
        const uint64_t* r = src.row(bb.y0 + y); signed char* o = &img[(size_t)(y + 1) * w + 1];
        for (int k = kx0; k <= kx1; k++) {
            uint64_t m = r[k]; if (k == kx0) m &= first; if (k == kx1) m &= last;
            if (!m) continue;
            for (int q = 0; q < 8; q++) { const uint64_t v = spread8((unsigned)(m >> (8 * q)) & 0xffu) & 0x0101010101010101ull; std::memcpy(o + ((k - kx0) << 6) + 8 * q, &v, 8); }
        }
    }
    const int d8[8] = {1, -w + 1, -w, -w - 1, -1, w - 1, w, w + 1};
    int deltas[16]; for (int i = 0; i < 16; i++) deltas[i] = d8[i & 7];
    static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    const int ox = xa - 1, oy = bb.y0 - 1;
    int nbd = 2;
    for (int y = 1; y < h - 1; y++) {
        int prev = 0, lnbd_x = 0;
        signed char* rowp = &img[(size_t)y * w];
        for (int x = 1; x < w; x++) {
            int p = rowp[x];
            if (p == prev) {            // nothing happens inside a run of equal values (background, blob interior): skip it eight pixels at a time
                const uint64_t pat = (uint64_t)(uint8_t)prev * 0x0101010101010101ull;
                while (x + 8 < w) { uint64_t v; std::memcpy(&v, rowp + x + 1, 8); if (v != pat) break; x += 8; }
                continue;
            }
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
    std::vector<uint32_t> cross;                 // (row << 16 | column + 1) of every edge crossing, sorted = rows in order, columns ascending
    for (const Contour* cp : cs) {
        const Contour& c = *cp; const size_t n = c.size();
        for (size_t i = 0; i < n; i++) {
            const PtI& p = c[i]; const PtI& q = c[(i + 1) % n];
            img.set_safe(p.x, p.y);
            if (p.y == q.y) continue;
            const PtI& top = p.y < q.y ? p : q;
            if (top.y >= 0 && top.y < img.h) cross.push_back(((uint32_t)top.y << 16) | (uint32_t)(std::min(std::max(top.x, -1), 65533) + 1));
        }
    }
    std::sort(cross.begin(), cross.end());
    for (size_t i = 0; i < cross.size();) {
        size_t j = i; const uint32_t yk = cross[i] >> 16;
        while (j < cross.size() && (cross[j] >> 16) == yk) j++;
        const int y = (int)yk; uint64_t* r = img.row(y);
        for (size_t k = i; k + 1 < j; k += 2) {
            const int a = std::max((int)(cross[k] & 0xffffu) - 1, 0), b = std::min((int)(cross[k + 1] & 0xffffu) - 1, img.w - 1);
            if (b < a) continue;
            const int ka = a >> 6, kb = b >> 6;
            for (int q = ka; q <= kb; q++) {
                uint64_t m = ~0ull;
                if (q == ka) m &= ~0ull << (a & 63);
                if (q == kb) m &= (b & 63) == 63 ? ~0ull : ((1ull << ((b & 63) + 1)) - 1);
                r[q] |= m;
            }
        }
        i = j;
    }
}
inline void draw_filled(BitImg& img, const Contour& c) { std::vector<const Contour*> one(1, &c); draw_filled(img, one); }

// floodFill(FLOODFILL_MASK_ONLY, 8-connected) on a two-valued image: `same` holds the pixels whose value equals the
// seed's value, `blocked` the non-zero mask pixels; newly filled pixels are added to `filled` and to `blocked`.
// Span version: the filled set of a flood fill does not depend on the visiting order, so rows are filled run by run on the bit
// words (allowed = same & ~blocked) and only run end points are pushed; cost ~ number of runs instead of number of pixels.
inline int flood_fill(const BitImg& same, BitImg& blocked, BitImg& filled, PtI seed) {
    const int W = same.w, H = same.h, wpr = same.wpr;
    if (seed.x < 0 || seed.y < 0 || seed.x >= W || seed.y >= H) return 0;
    if (blocked.get(seed.x, seed.y)) return 0;
    auto allowed_word = [&](int y, int k) -> uint64_t { uint64_t a = same.row(y)[k] & ~blocked.row(y)[k]; if (k == wpr - 1) a &= same.tail_mask(); return a; };
    // first allowed pixel in [x, xmax] of row y, or -1
    auto next_allowed = [&](int y, int x, int xmax) -> int {
        for (int k = x >> 6; k <= (xmax >> 6); k++) {
            uint64_t a = allowed_word(y, k); if (k == (x >> 6)) a &= ~0ull << (x & 63);
            if (a) { const int p = (k << 6) + __builtin_ctzll(a); return p <= xmax ? p : -1; }
        }
        return -1;
    };
    // last pixel of the allowed run that contains x (x is allowed), scanning right / left
    auto run_right = [&](int y, int x) -> int {
        for (int k = x >> 6; k < wpr; k++) {
            uint64_t na = ~allowed_word(y, k); if (k == (x >> 6)) na &= ~0ull << (x & 63);
            if (k == wpr - 1) na |= ~same.tail_mask();
            if (na) return (k << 6) + __builtin_ctzll(na) - 1;
        }
        return W - 1;
    };
    auto run_left = [&](int y, int x) -> int {
        for (int k = x >> 6; k >= 0; k--) {
            uint64_t na = ~allowed_word(y, k); if (k == (x >> 6)) na &= (x & 63) == 63 ? ~0ull : ((1ull << ((x & 63) + 1)) - 1);
            if (na) return (k << 6) + 64 - __builtin_clzll(na);
        }
        return 0;
    };
    auto mark = [&](int y, int a, int b) {
        uint64_t* bl = blocked.row(y); uint64_t* fl = filled.row(y);
        for (int k = a >> 6; k <= (b >> 6); k++) {
            uint64_t m = ~0ull; if (k == (a >> 6)) m &= ~0ull << (a & 63); if (k == (b >> 6)) m &= (b & 63) == 63 ? ~0ull : ((1ull << ((b & 63) + 1)) - 1);
            bl[k] |= m; fl[k] |= m;
        }
    };
    struct Span { int y, a, b; };
    std::vector<Span> st; int area = 0;
    // the seed pixel is filled unconditionally (like the queue version), then its row neighbourhood is scanned
    { uint64_t* bl = blocked.row(seed.y); uint64_t* fl = filled.row(seed.y); bl[seed.x >> 6] |= 1ull << (seed.x & 63); fl[seed.x >> 6] |= 1ull << (seed.x & 63); area++; }
    st.push_back({seed.y, seed.x - 1, seed.x + 1});
    if (seed.y > 0) st.push_back({seed.y - 1, seed.x - 1, seed.x + 1});
    if (seed.y + 1 < H) st.push_back({seed.y + 1, seed.x - 1, seed.x + 1});
    while (!st.empty()) {
        const Span s = st.back(); st.pop_back();
        int x = std::max(s.a, 0); const int xmax = std::min(s.b, W - 1);
        while (x <= xmax) {
            const int p = next_allowed(s.y, x, xmax); if (p < 0) break;
            const int l = run_left(s.y, p), r = run_right(s.y, p);
            mark(s.y, l, r); area += r - l + 1;
            if (s.y > 0) st.push_back({s.y - 1, l - 1, r + 1});
            if (s.y + 1 < H) st.push_back({s.y + 1, l - 1, r + 1});
            x = r + 2;
        }
    }
    return area;
}

}  // namespace sind
