// Host stage support: bit-packed binary images (64 pixels per word) with elliptical morphology, set algebra and
// bounding boxes.  Every mask DynaDetect manipulates between its per-pixel stages is two-valued, so the serial
// "graph" stages (contours, pieces, flood fill; see DESIGN.md "host stages") run on these instead of byte images:
// a 640x480 dilation is ~5 k word operations.  Semantics follow OpenCV: dilate/erode take the max/min over the
// element positions that fall INSIDE the image (morphologyDefaultBorderValue), element = getStructuringElement
// (MORPH_ELLIPSE, n x n) with anchor (n/2, n/2), no kernel flip.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sind {

struct Rect { int x0, y0, x1, y1; bool empty() const { return x1 < x0 || y1 < y0; } };   // inclusive

struct EllipseElem {
    int n = 1, ax = 0, ay = 0; std::vector<int> j1, j2;
    explicit EllipseElem(int n_ = 1) : n(n_), ax(n_ / 2), ay(n_ / 2), j1(n_, 0), j2(n_, 0) {
        if (n == 1) { j2[0] = 1; return; }
        const int r = n / 2, c = n / 2; const double inv_r2 = r ? 1. / ((double)r * r) : 0;
        for (int i = 0; i < n; i++) { const int dy = i - r; if (std::abs(dy) <= r) { const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2)); j1[i] = std::max(c - dx, 0); j2[i] = std::min(c + dx + 1, n); } }
    }
};

class BitImg {
public:
    int w = 0, h = 0, wpr = 0;
    std::vector<uint64_t> d;
    BitImg() {}
    BitImg(int w_, int h_) { create(w_, h_); }
    void create(int w_, int h_) { w = w_; h = h_; wpr = (w + 63) >> 6; d.assign((size_t)wpr * h, 0); }
    void clear() { std::fill(d.begin(), d.end(), 0); }
    uint64_t* row(int y) { return d.data() + (size_t)y * wpr; }
    const uint64_t* row(int y) const { return d.data() + (size_t)y * wpr; }
    bool get(int x, int y) const { return (row(y)[x >> 6] >> (x & 63)) & 1; }
    void set(int x, int y) { row(y)[x >> 6] |= 1ull << (x & 63); }
    void set_safe(int x, int y) { if (x >= 0 && y >= 0 && x < w && y < h) set(x, y); }
    uint64_t tail_mask() const { const int r = w & 63; return r ? ((1ull << r) - 1) : ~0ull; }

    static BitImg from_u8(const uint8_t* p, int w, int h, int stride) {
        BitImg b(w, h);
        for (int y = 0; y < h; y++) { const uint8_t* s = p + (size_t)y * stride; uint64_t* r = b.row(y);
            for (int x = 0; x < w; x++) if (s[x]) r[x >> 6] |= 1ull << (x & 63); }
        return b;
    }
    static BitImg from_equal(const uint8_t* p, int w, int h, int stride, uint8_t v) {
        BitImg b(w, h);
        for (int y = 0; y < h; y++) { const uint8_t* s = p + (size_t)y * stride; uint64_t* r = b.row(y);
            for (int x = 0; x < w; x++) if (s[x] == v) r[x >> 6] |= 1ull << (x & 63); }
        return b;
    }
    void to_u8(uint8_t* out, int stride, uint8_t on, uint8_t off = 0) const {
        for (int y = 0; y < h; y++) { const uint64_t* r = row(y); uint8_t* o = out + (size_t)y * stride; for (int x = 0; x < w; x++) o[x] = ((r[x >> 6] >> (x & 63)) & 1) ? on : off; }
    }
    void paint_u8(uint8_t* out, int stride, uint8_t v) const {       // out[p] = v where the bit is set
        for (int y = 0; y < h; y++) { const uint64_t* r = row(y); uint8_t* o = out + (size_t)y * stride;
            for (int k = 0; k < wpr; k++) { uint64_t m = r[k]; while (m) { const int b = __builtin_ctzll(m); m &= m - 1; o[(k << 6) + b] = v; } } }
    }
    int count() const { int n = 0; for (uint64_t v : d) n += __builtin_popcountll(v); return n; }
    bool any() const { for (uint64_t v : d) if (v) return true; return false; }
    BitImg& operator&=(const BitImg& o) { for (size_t i = 0; i < d.size(); i++) d[i] &= o.d[i]; return *this; }
    BitImg& operator|=(const BitImg& o) { for (size_t i = 0; i < d.size(); i++) d[i] |= o.d[i]; return *this; }
    BitImg& andnot(const BitImg& o) { for (size_t i = 0; i < d.size(); i++) d[i] &= ~o.d[i]; return *this; }       // this - o (saturating subtract of masks)
    BitImg inverted() const { BitImg r = *this; const uint64_t tm = tail_mask(); for (int y = 0; y < h; y++) { uint64_t* p = r.row(y); for (int k = 0; k < wpr; k++) p[k] = ~p[k]; p[wpr - 1] &= tm; } return r; }
    static int and_count(const BitImg& a, const BitImg& b) { int n = 0; for (size_t i = 0; i < a.d.size(); i++) n += __builtin_popcountll(a.d[i] & b.d[i]); return n; }
    Rect bbox() const {
        Rect r{w, h, -1, -1};
        for (int y = 0; y < h; y++) { const uint64_t* p = row(y);
            for (int k = 0; k < wpr; k++) if (p[k]) { r.y0 = std::min(r.y0, y); r.y1 = y; r.x0 = std::min(r.x0, (k << 6) + __builtin_ctzll(p[k])); r.x1 = std::max(r.x1, (k << 6) + 63 - __builtin_clzll(p[k])); } }
        return r;
    }

    // dst[x] |= src[x + s] for one row of wpr words (bits shifted out of the row are dropped)
    static void or_shift(uint64_t* dst, const uint64_t* src, int wpr, int s) {
        if (s >= 0) { const int ws = s >> 6, bs = s & 63;
            for (int k = 0; k + ws < wpr; k++) { uint64_t v = src[k + ws] >> bs; if (bs && k + ws + 1 < wpr) v |= src[k + ws + 1] << (64 - bs); dst[k] |= v; } }
        else { const int t = -s, ws = t >> 6, bs = t & 63;
            for (int k = wpr - 1; k - ws >= 0; k--) { uint64_t v = src[k - ws] << bs; if (bs && k - ws - 1 >= 0) v |= src[k - ws - 1] >> (64 - bs); dst[k] |= v; } }
    }

    // dilation by an elliptical element; rows outside [ry0, ry1] of the SOURCE are known to be empty (bbox hint)
    BitImg dilated(const EllipseElem& e, int ry0 = 0, int ry1 = -1) const {
        if (ry1 < 0) ry1 = h - 1;
        ry0 = std::max(ry0, 0); ry1 = std::min(ry1, h - 1);
        BitImg out(w, h);
        if (ry1 < ry0) return out;
        // distinct horizontal runs of the element
        std::vector<std::pair<int, int>> runs; std::vector<int> run_of(e.n, -1);
        for (int i = 0; i < e.n; i++) { if (e.j2[i] <= e.j1[i]) continue; std::pair<int, int> r(e.j1[i], e.j2[i]);
            size_t k = 0; for (; k < runs.size(); k++) if (runs[k] == r) break; if (k == runs.size()) runs.push_back(r); run_of[i] = (int)k; }
        const int nr = ry1 - ry0 + 1; const uint64_t tm = tail_mask();
        std::vector<std::vector<uint64_t>> H(runs.size(), std::vector<uint64_t>((size_t)nr * wpr, 0));
        std::vector<uint64_t> acc(wpr), tmp(wpr);
        for (size_t k = 0; k < runs.size(); k++) {
            const int a = runs[k].first - e.ax, L = runs[k].second - runs[k].first;     // shifts a .. a+L-1
            for (int y = ry0; y <= ry1; y++) {
                const uint64_t* s = row(y);
                bool nz = false; for (int q = 0; q < wpr; q++) nz |= s[q] != 0;
                if (!nz) continue;
                // every run of an elliptical element contains the anchor column (a <= 0 <= a+L-1): build the window as
                // OR_{t=0..mp} src[x+t]  |  OR_{t=0..mn} src[x-t], each by doubling, so partial windows at the borders survive
                uint64_t* hrow = &H[k][(size_t)(y - ry0) * wpr];
                for (int dir = 0; dir < 2; dir++) {
                    const int m = dir == 0 ? a + L - 1 : -a, sgn = dir == 0 ? 1 : -1;
                    std::copy(s, s + wpr, acc.begin());
                    int p = 1;                                          // acc covers t = 0..p-1
                    while (p * 2 <= m + 1) { std::copy(acc.begin(), acc.end(), tmp.begin()); or_shift(acc.data(), tmp.data(), wpr, sgn * p); p *= 2; }
                    if (p < m + 1) { std::copy(acc.begin(), acc.end(), tmp.begin()); or_shift(acc.data(), tmp.data(), wpr, sgn * (m + 1 - p)); }
                    for (int q = 0; q < wpr; q++) hrow[q] |= acc[q];
                }
                hrow[wpr - 1] &= tm;
            }
        }
        for (int i = 0; i < e.n; i++) {
            if (run_of[i] < 0) continue;
            const int dy = i - e.ay;                               // dst(y) takes src(y + dy)  ->  src row ys feeds dst row ys - dy
            for (int ys = ry0; ys <= ry1; ys++) { const int yd = ys - dy; if (yd < 0 || yd >= h) continue;
                const uint64_t* hrow = &H[run_of[i]][(size_t)(ys - ry0) * wpr]; uint64_t* o = out.row(yd);
                for (int q = 0; q < wpr; q++) o[q] |= hrow[q]; }
        }
        return out;
    }
    // erosion: positions outside the image are ignored, i.e. erode(X) = ~dilate(~X) inside the image
    BitImg eroded(const EllipseElem& e) const { return inverted().dilated(e).inverted(); }
    // erosion of an image whose set pixels all lie in rows [y0, y1]: the result is a subset of the source, so only those rows
    // are evaluated (complement rows further than the element height away cannot influence them)
    BitImg eroded_rows(const EllipseElem& e, int y0, int y1) const {
        y0 = std::max(y0, 0); y1 = std::min(y1, h - 1);
        BitImg out(w, h);
        if (y1 < y0) return out;
        const BitImg d = inverted().dilated(e, y0 - e.n, y1 + e.n);
        const uint64_t tm = tail_mask();
        for (int y = y0; y <= y1; y++) { const uint64_t* p = d.row(y); uint64_t* o = out.row(y); for (int k = 0; k < wpr; k++) o[k] = ~p[k]; o[wpr - 1] &= tm; }
        return out;
    }
    BitImg opened(const EllipseElem& e) const { return eroded(e).dilated(e); }
    BitImg closed(const EllipseElem& e) const { return dilated(e).eroded(e); }
};

}  // namespace sind
