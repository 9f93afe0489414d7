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
#include <emmintrin.h>      // SSE2 (x86-64 baseline): byte image <-> bit image conversions, 16 pixels per instruction

namespace sind {

// 64 consecutive bytes -> one word: bit i = (s[i] != 0) / (s[i] == v)
static inline uint64_t pack64_nonzero(const uint8_t* s) {
    const __m128i z = _mm_setzero_si128(); uint64_t m = 0;
    for (int q = 0; q < 4; q++) m |= (uint64_t)(uint16_t)~_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(s + 16 * q)), z)) << (16 * q);
    return m;
}
static inline uint64_t pack64_equal(const uint8_t* s, uint8_t v) {
    const __m128i c = _mm_set1_epi8((char)v); uint64_t m = 0;
    for (int q = 0; q < 4; q++) m |= (uint64_t)(uint16_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(s + 16 * q)), c)) << (16 * q);
    return m;
}
// 8 bits -> 8 bytes of 0x00 / 0xFF
static inline uint64_t spread8(unsigned b) {
    uint64_t x = (uint64_t)b * 0x0101010101010101ull & 0x8040201008040201ull;      // byte k holds bit k (at position k)
    x = ((x + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;                // any bit in the byte -> 0x80
    return (x >> 7) * 0xffull;
}

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
        const int full = w >> 6;
        for (int y = 0; y < h; y++) { const uint8_t* s = p + (size_t)y * stride; uint64_t* r = b.row(y);
            for (int k = 0; k < full; k++) r[k] = pack64_nonzero(s + 64 * k);
            for (int x = full << 6; x < w; x++) if (s[x]) r[x >> 6] |= 1ull << (x & 63); }
        return b;
    }
    static BitImg from_equal(const uint8_t* p, int w, int h, int stride, uint8_t v) {
        BitImg b(w, h);
        const int full = w >> 6;
        for (int y = 0; y < h; y++) { const uint8_t* s = p + (size_t)y * stride; uint64_t* r = b.row(y);
            for (int k = 0; k < full; k++) r[k] = pack64_equal(s + 64 * k, v);
            for (int x = full << 6; x < w; x++) if (s[x] == v) r[x >> 6] |= 1ull << (x & 63); }
        return b;
    }
    // one bit image per byte value v in [0, n): out[v] = (src == v); n passes of 16-pixel compares instead of a per-pixel scatter
    static void split_labels(const uint8_t* p, int w, int h, int stride, int first, int n, BitImg* out) {
        for (int v = 0; v < n; v++) out[v].create(w, h);
        const int full = w >> 6;
        for (int y = 0; y < h; y++) { const uint8_t* s = p + (size_t)y * stride;
            for (int k = 0; k < full; k++) for (int v = 0; v < n; v++) out[v].row(y)[k] = pack64_equal(s + 64 * k, (uint8_t)(first + v));
            for (int x = full << 6; x < w; x++) { const int v = (int)s[x] - first; if (v >= 0 && v < n) out[v].row(y)[x >> 6] |= 1ull << (x & 63); } }
    }
    void to_u8(uint8_t* out, int stride, uint8_t on, uint8_t off = 0) const {
        const int full = w >> 6; const uint64_t on8 = on * 0x0101010101010101ull, off8 = off * 0x0101010101010101ull;
        for (int y = 0; y < h; y++) { const uint64_t* r = row(y); uint8_t* o = out + (size_t)y * stride;
            for (int k = 0; k < full; k++) { const uint64_t m = r[k]; for (int q = 0; q < 8; q++) { const uint64_t f = spread8((unsigned)(m >> (8 * q)) & 0xffu), v = (f & on8) | (~f & off8); std::memcpy(o + 64 * k + 8 * q, &v, 8); } }
            for (int x = full << 6; x < w; x++) o[x] = ((r[x >> 6] >> (x & 63)) & 1) ? on : off; }
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
    // word range [k0, k1] that holds the set pixels of rows [y0, y1] (k1 < k0: none)
    void word_extent(int y0, int y1, int& k0, int& k1) const {
        k0 = wpr; k1 = -1;
        for (int y = y0; y <= y1; y++) { const uint64_t* s = row(y); for (int q = 0; q < wpr; q++) if (s[q]) { k0 = std::min(k0, q); k1 = std::max(k1, q); } }
    }
    // Dilation of the source rows [ry0, ry1].  Only the word window [kw0, kw1] is read and written: the caller guarantees that the
    // result is needed there only and that nothing outside the window can reach it except through the window's first / last word
    // (an element is far narrower than a word), or that the window edge is the image edge.
    BitImg dilated_win(const EllipseElem& e, int ry0, int ry1, int kw0, int kw1) const {
        BitImg out(w, h);
        kw0 = std::max(kw0, 0); kw1 = std::min(kw1, wpr - 1);
        if (ry1 < ry0 || kw1 < kw0) return out;
        const int nw = kw1 - kw0 + 1; const bool has_tail = kw1 == wpr - 1;
        // distinct horizontal runs of the element
        std::vector<std::pair<int, int>> runs; std::vector<int> run_of(e.n, -1);
        for (int i = 0; i < e.n; i++) { if (e.j2[i] <= e.j1[i]) continue; std::pair<int, int> r(e.j1[i], e.j2[i]);
            size_t k = 0; for (; k < runs.size(); k++) if (runs[k] == r) break; if (k == runs.size()) runs.push_back(r); run_of[i] = (int)k; }
        const int nr = ry1 - ry0 + 1; const uint64_t tm = tail_mask();
        std::vector<uint64_t> Hbuf(runs.size() * (size_t)nr * nw, 0), acc(nw), tmp(nw);
        // every run of an elliptical element contains the anchor column: run k reaches ext[0][k] pixels to the right and ext[1][k] to
        // the left of it.  Per source row the one-sided windows OR_{t=0..m} src[x +- t] are grown ONCE, from the smallest extent to the
        // largest (a window of length m + 1 extends by up to m + 1 pixels per shift-and-or), and every run picks its two windows up
        // on the way, so partial windows at the borders survive and no extent is built twice.
        std::vector<int> ext[2], ord[2];
        for (int dir = 0; dir < 2; dir++) {
            for (size_t k = 0; k < runs.size(); k++) ext[dir].push_back(dir == 0 ? runs[k].second - 1 - e.ax : e.ax - runs[k].first);
            ord[dir].resize(runs.size()); for (size_t k = 0; k < runs.size(); k++) ord[dir][k] = (int)k;
            std::sort(ord[dir].begin(), ord[dir].end(), [&](int a, int b) { return ext[dir][a] < ext[dir][b]; });
        }
        for (int y = ry0; y <= ry1; y++) {
            const uint64_t* s = row(y) + kw0;
            bool nz = false; for (int q = 0; q < nw; q++) nz |= s[q] != 0;
            if (!nz) continue;
            for (int dir = 0; dir < 2; dir++) {
                const int sgn = dir == 0 ? 1 : -1;
                std::copy(s, s + nw, acc.begin());
                int m = 0;                                           // acc covers t = 0..m
                for (int k : ord[dir]) {
                    const int target = std::max(ext[dir][k], 0);
                    while (m < target) { const int step = std::min(m + 1, target - m); std::copy(acc.begin(), acc.end(), tmp.begin()); or_shift(acc.data(), tmp.data(), nw, sgn * step); m += step; }
                    uint64_t* hrow = &Hbuf[((size_t)k * nr + (size_t)(y - ry0)) * nw];
                    for (int q = 0; q < nw; q++) hrow[q] |= acc[q];
                }
            }
            if (has_tail) for (size_t k = 0; k < runs.size(); k++) Hbuf[(k * nr + (size_t)(y - ry0)) * nw + nw - 1] &= tm;
        }
        for (int i = 0; i < e.n; i++) {
            if (run_of[i] < 0) continue;
            const int dy = i - e.ay;                               // dst(y) takes src(y + dy)  ->  src row ys feeds dst row ys - dy
            for (int ys = ry0; ys <= ry1; ys++) { const int yd = ys - dy; if (yd < 0 || yd >= h) continue;
                const uint64_t* hrow = &Hbuf[((size_t)run_of[i] * nr + (size_t)(ys - ry0)) * nw]; uint64_t* o = out.row(yd) + kw0;
                for (int q = 0; q < nw; q++) o[q] |= hrow[q]; }
        }
        return out;
    }
    BitImg dilated(const EllipseElem& e, int ry0 = 0, int ry1 = -1) const {
        if (ry1 < 0) ry1 = h - 1;
        ry0 = std::max(ry0, 0); ry1 = std::min(ry1, h - 1);
        if (ry1 < ry0) return BitImg(w, h);
        int k0, k1; word_extent(ry0, ry1, k0, k1);
        if (k1 < k0) return BitImg(w, h);
        return dilated_win(e, ry0, ry1, k0 - 1, k1 + 1);          // a dilation grows by less than a word: one spare word on each side
    }
    // dilated(e).get(x, y) without the dilation: is any pixel of the element's window around (x, y) set?  (dst(x, y) = OR over the element's rows i and columns j1[i] .. j2[i] - 1
    // of src(x + j - ax, y + i - ay), positions outside the image ignored)
    bool dilation_hits(const EllipseElem& e, int x, int y) const {
        for (int i = 0; i < e.n; i++) {
            const int yy = y + i - e.ay;
            if (yy < 0 || yy >= h || e.j2[i] <= e.j1[i]) continue;
            const int xa = std::max(x + e.j1[i] - e.ax, 0), xb = std::min(x + e.j2[i] - e.ax, w) - 1;      // [xa, xb]
            if (xb < xa) continue;
            const uint64_t* r = row(yy); const int ka = xa >> 6, kb = xb >> 6;
            for (int k = ka; k <= kb; k++) {
                uint64_t m = ~0ull; if (k == ka) m &= ~0ull << (xa & 63); if (k == kb) m &= (xb & 63) == 63 ? ~0ull : ((1ull << ((xb & 63) + 1)) - 1);
                if (r[k] & m) return true;
            }
        }
        return false;
    }
    // erosion: positions outside the image are ignored, i.e. erode(X) = ~dilate(~X) inside the image
    BitImg eroded(const EllipseElem& e) const { return inverted().dilated(e).inverted(); }
    // erosion of an image whose set pixels all lie in rows [y0, y1]: the result is a subset of the source, so only those rows
    // are evaluated (complement rows further than the element height away cannot influence them)
    BitImg eroded_rows(const EllipseElem& e, int y0, int y1) const {
        y0 = std::max(y0, 0); y1 = std::min(y1, h - 1);
        BitImg out(w, h);
        if (y1 < y0) return out;
        int k0, k1; word_extent(y0, y1, k0, k1);
        if (k1 < k0) return out;
        // the erosion is a subset of the source: only the words [k0, k1] of rows [y0, y1] are needed from the dilated complement
        const BitImg dil = inverted().dilated_win(e, std::max(y0 - e.n, 0), std::min(y1 + e.n, h - 1), k0 - 1, k1 + 1);
        const uint64_t tm = tail_mask();
        for (int y = y0; y <= y1; y++) { const uint64_t* p = dil.row(y); uint64_t* o = out.row(y); for (int k = k0; k <= k1; k++) o[k] = ~p[k]; o[wpr - 1] &= tm; }
        return out;
    }
    BitImg opened(const EllipseElem& e) const { return eroded(e).dilated(e); }
    // opening of an image whose set pixels all lie in rows [y0, y1] (the result is a subset of the source)
    BitImg opened_rows(const EllipseElem& e, int y0, int y1) const { return eroded_rows(e, y0, y1).dilated(e, y0, y1); }
    // row-range variants of the set operations (everything outside [y0, y1] is left untouched)
    void and_rows(const BitImg& o, int y0, int y1) { y0 = std::max(y0, 0); y1 = std::min(y1, h - 1); for (size_t i = (size_t)y0 * wpr; i < (size_t)(y1 + 1) * wpr; i++) d[i] &= o.d[i]; }
    void andnot_rows(const BitImg& o, int y0, int y1) { y0 = std::max(y0, 0); y1 = std::min(y1, h - 1); for (size_t i = (size_t)y0 * wpr; i < (size_t)(y1 + 1) * wpr; i++) d[i] &= ~o.d[i]; }
    int count_rows(int y0, int y1) const { y0 = std::max(y0, 0); y1 = std::min(y1, h - 1); int n = 0; for (size_t i = (size_t)y0 * wpr; i < (size_t)(y1 + 1) * wpr; i++) n += __builtin_popcountll(d[i]); return n; }
    BitImg closed(const EllipseElem& e) const { return dilated(e).eroded(e); }
};

}  // namespace sind
