// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// Restatement of the OpenCV 4.2.0 shape / region primitives on the DynaDetect depth side
// (SURVEY.md §8c call-site list): getStructuringElement, morphologyEx, findContours (Suzuki-Abe
// border following, CHAIN_APPROX_NONE), drawContours (FILLED / thickness 2), contourArea, arcLength,
// floodFill (mask-only), medianBlur(5, 32F), kmeans(KMEANS_USE_INITIAL_LABELS), calcHist/compareHist.
#pragma once
#include <deque>
#include "cvx_core.hpp"

namespace cvx {

// ---------------------------------------------------------------- structuring elements / morphology
struct StructElem { int n; std::vector<int> j1, j2; int ax, ay; };   // per row: columns [j1, j2)
// getStructuringElement(MORPH_ELLIPSE, Size(n,n)); anchor = (n/2, n/2)  (imgproc/morph.dispatch.cpp)
inline StructElem ellipse_elem(int n) {
    StructElem e; e.n = n; e.j1.assign(n, 0); e.j2.assign(n, 0); e.ax = n / 2; e.ay = n / 2;
    if (n == 1) { e.j2[0] = 1; return e; }
    int r = n / 2, c = n / 2; double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < n; i++) {
        int dy = i - r;
        if (std::abs(dy) <= r) {
            int dx = cvRound(c * std::sqrt((r * r - dy * dy) * inv_r2));
            e.j1[i] = std::max(c - dx, 0); e.j2[i] = std::min(c + dx + 1, n);
        }
    }
    return e;
}

// dilate: dst(x,y) = max over element of src(x + j - ax, y + i - ay); erode: min.  Pixels outside the
// image are ignored (morphologyDefaultBorderValue).  Every image morphed on this path is two-valued
// {0, V}; that is asserted and exploited (row prefix sums), the result is identical to the max/min filter.
inline void morph_binary(const Img8& src, Img8& dst, const StructElem& e, bool dilate) {
    const int w = src.w, h = src.h;
    uint8_t V = 0;
    for (uint8_t v : src.d) if (v) { if (V && v != V) { std::fprintf(stderr, "morph_binary: image is not two-valued\n"); std::abort(); } V = v; }
    Img8 out(w, h, 1, 0);
    if (!V) { dst = out; return; }
    // prefix[y][x] = number of "hit" pixels in row y, columns < x.  hit = nonzero (dilate) / zero (erode)
    std::vector<int> pre((size_t)(w + 1) * h);
    for (int y = 0; y < h; y++) {
        int* p = &pre[(size_t)y * (w + 1)]; const uint8_t* s = src.row(y); p[0] = 0;
        for (int x = 0; x < w; x++) p[x + 1] = p[x] + (dilate ? (s[x] != 0) : (s[x] == 0));
    }
    for (int y = 0; y < h; y++) {
        uint8_t* o = out.row(y);
        for (int x = 0; x < w; x++) {
            bool hit = false;
            for (int i = 0; i < e.n && !hit; i++) {
                if (e.j2[i] <= e.j1[i]) continue;
                int yy = y + i - e.ay; if (yy < 0 || yy >= h) continue;
                int xa = std::max(x + e.j1[i] - e.ax, 0), xb = std::min(x + e.j2[i] - e.ax, w);
                if (xa >= xb) continue;
                const int* p = &pre[(size_t)yy * (w + 1)];
                hit = p[xb] - p[xa] > 0;
            }
            o[x] = dilate ? (hit ? V : 0) : (hit ? 0 : V);   // erode: any zero under the element -> 0
        }
    }
    dst = std::move(out);
}
inline void dilate(const Img8& s, Img8& d, const StructElem& e) { morph_binary(s, d, e, true); }
inline void erode(const Img8& s, Img8& d, const StructElem& e) { morph_binary(s, d, e, false); }
inline void morph_open(const Img8& s, Img8& d, const StructElem& e) { Img8 t; erode(s, t, e); dilate(t, d, e); }
inline void morph_close(const Img8& s, Img8& d, const StructElem& e) { Img8 t; dilate(s, t, e); erode(t, d, e); }

inline int count_nonzero(const Img8& a) { int n = 0; for (uint8_t v : a.d) n += v != 0; return n; }

// ---------------------------------------------------------------- findContours (imgproc/contours.cpp)
typedef std::vector<Pt> Contour;
// Suzuki-Abe border following on a 0-padded copy; external_only = RETR_EXTERNAL, otherwise every
// outer and hole border is returned (RETR_CCOMP / RETR_LIST membership; hierarchy is not needed here).
// is_hole (optional) receives one flag per contour.
inline void find_contours(const Img8& src, std::vector<Contour>& contours, bool external_only, std::vector<char>* is_hole_out = nullptr) {
    contours.clear(); if (is_hole_out) is_hole_out->clear();
    const int w = src.w + 2, h = src.h + 2;
    std::vector<signed char> img((size_t)w * h, 0);
    for (int y = 0; y < src.h; y++) for (int x = 0; x < src.w; x++) img[(size_t)(y + 1) * w + x + 1] = src.at(y, x) ? 1 : 0;
    int deltas[16];
    const int d8[8] = {1, -w + 1, -w, -w - 1, -1, w - 1, w, w + 1};
    for (int i = 0; i < 16; i++) deltas[i] = d8[i & 7];
    static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int nbd = 2;
    for (int y = 1; y < h - 1; y++) {
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x < w; x++) {
            int p = img[(size_t)y * w + x];
            if (p == prev) { continue; }
            bool is_hole = false, start = true;
            if (!(prev == 0 && p == 1)) {                                // not an outer border start
                if (p != 0 || prev < 1) start = false;                   // not a hole border start either
                else { if (prev & -2) lnbd_x = x - 1; is_hole = true; }
            }
            if (start && external_only && (is_hole || img[(size_t)y * w + lnbd_x] > 0)) start = false;
            if (start) {
                lnbd_x = x - (is_hole ? 1 : 0);
                Pt origin = {x - (is_hole ? 1 : 0), y};
                signed char* i0 = &img[(size_t)origin.y * w + origin.x];
                signed char *i1, *i3, *i4 = nullptr;
                Contour c;
                Pt pt = origin;
                const int mark = external_only ? 2 : nbd;
                int s, s_end; s_end = s = is_hole ? 0 : 4;
                do { s = (s - 1) & 7; i1 = i0 + deltas[s]; } while (*i1 == 0 && s != s_end);
                if (s == s_end) {            // single pixel
                    *i0 = (signed char)(mark | -128);
                    c.push_back({pt.x - 1, pt.y - 1});
                } else {
                    i3 = i0;
                    for (;;) {
                        s_end = s;
                        s = std::min(s, 15);
                        while (s < 15) { i4 = i3 + deltas[++s]; if (*i4 != 0) break; }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(mark | -128);
                        else if (*i3 == 1) *i3 = (signed char)mark;
                        c.push_back({pt.x - 1, pt.y - 1});
                        pt.x += cdx[s]; pt.y += cdy[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                contours.push_back(std::move(c));
                if (is_hole_out) is_hole_out->push_back(is_hole);
                if (!external_only) { nbd++; if (nbd > 127) nbd = 2; }
                p = img[(size_t)y * w + x];   // the start pixel may have been re-marked
                if (is_hole) p = 0;
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
}

// cv::contourArea (unoriented): |shoelace| / 2
inline double contour_area(const Contour& c) {
    if (c.empty()) return 0;
    double a00 = 0; Pt prev = c.back();
    for (const Pt& p : c) { a00 += (double)prev.x * p.y - (double)prev.y * p.x; prev = p; }
    return std::fabs(a00 * 0.5);
}
// cv::arcLength(closed = true): float segment lengths accumulated in double
inline double arc_length_closed(const Contour& c) {
    if (c.size() <= 1) return 0;
    double per = 0; Pt prev = c.back();
    for (const Pt& p : c) { float dx = (float)p.x - (float)prev.x, dy = (float)p.y - (float)prev.y; per += std::sqrt(dx * dx + dy * dy); prev = p; }
    return per;
}

// drawContours(..., color, thickness = 2): every segment of the closed chain is a unit 8-neighbour step,
// so ThickLine(thickness 2) = a half-width-1 quad plus radius-1 discs at both ends; the union over the
// chain equals the chain pixels dilated by the 3x3 cross (derivation in DESIGN.md "drawContours").
inline void draw_contour_thick2(Img8& img, const Contour& c, uint8_t color) {
    for (const Pt& p : c) {
        static const int ox[5] = {0, -1, 1, 0, 0}, oy[5] = {0, 0, 0, -1, 1};
        for (int k = 0; k < 5; k++) { int x = p.x + ox[k], y = p.y + oy[k]; if (x >= 0 && y >= 0 && x < img.w && y < img.h) img.at(y, x) = color; }
    }
}
// drawContours(..., FILLED) for ONE contour: boundary polyline plus even-odd scanline interior
// (drawing.cpp CollectPolyEdges + FillEdgeCollection).  Several contours drawn in ONE call (contourIdx = -1)
// share one edge table: use draw_contours_filled_joint for that.
inline void draw_contours_filled_joint(Img8& img, const std::vector<Contour>& cs, uint8_t color) {
    std::vector<std::vector<int>> cross(img.h);
    for (const Contour& c : cs) {
        const size_t n = c.size();
        for (size_t i = 0; i < n; i++) {
            const Pt& p = c[i]; const Pt& q = c[(i + 1) % n];
            if (p.x >= 0 && p.y >= 0 && p.x < img.w && p.y < img.h) img.at(p.y, p.x) = color;
            if (p.y == q.y) continue;
            const Pt& top = p.y < q.y ? p : q;          // unit step: active on scanline top.y only
            if (top.y >= 0 && top.y < img.h) cross[top.y].push_back(top.x);
        }
    }
    for (int y = 0; y < img.h; y++) {
        std::vector<int>& xs = cross[y];
        if (xs.size() < 2) continue;
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2)
            for (int x = std::max(xs[k], 0); x <= std::min(xs[k + 1], img.w - 1); x++) img.at(y, x) = color;
    }
}
inline void draw_contour_filled(Img8& img, const Contour& c, uint8_t color) {
    std::vector<Contour> one(1, c); draw_contours_filled_joint(img, one, color);
}

// ---------------------------------------------------------------- floodFill, FLOODFILL_MASK_ONLY, floating range
// image is two-valued on this path and lo/up diff (5) is smaller than the value gap, so the filled set is
// the 8-connected component of equal-valued pixels with mask == 0.  mask is (h+2) x (w+2).
inline int flood_fill_mask_only(const Img8& image, Img8& mask, Pt seed, uint8_t new_mask_val, int diff) {
    if (seed.x < 0 || seed.y < 0 || seed.x >= image.w || seed.y >= image.h) return 0;
    if (mask.at(seed.y + 1, seed.x + 1) != 0) return 0;
    std::deque<Pt> q; q.push_back(seed); mask.at(seed.y + 1, seed.x + 1) = new_mask_val;
    int area = 0;
    while (!q.empty()) {
        Pt p = q.front(); q.pop_front(); area++;
        int v = image.at(p.y, p.x);
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            int x = p.x + dx, y = p.y + dy;
            if (x < 0 || y < 0 || x >= image.w || y >= image.h) continue;
            if (mask.at(y + 1, x + 1) != 0) continue;
            if (std::abs((int)image.at(y, x) - v) > diff) continue;
            mask.at(y + 1, x + 1) = new_mask_val; q.push_back({x, y});
        }
    }
    return area;
}

// ---------------------------------------------------------------- medianBlur(ksize 5) on CV_32F, BORDER_REPLICATE
inline void median5_f32(const ImgF& src, ImgF& dst) {
    dst.create(src.w, src.h);
    float v[25];
    for (int y = 0; y < src.h; y++) for (int x = 0; x < src.w; x++) {
        int n = 0;
        for (int dy = -2; dy <= 2; dy++) { const float* r = src.row(std::min(std::max(y + dy, 0), src.h - 1));
            for (int dx = -2; dx <= 2; dx++) v[n++] = r[std::min(std::max(x + dx, 0), src.w - 1)]; }
        std::nth_element(v, v + 12, v + 25);
        dst.at(y, x) = v[12];
    }
}

// ---------------------------------------------------------------- cv::kmeans (core/kmeans.cpp), KMEANS_USE_INITIAL_LABELS, 1 attempt
// data: N x dims float; labels in/out; centers K x dims out.  criteria EPS+COUNT.
inline void kmeans_initial_labels(const float* data, int N, int dims, int K, int* labels, int maxCount, double epsilon, std::vector<float>& centers) {
    epsilon = std::max(epsilon, 0.); epsilon *= epsilon;
    maxCount = std::min(std::max(maxCount, 2), 100);
    centers.assign((size_t)K * dims, 0.f);
    std::vector<float> old_centers((size_t)K * dims, 0.f), temp(dims);
    std::vector<int> counters(K);
    for (int iter = 0;;) {
        double max_center_shift = iter == 0 ? DBL_MAX : 0.0;
        std::swap(centers, old_centers);
        std::fill(centers.begin(), centers.end(), 0.f); std::fill(counters.begin(), counters.end(), 0);
        for (int i = 0; i < N; i++) {
            const float* sample = data + (size_t)i * dims; int k = labels[i]; float* center = &centers[(size_t)k * dims];
            for (int j = 0; j < dims; j++) center[j] += sample[j];
            counters[k]++;
        }
        for (int k = 0; k < K; k++) {
            if (counters[k] != 0) continue;
            int max_k = 0;
            for (int k1 = 1; k1 < K; k1++) if (counters[max_k] < counters[k1]) max_k = k1;
            double max_dist = 0; int farthest_i = -1;
            float* base_center = &centers[(size_t)max_k * dims];
            float scale = 1.f / counters[max_k];
            for (int j = 0; j < dims; j++) temp[j] = base_center[j] * scale;
            for (int i = 0; i < N; i++) {
                if (labels[i] != max_k) continue;
                const float* sample = data + (size_t)i * dims;
                double dist = 0; for (int j = 0; j < dims; j++) { float t = sample[j] - temp[j]; dist += t * t; }  // normL2Sqr_ (float acc)
                if (max_dist <= dist) { max_dist = dist; farthest_i = i; }
            }
            counters[max_k]--; counters[k]++; labels[farthest_i] = k;
            const float* sample = data + (size_t)farthest_i * dims; float* cur_center = &centers[(size_t)k * dims];
            for (int j = 0; j < dims; j++) { base_center[j] -= sample[j]; cur_center[j] += sample[j]; }
        }
        for (int k = 0; k < K; k++) {
            float* center = &centers[(size_t)k * dims];
            float scale = 1.f / counters[k];
            for (int j = 0; j < dims; j++) center[j] *= scale;
            if (iter > 0) {
                double dist = 0; const float* oc = &old_centers[(size_t)k * dims];
                for (int j = 0; j < dims; j++) { double t = center[j] - oc[j]; dist += t * t; }
                max_center_shift = std::max(max_center_shift, dist);
            }
        }
        bool isLastIter = (++iter == std::max(maxCount, 2) || max_center_shift <= epsilon);
        if (isLastIter) break;   // labels are NOT re-assigned on the last iteration
        // KMeansDistanceComputer<false>: nearest centre by squared L2 (float accumulation), first minimum wins
        for (int i = 0; i < N; i++) {
            const float* sample = data + (size_t)i * dims;
            int k_best = 0; float min_dist = FLT_MAX;
            for (int k = 0; k < K; k++) {
                const float* c = &centers[(size_t)k * dims];
                float dist = 0; for (int j = 0; j < dims; j++) { float t = sample[j] - c[j]; dist += t * t; }
                if (min_dist > dist) { min_dist = dist; k_best = k; }
            }
            labels[i] = k_best;
        }
    }
}

// ---------------------------------------------------------------- calcHist (masked, 256 bins) + compareHist
// ranges {0,255} with 256 bins (reference DynaDetect.cc:1691-1696): bin(v) = floor(v*256/255) = v for v<=254, 255 falls outside.
inline void masked_hist_0_255(const Img8& img, const Img8& mask, float hist[256]) {
    std::fill(hist, hist + 256, 0.f);
    for (size_t i = 0; i < img.d.size(); i++) if (mask.d[i] && img.d[i] < 255) hist[img.d[i]] += 1.f;
}
inline double compare_hist_correl(const float* h1, const float* h2, int n) {
    double s1 = 0, s2 = 0, s11 = 0, s12 = 0, s22 = 0;
    for (int j = 0; j < n; j++) { double a = h1[j], b = h2[j]; s12 += a * b; s1 += a; s11 += a * a; s2 += b; s22 += b * b; }
    double scale = 1. / n, num = s12 - s1 * s2 * scale, denom2 = (s11 - s1 * s1 * scale) * (s22 - s2 * s2 * scale);
    return std::fabs(denom2) > DBL_EPSILON ? num / std::sqrt(denom2) : 1.;
}
inline double compare_hist_bhattacharyya(const float* h1, const float* h2, int n) {
    double s1 = 0, s2 = 0, result = 0;
    for (int j = 0; j < n; j++) { double a = h1[j], b = h2[j]; result += std::sqrt(a * b); s1 += a; s2 += b; }
    s1 *= s2; s1 = std::fabs(s1) > FLT_EPSILON ? 1. / std::sqrt(s1) : 1.;
    return std::sqrt(std::max(1. - result * s1, 0.));
}
inline double compare_hist_intersect(const float* h1, const float* h2, int n) {
    double r = 0; for (int j = 0; j < n; j++) r += std::min(h1[j], h2[j]); return r;
}

}  // namespace cvx
