// Host stage: robust homography between PROSAC-ordered correspondences, replacing cv::findHomography(..., cv::RHO)
// at reference DynaDetect.cc:1235 (63 x 47 grid samples, serial and order-sensitive -> host).
//
// OpenCV's RHO (calib3d/rho.cpp) is not vendored by the reference and cannot be reproduced bit for bit offline, so
// this is a documented substitute with the same contract (DESIGN.md "homography"):
//   1. PROSAC sampling (Chum & Matas 2005 growth function, T_N = 2000) over quality-sorted pairs, cv::RNG recurrence,
//      fixed seed; 4-point minimal solver with h33 = 1 (Gaussian elimination, partial pivoting);
//   2. degenerate (near-collinear) samples rejected; support = transfer error <= 3 px; adaptive stop at 0.995;
//   3. refinement on the inliers: Hartley-normalised linear least squares, then <= 10 Gauss-Newton steps on the
//      transfer error; kept only if the support does not shrink;
//   4. H scaled so H(2,2) = 1; all zeros when fewer than 4 pairs or no model.
#include <algorithm>
#include <cmath>
#include <immintrin.h>
#include "host.hpp"

namespace sind {
namespace {

struct Rng { uint64_t st; uint32_t next() { st = (uint64_t)(uint32_t)st * 4164903690U + (st >> 32); return (uint32_t)st; } };

bool gauss_solve(int n, double* A, double* b, double* x) {
    for (int c = 0; c < n; c++) {
        int piv = c; double best = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) { const double v = std::fabs(A[r * n + c]); if (v > best) { best = v; piv = r; } }
        if (best < 1e-12) return false;
        if (piv != c) { for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]); std::swap(b[c], b[piv]); }
        for (int r = c + 1; r < n; r++) {
            const double f = A[r * n + c] / A[c * n + c];
            if (f == 0) continue;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) { double s = b[r]; for (int k = r + 1; k < n; k++) s -= A[r * n + k] * x[k]; x[r] = s / A[r * n + r]; }
    return true;
}
bool minimal_h(const Pt2f* s, const Pt2f* d, const int id[4], double H[9]) {
    double A[64], b[8], x[8];
    for (int i = 0; i < 4; i++) {
        const double X = s[id[i]].x, Y = s[id[i]].y, u = d[id[i]].x, v = d[id[i]].y;
        double* r0 = &A[(2 * i) * 8]; double* r1 = &A[(2 * i + 1) * 8];
        r0[0] = X; r0[1] = Y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * X; r0[7] = -u * Y; b[2 * i] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = X; r1[4] = Y; r1[5] = 1; r1[6] = -v * X; r1[7] = -v * Y; b[2 * i + 1] = v;
    }
    if (!gauss_solve(8, A, b, x)) return false;
    for (int i = 0; i < 8; i++) H[i] = x[i];
    H[8] = 1.0; return true;
}
double err2(const double H[9], const Pt2f& s, const Pt2f& d) {
    const double w = H[6] * s.x + H[7] * s.y + H[8];
    if (std::fabs(w) < 1e-12) return 1e30;
    const double px = (H[0] * s.x + H[1] * s.y + H[2]) / w, py = (H[3] * s.x + H[4] * s.y + H[5]) / w;
    const double ex = px - d.x, ey = py - d.y;
    return ex * ex + ey * ey;
}
// number of pairs with err2(H, s, d) <= t2: the same IEEE operations in the same order as err2, four pairs per AVX2 instruction
// (explicit mul / add / div intrinsics, nothing fused) -- the support count is the inner loop of the PROSAC search
int count_support(const double H[9], const Pt2f* s, const Pt2f* d, int N, double t2) {
    int cnt = 0, i = 0;
#if defined(__AVX2__)
    const __m256d h0 = _mm256_set1_pd(H[0]), h1 = _mm256_set1_pd(H[1]), h2 = _mm256_set1_pd(H[2]), h3 = _mm256_set1_pd(H[3]), h4 = _mm256_set1_pd(H[4]),
                  h5 = _mm256_set1_pd(H[5]), h6 = _mm256_set1_pd(H[6]), h7 = _mm256_set1_pd(H[7]), h8 = _mm256_set1_pd(H[8]);
    const __m256d tiny = _mm256_set1_pd(1e-12), lim = _mm256_set1_pd(t2), absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffll));
    for (; i + 4 <= N; i += 4) {
        const __m256 sv = _mm256_loadu_ps(&s[i].x), dv = _mm256_loadu_ps(&d[i].x);                 // x0 y0 x1 y1 | x2 y2 x3 y3
        const __m256 sx4 = _mm256_permutevar8x32_ps(sv, _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7)), dx4 = _mm256_permutevar8x32_ps(dv, _mm256_setr_epi32(0, 2, 4, 6, 1, 3, 5, 7));
        const __m256d sx = _mm256_cvtps_pd(_mm256_castps256_ps128(sx4)), sy = _mm256_cvtps_pd(_mm256_extractf128_ps(sx4, 1));
        const __m256d dx = _mm256_cvtps_pd(_mm256_castps256_ps128(dx4)), dy = _mm256_cvtps_pd(_mm256_extractf128_ps(dx4, 1));
        const __m256d w = _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(h6, sx), _mm256_mul_pd(h7, sy)), h8);
        const __m256d px = _mm256_div_pd(_mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(h0, sx), _mm256_mul_pd(h1, sy)), h2), w);
        const __m256d py = _mm256_div_pd(_mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(h3, sx), _mm256_mul_pd(h4, sy)), h5), w);
        const __m256d ex = _mm256_sub_pd(px, dx), ey = _mm256_sub_pd(py, dy);
        const __m256d e = _mm256_add_pd(_mm256_mul_pd(ex, ex), _mm256_mul_pd(ey, ey));
        const __m256d ok = _mm256_andnot_pd(_mm256_cmp_pd(_mm256_and_pd(w, absmask), tiny, _CMP_LT_OQ), _mm256_cmp_pd(e, lim, _CMP_LE_OQ));
        cnt += __builtin_popcount((unsigned)_mm256_movemask_pd(ok));
    }
#endif
    for (; i < N; i++) cnt += err2(H, s[i], d[i]) <= t2;
    return cnt;
}
bool collinear3(const Pt2f* p, const int id[4]) {
    for (int a = 0; a < 4; a++) for (int b = a + 1; b < 4; b++) for (int c = b + 1; c < 4; c++) {
        const double x1 = p[id[b]].x - p[id[a]].x, y1 = p[id[b]].y - p[id[a]].y;
        const double x2 = p[id[c]].x - p[id[a]].x, y2 = p[id[c]].y - p[id[a]].y;
        if (std::fabs(x1 * y2 - x2 * y1) < 1e-3 * (std::fabs(x1 * x2 + y1 * y2) + 1.0)) return true;
    }
    return false;
}
bool refine(const Pt2f* s, const Pt2f* d, const std::vector<int>& in, double H[9]) {
    const int n = (int)in.size();
    if (n < 4) return false;
    double cs[2] = {0, 0}, cd[2] = {0, 0};
    for (int i : in) { cs[0] += s[i].x; cs[1] += s[i].y; cd[0] += d[i].x; cd[1] += d[i].y; }
    cs[0] /= n; cs[1] /= n; cd[0] /= n; cd[1] /= n;
    double ms = 0, md = 0;
    for (int i : in) {
        ms += std::sqrt((s[i].x - cs[0]) * (s[i].x - cs[0]) + (s[i].y - cs[1]) * (s[i].y - cs[1]));
        md += std::sqrt((d[i].x - cd[0]) * (d[i].x - cd[0]) + (d[i].y - cd[1]) * (d[i].y - cd[1]));
    }
    if (ms < 1e-9 || md < 1e-9) return false;
    const double ss = std::sqrt(2.0) * n / ms, sd = std::sqrt(2.0) * n / md;
    double AtA[64], Atb[8], x[8];
    std::fill(AtA, AtA + 64, 0.0); std::fill(Atb, Atb + 8, 0.0);
    for (int i : in) {
        const double X = (s[i].x - cs[0]) * ss, Y = (s[i].y - cs[1]) * ss, u = (d[i].x - cd[0]) * sd, v = (d[i].y - cd[1]) * sd;
        const double r0[8] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y}, r1[8] = {0, 0, 0, X, Y, 1, -v * X, -v * Y};
        for (int a = 0; a < 8; a++) { for (int b = 0; b < 8; b++) AtA[a * 8 + b] += r0[a] * r0[b] + r1[a] * r1[b]; Atb[a] += r0[a] * u + r1[a] * v; }
    }
    if (!gauss_solve(8, AtA, Atb, x)) return false;
    const double Hn[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 1.0};
    const double Ts[9] = {ss, 0, -ss * cs[0], 0, ss, -ss * cs[1], 0, 0, 1};
    const double Tdi[9] = {1 / sd, 0, cd[0], 0, 1 / sd, cd[1], 0, 0, 1};
    double M[9], R[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Hn[r * 3 + k] * Ts[k * 3 + c]; M[r * 3 + c] = a; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Tdi[r * 3 + k] * M[k * 3 + c]; R[r * 3 + c] = a; }
    if (std::fabs(R[8]) < 1e-12) return false;
    for (int i = 0; i < 9; i++) H[i] = R[i] / R[8];
    for (int it = 0; it < 10; it++) {
        double JtJ[64], Jtr[8], dx[8];
        std::fill(JtJ, JtJ + 64, 0.0); std::fill(Jtr, Jtr + 8, 0.0);
        for (int i : in) {
            const double X = s[i].x, Y = s[i].y;
            const double w = H[6] * X + H[7] * Y + 1.0; if (std::fabs(w) < 1e-12) continue;
            const double iw = 1.0 / w, px = (H[0] * X + H[1] * Y + H[2]) * iw, py = (H[3] * X + H[4] * Y + H[5]) * iw;
            const double rx = d[i].x - px, ry = d[i].y - py;
            const double jx[8] = {X * iw, Y * iw, iw, 0, 0, 0, -X * px * iw, -Y * px * iw};
            const double jy[8] = {0, 0, 0, X * iw, Y * iw, iw, -X * py * iw, -Y * py * iw};
            for (int a = 0; a < 8; a++) { for (int b = 0; b < 8; b++) JtJ[a * 8 + b] += jx[a] * jx[b] + jy[a] * jy[b]; Jtr[a] += jx[a] * rx + jy[a] * ry; }
        }
        if (!gauss_solve(8, JtJ, Jtr, dx)) break;
        double step = 0;
        for (int a = 0; a < 8; a++) { H[a] += dx[a]; step += dx[a] * dx[a]; }
        if (step < 1e-20) break;
    }
    return true;
}
}  // namespace

bool find_homography_prosac(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst, double H[9]) {
    const double thresh = 3.0, confidence = 0.995; const int maxIters = 2000;
    const int N = (int)src.size();
    std::fill(H, H + 9, 0.0);
    if (N < 4) return false;
    const double t2 = thresh * thresh;
    Rng rng{0x9E3779B97F4A7C15ull};
    double Tn = maxIters;
    for (int i = 0; i < 4; i++) Tn *= (double)(4 - i) / (double)(N - i);
    int n = 4, Tn_prime = 1, best_cnt = 0, iters_needed = maxIters;
    double bestH[9];
    for (int t = 1; t <= iters_needed && t <= maxIters; t++) {
        if (t == Tn_prime && n < N) { const double Tn1 = Tn * (double)(n + 1) / (double)(n + 1 - 4); Tn_prime += (int)std::ceil(Tn1 - Tn); Tn = Tn1; n++; }
        int id[4];
        if (Tn_prime < t) { for (int k = 0; k < 4; k++) { bool dup; do { id[k] = (int)(rng.next() % (uint32_t)n); dup = false; for (int q = 0; q < k; q++) dup |= id[q] == id[k]; } while (dup); } }
        else { id[3] = n - 1; for (int k = 0; k < 3; k++) { bool dup; do { id[k] = (int)(rng.next() % (uint32_t)(n - 1)); dup = false; for (int q = 0; q < k; q++) dup |= id[q] == id[k]; } while (dup); } }
        if (collinear3(src.data(), id) || collinear3(dst.data(), id)) continue;
        double Hc[9];
        if (!minimal_h(src.data(), dst.data(), id, Hc)) continue;
        const int cnt = count_support(Hc, src.data(), dst.data(), N, t2);
        if (cnt > best_cnt) {
            best_cnt = cnt; std::copy(Hc, Hc + 9, bestH);
            const double eps = (double)cnt / N, p4 = eps * eps * eps * eps;
            if (p4 > 1.0 - 1e-12) iters_needed = t;
            else { const double k = std::log(1.0 - confidence) / std::log(1.0 - p4); iters_needed = (int)std::min<double>(maxIters, std::ceil(k)); }
        }
    }
    if (best_cnt < 4) return false;
    std::vector<int> inl;
    for (int i = 0; i < N; i++) if (err2(bestH, src[i], dst[i]) <= t2) inl.push_back(i);
    double Hr[9]; std::copy(bestH, bestH + 9, Hr);
    if (refine(src.data(), dst.data(), inl, Hr) && count_support(Hr, src.data(), dst.data(), N, t2) >= best_cnt) std::copy(Hr, Hr + 9, bestH);
    std::copy(bestH, bestH + 9, H);
    return true;
}

}  // namespace sind
