// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// Stand-ins for cv::findHomography(src, dst, noArray(), cv::RHO) (reference DynaDetect.cc:1235).  OpenCV 4.2's RHO estimator lives in
// calib3d/src/rho.cpp (~2.6 kLoC), is not vendored under /root/reference and cannot be restated bit for bit offline.  Two estimators are kept here:
//   find_homography_rho_scheme   (further down) RHO's PUBLISHED scheme -- PROSAC + SPRT + LM in float32; the oracle's default.
//                                The product's sindslam_amd/csrc/host/homography.cpp is a transliteration of it into the product's own types
//                                (same PRNG and warm-up, same SPRT constants, solver, LM loop and thresholds): the two agree bit for bit BY
//                                CONSTRUCTION, which checks the two implementations against each other and says nothing about OpenCV's output.
//   find_homography_prosac       round 1's lighter substitute (PROSAC + normalised least squares + Gauss-Newton); the only estimator here that is
//                                independent of the product, kept for tests/test_a8_sensitivity_cpu.py.
// What is known of rho.cpp itself, for whoever can read the 4.2.0 sources again.  [R] = recalled with confidence, [G] = guessed / uncertain.
//   call site   fundam.cpp findHomography(method == RHO) -> createAndRunRHORegistrator(confidence 0.995, maxIters 2000, threshold 3.0, ...)      [R]
//               -> rhoHest(src, dst, mask, N, maxD = 3.0f, maxI = 2000, rConvg = 2000, cfd = 0.995, minInl = 4, beta = 0.35,
//                          RHO_FLAG_ENABLE_NR | RHO_FLAG_ENABLE_FINAL_REFINEMENT, guess = NULL, H)                                            [R: the flags; G: beta 0.35]
//               findHomography's own LM refinement (createLMSolver, 10 iterations) is skipped for RHO; H is divided by H(2,2) afterwards.   [R]
//   PRNG        xorshift128+ ("fastRandom"), seeded by fastSeed(~0ULL) in rhoInit; a new estimator object is created per findHomography call,
//               so every call starts from the same state.                                                                                    [R: generator and seed; G: the seeding recipe]
//               (here: an xorshift128+ whose two words are derived from the seed by two constants and warmed up 20 draws -- NOT OpenCV's recipe)
//   sampling    PROSAC (Chum & Matas 2005) over the caller's order (best first) while NR is enabled: phNum starts at SMPL_SIZE = 4, grows when the
//               iteration count passes phEndI, phEndFpI follows the T'_n recurrence with T_N = rConvg; the 4th point is the phNum-th
//               correspondence, the other three are drawn below it.                                                                           [R: structure; G: rounding of phEndFpI]
//   degeneracy  isSampleDegenerate: orientation of the point triples must be preserved between source and destination (cross products).     [R]
//   model       hFuncRefC: 8x9 Gaussian elimination in float32, row pivoting only where a pivot vanishes.                                    [R: float32 elimination; G: pivoting rule]
//               (here: partial pivoting on the largest magnitude)
//   evaluation  SPRT (Matas & Chum 2005): SPRT_T_M = 25, SPRT_M_S = 1, SPRT_EPSILON = 0.1, SPRT_DELTA = 0.01; lambda multiplied point by point in
//               INDEX order over all N correspondences (no shuffle), rejection at lambda > A; epsilon / delta redesigned on a new best model
//               and on a rejected model whose inlier ratio differs by more than MIN_DELTA_CHNG = 0.1 relative.                                [R: constants; G: index order, the redesign rule's exact form]
//               (here: points are tested in a fixed random order and delta is redesigned at 5 % relative change -- in index order a dozen bad
//                leading pairs of this caller's ranking rejected every model)
//   stopping    updateBounds: the PROSAC non-randomness test (N*: the smallest n whose inlier count among the n best is not explicable by
//               chance, chi-square constants CHI_STAT = 2.706, CHI_SQ = 1.645, beta) shrinks the evaluated set and the iteration bound
//               together with the usual log(1 - cfd) / log(1 - eps^4) bound.                                                                  [R: that it exists and its constants; not restated here]
//   refinement  sacLMRefine on the inliers: at most 10 iterations, lambda starts at 0.01, gain ratio thresholds LM_GAIN_LO = 0.25 /
//               LM_GAIN_HI = 0.75 scale lambda by 2 / 0.5 (not 10 / 0.1), 8x8 damped Cholesky in float32 (sacChol8x8Damped, sacTRISolve8x8).   [R: thresholds and Cholesky; G: the scale factors]
//               (here: lambda *= 0.1 on success, *= 10 on failure)
// The sensitivity of the dynamic mask to these free choices is what tests/test_a8_sensitivity_cpu.py measures (profiles/r03/a8_sensitivity.txt).
// Every estimator returns H scaled so that H(2,2) = 1 as 9 doubles row-major, all-zero if estimation fails.
#pragma once
#include "cvx_core.hpp"

namespace cvx {

struct Pt2f { float x, y; };

namespace hdetail {
// Gaussian elimination with partial pivoting, n<=8.  Returns false if singular.
inline bool solve_linear(int n, double* A /*n x n row-major*/, double* b, double* x) {
    for (int c = 0; c < n; c++) {
        int piv = c; double best = std::fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) { double v = std::fabs(A[r * n + c]); if (v > best) { best = v; piv = r; } }
        if (best < 1e-12) return false;
        if (piv != c) { for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]); std::swap(b[c], b[piv]); }
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] / A[c * n + c];
            if (f == 0) continue;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * x[k];
        x[r] = s / A[r * n + r];
    }
    return true;
}
// H (h33 = 1) from 4 correspondences: 8x8 linear system.
inline bool h_from_4(const Pt2f* s, const Pt2f* d, const int idx[4], double H[9]) {
    double A[64], b[8], x[8];
    for (int i = 0; i < 4; i++) {
        double X = s[idx[i]].x, Y = s[idx[i]].y, u = d[idx[i]].x, v = d[idx[i]].y;
        double* r0 = &A[(2 * i) * 8]; double* r1 = &A[(2 * i + 1) * 8];
        r0[0] = X; r0[1] = Y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * X; r0[7] = -u * Y; b[2 * i] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = X; r1[4] = Y; r1[5] = 1; r1[6] = -v * X; r1[7] = -v * Y; b[2 * i + 1] = v;
    }
    if (!solve_linear(8, A, b, x)) return false;
    for (int i = 0; i < 8; i++) H[i] = x[i];
    H[8] = 1.0;
    return true;
}
inline double reproj_err2(const double H[9], const Pt2f& s, const Pt2f& d) {
    double w = H[6] * s.x + H[7] * s.y + H[8];
    if (std::fabs(w) < 1e-12) return 1e30;
    double px = (H[0] * s.x + H[1] * s.y + H[2]) / w, py = (H[3] * s.x + H[4] * s.y + H[5]) / w;
    double ex = px - d.x, ey = py - d.y;
    return ex * ex + ey * ey;
}
// three of the four sample points (nearly) collinear in either image -> degenerate
inline bool degenerate4(const Pt2f* p, const int idx[4]) {
    for (int a = 0; a < 4; a++) for (int b = a + 1; b < 4; b++) for (int c = b + 1; c < 4; c++) {
        double x1 = p[idx[b]].x - p[idx[a]].x, y1 = p[idx[b]].y - p[idx[a]].y;
        double x2 = p[idx[c]].x - p[idx[a]].x, y2 = p[idx[c]].y - p[idx[a]].y;
        if (std::fabs(x1 * y2 - x2 * y1) < 1e-3 * (std::fabs(x1 * x2 + y1 * y2) + 1.0)) return true;
    }
    return false;
}
// least-squares H (h33=1) over a subset, Hartley-normalised normal equations, then Gauss-Newton.
inline bool refine_h(const Pt2f* s, const Pt2f* d, const std::vector<int>& in, double H[9]) {
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
        double X = (s[i].x - cs[0]) * ss, Y = (s[i].y - cs[1]) * ss, u = (d[i].x - cd[0]) * sd, v = (d[i].y - cd[1]) * sd;
        double r0[8] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y}, r1[8] = {0, 0, 0, X, Y, 1, -v * X, -v * Y};
        for (int a = 0; a < 8; a++) {
            for (int b = 0; b < 8; b++) AtA[a * 8 + b] += r0[a] * r0[b] + r1[a] * r1[b];
            Atb[a] += r0[a] * u + r1[a] * v;
        }
    }
    if (!solve_linear(8, AtA, Atb, x)) return false;
    // de-normalise: H = Td^-1 * Hn * Ts
    double Hn[9] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], 1.0};
    double Ts[9] = {ss, 0, -ss * cs[0], 0, ss, -ss * cs[1], 0, 0, 1};
    double Tdi[9] = {1 / sd, 0, cd[0], 0, 1 / sd, cd[1], 0, 0, 1};
    double M[9], R[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Hn[r * 3 + k] * Ts[k * 3 + c]; M[r * 3 + c] = a; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Tdi[r * 3 + k] * M[k * 3 + c]; R[r * 3 + c] = a; }
    if (std::fabs(R[8]) < 1e-12) return false;
    for (int i = 0; i < 9; i++) H[i] = R[i] / R[8];
    // Gauss-Newton on the forward transfer error, 8 parameters, 10 iterations
    for (int it = 0; it < 10; it++) {
        double JtJ[64], Jtr[8], dx[8];
        std::fill(JtJ, JtJ + 64, 0.0); std::fill(Jtr, Jtr + 8, 0.0);
        for (int i : in) {
            double X = s[i].x, Y = s[i].y;
            double w = H[6] * X + H[7] * Y + 1.0; if (std::fabs(w) < 1e-12) continue;
            double iw = 1.0 / w, px = (H[0] * X + H[1] * Y + H[2]) * iw, py = (H[3] * X + H[4] * Y + H[5]) * iw;
            double rx = d[i].x - px, ry = d[i].y - py;
            double jx[8] = {X * iw, Y * iw, iw, 0, 0, 0, -X * px * iw, -Y * px * iw};
            double jy[8] = {0, 0, 0, X * iw, Y * iw, iw, -X * py * iw, -Y * py * iw};
            for (int a = 0; a < 8; a++) {
                for (int b = 0; b < 8; b++) JtJ[a * 8 + b] += jx[a] * jx[b] + jy[a] * jy[b];
                Jtr[a] += jx[a] * rx + jy[a] * ry;
            }
        }
        if (!solve_linear(8, JtJ, Jtr, dx)) break;
        double step = 0;
        for (int a = 0; a < 8; a++) { H[a] += dx[a]; step += dx[a] * dx[a]; }
        if (step < 1e-20) break;
    }
    return true;
}
}  // namespace hdetail

// src -> dst homography; points sorted by decreasing quality.  Returns false (H = 0) on failure.
inline bool find_homography_prosac(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst, double H[9],
                                   double thresh = 3.0, double confidence = 0.995, int maxIters = 2000, uint64_t seed = 0x9E3779B97F4A7C15ull) {
    using namespace hdetail;
    const int N = (int)src.size();
    std::fill(H, H + 9, 0.0);
    if (N < 4) return false;
    const double t2 = thresh * thresh;
    RNG rng(seed);                 // other seeds: tests/test_a8_sensitivity_cpu.py (how much does the mask depend on the draw order?)
    // PROSAC growth function
    double Tn = maxIters;
    for (int i = 0; i < 4; i++) Tn *= (double)(4 - i) / (double)(N - i);
    int n = 4, Tn_prime = 1, best_cnt = 0, iters_needed = maxIters;
    double bestH[9];
    for (int t = 1; t <= iters_needed && t <= maxIters; t++) {
        if (t == Tn_prime && n < N) {
            double Tn1 = Tn * (double)(n + 1) / (double)(n + 1 - 4);
            Tn_prime += (int)std::ceil(Tn1 - Tn);
            Tn = Tn1; n++;
        }
        int idx[4];
        if (Tn_prime < t) {            // plain RANSAC draw from the first n
            for (int k = 0; k < 4; k++) { bool dup; do { idx[k] = (int)(rng.next() % (uint32_t)n); dup = false; for (int q = 0; q < k; q++) dup |= idx[q] == idx[k]; } while (dup); }
        } else {                       // n-th point plus 3 from the first n-1
            idx[3] = n - 1;
            for (int k = 0; k < 3; k++) { bool dup; do { idx[k] = (int)(rng.next() % (uint32_t)(n - 1)); dup = false; for (int q = 0; q < k; q++) dup |= idx[q] == idx[k]; } while (dup); }
        }
        if (degenerate4(src.data(), idx) || degenerate4(dst.data(), idx)) continue;
        double Hc[9];
        if (!h_from_4(src.data(), dst.data(), idx, Hc)) continue;
        int cnt = 0;
        for (int i = 0; i < N; i++) cnt += reproj_err2(Hc, src[i], dst[i]) <= t2;
        if (cnt > best_cnt) {
            best_cnt = cnt; std::copy(Hc, Hc + 9, bestH);
            double eps = (double)cnt / N, p4 = eps * eps * eps * eps;
            if (p4 > 1.0 - 1e-12) iters_needed = t;
            else { double k = std::log(1.0 - confidence) / std::log(1.0 - p4); iters_needed = (int)std::min<double>(maxIters, std::ceil(k)); }
        }
    }
    if (best_cnt < 4) return false;
    std::vector<int> inl;
    for (int i = 0; i < N; i++) if (reproj_err2(bestH, src[i], dst[i]) <= t2) inl.push_back(i);
    double Hr[9]; std::copy(bestH, bestH + 9, Hr);
    if (refine_h(src.data(), dst.data(), inl, Hr)) {
        // keep the refinement only if it does not lose support
        int cnt = 0; for (int i = 0; i < N; i++) cnt += reproj_err2(Hr, src[i], dst[i]) <= t2;
        if (cnt >= best_cnt) std::copy(Hr, Hr + 9, bestH);
    }
    std::copy(bestH, bestH + 9, H);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The published scheme behind cv::RHO (Bazargani, Bilaniuk & Laganiere, "A fast and robust homography scheme for real-time planar target
// detection", 2015; OpenCV calib3d/src/rho.cpp), restated from the paper and from memory of rho.cpp's structure (see the header of this file for what
// is recalled and what is guessed): PROSAC sampling over the quality-sorted correspondences, a float32 minimal solver, the orientation test on the
// sample, SPRT evaluation (t_M = 25, m_S = 1, eps_0 = 0.1, delta_0 = 0.01), the confidence bound on the iteration count, Levenberg-Marquardt on the
// inliers in float32 with a damped Cholesky solve.  It is NOT bit-compatible with OpenCV.  The product's host/homography.cpp is a transliteration of
// THIS function (parity by construction); find_homography_prosac above is the independent one.
namespace rho_scheme {
struct Xs128 { uint64_t s[2];
    explicit Xs128(uint64_t seed) { s[0] = seed ^ 0x2545F4914F6CDD1Dull; s[1] = ~seed + 0x9E3779B97F4A7C15ull; for (int i = 0; i < 20; i++) next(); }
    uint64_t next() { uint64_t x = s[0]; const uint64_t y = s[1]; s[0] = y; x ^= x << 23; s[1] = x ^ y ^ (x >> 17) ^ (y >> 26); return s[1] + y; }
    unsigned below(unsigned n) { return (unsigned)((double)(next() >> 11) * (1.0 / 9007199254740992.0) * n); } };
inline bool solve4(const Pt2f* s, const Pt2f* d, const unsigned id[4], float H[9]) {      // 8 x 9 augmented system, float32 Gauss-Jordan with row pivoting
    float M[8][9];
    for (int i = 0; i < 4; i++) { const float X = s[id[i]].x, Y = s[id[i]].y, u = d[id[i]].x, v = d[id[i]].y;
        const float r0[9] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y, u}, r1[9] = {0, 0, 0, X, Y, 1, -v * X, -v * Y, v};
        for (int k = 0; k < 9; k++) { M[2 * i][k] = r0[k]; M[2 * i + 1][k] = r1[k]; } }
    for (int c = 0; c < 8; c++) {
        int p = c; for (int r = c + 1; r < 8; r++) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (std::fabs(M[p][c]) < 1e-9f) return false;
        if (p != c) for (int k = 0; k < 9; k++) std::swap(M[p][k], M[c][k]);
        const float inv = 1.0f / M[c][c];
        for (int k = c; k < 9; k++) M[c][k] *= inv;
        for (int r = 0; r < 8; r++) if (r != c) { const float f = M[r][c]; if (f != 0.f) for (int k = c; k < 9; k++) M[r][k] -= f * M[c][k]; }
    }
    for (int i = 0; i < 8; i++) { H[i] = M[i][8]; if (!(H[i] == H[i])) return false; }
    H[8] = 1.f; return true;
}
inline float orient(const Pt2f& a, const Pt2f& b, const Pt2f& c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }
inline bool sample_ok(const Pt2f* s, const Pt2f* d, const unsigned id[4]) {           // no three points collinear, orientation of every triple preserved
    static const int T[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    for (auto& t : T) { const float a = orient(s[id[t[0]]], s[id[t[1]]], s[id[t[2]]]), b = orient(d[id[t[0]]], d[id[t[1]]], d[id[t[2]]]);
        if (std::fabs(a) < 1e-3f || std::fabs(b) < 1e-3f || (a > 0) != (b > 0)) return false; }
    return true;
}
inline float err2(const float H[9], const Pt2f& s, const Pt2f& d) {
    const float w = H[6] * s.x + H[7] * s.y + H[8]; if (std::fabs(w) < 1e-12f) return 1e30f;
    const float iw = 1.0f / w, ex = (H[0] * s.x + H[1] * s.y + H[2]) * iw - d.x, ey = (H[3] * s.x + H[4] * s.y + H[5]) * iw - d.y;
    return ex * ex + ey * ey;
}
struct Sprt { double eps, delta, A, lamAccept, lamReject;
    void design(double e, double dl) { eps = e; delta = dl; const double tM = 25, mS = 1;
        const double C = (1 - delta) * std::log((1 - delta) / (1 - eps)) + delta * std::log(delta / eps), K = tM * C / mS + 1;
        double a = K; for (int i = 0; i < 10; i++) a = K + std::log(a);
        A = a; lamAccept = delta / eps; lamReject = (1 - delta) / (1 - eps); } };
inline bool chol8(float A[8][8], float b[8], float x[8]) {                            // A = L L^T in place (lower), then the two triangular solves
    for (int j = 0; j < 8; j++) { float sum = A[j][j]; for (int k = 0; k < j; k++) sum -= A[j][k] * A[j][k];
        if (!(sum > 0.f)) return false; A[j][j] = std::sqrt(sum);
        for (int i = j + 1; i < 8; i++) { float t = A[i][j]; for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k]; A[i][j] = t / A[j][j]; } }
    float y[8];
    for (int i = 0; i < 8; i++) { float t = b[i]; for (int k = 0; k < i; k++) t -= A[i][k] * y[k]; y[i] = t / A[i][i]; }
    for (int i = 7; i >= 0; i--) { float t = y[i]; for (int k = i + 1; k < 8; k++) t -= A[k][i] * x[k]; x[i] = t / A[i][i]; }
    return true;
}
inline float lm_cost(const Pt2f* s, const Pt2f* d, const std::vector<unsigned>& in, const float H[9], float JtJ[8][8], float Jtr[8]) {
    float cost = 0.f;
    if (JtJ) { for (int a = 0; a < 8; a++) { Jtr[a] = 0.f; for (int b = 0; b < 8; b++) JtJ[a][b] = 0.f; } }
    for (unsigned i : in) { const float X = s[i].x, Y = s[i].y, w = H[6] * X + H[7] * Y + 1.0f; if (std::fabs(w) < 1e-12f) continue;
        const float iw = 1.0f / w, px = (H[0] * X + H[1] * Y + H[2]) * iw, py = (H[3] * X + H[4] * Y + H[5]) * iw, rx = d[i].x - px, ry = d[i].y - py;
        cost += rx * rx + ry * ry;
        if (JtJ) { const float jx[8] = {X * iw, Y * iw, iw, 0, 0, 0, -X * px * iw, -Y * px * iw}, jy[8] = {0, 0, 0, X * iw, Y * iw, iw, -X * py * iw, -Y * py * iw};
            for (int a = 0; a < 8; a++) { Jtr[a] += jx[a] * rx + jy[a] * ry; for (int b = 0; b <= a; b++) JtJ[a][b] += jx[a] * jx[b] + jy[a] * jy[b]; } } }
    return cost;
}
}  // namespace rho_scheme

inline bool find_homography_rho_scheme(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst, double Hout[9],
                                       float maxD = 3.0f, double cfd = 0.995, unsigned maxI = 2000, uint64_t seed = ~0ull) {
    using namespace rho_scheme;
    const unsigned N = (unsigned)src.size(); std::fill(Hout, Hout + 9, 0.0);
    if (N < 4) return false;
    const Pt2f* s = src.data(); const Pt2f* d = dst.data(); const float maxD2 = maxD * maxD;
    Xs128 rng(seed); Sprt sprt; sprt.design(0.1, 0.01);
    // SPRT tests the points in a fixed random order (Matas & Chum): in index order a run of bad leading correspondences -- the PROSAC ranking is
    // only a prior -- would reject every model, the right one included, after a dozen points
    std::vector<unsigned> order(N); for (unsigned i = 0; i < N; i++) order[i] = i;
    for (unsigned i = N - 1; i > 0; i--) std::swap(order[i], order[rng.below(i + 1)]);
    // PROSAC (Chum & Matas 2005): the n best correspondences are sampled until T'_n hypotheses were drawn, then n grows
    double Tn = maxI; for (int i = 0; i < 4; i++) Tn *= (double)(4 - i) / (double)(N - i);
    unsigned n = 4, TnPrime = 1, bestInl = 0, limit = maxI; float best[9] = {0};
    for (unsigned it = 1; it <= limit; it++) {
        if (it > TnPrime && n < N) { const double Tn1 = Tn * (n + 1) / (double)(n + 1 - 4); TnPrime += (unsigned)std::ceil(Tn1 - Tn); Tn = Tn1; n++; }
        unsigned id[4]; const bool prosac = n < N;
        const unsigned pool = prosac ? n - 1 : N, need = prosac ? 3 : 4;
        for (unsigned k = 0; k < need; k++) { bool dup; do { id[k] = rng.below(pool); dup = false; for (unsigned q = 0; q < k; q++) dup |= id[q] == id[k]; } while (dup); }
        if (prosac) id[3] = n - 1;
        if (!sample_ok(s, d, id)) continue;
        float Hc[9]; if (!solve4(s, d, id, Hc)) continue;
        // SPRT: multiply the likelihood ratio point by point; a bad model is rejected after a few points
        double lambda = 1.0; unsigned inl = 0, tested = 0; bool rejected = false;
        for (unsigned q = 0; q < N; q++) { const unsigned i = order[q]; const bool in = err2(Hc, s[i], d[i]) <= maxD2; inl += in; tested++;
            lambda *= in ? sprt.lamAccept : sprt.lamReject;
            if (lambda > sprt.A) { rejected = true; break; } }
        if (rejected) { const double dl = (double)inl / tested; if (dl > 0 && std::fabs(dl - sprt.delta) / sprt.delta > 0.05 && dl < sprt.eps) sprt.design(sprt.eps, dl); continue; }
        if (inl > bestInl) { bestInl = inl; std::copy(Hc, Hc + 9, best);
            const double e = (double)inl / N; if (e > sprt.eps) sprt.design(e, sprt.delta);
            const double p4 = e * e * e * e; limit = p4 > 1 - 1e-12 ? it : (unsigned)std::min<double>(maxI, std::ceil(std::log(1 - cfd) / std::log(1 - p4))); }
    }
    if (bestInl < 4) return false;
    std::vector<unsigned> in; for (unsigned i = 0; i < N; i++) if (err2(best, s[i], d[i]) <= maxD2) in.push_back(i);
    // Levenberg-Marquardt on the inliers (8 parameters, h33 = 1), damped Cholesky, float32
    float H[9]; std::copy(best, best + 9, H); float lam = 0.01f, JtJ[8][8], Jtr[8]; float cost = lm_cost(s, d, in, H, JtJ, Jtr);
    for (int it = 0; it < 10; it++) {
        float A[8][8], b[8], dx[8];
        for (int a = 0; a < 8; a++) { b[a] = Jtr[a]; for (int c = 0; c <= a; c++) A[a][c] = JtJ[a][c]; A[a][a] += lam * JtJ[a][a] + 1e-12f; }
        if (!chol8(A, b, dx)) { lam *= 10.f; continue; }
        float Hn[9]; for (int a = 0; a < 8; a++) Hn[a] = H[a] + dx[a]; Hn[8] = 1.f;
        const float c2 = lm_cost(s, d, in, Hn, nullptr, nullptr);
        if (c2 < cost) { std::copy(Hn, Hn + 9, H); lam *= 0.1f; cost = lm_cost(s, d, in, H, JtJ, Jtr); } else lam *= 10.f;
    }
    for (int i = 0; i < 9; i++) Hout[i] = H[i];
    return true;
}

}  // namespace cvx
