// Host stage: robust homography between quality-sorted correspondences, in place of cv::findHomography(..., cv::RHO) at reference
// DynaDetect.cc:1235 (63 x 47 grid samples; serial, order-sensitive, a few kB of data -> host).
//
// OpenCV's RHO (calib3d/src/rho.cpp) is not vendored by the reference and cannot be reproduced bit for bit offline (PRNG seeding, the N*
// non-randomness test and several constants).  What is implemented here is RHO's PUBLISHED scheme (Bazargani, Bilaniuk & Laganiere 2015):
//   1. PROSAC sampling (Chum & Matas 2005 growth function, T_N = 2000): the newest point of the growing subset plus three of the earlier ones;
//   2. the sample must keep the orientation of every point triple (and no triple collinear); 4-point minimal solver with h33 = 1 in FP32;
//   3. SPRT evaluation (Matas & Chum: likelihood ratio per point in a fixed random order, a bad model is rejected after a handful of points; t_M = 25, m_S = 1,
//      eps_0 = 0.1, delta_0 = 0.01, both re-estimated on the way), support = transfer error <= 3 px, iteration bound from confidence 0.995;
//   4. Levenberg-Marquardt on the inliers of the best model (8 parameters, damped Cholesky, FP32), ten iterations.
// Why this and not a lighter PROSAC + least squares (round 1): measured on the synthetic stream, that estimator moved the image corners by up
// to 2.9 px between PRNG seeds and the dynamic mask with it (IoU down to 0.85, tests/test_a8_sensitivity_cpu.py), i.e. the mask depended on
// the draw order; this scheme agrees with itself to <= 0.06 px / IoU >= 0.9997 across seeds, so the stage no longer decides the mask.
// The CPU checker used by the tests carries its own restatement of the same scheme; both are run on the same pairs (tests/test_host_stages_cpu.py).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "host.hpp"

namespace sind {
namespace {

class RhoScheme {
public:
    RhoScheme(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst) : n_((unsigned)src.size()), sx_(n_), sy_(n_), dx_(n_), dy_(n_) {
        for (unsigned i = 0; i < n_; i++) { sx_[i] = src[i].x; sy_[i] = src[i].y; dx_[i] = dst[i].x; dy_[i] = dst[i].y; }
        rs_[0] = ~0ull ^ 0x2545F4914F6CDD1Dull; rs_[1] = ~(~0ull) + 0x9E3779B97F4A7C15ull;       // xorshift128+, fixed seed, 20 warm-up draws
        for (int i = 0; i < 20; i++) draw();
        design(0.1, 0.01);
        // SPRT visits the points in a fixed random order (Matas & Chum): in index order a run of bad leading pairs -- the PROSAC ranking is only
        // a prior -- would reject every model, the right one included, after a dozen points
        visit_.resize(n_); for (unsigned i = 0; i < n_; i++) visit_[i] = i;
        for (unsigned i = n_ > 0 ? n_ - 1 : 0; i > 0; i--) std::swap(visit_[i], visit_[below(i + 1)]);
    }
    bool run(double H[9]) {
        std::fill(H, H + 9, 0.0);
        if (n_ < 4) return false;
        search();
        if (best_support_ < 4) return false;
        polish();
        for (int i = 0; i < 9; i++) H[i] = best_[i];
        return true;
    }

private:
    static constexpr float kMaxD2 = 3.0f * 3.0f;
    static constexpr unsigned kMaxIter = 2000;
    unsigned n_; std::vector<float> sx_, sy_, dx_, dy_; std::vector<unsigned> visit_;
    uint64_t rs_[2];
    double eps_ = 0, delta_ = 0, A_ = 0, up_ = 0, down_ = 0;      // SPRT: likelihood-ratio factors of an inlier (up_) / outlier (down_), decision threshold A_
    float best_[9] = {0}; unsigned best_support_ = 0;

    uint64_t draw() { uint64_t x = rs_[0]; const uint64_t y = rs_[1]; rs_[0] = y; x ^= x << 23; rs_[1] = x ^ y ^ (x >> 17) ^ (y >> 26); return rs_[1] + y; }
    unsigned below(unsigned m) { return (unsigned)((double)(draw() >> 11) * (1.0 / 9007199254740992.0) * m); }
    void design(double eps, double delta) {
        eps_ = eps; delta_ = delta;
        const double C = (1 - delta) * std::log((1 - delta) / (1 - eps)) + delta * std::log(delta / eps), K = 25.0 * C / 1.0 + 1;
        double a = K; for (int i = 0; i < 10; i++) a = K + std::log(a);
        A_ = a; up_ = delta / eps; down_ = (1 - delta) / (1 - eps);
    }
    float tri(const std::vector<float>& x, const std::vector<float>& y, unsigned a, unsigned b, unsigned c) const { return (x[b] - x[a]) * (y[c] - y[a]) - (y[b] - y[a]) * (x[c] - x[a]); }
    bool sample_usable(const unsigned id[4]) const {
        static const int T[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
        for (const auto& t : T) {
            const float a = tri(sx_, sy_, id[t[0]], id[t[1]], id[t[2]]), b = tri(dx_, dy_, id[t[0]], id[t[1]], id[t[2]]);
            if (std::fabs(a) < 1e-3f || std::fabs(b) < 1e-3f || (a > 0) != (b > 0)) return false;
        }
        return true;
    }
    bool minimal(const unsigned id[4], float H[9]) const {         // 8 x 9 augmented system, FP32 Gauss-Jordan, row pivoting
        float M[8][9];
        for (int i = 0; i < 4; i++) {
            const float X = sx_[id[i]], Y = sy_[id[i]], u = dx_[id[i]], v = dy_[id[i]];
            const float r0[9] = {X, Y, 1, 0, 0, 0, -u * X, -u * Y, u}, r1[9] = {0, 0, 0, X, Y, 1, -v * X, -v * Y, v};
            std::memcpy(M[2 * i], r0, sizeof(r0)); std::memcpy(M[2 * i + 1], r1, sizeof(r1));
        }
        for (int c = 0; c < 8; c++) {
            int p = c; for (int r = c + 1; r < 8; r++) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
            if (std::fabs(M[p][c]) < 1e-9f) return false;
            if (p != c) for (int k = 0; k < 9; k++) std::swap(M[p][k], M[c][k]);
            const float inv = 1.0f / M[c][c];
            for (int k = c; k < 9; k++) M[c][k] *= inv;
            for (int r = 0; r < 8; r++) { if (r == c) continue; const float f = M[r][c]; if (f != 0.f) for (int k = c; k < 9; k++) M[r][k] -= f * M[c][k]; }
        }
        for (int i = 0; i < 8; i++) { H[i] = M[i][8]; if (!(H[i] == H[i])) return false; }
        H[8] = 1.f; return true;
    }
    float transfer2(const float H[9], unsigned i) const {
        const float w = H[6] * sx_[i] + H[7] * sy_[i] + H[8]; if (std::fabs(w) < 1e-12f) return 1e30f;
        const float iw = 1.0f / w, ex = (H[0] * sx_[i] + H[1] * sy_[i] + H[2]) * iw - dx_[i], ey = (H[3] * sx_[i] + H[4] * sy_[i] + H[5]) * iw - dy_[i];
        return ex * ex + ey * ey;
    }
    void search() {
        double Tn = kMaxIter; for (int i = 0; i < 4; i++) Tn *= (double)(4 - i) / (double)(n_ - i);
        unsigned grow = 4, TnPrime = 1, limit = kMaxIter;
        for (unsigned it = 1; it <= limit; it++) {
            if (it > TnPrime && grow < n_) { const double Tn1 = Tn * (grow + 1) / (double)(grow + 1 - 4); TnPrime += (unsigned)std::ceil(Tn1 - Tn); Tn = Tn1; grow++; }
            unsigned id[4]; const bool prosac = grow < n_; const unsigned pool = prosac ? grow - 1 : n_, need = prosac ? 3 : 4;
            for (unsigned k = 0; k < need; k++) { bool dup; do { id[k] = below(pool); dup = false; for (unsigned q = 0; q < k; q++) dup |= id[q] == id[k]; } while (dup); }
            if (prosac) id[3] = grow - 1;
            if (!sample_usable(id)) continue;
            float H[9]; if (!minimal(id, H)) continue;
            double ratio = 1.0; unsigned support = 0, seen = 0; bool rejected = false;
            for (unsigned q = 0; q < n_; q++) {
                const bool in = transfer2(H, visit_[q]) <= kMaxD2; support += in; seen++;
                ratio *= in ? up_ : down_;
                if (ratio > A_) { rejected = true; break; }
            }
            if (rejected) { const double dl = (double)support / seen; if (dl > 0 && std::fabs(dl - delta_) / delta_ > 0.05 && dl < eps_) design(eps_, dl); continue; }
            if (support > best_support_) {
                best_support_ = support; std::memcpy(best_, H, sizeof(best_));
                const double e = (double)support / n_; if (e > eps_) design(e, delta_);
                const double p4 = e * e * e * e;
                limit = p4 > 1 - 1e-12 ? it : (unsigned)std::min<double>(kMaxIter, std::ceil(std::log(1 - 0.995) / std::log(1 - p4)));
            }
        }
    }
    // sum of squared transfer errors over the inlier list; with JtJ != nullptr also the normal equations (lower triangle)
    float normal_eq(const std::vector<unsigned>& in, const float H[9], float JtJ[8][8], float Jtr[8]) const {
        float cost = 0.f;
        if (JtJ) { std::memset(Jtr, 0, 8 * sizeof(float)); std::memset(JtJ, 0, 64 * sizeof(float)); }
        for (unsigned i : in) {
            const float X = sx_[i], Y = sy_[i], w = H[6] * X + H[7] * Y + 1.0f; if (std::fabs(w) < 1e-12f) continue;
            const float iw = 1.0f / w, px = (H[0] * X + H[1] * Y + H[2]) * iw, py = (H[3] * X + H[4] * Y + H[5]) * iw, rx = dx_[i] - px, ry = dy_[i] - py;
            cost += rx * rx + ry * ry;
            if (!JtJ) continue;
            const float jx[8] = {X * iw, Y * iw, iw, 0, 0, 0, -X * px * iw, -Y * px * iw}, jy[8] = {0, 0, 0, X * iw, Y * iw, iw, -X * py * iw, -Y * py * iw};
            for (int a = 0; a < 8; a++) { Jtr[a] += jx[a] * rx + jy[a] * ry; for (int b = 0; b <= a; b++) JtJ[a][b] += jx[a] * jx[b] + jy[a] * jy[b]; }
        }
        return cost;
    }
    static bool cholesky_solve(float A[8][8], const float b[8], float x[8]) {
        for (int j = 0; j < 8; j++) {
            float sum = A[j][j]; for (int k = 0; k < j; k++) sum -= A[j][k] * A[j][k];
            if (!(sum > 0.f)) return false;
            A[j][j] = std::sqrt(sum);
            for (int i = j + 1; i < 8; i++) { float t = A[i][j]; for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k]; A[i][j] = t / A[j][j]; }
        }
        float y[8];
        for (int i = 0; i < 8; i++) { float t = b[i]; for (int k = 0; k < i; k++) t -= A[i][k] * y[k]; y[i] = t / A[i][i]; }
        for (int i = 7; i >= 0; i--) { float t = y[i]; for (int k = i + 1; k < 8; k++) t -= A[k][i] * x[k]; x[i] = t / A[i][i]; }
        return true;
    }
    void polish() {
        std::vector<unsigned> in; for (unsigned i = 0; i < n_; i++) if (transfer2(best_, i) <= kMaxD2) in.push_back(i);
        float H[9]; std::memcpy(H, best_, sizeof(H));
        float lam = 0.01f, JtJ[8][8], Jtr[8]; float cost = normal_eq(in, H, JtJ, Jtr);
        for (int it = 0; it < 10; it++) {
            float A[8][8], step[8];
            for (int a = 0; a < 8; a++) { for (int c = 0; c <= a; c++) A[a][c] = JtJ[a][c]; A[a][a] += lam * JtJ[a][a] + 1e-12f; }
            if (!cholesky_solve(A, Jtr, step)) { lam *= 10.f; continue; }
            float Hn[9]; for (int a = 0; a < 8; a++) Hn[a] = H[a] + step[a]; Hn[8] = 1.f;
            const float c2 = normal_eq(in, Hn, nullptr, nullptr);
            if (c2 < cost) { std::memcpy(H, Hn, sizeof(H)); lam *= 0.1f; cost = normal_eq(in, H, JtJ, Jtr); } else lam *= 10.f;
        }
        std::memcpy(best_, H, sizeof(best_));
    }
};

}  // namespace

bool find_homography_rho(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst, double H[9]) { return RhoScheme(src, dst).run(H); }

}  // namespace sind
