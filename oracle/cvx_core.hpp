// ORACLE — test infrastructure only.  CPU restatement of the OpenCV 4.2.0 primitives that
// qimao7213/SInDSLAM's DynaDetect / ORBextractor hot path calls (SURVEY.md §8c lists the call
// sites).  OpenCV itself is NOT vendored under /root/reference and is absent from this image, so
// every routine here restates the published OpenCV 4.2.0 algorithm from its documented behaviour.
// PARITY UNPINNED: the reference holds no golden vectors for this path.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under
// oracle/.  The product (sindslam_amd/) never links or imports it.
//
// Build flags that matter: -ffp-contract=off (OpenCV's scalar code paths are plain mul/add;
// the HIP kernels are built the same way so FP32 stages can be compared bit for bit).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace cvx {

// ---------------------------------------------------------------- image container
template <class T>
struct Img {
    int w = 0, h = 0, c = 1;
    std::vector<T> d;
    Img() {}
    Img(int w_, int h_, int c_ = 1, T v = T()) { create(w_, h_, c_, v); }
    void create(int w_, int h_, int c_ = 1, T v = T()) {
        w = w_; h = h_; c = c_;
        d.assign((size_t)w * h * c, v);
    }
    bool empty() const { return d.empty(); }
    size_t size() const { return d.size(); }
    T* row(int y) { return d.data() + (size_t)y * w * c; }
    const T* row(int y) const { return d.data() + (size_t)y * w * c; }
    T& at(int y, int x, int ch = 0) { return d[((size_t)y * w + x) * c + ch]; }
    const T& at(int y, int x, int ch = 0) const { return d[((size_t)y * w + x) * c + ch]; }
    void fill(T v) { std::fill(d.begin(), d.end(), v); }
};
using Img8 = Img<uint8_t>;
using Img16 = Img<uint16_t>;
using ImgF = Img<float>;
using ImgI = Img<int32_t>;

struct Pt { int x, y; };

// ---------------------------------------------------------------- scalar helpers
// cv::cvRound: round half to even (lrint under the default rounding mode).
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRoundf(float v) { return (int)std::lrintf(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline uint8_t sat_u8f(float v) { return sat_u8(cvRoundf(v)); }
static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
// BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba)
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

// ---------------------------------------------------------------- cvtColor(BGR2GRAY / RGB2GRAY), 8U
// OpenCV 8-bit path: fixed point, 14 fractional bits: (B*1868 + G*9617 + R*4899 + 8192) >> 14.
// swap_rb=true gives COLOR_RGB2GRAY applied to the same buffer (Tracking.cc:246-251 quirk).
inline void bgr2gray(const Img8& bgr, Img8& gray, bool swap_rb = false) {
    gray.create(bgr.w, bgr.h, 1);
    const int cb = swap_rb ? 4899 : 1868, cr = swap_rb ? 1868 : 4899;
    for (size_t i = 0, n = (size_t)bgr.w * bgr.h; i < n; i++) {
        const uint8_t* p = &bgr.d[i * 3];
        gray.d[i] = (uint8_t)((p[0] * cb + p[1] * 9617 + p[2] * cr + 8192) >> 14);
    }
}

// ---------------------------------------------------------------- resize, INTER_LINEAR
// Coordinate tables shared by all depths (imgproc/resize.cpp, cv::resize linear branch):
// fx = (dx+0.5)*scale-0.5, sx=floor(fx), clamp at both ends with fx:=0.
struct LinTab { std::vector<int> ofs; std::vector<float> a; int dmax; };
// vertical table: OpenCV keeps the raw (sy, fy) pair and clips the two source ROWS instead.
inline LinTab lin_tab_y(int ssize, int dsize) {
    LinTab t; t.ofs.resize(dsize); t.a.resize(dsize); t.dmax = dsize;
    double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int dy = 0; dy < dsize; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = cvFloor(fy);
        t.ofs[dy] = sy; t.a[dy] = fy - sy;
    }
    return t;
}
inline LinTab lin_tab(int ssize, int dsize) {
    LinTab t; t.ofs.resize(dsize); t.a.resize(dsize); t.dmax = dsize;
    double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int dx = 0; dx < dsize; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= ssize) {           // xmax: beyond here only S[sx] is read
            t.dmax = std::min(t.dmax, dx);
            if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; }
        }
        t.ofs[dx] = sx; t.a[dx] = fx;
    }
    return t;
}

// 8U: 11-bit fixed-point coefficients, HResize to int (x2048), VResize >>4, *b>>16, +2 >>2.
inline void resize_linear_u8(const Img8& src, Img8& dst, int dw, int dh) {
    if (dw == src.w && dh == src.h) { dst = src; return; }
    dst.create(dw, dh, 1);
    LinTab tx = lin_tab(src.w, dw), ty = lin_tab_y(src.h, dh);
    std::vector<short> ax(2 * dw), ay(2 * dh);
    for (int i = 0; i < dw; i++) { ax[2*i] = (short)cvRoundf((1.f - tx.a[i]) * 2048); ax[2*i+1] = (short)cvRoundf(tx.a[i] * 2048); }
    for (int i = 0; i < dh; i++) { ay[2*i] = (short)cvRoundf((1.f - ty.a[i]) * 2048); ay[2*i+1] = (short)cvRoundf(ty.a[i] * 2048); }
    std::vector<int> r0(dw), r1(dw);
    auto hrow = [&](int sy, std::vector<int>& out) {
        const uint8_t* S = src.row(sy);
        for (int dx = 0; dx < dw; dx++) {
            int sx = tx.ofs[dx];
            out[dx] = dx < tx.dmax ? S[sx] * ax[2*dx] + S[sx+1] * ax[2*dx+1] : S[sx] * 2048;
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy = clipi(ty.ofs[dy], 0, src.h), sy1 = clipi(ty.ofs[dy] + 1, 0, src.h);
        hrow(sy, r0); hrow(sy1, r1);
        int b0 = ay[2*dy], b1 = ay[2*dy+1];
        uint8_t* D = dst.row(dy);
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// 32F (1..n channels): float coefficients, D = S0*b0 + S1*b1 over H-resized float rows.
inline void resize_linear_f32(const ImgF& src, ImgF& dst, int dw, int dh) {
    const int cn = src.c;
    if (dw == src.w && dh == src.h) { dst = src; return; }
    dst.create(dw, dh, cn);
    LinTab tx = lin_tab(src.w, dw), ty = lin_tab_y(src.h, dh);
    std::vector<float> r0((size_t)dw * cn), r1((size_t)dw * cn);
    auto hrow = [&](int sy, std::vector<float>& out) {
        const float* S = src.row(sy);
        for (int dx = 0; dx < dw; dx++) {
            int sx = tx.ofs[dx]; float a1 = tx.a[dx], a0 = 1.f - a1;
            for (int k = 0; k < cn; k++)
                out[dx*cn+k] = dx < tx.dmax ? S[sx*cn+k] * a0 + S[(sx+1)*cn+k] * a1 : S[sx*cn+k] * 1.f;
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy = clipi(ty.ofs[dy], 0, src.h), sy1 = clipi(ty.ofs[dy] + 1, 0, src.h);
        hrow(sy, r0); hrow(sy1, r1);
        float b1 = ty.a[dy], b0 = 1.f - b1;
        float* D = dst.row(dy);
        for (int x = 0; x < dw * cn; x++) D[x] = r0[x] * b0 + r1[x] * b1;
    }
}

// 16U exact 2x decimation: cv::resize turns INTER_LINEAR into INTER_AREA when both scales are
// exactly 2 (resize.cpp "is_area_fast"); 2x2 box mean, (sum+2)>>2.  DynaDetect.cc:333.
inline void resize_half_u16(const Img16& src, Img16& dst) {
    int dw = src.w / 2, dh = src.h / 2;
    dst.create(dw, dh, 1);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int s = src.at(2*y, 2*x) + src.at(2*y, 2*x+1) + src.at(2*y+1, 2*x) + src.at(2*y+1, 2*x+1);
            dst.at(y, x) = (uint16_t)((s + 2) >> 2);
        }
}

// ---------------------------------------------------------------- Gaussian kernels / blur
inline std::vector<double> gaussian_kernel(int n, double sigma) {
    // getGaussianKernel (sigma>0): exp(-x^2/(2 sigma^2)) normalised to unit sum.
    std::vector<double> k(n); double sum = 0;
    for (int i = 0; i < n; i++) { double x = i - (n - 1) * 0.5; k[i] = std::exp(-0.5 * x * x / (sigma * sigma)); sum += k[i]; }
    for (auto& v : k) v /= sum;
    return k;
}

// 32F separable blur, BORDER_REFLECT_101, symmetric 3-tap form used by sepFilter2D:
// row: c*k0 + (l+r)*k1 ; column the same.  (deepflow.cpp pre-smoothing, sigma 0.6 -> ksize 3)
inline void gaussian_blur3_f32(const ImgF& src, ImgF& dst, double sigma) {
    std::vector<double> kd = gaussian_kernel(3, sigma);
    const float k0 = (float)kd[1], k1 = (float)kd[0];
    ImgF tmp(src.w, src.h, 1);
    for (int y = 0; y < src.h; y++) {
        const float* S = src.row(y); float* T = tmp.row(y);
        for (int x = 0; x < src.w; x++) {
            float l = S[reflect101(x - 1, src.w)], r = S[reflect101(x + 1, src.w)];
            T[x] = S[x] * k0 + (l + r) * k1;
        }
    }
    dst.create(src.w, src.h, 1);
    for (int y = 0; y < src.h; y++) {
        const float* U = tmp.row(reflect101(y - 1, src.h)); const float* C = tmp.row(y);
        const float* L = tmp.row(reflect101(y + 1, src.h)); float* D = dst.row(y);
        for (int x = 0; x < src.w; x++) D[x] = C[x] * k0 + (U[x] + L[x]) * k1;
    }
}

// 8U GaussianBlur fixed-point path (smooth.dispatch.cpp, OpenCV 4.2.0): kernel taps rounded to
// 8.8 fixed point (cvRound(k*256), no error diffusion in 4.2.0), horizontal pass in 8.8 (u16,
// saturating), vertical pass in 16.16 (u32), result (v + 2^15) >> 16, BORDER_REFLECT_101.
inline std::vector<int> gaussian_kernel_fx8(int n, double sigma) {
    std::vector<double> k = gaussian_kernel(n, sigma);
    std::vector<int> f(n);
    for (int i = 0; i < n; i++) f[i] = cvRound(k[i] * 256.0);
    return f;
}
inline void gaussian_blur_u8(const Img8& src, Img8& dst, int ksize, double sigma) {
    std::vector<int> k = gaussian_kernel_fx8(ksize, sigma);
    const int r = ksize / 2;
    Img<uint16_t> tmp(src.w, src.h, 1);
    for (int y = 0; y < src.h; y++) {
        const uint8_t* S = src.row(y);
        for (int x = 0; x < src.w; x++) {
            uint32_t s = 0;
            for (int i = -r; i <= r; i++) s += (uint32_t)k[i + r] * S[reflect101(x + i, src.w)];
            tmp.at(y, x) = (uint16_t)std::min<uint32_t>(s, 65535u);
        }
    }
    dst.create(src.w, src.h, 1);
    for (int y = 0; y < src.h; y++)
        for (int x = 0; x < src.w; x++) {
            uint32_t s = 0;
            for (int i = -r; i <= r; i++) s += (uint32_t)k[i + r] * tmp.at(reflect101(y + i, src.h), x);
            dst.at(y, x) = sat_u8((int)((s + (1u << 15)) >> 16));
        }
}

// copyMakeBorder(BORDER_REFLECT_101)
inline void pad_reflect101_u8(const Img8& src, Img8& dst, int b) {
    dst.create(src.w + 2 * b, src.h + 2 * b, 1);
    for (int y = 0; y < dst.h; y++) {
        const uint8_t* S = src.row(reflect101(y - b, src.h));
        uint8_t* D = dst.row(y);
        for (int x = 0; x < dst.w; x++) D[x] = S[reflect101(x - b, src.w)];
    }
}

// ---------------------------------------------------------------- histogram thresholds
inline void hist256(const Img8& img, int h[256]) {
    std::memset(h, 0, 256 * sizeof(int));
    for (uint8_t v : img.d) h[v]++;
}
// getThreshVal_Otsu_8u (imgproc/thresh.cpp)
inline double otsu_from_hist(const int h[256], int total) {
    const int N = 256;
    double mu = 0, scale = 1. / total;
    for (int i = 0; i < N; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < N; i++) {
        double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return max_val;
}
// getThreshVal_Triangle_8u (imgproc/thresh.cpp)
inline double triangle_from_hist(const int hin[256]) {
    const int N = 256;
    int h[N]; std::memcpy(h, hin, sizeof(h));
    int left_bound = 0, right_bound = 0, max_ind = 0, mx = 0;
    bool isflipped = false;
    for (int i = 0; i < N; i++) if (h[i] > 0) { left_bound = i; break; }
    if (left_bound > 0) left_bound--;
    for (int i = N - 1; i > 0; i--) if (h[i] > 0) { right_bound = i; break; }
    if (right_bound < N - 1) right_bound++;
    for (int i = 0; i < N; i++) if (h[i] > mx) { mx = h[i]; max_ind = i; }
    if (max_ind - left_bound < right_bound - max_ind) {
        isflipped = true;
        for (int i = 0, j = N - 1; i < j; i++, j--) std::swap(h[i], h[j]);
        left_bound = N - 1 - right_bound;
        max_ind = N - 1 - max_ind;
    }
    double thresh = left_bound, a = mx, b = left_bound - max_ind, dist = 0;
    for (int i = left_bound + 1; i <= max_ind; i++) {
        double tempdist = a * i + b * h[i];
        if (tempdist > dist) { dist = tempdist; thresh = i; }
    }
    thresh--;
    if (isflipped) thresh = N - 1 - thresh;
    return thresh;
}

// ---------------------------------------------------------------- cv::RNG (core/rand.cpp)
struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s = 0xffffffff) : state(s ? s : 0xffffffff) {}
    static uint64_t next_state(uint64_t x) { return (uint64_t)(uint32_t)x * 4164903690U + (x >> 32); }
    uint32_t next() { state = next_state(state); return (uint32_t)state; }
    // randn_0_1_32f: Marsaglia-Tsang ziggurat, tables generated at first use exactly as OpenCV does.
    float randn() {
        static uint32_t kn[128]; static float wn[128], fn[128]; static bool init = false;
        const float r = 3.442620f, rng_flt = 2.3283064365386962890625e-10f;
        if (!init) {
            const double m1 = 2147483648.0;
            double dn = 3.442619855899, tn = dn, vn = 9.91256303526217e-3;
            double q = vn / std::exp(-.5 * dn * dn);
            kn[0] = (uint32_t)((dn / q) * m1); kn[1] = 0;
            wn[0] = (float)(q / m1); wn[127] = (float)(dn / m1);
            fn[0] = 1.f; fn[127] = (float)std::exp(-.5 * dn * dn);
            for (int i = 126; i >= 1; i--) {
                dn = std::sqrt(-2. * std::log(vn / dn + std::exp(-.5 * dn * dn)));
                kn[i + 1] = (uint32_t)((dn / tn) * m1);
                tn = dn;
                fn[i] = (float)std::exp(-.5 * dn * dn);
                wn[i] = (float)(dn / m1);
            }
            init = true;
        }
        uint64_t temp = state; float x, y;
        for (;;) {
            int hz = (int)temp;
            temp = next_state(temp);
            int iz = hz & 127;
            x = hz * wn[iz];
            if ((uint32_t)std::abs(hz) < kn[iz]) break;
            if (iz == 0) {
                do {
                    x = (uint32_t)temp * rng_flt; temp = next_state(temp);
                    y = (uint32_t)temp * rng_flt; temp = next_state(temp);
                    x = (float)(-std::log(x + FLT_MIN) * 0.2904764);
                    y = (float)-std::log(y + FLT_MIN);
                } while (y + y < x * x);
                x = hz > 0 ? r + x : -r - x;
                break;
            }
            y = (uint32_t)temp * rng_flt; temp = next_state(temp);
            if (fn[iz] + y * (fn[iz - 1] - fn[iz]) < (float)std::exp(-.5 * x * x)) break;
        }
        state = temp;
        return x;
    }
    double gaussian(double sigma) { return randn() * sigma; }
};

// ---------------------------------------------------------------- cv::fastAtan2 (core/mathfuncs_core, degrees)
inline float fastAtan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

}  // namespace cvx
