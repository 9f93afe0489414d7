// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the two OpenCV 4.2.0 dense-flow classes DynaDetect calls:
//   cv::optflow::createOptFlow_DeepFlow()->calc        (reference DynaDetect.cc:1031, 1075, 1127)
//   cv::VariationalRefinement::create()->calc          (reference DynaDetect.cc:1133-1143)
// Algorithm sources restated: opencv_contrib 4.2.0 modules/optflow/src/deepflow.cpp and
// opencv 4.2.0 modules/video/src/variational_refinement.cpp (neither is vendored in /root/reference).
//
// OpenCV stores every buffer in a split red/black checkerboard layout; this restatement uses a
// plain row-major layout and reproduces the same arithmetic per pixel, including the ORDER in
// which the four smoothness contributions are added into A11/A22/b1/b2 (which depends on the
// pixel colour because OpenCV runs hor-red, hor-black, vert-red, vert-black passes).
#pragma once
#include "cvx_core.hpp"

namespace cvx {

struct VarRefParams {
    int fixedPointIterations = 5, sorIterations = 5;
    float alpha = 20.0f, delta = 5.0f, gamma = 10.0f, omega = 1.6f;
    float zeta = 0.1f, epsilon = 0.001f;
};

// cv::remap(INTER_LINEAR, BORDER_REPLICATE) with CV_32FC1 maps: coordinates are quantised to 1/32 px
// (INTER_BITS = 5) and the four taps are weighted with the float table (1-fy)(1-fx) ... fy*fx.
inline float remap_bilinear_replicate(const ImgF& src, float mx, float my) {
    int sx = cvRoundf(mx * 32.f), sy = cvRoundf(my * 32.f);
    int fx = sx & 31, fy = sy & 31;
    sx >>= 5; sy >>= 5;
    sx = std::max(-32768, std::min(32767, sx)); sy = std::max(-32768, std::min(32767, sy));
    float tx1 = fx * (1.f / 32), tx0 = 1.f - tx1, ty1 = fy * (1.f / 32), ty0 = 1.f - ty1;
    float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    int x0 = clipi(sx, 0, src.w), x1 = clipi(sx + 1, 0, src.w);
    int y0 = clipi(sy, 0, src.h), y1 = clipi(sy + 1, 0, src.h);
    float v0 = src.at(y0, x0), v1 = src.at(y0, x1), v2 = src.at(y1, x0), v3 = src.at(y1, x1);
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

struct VarRefBuffers {  // exposed so tests can compare GPU intermediates stage by stage
    ImgF warped, avg, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz;
    ImgF A11, A12, A22, b1, b2, wgt, dWu, dWv, tWu, tWv;
};

// Sobel(ksize=1, BORDER_REPLICATE): x[+1]-x[-1] along one axis.
inline void deriv_x(const ImgF& s, ImgF& d) {
    d.create(s.w, s.h);
    for (int y = 0; y < s.h; y++) for (int x = 0; x < s.w; x++)
        d.at(y, x) = s.at(y, std::min(x + 1, s.w - 1)) - s.at(y, std::max(x - 1, 0));
}
inline void deriv_y(const ImgF& s, ImgF& d) {
    d.create(s.w, s.h);
    for (int y = 0; y < s.h; y++) for (int x = 0; x < s.w; x++)
        d.at(y, x) = s.at(std::min(y + 1, s.h - 1), x) - s.at(std::max(y - 1, 0), x);
}

// VariationalRefinementImpl::prepareBuffers: warp once per calc, averaged image, derivatives.
inline void varref_prepare(const ImgF& I0, const ImgF& I1, const ImgF& Wu, const ImgF& Wv, VarRefBuffers& B) {
    const int w = I0.w, h = I0.h;
    B.warped.create(w, h); B.avg.create(w, h); B.Iz.create(w, h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        float wv = remap_bilinear_replicate(I1, x + Wu.at(y, x), y + Wv.at(y, x));
        B.warped.at(y, x) = wv;
        B.avg.at(y, x) = I0.at(y, x) * 0.5f + wv * 0.5f;   // addWeighted(I0,.5,warped,.5,0)
        B.Iz.at(y, x) = wv - I0.at(y, x);                  // subtract(warped, I0)
    }
    deriv_x(B.avg, B.Ix); deriv_y(B.avg, B.Iy);
    deriv_x(B.Iz, B.Ixz); deriv_y(B.Iz, B.Iyz);
    deriv_x(B.Ix, B.Ixx); deriv_y(B.Ix, B.Ixy); deriv_y(B.Iy, B.Iyy);
}

// ComputeDataTerm_ParBody: linear system coefficients from colour and gradient constancy.
inline void varref_data_term(const VarRefParams& P, VarRefBuffers& B) {
    const int w = B.Ix.w, h = B.Ix.h;
    const float zeta2 = P.zeta * P.zeta, eps2 = P.epsilon * P.epsilon, gamma2 = P.gamma / 2, delta2 = P.delta / 2;
    B.A11.create(w, h); B.A12.create(w, h); B.A22.create(w, h); B.b1.create(w, h); B.b2.create(w, h);
    for (size_t i = 0, n = (size_t)w * h; i < n; i++) {
        const float Ix = B.Ix.d[i], Iy = B.Iy.d[i], Iz = B.Iz.d[i], Ixx = B.Ixx.d[i], Ixy = B.Ixy.d[i],
                    Iyy = B.Iyy.d[i], Ixz = B.Ixz.d[i], Iyz = B.Iyz.d[i], dU = B.dWu.d[i], dV = B.dWv.d[i];
        float derivNorm = Ix * Ix + Iy * Iy + zeta2;
        float Ik1z = Iz + Ix * dU + Iy * dV;
        float weight = (delta2 / std::sqrt(Ik1z * Ik1z / derivNorm + eps2)) / derivNorm;
        float a11 = weight * (Ix * Ix) + zeta2;
        float a12 = weight * (Ix * Iy);
        float a22 = weight * (Iy * Iy) + zeta2;
        float b1 = -weight * (Iz * Ix);
        float b2 = -weight * (Iz * Iy);
        derivNorm = Ixx * Ixx + Ixy * Ixy + zeta2;
        float derivNorm2 = Iyy * Iyy + Ixy * Ixy + zeta2;
        float Ik1zx = Ixz + Ixx * dU + Ixy * dV;
        float Ik1zy = Iyz + Ixy * dU + Iyy * dV;
        weight = gamma2 / std::sqrt(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
        a11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
        a12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
        a22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
        b1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
        b2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
        B.A11.d[i] = a11; B.A12.d[i] = a12; B.A22.d[i] = a22; B.b1.d[i] = b1; B.b2.d[i] = b2;
    }
}

// ComputeSmoothnessTermHorPass / VertPass, gathered per pixel in OpenCV's accumulation order.
inline void varref_smooth_term(const VarRefParams& P, const ImgF& Wu, const ImgF& Wv, VarRefBuffers& B) {
    const int w = Wu.w, h = Wu.h;
    const float eps2 = P.epsilon * P.epsilon, alpha2 = P.alpha / 2;
    B.wgt.create(w, h);
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
        int jn = std::min(j + 1, w - 1), in = std::min(i + 1, h - 1);   // repeated borders of tempW
        float ux = B.tWu.at(i, jn) - B.tWu.at(i, j), vx = B.tWv.at(i, jn) - B.tWv.at(i, j);
        float uy = B.tWu.at(in, j) - B.tWu.at(i, j), vy = B.tWv.at(in, j) - B.tWv.at(i, j);
        B.wgt.at(i, j) = alpha2 / std::sqrt(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
    }
    for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
        const bool red = ((i + j) & 1) == 0;
        float a11 = B.A11.at(i, j), a22 = B.A22.at(i, j), b1 = B.b1.at(i, j), b2 = B.b2.at(i, j);
        const float wp = B.wgt.at(i, j);
        auto own_h = [&]() { if (j < w - 1) {
            b1 += wp * (Wu.at(i, j + 1) - Wu.at(i, j)); a11 += wp;
            b2 += wp * (Wv.at(i, j + 1) - Wv.at(i, j)); a22 += wp; } };
        auto left_h = [&]() { if (j > 0) { float wl = B.wgt.at(i, j - 1);
            b1 -= wl * (Wu.at(i, j) - Wu.at(i, j - 1)); a11 += wl;
            b2 -= wl * (Wv.at(i, j) - Wv.at(i, j - 1)); a22 += wl; } };
        auto own_v = [&]() { if (i < h - 1) {
            b1 += wp * (Wu.at(i + 1, j) - Wu.at(i, j)); a11 += wp;
            b2 += wp * (Wv.at(i + 1, j) - Wv.at(i, j)); a22 += wp; } };
        auto up_v = [&]() { if (i > 0) { float wu = B.wgt.at(i - 1, j);
            b1 -= wu * (Wu.at(i, j) - Wu.at(i - 1, j)); a11 += wu;
            b2 -= wu * (Wv.at(i, j) - Wv.at(i - 1, j)); a22 += wu; } };
        if (red) { own_h(); left_h(); own_v(); up_v(); }
        else     { left_h(); own_h(); up_v(); own_v(); }
        B.A11.at(i, j) = a11; B.A22.at(i, j) = a22; B.b1.at(i, j) = b1; B.b2.at(i, j) = b2;
    }
}

// RedBlackSOR_ParBody: one colour sweep.  Out-of-image neighbours contribute 0 (OpenCV: zero
// weight / zero increment in the buffer borders).
inline void varref_sor_sweep(const VarRefParams& P, VarRefBuffers& B, bool red_pass) {
    const int w = B.A11.w, h = B.A11.h;
    for (int i = 0; i < h; i++)
        for (int j = ((i & 1) == (red_pass ? 0 : 1)) ? 0 : 1; j < w; j += 2) {
            const float wp = B.wgt.at(i, j);
            const float wl = j > 0 ? B.wgt.at(i, j - 1) : 0.f, wu = i > 0 ? B.wgt.at(i - 1, j) : 0.f;
            const float ul = j > 0 ? B.dWu.at(i, j - 1) : 0.f, vl = j > 0 ? B.dWv.at(i, j - 1) : 0.f;
            const float ur = j < w - 1 ? B.dWu.at(i, j + 1) : 0.f, vr = j < w - 1 ? B.dWv.at(i, j + 1) : 0.f;
            const float uu = i > 0 ? B.dWu.at(i - 1, j) : 0.f, vu = i > 0 ? B.dWv.at(i - 1, j) : 0.f;
            const float ud = i < h - 1 ? B.dWu.at(i + 1, j) : 0.f, vd = i < h - 1 ? B.dWv.at(i + 1, j) : 0.f;
            float sigmaU = wl * ul + wp * ur + wu * uu + wp * ud;
            float sigmaV = wl * vl + wp * vr + wu * vu + wp * vd;
            float du = B.dWu.at(i, j), dv = B.dWv.at(i, j);
            du += P.omega * ((sigmaU + B.b1.at(i, j) - dv * B.A12.at(i, j)) / B.A11.at(i, j) - du);
            dv += P.omega * ((sigmaV + B.b2.at(i, j) - du * B.A12.at(i, j)) / B.A22.at(i, j) - dv);
            B.dWu.at(i, j) = du; B.dWv.at(i, j) = dv;
        }
}

// VariationalRefinementImpl::calcUV.  I0/I1 are float images (u8 inputs are converted by the caller,
// as prepareBuffers does with convertTo / mixed-type arithmetic).  Wu/Wv: initial flow in, refined out.
inline void varref_calc(const VarRefParams& P, const ImgF& I0, const ImgF& I1, ImgF& Wu, ImgF& Wv,
                        VarRefBuffers* keep = nullptr) {
    VarRefBuffers local; VarRefBuffers& B = keep ? *keep : local;
    const int w = I0.w, h = I0.h;
    varref_prepare(I0, I1, Wu, Wv, B);
    B.dWu.create(w, h, 1, 0.f); B.dWv.create(w, h, 1, 0.f);
    B.tWu = Wu; B.tWv = Wv;
    for (int it = 0; it < P.fixedPointIterations; it++) {
        varref_data_term(P, B);
        varref_smooth_term(P, Wu, Wv, B);
        for (int s = 0; s < P.sorIterations; s++) { varref_sor_sweep(P, B, true); varref_sor_sweep(P, B, false); }
        for (size_t i = 0, n = (size_t)w * h; i < n; i++) { B.tWu.d[i] = Wu.d[i] + B.dWu.d[i]; B.tWv.d[i] = Wv.d[i] + B.dWv.d[i]; }
    }
    Wu = B.tWu; Wv = B.tWv;
}

// ---------------------------------------------------------------- DeepFlow (variational part only)
struct DeepFlowParams {
    float sigma = 0.6f; int minSize = 25; float downscaleFactor = 0.95f;
    int fixedPointIterations = 5, sorIterations = 25;
    float alpha = 1.0f, delta = 0.5f, gamma = 5.0f, omega = 1.6f;
    int maxLayers = 200;
    int maxLevels = 0;      // build-side option (BASELINE.json config 5, "3-level flow pyramid"): keep only the finest maxLevels levels; 0 = OpenCV behaviour
};

// OpticalFlowDeepFlow::buildPyramid sizes: (int)(prev*0.95f + 0.5f) until a side <= minSize.
inline std::vector<std::pair<int, int>> deepflow_level_sizes(int w, int h, const DeepFlowParams& P = DeepFlowParams()) {
    std::vector<std::pair<int, int>> s; s.push_back({w, h});
    for (int i = 0; i < P.maxLayers; ) {
        int nw = (int)(s.back().first * P.downscaleFactor + 0.5f), nh = (int)(s.back().second * P.downscaleFactor + 0.5f);
        if (nh <= P.minSize || nw <= P.minSize) break;
        s.push_back({nw, nh});
    }
    if (P.maxLevels > 0 && (int)s.size() > P.maxLevels) s.resize(P.maxLevels);
    return s;
}

inline void deepflow_pyramid(const ImgF& src, std::vector<ImgF>& pyr, const DeepFlowParams& P) {
    pyr.clear(); pyr.push_back(src);
    auto sizes = deepflow_level_sizes(src.w, src.h, P);
    for (size_t l = 1; l < sizes.size(); l++) {
        ImgF next; resize_linear_f32(pyr.back(), next, sizes[l].first, sizes[l].second);
        pyr.push_back(std::move(next));
    }
}

// OpticalFlowDeepFlow::calc.  flow: 2-channel (u,v) interleaved like CV_32FC2.
inline void deepflow_calc(const Img8& I0u8, const Img8& I1u8, ImgF& flow, const DeepFlowParams& P = DeepFlowParams()) {
    const int w = I0u8.w, h = I0u8.h;
    ImgF I0(w, h), I1(w, h);
    for (size_t i = 0; i < I0.d.size(); i++) { I0.d[i] = (float)I0u8.d[i]; I1.d[i] = (float)I1u8.d[i]; }
    ImgF I0s, I1s;
    gaussian_blur3_f32(I0, I0s, P.sigma);    // kernelLen = floor(3*sigma)*2+1 = 3
    gaussian_blur3_f32(I1, I1s, P.sigma);
    std::vector<ImgF> p0, p1;
    deepflow_pyramid(I0s, p0, P); deepflow_pyramid(I1s, p1, P);
    const int L = (int)p0.size();
    ImgF Wu(p0[L - 1].w, p0[L - 1].h, 1, 0.f), Wv = Wu;
    VarRefParams V;
    V.alpha = 4 * P.alpha; V.delta = P.delta / 3; V.gamma = P.gamma / 3;
    V.fixedPointIterations = P.fixedPointIterations; V.sorIterations = P.sorIterations; V.omega = P.omega;
    const float inv_scale = 1.0f / P.downscaleFactor;
    for (int level = L - 1; level >= 0; --level) {
        varref_calc(V, p0[level], p1[level], Wu, Wv);
        if (level > 0) {
            ImgF tu, tv;
            resize_linear_f32(Wu, tu, p0[level - 1].w, p0[level - 1].h);
            resize_linear_f32(Wv, tv, p0[level - 1].w, p0[level - 1].h);
            for (auto& v : tu.d) v = v * inv_scale;
            for (auto& v : tv.d) v = v * inv_scale;
            Wu = std::move(tu); Wv = std::move(tv);
        }
    }
    flow.create(w, h, 2);
    for (size_t i = 0, n = (size_t)w * h; i < n; i++) { flow.d[2*i] = Wu.d[i]; flow.d[2*i+1] = Wv.d[i]; }
}

}  // namespace cvx
