// ORACLE (test infrastructure only; never linked or imported by the product path).
// CPU restatement of the mapping consumer's per-key-frame cloud generation, reference octomap_pub/src/pubPointCloud.cc:471-660
// (generatePointCloud with imgLabel): stride-2 back-projection, re-projection depth-consistency test per cluster (:556-607),
// cluster rejection rule (:641-663) and the world transform (pcl::transformPointCloud, :665).
// Third-party pieces restated from their published behaviour: Eigen 3.3 fixed-size Matrix3d * Vector3d is the coefficient-based
// product whose unrolled reduction evaluates a0*b0 + (a1*b1 + a2*b2); pcl::transformPointCloud<PointT, double> on a non-dense
// cloud copies non-finite points and maps the others with t(r,0)*x + t(r,1)*y + t(r,2)*z + t(r,3) in FP64.  Parity UNPINNED.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace cvx {

struct CloudPoint { float x, y, z; uint8_t b, g, r, a; };                  // pcl::PointXYZRGB payload (a = 255 by construction)

struct CloudParams { double fx, fy, cx, cy, depthScale; };

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

// poseRelative: 4x4 row-major (last -> current as the reference names it), Twc: 4x4 row-major.  occlusion[12]: vecOcclusion,
// labelCount[12]: countNonZero(imgLabel == i), kept[12].  Returns the cloud in the reference's order.
inline void generate_point_cloud(const CloudParams& P, const uint8_t* bgr, const uint16_t* depth, const uint16_t* depthLast, const uint8_t* dynaMask,
                                 const uint8_t* dynaMaskLast, const uint8_t* label, int width, int height, const double* poseRelative, const double* Twc,
                                 std::vector<CloudPoint>& out, double occlusion[12], int labelCount[12], uint8_t kept[12]) {
    std::vector<CloudPoint> cluster[12];
    for (int i = 0; i < 12; i++) { occlusion[i] = 0; labelCount[i] = 0; }
    const double K[3][3] = {{P.fx, 0.0, P.cx}, {0.0, P.fy, P.cy}, {0.0, 0.0, 1.0}};
    double R[3][3], t[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r][c] = poseRelative[4 * r + c]; t[r] = poseRelative[4 * r + 3]; }
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    for (int m = 0; m < height; m += 2) for (int n = 0; n < width; n += 2) {
        const size_t px = (size_t)m * width + n;
        const float dCurrent = (float)(depth[px] * (1.0 / P.depthScale));
        const int iLabel = label[px];
        if (iLabel >= 12) continue;
        const double p_xyz[3] = {(double)((n - (float)P.cx) * (float)dCurrent / (float)P.fx), (double)((m - (float)P.cy) * dCurrent / (float)P.fy), (double)dCurrent};
        double q[3]; for (int r = 0; r < 3; r++) q[r] = dot3(R[r], p_xyz) + t[r];
        double pt[3]; for (int r = 0; r < 3; r++) pt[r] = dot3(K[r], q);
        const double z = pt[2]; pt[0] /= z; pt[1] /= z;
        const float x_translate = (float)pt[0], y_translate = (float)pt[1];
        float dLast = 0.0f; bool isDynaLast = false;
        if (y_translate >= 0.0f && y_translate < height && x_translate >= 0.0f && x_translate < width) {
            const size_t pl = (size_t)(int)y_translate * width + (int)x_translate;
            dLast = (float)(depthLast[pl] * (1.0 / P.depthScale));
            isDynaLast = dynaMaskLast[pl] > 240;
        }
        if (dCurrent >= 0 && dCurrent < 10 && dLast >= 0 && dLast < 10) {
            const float diff = dCurrent - dLast;
            if ((diff * diff) > (0.13 * dCurrent) * (0.13 * dCurrent) || isDynaLast) occlusion[iLabel]++;
        }
        CloudPoint p; p.a = 255;
        if ((int)dynaMask[px] >= 240) p.x = p.y = p.z = qnan;
        else if (dCurrent < 0.01 || dCurrent > 10) p.x = p.y = p.z = qnan;
        else { p.z = dCurrent; p.x = (float)((n - P.cx) * p.z / P.fx); p.y = (float)((m - P.cy) * p.z / P.fy); }
        p.b = bgr[px * 3]; p.g = bgr[px * 3 + 1]; p.r = bgr[px * 3 + 2];
        cluster[iLabel].push_back(p);
    }
    for (size_t i = 0; i < (size_t)width * height; i++) if (label[i] < 12) labelCount[label[i]]++;
    std::vector<CloudPoint> tmp;
    for (int i = 0; i < 12; i++) {
        kept[i] = (i == 0) || (occlusion[i] * 9 <= 0.4 * labelCount[i]);
        if (kept[i]) tmp.insert(tmp.end(), cluster[i].begin(), cluster[i].end());
    }
    out = tmp;
    for (CloudPoint& p : out) {
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        const double x = p.x, y = p.y, z = p.z;
        p.x = (float)(Twc[0] * x + Twc[1] * y + Twc[2] * z + Twc[3]);
        p.y = (float)(Twc[4] * x + Twc[5] * y + Twc[6] * z + Twc[7]);
        p.z = (float)(Twc[8] * x + Twc[9] * y + Twc[10] * z + Twc[11]);
    }
}

}  // namespace cvx
