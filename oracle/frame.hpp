// ORACLE (test infrastructure only; never linked or imported by the product path).
// CPU restatement of the per-keypoint steps the reference's RGB-D Frame constructor runs right after ORB extraction
// (reference ORB_SLAM2/src/Frame.cc:136-170): UndistortKeyPoints (Frame.cc:477-509), ComputeStereoFromRGBD (Frame.cc:714-735),
// ComputeImageBounds (Frame.cc:511-541), PosInGrid (Frame.cc:453-463) and AssignFeaturesToGrid (Frame.cc:283-299).
// cv::undistortPoints(src, dst, K, D, noArray(), K) is restated from the published algorithm of OpenCV 4.2.0
// (cvUndistortPointsInternal: FP64, 5 fixed-point iterations, criteria MAX_ITER only).  Parity UNPINNED (no OpenCV here).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace cvx {

struct FrameCalib {
    float fx, fy, cx, cy;            // mK (CV_32F)
    float k1, k2, p1, p2, k3;        // mDistCoef (CV_32F, 4 or 5 entries; k3 = 0 when absent)
    float bf;                        // Camera.bf
    float depthMapFactor;            // 1 / DepthMapFactor (Tracking.cc:~150), applied by imDepth.convertTo(CV_32F, factor)
};

static const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;

// cv::undistortPoints with R = I and P = K on one point (float in, float out)
inline void undistort_point(const FrameCalib& c, float xin, float yin, float& xo, float& yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k[12] = {c.k1, c.k2, c.p1, c.p2, c.k3, 0, 0, 0, 0, 0, 0, 0};
    double x = xin, y = yin; const double u = x, v = y;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    xo = (float)(xx * ww); yo = (float)(yy * ww);
}

// ComputeImageBounds -> {mnMinX, mnMaxX, mnMinY, mnMaxY}
inline void frame_bounds(const FrameCalib& c, int W, int H, float b[4]) {
    if (c.k1 != 0.0f) {
        float m[4][2]; const float src[4][2] = {{0.f, 0.f}, {(float)W, 0.f}, {0.f, (float)H}, {(float)W, (float)H}};
        for (int i = 0; i < 4; i++) undistort_point(c, src[i][0], src[i][1], m[i][0], m[i][1]);
        b[0] = std::min(m[0][0], m[2][0]); b[1] = std::max(m[1][0], m[3][0]); b[2] = std::min(m[0][1], m[1][1]); b[3] = std::max(m[2][1], m[3][1]);
    } else { b[0] = 0.f; b[1] = (float)W; b[2] = 0.f; b[3] = (float)H; }
}

struct FramePost {
    std::vector<float> unx, uny, uRight, depth;      // mvKeysUn[i].pt, mvuRight, mvDepth
    std::vector<int> cell;                           // x * FRAME_GRID_ROWS + y, or -1 when PosInGrid fails
    std::vector<int> gridStart, gridIdx;             // mGrid as CSR over cell = x * 48 + y, indices in push_back order
};

inline void frame_post_orb(const FrameCalib& c, const float* kx, const float* ky, int N, const uint16_t* depth, int W, int H, FramePost& o) {
    o.unx.resize(N); o.uny.resize(N); o.uRight.assign(N, -1.f); o.depth.assign(N, -1.f); o.cell.assign(N, -1);
    for (int i = 0; i < N; i++) {                    // UndistortKeyPoints
        if (c.k1 == 0.0f) { o.unx[i] = kx[i]; o.uny[i] = ky[i]; }
        else undistort_point(c, kx[i], ky[i], o.unx[i], o.uny[i]);
    }
    for (int i = 0; i < N; i++) {                    // ComputeStereoFromRGBD: depth at the DISTORTED position, u_R from the undistorted x
        const int v = (int)ky[i], u = (int)kx[i];
        const float d = (float)depth[(size_t)v * W + u] * c.depthMapFactor;
        if (d > 0) { o.depth[i] = d; o.uRight[i] = o.unx[i] - c.bf / d; }
    }
    float b[4]; frame_bounds(c, W, H, b);
    const float wInv = (float)FRAME_GRID_COLS / (float)(b[1] - b[0]), hInv = (float)FRAME_GRID_ROWS / (float)(b[3] - b[2]);
    std::vector<std::vector<int>> grid(FRAME_GRID_COLS * FRAME_GRID_ROWS);
    for (int i = 0; i < N; i++) {                    // AssignFeaturesToGrid / PosInGrid
        const int px = (int)std::round((o.unx[i] - b[0]) * wInv), py = (int)std::round((o.uny[i] - b[2]) * hInv);
        if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) continue;
        o.cell[i] = px * FRAME_GRID_ROWS + py; grid[o.cell[i]].push_back(i);
    }
    o.gridStart.assign(grid.size() + 1, 0); o.gridIdx.clear();
    for (size_t g = 0; g < grid.size(); g++) { o.gridStart[g] = (int)o.gridIdx.size(); o.gridIdx.insert(o.gridIdx.end(), grid[g].begin(), grid[g].end()); }
    o.gridStart[grid.size()] = (int)o.gridIdx.size();
}

}  // namespace cvx
