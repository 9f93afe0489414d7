// ORACLE (test infrastructure only; never linked or imported by the product path).
// CPU restatement of ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
// (reference ORB_SLAM2/src/ORBmatcher.cc:1328-1470) with its helpers Frame::GetFeaturesInArea (src/Frame.cc:398-451),
// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1665) and ComputeThreeMaxima (:1601-1642).  The Frame / MapPoint
// object graph is flattened to arrays (see MatchInput).  The two cv::Mat products the function evaluates are restated from
// OpenCV 4.2.0's gemm: A(3x3)*b(3x1)+c without transposition takes the small-matrix path (FP32 row dot product, then
// (float)(t*alpha + c*beta) in FP64); -A^T*b takes the generic path (FP64 accumulation).  Parity UNPINNED (no OpenCV here).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace cvx {

struct MatchInput {
    // calibration / grid of the current frame
    float fx, fy, cx, cy, bf, mb, bounds[4];          // mnMinX, mnMaxX, mnMinY, mnMaxY
    const float* scaleFactors; int nlevels;           // CurrentFrame.mvScaleFactors
    float TcwCur[12], TcwLast[12];                    // rows 0..2 of the 4x4 poses (row-major [R | t])
    // last frame: N points
    int nLast; const float* x3Dw;                     // pMP->GetWorldPos() [nLast][3]
    const uint8_t* lastValid;                         // pMP != NULL && !mvbOutlier[i]
    const uint8_t* lastHasObs;                        // pMP->Observations() > 0
    const int* lastOctave; const float* lastAngle;    // LastFrame.mvKeys[i].octave, LastFrame.mvKeysUn[i].angle
    const uint8_t* lastDesc;                          // pMP->GetDescriptor() [nLast][32]
    // current frame: M keypoints
    int nCur; const float* curUnXY; const int* curOctave; const float* curAngle; const float* curURight; const uint8_t* curDesc;
    const int* gridStart; const int* gridIdx;         // mGrid as CSR over cell = x * 48 + y
    const uint8_t* curTaken;                          // CurrentFrame.mvpMapPoints[i2] && Observations() > 0 on entry (may be NULL)
    float th; bool mono, checkOrientation;
};

inline int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y; std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
        unsigned v = x ^ y; v = v - ((v >> 1) & 0x55555555); v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// A*b + c, small-matrix path of cv::gemm (T = [A | c] rows)
inline void mat_vec_add(const float* T, const float* b, float* d) {
    for (int r = 0; r < 3; r++) { const float t = T[4 * r] * b[0] + T[4 * r + 1] * b[1] + T[4 * r + 2] * b[2]; d[r] = (float)((double)t * 1.0 + (double)T[4 * r + 3] * 1.0); }
}
// -A^T * t (generic path, FP64 accumulators, alpha = -1)
inline void neg_rt_t(const float* T, float* d) {
    for (int r = 0; r < 3; r++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)T[4 * k + r] * (double)T[4 * k + 3]; d[r] = (float)(s * -1.0); }
}

inline void features_in_area(const MatchInput& in, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) {
    out.clear();
    const float wInv = 64.f / (float)(in.bounds[1] - in.bounds[0]), hInv = 48.f / (float)(in.bounds[3] - in.bounds[2]);
    const int x0 = std::max(0, (int)std::floor((x - in.bounds[0] - r) * wInv)); if (x0 >= 64) return;
    const int x1 = std::min(63, (int)std::ceil((x - in.bounds[0] + r) * wInv)); if (x1 < 0) return;
    const int y0 = std::max(0, (int)std::floor((y - in.bounds[2] - r) * hInv)); if (y0 >= 48) return;
    const int y1 = std::min(47, (int)std::ceil((y - in.bounds[2] + r) * hInv)); if (y1 < 0) return;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = x0; ix <= x1; ix++) for (int iy = y0; iy <= y1; iy++) {
        const int c = ix * 48 + iy;
        for (int j = in.gridStart[c]; j < in.gridStart[c + 1]; j++) {
            const int k = in.gridIdx[j];
            if (checkLevels) { if (in.curOctave[k] < minLevel) continue; if (maxLevel >= 0 && in.curOctave[k] > maxLevel) continue; }
            const float dx = in.curUnXY[2 * k] - x, dy = in.curUnXY[2 * k + 1] - y;
            if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(k);
        }
    }
}

// matchOfCur[nCur]: index of the last-frame point whose MapPoint ends up in CurrentFrame.mvpMapPoints[i2], or -1.  Returns nmatches.
inline int search_by_projection(const MatchInput& in, int* matchOfCur) {
    const int HISTO_LENGTH = 30, TH_HIGH = 100;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> taken(in.nCur, 0);
    for (int k = 0; k < in.nCur; k++) { matchOfCur[k] = -1; if (in.curTaken && in.curTaken[k]) taken[k] = 1; }
    float twc[3], tlc[3]; neg_rt_t(in.TcwCur, twc); mat_vec_add(in.TcwLast, twc, tlc);
    const bool bForward = tlc[2] > in.mb && !in.mono, bBackward = -tlc[2] > in.mb && !in.mono;
    std::vector<int> idx;
    for (int i = 0; i < in.nLast; i++) {
        if (!in.lastValid[i]) continue;
        float x3Dc[3]; mat_vec_add(in.TcwCur, in.x3Dw + 3 * i, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1], invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        const float u = in.fx * xc * invzc + in.cx, v = in.fy * yc * invzc + in.cy;
        if (u < in.bounds[0] || u > in.bounds[1]) continue;
        if (v < in.bounds[2] || v > in.bounds[3]) continue;
        const int oct = in.lastOctave[i];
        const float radius = in.th * in.scaleFactors[oct];
        if (bForward) features_in_area(in, u, v, radius, oct, -1, idx);
        else if (bBackward) features_in_area(in, u, v, radius, 0, oct, idx);
        else features_in_area(in, u, v, radius, oct - 1, oct + 1, idx);
        if (idx.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : idx) {
            if (taken[i2]) continue;                                        // mvpMapPoints[i2] && Observations() > 0
            if (in.curURight[i2] > 0) { const float ur = u - in.bf * invzc; const float er = std::fabs(ur - in.curURight[i2]); if (er > radius) continue; }
            const int dist = descriptor_distance(in.lastDesc + 32 * i, in.curDesc + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            matchOfCur[bestIdx2] = i; taken[bestIdx2] = in.lastHasObs[i]; nmatches++;
            if (in.checkOrientation) {
                float rot = in.lastAngle[i] - in.curAngle[bestIdx2]; if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor); if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (in.checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; } else if (max3 < 0.1f * (float)max1) ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) if (i != ind1 && i != ind2 && i != ind3)
            for (int k : rotHist[i]) { matchOfCur[k] = -1; nmatches--; }
    }
    return nmatches;
}

}  // namespace cvx
