// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// CPU restatement of ORB_SLAM2::ORBextractor as modified by SInDSLAM
// (reference ORB_SLAM2/src/ORBextractor.cc, include/ORBextractor.h) plus the OpenCV 4.2.0
// primitives it calls: cv::FAST (features2d/fast.cpp, fast_score.cpp), cv::resize, copyMakeBorder,
// cv::GaussianBlur (8U fixed point), cv::fastAtan2.
#pragma once
#include <list>
#include "cvx_core.hpp"
#include "../include/sind_brief_pattern.h"

namespace cvx {

struct KeyPoint {
    float x = 0, y = 0, size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

// ---------------------------------------------------------------- cv::FAST, TYPE_9_16, on a sub-image view
static const int FAST_RING[16][2] = {{0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},
                                     {0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3}};

// cornerScore<16> (fast_score.cpp): largest threshold for which the pixel is still a 9/16 corner.
inline int fast_corner_score(const uint8_t* ptr, int stride, int threshold) {
    const int N = 25; int d[N]; int v = ptr[0];
    for (int k = 0; k < N; k++) { const int* o = FAST_RING[k & 15]; d[k] = v - ptr[o[1] * stride + o[0]]; }
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k+1], d[k+2]); a = std::min(a, d[k+3]);
        if (a <= a0) continue;
        a = std::min(a, d[k+4]); a = std::min(a, d[k+5]); a = std::min(a, d[k+6]); a = std::min(a, d[k+7]); a = std::min(a, d[k+8]);
        a0 = std::max(a0, std::min(a, d[k])); a0 = std::max(a0, std::min(a, d[k+9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k+1], d[k+2]); b = std::max(b, d[k+3]); b = std::max(b, d[k+4]); b = std::max(b, d[k+5]);
        if (b >= b0) continue;
        b = std::max(b, d[k+6]); b = std::max(b, d[k+7]); b = std::max(b, d[k+8]);
        b0 = std::min(b0, std::max(b, d[k])); b0 = std::min(b0, std::max(b, d[k+9]));
    }
    return -b0 - 1;
}

inline bool fast_is_corner(const uint8_t* ptr, int stride, int t) {
    int v = ptr[0]; uint32_t br = 0, dk = 0;
    for (int k = 0; k < 16; k++) {
        int p = ptr[FAST_RING[k][1] * stride + FAST_RING[k][0]];
        if (p > v + t) br |= 1u << k;
        if (p < v - t) dk |= 1u << k;
    }
    auto run9 = [](uint32_t m) { m |= m << 16; uint32_t r = m; for (int i = 1; i < 9; i++) r &= m >> i; return (r & 0xffff) != 0; };
    return run9(br) || run9(dk);
}

// FAST_t<16> with nonmax suppression on the view (x0,y0,vw,vh) of img; keypoints row-major,
// coordinates relative to the view; 3-px view border is never a corner; NMS = strictly greater than
// all 8 neighbour scores (non-corners score 0).
inline void fast_view(const Img8& img, int x0, int y0, int vw, int vh, int threshold, std::vector<KeyPoint>& out) {
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (vw < 7 || vh < 7) return;
    std::vector<int> score((size_t)vw * vh, 0);
    const int stride = img.w;
    for (int y = 3; y < vh - 3; y++)
        for (int x = 3; x < vw - 3; x++) {
            const uint8_t* p = &img.d[(size_t)(y0 + y) * stride + x0 + x];
            if (fast_is_corner(p, stride, threshold)) score[(size_t)y * vw + x] = fast_corner_score(p, stride, threshold);
        }
    for (int y = 3; y < vh - 3; y++)
        for (int x = 3; x < vw - 3; x++) {
            int s = score[(size_t)y * vw + x];
            if (!s) continue;   // a corner's score is >= threshold; threshold 0 corners with score 0 cannot beat their neighbours
            const int* r0 = &score[(size_t)(y - 1) * vw + x]; const int* r1 = r0 + vw; const int* r2 = r1 + vw;
            if (s > r0[-1] && s > r0[0] && s > r0[1] && s > r1[-1] && s > r1[1] && s > r2[-1] && s > r2[0] && s > r2[1]) {
                KeyPoint k; k.x = (float)x; k.y = (float)y; k.size = 7.f; k.angle = -1; k.response = (float)s;
                out.push_back(k);
            }
        }
}

// ---------------------------------------------------------------- ORBextractor
struct ExtractorNode {
    std::vector<KeyPoint> vKeys;
    Pt UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0;   // creation order; stands in for the node address in the reference's (size, pointer) sort
    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) const {
        // reference ORBextractor.cc:481-537
        const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
        n1.UL = UL; n1.UR = {UL.x + halfX, UL.y}; n1.BL = {UL.x, UL.y + halfY}; n1.BR = {UL.x + halfX, UL.y + halfY};
        n2.UL = n1.UR; n2.UR = UR; n2.BL = n1.BR; n2.BR = {UR.x, UL.y + halfY};
        n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = BL; n3.BR = {n1.BR.x, BL.y};
        n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = BR;
        for (const KeyPoint& kp : vKeys) {
            if (kp.x < n1.UR.x) { if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp); }
            else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
            else n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};

struct ORBextractor {
    static const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<Img8> pyr_padded;               // level image with the 19-px border (mvImagePyramid's parent)
    std::vector<std::pair<int, int>> level_size;
    // exposed intermediates for stage-by-stage parity tests
    std::vector<std::vector<KeyPoint>> dbg_fast;      // per level: all cell-wise FAST keypoints (vToDistributeKeys)
    std::vector<std::vector<KeyPoint>> dbg_selected;  // per level: after the octree + orientation, level coords
    int dbg_fallback = 0;

    // reference ORBextractor.cc:410-470
    ORBextractor(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniThFAST(ini), minThFAST(mn) {
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRound(nDesired);
            sum += mnFeaturesPerLevel[level];
            nDesired *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0; ++v0;
        }
    }

    // reference ORBextractor.cc:1166-1191
    void ComputePyramid(const Img8& image) {
        pyr_padded.resize(nlevels); level_size.resize(nlevels);
        Img8 prev;
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            int sw = cvRound((float)image.w * scale), sh = cvRound((float)image.h * scale);
            level_size[level] = {sw, sh};
            Img8 cur;
            if (level != 0) resize_linear_u8(prev, cur, sw, sh); else cur = image;
            pad_reflect101_u8(cur, pyr_padded[level], EDGE_THRESHOLD);
            prev = std::move(cur);
        }
    }

    // reference ORBextractor.cc:77-104
    float IC_Angle(const Img8& padded, float px, float py) const {
        int m_01 = 0, m_10 = 0;
        const int step = padded.w;
        const uint8_t* center = &padded.d[(size_t)(cvRoundf(py) + EDGE_THRESHOLD) * step + cvRoundf(px) + EDGE_THRESHOLD];
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0, d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2((float)m_01, (float)m_10);
    }

    // reference ORBextractor.cc:539-763
    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX, int minY, int maxY, int N) {
        long seq = 0;
        const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
        const float hX = static_cast<float>(maxX - minX) / nIni;
        std::list<ExtractorNode> lNodes;
        std::vector<ExtractorNode*> vpIniNodes(nIni);
        for (int i = 0; i < nIni; i++) {
            ExtractorNode ni;
            ni.UL = {(int)(hX * static_cast<float>(i)), 0};
            ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
            ni.BL = {ni.UL.x, maxY - minY};
            ni.BR = {ni.UR.x, maxY - minY};
            ni.seq = seq++;
            lNodes.push_back(ni);
            vpIniNodes[i] = &lNodes.back();
        }
        for (const KeyPoint& kp : vToDistributeKeys) vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
        auto lit = lNodes.begin();
        while (lit != lNodes.end()) {
            if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
            else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
            else lit++;
        }
        bool bFinish = false;
        typedef std::pair<int, ExtractorNode*> SP;
        auto sp_less = [](const SP& a, const SP& b) { return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq; };
        std::vector<SP> vSizeAndPointerToNode;
        auto add_child = [&](ExtractorNode& n, int& nToExpand) {
            if (n.vKeys.size() > 0) {
                n.seq = seq++;
                lNodes.push_front(n);
                if (n.vKeys.size() > 1) {
                    nToExpand++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        };
        while (!bFinish) {
            int prevSize = (int)lNodes.size();
            lit = lNodes.begin();
            int nToExpand = 0;
            vSizeAndPointerToNode.clear();
            while (lit != lNodes.end()) {
                if (lit->bNoMore) { lit++; continue; }
                ExtractorNode n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                add_child(n1, nToExpand); add_child(n2, nToExpand); add_child(n3, nToExpand); add_child(n4, nToExpand);
                lit = lNodes.erase(lit);
            }
            if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            else if (((int)lNodes.size() + nToExpand * 3) > N) {
                while (!bFinish) {
                    prevSize = (int)lNodes.size();
                    std::vector<SP> vPrev = vSizeAndPointerToNode;
                    vSizeAndPointerToNode.clear();
                    std::sort(vPrev.begin(), vPrev.end(), sp_less);
                    for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                        ExtractorNode n1, n2, n3, n4; int dummy = 0;
                        vPrev[j].second->DivideNode(n1, n2, n3, n4);
                        add_child(n1, dummy); add_child(n2, dummy); add_child(n3, dummy); add_child(n4, dummy);
                        lNodes.erase(vPrev[j].second->lit);
                        if ((int)lNodes.size() >= N) break;
                    }
                    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
                }
            }
        }
        std::vector<KeyPoint> vResultKeys;
        for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
            const std::vector<KeyPoint>& vNodeKeys = it->vKeys;
            const KeyPoint* pKP = &vNodeKeys[0];
            float maxResponse = pKP->response;
            for (size_t k = 1; k < vNodeKeys.size(); k++)
                if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
            vResultKeys.push_back(*pKP);
        }
        return vResultKeys;
    }

    // reference ORBextractor.cc:765-853
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
        allKeypoints.resize(nlevels); dbg_fast.assign(nlevels, {});
        const float W = 30;
        for (int level = 0; level < nlevels; ++level) {
            const int lw = level_size[level].first, lh = level_size[level].second;
            const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
            const int maxBorderX = lw - EDGE_THRESHOLD + 3, maxBorderY = lh - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint> vToDistributeKeys;
            const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W), nRows = (int)(height / W);
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    std::vector<KeyPoint> vKeysCell;
                    const int vx = (int)iniX + EDGE_THRESHOLD, vy = (int)iniY + EDGE_THRESHOLD;
                    const int vw = (int)maxX - (int)iniX, vh = (int)maxY - (int)iniY;
                    fast_view(pyr_padded[level], vx, vy, vw, vh, iniThFAST, vKeysCell);
                    if (vKeysCell.empty()) fast_view(pyr_padded[level], vx, vy, vw, vh, minThFAST, vKeysCell);
                    for (KeyPoint& k : vKeysCell) { k.x += j * wCell; k.y += i * hCell; vToDistributeKeys.push_back(k); }
                }
            }
            dbg_fast[level] = vToDistributeKeys;
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (KeyPoint& k : keypoints) { k.x += minBorderX; k.y += minBorderY; k.octave = level; k.size = (float)scaledPatchSize; }
        }
        for (int level = 0; level < nlevels; ++level)
            for (KeyPoint& k : allKeypoints[level]) k.angle = IC_Angle(pyr_padded[level], k.x, k.y);
    }

    // reference ORBextractor.cc:108-147
    static void computeOrbDescriptor(const KeyPoint& kpt, const Img8& img, uint8_t* desc) {
        const float factorPI = (float)(M_PI / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
        const int step = img.w;
        const uint8_t* center = &img.d[(size_t)cvRoundf(kpt.y) * step + cvRoundf(kpt.x)];
        const signed char* pattern = SIND_BRIEF_PATTERN;
        auto val = [&](int idx) {
            float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
            return (int)center[cvRoundf(px * b + py * a) * step + cvRoundf(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int v = 0;
            for (int k = 0; k < 8; k++) v |= (val(2 * k) < val(2 * k + 1)) << k;
            desc[i] = (uint8_t)v;
        }
    }

    // reference ORBextractor.cc:1043-1164.  mask may be empty.  desc: n x 32.
    void extract(const Img8& image, const Img8& mask, std::vector<KeyPoint>& keypoints, std::vector<uint8_t>& desc) {
        keypoints.clear(); desc.clear(); dbg_fallback = 0;
        if (image.empty()) return;
        ComputePyramid(image);
        std::vector<std::vector<KeyPoint>> all;
        ComputeKeyPointsOctTree(all);
        dbg_selected = all;
        std::vector<std::vector<KeyPoint>> all_copy = all;
        if (!mask.empty()) {
            for (auto& lv : all)
                for (auto it = lv.begin(); it != lv.end();) {
                    float scale = (float)std::pow(scaleFactor, it->octave);
                    bool dyn = mask.at((int)(it->y * scale), (int)(it->x * scale)) == 255;
                    if (dyn) it = lv.erase(it); else ++it;
                }
        }
        int nk = 0; for (auto& lv : all) nk += (int)lv.size();
        if (nk < 250) { all = all_copy; dbg_fallback = 1; nk = 0; for (auto& lv : all) nk += (int)lv.size(); }
        desc.assign((size_t)nk * 32, 0);
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& kps = all[level];
            if (kps.empty()) continue;
            const int lw = level_size[level].first, lh = level_size[level].second;
            Img8 working(lw, lh), blurred;
            for (int y = 0; y < lh; y++) std::memcpy(working.row(y), pyr_padded[level].row(y + EDGE_THRESHOLD) + EDGE_THRESHOLD, lw);
            gaussian_blur_u8(working, blurred, 7, 2.0);
            for (size_t i = 0; i < kps.size(); i++) computeOrbDescriptor(kps[i], blurred, &desc[(size_t)(offset + i) * 32]);
            offset += (int)kps.size();
            if (level != 0) { float scale = mvScaleFactor[level]; for (KeyPoint& k : kps) { k.x *= scale; k.y *= scale; } }
            keypoints.insert(keypoints.end(), kps.begin(), kps.end());
        }
    }
};

}  // namespace cvx
