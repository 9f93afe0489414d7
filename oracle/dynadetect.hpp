// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// CPU restatement of ORB_SLAM2::DynaDetect (reference ORB_SLAM2/src/DynaDetect.cc, include/DynaDetect.h)
// and of the caller-side 15x15 dilation (Examples/RGB-D/rgbd_tum_noros.cc:108,138).  Function-by-function
// citations are on each member.  GUI calls, stdout timing and IMGSAVE dumps are dropped.
// Reference quirks are reproduced, except the two that are undefined behaviour there:
//   C-14 applyNMS sorts end-points by an uninitialised float -> here the scan order is kept (stable);
//   a-19 octree ties on node addresses                         -> creation order (see orb.hpp).
#pragma once
#include <chrono>
#include "cvx_core.hpp"
#include "cvx_shape.hpp"
#include "homography.hpp"
#include "peac.hpp"
#include "varflow.hpp"

namespace cvx {

struct DynaIntermediates {   // everything a stage-by-stage parity test wants to look at
    Img8 gray, grayLast, grayLastLast, grayMin, grayLastMin, grayLastLastMin;
    ImgF flowDeep;        // 384x288x2, already negated (DD:1080), before refinement, for the pair finally used
    ImgF flowRefined;     // 384x288x2 after VariationalRefinement
    ImgF flowFull;        // 640x480x2 after resize and *1/0.6
    bool largeMotion = false;
    int endFlow = 0, endFlow2 = 0;
    double H[9] = {0};
    int nPairs = 0;
    float maxError = 0, otsu = 0, triangle = 0, thr_low = 0, thr_high = 0;
    int hist[256] = {0};
    Img8 magU8, maskLow, maskHigh;     // maskLow: 0/128, maskHigh: 0/255 (stImgMasks)
    Img8 kmeansLabel;                  // after SegByKmeans, 0..11
    std::vector<float> centers;        // 12 x 3
    Img8 labelForSegEdge, totalArea, occluded1, occluded2, gradEdge, planeContours;
    int nClusters = 0;
    Img8 label, dyna;                  // outputs
};

class DynaDetect {
public:
    static const int numCluster = 12, nRowCluster = 3, nColCluster = 4;
    const float depth_weight = 1.5f;
    int width, height;
    float fx, fy, cx, cy, depthScale;
    Img8 imgRGB, imgRGBLast, imgRGBLastLast; Img16 imgDepth;
    Img8 imgGray, imgGrayLast, imgGrayLastLast;
    Img8 imgDyna, imgDynaLast, imgLabel, imgLabelLast, imgMaskHighErrorLast;
    std::vector<Pt> aroundPoint;
    StructElem element3, element4, element5, element7, element9, element10;
    peac::PlaneFitter pf;
    DynaIntermediates dbg;
    int flow_max_levels = 0;  // build-side option of BASELINE.json config 5 ("3-level flow pyramid"); 0 = the reference's full pyramid
    // a-8 (cv::findHomography(..., RHO), DD:1235): 0 = RHO's published scheme (PROSAC + SPRT + LM, homography.hpp; the product implements the same
    // scheme), 2 = round 1's lighter PROSAC + least-squares estimator, kept for the sensitivity comparison of tests/test_a8_sensitivity_cpu.py;
    // h_seed != 0 replaces the estimator's fixed PRNG seed (another draw order)
    int h_estimator = 0; uint64_t h_seed = 0;
    bool skip_flow = false;   // test hook: reuse dbg.flowFull supplied by the caller instead of computing it
    // wall seconds per stage in the reference's own breakdown (its stdout timers: "K-means timecost" DD:1421, "Calculate DepthEdge" DD:1499,
    // "SegAndMergeV2 timecost" DD:1518, "DenseFlow + Refine" DD:1161 -- here incl. the mask half of the flow thread --, "Dynamic detection
    // timecost" DD:1644 = fusion): {k-means, depth-edge, seg-and-merge, flow+refine, fusion}
    double t_stage[5] = {0, 0, 0, 0, 0};
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // reference DynaDetect.h:98-126
    DynaDetect(const Img8& imgLast, const Img8& imgLastLast, float fx_, float fy_, float cx_, float cy_, float depthScale_)
        : width(imgLast.w), height(imgLast.h), fx(fx_), fy(fy_), cx(cx_), cy(cy_), depthScale(depthScale_),
          imgRGBLast(imgLast), imgRGBLastLast(imgLastLast) {
        imgDyna.create(width, height, 1, 0); imgDynaLast = imgDyna; imgMaskHighErrorLast = imgDyna; imgLabelLast = imgDyna;
        const int ap[12][2] = {{0,-2},{1,-2},{2,-1},{2,0},{2,1},{1,2},{0,2},{-1,2},{-2,1},{-2,0},{-2,-1},{-1,-2}};
        for (auto& a : ap) aroundPoint.push_back({a[0], a[1]});
        element3 = ellipse_elem(3); element4 = ellipse_elem(4); element5 = ellipse_elem(5);
        element7 = ellipse_elem(7); element9 = ellipse_elem(9); element10 = ellipse_elem(10);
    }

    // ------------------------------------------------------------------ DD:1023-1374 flow -> masks
    // Part 1 (state free): resize, DeepFlow, large-motion test, refinement, upscale.
    void ComputeDenseFlow() {
        const float scale_element = 0.6f;
        const int fw = (int)(scale_element * width), fh = (int)(scale_element * height);
        resize_linear_u8(imgGray, dbg.grayMin, fw, fh);
        resize_linear_u8(imgGrayLast, dbg.grayLastMin, fw, fh);
        resize_linear_u8(imgGrayLastLast, dbg.grayLastLastMin, fw, fh);
        ImgF flow;
        DeepFlowParams dfp; dfp.maxLevels = flow_max_levels;
        deepflow_calc(dbg.grayMin, dbg.grayLastLastMin, flow, dfp);
        for (auto& v : flow.d) v *= -1.0f;
        bool largeMotion = false;
        {   // DD:1081-1114
            ImgF mag(fw, fh); float maxFlow = 0;
            for (size_t i = 0; i < mag.d.size(); i++) { float x = flow.d[2*i], y = flow.d[2*i+1]; mag.d[i] = std::sqrt(x * x + y * y); maxFlow = std::max(maxFlow, mag.d[i]); }
            const float a = (float)(255.0 / (double)maxFlow);
            int hist[256] = {0};
            for (float m : mag.d) hist[sat_u8f(m * a)]++;
            float ratio = 0.0f;
            int endFlow = (int)(10.0f * scale_element * 255.0f / (double)maxFlow), endFlow2 = 0;
            float totalpixel = width * height * scale_element * scale_element;
            for (int i = 0; i < 255; ++i) { ratio += (float)hist[i]; if (ratio > 0.3f * totalpixel) { endFlow2 = i; break; } }
            if (endFlow2 > endFlow) largeMotion = true;
            dbg.endFlow = endFlow; dbg.endFlow2 = endFlow2;
        }
        if (largeMotion) { deepflow_calc(dbg.grayMin, dbg.grayLastMin, flow, dfp); for (auto& v : flow.d) v *= -1.0f; }
        dbg.largeMotion = largeMotion; dbg.flowDeep = flow;
        {   // DD:1133-1143 VariationalRefinement::create()->calc on the u8 images, defaults
            const Img8& other = largeMotion ? dbg.grayLastMin : dbg.grayLastLastMin;
            ImgF I0(fw, fh), I1(fw, fh), Wu(fw, fh), Wv(fw, fh);
            for (size_t i = 0; i < I0.d.size(); i++) { I0.d[i] = dbg.grayMin.d[i]; I1.d[i] = other.d[i]; Wu.d[i] = flow.d[2*i]; Wv.d[i] = flow.d[2*i+1]; }
            varref_calc(VarRefParams(), I0, I1, Wu, Wv);
            for (size_t i = 0; i < I0.d.size(); i++) { flow.d[2*i] = Wu.d[i]; flow.d[2*i+1] = Wv.d[i]; }
        }
        dbg.flowRefined = flow;
        resize_linear_f32(flow, dbg.flowFull, width, height);       // DD:1144
        const float inv = 1.0f / scale_element;
        for (auto& v : dbg.flowFull.d) v *= inv;                    // DD:1147
    }

    // Part 2 (stateful): sample weights from the previous frame's result, homography, residual, thresholds.
    void FlowToMasks(Img8& maskLow, Img8& maskHigh) {
        const ImgF& flow = dbg.flowFull;
        RNG rng(12345);
        struct PW { int x, y; float weight; };
        std::vector<PW> pts;
        std::vector<float> clusterWeight(numCluster, 0.0f);
        for (int i = 1; i < numCluster; i++) {      // DD:1169-1177
            int nC = 0, nD = 0;
            for (size_t k = 0; k < imgLabelLast.d.size(); k++) if (imgLabelLast.d[k] == i) { nC++; nD += imgDynaLast.d[k] == 255; }
            clusterWeight[i] = (float)nD / (float)(nC + 1.0f);
        }
        for (int row = 10; row < height; row += 10) for (int col = 10; col < width; col += 10) {   // DD:1182-1204
            float randomd = (float)rng.gaussian(0.5);
            uint8_t dl = imgDynaLast.at(row, col);
            if (dl < 20) pts.push_back({col, row, randomd + 1.0f});
            else if ((unsigned)(dl - 20) <= 230 - 20) { int label = imgLabelLast.at(row, col); pts.push_back({col, row, randomd + 1.2f * (1.0f - clusterWeight[label])}); }
            else pts.push_back({col, row, randomd + 0.4f});
        }
        std::sort(pts.begin(), pts.end(), [](const PW& a, const PW& b) { return a.weight > b.weight; });
        std::vector<Pt2f> in, inLast;
        for (const PW& p : pts) {                   // DD:1220-1231
            float ptCol = (float)p.x, ptRow = (float)p.y;
            float fxv = flow.at(p.y, p.x, 0), fyv = flow.at(p.y, p.x, 1);
            int r = (int)(ptRow - fyv), c = (int)(ptCol - fxv);
            if ((unsigned)r <= (unsigned)height && (unsigned)c <= (unsigned)width) { in.push_back({ptCol, ptRow}); inLast.push_back({ptCol - fxv, ptRow - fyv}); }
        }
        dbg.nPairs = (int)in.size();
        double H[9];
        if (h_estimator == 2) { if (h_seed) find_homography_prosac(in, inLast, H, 3.0, 0.995, 2000, h_seed); else find_homography_prosac(in, inLast, H); }
        else find_homography_rho_scheme(in, inLast, H, 3.0f, 0.995, 2000, h_seed ? h_seed : ~0ull);
        std::copy(H, H + 9, dbg.H);
        ImgF mag(width, height); float maxErr = 0;
        for (int row = 0; row < height; ++row) for (int col = 0; col < width; ++col) {   // DD:1252-1271
            double den = H[6] * col + H[7] * row + H[8];
            double flowX2 = (col - (H[0] * col + H[1] * row + H[2]) / den);
            double flowY2 = (row - (H[3] * col + H[4] * row + H[5]) / den);
            float dx = flow.at(row, col, 0) - (float)flowX2, dy = flow.at(row, col, 1) - (float)flowY2;
            float m = std::sqrt(dx * dx + dy * dy);
            mag.at(row, col) = m; maxErr = std::max(maxErr, m);
        }
        const float maxErrorf = maxErr;
        const float a = (float)(255.0 / (double)maxErr);
        Img8 magN(width, height);
        for (size_t i = 0; i < mag.d.size(); i++) magN.d[i] = sat_u8f(mag.d[i] * a);
        hist256(magN, dbg.hist);
        float thred1 = (float)otsu_from_hist(dbg.hist, width * height);
        float thred2 = (float)triangle_from_hist(dbg.hist);
        dbg.maxError = maxErrorf; dbg.otsu = thred1; dbg.triangle = thred2; dbg.magU8 = magN;
        auto gt = [&](float t, Img8& out) { out.create(width, height); int n = 0; for (size_t i = 0; i < magN.d.size(); i++) { bool b = (double)magN.d[i] > (double)t; out.d[i] = b ? 255 : 0; n += b; } return n; };
        Img8 th1, th2;
        if (thred1 < thred2) {                      // DD:1309-1336
            if (thred1 < 1.7f * 255.0f / maxErrorf) thred1 = 1.7f * 255.0f / maxErrorf;
            else if (thred1 > 3.0f * 255.0f / maxErrorf) thred1 = 3.0f * 255.0f / maxErrorf;
            int n = gt(thred1, th1);
            if (n > 0.5 * width * height) { thred1 = thred1 + 0.2f * 255.0f / maxErrorf; gt(thred1, th1); }
            if (thred2 < std::max(3.0f * 255.0f / maxErrorf, thred1 * 1.2f)) thred2 = std::max(3.0f * 255.0f / maxErrorf, thred1 * 1.2f);
            else if (thred2 > 10.0f * 255.0f / maxErrorf) thred2 = 10.0f * 255.0f / maxErrorf;
            gt(thred2, th2);
            dbg.thr_low = thred1; dbg.thr_high = thred2;
            maskLow = th1; maskHigh = th2;
        } else {                                    // DD:1337-1367 (countNonZero(thred2) quirk: never relaxes, App. C-3)
            if (thred2 < 1.7f * 255.0f / maxErrorf) thred2 = 1.7f * 255.0f / maxErrorf;
            else if (thred2 > 3.0f * 255.0f / maxErrorf) thred2 = 3.0f * 255.0f / maxErrorf;
            gt(thred2, th2);
            if (thred1 < std::max(3.0f * 255.0f / maxErrorf, thred2 * 1.2f)) thred1 = std::max(3.0f * 255.0f / maxErrorf, thred2 * 1.2f);
            else if (thred1 > 10.0f * 255.0f / maxErrorf) thred1 = 10.0f * 255.0f / maxErrorf;
            gt(thred1, th1);
            dbg.thr_low = thred2; dbg.thr_high = thred1;
            maskLow = th2; maskHigh = th1;
        }
        for (auto& v : maskLow.d) v = v ? 128 : 0;  // "* 0.5": saturate_cast<uchar>(127.5) = 128
        dbg.maskLow = maskLow; dbg.maskHigh = maskHigh;
    }

    // ------------------------------------------------------------------ DD:315-420
    void SegByKmeans(ImgI& labelOut, std::vector<float>& points, std::vector<float>& centers) {
        const int pyramid_num = 4; const float scales[] = {1.0f, 0.5f, 0.25f, 0.125f};
        std::vector<Img16> depth_pyr(pyramid_num); depth_pyr[0] = imgDepth;
        for (int l = 1; l < pyramid_num; l++) resize_half_u16(depth_pyr[l - 1], depth_pyr[l]);
        std::vector<ImgI> label(pyramid_num);
        for (int level = pyramid_num - 1; level >= 0; level--) {
            const int hp = (int)(height * scales[level]), wp = (int)(width * scales[level]);
            const Img16& dp = depth_pyr[level];
            std::vector<int> labels((size_t)hp * wp); std::vector<float> pts((size_t)hp * wp * 3, 0.f), ctr;
            for (int row = 0; row < hp; ++row) for (int col = 0; col < wp; ++col) {
                int index = row * wp + col;
                uint16_t depth = (uint16_t)(dp.at(row, col) * scales[level]);
                if (depth / depthScale >= (uint16_t)6 || depth == 0) { pts[3*index] = pts[3*index+1] = pts[3*index+2] = 0; }
                else {
                    float depth2 = (float)(depth) * (1.0f / depthScale);
                    pts[3*index+2] = (float)(depth2 * depth_weight);
                    pts[3*index+0] = (float)((col - cx * scales[level]) * depth2 * (1.0f / (fx * scales[level])));
                    pts[3*index+1] = (float)((row - cy * scales[level]) * depth2 * (1.0f / (fy * scales[level])));
                }
            }
            if (level == pyramid_num - 1) {
                if (count_nonzero(imgLabelLast) == 0) {
                    float batch_rows = (float)hp / nRowCluster, batch_cols = (float)wp / nColCluster;
                    for (int i = 0; i < hp; i++) for (int j = 0; j < wp; j++) labels[i * wp + j] = cvFloor(i / batch_rows) * nColCluster + cvFloor(j / batch_cols);
                } else {
                    ImgF f(width, height), r; for (size_t k = 0; k < f.d.size(); k++) f.d[k] = imgLabelLast.d[k];
                    resize_linear_f32(f, r, wp, hp);
                    for (size_t k = 0; k < r.d.size(); k++) labels[k] = cvRoundf(r.d[k]);
                }
            } else {
                const ImgI& up = label[level + 1];
                ImgF f(up.w, up.h), r; for (size_t k = 0; k < f.d.size(); k++) f.d[k] = (float)up.d[k];
                resize_linear_f32(f, r, wp, hp);
                for (size_t k = 0; k < r.d.size(); k++) labels[k] = cvRoundf(r.d[k]);
            }
            kmeans_initial_labels(pts.data(), hp * wp, 3, numCluster, labels.data(), 4, 0.07, ctr);
            label[level].create(wp, hp); label[level].d.assign(labels.begin(), labels.end());
            if (level == 0) { points = pts; centers = ctr; }
        }
        labelOut = label[0];
    }

    // ------------------------------------------------------------------ DD:110-143 (scan order kept, see header)
    static void applyNMS(std::vector<Pt>& endpoints, float distanceThreshold) {
        std::vector<Pt> sel;
        for (const Pt& e : endpoints) {
            bool overlap = false;
            for (const Pt& s : sel) { int dx = e.x - s.x, dy = e.y - s.y; float d2 = (float)dx * dx + dy * dy; if (d2 < distanceThreshold * distanceThreshold) { overlap = true; break; } }
            if (!overlap) sel.push_back(e);
        }
        endpoints = sel;
    }

    // ------------------------------------------------------------------ DD:429-642
    void CalOccluded(Img8& imgTotalArea, Img8& imgOccluded1, Img8& imgOccluded2) {
        ImgF d1(width, height), filt; for (size_t i = 0; i < d1.d.size(); i++) d1.d[i] = imgDepth.d[i];
        median5_f32(d1, filt);
        float depth_max = 0; for (float v : filt.d) depth_max = std::max(depth_max, v);
        Img8 occ(width, height, 1, 0);
        const int range = 3;
        for (int row = range; row < height - range; ++row) for (int col = range; col < width - range; ++col) {
            float val_max = 0.0f, depth1 = filt.at(row, col);
            if (depth1 > 0.0f && depth1 / depthScale < 6.0f) imgTotalArea.at(row, col) = 255;
            for (int i = 0; i < 2 * range - 1; i++) for (int j = 0; j < 2 * range - 1; j++) {
                float nb = filt.at(row + i - range + 1, col + j - range + 1);
                if ((depth1 - nb) > (float)depth_max * 0.5f) continue;
                val_max = (std::fabs(val_max) > std::fabs(depth1 - nb)) ? std::fabs(val_max) : std::fabs(depth1 - nb);
            }
            if (val_max > depth1 * 0.03f && val_max > 400.0f) occ.at(row, col) = 255;
        }
        morph_open(occ, occ, element4);
        dbg.gradEdge = occ;
        std::vector<Pt> endPoints;
        for (int row = 3; row < height - 3; ++row) for (int col = 3; col < width - 3; ++col) {
            if (occ.at(row, col) != 255) continue;
            int aroundSum = 0;
            for (int i = 0; i < 12; ++i) if (occ.at(row + aroundPoint[i].y, col + aroundPoint[i].x) == 255) aroundSum++;
            if (aroundSum <= 4) endPoints.push_back({col, row});
        }
        Img8 occForPlane = occ;
        applyNMS(endPoints, 6.0f);
        // organised cloud (DD:558-589) and PEAC plane contours (DD:592-593)
        std::vector<float> cloud((size_t)width * height * 3);
        for (int v = 0; v < height; v++) for (int u = 0; u < width; u++) {
            float d = (float)imgDepth.at(v, u); float* p = &cloud[((size_t)v * width + u) * 3];
            if (d < 1e-3f) { p[0] = p[1] = p[2] = std::nanf(""); continue; }
            float z = d * (1.0f / depthScale);
            p[2] = z; p[0] = (u - cx) * z / fx; p[1] = (v - cy) * z / fy;
        }
        Img8 edgeByPlane(width, height, 1, 0);
        peac::Cloud pc{width, height, cloud.data()};
        pf.run(pc, edgeByPlane);
        dbg.planeContours = edgeByPlane;
        for (size_t i = 0; i < edgeByPlane.d.size(); i++) edgeByPlane.d[i] = sat_u8((int)edgeByPlane.d[i] - occForPlane.d[i]);   // DD:599
        {
            std::vector<Contour> contours; find_contours(edgeByPlane, contours, true);
            Img8 one(width, height, 1, 0), tmp;
            edgeByPlane.fill(0);
            for (const Contour& c : contours) {
                if (c.size() < 25) continue;
                one.fill(0);
                draw_contour_thick2(one, c, 255);
                dilate(one, one, element10);
                bool isEnd = false;
                for (const Pt& e : endPoints) if (one.at(e.y, e.x) == 255) { isEnd = true; break; }
                if (isEnd) { erode(one, one, element7); for (size_t i = 0; i < one.d.size(); i++) edgeByPlane.d[i] = sat_u8(edgeByPlane.d[i] + one.d[i]); }
            }
        }
        imgOccluded2 = edgeByPlane;
        Img8 u(width, height); for (size_t i = 0; i < u.d.size(); i++) u.d[i] = occ.d[i] | edgeByPlane.d[i];
        morph_close(u, imgOccluded1, element3);
    }

    // ------------------------------------------------------------------ DD:256-305, 1685-1739
    struct Cluster { Img8 img, dil, lianjie; float area = 0, score = -10; float center[3] = {0, 0, 0}; };
    static void calCenterPoint(Cluster& c, const std::vector<float>& points) {
        float totalCount = (float)count_nonzero(c.img);
        float f1 = 0, f2 = 0, f3 = 0;
        for (size_t i = 0; i < c.img.d.size(); i++) if (c.img.d[i]) { f1 += points[3*i]; f2 += points[3*i+1]; f3 += points[3*i+2]; }
        c.center[0] = f1 / totalCount; c.center[1] = f2 / totalCount; c.center[2] = f3 / totalCount;
    }
    static void cal_hist(const Img8& img1, const Img8& img2, const Img8& depthN, double out[3]) {
        float h1[256], h2[256];
        masked_hist_0_255(depthN, img1, h1); masked_hist_0_255(depthN, img2, h2);
        double m1 = 0, m2 = 0, mn1 = DBL_MAX, mn2 = DBL_MAX;
        for (int i = 0; i < 256; i++) { m1 = std::max<double>(m1, h1[i]); m2 = std::max<double>(m2, h2[i]); mn1 = std::min<double>(mn1, h1[i]); mn2 = std::min<double>(mn2, h2[i]); }
        const int hist_h = 400;
        auto norm_minmax = [&](float* h, double mn, double mx) {   // cv::normalize(NORM_MINMAX, 0..400): convertTo with float scale/shift
            double scale = (mx - mn) > DBL_EPSILON ? (double)hist_h / (mx - mn) : 0, shift = 0 - mn * scale;
            for (int i = 0; i < 256; i++) h[i] = h[i] * (float)scale + (float)shift;
        };
        if (m1 > m2) { norm_minmax(h1, mn1, m1); float s = (float)(1.0 / (m1 / hist_h)); for (int i = 0; i < 256; i++) h2[i] = h2[i] * s; }
        else         { norm_minmax(h2, mn2, m2); float s = (float)(1.0 / (m2 / hist_h)); for (int i = 0; i < 256; i++) h1[i] = h1[i] * s; }
        out[0] = compare_hist_correl(h1, h2, 256);
        out[1] = 1 - compare_hist_bhattacharyya(h1, h2, 256);
        out[2] = compare_hist_intersect(h1, h2, 256);
    }

    // ------------------------------------------------------------------ DD:653-1018
    void SegAndMergeV2(const std::vector<Img8>& allLabels, const Img8& imgOccluded, const Img8& imgOccluded2,
                       const Img8& imgLabelForSegEdge, const std::vector<float>& points, Img8& imgLabelNew) {
        std::vector<Cluster> all;
        Img8 occDil; dilate(imgOccluded, occDil, element10);
        for (int i = 0; i + 1 < (int)allLabels.size(); i++) {       // the last (farthest/invalid) label is skipped
            const Img8& orig = allLabels[i];
            Img8 each(width, height); for (size_t k = 0; k < each.d.size(); k++) each.d[k] = sat_u8((int)orig.d[k] - imgOccluded.d[k]);
            morph_open(each, each, element4);
            std::vector<Contour> contours; find_contours(each, contours, true);
            for (const Contour& c : contours) {
                if (!(c.size() > 50 && contour_area(c) > 80)) continue;
                Img8 temp(width, height, 1, 0), drawC(width, height, 1, 0);
                draw_contour_filled(temp, c, 255);
                dilate(temp, temp, element9);
                for (size_t k = 0; k < temp.d.size(); k++) temp.d[k] &= orig.d[k];
                Cluster nc; nc.img = temp; nc.area = (float)count_nonzero(temp);
                dilate(temp, temp, element7); nc.dil = temp;
                draw_contour_thick2(drawC, c, 255);
                Img8 temp1(width, height); for (size_t k = 0; k < temp1.d.size(); k++) temp1.d[k] = sat_u8((int)drawC.d[k] - occDil.d[k]) & imgLabelForSegEdge.d[k];
                if (count_nonzero(temp1) > 20) {
                    std::vector<Contour> c2; find_contours(temp1, c2, true);
                    std::vector<Contour> kept; for (auto& q : c2) if (q.size() >= 30) kept.push_back(q);
                    if (!kept.empty()) { temp1.fill(0); draw_contours_filled_joint(temp1, kept, 255); nc.lianjie = temp1; }
                }
                calCenterPoint(nc, points);
                all.push_back(std::move(nc));
            }
        }
        const int C = (int)all.size();
        dbg.nClusters = C;
        Img8 total(width, height, 1, (uint8_t)(C + 1));
        for (auto& c : all) c.score = (float)(c.area * 0.0003f - c.center[2]);
        std::sort(all.begin(), all.end(), [](const Cluster& a, const Cluster& b) { return a.score > b.score; });
        for (int i = 0; i < C; i++) for (size_t k = 0; k < total.d.size(); k++) if (all[i].img.d[k]) total.d[k] = (uint8_t)i;
        // DD:765-768 imgDepth/depth_max*255 evaluated in CV_16U with a float scale, then to 8U
        uint16_t dmax = 0; for (uint16_t v : imgDepth.d) dmax = std::max(dmax, v);
        Img8 depthN(width, height);
        { const float a = (float)((1.0 / (double)dmax) * 255); for (size_t k = 0; k < depthN.d.size(); k++) { int v = cvRoundf((float)imgDepth.d[k] * a); depthN.d[k] = sat_u8(std::min(std::max(v, 0), 65535)); } }
        const int M = C + 1;
        std::vector<float> mTotal((size_t)M * M, 0.f), m1((size_t)M * M, 0.f), m2((size_t)M * M, 0.f), m3((size_t)M * M, 0.f), wgt((size_t)M * M, 1.f), rej((size_t)M * M, 1.f);
        const float thredshold = 0.9f;
        const int smallLabel = (int)std::min(0.7f * C, 15.0f);
        std::vector<int> lianjieArea(C, 0); for (int i = 0; i < C; i++) if (!all[i].lianjie.empty()) lianjieArea[i] = count_nonzero(all[i].lianjie);
        for (int i = 0; i < C; i++) for (int j = i + 1; j < C; j++) {
            float v1 = 0, v2 = 0, v3 = 0; float lessArea; int lessLabel;
            if (all[i].area < all[j].area) { lessArea = all[i].area; lessLabel = i; } else { lessArea = all[j].area; lessLabel = j; }
            if (lessLabel < 10) wgt[i * M + j] = wgt[j * M + i] = 0.7f;
            else if (lessLabel > smallLabel) wgt[i * M + j] = wgt[j * M + i] = 2.0f;
            int overlap = 0, overlapPlane = 0;
            for (size_t k = 0; k < total.d.size(); k++) if (all[i].dil.d[k] & all[j].dil.d[k]) { overlap++; overlapPlane += imgOccluded2.d[k] != 0; }
            if (overlap > std::min(200.0f, lessArea * 0.4f)) {
                v1 = 1.0f;
                double isMerge[3]; cal_hist(all[i].img, all[j].img, depthN, isMerge);
                v3 = (float)(isMerge[0] + isMerge[1] + isMerge[2] * 0.0005);
                if (overlapPlane > 100 && lessLabel < smallLabel) { rej[i * M + j] = rej[j * M + i] = 0.f; continue; }
                else if (v3 < 0.19f && lessLabel < smallLabel) { rej[i * M + j] = rej[j * M + i] = 0.f; continue; }
                if (!all[i].lianjie.empty() && !all[j].lianjie.empty()) {
                    int ov = 0; for (size_t k = 0; k < total.d.size(); k++) ov += (all[i].lianjie.d[k] & all[j].lianjie.d[k]) != 0;
                    if (ov > 0) {
                        int a1 = lianjieArea[i], a2 = lianjieArea[j];
                        if (ov > std::min(50, (int)(0.5 * std::min(a1, a2)))) {
                            v2 = (float)ov;
                            if ((ov > 0.62 * a1) || (ov > 0.62 * a2)) v2 = (float)std::max(250, ov);
                        }
                    }
                }
                m1[i * M + j] = m1[j * M + i] = v1; m2[i * M + j] = m2[j * M + i] = v2; m3[i * M + j] = m3[j * M + i] = v3;
            }
        }
        for (size_t k = 0; k < mTotal.size(); k++) mTotal[k] = ((m2[k] * 0.01f + m3[k]) * rej[k]) * wgt[k];
        int countMerged = 0;
        std::vector<std::vector<int>> merge(M); std::vector<int> mergeSit(M, 0);
        auto fold = [&](int dst, int j) {      // column/row j of the RAG is added into dst and cleared (DD:962-966)
            std::vector<float> oneCol(M); for (int r = 0; r < M; r++) oneCol[r] = mTotal[r * M + j];
            for (int r = 0; r < M; r++) mTotal[r * M + dst] += oneCol[r];
            for (int c = 0; c < M; c++) mTotal[dst * M + c] += oneCol[c];
            for (int r = 0; r < M; r++) mTotal[r * M + j] = 0.f;
            for (int c = 0; c < M; c++) mTotal[j * M + c] = 0.f;
        };
        for (int i = 0; i < std::min(numCluster - 1 + countMerged, C); i++)
            for (int j = i + 1; j < std::min(numCluster - 1 + countMerged, C); j++) {
                float sorce = mTotal[j * M + i];
                if (sorce > thredshold) {
                    int toMerge = i; float toMergeValue = mTotal[j * M + i];
                    for (int k = 0; k < j; k++) if (mTotal[k * M + j] > toMergeValue) toMerge = k;   // quirk C-7: last k above the INITIAL value
                    mergeSit[j] = 1; merge[toMerge].push_back(j);
                    fold(toMerge, j);
                    countMerged++;
                }
            }
        for (int i = std::min(numCluster - 1 + countMerged, C); i < C; i++) {
            int mergeCluster = C; float maxScore = 0.2f;
            for (int j = 0; j < i; j++) { float score = mTotal[j * M + i]; if (score > maxScore) { maxScore = score; mergeCluster = j; } }
            mergeSit[i] = 1; merge[mergeCluster].push_back(i);
            fold(mergeCluster, i);
        }
        int labelindex = 1;
        for (int i = 0; i < C; i++) {
            if (mergeSit[i]) continue;
            std::vector<char> sel(M + 1, 0); sel[i] = 1;
            for (int mj : merge[i]) { sel[mj] = 1; for (int mk : merge[mj]) sel[mk] = 1; }
            for (size_t k = 0; k < total.d.size(); k++) if (sel[total.d[k]]) imgLabelNew.d[k] = (uint8_t)labelindex;
            labelindex++;
        }
    }

    // ------------------------------------------------------------------ DD:1377-1666
    void DetectDynaArea(const Img8& img, const Img16& depth, Img8& imgDynaOut, Img8& imgLabelOut) {
        imgRGB = img; imgDepth = depth;
        bgr2gray(imgRGB, imgGray); bgr2gray(imgRGBLast, imgGrayLast); bgr2gray(imgRGBLastLast, imgGrayLastLast);
        dbg.gray = imgGray; dbg.grayLast = imgGrayLast; dbg.grayLastLast = imgGrayLastLast;
        imgDyna.fill(0);
        Img8 maskLow, maskHigh;
        double ts = now_s();
        #define ORC_LAP(i) { const double t_ = now_s(); t_stage[i] += t_ - ts; ts = t_; }
        if (!skip_flow) ComputeDenseFlow();
        FlowToMasks(maskLow, maskHigh);          // the reference runs this in a side thread; it only reads *Last state
        ORC_LAP(3)
        // k-means
        ImgI labelI; std::vector<float> points, centers;
        SegByKmeans(labelI, points, centers);
        imgLabel.create(width, height); for (size_t k = 0; k < labelI.d.size(); k++) imgLabel.d[k] = sat_u8(labelI.d[k]);
        dbg.kmeansLabel = imgLabel; dbg.centers = centers;
        // sort clusters by centre z (DD:1428-1438)
        std::vector<float> depth_vals(numCluster); for (int i = 0; i < numCluster; i++) { depth_vals[i] = centers[3*i+2]; if (depth_vals[i] < 0.2) depth_vals[i] += 20.0f; }
        std::vector<int> sortDepth(numCluster); for (int i = 0; i < numCluster; i++) sortDepth[i] = i;
        std::stable_sort(sortDepth.begin(), sortDepth.end(), [&](int a, int b) { return depth_vals[a] < depth_vals[b]; });
        Img8 labelForSegEdge(width, height, 1, 0), totalArea(width, height, 1, 0);
        std::vector<Img8> allLabels; float ratioArea = 0.0f; const float TotalArea = (float)(height * width); int count0 = 0;
        for (int i = 0; i < numCluster; i++) {
            int idx = sortDepth[i];
            Img8 each(width, height); int cnt = 0; for (size_t k = 0; k < each.d.size(); k++) { each.d[k] = imgLabel.d[k] == idx ? 255 : 0; cnt += each.d[k] != 0; }
            if (cnt < 60) continue;
            allLabels.push_back(each);
            float ratio = (float)cnt * (1.0f / TotalArea); ratioArea += ratio;
            if (count0 <= 5 && ratioArea < 0.6f) { for (size_t k = 0; k < each.d.size(); k++) labelForSegEdge.d[k] |= each.d[k]; ++count0; }
        }
        dilate(labelForSegEdge, labelForSegEdge, element7);
        ORC_LAP(0)
        Img8 occ1(width, height, 1, 0), occ2(width, height, 1, 0);
        CalOccluded(totalArea, occ1, occ2);
        ORC_LAP(1)
        dbg.labelForSegEdge = labelForSegEdge; dbg.totalArea = totalArea; dbg.occluded1 = occ1; dbg.occluded2 = occ2;
        Img8 label3(width, height, 1, 0);
        if (!allLabels.empty()) SegAndMergeV2(allLabels, occ1, occ2, labelForSegEdge, points, label3);
        imgLabel = label3;
        ORC_LAP(2)
        int maxNum = 0; for (uint8_t v : label3.d) maxNum = std::max<int>(maxNum, v);
        // fusion (DD:1553-1636)
        for (size_t k = 0; k < maskLow.d.size(); k++) { uint8_t v = imgMaskHighErrorLast.d[k] | maskLow.d[k]; maskLow.d[k] = (v ? 128 : 0) & totalArea.d[k]; }
        dilate(maskLow, maskLow, element5);
        for (int n = 1; n <= maxNum; n++) {
            Img8 one(width, height); int oneCnt = 0; for (size_t k = 0; k < one.d.size(); k++) { one.d[k] = imgLabel.d[k] == n ? 255 : 0; oneCnt += one.d[k] != 0; }
            Img8 border(width + 2, height + 2, 1, 255);
            for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) border.at(y + 1, x + 1) = one.at(y, x) ? 0 : 255;
            Img8 oneHigh(width, height); int totalA = 0; for (size_t k = 0; k < one.d.size(); k++) { oneHigh.d[k] = one.d[k] & maskHigh.d[k]; totalA += oneHigh.d[k] != 0; }
            if (totalA > 100) {
                std::vector<Contour> c2; find_contours(oneHigh, c2, false);
                for (const Contour& c : c2) {
                    double area = contour_area(c), len = arc_length_closed(c), roundness = (4 * M_PI * area) / (len * len);
                    Pt seed = {0, 0};
                    for (const Pt& p : c) if (maskLow.at(p.y, p.x) == 128) { seed = p; break; }
                    if ((area > 100.0 && roundness > 0.2) || area > 2000.0) flood_fill_mask_only(maskLow, border, seed, 50, 5);
                }
            }
            int filled = 0; for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) filled += border.at(y + 1, x + 1) == 50;
            if (filled > 0.5 * oneCnt) { for (size_t k = 0; k < one.d.size(); k++) imgDyna.d[k] |= one.d[k]; }
            else { for (int y = 0; y < height; y++) for (int x = 0; x < width; x++) if (border.at(y + 1, x + 1) == 50) imgDyna.at(y, x) = 255; }
        }
        dilate(imgDyna, imgDyna, element9);
        for (size_t k = 0; k < imgDyna.d.size(); k++) if (!imgDyna.d[k] && totalArea.d[k]) imgDyna.d[k] = 125;   // DD:1633-1634
        imgDynaOut = imgDyna; imgLabelOut = imgLabel;
        ORC_LAP(4)
        #undef ORC_LAP
        dbg.dyna = imgDyna; dbg.label = imgLabel;
        imgDynaLast = imgDyna; imgRGBLastLast = imgRGBLast; imgRGBLast = imgRGB; imgMaskHighErrorLast = maskHigh; imgLabelLast = imgLabel;
    }
};

// caller-side dilation before tracking (rgbd_tum_noros.cc:108,138): 15x15 ellipse on the 0/125/255 image.
// Three-valued input, so a plain max filter is used here.
inline void dilate_ellipse15(const Img8& src, Img8& dst) {
    StructElem e = ellipse_elem(15);
    Img8 out(src.w, src.h, 1, 0);
    for (int y = 0; y < src.h; y++) for (int x = 0; x < src.w; x++) {
        uint8_t m = 0;
        for (int i = 0; i < e.n; i++) { int yy = y + i - e.ay; if (yy < 0 || yy >= src.h) continue;
            for (int j = e.j1[i]; j < e.j2[i]; j++) { int xx = x + j - e.ax; if (xx < 0 || xx >= src.w) continue; m = std::max(m, src.at(yy, xx)); } }
        out.at(y, x) = m;
    }
    dst = std::move(out);
}

}  // namespace cvx
