// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
// C entry points so tests/ (ctypes), smoke() and bench.py's cpu_baseline leg can drive the CPU restatement.
#include <chrono>
#include "dynadetect.hpp"
#include "orb.hpp"
#include "frame.hpp"
#include "matcher.hpp"
#include "cloud.hpp"

using namespace cvx;

static Img8 wrap8(const uint8_t* p, int w, int h, int c = 1) { Img8 m; m.w = w; m.h = h; m.c = c; m.d.assign(p, p + (size_t)w * h * c); return m; }
static Img16 wrap16(const uint16_t* p, int w, int h) { Img16 m; m.w = w; m.h = h; m.c = 1; m.d.assign(p, p + (size_t)w * h); return m; }
template <class T> static void put(const Img<T>& m, T* out) { if (out && !m.d.empty()) std::memcpy(out, m.d.data(), m.d.size() * sizeof(T)); }

extern "C" {

// ---------------------------------------------------------------- primitives
void orc_bgr2gray(const uint8_t* bgr, int w, int h, int swap_rb, uint8_t* out) { Img8 g; bgr2gray(wrap8(bgr, w, h, 3), g, swap_rb != 0); put(g, out); }
void orc_resize_u8(const uint8_t* src, int sw, int sh, int dw, int dh, uint8_t* out) { Img8 d; resize_linear_u8(wrap8(src, sw, sh), d, dw, dh); put(d, out); }
void orc_resize_f32(const float* src, int sw, int sh, int cn, int dw, int dh, float* out) {
    ImgF s; s.w = sw; s.h = sh; s.c = cn; s.d.assign(src, src + (size_t)sw * sh * cn); ImgF d; resize_linear_f32(s, d, dw, dh); put(d, out); }
void orc_gaussian_blur_u8(const uint8_t* src, int w, int h, int ksize, double sigma, uint8_t* out) { Img8 d; gaussian_blur_u8(wrap8(src, w, h), d, ksize, sigma); put(d, out); }
void orc_gaussian_blur3_f32(const float* src, int w, int h, double sigma, float* out) { ImgF s(w, h); s.d.assign(src, src + (size_t)w * h); ImgF d; gaussian_blur3_f32(s, d, sigma); put(d, out); }
double orc_otsu(const int* hist, int total) { return otsu_from_hist(hist, total); }
double orc_triangle(const int* hist) { return triangle_from_hist(hist); }
void orc_rng_gaussian(uint64_t seed, double sigma, int n, float* out) { RNG r(seed); for (int i = 0; i < n; i++) out[i] = (float)r.gaussian(sigma); }
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int orc_deepflow_levels(int w, int h, int* ws, int* hs, int cap) { auto s = deepflow_level_sizes(w, h); for (int i = 0; i < (int)s.size() && i < cap; i++) { ws[i] = s[i].first; hs[i] = s[i].second; } return (int)s.size(); }

// VariationalRefinement on float images.  wu/wv in-out.  Optional intermediates (each w*h floats or NULL):
// order: warped, Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz, A11, A12, A22, b1, b2, wgt, dWu, dWv  (state after the LAST fixed-point iteration)
void orc_varref(const float* i0, const float* i1, int w, int h, float* wu, float* wv, int fp_iters, int sor_iters,
                float alpha, float delta, float gamma, float omega, float** inter) {
    VarRefParams P; P.fixedPointIterations = fp_iters; P.sorIterations = sor_iters; P.alpha = alpha; P.delta = delta; P.gamma = gamma; P.omega = omega;
    ImgF I0(w, h), I1(w, h), Wu(w, h), Wv(w, h);
    I0.d.assign(i0, i0 + (size_t)w * h); I1.d.assign(i1, i1 + (size_t)w * h); Wu.d.assign(wu, wu + (size_t)w * h); Wv.d.assign(wv, wv + (size_t)w * h);
    VarRefBuffers B; varref_calc(P, I0, I1, Wu, Wv, &B);
    put(Wu, wu); put(Wv, wv);
    if (inter) { const ImgF* a[17] = {&B.warped, &B.Ix, &B.Iy, &B.Iz, &B.Ixx, &B.Ixy, &B.Iyy, &B.Ixz, &B.Iyz, &B.A11, &B.A12, &B.A22, &B.b1, &B.b2, &B.wgt, &B.dWu, &B.dWv};
        for (int k = 0; k < 17; k++) if (inter[k]) put(*a[k], inter[k]); }
}
void orc_deepflow(const uint8_t* i0, const uint8_t* i1, int w, int h, float* flow /* w*h*2 */) { ImgF f; deepflow_calc(wrap8(i0, w, h), wrap8(i1, w, h), f); put(f, flow); }
void orc_deepflow_levels_capped(const uint8_t* i0, const uint8_t* i1, int w, int h, int max_levels, float* flow) { ImgF f; DeepFlowParams P; P.maxLevels = max_levels; deepflow_calc(wrap8(i0, w, h), wrap8(i1, w, h), f, P); put(f, flow); }

int orc_find_homography(const float* src, const float* dst, int n, double* H) {
    std::vector<Pt2f> s(n), d(n); for (int i = 0; i < n; i++) { s[i] = {src[2*i], src[2*i+1]}; d[i] = {dst[2*i], dst[2*i+1]}; }
    return find_homography_rho_scheme(s, d, H) ? 1 : 0;
}
int orc_find_homography_prosac_ls(const float* src, const float* dst, int n, double* H) {          // round 1's estimator (sensitivity comparison only)
    std::vector<Pt2f> s(n), d(n); for (int i = 0; i < n; i++) { s[i] = {src[2*i], src[2*i+1]}; d[i] = {dst[2*i], dst[2*i+1]}; }
    return find_homography_prosac(s, d, H) ? 1 : 0;
}

// property-test access to the k-means and flood-fill restatements
void orc_kmeans(const float* data, int N, int dims, int K, int* labels /* in: initial, out: final */, int maxCount, double eps, float* centers /* K x dims */) {
    std::vector<float> c; kmeans_initial_labels(data, N, dims, K, labels, maxCount, eps, c); std::copy(c.begin(), c.end(), centers);
}
int orc_flood_fill_mask_only(const uint8_t* image, int w, int h, uint8_t* mask /* (h+2) x (w+2), in/out */, int sx, int sy, int new_val, int diff) {
    Img8 img = wrap8(image, w, h), m = wrap8(mask, w + 2, h + 2);
    const int area = flood_fill_mask_only(img, m, Pt{sx, sy}, (uint8_t)new_val, diff); put(m, mask); return area;
}

// morphology / contours for unit tests
void orc_morph(const uint8_t* src, int w, int h, int n, int op /*0 dilate 1 erode 2 open 3 close*/, uint8_t* out) {
    Img8 s = wrap8(src, w, h), d; StructElem e = ellipse_elem(n);
    if (op == 0) dilate(s, d, e); else if (op == 1) erode(s, d, e); else if (op == 2) morph_open(s, d, e); else morph_close(s, d, e);
    put(d, out);
}
int orc_find_contours(const uint8_t* src, int w, int h, int external_only, int* pts_xy, int cap_pts, int* lens, int cap_contours) {
    std::vector<Contour> cs; find_contours(wrap8(src, w, h), cs, external_only != 0);
    int np = 0, nc = 0;
    for (auto& c : cs) { if (nc >= cap_contours) break; lens[nc++] = (int)c.size(); for (auto& p : c) { if (np < cap_pts) { pts_xy[2*np] = p.x; pts_xy[2*np+1] = p.y; } np++; } }
    return (int)cs.size();
}
void orc_median5_f32(const float* src, int w, int h, float* out) { ImgF s(w, h); s.d.assign(src, src + (size_t)w * h); ImgF d; median5_f32(s, d); put(d, out); }
void orc_dilate15(const uint8_t* src, int w, int h, uint8_t* out) { Img8 d; dilate_ellipse15(wrap8(src, w, h), d); put(d, out); }

// ---------------------------------------------------------------- DynaDetect
struct OrcDyna { DynaDetect* dd; };
void* orc_dyna_create(const uint8_t* bgr_last, const uint8_t* bgr_lastlast, int w, int h, float fx, float fy, float cx, float cy, float depthScale) {
    return new DynaDetect(wrap8(bgr_last, w, h, 3), wrap8(bgr_lastlast, w, h, 3), fx, fy, cx, cy, depthScale);
}
void orc_dyna_destroy(void* p) { delete (DynaDetect*)p; }
void orc_dyna_set_h_estimator(void* p, int which, uint64_t seed) { ((DynaDetect*)p)->h_estimator = which; ((DynaDetect*)p)->h_seed = seed; }
// a second detector that continues from the same inter-frame state (the five *Last images of DynaDetect.h:172-178)
void* orc_dyna_fork(void* p) {
    DynaDetect* a = (DynaDetect*)p; DynaDetect* b = new DynaDetect(a->imgRGBLast, a->imgRGBLastLast, a->fx, a->fy, a->cx, a->cy, a->depthScale);
    b->imgDynaLast = a->imgDynaLast; b->imgLabelLast = a->imgLabelLast; b->imgMaskHighErrorLast = a->imgMaskHighErrorLast; b->flow_max_levels = a->flow_max_levels;
    return b;
}
void orc_dyna_set_flow_max_levels(void* p, int n) { ((DynaDetect*)p)->flow_max_levels = n; }
void orc_dyna_detect(void* p, const uint8_t* bgr, const uint16_t* depth, uint8_t* dyna_out, uint8_t* label_out) {
    DynaDetect* d = (DynaDetect*)p; Img8 dy, lb;
    d->DetectDynaArea(wrap8(bgr, d->width, d->height, 3), wrap16(depth, d->width, d->height), dy, lb);
    put(dy, dyna_out); put(lb, label_out);
}
// state-free front half only (gray, resize, DeepFlow, large-motion choice, refinement, upscale): flow_full out (w*h*2)
void orc_dyna_flow_only(void* p, const uint8_t* bgr, float* flow_full, float* flow_deep, float* flow_refined, int* large_motion) {
    DynaDetect* d = (DynaDetect*)p;
    d->imgRGB = wrap8(bgr, d->width, d->height, 3);
    bgr2gray(d->imgRGB, d->imgGray); bgr2gray(d->imgRGBLast, d->imgGrayLast); bgr2gray(d->imgRGBLastLast, d->imgGrayLastLast);
    d->ComputeDenseFlow();
    put(d->dbg.flowFull, flow_full); put(d->dbg.flowDeep, flow_deep); put(d->dbg.flowRefined, flow_refined);
    if (large_motion) *large_motion = d->dbg.largeMotion;
}
// run DetectDynaArea with a caller-supplied 640x480x2 flow (skips the dense-flow front half)
void orc_dyna_detect_with_flow(void* p, const uint8_t* bgr, const uint16_t* depth, const float* flow_full, uint8_t* dyna_out, uint8_t* label_out) {
    DynaDetect* d = (DynaDetect*)p;
    d->dbg.flowFull.create(d->width, d->height, 2); std::memcpy(d->dbg.flowFull.d.data(), flow_full, d->dbg.flowFull.d.size() * sizeof(float));
    d->skip_flow = true; orc_dyna_detect(p, bgr, depth, dyna_out, label_out); d->skip_flow = false;
}
// intermediates of the last call; any pointer may be NULL
void orc_dyna_debug(void* p, float* flow_full, double* H, float* thr /*maxError, otsu, triangle, low, high*/, int* hist, uint8_t* mask_low,
                    uint8_t* mask_high, uint8_t* kmeans_label, float* centers, uint8_t* occ1, uint8_t* occ2, uint8_t* total_area, uint8_t* mag_u8, int* info /*largeMotion,nPairs,nClusters*/) {
    DynaDetect* d = (DynaDetect*)p; const DynaIntermediates& g = d->dbg;
    put(g.flowFull, flow_full);
    if (H) std::copy(g.H, g.H + 9, H);
    if (thr) { thr[0] = g.maxError; thr[1] = g.otsu; thr[2] = g.triangle; thr[3] = g.thr_low; thr[4] = g.thr_high; }
    if (hist) std::copy(g.hist, g.hist + 256, hist);
    put(g.maskLow, mask_low); put(g.maskHigh, mask_high); put(g.kmeansLabel, kmeans_label);
    if (centers && !g.centers.empty()) std::copy(g.centers.begin(), g.centers.end(), centers);
    put(g.occluded1, occ1); put(g.occluded2, occ2); put(g.totalArea, total_area); put(g.magU8, mag_u8);
    if (info) { info[0] = g.largeMotion; info[1] = g.nPairs; info[2] = g.nClusters; }
}

void orc_dyna_debug2(void* p, uint8_t* grad_edge, uint8_t* plane_contours, uint8_t* label_for_seg_edge) {
    DynaDetect* d = (DynaDetect*)p; put(d->dbg.gradEdge, grad_edge); put(d->dbg.planeContours, plane_contours); put(d->dbg.labelForSegEdge, label_for_seg_edge);
}

// ---------------------------------------------------------------- ORBextractor
struct OrcKp { float x, y, size, angle, response; int octave, class_id; };
void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) { return new ORBextractor(nfeatures, scaleFactor, nlevels, iniTh, minTh); }
void orc_orb_destroy(void* p) { delete (ORBextractor*)p; }
static int put_kps(const std::vector<KeyPoint>& v, OrcKp* out, int cap) {
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = {v[i].x, v[i].y, v[i].size, v[i].angle, v[i].response, v[i].octave, v[i].class_id};
    return (int)v.size();
}
int orc_orb_extract(void* p, const uint8_t* gray, int w, int h, const uint8_t* mask_or_null, OrcKp* kps, int cap, uint8_t* desc) {
    ORBextractor* o = (ORBextractor*)p; std::vector<KeyPoint> k; std::vector<uint8_t> d;
    Img8 mask; if (mask_or_null) mask = wrap8(mask_or_null, w, h);
    o->extract(wrap8(gray, w, h), mask, k, d);
    int n = put_kps(k, kps, cap);
    if (desc) std::memcpy(desc, d.data(), (size_t)std::min(n, cap) * 32);
    return n;
}
int orc_orb_level_size(void* p, int level, int* w, int* h) { ORBextractor* o = (ORBextractor*)p; if (level < 0 || level >= (int)o->level_size.size()) return -1; *w = o->level_size[level].first; *h = o->level_size[level].second; return 0; }
void orc_orb_level_padded(void* p, int level, uint8_t* out) { put(((ORBextractor*)p)->pyr_padded[level], out); }
int orc_orb_fast_keypoints(void* p, int level, OrcKp* out, int cap) { return put_kps(((ORBextractor*)p)->dbg_fast[level], out, cap); }
int orc_orb_selected(void* p, int level, OrcKp* out, int cap) { return put_kps(((ORBextractor*)p)->dbg_selected[level], out, cap); }
void orc_orb_tables(void* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* per_level, int* umax) {
    ORBextractor* o = (ORBextractor*)p;
    for (int i = 0; i < o->nlevels; i++) { scale[i] = o->mvScaleFactor[i]; inv_scale[i] = o->mvInvScaleFactor[i]; sigma2[i] = o->mvLevelSigma2[i]; inv_sigma2[i] = o->mvInvLevelSigma2[i]; per_level[i] = o->mnFeaturesPerLevel[i]; }
    for (int i = 0; i < 16; i++) umax[i] = o->umax[i];
}

// ---------------------------------------------------------------- Frame post-ORB steps (Frame.cc:143-170)
// calib = {fx, fy, cx, cy, k1, k2, p1, p2, k3, bf, depthMapFactor}; outputs sized N (grid_start 3073, grid_idx N); returns the grid entry count
int orc_frame_post_orb(const float* calib11, const float* kx, const float* ky, int N, const uint16_t* depth, int w, int h,
                       float* un_xy, float* u_right, float* depth_out, int* cell, int* grid_start, int* grid_idx, float* bounds4) {
    FrameCalib c{calib11[0], calib11[1], calib11[2], calib11[3], calib11[4], calib11[5], calib11[6], calib11[7], calib11[8], calib11[9], calib11[10]};
    FramePost o; frame_post_orb(c, kx, ky, N, depth, w, h, o);
    for (int i = 0; i < N; i++) { un_xy[2 * i] = o.unx[i]; un_xy[2 * i + 1] = o.uny[i]; u_right[i] = o.uRight[i]; depth_out[i] = o.depth[i]; cell[i] = o.cell[i]; }
    std::memcpy(grid_start, o.gridStart.data(), o.gridStart.size() * sizeof(int));
    if (!o.gridIdx.empty()) std::memcpy(grid_idx, o.gridIdx.data(), o.gridIdx.size() * sizeof(int));
    if (bounds4) frame_bounds(c, w, h, bounds4);
    return (int)o.gridIdx.size();
}

// ---------------------------------------------------------------- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:1328-1470)
// cam10 = {fx, fy, cx, cy, bf, mb, minX, maxX, minY, maxY}; Tcw: 4x4 row-major
int orc_search_by_projection(const float* cam10, const float* scale, int nlevels, const float* TcwCur, const float* TcwLast, int nLast, const float* x3Dw,
                             const uint8_t* lastValid, const uint8_t* lastHasObs, const int* lastOctave, const float* lastAngle, const uint8_t* lastDesc, int nCur,
                             const float* curUnXY, const int* curOctave, const float* curAngle, const float* curURight, const uint8_t* curDesc, const int* gridStart,
                             const int* gridIdx, const uint8_t* curTaken, float th, int mono, int checkOrientation, int* matchOfCur) {
    MatchInput in{}; in.fx = cam10[0]; in.fy = cam10[1]; in.cx = cam10[2]; in.cy = cam10[3]; in.bf = cam10[4]; in.mb = cam10[5];
    for (int i = 0; i < 4; i++) in.bounds[i] = cam10[6 + i];
    in.scaleFactors = scale; in.nlevels = nlevels; std::memcpy(in.TcwCur, TcwCur, 48); std::memcpy(in.TcwLast, TcwLast, 48);
    in.nLast = nLast; in.x3Dw = x3Dw; in.lastValid = lastValid; in.lastHasObs = lastHasObs; in.lastOctave = lastOctave; in.lastAngle = lastAngle; in.lastDesc = lastDesc;
    in.nCur = nCur; in.curUnXY = curUnXY; in.curOctave = curOctave; in.curAngle = curAngle; in.curURight = curURight; in.curDesc = curDesc; in.gridStart = gridStart; in.gridIdx = gridIdx;
    in.curTaken = curTaken; in.th = th; in.mono = mono != 0; in.checkOrientation = checkOrientation != 0;
    return search_by_projection(in, matchOfCur);
}
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// ---------------------------------------------------------------- octomap_pub generatePointCloud (pubPointCloud.cc:471-660)
// cam5 = {fx, fy, cx, cy, depthScale} (doubles); returns the point count (points_out may be NULL to query it; 16 B per point)
int orc_generate_point_cloud(const double* cam5, const uint8_t* bgr, const uint16_t* depth, const uint16_t* depthLast, const uint8_t* dyna, const uint8_t* dynaLast,
                             const uint8_t* label, int w, int h, const double* poseRelative16, const double* Twc16, void* points_out, int cap, double* occlusion12,
                             int* labelCount12, uint8_t* kept12) {
    CloudParams P{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4]}; std::vector<CloudPoint> out;
    generate_point_cloud(P, bgr, depth, depthLast, dyna, dynaLast, label, w, h, poseRelative16, Twc16, out, occlusion12, labelCount12, kept12);
    if (points_out) std::memcpy(points_out, out.data(), std::min(out.size(), (size_t)cap) * sizeof(CloudPoint));
    return (int)out.size();
}

static int put_kps(const std::vector<KeyPoint>& v, OrcKp* out, int cap);
// ---------------------------------------------------------------- CPU baseline: frames through DynaDetect + dilate + ORB, seconds out
// bgr: n frames (w*h*3 each), depth: n frames.  Frames 0,1 prime the detector; pairs = n-2.  Gray for ORB = BGR2GRAY (Camera.RGB: 0).
double orc_baseline_run(const uint8_t* bgr, const uint16_t* depth, int n, int w, int h, float fx, float fy, float cx, float cy, float depthScale,
                        int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, double* stage_seconds /*flow, tail, orb, then the reference's breakdown: kmeans, depth-edge, seg-merge, flow+refine+masks, fusion, orb (9 values)*/,
                        int orb_gray_rgb_order, uint8_t* dyna_out /* (n-2) x h x w or NULL */, int* nkp_out /* n-2 or NULL */,
                        OrcKp* kps_out /* (n-2) x kp_cap or NULL */, int kp_cap, int skip_first /* untimed warm-up pairs */, int flow_max_levels) {
    const size_t fb = (size_t)w * h * 3, fd = (size_t)w * h;
    DynaDetect dd(wrap8(bgr + fb, w, h, 3), wrap8(bgr, w, h, 3), fx, fy, cx, cy, depthScale);
    dd.flow_max_levels = flow_max_levels;
    ORBextractor orb(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), tf = 0, tt = 0, to = 0;
    if (skip_first < 0) skip_first = 0;
    for (int i = 2; i < n; i++) {
        if (i - 2 == skip_first) { t0 = now(); tf = tt = to = 0; for (double& q : dd.t_stage) q = 0; }       // warm-up pairs end here
        Img8 img = wrap8(bgr + fb * i, w, h, 3); Img16 dp = wrap16(depth + fd * i, w, h);
        Img8 dy, lb, dil, gray; std::vector<KeyPoint> k; std::vector<uint8_t> d;
        double a = now();
        dd.imgRGB = img; bgr2gray(dd.imgRGB, dd.imgGray); bgr2gray(dd.imgRGBLast, dd.imgGrayLast); bgr2gray(dd.imgRGBLastLast, dd.imgGrayLastLast);
        dd.ComputeDenseFlow();
        double b = now();
        dd.skip_flow = true; dd.DetectDynaArea(img, dp, dy, lb); dd.skip_flow = false;
        dilate_ellipse15(dy, dil);
        double c = now();
        bgr2gray(img, gray, orb_gray_rgb_order != 0); orb.extract(gray, dil, k, d);
        double e = now();
        if (dyna_out) std::memcpy(dyna_out + fd * (i - 2), dy.d.data(), fd);
        if (nkp_out) nkp_out[i - 2] = (int)k.size();
        if (kps_out) put_kps(k, kps_out + (size_t)(i - 2) * kp_cap, kp_cap);
        tf += b - a; tt += c - b; to += e - c;
    }
    if (stage_seconds) { stage_seconds[0] = tf; stage_seconds[1] = tt; stage_seconds[2] = to;
        for (int k = 0; k < 5; k++) stage_seconds[3 + k] = dd.t_stage[k];
        stage_seconds[3 + 3] += tf;                       /* the dense flow was computed outside DetectDynaArea (skip_flow) */
        stage_seconds[8] = to; }
    return now() - t0;
}

}  // extern "C"
