// DynaDetect engine: orchestration of the HIP stages and the serial host stages.
// Reference walk: ORB_SLAM2/src/DynaDetect.cc DetectDynaArea :1377-1666, DetectDynaByDenseOpticalFLow :1023-1374,
// SegByKmeans :315-420, CalOccluded :429-642, SegAndMergeV2 :653-1018, cal_hist :1685-1739.
// Per-pixel work runs in flow_kernels.hip / depth_kernels.hip; contour / graph / flood-fill logic is serial, tiny and
// order-defined and runs on bit-packed masks on the host (DESIGN.md "host stages").  No CPU fallback exists for any
// device stage: every launch error is returned to the caller.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <atomic>
#include <thread>
#include "dyna.hpp"
#include "host/statehash.hpp"
#include <chrono>
#include "host/rng.hpp"

namespace sind {

static inline double tick_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ======================================================================================================= front
int DynaFront::init(const DynaConfig& c, int maxB_, hipStream_t s) {
    cfg = c; maxB = maxB_; stream = s;
    const float scale_element = 0.6f;                                   // DD:1033
    fw = (int)(scale_element * c.W); fh = (int)(scale_element * c.H);
    SIND_TRY(flow.init(fw, fh, maxB, s));
    const size_t nf = (size_t)fw * fh * maxB;
    SIND_TRY(g0.alloc(nf)); SIND_TRY(g1.alloc(nf)); SIND_TRY(u.alloc(nf)); SIND_TRY(v.alloc(nf)); SIND_TRY(u2.alloc(nf)); SIND_TRY(v2.alloc(nf)); SIND_TRY(mag.alloc(nf));
    SIND_TRY(maxbits.alloc(maxB)); SIND_TRY(hist.alloc((size_t)maxB * 256)); SIND_TRY(idx_dev.alloc(maxB));
    return SIND_OK;
}

int DynaFront::gray_and_min(const uint8_t* bgr, int n, uint8_t* gray, uint8_t* grayMin) {
    const size_t np = (size_t)cfg.W * cfg.H;
    SIND_TRY(launch_bgr2gray(stream, bgr, gray, np * n, false));
    SIND_TRY(launch_resize_u8(stream, gray, grayMin, cfg.W, cfg.H, fw, fh, n, cfg.W, fw, np, (size_t)fw * fh));
    return SIND_OK;
}

int DynaFront::gather(const uint8_t* pool, const int* idx, int B, uint8_t* out) {
    const size_t fb = (size_t)fw * fh;
    if (fb % 16 == 0) return launch_gather_frames(stream, pool, idx, B, out, fb);
    for (int b = 0; b < B; b++) HIP_TRY(hipMemcpyAsync(out + fb * b, pool + fb * idx[b], fb, hipMemcpyDeviceToDevice, stream));
    return SIND_OK;
}

int DynaFront::dense_flow(const uint8_t* pool, const int* cur, const int* prev1, const int* prev2, int B, float* U, float* V, int* large_motion,
                          float* dbg_du, float* dbg_dv, float* dbg_ru, float* dbg_rv) {
    if (B < 1 || B > maxB) { sind_set_error("dense_flow: batch %d outside [1,%d]", B, maxB); return SIND_E_ARG; }
    const int nf = fw * fh; const size_t nfB = (size_t)nf * B;
    // pass 1: flow(n, n-2) for every pair (DD:1075); with `speculate` the candidates of pass 2 ride along as pairs B .. 2 B - 1
    const bool spec = speculate && 2 * B <= maxB;
    if (spec) {
        std::vector<int> c2(2 * B), p2x(2 * B);
        for (int b = 0; b < B; b++) { c2[b] = c2[B + b] = cur[b]; p2x[b] = prev2[b]; p2x[B + b] = prev1[b]; }
        SIND_TRY(gather(pool, c2.data(), 2 * B, g0.p)); SIND_TRY(gather(pool, p2x.data(), 2 * B, g1.p));
        SIND_TRY(flow.deepflow(g0.p, g1.p, 2 * B, u.p, v.p));
    } else {
        SIND_TRY(gather(pool, cur, B, g0.p)); SIND_TRY(gather(pool, prev2, B, g1.p));
        SIND_TRY(flow.deepflow(g0.p, g1.p, B, u.p, v.p));
    }
    // large-motion test on |flow| (DD:1081-1114): max, u8 normalisation, histogram on the GPU, percentile test here
    SIND_TRY(launch_mag_stats(stream, u.p, v.p, mag.p, maxbits.p, hist.p, nullptr, nf, B));
    std::vector<unsigned> h_max(B); std::vector<int> h_hist((size_t)B * 256);
    HIP_TRY(hipMemcpyAsync(h_max.data(), maxbits.p, B * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_hist.data(), hist.p, (size_t)B * 256 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    std::vector<int> flagged; std::vector<int> lm(B, 0);
    const float scale_element = 0.6f;
    for (int b = 0; b < B; b++) {
        float maxFlowf; std::memcpy(&maxFlowf, &h_max[b], 4);
        const double maxFlow = maxFlowf;
        const int endFlow = (int)(10.0f * scale_element * 255.0f / maxFlow);
        int endFlow2 = 0; float ratio = 0.0f;
        const float totalpixel = cfg.W * cfg.H * scale_element * scale_element;
        for (int i = 0; i < 255; ++i) { ratio += (float)h_hist[(size_t)b * 256 + i]; if (ratio > 0.3f * totalpixel) { endFlow2 = i; break; } }
        if (endFlow2 > endFlow) { lm[b] = 1; flagged.push_back(b); }
    }
    if (large_motion) std::copy(lm.begin(), lm.end(), large_motion);
    // pass 2: flow(n, n-1) for the flagged pairs only (DD:1121-1131)
    if (!flagged.empty() && spec) {
        for (int b : flagged) {
            HIP_TRY(hipMemcpyAsync(u.p + (size_t)nf * b, u.p + (size_t)nf * (B + b), nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(v.p + (size_t)nf * b, v.p + (size_t)nf * (B + b), nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
    } else if (!flagged.empty()) {
        const int B2 = (int)flagged.size();
        std::vector<int> c2(B2), p2(B2);
        for (int k = 0; k < B2; k++) { c2[k] = cur[flagged[k]]; p2[k] = prev1[flagged[k]]; }
        SIND_TRY(gather(pool, c2.data(), B2, g0.p)); SIND_TRY(gather(pool, p2.data(), B2, g1.p));
        SIND_TRY(flow.deepflow(g0.p, g1.p, B2, u2.p, v2.p));
        for (int k = 0; k < B2; k++) {
            HIP_TRY(hipMemcpyAsync(u.p + (size_t)nf * flagged[k], u2.p + (size_t)nf * k, nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(v.p + (size_t)nf * flagged[k], v2.p + (size_t)nf * k, nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
    }
    SIND_TRY(launch_scale2(stream, u.p, v.p, -1.0f, nfB));                                   // imgDenseFlow *= -1 (DD:1080, 1129)
    if (dbg_du) { HIP_TRY(hipMemcpyAsync(dbg_du, u.p, nfB * 4, hipMemcpyDeviceToHost, stream)); HIP_TRY(hipMemcpyAsync(dbg_dv, v.p, nfB * 4, hipMemcpyDeviceToHost, stream)); }
    // refinement against the frame actually used (DD:1133-1143)
    std::vector<int> sel(B); for (int b = 0; b < B; b++) sel[b] = lm[b] ? prev1[b] : prev2[b];
    SIND_TRY(gather(pool, cur, B, g0.p)); SIND_TRY(gather(pool, sel.data(), B, g1.p));
    SIND_TRY(flow.refine(g0.p, g1.p, B, u.p, v.p));
    if (dbg_ru) { HIP_TRY(hipMemcpyAsync(dbg_ru, u.p, nfB * 4, hipMemcpyDeviceToHost, stream)); HIP_TRY(hipMemcpyAsync(dbg_rv, v.p, nfB * 4, hipMemcpyDeviceToHost, stream)); }
    // resize to the full frame and undo the 0.6 scale (DD:1144-1147)
    const float inv = 1.0f / scale_element;
    SIND_TRY(launch_resize_f32(stream, u.p, U, fw, fh, cfg.W, cfg.H, B, inv, true));
    SIND_TRY(launch_resize_f32(stream, v.p, V, fw, fh, cfg.W, cfg.H, B, inv, true));
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}

// ======================================================================================================= tail
int DynaTail::init(const DynaConfig& c, hipStream_t s) {
    cfg = c; stream = s; W = c.W; H = c.H; N = W * H;
    invDepthScale = 1.0f / cfg.depthScale; zInvalidFrom = 65536;
    for (int d = 65535; d >= 0; d--) { if ((float)(uint16_t)d / cfg.depthScale >= (float)(uint16_t)6) zInvalidFrom = d; else break; }
    if (W % 64 != 0 || W % 8 != 0 || H % 8 != 0) { sind_set_error("DynaTail: width must be a multiple of 64 and height of 8 (got %dx%d)", W, H); return SIND_E_ARG; }
    reset();
    for (int l = 1; l < 4; l++) SIND_TRY(dpyr[l].alloc((size_t)N >> (2 * l)));
    for (int l = 0; l < 4; l++) SIND_TRY(lab[l].alloc((size_t)N >> (2 * l)));
    SIND_TRY(filt.alloc(N)); SIND_TRY(px.alloc(N)); SIND_TRY(py.alloc(N)); SIND_TRY(pz.alloc(N)); SIND_TRY(lab8.alloc(N)); SIND_TRY(labPrev8.alloc(N));
    SIND_TRY(edge.alloc(N)); SIND_TRY(edgeTmp.alloc(N)); SIND_TRY(total.alloc(N)); SIND_TRY(depthN.alloc(N)); SIND_TRY(occ2_d.alloc(N));
    SIND_TRY(magu8.alloc(N)); SIND_TRY(low_d.alloc(((size_t)2 * N + 15) / 16 * 16 + 1088)); SIND_TRY(mag.alloc(N));       // low_d: low mask, high mask, then (16-byte aligned) the thresholds' 261-word result block: ONE D2H for the stage
    SIND_TRY(kpart.alloc((size_t)(KM_MAX_BLOCKS * 4 + 1) * KM_K + 64));       // count table, totals row, the 36 sums
    SIND_TRY(kcomp.alloc((size_t)3 * N + 8)); /* + 8: k_km_seqsum's window loads may touch up to 7 floats behind the last run */ SIND_TRY(umax_d.alloc(2));
    SIND_TRY(h_grid.alloc((size_t)2 * ((W - 1) / 10) * ((H - 1) / 10) + 2)); SIND_TRY(h_hist.alloc(261)); SIND_TRY(h_ab.alloc(((size_t)2 * N + 15) / 16 * 16 + 1088)); SIND_TRY(h_lab8.alloc(N));
    SIND_TRY(h_kstate.alloc(4)); SIND_TRY(h_blocks.alloc((size_t)(W / 16) * (H / 16)));
    { // RAG workspaces for up to 64 pieces up front: a later (re)allocation synchronises the whole device, i.e. waits for the
      // flow solver of the next step when tails and dense flow overlap
      const size_t cap = 64, pw = (size_t)H * (W / 64), nr = 3 * cap * cap + cap + cap * 256;
      SIND_TRY(h_planes.alloc(3 * cap * pw)); SIND_TRY(planes_d.alloc(3 * cap * pw)); SIND_TRY(h_rag.alloc(nr)); SIND_TRY(rag_d.alloc(nr)); }
    SIND_TRY(kstate.alloc(4)); SIND_TRY(depth_fix.alloc(N));
    SIND_TRY(hist_d.alloc(264 + 261));       // working block [0..256] = residual histogram + maximum (float bits); result block at 264: the same + thresholds lo, hi, otsu, triangle (floats)
    SIND_TRY(grid_d.alloc((size_t)2 * ((W - 1) / 10) * ((H - 1) / 10) + 2)); SIND_TRY(blocks_d.alloc((size_t)(W / 16) * (H / 16)));
    // every kernel operand of the tail must exist before the first launch (a missing workspace would be a wild device write)
    const void* ws[] = {dpyr[1].p, dpyr[2].p, dpyr[3].p, lab[0].p, lab[1].p, lab[2].p, lab[3].p, filt.p, px.p, py.p, pz.p, lab8.p, labPrev8.p, edge.p, edgeTmp.p, total.p,
                        depthN.p, occ2_d.p, magu8.p, low_d.p, mag.p, kpart.p, kcomp.p, umax_d.p, h_grid.p, h_hist.p, h_ab.p, h_lab8.p, h_kstate.p, h_blocks.p, h_planes.p, planes_d.p,
                        h_rag.p, rag_d.p, kstate.p, depth_fix.p, hist_d.p, grid_d.p, blocks_d.p};
    for (const void* q : ws) if (!q) { sind_set_error("DynaTail::init: a workspace was not allocated"); return SIND_E_STATE; }
    return SIND_OK;
}
void DynaTail::reset() { dynaLast.assign(N, 0); labelLast.assign(N, 0); kmLabelLast.assign(N, 0); highLast.create(W, H); kmLabelLastAny = false; std::memset(lastCnt, 0, sizeof(lastCnt)); std::memset(lastDyn, 0, sizeof(lastDyn)); }

void DynaTail::save_state(uint8_t* buf, bool flow_half, bool depth_half) const {
    uint8_t* q = buf;
    if (flow_half) { std::memcpy(q, dynaLast.data(), N); std::memcpy(q + N, labelLast.data(), N); highLast.to_u8(q + 2 * (size_t)N, W, 255);
                     std::memcpy(q + 3 * (size_t)N, lastCnt, sizeof(lastCnt)); std::memcpy(q + 3 * (size_t)N + sizeof(lastCnt), lastDyn, sizeof(lastDyn)); }
    q += 3 * (size_t)N + sizeof(lastCnt) + sizeof(lastDyn);
    if (depth_half) { std::memcpy(q, kmLabelLast.data(), N); const int any = kmLabelLastAny ? 1 : 0; std::memcpy(q + N, &any, sizeof(int)); }
}
void DynaTail::load_state(const uint8_t* buf, bool flow_half, bool depth_half) {
    const uint8_t* q = buf;
    if (flow_half) { std::memcpy(dynaLast.data(), q, N); std::memcpy(labelLast.data(), q + N, N); highLast = BitImg::from_u8(q + 2 * (size_t)N, W, H, W);
                     std::memcpy(lastCnt, q + 3 * (size_t)N, sizeof(lastCnt)); std::memcpy(lastDyn, q + 3 * (size_t)N + sizeof(lastCnt), sizeof(lastDyn)); }
    q += 3 * (size_t)N + sizeof(lastCnt) + sizeof(lastDyn);
    if (depth_half) { std::memcpy(kmLabelLast.data(), q, N); int any = 0; std::memcpy(&any, q + N, sizeof(int)); kmLabelLastAny = any != 0; }
}

// ---- DD:1163-1367: sample weights -> PROSAC pairs -> homography -> residual -> Otsu / Triangle thresholds -> masks
int DynaTail::flow_masks(const float* U, const float* V, BitImg& low, BitImg& high, const float* gridFlowPre) {
    const int numCluster = KM_K;
    const int gx = (W - 1) / 10, gy = (H - 1) / 10;
    const float* gridFlow = gridFlowPre;                      // the pipeline gathers the sample grid of all frames right after the dense flow
    if (!gridFlow) {
        SIND_TRY(launch_gather_grid(stream, U, V, grid_d.p, W, H, 10));
        HIP_TRY(hipMemcpyAsync(h_grid.p, grid_d.p, (size_t)2 * gx * gy * sizeof(float), hipMemcpyDeviceToHost, stream));
        gridFlow = h_grid.p;
    }
    double tq = tick_ms();
    #define QLAP(i) { const double t_ = tick_ms(); t_fine[i] += t_ - tq; tq = t_; }
    // previous-frame dynamic ratio per cluster (DD:1169-1177)
    std::vector<float> clusterWeight(numCluster, 0.0f);
    for (int i = 1; i < numCluster; i++) clusterWeight[i] = (float)lastDyn[i] / (float)(lastCnt[i] + 1.0f);       // counts of the previous frame's labels (see process())
    struct PW { int x, y; float weight; };
    std::vector<PW> pts; pts.reserve((size_t)gx * gy);
    CvRng rng(12345);
    for (int row = 10; row < H; row += 10) for (int col = 10; col < W; col += 10) {
        const float randomd = (float)rng.gaussian(0.5);
        const uint8_t dl = dynaLast[(size_t)row * W + col];
        if (dl < 20) pts.push_back({col, row, randomd + 1.0f});
        else if ((unsigned)(dl - 20) <= 230 - 20) { const int label = labelLast[(size_t)row * W + col]; pts.push_back({col, row, randomd + 1.2f * (1.0f - clusterWeight[label])}); }
        else pts.push_back({col, row, randomd + 0.4f});
    }
    QLAP(20)
    std::sort(pts.begin(), pts.end(), [](const PW& a, const PW& b) { return a.weight > b.weight; });
    if (!gridFlowPre) HIP_TRY(sind_stream_wait(stream));
    QLAP(21)
    std::vector<Pt2f> in, inLast;
    for (const PW& p : pts) {
        const int gi = (p.y / 10 - 1) * gx + (p.x / 10 - 1);
        const float ptCol = (float)p.x, ptRow = (float)p.y, fxv = gridFlow[2 * gi], fyv = gridFlow[2 * gi + 1];
        const int r = (int)(ptRow - fyv), c = (int)(ptCol - fxv);
        if ((unsigned)r <= (unsigned)H && (unsigned)c <= (unsigned)W) { in.push_back({ptCol, ptRow}); inLast.push_back({ptCol - fxv, ptRow - fyv}); }
    }
    double Hm[9];
    tq = tick_ms();
    find_homography_rho(in, inLast, Hm);
    QLAP(22)
    // working histogram hist_d[0..256] (+ maximum): zero whenever the previous frame's threshold kernel completed (it clears what it read); a frame that
    // failed in between leaves it unknown, hence the flag
    SIND_TRY(launch_residual(stream, U, V, Hm, mag.p, (unsigned*)(hist_d.p + 256), hist_d.p, magu8.p, W, H, hist_clean));
    hist_clean = false;
    // thresholds (Otsu / triangle + the clamping of DD:1309-1367) and the two masks on the device: one host round trip for the stage
    // result block (histogram, maximum, four thresholds) right behind the two masks: it travels with them (a copy of 1 KB of its own was a blit kernel and a launch per frame)
    const size_t res_off = ((size_t)2 * N + 15) / 16 * 16;
    int* res = reinterpret_cast<int*>(low_d.p + res_off);
    SIND_TRY(launch_flow_thresholds_and_masks(stream, hist_d.p, W, H, res, magu8.p, low_d.p, (low_d.p + N)));
    HIP_TRY(hipMemcpyAsync(h_ab.p, low_d.p, res_off + 261 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    hist_clean = true;
    {
        const int* hh = reinterpret_cast<const int*>(h_ab.p + res_off);
        float f[5]; std::memcpy(&f[0], hh + 256, 4); std::memcpy(&f[1], hh + 257, 16);
        dbg.maxError = f[0]; dbg.thr_low = f[1]; dbg.thr_high = f[2]; dbg.otsu = f[3]; dbg.triangle = f[4];
        dbg.nPairs = (int)in.size(); std::copy(Hm, Hm + 9, dbg.H); std::copy(hh, hh + 256, dbg.hist);
    }
    tq = tick_ms();
    low = BitImg::from_u8(h_ab.p, W, H, W); high = BitImg::from_u8((h_ab.p + N), W, H, W);
    QLAP(23)
    #undef QLAP
    if (keep_debug) { dbg.maskLow.assign(h_ab.p, h_ab.p + N); dbg.maskHigh.assign((h_ab.p + N), (h_ab.p + N) + N); }
    return SIND_OK;
}

// ---- DD:315-420: 4-level k-means, K = 12, criteria (EPS+COUNT, 4, 0.07), KMEANS_USE_INITIAL_LABELS.
// The whole loop is enqueued without a host round trip: the centre step, empty-cluster repair and the stop test of cv::kmeans
// run in k_km_update (one workgroup) on the device; one D2H of the level-0 state + labels at the end.
// The ~50 launches of a frame's k-means are recorded once per tail as a HIP graph (two variants: grid labels for the very first
// frame, previous-frame labels afterwards) and replayed with one hipGraphLaunch: the packets reach the queue in one go instead of
// one host submission per kernel.  The graph reads the depth frame from a fixed per-tail buffer.
int DynaTail::kmeans_enqueue(const uint16_t* depth0, bool prevLabels) {
    const float scales[4] = {1.0f, 0.5f, 0.25f, 0.125f};
    const uint16_t* dl[4] = {depth0, dpyr[1].p, dpyr[2].p, dpyr[3].p};
    for (int l = 1; l < 4; l++) SIND_TRY(launch_depth_half(stream, dl[l - 1], dpyr[l].p, W >> l, H >> l));
    for (int level = 3; level >= 0; level--) {
        const int hp = (int)(H * scales[level]), wp = (int)(W * scales[level]), n = hp * wp;
        SIND_TRY(launch_points(stream, dl[level], px.p, py.p, pz.p, wp, hp, scales[level], cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale));
        if (level == 3) { if (!prevLabels) SIND_TRY(launch_labels_grid(stream, lab[3].p, wp, hp)); else SIND_TRY(launch_labels_resize_u8(stream, labPrev8.p, lab[3].p, W, H, wp, hp)); }
        else SIND_TRY(launch_labels_resize_i32(stream, lab[level + 1].p, lab[level].p, wp / 2, hp / 2, wp, hp));
        SIND_TRY(launch_kmeans_level(stream, km_fuse, px.p, py.p, pz.p, lab[level].p, n, kpart.p, kcomp.p, kstate.p + level, 4, 0.07 * 0.07));
    }
    SIND_TRY(launch_labels_to_u8(stream, lab[0].p, lab8.p, N));
    return SIND_OK;
}
int DynaTail::kmeans(const uint16_t* depth_dev, std::vector<uint8_t>& label8, float centers[KM_K][3], int counts[KM_K]) {
    static const bool use_graph = !(sind_lab_env("SIND_KM_GRAPH") && atoi(sind_lab_env("SIND_KM_GRAPH")) == 0);
    if (kmLabelLastAny) { std::memcpy(h_lab8.p, kmLabelLast.data(), N); HIP_TRY(hipMemcpyAsync(labPrev8.p, h_lab8.p, N, hipMemcpyHostToDevice, stream)); }
    bool graphed = false;
    if (use_graph && !kmGraphBroken) {
        const int v = kmLabelLastAny ? 1 : 0;
        if (!kmGraph[v]) {                       // record once; any failure here only means "launch kernel by kernel from now on" (still the GPU path)
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = kmeans_enqueue(depth_fix.p, v != 0);
                const hipError_t ec = hipStreamEndCapture(stream, &g);
                if (rc != SIND_OK || ec != hipSuccess || hipGraphInstantiate(&kmGraph[v], g, nullptr, nullptr, 0) != hipSuccess) { kmGraph[v] = nullptr; kmGraphBroken = true; }
                if (g) (void)hipGraphDestroy(g);
            } else kmGraphBroken = true;
            if (kmGraphBroken) { const hipError_t le = hipGetLastError(); fprintf(stderr, "[sind] k-means HIP graph unavailable (%s); launching the chain kernel by kernel\n", hipGetErrorString(le)); }
        }
        if (!kmGraphBroken) {
            HIP_TRY(hipMemcpyAsync(depth_fix.p, depth_dev, (size_t)N * sizeof(uint16_t), hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipGraphLaunch(kmGraph[v], stream));
            graphed = true;
        }
    }
    if (!graphed) SIND_TRY(kmeans_enqueue(depth_dev, kmLabelLastAny));
    label8.resize(N);
    HIP_TRY(hipMemcpyAsync(h_kstate.p, kstate.p, 4 * sizeof(KmState), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_ab.p, lab8.p, N, hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    const KmState* st = h_kstate.p; std::memcpy(label8.data(), h_ab.p, N);
    std::memcpy(centers, st[0].ctr, sizeof(st[0].ctr)); std::memcpy(counts, st[0].cnt, sizeof(st[0].cnt));
    return SIND_OK;
}

// ---- the same chain for one frame of every stream (see dyna.hpp)
int KMeansBatch::init(const DynaConfig& c, int maxB_, hipStream_t s) {
    cfg = c; W = c.W; H = c.H; N = W * H; maxB = maxB_; stream = s;
    const size_t B = (size_t)maxB;
    for (int l = 1; l < 4; l++) SIND_TRY(dpyr[l].alloc(((size_t)N >> (2 * l)) * B));
    for (int l = 0; l < 4; l++) SIND_TRY(lab[l].alloc(((size_t)N >> (2 * l)) * B));
    SIND_TRY(px.alloc((size_t)N * B)); SIND_TRY(py.alloc((size_t)N * B)); SIND_TRY(pz.alloc((size_t)N * B)); SIND_TRY(comp.alloc((size_t)3 * N * B + 8));
    SIND_TRY(seg.alloc((size_t)KM_SEG_WORDS * B)); SIND_TRY(use_prev_d.alloc(B)); SIND_TRY(labPrev8.alloc((size_t)N * B)); SIND_TRY(lab8.alloc((size_t)N * B)); SIND_TRY(kstate.alloc(4 * B));
    SIND_TRY(h_prev.alloc((size_t)N * B)); SIND_TRY(h_lab8.alloc((size_t)N * B)); SIND_TRY(h_state.alloc(4 * B)); SIND_TRY(h_use_prev.alloc(B));
    res.resize(maxB);
    return SIND_OK;
}
int KMeansBatch::run(const uint16_t* depth_base, size_t depth_stride, int B, const uint8_t* const* prev) {
    if (B < 1 || B > maxB) { sind_set_error("KMeansBatch::run: batch %d outside [1,%d]", B, maxB); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(cfg.device));
    bool any_prev = false;
    for (int b = 0; b < B; b++) { h_use_prev.p[b] = prev[b] ? 1 : 0; if (prev[b]) { std::memcpy(h_prev.p + (size_t)N * b, prev[b], N); any_prev = true; } }
    HIP_TRY(hipMemcpyAsync(use_prev_d.p, h_use_prev.p, B * sizeof(int), hipMemcpyHostToDevice, stream));
    if (any_prev) HIP_TRY(hipMemcpyAsync(labPrev8.p, h_prev.p, (size_t)N * B, hipMemcpyHostToDevice, stream));
    const float scales[4] = {1.0f, 0.5f, 0.25f, 0.125f};
    const uint16_t* dl[4] = {depth_base, dpyr[1].p, dpyr[2].p, dpyr[3].p};
    const size_t ds[4] = {depth_stride, (size_t)N >> 2, (size_t)N >> 4, (size_t)N >> 6};
    for (int l = 1; l < 4; l++) SIND_TRY(launch_depth_half(stream, dl[l - 1], dpyr[l].p, W >> l, H >> l, B, ds[l - 1], ds[l]));
    for (int level = 3; level >= 0; level--) {
        const int hp = (int)(H * scales[level]), wp = (int)(W * scales[level]), n = hp * wp; const size_t ls = (size_t)N >> (2 * level);
        SIND_TRY(launch_points(stream, dl[level], px.p, py.p, pz.p, wp, hp, scales[level], cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale, B, ds[level], N));
        if (level == 3) { SIND_TRY(launch_labels_grid(stream, lab[3].p, wp, hp, B, ls, use_prev_d.p)); SIND_TRY(launch_labels_resize_u8(stream, labPrev8.p, lab[3].p, W, H, wp, hp, B, N, ls, use_prev_d.p)); }
        else SIND_TRY(launch_labels_resize_i32(stream, lab[level + 1].p, lab[level].p, wp / 2, hp / 2, wp, hp, B, (size_t)N >> (2 * (level + 1)), ls));
        SIND_TRY(launch_kmeans_level(stream, km_fuse, px.p, py.p, pz.p, lab[level].p, n, seg.p, comp.p, kstate.p + level, 4, 0.07 * 0.07, B, N, ls, KM_SEG_WORDS, (size_t)3 * N, 4));
    }
    SIND_TRY(launch_labels_to_u8(stream, lab[0].p, lab8.p, N, B, N, N));
    HIP_TRY(hipMemcpyAsync(h_state.p, kstate.p, (size_t)4 * B * sizeof(KmState), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_lab8.p, lab8.p, (size_t)N * B, hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    for (int b = 0; b < B; b++) { const KmState& st = h_state.p[4 * b];
        res[b].label8 = h_lab8.p + (size_t)N * b; std::memcpy(res[b].centers, st.ctr, sizeof(st.ctr)); std::memcpy(res[b].counts, st.cnt, sizeof(st.cnt)); }
    return SIND_OK;
}

// ---- DD:429-642: depth-gradient edges (GPU), end points, PEAC plane contours, plane-edge filtering (host).  Two host halves around the PEAC region grow, which runs
// on the GPU (peac_kernels.hip): first half = GPU stencils (unless the batch produced them), packing, end points, PEAC graph clustering and the grow's input block;
// second half = PEAC's last merge + plane contours, contour filter, closing.  grow_block == nullptr: the half grows on the host right away (no GPU grow wanted).
int DynaTail::cal_occluded_p1(const uint16_t* depth_host, const uint16_t* depth_dev, OccCtx& c, const OccGpuOut* pre, uint8_t* grow_block, int depth_index) {
    double tf = tick_ms();
    #define FLAP(i) { const double t_ = tick_ms(); t_fine[i] += t_ - tf; tf = t_; }
    const uint8_t* edge_h = h_ab.p; const uint8_t* total_h = h_ab.p + N; const PeacBlockStats* blocks_h = h_blocks.data();
    if (pre) { edge_h = pre->edge; total_h = pre->total; blocks_h = pre->blocks; }      // the GPU half ran for all frames of the step at once (OccBatch)
    else {
    SIND_TRY(launch_median5(stream, depth_dev, filt.p, W, H));
    SIND_TRY(launch_max_u16(stream, filt.p, N, umax_d.p));
    SIND_TRY(launch_grad_edge(stream, filt.p, umax_d.p, edge.p, total.p, W, H, cfg.depthScale));
    SIND_TRY(launch_morph(stream, edge.p, edgeTmp.p, W, H, 4, false));      // MORPH_OPEN, element4 = erode then dilate
    SIND_TRY(launch_morph(stream, edgeTmp.p, edge.p, W, H, 4, true));
    SIND_TRY(launch_peac_block_stats(stream, depth_dev, W, H, 16, 16, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale, blocks_d.p));
    PinnedBuf<PeacBlockStats>& blocks = h_blocks;
    HIP_TRY(hipMemcpyAsync(h_ab.p, edge.p, N, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync((h_ab.p + N), total.p, N, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(blocks.data(), blocks_d.p, blocks.n * sizeof(PeacBlockStats), hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    }
    FLAP(0)
    c.occ = BitImg::from_u8(edge_h, W, H, W);
    c.totalArea = BitImg::from_u8(total_h, W, H, W);
    if (keep_debug) dbg.gradEdge.assign(edge_h, edge_h + N);
    FLAP(1)
    // end points: edge pixels with at most 4 of the 12 radius-2 ring pixels set (DD:498-532), greedy NMS radius 6 in scan order
    static const int ring[12][2] = {{0,-2},{1,-2},{2,-1},{2,0},{2,1},{1,2},{0,2},{-1,2},{-2,1},{-2,0},{-2,-1},{-1,-2}};
    const BitImg& occ = c.occ;
    std::vector<PtI>& endPoints = c.endPoints; endPoints.clear();
    for (int row = 3; row < H - 3; ++row) {                                     // (edge pixels are sparse: walk the set bits of the row's words, in column order)
        const uint64_t* r = occ.row(row);
        for (int k = 0; k < occ.wpr; k++) for (uint64_t bits = r[k]; bits; bits &= bits - 1) {
            const int col = (k << 6) + __builtin_ctzll(bits);
            if (col < 3 || col >= W - 3) continue;
            int s = 0; for (int i = 0; i < 12; i++) s += occ.get(col + ring[i][0], row + ring[i][1]);
            if (s <= 4) endPoints.push_back({col, row});
        }
    }
    { std::vector<PtI> sel; for (const PtI& e : endPoints) { bool ov = false; for (const PtI& q : sel) { const int dx = e.x - q.x, dy = e.y - q.y; if ((float)dx * dx + dy * dy < 6.0f * 6.0f) { ov = true; break; } } if (!ov) sel.push_back(e); } endPoints.swap(sel); }
    FLAP(2)
    // PEAC (DD:558-593), first part: graph clustering, planes, seeds of the region grow
    PeacInput pin{blocks_h, depth_host, W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale};
    c.fit.reset(new PeacFitter(pin)); c.fit->part1();
    c.grown_on_host = false;
    if (grow_block) peac_grow_pack(*c.fit, depth_index, grow_block);
    if (!grow_block || !c.fit->gpu_ok) { c.fit->grow_host(c.m8, c.m16, c.pairs); c.grown_on_host = true; }      // beyond the kernel's capacities (or no GPU grow asked for)
    FLAP(3)
    #undef FLAP
    return SIND_OK;
}
int DynaTail::cal_occluded_p2(OccCtx& c, const int8_t* member8, const uint8_t* pair_seen, const int* grow_status, BitImg& occ1, BitImg& occ2) {
    double tf = tick_ms();
    #define FLAP(i) { const double t_ = tick_ms(); t_fine[i] += t_ - tf; tf = t_; }
    BitImg planeC;
    if (!c.fit) { sind_set_error("CalOccluded, second half: the frame has no first half (its PEAC fitter is missing)"); return SIND_E_STATE; }
    if (!c.grown_on_host && !(grow_status && grow_status[0] == PG_OK && member8 && pair_seen)) {
        // the kernel reported a capacity overflow for this frame (status 1..3): the host statement of the same FIFO takes over
        c.fit->grow_host(c.m8, c.m16, c.pairs); c.grown_on_host = true; n_grow_fallback++;
    }
    if (c.grown_on_host) c.fit->part2(c.m8.empty() ? nullptr : c.m8.data(), c.m16.empty() ? nullptr : c.m16.data(), c.pairs.data(), planeC);
    else c.fit->part2(member8, nullptr, pair_seen, planeC);
    c.fit.reset(); c.m8.clear(); c.m16.clear();
    FLAP(3)
    const BitImg& occ = c.occ;
    BitImg edgeByPlane = planeC; edgeByPlane.andnot(occ);                       // DD:599
    std::vector<Contour> contours; find_contours(edgeByPlane, contours, true);
    BitImg acc(W, H);
    const EllipseElem e10(10), e7(7), e3(3);
    for (const Contour& k : contours) {
        if (k.size() < 25) continue;
        BitImg one(W, H); draw_thick2(one, k);
        const Rect bb = contour_bbox(k);
        // "an end point lies in the 10 x 10 dilation of the contour" is asked of the contour itself (the element's window around the end point): the dilation and the
        // erosion are only made for the contours that pass (most do not: the stage was 0.6 ms of a 640 x 480 frame's host time, 1.7 ms at 1280 x 720)
        bool isEnd = false;
        for (const PtI& e : c.endPoints) if (e.y >= bb.y0 - 1 - e10.n && e.y <= bb.y1 + 1 + e10.n && e.x >= bb.x0 - 1 - e10.n && e.x <= bb.x1 + 1 + e10.n && one.dilation_hits(e10, e.x, e.y)) { isEnd = true; break; }
        if (isEnd) { one = one.dilated(e10, bb.y0 - 1, bb.y1 + 1); acc |= one.eroded_rows(e7, bb.y0 - 7, bb.y1 + 7); }
    }
    FLAP(4)
    occ2 = acc;
    BitImg u = occ; u |= acc; occ1 = u.closed(e3);
    FLAP(5)
    #undef FLAP
    if (keep_debug) { dbg.planeContours.resize(N); planeC.to_u8(dbg.planeContours.data(), W, 255); }
    return SIND_OK;
}
// one frame on this tail's own stream: both halves with a one-frame launch of the grow kernel in between
int DynaTail::cal_occluded(const uint16_t* depth_host, const uint16_t* depth_dev, BitImg& totalArea, BitImg& occ1, BitImg& occ2, const OccGpuOut* pre) {
    // sizes the grow kernel does not take (PeacGrowBatch::supports) are grown on the host in the first half, like before the kernel existed
    const bool gpu_grow = PeacGrowBatch::supports(W, H);
    if (gpu_grow && !own_grow) {
        std::unique_ptr<OwnGrow> g(new OwnGrow());             // assigned only when every piece exists: a failed init must not leave a half-built workspace behind
        SIND_TRY(g->batch.init(W, H, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale, 1));
        SIND_TRY(g->in_h.alloc(PG_IN_STRIDE)); SIND_TRY(g->member_h.alloc(N)); SIND_TRY(g->pair_h.alloc((size_t)PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES)); SIND_TRY(g->status_h.alloc(4));
        own_grow = std::move(g);
    }
    OccCtx c;
    SIND_TRY(cal_occluded_p1(depth_host, depth_dev, c, pre, gpu_grow ? own_grow->in_h.p : nullptr, 0));
    if (!c.grown_on_host) {
        const double t0 = tick_ms();
        SIND_TRY(own_grow->batch.run(stream, own_grow->in_h.p, depth_dev, 1, own_grow->member_h.p, own_grow->pair_h.p, own_grow->status_h.p));
        HIP_TRY(sind_stream_wait(stream));
        t_fine[33] += tick_ms() - t0;
    }
    SIND_TRY(cal_occluded_p2(c, gpu_grow ? own_grow->member_h.p : nullptr, gpu_grow ? own_grow->pair_h.p : nullptr, gpu_grow ? own_grow->status_h.p : nullptr, occ1, occ2));
    totalArea = c.totalArea;
    return SIND_OK;
}

// ---- DD:653-1018: split every depth cluster on edges into pieces, region-adjacency statistics on the GPU, greedy merge on the host
int DynaTail::seg_and_merge(const std::vector<BitImg>& allLabels, const BitImg& occ1, const BitImg& occ2, const BitImg& labelForSegEdge,
                            const uint16_t* depth_host, const uint16_t* depth_dev, std::vector<uint8_t>& labelNew, const OccResult* pre) {
    struct Piece { BitImg img, lianjie; bool hasLianjie = false; float area = 0, score = -10, cz = 0; };
    std::vector<Piece> all;
    double tf = tick_ms();
    #define FLAP(i) { const double t_ = tick_ms(); t_fine[i] += t_ - tf; tf = t_; }
    const EllipseElem e4(4), e9(9), e10(10);
    const BitImg occDil = occ1.dilated(e10);
    const float depth_weight = 1.5f;
    // pieces of one depth cluster (DD:668-760); the clusters are independent, the result keeps the cluster order
    auto pieces_of = [&](int i, std::vector<Piece>& out, bool timed) {
        const BitImg& orig = allLabels[i];
        double tq = tick_ms();
        #define QLAP(i) { if (timed) { const double t_ = tick_ms(); t_fine[i] += t_ - tq; tq = t_; } }
        BitImg each = orig; each.andnot(occ1);
        { const Rect eb = each.bbox(); if (eb.empty()) return; each = each.opened_rows(e4, eb.y0, eb.y1); }
        QLAP(12)
        std::vector<Contour> contours; find_contours(each, contours, true);
        QLAP(13)
        for (const Contour& c : contours) {
            if (!(c.size() > 50 && contour_area(c) > 80)) continue;
            tq = tick_ms();
            const Rect bb = contour_bbox(c);
            Piece p; p.img.create(W, H); draw_filled(p.img, c);
            p.img = p.img.dilated(e9, bb.y0, bb.y1); p.img.and_rows(orig, bb.y0 - 4, bb.y1 + 4);       // the 9x9 element reaches 4 rows up and down
            p.area = (float)p.img.count_rows(bb.y0 - 4, bb.y1 + 4);
            QLAP(14)
            BitImg t1(W, H); draw_thick2(t1, c); t1.andnot_rows(occDil, bb.y0 - 1, bb.y1 + 1); t1.and_rows(labelForSegEdge, bb.y0 - 1, bb.y1 + 1);
            if (t1.count_rows(bb.y0 - 1, bb.y1 + 1) > 20) {
                std::vector<Contour> c2; const Rect r2{std::max(bb.x0 - 2, 0), std::max(bb.y0 - 2, 0), std::min(bb.x1 + 2, W - 1), std::min(bb.y1 + 2, H - 1)};
                find_contours(t1, c2, true, &r2);
                std::vector<const Contour*> kept; for (const Contour& q : c2) if (q.size() >= 30) kept.push_back(&q);
                if (!kept.empty()) { p.lianjie.create(W, H); draw_filled(p.lianjie, kept); p.hasLianjie = true; }
            }
            QLAP(15)
            // myCluster::calCenterPoint (DD:256-293): float sums in row-major order; only z is used afterwards
            { float f3 = 0.f; const Rect ib = p.img.bbox();
              for (int y = ib.y0; y <= ib.y1; y++) { const uint64_t* r = p.img.row(y);
                for (int k = 0; k < p.img.wpr; k++) { uint64_t m = r[k]; while (m) { const int x = (k << 6) + __builtin_ctzll(m); m &= m - 1;
                    const uint16_t d = depth_host[(size_t)y * W + x];
                    // (float)d / depthScale >= 6 is monotonic in d: compared against the first raw value that satisfies it (zInvalidFrom, init())
                    float pzv = 0.f; if (!((int)d >= zInvalidFrom || d == 0)) { const float depth2 = (float)d * invDepthScale; pzv = (float)(depth2 * depth_weight); }
                    f3 += pzv; } } }
              p.cz = f3 / p.area; }
            QLAP(16)
            out.push_back(std::move(p));
        }
        #undef QLAP
    };
    const int nCl = std::max(0, (int)allLabels.size() - 1);
    if (piece_threads > 1 && nCl > 1) {        // a stream alone on the GPU (in-order sequence mode) leaves host cores idle: the clusters go to a few helper threads
        std::vector<std::vector<Piece>> per(nCl); std::atomic<int> next{0}; std::vector<std::thread> team;
        auto work = [&] { for (int i; (i = next.fetch_add(1)) < nCl;) pieces_of(i, per[i], false); };
        for (int t = 1; t < std::min(piece_threads, nCl); t++) team.emplace_back(work);
        work();
        for (auto& t : team) t.join();
        for (auto& v : per) for (Piece& q : v) all.push_back(std::move(q));
    } else for (int i = 0; i < nCl; i++) pieces_of(i, all, true);
    FLAP(6)
    const int C = (int)all.size();
    dbg.nClusters = C;
    labelNew.assign(N, 0);
    if (C == 0) return SIND_OK;
    if (C > 254) { sind_set_error("seg_and_merge: %d pieces exceed the 8-bit label range of the reference", C); return SIND_E_CAPACITY; }
    for (Piece& p : all) p.score = (float)(p.area * 0.0003f - p.cz);
    std::sort(all.begin(), all.end(), [](const Piece& a, const Piece& b) { return a.score > b.score; });
    std::vector<uint8_t> pieceOf(N, (uint8_t)(C + 1));       // index of the piece that owns a pixel (the reference paints the pieces in score order)
    for (int i = 0; i < C; i++) all[i].img.paint_u8(pieceOf.data(), W, (uint8_t)i);
    // ---- RAG statistics on the GPU: upload the 3*C bit planes, one pass over the frame
    const int wpr = W / 64; const size_t pw = (size_t)H * wpr;
    SIND_TRY(h_planes.alloc((size_t)3 * C * pw)); SIND_TRY(h_rag.alloc((size_t)3 * C * C + C + (size_t)C * 256));
    unsigned long long* planes = h_planes.p; const size_t planes_n = (size_t)3 * C * pw;
    // plane sets: [0, C) pieces, [C, 2C) their 7x7 dilations (formed on the GPU from the first set), [2C, 3C) connection areas
    for (int i = 0; i < C; i++) if (!all[i].hasLianjie) std::memset(&planes[((size_t)2 * C + i) * pw], 0, pw * 8);
    for (int i = 0; i < C; i++) {
        std::memcpy(&planes[((size_t)0 * C + i) * pw], all[i].img.d.data(), pw * 8);
        if (all[i].hasLianjie) std::memcpy(&planes[((size_t)2 * C + i) * pw], all[i].lianjie.d.data(), pw * 8);
    }
    FLAP(10)
    SIND_TRY(planes_d.alloc(planes_n)); SIND_TRY(rag_d.alloc((size_t)3 * C * C + C + (size_t)C * 256));
    FLAP(11)
    HIP_TRY(hipMemcpyAsync(planes_d.p, planes, (size_t)C * pw * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(planes_d.p + (size_t)2 * C * pw, planes + (size_t)2 * C * pw, (size_t)C * pw * 8, hipMemcpyHostToDevice, stream));
    SIND_TRY(launch_dilate_planes(stream, planes_d.p, planes_d.p + (size_t)C * pw, C, W, H, 7, rag_d.p, (int)((size_t)3 * C * C + C + (size_t)C * 256)));      // also clears the RAG accumulators
    const uint8_t* occ2_use = pre && pre->occ2_dev ? pre->occ2_dev : occ2_d.p;
    if (pre && pre->occ2_dev && pre->occ2_event) HIP_TRY(hipStreamWaitEvent(stream, pre->occ2_event, 0));       // uploaded on the CalOccluded runner's stream without a host wait
    if (!(pre && pre->occ2_dev)) { occ2.to_u8((h_ab.p + N), W, 255); HIP_TRY(hipMemcpyAsync(occ2_d.p, (h_ab.p + N), N, hipMemcpyHostToDevice, stream)); }
    FLAP(7)
    const uint8_t* depthN_use = pre && pre->depthN_dev ? pre->depthN_dev : depthN.p;
    if (!(pre && pre->depthN_dev)) { SIND_TRY(launch_max_u16(stream, depth_dev, N, umax_d.p + 1)); SIND_TRY(launch_depth_norm(stream, depth_dev, umax_d.p + 1, depthN.p, N)); }
    int* ov_d = rag_d.p; int* ovp_d = ov_d + C * C; int* lj_d = ovp_d + C * C; int* la_d = lj_d + C * C; int* hist_dd = la_d + C;
    SIND_TRY(launch_rag_stats(stream, planes_d.p, C, W, H, wpr, occ2_use, depthN_use, ov_d, ovp_d, lj_d, la_d, hist_dd, true));
    PinnedBuf<int>& rag = h_rag;
    HIP_TRY(hipMemcpyAsync(rag.data(), rag_d.p, ((size_t)3 * C * C + C + (size_t)C * 256) * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    FLAP(8)
    const int* ov = rag.data(); const int* ovp = ov + C * C; const int* ljo = ovp + C * C; const int* lja = ljo + C * C; const int* hst = lja + C;
    // cal_hist (DD:1685-1739) from the two masked histograms
    auto cal_hist = [&](int a, int b, double out[3]) {
        float h1[256], h2[256]; double m1 = 0, m2 = 0, mn1 = DBL_MAX, mn2 = DBL_MAX;
        for (int i = 0; i < 256; i++) { h1[i] = (float)hst[a * 256 + i]; h2[i] = (float)hst[b * 256 + i]; m1 = std::max<double>(m1, h1[i]); m2 = std::max<double>(m2, h2[i]); mn1 = std::min<double>(mn1, h1[i]); mn2 = std::min<double>(mn2, h2[i]); }
        const int hist_h = 400;
        auto norm_minmax = [&](float* h, double mn, double mx) { const double scale = (mx - mn) > DBL_EPSILON ? (double)hist_h / (mx - mn) : 0, shift = 0 - mn * scale; for (int i = 0; i < 256; i++) h[i] = h[i] * (float)scale + (float)shift; };
        if (m1 > m2) { norm_minmax(h1, mn1, m1); const float s = (float)(1.0 / (m1 / hist_h)); for (int i = 0; i < 256; i++) h2[i] = h2[i] * s; }
        else         { norm_minmax(h2, mn2, m2); const float s = (float)(1.0 / (m2 / hist_h)); for (int i = 0; i < 256; i++) h1[i] = h1[i] * s; }
        double s1 = 0, s2 = 0, s11 = 0, s12 = 0, s22 = 0, bh = 0, inter = 0;
        for (int j = 0; j < 256; j++) { const double x = h1[j], y = h2[j]; s12 += x * y; s1 += x; s11 += x * x; s2 += y; s22 += y * y; bh += std::sqrt(x * y); inter += std::min(h1[j], h2[j]); }
        const double scale = 1. / 256, num = s12 - s1 * s2 * scale, denom2 = (s11 - s1 * s1 * scale) * (s22 - s2 * s2 * scale);
        out[0] = std::fabs(denom2) > DBL_EPSILON ? num / std::sqrt(denom2) : 1.;
        double ss = s1 * s2; ss = std::fabs(ss) > FLT_EPSILON ? 1. / std::sqrt(ss) : 1.;
        out[1] = 1 - std::sqrt(std::max(1. - bh * ss, 0.));
        out[2] = inter;
    };
    const int M = C + 1, numCluster = KM_K;
    std::vector<float> mTotal((size_t)M * M, 0.f), m2((size_t)M * M, 0.f), m3((size_t)M * M, 0.f), wgt((size_t)M * M, 1.f), rej((size_t)M * M, 1.f);
    const float thredshold = 0.9f; const int smallLabel = (int)std::min(0.7f * C, 15.0f);
    for (int i = 0; i < C; i++) for (int j = i + 1; j < C; j++) {
        float v2 = 0, v3 = 0, lessArea; int lessLabel;
        if (all[i].area < all[j].area) { lessArea = all[i].area; lessLabel = i; } else { lessArea = all[j].area; lessLabel = j; }
        if (lessLabel < 10) wgt[i * M + j] = wgt[j * M + i] = 0.7f;
        else if (lessLabel > smallLabel) wgt[i * M + j] = wgt[j * M + i] = 2.0f;
        const int overlap = ov[i * C + j], overlapPlane = ovp[i * C + j];
        if (overlap > std::min(200.0f, lessArea * 0.4f)) {
            double isMerge[3]; cal_hist(i, j, isMerge);
            v3 = (float)(isMerge[0] + isMerge[1] + isMerge[2] * 0.0005);
            if (overlapPlane > 100 && lessLabel < smallLabel) { rej[i * M + j] = rej[j * M + i] = 0.f; continue; }
            else if (v3 < 0.19f && lessLabel < smallLabel) { rej[i * M + j] = rej[j * M + i] = 0.f; continue; }
            if (all[i].hasLianjie && all[j].hasLianjie) {
                const int o = ljo[i * C + j];
                if (o > 0) { const int a1 = lja[i], a2 = lja[j];
                    if (o > std::min(50, (int)(0.5 * std::min(a1, a2)))) { v2 = (float)o; if ((o > 0.62 * a1) || (o > 0.62 * a2)) v2 = (float)std::max(250, o); } }
            }
            m2[i * M + j] = m2[j * M + i] = v2; m3[i * M + j] = m3[j * M + i] = v3;
        }
    }
    for (size_t k = 0; k < mTotal.size(); k++) mTotal[k] = ((m2[k] * 0.01f + m3[k]) * rej[k]) * wgt[k];
    int countMerged = 0;
    std::vector<std::vector<int>> merge(M); std::vector<int> mergeSit(M, 0);
    auto fold = [&](int dst, int j) {
        std::vector<float> col(M); for (int r = 0; r < M; r++) col[r] = mTotal[r * M + j];
        for (int r = 0; r < M; r++) mTotal[r * M + dst] += col[r];
        for (int c = 0; c < M; c++) mTotal[dst * M + c] += col[c];
        for (int r = 0; r < M; r++) mTotal[r * M + j] = 0.f;
        for (int c = 0; c < M; c++) mTotal[j * M + c] = 0.f;
    };
    for (int i = 0; i < std::min(numCluster - 1 + countMerged, C); i++)
        for (int j = i + 1; j < std::min(numCluster - 1 + countMerged, C); j++) {
            const float sorce = mTotal[j * M + i];
            if (sorce > thredshold) {
                int toMerge = i; const float toMergeValue = mTotal[j * M + i];
                for (int k = 0; k < j; k++) if (mTotal[k * M + j] > toMergeValue) toMerge = k;
                mergeSit[j] = 1; merge[toMerge].push_back(j); fold(toMerge, j); countMerged++;
            }
        }
    for (int i = std::min(numCluster - 1 + countMerged, C); i < C; i++) {
        int mergeCluster = C; float maxScore = 0.2f;
        for (int j = 0; j < i; j++) { const float score = mTotal[j * M + i]; if (score > maxScore) { maxScore = score; mergeCluster = j; } }
        mergeSit[i] = 1; merge[mergeCluster].push_back(i); fold(mergeCluster, i);
    }
    int labelindex = 1; std::vector<uint8_t> lut(256, 0);
    for (int i = 0; i < C; i++) {
        if (mergeSit[i]) continue;
        std::vector<char> sel(M + 1, 0); sel[i] = 1;
        for (int mj : merge[i]) { sel[mj] = 1; for (int mk : merge[mj]) sel[mk] = 1; }
        // the reference paints labels in increasing order, a later group overwrites an earlier one on shared pieces
        for (int q = 0; q <= M; q++) if (sel[q]) lut[q] = (uint8_t)labelindex;
        labelindex++;
    }
    for (int k = 0; k < N; k++) labelNew[k] = lut[pieceOf[k]];
    FLAP(9)
    #undef FLAP
    return SIND_OK;
}

// ---- DD:1377-1666
// ---- GPU half of CalOccluded (and the 8-bit normalised depth of the RAG statistics) for a chunk of frames in seven launches: the stage has no
// state, so the pipeline runs it for all frames of a step at the step's start instead of seven launches + a stream wait per frame (DD:429-482, 765-768)
int OccBatch::init(const DynaConfig& c, int chunk) {
    cfg = c; cap = chunk; const size_t n = (size_t)c.W * c.H;
    SIND_TRY(filt.alloc(n * cap)); SIND_TRY(edge.alloc(n * cap)); SIND_TRY(edgeTmp.alloc(n * cap)); SIND_TRY(total.alloc(n * cap));
    SIND_TRY(umax.alloc((size_t)2 * cap)); SIND_TRY(blocks.alloc((size_t)(c.W / 16) * (c.H / 16) * cap));
    return SIND_OK;
}
int OccBatch::run(hipStream_t s, const uint16_t* depth_dev, int B, uint8_t* depthN_dev, uint8_t* edge_h, uint8_t* total_h, PeacBlockStats* blocks_h) {
    if (B < 1 || B > cap) { sind_set_error("OccBatch::run: %d frames (capacity %d)", B, cap); return SIND_E_ARG; }
    const int W = cfg.W, H = cfg.H, N = W * H; const size_t nb = (size_t)(W / 16) * (H / 16);
    if (depthN_dev) { SIND_TRY(launch_max_u16(s, depth_dev, N, umax.p + cap, B, 1)); SIND_TRY(launch_depth_norm(s, depth_dev, umax.p + cap, depthN_dev, N, B, 1)); }
    SIND_TRY(launch_median5(s, depth_dev, filt.p, W, H, B));
    SIND_TRY(launch_max_u16(s, filt.p, N, umax.p, B, 1));
    SIND_TRY(launch_grad_edge(s, filt.p, umax.p, edge.p, total.p, W, H, cfg.depthScale, B, 1));
    SIND_TRY(launch_morph(s, edge.p, edgeTmp.p, W, H, 4, false, B));       // MORPH_OPEN, element4 = erode then dilate
    SIND_TRY(launch_morph(s, edgeTmp.p, edge.p, W, H, 4, true, B));
    SIND_TRY(launch_peac_block_stats(s, depth_dev, W, H, 16, 16, cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.depthScale, blocks.p, B));
    HIP_TRY(hipMemcpyAsync(edge_h, edge.p, (size_t)N * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(total_h, total.p, (size_t)N * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(blocks_h, blocks.p, nb * B * sizeof(PeacBlockStats), hipMemcpyDeviceToHost, s));
    return SIND_OK;
}

int DynaTail::compute_occluded(const uint16_t* depth_host, const uint16_t* depth_dev, OccResult& out, const OccGpuOut* pre) {
    HIP_TRY(hipSetDevice(cfg.device));
    // the depth-only inputs of the RAG statistics go first so that they are finished by the time cal_occluded waits on the stream
    if (out.depthN_dev && !pre) { SIND_TRY(launch_max_u16(stream, depth_dev, N, umax_d.p + 1)); SIND_TRY(launch_depth_norm(stream, depth_dev, umax_d.p + 1, out.depthN_dev, N)); }
    SIND_TRY(cal_occluded(depth_host, depth_dev, out.totalArea, out.occ1, out.occ2, pre));
    return finish_occluded(out, pre);
}
// batched pipeline: first half of a frame (the grow of its chunk is launched by whoever finishes the chunk's last first half) ...
int DynaTail::compute_occluded_p1(const uint16_t* depth_host, const uint16_t* depth_dev, OccCtx& c, const OccGpuOut* pre, uint8_t* grow_block, int depth_index) {
    HIP_TRY(hipSetDevice(cfg.device));
    return cal_occluded_p1(depth_host, depth_dev, c, pre, grow_block, depth_index);
}
// ... and the second half once the chunk's grow results are on the host
int DynaTail::compute_occluded_p2(OccCtx& c, const int8_t* member8, const uint8_t* pair_seen, const int* grow_status, OccResult& out, const OccGpuOut* pre) {
    HIP_TRY(hipSetDevice(cfg.device));
    SIND_TRY(cal_occluded_p2(c, member8, pair_seen, grow_status, out.occ1, out.occ2));
    out.totalArea = c.totalArea;
    return finish_occluded(out, pre);
}
int DynaTail::finish_occluded(OccResult& out, const OccGpuOut* pre) {
    if (out.occ2_dev && pre && pre->occ2_stage && pre->occ2_event) {      // batched path: page-locked staging of the frame's own, consumers wait on the event, not this thread
        out.occ2.to_u8(pre->occ2_stage, W, 255); HIP_TRY(hipMemcpyAsync(out.occ2_dev, pre->occ2_stage, N, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(pre->occ2_event, stream)); out.occ2_event = pre->occ2_event;
    } else if (out.occ2_dev) { out.occ2.to_u8(h_ab.p + N, W, 255); HIP_TRY(hipMemcpyAsync(out.occ2_dev, h_ab.p + N, N, hipMemcpyHostToDevice, stream)); HIP_TRY(sind_stream_wait(stream)); out.occ2_event = nullptr; }
    out.ready = true; return SIND_OK;
}

int DynaTail::process(const uint16_t* depth_host, const uint16_t* depth_dev, const float* U, const float* V, uint8_t* dyna_out, uint8_t* label_out,
                      const OccResult* pre, DynaTail* depth_half, const KmFrameResult* km) {
    // same stage order as the reference (flow masks, clustering, fusion); the order also matters for throughput: a pool of tails that
    // all start with the 50-launch k-means chain was measured 20 ms per step slower than one that starts with the host-side pair sorting
    HIP_TRY(hipSetDevice(cfg.device));
    BitImg maskLow, maskHigh; n_frames++;
    { const double t0 = tick_ms(); SIND_TRY(flow_masks(U, V, maskLow, maskHigh, pre ? pre->gridFlow : nullptr)); t_stage[0] += tick_ms() - t0; }
    DepthStageOut d;
    SIND_TRY((depth_half ? depth_half : this)->depth_stage(depth_host, depth_dev, pre, d, km));
    return fuse(maskLow, maskHigh, d, dyna_out, label_out);
}

// flow-independent half: k-means (DD:1410-1414), cluster order (DD:1428-1491), CalOccluded (DD:1493), SegAndMerge (DD:1495-1551)
int DynaTail::depth_stage(const uint16_t* depth_host, const uint16_t* depth_dev, const OccResult* pre, DepthStageOut& out, const KmFrameResult* km) {
    HIP_TRY(hipSetDevice(cfg.device));
    double tk = tick_ms();
    #define LAP(i) { const double t_ = tick_ms(); t_stage[i] += t_ - tk; tk = t_; }
    std::vector<uint8_t> label8; float centers[KM_K][3]; int counts[KM_K];
    if (km) { label8.assign(km->label8, km->label8 + N); std::memcpy(centers, km->centers, sizeof(centers)); std::memcpy(counts, km->counts, sizeof(counts)); }
    else SIND_TRY(kmeans(depth_dev, label8, centers, counts));
    if (keep_debug) { dbg.kmeansLabel = label8; std::memcpy(dbg.centers, centers, sizeof(centers)); }
    LAP(1)
    // nearest clusters first (DD:1428-1491)
    float depth_vals[KM_K]; int order[KM_K];
    for (int i = 0; i < KM_K; i++) { depth_vals[i] = centers[i][2]; if (depth_vals[i] < 0.2) depth_vals[i] += 20.0f; order[i] = i; }
    std::stable_sort(order, order + KM_K, [&](int a, int b) { return depth_vals[a] < depth_vals[b]; });
    std::vector<BitImg> labelMask(KM_K); BitImg::split_labels(label8.data(), W, H, W, 0, KM_K, labelMask.data());
    std::vector<BitImg> allLabels; BitImg labelForSegEdge(W, H);
    float ratioArea = 0.0f; const float TotalArea = (float)(H * W); int count0 = 0;
    for (int i = 0; i < KM_K; i++) {
        const int idx = order[i]; const int cnt = labelMask[idx].count();
        if (cnt < 60) continue;
        allLabels.push_back(labelMask[idx]);
        const float ratio = (float)cnt * (1.0f / TotalArea); ratioArea += ratio;
        if (count0 <= 5 && ratioArea < 0.6f) { labelForSegEdge |= labelMask[idx]; ++count0; }
    }
    labelForSegEdge = labelForSegEdge.dilated(EllipseElem(7));
    BitImg occ1, occ2;
    LAP(2)
    if (pre && pre->ready) { out.totalArea = pre->totalArea; occ1 = pre->occ1; occ2 = pre->occ2; }
    else SIND_TRY(cal_occluded(depth_host, depth_dev, out.totalArea, occ1, occ2));
    LAP(3)
    out.label3.assign(N, 0);
    if (!allLabels.empty()) SIND_TRY(seg_and_merge(allLabels, occ1, occ2, labelForSegEdge, depth_host, depth_dev, out.label3, pre && pre->ready ? pre : nullptr));
    LAP(4)
    #undef LAP
    int maxNum = 0; for (uint8_t v : out.label3) maxNum = std::max<int>(maxNum, v);
    out.maxNum = maxNum; out.ready = true;
    if (keep_debug) { dbg.occ1.resize(N); occ1.to_u8(dbg.occ1.data(), W, 255); dbg.occ2.resize(N); occ2.to_u8(dbg.occ2.data(), W, 255); dbg.totalArea.resize(N); out.totalArea.to_u8(dbg.totalArea.data(), W, 255); }
    kmLabelLast = out.label3; kmLabelLastAny = maxNum > 0;          // imgLabelLast for the next frame's k-means (DD:1660-1664)
    return SIND_OK;
}

// flow-dependent half: flow masks (DD:1163-1367) and fusion (DD:1553-1636), then the state roll (DD:1660-1664)
int DynaTail::flow_stage(const float* U, const float* V, const DepthStageOut& d, uint8_t* dyna_out, uint8_t* label_out, const float* gridFlowPre) {
    HIP_TRY(hipSetDevice(cfg.device));
    if (!d.ready) { sind_set_error("DynaTail::flow_stage: the depth stage of this frame has not run"); return SIND_E_STATE; }
    BitImg maskLow, maskHigh; n_frames++;
    { const double t0 = tick_ms(); SIND_TRY(flow_masks(U, V, maskLow, maskHigh, gridFlowPre)); t_stage[0] += tick_ms() - t0; }
    return fuse(maskLow, maskHigh, d, dyna_out, label_out);
}

// fusion (DD:1553-1636) and the state roll (DD:1660-1664)
int DynaTail::fuse(const BitImg& maskLow, const BitImg& maskHigh, const DepthStageOut& d, uint8_t* dyna_out, uint8_t* label_out) {
    double tk = tick_ms();
    #define LAP(i) { const double t_ = tick_ms(); t_stage[i] += t_ - tk; tk = t_; }
    const std::vector<uint8_t>& label3 = d.label3; const int maxNum = d.maxNum; const BitImg& totalArea = d.totalArea;
    double tq = tick_ms();
    #define QLAP(i) { const double t_ = tick_ms(); t_fine[i] += t_ - tq; tq = t_; }
    // fusion (DD:1553-1636)
    BitImg low = highLast; low |= maskLow; low &= totalArea;
    low = low.dilated(EllipseElem(5));
    const BitImg notLow = low.inverted();
    QLAP(25)
    BitImg dyna(W, H);
    std::vector<BitImg> clusters(maxNum + 1); clusters[0].create(W, H);
    if (maxNum > 0) BitImg::split_labels(label3.data(), W, H, W, 1, maxNum, clusters.data() + 1);
    QLAP(26)
    for (int n = 1; n <= maxNum; n++) {
        const BitImg& one = clusters[n]; const int oneCnt = one.count();
        BitImg blocked = one.inverted(), filled(W, H);
        BitImg oneHigh = one; oneHigh &= maskHigh;
        if (oneHigh.count() > 100) {
            std::vector<Contour> cs; find_contours(oneHigh, cs, false);
            for (const Contour& c : cs) {
                const double area = contour_area(c), len = arc_length_closed(c), roundness = (4 * M_PI * area) / (len * len);
                PtI seed{0, 0};
                for (const PtI& p : c) if (low.get(p.x, p.y)) { seed = p; break; }
                if ((area > 100.0 && roundness > 0.2) || area > 2000.0) flood_fill(low.get(seed.x, seed.y) ? low : notLow, blocked, filled, seed);
            }
        }
        if (filled.count() > 0.5 * oneCnt) dyna |= one; else dyna |= filled;
    }
    QLAP(27)
    dyna = dyna.dilated(EllipseElem(9));
    totalArea.to_u8(dyna_out, W, 125); dyna.paint_u8(dyna_out, W, 255);
    std::memcpy(label_out, label3.data(), N);
    // roll the state (DD:1660-1664)
    std::memcpy(dynaLast.data(), dyna_out, N); labelLast = label3; highLast = maskHigh;
    std::memset(lastCnt, 0, sizeof(lastCnt)); std::memset(lastDyn, 0, sizeof(lastDyn));       // per-label pixel / dynamic-pixel counts for the next frame's sample weights
    for (int n = 1; n <= maxNum && n < 256; n++) { lastCnt[n] = clusters[n].count(); lastDyn[n] = BitImg::and_count(clusters[n], dyna); }
    if (hash_state) {
        // imgDynaLast = 255 on `dyna`, 125 on totalArea - dyna, else 0: hashed as those two bit images (canonical: equal byte images give equal words)
        StateHash hs;
        for (size_t i = 0; i < dyna.d.size(); i++) { hs.word(dyna.d[i]); hs.word(totalArea.d[i] & ~dyna.d[i]); hs.word(maskHigh.d[i]); }
        hs.bytes(label3.data(), N); hs.bytes(lastCnt, sizeof(lastCnt)); hs.bytes(lastDyn, sizeof(lastDyn)); hs.word(maxNum > 0 ? 1 : 0);
        hs.finish(state_hash);
    }
    QLAP(28)
    #undef QLAP
    LAP(5)
    #undef LAP
    return SIND_OK;
}

}  // namespace sind
