// C ABI: DynaDetect, single stream, one frame per call (include/sind_hip.h).
#include <cstring>
#include "../../include/sind_hip.h"
#include "dyna.hpp"
#include "peac_grow.hpp"

struct sind_dyna {
    sind::DynaConfig cfg; hipStream_t stream = nullptr, tail_stream = nullptr, km_stream = nullptr; hipEvent_t flow_done = nullptr, depth_up = nullptr; sind::DynaFront front; sind::DynaTail tail;
    sind::KMeansBatch kmb;       // the frame's k-means on its own stream beside CalOccluded (both need the depth frame only)
    DevBuf<uint8_t> bgr, gray, pool, dil_a, dil_b; DevBuf<uint16_t> depth; DevBuf<float> U, V;
    PinnedBuf<uint16_t> depth_h; PinnedBuf<uint8_t> bgr_h, dil_h;       // page-locked staging of the caller's frames (a copy from pageable memory is staged by the runtime and blocks the call)
    int t = 0; bool primed = false; int largeMotion = 0; bool debug = false, overlap = true;
    std::vector<float> deep, refined;
    double t_ms[6] = {0, 0, 0, 0, 0, 0}; long n_timed = 0;       // summed per-call times: upload, dense flow (main thread), wait for the depth half, flow masks + fusion, depth half (its own thread), total
};
#include <string>
#include <thread>
namespace { struct SpinScope { int keep; SpinScope() : keep(t_sind_spin_us) { t_sind_spin_us = 2000; } ~SpinScope() { t_sind_spin_us = keep; } }; }      // one camera, idle host: poll before sleeping (common.hpp)

extern "C" {

int sind_dyna_create(int w, int h, float fx, float fy, float cx, float cy, float ds, int device, sind_dyna** out) {
    if (!out || w < 64 || h < 64 || ds <= 0) { sind_set_error("sind_dyna_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    sind_dyna* d = new sind_dyna();
    d->cfg.W = w; d->cfg.H = h; d->cfg.fx = fx; d->cfg.fy = fy; d->cfg.cx = cx; d->cfg.cy = cy; d->cfg.depthScale = ds; d->cfg.device = device;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { delete d; sind_set_error("sind_dyna_create: hipStreamCreate failed"); return SIND_E_HIP; }
    // the depth half of a frame (k-means, CalOccluded, SegAndMerge: everything that does not need the flow) runs on its own stream and host thread beside the dense flow, as
    // the reference runs its flow thread beside the segmentation (DynaDetect.cc:1396-1398, 1553-1554)
    if (hipStreamCreateWithFlags(&d->tail_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->flow_done, hipEventDisableTiming) != hipSuccess) {
        hipStream_t a = d->stream, b = d->tail_stream; delete d; (void)hipStreamDestroy(a); if (b) (void)hipStreamDestroy(b); sind_set_error("sind_dyna_create: hipStreamCreate failed"); return SIND_E_HIP; }
    int r = d->front.init(d->cfg, 2, d->stream);       // (2: the candidate of the large-motion pass rides along with every frame, see DynaFront::speculate)
    d->front.speculate = true;
    if (r == SIND_OK) r = d->tail.init(d->cfg, d->tail_stream);
    if (r == SIND_OK && (hipStreamCreateWithFlags(&d->km_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->depth_up, hipEventDisableTiming) != hipSuccess)) { sind_set_error("sind_dyna_create: hipStreamCreate failed"); r = SIND_E_HIP; }
    if (r == SIND_OK) r = d->kmb.init(d->cfg, 1, d->km_stream);
    if (r == SIND_OK) r = d->depth_h.alloc((size_t)w * h);
    if (r == SIND_OK) r = d->bgr_h.alloc((size_t)w * h * 3);
    if (r == SIND_OK) r = d->dil_h.alloc((size_t)w * h);
    const size_t np = (size_t)w * h;
    if (r == SIND_OK) r = d->bgr.alloc(np * 3);
    if (r == SIND_OK) r = d->gray.alloc(np);
    if (r == SIND_OK) r = d->pool.alloc((size_t)d->front.fw * d->front.fh * 3);
    if (r == SIND_OK) r = d->depth.alloc(np);
    if (r == SIND_OK) r = d->U.alloc(np);
    if (r == SIND_OK) r = d->V.alloc(np);
    if (r == SIND_OK) r = d->dil_a.alloc(np);
    if (r == SIND_OK) r = d->dil_b.alloc(np);
    if (r != SIND_OK) { hipStream_t st = d->stream, ts = d->tail_stream, ks = d->km_stream; hipEvent_t ev = d->flow_done, e2 = d->depth_up; delete d; (void)hipStreamDestroy(st); (void)hipStreamDestroy(ts); if (ks) (void)hipStreamDestroy(ks); (void)hipEventDestroy(ev); if (e2) (void)hipEventDestroy(e2); return r; }
    d->tail.keep_debug = false; d->tail.piece_threads = 4;      // (one camera per handle: the host cores are idle while a frame is in flight)
    *out = d; return SIND_OK;
}
int sind_debug_seqsum(const float* x, int n, int device, float* out) {
    if (!x || !out || n < 0) return SIND_E_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf<float> xd; DevBuf<int> scratch; SIND_TRY(xd.alloc((size_t)std::max(n, 1) + 8)); SIND_TRY(scratch.alloc(2 * KM_K + 64 + sizeof(sind::KmState) / 4 + 4));
    HIP_TRY(hipMemcpy(xd.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return sind::debug_seqsum(nullptr, xd.p, n, scratch.p, out);
}
// parity-test / A-B access: k-means levels of at most n points run in the fused one-launch kernel (default 81920; 0 = the per-pass kernels everywhere).  These set the DEFAULTS
// that handles created afterwards copy; a handle that exists keeps what it was created with.
int sind_debug_set_kmeans_fused_max(int n) { if (n < 0) return SIND_E_ARG; sind::g_km_fuse_default.max_points = n; return SIND_OK; }
int sind_debug_set_kmeans_fused_min_batch(int b) { if (b < 1) return SIND_E_ARG; sind::g_km_fuse_default.min_batch = b; return SIND_OK; }
// debug != 0: keep what sind_dyna_debug reports (intermediate images of every stage, the raw and refined flow copied back every frame); off by default
int sind_dyna_set_debug(sind_dyna* d, int on) { if (!d) return SIND_E_ARG; d->debug = on != 0; d->tail.keep_debug = d->debug; return SIND_OK; }
// overlap != 0 (default): the depth half of a frame runs beside its dense flow (own stream, own host thread); 0 = one after the other.  Same results.
int sind_dyna_set_overlap(sind_dyna* d, int on) { if (!d) return SIND_E_ARG; d->overlap = on != 0; return SIND_OK; }
// mean milliseconds per sind_dyna_detect call since the last reset: [0..5] upload, dense flow (calling thread), wait for the depth half after the flow, flow masks + fusion,
// depth half (its own thread when overlapped), whole call; [6..11] the tail's stages: flow masks, k-means, label preparation, CalOccluded, SegAndMerge, fusion.
// Returns the number of calls averaged
int sind_dyna_timing(sind_dyna* d, double ms12[12], int reset) {
    if (!d || !ms12) return SIND_E_ARG;
    for (int i = 0; i < 6; i++) ms12[i] = d->n_timed ? d->t_ms[i] / d->n_timed : 0.0;
    for (int i = 0; i < 6; i++) ms12[6 + i] = d->n_timed ? d->tail.t_stage[i] / d->n_timed : 0.0;
    const int n = (int)d->n_timed;
    if (reset) { for (double& v : d->t_ms) v = 0; for (double& v : d->tail.t_stage) v = 0; d->n_timed = 0; }
    return n;
}
int sind_dyna_timing_fine(sind_dyna* d, double ms40[40], int reset) {
    if (!d || !ms40) return SIND_E_ARG;
    for (int i = 0; i < 40; i++) ms40[i] = d->tail.t_fine[i];
    if (reset) for (double& v : d->tail.t_fine) v = 0;
    return SIND_OK;
}
int sind_dyna_set_flow_max_levels(sind_dyna* d, int n) { if (!d || n < 0) return SIND_E_ARG; d->front.flow.max_levels = n; return SIND_OK; }
int sind_dyna_destroy(sind_dyna* d) {
    if (!d) return SIND_OK;
    (void)hipSetDevice(d->cfg.device);
    hipStream_t s = d->stream, ts = d->tail_stream, ks = d->km_stream; hipEvent_t ev = d->flow_done, e2 = d->depth_up;
    if (s) (void)hipStreamSynchronize(s);
    if (ts) (void)hipStreamSynchronize(ts);
    if (ks) (void)hipStreamSynchronize(ks);
    delete d; if (s) (void)hipStreamDestroy(s); if (ts) (void)hipStreamDestroy(ts); if (ks) (void)hipStreamDestroy(ks); if (ev) (void)hipEventDestroy(ev); if (e2) (void)hipEventDestroy(e2);
    return SIND_OK;
}
static int upload_bgr(sind_dyna* d, const uint8_t* bgr, int stride, int slot) {
    const int w = d->cfg.W, h = d->cfg.H;
    HIP_TRY(hipStreamSynchronize(d->stream));            // (the staging buffer of the previous upload is free)
    for (int y = 0; y < h; y++) std::memcpy(d->bgr_h.p + (size_t)y * w * 3, bgr + (size_t)y * stride, (size_t)w * 3);
    HIP_TRY(hipMemcpyAsync(d->bgr.p, d->bgr_h.p, (size_t)w * h * 3, hipMemcpyHostToDevice, d->stream));
    return d->front.gray_and_min(d->bgr.p, 1, d->gray.p, d->pool.p + (size_t)d->front.fw * d->front.fh * slot);
}
int sind_dyna_prime(sind_dyna* d, const uint8_t* last, const uint8_t* lastlast, int stride) {
    if (!d || !last || !lastlast) { sind_set_error("sind_dyna_prime: null argument"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(d->cfg.device));
    if (stride <= 0) stride = d->cfg.W * 3;
    SIND_TRY(upload_bgr(d, lastlast, stride, 0));       // frame n-2
    SIND_TRY(upload_bgr(d, last, stride, 1));           // frame n-1
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->t = 2; d->primed = true; d->tail.reset();
    return SIND_OK;
}
int sind_dyna_detect(sind_dyna* d, const uint8_t* bgr, int bstride, const uint16_t* depth, int dstride, uint8_t* dyna_out, uint8_t* label_out, int) {
    if (!d || !bgr || !depth || !dyna_out || !label_out) { sind_set_error("sind_dyna_detect: null argument"); return SIND_E_ARG; }
    if (!d->primed) { sind_set_error("sind_dyna_detect: call sind_dyna_prime first (the reference constructor takes the two previous frames)"); return SIND_E_STATE; }
    HIP_TRY(hipSetDevice(d->cfg.device));
    const int w = d->cfg.W, h = d->cfg.H;
    if (bstride <= 0) bstride = w * 3;
    if (dstride <= 0) dstride = w * 2;
    const int cur = d->t % 3, p1 = (d->t + 2) % 3, p2 = (d->t + 1) % 3;
    SpinScope spin;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now_ms(); double t_depth = 0;
    SIND_TRY(upload_bgr(d, bgr, bstride, cur));
    uint16_t* dh = d->depth_h.p;
    for (int y = 0; y < h; y++) std::memcpy(dh + (size_t)y * w, (const uint8_t*)depth + (size_t)y * dstride, (size_t)w * 2);
    HIP_TRY(hipMemcpyAsync(d->depth.p, dh, (size_t)w * h * 2, hipMemcpyHostToDevice, d->tail_stream));      // the depth frame is the tail's input only
    HIP_TRY(hipEventRecord(d->depth_up, d->tail_stream)); HIP_TRY(hipStreamWaitEvent(d->km_stream, d->depth_up, 0));
    const size_t nf = (size_t)d->front.fw * d->front.fh;
    float *du = nullptr, *dv = nullptr, *ru = nullptr, *rv = nullptr;
    if (d->debug) { d->deep.resize(nf * 2); d->refined.resize(nf * 2); du = d->deep.data(); dv = du + nf; ru = d->refined.data(); rv = ru + nf; }
    sind::DepthStageOut dso; int rc_depth = SIND_OK; std::string err_depth;
    std::thread side;
    const double t1 = now_ms();
    // the depth half: CalOccluded (GPU edges + PEAC: ~7 ms of a frame's ~10) on this thread and the tail's stream, the k-means (~1.7 ms) beside it on a third thread and stream
    // -- neither needs the other (DynaDetect.cc:1410 / :1497) --, then the label preparation and SegAndMerge, which need both
    if (d->overlap) side = std::thread([&] {
        SpinScope sp; const double a = now_ms(); (void)hipSetDevice(d->cfg.device);
        int rc_km = SIND_OK; std::string err_km;
        std::thread kmt([&] { SpinScope s2; (void)hipSetDevice(d->cfg.device); const uint8_t* prev = d->tail.prev_km_labels(); rc_km = d->kmb.run(d->depth.p, (size_t)w * h, 1, &prev); if (rc_km != SIND_OK) err_km = sind_last_error(); });
        sind::OccResult occ; rc_depth = d->tail.compute_occluded(dh, d->depth.p, occ, nullptr);
        if (rc_depth != SIND_OK) err_depth = sind_last_error();
        kmt.join();
        if (rc_depth == SIND_OK && rc_km != SIND_OK) { rc_depth = rc_km; err_depth = err_km; }
        if (rc_depth == SIND_OK) { rc_depth = d->tail.depth_stage(dh, d->depth.p, &occ, dso, &d->kmb.result(0)); if (rc_depth != SIND_OK) err_depth = sind_last_error(); }
        t_depth = now_ms() - a; });
    const int rc_flow = d->front.dense_flow(d->pool.p, &cur, &p1, &p2, 1, d->U.p, d->V.p, &d->largeMotion, du, dv, ru, rv);
    const double t2 = now_ms();
    if (side.joinable()) side.join();
    SIND_TRY(rc_flow);
    if (!d->overlap) { rc_depth = d->tail.depth_stage(dh, d->depth.p, nullptr, dso, nullptr); t_depth = now_ms() - t2; }
    else if (rc_depth != SIND_OK) sind_set_error("%s", err_depth.c_str());
    SIND_TRY(rc_depth);
    HIP_TRY(hipEventRecord(d->flow_done, d->stream));
    HIP_TRY(hipStreamWaitEvent(d->tail_stream, d->flow_done, 0));       // the flow masks read U / V on the tail's stream
    const double t3 = now_ms();
    SIND_TRY(d->tail.flow_stage(d->U.p, d->V.p, dso, dyna_out, label_out));
    const double t4 = now_ms();
    d->t_ms[0] += t1 - t0; d->t_ms[1] += t2 - t1; d->t_ms[2] += t3 - t2; d->t_ms[3] += t4 - t3; d->t_ms[4] += t_depth; d->t_ms[5] += t4 - t0; d->n_timed++;
    d->t++;
    return SIND_OK;
}
int sind_dyna_dilate15(sind_dyna* d, uint8_t* io) {
    if (!d || !io) return SIND_E_ARG;
    HIP_TRY(hipSetDevice(d->cfg.device));
    const size_t np = (size_t)d->cfg.W * d->cfg.H;
    SpinScope spin;
    std::memcpy(d->dil_h.p, io, np);
    HIP_TRY(hipMemcpyAsync(d->dil_a.p, d->dil_h.p, np, hipMemcpyHostToDevice, d->stream));
    SIND_TRY(sind::launch_morph(d->stream, d->dil_a.p, d->dil_b.p, d->cfg.W, d->cfg.H, 15, true));
    HIP_TRY(hipMemcpyAsync(d->dil_h.p, d->dil_b.p, np, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(sind_stream_wait(d->stream));
    std::memcpy(io, d->dil_h.p, np);
    return SIND_OK;
}
int sind_dyna_debug(sind_dyna* d, float* flow_deep, float* flow_refined, float* flow_full, double* H9, float* thr5, int* hist256, uint8_t* mask_low,
                    uint8_t* mask_high, uint8_t* kmeans_label, float* centers36, uint8_t* occ1, uint8_t* occ2, uint8_t* total_area, uint8_t* grad_edge,
                    uint8_t* plane_contours, int* info3) {
    if (!d) return SIND_E_ARG;
    HIP_TRY(hipSetDevice(d->cfg.device));
    const size_t np = (size_t)d->cfg.W * d->cfg.H;
    const sind::DynaDebug& g = d->tail.dbg;
    if (flow_deep && !d->deep.empty()) std::memcpy(flow_deep, d->deep.data(), d->deep.size() * 4);
    if (flow_refined && !d->refined.empty()) std::memcpy(flow_refined, d->refined.data(), d->refined.size() * 4);
    if (flow_full) { HIP_TRY(hipMemcpy(flow_full, d->U.p, np * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(flow_full + np, d->V.p, np * 4, hipMemcpyDeviceToHost)); }
    if (H9) std::memcpy(H9, g.H, sizeof(g.H));
    if (thr5) { thr5[0] = g.maxError; thr5[1] = g.otsu; thr5[2] = g.triangle; thr5[3] = g.thr_low; thr5[4] = g.thr_high; }
    if (hist256) std::memcpy(hist256, g.hist, sizeof(g.hist));
    auto put = [&](uint8_t* dst, const std::vector<uint8_t>& src) { if (dst && src.size() == np) std::memcpy(dst, src.data(), np); };
    put(mask_low, g.maskLow); put(mask_high, g.maskHigh); put(kmeans_label, g.kmeansLabel); put(occ1, g.occ1); put(occ2, g.occ2);
    put(total_area, g.totalArea); put(grad_edge, g.gradEdge); put(plane_contours, g.planeContours);
    if (centers36) std::memcpy(centers36, g.centers, sizeof(g.centers));
    if (info3) { info3[0] = d->largeMotion; info3[1] = g.nPairs; info3[2] = g.nClusters; }
    return SIND_OK;
}

// parity-test access to the bit-plane dilation of the RAG stage (k_dilate_planes): planes are [nplanes][height][ceil(width / 64)] words, host memory
int sind_debug_dilate_planes(const unsigned long long* planes, int nplanes, int width, int height, int n, int device, unsigned long long* out) {
    if (!planes || !out || nplanes < 1 || width < 1 || height < 1 || n < 1 || n > MORPH_MAX) { sind_set_error("sind_debug_dilate_planes: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    const size_t words = (size_t)nplanes * height * ((width + 63) / 64);
    DevBuf<unsigned long long> a, b; SIND_TRY(a.alloc(words)); SIND_TRY(b.alloc(words));
    HIP_TRY(hipMemcpy(a.p, planes, words * 8, hipMemcpyHostToDevice));
    SIND_TRY(sind::launch_dilate_planes(nullptr, a.p, b.p, nplanes, width, height, n));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, b.p, words * 8, hipMemcpyDeviceToHost));
    return SIND_OK;
}

// parity-test access to the GPU region grow of the PEAC refinement (k_peac_grow): n depth frames (host, [n][height][width] u16) go through the block
// statistics kernel and the host graph clustering (PeacFitter::part1); then ONE launch grows all frames on the GPU and the host statement of the same
// FIFO grows them one by one.  member_* [n][height*width] int8 (plane or -1), pair_* [n][127*127] (row stride = the frame's plane count), status [n][4] =
// kernel status, BFS levels, seeds processed, plane count.  A frame the kernel skips (capacities) reports status 4 and undefined GPU outputs.
int sind_debug_peac_grow(const uint16_t* depth, int n, int width, int height, float fx, float fy, float cx, float cy, float depth_scale, int device,
                         int8_t* member_gpu, int8_t* member_host, uint8_t* pair_gpu, uint8_t* pair_host, int* status) {
    if (!depth || n < 1 || !member_gpu || !member_host || !pair_gpu || !pair_host || !status) { sind_set_error("sind_debug_peac_grow: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    const size_t N = (size_t)width * height, nblk = (size_t)(width / 16) * (height / 16), PP = (size_t)PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES;
    DevBuf<uint16_t> dd; DevBuf<sind::PeacBlockStats> bd; SIND_TRY(dd.alloc(N * n)); SIND_TRY(bd.alloc(nblk * n));
    HIP_TRY(hipMemcpy(dd.p, depth, N * n * 2, hipMemcpyHostToDevice));
    SIND_TRY(sind::launch_peac_block_stats(nullptr, dd.p, width, height, 16, 16, fx, fy, cx, cy, depth_scale, bd.p, n));
    std::vector<sind::PeacBlockStats> bh(nblk * n);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(bh.data(), bd.p, bh.size() * sizeof(sind::PeacBlockStats), hipMemcpyDeviceToHost));
    sind::PeacGrowBatch batch; SIND_TRY(batch.init(width, height, fx, fy, cx, cy, depth_scale, n));
    PinnedBuf<uint8_t> in_h, pair_h; PinnedBuf<int8_t> mem_h; PinnedBuf<int> st_h;
    SIND_TRY(in_h.alloc((size_t)n * PG_IN_STRIDE)); SIND_TRY(pair_h.alloc(PP * n)); SIND_TRY(mem_h.alloc(N * n)); SIND_TRY(st_h.alloc((size_t)4 * n));
    std::vector<std::unique_ptr<sind::PeacFitter>> fit(n);
    for (int k = 0; k < n; k++) {
        sind::PeacInput pin{bh.data() + nblk * k, depth + N * k, width, height, fx, fy, cx, cy, depth_scale};
        fit[k].reset(new sind::PeacFitter(pin)); fit[k]->part1();
        sind::peac_grow_pack(*fit[k], k, in_h.p + (size_t)k * PG_IN_STRIDE);
    }
    SIND_TRY(batch.run(nullptr, in_h.p, dd.p, n, mem_h.p, pair_h.p, st_h.p));
    HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < N * n; i++) member_gpu[i] = mem_h.p[i] < 0 ? (int8_t)-1 : mem_h.p[i];       // the kernel keeps the number of failed tries in the negative values (-2 .. -6)
    std::memcpy(pair_gpu, pair_h.p, PP * n);
    for (int k = 0; k < n; k++) {
        for (int q = 0; q < 3; q++) status[4 * k + q] = st_h.p[4 * k + q];
        status[4 * k + 3] = fit[k]->n_planes();
        std::vector<int8_t> m8; std::vector<int16_t> m16; std::vector<uint8_t> seen;
        fit[k]->grow_host(m8, m16, seen);
        std::memset(pair_host + PP * k, 0, PP);
        if (!m8.empty()) { std::memcpy(member_host + N * k, m8.data(), N); std::memcpy(pair_host + PP * k, seen.data(), seen.size()); }
        else std::memset(member_host + N * k, 0x80, N);       // more than 127 planes: no int8 map (and the kernel skipped the frame)
    }
    return SIND_OK;
}

}  // extern "C"
