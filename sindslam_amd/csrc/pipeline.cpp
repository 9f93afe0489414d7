// Batched multi-stream pipeline (include/sind_hip.h "sind_pipe"): the frame-loop body of the reference's
// Examples/RGB-D/rgbd_tum_noros.cc:110-170 (DetectDynaArea -> 15x15 dilate -> ORBextractor via Frame::ExtractORB2) for
// S independent streams x T frames per step.
//   phase A (state free, one batch of S*T frames on the shared HIP stream): gray, 0.6 resize, dense flow, ORB front
//   phase B (stateful, frame order inside a stream, streams in parallel on host threads + their own HIP streams):
//            DynaDetect tail, dilation, dynamic-mask erasure of the ORB keypoints.
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include "../../include/sind_hip.h"
#include "dyna.hpp"
#include "orb.hpp"

using namespace sind;

struct sind_pipe {
    sind_pipe_config c{}; DynaConfig dc; int S = 0, T = 0, fw = 0, fh = 0;
    hipStream_t stream = nullptr; std::vector<hipStream_t> tail_streams;
    DynaFront front; OrbEngine orb; std::vector<std::unique_ptr<DynaTail>> tails;
    DevBuf<uint8_t> bgr_d, gray, gray_orb, pool; DevBuf<uint16_t> depth_d; DevBuf<float> U, V;
    std::vector<uint16_t> depth_h; std::vector<char> primed;
    double stage_ms[6] = {0}; double sor_ms = 0, sor_bytes = 0; long long sor_launches = 0;
};

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// max filter of the 0/125/255 image with the 15x15 ellipse = two binary dilations (>=125, ==255)
static void dilate15_codes(const uint8_t* src, int W, int H, uint8_t* dst) {
    const EllipseElem e15(15);
    BitImg hi = BitImg::from_equal(src, W, H, W, 255), any = BitImg::from_u8(src, W, H, W);
    hi = hi.dilated(e15); any = any.dilated(e15);
    std::memset(dst, 0, (size_t)W * H);
    any.paint_u8(dst, W, 125); hi.paint_u8(dst, W, 255);
}

extern "C" {

int sind_pipe_create(const sind_pipe_config* cfg, sind_pipe** out) {
    if (!cfg || !out || cfg->streams < 1 || cfg->frames_per_step < 1 || cfg->width < 64 || cfg->height < 64) { sind_set_error("sind_pipe_create: bad configuration"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(cfg->device));
    std::unique_ptr<sind_pipe> p(new sind_pipe());
    p->c = *cfg; p->S = cfg->streams; p->T = cfg->frames_per_step;
    p->dc.W = cfg->width; p->dc.H = cfg->height; p->dc.fx = cfg->fx; p->dc.fy = cfg->fy; p->dc.cx = cfg->cx; p->dc.cy = cfg->cy; p->dc.depthScale = cfg->depth_scale; p->dc.device = cfg->device;
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    const int B = p->S * p->T; const size_t np = (size_t)cfg->width * cfg->height;
    SIND_TRY(p->front.init(p->dc, B, p->stream));
    p->fw = p->front.fw; p->fh = p->front.fh;
    SIND_TRY(p->orb.init(cfg->width, cfg->height, cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th_fast, cfg->min_th_fast, B, p->stream));
    p->tail_streams.resize(p->S); p->tails.resize(p->S);
    for (int s = 0; s < p->S; s++) {
        HIP_TRY(hipStreamCreateWithFlags(&p->tail_streams[s], hipStreamNonBlocking));
        p->tails[s].reset(new DynaTail()); SIND_TRY(p->tails[s]->init(p->dc, p->tail_streams[s]));
    }
    SIND_TRY(p->gray.alloc(np * B)); SIND_TRY(p->pool.alloc((size_t)p->fw * p->fh * p->S * (p->T + 2)));
    if (cfg->orb_gray_rgb_order) SIND_TRY(p->gray_orb.alloc(np * B));
    SIND_TRY(p->U.alloc(np * B)); SIND_TRY(p->V.alloc(np * B));
    p->depth_h.resize(np * B); p->primed.assign(p->S, 0);
    *out = p.release(); return SIND_OK;
}
int sind_pipe_destroy(sind_pipe* p) {
    if (!p) return SIND_OK;
    (void)hipSetDevice(p->c.device);
    (void)hipDeviceSynchronize();
    std::vector<hipStream_t> ss = p->tail_streams; ss.push_back(p->stream);
    delete p;
    for (hipStream_t s : ss) if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}
int sind_pipe_prime(sind_pipe* p, int s, const uint8_t* last, const uint8_t* lastlast) {
    if (!p || s < 0 || s >= p->S || !last || !lastlast) { sind_set_error("sind_pipe_prime: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    const size_t np = (size_t)p->c.width * p->c.height, fb = (size_t)p->fw * p->fh;
    SIND_TRY(p->bgr_d.alloc(np * 3 * 2));
    HIP_TRY(hipMemcpyAsync(p->bgr_d.p, lastlast, np * 3, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->bgr_d.p + np * 3, last, np * 3, hipMemcpyHostToDevice, p->stream));
    SIND_TRY(p->front.gray_and_min(p->bgr_d.p, 2, p->gray.p, p->pool.p + fb * (size_t)s * (p->T + 2)));   // slots 0 (n-2), 1 (n-1)
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->tails[s]->reset(); p->primed[s] = 1;
    return SIND_OK;
}

int sind_pipe_process_dev(sind_pipe* p, const uint8_t* bgr_dev, const uint16_t* depth_dev, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil,
                          sind_keypoint* kps, int cap, int* nkp, uint8_t* desc) {
    if (!p || !bgr_dev || !depth_dev) { sind_set_error("sind_pipe_process: null input"); return SIND_E_ARG; }
    for (int s = 0; s < p->S; s++) if (!p->primed[s]) { sind_set_error("sind_pipe_process: stream %d was not primed", s); return SIND_E_STATE; }
    HIP_TRY(hipSetDevice(p->c.device));
    const int S = p->S, T = p->T, B = S * T, W = p->c.width, H = p->c.height;
    const size_t np = (size_t)W * H, fb = (size_t)p->fw * p->fh;
    const double t0 = now_ms();
    // ---- phase A1: gray for all frames, 0.6-scaled gray into the per-stream pools behind the two history slots
    SIND_TRY(launch_bgr2gray(p->stream, bgr_dev, p->gray.p, np * B, false));
    for (int s = 0; s < S; s++)
        SIND_TRY(launch_resize_u8(p->stream, p->gray.p + np * (size_t)s * T, p->pool.p + fb * ((size_t)s * (T + 2) + 2), W, H, p->fw, p->fh, T, W, p->fw, np, fb));
    const uint8_t* gray_for_orb = p->gray.p;
    if (p->c.orb_gray_rgb_order) { SIND_TRY(launch_bgr2gray(p->stream, bgr_dev, p->gray_orb.p, np * B, true)); gray_for_orb = p->gray_orb.p; }
    // depth to the host (PEAC region grow and cluster centres read single pixels); overlaps with the flow below
    HIP_TRY(hipMemcpyAsync(p->depth_h.data(), depth_dev, np * B * sizeof(uint16_t), hipMemcpyDeviceToHost, p->stream));
    const double t1 = now_ms();
    // ---- phase A2: dense flow for every (n, n-2) pair, second pass for large-motion pairs, refinement, up-scale
    std::vector<int> cur(B), p1(B), p2(B);
    for (int s = 0; s < S; s++) for (int t = 0; t < T; t++) { const int k = s * T + t, base = s * (T + 2) + t; cur[k] = base + 2; p1[k] = base + 1; p2[k] = base; }
    p->front.flow.sor_timer.enabled = true; p->front.flow.sor_timer.reset();
    SIND_TRY(p->front.dense_flow(p->pool.p, cur.data(), p1.data(), p2.data(), B, p->U.p, p->V.p, nullptr));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->sor_ms = p->front.flow.sor_timer.collect_ms(); p->sor_bytes = p->front.flow.sor_timer.alg_bytes; p->sor_launches = p->front.flow.sor_timer.launches;
    const double t2 = now_ms();
    // ---- phase A3: ORB front (pyramid, FAST, octree, orientation, blur, BRIEF) for all frames
    std::vector<OrbFrameResult> orb_all;
    SIND_TRY(p->orb.extract_all(gray_for_orb, B, orb_all));
    // roll the gray history: the last two frames of every stream become slots 0, 1
    for (int s = 0; s < S; s++) {
        uint8_t* base = p->pool.p + fb * (size_t)s * (T + 2);
        if (T >= 2) HIP_TRY(hipMemcpyAsync(base, base + fb * T, fb * 2, hipMemcpyDeviceToDevice, p->stream));
        else { HIP_TRY(hipMemcpyAsync(base, base + fb, fb, hipMemcpyDeviceToDevice, p->stream)); HIP_TRY(hipMemcpyAsync(base + fb, base + fb * 2, fb, hipMemcpyDeviceToDevice, p->stream)); }
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    const double t3 = now_ms();
    // ---- phase B: stateful tails, one host thread per stream (or a bounded pool)
    std::vector<int> rc(S, SIND_OK); std::vector<std::string> err(S);
    int nthreads = p->c.host_threads > 0 ? std::min(p->c.host_threads, S) : S;
    std::vector<std::thread> th;
    auto work = [&](int tid) {
        for (int s = tid; s < S; s += nthreads) {
            std::vector<uint8_t> dy(np), lb(np), dil(np);
            for (int t = 0; t < T && rc[s] == SIND_OK; t++) {
                const int k = s * T + t;
                int r = p->tails[s]->process(p->depth_h.data() + np * k, depth_dev + np * k, p->U.p + np * k, p->V.p + np * k, dy.data(), lb.data());
                if (r != SIND_OK) { rc[s] = r; err[s] = sind_last_error(); break; }
                dilate15_codes(dy.data(), W, H, dil.data());
                if (dyna) std::memcpy(dyna + np * k, dy.data(), np);
                if (label) std::memcpy(label + np * k, lb.data(), np);
                if (mask_dil) std::memcpy(mask_dil + np * k, dil.data(), np);
                std::vector<OrbKeyPoint> kk; std::vector<uint8_t> dd;
                p->orb.finish(orb_all[k], dil.data(), W, kk, dd);
                if ((int)kk.size() > cap && kps) { rc[s] = SIND_E_CAPACITY; err[s] = "keypoint capacity exceeded"; break; }
                if (nkp) nkp[k] = (int)kk.size();
                if (kps) std::memcpy(kps + (size_t)k * cap, kk.data(), kk.size() * sizeof(sind_keypoint));
                if (desc) std::memcpy(desc + (size_t)k * cap * 32, dd.data(), dd.size());
            }
        }
    };
    for (int i = 0; i < nthreads; i++) th.emplace_back(work, i);
    for (auto& t : th) t.join();
    const double t4 = now_ms();
    for (int s = 0; s < S; s++) if (rc[s] != SIND_OK) { sind_set_error("stream %d: %s", s, err[s].c_str()); return rc[s]; }
    p->stage_ms[0] = t1 - t0; p->stage_ms[1] = t2 - t1; p->stage_ms[2] = t3 - t2; p->stage_ms[3] = 0; p->stage_ms[4] = t4 - t3; p->stage_ms[5] = t4 - t0;
    return SIND_OK;
}

int sind_pipe_process(sind_pipe* p, const uint8_t* bgr, const uint16_t* depth, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil, sind_keypoint* kps,
                      int cap, int* nkp, uint8_t* desc) {
    if (!p || !bgr || !depth) { sind_set_error("sind_pipe_process: null input"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    const size_t np = (size_t)p->c.width * p->c.height, B = (size_t)p->S * p->T;
    SIND_TRY(p->bgr_d.alloc(np * 3 * B)); SIND_TRY(p->depth_d.alloc(np * B));
    HIP_TRY(hipMemcpy(p->bgr_d.p, bgr, np * 3 * B, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->depth_d.p, depth, np * B * 2, hipMemcpyHostToDevice));
    return sind_pipe_process_dev(p, p->bgr_d.p, p->depth_d.p, dyna, label, mask_dil, kps, cap, nkp, desc);
}

int sind_pipe_stats(sind_pipe* p, double* stage_ms6, long long* sor_launches, double* sor_ms, double* sor_alg_bytes) {
    if (!p) return SIND_E_ARG;
    if (stage_ms6) std::memcpy(stage_ms6, p->stage_ms, sizeof(p->stage_ms));
    if (sor_launches) *sor_launches = p->sor_launches;
    if (sor_ms) *sor_ms = p->sor_ms;
    if (sor_alg_bytes) *sor_alg_bytes = p->sor_bytes;
    return SIND_OK;
}

}  // extern "C"
