// C ABI: dense flow stage (include/sind_hip.h).
#include "../../include/sind_hip.h"
#include "flow.hpp"

struct sind_flow {
    int device = 0; hipStream_t stream = nullptr; sind::FlowEngine eng;
    DevBuf<uint8_t> g0, g1; DevBuf<float> u, v, f0, f1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#ifdef SIND_LAB
namespace sind { int debug_ss_profile(unsigned long long* out, int reset); }
#endif

extern "C" {

int sind_flow_set_max_levels(sind_flow* f, int n) { if (!f || n < 0) return SIND_E_ARG; f->eng.max_levels = n; return SIND_OK; }
int sind_flow_set_coarse_chain(sind_flow* f, int on) { if (!f) return SIND_E_ARG; f->eng.coarse_chain = on != 0; return SIND_OK; }
int sind_flow_set_level_up(sind_flow* f, int on) { if (!f) return SIND_E_ARG; f->eng.level_up = on != 0; return SIND_OK; }
int sind_flow_set_latency_tiles(sind_flow* f, int on) { if (!f) return SIND_E_ARG; f->eng.latency_tiles = on != 0; return SIND_OK; }
int sind_device_count(int* count) { if (!count) return SIND_E_ARG; HIP_TRY(hipGetDeviceCount(count)); return SIND_OK; }

int sind_flow_create(int fw, int fh, int max_batch, int device, sind_flow** out) {
    if (!out || fw < 32 || fh < 32 || max_batch < 1) { sind_set_error("sind_flow_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    sind_flow* f = new sind_flow(); f->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&f->ev0)); HIP_TRY(hipEventCreate(&f->ev1));
    int r = f->eng.init(fw, fh, max_batch, f->stream);
    if (r != SIND_OK) { delete f; return r; }
    *out = f; return SIND_OK;
}
int sind_flow_destroy(sind_flow* f) {
    if (!f) return SIND_OK;
    (void)hipSetDevice(f->device);
    if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    delete f; return SIND_OK;
}
int sind_flow_levels(sind_flow* f, int* ws, int* hs, int cap) {
    if (!f) return SIND_E_ARG;
    for (int i = 0; i < (int)f->eng.levels.size() && i < cap; i++) { ws[i] = f->eng.levels[i].first; hs[i] = f->eng.levels[i].second; }
    return (int)f->eng.levels.size();
}
static int stage_in(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B) {
    const size_t n = (size_t)f->eng.fw * f->eng.fh * B;
    SIND_TRY(f->g0.alloc(n)); SIND_TRY(f->g1.alloc(n)); SIND_TRY(f->u.alloc(n)); SIND_TRY(f->v.alloc(n));
    HIP_TRY(hipMemcpyAsync(f->g0.p, i0, n, hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->g1.p, i1, n, hipMemcpyHostToDevice, f->stream));
    return SIND_OK;
}
int sind_flow_deepflow(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v) {
    if (!f || !i0 || !i1 || !u || !v || B < 1 || B > f->eng.maxB) { sind_set_error("sind_flow_deepflow: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    SIND_TRY(stage_in(f, i0, i1, B));
    SIND_TRY(f->eng.deepflow(f->g0.p, f->g1.p, B, f->u.p, f->v.p));
    const size_t n = (size_t)f->eng.fw * f->eng.fh * B;
    HIP_TRY(hipMemcpyAsync(u, f->u.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipMemcpyAsync(v, f->v.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    return SIND_OK;
}
int sind_flow_refine(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v) {
    if (!f || !i0 || !i1 || !u || !v || B < 1 || B > f->eng.maxB) { sind_set_error("sind_flow_refine: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    SIND_TRY(stage_in(f, i0, i1, B));
    const size_t n = (size_t)f->eng.fw * f->eng.fh * B;
    HIP_TRY(hipMemcpyAsync(f->u.p, u, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->v.p, v, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    SIND_TRY(f->eng.refine(f->g0.p, f->g1.p, B, f->u.p, f->v.p));
    HIP_TRY(hipMemcpyAsync(u, f->u.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipMemcpyAsync(v, f->v.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    return SIND_OK;
}
int sind_flow_varref_f32(sind_flow* f, const float* i0, const float* i1, int w, int h, int B, float* u, float* v,
                         int fp, int sor, float alpha, float delta, float gamma, float omega) {
    if (!f || !i0 || !i1 || !u || !v || B < 1 || B > f->eng.maxB || w < 2 || h < 2 || (size_t)w * h > (size_t)f->eng.fw * f->eng.fh) { sind_set_error("sind_flow_varref_f32: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    const size_t n = (size_t)w * h * B;
    SIND_TRY(f->f0.alloc(n)); SIND_TRY(f->f1.alloc(n)); SIND_TRY(f->u.alloc(n)); SIND_TRY(f->v.alloc(n));
    HIP_TRY(hipMemcpyAsync(f->f0.p, i0, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->f1.p, i1, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->u.p, u, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->v.p, v, n * sizeof(float), hipMemcpyHostToDevice, f->stream));
    sind::VarParams V; V.fixedPointIterations = fp; V.sorIterations = sor; V.alpha = alpha; V.delta = delta; V.gamma = gamma; V.omega = omega;
    SIND_TRY(f->eng.varref_f32(f->f0.p, f->f1.p, w, h, B, f->u.p, f->v.p, V));
    HIP_TRY(hipMemcpyAsync(u, f->u.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipMemcpyAsync(v, f->v.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    return SIND_OK;
}
int sind_flow_deepflow_dev(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v) {
    if (!f || !i0 || !i1 || !u || !v) { sind_set_error("sind_flow_deepflow_dev: null argument"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    return f->eng.deepflow(i0, i1, B, u, v);
}
int sind_flow_refine_dev(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v) {
    if (!f || !i0 || !i1 || !u || !v) { sind_set_error("sind_flow_refine_dev: null argument"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    return f->eng.refine(i0, i1, B, u, v);
}
int sind_debug_rcp_scan(int device, int exp_lo, int exp_hi, unsigned long long out[3]) {
    if (!out || exp_lo < -100 || exp_hi > 100 || exp_lo > exp_hi) { sind_set_error("sind_debug_rcp_scan: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    DevBuf<unsigned long long> d; SIND_TRY(d.alloc(3));
    const unsigned long long init[3] = {0, 0, ~0ull};
    HIP_TRY(hipMemcpy(d.p, init, sizeof(init), hipMemcpyHostToDevice));
    SIND_TRY(sind::debug_rcp_scan(nullptr, exp_lo, exp_hi, d.p));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d.p, sizeof(init), hipMemcpyDeviceToHost));
    return SIND_OK;
}
int sind_debug_coef_math_scan(int device, int exp_lo, int exp_hi, const float numer[3], unsigned long long out[2]) {
    if (!out || !numer || exp_lo < -96 || exp_hi > 100 || exp_lo > exp_hi) { sind_set_error("sind_debug_coef_math_scan: bad arguments (exponents -96 .. 100)"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    DevBuf<unsigned long long> d; SIND_TRY(d.alloc(2));
    HIP_TRY(hipMemset(d.p, 0, 2 * sizeof(unsigned long long)));
    SIND_TRY(sind::debug_coef_math_scan(nullptr, exp_lo, exp_hi, numer, d.p));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return SIND_OK;
}
#ifdef SIND_LAB
int sind_lab_ss_profile(unsigned long long* out96, int reset) { return sind::debug_ss_profile(out96, reset); }      // lab builds only: k_sor_stream's per-wave step cycles
#endif

int sind_debug_flow_thresholds(const int* hist, int n, int width, int height, int variant, int device, int* res, double* mu1) {
    if (!hist || !res || n < 1 || variant < 0 || variant > 2 || width < 1 || height < 1) { sind_set_error("sind_debug_flow_thresholds: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    DevBuf<int> dh, dr; DevBuf<double> dm; SIND_TRY(dh.alloc((size_t)n * 257)); SIND_TRY(dr.alloc((size_t)n * 261)); if (mu1) SIND_TRY(dm.alloc((size_t)n * 256));
    HIP_TRY(hipMemcpy(dh.p, hist, (size_t)n * 257 * sizeof(int), hipMemcpyHostToDevice)); HIP_TRY(hipMemset(dr.p, 0, (size_t)n * 261 * sizeof(int)));
    SIND_TRY(sind::debug_flow_thresholds(nullptr, dh.p, n, width, height, variant, dr.p, mu1 ? dm.p : nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(res, dr.p, (size_t)n * 261 * sizeof(int), hipMemcpyDeviceToHost));
    if (mu1) HIP_TRY(hipMemcpy(mu1, dm.p, (size_t)n * 256 * sizeof(double), hipMemcpyDeviceToHost));
    if (variant == 2) {                                 // the working histograms must have been cleared: hand them back in the first words of every result block
        std::vector<int> back((size_t)n * 257); HIP_TRY(hipMemcpy(back.data(), dh.p, back.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) for (int k = 0; k < 257; k++) res[(size_t)i * 261 + k] = back[(size_t)i * 257 + k];
    }
    return SIND_OK;
}
int sind_flow_set_sor_tiled(sind_flow* f, int mode, int fuse, int tile_w, int tile_h) {
    if (!f) { sind_set_error("sind_flow_set_sor_tiled: null handle"); return SIND_E_ARG; }
    const int nt = tile_w * tile_h / 8;
    if (mode < 0 || mode > 6 || fuse < 0 || fuse > 12 || tile_w < 16 || tile_w % 8 || tile_h < 8 || tile_h % 2 || nt % 128 || nt > 1024 || 4 * fuse >= tile_w || 4 * fuse >= tile_h ||
        (mode == 3 && nt != 256 && nt != 384 && nt != 768)) {
        sind_set_error("sind_flow_set_sor_tiled: bad arguments (mode %d, fuse %d, tile %d x %d)", mode, fuse, tile_w, tile_h); return SIND_E_ARG;
    }
#ifndef SIND_LAB
    // the shipped library carries the per-colour reference (0), the tiled kernel with the reciprocal formed on the fly (4, fixed fuse depth) and the streaming
    // kernel (5); IEEE-division, reciprocal-plane and 1 x 4-strip variants and the per-level fuse plans are lab builds (make -C sindslam_amd/csrc lab)
    if ((mode != 0 && mode != 4 && mode != 5 && mode != 6) || fuse == 0) { sind_set_error("sind_flow_set_sor_tiled: solver variant (mode %d, fuse %d) exists in lab builds only", mode, fuse); return SIND_E_ARG; }
#endif
    sind::SolverCfg& C = f->eng.solver; C.mode = mode; C.fuse = fuse; C.tile_w = tile_w; C.tile_h = tile_h; return SIND_OK;
}
int sind_flow_set_solver_workgroups(sind_flow* f, int cap) { if (!f || cap < 0) return SIND_E_ARG; f->eng.solver.stream_wg_cap = cap; return SIND_OK; }
int sind_flow_set_wave_solver(sind_flow* f, int on, int target_items, int bands) {
    if (!f || target_items < 0 || bands < 0 || on < 0 || on > 4) return SIND_E_ARG;
    if (on > 1) f->eng.solver.wave_prefetch = std::min(on - 1, 3);       // (experiment: on = 2 / 3 / 4 selects 1 / 2 / 3 rows in flight)
    f->eng.solver.wave = on ? 1 : 0; f->eng.solver.wave_items = target_items > 0 ? target_items : sind::SolverCfg().wave_items; f->eng.solver.wave_bands = bands; return SIND_OK;
}
int sind_flow_wave_layout(int w, int h, int B, int target_items, int bands, int out[4]) {
    if (w < 1 || h < 1 || B < 1 || target_items < 0 || bands < 0 || !out) return SIND_E_ARG;
    sind::sor_wave_layout(w, h, B, target_items > 0 ? target_items : sind::SolverCfg().wave_items, bands, &out[0], &out[1], &out[2], &out[3]); return SIND_OK;
}
int sind_flow_set_coef_kernel(sind_flow* f, int variant) { if (!f || variant < 0 || variant > 3) return SIND_E_ARG; f->eng.solver.coef_kernel = variant == 3 ? 1 : variant; f->eng.solver.coef_xcd = variant == 3 ? 0 : 1; return SIND_OK; }
int sind_flow_set_sor(sind_flow* f, int mode, int fuse, int tile_w) {
    if (tile_w != 64 && tile_w != 128) { sind_set_error("sind_flow_set_sor: bad arguments"); return SIND_E_ARG; }
    return sind_flow_set_sor_tiled(f, mode, fuse, tile_w, mode == 3 ? (tile_w == 64 ? 48 : 48) : 64);
}
int sind_lab_build(void) {
#ifdef SIND_LAB
    return 1;
#else
    return 0;
#endif
}
int sind_flow_sync(sind_flow* f) { if (!f) return SIND_E_ARG; HIP_TRY(hipStreamSynchronize(f->stream)); return SIND_OK; }
int sind_flow_timer_begin(sind_flow* f) { if (!f) return SIND_E_ARG; HIP_TRY(hipEventRecord(f->ev0, f->stream)); return SIND_OK; }
int sind_flow_timer_end(sind_flow* f, float* ms) {
    if (!f || !ms) return SIND_E_ARG;
    HIP_TRY(hipEventRecord(f->ev1, f->stream)); HIP_TRY(hipEventSynchronize(f->ev1)); HIP_TRY(hipEventElapsedTime(ms, f->ev0, f->ev1));
    return SIND_OK;
}

}  // extern "C"
