// Host orchestration of the batched dense-flow path (OpticalFlowDeepFlow::calc and VariationalRefinement::calc
// call structure, opencv_contrib 4.2.0 deepflow.cpp / opencv 4.2.0 variational_refinement.cpp; call sites in the
// reference: DynaDetect.cc:1031, 1075, 1127, 1133-1143).  All work is enqueued on one HIP stream; nothing here syncs.
#include <cmath>
#include <cstdarg>
#include "flow.hpp"

namespace sind {

static std::vector<std::pair<int, int>> deepflow_sizes(int w, int h) {
    const float downscale = 0.95f; const int minSize = 25, maxLayers = 200;
    std::vector<std::pair<int, int>> s; s.push_back({w, h});
    for (int i = 0; i < maxLayers; i++) {
        int nw = (int)(s.back().first * downscale + 0.5f), nh = (int)(s.back().second * downscale + 0.5f);
        if (nh <= minSize || nw <= minSize) break;
        s.push_back({nw, nh});
    }
    return s;
}

int FlowEngine::init(int fw_, int fh_, int maxB_, hipStream_t s) {
    fw = fw_; fh = fh_; maxB = maxB_; stream = s;
    if (const char* e = sind_lab_env("SIND_SOR_MODE")) solver.mode = atoi(e);       // see SolverCfg (flow.hpp); the tile / fuse variables are for A/B timing
    if (const char* e = sind_lab_env("SIND_SOR_TILEW")) solver.tile_w = atoi(e);
    if (const char* e = sind_lab_env("SIND_SOR_FUSE")) solver.fuse = std::max(0, std::min(atoi(e), 12));
    if (const char* e = sind_lab_env("SIND_SOR_PLAN_COST")) solver.plan_cost = std::max(0.0, atof(e));
    if (const char* e = sind_lab_env("SIND_SOR_XCD")) solver.xcd = atoi(e) != 0;
    if (const char* e = sind_lab_env("SIND_SOR_TILEH")) solver.tile_h = atoi(e);
    if (const char* e = sind_lab_env("SIND_SOR_STREAM_MINB")) solver.stream_min_b = atoi(e);
    if (const char* e = sind_lab_env("SIND_SOR_STREAM_MINPX")) solver.stream_min_px = atoi(e);       // images per launch from which the tiled levels take the streaming kernel
    if (const char* e = sind_lab_env("SIND_LAUNCH_AHEAD")) launch_ahead = std::max(0, atoi(e));       // 0: unbounded
    levels = deepflow_sizes(fw, fh);
    level_off.clear(); pyr_pixels = 0;
    for (auto& l : levels) { level_off.push_back(pyr_pixels); pyr_pixels += (size_t)l.first * l.second; }
    const size_t n0 = (size_t)fw * fh * maxB;
    static_assert(sizeof(FlowPlanes) == 18 * sizeof(float*), "FlowPlanes is filled as an array of 18 plane pointers");
    SIND_TRY(plane_store.alloc(n0 * 18 + 16));           // + 16: the streaming solver's row loader reads whole 16-byte chunks (up to 3 floats past a row's end)
    if (hipMemsetAsync(plane_store.p, 0, (n0 * 18 + 16) * sizeof(float), stream) != hipSuccess) { sind_set_error("hipMemset(plane store) failed"); return SIND_E_HIP; }     // (the loader's over-reads must find finite values)
    float** f = reinterpret_cast<float**>(&planes);
    for (int i = 0; i < 18; i++) f[i] = plane_store.p + n0 * i;
    SIND_TRY(pyr0.alloc(pyr_pixels * maxB));
    SIND_TRY(pyr1.alloc(pyr_pixels * maxB));
    return SIND_OK;
}

int FlowEngine::deepflow(const uint8_t* g0, const uint8_t* g1, int B, float* u, float* v) {
    if (B < 1 || B > maxB) { sind_set_error("deepflow: batch %d outside [1,%d]", B, maxB); return SIND_E_ARG; }
    // getGaussianKernel(3, 0.6): exp(-x^2/(2 sigma^2)) normalised, cast to float
    const double sigma = 0.6f; const double e = std::exp(-0.5 * 1.0 * 1.0 / (sigma * sigma)), sum = (e + 1.0) + e;
    const float k0 = (float)(1.0 / sum), k1 = (float)(e / sum);
    const int L = max_levels > 0 ? std::min(max_levels, (int)levels.size()) : (int)levels.size();
    SIND_TRY(launch_u8_to_f32_blur3(stream, g0, level_ptr(pyr0, 0, B), fw, fh, B, k0, k1, true));
    SIND_TRY(launch_u8_to_f32_blur3(stream, g1, level_ptr(pyr1, 0, B), fw, fh, B, k0, k1, true));
    // 0.95 pyramid of both images: one launch per level for the pair while the level is large, the small levels (<= 12 k pixels) in one launch
    int lt = 1; while (lt < L && (size_t)levels[lt - 1].first * levels[lt - 1].second > 12288) lt++;
    lt = std::max(lt, L - 1 - 40 + 1);                          // the tail kernel takes at most 40 levels
    for (int l = 1; l < std::min(lt, L); l++)
        SIND_TRY(launch_resize_f32_pair(stream, level_ptr(pyr0, l - 1, B), level_ptr(pyr0, l, B), level_ptr(pyr1, l - 1, B), level_ptr(pyr1, l, B), levels[l - 1].first, levels[l - 1].second,
                                        levels[l].first, levels[l].second, B, 1.f, false));
    if (lt < L) SIND_TRY(launch_pyramid_tail(stream, pyr0.p, pyr1.p, levels, level_off, lt - 1, L - 1, B));
    VarParams V;   // OpticalFlowDeepFlow defaults: alpha 1, delta 0.5, gamma 5 -> 4*alpha, delta/3, gamma/3; 5 x 25, omega 1.6
    V.alpha = 4 * 1.0f; V.delta = 0.5f / 3; V.gamma = 5.0f / 3; V.fixedPointIterations = 5; V.sorIterations = 25; V.omega = 1.6f;
    const float inv_scale = 1.0f / 0.95f;
    FlowPlanes& P = planes;
    // the coarsest levels that are one workgroup's work each go through ONE launch (flow_coarse.hip), which leaves the flow up-sampled to the first level above them
    int lc = L;
    if (coarse_chain && V.epsilon >= 1e-12f) while (lc > 0 && L - lc < 32 && coarse_level_P(levels[lc - 1].first, levels[lc - 1].second)) lc--;
    while ((int)level_done.size() < L) { hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); level_done.push_back(ev); }
    if (lc < L) {
        SIND_TRY(launch_coarse_chain(stream, P, pyr0.p, pyr1.p, levels, level_off, L - 1, lc, B, V, true, lc > 0, inv_scale, P.dWu2, P.dWv2));
        if (lc > 0) { std::swap(P.Wu, P.dWu2); std::swap(P.Wv, P.dWv2); }
        for (int l = L - 1; l >= lc; --l) HIP_TRY(hipEventRecord(level_done[l], stream));
    } else {
        const size_t nc = (size_t)levels[L - 1].first * levels[L - 1].second * B;
        HIP_TRY(hipMemsetAsync(P.Wu, 0, nc * sizeof(float), stream));
        HIP_TRY(hipMemsetAsync(P.Wv, 0, nc * sizeof(float), stream));
    }
    // One level is ~36 launches.  A thread that enqueues the whole pyramid runs far ahead of the GPU, fills the queue and then SPINS inside
    // the launch call for the rest of the solve (measured: a slice thread burnt a full core, 215 ms per step); so the thread stays at
    // most `launch_ahead` levels ahead and sleeps on the level events instead.
    bool have_buffers = false;
    for (int l = lc - 1; l >= 0; --l) {
        const int w = levels[l].first, h = levels[l].second;
        if (launch_ahead > 0 && l + launch_ahead < L) HIP_TRY(sind_event_wait(level_done[l + launch_ahead]));
        const bool fuse_up = level_up && l > 0 && solver.mode != 0;
        SIND_TRY(varref_level(stream, P, level_ptr(pyr0, l, B), level_ptr(pyr1, l, B), w, h, B, V, &sor_timer, opts(), have_buffers, fuse_up));
        HIP_TRY(hipEventRecord(level_done[l], stream));
        have_buffers = false;
        if (l > 0) {
            const int nw = levels[l - 1].first, nh = levels[l - 1].second;
            if (fuse_up) { SIND_TRY(launch_level_up(stream, P, w, h, level_ptr(pyr0, l - 1, B), level_ptr(pyr1, l - 1, B), nw, nh, B, inv_scale)); have_buffers = true; }
            else {
                SIND_TRY(launch_resize_f32_pair(stream, P.Wu, P.tWu, P.Wv, P.tWv, w, h, nw, nh, B, inv_scale, true));
                std::swap(P.Wu, P.tWu); std::swap(P.Wv, P.tWv);
            }
        }
    }
    const size_t n = (size_t)fw * fh * B;
    HIP_TRY(hipMemcpyAsync(u, P.Wu, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v, P.Wv, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SIND_OK;
}

int FlowEngine::refine(const uint8_t* g0, const uint8_t* g1, int B, float* u, float* v) {
    if (B < 1 || B > maxB) { sind_set_error("refine: batch %d outside [1,%d]", B, maxB); return SIND_E_ARG; }
    float* I0 = level_ptr(pyr0, 0, B); float* I1 = level_ptr(pyr1, 0, B);
    SIND_TRY(launch_u8_to_f32_blur3(stream, g0, I0, fw, fh, B, 1.f, 0.f, false));
    SIND_TRY(launch_u8_to_f32_blur3(stream, g1, I1, fw, fh, B, 1.f, 0.f, false));
    return varref_f32(I0, I1, fw, fh, B, u, v, VarParams());
}

int FlowEngine::varref_f32(const float* I0, const float* I1, int w, int h, int B, float* u, float* v, const VarParams& V) {
    if (B < 1 || B > maxB || (size_t)w * h > (size_t)fw * fh) { sind_set_error("varref_f32: bad shape"); return SIND_E_ARG; }
    const size_t n = (size_t)w * h * B;
    HIP_TRY(hipMemcpyAsync(planes.Wu, u, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(planes.Wv, v, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    SIND_TRY(varref_level(stream, planes, I0, I1, w, h, B, V, &sor_timer, opts()));
    HIP_TRY(hipMemcpyAsync(u, planes.Wu, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v, planes.Wv, n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SIND_OK;
}

}  // namespace sind

// ---------------------------------------------------------------- process set-up
// HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (runtime default 4).  A pipeline handle drives ~45 streams; with four queues its flow slices share queues
// with the tail streams and a stand-alone sequence job runs 17 % slower, six keep them apart, ten and more are catastrophic (profiles/r04/hw_queues.txt).  The runtime reads
// the variable when it initialises (the first HIP call of the process), so the library sets it when it is LOADED -- for the C++ integrator who links libsind_hip.so as much
// as for the Python mirror.  An explicit setting in the environment wins; a process whose HIP runtime is already up keeps what it started with.
__attribute__((constructor)) static void sind_process_setup() { setenv("GPU_MAX_HW_QUEUES", "6", 0); }

// ---------------------------------------------------------------- error string (shared by the whole library)
static thread_local char g_err[512] = "";
std::atomic<long long> g_sind_wait_ns{0}, g_sind_wait_calls{0};
thread_local SindHostGate* t_sind_gate = nullptr;
thread_local int t_sind_spin_us = 0;

#include <dlfcn.h>
#include <cstring>
#include <cstdlib>
namespace {
struct Roctx { int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    // The marker library is only looked for when a profiler is attached (rocprofv3 preloads librocprofiler-sdk-tool / sets ROCP_TOOL_LIBRARIES) or SIND_ROCTX=1
    // asks for it: a production process does not dlopen profiler libraries.
    static bool wanted() {
        if (const char* e = getenv("SIND_ROCTX")) return atoi(e) != 0;
        for (const char* v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH"}) if (getenv(v)) return true;
        const char* pre = getenv("LD_PRELOAD"); return pre && strstr(pre, "rocprofiler");
    }
    Roctx() {
        if (!wanted()) return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void* so = dlopen(lib, RTLD_NOW | RTLD_LOCAL)) {
                push = (int (*)(const char*))dlsym(so, "roctxRangePushA"); pop = (int (*)())dlsym(so, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr; (void)dlclose(so);
            }
        }
    } };
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void sind_range_push(const char* name) { if (roctx().push) (void)roctx().push(name); }
void sind_range_pop() { if (roctx().pop) (void)roctx().pop(); }
void sind_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
extern "C" const char* sind_last_error() { return g_err; }
