// Flow engine interface (host side of flow_kernels.hip).
#pragma once
#include "common.hpp"

namespace sind {

struct VarParams {      // cv::VariationalRefinement parameters (variational_refinement.cpp defaults)
    int fixedPointIterations = 5, sorIterations = 5;
    float alpha = 20.0f, delta = 5.0f, gamma = 10.0f, omega = 1.6f, zeta = 0.1f, epsilon = 0.001f;
};
struct HMat { double h[9]; };

// per-level working planes (16), each [B][h][w] float, allocated once for the finest level
struct FlowPlanes {
    float *avg, *Iz;                                          // warped-average image and temporal difference (k_coef derives the Sobel images from them)
    float *A11, *A12, *A22, *b1, *b2, *wgt;                   // linear system + smoothness weights
    float *Wu, *Wv, *dWu, *dWv, *tWu, *tWv;                    // level flow, increment, W + dW
    float *dWu2, *dWv2;                                       // ping-pong partner of the increment (tiled fused SOR)
    float *r11, *r22;                                         // RN(1 / A11), RN(1 / A22): the tiled solver divides through them (Markstein)
};

// HIP-event brackets around the SOR launch groups of the flow solver (bench.py's roofline leg): events are recorded on the
// stream the kernels run on; collect() sums the elapsed times after the stream has been synchronised.
struct SorTimer {
    std::vector<hipEvent_t> ev; std::vector<char> kind; size_t used = 0; bool enabled = false;
    // kind 0: launch groups of the STREAMING solver (k_sor_stream, the kernel the roofline object describes); kind 1: every other solver kernel (tiles, one-workgroup levels)
    double alg_bytes = 0, alg_bytes_other = 0; long long launches = 0, launches_other = 0;
    void begin(hipStream_t s) { if (!enabled) return; if (used + 2 > ev.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); ev.push_back(a); ev.push_back(b); kind.push_back(0); } (void)hipEventRecord(ev[used], s); }
    void end(hipStream_t s, long long n_launches, double bytes, int k = 0) {
        if (!enabled) return;
        (void)hipEventRecord(ev[used + 1], s); kind[used / 2] = (char)k; used += 2;
        if (k == 0) { launches += n_launches; alg_bytes += bytes; } else { launches_other += n_launches; alg_bytes_other += bytes; }
    }
    void reset() { used = 0; alg_bytes = alg_bytes_other = 0; launches = launches_other = 0; }
    double collect_ms(int k = 0) { double t = 0; for (size_t i = 0; i + 1 < used; i += 2) { if (kind[i / 2] != k) continue; float ms = 0; if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) t += ms; } return t; }
    // [start, end] of every bracketed launch group of kind k in ms after `base` (an event with timing, recorded earlier on any stream)
    void intervals(hipEvent_t base, std::vector<std::pair<double, double>>& out, int k = 0) { for (size_t i = 0; i + 1 < used; i += 2) { if (kind[i / 2] != k) continue; float a = 0, b = 0; if (hipEventElapsedTime(&a, base, ev[i]) == hipSuccess && hipEventElapsedTime(&b, base, ev[i + 1]) == hipSuccess) out.emplace_back(a, b); } }
    ~SorTimer() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};

int launch_gather_frames(hipStream_t s, const uint8_t* pool, const int* idx_host, int B, uint8_t* out, size_t frame_bytes);
int launch_u8_to_f32_blur3(hipStream_t s, const uint8_t* src, float* dst, int w, int h, int B, float k0, float k1, bool blur);
int launch_resize_f32_pair(hipStream_t s, const float* srcA, float* dstA, const float* srcB, float* dstB, int sw, int sh, int dw, int dh, int B, float post, bool has_post);
int launch_pyramid_tail(hipStream_t s, float* pyrA, float* pyrB, const std::vector<std::pair<int, int>>& levels, const std::vector<size_t>& level_off, int first, int last, int B);
int launch_resize_f32(hipStream_t s, const float* src, float* dst, int sw, int sh, int dw, int dh, int B, float post, bool has_post);
int launch_resize_u8(hipStream_t s, const uint8_t* src, uint8_t* dst, int sw, int sh, int dw, int dh, int B, int s_stride, int d_stride, size_t s_img, size_t d_img, int d_group = 0, int d_skip = 0);       // d_group > 0: d_skip destination slots stay free after every d_group images
int launch_roll_history(hipStream_t s, uint8_t* pool, int S, int T, size_t frame_bytes);      // history slots 0, 1 <- slots T, T + 1 of every stream's T + 2 pool slots
int launch_bgr2gray(hipStream_t s, const uint8_t* bgr, uint8_t* gray, size_t npix, bool swap_rb);
int debug_rcp_scan(hipStream_t s, int exp_lo, int exp_hi, unsigned long long* out_dev);
int debug_coef_math_scan(hipStream_t s, int exp_lo, int exp_hi, const float numer[3], unsigned long long* out_dev);
struct SolverCfg;
int sor_iterations(hipStream_t s, FlowPlanes& P, int w, int h, int B, int total, float omega, long long* nlaunch, const SolverCfg& C, int* streamed = nullptr);
#define FLOW_OPT_COARSE_CHAIN 1      /* levels of <= 4096 pixels: the whole level (all of them, in a pyramid) in one launch (flow_coarse.hip) */
#define FLOW_OPT_LATENCY_TILES 2
#define FLOW_OPT_LEVEL_UP 4          /* W += dW, the up-sampling and the next level's warp in one launch (k_level_up) instead of three */     /* tiled levels with a compute unit per tile (few images): 1024-thread tiles, up to 13 iterations per launch (k_sor_tile) */
// Solver settings of ONE flow handle (every variant returns the same bits; sind_flow_set_sor_tiled / _solver_workgroups / _coef_kernel / _coarse_chain / _latency_tiles).
struct SolverCfg {
    int mode = 4;              // fused register-resident SOR with 1x8 strips: 4 = divisions through a reciprocal formed on the fly (hardware estimate + Newton step, then Markstein's
                               // correction; default), 5 = the streaming kernel on every level it fits, 6 = the one-wave pipeline (k_sor_wave) on every level beyond one workgroup, 0 = one launch per colour (cross-check); lab builds: 1 = IEEE division, 3 = reciprocal
                               // planes held in registers (three waves per SIMD), 2 = 1x4 strips + reciprocal division
    int fuse = 5;              // iterations per launch on the tiled levels; 0 = per-level plan (sor_fuse_plan: measured 1-2 % faster, 10 % more launches; lab builds)
    int tile_w = 64, tile_h = 64;      // extended tile (multiple of 8 wide, even height, tile_w * tile_h / 8 threads)
    int xcd = 1;               // XCD-aware tile order of the fused kernel (0 = plain blockIdx order, for A/B timing)
    int stream_min_b = 80;     // mode 4: tiled levels go to the streaming kernel (one workgroup per image and column strip) from this many images per launch on
                               // (profiles/r05/stream_min_batch.txt: 48 images per launch 907 pairs/s streamed vs 1093-1110 tiled; 112 images 1356 vs 1237; 170 images 1514 vs 1296)
    int stream_min_px = 0;     // ... and only for levels of at least this many pixels
    int stream_wg_cap = 0;     // k_sor_stream: at most this many (persistent) workgroups per launch (0 = one per item)
    int wave = 1;              // mode 4: levels that would go to the streaming kernel go to the one-wave pipeline instead (k_sor_wave, flow_wave.hip); 0 = k_sor_stream (cross-check, A/B timing)
    int wave_items = 1024;     // k_sor_wave: row bands are cut while a launch has fewer waves than this (2048 fill the chip; every cut recomputes 20 rows, and the slices of a step run side by side:
                               // headline step 1571-1607 pairs/s at 1024, 1522-1578 at 680, 1534-1542 at 2048, profiles/r05/ab_wave_bench.txt)
    int wave_bands = 0;        // > 0: exactly this many row bands (tests)
    int wave_prefetch = 2;     // k_sor_wave: steps between a row's request and its take-over (1 .. 3; 16 registers per row in flight)
    int coef_kernel = 1;       // 1: k_coef_lanes (neighbours from lanes; short forms of sqrt and c / sqrt), 2: k_coef_lanes with the IEEE forms, 0: k_coef (neighbours from memory)
    int coef_xcd = 1;          // k_coef_lanes: the tiles of a pair go to one XCD (1) or round-robin over the eight in grid order (0: A/B timing)
    double plan_cost = 14;     // prologue of a tile in iterations (sor_fuse_plan)
    int opts = FLOW_OPT_COARSE_CHAIN | FLOW_OPT_LATENCY_TILES | FLOW_OPT_LEVEL_UP;
};
int varref_level(hipStream_t s, FlowPlanes& P, const float* I0, const float* I1, int w, int h, int B, const VarParams& V, SorTimer* timer, const SolverCfg& C,
                 bool have_buffers = false, bool leave_increment = false);
int launch_level_up(hipStream_t s, FlowPlanes& P, int sw, int sh, const float* I0_next, const float* I1_next, int dw, int dh, int B, float post);
// flow_coarse.hip: the one-workgroup levels of a pyramid (or one such level) in one launch
int coarse_level_P(int w, int h);
int launch_sor_tile(hipStream_t s, FlowPlanes& P, int w, int h, int B, int iters, float omega);
// flow_wave.hip: 5 iterations of a level as one-wave row pipelines (column strips x row bands x images); the result is in P.dWu / P.dWv (swapped with their partners)
int launch_sor_wave(hipStream_t s, FlowPlanes& P, int w, int h, int B, float omega, int target_items, int force_bands, int prefetch);
void sor_wave_layout(int w, int h, int B, int target_items, int force_bands, int* strips, int* IW, int* bands, int* BH);
int sor_tile_count(int w, int h, int iters);
int launch_coarse_chain(hipStream_t s, FlowPlanes& Pl, const float* pyr0, const float* pyr1, const std::vector<std::pair<int, int>>& levels, const std::vector<size_t>& level_off,
                        int first, int last, int B, const VarParams& V, bool init_zero, bool upsample_last, float post, float* out_u, float* out_v);
int launch_mag_stats(hipStream_t s, const float* u, const float* v, float* mag, unsigned* maxbits, int* hist, uint8_t* out_u8, int n, int B);
int launch_residual(hipStream_t s, const float* u, const float* v, const double H[9], float* mag, unsigned* maxbits, int* hist, uint8_t* magu8, int w, int h, bool already_zero = false);
// hist: 257 working words (zeroed again by the kernel), res: 261 words = histogram, maximum, lo / hi / otsu / triangle
int launch_flow_thresholds_and_masks(hipStream_t s, int* hist, int W, int H, int* res, const uint8_t* magu8, uint8_t* low, uint8_t* high);
int debug_flow_thresholds(hipStream_t s, int* hist, int n, int W, int H, int variant, int* res, double* mu1);
int launch_threshold_masks(hipStream_t s, const uint8_t* magu8, float lo, float hi, uint8_t* low, uint8_t* high, int n);
int launch_gather_grid(hipStream_t s, const float* u, const float* v, float* out, int w, int h, int step, int B = 1);
int launch_scale2(hipStream_t s, float* a, float* b, float sc, size_t n);

// Batched DeepFlow + VariationalRefinement for B frame pairs at the flow grid (fw x fh).
class FlowEngine {
public:
    int fw = 0, fh = 0, maxB = 0;
    hipStream_t stream = nullptr;
    std::vector<std::pair<int, int>> levels;     // OpticalFlowDeepFlow::buildPyramid sizes
    std::vector<size_t> level_off;               // offset (in pixels per image) of each level inside a pyramid
    size_t pyr_pixels = 0;
    int init(int fw, int fh, int maxB, hipStream_t s);
    // g0/g1: device u8 [B][fh][fw].  u/v: device f32 [B][fh*fw], raw DeepFlow output (not negated).
    int deepflow(const uint8_t* g0, const uint8_t* g1, int B, float* u, float* v);
    // cv::VariationalRefinement::create()->calc(g0, g1, flow) with defaults; u/v in-out.
    int refine(const uint8_t* g0, const uint8_t* g1, int B, float* u, float* v);
    // expose one-level refinement on float images for stage-level parity tests
    int varref_f32(const float* I0, const float* I1, int w, int h, int B, float* u, float* v, const VarParams& V);
    FlowPlanes planes{};
    SorTimer sor_timer;
    int max_levels = 0;                          // > 0: use only the finest max_levels pyramid levels, zero flow at the coarsest of them (DeepFlow's maxLayers knob made
                                                 // effective -- OpenCV 4.2 never increments its layer counter; BASELINE.json config 5 "3-level flow pyramid"); 0 = all levels
    int launch_ahead = 3;                        // pyramid levels the launching thread may be ahead of the GPU (0 = unbounded)
    SolverCfg solver;                            // this handle's solver settings
    bool latency_tiles = true;                   // tiled levels of few images: 1024-thread tiles and deep halos (k_sor_tile) where every tile has a compute unit to itself
    bool level_up = true;                        // W += dW, up-sampling and the next level's warp in one launch (k_level_up); false: the three kernels (cross-check)
    const SolverCfg& opts() { solver.opts = (coarse_chain ? FLOW_OPT_COARSE_CHAIN : 0) | (latency_tiles ? FLOW_OPT_LATENCY_TILES : 0) | (level_up ? FLOW_OPT_LEVEL_UP : 0); return solver; }
    bool coarse_chain = true;                    // the one-workgroup levels (<= ~8 k pixels) run in ONE launch (k_coarse_chain); false: per-stage kernels everywhere (cross-check, A/B timing)
    ~FlowEngine() { for (hipEvent_t e : level_done) (void)hipEventDestroy(e); }
private:
    DevBuf<float> plane_store, pyr0, pyr1;
    std::vector<hipEvent_t> level_done;          // one event per pyramid level
    float* level_ptr(DevBuf<float>& pyr, int l, int B) { return pyr.p + level_off[l] * (size_t)B; }
};

}  // namespace sind
