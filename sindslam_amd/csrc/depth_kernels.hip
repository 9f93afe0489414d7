// Depth-side device stages of DynaDetect for gfx950 (reference ORB_SLAM2/src/DynaDetect.cc):
//   SegByKmeans (:315-420): depth pyramid, back-projection, label up-sampling, cv::kmeans assignment + centre sums
//   CalOccluded (:429-482): medianBlur(5), 5x5 max-difference depth edge, valid-area mask
//   morphologyEx with elliptical elements (:51-59 and every MORPH_* call)
//   PEAC initial 16x16 block statistics (PEAC/AHCPlaneSeg.hpp:180-262)
//   SegAndMergeV2 region-adjacency statistics (:784-893, cal_hist :1685-1739) from per-pixel membership words
// Integer stages are bit-exact; FP32 expressions are written in the reference's operation order (-ffp-contract=off).
#include <mutex>
#include "common.hpp"
#include "depth.hpp"
#include "host/peac_fit.hpp"

namespace sind {

// ---------------------------------------------------------------- cv::resize(INTER_LINEAR) with exact 2x -> INTER_AREA fast path, CV_16U
// Every kernel of the k-means chain takes a batch of frames: blockIdx.z (2-D grids) or blockIdx.y (1-D grids) is the frame, *_stride the
// distance between two frames' planes in elements (the pipeline runs the chain for all streams' frame t at once; one frame = batch of 1).
__global__ void k_depth_half(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int dw, int dh, size_t src_stride, size_t dst_stride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    src += blockIdx.z * src_stride; dst += blockIdx.z * dst_stride;
    if (x >= dw) return;
    const int sw = dw * 2;
    const int s = src[(2 * y) * sw + 2 * x] + src[(2 * y) * sw + 2 * x + 1] + src[(2 * y + 1) * sw + 2 * x] + src[(2 * y + 1) * sw + 2 * x + 1];
    dst[y * dw + x] = (uint16_t)((s + 2) >> 2);
}

// ---------------------------------------------------------------- back-projection (DD:347-369), SoA points
__global__ void k_points(const uint16_t* __restrict__ depth, float* __restrict__ px, float* __restrict__ py, float* __restrict__ pz,
                         int w, int h, float scale, float fx, float fy, float cx, float cy, float depthScale, float depth_weight, size_t depth_stride, size_t pt_stride) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    depth += blockIdx.z * depth_stride; px += blockIdx.z * pt_stride; py += blockIdx.z * pt_stride; pz += blockIdx.z * pt_stride;
    if (col >= w) return;
    const int i = row * w + col;
    const uint16_t d = (uint16_t)(depth[i] * scale);
    if ((float)d / depthScale >= (float)(uint16_t)6 || d == 0) { px[i] = 0.f; py[i] = 0.f; pz[i] = 0.f; return; }
    const float depth2 = (float)d * (1.0f / depthScale);
    pz[i] = (float)(depth2 * depth_weight);
    px[i] = (float)((col - cx * scale) * depth2 * (1.0f / (fx * scale)));
    py[i] = (float)((row - cy * scale) * depth2 * (1.0f / (fy * scale)));
}

// ---------------------------------------------------------------- initial labels
__global__ void k_labels_grid(int* __restrict__ labels, int w, int h, float batch_rows, float batch_cols, int ncol, size_t stride, const int* __restrict__ use_prev) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w || (use_prev && use_prev[blockIdx.z])) return;             // frames with labels from their previous frame are filled by k_labels_resize
    labels[blockIdx.z * stride + i * w + j] = (int)floorf(i / batch_rows) * ncol + (int)floorf(j / batch_cols);
}
// bilinear resize of a label image converted to float, then cvRound (DD:390-394, 402-406)
template <class T>
__global__ void k_labels_resize(const T* __restrict__ src, int* __restrict__ dst, int sw, int sh, int dw, int dh, double scale_x, double scale_y,
                                size_t src_stride, size_t dst_stride, const int* __restrict__ use_prev) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    src += blockIdx.z * src_stride; dst += blockIdx.z * dst_stride;
    if (dx >= dw || (use_prev && !use_prev[blockIdx.z])) return;          // first frame of a stream: the 3 x 4 grid labels (k_labels_grid)
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = d_cvFloorf(fx); fx -= sx;
    const bool two = sx + 1 < sw;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = d_cvFloorf(fy); fy -= sy;
    const int y0 = d_clip(sy, 0, sh), y1 = d_clip(sy + 1, 0, sh);
    const float a1 = fx, a0 = 1.f - a1, b1 = fy, b0 = 1.f - b1;
    const T* R0 = src + (size_t)y0 * sw; const T* R1 = src + (size_t)y1 * sw;
    float r0, r1;
    if (two) { r0 = (float)R0[sx] * a0 + (float)R0[sx + 1] * a1; r1 = (float)R1[sx] * a0 + (float)R1[sx + 1] * a1; }
    else     { r0 = (float)R0[sx] * 1.f;                         r1 = (float)R1[sx] * 1.f; }
    dst[dy * dw + dx] = d_cvRound(r0 * b0 + r1 * b1);
}

// ---------------------------------------------------------------- k-means (cv::kmeans, KMEANS_USE_INITIAL_LABELS), device resident
// Centre sums are EXACTLY cv::kmeans': one FP32 accumulator per centre coordinate, samples added in index order (kmeans.cpp "compute centers").
// FP32 addition is not associative and DynaDetect amplifies a centre that is off by one ulp (a handful of border pixels change cluster,
// SegAndMerge merges differently, the next frame's warm labels start another local optimum: measured IoU 0.38 against the oracle twenty
// frames later with the former FP64 tree sums), so the order is reproduced instead of approximated:
//   k_km_count / k_km_assign_count   per wave-segment (a contiguous index range) counts of every cluster      -- parallel
//   k_km_compact                     ordered scatter of the coordinates into per-cluster runs (stable partition) -- parallel
//   k_km_seqsum                      one workgroup per run (12 clusters x 3 coordinates): the sequential FP32 sum of the run, exactly, by windows of
//                                    512 samples in integer arithmetic inside one binade; the LAST workgroup of a level to finish then
//                                    takes the centre step with cv::kmeans' operations (km_try_finalize: empty-cluster repair, scale, shift test,
//                                    last-iteration decision).  One KmState per pyramid level and frame.
__device__ void km_try_finalize(KmState* st) {
    // look for an empty cluster; if there is one, request a farthest-point search and return
    for (int k = 0; k < KM_K; k++) {
        if (st->cnt[k] != 0) continue;
        int max_k = 0; for (int k1 = 1; k1 < KM_K; k1++) if (st->cnt[max_k] < st->cnt[k1]) max_k = k1;
        const float sc = 1.f / st->cnt[max_k];
        for (int j = 0; j < 3; j++) st->base[j] = st->ctr[max_k][j] * sc;
        st->fix_k = k; st->max_k = max_k; st->far = 0ull;
        return;
    }
    st->fix_k = -1;
    double max_center_shift = st->iter == 0 ? 1.7976931348623157e308 : 0.0;
    for (int k = 0; k < KM_K; k++) {
        const float sc = 1.f / st->cnt[k];
        for (int j = 0; j < 3; j++) st->ctr[k][j] *= sc;
        if (st->iter > 0) { double dist = 0; for (int j = 0; j < 3; j++) { const double t = st->ctr[k][j] - st->old[k][j]; dist += t * t; } max_center_shift = max_center_shift > dist ? max_center_shift : dist; }
    }
    st->iter++;
    st->phase = 1;                                        // centres of this iteration are final
    if (st->iter == (st->maxCount > 2 ? st->maxCount : 2) || max_center_shift <= st->eps2) st->done = 1;
}
// Segment geometry shared by the count / compact kernels: the points are cut into nseg contiguous segments of seg_len (a multiple of 64),
// one per wave; segment s = blockIdx.x * 4 + wave.
#define KM_WAVES 4
__device__ __forceinline__ void km_count_store(const int (&c)[KM_K], int* __restrict__ segcnt, int seg, int lane) {
    if (lane < KM_K) { int v = 0;
        #pragma unroll
        for (int k = 0; k < KM_K; k++) if (lane == k) v = c[k];
        segcnt[seg * KM_K + lane] = v; }
}
// first centre pass of a level: counts of the given labels
// per-frame strides of the k-means buffers (elements): points / labels planes, count table, compacted runs, state records
struct KmStride { size_t pt, lab, seg, comp, st; };
// first centre pass of a level: counts of the given labels; its first thread also resets the level's state record and the pass ticket (no kernel of the
// level has run yet, so nobody reads them concurrently)
__global__ void __launch_bounds__(64 * KM_WAVES) k_km_count(const int* __restrict__ labels, int n, int seg_len, int* __restrict__ segcnt, KmState* __restrict__ st, KmStride ks,
                                                            int maxCount, double eps2, int ticket_word) {
    labels += blockIdx.y * ks.lab; segcnt += blockIdx.y * ks.seg; st += blockIdx.y * ks.st;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->iter = 0; st->done = 0; st->phase = 0; st->overflow = 0; st->fix_k = -1; st->max_k = 0; st->far = 0ull; st->maxCount = maxCount; st->eps2 = eps2;
        for (int k = 0; k < KM_K; k++) { st->cnt[k] = 0; for (int j = 0; j < 3; j++) { st->ctr[k][j] = 0.f; st->old[k][j] = 0.f; } }
        segcnt[ticket_word] = 0;
    }
    const int lane = threadIdx.x & 63, seg = blockIdx.x * KM_WAVES + (threadIdx.x >> 6);
    const int lo = seg * seg_len, hi = min(n, lo + seg_len);
    int c[KM_K];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) c[k] = 0;
    for (int i = lo + lane; i - lane < hi; i += 64) {                     // wave-uniform trip count (the ballots below need every lane)
        const int l = i < hi ? labels[i] : -1;
        #pragma unroll
        for (int k = 0; k < KM_K; k++) c[k] += __popcll(__ballot(l == k));
    }
    km_count_store(c, segcnt, seg, lane);
}
// re-assignment to the nearest centre (cv::kmeans' KMeansDistanceComputer: float accumulation, first minimum wins) + the counts of the new labels
__global__ void __launch_bounds__(64 * KM_WAVES) k_km_assign_count(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                                   int* __restrict__ labels, int n, int seg_len, int* __restrict__ segcnt, const KmState* __restrict__ st, KmStride ks) {
    px += blockIdx.y * ks.pt; py += blockIdx.y * ks.pt; pz += blockIdx.y * ks.pt; labels += blockIdx.y * ks.lab; segcnt += blockIdx.y * ks.seg; st += blockIdx.y * ks.st;
    if (st->done || st->phase != 1) return;
    __shared__ float ctr[KM_K][3];
    const int tid = threadIdx.x, lane = tid & 63, seg = blockIdx.x * KM_WAVES + (tid >> 6);
    if (tid < KM_K * 3) ctr[tid / 3][tid % 3] = st->ctr[tid / 3][tid % 3];
    __syncthreads();
    const int lo = seg * seg_len, hi = min(n, lo + seg_len);
    int c[KM_K];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) c[k] = 0;
    for (int i = lo + lane; i - lane < hi; i += 64) {
        int best = -1;
        if (i < hi) {
            const float xf = px[i], yf = py[i], zf = pz[i];
            float md = 3.402823466e+38f; best = 0;
            #pragma unroll
            for (int k = 0; k < KM_K; k++) {
                float t = xf - ctr[k][0]; float dist = 0.f; dist += t * t;
                t = yf - ctr[k][1]; dist += t * t;
                t = zf - ctr[k][2]; dist += t * t;
                if (md > dist) { md = dist; best = k; }
            }
            labels[i] = best;
        }
        #pragma unroll
        for (int k = 0; k < KM_K; k++) c[k] += __popcll(__ballot(best == k));
    }
    km_count_store(c, segcnt, seg, lane);
}
// Stable partition by label: cluster k's samples, in index order, become the run comp[j][start_k .. start_k + n_k) of every coordinate
// plane j (start_k = samples of the clusters before k).  A wave derives its write positions from the count table alone.
__global__ void __launch_bounds__(64 * KM_WAVES) k_km_compact(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                              const int* __restrict__ labels, int n, int seg_len, int nseg, const int* __restrict__ segcnt,
                                                              float* __restrict__ comp, int* __restrict__ tot_out, const KmState* __restrict__ st, KmStride ks) {
    px += blockIdx.y * ks.pt; py += blockIdx.y * ks.pt; pz += blockIdx.y * ks.pt; labels += blockIdx.y * ks.lab; segcnt += blockIdx.y * ks.seg; comp += blockIdx.y * ks.comp;
    tot_out += blockIdx.y * ks.seg; st += blockIdx.y * ks.st;
    if (st->done) return;
    const int lane = threadIdx.x & 63, seg = blockIdx.x * KM_WAVES + (threadIdx.x >> 6);
    // per cluster: samples in the segments before this one (before) and in all segments (total); lanes stride over the table rows
    int before[KM_K], total[KM_K];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) { before[k] = 0; total[k] = 0; }
    for (int q = lane; q < nseg; q += 64) {
        #pragma unroll
        for (int k = 0; k < KM_K; k++) { const int v = segcnt[q * KM_K + k]; total[k] += v; if (q < seg) before[k] += v; }
    }
    #pragma unroll
    for (int k = 0; k < KM_K; k++) for (int o = 32; o > 0; o >>= 1) { before[k] += __shfl_xor(before[k], o); total[k] += __shfl_xor(total[k], o); }
    int pos[KM_K]; int run = 0;
    #pragma unroll
    for (int k = 0; k < KM_K; k++) { pos[k] = run + before[k]; run += total[k]; }
    if (seg == 0 && lane < KM_K) { int v = 0;              // cluster sizes for k_km_seqsum / k_km_update, behind the table's last row
        #pragma unroll
        for (int k = 0; k < KM_K; k++) if (lane == k) v = total[k];
        tot_out[lane] = v; }
    float* cx = comp; float* cy = comp + n; float* cz = comp + 2 * (size_t)n;
    const int lo = seg * seg_len, hi = min(n, lo + seg_len);
    // four turns of 64 samples are requested together (a turn's loads are one memory round trip; a wave spent its life waiting for ~13 of them in a row)
    for (int i0 = lo; i0 < hi; i0 += 4 * 64) {
        int l4[4]; float x4[4], y4[4], z4[4];
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * 64 + lane; const bool in = i < hi;
            l4[u] = in ? labels[i] : -1; x4[u] = in ? px[i] : 0.f; y4[u] = in ? py[i] : 0.f; z4[u] = in ? pz[i] : 0.f;
        }
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + u * 64 >= hi) break;                      // wave-uniform
            const int l = l4[u];
            #pragma unroll
            for (int k = 0; k < KM_K; k++) {
                const unsigned long long m = __ballot(l == k);
                if (l == k) { const int d = pos[k] + __popcll(m & ((1ull << lane) - 1ull)); cx[d] = x4[u]; cy[d] = y4[u]; cz[d] = z4[u]; }
                pos[k] += __popcll(m);
            }
        }
    }
}
// ---- the sequential FP32 sums  acc = 0; for (i) acc = acc + x[i]  (round to nearest even), bit for bit, by one wave per run.
// A dependent FP32 add costs ~8 cycles, a run has up to ~10^5 samples and a frame needs 16 passes: adding one sample at a time was measured
// at 0.3-0.5 ms per pass.  The additions are therefore done a WINDOW (256 samples) at a time in exact integer arithmetic:
//   while the accumulator stays inside one binade [2^e, 2^(e+1)) it is an integer S in [2^23, 2^24) times u = 2^(e-23), and
//   RN(S u + x) = (S + rint(x / u)) u  unless  x / u is an exact tie (the even-mantissa rule then depends on S) or the result leaves the binade.
// A window step scales its samples by 1 / u (exact), rounds them, prefix-sums the integers over the wave and looks for the FIRST sample that
// breaks a premise (tie, |x / u| >= 2^22, result outside (2^23, 2^24), or exactly 2^23 reached from above, where the spacing halves).
// Everything before that sample is the sequential result by induction; the sample itself is added by the hardware FP32 add, which re-bases
// the binade, and the next window starts behind it.  Premises fail ~20-30 times per run (once per binade the sum grows through), so a run
// costs ~n / 256 steps of ~0.2 us instead of n dependent adds.  acc == 0 (leading zero coordinates of the invalid points) is handled by
// skipping to the first non-zero sample.  Samples are staged through an LDS ring by the other three waves of the workgroup (a window start is
// arbitrary after a re-base, and one wave cannot hide the memory latency of its own stream).
#define KM_EPL 8
#define KM_WIN (64 * KM_EPL)      /* 512 samples per window step */
#define KM_CH 2048                /* samples per LDS chunk (a multiple of KM_WIN, at least 2 * KM_WIN + KM_SER_MAX) */
#define KM_RING (4 * KM_CH)       /* four chunk slots: two being read (a window or a serial stretch may straddle), one being written, one spare */
#define KM_TICKET 40               /* word behind the 36 sums that counts the finished runs of a pass */
#define KM_SER_MAX 512            /* longest stretch of plain one-by-one adds after a broken premise */
// ring position of sample i: one pad word per eight samples, so that lane l's samples pos + 8 l + q (q fixed) sit 9 words apart -- an odd stride,
// 32 consecutive lanes hit 32 different banks (the plain layout puts them 8 apart: a 16-way conflict on every window read)
#define KM_RING_WORDS (KM_RING + KM_RING / 8)
__device__ __forceinline__ int km_at(int i) { const int j = i & (KM_RING - 1); return j + (j >> 3); }
// The step state (accumulator, position, stretch length) is wave-uniform: pinning it to scalar registers turns the step's control flow into
// scalar branches.  Lane reads with a uniform index are v_readlane (a generic __shfl is a ds_bpermute round trip of ~100 cycles, and a step is
// one long dependency chain).
__device__ __forceinline__ int km_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float km_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ int km_rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float km_rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
// sample / integer `slot` of lane `lane` (both uniform): eight lane reads and a scalar select -- never a dynamically indexed register array,
// which the compiler would spill to scratch memory
__device__ __forceinline__ float km_pick(const float (&v)[KM_EPL], int lane, int slot) { float o = km_rlf(v[0], lane);
    #pragma unroll
    for (int i = 1; i < KM_EPL; i++) { const float t = km_rlf(v[i], lane); o = slot == i ? t : o; }
    return o; }
__device__ __forceinline__ int km_picki(const int (&v)[KM_EPL], int lane, int slot) { int o = km_rl(v[0], lane);
    #pragma unroll
    for (int i = 1; i < KM_EPL; i++) { const int t = km_rl(v[i], lane); o = slot == i ? t : o; }
    return o; }
// inclusive prefix sum over the 64 lanes: DPP row shifts inside the four rows of 16, then the row totals through readlane
__device__ __forceinline__ int km_wave_scan(int v, int lane) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source receive 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    const int r0 = km_rl(v, 15), r1 = km_rl(v, 31), r2 = km_rl(v, 47);
    return v + (lane >= 48 ? r0 + r1 + r2 : lane >= 32 ? r0 + r1 : lane >= 16 ? r0 : 0);
}
// lane l's KM_EPL consecutive samples of the window starting at pos (the ring is zero behind the run's end)
__device__ __forceinline__ void km_window(const float* __restrict__ ring, int lane, int pos, float (&x)[KM_EPL]) {
    const int g0 = pos + KM_EPL * lane;
    #pragma unroll
    for (int q = 0; q < KM_EPL; q++) x[q] = ring[km_at(g0 + q)];
}
// One window step of the wave on the samples x = [pos, pos + KM_WIN); acc, pos are uniform.  Returns the next position; `broke` tells
// whether a premise failed (the failing sample has then been added by the hardware add).  The arithmetic runs on magnitudes: the samples are
// scaled by sign(acc) / u, so the running integer stays positive and every premise is a plain range test.
__device__ __forceinline__ int km_seq_step(const float (&x)[KM_EPL], int lane, float& acc, int pos, bool& broke) {
    broke = false;
    const unsigned ab = __float_as_uint(acc);
    if ((ab << 1) == 0u) {                                 // acc == 0: zeros leave it there, the first non-zero sample becomes the accumulator
        int loc = KM_EPL;
        #pragma unroll
        for (int q = KM_EPL - 1; q >= 0; q--) loc = x[q] != 0.f ? q : loc;
        const unsigned long long m = __ballot(loc < KM_EPL);
        if (!m) return pos + KM_WIN;
        const int fl = __ffsll((long long)m) - 1, floc = km_rl(loc, fl);
        acc = km_unif(0.f + km_pick(x, fl, floc));
        return pos + KM_EPL * fl + floc + 1;
    }
    const int ef = (int)((ab >> 23) & 0xffu);
    if (ef < 127 - 80 || ef > 127 + 100) { acc = km_unif(acc + km_rlf(x[0], 0)); broke = true; return pos + 1; }       // far-out exponents (and denormal / inf / nan): plain add
    const unsigned sb = ab & 0x80000000u;                  // sign of the accumulator, folded into the scale factors
    const float scale = __uint_as_float(((unsigned)(127 + 23 + 127 - ef) << 23) | sb), unit = __uint_as_float(((unsigned)(ef - 23) << 23) | sb);
    const int S_in = (int)(acc * scale);                   // exact: |acc| / u, an integer in [2^23, 2^24)
    int r[KM_EPL]; bool bad[KM_EPL], down[KM_EPL];
    #pragma unroll
    for (int q = 0; q < KM_EPL; q++) {
        const float xs = x[q] * scale, rf = rintf(xs);     // power-of-two scaling: exact (or a harmless underflow towards 0)
        bad[q] = !(fabsf(xs) < 4194304.f) || fabsf(xs - rf) == 0.5f;
        r[q] = bad[q] ? 0 : (int)rf; down[q] = xs < 0.f;
    }
    int c[KM_EPL]; c[0] = r[0];
    #pragma unroll
    for (int q = 1; q < KM_EPL; q++) c[q] = c[q - 1] + r[q];
    const int base = S_in + km_wave_scan(c[KM_EPL - 1], lane) - c[KM_EPL - 1];
    int Sq[KM_EPL]; int loc = KM_EPL;
    #pragma unroll
    for (int q = KM_EPL - 1; q >= 0; q--) {
        Sq[q] = base + c[q];
        // valid results lie in (2^23, 2^24); exactly 2^23 only when reached without moving down (below it the spacing halves)
        const bool fail = bad[q] || (unsigned)(Sq[q] - 8388608) >= 8388608u || (Sq[q] == 8388608 && down[q]);
        loc = fail ? q : loc;
    }
    const unsigned long long fm = __ballot(loc < KM_EPL);
    if (!fm) { acc = km_unif((float)km_rl(Sq[KM_EPL - 1], 63) * unit); return pos + KM_WIN; }      // clean window
    const int fl = __ffsll((long long)fm) - 1, f = km_uni(KM_EPL * fl + km_rl(loc, fl));
    int Sprev = S_in;
    if (f > 0) Sprev = km_picki(Sq, (f - 1) / KM_EPL, (f - 1) & (KM_EPL - 1));
    const float before = (float)Sprev * unit;              // exact
    acc = km_unif(before + km_pick(x, f / KM_EPL, f & (KM_EPL - 1)));
    broke = true;
    return pos + f + 1;
}
// plain sequential adds of the samples [pos, pos + cnt), cnt a multiple of 64: the mode for stretches in which the sum keeps changing binade
// (e.g. a cluster astride the optical axis: every image row drags its x-sum down and up through zero).  One conflict-free LDS read hands 64
// samples to the lanes; they reach the accumulator through v_readlane with constant lane numbers, so a sample costs its dependent add and
// one scalar-register read.
__device__ __forceinline__ void km_seq_serial(const float* __restrict__ ring, int lane, float& acc, int pos, int cnt) {
    float a = acc;
    float v = ring[km_at(pos + lane)];
    for (int i = 0; i < cnt; i += 64) {
        const float vn = ring[km_at(pos + i + 64 + lane)];      // the next 64 are on their way while these are added
        #pragma unroll
        for (int q = 0; q < 64; q++) a = a + km_rlf(v, q);
        v = vn;
    }
    acc = km_unif(a);
}
// chunk ch of the run (zero-filled behind its end): global -> registers, registers -> ring slot
#define KM_NR ((KM_CH / 4 + 191) / 192)          /* float4 pieces per thread of the three loader waves (the four-wave prologue needs fewer) */
__device__ __forceinline__ void km_chunk_fetch(const float* __restrict__ p, int nk, int ch, int tid, int nthreads, float4 (&R)[KM_NR]) {
    struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
    #pragma unroll
    for (int u = 0; u < KM_NR; u++) {
        const int v4 = tid + u * nthreads; float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v4 < KM_CH / 4) { const int g = ch * KM_CH + 4 * v4;
            if (g + 4 <= nk) { const F4u a = *reinterpret_cast<const F4u*>(p + g); v = make_float4(a.x, a.y, a.z, a.w); }
            else { if (g < nk) v.x = p[g]; if (g + 1 < nk) v.y = p[g + 1]; if (g + 2 < nk) v.z = p[g + 2]; } }
        R[u] = v;
    }
}
__device__ __forceinline__ void km_chunk_store(float* __restrict__ ring, int ch, int tid, int nthreads, const float4 (&R)[KM_NR]) {
    #pragma unroll
    for (int u = 0; u < KM_NR; u++) {
        const int v4 = tid + u * nthreads;
        if (v4 < KM_CH / 4) { const int o = km_at(ch * KM_CH + 4 * v4);      // four samples of one group of eight: contiguous words
            ring[o] = R[u].x; ring[o + 1] = R[u].y; ring[o + 2] = R[u].z; ring[o + 3] = R[u].w; }
    }
}
// Centre step of one k-means iteration in ONE workgroup: take the 36 sequential sums, then repair every empty cluster (block-wide farthest-point search over the biggest cluster, as cv::kmeans
// does, repeated until no cluster is empty), scale, shift test, stop decision.  No host round trip and no provisioning limit.
// Runs in the workgroup of k_km_seqsum that finishes LAST for its frame (all pointers already belong to that frame), any block size.
__device__ void km_centre_step(const int* __restrict__ segcnt, int nseg, const float* __restrict__ seqsums, KmState* __restrict__ gst,
                               const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz, int* __restrict__ labels, int n) {
    __shared__ float sums[KM_K * 3];
    __shared__ int tot[KM_K + 1];
    __shared__ unsigned long long wbest[16];
    __shared__ int s_fix;
    __shared__ KmState S;                                  // the state lives in LDS while the centre step runs: the serial part below touches it
                                                           // ~100 times, and every touch of the global copy would be a dependent ~1 us round trip
    static_assert(sizeof(KmState) % 4 == 0, "KmState is copied word by word");
    const int t = threadIdx.x;
    for (int i = t; i < (int)(sizeof(KmState) / 4); i += blockDim.x) reinterpret_cast<unsigned*>(&S)[i] = reinterpret_cast<const unsigned*>(gst)[i];
    if (t < KM_K) tot[t] = segcnt[nseg * KM_K + t];
    __syncthreads();
    KmState* st = &S;
    if (t < KM_K * 3) sums[t] = seqsums[t];
    __syncthreads();
    if (t == 0) {
        st->phase = 0;
        for (int k = 0; k < KM_K; k++) { for (int j = 0; j < 3; j++) { st->old[k][j] = st->ctr[k][j]; st->ctr[k][j] = sums[k * 3 + j]; } st->cnt[k] = tot[k]; }
        km_try_finalize(st);
        s_fix = st->fix_k;
    }
    __syncthreads();
    while (s_fix >= 0) {                                   // uniform: s_fix is shared
        const int which = st->max_k; const float c0 = st->base[0], c1 = st->base[1], c2 = st->base[2];
        unsigned long long b = 0;
        // four points per lane and step with 16-byte loads: the scan is one workgroup walking the whole level, so its cost is the number
        // of dependent memory round trips (it was ~280 us at 640x480 with scalar loads, and empty clusters are not rare)
        auto consider = [&](int l, float x, float y, float z, int i) {
            if (l != which) return;
            float d0 = x - c0; float d = 0.f; d += d0 * d0; d0 = y - c1; d += d0 * d0; d0 = z - c2; d += d0 * d0;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;      // ties -> largest index ("max_dist <= dist")
            b = b > key ? b : key;
        };
        const int n4 = n >> 2;
        #pragma unroll 2
        for (int q = t; q < n4; q += blockDim.x) {
            const int4 L = reinterpret_cast<const int4*>(labels)[q];
            const float4 X = reinterpret_cast<const float4*>(px)[q], Y = reinterpret_cast<const float4*>(py)[q], Z = reinterpret_cast<const float4*>(pz)[q];
            consider(L.x, X.x, Y.x, Z.x, 4 * q); consider(L.y, X.y, Y.y, Z.y, 4 * q + 1); consider(L.z, X.z, Y.z, Z.z, 4 * q + 2); consider(L.w, X.w, Y.w, Z.w, 4 * q + 3);
        }
        for (int i = (n4 << 2) + t; i < n; i += blockDim.x) consider(labels[i], px[i], py[i], pz[i], i);
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(b, o); b = b > v ? b : v; }
        if ((t & 63) == 0) wbest[t >> 6] = b;
        __syncthreads();
        if (t == 0) {
            unsigned long long m = 0; for (int q = 0; q < (int)(blockDim.x >> 6); q++) m = m > wbest[q] ? m : wbest[q];
            const int fi = (int)(m & 0xffffffffull), k = st->fix_k, max_k = st->max_k;
            const float smp[3] = {px[fi], py[fi], pz[fi]};
            labels[fi] = k;
            st->cnt[max_k]--; st->cnt[k]++;
            for (int j = 0; j < 3; j++) { st->ctr[max_k][j] -= smp[j]; st->ctr[k][j] += smp[j]; }
            km_try_finalize(st);
            s_fix = st->fix_k;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int i = t; i < (int)(sizeof(KmState) / 4); i += blockDim.x) reinterpret_cast<unsigned*>(gst)[i] = reinterpret_cast<const unsigned*>(&S)[i];
}
// grid = 36 runs (cluster k = blockIdx.x / 3, coordinate j = blockIdx.x % 3), 256 threads: wave 0 sums, waves 1..3 keep the ring ahead of it
// (chunk ch + 2 goes into the ring while chunk ch is summed, chunk ch + 3 is already on its way in registers)
// The workgroup that finishes last for its frame (ticket counter behind the sums, agent-scope fences around it) runs the centre step right away:
// one launch per pass less, and no second kernel whose only job is to wait for this one.
__global__ void __launch_bounds__(256) k_km_seqsum(const int* __restrict__ segcnt, int nseg, const float* __restrict__ comp, int n, float* __restrict__ seqsums,
                                                   KmState* __restrict__ st, const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                   int* __restrict__ labels, KmStride ks) {
    segcnt += blockIdx.y * ks.seg; comp += blockIdx.y * ks.comp; seqsums += blockIdx.y * ks.seg; st += blockIdx.y * ks.st;
    if (st->done) return;
    __shared__ float ring[KM_RING_WORDS];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, k = blockIdx.x / 3, j = blockIdx.x % 3;
    const int* tot = segcnt + nseg * KM_K;                 // cluster sizes, left behind the count table by k_km_compact
    int start = 0; for (int q = 0; q < k; q++) start += tot[q];
    const int nk = tot[k]; const float* p = comp + (size_t)j * n + start;
    const int nch = (nk + KM_CH - 1) / KM_CH;              // chunks 0 .. nch are staged: the one behind the run's end is all zero (windows and stretches read past the end)
    float4 R[KM_NR];
    { float4 R1[KM_NR];                                    // chunks 0 and 1 by all four waves, both fetches in flight together
      km_chunk_fetch(p, nk, 0, t, 256, R); km_chunk_fetch(p, nk, 1, t, 256, R1);
      km_chunk_store(ring, 0, t, 256, R); km_chunk_store(ring, 1, t, 256, R1); }
    if (wv > 0) km_chunk_fetch(p, nk, 2, t - 64, 192, R);
    __syncthreads();
    float acc = 0.f; int pos = 0, ser = 0;
    for (int ch = 0; ch < nch; ch++) {
        if (wv == 0) {
            const int lim = min(nk, (ch + 1) * KM_CH);
            float x[KM_EPL]; km_window(ring, lane, pos, x);
            while (pos < lim) {
                float xn[KM_EPL]; km_window(ring, lane, pos + KM_WIN, xn);            // the next window, read while this one is worked on (stays inside chunk ch + 1)
                bool broke; const int from = pos;
                pos = km_uni(km_seq_step(x, lane, acc, pos, broke));
                if (broke) {                               // plain adds for a while; the stretch doubles as long as windows keep breaking
                    ser = km_uni(ser ? min(2 * ser, KM_SER_MAX) : 64);
                    km_seq_serial(ring, lane, acc, pos, ser); pos += ser;
                    km_window(ring, lane, pos, x);
                } else if (pos == from + KM_WIN) {
                    ser = 0;
                    #pragma unroll
                    for (int q = 0; q < KM_EPL; q++) x[q] = xn[q];
                } else km_window(ring, lane, pos, x);          // zero accumulator: the window ended at the first non-zero sample
            }
        } else { if (ch + 2 <= nch) km_chunk_store(ring, ch + 2, t - 64, 192, R); if (ch + 3 <= nch) km_chunk_fetch(p, nk, ch + 3, t - 64, 192, R); }
        __syncthreads();
    }
    __shared__ int s_last;
    if (t == 0) {
        seqsums[blockIdx.x] = acc;
        __threadfence();                                   // release: the sum is visible device-wide before the ticket is taken
        int* ticket = reinterpret_cast<int*>(seqsums) + KM_TICKET;
        const int got = atomicAdd(ticket, 1);
        s_last = got == (int)gridDim.x - 1;
        if (s_last) *ticket = 0;                           // next pass starts from zero again (nobody else touches it any more)
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                       // acquire: the other 35 sums (written by other CUs) are read from memory, not from a stale L1 line
    if (px) km_centre_step(segcnt, nseg, seqsums, st, px + blockIdx.y * ks.pt, py + blockIdx.y * ks.pt, pz + blockIdx.y * ks.pt, labels + blockIdx.y * ks.lab, n);
}
// ---- coarse pyramid levels: cv::kmeans of one level in ONE launch, one workgroup (16 waves) per frame.
// The chain of a frame's k-means is ~60 dependent launches (count, then <= 4 x (assign + count, compact, sums + centre step) on each of four levels); on the three coarse
// levels (60 x 80 ... 240 x 320) a launch has a few microseconds of work, so the chain's time there is launch latency -- and next to the flow solver, whose workgroups
// hold every CU for the length of a launch, each of those small kernels also waits for a slot.  Here the passes of a level are phases of one workgroup separated by
// barriers: the same assignment arithmetic, the same stable partition, the same exact sequential sums (km_seq_step on windows read from memory instead of the LDS ring:
// each wave sums whole runs by itself, 36 runs over 16 waves) and the same centre step (km_centre_step), so the labels and centres are bit for bit those of the
// per-pass kernels (tests/test_dyna_gpu.py, test_kmeans_fused_gpu.py).
#define KMF_WAVES 16
__device__ __forceinline__ void km_window_g(const float* __restrict__ p, int nk, int lane, int pos, float (&x)[KM_EPL]) {
    struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
    const int g0 = pos + KM_EPL * lane;
    if (g0 + KM_EPL <= nk) { const F4u a = *reinterpret_cast<const F4u*>(p + g0), b = *reinterpret_cast<const F4u*>(p + g0 + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w; }
    else {
        #pragma unroll
        for (int q = 0; q < KM_EPL; q++) x[q] = (g0 + q < nk) ? p[g0 + q] : 0.f;       // behind the run's end: zeros, like the ring
    }
}
__device__ __forceinline__ void km_seq_serial_g(const float* __restrict__ p, int nk, int lane, float& acc, int pos, int cnt) {
    float a = acc;
    float v = (pos + lane < nk) ? p[pos + lane] : 0.f;
    for (int i = 0; i < cnt; i += 64) {
        const int gn = pos + i + 64 + lane; const float vn = (gn < nk) ? p[gn] : 0.f;
        #pragma unroll
        for (int q = 0; q < 64; q++) a = a + km_rlf(v, q);
        v = vn;
    }
    acc = km_unif(a);
}
// the sequential FP32 sum of p[0 .. nk) by ONE wave: the loop of k_km_seqsum's summing wave, windows and stretches read from memory
__device__ float km_run_sum_wave(const float* __restrict__ p, int nk, int lane) {
    static_assert(KM_EPL == 8, "km_window_g loads two 16-byte pieces per lane");
    float acc = 0.f; int pos = 0, ser = 0;
    if (nk <= 0) return acc;
    float x[KM_EPL]; km_window_g(p, nk, lane, pos, x);
    while (pos < nk) {
        float xn[KM_EPL]; km_window_g(p, nk, lane, pos + KM_WIN, xn);
        bool broke; const int from = pos;
        pos = km_uni(km_seq_step(x, lane, acc, pos, broke));
        if (broke) {
            ser = km_uni(ser ? min(2 * ser, KM_SER_MAX) : 64);
            km_seq_serial_g(p, nk, lane, acc, pos, ser); pos += ser;
            km_window_g(p, nk, lane, pos, x);
        } else if (pos == from + KM_WIN) {
            ser = 0;
            #pragma unroll
            for (int q = 0; q < KM_EPL; q++) x[q] = xn[q];
        } else km_window_g(p, nk, lane, pos, x);
    }
    return acc;
}
__global__ void __launch_bounds__(64 * KMF_WAVES) k_km_level_fused(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz, int* __restrict__ labels, int n, int seg_len,
                                                                   int* __restrict__ segcnt, float* __restrict__ comp, KmState* __restrict__ st, KmStride ks, int maxCount, double eps2, int iters) {
    px += blockIdx.y * ks.pt; py += blockIdx.y * ks.pt; pz += blockIdx.y * ks.pt; labels += blockIdx.y * ks.lab; segcnt += blockIdx.y * ks.seg; comp += blockIdx.y * ks.comp; st += blockIdx.y * ks.st;
    float* seqsums = reinterpret_cast<float*>(segcnt + (KM_MAX_BLOCKS * 4 + 1) * KM_K);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nseg = KMF_WAVES;
    const int lo = wv * seg_len, hi = min(n, lo + seg_len);
    __shared__ float ctr[KM_K][3];
    __shared__ int s_go[2];                                // {the level goes on (not done), the assignment pass applies (phase == 1)}
    if (tid == 0) {
        st->iter = 0; st->done = 0; st->phase = 0; st->overflow = 0; st->fix_k = -1; st->max_k = 0; st->far = 0ull; st->maxCount = maxCount; st->eps2 = eps2;
        for (int k = 0; k < KM_K; k++) { st->cnt[k] = 0; for (int j = 0; j < 3; j++) { st->ctr[k][j] = 0.f; st->old[k][j] = 0.f; } }
    }
    {   // counts of the given labels (k_km_count)
        int c[KM_K];
        #pragma unroll
        for (int k = 0; k < KM_K; k++) c[k] = 0;
        for (int i = lo + lane; i - lane < hi; i += 64) {
            const int l = i < hi ? labels[i] : -1;
            #pragma unroll
            for (int k = 0; k < KM_K; k++) c[k] += __popcll(__ballot(l == k));
        }
        km_count_store(c, segcnt, wv, lane);
    }
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (tid == 0) { s_go[0] = !st->done; s_go[1] = !st->done && st->phase == 1; }
        if (tid < KM_K * 3) ctr[tid / 3][tid % 3] = st->ctr[tid / 3][tid % 3];
        __syncthreads();
        if (!s_go[0]) break;                               // uniform
        if (it > 0 && s_go[1]) {                           // re-assignment to the nearest centre + the counts of the new labels (k_km_assign_count)
            int c[KM_K];
            #pragma unroll
            for (int k = 0; k < KM_K; k++) c[k] = 0;
            for (int i = lo + lane; i - lane < hi; i += 64) {
                int best = -1;
                if (i < hi) {
                    const float xf = px[i], yf = py[i], zf = pz[i];
                    float md = 3.402823466e+38f; best = 0;
                    #pragma unroll
                    for (int k = 0; k < KM_K; k++) {
                        float t = xf - ctr[k][0]; float dist = 0.f; dist += t * t;
                        t = yf - ctr[k][1]; dist += t * t;
                        t = zf - ctr[k][2]; dist += t * t;
                        if (md > dist) { md = dist; best = k; }
                    }
                    labels[i] = best;
                }
                #pragma unroll
                for (int k = 0; k < KM_K; k++) c[k] += __popcll(__ballot(best == k));
            }
            km_count_store(c, segcnt, wv, lane);
        }
        __syncthreads();
        {   // stable partition by label into per-cluster runs (k_km_compact)
            int before[KM_K], total[KM_K];
            #pragma unroll
            for (int k = 0; k < KM_K; k++) { before[k] = 0; total[k] = 0; }
            if (lane < nseg) {
                #pragma unroll
                for (int k = 0; k < KM_K; k++) { const int v = segcnt[lane * KM_K + k]; total[k] = v; if (lane < wv) before[k] = v; }
            }
            #pragma unroll
            for (int k = 0; k < KM_K; k++) for (int o = 32; o > 0; o >>= 1) { before[k] += __shfl_xor(before[k], o); total[k] += __shfl_xor(total[k], o); }
            int pos[KM_K]; int run = 0;
            #pragma unroll
            for (int k = 0; k < KM_K; k++) { pos[k] = run + before[k]; run += total[k]; }
            if (wv == 0 && lane < KM_K) { int v = 0;
                #pragma unroll
                for (int k = 0; k < KM_K; k++) if (lane == k) v = total[k];
                segcnt[nseg * KM_K + lane] = v; }
            float* cx = comp; float* cy = comp + n; float* cz = comp + 2 * (size_t)n;
            for (int i = lo + lane; i - lane < hi; i += 64) {
                const bool in = i < hi; const int l = in ? labels[i] : -1;
                const float x = in ? px[i] : 0.f, y = in ? py[i] : 0.f, z = in ? pz[i] : 0.f;
                #pragma unroll
                for (int k = 0; k < KM_K; k++) {
                    const unsigned long long m = __ballot(l == k);
                    if (l == k) { const int d = pos[k] + __popcll(m & ((1ull << lane) - 1ull)); cx[d] = x; cy[d] = y; cz[d] = z; }
                    pos[k] += __popcll(m);
                }
            }
        }
        __syncthreads();
        {   // the 36 sequential sums: wave w takes the runs w, w + 16, w + 32
            const int* tot = segcnt + nseg * KM_K;
            for (int r = wv; r < KM_K * 3; r += KMF_WAVES) {
                const int k = r / 3, j = r % 3;
                int start = 0; for (int q = 0; q < k; q++) start += tot[q];
                const float acc = km_run_sum_wave(comp + (size_t)j * n + start, tot[k], lane);
                if (lane == 0) seqsums[r] = acc;
            }
        }
        __syncthreads();
        km_centre_step(segcnt, nseg, seqsums, st, px, py, pz, labels, n);
        __syncthreads();
    }
}
__global__ void k_labels_to_u8(const int* __restrict__ labels, uint8_t* __restrict__ out, int n, size_t lab_stride, size_t out_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    labels += blockIdx.y * lab_stride; out += blockIdx.y * out_stride;
    if (i < n) { const int v = labels[i]; out[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}

// ---------------------------------------------------------------- medianBlur(5) on the depth image (values are integers, so the float median == u16 median)
// (the CalOccluded kernels take the frame from blockIdx.z: the pipeline runs them once for all frames of a step)
// A thread makes MED_ROWS outputs of one column: the 5 x (MED_ROWS + 4) window is loaded once (10 loads per pixel instead of 25) and a launch has a quarter of the waves.
#define MED_ROWS 4
__global__ void k_median5_u16(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * MED_ROWS;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; dst += (size_t)blockIdx.z * w * h;
    int xs[5];
    #pragma unroll
    for (int dx = -2; dx <= 2; dx++) xs[dx + 2] = min(max(x + dx, 0), w - 1);
    int win[MED_ROWS + 4][5];
    #pragma unroll
    for (int j = 0; j < MED_ROWS + 4; j++) {
        const uint16_t* r = src + (size_t)min(max(y0 - 2 + j, 0), h - 1) * w;
        #pragma unroll
        for (int k = 0; k < 5; k++) win[j][k] = r[xs[k]];
    }
    #pragma unroll
    for (int q = 0; q < MED_ROWS; q++) {
        const int y = y0 + q;
        if (y >= h) break;
        int p[25];
        #pragma unroll
        for (int j = 0; j < 5; j++) {
            #pragma unroll
            for (int k = 0; k < 5; k++) p[j * 5 + k] = win[q + j][k];
        }
        // 99 compare-exchanges instead of the 625 comparisons of a rank count (median25_net.inc)
        #define S(a, b) { const int lo_ = min(p[a], p[b]), hi_ = max(p[a], p[b]); p[a] = lo_; p[b] = hi_; }
        #include "median25_net.inc"
        #undef S
        dst[y * w + x] = (uint16_t)p[12];
    }
}
// (16-byte loads, eight samples each: with one 2-byte load per lane and turn the kernel took 0.5 ms for the 39 MB of a 64-frame round and 3 % of a step's wave cycles)
__global__ void k_max_u16(const uint16_t* __restrict__ src, int n, unsigned* __restrict__ out, int out_stride, int wide) {
    src += (size_t)blockIdx.y * n; out += (size_t)blockIdx.y * out_stride;
    unsigned m = 0;
    const int n8 = wide ? n >> 3 : 0; const uint4* s8 = reinterpret_cast<const uint4*>(src);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
        const uint4 v = s8[i];
        m = max(m, max(max(max(v.x & 0xffffu, v.x >> 16), max(v.y & 0xffffu, v.y >> 16)), max(max(v.z & 0xffffu, v.z >> 16), max(v.w & 0xffffu, v.w >> 16))));
    }
    for (int i = (n8 << 3) + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, (unsigned)src[i]);      // n % 8 samples at the end
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// 5x5 max-difference depth edge + valid-area mask (DD:443-482); the 3-px frame is left 0 in both outputs
#define GE_ROWS 4
__global__ void k_grad_edge(const uint16_t* __restrict__ filt, const unsigned* __restrict__ dmax, uint8_t* __restrict__ edge,
                            uint8_t* __restrict__ total_area, int w, int h, float depthScale, int dmax_stride) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row0 = blockIdx.y * GE_ROWS;
    if (col >= w) return;
    { const size_t fo = (size_t)blockIdx.z * w * h; filt += fo; edge += fo; total_area += fo; dmax += (size_t)blockIdx.z * dmax_stride; }
    const bool col_in = col >= 3 && col < w - 3;
    // the 5 x (GE_ROWS + 4) window of this column's rows, loaded once (rows / columns outside the image are never used: the 3-pixel frame has no window)
    float win[GE_ROWS + 4][5];
    #pragma unroll
    for (int j = 0; j < GE_ROWS + 4; j++) {
        const int r = row0 - 2 + j; const bool in = col_in && r >= 0 && r < h;
        #pragma unroll
        for (int k = 0; k < 5; k++) win[j][k] = in ? (float)filt[r * w + col + k - 2] : 0.f;
    }
    const float depth_max = (float)(*dmax);
    #pragma unroll
    for (int q = 0; q < GE_ROWS; q++) {
        const int row = row0 + q;
        if (row >= h) break;
        uint8_t e = 0, t = 0;
        if (row >= 3 && row < h - 3 && col_in) {
            const float depth1 = win[q + 2][2];
            if (depth1 > 0.0f && depth1 / depthScale < 6.0f) t = 255;
            float val_max = 0.0f;
            #pragma unroll
            for (int i = -2; i <= 2; i++)
                #pragma unroll
                for (int j = -2; j <= 2; j++) {
                    const float nb = win[q + 2 + i][j + 2];
                    if ((depth1 - nb) > depth_max * 0.5f) continue;
                    const float a = fabsf(depth1 - nb);
                    val_max = fabsf(val_max) > a ? fabsf(val_max) : a;
                }
            if (val_max > depth1 * 0.03f && val_max > 400.0f) e = 255;
        }
        edge[row * w + col] = e; total_area[row * w + col] = t;
    }
}

// ---------------------------------------------------------------- morphology with an elliptical element (max / min over in-image pixels)
__global__ void k_morph(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, MorphElem E, int is_dilate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; dst += (size_t)blockIdx.z * w * h;
    int m = is_dilate ? 0 : 255;
    for (int i = 0; i < E.n; i++) {
        const int yy = y + i - E.ay;
        if (yy < 0 || yy >= h || E.j2[i] <= E.j1[i]) continue;
        const uint8_t* r = src + (size_t)yy * w;
        const int xa = max(x + E.j1[i] - E.ax, 0), xb = min(x + E.j2[i] - E.ax, w);
        for (int xx = xa; xx < xb; xx++) m = is_dilate ? max(m, (int)r[xx]) : min(m, (int)r[xx]);
    }
    dst[y * w + x] = (uint8_t)m;
}

// The 4 x 4 ellipse of CalOccluded's MORPH_OPEN ({(0, 0)} in the row two above, columns -2 .. +1 in the rows -1, 0, +1), a thread walking MO_ROWS rows of one column: four
// byte loads per source row serve every output row that sees it (5.5 loads per pixel instead of 13) and a launch has an eighth of the waves.  Same values as k_morph.
#define MO_ROWS 8
__global__ void k_morph_e4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, int is_dilate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * MO_ROWS;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; dst += (size_t)blockIdx.z * w * h;
    const int none = is_dilate ? 0 : 255;
    auto op = [&](int a, int b) { return is_dilate ? max(a, b) : min(a, b); };
    int own[MO_ROWS + 3], four[MO_ROWS + 3];                 // source row y0 - 2 + j: the pixel itself, the extreme of its columns x - 2 .. x + 1 (inside the image)
    #pragma unroll
    for (int j = 0; j < MO_ROWS + 3; j++) {
        const int yy = y0 - 2 + j;
        own[j] = none; four[j] = none;
        if (yy >= 0 && yy < h) {
            const uint8_t* r = src + (size_t)yy * w;
            const int c = r[x]; own[j] = c; int m = c;
            if (x >= 2) m = op(m, (int)r[x - 2]);
            if (x >= 1) m = op(m, (int)r[x - 1]);
            if (x + 1 < w) m = op(m, (int)r[x + 1]);
            four[j] = m;
        }
    }
    #pragma unroll
    for (int q = 0; q < MO_ROWS; q++) {
        const int y = y0 + q;
        if (y >= h) break;
        dst[y * w + x] = (uint8_t)op(op(own[q], four[q + 1]), op(four[q + 2], four[q + 3]));
    }
}
// ---------------------------------------------------------------- PEAC initial block statistics (16 x 16 blocks): one wave per SEVEN blocks.
// Phase 1, all lanes: the points of the wave's blocks (x, y, z as the FLOATS the reference forms before it widens them) and each block's validity (a block with a missing
// point or a depth jump to the right / lower neighbour is dropped as a whole, so the order of that test is free) go to LDS.  Phase 2: the nine FP64 moments of a block are
// accumulated by nine lanes, each adding its 256 terms in the reference's row-major order (bit-exact with a sequential CPU loop) -- 63 lanes = 7 blocks x 9 moments, and a term
// is two LDS reads, two conversions and one product.  (One block per wave with every term's point re-derived inside the sequential loop -- two float divisions per term on 9 of
// 64 lanes -- was 6.9 % of ALL VALU cycles of a step and 2.8 % of its CU-busy cycles: profiles/r04/pmc_by_kernel.txt.)
#define PEAC_BW 16
#define PEAC_BPW 7                                                  // blocks per wave
__global__ void __launch_bounds__(64) k_peac_block_stats(const uint16_t* __restrict__ depth, int w, int h, int Nw, int nblk, float fx, float fy, float cx, float cy,
                                                         float depthScale, double depthAlpha, double depthChangeTol, PeacBlockStats* __restrict__ out) {
    __shared__ float pt[PEAC_BPW][3][PEAC_BW * PEAC_BW];            // x, y, z per point of the wave's blocks
    const int lane = threadIdx.x, blk0 = blockIdx.x * PEAC_BPW;
    depth += (size_t)blockIdx.y * w * h; out += (size_t)blockIdx.y * nblk;
    const float inv = 1.0f / depthScale;
    auto zat = [&](int i, int j, float& zf) -> bool { const float d = (float)depth[i * w + j]; if (d < 1e-3f) return false; zf = d * inv; return true; };
    unsigned validbits = 0;                                         // wave-uniform
    for (int b = 0; b < PEAC_BPW; b++) {
        const int blk = blk0 + b;
        if (blk >= nblk) break;
        const int by = blk / Nw, bx = blk - by * Nw;
        bool ok = true;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = k * 64 + lane, i = by * PEAC_BW + (p >> 4), j = bx * PEAC_BW + (p & 15);
            float zf = 0.f, zn;
            if (!zat(i, j, zf)) ok = false;
            else {
                const double z = (double)zf, tol = depthAlpha * fabs(z) + depthChangeTol;
                if (j + 1 < w && zat(i, j + 1, zn) && fabs(z - (double)zn) > tol) ok = false;
                if (i + 1 < h && zat(i + 1, j, zn) && fabs(z - (double)zn) > tol) ok = false;
            }
            pt[b][0][p] = (j - cx) * zf / fx; pt[b][1][p] = (i - cy) * zf / fy; pt[b][2][p] = zf;
        }
        if (__all(ok)) validbits |= 1u << b;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0): the wave's own LDS stores (wave-synchronous, no barrier needed)
    __builtin_amdgcn_wave_barrier();
    const int b = lane / 9, m = lane - 9 * b, blk = blk0 + b;
    if (b >= PEAC_BPW || blk >= nblk) return;
    const bool valid = (validbits >> b) & 1u;
    // lane (b, m) accumulates moment m of block b: sx sy sz sxx syy szz sxy syz sxz = the sums of A, or of A x B
    const int ia = (m == 1 || m == 4 || m == 7) ? 1 : (m == 2 || m == 5) ? 2 : 0, ib = (m == 3) ? 0 : (m == 4 || m == 6) ? 1 : 2;      // (x, y, z) = (0, 1, 2): xx yy zz xy yz xz
    const bool single = m < 3;
    double acc = 0;
    if (valid) {
        const float* A = pt[b][ia]; const float* B = pt[b][ib];
        #pragma unroll 4
        for (int p = 0; p < PEAC_BW * PEAC_BW; p++) {
            const double av = (double)A[p], bv = (double)B[p];
            acc += single ? av : av * bv;
        }
    }
    double* o = &out[blk].sx; o[m] = acc;
    if (m == 0) { out[blk].N = valid ? PEAC_BW * PEAC_BW : 0; out[blk].valid = valid ? 1 : 0; }
}
// Plane fit of every valid block (host/peac_fit.hpp, the host's own function): the 1200 (640 x 480) to 3600 (1280 x 720) initial nodes are a third of all the
// fits of a frame's graph clustering (~1 us each on a host core: a 3 x 3 Jacobi iteration).  One thread per block.
__global__ void k_peac_block_fit(PeacBlockStats* __restrict__ blocks, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PeacBlockStats& S = blocks[i];
    int fitted = 0;
    if (S.valid && S.N >= 4) { double c[3], nrm[3], mse; peac_fit(&S.sx, S.N, c, nrm, mse); S.mse = mse; for (int k = 0; k < 3; k++) { S.center[k] = c[k]; S.normal[k] = nrm[k]; } fitted = 1; }
    S.fitted = fitted; S.pad = 0;
}

// ---------------------------------------------------------------- imgDepth/depth_max*255 -> 8U (DD:765-768): u16 * (float)((1/max)*255), cvRound, saturate
__global__ void k_depth_norm(const uint16_t* __restrict__ depth, const unsigned* __restrict__ dmax, uint8_t* __restrict__ out, int n, int dmax_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    depth += (size_t)blockIdx.y * n; out += (size_t)blockIdx.y * n; dmax += (size_t)blockIdx.y * dmax_stride;
    const float a = (float)((1.0 / (double)(*dmax)) * 255);
    int v = d_cvRound((float)depth[i] * a);
    v = min(max(v, 0), 65535);
    out[i] = (uint8_t)min(v, 255);
}

// ---------------------------------------------------------------- region-adjacency statistics (SegAndMergeV2 RAG build + cal_hist inputs)
// Input: bit planes [3][C][h*wpr] (img = piece mask, dil = 7-dilated piece, lj = "lianjie" fake-edge mask), 64 px per word.
// Each lane takes one pixel; the 64 lanes of a wavefront share the plane words (broadcast loads) and assemble the pixel's
// membership words.  One pass produces overlap[i][j] = |dil_i & dil_j|, overlapPlane[i][j] = |dil_i & dil_j & occ2|,
// ljOverlap[i][j], ljArea[i] and hist[i][v] = 256-bin histogram of the normalised depth over img_i (value 255 dropped:
// calcHist ranges {0,255}).  Counters are privatised in LDS when they fit (C <= 64) and flushed once per workgroup.
template <bool USE_LDS>
__global__ void __launch_bounds__(256) k_rag_stats(const unsigned long long* __restrict__ planes, int C, int w, int h, int wpr,
                                                   const uint8_t* __restrict__ occ2, const uint8_t* __restrict__ depthN,
                                                   int* __restrict__ overlap, int* __restrict__ overlapPlane, int* __restrict__ ljOverlap,
                                                   int* __restrict__ ljArea, int* __restrict__ hist) {
    extern __shared__ int sm[];
    int *lh = hist, *lo = overlap, *lp = overlapPlane, *ll = ljOverlap, *la = ljArea;
    if (USE_LDS) {
        lh = sm; lo = lh + C * 256; lp = lo + C * C; ll = lp + C * C; la = ll + C * C;
        const int total = C * 256 + 3 * C * C + C;
        for (int i = threadIdx.x; i < total; i += 256) sm[i] = 0;
        __syncthreads();
    }
    const int NW = (C + 63) >> 6;
    const size_t pw = (size_t)h * wpr;           // words per plane
    const int n = w * h;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w; const size_t widx = (size_t)y * wpr + (x >> 6); const int bit = x & 63;
        // The 64 lanes of a wave sit on the 64 pixels of ONE plane word (w % 64 == 0 and every stride is a multiple of 64): lane c fetches
        // the three words of piece c once, and each lane then picks its own bit out of the words broadcast with readlane -- 3 loads per
        // 64 pieces instead of 3*C dependent loads per pixel, which is what this kernel's time used to be.
        unsigned long long mi[4] = {0, 0, 0, 0}, md[4] = {0, 0, 0, 0}, ml[4] = {0, 0, 0, 0};
        const int lane = threadIdx.x & 63;
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q >= NW) break;
            const int c = (q << 6) + lane;
            unsigned long long wi = 0, wd = 0, wl = 0;
            if (c < C) { wi = planes[((size_t)0 * C + c) * pw + widx]; wd = planes[((size_t)1 * C + c) * pw + widx]; wl = planes[((size_t)2 * C + c) * pw + widx]; }
            const int cnt = min(64, C - (q << 6));
            // lane `bit` wants bit `bit` of piece cc's words as bit cc of its own: the half of the word the lane's pixel lies in is fixed per lane, the half of the result per cc --
            // 32-bit operations throughout (three instead of six per plane and piece: the 64-bit shifts by a lane-varying and by a uniform count were two passes each)
            const bool upper = bit >= 32; const unsigned sh = (unsigned)bit & 31u;
            unsigned ri[2] = {0u, 0u}, rd[2] = {0u, 0u}, rl[2] = {0u, 0u};
            #pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int c0 = hh * 32, c1 = min(cnt, c0 + 32);
                for (int cc = c0; cc < c1; cc++) {
                    const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)wi, cc), ahi = (unsigned)__builtin_amdgcn_readlane((int)(wi >> 32), cc);
                    const unsigned dlo = (unsigned)__builtin_amdgcn_readlane((int)wd, cc), dhi = (unsigned)__builtin_amdgcn_readlane((int)(wd >> 32), cc);
                    const unsigned llo = (unsigned)__builtin_amdgcn_readlane((int)wl, cc), lhi = (unsigned)__builtin_amdgcn_readlane((int)(wl >> 32), cc);
                    const unsigned k = (unsigned)(cc - c0);
                    ri[hh] |= (((upper ? ahi : alo) >> sh) & 1u) << k; rd[hh] |= (((upper ? dhi : dlo) >> sh) & 1u) << k; rl[hh] |= (((upper ? lhi : llo) >> sh) & 1u) << k;
                }
            }
            mi[q] = ((unsigned long long)ri[1] << 32) | ri[0]; md[q] = ((unsigned long long)rd[1] << 32) | rd[0]; ml[q] = ((unsigned long long)rl[1] << 32) | rl[0];
        }
        const int dv = depthN[i];
        if (dv < 255) {
            #pragma unroll
            for (int q = 0; q < 4; q++) { if (q >= NW) break; unsigned long long m = mi[q]; while (m) { const int c = (q << 6) + __ffsll((long long)m) - 1; m &= m - 1; atomicAdd(&lh[c * 256 + dv], 1); } }
        }
        int nd = 0, nl = 0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { nd += __popcll(md[q]); nl += __popcll(ml[q]); }
        if (nd >= 2) {
            const bool pl = occ2[i] != 0;
            for (int ci = 0; ci < C; ci++) { if (!((md[ci >> 6] >> (ci & 63)) & 1ull)) continue;
                for (int cj = ci + 1; cj < C; cj++) { if (!((md[cj >> 6] >> (cj & 63)) & 1ull)) continue; atomicAdd(&lo[ci * C + cj], 1); if (pl) atomicAdd(&lp[ci * C + cj], 1); } }
        }
        if (nl >= 1) {
            for (int ci = 0; ci < C; ci++) { if (!((ml[ci >> 6] >> (ci & 63)) & 1ull)) continue; atomicAdd(&la[ci], 1);
                for (int cj = ci + 1; cj < C; cj++) { if (!((ml[cj >> 6] >> (cj & 63)) & 1ull)) continue; atomicAdd(&ll[ci * C + cj], 1); } }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < C * 256; i += 256) if (lh[i]) atomicAdd(&hist[i], lh[i]);
        for (int i = threadIdx.x; i < C * C; i += 256) { if (lo[i]) atomicAdd(&overlap[i], lo[i]); if (lp[i]) atomicAdd(&overlapPlane[i], lp[i]); if (ll[i]) atomicAdd(&ljOverlap[i], ll[i]); }
        for (int i = threadIdx.x; i < C; i += 256) if (la[i]) atomicAdd(&ljArea[i], la[i]);
    }
}

// Elliptical dilation of bit planes (64 pixels per word), dst(x, y) = OR over the element of src(x + j - ax, y + i - ay) with positions
// outside the image ignored: the 7x7 dilation of every piece of SegAndMergeV2 (DD:760-ish "imgEachClusterDilate") for the region-adjacency
// statistics.  One thread per output word; a row of the element is a run of <= 15 shifts over a three-word window.
// zero / nzero (optional): a block of ints cleared on the way -- the accumulators of k_rag_stats, which follows on the same stream (one fill launch less per frame).
__global__ void k_dilate_planes(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int nplanes, int wpr, int H, MorphElem e,
                                unsigned long long tail_mask, int* __restrict__ zero, int nzero) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, pw = (size_t)wpr * H;
    for (size_t z = idx; z < (size_t)nzero; z += (size_t)gridDim.x * blockDim.x) zero[z] = 0;
    if (idx >= pw * nplanes) return;
    const int c = (int)(idx / pw), rem = (int)(idx - (size_t)c * pw), y = rem / wpr, k = rem - y * wpr;
    const unsigned long long* sp = src + (size_t)c * pw;
    unsigned long long acc = 0ull;
    for (int i = 0; i < e.n; i++) {
        const int ys = y + i - e.ay;
        if (ys < 0 || ys >= H || e.j2[i] <= e.j1[i]) continue;
        const unsigned long long* r = sp + (size_t)ys * wpr;
        const unsigned long long cur = r[k], prev = k > 0 ? r[k - 1] : 0ull, next = k + 1 < wpr ? r[k + 1] : 0ull;
        for (int j = e.j1[i]; j < e.j2[i]; j++) {
            const int t = j - e.ax;                      // dst bit x takes src bit x + t
            if (t == 0) acc |= cur;
            else if (t > 0) acc |= (cur >> t) | (next << (64 - t));
            else acc |= (cur << (-t)) | (prev >> (64 + t));
        }
    }
    if (k == wpr - 1) acc &= tail_mask;
    dst[idx] = acc;
}
int launch_dilate_planes(hipStream_t s, const unsigned long long* src, unsigned long long* dst, int nplanes, int w, int h, int n, int* zero, int nzero) {
    const int wpr = (w + 63) / 64; const size_t total = (size_t)wpr * h * nplanes;
    const unsigned long long tm = (w & 63) ? ((1ull << (w & 63)) - 1) : ~0ull;
    hipLaunchKernelGGL(k_dilate_planes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, nplanes, wpr, h, make_ellipse(n), tm, zero, zero ? nzero : 0);
    return SIND_OK;
}

// ---------------------------------------------------------------- launchers
int launch_depth_half(hipStream_t s, const uint16_t* src, uint16_t* dst, int dw, int dh, int B, size_t src_stride, size_t dst_stride) {
    hipLaunchKernelGGL(k_depth_half, dim3(divup(dw, 128), dh, B), dim3(128), 0, s, src, dst, dw, dh, src_stride, dst_stride); return SIND_OK; }
int launch_points(hipStream_t s, const uint16_t* depth, float* px, float* py, float* pz, int w, int h, float scale, float fx, float fy, float cx, float cy, float depthScale,
                  int B, size_t depth_stride, size_t pt_stride) {
    hipLaunchKernelGGL(k_points, dim3(divup(w, 128), h, B), dim3(128), 0, s, depth, px, py, pz, w, h, scale, fx, fy, cx, cy, depthScale, 1.5f, depth_stride, pt_stride); return SIND_OK; }
int launch_labels_grid(hipStream_t s, int* labels, int w, int h, int B, size_t stride, const int* use_prev) {
    hipLaunchKernelGGL(k_labels_grid, dim3(divup(w, 128), h, B), dim3(128), 0, s, labels, w, h, (float)h / 3, (float)w / 4, 4, stride, use_prev); return SIND_OK; }
int launch_labels_resize_u8(hipStream_t s, const uint8_t* src, int* dst, int sw, int sh, int dw, int dh, int B, size_t src_stride, size_t dst_stride, const int* use_prev) {
    hipLaunchKernelGGL(k_labels_resize<uint8_t>, dim3(divup(dw, 128), dh, B), dim3(128), 0, s, src, dst, sw, sh, dw, dh, 1. / ((double)dw / sw), 1. / ((double)dh / sh), src_stride, dst_stride, use_prev); return SIND_OK; }
int launch_labels_resize_i32(hipStream_t s, const int* src, int* dst, int sw, int sh, int dw, int dh, int B, size_t src_stride, size_t dst_stride) {
    hipLaunchKernelGGL(k_labels_resize<int>, dim3(divup(dw, 128), dh, B), dim3(128), 0, s, src, dst, sw, sh, dw, dh, 1. / ((double)dw / sw), 1. / ((double)dh / sh), src_stride, dst_stride, (const int*)nullptr); return SIND_OK; }
KmFuse g_km_fuse_default;            // what a handle created from now on starts with (depth.hpp)
int launch_kmeans_level(hipStream_t s, const KmFuse& fuse, const float* px, const float* py, const float* pz, int* labels, int n, int* segcnt, float* comp, KmState* st,
                        int maxCount, double eps2, int B, size_t pt_stride, size_t lab_stride, size_t seg_stride, size_t comp_stride, size_t st_stride) {
    // nseg wave-segments of seg_len (multiple of 64) contiguous points; at most KM_MAX_BLOCKS * KM_WAVES rows in the count table, then the totals row
    // and the 36 sequential sums of the pass
    const int nb = std::min(divup(n, 1024), KM_MAX_BLOCKS), nseg = nb * 4, seg_len = divup(divup(n, nseg), 64) * 64, iters = std::max(maxCount, 2);
    float* seqsums = reinterpret_cast<float*>(segcnt + (KM_MAX_BLOCKS * 4 + 1) * KM_K);
    const KmStride ks{pt_stride, lab_stride, seg_stride, comp_stride, st_stride};
    // A coarse level of a BATCH of frames: every pass in one launch, one workgroup per frame (k_km_level_fused).  Measured (profiles/r04/kmeans_fused.txt): next to the flow
    // solver a round of 128 frames takes 17-18 ms instead of 23.4 (the ~40 small launches it replaces each wait for a CU slot); a single frame or a small batch on an idle GPU is
    // FASTER with the per-pass kernels, whose blocks spread over many CUs (in-order mode 245 frames/s against 190; a round of 13 replayed frames 2.4 ms against 3.4).
    if (n <= fuse.max_points && B >= fuse.min_batch && KMF_WAVES <= KM_MAX_BLOCKS * 4) {
        const int fseg = divup(divup(n, KMF_WAVES), 64) * 64;
        hipLaunchKernelGGL(k_km_level_fused, dim3(1, B), dim3(64 * KMF_WAVES), 0, s, px, py, pz, labels, n, fseg, segcnt, comp, st, ks, maxCount, eps2, iters);
        return SIND_OK;
    }
    const int ticket_word = (KM_MAX_BLOCKS * 4 + 1) * KM_K + KM_TICKET;
    hipLaunchKernelGGL(k_km_count, dim3(nb, B), dim3(256), 0, s, labels, n, seg_len, segcnt, st, ks, maxCount, eps2, ticket_word);
    for (int it = 0; it < iters; it++) {           // every kernel is a no-op for a frame whose centre step has set st->done
        if (it > 0) hipLaunchKernelGGL(k_km_assign_count, dim3(nb, B), dim3(256), 0, s, px, py, pz, labels, n, seg_len, segcnt, st, ks);
        hipLaunchKernelGGL(k_km_compact, dim3(nb, B), dim3(256), 0, s, px, py, pz, labels, n, seg_len, nseg, segcnt, comp, segcnt + nseg * KM_K, st, ks);
        hipLaunchKernelGGL(k_km_seqsum, dim3(KM_K * 3, B), dim3(256), 0, s, segcnt, nseg, comp, n, seqsums, st, px, py, pz, labels, ks);
    }
    return SIND_OK;
}
// parity-test access: the sequential FP32 sum of n host floats through k_km_seqsum's window arithmetic (one run)
int debug_seqsum(hipStream_t s, const float* x_dev, int n, int* scratch_dev /* 2 * KM_K ints + 64 floats + a KmState */, float* out_host) {
    int* segcnt = scratch_dev; float* sums = reinterpret_cast<float*>(scratch_dev + 2 * KM_K); KmState* st = reinterpret_cast<KmState*>(scratch_dev + 2 * KM_K + 64);
    int h[2 * KM_K] = {n}; h[KM_K] = n;                     // one table row + the totals row: everything is cluster 0, run 0 = coordinate plane 0
    HIP_TRY(hipMemsetAsync(scratch_dev, 0, (2 * KM_K + 64) * sizeof(int), s));
    HIP_TRY(hipMemcpyAsync(segcnt, h, sizeof(h), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(st, 0, sizeof(KmState), s));
    hipLaunchKernelGGL(k_km_seqsum, dim3(1), dim3(256), 0, s, segcnt, 1, x_dev, n, sums, st, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (int*)nullptr, KmStride{0, 0, 0, 0, 0});
    HIP_TRY(hipMemcpyAsync(out_host, sums, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SIND_OK;
}
int launch_labels_to_u8(hipStream_t s, const int* labels, uint8_t* out, int n, int B, size_t lab_stride, size_t out_stride) {
    hipLaunchKernelGGL(k_labels_to_u8, dim3(divup(n, 256), B), dim3(256), 0, s, labels, out, n, lab_stride, out_stride); return SIND_OK; }
// B frames per launch (frame b at offset b * w * h of every image argument; maxima at out + b * out_stride)
int launch_median5(hipStream_t s, const uint16_t* src, uint16_t* dst, int w, int h, int B) { hipLaunchKernelGGL(k_median5_u16, dim3(divup(w, 64), divup(h, MED_ROWS), B), dim3(64), 0, s, src, dst, w, h); return SIND_OK; }
int launch_max_u16(hipStream_t s, const uint16_t* src, int n, unsigned* out, int B, int out_stride) {
    if (B == 1 || out_stride == 1) HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * sizeof(unsigned), s));
    else for (int b = 0; b < B; b++) HIP_TRY(hipMemsetAsync(out + (size_t)b * out_stride, 0, sizeof(unsigned), s));
    // 16-byte loads need every image of the launch to start on a 16-byte boundary: the base pointer aligned (a per-frame pointer into a batch buffer is only when the frame
    // size is a multiple of 8 samples) and, for a batch, a multiple of 8 samples per image; anything else takes the 2-byte loop
    const int wide = ((uintptr_t)src % 16 == 0 && (B == 1 || n % 8 == 0)) ? 1 : 0;
    hipLaunchKernelGGL(k_max_u16, dim3(std::max(1, std::min(divup(n, 8 * 256), 64)), B), dim3(256), 0, s, src, n, out, out_stride, wide); return SIND_OK; }
int launch_grad_edge(hipStream_t s, const uint16_t* filt, const unsigned* dmax, uint8_t* edge, uint8_t* total_area, int w, int h, float depthScale, int B, int dmax_stride) {
    hipLaunchKernelGGL(k_grad_edge, dim3(divup(w, 128), divup(h, GE_ROWS), B), dim3(128), 0, s, filt, dmax, edge, total_area, w, h, depthScale, dmax_stride); return SIND_OK; }
MorphElem make_ellipse(int n) {
    MorphElem e; e.n = n; e.ax = n / 2; e.ay = n / 2;
    for (int i = 0; i < MORPH_MAX; i++) { e.j1[i] = 0; e.j2[i] = 0; }
    if (n == 1) { e.j2[0] = 1; return e; }
    const int r = n / 2, c = n / 2; const double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < n; i++) {
        const int dy = i - r;
        if (std::abs(dy) <= r) { const int dx = (int)std::lrint(c * std::sqrt((r * r - dy * dy) * inv_r2)); e.j1[i] = std::max(c - dx, 0); e.j2[i] = std::min(c + dx + 1, n); }
    }
    return e;
}
int launch_morph(hipStream_t s, const uint8_t* src, uint8_t* dst, int w, int h, int n, bool dilate, int B) {
    if (n == 4) { hipLaunchKernelGGL(k_morph_e4, dim3(divup(w, 128), divup(h, MO_ROWS), B), dim3(128), 0, s, src, dst, w, h, dilate ? 1 : 0); return SIND_OK; }
    hipLaunchKernelGGL(k_morph, dim3(divup(w, 128), h, B), dim3(128), 0, s, src, dst, w, h, make_ellipse(n), dilate ? 1 : 0); return SIND_OK; }
int launch_peac_block_stats(hipStream_t s, const uint16_t* depth, int w, int h, int bw, int bh, float fx, float fy, float cx, float cy, float depthScale, PeacBlockStats* out, int B) {
    if (bw != PEAC_BW || bh != PEAC_BW) { sind_set_error("peac_block_stats: %d x %d blocks (only %d x %d)", bw, bh, PEAC_BW, PEAC_BW); return SIND_E_ARG; }
    const int nb = (w / bw) * (h / bh);
    hipLaunchKernelGGL(k_peac_block_stats, dim3(divup(nb, PEAC_BPW), B), dim3(64), 0, s, depth, w, h, w / bw, nb, fx, fy, cx, cy, depthScale, 0.04, 0.02 * 1000, out);
    hipLaunchKernelGGL(k_peac_block_fit, dim3(divup(nb * B, 64)), dim3(64), 0, s, out, nb * B); return SIND_OK; }
int launch_depth_norm(hipStream_t s, const uint16_t* depth, const unsigned* dmax, uint8_t* out, int n, int B, int dmax_stride) {
    hipLaunchKernelGGL(k_depth_norm, dim3(divup(n, 256), B), dim3(256), 0, s, depth, dmax, out, n, dmax_stride); return SIND_OK; }
int launch_rag_stats(hipStream_t s, const unsigned long long* planes, int C, int w, int h, int wpr, const uint8_t* occ2, const uint8_t* depthN,
                     int* overlap, int* overlapPlane, int* ljOverlap, int* ljArea, int* hist, bool already_zero) {
    if (C < 1 || C > 254) { sind_set_error("rag_stats: %d pieces unsupported (1..254)", C); return SIND_E_ARG; }
    if (already_zero) {}                                  // cleared by the dilation kernel ahead of this one on the stream
    else if (overlapPlane == overlap + C * C && ljOverlap == overlapPlane + C * C && ljArea == ljOverlap + C * C && hist == ljArea + C)
        HIP_TRY(hipMemsetAsync(overlap, 0, ((size_t)3 * C * C + C + (size_t)C * 256) * sizeof(int), s));        // the caller's five outputs are one block: one fill
    else {
        HIP_TRY(hipMemsetAsync(overlap, 0, (size_t)C * C * sizeof(int), s)); HIP_TRY(hipMemsetAsync(overlapPlane, 0, (size_t)C * C * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(ljOverlap, 0, (size_t)C * C * sizeof(int), s)); HIP_TRY(hipMemsetAsync(ljArea, 0, (size_t)C * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(hist, 0, (size_t)C * 256 * sizeof(int), s));
    }
    if (C <= 64) {
        const size_t shm = ((size_t)C * 256 + 3 * (size_t)C * C + C) * sizeof(int);      // <= 64 KB + 48 KB + 256 B of the 160 KB LDS
        static SindPerDeviceInit attr_init;       // the pool's workers call this concurrently; once per device
        HIP_TRY(attr_init.run([] { return hipFuncSetAttribute((const void*)k_rag_stats<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); }));
        hipLaunchKernelGGL(k_rag_stats<true>, dim3(128), dim3(256), shm, s, planes, C, w, h, wpr, occ2, depthN, overlap, overlapPlane, ljOverlap, ljArea, hist);
    } else {
        hipLaunchKernelGGL(k_rag_stats<false>, dim3(256), dim3(256), 0, s, planes, C, w, h, wpr, occ2, depthN, overlap, overlapPlane, ljOverlap, ljArea, hist);
    }
    return SIND_OK;
}

}  // namespace sind
