// Depth-side device stages of DynaDetect for gfx950 (reference ORB_SLAM2/src/DynaDetect.cc):
//   SegByKmeans (:315-420): depth pyramid, back-projection, label up-sampling, cv::kmeans assignment + centre sums
//   CalOccluded (:429-482): medianBlur(5), 5x5 max-difference depth edge, valid-area mask
//   morphologyEx with elliptical elements (:51-59 and every MORPH_* call)
//   PEAC initial 16x16 block statistics (PEAC/AHCPlaneSeg.hpp:180-262)
//   SegAndMergeV2 region-adjacency statistics (:784-893, cal_hist :1685-1739) from per-pixel membership words
// Integer stages are bit-exact; FP32 expressions are written in the reference's operation order (-ffp-contract=off).
#include "common.hpp"
#include "depth.hpp"

namespace sind {

// ---------------------------------------------------------------- cv::resize(INTER_LINEAR) with exact 2x -> INTER_AREA fast path, CV_16U
__global__ void k_depth_half(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int sw = dw * 2;
    const int s = src[(2 * y) * sw + 2 * x] + src[(2 * y) * sw + 2 * x + 1] + src[(2 * y + 1) * sw + 2 * x] + src[(2 * y + 1) * sw + 2 * x + 1];
    dst[y * dw + x] = (uint16_t)((s + 2) >> 2);
}

// ---------------------------------------------------------------- back-projection (DD:347-369), SoA points
__global__ void k_points(const uint16_t* __restrict__ depth, float* __restrict__ px, float* __restrict__ py, float* __restrict__ pz,
                         int w, int h, float scale, float fx, float fy, float cx, float cy, float depthScale, float depth_weight) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
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
__global__ void k_labels_grid(int* __restrict__ labels, int w, int h, float batch_rows, float batch_cols, int ncol) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= w) return;
    labels[i * w + j] = (int)floorf(i / batch_rows) * ncol + (int)floorf(j / batch_cols);
}
// bilinear resize of a label image converted to float, then cvRound (DD:390-394, 402-406)
template <class T>
__global__ void k_labels_resize(const T* __restrict__ src, int* __restrict__ dst, int sw, int sh, int dw, int dh, double scale_x, double scale_y) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
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
// Centre sums: per-workgroup partials in FP64 with a fixed-order final sum (deterministic; OpenCV sums sequentially in FP32, so
// centres agree to ~1e-7 relative and labels on all but <=1e-3 of the pixels).  One KmState per pyramid level; k_km_update runs
// the centre step (sums -> centres, empty-cluster repair, shift test, last-iteration decision) with cv::kmeans' operations.
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
// Wave-level sum of the 48 per-lane accumulators (12 clusters x {x, y, z, count}) as a reduce-scatter: in every step a lane hands half
// of its values to its partner and keeps the other half, so 51 shuffles replace 48 x 6; after the xor-4 step three values are left per
// lane, finished by two plain exchange steps.  Lane l (l % 4 == 0) ends up with the sums of accumulators j + 3*b2 + 6*b3 + 12*b4 + 24*b5
// (b = bits of l), which it writes to acc[...][wave].  Fixed order, hence deterministic.
__device__ __forceinline__ void km_wave_reduce_store(double (&s)[KM_K][4], double (*acc)[4][4], int lane, int wv) {
    double v[48];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) { v[4 * k] = s[k][0]; v[4 * k + 1] = s[k][1]; v[4 * k + 2] = s[k][2]; v[4 * k + 3] = s[k][3]; }
    #define KM_STEP(HALF, MASK)                                                                               \
        _Pragma("unroll")                                                                                     \
        for (int j = 0; j < (HALF); j++) {                                                                    \
            const bool up = (lane & (MASK)) != 0;                                                             \
            const double send = up ? v[j] : v[j + (HALF)], keep = up ? v[j + (HALF)] : v[j];                  \
            v[j] = keep + __shfl_xor(send, (MASK));                                                           \
        }
    KM_STEP(24, 32) KM_STEP(12, 16) KM_STEP(6, 8) KM_STEP(3, 4)
    #undef KM_STEP
    #pragma unroll
    for (int j = 0; j < 3; j++) { v[j] += __shfl_xor(v[j], 2); v[j] += __shfl_xor(v[j], 1); }
    if ((lane & 3) == 0) {
        const int base = ((lane >> 2) & 1) * 3 + ((lane >> 3) & 1) * 6 + ((lane >> 4) & 1) * 12 + ((lane >> 5) & 1) * 24;
        #pragma unroll
        for (int j = 0; j < 3; j++) { const int a = base + j; acc[a >> 2][a & 3][wv] = v[j]; }
    }
}
// Centre step of one k-means iteration in ONE workgroup: reduce the per-block partial sums, repair every empty cluster
// (block-wide farthest-point search over the biggest cluster, as cv::kmeans does, repeated until no cluster is empty), scale,
// shift test, stop decision.  No host round trip and no provisioning limit.
__global__ void __launch_bounds__(1024) k_km_update(const double* __restrict__ partial, int nblocks, KmState* __restrict__ gst,
                                                    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                    int* __restrict__ labels, int n) {
    __shared__ double sums[KM_K * 4];
    __shared__ double stage[KM_MAX_BLOCKS * KM_K * 4];
    __shared__ unsigned long long wbest[16];
    __shared__ int s_fix;
    __shared__ KmState S;                                  // the state lives in LDS while the centre step runs: the serial part below touches it
                                                           // ~100 times, and every touch of the global copy would be a dependent ~1 us round trip
    static_assert(sizeof(KmState) % 4 == 0, "KmState is copied word by word");
    if (gst->done) return;
    const int t = threadIdx.x;
    for (int i = t; i < (int)(sizeof(KmState) / 4); i += blockDim.x) reinterpret_cast<unsigned*>(&S)[i] = reinterpret_cast<const unsigned*>(gst)[i];
    for (int i = t; i < nblocks * KM_K * 4; i += blockDim.x) stage[i] = partial[i];     // parallel fetch, then a fixed-order (deterministic) sum
    __syncthreads();
    KmState* st = &S;
    if (t < KM_K * 4) { double v = 0; for (int b = 0; b < nblocks; b++) v += stage[b * KM_K * 4 + t]; sums[t] = v; }
    __syncthreads();
    if (t == 0) {
        st->phase = 0;
        for (int k = 0; k < KM_K; k++) { for (int j = 0; j < 3; j++) { st->old[k][j] = st->ctr[k][j]; st->ctr[k][j] = (float)sums[k * 4 + j]; } st->cnt[k] = (int)sums[k * 4 + 3]; }
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
__global__ void __launch_bounds__(256) k_km_partial_dev(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                        const int* __restrict__ labels, int n, double* __restrict__ partial, const KmState* __restrict__ st) {
    if (st->done) return;
    __shared__ double acc[KM_K][4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double s[KM_K][4];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) { s[k][0] = s[k][1] = s[k][2] = s[k][3] = 0.0; }
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const int l = labels[i]; const double x = px[i], y = py[i], z = pz[i];
        #pragma unroll
        for (int k = 0; k < KM_K; k++) if (l == k) { s[k][0] += x; s[k][1] += y; s[k][2] += z; s[k][3] += 1.0; }
    }
    km_wave_reduce_store(s, acc, lane, wv);
    __syncthreads();
    if (tid < KM_K * 4) { const int k = tid >> 2, c = tid & 3; partial[((size_t)blockIdx.x * KM_K + k) * 4 + c] = ((acc[k][c][0] + acc[k][c][1]) + acc[k][c][2]) + acc[k][c][3]; }
}
// re-assignment to the nearest centre fused with the centre sums of the next iteration (cv::kmeans' assignment step followed by the sums of
// k_km_partial_dev: one pass over the points and one launch instead of two)
__global__ void __launch_bounds__(256) k_km_assign_partial(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                                           int* __restrict__ labels, int n, double* __restrict__ partial, const KmState* __restrict__ st) {
    if (st->done || st->phase != 1) return;
    __shared__ double acc[KM_K][4][4];
    __shared__ float ctr[KM_K][3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < KM_K * 3) ctr[tid / 3][tid % 3] = st->ctr[tid / 3][tid % 3];
    __syncthreads();
    double s[KM_K][4];
    #pragma unroll
    for (int k = 0; k < KM_K; k++) { s[k][0] = s[k][1] = s[k][2] = s[k][3] = 0.0; }
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const float xf = px[i], yf = py[i], zf = pz[i];
        int best = 0; float md = 3.402823466e+38f;
        #pragma unroll
        for (int k = 0; k < KM_K; k++) {
            float t = xf - ctr[k][0]; float dist = 0.f; dist += t * t;
            t = yf - ctr[k][1]; dist += t * t;
            t = zf - ctr[k][2]; dist += t * t;
            if (md > dist) { md = dist; best = k; }
        }
        labels[i] = best;
        const double x = xf, y = yf, z = zf;
        #pragma unroll
        for (int k = 0; k < KM_K; k++) if (best == k) { s[k][0] += x; s[k][1] += y; s[k][2] += z; s[k][3] += 1.0; }
    }
    km_wave_reduce_store(s, acc, lane, wv);
    __syncthreads();
    if (tid < KM_K * 4) { const int k = tid >> 2, c = tid & 3; partial[((size_t)blockIdx.x * KM_K + k) * 4 + c] = ((acc[k][c][0] + acc[k][c][1]) + acc[k][c][2]) + acc[k][c][3]; }
}
__global__ void k_km_reset(KmState* st, int maxCount, double eps2) {
    if (threadIdx.x == 0) { st->iter = 0; st->done = 0; st->phase = 0; st->overflow = 0; st->fix_k = -1; st->max_k = 0; st->far = 0ull; st->maxCount = maxCount; st->eps2 = eps2;
        for (int k = 0; k < KM_K; k++) { st->cnt[k] = 0; for (int j = 0; j < 3; j++) { st->ctr[k][j] = 0.f; st->old[k][j] = 0.f; } } }
}

__global__ void k_labels_to_u8(const int* __restrict__ labels, uint8_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = labels[i]; out[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}

// ---------------------------------------------------------------- medianBlur(5) on the depth image (values are integers, so the float median == u16 median)
__global__ void k_median5_u16(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int v[25];
    #pragma unroll
    for (int dy = -2; dy <= 2; dy++) {
        const uint16_t* r = src + (size_t)min(max(y + dy, 0), h - 1) * w;
        #pragma unroll
        for (int dx = -2; dx <= 2; dx++) v[(dy + 2) * 5 + dx + 2] = r[min(max(x + dx, 0), w - 1)];
    }
    int med = v[0];
    #pragma unroll
    for (int i = 0; i < 25; i++) {
        int lt = 0, le = 0;
        #pragma unroll
        for (int j = 0; j < 25; j++) { lt += v[j] < v[i]; le += v[j] <= v[i]; }
        if (lt <= 12 && 12 < le) med = v[i];
    }
    dst[y * w + x] = (uint16_t)med;
}
__global__ void k_max_u16(const uint16_t* __restrict__ src, int n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, (unsigned)src[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// 5x5 max-difference depth edge + valid-area mask (DD:443-482); the 3-px frame is left 0 in both outputs
__global__ void k_grad_edge(const uint16_t* __restrict__ filt, const unsigned* __restrict__ dmax, uint8_t* __restrict__ edge,
                            uint8_t* __restrict__ total_area, int w, int h, float depthScale) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (col >= w) return;
    uint8_t e = 0, t = 0;
    if (row >= 3 && row < h - 3 && col >= 3 && col < w - 3) {
        const float depth_max = (float)(*dmax);
        const float depth1 = (float)filt[row * w + col];
        if (depth1 > 0.0f && depth1 / depthScale < 6.0f) t = 255;
        float val_max = 0.0f;
        #pragma unroll
        for (int i = -2; i <= 2; i++)
            #pragma unroll
            for (int j = -2; j <= 2; j++) {
                const float nb = (float)filt[(row + i) * w + col + j];
                if ((depth1 - nb) > depth_max * 0.5f) continue;
                const float a = fabsf(depth1 - nb);
                val_max = fabsf(val_max) > a ? fabsf(val_max) : a;
            }
        if (val_max > depth1 * 0.03f && val_max > 400.0f) e = 255;
    }
    edge[row * w + col] = e; total_area[row * w + col] = t;
}

// ---------------------------------------------------------------- morphology with an elliptical element (max / min over in-image pixels)
__global__ void k_morph(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int w, int h, MorphElem E, int is_dilate) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
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

// ---------------------------------------------------------------- PEAC initial block statistics: ONE thread per 16x16 block so that the FP64
// sums are accumulated in the reference's row-major order (bit-exact with a sequential CPU loop; 1200 blocks -> negligible time).
__global__ void k_peac_block_stats(const uint16_t* __restrict__ depth, int w, int h, int bw, int bh, float fx, float fy, float cx, float cy,
                                   float depthScale, double depthAlpha, double depthChangeTol, PeacBlockStats* __restrict__ out) {
    const int Nw = w / bw, Nh = h / bh;
    const int blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= Nw * Nh) return;
    const int by = blk / Nw, bx = blk - by * Nw;
    PeacBlockStats S; S.sx = S.sy = S.sz = S.sxx = S.syy = S.szz = S.sxy = S.syz = S.sxz = 0; S.N = 0; S.valid = 1;
    auto getz = [&](int i, int j, double& x, double& y, double& z) -> bool {
        const float d = (float)depth[i * w + j];
        if (d < 1e-3f) return false;                     // NaN point in the reference's organised cloud
        const float zf = d * (1.0f / depthScale);
        x = (double)((j - cx) * zf / fx); y = (double)((i - cy) * zf / fy); z = (double)zf;
        return true;
    };
    for (int i = by * bh; i < (by + 1) * bh && i < h && S.valid; i++)
        for (int j = bx * bw; j < (bx + 1) * bw && j < w; j++) {
            double x, y, z, xn, yn, zn;
            if (!getz(i, j, x, y, z)) { S.valid = 0; break; }
            if (j + 1 < w && getz(i, j + 1, xn, yn, zn) && fabs(z - zn) > depthAlpha * fabs(z) + depthChangeTol) { S.valid = 0; break; }
            if (i + 1 < h && getz(i + 1, j, xn, yn, zn) && fabs(z - zn) > depthAlpha * fabs(z) + depthChangeTol) { S.valid = 0; break; }
            S.sx += x; S.sy += y; S.sz += z; S.sxx += x * x; S.syy += y * y; S.szz += z * z; S.sxy += x * y; S.syz += y * z; S.sxz += x * z; S.N++;
        }
    out[blk] = S;
}

// ---------------------------------------------------------------- imgDepth/depth_max*255 -> 8U (DD:765-768): u16 * (float)((1/max)*255), cvRound, saturate
__global__ void k_depth_norm(const uint16_t* __restrict__ depth, const unsigned* __restrict__ dmax, uint8_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
            for (int cc = 0; cc < cnt; cc++) {
                const unsigned long long a = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(wi >> 32), cc) << 32) | (unsigned)__builtin_amdgcn_readlane((int)wi, cc);
                const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(wd >> 32), cc) << 32) | (unsigned)__builtin_amdgcn_readlane((int)wd, cc);
                const unsigned long long l = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(wl >> 32), cc) << 32) | (unsigned)__builtin_amdgcn_readlane((int)wl, cc);
                mi[q] |= ((a >> bit) & 1ull) << cc; md[q] |= ((d >> bit) & 1ull) << cc; ml[q] |= ((l >> bit) & 1ull) << cc;
            }
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
__global__ void k_dilate_planes(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int nplanes, int wpr, int H, MorphElem e,
                                unsigned long long tail_mask) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, pw = (size_t)wpr * H;
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
int launch_dilate_planes(hipStream_t s, const unsigned long long* src, unsigned long long* dst, int nplanes, int w, int h, int n) {
    const int wpr = (w + 63) / 64; const size_t total = (size_t)wpr * h * nplanes;
    const unsigned long long tm = (w & 63) ? ((1ull << (w & 63)) - 1) : ~0ull;
    hipLaunchKernelGGL(k_dilate_planes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, nplanes, wpr, h, make_ellipse(n), tm);
    return SIND_OK;
}

// ---------------------------------------------------------------- launchers
int launch_depth_half(hipStream_t s, const uint16_t* src, uint16_t* dst, int dw, int dh) { hipLaunchKernelGGL(k_depth_half, dim3(divup(dw, 128), dh), dim3(128), 0, s, src, dst, dw, dh); return SIND_OK; }
int launch_points(hipStream_t s, const uint16_t* depth, float* px, float* py, float* pz, int w, int h, float scale, float fx, float fy, float cx, float cy, float depthScale) {
    hipLaunchKernelGGL(k_points, dim3(divup(w, 128), h), dim3(128), 0, s, depth, px, py, pz, w, h, scale, fx, fy, cx, cy, depthScale, 1.5f); return SIND_OK; }
int launch_labels_grid(hipStream_t s, int* labels, int w, int h) {
    hipLaunchKernelGGL(k_labels_grid, dim3(divup(w, 128), h), dim3(128), 0, s, labels, w, h, (float)h / 3, (float)w / 4, 4); return SIND_OK; }
int launch_labels_resize_u8(hipStream_t s, const uint8_t* src, int* dst, int sw, int sh, int dw, int dh) {
    hipLaunchKernelGGL(k_labels_resize<uint8_t>, dim3(divup(dw, 128), dh), dim3(128), 0, s, src, dst, sw, sh, dw, dh, 1. / ((double)dw / sw), 1. / ((double)dh / sh)); return SIND_OK; }
int launch_labels_resize_i32(hipStream_t s, const int* src, int* dst, int sw, int sh, int dw, int dh) {
    hipLaunchKernelGGL(k_labels_resize<int>, dim3(divup(dw, 128), dh), dim3(128), 0, s, src, dst, sw, sh, dw, dh, 1. / ((double)dw / sw), 1. / ((double)dh / sh)); return SIND_OK; }
int launch_kmeans_level(hipStream_t s, const float* px, const float* py, const float* pz, int* labels, int n, double* partial, KmState* st,
                        int maxCount, double eps2) {
    const int nb = std::min(divup(n, 256), KM_MAX_BLOCKS), iters = std::max(maxCount, 2);
    hipLaunchKernelGGL(k_km_reset, dim3(1), dim3(64), 0, s, st, maxCount, eps2);
    hipLaunchKernelGGL(k_km_partial_dev, dim3(nb), dim3(256), 0, s, px, py, pz, labels, n, partial, st);
    hipLaunchKernelGGL(k_km_update, dim3(1), dim3(1024), 0, s, partial, nb, st, px, py, pz, labels, n);
    for (int it = 1; it < iters; it++) {           // every later kernel is a no-op once the centre step has set st->done
        hipLaunchKernelGGL(k_km_assign_partial, dim3(nb), dim3(256), 0, s, px, py, pz, labels, n, partial, st);
        hipLaunchKernelGGL(k_km_update, dim3(1), dim3(1024), 0, s, partial, nb, st, px, py, pz, labels, n);
    }
    return SIND_OK;
}
int launch_labels_to_u8(hipStream_t s, const int* labels, uint8_t* out, int n) { hipLaunchKernelGGL(k_labels_to_u8, dim3(divup(n, 256)), dim3(256), 0, s, labels, out, n); return SIND_OK; }
int launch_median5(hipStream_t s, const uint16_t* src, uint16_t* dst, int w, int h) { hipLaunchKernelGGL(k_median5_u16, dim3(divup(w, 64), h), dim3(64), 0, s, src, dst, w, h); return SIND_OK; }
int launch_max_u16(hipStream_t s, const uint16_t* src, int n, unsigned* out) {
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_max_u16, dim3(std::min(divup(n, 256), 256)), dim3(256), 0, s, src, n, out); return SIND_OK; }
int launch_grad_edge(hipStream_t s, const uint16_t* filt, const unsigned* dmax, uint8_t* edge, uint8_t* total_area, int w, int h, float depthScale) {
    hipLaunchKernelGGL(k_grad_edge, dim3(divup(w, 128), h), dim3(128), 0, s, filt, dmax, edge, total_area, w, h, depthScale); return SIND_OK; }
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
int launch_morph(hipStream_t s, const uint8_t* src, uint8_t* dst, int w, int h, int n, bool dilate) {
    hipLaunchKernelGGL(k_morph, dim3(divup(w, 128), h), dim3(128), 0, s, src, dst, w, h, make_ellipse(n), dilate ? 1 : 0); return SIND_OK; }
int launch_peac_block_stats(hipStream_t s, const uint16_t* depth, int w, int h, int bw, int bh, float fx, float fy, float cx, float cy, float depthScale, PeacBlockStats* out) {
    const int nb = (w / bw) * (h / bh);
    hipLaunchKernelGGL(k_peac_block_stats, dim3(divup(nb, 64)), dim3(64), 0, s, depth, w, h, bw, bh, fx, fy, cx, cy, depthScale, 0.04, 0.02 * 1000, out); return SIND_OK; }
int launch_depth_norm(hipStream_t s, const uint16_t* depth, const unsigned* dmax, uint8_t* out, int n) { hipLaunchKernelGGL(k_depth_norm, dim3(divup(n, 256)), dim3(256), 0, s, depth, dmax, out, n); return SIND_OK; }
int launch_rag_stats(hipStream_t s, const unsigned long long* planes, int C, int w, int h, int wpr, const uint8_t* occ2, const uint8_t* depthN,
                     int* overlap, int* overlapPlane, int* ljOverlap, int* ljArea, int* hist) {
    if (C < 1 || C > 254) { sind_set_error("rag_stats: %d pieces unsupported (1..254)", C); return SIND_E_ARG; }
    if (overlapPlane == overlap + C * C && ljOverlap == overlapPlane + C * C && ljArea == ljOverlap + C * C && hist == ljArea + C)
        HIP_TRY(hipMemsetAsync(overlap, 0, ((size_t)3 * C * C + C + (size_t)C * 256) * sizeof(int), s));        // the caller's five outputs are one block: one fill
    else {
        HIP_TRY(hipMemsetAsync(overlap, 0, (size_t)C * C * sizeof(int), s)); HIP_TRY(hipMemsetAsync(overlapPlane, 0, (size_t)C * C * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(ljOverlap, 0, (size_t)C * C * sizeof(int), s)); HIP_TRY(hipMemsetAsync(ljArea, 0, (size_t)C * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(hist, 0, (size_t)C * 256 * sizeof(int), s));
    }
    if (C <= 64) {
        const size_t shm = ((size_t)C * 256 + 3 * (size_t)C * C + C) * sizeof(int);      // <= 64 KB + 48 KB + 256 B of the 160 KB LDS
        static bool attr_set = false;
        if (!attr_set) { HIP_TRY(hipFuncSetAttribute((const void*)k_rag_stats<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024)); attr_set = true; }
        hipLaunchKernelGGL(k_rag_stats<true>, dim3(128), dim3(256), shm, s, planes, C, w, h, wpr, occ2, depthN, overlap, overlapPlane, ljOverlap, ljArea, hist);
    } else {
        hipLaunchKernelGGL(k_rag_stats<false>, dim3(256), dim3(256), 0, s, planes, C, w, h, wpr, occ2, depthN, overlap, overlapPlane, ljOverlap, ljArea, hist);
    }
    return SIND_OK;
}

}  // namespace sind
