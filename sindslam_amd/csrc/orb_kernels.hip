// ORBextractor device stages for gfx950 (reference ORB_SLAM2/src/ORBextractor.cc):
//   ComputePyramid (:1166-1191)  -> k_resize_u8 (flow_kernels.hip) + k_pad_reflect101
//   cell-wise cv::FAST + NMS (:789-828, OpenCV features2d fast.cpp / fast_score.cpp) -> k_fast_cells
//   IC_Angle (:77-104) + cv::fastAtan2 -> k_ic_angle
//   GaussianBlur 7x7 sigma 2, 8U fixed point (:1145) -> k_blur7_h / k_blur7_v
//   computeOrbDescriptor (:108-147) -> k_brief
// All integer stages are bit-exact restatements; batched over B frames (blockIdx.z / .y = frame).
// Memory: per frame one "slab" holding the 8 padded levels back to back (19-px REFLECT_101 border each).
#include "common.hpp"
#include "orb.hpp"
#include "../../include/sind_brief_pattern.h"

namespace sind {

__constant__ signed char c_brief[1024];
__constant__ int c_umax[16];

// copyMakeBorder(BORDER_REFLECT_101) of the level interior (already written at offset (19,19)) into its border.
// One thread per BORDER pixel (the interior is 7/8 of a 640 x 480 level: a thread per padded pixel started eight waves to move one): border pixel k of the padded image, the
// 2 pad full rows above and below first, then the 2 pad columns beside each interior row.
__global__ void k_pad_reflect101(uint8_t* __restrict__ slab, size_t slab_stride, size_t off, int lw, int lh, int pad) {
    const int pw = lw + 2 * pad, nrow = 2 * pad * pw, nall = nrow + 2 * pad * lh;
    const int k = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
    if (k >= nall) return;
    int x, y;
    if (k < nrow) { const int r = k / pw; x = k - r * pw; y = r < pad ? r : lh + r; }                                  // rows 0 .. pad - 1 and lh + pad .. lh + 2 pad - 1
    else { const int q = k - nrow, r = q / (2 * pad), c = q - r * (2 * pad); y = pad + r; x = c < pad ? c : lw + c; }     // columns 0 .. pad - 1 and lw + pad .. lw + 2 pad - 1
    const int ix = x - pad, iy = y - pad;
    uint8_t* L = slab + (size_t)b * slab_stride + off;
    const int sx = d_reflect101(ix, lw) + pad, sy = d_reflect101(iy, lh) + pad;
    L[(size_t)y * pw + x] = L[(size_t)sy * pw + sx];
}
// level 0 interior: copy the gray image into the slab
__global__ void k_copy_into_slab(const uint8_t* __restrict__ gray, uint8_t* __restrict__ slab, size_t slab_stride, size_t off, int w, int h, int pad) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    slab[(size_t)b * slab_stride + off + (size_t)(y + pad) * (w + 2 * pad) + x + pad] = gray[((size_t)b * h + y) * w + x];
}

// ---------------------------------------------------------------------------------------------------------
// FAST-9/16 score (cornerScore<16>): max(threshold, max over 9-arcs of min(d), max over 9-arcs of min(-d)) - 1,
// 0 if the pixel is not a corner at `threshold`.
__device__ __forceinline__ int fast_score(const uint8_t* win, int stride, int idx, int threshold) {
    const int v = win[idx];
    // A 9-arc of the 16-ring holds at least two of the four compass pixels (0, 4, 8, 12): unless two of them are brighter than v + threshold or two darker than v - threshold
    // no arc passes, and the score is 0 without looking at the other twelve (most pixels of an image).  The score of the pixels that do pass is computed as before.
    {
        const int c0 = v - win[idx + 3 * stride], c4 = v - win[idx + 3], c8 = v - win[idx - 3 * stride], c12 = v - win[idx - 3];
        const int darker = (c0 > threshold) + (c4 > threshold) + (c8 > threshold) + (c12 > threshold);             // ring pixel darker than the centre by more than the threshold
        const int brighter = (c0 < -threshold) + (c4 < -threshold) + (c8 < -threshold) + (c12 < -threshold);
        if (darker < 2 && brighter < 2) return 0;
    }
    int d[16];
    d[0] = v - win[idx + 3 * stride];       d[1] = v - win[idx + 3 * stride + 1];  d[2] = v - win[idx + 2 * stride + 2];
    d[3] = v - win[idx + stride + 3];       d[4] = v - win[idx + 3];               d[5] = v - win[idx - stride + 3];
    d[6] = v - win[idx - 2 * stride + 2];   d[7] = v - win[idx - 3 * stride + 1];  d[8] = v - win[idx - 3 * stride];
    d[9] = v - win[idx - 3 * stride - 1];   d[10] = v - win[idx - 2 * stride - 2]; d[11] = v - win[idx - stride - 3];
    d[12] = v - win[idx - 3];               d[13] = v - win[idx + stride - 3];     d[14] = v - win[idx + 2 * stride - 2];
    d[15] = v - win[idx + 3 * stride - 1];
    // extremes of the sixteen 9-arcs by doubling (arcs of 2, 4, 8, then 8 + 1): 64 instead of 128 comparisons per side, the same minima and maxima
    int n2[16], x2[16], n4[16], x4[16];
    #pragma unroll
    for (int s = 0; s < 16; s++) { n2[s] = min(d[s], d[(s + 1) & 15]); x2[s] = max(d[s], d[(s + 1) & 15]); }
    #pragma unroll
    for (int s = 0; s < 16; s++) { n4[s] = min(n2[s], n2[(s + 2) & 15]); x4[s] = max(x2[s], x2[(s + 2) & 15]); }
    int A = -256, Bm = -256;
    #pragma unroll
    for (int s = 0; s < 16; s++) {
        const int mn = min(min(n4[s], n4[(s + 4) & 15]), d[(s + 8) & 15]), mx = max(max(x4[s], x4[(s + 4) & 15]), d[(s + 8) & 15]);
        A = max(A, mn); Bm = max(Bm, -mx);
    }
    const int best = max(A, Bm);
    return best > threshold ? best - 1 : 0;      // corner iff some arc exceeds the threshold strictly
}

// One workgroup per (cell, frame).  Window (<= 40x40) -> LDS, scores for iniTh (retry with minTh if the cell has
// no corner at all... note: OpenCV's retry condition is "no keypoint AFTER nms", reproduced), 3x3 strict-max NMS,
// row-major compaction with wave ballots.  Output: per cell up to cell_cap (x, y, score) records (cell_cap >= the NMS bound of the largest cell: orb.hpp).
__global__ void __launch_bounds__(256) k_fast_cells(const uint8_t* __restrict__ slab, size_t slab_stride, const OrbCell* __restrict__ cells,
                                                    int ncells, int cell_cap, int iniTh, int minTh, OrbRawKp* __restrict__ out, int* __restrict__ counts) {
    __shared__ uint8_t win[ORB_WIN_MAX * ORB_WIN_MAX];
    __shared__ uint8_t sc[ORB_WIN_MAX * ORB_WIN_MAX];
    __shared__ int wave_cnt[4];
    __shared__ int total;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const OrbCell cell = cells[c];
    const uint8_t* L = slab + (size_t)b * slab_stride + cell.level_off;
    const int vw = cell.vw, vh = cell.vh, npx = vw * vh;
    for (int i = tid; i < npx; i += 256) { const int y = i / vw, x = i - y * vw; win[i] = L[(size_t)(cell.y0 + y) * cell.pitch + cell.x0 + x]; }
    __syncthreads();
    OrbRawKp* dst = out + ((size_t)b * ncells + c) * cell_cap;
    int emitted = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int th = pass == 0 ? iniTh : minTh;
        for (int i = tid; i < npx; i += 256) {
            const int y = i / vw, x = i - y * vw;
            int s = 0;
            if (x >= 3 && y >= 3 && x < vw - 3 && y < vh - 3) s = fast_score(win, vw, i, th);
            sc[i] = (uint8_t)s;
        }
        __syncthreads();
        int base = 0;
        for (int i0 = 0; i0 < npx; i0 += 256) {
            const int i = i0 + tid;
            bool keep = false; int s = 0, x = 0, y = 0;
            if (i < npx) {
                y = i / vw; x = i - y * vw; s = sc[i];
                if (s > 0) {       // non-zero score implies the 3-px margin, so all 8 neighbours are inside the window
                    keep = s > sc[i - 1] && s > sc[i + 1] && s > sc[i - vw - 1] && s > sc[i - vw] && s > sc[i - vw + 1] &&
                           s > sc[i + vw - 1] && s > sc[i + vw] && s > sc[i + vw + 1];
                }
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_cnt[wv] = __popcll(m);
            __syncthreads();
            int off = base; for (int k = 0; k < wv; k++) off += wave_cnt[k];
            const int rank = off + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && rank < cell_cap) { OrbRawKp k; k.x = (short)(x + cell.shift_x); k.y = (short)(y + cell.shift_y); k.score = (short)s; k.level = (short)cell.level; dst[rank] = k; }
            base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        if (tid == 0) total = base;
        __syncthreads();
        emitted = total;
        if (emitted > 0) break;
        __syncthreads();
    }
    if (tid == 0) counts[(size_t)b * ncells + c] = min(emitted, cell_cap) | (emitted > cell_cap ? 0x40000000 : 0);
}

// exclusive scan of the per-cell counts of one frame (ncells <= 4096) and compaction into a dense per-frame list
// ordered exactly like the reference's vToDistributeKeys (level, cell row, cell column, row-major inside the cell).
__global__ void __launch_bounds__(256) k_compact_cells(const OrbRawKp* __restrict__ raw, const int* __restrict__ counts, int ncells, int cell_cap,
                                                       OrbRawKp* __restrict__ dense, int cap, int* __restrict__ frame_total,
                                                       int* __restrict__ cell_offsets) {
    __shared__ int part[256];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncells; c0 += 256) {
        const int c = c0 + tid;
        const int n = c < ncells ? (counts[(size_t)b * ncells + c] & 0xffff) : 0;
        part[tid] = n; __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { int t = tid >= o ? part[tid - o] : 0; __syncthreads(); part[tid] += t; __syncthreads(); }
        const int excl = carry + part[tid] - n;
        if (c < ncells) {
            cell_offsets[(size_t)b * ncells + c] = excl;
            const OrbRawKp* src = raw + ((size_t)b * ncells + c) * cell_cap;
            for (int k = 0; k < n; k++) if (excl + k < cap) dense[(size_t)b * cap + excl + k] = src[k];
        }
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) frame_total[b] = carry;
}

// ---------------------------------------------------------------------------------------------------------
// IC_Angle: one wavefront per keypoint; lanes split the 31 rows of the radius-15 disc, integer moments,
// shuffle reduction, cv::fastAtan2 on lane 0.
__device__ __forceinline__ float d_fastAtan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y); float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else          { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
__global__ void __launch_bounds__(256) k_ic_angle(const uint8_t* __restrict__ slab, size_t slab_stride, const OrbLevel* __restrict__ levels,
                                                  const OrbSelKp* __restrict__ sel, const int* __restrict__ nsel, int cap, float* __restrict__ angle) {
    const int b = blockIdx.y, kp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (kp >= nsel[b]) return;
    const OrbSelKp k = sel[(size_t)b * cap + kp];
    const OrbLevel lv = levels[k.level];
    const int pitch = lv.w + 2 * ORB_PAD;
    const uint8_t* center = slab + (size_t)b * slab_stride + lv.off + (size_t)(d_cvRound(k.y) + ORB_PAD) * pitch + d_cvRound(k.x) + ORB_PAD;
    int m10 = 0, m01 = 0;
    // lanes 0..30 each take one row v = lane - 15; the remaining lanes idle
    if (lane < 31) {
        const int v = lane - 15, dmax = c_umax[v < 0 ? -v : v];
        int rs = 0;
        for (int u = -dmax; u <= dmax; u++) { const int val = center[v * pitch + u]; m10 += u * val; rs += val; }
        m01 = v * rs;
    }
    for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    if (lane == 0) angle[(size_t)b * cap + kp] = d_fastAtan2((float)m01, (float)m10);
}

// ---------------------------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma 2) on the level interior, BORDER_REFLECT_101 at the INTERIOR's edges (the reference
// blurs a clone of the ROI): horizontal pass 8.8 fixed point (u16), vertical pass 16.16, (v + 2^15) >> 16.
struct Taps7 { int k[7]; };
// A thread makes BH_COLS neighbouring outputs of a row from one window of BH_COLS + 6 bytes (2.5 loads per pixel instead of 7, a quarter of the waves) ...
#define BH_COLS 4
__global__ void k_blur7_h(const uint8_t* __restrict__ slab, size_t slab_stride, size_t off, int lw, int lh, Taps7 T,
                          uint16_t* __restrict__ tmp, size_t tmp_stride, size_t tmp_off) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * BH_COLS, y = blockIdx.y, b = blockIdx.z;
    if (x0 >= lw) return;
    const int pitch = lw + 2 * ORB_PAD;
    const uint8_t* R = slab + (size_t)b * slab_stride + off + (size_t)(y + ORB_PAD) * pitch + ORB_PAD;
    unsigned win[BH_COLS + 6];
    #pragma unroll
    for (int j = 0; j < BH_COLS + 6; j++) win[j] = (x0 - 3 + j <= lw + 2) ? (unsigned)R[d_reflect101(x0 - 3 + j, lw)] : 0u;      // (columns right of x = lw - 1 + 3 belong to no output)
    uint16_t* o = tmp + (size_t)b * tmp_stride + tmp_off + (size_t)y * lw + x0;
    #pragma unroll
    for (int q = 0; q < BH_COLS; q++) {
        if (x0 + q >= lw) break;
        unsigned s = 0;
        #pragma unroll
        for (int i = 0; i < 7; i++) s += (unsigned)T.k[i] * win[q + i];
        o[q] = (uint16_t)min(s, 65535u);
    }
}
// ... and BV_ROWS outputs of a column from one window of BV_ROWS + 6 rows (1.75 loads per pixel instead of 7, an eighth of the waves)
#define BV_ROWS 8
__global__ void k_blur7_v(const uint16_t* __restrict__ tmp, size_t tmp_stride, size_t tmp_off, int lw, int lh, Taps7 T,
                          uint8_t* __restrict__ blurred, size_t bl_stride, size_t bl_off) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * BV_ROWS, b = blockIdx.z;
    if (x >= lw) return;
    const uint16_t* P = tmp + (size_t)b * tmp_stride + tmp_off;
    unsigned win[BV_ROWS + 6];
    #pragma unroll
    for (int j = 0; j < BV_ROWS + 6; j++) win[j] = (y0 - 3 + j <= lh + 2) ? (unsigned)P[(size_t)d_reflect101(y0 - 3 + j, lh) * lw + x] : 0u;
    #pragma unroll
    for (int q = 0; q < BV_ROWS; q++) {
        const int y = y0 + q;
        if (y >= lh) break;
        unsigned s = 0;
        #pragma unroll
        for (int i = 0; i < 7; i++) s += (unsigned)T.k[i] * win[q + i];
        const unsigned r = (s + (1u << 15)) >> 16;
        blurred[(size_t)b * bl_stride + bl_off + (size_t)y * lw + x] = (uint8_t)min(r, 255u);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Steered BRIEF-256: 32 lanes per keypoint (one descriptor byte per lane), two keypoints per wavefront.
__global__ void __launch_bounds__(256) k_brief(const uint8_t* __restrict__ blurred, size_t bl_stride, const OrbLevel* __restrict__ levels,
                                               const OrbSelKp* __restrict__ sel, const int* __restrict__ nsel, int cap,
                                               const float* __restrict__ angle, uint8_t* __restrict__ desc) {
    const int b = blockIdx.y, kp = blockIdx.x * 8 + (threadIdx.x >> 5), byte = threadIdx.x & 31;
    if (kp >= nsel[b]) return;
    const OrbSelKp k = sel[(size_t)b * cap + kp];
    const OrbLevel lv = levels[k.level];
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = angle[(size_t)b * cap + kp] * factorPI;
    const float a = (float)cos((double)ang), bb = (float)sin((double)ang);
    const int step = lv.w;
    const uint8_t* center = blurred + (size_t)b * bl_stride + lv.blur_off + (size_t)d_cvRound(k.y) * step + d_cvRound(k.x);
    const signed char* pat = c_brief + byte * 32;
    int val = 0;
    #pragma unroll
    for (int j = 0; j < 8; j++) {
        const float x0 = (float)pat[4 * j], y0 = (float)pat[4 * j + 1], x1 = (float)pat[4 * j + 2], y1 = (float)pat[4 * j + 3];
        const int t0 = center[d_cvRound(x0 * bb + y0 * a) * step + d_cvRound(x0 * a - y0 * bb)];
        const int t1 = center[d_cvRound(x1 * bb + y1 * a) * step + d_cvRound(x1 * a - y1 * bb)];
        val |= (t0 < t1) << j;
    }
    desc[((size_t)b * cap + kp) * 32 + byte] = (uint8_t)val;
}

// ---------------------------------------------------------------------------------------------------------
// launchers
int orb_upload_constants(const int umax[16]) {
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_brief), SIND_BRIEF_PATTERN, 1024));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax, 16 * sizeof(int)));
    return SIND_OK;
}
int launch_pad(hipStream_t s, uint8_t* slab, size_t slab_stride, size_t off, int lw, int lh, int B) {
    hipLaunchKernelGGL(k_pad_reflect101, dim3(divup(2 * ORB_PAD * (lw + 2 * ORB_PAD) + 2 * ORB_PAD * lh, 256), 1, B), dim3(256), 0, s, slab, slab_stride, off, lw, lh, ORB_PAD);
    return SIND_OK;
}
int launch_copy_into_slab(hipStream_t s, const uint8_t* gray, uint8_t* slab, size_t slab_stride, size_t off, int w, int h, int B) {
    hipLaunchKernelGGL(k_copy_into_slab, dim3(divup(w, 128), h, B), dim3(128), 0, s, gray, slab, slab_stride, off, w, h, ORB_PAD);
    return SIND_OK;
}
int launch_fast_cells(hipStream_t s, const uint8_t* slab, size_t slab_stride, const OrbCell* cells, int ncells, int cell_cap, int iniTh, int minTh,
                      OrbRawKp* raw, int* counts, OrbRawKp* dense, int cap, int* frame_total, int* cell_offsets, int B) {
    hipLaunchKernelGGL(k_fast_cells, dim3(ncells, B), dim3(256), 0, s, slab, slab_stride, cells, ncells, cell_cap, iniTh, minTh, raw, counts);
    hipLaunchKernelGGL(k_compact_cells, dim3(B), dim3(256), 0, s, raw, counts, ncells, cell_cap, dense, cap, frame_total, cell_offsets);
    return SIND_OK;
}
int launch_ic_angle(hipStream_t s, const uint8_t* slab, size_t slab_stride, const OrbLevel* levels, const OrbSelKp* sel, const int* nsel,
                    int cap, int max_n, float* angle, int B) {
    if (max_n <= 0) return SIND_OK;
    hipLaunchKernelGGL(k_ic_angle, dim3(divup(max_n, 4), B), dim3(256), 0, s, slab, slab_stride, levels, sel, nsel, cap, angle);
    return SIND_OK;
}
int launch_blur7(hipStream_t s, const uint8_t* slab, size_t slab_stride, size_t off, int lw, int lh, const int taps[7], uint16_t* tmp,
                 size_t tmp_stride, size_t tmp_off, uint8_t* blurred, size_t bl_stride, size_t bl_off, int B) {
    Taps7 T; for (int i = 0; i < 7; i++) T.k[i] = taps[i];
    hipLaunchKernelGGL(k_blur7_h, dim3(divup(divup(lw, BH_COLS), 128), lh, B), dim3(128), 0, s, slab, slab_stride, off, lw, lh, T, tmp, tmp_stride, tmp_off);
    hipLaunchKernelGGL(k_blur7_v, dim3(divup(lw, 128), divup(lh, BV_ROWS), B), dim3(128), 0, s, tmp, tmp_stride, tmp_off, lw, lh, T, blurred, bl_stride, bl_off);
    return SIND_OK;
}
int launch_brief(hipStream_t s, const uint8_t* blurred, size_t bl_stride, const OrbLevel* levels, const OrbSelKp* sel, const int* nsel, int cap,
                 int max_n, const float* angle, uint8_t* desc, int B) {
    if (max_n <= 0) return SIND_OK;
    hipLaunchKernelGGL(k_brief, dim3(divup(max_n, 8), B), dim3(256), 0, s, blurred, bl_stride, levels, sel, nsel, cap, angle, desc);
    return SIND_OK;
}

}  // namespace sind
