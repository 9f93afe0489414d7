// Dense variational optical flow for DynaDetect on gfx950 (CDNA4): batched DeepFlow (variational part) and
// VariationalRefinement, i.e. the path reference DynaDetect.cc:1031-1147 spends ~99 % of its bytes in
// (SURVEY.md §8 a-3..a-7).  Every kernel is batched over B frame pairs: planes are laid out [B][h][w] in HBM,
// blockIdx.z (or .y for 1-D kernels) is the pair index, consecutive lanes walk consecutive x (coalesced rows).
//
// Arithmetic contract: built with -ffp-contract=off and IEEE divide/sqrt so that every FP32 expression below is
// evaluated exactly like OpenCV's scalar code paths (variational_refinement.cpp, deepflow.cpp, resize.cpp,
// imgwarp.cpp remap); the parity tests compare these kernels with the CPU oracle bit for bit.
//
// Roofline: all kernels here are HBM/L2-bound stencil passes (no MFMA: nothing is a contraction).
#include <mutex>
#include "common.hpp"
#include "flow.hpp"
#include "flow_dev.hpp"
#include <functional>
#include <array>
#include <map>

namespace sind {

// ---------------------------------------------------------------------------------------------------------
// u8 -> f32 (+ optional 3x3 Gaussian, BORDER_REFLECT_101): deepflow.cpp pre-smoothing with sigma = 0.6.
// Separable form kept (row pass value T, then column pass over T) so rounding matches sepFilter2D.
__global__ void k_u8_to_f32_blur3(const uint8_t* __restrict__ src, float* __restrict__ dst, int w, int h, float k0, float k1, int do_blur) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const uint8_t* S = src + (size_t)b * w * h;
    float* D = dst + (size_t)b * w * h;
    if (!do_blur) { D[y * w + x] = (float)S[y * w + x]; return; }
    const int xl = d_reflect101(x - 1, w), xr = d_reflect101(x + 1, w);
    const int yu = d_reflect101(y - 1, h), yd = d_reflect101(y + 1, h);
    auto rowv = [&](int yy) { const uint8_t* r = S + yy * w; return (float)r[x] * k0 + ((float)r[xl] + (float)r[xr]) * k1; };
    const float U = rowv(yu), C = rowv(y), L = rowv(yd);
    D[y * w + x] = C * k0 + (U + L) * k1;
}

// ---------------------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) on CV_32F (resize.cpp): used for the 0.95 pyramid, the flow up-sampling between
// levels (post = 1/0.95) and the final 384x288 -> 640x480 up-scale (post = 1/0.6).  scale_* are doubles computed
// on the host exactly as OpenCV does (1 / ((double)dsize/ssize)).
// Two batches with the same geometry in one launch (both images of the pairs for the pyramid, both flow components for the up-sampling): blockIdx.z =
// image + B * (0 | 1).
// A thread makes RZ_ROWS destination pixels of one column: the horizontal weights are formed once, and a launch has a quarter of the workgroups (one pixel per thread was
// bound by workgroup dispatch: 294 k workgroups of 128 threads for the top level of 170 pairs, 131 us for 0.3 GB).  The per-pixel expressions are resize_px's.
#define RZ_ROWS 4
__global__ void k_resize_f32_pair(const float* __restrict__ srcA, float* __restrict__ dstA, const float* __restrict__ srcB, float* __restrict__ dstB, int B,
                                  int sw, int sh, int dw, int dh, double scale_x, double scale_y, float post, int has_post) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy0 = blockIdx.y * RZ_ROWS; int b = blockIdx.z;
    if (dx >= dw) return;
    const float* src = srcA; float* dst = dstA;
    if (b >= B) { b -= B; src = srcB; dst = dstB; }
    const float* S = src + (size_t)b * sw * sh; float* D = dst + (size_t)b * dw * dh;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = d_cvFloorf(fx); fx -= sx;
    const bool two = sx + 1 < sw;            // dx < xmax in OpenCV's HResizeLinear
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const float a1 = fx, a0 = 1.f - a1;
    #pragma unroll
    for (int r = 0; r < RZ_ROWS; r++) {
        const int dy = dy0 + r;
        if (dy >= dh) break;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = d_cvFloorf(fy); fy -= sy;
        const int y0 = d_clip(sy, 0, sh), y1 = d_clip(sy + 1, 0, sh);
        const float b1 = fy, b0 = 1.f - b1;
        const float* R0 = S + (size_t)y0 * sw; const float* R1 = S + (size_t)y1 * sw;
        float r0, r1;
        if (two) { r0 = R0[sx] * a0 + R0[sx + 1] * a1; r1 = R1[sx] * a0 + R1[sx + 1] * a1; }
        else     { r0 = R0[sx] * 1.f;                  r1 = R1[sx] * 1.f; }
        float v = r0 * b0 + r1 * b1;
        if (has_post) v = v * post;
        D[(size_t)dy * dw + dx] = v;
    }
}
// The small levels of the 0.95 pyramid in ONE launch: a workgroup walks levels first+1 .. first+n of one image (level l from level l-1, a barrier in
// between; the image is <= 12 k pixels there) instead of one launch per level and image -- the chain of ~2 x 27 dependent launches per slice sat at the
// start of every step, when nothing else of the step can run yet.
#define PYR_TAIL_MAX 40
struct PyrTail { int n, B; int w[PYR_TAIL_MAX + 1], h[PYR_TAIL_MAX + 1]; unsigned long long off[PYR_TAIL_MAX + 1]; double sx[PYR_TAIL_MAX + 1], sy[PYR_TAIL_MAX + 1]; };
__global__ void __launch_bounds__(256) k_pyramid_tail(float* __restrict__ pyrA, float* __restrict__ pyrB, PyrTail T) {
    int b = blockIdx.x; float* pyr = pyrA;
    if (b >= T.B) { b -= T.B; pyr = pyrB; }
    for (int k = 1; k <= T.n; k++) {
        const int sw = T.w[k - 1], sh = T.h[k - 1], dw = T.w[k], dh = T.h[k];
        const float* S = pyr + T.off[k - 1] + (size_t)b * sw * sh; float* D = pyr + T.off[k] + (size_t)b * dw * dh;
        for (int i = threadIdx.x; i < dw * dh; i += 256) { const int dy = i / dw, dx = i - dy * dw; D[i] = resize_px(S, sw, sh, dx, dy, T.sx[k], T.sy[k]); }
        __syncthreads();                     // level k complete (and visible to the workgroup) before level k + 1 reads it
    }
}
__global__ void k_resize_f32(const float* __restrict__ src, float* __restrict__ dst, int sw, int sh, int dw, int dh,
                             double scale_x, double scale_y, float post, int has_post) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, b = blockIdx.z;
    if (dx >= dw) return;
    const float* S = src + (size_t)b * sw * sh;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = d_cvFloorf(fx); fx -= sx;
    const bool two = sx + 1 < sw;            // dx < xmax in OpenCV's HResizeLinear
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = d_cvFloorf(fy); fy -= sy;
    const int y0 = d_clip(sy, 0, sh), y1 = d_clip(sy + 1, 0, sh);
    const float a1 = fx, a0 = 1.f - a1, b1 = fy, b0 = 1.f - b1;
    const float* R0 = S + (size_t)y0 * sw; const float* R1 = S + (size_t)y1 * sw;
    float r0, r1;
    if (two) { r0 = R0[sx] * a0 + R0[sx + 1] * a1; r1 = R1[sx] * a0 + R1[sx + 1] * a1; }
    else     { r0 = R0[sx] * 1.f;                  r1 = R1[sx] * 1.f; }
    float v = r0 * b0 + r1 * b1;
    if (has_post) v = v * post;
    dst[(size_t)b * dw * dh + (size_t)dy * dw + dx] = v;
}

// ---------------------------------------------------------------------------------------------------------
// VariationalRefinementImpl::prepareBuffers, part 1: warp I1 by the level's initial flow (cv::remap, INTER_LINEAR,
// BORDER_REPLICATE, coordinates quantised to 1/32 px), averaged image and temporal difference.
#define WARP_ROWS 4                                             // rows of its column a thread makes (a quarter of the waves; the per-pixel expressions are unchanged)
__global__ void k_warp_avg_iz(const float* __restrict__ I0, const float* __restrict__ I1, const float* __restrict__ Wu,
                              const float* __restrict__ Wv, float* __restrict__ avg, float* __restrict__ Iz, float* __restrict__ dWu,
                              float* __restrict__ dWv, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yb = blockIdx.y * WARP_ROWS, b = blockIdx.z;
    if (x >= w) return;
    const size_t base = (size_t)b * w * h;
    const float* S = I1 + base;
    #pragma unroll
    for (int r = 0; r < WARP_ROWS; r++) {
        const int y = yb + r;
        if (y >= h) break;
        const int i = y * w + x;
        dWu[base + i] = 0.f; dWv[base + i] = 0.f;           // the level's flow increment starts at zero (two fills less per level)
        warp_px(S, I0[base + i], Wu[base + i], Wv[base + i], x, y, w, h, avg[base + i], Iz[base + i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// One fixed-point iteration set-up: ComputeDataTerm + ComputeSmoothnessTerm{Hor,Vert}Pass gathered per pixel.
// The four smoothness contributions are added in the order OpenCV's red/black passes produce for the pixel's colour.
// A thread walks KC_ROWS consecutive rows of its column: the smoothness weight of a pixel is also its right neighbour's "left" weight and its lower
// neighbour's "upper" weight, so the left one comes from the neighbouring lane (wave shuffle; the first lane of a wave computes it) and the upper one
// from the thread's previous row (the first row of a block computes it) -- 1.3 instead of 3 weight evaluations (two loads rows, a square root and a
// division each) per pixel, same values bit for bit.
#define KC_ROWS 4
// IN: every pixel this wave touches in these rows -- the four rows of its 64 columns and their stencils (two columns / rows to every side) -- lies inside the image: no index is
// clamped and no border case exists, so every neighbour is a load at a CONSTANT offset from one of a few row pointers.  (The clamped form spends ~3 integer VALU instructions on
// each of its 49 loads per pixel: ~150 of the kernel's 376 VALU instructions per pixel, disassembly of round 4; the interior is > 95 % of a 384 x 288 level.)  Same loads of the
// same values, same arithmetic: bit-identical by construction (tests/test_flow_gpu.py).
template <bool IN>
__device__ __forceinline__ void kc_rows(const VarParams& P, int w, int h, int x, int y0, size_t base, const float* __restrict__ gAvg, const float* __restrict__ gIz, const float* __restrict__ gWu,
                       const float* __restrict__ gWv,
                       const float* __restrict__ gdWu, const float* __restrict__ gdWv, float* __restrict__ A11, float* __restrict__ A12,
                       float* __restrict__ A22, float* __restrict__ B1, float* __restrict__ B2, float* __restrict__ Wgt, float* __restrict__ R11,
                       float* __restrict__ R22) {
    const float zeta2 = P.zeta * P.zeta, eps2 = P.epsilon * P.epsilon, gamma2 = P.gamma / 2, delta2 = P.delta / 2, alpha2 = P.alpha / 2;
    // The seven Sobel(ksize = 1, BORDER_REPLICATE) derivative images of prepareBuffers are formed here from the warped average and
    // the temporal difference (the float operations k_derivs used to store, evaluated at the same replicated positions): two planes
    // are read through the cache instead of eight from memory, in each of the five fixed-point iterations of a level.
    const float* A = gAvg + base; const float* Z = gIz + base;
    auto cx = [&](int v) { return IN ? v : min(max(v, 0), w - 1); };
    auto cy = [&](int v) { return IN ? v : min(max(v, 0), h - 1); };
    auto dX = [&](const float* Pp, int yy, int xx) { return Pp[yy * w + cx(xx + 1)] - Pp[yy * w + cx(xx - 1)]; };
    auto dY = [&](const float* Pp, int yy, int xx) { return Pp[cy(yy + 1) * w + xx] - Pp[cy(yy - 1) * w + xx]; };
    // tempW = W + dW is formed on the fly (OpenCV keeps it in a buffer that it refreshes after every fixed-point iteration with this
    // very addition): two planes less to read here and no k_add_flow pass between the iterations.  Before the first iteration
    // dW = 0, and W + 0 differs from W only in the sign of a zero, which the squared differences below cannot see.
    const float* WU = gWu + base; const float* WV = gWv + base; const float* DU = gdWu + base; const float* DV = gdWv + base;
    auto wgt_at = [&](int yy, int xx) {
        const int xn = IN ? xx + 1 : min(xx + 1, w - 1), yn = IN ? yy + 1 : min(yy + 1, h - 1);
        const int ic = yy * w + xx, ix = yy * w + xn, iy = yn * w + xx;
        const float c_u = WU[ic] + DU[ic], c_v = WV[ic] + DV[ic];
        const float ux = (WU[ix] + DU[ix]) - c_u, vx = (WV[ix] + DV[ix]) - c_v;
        const float uy = (WU[iy] + DU[iy]) - c_u, vy = (WV[iy] + DV[iy]) - c_v;
        return alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
    };
    const bool wave_first = (threadIdx.x & 63) == 0;
    float w_up = 0.f;                                       // weight of (y - 1, x): computed for the block's first row, carried down afterwards
    #pragma unroll
    for (int r = 0; r < KC_ROWS; r++) {
        const int y = y0 + r;
        if (!IN && y >= h) break;                           // uniform over the block
        const int i = y * w + x;
        const float Ix = dX(A, y, x), Iy = dY(A, y, x), Iz = Z[i], Ixz = dX(Z, y, x), Iyz = dY(Z, y, x);
        const float Ixx = dX(A, y, cx(x + 1)) - dX(A, y, cx(x - 1));
        const float Ixy = dX(A, cy(y + 1), x) - dX(A, cy(y - 1), x);
        const float Iyy = dY(A, cy(y + 1), x) - dY(A, cy(y - 1), x);
        const float dU = gdWu[base + i], dV = gdWv[base + i];
        float derivNorm = Ix * Ix + Iy * Iy + zeta2;
        const float Ik1z = Iz + Ix * dU + Iy * dV;
        const float rN0 = kc_rcp(derivNorm);
        float weight = kc_div(delta2 / sqrtf(kc_div(Ik1z * Ik1z, derivNorm, rN0) + eps2), derivNorm, rN0);
        float a11 = weight * (Ix * Ix) + zeta2;
        float a12 = weight * (Ix * Iy);
        float a22 = weight * (Iy * Iy) + zeta2;
        float b1 = -weight * (Iz * Ix);
        float b2 = -weight * (Iz * Iy);
        derivNorm = Ixx * Ixx + Ixy * Ixy + zeta2;
        const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + zeta2;
        const float Ik1zx = Ixz + Ixx * dU + Ixy * dV;
        const float Ik1zy = Iyz + Ixy * dU + Iyy * dV;
        const float rN1 = kc_rcp(derivNorm), rN2 = kc_rcp(derivNorm2);
        #define D1(n) kc_div((n), derivNorm, rN1)
        #define D2(n) kc_div((n), derivNorm2, rN2)
        weight = gamma2 / sqrtf(D1(Ik1zx * Ik1zx) + D2(Ik1zy * Ik1zy) + eps2);
        a11 += weight * (D1(Ixx * Ixx) + D2(Ixy * Ixy));
        a12 += weight * (D1(Ixx * Ixy) + D2(Ixy * Iyy));
        a22 += weight * (D1(Ixy * Ixy) + D2(Iyy * Iyy));
        b1 += -weight * (D1(Ixx * Ixz) + D2(Ixy * Iyz));
        b2 += -weight * (D1(Ixy * Ixz) + D2(Iyy * Iyz));
        #undef D1
        #undef D2

        const float wp = wgt_at(y, x);
        const float w_from_lane = __shfl_up(wp, 1);          // the left neighbour's own weight (same row, lane - 1)
        const float wl = (IN || x > 0) ? (wave_first ? wgt_at(y, x - 1) : w_from_lane) : 0.f;
        const float wq = (IN || y > 0) ? (r == 0 ? wgt_at(y - 1, x) : w_up) : 0.f;
        const float wu_c = WU[i], wv_c = WV[i];
        const bool red = ((x + y) & 1) == 0;
        // the four link updates (no-ops at the image border)
        #define OWN_H() if (IN || x < w - 1) { b1 += wp * (WU[i + 1] - wu_c); a11 += wp; b2 += wp * (WV[i + 1] - wv_c); a22 += wp; }
        #define LEFT_H() if (IN || x > 0) { b1 -= wl * (wu_c - WU[i - 1]); a11 += wl; b2 -= wl * (wv_c - WV[i - 1]); a22 += wl; }
        #define OWN_V() if (IN || y < h - 1) { b1 += wp * (WU[i + w] - wu_c); a11 += wp; b2 += wp * (WV[i + w] - wv_c); a22 += wp; }
        #define UP_V() if (IN || y > 0) { b1 -= wq * (wu_c - WU[i - w]); a11 += wq; b2 -= wq * (wv_c - WV[i - w]); a22 += wq; }
        if (red) { OWN_H() LEFT_H() OWN_V() UP_V() }
        else     { LEFT_H() OWN_H() UP_V() OWN_V() }
        #undef OWN_H
        #undef LEFT_H
        #undef OWN_V
        #undef UP_V
        A11[base + i] = a11; A12[base + i] = a12; A22[base + i] = a22; B1[base + i] = b1; B2[base + i] = b2; Wgt[base + i] = wp;
        if (R11) { R11[base + i] = 1.f / a11; R22[base + i] = 1.f / a22; }     // solver mode 3 only: correctly rounded (Markstein's division needs exactly RN(1 / a))
        w_up = wp;
    }
}
__global__ void k_coef(VarParams P, int w, int h, const float* __restrict__ gAvg, const float* __restrict__ gIz, const float* __restrict__ gWu,
                       const float* __restrict__ gWv,
                       const float* __restrict__ gdWu, const float* __restrict__ gdWv, float* __restrict__ A11, float* __restrict__ A12,
                       float* __restrict__ A22, float* __restrict__ B1, float* __restrict__ B2, float* __restrict__ Wgt, float* __restrict__ R11,
                       float* __restrict__ R22) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z, y0 = blockIdx.y * KC_ROWS;
    if (x >= w) return;
    const size_t base = (size_t)b * w * h;
    // interior: rows y0 - 2 .. y0 + KC_ROWS + 1 and, for every lane of the wave, columns x - 2 .. x + 2 exist (wave-uniform: a ballot over the wave's live lanes)
    const bool in = y0 >= 2 && y0 + KC_ROWS + 1 < h && __builtin_amdgcn_ballot_w64(x < 2 || x + 2 >= w) == 0ull;
    if (in) kc_rows<true>(P, w, h, x, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22);
    else kc_rows<false>(P, w, h, x, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22);
}
// grid (tiles of KL_COLS columns, groups of 4 x KL_ROWS rows, pairs), 256 threads: wave k of a workgroup takes the rows (4 * blockIdx.y + k) * KL_ROWS ...
__global__ __launch_bounds__(256) void k_coef_lanes(VarParams P, int w, int h, const float* __restrict__ gAvg, const float* __restrict__ gIz, const float* __restrict__ gWu,
                       const float* __restrict__ gWv, const float* __restrict__ gdWu, const float* __restrict__ gdWv, float* __restrict__ A11, float* __restrict__ A12,
                       float* __restrict__ A22, float* __restrict__ B1, float* __restrict__ B2, float* __restrict__ Wgt, float* __restrict__ R11, float* __restrict__ R22, int fast_math,
                       int tiles_x, int tiles_y, int n_tiles, int xcd) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // 1-D grid (a multiple of 8): tile = x + tiles_x * (y + tiles_y * pair).  xcd: consecutive tiles -- the tiles of one pair, which share their halo rows and columns -- go to the
    // SAME XCD (workgroups are handed to the eight XCDs round-robin), so the second reader of a halo line finds it in that XCD's L2
    const int tile = xcd ? ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile >= n_tiles) return;
    const int bx = tile % tiles_x, byz = tile / tiles_x, by = byz % tiles_y, bz = byz / tiles_y;
    const int c0 = bx * KL_COLS - 2, y0 = (by * 4 + wave) * KL_ROWS;
    if (y0 >= h) return;                                    // whole wave
    const size_t base = (size_t)bz * w * h;
    const int col = c0 + lane;
    const bool store_lane = lane >= 2 && lane < 2 + KL_COLS && col < w;
    // the short forms of sqrt and c / sqrt need arguments >= 2^-96: every argument carries epsilon^2 (1e-6 with the reference's parameters); any other epsilon takes the IEEE forms
    const bool interior = c0 >= 0 && c0 + 63 < w && y0 >= 2 && y0 + KL_ROWS + 1 < h, fastm = P.epsilon >= 1e-12f && fast_math != 0;
    if (fastm) {
        if (interior) kc_lanes<false, true>(P, w, h, col, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22, store_lane);
        else kc_lanes<true, true>(P, w, h, col, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22, store_lane);
    } else {
        if (interior) kc_lanes<false, false>(P, w, h, col, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22, store_lane);
        else kc_lanes<true, false>(P, w, h, col, y0, base, gAvg, gIz, gWu, gWv, gdWu, gdWv, A11, A12, A22, B1, B2, Wgt, R11, R22, store_lane);
    }
}

// ---------------------------------------------------------------------------------------------------------
// RedBlackSOR_ParBody: one colour of one SOR iteration (plain version: one thread per pixel of the colour).
// Out-of-image neighbours contribute exactly 0 (zero weight / zero increment in OpenCV's buffer borders).
__global__ void k_sor_color(int w, int h, float omega, int color, const float* __restrict__ A11, const float* __restrict__ A12,
                            const float* __restrict__ A22, const float* __restrict__ B1, const float* __restrict__ B2,
                            const float* __restrict__ Wgt, float* __restrict__ dWu, float* __restrict__ dWv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int x = 2 * k + ((y + color) & 1);
    if (x >= w) return;
    const size_t base = (size_t)b * w * h; const int i = y * w + x;
    const float* Wg = Wgt + base; float* U = dWu + base; float* V = dWv + base;
    const float wp = Wg[i];
    const float wl = x > 0 ? Wg[i - 1] : 0.f, wu = y > 0 ? Wg[i - w] : 0.f;
    const float ul = x > 0 ? U[i - 1] : 0.f, vl = x > 0 ? V[i - 1] : 0.f;
    const float ur = x < w - 1 ? U[i + 1] : 0.f, vr = x < w - 1 ? V[i + 1] : 0.f;
    const float uu = y > 0 ? U[i - w] : 0.f, vu = y > 0 ? V[i - w] : 0.f;
    const float ud = y < h - 1 ? U[i + w] : 0.f, vd = y < h - 1 ? V[i + w] : 0.f;
    const float sigmaU = wl * ul + wp * ur + wu * uu + wp * ud;
    const float sigmaV = wl * vl + wp * vr + wu * vu + wp * vd;
    float du = U[i], dv = V[i];
    const float a12 = A12[base + i];
    du += omega * ((sigmaU + B1[base + i] - dv * a12) / A11[base + i] - du);
    dv += omega * ((sigmaV + B2[base + i] - du * a12) / A22[base + i] - dv);
    U[i] = du; V[i] = dv;
}

// ---------------------------------------------------------------------------------------------------------
// Fused red-black SOR: `iters` full iterations per launch with the linear system held in REGISTERS.
//   * a workgroup owns an extended tile E (EW x EH pixels); every thread owns a 1x8 pixel strip of one row and keeps
//     its 8 x (A11, A12, A22, b1, b2, w, w_up) + w_left in VGPRs for the whole launch (the CU's 512 KB register file is
//     the largest on-chip store; LDS only carries the flow increments);
//   * the increments (du, dv) live in LDS in a checkerboard-split layout (plane[parity][component][row][x/2]) so that the
//     four up / down neighbours of a thread's four same-colour pixels are ONE ds_read_b128 per plane; horizontal
//     neighbours are the thread's own other-colour registers plus one LDS word at the strip edge;
//   * waves are row-parity uniform (first half of the block = even rows, second half = odd rows), so the choice of which
//     four strip pixels carry the active colour is wave-uniform: no divergence;
//   * tiles overlap by a halo of 2*iters pixels: stale values creep inwards one pixel per half-sweep from the tile edge, so
//     the interior (written back) is exact; at image borders E is clipped and the boundary condition is exact.
// Arithmetic per pixel is identical to k_sor_color / OpenCV RedBlackSOR_ParBody, so results stay bit-exact.
#define SOR_PX 8
#define SOR_NT 1024
// Row stride of the LDS planes in float4: strips + a guard on each side, padded to 4 (mod 8).  A wave's ds_read_b128 is served in four 16-lane groups
// that each touch four half rows (16 banks) of four different rows two apart; with a stride of 4 (mod 8) float4 the four pieces fall into the four
// different quarters of the 64 banks (stride 10, the unpadded 64-pixel tile, put two of them on the same quarter: 40 % of the LDS cycles were conflicts).
__host__ __device__ constexpr int sor_row_stride(int EW) { return (EW / 8 + 2) + ((4 - (EW / 8 + 2)) % 8 + 8) % 8; }
__device__ __forceinline__ void ld8(const float* __restrict__ p, float (&d)[SOR_PX]) {
    const F4u a = *reinterpret_cast<const F4u*>(p), c = *reinterpret_cast<const F4u*>(p + 4);
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = c.x; d[5] = c.y; d[6] = c.z; d[7] = c.w;
}
__global__ void k_debug_rcp_scan(int exp_lo, int exp_hi, unsigned long long* __restrict__ out) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;           // all 2^23 significands
    if (m >= (1u << 23)) return;
    unsigned long long bad_r = 0, bad_q = 0; unsigned rng = m * 2654435761u + 12345u;
    for (int e = exp_lo; e <= exp_hi; e++) {
        const float a = __uint_as_float(((unsigned)(127 + e) << 23) | m);
        const float y = sor_rcp(a), ref = 1.f / a;
        if (__float_as_uint(y) != __float_as_uint(ref)) bad_r++;
        for (int k = 0; k < 8; k++) {                                   // quotients through the reciprocal (Markstein) against the IEEE division
            rng = rng * 1664525u + 1013904223u;
            const float n = __uint_as_float((rng & 0x807fffffu) | ((unsigned)(127 + e - 40 + (int)(((rng >> 23) & 127) % 81)) << 23));
            const float q = sor_div(n, a, y), qr = n / a;
            if (__float_as_uint(q) != __float_as_uint(qr)) bad_q++;
            const float q2 = sor_div(n, -a, -y), qr2 = n / -a;
            if (__float_as_uint(q2) != __float_as_uint(qr2)) bad_q++;
        }
    }
    if (bad_r) atomicAdd(&out[0], bad_r);
    if (bad_q) atomicAdd(&out[1], bad_q);
    if (bad_r) atomicMin(&out[2], (unsigned long long)m);
}
// kc_sqrt against sqrtf, and c / b through kc_rcp + kc_div against the IEEE division for three numerators, for every significand and the binary exponents exp_lo .. exp_hi
__global__ void k_debug_coef_math_scan(int exp_lo, int exp_hi, float n0, float n1, float n2, unsigned long long* __restrict__ out) {
    const unsigned m = blockIdx.x * blockDim.x + threadIdx.x;           // all 2^23 significands
    if (m >= (1u << 23)) return;
    unsigned long long bad_s = 0, bad_q = 0;
    for (int e = exp_lo; e <= exp_hi; e++) {
        const float x = __uint_as_float(((unsigned)(127 + e) << 23) | m);
        if (__float_as_uint(kc_sqrt(x)) != __float_as_uint(sqrtf(x))) bad_s++;
        const float r = kc_rcp(x);
        if (__float_as_uint(kc_div(n0, x, r)) != __float_as_uint(n0 / x)) bad_q++;
        if (__float_as_uint(kc_div(n1, x, r)) != __float_as_uint(n1 / x)) bad_q++;
        if (__float_as_uint(kc_div(n2, x, r)) != __float_as_uint(n2 / x)) bad_q++;
    }
    if (bad_s) atomicAdd(&out[0], bad_s);
    if (bad_q) atomicAdd(&out[1], bad_q);
}
int debug_coef_math_scan(hipStream_t s, int exp_lo, int exp_hi, const float numer[3], unsigned long long* out_dev) {
    hipLaunchKernelGGL(k_debug_coef_math_scan, dim3((1u << 23) / 256), dim3(256), 0, s, exp_lo, exp_hi, numer[0], numer[1], numer[2], out_dev);
    HIP_TRY(hipGetLastError()); return SIND_OK;
}
int debug_rcp_scan(hipStream_t s, int exp_lo, int exp_hi, unsigned long long* out_dev) {
    hipLaunchKernelGGL(k_debug_rcp_scan, dim3((1u << 23) / 256), dim3(256), 0, s, exp_lo, exp_hi, out_dev);
    HIP_TRY(hipGetLastError()); return SIND_OK;
}
// RCP: the two IEEE divisions of a pixel update (ten VALU instructions each, one of them quarter rate) become Markstein's three-operation
// sequence on the reciprocals of A11 / A22, formed once per launch and held in registers (see sor_div / k_sor_fused4 below); a pixel outside the
// image gets the reciprocal 0, which makes its update return exactly 0 and removes the per-pixel validity select from the loop.
// The ten persistent values per pixel (80 registers per strip) do not fit the 128-register budget of four waves per SIMD, so the RCP instance runs
// three waves per SIMD (<= 168 registers): MAXT threads per workgroup, WPE waves per SIMD.
// CEW x CNR: extended tile known at compile time (0 = run-time sizes, the one-workgroup levels): every LDS address then is one base register plus an
// immediate offset instead of a dozen address registers held across the loop.
template <int DIV, int MAXT, int WPE, int CEW, int CNR>        // DIV: 0 = IEEE division, 1 = reciprocal planes in registers (RCP), 2 = reciprocal formed on the fly (sor_rcp)
__global__ void __attribute__((amdgpu_flat_work_group_size(64, MAXT), amdgpu_waves_per_eu(WPE, WPE)))
k_sor_fused(int w, int h, int EW, int EH, int IW, int IH, int halo_x, int halo, int ntx, int iters, int xcd_remap, float omega,
            const float* __restrict__ gA11, const float* __restrict__ gA12, const float* __restrict__ gA22, const float* __restrict__ gB1,
            const float* __restrict__ gB2, const float* __restrict__ gW, const float* __restrict__ gR11, const float* __restrict__ gR22,
            const float* __restrict__ gUin, const float* __restrict__ gVin, float* __restrict__ gUout, float* __restrict__ gVout) {
    constexpr bool RCP = DIV == 1;
    extern __shared__ float4 lds4[];             // float4-indexed so that every strip access is one ds_read/write_b128
    float* lds = reinterpret_cast<float*>(lds4);
    const int SW = (CEW ? CEW : EW) / SOR_PX;    // strips per row
    const int RS4 = sor_row_stride(CEW ? CEW : EW);      // LDS row stride in float4 (one guard float4 on each side + bank padding)
    const int half = CEW ? CEW * CNR / (2 * SOR_PX) : (int)(blockDim.x >> 1);
    const int NR = CEW ? CNR : 2 * (half / SW);  // rows covered by the thread block (>= EH; surplus rows stay zero)
    const int PL4 = (NR + 2) * RS4;              // one plane (guard row above and below)
    const int tid = threadIdx.x;
    const int idx = tid < half ? tid : tid - half;
    const int j = idx % SW, ly = 2 * (idx / SW) + (tid < half ? 0 : 1);
    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (each with its own L2) in linear-id order, so neighbouring ids never
    // share an L2.  Re-map id -> (image, tile) such that XCD q walks the q-th eighth of all tiles in order: tiles that overlap in their
    // halos then run on the same XCD at about the same time and the halo is fetched from HBM once.
    int tile = blockIdx.x, b = blockIdx.y;
    {
        const long long ntile = gridDim.x, lin = blockIdx.x + (long long)blockIdx.y * ntile, per = ntile * gridDim.y / 8;
        if (xcd_remap && lin < per * 8) { const long long logical = (lin & 7) * per + (lin >> 3); b = (int)(logical / ntile); tile = (int)(logical - (long long)b * ntile); }
    }
    const int tx = tile % ntx, ty = tile / ntx;
    const int ex0 = tx * IW - halo_x, ey0 = ty * IH - halo;         // E origin in image coordinates (may be negative)
    const int gy = ey0 + ly, gx0 = ex0 + SOR_PX * j;
    const size_t base = (size_t)b * w * h;
    const int off = (ex0 + ey0) & 1;             // local parity of the globally "red" pixels

    float a11[SOR_PX], a12[SOR_PX], a22[SOR_PX], b1[SOR_PX], b2[SOR_PX], wp[SOR_PX], du[SOR_PX], dv[SOR_PX];
    float wtop[SOR_PX];                          // weights of the image row above the tile, loaded by the tile's first row only
    float r11[RCP ? SOR_PX : 1], r22[RCP ? SOR_PX : 1];
    float wl0 = 0.f;
    unsigned valid = 0;      // a pixel outside the image keeps the reciprocal 0 (RCP): its update returns exactly 0, with no select in the loop
    const bool row_ok = ly < EH && gy >= 0 && gy < h;
    #pragma unroll
    for (int i = 0; i < SOR_PX; i++) { a11[i] = 1.f; a12[i] = 0.f; a22[i] = 1.f; b1[i] = 0.f; b2[i] = 0.f; wp[i] = 0.f; wtop[i] = 0.f; du[i] = 0.f; dv[i] = 0.f; }
    if constexpr (RCP) {
        #pragma unroll
        for (int i = 0; i < SOR_PX; i++) { r11[i] = 0.f; r22[i] = 0.f; }
    }
    if (row_ok && gx0 >= 0 && gx0 + SOR_PX <= w) {        // whole strip inside the image: 16-byte loads
        const size_t g = base + (size_t)gy * w + gx0;
        valid = 0xffu;
        ld8(gA11 + g, a11); ld8(gA12 + g, a12); ld8(gA22 + g, a22); ld8(gB1 + g, b1); ld8(gB2 + g, b2); ld8(gW + g, wp); ld8(gUin + g, du); ld8(gVin + g, dv);
        if constexpr (RCP) { ld8(gR11 + g, r11); ld8(gR22 + g, r22); }
        if (ly == 0 && gy > 0) ld8(gW + g - w, wtop);
    } else if (row_ok) {
        #pragma unroll
        for (int i = 0; i < SOR_PX; i++) {
            const int gx = gx0 + i;
            if (gx >= 0 && gx < w) {
                const size_t g = base + (size_t)gy * w + gx;
                valid |= 1u << i;
                a11[i] = gA11[g]; a12[i] = gA12[g]; a22[i] = gA22[g]; b1[i] = gB1[g]; b2[i] = gB2[g]; wp[i] = gW[g];
                if (ly == 0 && gy > 0) wtop[i] = gW[g - w];
                du[i] = gUin[g]; dv[i] = gVin[g];
                if constexpr (RCP) { r11[i] = gR11[g]; r22[i] = gR22[g]; }
            }
        }
    }
    // zero the guard ring of the six planes (top and bottom row, the cells left and right of the strips; every other cell is written below by the
    // thread that owns it) while the loads above are in flight
    {
        const int GC = RS4 - SW, G = 2 * RS4 + NR * GC;          // guard cells per row / per plane
        for (int i = tid; i < 6 * G; i += blockDim.x) {
            const int pl = i / G, c = i - pl * G;
            int cell;
            if (c < 2 * RS4) cell = c < RS4 ? c : (NR + 1) * RS4 + (c - RS4);
            else { const int r = (c - 2 * RS4) / GC, g = (c - 2 * RS4) - r * GC; cell = (r + 1) * RS4 + (g == 0 ? 0 : SW + g); }
            lds4[pl * PL4 + cell] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    // LDS addressing (float4 units): plane(parity q, component c) at (q*2 + c)*PL4; strip j of row ly at (ly+1)*RS4 + 1 + j
    const int ro4 = (ly + 1) * RS4 + 1 + j;
    const int s0 = ly & 1;                       // strip pixels i with ((i + ly) & 1) == q have local parity q
    {   // initial state of both parities: strip pixels s0, s0+2, ... are local-even, the others local-odd
        const float4 au = make_float4(du[0], du[2], du[4], du[6]), bu = make_float4(du[1], du[3], du[5], du[7]);
        const float4 av = make_float4(dv[0], dv[2], dv[4], dv[6]), bv = make_float4(dv[1], dv[3], dv[5], dv[7]);
        const int pe = s0 == 0 ? 0 : 2, po = s0 == 0 ? 2 : 0;      // plane pair receiving the i-even / i-odd pixels
        lds4[(pe + 0) * PL4 + ro4] = au; lds4[(pe + 1) * PL4 + ro4] = av;
        lds4[(po + 0) * PL4 + ro4] = bu; lds4[(po + 1) * PL4 + ro4] = bv;
        // smoothness weights, same checkerboard split (plane 4 + parity): a pixel's upper weight is read back from the row above, so the
        // w_up plane is neither loaded from memory (one ninth of the tile's load requests) nor held in registers
        lds4[(4 + (pe >> 1)) * PL4 + ro4] = make_float4(wp[0], wp[2], wp[4], wp[6]);
        lds4[(4 + (po >> 1)) * PL4 + ro4] = make_float4(wp[1], wp[3], wp[5], wp[7]);
        if (ly == 0) {                           // guard row = image row above the tile: pixel parities are those of local row -1
            lds4[(4 + (po >> 1)) * PL4 + ro4 - RS4] = make_float4(wtop[0], wtop[2], wtop[4], wtop[6]);
            lds4[(4 + (pe >> 1)) * PL4 + ro4 - RS4] = make_float4(wtop[1], wtop[3], wtop[5], wtop[7]);
        }
    }
    __syncthreads();
    // weight of the pixel left of the strip = last pixel of the neighbouring strip (LDS); at the tile's left edge it reads the zero
    // guard, which only touches the outermost halo ring (never part of the written interior) or lies outside the image (weight 0)
    wl0 = lds[4 * ((4 + (s0 ^ 1)) * PL4 + ro4) - 1];

    // one half-sweep over the strip pixels START, START+2, START+4, START+6 (compile-time START keeps register indices static);
    // Q = local parity being updated.  Invalid pixels (outside the image / surplus rows) keep du = dv = 0.
    #define SOR_HALF(START, Q)                                                                                                    \
        {                                                                                                                         \
            const int oq = (Q) ^ 1;                                                                                               \
            const float4 t0 = lds4[(oq * 2 + 0) * PL4 + ro4 - RS4], t1 = lds4[(oq * 2 + 1) * PL4 + ro4 - RS4];                    \
            const float4 t2 = lds4[(oq * 2 + 0) * PL4 + ro4 + RS4], t3 = lds4[(oq * 2 + 1) * PL4 + ro4 + RS4];                    \
            const float uu[4] = {t0.x, t0.y, t0.z, t0.w}, vu[4] = {t1.x, t1.y, t1.z, t1.w};                                       \
            const float ud[4] = {t2.x, t2.y, t2.z, t2.w}, vd[4] = {t3.x, t3.y, t3.z, t3.w};                                       \
            const float4 tw = lds4[(4 + oq) * PL4 + ro4 - RS4]; const float wu[4] = {tw.x, tw.y, tw.z, tw.w};                     \
            /* strip-edge horizontal neighbour: left of pixel 0 (START == 0) or right of pixel 7 (START == 1) */                   \
            const float eu = lds[4 * ((oq * 2 + 0) * PL4 + ro4) + ((START) == 0 ? -1 : 4)];                                       \
            const float ev = lds[4 * ((oq * 2 + 1) * PL4 + ro4) + ((START) == 0 ? -1 : 4)];                                       \
            _Pragma("unroll")                                                                                                     \
            for (int k = 0; k < 4; k++) {                                                                                         \
                const int i = (START) + 2 * k;                                                                                    \
                const float wl = i == 0 ? wl0 : wp[i == 0 ? 0 : i - 1];                                                           \
                const float ul = i == 0 ? eu : du[i == 0 ? 0 : i - 1], vl = i == 0 ? ev : dv[i == 0 ? 0 : i - 1];                \
                const float ur = i == 7 ? eu : du[i == 7 ? 7 : i + 1], vr = i == 7 ? ev : dv[i == 7 ? 7 : i + 1];                \
                const float sigmaU = wl * ul + wp[i] * ur + wu[k] * uu[k] + wp[i] * ud[k];                                        \
                const float sigmaV = wl * vl + wp[i] * vr + wu[k] * vu[k] + wp[i] * vd[k];                                        \
                float nu = du[i], nv = dv[i];                                                                                     \
                if constexpr (RCP) {                                                                                              \
                    nu += omega * (sor_div(sigmaU + b1[i] - nv * a12[i], a11[i], r11[i]) - nu);                                   \
                    nv += omega * (sor_div(sigmaV + b2[i] - nu * a12[i], a22[i], r22[i]) - nv);                                   \
                    du[i] = nu; dv[i] = nv;                                                                                       \
                } else if constexpr (DIV == 2) {                                                                                  \
                    nu += omega * (sor_div(sigmaU + b1[i] - nv * a12[i], a11[i], sor_rcp(a11[i])) - nu);                          \
                    nv += omega * (sor_div(sigmaV + b2[i] - nu * a12[i], a22[i], sor_rcp(a22[i])) - nv);                          \
                    const bool ok = (valid >> i) & 1u;                                                                            \
                    du[i] = ok ? nu : 0.f; dv[i] = ok ? nv : 0.f;                                                                 \
                } else {                                                                                                          \
                    nu += omega * ((sigmaU + b1[i] - nv * a12[i]) / a11[i] - nu);                                                 \
                    nv += omega * ((sigmaV + b2[i] - nu * a12[i]) / a22[i] - nv);                                                 \
                    const bool ok = (valid >> i) & 1u;                                                                            \
                    du[i] = ok ? nu : 0.f; dv[i] = ok ? nv : 0.f;                                                                 \
                }                                                                                                                 \
            }                                                                                                                     \
            lds4[((Q) * 2 + 0) * PL4 + ro4] = make_float4(du[START], du[(START) + 2], du[(START) + 4], du[(START) + 6]);         \
            lds4[((Q) * 2 + 1) * PL4 + ro4] = make_float4(dv[START], dv[(START) + 2], dv[(START) + 4], dv[(START) + 6]);         \
        }

    // globally red pixels have local parity `off`; in this thread's row they are the strip pixels i == c (mod 2), c wave-uniform
    if ((off ^ s0) == 0) {
        for (int it = 0; it < iters; it++) { SOR_HALF(0, off) __syncthreads(); SOR_HALF(1, off ^ 1) __syncthreads(); }
    } else {
        for (int it = 0; it < iters; it++) { SOR_HALF(1, off) __syncthreads(); SOR_HALF(0, off ^ 1) __syncthreads(); }
    }
    #undef SOR_HALF
    if (row_ok) {
        const int ix0 = tx * IW, iy0 = ty * IH;
        if (gy >= iy0 && gy < iy0 + IH) {
            // 16-byte stores for the half strips that lie inside the written interior (the interior starts at a multiple of 44 = 4 x 11 pixels, so its
            // quads are whole): the per-pixel stores of a whole strip cost the L2 ~17 masked 64-byte write requests per wave instruction, and the
            // write requests outnumbered the read requests of the tile loads (TCP_TCC_WRITE_REQ 1.17e10 vs READ 1.02e10 over the same launches)
            #pragma unroll
            for (int q = 0; q < SOR_PX; q += 4) {
                const int gxa = gx0 + q; const size_t g = base + (size_t)gy * w + gxa;
                if (((valid >> q) & 0xfu) == 0xfu && gxa >= ix0 && gxa + 4 <= ix0 + IW) {
                    *reinterpret_cast<F4u*>(gUout + g) = F4u{du[q], du[q + 1], du[q + 2], du[q + 3]};
                    *reinterpret_cast<F4u*>(gVout + g) = F4u{dv[q], dv[q + 1], dv[q + 2], dv[q + 3]};
                } else {
                    #pragma unroll
                    for (int i = q; i < q + 4; i++) {
                        const int gx = gx0 + i;
                        if (((valid >> i) & 1u) && gx >= ix0 && gx < ix0 + IW) { gUout[g + (i - q)] = du[i]; gVout[g + (i - q)] = dv[i]; }
                    }
                }
            }
        }
    }
}

#ifdef SIND_LAB      /* dormant variant of the measurement rounds (make lab): kept for A/B timing, bit-identical to the shipped kernels */
// ---------------------------------------------------------------------------------------------------------
// Second generation of the fused kernel for the tiled levels: 1x4 pixel strips (1024 threads per 64x64 tile) leave room in the
// 128-VGPR budget for the RECIPROCALS of A11 / A22, so the two IEEE divisions of a pixel update (~14 VALU operations each, and the
// kernel is VALU-bound once the coefficients come from L2) become Markstein's three-operation sequence
//     q0 = n * r;  e = fma(-a, q0, n);  q = fma(e, r, q0)        with r = RN(1 / a)
// which returns the correctly rounded quotient RN(n / a) for a correctly rounded reciprocal (Markstein 1990; no overflow /
// underflow: the system is O(1)).  Same LDS scheme as above with float2 instead of float4 (two pixels per colour and strip).
// tests/test_flow_gpu.py::test_sor_variants_agree_bitwise holds this kernel to the per-colour reference kernel bit for bit.
#define SOR4_PX 4
__global__ void __launch_bounds__(1024) k_sor_fused4(int w, int h, int EW, int EH, int IW, int IH, int halo_x, int halo, int ntx, int iters, int xcd_remap, float omega,
                                                     const float* __restrict__ gA11, const float* __restrict__ gA12, const float* __restrict__ gA22,
                                                     const float* __restrict__ gB1, const float* __restrict__ gB2, const float* __restrict__ gW,
                                                     const float* __restrict__ gUin, const float* __restrict__ gVin, float* __restrict__ gUout,
                                                     float* __restrict__ gVout) {
    extern __shared__ float4 lds4[];
    float2* lds2 = reinterpret_cast<float2*>(lds4);
    float* lds = reinterpret_cast<float*>(lds4);
    const int SW = EW / SOR4_PX;                 // strips per row
    const int RS2 = EW / 4 + 2;                  // LDS row stride in float2 (one guard float2 on each side)
    const int half = blockDim.x >> 1;
    const int NR = 2 * (half / SW);
    const int PL2 = (NR + 2) * RS2;
    const int tid = threadIdx.x;
    const int idx = tid < half ? tid : tid - half;
    const int j = idx % SW, ly = 2 * (idx / SW) + (tid < half ? 0 : 1);
    int tile = blockIdx.x, b = blockIdx.y;
    {
        const long long ntile = gridDim.x, lin = blockIdx.x + (long long)blockIdx.y * ntile, per = ntile * gridDim.y / 8;
        if (xcd_remap && lin < per * 8) { const long long logical = (lin & 7) * per + (lin >> 3); b = (int)(logical / ntile); tile = (int)(logical - (long long)b * ntile); }
    }
    const int tx = tile % ntx, ty = tile / ntx;
    const int ex0 = tx * IW - halo_x, ey0 = ty * IH - halo;
    const int gy = ey0 + ly, gx0 = ex0 + SOR4_PX * j;
    const size_t base = (size_t)b * w * h;
    const int off = (ex0 + ey0) & 1;

    for (int i = tid; i < 4 * PL2; i += blockDim.x) lds2[i] = make_float2(0.f, 0.f);
    float a11[4], a12[4], a22[4], b1[4], b2[4], wp[4], wu[4], du[4], dv[4], r11[4], r22[4];
    float wl0 = 0.f;
    unsigned valid = 0;
    const bool row_ok = ly < EH && gy >= 0 && gy < h;
    #pragma unroll
    for (int i = 0; i < 4; i++) { a11[i] = 1.f; a12[i] = 0.f; a22[i] = 1.f; b1[i] = 0.f; b2[i] = 0.f; wp[i] = 0.f; wu[i] = 0.f; du[i] = 0.f; dv[i] = 0.f; }
    #define LD4(P, D) { const F4u t_ = *reinterpret_cast<const F4u*>(P); D[0] = t_.x; D[1] = t_.y; D[2] = t_.z; D[3] = t_.w; }
    if (row_ok && gx0 >= 0 && gx0 + 4 <= w) {
        const size_t g = base + (size_t)gy * w + gx0;
        valid = 0xfu;
        LD4(gA11 + g, a11) LD4(gA12 + g, a12) LD4(gA22 + g, a22) LD4(gB1 + g, b1) LD4(gB2 + g, b2) LD4(gW + g, wp) LD4(gUin + g, du) LD4(gVin + g, dv)
        if (gy > 0) LD4(gW + g - w, wu)
    } else if (row_ok) {
        #pragma unroll
        for (int i = 0; i < 4; i++) {
            const int gx = gx0 + i;
            if (gx >= 0 && gx < w) {
                const size_t g = base + (size_t)gy * w + gx;
                valid |= 1u << i;
                a11[i] = gA11[g]; a12[i] = gA12[g]; a22[i] = gA22[g]; b1[i] = gB1[g]; b2[i] = gB2[g]; wp[i] = gW[g];
                wu[i] = gy > 0 ? gW[g - w] : 0.f;
                du[i] = gUin[g]; dv[i] = gVin[g];
            }
        }
    }
    #undef LD4
    if (row_ok && gx0 - 1 >= 0 && gx0 - 1 < w) wl0 = gW[base + (size_t)gy * w + gx0 - 1];
    #pragma unroll
    for (int i = 0; i < 4; i++) { r11[i] = 1.0f / a11[i]; r22[i] = 1.0f / a22[i]; }          // correctly rounded (IEEE divide flag of the build)
    __syncthreads();
    const int ro2 = (ly + 1) * RS2 + 1 + j;
    const int s0 = ly & 1;
    {
        const int pe = s0 == 0 ? 0 : 2, po = s0 == 0 ? 2 : 0;
        lds2[(pe + 0) * PL2 + ro2] = make_float2(du[0], du[2]); lds2[(pe + 1) * PL2 + ro2] = make_float2(dv[0], dv[2]);
        lds2[(po + 0) * PL2 + ro2] = make_float2(du[1], du[3]); lds2[(po + 1) * PL2 + ro2] = make_float2(dv[1], dv[3]);
    }
    __syncthreads();

    #define SOR4_HALF(START, Q)                                                                                                   \
        {                                                                                                                         \
            const int oq = (Q) ^ 1;                                                                                               \
            const float2 t0 = lds2[(oq * 2 + 0) * PL2 + ro2 - RS2], t1 = lds2[(oq * 2 + 1) * PL2 + ro2 - RS2];                    \
            const float2 t2 = lds2[(oq * 2 + 0) * PL2 + ro2 + RS2], t3 = lds2[(oq * 2 + 1) * PL2 + ro2 + RS2];                    \
            const float uu[2] = {t0.x, t0.y}, vu[2] = {t1.x, t1.y}, ud[2] = {t2.x, t2.y}, vd[2] = {t3.x, t3.y};                   \
            const float eu = lds[2 * ((oq * 2 + 0) * PL2 + ro2) + ((START) == 0 ? -1 : 2)];                                       \
            const float ev = lds[2 * ((oq * 2 + 1) * PL2 + ro2) + ((START) == 0 ? -1 : 2)];                                       \
            _Pragma("unroll")                                                                                                     \
            for (int k = 0; k < 2; k++) {                                                                                         \
                const int i = (START) + 2 * k;                                                                                    \
                const float wl = i == 0 ? wl0 : wp[i == 0 ? 0 : i - 1];                                                           \
                const float ul = i == 0 ? eu : du[i == 0 ? 0 : i - 1], vl = i == 0 ? ev : dv[i == 0 ? 0 : i - 1];                \
                const float ur = i == 3 ? eu : du[i == 3 ? 3 : i + 1], vr = i == 3 ? ev : dv[i == 3 ? 3 : i + 1];                \
                const float sigmaU = wl * ul + wp[i] * ur + wu[i] * uu[k] + wp[i] * ud[k];                                        \
                const float sigmaV = wl * vl + wp[i] * vr + wu[i] * vu[k] + wp[i] * vd[k];                                        \
                float nu = du[i], nv = dv[i];                                                                                     \
                nu += omega * (sor_div(sigmaU + b1[i] - nv * a12[i], a11[i], r11[i]) - nu);                                       \
                nv += omega * (sor_div(sigmaV + b2[i] - nu * a12[i], a22[i], r22[i]) - nv);                                       \
                const bool ok = (valid >> i) & 1u;                                                                                \
                du[i] = ok ? nu : 0.f; dv[i] = ok ? nv : 0.f;                                                                     \
            }                                                                                                                     \
            lds2[((Q) * 2 + 0) * PL2 + ro2] = make_float2(du[START], du[(START) + 2]);                                            \
            lds2[((Q) * 2 + 1) * PL2 + ro2] = make_float2(dv[START], dv[(START) + 2]);                                            \
        }
    if ((off ^ s0) == 0) {
        for (int it = 0; it < iters; it++) { SOR4_HALF(0, off) __syncthreads(); SOR4_HALF(1, off ^ 1) __syncthreads(); }
    } else {
        for (int it = 0; it < iters; it++) { SOR4_HALF(1, off) __syncthreads(); SOR4_HALF(0, off ^ 1) __syncthreads(); }
    }
    #undef SOR4_HALF
    if (row_ok) {
        const int ix0 = tx * IW, iy0 = ty * IH;
        if (gy >= iy0 && gy < iy0 + IH) {
            #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gx = gx0 + i;
                if (((valid >> i) & 1u) && gx >= ix0 && gx < ix0 + IW) { const size_t g = base + (size_t)gy * w + gx; gUout[g] = du[i]; gVout[g] = dv[i]; }
            }
        }
    }
}

// tempW = W + dW (end of a fixed-point iteration); with commit != 0 also W = tempW (end of the level)

#endif  // SIND_LAB

// ---------------------------------------------------------------------------------------------------------
// Third generation for the large levels: the solver STREAMS down the image instead of tiling it.
// The tiled kernel above pays for its halo twice: a 64 x 64 tile with a 10-pixel halo computes (64 / 44)^2 = 2.1 x the pixel updates it keeps, and a
// workgroup can neither load the next tile's coefficients nor store its result while it iterates (loads + write-back are 58 % of a tiled launch).
// Here ONE workgroup owns one column strip of an image (up to 152 columns: 128 kept + 12 halo columns on each cut side) and walks it top to bottom as a
// software pipeline in time:
//   * half-sweep s (s = 0 .. 2 * iters - 1, red first) of image row y runs at step t = y + 2 s.  Row y then has rows y - 1 and y + 1 exactly after
//     half-sweep s - 1 and before s + 1 -- what the sequential red-black order defines -- and the rows updated in one step (all of t's parity) never
//     read each other: every pixel update is the same arithmetic on the same operands as in k_sor_color, so the result is bit-identical, with NO
//     redundant update and every coefficient read from memory exactly once per launch;
//   * a thread owns a 1 x 4 strip of TWO consecutive rows (2 p, 2 p + 1) for the SS_NQ steps they spend in the pipeline (every step: a half-sweep of its odd
//     row, then one of its even row), keeps their system AND their du / dv in registers, then takes the pair SS_NQ pairs further down;
//   * du, dv and the smoothness weight also live in LDS rings of SS_RING rows, split by column parity (the two pixels a strip updates in a half-sweep and
//     their vertical neighbours are one 8-byte access per plane): a thread reads from them only what OTHER threads own -- the row above its even row, the
//     row below its odd row and the strip-edge neighbours; threads are grouped by the parity of their pair slot (whole waves per group) so that the active
//     colour is a scalar per wave;
//   * the rows ahead of the pipeline are fetched by two LOADER waves (one per row of a pair): every step a loader parks the 16-byte pieces it requested two
//     steps earlier in LDS (coefficients and the reciprocals of A11 / A22, which it forms: staging ring, read once by the row's next owner; du / dv / w:
//     the rings) and requests the pieces of the pair three steps ahead.  The step barrier waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier), so those
//     loads stay in flight across steps.  (One loader wave for both rows was the step time: DESIGN 3.1-9, profiles/r03/v2_step_probes.txt);
//   * a finished row goes from LDS to the output planes (ping-pong with the input: neighbouring column strips read each other's halo columns).
#define SS_NQ 10           /* pair slots = rows in flight / 2 = half-sweeps per launch (5 iterations) */
#define SS_RING 28         /* rows of du / dv / w resident in LDS: the rows of pair p are parked during step p - 1 and read until step p + SS_NQ */
#define SS_STG 4           /* rows of the coefficient staging ring */
#define SS_MAXSW 38        /* widest column strip in 4-pixel strips: 6 compute waves + 2 loader waves = 512 threads, two workgroups per CU */
#define SS_NST 7           /* staging planes: A11, A12, A22, b1, b2 and the reciprocals of A11 and A22 (formed by the loader wave) */
typedef float ss_f4 __attribute__((ext_vector_type(4)));
typedef float ss_f2 __attribute__((ext_vector_type(2)));
typedef float ss_f2a __attribute__((ext_vector_type(2), aligned(4)));      // two neighbouring floats at any 4-byte address (ds_read2_b32)
__device__ __forceinline__ void ss_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef SIND_LAB
// lab builds: where a step's cycles go, per wave of the workgroups (0, y): [wave][0] cycles from the step's start to its barrier, [1] cycles in the barrier,
// [2] steps, [3] / [4] the same two sums over the steps in which the wave holds threads changing row pairs; waves 14 / 15 = the loaders
// ([3] there: the wait for the loads of two steps ago)
__device__ unsigned long long g_ss_prof[16][6];
#define SS_PROF_BEGIN unsigned long long pr_b = 0, pr_w = 0, pr_n = 0, pr_hb = 0, pr_hn = 0;
#define SS_PROF_T0 const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime();
#define SS_PROF_MARK { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pr_hb += __builtin_amdgcn_s_memtime() - pr_t0; pr_hn++; }     /* (loader: the wait for the loads of two steps ago) */
#define SS_PROF_BARRIER(HEAVY) { const unsigned long long pr_t1 = __builtin_amdgcn_s_memtime(); ss_lds_barrier(); const unsigned long long pr_t2 = __builtin_amdgcn_s_memtime(); \
        pr_b += pr_t1 - pr_t0; pr_w += pr_t2 - pr_t1; pr_n++; if (HEAVY) { pr_hb += pr_t1 - pr_t0; pr_hn++; } }
#define SS_PROF_END(WAVE) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicAdd(&g_ss_prof[WAVE][0], pr_b); atomicAdd(&g_ss_prof[WAVE][1], pr_w); atomicAdd(&g_ss_prof[WAVE][2], pr_n); \
        atomicAdd(&g_ss_prof[WAVE][3], pr_hb); atomicAdd(&g_ss_prof[WAVE][4], pr_hn); }
#else
#define SS_PROF_BEGIN
#define SS_PROF_T0
#define SS_PROF_MARK
#define SS_PROF_BARRIER(HEAVY) ss_lds_barrier();
#define SS_PROF_END(WAVE)
#endif
struct SsRow { float a11[4], a12[4], a22[4], b1[4], b2[4], wp[4], r11[4], r22[4], du[4], dv[4]; float wl0; };      // du / dv: the owner's copy (the rings hold the same values for the neighbours and the write-back); r = RN(1 / a), formed once per row by the loader wave (0 for a pixel outside the image: its update returns exactly 0)
// MAXSW: widest column strip (in 4-pixel strips) the instance is laid out for.  The LDS layout is fixed at compile time (every plane at a constant offset:
// an access is one base register per ring row plus an immediate); 38 strips = 152 columns need 69 KB and 512 threads (6 compute waves + 2 loaders):
// two workgroups per CU.  Wider levels are cut into column strips (grid x), each with its own pipeline.
// One (column strip, image) of a launch: the whole row pipeline of that strip, top to bottom.
template <int MAXSW>
__device__ __forceinline__ void ss_item(float* lds, const int strip, const int image, int w, int h, int SW, int HT, int IW, float omega, const float* __restrict__ gA11, const float* __restrict__ gA12,
                                                     const float* __restrict__ gA22, const float* __restrict__ gB1, const float* __restrict__ gB2,
                                                     const float* __restrict__ gW, const float* __restrict__ gU, const float* __restrict__ gV, float* __restrict__ gUo, float* __restrict__ gVo) {
    constexpr int HS = 2 * MAXSW + 4, EWS = 4 * MAXSW;            // floats per row: split plane (two guard floats on each side), staging
    constexpr int PL = SS_RING * HS;                              // one parity plane of a ring
    constexpr int O_DU = 2, O_DV = 2 * PL + 2, O_W = 4 * PL + 2, O_ST = 6 * PL, STP = SS_STG * EWS;      // float offsets (rings: + parity * PL + slot * HS + x / 2)
    constexpr int total = 6 * PL + SS_NST * STP;
    int tid_ = threadIdx.x; asm volatile("" : "+v"(tid_));     // (opaque per item: nothing derived from it is hoisted out of the kernel's item loop and kept in registers across items)
    const int tid = tid_;
    const size_t base = (size_t)image * w * h;
    // column strip of this workgroup: it keeps the columns [ix0, ix1) and works on [ex0, ex0 + 4 SW), 12 columns more on every side that is not an image
    // border (a cut edge reads zeros from outside; what that falsifies creeps inwards one column per half-sweep, 10 columns in a launch; 12 keeps ex0 a multiple of 4)
    const int ix0 = strip * IW, ix1 = min(ix0 + IW, w), ex0 = max(ix0 - 12, 0);
    for (int i = tid; i < total; i += blockDim.x) lds[i] = 0.f;   // guards, the row above the image, and everything not yet loaded read as zero
    // roles: threads [0, 10 SW) compute -- slot group g (parity of the pair index: first 5 SW threads even slots, next 5 SW odd), slot within the group, strip;
    // the LAST TWO waves of the block are the loaders, one for the even and one for the odd row of every pair (their own registers hold two rows of 16-byte
    // pieces in flight each; the compute waves hold none).  One loader wave for both rows ran ~3500 cycles a step against ~1900 of a compute wave and was
    // the step time (s_memtime probes of the lab build, profiles/tools/ss_step_profile.py)
    const int CT = (int)blockDim.x - 128;                         // compute threads (padded to whole waves)
    const int lw = (tid - CT) >> 6;                               // loader index (0, 1) if this wave loads
    const bool is_loader = tid >= CT;
    const int ct = tid;
    const int g = ct >= HT ? 1 : 0, idx = g ? ct - HT : ct, qs = idx / SW, j = idx - qs * SW;
    int p = (!is_loader && qs < SS_NQ / 2) ? 2 * qs + g : (1 << 28);      // current row pair; padding lanes and the loader never get one
    const int k2 = 2 * j, x0 = 4 * j;
    // loader: lane l takes the items l, l + 64, ... of a row's 8 SW pieces (plane 0..4 coefficients, 5 weight, 6 du, 7 dv; 4-pixel chunk)
    const int llane = tid & 63, lrow = lw;
    constexpr int SS_NC = (8 * MAXSW + 63) / 64;                  // pieces per loader lane and row
    ss_f4 pf[2][SS_NC];                                           // [step parity][piece]: this loader's row of two row pairs in flight
    #pragma unroll
    for (int a = 0; a < 2; a++) {
        #pragma unroll
        for (int c = 0; c < SS_NC; c++) pf[a][c] = ss_f4{0.f, 0.f, 0.f, 0.f};
    }
    SsRow A, B;
    #pragma unroll
    for (int i = 0; i < 4; i++) { A.a11[i] = A.a22[i] = B.a11[i] = B.a22[i] = 1.f; A.a12[i] = A.b1[i] = A.b2[i] = A.wp[i] = A.r11[i] = A.r22[i] = 0.f; B.a12[i] = B.b1[i] = B.b2[i] = B.wp[i] = B.r11[i] = B.r22[i] = 0.f; }
    A.wl0 = B.wl0 = 0.f;
    #pragma unroll
    for (int i = 0; i < 4; i++) A.du[i] = A.dv[i] = B.du[i] = B.dv[i] = 0.f;
    unsigned smask = 0u;                                         // strip pixels inside the columns this workgroup keeps
    #pragma unroll
    for (int i = 0; i < 4; i++) if (ex0 + x0 + i >= ix0 && ex0 + x0 + i < ix1) smask |= 1u << i;
    __syncthreads();

    // row ROWY (ring slot SLOT) leaves LDS for registers (its owner, at the row's first step)
    #define SS_LOAD(R, ROWY, RB)                                                                                                   \
        {                                                                                                                          \
            const float* st_ = lds + O_ST + ((ROWY) & (SS_STG - 1)) * EWS + k2; const float* rb_ = (RB);                          \
            /* the staging rows are split by column parity like the rings: (pixel 0, pixel 2) and (pixel 1, pixel 3) arrive as the register pairs the    \
               packed FP32 operations of a half-sweep take (a float4 per plane left the pairs to be rebuilt by moves in every half-sweep) */            \
            const float2 e0 = *reinterpret_cast<const float2*>(st_), o0 = *reinterpret_cast<const float2*>(st_ + EWS / 2);                             \
            const float2 e1 = *reinterpret_cast<const float2*>(st_ + STP), o1 = *reinterpret_cast<const float2*>(st_ + STP + EWS / 2);                 \
            const float2 e2 = *reinterpret_cast<const float2*>(st_ + 2 * STP), o2 = *reinterpret_cast<const float2*>(st_ + 2 * STP + EWS / 2);         \
            const float2 e3 = *reinterpret_cast<const float2*>(st_ + 3 * STP), o3 = *reinterpret_cast<const float2*>(st_ + 3 * STP + EWS / 2);         \
            const float2 e4 = *reinterpret_cast<const float2*>(st_ + 4 * STP), o4 = *reinterpret_cast<const float2*>(st_ + 4 * STP + EWS / 2);         \
            const float2 e5 = *reinterpret_cast<const float2*>(st_ + 5 * STP), o5 = *reinterpret_cast<const float2*>(st_ + 5 * STP + EWS / 2);         \
            const float2 e6 = *reinterpret_cast<const float2*>(st_ + 6 * STP), o6 = *reinterpret_cast<const float2*>(st_ + 6 * STP + EWS / 2);         \
            const float2 we = *reinterpret_cast<const float2*>(rb_ + O_W), wo = *reinterpret_cast<const float2*>(rb_ + O_W + PL); \
            R.wl0 = rb_[O_W + PL - 1];                                                                                             \
            const float2 ue_ = *reinterpret_cast<const float2*>(rb_ + O_DU), uo_ = *reinterpret_cast<const float2*>(rb_ + O_DU + PL); \
            const float2 ve_ = *reinterpret_cast<const float2*>(rb_ + O_DV), vo_ = *reinterpret_cast<const float2*>(rb_ + O_DV + PL); \
            R.du[0] = ue_.x; R.du[2] = ue_.y; R.du[1] = uo_.x; R.du[3] = uo_.y; R.dv[0] = ve_.x; R.dv[2] = ve_.y; R.dv[1] = vo_.x; R.dv[3] = vo_.y; \
            R.a11[0] = e0.x; R.a11[2] = e0.y; R.a11[1] = o0.x; R.a11[3] = o0.y; R.a12[0] = e1.x; R.a12[2] = e1.y; R.a12[1] = o1.x; R.a12[3] = o1.y; \
            R.a22[0] = e2.x; R.a22[2] = e2.y; R.a22[1] = o2.x; R.a22[3] = o2.y; R.b1[0] = e3.x; R.b1[2] = e3.y; R.b1[1] = o3.x; R.b1[3] = o3.y;     \
            R.b2[0] = e4.x; R.b2[2] = e4.y; R.b2[1] = o4.x; R.b2[3] = o4.y;                                                        \
            R.wp[0] = we.x; R.wp[2] = we.y; R.wp[1] = wo.x; R.wp[3] = wo.y;                                                        \
            R.r11[0] = e5.x; R.r11[2] = e5.y; R.r11[1] = o5.x; R.r11[3] = o5.y; R.r22[0] = e6.x; R.r22[2] = e6.y; R.r22[1] = o6.x; R.r22[3] = o6.y; \
        }
    // one half-sweep of a row: the two strip pixels of column parity START (the strip starts at an even column).  The thread holds the values of its two rows
    // in registers (the rings carry the same values for the neighbours): the odd row reads its upper neighbour and that row's weights from the even row's
    // registers, the even row its lower neighbour from the odd row's; LDS gives only the row of the OTHER thread (above the even row, below the odd one) and
    // the strip-edge neighbour.  UU / VU / WU, UD / VD: the vertical neighbours' values at the two updated columns.
    #define SS_UPDATE(R, RB, START, UU0, UU1, VU0, VU1, WU0, WU1, UD0, UD1, VD0, VD1)                                              \
        {                                                                                                                          \
            float* rb_ = (RB) + (START) * PL;                                                                                      \
            /* strip-edge horizontal neighbour: left of pixel 0 (an odd column) or right of pixel 3 (the next strip's first, even column) */ \
            /* ... read TOGETHER with the own pixel beside it (pixel 1 resp. pixel 2: the ring holds the thread's own values too, written a step ago), as the register pair the   \
               packed operations take: (edge, 1) resp. (2, edge) used to be put together by two moves per plane and half-sweep */                                              \
            const ss_f2a eup = *reinterpret_cast<const ss_f2a*>((START) == 0 ? rb_ + O_DU + PL - 1 : rb_ + O_DU - PL + 1);           \
            const ss_f2a evp = *reinterpret_cast<const ss_f2a*>((START) == 0 ? rb_ + O_DV + PL - 1 : rb_ + O_DV - PL + 1);           \
            const float uua[2] = {UU0, UU1}, uda[2] = {UD0, UD1}, vua[2] = {VU0, VU1}, vda[2] = {VD0, VD1}, wua[2] = {WU0, WU1};     \
            /* START 0: pixels 0, 2 between (edge, 1) and (1, 3); START 1: pixels 1, 3 between (0, 2) and (2, edge) */             \
            const float ula[2] = {(START) == 0 ? eup.x : R.du[0], (START) == 0 ? eup.y : R.du[2]}, ura[2] = {(START) == 0 ? R.du[1] : eup.x, (START) == 0 ? R.du[3] : eup.y}; \
            const float vla[2] = {(START) == 0 ? evp.x : R.dv[0], (START) == 0 ? evp.y : R.dv[2]}, vra[2] = {(START) == 0 ? R.dv[1] : evp.x, (START) == 0 ? R.dv[3] : evp.y}; \
            _Pragma("unroll")                                                                                                      \
            for (int k = 0; k < 2; k++) {                                                                                          \
                const int i = (START) + 2 * k;                                                                                     \
                const float wl = i == 0 ? R.wl0 : R.wp[i == 0 ? 0 : i - 1];                                                        \
                const float sigmaU = wl * ula[k] + R.wp[i] * ura[k] + wua[k] * uua[k] + R.wp[i] * uda[k];                          \
                const float sigmaV = wl * vla[k] + R.wp[i] * vra[k] + wua[k] * vua[k] + R.wp[i] * vda[k];                          \
                float nu = R.du[i], nv = R.dv[i];                                                                                  \
                nu += omega * (sor_div(sigmaU + R.b1[i] - nv * R.a12[i], R.a11[i], R.r11[i]) - nu);                                \
                nv += omega * (sor_div(sigmaV + R.b2[i] - nu * R.a12[i], R.a22[i], R.r22[i]) - nv);                                \
                R.du[i] = nu; R.dv[i] = nv;                                                                                        \
            }                                                                                                                      \
            /* (one 8-byte vector store each: as HIP float2 structs the stores were split into scalars, sunk below the two colours' code and came out as four ds_write_b32 with four address adds) */ \
            *reinterpret_cast<ss_f2*>(rb_ + O_DU) = ss_f2{R.du[START], R.du[(START) + 2]}; *reinterpret_cast<ss_f2*>(rb_ + O_DV) = ss_f2{R.dv[START], R.dv[(START) + 2]}; \
        }
    #define SS_HALF_A(START)                                                                                                       \
        {                                                                                                                          \
            const float* ru_ = pU + (START) * PL;                                                                                  \
            const float2 uu = *reinterpret_cast<const float2*>(ru_ + O_DU), vu = *reinterpret_cast<const float2*>(ru_ + O_DV), wu = *reinterpret_cast<const float2*>(ru_ + O_W); \
            SS_UPDATE(A, pA, START, uu.x, uu.y, vu.x, vu.y, wu.x, wu.y, B.du[START], B.du[(START) + 2], B.dv[START], B.dv[(START) + 2]) \
        }
    #define SS_HALF_B(START)                                                                                                       \
        {                                                                                                                          \
            const float* rd_ = pD + (START) * PL;                                                                                  \
            const float2 ud = *reinterpret_cast<const float2*>(rd_ + O_DU), vd = *reinterpret_cast<const float2*>(rd_ + O_DV);     \
            SS_UPDATE(B, pB, START, A.du[START], A.du[(START) + 2], A.dv[START], A.dv[(START) + 2], A.wp[START], A.wp[(START) + 2], ud.x, ud.y, vd.x, vd.y) \
        }
    // a finished row: LDS (both column parities) -> global memory
    #define SS_STORE(ROWY, RB)                                                                                                     \
        {                                                                                                                          \
            const float* rb_ = (RB);                                                                                               \
            const float2 ue = *reinterpret_cast<const float2*>(rb_ + O_DU), uo = *reinterpret_cast<const float2*>(rb_ + O_DU + PL); \
            const float2 ve = *reinterpret_cast<const float2*>(rb_ + O_DV), vo = *reinterpret_cast<const float2*>(rb_ + O_DV + PL); \
            const float du_[4] = {ue.x, uo.x, ue.y, uo.y}, dv_[4] = {ve.x, vo.x, ve.y, vo.y};                                       \
            const size_t go = base + (size_t)(ROWY) * w + ex0 + x0;                                                                \
            if (smask == 0xfu) { *reinterpret_cast<F4u*>(gUo + go) = F4u{du_[0], du_[1], du_[2], du_[3]}; *reinterpret_cast<F4u*>(gVo + go) = F4u{dv_[0], dv_[1], dv_[2], dv_[3]}; } \
            else { _Pragma("unroll") for (int i = 0; i < 4; i++) if ((smask >> i) & 1u) { gUo[go + i] = du_[i]; gVo[go + i] = dv_[i]; } } \
        }

    // The loader wave and the compute waves run their own copy of the step loop (same number of steps, one barrier each): the register allocation of one
    // role does not see the other's live values (two rows of pieces in flight there, two rows of coefficients here)
    SS_PROF_BEGIN
    if (is_loader) {
        // per piece, fixed for the launch: source pointer of row 0, LDS offset without the row part, kind (0 staging, 1 ring, -1 none), in-image mask of the 4 pixels
        // an A11 / A22 piece also leaves its reciprocals in the staging planes 5 / 6 (prd: distance to that plane, 0 = none): the step's one heavy job of
        // a compute thread -- taking over a new row pair -- is then LDS reads only.  (With the reciprocals formed by the row's owner the wave holding the
        // threads that change pairs in a step ran ~2.8 x the instructions of the others, and every step has such a wave: the barrier made its path the step time.)
        // Pieces 0 .. NCS-1 hold the staging planes' chunks (A11, A22, A12, b1, b2 -- the two that need reciprocals first), the rest the ring planes' (w, du, dv):
        // what a piece is, is known when the code is compiled, and everything that is the same for the whole wave is kept scalar (readfirstlane)
        constexpr int NCS = (5 * MAXSW + 63) / 64, NCR = (3 * MAXSW + 63) / 64, NCRCP = (2 * MAXSW + 63) / 64;
        static_assert(NCS + NCR == SS_NC, "k_sor_stream: pieces per loader lane");
        const int srow = __builtin_amdgcn_readfirstlane(lrow);
        const float* psrc[SS_NC]; int pdst[SS_NC], prd[SS_NC]; unsigned pmask[SS_NC]; bool pact[SS_NC];
        int slowbits = 0, rcpbits = 0;                            // per piece: holds du / dv chunks that stick out of the image; holds A11 / A22 chunks
        #pragma unroll
        for (int c = 0; c < SS_NC; c++) {
            const bool stg = c < NCS;
            const int id = llane + 64 * (stg ? c : c - NCS), lp = id / SW, lch = id - lp * SW, gx = ex0 + 4 * lch;
            pact[c] = id < (stg ? 5 : 3) * SW;
            const float* plane = stg ? (lp == 0 ? gA11 : lp == 1 ? gA22 : lp == 2 ? gA12 : lp == 3 ? gB1 : gB2) : (lp == 0 ? gW : lp == 1 ? gU : gV);
            psrc[c] = (pact[c] ? plane : gA11) + base + min(gx, w - 1);      // (a chunk wholly right of the image re-reads around the last pixel; inactive lanes read somewhere valid)
            // staging planes in LDS: 0 A11, 1 A12, 2 A22, 3 b1, 4 b2, 5 1 / A11, 6 1 / A22; every plane split by column parity: even columns first
            pdst[c] = (stg ? O_ST + (lp == 0 ? 0 : lp == 1 ? 2 : lp == 2 ? 1 : lp) * STP : (lp == 0 ? O_W : lp == 1 ? O_DU : O_DV)) + 2 * lch;
            pmask[c] = (gx < w ? 1u : 0u) | (gx + 1 < w ? 2u : 0u) | (gx + 2 < w ? 4u : 0u) | (gx + 3 < w ? 8u : 0u);
            prd[c] = (stg && pact[c] && lp == 0) ? 5 * STP : (stg && pact[c] && lp == 1) ? 4 * STP : 0;
            // what has to read as zero outside the image is du / dv (a neighbour's sum takes them) and the reciprocals (an outside pixel's update is then
            // exactly 0 whatever finite coefficients it reads: the row's continuation in memory, or the zeroed padding behind the planes)
            if (__builtin_amdgcn_ballot_w64(!stg && pact[c] && lp >= 1 && pmask[c] != 0xfu) != 0ull) slowbits |= 1 << c;
            if (__builtin_amdgcn_ballot_w64(prd[c] != 0) != 0ull) rcpbits |= 1 << c;
        }
        slowbits = __builtin_amdgcn_readfirstlane(slowbits); rcpbits = __builtin_amdgcn_readfirstlane(rcpbits);
        for (int T0 = -4; T0 < (h + 1) / 2 + SS_NQ + 2; T0 += 2) {
            #pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                // step T: the pieces of row pair T + 1 (requested two steps ago) go to LDS, then the pieces of pair T + 3 are requested: SS_NC loads per step and loader,
                // always (rows outside the image re-read row 0 / h - 1 and are masked when parked), so that "at most SS_NC loads outstanding" means exactly
                // "the loads of two steps ago have landed".  The loads are inline assembly: the compiler's own s_waitcnt would be vmcnt(0) at every use of a
                // loaded register in this loop (it cannot count across the back edge), i.e. one full memory latency per piece instead of per step.
                const int T = T0 + tt;
                SS_PROF_T0
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SS_NC) : "memory");
                SS_PROF_MARK
                {
                    const int ys = 2 * (T + 1) + srow;
                    const bool park = ys >= 0 && ys <= h + 1, rowin = ys < h;       // the two rows below the image read as zero
                    const int stg_off = (ys & (SS_STG - 1)) * EWS, ring_off = (((ys % SS_RING) + SS_RING) % SS_RING) * HS;
                    #pragma unroll
                    for (int c = 0; c < SS_NC; c++) {
                        ss_f4 q = pf[tt][c];
                        asm volatile("" : "+v"(q));                                   // (uses stay behind the wait above)
                        if (!park) continue;
                        if (c < NCS) {
                            if (!rowin) continue;                                     // (nobody takes over a row below the image)
                            float* dst = lds + pdst[c] + stg_off;
                            if (pact[c]) { *reinterpret_cast<float2*>(dst) = make_float2(q.x, q.z); *reinterpret_cast<float2*>(dst + EWS / 2) = make_float2(q.y, q.w); }
                            if (c < NCRCP && ((rcpbits >> c) & 1)) {
                                const float r0 = sor_rcp(q.x), r1 = sor_rcp(q.y), r2 = sor_rcp(q.z), r3 = sor_rcp(q.w);
                                const unsigned m = pmask[c];
                                if (prd[c]) {
                                    *reinterpret_cast<float2*>(dst + prd[c]) = make_float2((m & 1u) ? r0 : 0.f, (m & 4u) ? r2 : 0.f);
                                    *reinterpret_cast<float2*>(dst + prd[c] + EWS / 2) = make_float2((m & 2u) ? r1 : 0.f, (m & 8u) ? r3 : 0.f);
                                }
                            }
                        } else {
                            float* dst = lds + pdst[c] + ring_off;
                            if (((slowbits >> c) & 1) || !rowin) {
                                const unsigned m = rowin ? pmask[c] : 0u;                // (a weight chunk is masked with its piece: nobody reads a weight outside the image)
                                if (pact[c]) { *reinterpret_cast<float2*>(dst) = make_float2((m & 1u) ? q.x : 0.f, (m & 4u) ? q.z : 0.f); *reinterpret_cast<float2*>(dst + PL) = make_float2((m & 2u) ? q.y : 0.f, (m & 8u) ? q.w : 0.f); }
                            } else if (pact[c]) { *reinterpret_cast<float2*>(dst) = make_float2(q.x, q.z); *reinterpret_cast<float2*>(dst + PL) = make_float2(q.y, q.w); }
                        }
                    }
                }
                {
                    const int yl = min(max(2 * (T + 3) + srow, 0), h - 1);
                    const size_t src_off = (size_t)yl * w;
                    #pragma unroll
                    for (int c = 0; c < SS_NC; c++) { const float* a = psrc[c] + src_off; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pf[tt][c]) : "v"(a) : "memory"); }
                }
                SS_PROF_BARRIER(false)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SS_PROF_END(14 + lrow)
        return;
    }
    // Per-thread pipeline state, touched only when the thread changes pairs (every SS_NQ steps): the ring rows of its pair (pA, pB), of the row above (pU)
    // and below (pD), whether the pair's rows exist, and s = T - p counted up step by step
    float *pA, *pB, *pU, *pD, *pS = lds; int yS = 0; bool okA, okB, stS = false;
    #define SS_PAIR()                                                                                                              \
        {                                                                                                                          \
            const int sa_ = (2 * p) % SS_RING;                                                                                     \
            pA = lds + sa_ * HS + k2; pB = lds + (sa_ + 1) * HS + k2;        /* (2 p is even and SS_RING is: the odd row never wraps) */ \
            pU = lds + (sa_ == 0 ? SS_RING - 1 : sa_ - 1) * HS + k2; pD = lds + (sa_ + 2 == SS_RING ? 0 : sa_ + 2) * HS + k2;      \
            okA = 2 * p < h; okB = 2 * p + 1 < h;                                                                                  \
        }
    SS_PAIR()
    int sg = -4 - p;
    const int sgrp = __builtin_amdgcn_readfirstlane(g);        // (a slot group is whole waves: the colour of a step is a scalar)
    for (int T0 = -4; T0 < (h + 1) / 2 + SS_NQ + 2; T0 += 2) {
        #pragma unroll 1
        for (int tt = 0; tt < 2; tt++) {
            const int par = (tt + sgrp) & 1;
            // step T of this thread's row pair p: s = T - p.  First its odd row's half-sweep s - 1, then its even row's half-sweep s (the even row's vertical
            // neighbours in the odd row are of the colour just updated -- same thread, same columns).  Both update the column parity s & 1 = (T + g) & 1.
            SS_PROF_T0
            if (sg == 1 && stS) SS_STORE(yS, pS)               // the odd row of the pair left in the last step (the ring keeps a finished row for two steps)
            if (sg >= 1 && okB) { if (par == 0) SS_HALF_B(0) else SS_HALF_B(1) }
            if (sg == SS_NQ) {                                  // the pair is through: the thread takes the pair SS_NQ further down (same parity)
                pS = pB; yS = 2 * p + 1; stS = okB;
                p += SS_NQ; sg = 0;
                SS_PAIR()
            }
            if (sg == 0 && okA) {
                SS_LOAD(A, 2 * p, pA)
                if (okB) SS_LOAD(B, 2 * p + 1, pB)
                else { _Pragma("unroll") for (int i = 0; i < 4; i++) B.du[i] = B.dv[i] = 0.f; }      // the row below the image reads as zero
            }
            if (sg >= 0 && okA) { if (par == 0) SS_HALF_A(0) else SS_HALF_A(1) }
            if (sg == SS_NQ - 1 && okA) SS_STORE(2 * p, pA)
            SS_PROF_BARRIER(__builtin_amdgcn_ballot_w64(sg == 0 && okA) != 0ull)
            sg++;
        }
    }
    #undef SS_PAIR
    SS_PROF_END(ct >> 6)
    #undef SS_LOAD
    #undef SS_HALF_A
    #undef SS_HALF_B
    #undef SS_UPDATE
    #undef SS_STORE
}
// The launch: workgroup k takes the items k, k + gridDim.x, ... (item = strip + strips * image).  With as many workgroups as items (the default) a workgroup has one item;
// with FEWER (sor_iterations: C.stream_wg_cap) the workgroups are persistent: they keep their place on their compute units for the whole launch instead of handing it back
// after every item -- and a place that is handed back while other kernels' small workgroups are queued at a higher stream priority comes back in pieces.
template <int MAXSW>
__global__ void __attribute__((amdgpu_flat_work_group_size(64, 512), amdgpu_waves_per_eu(4, 4))) k_sor_stream(int strips, int items, int w, int h, int SW, int HT, int IW, float omega, const float* __restrict__ gA11, const float* __restrict__ gA12,
                                                     const float* __restrict__ gA22, const float* __restrict__ gB1, const float* __restrict__ gB2,
                                                     const float* __restrict__ gW, const float* __restrict__ gU, const float* __restrict__ gV, float* __restrict__ gUo, float* __restrict__ gVo) {
    extern __shared__ float4 lds4s[];
    float* lds = reinterpret_cast<float*>(lds4s);
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int image = item / strips, strip = item - image * strips;
        ss_item<MAXSW>(lds, strip, image, w, h, SW, HT, IW, omega, gA11, gA12, gA22, gB1, gB2, gW, gU, gV, gUo, gVo);
        __syncthreads();                                          // every wave is through with the item's LDS before the next item clears it
    }
}

#ifdef SIND_LAB
int debug_ss_profile(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ss_prof), sizeof(g_ss_prof)) != hipSuccess) return SIND_E_HIP;
    if (reset) { static const unsigned long long z[16][6] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ss_prof), z, sizeof(z)) != hipSuccess) return SIND_E_HIP; }
    return SIND_OK;
}
#endif

// ---------------------------------------------------------------------------------------------------------
// Level transition of the DeepFlow loop in ONE launch instead of three (k_add_flow, k_resize_f32_pair, k_warp_avg_iz of the next level; deepflow.cpp: W += dW; resize(W) / 0.95;
// prepareBuffers of the next level).  A thread makes LU_ROWS pixels of one column of the NEXT level: the up-sampled flow from the four taps of fl(W + dW) -- the very values
// k_add_flow would have stored --, times `post`; then the warp of I1 by that flow, the averaged image and the temporal difference; and the next level's zero increment, which goes
// to the ping-pong partner of dW (other threads still read this level's dW as taps).  Same float operations on the same values as the three kernels: same bits.
#define LU_ROWS 4
__global__ void k_level_up(const float* __restrict__ Wu, const float* __restrict__ Wv, const float* __restrict__ dWu, const float* __restrict__ dWv, int sw, int sh,
                           const float* __restrict__ I0, const float* __restrict__ I1, float* __restrict__ nWu, float* __restrict__ nWv, float* __restrict__ ndWu, float* __restrict__ ndWv,
                           float* __restrict__ avg, float* __restrict__ Iz, int dw, int dh, double scale_x, double scale_y, float post) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy0 = blockIdx.y * LU_ROWS, b = blockIdx.z;
    if (dx >= dw) return;
    const size_t sbase = (size_t)b * sw * sh, dbase = (size_t)b * dw * dh;
    const float* SU = Wu + sbase; const float* SV = Wv + sbase; const float* DU = dWu + sbase; const float* DV = dWv + sbase;
    const float* S1 = I1 + dbase;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = d_cvFloorf(fx); fx -= sx;
    const bool two = sx + 1 < sw;            // dx < xmax in OpenCV's HResizeLinear
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const float a1 = fx, a0 = 1.f - a1;
    #pragma unroll
    for (int r = 0; r < LU_ROWS; r++) {
        const int dy = dy0 + r;
        if (dy >= dh) break;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = d_cvFloorf(fy); fy -= sy;
        const int y0 = d_clip(sy, 0, sh), y1 = d_clip(sy + 1, 0, sh);
        const float b1 = fy, b0 = 1.f - b1;
        const int i00 = y0 * sw + sx, i10 = y1 * sw + sx;
        float ru0, ru1, rv0, rv1;
        if (two) {
            ru0 = (SU[i00] + DU[i00]) * a0 + (SU[i00 + 1] + DU[i00 + 1]) * a1; ru1 = (SU[i10] + DU[i10]) * a0 + (SU[i10 + 1] + DU[i10 + 1]) * a1;
            rv0 = (SV[i00] + DV[i00]) * a0 + (SV[i00 + 1] + DV[i00 + 1]) * a1; rv1 = (SV[i10] + DV[i10]) * a0 + (SV[i10 + 1] + DV[i10 + 1]) * a1;
        } else {
            ru0 = (SU[i00] + DU[i00]) * 1.f; ru1 = (SU[i10] + DU[i10]) * 1.f; rv0 = (SV[i00] + DV[i00]) * 1.f; rv1 = (SV[i10] + DV[i10]) * 1.f;
        }
        float vu = ru0 * b0 + ru1 * b1, vv = rv0 * b0 + rv1 * b1;
        vu = vu * post; vv = vv * post;
        const size_t o = dbase + (size_t)dy * dw + dx;
        nWu[o] = vu; nWv[o] = vv; ndWu[o] = 0.f; ndWv[o] = 0.f;
        warp_px(S1, I0[o], vu, vv, dx, dy, dw, dh, avg[o], Iz[o]);
    }
}
// after the call the planes of P describe the next level: W = the up-sampled flow, dW = 0, avg / Iz = its warped buffers
int launch_level_up(hipStream_t s, FlowPlanes& P, int sw, int sh, const float* I0, const float* I1, int dw, int dh, int B, float post) {
    hipLaunchKernelGGL(k_level_up, dim3(divup(dw, 128), divup(dh, LU_ROWS), B), dim3(128), 0, s, P.Wu, P.Wv, P.dWu, P.dWv, sw, sh, I0, I1, P.tWu, P.tWv, P.dWu2, P.dWv2, P.avg, P.Iz, dw, dh,
                       1. / ((double)dw / sw), 1. / ((double)dh / sh), post);
    HIP_TRY(hipGetLastError());
    std::swap(P.Wu, P.tWu); std::swap(P.Wv, P.tWv); std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);
    return SIND_OK;
}

// four elements per thread (16-byte accesses; `n4` whole quads, the n % 4 elements behind them by the last threads)
__global__ void k_add_flow(const float* Wu, const float* Wv, const float* __restrict__ dWu,
                           const float* __restrict__ dWv, float* tWu, float* tWv, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n4 = n >> 2;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(Wu)[i], b = reinterpret_cast<const float4*>(dWu)[i], c = reinterpret_cast<const float4*>(Wv)[i], d = reinterpret_cast<const float4*>(dWv)[i];
        reinterpret_cast<float4*>(tWu)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        reinterpret_cast<float4*>(tWv)[i] = make_float4(c.x + d.x, c.y + d.y, c.z + d.z, c.w + d.w);
    } else {
        const size_t k = (n4 << 2) + (i - n4);
        if (k < n) { tWu[k] = Wu[k] + dWu[k]; tWv[k] = Wv[k] + dWv[k]; }
    }
}
__global__ void k_add_flow_1(const float* Wu, const float* Wv, const float* __restrict__ dWu,
                             const float* __restrict__ dWv, float* tWu, float* tWv, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tWu[i] = Wu[i] + dWu[i]; tWv[i] = Wv[i] + dWv[i];
}

__global__ void k_scale(float* __restrict__ a, float* __restrict__ b, float s, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = a[i] * s; b[i] = b[i] * s;
}

// ---------------------------------------------------------------------------------------------------------
// Large-motion test, reference DynaDetect.cc:1081-1114: magnitude, max, u8 normalisation, 256-bin histogram.
// Pass 1: per-image max of |flow| (non-negative floats order like their bit patterns -> integer atomicMax).
__global__ void k_mag_max(const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ mag, unsigned* __restrict__ maxbits, int n) {
    const int b = blockIdx.y;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = u[(size_t)b * n + i], y = v[(size_t)b * n + i];
        const float g = sqrtf(x * x + y * y);
        if (mag) mag[(size_t)b * n + i] = g;
        m = fmaxf(m, g);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&maxbits[b], __float_as_uint(m));
}
// Pass 2: histogram of saturate_cast<uchar>(mag * (float)(255.0/max)); LDS histogram per workgroup, one global
// atomic per bin per workgroup.  Optionally stores the u8 image (residual stage).
__global__ void k_mag_hist(const float* __restrict__ mag, const unsigned* __restrict__ maxbits, int* __restrict__ hist,
                           uint8_t* __restrict__ out_u8, int n) {
    __shared__ int lh[256];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const float a = (float)(255.0 / (double)__uint_as_float(maxbits[b]));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int q = d_cvRound(mag[(size_t)b * n + i] * a);
        q = min(max(q, 0), 255);
        if (out_u8) out_u8[(size_t)b * n + i] = (uint8_t)q;
        atomicAdd(&lh[q], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) if (lh[i]) atomicAdd(&hist[b * 256 + i], lh[i]);
}

// ---------------------------------------------------------------------------------------------------------
// Residual stage, reference DynaDetect.cc:1252-1271: flow - (p - H p), homography part in FP64 then cast to FP32.
// A thread takes RM_ROWS pixels of one column (2 400 workgroups of one row segment each were mostly workgroup turnover: 11 us of the whole GPU per 640 x 480 frame, 512 times a step)
#define RM_ROWS 4
__global__ void k_residual_mag(const float* __restrict__ u, const float* __restrict__ v, HMat Hm, float* __restrict__ mag,
                               unsigned* __restrict__ maxbits, int w, int h) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row0 = blockIdx.y * RM_ROWS;
    float m = 0.f;
    if (col < w) {
        const double* H = Hm.h;
        #pragma unroll
        for (int r = 0; r < RM_ROWS; r++) {
            const int row = row0 + r;
            if (row >= h) break;
            const double den = H[6] * col + H[7] * row + H[8];
            const double fx2 = (col - (H[0] * col + H[1] * row + H[2]) / den);
            const double fy2 = (row - (H[3] * col + H[4] * row + H[5]) / den);
            const float dx = u[row * w + col] - (float)fx2, dy = v[row * w + col] - (float)fy2;
            const float mm = sqrtf(dx * dx + dy * dy);
            mag[row * w + col] = mm;
            m = fmaxf(m, mm);
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // thousands of waves hitting one address serialise in L2: a wave only issues the atomic when it can still raise the maximum
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *(volatile unsigned*)maxbits) atomicMax(maxbits, __float_as_uint(m));
}
// masks from the u8 residual: low -> 128, high -> 255 (stImgMasks), thresholds decided on the host from the histogram
__global__ void k_threshold_masks(const uint8_t* __restrict__ magu8, float thr_low, float thr_high, uint8_t* __restrict__ low,
                                  uint8_t* __restrict__ high, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q = (double)magu8[i];
    low[i] = q > (double)thr_low ? 128 : 0;
    high[i] = q > (double)thr_high ? 255 : 0;
}
// The two thresholds of stImgMasks from the residual's 256-bin histogram, on the device so that the flow-mask stage needs ONE host round trip (masks,
// histogram and thresholds come back together): cv::threshold's THRESH_OTSU / THRESH_TRIANGLE return values (imgproc/thresh.cpp) and the clamping of
// DD:1309-1367, with the host code's FP64 / FP32 operations in the same order (no contraction: the library is built with -ffp-contract=off).
// k_flow_thresholds_serial is the one-thread statement of it (83 us per frame: every iteration of Otsu's loop carries an FP64 division) and stays as the
// in-library reference; k_flow_thresholds gives the same bits from ONE WAVE:
//   * sums of integers below 2^53 (the mean, Triangle's distances, the pixel counts) are exact in FP64 in any order -> lane-parallel + wave reduction;
//   * q1 += p_i is a rounding chain -> walked in order, but it does not depend on mu1, so it runs first (one dependent add per bin);
//   * mu1 = (mu1 * q1_old + i * p_i) / q1 is the only chain with a division: the divisor q1_i is known beforehand, so every lane prepares the refined
//     reciprocal y_i of its four bins with the compiler's own FP64 division sequence (v_rcp_f64 + two Newton steps), and the chain keeps only that
//     sequence's last three operations (q0 = n y, r = fma(-q1, q0, n), q = fma(r, y, q0)); operands are far from the exponent range where
//     v_div_scale / v_div_fixup would act (q1 in [1.2e-7, 1], n in [0, 255]);
//   * sigma, its arg max (first maximum wins, like the sequential `>` test) and Triangle's arg max are lane-parallel again.
// hist: [256] counts + [256] = bit pattern of the maximal residual; res: [261] = the same 257 words + lo, hi, otsu, triangle; the kernel leaves hist
// ZEROED for the next frame (the residual kernels accumulate into it).  dbg_mu1 (optional): the 256 values of the mu1 chain.  One block = one histogram.
__global__ void k_flow_thresholds_serial(const int* __restrict__ hist, int W, int H, float* __restrict__ out) {
    hist += (size_t)blockIdx.x * 257; out += (size_t)blockIdx.x * 4;
    if (threadIdx.x != 0) return;
    const int N = W * H;
    const float maxErrorf = __int_as_float(hist[256]);
    double otsu_v;
    {
        double mu = 0; const double scale = 1. / N;
        for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
        mu *= scale;
        double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
        for (int i = 0; i < 256; i++) {
            const double p_i = hist[i] * scale; mu1 *= q1; q1 += p_i; const double q2 = 1. - q1;
            if (fmin(q1, q2) < (double)1.1920928955078125e-7f || fmax(q1, q2) > 1. - (double)1.1920928955078125e-7f) continue;
            mu1 = (mu1 + i * p_i) / q1; const double mu2 = (mu - q1 * mu1) / q2, sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
            if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
        }
        otsu_v = max_val;
    }
    double tri_v;
    {
        int left = 0, right = 0, max_ind = 0, mx = 0; bool flipped = false;
        for (int i = 0; i < 256; i++) if (hist[i] > 0) { left = i; break; }
        if (left > 0) left--;
        for (int i = 255; i > 0; i--) if (hist[i] > 0) { right = i; break; }
        if (right < 255) right++;
        for (int i = 0; i < 256; i++) if (hist[i] > mx) { mx = hist[i]; max_ind = i; }
        if (max_ind - left < right - max_ind) { flipped = true; left = 255 - right; max_ind = 255 - max_ind; }
        double thresh = left, a = mx, b = left - max_ind, dist = 0;
        for (int i = left + 1; i <= max_ind; i++) { const double t = a * i + b * hist[flipped ? 255 - i : i]; if (t > dist) { dist = t; thresh = i; } }
        thresh--;
        if (flipped) thresh = 255 - thresh;
        tri_v = thresh;
    }
    float thred1 = (float)otsu_v, thred2 = (float)tri_v;
    out[2] = thred1; out[3] = thred2;
    auto count_gt = [&](float t) { int n = 0; for (int v = 0; v < 256; v++) if ((double)v > (double)t) n += hist[v]; return n; };
    float lo, hi;
    if (thred1 < thred2) {                                    // DD:1309-1336
        if (thred1 < 1.7f * 255.0f / maxErrorf) thred1 = 1.7f * 255.0f / maxErrorf;
        else if (thred1 > 3.0f * 255.0f / maxErrorf) thred1 = 3.0f * 255.0f / maxErrorf;
        if (count_gt(thred1) > 0.5 * W * H) thred1 = thred1 + 0.2f * 255.0f / maxErrorf;
        if (thred2 < fmaxf(3.0f * 255.0f / maxErrorf, thred1 * 1.2f)) thred2 = fmaxf(3.0f * 255.0f / maxErrorf, thred1 * 1.2f);
        else if (thred2 > 10.0f * 255.0f / maxErrorf) thred2 = 10.0f * 255.0f / maxErrorf;
        lo = thred1; hi = thred2;
    } else {                                                  // DD:1337-1367 (the relaxation test there is dead code)
        if (thred2 < 1.7f * 255.0f / maxErrorf) thred2 = 1.7f * 255.0f / maxErrorf;
        else if (thred2 > 3.0f * 255.0f / maxErrorf) thred2 = 3.0f * 255.0f / maxErrorf;
        if (thred1 < fmaxf(3.0f * 255.0f / maxErrorf, thred2 * 1.2f)) thred1 = fmaxf(3.0f * 255.0f / maxErrorf, thred2 * 1.2f);
        else if (thred1 > 10.0f * 255.0f / maxErrorf) thred1 = 10.0f * 255.0f / maxErrorf;
        lo = thred2; hi = thred1;
    }
    out[0] = lo; out[1] = hi;
}

__device__ __forceinline__ double wave_sum_f64(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }      // exact when every partial sum is an integer below 2^53
__device__ __forceinline__ int wave_sum_i32(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int wave_min_i32(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_max_i32(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }
// arg max with "first maximum wins": larger value, on equal values the smaller index
__device__ __forceinline__ void wave_argmax_f64(double& v, int& idx) {
    for (int o = 32; o > 0; o >>= 1) { const double ov = __shfl_xor(v, o); const int oi = __shfl_xor(idx, o); if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; } }
}
__global__ void __launch_bounds__(64) k_flow_thresholds(int* __restrict__ hist, int W, int H, int* __restrict__ res, double* __restrict__ dbg_mu1, int keep_hist) {
    hist += (size_t)blockIdx.x * 257; res += (size_t)blockIdx.x * 261; if (dbg_mu1) dbg_mu1 += (size_t)blockIdx.x * 256;
    __shared__ double s_p[256];
    __shared__ double4 s_c[256];                        // per bin: q1_i, i * p_i, y_i, skip flag
    const int lane = threadIdx.x, N = W * H;
    const int4 h4 = *reinterpret_cast<const int4*>(hist + 4 * lane);
    const int hmax_bits = hist[256];
    const int h[4] = {h4.x, h4.y, h4.z, h4.w};
    // result block (the working histogram is zeroed at the very end, when every lane's loads have long been consumed)
    *reinterpret_cast<int4*>(res + 4 * lane) = h4; if (lane == 0) res[256] = hmax_bits;
    const float maxErrorf = __int_as_float(hmax_bits);
    // ---- Otsu
    const double scale = 1. / N;
    double mu = 0, pv[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) { mu += (4 * lane + k) * (double)h[k]; pv[k] = h[k] * scale; s_p[4 * lane + k] = pv[k]; }
    mu = wave_sum_f64(mu) * scale;
    __syncthreads();
    // The chains below are walked by every lane (wave-uniform values); a lane keeps the four values of its own bins.  No LDS store inside the loops: the
    // loads of the coming bins do not depend on the chain and must be free to move ahead of it (an LDS store in between pins them behind it: measured
    // 37 us for the kernel with the stores, the LDS latency then sits on the chain 512 times).
    double q1v[4] = {0, 0, 0, 0};
    {   // q1 chain (same hand pipelining as the mu1 chain below)
        double q1 = 0;
        double4 cur = *reinterpret_cast<const double4*>(&s_p[0]), nxt;
        for (int L = 0; L < 64; L++) {
            nxt = *reinterpret_cast<const double4*>(&s_p[4 * min(L + 1, 63)]);
            __builtin_amdgcn_sched_barrier(0);
            const bool mine = L == lane;
            q1 += cur.x; if (mine) q1v[0] = q1;
            q1 += cur.y; if (mine) q1v[1] = q1;
            q1 += cur.z; if (mine) q1v[2] = q1;
            q1 += cur.w; if (mine) q1v[3] = q1;
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    double q2v[4]; bool skip[4];
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = 4 * lane + k; const double q1 = q1v[k], q2 = 1. - q1; q2v[k] = q2;
        skip[k] = fmin(q1, q2) < (double)1.1920928955078125e-7f || fmax(q1, q2) > 1. - (double)1.1920928955078125e-7f;
        // the compiler's FP64 division up to the refined reciprocal (AMDGPU legalizeFDIV64: rcp, two Newton steps in FMA arithmetic)
        const double b = skip[k] ? 1.0 : q1, r0 = __builtin_amdgcn_rcp(b), e0 = fma(-b, r0, 1.0), r1 = fma(r0, e0, r0), e1 = fma(-b, r1, 1.0), y = fma(r1, e1, r1);
        s_c[i] = make_double4(q1, i * pv[k], y, skip[k] ? 1.0 : 0.0);
    }
    __syncthreads();
    double mu1v[4] = {0, 0, 0, 0};
    {   // mu1 chain: mu1 *= q1_old; then either stays (skipped bin) or becomes (mu1 + i p_i) / q1_i.  Software-pipelined by hand: the four bins of lane
        // L + 1 are fetched before the chain walks the bins of lane L (the scheduler otherwise issues each read right before its use)
        double mu1 = 0, q1_old = 0;
        double4 cur[4], nxt[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = s_c[k];
        for (int L = 0; L < 64; L++) {
            const int Ln = min(L + 1, 63);
            #pragma unroll
            for (int k = 0; k < 4; k++) nxt[k] = s_c[4 * Ln + k];
            __builtin_amdgcn_sched_barrier(0);
            const bool mine = L == lane;
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                const double4 c = cur[k];
                const double m = mu1 * q1_old, n = m + c.y, q0 = n * c.z, r = fma(-c.x, q0, n), q = fma(r, c.z, q0);
                mu1 = c.w != 0.0 ? m : q; q1_old = c.x;
                if (mine) mu1v[k] = mu1;
            }
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = nxt[k];
        }
    }
    double best = 0; int best_i = 0x7fffffff;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = 4 * lane + k; const double mu1 = mu1v[k];
        if (dbg_mu1) dbg_mu1[i] = mu1;
        if (skip[k]) continue;
        const double q1 = q1v[k], q2 = q2v[k], mu2 = (mu - q1 * mu1) / q2, sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > best) { best = sigma; best_i = i; }
    }
    wave_argmax_f64(best, best_i);
    const double otsu_v = best > 0 ? (double)best_i : 0.0;
    // ---- Triangle
    double tri_v;
    {
        int lf = 256, rt = 0, mx = 0, mi = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++) { const int i = 4 * lane + k; if (h[k] > 0) { lf = min(lf, i); if (i > 0) rt = max(rt, i); } if (h[k] > mx) { mx = h[k]; mi = i; } }
        int left = wave_min_i32(lf); if (left == 256) left = 0;
        int right = wave_max_i32(rt);
        { double mxd = mx; int mid = mx > 0 ? mi : 0x7fffffff; wave_argmax_f64(mxd, mid); mx = (int)mxd; mi = mx > 0 ? mid : 0; }
        int max_ind = mi;
        if (left > 0) left--;
        if (right < 255) right++;
        bool flipped = false;
        if (max_ind - left < right - max_ind) { flipped = true; left = 255 - right; max_ind = 255 - max_ind; }
        const double a = mx, b = left - max_ind;
        double dist = 0; int thr_i = 0x7fffffff;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 4 * lane + k, i = flipped ? 255 - j : j;          // position i of the (possibly mirrored) histogram holds bin j
            if (i < left + 1 || i > max_ind) continue;
            const double t = a * i + b * h[k];
            if (t > dist || (t == dist && t > 0 && i < thr_i)) { dist = t; thr_i = i; }
        }
        wave_argmax_f64(dist, thr_i);
        double thresh = dist > 0 ? (double)thr_i : (double)left;
        thresh--;
        if (flipped) thresh = 255 - thresh;
        tri_v = thresh;
    }
    // ---- clamping (wave-uniform scalars)
    float thred1 = (float)otsu_v, thred2 = (float)tri_v;
    const float otsu_f = thred1, tri_f = thred2;
    float lo, hi;
    if (thred1 < thred2) {                                    // DD:1309-1336
        if (thred1 < 1.7f * 255.0f / maxErrorf) thred1 = 1.7f * 255.0f / maxErrorf;
        else if (thred1 > 3.0f * 255.0f / maxErrorf) thred1 = 3.0f * 255.0f / maxErrorf;
        int cnt = 0;
        #pragma unroll
        for (int k = 0; k < 4; k++) if ((double)(4 * lane + k) > (double)thred1) cnt += h[k];
        cnt = wave_sum_i32(cnt);
        if (cnt > 0.5 * W * H) thred1 = thred1 + 0.2f * 255.0f / maxErrorf;
        if (thred2 < fmaxf(3.0f * 255.0f / maxErrorf, thred1 * 1.2f)) thred2 = fmaxf(3.0f * 255.0f / maxErrorf, thred1 * 1.2f);
        else if (thred2 > 10.0f * 255.0f / maxErrorf) thred2 = 10.0f * 255.0f / maxErrorf;
        lo = thred1; hi = thred2;
    } else {                                                  // DD:1337-1367 (the relaxation test there is dead code)
        if (thred2 < 1.7f * 255.0f / maxErrorf) thred2 = 1.7f * 255.0f / maxErrorf;
        else if (thred2 > 3.0f * 255.0f / maxErrorf) thred2 = 3.0f * 255.0f / maxErrorf;
        if (thred1 < fmaxf(3.0f * 255.0f / maxErrorf, thred2 * 1.2f)) thred1 = fmaxf(3.0f * 255.0f / maxErrorf, thred2 * 1.2f);
        else if (thred1 > 10.0f * 255.0f / maxErrorf) thred1 = 10.0f * 255.0f / maxErrorf;
        lo = thred2; hi = thred1;
    }
    if (lane == 0) { float* out = reinterpret_cast<float*>(res + 257); out[0] = lo; out[1] = hi; out[2] = otsu_f; out[3] = tri_f; }
    if (!keep_hist) { *reinterpret_cast<int4*>(hist + 4 * lane) = make_int4(0, 0, 0, 0); if (lane == 0) hist[256] = 0; }
}
// (16 pixels per thread where the planes allow 16-byte accesses: one pixel per thread was 1 200 workgroups of turnover per 640 x 480 frame)
__global__ void k_threshold_masks_dev(const uint8_t* __restrict__ magu8, const float* __restrict__ thr, uint8_t* __restrict__ low, uint8_t* __restrict__ high, int n, int wide) {
    const double tl = (double)thr[0], th = (double)thr[1];
    if (wide) {
        const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
        if (i >= n) return;
        if (i + 16 <= n) {
            const uint4 q4 = *reinterpret_cast<const uint4*>(magu8 + i);
            const unsigned qs[4] = {q4.x, q4.y, q4.z, q4.w}; unsigned lo[4], hi[4];
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                lo[k] = hi[k] = 0u;
                #pragma unroll
                for (int b = 0; b < 4; b++) { const double q = (double)((qs[k] >> (8 * b)) & 255u); lo[k] |= (q > tl ? 128u : 0u) << (8 * b); hi[k] |= (q > th ? 255u : 0u) << (8 * b); }
            }
            *reinterpret_cast<uint4*>(low + i) = make_uint4(lo[0], lo[1], lo[2], lo[3]); *reinterpret_cast<uint4*>(high + i) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            return;
        }
        for (int k = i; k < n; k++) { const double q = (double)magu8[k]; low[k] = q > tl ? 128 : 0; high[k] = q > th ? 255 : 0; }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q = (double)magu8[i];
    low[i] = q > tl ? 128 : 0;
    high[i] = q > th ? 255 : 0;
}
// gather flow at the 63x47 sample grid (DD:1182-1204) so the host can build the PROSAC-ordered pairs
__global__ void k_gather_grid(const float* __restrict__ u, const float* __restrict__ v, float* __restrict__ out, int w, int h, int step) {
    const int gx = (w - 1) / step, gy = (h - 1) / step;   // points at step, 2*step, ... < w
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;      // blockIdx.y = frame of a batch (dense [B][h][w] planes)
    if (i >= gx * gy) return;
    const int r = (i / gx + 1) * step, c = (i % gx + 1) * step;
    const size_t fo = (size_t)b * w * h;
    out[((size_t)b * gx * gy + i) * 2] = u[fo + r * w + c]; out[((size_t)b * gx * gy + i) * 2 + 1] = v[fo + r * w + c];
}

// ---------------------------------------------------------------------------------------------------------
// cvtColor(BGR2GRAY) 8U fixed point and cv::resize(INTER_LINEAR) 8U (11-bit coefficients), reference DD:1390-1392, 1037-1039
__global__ void k_bgr2gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t n, int cb, int cr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = bgr + i * 3;
    gray[i] = (uint8_t)((p[0] * cb + p[1] * 9617 + p[2] * cr + 8192) >> 14);
}
// four pixels per thread: three 4-byte loads, one 4-byte store (n4 = whole quads; needs 4-byte aligned images)
__global__ void k_bgr2gray4(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t n4, int cb, int cr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const unsigned* p = reinterpret_cast<const unsigned*>(bgr) + i * 3;
    const unsigned a = p[0], b = p[1], c = p[2];             // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    auto g = [&](unsigned bb, unsigned gg, unsigned rr) { return (unsigned)(((int)bb * cb + (int)gg * 9617 + (int)rr * cr + 8192) >> 14) & 0xffu; };
    const unsigned g0 = g(a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xff), g1 = g(a >> 24, b & 0xff, (b >> 8) & 0xff);
    const unsigned g2 = g((b >> 16) & 0xff, b >> 24, c & 0xff), g3 = g((c >> 8) & 0xff, (c >> 16) & 0xff, c >> 24);
    reinterpret_cast<unsigned*>(gray)[i] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}
__global__ void k_resize_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int sw, int sh, int dw, int dh,
                            double scale_x, double scale_y, int s_stride, int d_stride, size_t s_img, size_t d_img, int d_group, int d_skip) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy0 = blockIdx.y * RZ_ROWS, b = blockIdx.z;      // RZ_ROWS destination rows of a column per thread, like k_resize_f32_pair
    if (dx >= dw) return;
    const uint8_t* S = src + (size_t)b * s_img;
    const int db = d_group > 0 ? b + (b / d_group) * d_skip : b;       // destination slot: groups of d_group images, d_skip slots left free after each group
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = d_cvFloorf(fx); fx -= sx;
    const bool two = sx + 1 < sw;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    const int ax0 = (short)d_cvRound((1.f - fx) * 2048), ax1 = (short)d_cvRound(fx * 2048);
    #pragma unroll
    for (int r = 0; r < RZ_ROWS; r++) {
        const int dy = dy0 + r;
        if (dy >= dh) break;
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = d_cvFloorf(fy); fy -= sy;
        const int y0 = d_clip(sy, 0, sh), y1 = d_clip(sy + 1, 0, sh);
        const int b0 = (short)d_cvRound((1.f - fy) * 2048), b1 = (short)d_cvRound(fy * 2048);
        const uint8_t* R0 = S + (size_t)y0 * s_stride; const uint8_t* R1 = S + (size_t)y1 * s_stride;
        int r0, r1;
        if (two) { r0 = R0[sx] * ax0 + R0[sx + 1] * ax1; r1 = R1[sx] * ax0 + R1[sx + 1] * ax1; }
        else     { r0 = R0[sx] * 2048;                   r1 = R1[sx] * 2048; }
        dst[(size_t)db * d_img + (size_t)dy * d_stride + dx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

// frames of the gray history pool picked by index into a dense batch: out[b] = pool[idx[b]] (16 bytes per thread; the indices travel as
// kernel arguments, up to 256 per launch) -- one launch instead of one device-to-device copy per frame (3 000 copies per 512-pair step)
struct GatherIdx { int v[256]; };
// End of a step: the last two flow-grid frames of every stream become its history slots 0, 1 (the pool holds T + 2 slots per stream).  One thread moves the
// same 16 bytes of both frames and reads before it writes, so T = 1 (slot 1 is source and destination) needs no second buffer.
__global__ void k_roll_history(uint8_t* __restrict__ pool, int T, size_t fb16) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fb16) return;
    uint4* base = reinterpret_cast<uint4*>(pool) + (size_t)blockIdx.y * (T + 2) * fb16;
    const uint4 a = base[(size_t)T * fb16 + i], b = base[(size_t)(T + 1) * fb16 + i];
    base[i] = a; base[fb16 + i] = b;
}
__global__ void k_gather_frames(const uint8_t* __restrict__ pool, GatherIdx idx, uint8_t* __restrict__ out, size_t fb) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= fb) return;
    const uint8_t* s = pool + (size_t)idx.v[blockIdx.y] * fb + i; uint8_t* d = out + (size_t)blockIdx.y * fb + i;
    if (i + 16 <= fb) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    else for (size_t k = 0; i + k < fb; k++) d[k] = s[k];
}

// ---------------------------------------------------------------------------------------------------------
// host-side launchers
static inline dim3 grid2d(int w, int h, int B, int bx = 128) { return dim3(divup(w, bx), h, B); }

int launch_gather_frames(hipStream_t s, const uint8_t* pool, const int* idx_host, int B, uint8_t* out, size_t fb) {
    if (fb % 16 != 0) { sind_set_error("launch_gather_frames: frame size %zu is not a multiple of 16 bytes", fb); return SIND_E_ARG; }
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int nb = std::min(256, B - b0); GatherIdx a;
        for (int i = 0; i < nb; i++) a.v[i] = idx_host[b0 + i];
        hipLaunchKernelGGL(k_gather_frames, dim3((unsigned)((fb / 16 + 255) / 256), nb), dim3(256), 0, s, pool, a, out + (size_t)b0 * fb, fb);
    }
    return SIND_OK;
}
int launch_u8_to_f32_blur3(hipStream_t s, const uint8_t* src, float* dst, int w, int h, int B, float k0, float k1, bool blur) {
    hipLaunchKernelGGL(k_u8_to_f32_blur3, grid2d(w, h, B), dim3(128), 0, s, src, dst, w, h, k0, k1, blur ? 1 : 0);
    return SIND_OK;
}
int launch_resize_f32(hipStream_t s, const float* src, float* dst, int sw, int sh, int dw, int dh, int B, float post, bool has_post) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    hipLaunchKernelGGL(k_resize_f32, grid2d(dw, dh, B), dim3(128), 0, s, src, dst, sw, sh, dw, dh, scale_x, scale_y, post, has_post ? 1 : 0);
    return SIND_OK;
}
int launch_resize_f32_pair(hipStream_t s, const float* srcA, float* dstA, const float* srcB, float* dstB, int sw, int sh, int dw, int dh, int B, float post, bool has_post) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    hipLaunchKernelGGL(k_resize_f32_pair, grid2d(dw, divup(dh, RZ_ROWS), 2 * B), dim3(128), 0, s, srcA, dstA, srcB, dstB, B, sw, sh, dw, dh, scale_x, scale_y, post, has_post ? 1 : 0);
    return SIND_OK;
}
// levels first+1 .. last of both pyramids (level l of image b at pyr + off[l] * B + b * w[l] * h[l]) from level `first`, which must be complete
int launch_pyramid_tail(hipStream_t s, float* pyrA, float* pyrB, const std::vector<std::pair<int, int>>& levels, const std::vector<size_t>& level_off, int first, int last, int B) {
    if (last <= first) return SIND_OK;
    if (last - first > PYR_TAIL_MAX) { sind_set_error("launch_pyramid_tail: %d levels (at most %d)", last - first, PYR_TAIL_MAX); return SIND_E_ARG; }
    PyrTail T; T.n = last - first; T.B = B;
    for (int k = 0; k <= T.n; k++) {
        const int l = first + k; T.w[k] = levels[l].first; T.h[k] = levels[l].second; T.off[k] = (unsigned long long)level_off[l] * B;
        if (k > 0) { T.sx[k] = 1. / ((double)T.w[k] / T.w[k - 1]); T.sy[k] = 1. / ((double)T.h[k] / T.h[k - 1]); } else { T.sx[k] = T.sy[k] = 1; }
    }
    hipLaunchKernelGGL(k_pyramid_tail, dim3(2 * B), dim3(256), 0, s, pyrA, pyrB, T);
    return SIND_OK;
}
int launch_roll_history(hipStream_t s, uint8_t* pool, int S, int T, size_t frame_bytes) {
    if (frame_bytes % 16) { sind_set_error("roll_history: frame size %zu is not a multiple of 16", frame_bytes); return SIND_E_ARG; }
    const size_t fb16 = frame_bytes / 16;
    hipLaunchKernelGGL(k_roll_history, dim3((unsigned)((fb16 + 255) / 256), S), dim3(256), 0, s, pool, T, fb16);
    return SIND_OK;
}
int launch_resize_u8(hipStream_t s, const uint8_t* src, uint8_t* dst, int sw, int sh, int dw, int dh, int B, int s_stride, int d_stride, size_t s_img, size_t d_img, int d_group, int d_skip) {
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    hipLaunchKernelGGL(k_resize_u8, grid2d(dw, divup(dh, RZ_ROWS), B), dim3(128), 0, s, src, dst, sw, sh, dw, dh, scale_x, scale_y, s_stride, d_stride, s_img, d_img, d_group, d_skip);
    return SIND_OK;
}
int launch_bgr2gray(hipStream_t s, const uint8_t* bgr, uint8_t* gray, size_t npix, bool swap_rb) {
    if ((npix & 3) == 0 && (((uintptr_t)bgr | (uintptr_t)gray) & 3) == 0)
        hipLaunchKernelGGL(k_bgr2gray4, dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, s, bgr, gray, npix / 4, swap_rb ? 4899 : 1868, swap_rb ? 1868 : 4899);
    else
        hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, bgr, gray, npix, swap_rb ? 4899 : 1868, swap_rb ? 1868 : 4899);
    return SIND_OK;
}

// `total` red-black SOR iterations on the level's system.  Small levels: the whole image is one tile and all iterations run
// in one launch; larger levels: 64 x 64 tiles, SOR_FUSE iterations per launch with a 2*SOR_FUSE halo, ping-pong between the
// two increment buffers (a tile reads its halo from neighbours that another workgroup of the same launch rewrites).
// (the solver settings -- variant, iterations per launch, tile, streaming threshold -- are a SolverCfg per flow handle: flow.hpp)
// How the `total` iterations of a level are cut into launches.  A launch of f iterations needs a halo of 2 f pixels, so it covers the level with
// ceil(w / (EW - 4 f)) x ceil(h / (EH - 4 f)) tiles, and a tile costs its prologue (coefficient loads, LDS set-up, write-back: C.plan_cost iterations'
// worth, fitted to the measured 4 / 5 / 6-iteration runs) plus f iterations.  With a fixed f = 5 the 26 tiled levels of the 384 x 288 pyramid compute
// 2.5 x their pixels (levels just above a multiple of the 44-pixel interior up to 3.5 x); the cheapest partition per level (dynamic programme over
// the remaining iterations) picks e.g. 1 + 6 x 4 iterations for 178 x 133: 12 tiles per launch instead of 20.  Measured (bench --sync, solver busy per step):
// 370 ms with f = 5 everywhere, 364-369 ms with the plans for a prologue cost of 7-20 iterations, 392 ms for 4 -- the model's 6 % shrink to 1-2 %, so the plan
// is an option (C.fuse = 0), not the default.
static std::vector<int> sor_fuse_plan(const SolverCfg& C, int w, int h, int EW, int EH, int total) {
    if (C.fuse > 0) return std::vector<int>((size_t)divup(total, C.fuse), C.fuse);
    static std::mutex mu; static std::map<std::array<int, 6>, std::vector<int>> cache;
    const std::array<int, 6> key{w, h, EW, EH, total, (int)std::lrint(C.plan_cost * 16)};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int fmax = std::min(10, (std::min(EW, EH) - 1) / 4);
    std::vector<double> cost((size_t)total + 1, 0.0); std::vector<int> pick((size_t)total + 1, 1);
    for (int rem = 1; rem <= total; rem++) {
        double best = -1; int bf = 1;
        for (int f = 1; f <= std::min(fmax, rem); f++) {
            const int IW = EW - 4 * f, IH = EH - 4 * f;
            const double c = (double)divup(w, IW) * divup(h, IH) * (C.plan_cost + f) + cost[(size_t)rem - f];
            if (best < 0 || c < best) { best = c; bf = f; }
        }
        cost[(size_t)rem] = best; pick[(size_t)rem] = bf;
    }
    std::vector<int> plan;
    for (int rem = total; rem > 0; rem -= pick[(size_t)rem]) plan.push_back(pick[(size_t)rem]);
    std::sort(plan.begin(), plan.end(), std::greater<int>());            // the order of the launches is free: same iterations, same bits
    cache[key] = plan;
    return plan;
}
// Few images: the launch is latency-bound (a tile per compute unit, most of the chip idle).  Then 1024-thread tiles (k_sor_tile) run MORE iterations per launch on deeper
// halos: the fewest launches whose tiles all still find a compute unit of their own (a launch costs ~6 us before its first iteration; an iteration ~0.8 us).
// Returns the iterations of each launch, or nothing if the level at this batch size is throughput-bound (the kernels below).
static std::vector<int> sor_latency_plan(int w, int h, int B, int total) {
    const int slots = 256;                                   // compute units: one 1024-thread workgroup each
    for (int n = 1; n <= total; n++) {
        const int base = total / n, rem = total % n, kmax = base + (rem ? 1 : 0);
        if (kmax > 13) continue;                             // 64 - 4 k >= 12 kept pixels per tile edge
        if ((long long)sor_tile_count(w, h, kmax) * B > slots) { if (kmax <= 5) break; continue; }
        std::vector<int> plan((size_t)n, base);
        for (int i = 0; i < rem; i++) plan[(size_t)i]++;
        return plan;
    }
    return {};
}
int sor_iterations(hipStream_t s, FlowPlanes& P, int w, int h, int B, int total, float omega, long long* nlaunch, const SolverCfg& C, int* streamed) {
    const bool latency_tiles = (C.opts & FLOW_OPT_LATENCY_TILES) != 0;
    if (streamed) *streamed = 0;
    if (C.mode == 0) {
        const dim3 gs(divup(divup(w, 2), 64), h, B), bs(64);
        for (int k = 0; k < total; k++) {
            hipLaunchKernelGGL(k_sor_color, gs, bs, 0, s, w, h, omega, 0, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, P.dWu, P.dWv);
            hipLaunchKernelGGL(k_sor_color, gs, bs, 0, s, w, h, omega, 1, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, P.dWu, P.dWv);
        }
        *nlaunch += 2LL * total; return SIND_OK;
    }
    // the flow slices call this concurrently from their own threads: set the (idempotent) attribute once per device
    static SindPerDeviceInit attr_init;
    HIP_TRY(attr_init.run([] {
        hipError_t attr_rc = hipSuccess;
        const void* fs[] = {(const void*)k_sor_fused<2, 1024, 4, 0, 0>, (const void*)k_sor_fused<2, 512, 2, 0, 0>, (const void*)k_sor_fused<2, 512, 4, 64, 64>, (const void*)k_sor_stream<SS_MAXSW>,
#ifdef SIND_LAB
                            (const void*)k_sor_fused<0, 1024, 4, 0, 0>, (const void*)k_sor_fused<0, 512, 4, 64, 64>, (const void*)k_sor_fused<1, 384, 3, 0, 0>, (const void*)k_sor_fused<1, 768, 3, 0, 0>,
                            (const void*)k_sor_fused<1, 256, 3, 0, 0>,
#endif
        };
        for (const void* f : fs) if (attr_rc == hipSuccess) attr_rc = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        return attr_rc;
    }));
    if (latency_tiles && C.mode == 4) {       // (mode 5 asks for the streaming kernel wherever it fits)
        const std::vector<int> plan = sor_latency_plan(w, h, B, total);
        if (!plan.empty() && !(plan.size() > 1 && 2 * divup(divup(w, 8) * ((h + 1) / 2), 64) * 64 <= 512)) {      // (a level of <= 4096 pixels that is not in the chain: one launch of the one-workgroup kernel below is as good)
            for (int k : plan) { SIND_TRY(launch_sor_tile(s, P, w, h, B, k, omega)); *nlaunch += 1; }
            return SIND_OK;
        }
    }
    auto sor_lds_bytes = [](int EW, int nt) { const int NR = 2 * ((nt / 2) / (EW / SOR_PX)); return (size_t)6 * (NR + 2) * sor_row_stride(EW) * sizeof(float4); };
    auto threads_for = [](int EW, int EH) { const int halfn = (EW / SOR_PX) * ((EH + 1) / 2); return 2 * ((halfn + 63) / 64 * 64); };
    const int EWw = (w + SOR_PX - 1) / SOR_PX * SOR_PX;
    if (threads_for(EWw, h) <= SOR_NT) {               // whole image in one workgroup: every iteration in one launch, in place
        const int EW = EWw, EH = h, nt = threads_for(EW, EH);
        const size_t shm = sor_lds_bytes(EW, nt);
        if (shm > 150 * 1024) { sind_set_error("sor_iterations: a %d x %d level needs %zu bytes of LDS", w, h, shm); return SIND_E_ARG; }
        // <= 512 threads: two waves per SIMD, i.e. up to 256 registers -- room for the run-time tile sizes AND the reciprocal division without spills
        // (all 25 iterations run in this launch, so here the loop is the cost); larger blocks keep the four-waves-per-SIMD IEEE instance
#ifdef SIND_LAB
        auto kern1 = (nt <= 512 && (C.mode == 4 || C.mode == 5)) ? k_sor_fused<2, 512, 2, 0, 0> : (C.mode == 4 || C.mode == 5) ? k_sor_fused<2, 1024, 4, 0, 0> : k_sor_fused<0, 1024, 4, 0, 0>;
#else
        auto kern1 = nt <= 512 ? k_sor_fused<2, 512, 2, 0, 0> : k_sor_fused<2, 1024, 4, 0, 0>;
#endif
        hipLaunchKernelGGL(kern1, dim3(1, B), dim3(nt), shm, s, w, h, EW, EH, EW, EH, 0, 0, 1, total, 0, omega, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt,
                           P.r11, P.r22, P.dWu, P.dWv, P.dWu, P.dWv);
        *nlaunch += 1; return SIND_OK;
    }
    // large levels, many images: one-wave row pipelines (flow_wave.hip) -- no workgroup barrier anywhere
    if (total % 5 == 0 && (C.mode == 6 || (C.mode == 4 && C.wave && B >= C.stream_min_b && w * h >= C.stream_min_px))) {
        for (int done = 0; done < total; done += 5) { SIND_TRY(launch_sor_wave(s, P, w, h, B, omega, C.wave_items, C.wave_bands, C.wave_prefetch)); *nlaunch += 1; }
        if (streamed) *streamed = 1;
        return SIND_OK;
    }
    // ... or the streaming kernel (one workgroup per image and column strip; no halo, loads and stores overlapped with the iterations)
    {
        // column strips: n = fewest strips whose working width (kept columns + 12 on each cut side, a multiple of 4) fits 4 SS_MAXSW = 152 columns
        int n = 1, IW = divup(w, 4) * 4, SW = IW / 4;
        while (SW > SS_MAXSW) { n++; IW = divup(divup(w, n), 4) * 4; SW = (IW + 24) / 4; }
        const int HT = divup(5 * SW, 64) * 64, CT = 2 * HT;       // threads of one slot group (whole waves: a wave holding both groups would run both colours' code every step), compute threads; + two loader waves
        const bool fits = h >= 4 && total % (SS_NQ / 2) == 0;
        if (fits && (C.mode == 5 || (C.mode == 4 && B >= C.stream_min_b && w * h >= C.stream_min_px))) {
            const size_t shm = ((size_t)6 * SS_RING * (2 * SS_MAXSW + 4) + (size_t)SS_NST * SS_STG * 4 * SS_MAXSW) * sizeof(float);
            for (int done = 0; done < total; done += SS_NQ / 2) {
                const int items = n * B, wgs = (C.stream_wg_cap > 0 && items > C.stream_wg_cap) ? divup(items, divup(items, C.stream_wg_cap)) : items;      // persistent: every workgroup the same number of items (+- 1)
                hipLaunchKernelGGL(k_sor_stream<SS_MAXSW>, dim3(wgs), dim3(CT + 128), shm, s, n, items, w, h, SW, HT, IW, omega, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, P.dWu, P.dWv, P.dWu2, P.dWv2);
                std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);      // column strips read each other's halo columns: not in place
                *nlaunch += 1;
            }
            if (streamed) *streamed = 1;
            return SIND_OK;
        }
        if (C.mode == 5 && !fits) { /* wider than one workgroup's strip: the tiled kernel below */ }
    }
#ifdef SIND_LAB
    if (C.mode == 2) {                             // 1x4 strips + reciprocal division (k_sor_fused4), 64 x 64 tiles, 1024 threads
        static SindPerDeviceInit attr4_init;
        HIP_TRY(attr4_init.run([] { return hipFuncSetAttribute((const void*)k_sor_fused4, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); }));
        const int EW = 64, EH = 64, nt = 1024;
        const size_t shm = (size_t)4 * (EH + 2) * (EW / 4 + 2) * sizeof(float2);
        for (int done = 0; done < total;) {
            const int k = std::min(C.fuse > 0 ? C.fuse : 5, total - done), halo = 2 * k, halo_x = halo, IW = EW - 2 * halo_x, IH = EH - 2 * halo;
            const int ntx = divup(w, IW), nty = divup(h, IH);
            hipLaunchKernelGGL(k_sor_fused4, dim3(ntx * nty, B), dim3(nt), shm, s, w, h, EW, EH, IW, IH, halo_x, halo, ntx, k, C.xcd, omega, P.A11, P.A12, P.A22, P.b1, P.b2,
                               P.wgt, P.dWu, P.dWv, P.dWu2, P.dWv2);
            std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);
            done += k; *nlaunch += 1;
        }
        return SIND_OK;
    }
#endif
    // tiled levels.  Reciprocal instance (default): EW x EH tiles of EW * EH / 8 threads at three waves per SIMD; IEEE-division instance: four waves per SIMD
    const int EW = C.tile_w, EH = C.tile_h, nt = threads_for(EW, EH);
    const bool rcp = C.mode == 3;
    if (rcp && nt != 256 && nt != 384 && nt != 768) { sind_set_error("sor_iterations: %d x %d tiles (%d threads) have no reciprocal-solver instance", EW, EH, nt); return SIND_E_ARG; }
    if (nt > SOR_NT || (C.fuse > 0 && 2 * 2 * C.fuse >= std::min(EW, EH))) { sind_set_error("sor_iterations: tile %d x %d / %d fused iterations not supported", EW, EH, C.fuse); return SIND_E_ARG; }
    const size_t shm = sor_lds_bytes(EW, nt);
    if (shm > 150 * 1024) { sind_set_error("sor_iterations: %d x %d tiles need %zu bytes of LDS", EW, EH, shm); return SIND_E_ARG; }
    const std::vector<int> plan = sor_fuse_plan(C, w, h, EW, EH, total);
    size_t step = 0;
    for (int done = 0; done < total;) {
        const int k = std::min(plan[step++], total - done), halo = 2 * k, halo_x = halo, IW = EW - 2 * halo_x, IH = EH - 2 * halo;
        const int ntx = divup(w, IW), nty = divup(h, IH);
        const bool t64 = EW == 64 && EH == 64 && nt == 512;           // the default tile has instances with compile-time sizes
#ifdef SIND_LAB
        auto kern = (C.mode == 4 || C.mode == 5) ? (t64 ? k_sor_fused<2, 512, 4, 64, 64> : k_sor_fused<2, 1024, 4, 0, 0>) : !rcp ? (t64 ? k_sor_fused<0, 512, 4, 64, 64> : k_sor_fused<0, 1024, 4, 0, 0>)
                    : nt == 384 ? k_sor_fused<1, 384, 3, 0, 0> : nt == 768 ? k_sor_fused<1, 768, 3, 0, 0> : k_sor_fused<1, 256, 3, 0, 0>;
#else
        auto kern = t64 ? k_sor_fused<2, 512, 4, 64, 64> : k_sor_fused<2, 1024, 4, 0, 0>;
#endif
        static const int dry = sind_lab_env("SIND_SOR_DRY") ? atoi(sind_lab_env("SIND_SOR_DRY")) : 0;       // timing experiment: 1 = no iterations (prologue + write-back only; results are wrong)
        hipLaunchKernelGGL(kern, dim3(ntx * nty, B), dim3(nt), shm, s, w, h, EW, EH, IW, IH, halo_x, halo, ntx, dry ? 0 : k, C.xcd, omega, P.A11, P.A12, P.A22, P.b1, P.b2,
                           P.wgt, P.r11, P.r22, P.dWu, P.dWv, P.dWu2, P.dWv2);
        std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);
        done += k; *nlaunch += 1;
    }
    return SIND_OK;
}

// VariationalRefinement::calcUV on one pyramid level for B pairs.  Wu/Wv: initial flow in, refined flow out.
int varref_level(hipStream_t s, FlowPlanes& P, const float* I0, const float* I1, int w, int h, int B, const VarParams& V, SorTimer* timer, const SolverCfg& C, bool have_buffers, bool leave_increment) {
    const bool coarse_chain = (C.opts & FLOW_OPT_COARSE_CHAIN) != 0;
    if (coarse_chain && V.epsilon >= 1e-12f && C.mode != 0 && coarse_level_P(w, h)) {       // one workgroup's work: the whole level in one launch (flow_coarse.hip)
        const std::vector<std::pair<int, int>> lv{{w, h}}; const std::vector<size_t> off{0};
        return launch_coarse_chain(s, P, I0, I1, lv, off, 0, 0, B, V, false, false, 1.f, nullptr, nullptr);
    }
    const size_t n = (size_t)w * h * B;
    const dim3 blk(128);
    // have_buffers: the level transition (k_level_up) has left avg / Iz and a zero increment already; leave_increment: W += dW is the next transition's business
    if (!have_buffers) hipLaunchKernelGGL(k_warp_avg_iz, grid2d(w, divup(h, WARP_ROWS), B), blk, 0, s, I0, I1, P.Wu, P.Wv, P.avg, P.Iz, P.dWu, P.dWv, w, h);
    for (int it = 0; it < V.fixedPointIterations; it++) {
        if (C.coef_kernel) {
            const int tx = divup(w, KL_COLS), ty = divup(h, 4 * KL_ROWS), nt = tx * ty * B;
            hipLaunchKernelGGL(k_coef_lanes, dim3(divup(nt, 8) * 8), dim3(256), 0, s, V, w, h, P.avg, P.Iz, P.Wu, P.Wv,
                               P.dWu, P.dWv, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, C.mode == 3 ? P.r11 : nullptr, C.mode == 3 ? P.r22 : nullptr, C.coef_kernel == 2 ? 0 : 1, tx, ty, nt, C.coef_xcd);
        }
        else
            hipLaunchKernelGGL(k_coef, dim3(divup(w, 128), divup(h, KC_ROWS), B), blk, 0, s, V, w, h, P.avg, P.Iz, P.Wu, P.Wv,
                               P.dWu, P.dWv, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, C.mode == 3 ? P.r11 : nullptr, C.mode == 3 ? P.r22 : nullptr);
        if (timer) timer->begin(s);
        long long nlaunch = 0; int streamed = 0;
        SIND_TRY(sor_iterations(s, P, w, h, B, V.sorIterations, V.omega, &nlaunch, C, &streamed));
        // algorithmic bytes: 44 B per pixel per red+black iteration (9 reads + 2 writes of f32), SURVEY.md §8d
        if (timer) timer->end(s, nlaunch, 44.0 * (double)w * h * B * V.sorIterations, streamed ? 0 : 1);
    }
    if (leave_increment) { HIP_TRY(hipGetLastError()); return SIND_OK; }
    if ((((uintptr_t)P.Wu | (uintptr_t)P.Wv | (uintptr_t)P.dWu | (uintptr_t)P.dWv) & 15) == 0)
        hipLaunchKernelGGL(k_add_flow, dim3((unsigned)(((n >> 2) + (n & 3) + 255) / 256)), dim3(256), 0, s, P.Wu, P.Wv, P.dWu, P.dWv, P.Wu, P.Wv, n);      // W = W + dW, in place
    else
        hipLaunchKernelGGL(k_add_flow_1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P.Wu, P.Wv, P.dWu, P.dWv, P.Wu, P.Wv, n);
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}

int launch_mag_stats(hipStream_t s, const float* u, const float* v, float* mag, unsigned* maxbits, int* hist, uint8_t* out_u8, int n, int B) {
    HIP_TRY(hipMemsetAsync(maxbits, 0, B * sizeof(unsigned), s));
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t)B * 256 * sizeof(int), s));
    const int gx = std::min(divup(n, 256), 64);
    hipLaunchKernelGGL(k_mag_max, dim3(gx, B), dim3(256), 0, s, u, v, mag, maxbits, n);
    hipLaunchKernelGGL(k_mag_hist, dim3(gx, B), dim3(256), 0, s, mag, maxbits, hist, out_u8, n);
    return SIND_OK;
}
int launch_residual(hipStream_t s, const float* u, const float* v, const double H[9], float* mag, unsigned* maxbits, int* hist, uint8_t* magu8, int w, int h, bool already_zero) {
    HMat Hm; for (int i = 0; i < 9; i++) Hm.h[i] = H[i];
    if (already_zero) {}                                  // the consumer of the previous frame (k_flow_thresholds) left the block zeroed
    else if ((const void*)maxbits == (const void*)(hist + 256)) HIP_TRY(hipMemsetAsync(hist, 0, 257 * sizeof(int), s));       // histogram and maximum in one block: one fill
    else { HIP_TRY(hipMemsetAsync(maxbits, 0, sizeof(unsigned), s)); HIP_TRY(hipMemsetAsync(hist, 0, 256 * sizeof(int), s)); }
    hipLaunchKernelGGL(k_residual_mag, dim3(divup(w, 128), divup(h, RM_ROWS)), dim3(128), 0, s, u, v, Hm, mag, maxbits, w, h);
    const int n = w * h, gx = std::min(divup(n, 256), 64);
    hipLaunchKernelGGL(k_mag_hist, dim3(gx, 1), dim3(256), 0, s, mag, maxbits, hist, magu8, n);
    return SIND_OK;
}
int launch_threshold_masks(hipStream_t s, const uint8_t* magu8, float lo, float hi, uint8_t* low, uint8_t* high, int n) {
    hipLaunchKernelGGL(k_threshold_masks, dim3(divup(n, 256)), dim3(256), 0, s, magu8, lo, hi, low, high, n);
    return SIND_OK;
}
int launch_flow_thresholds_and_masks(hipStream_t s, int* hist, int W, int H, int* res, const uint8_t* magu8, uint8_t* low, uint8_t* high) {
    hipLaunchKernelGGL(k_flow_thresholds, dim3(1), dim3(64), 0, s, hist, W, H, res, (double*)nullptr, 0);
    const int wide = ((((uintptr_t)magu8 | (uintptr_t)low | (uintptr_t)high) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_threshold_masks_dev, dim3(divup(wide ? divup(W * H, 16) : W * H, 256)), dim3(256), 0, s, magu8, reinterpret_cast<const float*>(res + 257), low, high, W * H, wide);
    return SIND_OK;
}
// test entry: n histograms (257 words each, device) through the one-wave kernel (variant 1; mu1 chain optionally dumped) or the serial reference (variant 0)
int debug_flow_thresholds(hipStream_t s, int* hist, int n, int W, int H, int variant, int* res, double* mu1) {
    if (variant == 0) hipLaunchKernelGGL(k_flow_thresholds_serial, dim3(n), dim3(64), 0, s, hist, W, H, reinterpret_cast<float*>(res));
    else hipLaunchKernelGGL(k_flow_thresholds, dim3(n), dim3(64), 0, s, hist, W, H, res, mu1, variant == 2 ? 0 : 1);
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}
int launch_gather_grid(hipStream_t s, const float* u, const float* v, float* out, int w, int h, int step, int B) {
    const int cnt = ((w - 1) / step) * ((h - 1) / step);
    hipLaunchKernelGGL(k_gather_grid, dim3(divup(cnt, 256), B), dim3(256), 0, s, u, v, out, w, h, step);
    return SIND_OK;
}
int launch_scale2(hipStream_t s, float* a, float* b, float sc, size_t n) {
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, sc, n);
    return SIND_OK;
}

}  // namespace sind
