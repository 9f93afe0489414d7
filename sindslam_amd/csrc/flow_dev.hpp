// Device-side pieces of the dense-flow path that more than one translation unit needs (flow_kernels.hip: the per-stage kernels; flow_coarse.hip: the one-launch
// chain over the one-workgroup pyramid levels).  Every function here is the ONE statement of its arithmetic: both units evaluate the same float operations in the
// same order (-ffp-contract=off), which is what keeps them bit-identical to each other and to the oracle.
#pragma once
#include "common.hpp"
#include "flow.hpp"

namespace sind {

// One output pixel of cv::resize(INTER_LINEAR, CV_32F): S = source image, (dx, dy) = destination pixel.
__device__ __forceinline__ float resize_px(const float* __restrict__ S, int sw, int sh, int dx, int dy, double scale_x, double scale_y) {
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
    return r0 * b0 + r1 * b1;
}

// VariationalRefinementImpl::prepareBuffers, one pixel: I1 warped by the level's initial flow (cv::remap, INTER_LINEAR, BORDER_REPLICATE, coordinates quantised to 1/32 px),
// then the averaged image and the temporal difference.  S = the second image of the pair (w x h), i0 = the first image's pixel.
__device__ __forceinline__ void warp_px(const float* __restrict__ S, float i0, float fu, float fv, int x, int y, int w, int h, float& avg, float& iz) {
    const float mx = x + fu, my = y + fv;
    int sx = d_cvRound(mx * 32.f), sy = d_cvRound(my * 32.f);
    const int fxq = sx & 31, fyq = sy & 31;
    sx >>= 5; sy >>= 5;
    sx = max(-32768, min(32767, sx)); sy = max(-32768, min(32767, sy));
    const float tx1 = fxq * (1.f / 32), tx0 = 1.f - tx1, ty1 = fyq * (1.f / 32), ty0 = 1.f - ty1;
    const float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
    const int x0 = d_clip(sx, 0, w), x1 = d_clip(sx + 1, 0, w), y0 = d_clip(sy, 0, h), y1 = d_clip(sy + 1, 0, h);
    const float v0 = S[y0 * w + x0], v1 = S[y0 * w + x1], v2 = S[y1 * w + x0], v3 = S[y1 * w + x1];
    const float wv = v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
    avg = i0 * 0.5f + wv * 0.5f;
    iz = wv - i0;
}

// Divisions that share a divisor (the three gradient norms of a pixel divide 14 numerators) go through the divisor's reciprocal: hardware estimate +
// one Newton step = exactly RN(1 / a) for every float significand, then Markstein's correction = the correctly rounded quotient (sor_rcp / sor_div
// below the solver kernels; checked exhaustively by sind_debug_rcp_scan) -- 3 instructions per quotient instead of the 10 of an IEEE division.
__device__ __forceinline__ float kc_rcp(float a) { const float y0 = __builtin_amdgcn_rcpf(a); const float e = fmaf(-a, y0, 1.f); return fmaf(e, y0, y0); }
__device__ __forceinline__ float kc_div(float n, float a, float r) { const float q0 = n * r; const float e = fmaf(-a, q0, n); return fmaf(e, r, q0); }
// RN(sqrt(x)) for 2^-96 <= x < inf: the compiler's own correctly rounded sequence (hardware estimate within 1 ulp, then the neighbour whose residual says so) without its
// guards for tiny, zero and infinite arguments -- 8 instructions instead of 15.  c / sqrt(x) then goes through the root's reciprocal like every other quotient of the kernel
// (kc_rcp + kc_div: 6 instructions instead of the IEEE division's 11).  sind_debug_coef_math_scan checks both against sqrtf and the IEEE division for EVERY float in a range
// of binary exponents (tests/test_flow_gpu.py); the arguments here are >= epsilon^2 = 1e-6 and the roots lie in [1e-3, ~1e4].
__device__ __forceinline__ float kc_sqrt(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (0.f >= rd) ? sd : s;
    r = (0.f < ru) ? su : r;
    return r;
}
template <bool FAST> __device__ __forceinline__ float kc_over_sqrt(float c, float x) {
    if (FAST) { const float s = kc_sqrt(x); return kc_div(c, s, kc_rcp(s)); }
    return c / sqrtf(x);
}

// ---- k_coef_lanes: the same coefficients, neighbours taken from the neighbouring LANES instead of from memory ----------------------------------------------------------------------
// k_coef above is bound by the texture-address path, not by arithmetic: 49 dword loads + 8 stores per pixel are 57 x 4 = 228 address cycles per 64-pixel wave row against ~94 CU
// cycles of VALU work (376 instructions per pixel over four SIMDs): 170 pairs of 384 x 288 -> 94 us predicted, 85 - 91 us measured (profiles/r04/k_coef.txt).  Here a wave owns
// 64 consecutive columns of KL_ROWS rows and loads each plane's rows ONCE (one dword per lane and row: 8 rows of the average, 6 of Iz and of each of the four flow planes for four
// output rows: 9.5 loads per pixel); x +- 1 and x +- 2 come from the lanes to the left and right with whole-wave DPP shifts (gfx9 wave_shl / wave_shr, VALU rate), y +- 1 and
// y +- 2 from the rows held in registers.  The two lanes at either end only feed their neighbours (60 of 64 lanes store).  Values and the order of every float operation are
// those of kc_rows -- except that sqrt and c / sqrt take their short correctly rounded forms (kc_sqrt / kc_over_sqrt above: the same VALUES for every float) --: the results are
// bit-identical (tests/test_flow_gpu.py compares the kernels, with and without the short forms, and all of them with the oracle).
#define KL_ROWS 4
#define KL_COLS 60
__device__ __forceinline__ float lane_next(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false)); }   // wave_shl:1 -- the value of lane + 1
__device__ __forceinline__ float lane_prev(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false)); }   // wave_shr:1 -- the value of lane - 1
// BORDER: the wave touches the image border (columns or rows clamp; BORDER_REPLICATE applies to the DERIVATIVE images too, hence the explicit cases below)
template <bool BORDER, bool FASTM>
__device__ __forceinline__ void kc_lanes(const VarParams& P, int w, int h, int col, int y0, size_t base, const float* __restrict__ gAvg, const float* __restrict__ gIz, const float* __restrict__ gWu,
                       const float* __restrict__ gWv, const float* __restrict__ gdWu, const float* __restrict__ gdWv, float* __restrict__ A11, float* __restrict__ A12,
                       float* __restrict__ A22, float* __restrict__ B1, float* __restrict__ B2, float* __restrict__ Wgt, float* __restrict__ R11, float* __restrict__ R22, bool store_lane) {
    const float zeta2 = P.zeta * P.zeta, eps2 = P.epsilon * P.epsilon, gamma2 = P.gamma / 2, delta2 = P.delta / 2, alpha2 = P.alpha / 2;
    const unsigned xc = BORDER ? (unsigned)min(max(col, 0), w - 1) : (unsigned)col;          // the column this lane loads (replicated outside the image)
    auto row_of = [&](int yy) { return (unsigned)(BORDER ? min(max(yy, 0), h - 1) : yy) * (unsigned)w * 4u; };   // byte offset of a row in its plane; wave-uniform: scalar arithmetic
    // rows held in registers: a[j] = average at row y0 - 2 + j; z[j], wu[j], ... at row y0 - 1 + j
    float a[KL_ROWS + 4], z[KL_ROWS + 2], wu[KL_ROWS + 2], wv[KL_ROWS + 2], du[KL_ROWS + 2], dv[KL_ROWS + 2];
    // one buffer descriptor per plane of this pair (scalar registers), the row as the scalar offset, this lane's column as the 32-bit vector offset: addressing without vector
    // arithmetic (as flat loads the 38 + 24 addresses are 64-bit VALU adds, ~30 of 270 instructions per pixel)
    const unsigned xoff = xc * 4u, plane_bytes = (unsigned)w * h * 4u;
    auto rs = [&](const float* plane) { return __builtin_amdgcn_make_buffer_rsrc((void*)(plane + base), 0, plane_bytes, 0x00020000); };
    const auto rA = rs(gAvg), rZ = rs(gIz), rWu = rs(gWu), rWv = rs(gWv), rDu = rs(gdWu), rDv = rs(gdWv);
    const auto oA11 = rs(A11), oA12 = rs(A12), oA22 = rs(A22), oB1 = rs(B1), oB2 = rs(B2), oW = rs(Wgt);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, unsigned row_bytes) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, xoff, row_bytes, 0)); };
    #pragma unroll
    for (int j = 0; j < KL_ROWS + 4; j++) a[j] = ld(rA, row_of(y0 - 2 + j));
    #pragma unroll
    for (int j = 0; j < KL_ROWS + 2; j++) {
        const unsigned ro = row_of(y0 - 1 + j);
        z[j] = ld(rZ, ro); wu[j] = ld(rWu, ro); wv[j] = ld(rWv, ro); du[j] = ld(rDu, ro); dv[j] = ld(rDv, ro);
    }
    // x derivative of the average on the rows y0 - 1 .. y0 + KL_ROWS (dxa[j] at row y0 - 1 + j), tempW = W + dW on the same rows
    float dxa[KL_ROWS + 2], tu[KL_ROWS + 2], tv[KL_ROWS + 2];
    #pragma unroll
    for (int j = 0; j < KL_ROWS + 2; j++) { dxa[j] = lane_next(a[j + 1]) - lane_prev(a[j + 1]); tu[j] = wu[j] + du[j]; tv[j] = wv[j] + dv[j]; }
    // own weight of (row y0 - 1 + j, this column); wgt_row(0) only serves as the upper neighbour of the first row
    auto wgt_row = [&](int j) {
        const float c_u = tu[j], c_v = tv[j];
        const float ux = lane_next(c_u) - c_u, vx = lane_next(c_v) - c_v;
        const float uy = tu[j + 1] - c_u, vy = tv[j + 1] - c_v;
        return kc_over_sqrt<FASTM>(alpha2, ux * ux + vx * vx + uy * uy + vy * vy + eps2);
    };
    float w_up = wgt_row(0);
    #pragma unroll
    for (int r = 0; r < KL_ROWS; r++) {
        const int y = y0 + r;
        if (BORDER && y >= h) break;                        // wave-uniform
        const float Ix = dxa[r + 1], Iy = a[r + 3] - a[r + 1], Iz = z[r + 1];
        const float Ixz = lane_next(z[r + 1]) - lane_prev(z[r + 1]), Iyz = z[r + 2] - z[r];
        // second derivatives = the first-derivative images differenced again, each replicated at ITS border
        const float dx_next = lane_next(dxa[r + 1]), dx_prev = lane_prev(dxa[r + 1]);      // every lane takes part in a shift: never inside a per-lane condition
        const float dx_hi = (BORDER && col == w - 1) ? dxa[r + 1] : dx_next, dx_lo = (BORDER && col == 0) ? dxa[r + 1] : dx_prev;
        const float Ixx = dx_hi - dx_lo;
        const float Ixy = dxa[r + 2] - dxa[r];
        const float dy_hi = (BORDER && y == h - 1) ? Iy : a[r + 4] - a[r + 2], dy_lo = (BORDER && y == 0) ? Iy : a[r + 2] - a[r];
        const float Iyy = dy_hi - dy_lo;
        const float dU = du[r + 1], dV = dv[r + 1];
        float derivNorm = Ix * Ix + Iy * Iy + zeta2;
        const float Ik1z = Iz + Ix * dU + Iy * dV;
        const float rN0 = kc_rcp(derivNorm);
        float weight = kc_div(kc_over_sqrt<FASTM>(delta2, kc_div(Ik1z * Ik1z, derivNorm, rN0) + eps2), derivNorm, rN0);
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
        weight = kc_over_sqrt<FASTM>(gamma2, D1(Ik1zx * Ik1zx) + D2(Ik1zy * Ik1zy) + eps2);
        a11 += weight * (D1(Ixx * Ixx) + D2(Ixy * Ixy));
        a12 += weight * (D1(Ixx * Ixy) + D2(Ixy * Iyy));
        a22 += weight * (D1(Ixy * Ixy) + D2(Iyy * Iyy));
        b1 += -weight * (D1(Ixx * Ixz) + D2(Ixy * Iyz));
        b2 += -weight * (D1(Ixy * Ixz) + D2(Iyy * Iyz));
        #undef D1
        #undef D2

        const float wp = wgt_row(r + 1);
        const float wl = lane_prev(wp), wq = w_up;
        const float wu_c = wu[r + 1], wv_c = wv[r + 1];
        const float wu_r = lane_next(wu_c), wu_l = lane_prev(wu_c), wv_r = lane_next(wv_c), wv_l = lane_prev(wv_c);
        const bool red = ((col + y) & 1) == 0;
        // the four link updates (no-ops at the image border)
        #define OWN_H() if (!BORDER || col < w - 1) { b1 += wp * (wu_r - wu_c); a11 += wp; b2 += wp * (wv_r - wv_c); a22 += wp; }
        #define LEFT_H() if (!BORDER || col > 0) { b1 -= wl * (wu_c - wu_l); a11 += wl; b2 -= wl * (wv_c - wv_l); a22 += wl; }
        #define OWN_V() if (!BORDER || y < h - 1) { b1 += wp * (wu[r + 2] - wu_c); a11 += wp; b2 += wp * (wv[r + 2] - wv_c); a22 += wp; }
        #define UP_V() if (!BORDER || y > 0) { b1 -= wq * (wu_c - wu[r]); a11 += wq; b2 -= wq * (wv_c - wv[r]); a22 += wq; }
        if (red) { OWN_H() LEFT_H() OWN_V() UP_V() }
        else     { LEFT_H() OWN_H() UP_V() OWN_V() }
        #undef OWN_H
        #undef LEFT_H
        #undef OWN_V
        #undef UP_V
        if (store_lane) {
            const unsigned ro = (unsigned)y * (unsigned)w * 4u;
            auto st = [&](const __amdgpu_buffer_rsrc_t& r, float val) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(val), r, xoff, ro, 0); };
            st(oA11, a11); st(oA12, a12); st(oA22, a22); st(oB1, b1); st(oB2, b2); st(oW, wp);
            if (R11) { st(rs(R11), 1.f / a11); st(rs(R22), 1.f / a22); }
        }
        w_up = wp;
    }
}

struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };     // 4-byte aligned 16-byte load (gfx950 allows unaligned dwordx4)
__device__ __forceinline__ float sor_div(float n, float a, float r) { const float q0 = n * r; const float e = fmaf(-a, q0, n); return fmaf(e, r, q0); }
// RN(1 / a) without the IEEE division sequence: hardware reciprocal (<= 1 ulp) + one Newton step in FMA arithmetic.  sind_debug_rcp_scan checks it
// against the correctly rounded division for EVERY float significand (the step is invariant under scaling by powers of two while nothing is denormal).
// (volatile asm: the compiler must not hoist the loop-invariant reciprocals out of the solver loop, where they would cost 16 registers and spill;
// s_nop: the transcendental unit's result needs one wait state before a VALU read, which the hazard pass cannot see through inline asm)
__device__ __forceinline__ float sor_rcp(float a) {
    float y0; asm volatile("v_rcp_f32 %0, %1\n\ts_nop 0" : "=v"(y0) : "v"(a));
    const float e = fmaf(-a, y0, 1.f); return fmaf(e, y0, y0);
}

}  // namespace sind
