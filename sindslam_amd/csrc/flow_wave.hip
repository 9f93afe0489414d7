// k_sor_wave: the red-black SOR iterations of one launch (5 iterations = 10 half-sweeps) as a row pipeline in time that lives in ONE WAVE.
// (VariationalRefinementImpl::RedBlackSOR_ParBody of OpenCV's variational_refinement.cpp, the solver under the reference's DeepFlow call, DynaDetect.cc:1031 / 1075.)
//
// k_sor_stream (flow_kernels.hip) spreads the same pipeline over the 512 threads of a workgroup: a thread owns two rows of a four-pixel column strip, du / dv / weights
// travel between threads through LDS rings, and every step -- two half-sweeps of two rows, ~150 instructions -- ends in a workgroup barrier.  Measured there: 54 % of the wave
// cycles wait, the VALU issues 0.14 - 0.19 of its peak by counts.  Here nothing is shared between waves, so there is no barrier at all:
//   * a wave owns a column strip of 128 columns (lane l: the pixels ex0 + 2 l, ex0 + 2 l + 1) of a band of rows of one image and walks down it one row per step;
//   * step t runs half-sweep s on row t - s for s = 0 .. 9, in this order: row t - s has then seen half-sweep s - 1 of row t - s + 1 (earlier in the same step) and of row
//     t - s - 1 (a step ago), and not yet half-sweep s + 1 of row t - s - 1 (later in this step) -- exactly the values the sequential red pass / black pass order reads;
//   * a pixel is updated in half-sweep s iff (x + y + s) is even; ex0 is even, so EVERY stage of step t updates the lane's pixel t & 1: the 12-step unrolled loop body
//     has no per-lane colour select anywhere;
//   * the 10 rows in flight keep a11, a22, their reciprocals, the weight, du and dv in REGISTERS (a 12-slot window addressed by compile-time indices: slot = row mod 12, the
//     loop body is 12 steps); a12, b1, b2 -- read once per update -- stay in LDS (wave-private ring, 18 KB), so that the wave fits 256 registers and two waves share a SIMD
//     (one wave reaches half of the VALU's issue rate: profiles/r05/valu_rate.txt);
//   * horizontal neighbours across lanes come through whole-wave DPP shifts of the PRODUCT weight x value formed in the neighbour's lane (same two operands, same bits);
//     a lane outside the wave reads 0, which is the image border's value -- or a cut edge's: what a cut falsifies creeps inwards one pixel per half-sweep, so a strip / band
//     computes 10 columns / rows beyond what it keeps on every cut side;
//   * a row is requested PF = 2 steps before it enters the window, into a separate set of registers (16 per row in flight; a step is 1 - 2 us); reciprocals RN(1 / a) are
//     formed at the take-over (sor_rcp), 0 for pixels outside the image: their update returns exactly 0;
//   * consecutive workgroups go to the same XCD, so the strips of an image share their halo columns and seam cache lines in that XCD's L2.
// Same float operations on the same operands in the same order as SS_UPDATE / sor_pixel: bit-identical results (tests/test_flow_gpu.py, against the oracle).
// Where it stands: 227 us per launch of 512 pairs against 192 us with a tenth of the arithmetic (memory alone) and 171 us with every load from the cache
// (instructions alone); -DSWV_SKIP=n / -DSWV_ROWFIX build those two experiments (profiles/tools/wave_skip.sh).
#include "flow_dev.hpp"

namespace sind {

#define SWV_NS 10                    /* half-sweeps per launch */
#define SWV_NW 12                    /* register window: rows t + 1 .. t - 10 */
#define SWV_ROWF (2 * 3 * 64)        /* floats of one LDS row: [pixel 0 / 1][a12, b1, b2][lane] */
#define SWV_HALO 10

struct SwvRow { float a11[2], a22[2], r11[2], r22[2], wp[2], du[2], dv[2]; };
typedef unsigned swv_u2 __attribute__((ext_vector_type(2)));

// half-sweep of pixel I of row R (U: the row above, D: the row below); a12, b1, b2: the pixel's entries of the row's LDS record
template <int I>
__device__ __forceinline__ void swv_update(SwvRow& R, const SwvRow& U, const SwvRow& D, float a12, float b1, float b2, float omega) {
    float lu, lv, ru, rv;            // weight x value of the left and right neighbours
    if (I == 0) { lu = lane_prev(R.wp[1] * R.du[1]); lv = lane_prev(R.wp[1] * R.dv[1]); ru = R.wp[0] * R.du[1]; rv = R.wp[0] * R.dv[1]; }
    else        { lu = R.wp[0] * R.du[0]; lv = R.wp[0] * R.dv[0]; ru = R.wp[1] * lane_next(R.du[0]); rv = R.wp[1] * lane_next(R.dv[0]); }
    const float sigmaU = lu + ru + U.wp[I] * U.du[I] + R.wp[I] * D.du[I];
    const float sigmaV = lv + rv + U.wp[I] * U.dv[I] + R.wp[I] * D.dv[I];
    float nu = R.du[I], nv = R.dv[I];
    nu += omega * (sor_div(sigmaU + b1 - nv * a12, R.a11[I], R.r11[I]) - nu);
    nv += omega * (sor_div(sigmaV + b2 - nu * a12, R.a22[I], R.r22[I]) - nv);
    R.du[I] = nu; R.dv[I] = nv;
}

extern __shared__ float swv_lds[];
struct SwvFlight { swv_u2 a11, a12, a22, b1, b2, w, u, v; };          // the row in flight (requested a step before it is taken over)
// buffer descriptors of this wave's image (scalar registers; an access is descriptor + scalar row offset + the lane's constant column offset), lane constants
struct SwvCtx {
    __amdgpu_buffer_rsrc_t pA11, pA12, pA22, pB1, pB2, pW, pU, pV, oU, oV;
    float* lbase; float omega; unsigned xoff, row_bytes; int ys; unsigned nrows; int by0, by1; bool in0, in1, keep0, keep1;
};
typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t(), 0, 0, 0)) swv_b64;

// (a12, b1, b2 of stage S are read from LDS one stage ahead -- during stage S - 1, or before the take-over for stage 0)
struct SwvRec { float a12, b1, b2; };
template <int K, int S>
__device__ __forceinline__ SwvRec swv_record(const SwvCtx& C) {
    constexpr int NW = SWV_NW, sl = (K - S + 2 * NW) % NW;
    const float* lrow = C.lbase + sl * SWV_ROWF + (K & 1) * 192;
    return SwvRec{lrow[0], lrow[64], lrow[128]};
}
// half-sweeps S .. 9 of step K of the loop body (t = t0 + K): every window index is a compile-time constant.  Every stage tests its row against the band (scalar
// compare and branch).  An instance without the tests for the steady state -- twelve steps in one basic block -- was measured: same time (the kernel waits for memory, see
// DESIGN.md), twice the code.
template <int K, int S>
__device__ __forceinline__ void swv_stages(SwvRow (&win)[SWV_NW], const SwvCtx& C, int t, SwvRec rec) {
    if constexpr (S < SWV_NS) {
        const int y = t - S;
        SwvRec next{};
        if constexpr (S + 1 < SWV_NS) next = swv_record<K, S + 1>(C);
        if ((unsigned)(y - C.ys) < C.nrows) {
            constexpr int NW = SWV_NW, sl = (K - S + 2 * NW) % NW, up = (sl + NW - 1) % NW, dn = (sl + 1) % NW;
#ifdef SWV_SKIP       /* timing experiment (wrong results): only the first SWV_SKIP half-sweeps compute; loads, stores and the pipeline stay */
            if (S < SWV_SKIP)
#endif
            swv_update<K & 1>(win[sl], win[up], win[dn], rec.a12, rec.b1, rec.b2, C.omega);
            if (S == SWV_NS - 1 && y >= C.by0 && y < C.by1) {                   // the row is through: its kept pixels go to memory
                const unsigned ro = (unsigned)y * C.row_bytes;
                const SwvRow& R = win[sl];
                if (C.keep1) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(swv_b64, swv_u2{__float_as_uint(R.du[0]), __float_as_uint(R.du[1])}), C.oU, C.xoff, ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(swv_b64, swv_u2{__float_as_uint(R.dv[0]), __float_as_uint(R.dv[1])}), C.oV, C.xoff, ro, 0);
                } else if (C.keep0) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(R.du[0]), C.oU, C.xoff, ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(R.dv[0]), C.oV, C.xoff, ro, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);              // (the next stage's LDS reads stay where they were issued: a stage ahead of their use)
        swv_stages<K, S + 1>(win, C, t, next);
    }
}
// steps K .. 11 of the loop body
template <int K, int PF>
__device__ __forceinline__ void swv_steps(SwvRow (&win)[SWV_NW], SwvFlight (&FL)[PF], const SwvCtx& C, int t0) {
    if constexpr (K < SWV_NW) {
        const int t = t0 + K;
        const SwvRec rec0 = swv_record<K, 0>(C);                            // (row t's record was written a step ago)
        SwvFlight& F = FL[K % PF];                                          // requested PF steps ago: row t + 1
        // (1) the row requested PF steps ago becomes row t + 1 of the window
        {
            SwvRow& N = win[(K + 1) % SWV_NW];
            const bool ok = (unsigned)(t + 1 - C.ys) < C.nrows;   // (wave-uniform) rows outside the band are rows of zeros
            const bool m0 = ok && C.in0, m1 = ok && C.in1;
            N.du[0] = m0 ? __uint_as_float(F.u.x) : 0.f; N.du[1] = m1 ? __uint_as_float(F.u.y) : 0.f;
            N.dv[0] = m0 ? __uint_as_float(F.v.x) : 0.f; N.dv[1] = m1 ? __uint_as_float(F.v.y) : 0.f;
            N.a11[0] = __uint_as_float(F.a11.x); N.a11[1] = __uint_as_float(F.a11.y); N.a22[0] = __uint_as_float(F.a22.x); N.a22[1] = __uint_as_float(F.a22.y);
            N.wp[0] = __uint_as_float(F.w.x); N.wp[1] = __uint_as_float(F.w.y);
            const float q0 = sor_rcp(N.a11[0]), q1 = sor_rcp(N.a11[1]), q2 = sor_rcp(N.a22[0]), q3 = sor_rcp(N.a22[1]);
            N.r11[0] = m0 ? q0 : 0.f; N.r11[1] = m1 ? q1 : 0.f; N.r22[0] = m0 ? q2 : 0.f; N.r22[1] = m1 ? q3 : 0.f;
            float* lr = C.lbase + ((K + 1) % SWV_NW) * SWV_ROWF;
            lr[0] = __uint_as_float(F.a12.x); lr[64] = __uint_as_float(F.b1.x); lr[128] = __uint_as_float(F.b2.x);
            lr[192] = __uint_as_float(F.a12.y); lr[256] = __uint_as_float(F.b1.y); lr[320] = __uint_as_float(F.b2.y);
        }
        // (2) request row t + 1 + PF into the registers just freed
        if ((unsigned)(t + 1 + PF - C.ys) < C.nrows && C.in0) {  // (lanes right of the image load nothing: their pixels stay zero)
#ifdef SWV_ROWFIX      /* timing experiment (wrong results): every request reads the band's first row -- cache hits: what is left is the instruction stream's own time */
            const unsigned ro = (unsigned)C.ys * C.row_bytes;
#else
            const unsigned ro = (unsigned)(t + 1 + PF) * C.row_bytes;
#endif
            auto ld = [&](const __amdgpu_buffer_rsrc_t& r) { return __builtin_bit_cast(swv_u2, __builtin_amdgcn_raw_buffer_load_b64(r, C.xoff, ro, 0)); };
            F.u = ld(C.pU); F.v = ld(C.pV); F.a11 = ld(C.pA11); F.a22 = ld(C.pA22); F.w = ld(C.pW); F.a12 = ld(C.pA12); F.b1 = ld(C.pB1); F.b2 = ld(C.pB2);
        }
        // (3) the ten half-sweeps of this step
        __builtin_amdgcn_sched_barrier(0);
        swv_stages<K, 0>(win, C, t, rec0);
        swv_steps<K + 1, PF>(win, FL, C, t0);
    }
}

// PF: rows in flight (a row is requested PF steps before it is taken over; 16 registers each)
template <int PF>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_sor_wave(int strips, int bands, int items, int w, int h, int IW, int BH, float omega, const float* __restrict__ gA11, const float* __restrict__ gA12, const float* __restrict__ gA22,
           const float* __restrict__ gB1, const float* __restrict__ gB2, const float* __restrict__ gW, const float* __restrict__ gU, const float* __restrict__ gV,
           float* __restrict__ gUo, float* __restrict__ gVo) {
    const int lane = threadIdx.x;
    // consecutive items -- the column strips of one band of one image, which share their halo columns and the cache lines at their seams -- go to the SAME XCD (workgroups are
    // handed to the eight XCDs round-robin): the second reader of a line finds it in that XCD's L2
    const int per_xcd = (int)gridDim.x >> 3, item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (item >= items) return;
    const int strip = item % strips, rest = item / strips, band = rest % bands, image = rest / bands;
    // columns: keeps [ix0, ix1), works on [ex0, ex0 + 128); rows: keeps [by0, by1), works on [ys, ye)
    const int ix0 = strip * IW, ix1 = min(ix0 + IW, w), ex0 = max(ix0 - SWV_HALO, 0);
    const int by0 = band * BH, by1 = min(by0 + BH, h), ys = max(by0 - SWV_HALO, 0), ye = min(by1 + SWV_HALO, h);
    const unsigned nrows = (unsigned)(ye - ys);
    const int x0 = ex0 + 2 * lane;
    const bool in0 = x0 < w, in1 = x0 + 1 < w;
    const bool keep0 = x0 >= ix0 && x0 < ix1, keep1 = x0 + 1 >= ix0 && x0 + 1 < ix1;
    const size_t base = (size_t)image * w * h;
    const unsigned plane_bytes = (unsigned)w * (unsigned)h * 4u;
    auto rs = [&](const float* plane) { return __builtin_amdgcn_make_buffer_rsrc((void*)(plane + base), 0, plane_bytes, 0x00020000); };
    const SwvCtx C{rs(gA11), rs(gA12), rs(gA22), rs(gB1), rs(gB2), rs(gW), rs(gU), rs(gV), rs(gUo), rs(gVo),
                   swv_lds + lane, omega, (unsigned)x0 * 4u, (unsigned)w * 4u, ys, nrows, by0, by1, in0, in1, keep0, keep1};
    SwvRow win[SWV_NW] = {};
    SwvFlight FL[PF] = {};
    const int Tb = (ys - 1 - PF) & ~1, Tend = ye + SWV_NS - 1;    // steps Tb .. Tend - 1 (Tb even: step K of the loop body updates pixel K & 1; row ys is requested in step ys - 1 - PF)
    for (int t0 = Tb; t0 < Tend; t0 += SWV_NW) swv_steps<0, PF>(win, FL, C, t0);
}

// Column strips and row bands of a w x h level.  One strip: w <= 128; two: the kept width + 10 <= 128; more: + 20.  The bands are the launch's parallelism knob: every cut
// costs 20 rows of recomputation, so there are only as many as it takes to give the chip `target_items` waves (8 per compute unit = 2048 fill it).
void sor_wave_layout(int w, int h, int B, int target_items, int force_bands, int* strips, int* IW, int* bands, int* BH) {
    int n = 1, iw = (w + 1) & ~1;
    if (w > 128) { for (n = 2;; n++) { iw = (divup(w, n) + 1) & ~1; if (iw + (n == 2 ? SWV_HALO : 2 * SWV_HALO) <= 128) break; } }
    int nb = force_bands > 0 ? force_bands : (int)std::min<long long>(std::max(1, h / 40), std::max<long long>(1, divup(target_items, n * std::max(B, 1))));
    nb = std::max(1, std::min(nb, h));
    const int bh = divup(h, nb);
    nb = divup(h, bh);
    *strips = n; *IW = iw; *bands = nb; *BH = bh;
}

int launch_sor_wave(hipStream_t s, FlowPlanes& P, int w, int h, int B, float omega, int target_items, int force_bands, int prefetch) {
    int strips, IW, bands, BH;
    sor_wave_layout(w, h, B, target_items, force_bands, &strips, &IW, &bands, &BH);
    if ((long long)w * h > (1ll << 28)) { sind_set_error("launch_sor_wave: %d x %d level", w, h); return SIND_E_ARG; }
    const size_t shm = (size_t)SWV_NW * SWV_ROWF * sizeof(float);
    const int items = strips * bands * B;
    auto kern = prefetch <= 1 ? k_sor_wave<1> : prefetch == 2 ? k_sor_wave<2> : k_sor_wave<3>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(divup(items, 8) * 8)), dim3(64), shm, s, strips, bands, items, w, h, IW, BH, omega, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, P.dWu, P.dWv, P.dWu2, P.dWv2);
    std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);       // strips and bands read each other's halo: not in place
    return SIND_OK;
}

}  // namespace sind
