// The one-workgroup levels of the DeepFlow pyramid in ONE launch (reference call structure: OpticalFlowDeepFlow::calc's level loop around VariationalRefinement::calcUV,
// opencv_contrib 4.2.0 deepflow.cpp / opencv 4.2.0 variational_refinement.cpp; call sites DynaDetect.cc:1031, 1075, 1127).
//
// A level of at most ~8 k pixels is one workgroup's work whatever the batch: as separate kernels it was 13 dependent launches (warp, 5 x (coefficients, 25 SOR iterations),
// W += dW, up-sampling) of 3 - 57 us each with ONE workgroup per image, and the 23 such levels of the 384 x 288 pyramid were a third of a DeepFlow's latency at every batch
// size (profiles/r05/flow_alone_by_batch.txt).  Here workgroup b walks image b through ALL of them, coarse to fine: every stage is a phase of the same kernel, phases are
// separated by workgroup barriers, the planes between phases stay in L2 / L1 (global memory written and re-read by the SAME workgroup: visible after s_waitcnt vmcnt(0) +
// s_barrier, the waves of a workgroup share their CU's vector L1).  Per-pixel arithmetic is that of the per-stage kernels (flow_dev.hpp holds the one statement of each), so
// the results are the same bits.
//
// The solver phase differs from k_sor_fused in its GRANULARITY, not in its arithmetic: 1024 threads whatever the level, a thread owns a 1 x 2P strip (P = 1, 2 or 4 pixels of
// each colour, the smallest that covers the level), keeps the strip's linear system, the reciprocals of A11 / A22 and all four neighbour weights in registers, and one
// half-sweep is P pixel updates per thread -- the half-sweep of a small level is latency (LDS read -> ~40 dependent float operations -> LDS write -> barrier), not throughput,
// so fewer pixels per thread and more waves per SIMD is what shortens it.
#include "common.hpp"
#include "flow.hpp"
#include "flow_dev.hpp"

namespace sind {

#define CL_NT 1024          /* threads of the chain kernel: 16 waves, 4 per SIMD, <= 128 VGPRs */
#define CL_PL 64            /* floats per plane and row in LDS: P guard floats, P * SW strip cells, one guard float */
#define CL_MAXLV 32

struct CoarseLevel {
    int w, h, P, SW, half, RS;          // size; pixels of one colour per thread; strips per row; threads of one row parity (whole waves); LDS row stride in floats
    int nw, nh;                         // size of the next finer level (the up-sampling target)
    unsigned long long off;             // offset of image 0 of this level inside the two pyramids (floats); image b follows at b * w * h
    double sx, sy;                      // cv::resize scales to the next finer level (inv of the size ratios, formed on the host as OpenCV does)
};
struct CoarseChain {
    int n, stride, init_zero, upsample_last; float post; int fp_iters, sor_iters;
    VarParams V; CoarseLevel L[CL_MAXLV];
};

typedef float cl_f2 __attribute__((ext_vector_type(2)));
typedef float cl_f4 __attribute__((ext_vector_type(4)));
template <int P> struct ClVec;
template <> struct ClVec<1> { typedef float T; };
template <> struct ClVec<2> { typedef cl_f2 T; };
template <> struct ClVec<4> { typedef cl_f4 T; };
template <int P> __device__ __forceinline__ void cl_unpack(const typename ClVec<P>::T& v, float (&d)[P]);
template <> __device__ __forceinline__ void cl_unpack<1>(const float& v, float (&d)[1]) { d[0] = v; }
template <> __device__ __forceinline__ void cl_unpack<2>(const cl_f2& v, float (&d)[2]) { d[0] = v.x; d[1] = v.y; }
template <> __device__ __forceinline__ void cl_unpack<4>(const cl_f4& v, float (&d)[4]) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
template <int P> __device__ __forceinline__ typename ClVec<P>::T cl_pack(const float* s);          // s[0], s[2], s[4], ... (every other strip pixel)
template <> __device__ __forceinline__ float cl_pack<1>(const float* s) { return s[0]; }
template <> __device__ __forceinline__ cl_f2 cl_pack<2>(const float* s) { return cl_f2{s[0], s[2]}; }
template <> __device__ __forceinline__ cl_f4 cl_pack<4>(const float* s) { return cl_f4{s[0], s[2], s[4], s[6]}; }

// global memory written by this workgroup becomes readable by all of its waves: every wave's stores have left (vmcnt), then the barrier
__device__ __forceinline__ void cl_sync_global() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void cl_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// `iters` red-black SOR iterations on one image's level (RedBlackSOR_ParBody; per-pixel arithmetic of k_sor_color / k_sor_fused).  du / dv live in LDS split by pixel parity:
// plane (q, c) -- q = (x + y) & 1, c = u | v -- holds pixel (y, x) at row (y + 1) * RS + (2 q + c) * CL_PL + P + x / 2, so the vertical neighbours of a strip's P pixels of
// one colour are ONE vector read per plane of the other colour, and a plane's row offset is a compile-time immediate.  Guard cells (row 0 and h + 1, the cell left of a
// row's first strip and right of its last) are zero: a neighbour outside the image contributes exactly 0, as in OpenCV's zero-bordered buffers.
template <int P>
__device__ __forceinline__ void coarse_sor(float* lds, int w, int h, int SW, int half, int RS, int iters, float omega, const float* __restrict__ gA11, const float* __restrict__ gA12,
                                           const float* __restrict__ gA22, const float* __restrict__ gB1, const float* __restrict__ gB2, const float* __restrict__ gW,
                                           float* __restrict__ gU, float* __restrict__ gV) {
    typedef typename ClVec<P>::T VT;
    constexpr int NP = 2 * P;
    const int tid = threadIdx.x;
    const bool lowh = tid < half; const int idx = lowh ? tid : tid - half;
    const int r2 = idx / SW, j = idx - r2 * SW, ly = 2 * r2 + (lowh ? 0 : 1), x0 = NP * j;
    const bool act = tid < 2 * half && ly < h;
    float a11[NP], a12[NP], a22[NP], b1[NP], b2[NP], wp[NP], wu[NP], r11[NP], r22[NP], du[NP], dv[NP];
    float wl0 = 0.f;
    #pragma unroll
    for (int i = 0; i < NP; i++) { a11[i] = 1.f; a12[i] = 0.f; a22[i] = 1.f; b1[i] = 0.f; b2[i] = 0.f; wp[i] = 0.f; wu[i] = 0.f; r11[i] = 0.f; r22[i] = 0.f; du[i] = 0.f; dv[i] = 0.f; }
    if (act) {
        const int g0 = ly * w + x0;
        #pragma unroll
        for (int i = 0; i < NP; i++) {
            if (x0 + i < w) {
                const int g = g0 + i;
                a11[i] = gA11[g]; a12[i] = gA12[g]; a22[i] = gA22[g]; b1[i] = gB1[g]; b2[i] = gB2[g]; wp[i] = gW[g]; du[i] = gU[g]; dv[i] = gV[g];
                if (ly > 0) wu[i] = gW[g - w];
            }
        }
        if (x0 > 0 && x0 - 1 < w) wl0 = gW[g0 - 1];
        // a pixel outside the image keeps the reciprocal 0: its update returns exactly 0 (q0 = n * 0, e = n, q = fma(n, 0, 0)), no select in the loop
        #pragma unroll
        for (int i = 0; i < NP; i++) if (x0 + i < w) { r11[i] = sor_rcp(a11[i]); r22[i] = sor_rcp(a22[i]); }
    }
    const int s0 = ly & 1;
    // this thread's cells: row (ly + 1), column offset P + P * j in each plane; rU / rD: the rows above / below
    float* const rM = lds + (ly + 1) * RS + P + P * j; float* const rU = rM - RS; float* const rD = rM + RS;
    if (act) {
        // strip pixels i with (i + ly) even are of parity 0
        *reinterpret_cast<VT*>(rM + (2 * s0 + 0) * CL_PL) = cl_pack<P>(du); *reinterpret_cast<VT*>(rM + (2 * s0 + 1) * CL_PL) = cl_pack<P>(dv);
        *reinterpret_cast<VT*>(rM + (2 * (s0 ^ 1) + 0) * CL_PL) = cl_pack<P>(du + 1); *reinterpret_cast<VT*>(rM + (2 * (s0 ^ 1) + 1) * CL_PL) = cl_pack<P>(dv + 1);
    }
    cl_sync_lds();
    // one half-sweep over the strip pixels START, START + 2, ...; Q = the parity being updated, its vertical and strip-edge neighbours are of parity Q ^ 1
    #define CL_HALF(START, Q)                                                                                                     \
        {                                                                                                                         \
            constexpr int oq = (Q) ^ 1;                                                                                           \
            float uu[P], vu[P], ud[P], vd[P];                                                                                     \
            cl_unpack<P>(*reinterpret_cast<const VT*>(rU + (2 * oq + 0) * CL_PL), uu); cl_unpack<P>(*reinterpret_cast<const VT*>(rU + (2 * oq + 1) * CL_PL), vu); \
            cl_unpack<P>(*reinterpret_cast<const VT*>(rD + (2 * oq + 0) * CL_PL), ud); cl_unpack<P>(*reinterpret_cast<const VT*>(rD + (2 * oq + 1) * CL_PL), vd); \
            /* strip-edge horizontal neighbour: left of pixel 0 (START == 0) or right of pixel 2P - 1 (START == 1) */              \
            const float eu = rM[(2 * oq + 0) * CL_PL + ((START) == 0 ? -1 : P)], ev = rM[(2 * oq + 1) * CL_PL + ((START) == 0 ? -1 : P)]; \
            _Pragma("unroll")                                                                                                     \
            for (int k = 0; k < P; k++) {                                                                                         \
                const int i = (START) + 2 * k;                                                                                    \
                const float wl = i == 0 ? wl0 : wp[i == 0 ? 0 : i - 1];                                                           \
                const float ul = i == 0 ? eu : du[i == 0 ? 0 : i - 1], vl = i == 0 ? ev : dv[i == 0 ? 0 : i - 1];                \
                const float ur = i == NP - 1 ? eu : du[i == NP - 1 ? NP - 1 : i + 1], vr = i == NP - 1 ? ev : dv[i == NP - 1 ? NP - 1 : i + 1]; \
                const float sigmaU = wl * ul + wp[i] * ur + wu[i] * uu[k] + wp[i] * ud[k];                                        \
                const float sigmaV = wl * vl + wp[i] * vr + wu[i] * vu[k] + wp[i] * vd[k];                                        \
                float nu = du[i], nv = dv[i];                                                                                     \
                nu += omega * (sor_div(sigmaU + b1[i] - nv * a12[i], a11[i], r11[i]) - nu);                                       \
                nv += omega * (sor_div(sigmaV + b2[i] - nu * a12[i], a22[i], r22[i]) - nv);                                       \
                du[i] = nu; dv[i] = nv;                                                                                           \
            }                                                                                                                     \
            *reinterpret_cast<VT*>(rM + (2 * (Q) + 0) * CL_PL) = cl_pack<P>(du + (START));                                        \
            *reinterpret_cast<VT*>(rM + (2 * (Q) + 1) * CL_PL) = cl_pack<P>(dv + (START));                                        \
        }
    // red = parity 0 first; in this thread's row the pixels of parity Q are the strip pixels i == Q ^ s0 (mod 2); s0 is the same for a whole wave
    for (int it = 0; it < iters; it++) {
        if (act) { if (s0 == 0) CL_HALF(0, 0) else CL_HALF(1, 0) }
        cl_sync_lds();
        if (act) { if (s0 == 0) CL_HALF(1, 1) else CL_HALF(0, 1) }
        cl_sync_lds();
    }
    #undef CL_HALF
    if (act) {
        const int g0 = ly * w + x0;
        #pragma unroll
        for (int i = 0; i < NP; i++) if (x0 + i < w) { gU[g0 + i] = du[i]; gV[g0 + i] = dv[i]; }
    }
}

__global__ void __launch_bounds__(CL_NT) k_coarse_chain(const float* __restrict__ pyr0, const float* __restrict__ pyr1, FlowPlanes Pl, float* __restrict__ outWu, float* __restrict__ outWv, CoarseChain C) {
    extern __shared__ float4 cl_lds4[];
    float* lds = reinterpret_cast<float*>(cl_lds4);
    const int tid = threadIdx.x, b = blockIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t so = (size_t)b * C.stride;
    float* Wu = Pl.Wu + so; float* Wv = Pl.Wv + so; float* tWu = Pl.tWu + so; float* tWv = Pl.tWv + so;
    float* dWu = Pl.dWu + so; float* dWv = Pl.dWv + so; float* avg = Pl.avg + so; float* Iz = Pl.Iz + so;
    float* A11 = Pl.A11 + so; float* A12 = Pl.A12 + so; float* A22 = Pl.A22 + so; float* B1 = Pl.b1 + so; float* B2 = Pl.b2 + so; float* Wg = Pl.wgt + so;
    for (int k = 0; k < C.n; k++) {
        const CoarseLevel& Lv = C.L[k];
        const int w = Lv.w, h = Lv.h, n = w * h, RS = Lv.RS;
        const float* I0 = pyr0 + Lv.off + (size_t)b * n; const float* I1 = pyr1 + Lv.off + (size_t)b * n;
        // ---- prepareBuffers: warp, average, temporal difference; the increment starts at zero.  The LDS planes of this level are cleared (guards read as zero)
        const bool zero_w = k == 0 && C.init_zero;
        for (int i = tid; i < n; i += CL_NT) {
            const int y = i / w, x = i - y * w;
            float fu = 0.f, fv = 0.f;
            if (zero_w) { Wu[i] = 0.f; Wv[i] = 0.f; } else { fu = Wu[i]; fv = Wv[i]; }
            dWu[i] = 0.f; dWv[i] = 0.f;
            warp_px(I1, I0[i], fu, fv, x, y, w, h, avg[i], Iz[i]);
        }
        for (int i = tid; i < (h + 2) * RS; i += CL_NT) lds[i] = 0.f;
        cl_sync_global();
        for (int it = 0; it < C.fp_iters; it++) {
            // ---- ComputeDataTerm + ComputeSmoothnessTerm: a wave takes tiles of KL_COLS columns x KL_ROWS rows (k_coef_lanes' work item; every tile of a level this
            // small touches the image border)
            const int tiles_x = (w + KL_COLS - 1) / KL_COLS, items = tiles_x * ((h + KL_ROWS - 1) / KL_ROWS);
            for (int item = wave; item < items; item += CL_NT / 64) {
                const int ty = item / tiles_x, tx = item - ty * tiles_x;
                const int c0 = tx * KL_COLS - 2, y0 = ty * KL_ROWS, col = c0 + lane;
                const bool store_lane = lane >= 2 && lane < 2 + KL_COLS && col < w;
                kc_lanes<true, true>(C.V, w, h, col, y0, 0, avg, Iz, Wu, Wv, dWu, dWv, A11, A12, A22, B1, B2, Wg, nullptr, nullptr, store_lane);
            }
            cl_sync_global();
            // ---- RedBlackSOR
            if (Lv.P == 1) coarse_sor<1>(lds, w, h, Lv.SW, Lv.half, RS, C.sor_iters, C.V.omega, A11, A12, A22, B1, B2, Wg, dWu, dWv);
            else coarse_sor<2>(lds, w, h, Lv.SW, Lv.half, RS, C.sor_iters, C.V.omega, A11, A12, A22, B1, B2, Wg, dWu, dWv);
            cl_sync_global();
        }
        // ---- W += dW
        for (int i = tid; i < n; i += CL_NT) { Wu[i] = Wu[i] + dWu[i]; Wv[i] = Wv[i] + dWv[i]; }
        cl_sync_global();
        // ---- flow of the next finer level: cv::resize(INTER_LINEAR) of both components, then * post (1 / 0.95)
        if (k + 1 < C.n || C.upsample_last) {
            const int nw = Lv.nw, nh = Lv.nh, nn = nw * nh;
            const bool last = k + 1 == C.n;
            float* du_ = last ? outWu + (size_t)b * nn : tWu; float* dv_ = last ? outWv + (size_t)b * nn : tWv;
            for (int i = tid; i < nn; i += CL_NT) {
                const int dy = i / nw, dx = i - dy * nw;
                float vu = resize_px(Wu, w, h, dx, dy, Lv.sx, Lv.sy), vv = resize_px(Wv, w, h, dx, dy, Lv.sx, Lv.sy);
                vu = vu * C.post; vv = vv * C.post;
                du_[i] = vu; dv_[i] = vv;
            }
            cl_sync_global();
            float* t = Wu; Wu = tWu; tWu = t; t = Wv; Wv = tWv; tWv = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tiled levels when the launch is LATENCY-bound (few images: every tile of the launch has a compute unit to itself).  k_sor_fused's scheme -- an extended tile per workgroup,
// `iters` iterations per launch inside a halo of 2 * iters pixels, ping-pong between two increment buffers -- at the granularity of the chain kernel above: 64 x 64 extended
// tiles of 1024 threads (1 x 4 strips, two pixel updates per thread and half-sweep instead of four; system, reciprocals and neighbour weights in registers).  A launch costs
// ~6 us before its first iteration whatever it does afterwards, and idle compute units make redundant halo work free, so the plan (sor_iterations) runs MORE iterations per
// launch on deeper halos -- 13 + 12 instead of 5 x 5 where the tiles still fit the chip.  Same arithmetic per pixel update, same bits.
#define ST_E 64                      /* extended tile edge */
#define ST_P 2
#define ST_SW (ST_E / (2 * ST_P))    /* strips per row */
#define ST_PL 40                     /* floats per plane and row: 2 guard floats, 32 cells, guard */
#define ST_RS (4 * ST_PL + 16)       /* row stride: rows two apart (a wave's next 16 lanes) fall into the other half of the 64 banks */
__global__ void __launch_bounds__(1024) k_sor_tile(int w, int h, int IW, int IH, int halo, int ntx, int iters, float omega, const float* __restrict__ gA11, const float* __restrict__ gA12,
                                                   const float* __restrict__ gA22, const float* __restrict__ gB1, const float* __restrict__ gB2, const float* __restrict__ gW,
                                                   const float* __restrict__ gUin, const float* __restrict__ gVin, float* __restrict__ gUout, float* __restrict__ gVout) {
    constexpr int P = ST_P, NP = 2 * P;
    typedef ClVec<P>::T VT;
    __shared__ float4 lds4[(ST_E + 2) * ST_RS / 4];
    float* lds = reinterpret_cast<float*>(lds4);
    const int tid = threadIdx.x, tile = blockIdx.x;
    const size_t base = (size_t)blockIdx.y * w * h;
    const int tx = tile % ntx, ty = tile / ntx;
    const int ex0 = tx * IW - halo, ey0 = ty * IH - halo;          // origin of the extended tile in the image (may be negative)
    const bool lowh = tid < 512; const int idx = tid & 511, r2 = idx / ST_SW, j = idx - r2 * ST_SW;
    const int s0 = __builtin_amdgcn_readfirstlane(lowh ? 0 : 1), ly = 2 * r2 + s0, x0 = NP * j;
    const int gy = ey0 + ly, gx0 = ex0 + x0;
    const int off = (ex0 + ey0) & 1;                               // local parity of the globally red pixels
    float a11[NP], a12[NP], a22[NP], b1[NP], b2[NP], wp[NP], wu[NP], r11[NP], r22[NP], du[NP], dv[NP];
    float wl0 = 0.f; unsigned valid = 0;
    #pragma unroll
    for (int i = 0; i < NP; i++) { a11[i] = 1.f; a12[i] = 0.f; a22[i] = 1.f; b1[i] = 0.f; b2[i] = 0.f; wp[i] = 0.f; wu[i] = 0.f; r11[i] = 0.f; r22[i] = 0.f; du[i] = 0.f; dv[i] = 0.f; }
    const bool row_ok = gy >= 0 && gy < h;
    if (row_ok && gx0 >= 0 && gx0 + NP <= w) {                     // whole strip inside the image: 16-byte loads
        const size_t g = base + (size_t)gy * w + gx0;
        valid = 0xfu;
        #define ST_LD(PL, D) { const F4u t_ = *reinterpret_cast<const F4u*>(PL + g); D[0] = t_.x; D[1] = t_.y; D[2] = t_.z; D[3] = t_.w; }
        ST_LD(gA11, a11) ST_LD(gA12, a12) ST_LD(gA22, a22) ST_LD(gB1, b1) ST_LD(gB2, b2) ST_LD(gW, wp) ST_LD(gUin, du) ST_LD(gVin, dv)
        if (gy > 0) { const F4u t_ = *reinterpret_cast<const F4u*>(gW + g - w); wu[0] = t_.x; wu[1] = t_.y; wu[2] = t_.z; wu[3] = t_.w; }
        #undef ST_LD
    } else if (row_ok) {
        #pragma unroll
        for (int i = 0; i < NP; i++) {
            const int gx = gx0 + i;
            if (gx >= 0 && gx < w) {
                const size_t g = base + (size_t)gy * w + gx;
                valid |= 1u << i;
                a11[i] = gA11[g]; a12[i] = gA12[g]; a22[i] = gA22[g]; b1[i] = gB1[g]; b2[i] = gB2[g]; wp[i] = gW[g]; du[i] = gUin[g]; dv[i] = gVin[g];
                if (gy > 0) wu[i] = gW[g - w];
            }
        }
    }
    if (row_ok && gx0 - 1 >= 0 && gx0 - 1 < w) wl0 = gW[base + (size_t)gy * w + gx0 - 1];
    // a pixel outside the image keeps the reciprocal 0: its update returns exactly 0, with no select in the loop
    #pragma unroll
    for (int i = 0; i < NP; i++) if ((valid >> i) & 1u) { r11[i] = sor_rcp(a11[i]); r22[i] = sor_rcp(a22[i]); }
    for (int i = tid; i < (ST_E + 2) * ST_RS / 4; i += 1024) lds4[i] = make_float4(0.f, 0.f, 0.f, 0.f);      // guards (and everything else, overwritten below)
    cl_sync_lds();
    float* const rM = lds + (ly + 1) * ST_RS + P + P * j; float* const rU = rM - ST_RS; float* const rD = rM + ST_RS;
    *reinterpret_cast<VT*>(rM + (2 * s0 + 0) * ST_PL) = cl_pack<P>(du); *reinterpret_cast<VT*>(rM + (2 * s0 + 1) * ST_PL) = cl_pack<P>(dv);
    *reinterpret_cast<VT*>(rM + (2 * (s0 ^ 1) + 0) * ST_PL) = cl_pack<P>(du + 1); *reinterpret_cast<VT*>(rM + (2 * (s0 ^ 1) + 1) * ST_PL) = cl_pack<P>(dv + 1);
    cl_sync_lds();
    #define ST_HALF(START, Q)                                                                                                     \
        {                                                                                                                         \
            constexpr int oq = (Q) ^ 1;                                                                                           \
            float uu[P], vu[P], ud[P], vd[P];                                                                                     \
            cl_unpack<P>(*reinterpret_cast<const VT*>(rU + (2 * oq + 0) * ST_PL), uu); cl_unpack<P>(*reinterpret_cast<const VT*>(rU + (2 * oq + 1) * ST_PL), vu); \
            cl_unpack<P>(*reinterpret_cast<const VT*>(rD + (2 * oq + 0) * ST_PL), ud); cl_unpack<P>(*reinterpret_cast<const VT*>(rD + (2 * oq + 1) * ST_PL), vd); \
            const float eu = rM[(2 * oq + 0) * ST_PL + ((START) == 0 ? -1 : P)], ev = rM[(2 * oq + 1) * ST_PL + ((START) == 0 ? -1 : P)]; \
            _Pragma("unroll")                                                                                                     \
            for (int k = 0; k < P; k++) {                                                                                         \
                const int i = (START) + 2 * k;                                                                                    \
                const float wl = i == 0 ? wl0 : wp[i == 0 ? 0 : i - 1];                                                           \
                const float ul = i == 0 ? eu : du[i == 0 ? 0 : i - 1], vl = i == 0 ? ev : dv[i == 0 ? 0 : i - 1];                \
                const float ur = i == NP - 1 ? eu : du[i == NP - 1 ? NP - 1 : i + 1], vr = i == NP - 1 ? ev : dv[i == NP - 1 ? NP - 1 : i + 1]; \
                const float sigmaU = wl * ul + wp[i] * ur + wu[i] * uu[k] + wp[i] * ud[k];                                        \
                const float sigmaV = wl * vl + wp[i] * vr + wu[i] * vu[k] + wp[i] * vd[k];                                        \
                float nu = du[i], nv = dv[i];                                                                                     \
                nu += omega * (sor_div(sigmaU + b1[i] - nv * a12[i], a11[i], r11[i]) - nu);                                       \
                nv += omega * (sor_div(sigmaV + b2[i] - nu * a12[i], a22[i], r22[i]) - nv);                                       \
                du[i] = nu; dv[i] = nv;                                                                                           \
            }                                                                                                                     \
            *reinterpret_cast<VT*>(rM + (2 * (Q) + 0) * ST_PL) = cl_pack<P>(du + (START));                                        \
            *reinterpret_cast<VT*>(rM + (2 * (Q) + 1) * ST_PL) = cl_pack<P>(dv + (START));                                        \
        }
    // the globally red pixels have local parity `off`; in this thread's row the pixels of local parity Q are the strip pixels i == Q ^ s0 (mod 2); both are scalars
    if (off == 0) {
        if (s0 == 0) for (int it = 0; it < iters; it++) { ST_HALF(0, 0) cl_sync_lds(); ST_HALF(1, 1) cl_sync_lds(); }
        else         for (int it = 0; it < iters; it++) { ST_HALF(1, 0) cl_sync_lds(); ST_HALF(0, 1) cl_sync_lds(); }
    } else {
        if (s0 == 0) for (int it = 0; it < iters; it++) { ST_HALF(1, 1) cl_sync_lds(); ST_HALF(0, 0) cl_sync_lds(); }
        else         for (int it = 0; it < iters; it++) { ST_HALF(0, 1) cl_sync_lds(); ST_HALF(1, 0) cl_sync_lds(); }
    }
    #undef ST_HALF
    const int ix0 = tx * IW, iy0 = ty * IH;
    if (row_ok && gy >= iy0 && gy < iy0 + IH) {
        const size_t g = base + (size_t)gy * w + gx0;
        if (valid == 0xfu && gx0 >= ix0 && gx0 + NP <= ix0 + IW) {
            *reinterpret_cast<F4u*>(gUout + g) = F4u{du[0], du[1], du[2], du[3]}; *reinterpret_cast<F4u*>(gVout + g) = F4u{dv[0], dv[1], dv[2], dv[3]};
        } else {
            #pragma unroll
            for (int i = 0; i < NP; i++) if (((valid >> i) & 1u) && gx0 + i >= ix0 && gx0 + i < ix0 + IW) { gUout[g + i] = du[i]; gVout[g + i] = dv[i]; }
        }
    }
}
// `iters` iterations of one launch over the B images of a level: grid (tiles, B)
int launch_sor_tile(hipStream_t s, FlowPlanes& P, int w, int h, int B, int iters, float omega) {
    const int halo = 2 * iters, IW = ST_E - 2 * halo, IH = ST_E - 2 * halo;
    if (iters < 1 || IW < 4) { sind_set_error("launch_sor_tile: %d iterations per launch do not fit a %d-pixel tile", iters, ST_E); return SIND_E_ARG; }
    const int ntx = divup(w, IW), nty = divup(h, IH);
    hipLaunchKernelGGL(k_sor_tile, dim3(ntx * nty, B), dim3(1024), 0, s, w, h, IW, IH, halo, ntx, iters, omega, P.A11, P.A12, P.A22, P.b1, P.b2, P.wgt, P.dWu, P.dWv, P.dWu2, P.dWv2);
    std::swap(P.dWu, P.dWu2); std::swap(P.dWv, P.dWv2);
    return SIND_OK;
}
int sor_tile_count(int w, int h, int iters) { const int I = ST_E - 4 * iters; return I < 4 ? (1 << 30) : divup(w, I) * divup(h, I); }

// P for a level, or 0 if the level is not one workgroup's work in the chain kernel
int coarse_level_P(int w, int h) {
    // (P = 4 -- eight pixels per thread, the levels of 4 k to 8 k pixels -- holds 88 persistent registers per thread and spills at 128: those levels keep the per-stage kernels)
    for (int P : {1, 2}) {
        const int SW = divup(w, 2 * P), half = divup(SW * ((h + 1) / 2), 64) * 64;
        if (2 * half <= CL_NT && P * SW + P + 1 <= CL_PL) return P;
    }
    return 0;
}
static void coarse_fill(CoarseLevel& L, int w, int h, int nw, int nh, size_t off) {
    L.w = w; L.h = h; L.P = coarse_level_P(w, h); L.SW = divup(w, 2 * L.P); L.half = divup(L.SW * ((h + 1) / 2), 64) * 64;
    // row stride: four planes + a pad that lets the rows a wave covers (two apart) continue each other's LDS banks (2 * RS = P * SW mod 64), a multiple of 4 floats
    const int pad = divup(divup(L.P * L.SW, 2), 4) * 4;
    L.RS = 4 * CL_PL + pad; L.nw = nw; L.nh = nh; L.off = off;
    L.sx = nw > 0 ? 1.0 / ((double)nw / w) : 1.0; L.sy = nh > 0 ? 1.0 / ((double)nh / h) : 1.0;
}
static size_t coarse_lds_bytes(const CoarseLevel& L) { return (size_t)(L.h + 2) * L.RS * sizeof(float); }

// Levels first .. last (indices into `levels`, first = the coarsest of the chain, last <= first the finest) of B pairs in one launch.  The flow enters in Pl.Wu / Pl.Wv with
// `stride` floats per image (ignored with init_zero) and leaves
//   * upsample_last: up-sampled to level last - 1, in out_u / out_v, laid out [B][nh][nw];
//   * otherwise: in Pl.Wu / Pl.Wv (the chain swaps the two buffer pairs once per level inside; an even number of swaps for one level -> none).
int launch_coarse_chain(hipStream_t s, FlowPlanes& Pl, const float* pyr0, const float* pyr1, const std::vector<std::pair<int, int>>& levels, const std::vector<size_t>& level_off,
                        int first, int last, int B, const VarParams& V, bool init_zero, bool upsample_last, float post, float* out_u, float* out_v) {
    const int n = first - last + 1;
    if (n < 1 || n > CL_MAXLV || last < 0 || first >= (int)levels.size() || (upsample_last && (last == 0 || !out_u || !out_v))) { sind_set_error("launch_coarse_chain: bad level range"); return SIND_E_ARG; }
    if (V.epsilon < 1e-12f) { sind_set_error("launch_coarse_chain: epsilon below the short forms' range"); return SIND_E_ARG; }
    CoarseChain C;
    C.n = n; C.init_zero = init_zero; C.upsample_last = upsample_last; C.post = post; C.fp_iters = V.fixedPointIterations; C.sor_iters = V.sorIterations; C.V = V;
    size_t shm = 0; int stride = 0;
    for (int k = 0; k < n; k++) {
        const int l = first - k, w = levels[l].first, h = levels[l].second;
        if (!coarse_level_P(w, h)) { sind_set_error("launch_coarse_chain: level %d x %d is not one workgroup's work", w, h); return SIND_E_ARG; }
        const bool up = l > 0 && (k + 1 < n || upsample_last);
        coarse_fill(C.L[k], w, h, up ? levels[l - 1].first : 0, up ? levels[l - 1].second : 0, level_off[l] * (size_t)B);
        shm = std::max(shm, coarse_lds_bytes(C.L[k])); stride = std::max(stride, w * h);
    }
    C.stride = stride;
    static SindPerDeviceInit attr_init;
    HIP_TRY(attr_init.run([] { return hipFuncSetAttribute((const void*)k_coarse_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }));
    if (shm > 150 * 1024) { sind_set_error("launch_coarse_chain: %zu bytes of LDS", shm); return SIND_E_ARG; }
    hipLaunchKernelGGL(k_coarse_chain, dim3(B), dim3(CL_NT), shm, s, pyr0, pyr1, Pl, out_u, out_v, C);
    HIP_TRY(hipGetLastError());
    // the buffer pairs (Wu, Wv) / (tWu, tWv) change roles once per up-sampling inside the kernel
    if ((n - 1) & 1) { std::swap(Pl.Wu, Pl.tWu); std::swap(Pl.Wv, Pl.tWv); }
    return SIND_OK;
}

}  // namespace sind
