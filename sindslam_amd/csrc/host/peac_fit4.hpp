// Four plane fits at once: peac_fit (peac_fit.hpp) with the cyclic Jacobi iteration carried out on four candidates in the four lanes of an AVX2 register.
// The merge candidates of one node of the PEAC graph (host/peac.cpp cluster(): 4.3 on average, 2 600 fits per 640 x 480 frame, 10 000 at 1280 x 720) are
// independent, and one fit is a chain of ~20 dependent rotations (two sqrt and two divisions each) that leaves the core's pipes idle: four lanes cost what
// one costs.  Every lane performs exactly the scalar code's IEEE operations in the scalar code's order -- lanes that have converged, or whose pivot is zero,
// keep their values through a mask -- so the results are those of peac_fit bit for bit (tests/test_host_stages_cpu.py::test_four_lane_plane_fit_equals_scalar).
// Host only (the device fits its blocks one per thread with peac_fit itself).
#pragma once
#include "peac_fit.hpp"
#if defined(__AVX2__)
#include <immintrin.h>
#endif

namespace sind {

struct PeacFitIn { double m[9]; int N; };                    // moments {sx, sy, sz, sxx, syy, szz, sxy, syz, sxz} of N points
struct PeacFitOut { double center[3], normal[3], mse; };

#if defined(__AVX2__)
// n <= 4 fits (unused lanes repeat the first one)
inline void peac_fit4(const PeacFitIn* in, int n, PeacFitOut* out) {
    alignas(32) double Kl[6][4], sc[4];                       // K00 K01 K02 K11 K12 K22 per lane
    for (int l = 0; l < 4; l++) {
        const PeacFitIn& f = in[l < n ? l : 0];
        const double sx = f.m[0], sy = f.m[1], sz = f.m[2], sxx = f.m[3], syy = f.m[4], szz = f.m[5], sxy = f.m[6], syz = f.m[7], sxz = f.m[8];
        const double s = 1.0 / f.N; sc[l] = s;
        Kl[0][l] = sxx - sx * sx * s; Kl[1][l] = sxy - sx * sy * s; Kl[2][l] = sxz - sx * sz * s; Kl[3][l] = syy - sy * sy * s; Kl[4][l] = syz - sy * sz * s; Kl[5][l] = szz - sz * sz * s;
    }
    typedef __m256d V;
    const V one = _mm256_set1_pd(1.0), zero = _mm256_setzero_pd(), two = _mm256_set1_pd(2.0), tiny = _mm256_set1_pd(1e-32);
    const V absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffLL));
    V A[3][3], E[3][3];
    A[0][0] = _mm256_load_pd(Kl[0]); A[0][1] = A[1][0] = _mm256_load_pd(Kl[1]); A[0][2] = A[2][0] = _mm256_load_pd(Kl[2]);
    A[1][1] = _mm256_load_pd(Kl[3]); A[1][2] = A[2][1] = _mm256_load_pd(Kl[4]); A[2][2] = _mm256_load_pd(Kl[5]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E[i][j] = i == j ? one : zero;
    #define MUL _mm256_mul_pd
    #define ADD _mm256_add_pd
    #define SUB _mm256_sub_pd
    V live = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));    // lanes still sweeping
    for (int sweep = 0; sweep < 60; sweep++) {
        const V off = ADD(ADD(MUL(A[0][1], A[0][1]), MUL(A[0][2], A[0][2])), MUL(A[1][2], A[1][2]));
        const V diag = ADD(ADD(MUL(A[0][0], A[0][0]), MUL(A[1][1], A[1][1])), MUL(A[2][2], A[2][2]));
        const V stop = _mm256_or_pd(_mm256_cmp_pd(off, MUL(tiny, diag), _CMP_LE_OQ), _mm256_cmp_pd(off, zero, _CMP_EQ_OQ));
        live = _mm256_andnot_pd(stop, live);
        if (_mm256_movemask_pd(live) == 0) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            const V act = _mm256_and_pd(live, _mm256_cmp_pd(A[p][q], zero, _CMP_NEQ_UQ));      // scalar: if (A[p][q] == 0) continue
            if (_mm256_movemask_pd(act) == 0) continue;
            const V theta = _mm256_div_pd(SUB(A[q][q], A[p][p]), MUL(two, A[p][q]));
            const V sgn = _mm256_blendv_pd(_mm256_set1_pd(-1.0), one, _mm256_cmp_pd(theta, zero, _CMP_GE_OQ));
            const V t = _mm256_div_pd(sgn, ADD(_mm256_and_pd(theta, absmask), _mm256_sqrt_pd(ADD(MUL(theta, theta), one))));
            const V c = _mm256_div_pd(one, _mm256_sqrt_pd(ADD(MUL(t, t), one))), sn = MUL(t, c);
            #define ROT(X, Y) { const V a_ = X, b_ = Y; X = _mm256_blendv_pd(a_, SUB(MUL(c, a_), MUL(sn, b_)), act); Y = _mm256_blendv_pd(b_, ADD(MUL(sn, a_), MUL(c, b_)), act); }
            for (int k = 0; k < 3; k++) ROT(A[k][p], A[k][q])
            for (int k = 0; k < 3; k++) ROT(A[p][k], A[q][k])
            for (int k = 0; k < 3; k++) ROT(E[k][p], E[k][q])
            #undef ROT
        }
    }
    #undef MUL
    #undef ADD
    #undef SUB
    alignas(32) double e[3][4], Vv[3][3][4];
    for (int i = 0; i < 3; i++) { _mm256_store_pd(e[i], A[i][i]); for (int j = 0; j < 3; j++) _mm256_store_pd(Vv[i][j], E[i][j]); }
    for (int l = 0; l < n; l++) {
        int o[3] = {0, 1, 2};
        for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (e[o[i]][l] > e[o[j]][l]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
        const PeacFitIn& f = in[l]; PeacFitOut& r = out[l];
        r.center[0] = f.m[0] * sc[l]; r.center[1] = f.m[1] * sc[l]; r.center[2] = f.m[2] * sc[l];
        const double v0 = Vv[0][o[0]][l], v1 = Vv[1][o[0]][l], v2 = Vv[2][o[0]][l];
        const double sgn = (v0 * r.center[0] + v1 * r.center[1] + v2 * r.center[2] <= 0) ? 1.0 : -1.0;
        r.normal[0] = sgn * v0; r.normal[1] = sgn * v1; r.normal[2] = sgn * v2;
        r.mse = e[o[0]][l] * sc[l];
    }
}
#else
inline void peac_fit4(const PeacFitIn* in, int n, PeacFitOut* out) { for (int l = 0; l < n; l++) peac_fit(in[l].m, in[l].N, out[l].center, out[l].normal, out[l].mse); }
#endif

}  // namespace sind
