// Plane fit of one PEAC node from its second-order statistics (reference include/PEAC/AHCPlaneSeg.hpp:103-134 PCA; Eigen's 3x3 self-adjoint solver replaced by
// a cyclic Jacobi iteration, see host/peac.cpp).  ONE source for the host (graph clustering: the fits of merge candidates) and the device (the 1200-3600
// initial block fits of a frame, k_peac_block_fit): IEEE FP64 add / mul / div / sqrt on both sides and no contraction (-ffp-contract=off), so the two
// give the same bits.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define SIND_HD __host__ __device__
#else
#define SIND_HD
#endif

namespace sind {

SIND_HD inline void peac_jacobi3(const double K[3][3], double s[3], double V[3][3]) {
    double A[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A[i][j] = K[i][j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (A[p][q] == 0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            const double c = 1 / sqrt(t * t + 1), sn = t * c;
            for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - sn * b; A[k][q] = sn * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - sn * b; A[q][k] = sn * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
        }
    }
    int o[3] = {0, 1, 2}; const double e[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (e[o[i]] > e[o[j]]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
    double T[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i][j] = V[i][o[j]];
    for (int i = 0; i < 3; i++) { s[i] = e[o[i]]; for (int j = 0; j < 3; j++) V[i][j] = T[i][j]; }
}

// moments m = {sx, sy, sz, sxx, syy, szz, sxy, syz, sxz} of N points -> centre, unit normal towards the camera, mean squared error
SIND_HD inline void peac_fit(const double m[9], int N, double center[3], double normal[3], double& mse) {
    const double sx = m[0], sy = m[1], sz = m[2], sxx = m[3], syy = m[4], szz = m[5], sxy = m[6], syz = m[7], sxz = m[8];
    const double sc = 1.0 / N;
    center[0] = sx * sc; center[1] = sy * sc; center[2] = sz * sc;
    double K[3][3] = {{sxx - sx * sx * sc, sxy - sx * sy * sc, sxz - sx * sz * sc}, {0, syy - sy * sy * sc, syz - sy * sz * sc}, {0, 0, szz - sz * sz * sc}};
    K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
    double sv[3], V[3][3]; peac_jacobi3(K, sv, V);
    const double sgn = (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) ? 1.0 : -1.0;
    normal[0] = sgn * V[0][0]; normal[1] = sgn * V[1][0]; normal[2] = sgn * V[2][0];
    mse = sv[0] * sc;
}

}  // namespace sind
