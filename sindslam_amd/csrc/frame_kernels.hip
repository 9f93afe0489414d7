// Frame post-ORB steps on the GPU: the per-keypoint work the reference's RGB-D Frame constructor does
// after ExtractORB (reference src/Frame.cc:143-170): UndistortKeyPoints (:477-509), ComputeStereoFromRGBD (:714-735),
// ComputeImageBounds (:511-541) and AssignFeaturesToGrid (:283-299 with PosInGrid :453-463).
// One workgroup per frame: undistort + depth gather + cell id per keypoint, then the 64x48 grid as CSR (count in LDS,
// scan, scatter, per-cell index sort = the reference's push_back order).  FP64 restatement of cv::undistortPoints
// (5 fixed-point iterations, R = I, P = K) in OpenCV's scalar operation order; built with -ffp-contract=off.
#include "frame.hpp"

namespace sind {

#define FR_NT 1024

__device__ __forceinline__ void d_undistort_point(const FrameCalib& c, float xin, float yin, float& xo, float& yo) {
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2, k4 = c.k3;
    double x = xin, y = yin; const double u = x, v = y;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
        x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    xo = (float)(xx * ww); yo = (float)(yy * ww);
}

__global__ __launch_bounds__(FR_NT) void k_frame_post_orb(FrameCalib c, const float* __restrict__ kxy, const int* __restrict__ nkp, int cap,
                                                         const uint16_t* __restrict__ depth, int W, int H, float* __restrict__ un_xy,
                                                         float* __restrict__ u_right, float* __restrict__ depth_out, int* __restrict__ cell_out,
                                                         int* __restrict__ grid_start, int* __restrict__ grid_idx, float* __restrict__ bounds_out) {
    __shared__ int cnt[FRAME_CELLS]; __shared__ int cur[FRAME_CELLS]; __shared__ int part[FR_NT]; __shared__ float bnd[4]; __shared__ float corner[4][2];
    const int b = blockIdx.x, t = threadIdx.x, N = min(nkp[b], cap);
    kxy += (size_t)b * cap * 2; un_xy += (size_t)b * cap * 2; u_right += (size_t)b * cap; depth_out += (size_t)b * cap; cell_out += (size_t)b * cap;
    grid_start += (size_t)b * (FRAME_CELLS + 1); grid_idx += (size_t)b * cap; depth += (size_t)b * W * H;
    const bool distorted = c.k1 != 0.0f;
    for (int i = t; i < FRAME_CELLS; i += FR_NT) cnt[i] = 0;
    if (t < 4) {                                                       // ComputeImageBounds: undistorted image corners
        const float sx = (t & 1) ? (float)W : 0.f, sy = (t & 2) ? (float)H : 0.f;
        if (distorted) d_undistort_point(c, sx, sy, corner[t][0], corner[t][1]); else { corner[t][0] = sx; corner[t][1] = sy; }
    }
    __syncthreads();
    if (t == 0) {
        if (distorted) { bnd[0] = fminf(corner[0][0], corner[2][0]); bnd[1] = fmaxf(corner[1][0], corner[3][0]); bnd[2] = fminf(corner[0][1], corner[1][1]); bnd[3] = fmaxf(corner[2][1], corner[3][1]); }
        else { bnd[0] = 0.f; bnd[1] = (float)W; bnd[2] = 0.f; bnd[3] = (float)H; }
        if (b == 0 && bounds_out) { bounds_out[0] = bnd[0]; bounds_out[1] = bnd[1]; bounds_out[2] = bnd[2]; bounds_out[3] = bnd[3]; }
    }
    __syncthreads();
    const float minX = bnd[0], minY = bnd[2];
    const float wInv = (float)FRAME_GRID_COLS / (float)(bnd[1] - bnd[0]), hInv = (float)FRAME_GRID_ROWS / (float)(bnd[3] - bnd[2]);
    for (int i = t; i < N; i += FR_NT) {
        const float x = kxy[2 * i], y = kxy[2 * i + 1]; float ux = x, uy = y;
        if (distorted) d_undistort_point(c, x, y, ux, uy);
        un_xy[2 * i] = ux; un_xy[2 * i + 1] = uy;
        const int v = min(max((int)y, 0), H - 1), u = min(max((int)x, 0), W - 1);     // keypoints lie inside the image; clamp = memory safety only
        const float d = (float)depth[(size_t)v * W + u] * c.depthMapFactor;
        float dep = -1.f, ur = -1.f;
        if (d > 0) { dep = d; ur = ux - c.bf / d; }
        depth_out[i] = dep; u_right[i] = ur;
        const int px = (int)roundf((ux - minX) * wInv), py = (int)roundf((uy - minY) * hInv);
        int cell = -1;
        if (!(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS)) { cell = px * FRAME_GRID_ROWS + py; atomicAdd(&cnt[cell], 1); }
        cell_out[i] = cell;
    }
    __syncthreads();
    // exclusive scan of the 3072 cell counts: 3 per thread + block scan of the partial sums
    const int c0 = cnt[3 * t], c1 = cnt[3 * t + 1], c2 = cnt[3 * t + 2];
    part[t] = c0 + c1 + c2;
    __syncthreads();
    for (int off = 1; off < FR_NT; off <<= 1) { const int v = t >= off ? part[t - off] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    const int base = part[t] - (c0 + c1 + c2);
    cur[3 * t] = base; cur[3 * t + 1] = base + c0; cur[3 * t + 2] = base + c0 + c1;
    grid_start[3 * t] = base; grid_start[3 * t + 1] = base + c0; grid_start[3 * t + 2] = base + c0 + c1;
    if (t == FR_NT - 1) grid_start[FRAME_CELLS] = part[t];
    __syncthreads();
    for (int i = t; i < N; i += FR_NT) { const int cell = cell_out[i]; if (cell >= 0) grid_idx[atomicAdd(&cur[cell], 1)] = i; }
    __syncthreads();
    for (int g = t; g < FRAME_CELLS; g += FR_NT) {                     // restore push_back order (ascending keypoint index) inside each cell
        const int n = cnt[g], s = cur[g] - n;
        for (int a = 1; a < n; a++) { const int key = grid_idx[s + a]; int q = a - 1; while (q >= 0 && grid_idx[s + q] > key) { grid_idx[s + q + 1] = grid_idx[s + q]; q--; } grid_idx[s + q + 1] = key; }
    }
}

int launch_frame_post_orb(const FrameCalib& c, const float* kxy, const int* nkp, int B, int cap, const uint16_t* depth, int W, int H, float* un_xy,
                          float* u_right, float* depth_out, int* cell, int* grid_start, int* grid_idx, float* bounds, hipStream_t s) {
    static_assert(FRAME_CELLS == 3 * FR_NT, "scan assumes three cells per thread");
    hipLaunchKernelGGL(k_frame_post_orb, dim3(B), dim3(FR_NT), 0, s, c, kxy, nkp, cap, depth, W, H, un_xy, u_right, depth_out, cell, grid_start, grid_idx, bounds);
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}

}  // namespace sind
