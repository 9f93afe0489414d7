// Key-frame cloud generation of the mapping consumer on the GPU (reference octomap_pub/src/pubPointCloud.cc:471-660):
// stride-2 back-projection, re-projection depth-consistency vote per cluster (:556-607), cluster rejection (:641-663), world
// transform (:665).  Three launches per batch: (1) per-pixel vote + per-chunk label histogram, (2) cluster decision and the exclusive
// offsets that reproduce the reference's output order (cluster 0, then the kept clusters, raster order inside a cluster),
// (3) stable scatter of the transformed points.  FP64 where the reference uses Eigen / PCL doubles, reduction order
// a0*b0 + (a1*b1 + a2*b2) for the fixed-size products, built with -ffp-contract=off.  HBM-bound: ~10 B read + 4 B written per pixel.
#include "cloud.hpp"

namespace sind {

__device__ __forceinline__ double d_dot3(double a0, double a1, double a2, double b0, double b1, double b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }

// stride-2 pixel index -> (m, n); returns false outside the grid
__device__ __forceinline__ bool d_grid_px(int idx, int gw, int np, int& m, int& n) { if (idx >= np) return false; m = (idx / gw) * 2; n = (idx % gw) * 2; return true; }

__global__ __launch_bounds__(CLOUD_CHUNK) void k_cloud_vote(CloudCam cam, CloudArrays a, int W, int H, int gw, int np, int nchunks) {
    __shared__ int cnt[CLOUD_LABELS], occ[CLOUD_LABELS], full[CLOUD_LABELS];
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    if (t < CLOUD_LABELS) { cnt[t] = 0; occ[t] = 0; full[t] = 0; }
    __syncthreads();
    const size_t fo = (size_t)b * W * H;
    const uint16_t* depth = a.depth + fo; const uint16_t* depthLast = a.depthLast + fo; const uint8_t* dynaLast = a.dynaLast + fo; const uint8_t* label = a.label + fo;
    int m, n;
    if (d_grid_px(chunk * CLOUD_CHUNK + t, gw, np, m, n)) {
        for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++)                    // countNonZero(imgLabel == i) runs over every pixel
            if (m + dy < H && n + dx < W) { const int l = label[(size_t)(m + dy) * W + n + dx]; if (l < CLOUD_LABELS) atomicAdd(&full[l], 1); }
        const size_t px = (size_t)m * W + n;
        const int iLabel = label[px];
        if (iLabel < CLOUD_LABELS) {
            atomicAdd(&cnt[iLabel], 1);
            const CloudPose& ps = a.pose[b];
            const float dCurrent = (float)(depth[px] * (1.0 / cam.depthScale));
            const double p0 = (double)((n - (float)cam.cx) * (float)dCurrent / (float)cam.fx), p1 = (double)((m - (float)cam.cy) * dCurrent / (float)cam.fy), p2 = (double)dCurrent;
            const double q0 = d_dot3(ps.rel[0], ps.rel[1], ps.rel[2], p0, p1, p2) + ps.rel[3], q1 = d_dot3(ps.rel[4], ps.rel[5], ps.rel[6], p0, p1, p2) + ps.rel[7],
                         q2 = d_dot3(ps.rel[8], ps.rel[9], ps.rel[10], p0, p1, p2) + ps.rel[11];
            double t0 = d_dot3(cam.fx, 0.0, cam.cx, q0, q1, q2), t1 = d_dot3(0.0, cam.fy, cam.cy, q0, q1, q2); const double t2 = d_dot3(0.0, 0.0, 1.0, q0, q1, q2);
            t0 /= t2; t1 /= t2;
            const float xt = (float)t0, yt = (float)t1;
            float dLast = 0.0f; bool isDynaLast = false;
            if (yt >= 0.0f && yt < (float)H && xt >= 0.0f && xt < (float)W) {
                const size_t pl = (size_t)(int)yt * W + (int)xt;
                dLast = (float)(depthLast[pl] * (1.0 / cam.depthScale)); isDynaLast = dynaLast[pl] > 240;
            }
            if (dCurrent >= 0 && dCurrent < 10 && dLast >= 0 && dLast < 10) {
                const float diff = dCurrent - dLast;
                if ((double)(diff * diff) > (0.13 * dCurrent) * (0.13 * dCurrent) || isDynaLast) atomicAdd(&occ[iLabel], 1);
            }
        }
    }
    __syncthreads();
    if (t < CLOUD_LABELS) {
        a.chunkCnt[((size_t)b * nchunks + chunk) * CLOUD_LABELS + t] = cnt[t];
        if (occ[t]) atomicAdd(&a.occlusion[b * CLOUD_LABELS + t], occ[t]);
        if (full[t]) atomicAdd(&a.labelCount[b * CLOUD_LABELS + t], full[t]);
    }
}

__global__ void k_cloud_offsets(CloudArrays a, int nchunks) {
    __shared__ int base[CLOUD_LABELS + 1], keep[CLOUD_LABELS], sum[CLOUD_LABELS];
    const int b = blockIdx.x, l = threadIdx.x;
    if (l < CLOUD_LABELS) {
        int s = 0; for (int c = 0; c < nchunks; c++) s += a.chunkCnt[((size_t)b * nchunks + c) * CLOUD_LABELS + l];
        sum[l] = s;
        keep[l] = (l == 0) || ((double)a.occlusion[b * CLOUD_LABELS + l] * 9 <= 0.4 * a.labelCount[b * CLOUD_LABELS + l]);      // pubPointCloud.cc:651
        a.kept[b * CLOUD_LABELS + l] = keep[l];
    }
    __syncthreads();
    if (l == 0) { int r = 0; for (int i = 0; i < CLOUD_LABELS; i++) { base[i] = r; if (keep[i]) r += sum[i]; } base[CLOUD_LABELS] = r; a.total[b] = r; }
    __syncthreads();
    if (l < CLOUD_LABELS) {
        int r = base[l];
        for (int c = 0; c < nchunks; c++) { const size_t k = ((size_t)b * nchunks + c) * CLOUD_LABELS + l; a.chunkOff[k] = keep[l] ? r : -1; r += a.chunkCnt[k]; }
    }
}

__global__ __launch_bounds__(CLOUD_CHUNK) void k_cloud_scatter(CloudCam cam, CloudArrays a, int W, int H, int gw, int np, int nchunks) {
    __shared__ int wcnt[CLOUD_CHUNK / 64][CLOUD_LABELS];
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t fo = (size_t)b * W * H;
    int m = 0, n = 0, iLabel = 255;
    const bool in = d_grid_px(chunk * CLOUD_CHUNK + t, gw, np, m, n);
    const size_t px = (size_t)m * W + n;
    if (in) iLabel = a.label[fo + px];
    int rank = 0;
    for (int l = 0; l < CLOUD_LABELS; l++) {                              // stable rank inside the chunk: raster order = thread order
        const unsigned long long mask = __ballot(iLabel == l);
        if (lane == 0) wcnt[wave][l] = __popcll(mask);
        if (iLabel == l) rank = __popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (!in || iLabel >= CLOUD_LABELS) return;
    const int off = a.chunkOff[((size_t)b * nchunks + chunk) * CLOUD_LABELS + iLabel];
    if (off < 0) return;
    for (int w = 0; w < wave; w++) rank += wcnt[w][iLabel];
    const float dCurrent = (float)(a.depth[fo + px] * (1.0 / cam.depthScale));
    CloudPoint p; p.a = 255;
    const uint8_t* c = a.bgr + (fo + px) * 3; p.b = c[0]; p.g = c[1]; p.r = c[2];
    const float qnan = __int_as_float(0x7fc00000);
    if ((int)a.dyna[fo + px] >= 240 || dCurrent < 0.01 || dCurrent > 10) p.x = p.y = p.z = qnan;
    else {
        const float z = dCurrent, x = (float)((n - cam.cx) * z / cam.fx), y = (float)((m - cam.cy) * z / cam.fy);
        const double* T = a.pose[b].twc; const double X = x, Y = y, Z = z;               // pcl::transformPointCloud<PointT, double>
        p.x = (float)(T[0] * X + T[1] * Y + T[2] * Z + T[3]); p.y = (float)(T[4] * X + T[5] * Y + T[6] * Z + T[7]); p.z = (float)(T[8] * X + T[9] * Y + T[10] * Z + T[11]);
    }
    a.out[(size_t)b * np + off + rank] = p;
}

int launch_cloud(const CloudCam& cam, const CloudArrays& a, int W, int H, int B, hipStream_t s) {
    const int gw = (W + 1) / 2, np = cloud_grid_points(W, H), nchunks = divup(np, CLOUD_CHUNK);
    HIP_TRY(hipMemsetAsync(a.occlusion, 0, (size_t)B * CLOUD_LABELS * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(a.labelCount, 0, (size_t)B * CLOUD_LABELS * sizeof(int), s));
    hipLaunchKernelGGL(k_cloud_vote, dim3(nchunks, B), dim3(CLOUD_CHUNK), 0, s, cam, a, W, H, gw, np, nchunks);
    hipLaunchKernelGGL(k_cloud_offsets, dim3(B), dim3(64), 0, s, a, nchunks);
    hipLaunchKernelGGL(k_cloud_scatter, dim3(nchunks, B), dim3(CLOUD_CHUNK), 0, s, cam, a, W, H, gw, np, nchunks);
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}

}  // namespace sind
