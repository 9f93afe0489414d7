// ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) on the GPU (reference
// src/ORBmatcher.cc:1328-1470, Frame::GetFeaturesInArea src/Frame.cc:398-451, DescriptorDistance src/ORBmatcher.cc:1647-1665,
// ComputeThreeMaxima :1601-1642).  One workgroup per frame pair, one thread per last-frame map point.
//
// The reference loop is sequential in one respect: a current keypoint that already holds a MapPoint with observations is skipped
// by later points.  Candidate c is therefore closed for point i exactly when some j < i with observations chose c.  The kernel
// solves that recurrence by rounds: every round recomputes all choices against the owners of the previous round
// (minOwner[c] = smallest such j); after round r the first r points are final, and a round without change is the sequential
// result.  Real frames need 2-3 rounds.  The window search keeps the reference's traversal order (cells x-major, push_back order
// inside a cell, strict '<' on the distance), so ties resolve identically.  Integer / bit work: v_bcnt popcounts, L2-resident.
#include "match.hpp"

namespace sind {

#define MT_NT 1024
#define HISTO_LENGTH 30
#define TH_HIGH 100

struct Proj { float u, v, invzc, radius; int minL, maxL, ok; };

__device__ __forceinline__ Proj d_project(const MatchParams& p, const MatchPose& ps, const float* X, int oct) {
    Proj r; r.ok = 0;
    float xc[3];
    for (int k = 0; k < 3; k++) {                                      // cv::gemm small-matrix path: FP32 row product, FP64 alpha/beta
        const float t = ps.Tcw[4 * k] * X[0] + ps.Tcw[4 * k + 1] * X[1] + ps.Tcw[4 * k + 2] * X[2];
        xc[k] = (float)((double)t * 1.0 + (double)ps.Tcw[4 * k + 3] * 1.0);
    }
    r.invzc = (float)(1.0 / xc[2]);
    if (r.invzc < 0) return r;
    r.u = p.fx * xc[0] * r.invzc + p.cx; r.v = p.fy * xc[1] * r.invzc + p.cy;
    if (r.u < p.bounds[0] || r.u > p.bounds[1]) return r;
    if (r.v < p.bounds[2] || r.v > p.bounds[3]) return r;
    r.radius = p.th * p.scale[oct];
    if (ps.forward) { r.minL = oct; r.maxL = -1; } else if (ps.backward) { r.minL = 0; r.maxL = oct; } else { r.minL = oct - 1; r.maxL = oct + 1; }
    r.ok = 1; return r;
}

__device__ __forceinline__ int d_hamming(const uint32_t* a, const uint4 b0, const uint4 b1) {
    const uint4 a0 = *(const uint4*)a, a1 = *(const uint4*)(a + 4);
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(MT_NT) void k_search_by_projection(MatchParams p, MatchArrays a) {
    __shared__ int changed, hist[HISTO_LENGTH], keep[HISTO_LENGTH], nmatch;
    const int b = blockIdx.x, t = threadIdx.x, nL = min(a.nLast[b], p.capLast), nC = min(a.nCur[b], p.capCur);
    const MatchPose ps = a.pose[b];
    const float* x3Dw = a.x3Dw + (size_t)b * p.capLast * 3; const uint8_t* lflags = a.lastFlags + (size_t)b * p.capLast; const int* loct = a.lastOctave + (size_t)b * p.capLast;
    const float* lang = a.lastAngle + (size_t)b * p.capLast; const uint32_t* ldesc = a.lastDesc + (size_t)b * p.capLast * 8;
    const float* cxy = a.curUnXY + (size_t)b * p.capCur * 2; const int* coct = a.curOctave + (size_t)b * p.capCur; const float* cang = a.curAngle + (size_t)b * p.capCur;
    const float* cur = a.curURight + (size_t)b * p.capCur; const uint32_t* cdesc = a.curDesc + (size_t)b * p.capCur * 8;
    const int* gs = a.gridStart + (size_t)b * 3073; const int* gi = a.gridIdx + (size_t)b * p.capCur; const uint8_t* taken0 = a.curTaken ? a.curTaken + (size_t)b * p.capCur : nullptr;
    int* choice = a.choice + (size_t)b * p.capLast; int* minOwner = a.minOwner + (size_t)b * p.capCur; int* matchOfCur = a.matchOfCur + (size_t)b * p.capCur;
    const float minX = p.bounds[0], minY = p.bounds[2];
    const float wInv = 64.f / (float)(p.bounds[1] - p.bounds[0]), hInv = 48.f / (float)(p.bounds[3] - p.bounds[2]);

    for (int i = t; i < nL; i += MT_NT) choice[i] = -1;
    int round = 0;
    for (;;) {
        for (int c = t; c < nC; c += MT_NT) minOwner[c] = 0x7fffffff;
        if (t == 0) changed = 0;
        __syncthreads();
        for (int i = t; i < nL; i += MT_NT) { const int c = choice[i]; if (c >= 0 && (lflags[i] & 2)) atomicMin(&minOwner[c], i); }
        __syncthreads();
        for (int i = t; i < nL; i += MT_NT) {
            int best = -1;
            if (lflags[i] & 1) {
                const int oct = loct[i];
                const Proj pr = d_project(p, ps, x3Dw + 3 * i, oct);
                if (pr.ok) {
                    const float x = pr.u, y = pr.v, r = pr.radius;
                    const int x0 = max(0, (int)floorf((x - minX - r) * wInv)), x1 = min(63, (int)ceilf((x - minX + r) * wInv));
                    const int y0 = max(0, (int)floorf((y - minY - r) * hInv)), y1 = min(47, (int)ceilf((y - minY + r) * hInv));
                    if (x0 < 64 && x1 >= 0 && y0 < 48 && y1 >= 0) {
                        const bool checkLevels = (pr.minL > 0) || (pr.maxL >= 0);
                        const uint4 d0 = *(const uint4*)(ldesc + 8 * i), d1 = *(const uint4*)(ldesc + 8 * i + 4);
                        const float ur = x - p.bf * pr.invzc;
                        int bestDist = 256;
                        for (int ix = x0; ix <= x1; ix++) {
                            const int jb = gs[ix * 48 + y0], je = gs[ix * 48 + y1 + 1];           // cells (ix, y0..y1) are contiguous in the CSR
                            for (int j = jb; j < je; j++) {
                                const int k = gi[j];
                                if (checkLevels) { const int o = coct[k]; if (o < pr.minL) continue; if (pr.maxL >= 0 && o > pr.maxL) continue; }
                                const float dx = cxy[2 * k] - x, dy = cxy[2 * k + 1] - y;
                                if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                                if ((taken0 && taken0[k]) || minOwner[k] < i) continue;
                                const float urk = cur[k];
                                if (urk > 0) { const float er = fabsf(ur - urk); if (er > r) continue; }
                                const int dist = d_hamming(cdesc + 8 * k, d0, d1);
                                if (dist < bestDist) { bestDist = dist; best = k; }
                            }
                        }
                        if (bestDist > TH_HIGH) best = -1;
                    }
                }
            }
            if (best != choice[i]) { choice[i] = best; changed = 1; }
        }
        __syncthreads();
        round++;
        const int ch = changed;
        __syncthreads();
        if (!ch || round > nL) break;
    }
    // assignments -> CurrentFrame.mvpMapPoints (the later point wins), rotation histogram, three maxima, removal
    for (int c = t; c < nC; c += MT_NT) matchOfCur[c] = -1;
    if (t < HISTO_LENGTH) hist[t] = 0;
    if (t == 0) nmatch = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = t; i < nL; i += MT_NT) {
        const int c = choice[i]; if (c < 0) continue;
        atomicMax(&matchOfCur[c], i); atomicAdd(&nmatch, 1);
        if (p.checkOrientation) { float rot = lang[i] - cang[c]; if (rot < 0.0f) rot += 360.0f; int bin = (int)roundf(rot * factor); if (bin == HISTO_LENGTH) bin = 0; atomicAdd(&hist[bin], 1); }
    }
    __syncthreads();
    if (p.checkOrientation) {
        if (t == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; } else if (max3 < 0.1f * (float)max1) ind3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) keep[i] = (i == ind1 || i == ind2 || i == ind3);
        }
        __syncthreads();
        for (int i = t; i < nL; i += MT_NT) {
            const int c = choice[i]; if (c < 0) continue;
            float rot = lang[i] - cang[c]; if (rot < 0.0f) rot += 360.0f; int bin = (int)roundf(rot * factor); if (bin == HISTO_LENGTH) bin = 0;
            if (!keep[bin]) { matchOfCur[c] = -2; atomicAdd(&nmatch, -1); }              // -2 < every index: a removal always wins
        }
        __syncthreads();
        for (int c = t; c < nC; c += MT_NT) if (matchOfCur[c] == -2) matchOfCur[c] = -1;
    }
    __syncthreads();
    if (t == 0) { a.nmatches[b] = nmatch; a.rounds[b] = round; }
}

int launch_search_by_projection(const MatchParams& p, const MatchArrays& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_search_by_projection, dim3(B), dim3(MT_NT), 0, s, p, a);
    HIP_TRY(hipGetLastError());
    return SIND_OK;
}

}  // namespace sind
