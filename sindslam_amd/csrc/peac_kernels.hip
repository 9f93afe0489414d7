// PEAC region grow on the GPU: the pixel-level refinement of the plane borders (reference include/PEAC/AHCPlaneFitter.hpp:546-601 floodFill, called from
// refineDetails :274-400), the largest serial host stage of CalOccluded until round 3 (4.6 ms of host CPU per frame: a FIFO over ~190 k seeds, ~760 k visits).
//
// The reference pops seeds (pixel, plane) from a FIFO; a seed visits its left / right / upper / lower neighbour in that order; a visit runs a small state
// machine on the VISITED pixel alone (owner plane or number of failed tries, best distance so far): tries are capped at five, a closer plane takes the pixel
// over and pushes it as a new seed.  What a visit does depends on the pixel's own state and on the visiting plane's constants only, so the FIFO order matters
// exactly through the ORDER OF THE VISITS TO EACH PIXEL.  That order is reproduced without the FIFO:
//   * the FIFO is a breadth-first traversal: level 0 = the initial seeds in their given order, level k + 1 = the seeds pushed while level k is processed, in
//     the order (rank of the pushing seed, neighbour index 0..3) = "visit key";
//   * within a level, the visits to pixel c come from seeds that sit on c's four neighbours; every seed of the level is entered in a per-pixel slot map
//     (rank + plane, up to PEAC_GROW_SLOTS seeds per pixel and level), so the thread of ANY visit to c can list all of them; the thread that holds the
//     smallest key owns the pixel for this level, sorts the visits by key and runs the reference's state machine over them in that order;
//   * a visit that pushes marks payload[key]; an exclusive scan over the keys gives the ranks of the next level (stable compaction).
// One workgroup per frame walks all levels of its frame (barriers between the three phases of a level; every level of a 640 x 480 frame is a few thousand
// visits at most, ~140-500 levels), a launch takes a chunk of frames.  Capacities are fixed (planes, seeds per level, seeds per pixel and level); a frame
// that exceeds one reports it in `status` and the caller runs the host statement of the same FIFO (host/peac.cpp grow_host) for that frame.
// Bit-exactness: the distance test repeats the host's operations (float cloud point, FP64 dot product, float cast; the library is built -ffp-contract=off).
#include "peac_grow.hpp"

namespace sind {

static constexpr unsigned PG_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned pg_block_excl_scan(unsigned flag, unsigned* s_wave /* >= 17 words */, unsigned& total) {
    // exclusive scan of one flag per thread over the workgroup (<= 1024 threads): ballot prefix inside a wave, wave totals through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long b = __ballot(flag != 0);
    const unsigned pre = __popcll(b & ((1ull << lane) - 1));
    if (lane == 0) s_wave[wv] = __popcll(b);
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const unsigned c = s_wave[w]; if (w < wv) base += c; tot += c; }
    __syncthreads();
    total = tot;
    return base + pre;
}

__global__ void __launch_bounds__(PG_THREADS) k_peac_grow(PeacGrowArgs A) {
    const int f = blockIdx.x, tid = threadIdx.x, W = A.W, H = A.H, N = W * H, Nw = W / 16, NB = Nw * (H / 16);
    const uint8_t* in = A.in + (size_t)f * A.in_stride;
    const PeacGrowHdr hdr = *reinterpret_cast<const PeacGrowHdr*>(in);
    int* status = reinterpret_cast<int*>(A.status) + 4 * f;            // status, levels, seeds processed, reserved
    int8_t* member = A.member + (size_t)f * N;
    uint8_t* pairSeen = A.pair_seen + (size_t)f * PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES;
    if (hdr.skip) { if (tid == 0) { status[0] = PG_SKIPPED; status[1] = 0; status[2] = 0; } return; }      // the caller grows this frame on the host
    const uint16_t* depth = A.depth_base + (size_t)hdr.depth_index * N;
    float* dist = A.dist + (size_t)f * N;
    unsigned* slotTag = A.slot_tag + (size_t)f * N;
    unsigned* slotEnt = A.slot_ent + (size_t)f * N * PEAC_GROW_SLOTS;
    unsigned* front[2] = {A.frontier + (size_t)f * 2 * PG_FRONT_CAP, A.frontier + (size_t)f * 2 * PG_FRONT_CAP + PG_FRONT_CAP};
    unsigned* payload = A.payload + (size_t)f * 4 * PG_FRONT_CAP;
    const int nPl = hdr.nPl;

    __shared__ PeacGrowPlane s_pl[PEAC_GROW_MAX_PLANES];
    __shared__ int8_t s_blk[PG_MAX_BLOCKS];
    __shared__ unsigned s_wave[20];
    __shared__ int s_err;
    {
        const double* src = reinterpret_cast<const double*>(in + PG_OFF_PLANES); double* dst = reinterpret_cast<double*>(s_pl);
        for (int i = tid; i < nPl * 8; i += blockDim.x) dst[i] = src[i];
        const int8_t* bsrc = reinterpret_cast<const int8_t*>(in + PG_OFF_BLOCKS);
        for (int i = tid; i < NB; i += blockDim.x) s_blk[i] = bsrc[i];
        if (tid == 0) s_err = 0;
    }
    __syncthreads();
    // membership from the eroded block map, best distance = "none yet", empty slot map, nobody met anybody
    for (int i = tid; i < N / 4; i += blockDim.x) {
        const int p = 4 * i, y = p / W, x = p - y * W;                  // four pixels of one row and one block (W is a multiple of 16)
        const int8_t m = s_blk[(y >> 4) * Nw + (x >> 4)];
        const unsigned mm = (uint8_t)m; reinterpret_cast<unsigned*>(member)[i] = mm | (mm << 8) | (mm << 16) | (mm << 24);
        reinterpret_cast<float4*>(dist)[i] = make_float4(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
        reinterpret_cast<uint4*>(slotTag)[i] = make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < nPl * nPl; i += blockDim.x) pairSeen[i] = 0;
    // level 1: the initial seeds, in the host's order
    int n = hdr.nSeeds, level = 1, cur = 0; long long processed = 0;
    const unsigned* seeds0 = reinterpret_cast<const unsigned*>(in + PG_OFF_SEEDS(NB));
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) {
        const unsigned w = seeds0[r]; front[0][r] = w;
        const unsigned c = w & 0xFFFFFu, pl = (w >> 20) & 0xFFu, idx = (w >> 28) & 3u, cnt = (w >> 30) + 1u;
        slotEnt[(size_t)c * PEAC_GROW_SLOTS + idx] = ((unsigned)r << 8) | pl; slotTag[c] = ((unsigned)level << 3) | cnt;
    }
    __syncthreads();
    while (n > 0) {
        processed += n;
        // ---- phase 2: the visits of this level
        const unsigned* F = front[cur];
        const int nvis = 4 * n;
        for (int v = tid; v < nvis; v += blockDim.x) {
            const int r = v >> 2, j = v & 3;
            const unsigned w = F[r]; const int cs = (int)(w & 0xFFFFFu), sy = cs / W, sx = cs - sy * W;
            int cx = sx, cy = sy;
            if (j == 0) cx--; else if (j == 1) cx++; else if (j == 2) cy--; else cy++;
            if (cx < 0 || cx >= W || cy < 0 || cy >= H) { payload[v] = PG_EMPTY; continue; }        // the reference does not visit outside the image
            const int c = cy * W + cx;
            // every visit to c in this level: seeds on c's four neighbours (q left of c visits c as its right neighbour, j = 1; and so on)
            unsigned key[4 * PEAC_GROW_SLOTS]; unsigned char vpl[4 * PEAC_GROW_SLOTS]; int nv = 0;
            const int qx[4] = {cx - 1, cx + 1, cx, cx}, qy[4] = {cy, cy, cy - 1, cy + 1}; const int jrel[4] = {1, 0, 3, 2};
            unsigned tags[4];
            #pragma unroll
            for (int d = 0; d < 4; d++) tags[d] = (qx[d] >= 0 && qx[d] < W && qy[d] >= 0 && qy[d] < H) ? slotTag[qy[d] * W + qx[d]] : 0u;
            #pragma unroll
            for (int d = 0; d < 4; d++) {
                if ((tags[d] >> 3) != (unsigned)level) continue;
                const int cntq = (int)(tags[d] & 7u); const size_t q = (size_t)(qy[d] * W + qx[d]) * PEAC_GROW_SLOTS;
                for (int e = 0; e < cntq; e++) { const unsigned ent = slotEnt[q + e]; key[nv] = (ent >> 8) * 4u + (unsigned)jrel[d]; vpl[nv] = (unsigned char)(ent & 0xFFu); nv++; }
            }
            unsigned kmin = key[0]; for (int i = 1; i < nv; i++) kmin = min(kmin, key[i]);
            if (kmin != (unsigned)v) continue;                          // another visit's thread owns pixel c in this level
            for (int a = 1; a < nv; a++) { const unsigned k = key[a]; const unsigned char p = vpl[a]; int b = a - 1; while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; vpl[b + 1] = vpl[b]; b--; } key[b + 1] = k; vpl[b + 1] = p; }
            if (s_blk[(cy >> 4) * Nw + (cx >> 4)] >= 0) { for (int i = 0; i < nv; i++) payload[key[i]] = PG_EMPTY; continue; }      // pixel of an eroded block: never revisited
            int trail = member[c]; float od = dist[c];
            const float dd = (float)depth[c]; const bool has_pt = !(dd < 1e-3f);
            double px = 0, py = 0, pz = 0;
            if (has_pt) { const float z = dd * A.inv_scale; px = (double)((cx - A.cx) * z / A.fx); py = (double)((cy - A.cy) * z / A.fy); pz = (double)z; }
            unsigned pushMask = 0; int npush = 0;
            for (int i = 0; i < nv; i++) {
                const int pl = vpl[i];
                if (trail <= -6 || trail == pl) continue;
                const PeacGrowPlane& S = s_pl[pl];
                float cdist = -1.f;
                if (has_pt) cdist = (float)fabs(S.n[0] * (px - S.c[0]) + S.n[1] * (py - S.c[1]) + S.n[2] * (pz - S.c[2]));
                if (has_pt && (double)cdist * (double)cdist < S.thr) {
                    if (trail >= 0) pairSeen[pl * nPl + trail] = 1;
                    if (cdist < od) { trail = pl; od = cdist; pushMask |= 1u << i; npush++; }
                    else if (trail < 0) trail -= 1;
                } else if (trail < 0) trail -= 1;
            }
            member[c] = (int8_t)trail; dist[c] = od;
            if (npush > PEAC_GROW_SLOTS) { s_err = PG_ERR_SLOTS; npush = PEAC_GROW_SLOTS; }
            int idx = 0;
            for (int i = 0; i < nv; i++) {
                unsigned out = PG_EMPTY;
                if ((pushMask >> i) & 1u) { if (idx < PEAC_GROW_SLOTS) out = ((unsigned)(npush - 1) << 30) | ((unsigned)idx << 28) | ((unsigned)vpl[i] << 20) | (unsigned)c; idx++; }
                payload[key[i]] = out;
            }
        }
        __syncthreads();
        // ---- phase 3: ranks of the next level = exclusive scan over the pushing visits; its frontier and its slots
        unsigned* Fn = front[cur ^ 1];
        unsigned base = 0;
        for (int v0 = 0; v0 < nvis; v0 += blockDim.x) {
            const int v = v0 + tid; const unsigned w = v < nvis ? payload[v] : PG_EMPTY;
            unsigned tot; const unsigned pos = base + pg_block_excl_scan(w != PG_EMPTY, s_wave, tot);
            if (w != PG_EMPTY) {
                if (pos < (unsigned)PG_FRONT_CAP) {
                    Fn[pos] = w;
                    const unsigned c = w & 0xFFFFFu, pl = (w >> 20) & 0xFFu, idx = (w >> 28) & 3u, cnt = (w >> 30) + 1u;
                    slotEnt[(size_t)c * PEAC_GROW_SLOTS + idx] = (pos << 8) | pl; slotTag[c] = ((unsigned)(level + 1) << 3) | cnt;
                } else s_err = PG_ERR_FRONTIER;
            }
            base += tot;
        }
        __syncthreads();
        if (s_err || level >= PG_MAX_LEVELS) break;
        n = (int)min(base, (unsigned)PG_FRONT_CAP); cur ^= 1; level++;
    }
    if (tid == 0) { status[0] = s_err ? s_err : (n > 0 ? PG_ERR_LEVELS : PG_OK); status[1] = level; status[2] = (int)min(processed, (long long)0x7fffffff); }
}

int launch_peac_grow(hipStream_t s, const PeacGrowArgs& A, int frames) {
    if (frames < 1 || A.W % 16 || A.H % 16 || (A.W / 16) * (A.H / 16) > PG_MAX_BLOCKS || (size_t)A.W * A.H > (1u << 20)) { sind_set_error("peac_grow: unsupported size %d x %d", A.W, A.H); return SIND_E_ARG; }
    hipLaunchKernelGGL(k_peac_grow, dim3(frames), dim3(PG_THREADS), 0, s, A);
    return SIND_OK;
}

}  // namespace sind
