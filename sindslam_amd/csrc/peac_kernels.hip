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
#define PG_U 4                      /* visits (phase 2) and keys (phase 3) a thread works on per pass: their loads are in flight together */
#ifndef PG_LDS_N
#define PG_LDS_N 2048               /* levels of at most this many seeds keep their frontier and their visit keys in LDS (most levels are a few hundred seeds) */
#endif

// slot of a pixel: the seeds that sit on it in one level.  One 64-bit word carries the level tag, the count and the first seed, so the common case (one
// seed) costs one load; further seeds of the same pixel and level (a pixel taken over twice within a level) go to a side array.
//   bits 63..37 level | 36..35 count - 1 | 34..8 rank of seed 0 | 7..0 plane of seed 0
__device__ __forceinline__ unsigned long long pg_slot(unsigned level, unsigned cnt, unsigned rank, unsigned pl) {
    return ((unsigned long long)level << 37) | ((unsigned long long)(cnt - 1) << 35) | ((unsigned long long)rank << 8) | pl;
}

// one visit of the reference's state machine on the visited pixel (AHCPlaneFitter.hpp:560-594): true = the plane takes the pixel over and it becomes a seed
__device__ __forceinline__ bool pg_visit(int& trail, float& od, int pl, bool has_pt, double px, double py, double pz, const PeacGrowPlane* s_pl, uint8_t* pairSeen, int nPl) {
    if (trail <= -6 || trail == pl) return false;
    const PeacGrowPlane& S = s_pl[pl];
    float cdist = -1.f;
    if (has_pt) cdist = (float)fabs(S.n[0] * (px - S.c[0]) + S.n[1] * (py - S.c[1]) + S.n[2] * (pz - S.c[2]));
    if (has_pt && (double)cdist * (double)cdist < S.thr) {
        if (trail >= 0) pairSeen[pl * nPl + trail] = 1;
        if (cdist < od) { trail = pl; od = cdist; return true; }
        if (trail < 0) trail -= 1;
    } else if (trail < 0) trail -= 1;
    return false;
}
struct PgPixel { int c, cx, cy, trail0; float od0; unsigned dep; };
// general form of a pixel's level (any number of seeds per neighbour): list, insertion sort, state machine.  Rare (a neighbour holds two seeds of the level),
// kept out of line so that its dynamically indexed arrays do not weigh on the common path below.
__device__ __attribute__((noinline)) void pg_pixel_general(const PgPixel P, unsigned long long sl0, unsigned long long sl1, unsigned long long sl2, unsigned long long sl3, unsigned level,
                                                           int v, int W, const PeacGrowArgs& A, const PeacGrowPlane* s_pl, uint8_t* pairSeen, int nPl, const unsigned* slotExt,
                                                           int8_t* member, float* dist, unsigned* payload, int* s_err) {
    unsigned key[4 * PEAC_GROW_SLOTS]; unsigned char vpl[4 * PEAC_GROW_SLOTS]; int nv = 0;
    const int qoff[4] = {-1, 1, -W, W}; const unsigned jrel[4] = {1u, 0u, 3u, 2u}; const unsigned long long sl[4] = {sl0, sl1, sl2, sl3};
    for (int d = 0; d < 4; d++) {
        const unsigned long long s64 = sl[d];
        if ((unsigned)(s64 >> 37) != level) continue;
        key[nv] = ((unsigned)(s64 >> 8) & 0x7FFFFFFu) * 4u + jrel[d]; vpl[nv] = (unsigned char)(s64 & 0xFFu); nv++;
        const int extra = (int)((s64 >> 35) & 3u);
        for (int e = 0; e < extra; e++) { const unsigned ent = slotExt[(size_t)(P.c + qoff[d]) * (PEAC_GROW_SLOTS - 1) + e]; key[nv] = (ent >> 8) * 4u + jrel[d]; vpl[nv] = (unsigned char)(ent & 0xFFu); nv++; }
    }
    unsigned kown = PG_EMPTY; for (int i = 0; i < nv; i++) if ((int)vpl[i] != P.trail0) kown = min(kown, key[i]);
    if (kown != (unsigned)v) return;                        // another active visit's thread owns the pixel in this level
    for (int a = 1; a < nv; a++) { const unsigned k = key[a]; const unsigned char p = vpl[a]; int b = a - 1; while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; vpl[b + 1] = vpl[b]; b--; } key[b + 1] = k; vpl[b + 1] = p; }
    int trail = P.trail0; float od = P.od0;
    const float dd = (float)P.dep; const bool has_pt = !(dd < 1e-3f);
    double px = 0, py = 0, pz = 0;
    if (has_pt) { const float z = dd * A.inv_scale; px = (double)((P.cx - A.cx) * z / A.fx); py = (double)((P.cy - A.cy) * z / A.fy); pz = (double)z; }
    unsigned pushMask = 0; int npush = 0;
    for (int i = 0; i < nv; i++) if (pg_visit(trail, od, vpl[i], has_pt, px, py, pz, s_pl, pairSeen, nPl)) { pushMask |= 1u << i; npush++; }
    if (trail != P.trail0) member[P.c] = (int8_t)trail;
    if (npush) dist[P.c] = od;
    if (npush > PEAC_GROW_SLOTS) { *s_err = PG_ERR_SLOTS; npush = PEAC_GROW_SLOTS; }
    int idx = 0;
    for (int i = 0; i < nv; i++)
        if ((pushMask >> i) & 1u) { if (idx < PEAC_GROW_SLOTS) payload[key[i]] = ((unsigned)(npush - 1) << 30) | ((unsigned)idx << 28) | ((unsigned)vpl[i] << 20) | (unsigned)P.c; idx++; }
}

// exclusive scan of one count per thread over the workgroup (<= 1024 threads): DPP-free wave scan by shuffles, wave totals through LDS
__device__ __forceinline__ unsigned pg_block_excl_scan(unsigned cnt, unsigned* s_wave /* >= 17 words */, unsigned& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned inc = cnt;
    #pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { const unsigned c = s_wave[w]; if (w < wv) base += c; tot += c; }
    __syncthreads();
    total = tot;
    return base + inc - cnt;
}

// NT: threads of the workgroup.  512 for launches of many frames (a workgroup per frame, several per compute unit); 1024 when a launch holds one or two frames (the one-frame
// call of the drop-in path): a level of a 1280 x 720 frame is ~2 000 seeds = 8 000 visits, ~3 500 of them active -- 17.0 -> 14.5 ms per frame there, 640 x 480 unchanged at 3 - 5 ms
// (phase 2b, two dependent rounds of global loads per batch of active visits, is 2/3 of a level's ~30 us: latency, not bandwidth; a larger LDS level capacity changed nothing)
template <int NT>
__global__ void __launch_bounds__(NT) k_peac_grow(PeacGrowArgs A) {
    const int f = blockIdx.x, tid = threadIdx.x, W = A.W, H = A.H, N = W * H, Nw = W / 16, NB = Nw * (H / 16);
    const uint8_t* in = A.in + (size_t)f * A.in_stride;
    const PeacGrowHdr hdr = *reinterpret_cast<const PeacGrowHdr*>(in);
    int* status = reinterpret_cast<int*>(A.status) + 4 * f;            // status, levels, seeds processed, reserved
    int8_t* member = A.member + (size_t)f * N;
    uint8_t* pairSeen = A.pair_seen + (size_t)f * PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES;
    if (hdr.skip) { if (tid == 0) { status[0] = PG_SKIPPED; status[1] = 0; status[2] = 0; } return; }      // the caller grows this frame on the host
    const uint16_t* depth = A.depth_base + (size_t)hdr.depth_index * N;
    float* dist = A.dist + (size_t)f * N;
    unsigned long long* slot = A.slot + (size_t)f * N;
    unsigned* slotExt = A.slot_ext + (size_t)f * N * (PEAC_GROW_SLOTS - 1);
    unsigned* front[2] = {A.frontier + (size_t)f * 2 * PG_FRONT_CAP, A.frontier + (size_t)f * 2 * PG_FRONT_CAP + PG_FRONT_CAP};
    unsigned* payload = A.payload + (size_t)f * 4 * PG_FRONT_CAP;
    unsigned* active_g = A.active + (size_t)f * 4 * PG_FRONT_CAP;
    const int nPl = hdr.nPl;

    __shared__ PeacGrowPlane s_pl[PEAC_GROW_MAX_PLANES];
    __shared__ int8_t s_blk[PG_MAX_BLOCKS];
    __shared__ unsigned s_wave[20];
    __shared__ int s_err;
    // A level costs global round trips, not bandwidth (a few hundred visits, ~500 levels deep): with the frontier and the visit keys of the small levels in
    // LDS a level is one round of loads (slots, state, depth) and one round of stores instead of three and two (measured: 13 -> ~5 us per level)
    __shared__ unsigned s_front[2][PG_LDS_N];
    __shared__ __attribute__((aligned(16))) unsigned s_pay[4 * PG_LDS_N];
    __shared__ unsigned s_act[4 * PG_LDS_N];             // this level's active visits (phase 2a -> 2b)
    __shared__ unsigned s_nact;
    {
        const double* src = reinterpret_cast<const double*>(in + PG_OFF_PLANES); double* dst = reinterpret_cast<double*>(s_pl);
        for (int i = tid; i < nPl * 8; i += blockDim.x) dst[i] = src[i];
        const int8_t* bsrc = reinterpret_cast<const int8_t*>(in + PG_OFF_BLOCKS);
        for (int i = tid; i < NB; i += blockDim.x) s_blk[i] = bsrc[i];
        if (tid == 0) { s_err = 0; s_nact = 0; }
    }
    __syncthreads();
    // membership from the eroded block map, best distance = "none yet", empty slot map, nobody met anybody
    for (int i = tid; i < N / 4; i += blockDim.x) {
        const int p = 4 * i, y = p / W, x = p - y * W;                  // four pixels of one row and one block (W is a multiple of 16)
        const int8_t m = s_blk[(y >> 4) * Nw + (x >> 4)];
        const unsigned mm = (uint8_t)m; reinterpret_cast<unsigned*>(member)[i] = mm | (mm << 8) | (mm << 16) | (mm << 24);
        reinterpret_cast<float4*>(dist)[i] = make_float4(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
        reinterpret_cast<uint4*>(slot)[2 * i] = make_uint4(0, 0, 0, 0); reinterpret_cast<uint4*>(slot)[2 * i + 1] = make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < nPl * nPl; i += blockDim.x) pairSeen[i] = 0;
    // level 1: the initial seeds, in the host's order
    int n = hdr.nSeeds, level = 1, cur = 0; long long processed = 0;
#ifdef PG_PROFILE
    long long tp[5] = {0, 0, 0, 0, 0}, tl = wall_clock64(); long long nactsum = 0; int nsmall = 0;
    #define PG_LAP(i) { const long long t_ = wall_clock64(); tp[i] += t_ - tl; tl = t_; }
#else
    #define PG_LAP(i)
#endif
    const unsigned* seeds0 = reinterpret_cast<const unsigned*>(in + PG_OFF_SEEDS(NB));
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) {
        const unsigned w = seeds0[r]; front[0][r] = w; if (r < PG_LDS_N) s_front[0][r] = w;
        const unsigned c = w & 0xFFFFFu, pl = (w >> 20) & 0xFFu, idx = (w >> 28) & 3u, cnt = (w >> 30) + 1u;
        if (idx == 0) slot[c] = pg_slot(level, cnt, r, pl); else slotExt[(size_t)c * (PEAC_GROW_SLOTS - 1) + idx - 1] = ((unsigned)r << 8) | pl;
    }
    __syncthreads();
    PG_LAP(0)
    while (n > 0) {
        processed += n;
        const unsigned* F = front[cur];
        const int nvis = 4 * n;
        const bool small = n <= PG_LDS_N;                    // wave-uniform: this level's frontier, keys and active list live in LDS
        // ---- phase 2a: classify the visits.  Three of four visits of a flood fill return at once (the pixel already belongs to the visiting plane, lies in an
        // eroded block or has used up its tries) -- unless another plane visits the same pixel in this level, and that visit is then ACTIVE itself.  So only the
        // active visits (plane != the pixel's owner at the start of the level) go on to phase 2b, compacted, and every lane there has work to do.
        for (int v0 = 0; v0 < nvis; v0 += PG_U * blockDim.x) {
            int vv[PG_U]; unsigned wseed[PG_U];
            #pragma unroll
            for (int u = 0; u < PG_U; u++) { vv[u] = v0 + u * blockDim.x + tid; wseed[u] = vv[u] < nvis ? (small ? s_front[cur][vv[u] >> 2] : F[vv[u] >> 2]) : 0u; }
            int mem[PG_U]; bool cand[PG_U];
            #pragma unroll
            for (int u = 0; u < PG_U; u++) {
                const int j = vv[u] & 3, cs = (int)(wseed[u] & 0xFFFFFu), sy = (int)__umulhi((unsigned)cs, A.w_magic), sx = cs - sy * W;
                int cx = sx, cy = sy;
                if (j == 0) cx--; else if (j == 1) cx++; else if (j == 2) cy--; else cy++;
                cand[u] = vv[u] < nvis && cx >= 0 && cx < W && cy >= 0 && cy < H && s_blk[(cy >> 4) * Nw + (cx >> 4)] < 0;      // inside the image, not in an eroded block
                mem[u] = cand[u] ? (int)member[cy * W + cx] : 0;
            }
            #pragma unroll
            for (int u = 0; u < PG_U; u++) {
                if (vv[u] < nvis) { if (small) s_pay[vv[u]] = PG_EMPTY; else payload[vv[u]] = PG_EMPTY; }      // "no push" until phase 2b says otherwise
                const bool active = cand[u] && mem[u] > -6 && mem[u] != (int)((wseed[u] >> 20) & 0xFFu);
                const unsigned long long b = __ballot(active);
                if (b) {
                    unsigned at = 0;
                    if ((tid & 63) == 0) at = atomicAdd(&s_nact, (unsigned)__popcll(b));
                    at = __shfl(at, 0) + __popcll(b & ((1ull << (tid & 63)) - 1));
                    // the pixel's owner as of the START of the level travels with the entry: phase 2b decides ownership from it, while owners already rewrite member[]
                    if (active) { const unsigned e = ((unsigned)(mem[u] & 0xFF) << 24) | (unsigned)vv[u]; if (small) s_act[at] = e; else active_g[at] = e; }
                }
            }
        }
        __syncthreads();
        PG_LAP(1)
        // ---- phase 2b: the pixels with at least one active visit.  The active visit with the smallest key owns the pixel and runs ALL its visits in key order.
        const int nact = (int)s_nact;
        for (int i0 = 0; i0 < nact; i0 += 2 * blockDim.x) {
            int vv[2], cxa[2], cya[2]; bool have[2];
            unsigned long long sl[2][4]; int mem[2]; float odv[2]; unsigned dep[2];
            #pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * blockDim.x + tid; have[u] = i < nact;
                const unsigned e = have[u] ? (small ? s_act[i] : active_g[i]) : 0u;
                vv[u] = (int)(e & 0xFFFFFFu); mem[u] = (int)(int8_t)(e >> 24);
            }
            #pragma unroll
            for (int u = 0; u < 2; u++) {
                const unsigned wseed = small ? s_front[cur][vv[u] >> 2] : F[vv[u] >> 2];
                const int j = vv[u] & 3, cs = (int)(wseed & 0xFFFFFu), sy = (int)__umulhi((unsigned)cs, A.w_magic), sx = cs - sy * W;
                int cx = sx, cy = sy;
                if (j == 0) cx--; else if (j == 1) cx++; else if (j == 2) cy--; else cy++;
                if (!have[u]) { cx = 0; cy = 0; }
                cxa[u] = cx; cya[u] = cy;
                const int c = cy * W + cx;
                // the seeds on c's four neighbours (left, right, upper, lower), c's own state and depth
                sl[u][0] = (have[u] && cx > 0) ? slot[c - 1] : 0ull; sl[u][1] = (have[u] && cx < W - 1) ? slot[c + 1] : 0ull;
                sl[u][2] = (have[u] && cy > 0) ? slot[c - W] : 0ull; sl[u][3] = (have[u] && cy < H - 1) ? slot[c + W] : 0ull;
                odv[u] = dist[c]; dep[u] = depth[c];
            }
            #pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!have[u]) continue;
                const int v = vv[u], cx = cxa[u], cy = cya[u], c = cy * W + cx, trail0 = mem[u];
                // every visit to c in this level: at most one seed per neighbour in the common case (q left of c visits c as its right neighbour, j = 1; and so on)
                const unsigned jrel[4] = {1u, 0u, 3u, 2u};
                unsigned k4[4]; int p4[4]; bool extra = false; unsigned kown = PG_EMPTY;
                #pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned long long s64 = sl[u][d]; const bool on = (unsigned)(s64 >> 37) == (unsigned)level;
                    k4[d] = on ? ((unsigned)(s64 >> 8) & 0x7FFFFFFu) * 4u + jrel[d] : PG_EMPTY; p4[d] = (int)(s64 & 0xFFu);
                    extra |= on && ((s64 >> 35) & 3u) != 0;
                    if (on && p4[d] != trail0) kown = min(kown, k4[d]);
                }
                if (extra) { const PgPixel P{c, cx, cy, trail0, odv[u], dep[u]}; pg_pixel_general(P, sl[u][0], sl[u][1], sl[u][2], sl[u][3], (unsigned)level, v, W, A, s_pl, pairSeen, nPl, slotExt, member, dist, small ? s_pay : payload, &s_err); continue; }
                if (kown != (unsigned)v) continue;                      // another active visit's thread owns pixel c in this level
                int trail = trail0; float od = odv[u];
                const float dd = (float)dep[u]; const bool has_pt = !(dd < 1e-3f);
                double px = 0, py = 0, pz = 0;
                if (has_pt) { const float z = dd * A.inv_scale; px = (double)((cx - A.cx) * z / A.fx); py = (double)((cy - A.cy) * z / A.fy); pz = (double)z; }
                unsigned rem[4] = {k4[0], k4[1], k4[2], k4[3]}, out4[4] = {PG_EMPTY, PG_EMPTY, PG_EMPTY, PG_EMPTY}; int npush = 0;
                #pragma unroll
                for (int t = 0; t < 4; t++) {                           // the visits in key order: smallest remaining key first
                    const unsigned km = min(min(rem[0], rem[1]), min(rem[2], rem[3]));
                    if (km == PG_EMPTY) break;
                    const int d = km == rem[0] ? 0 : km == rem[1] ? 1 : km == rem[2] ? 2 : 3;
                    const int pl = d == 0 ? p4[0] : d == 1 ? p4[1] : d == 2 ? p4[2] : p4[3];
                    if (pg_visit(trail, od, pl, has_pt, px, py, pz, s_pl, pairSeen, nPl)) {
                        const unsigned wv = ((unsigned)npush << 28) | ((unsigned)pl << 20) | (unsigned)c; npush++;
                        #pragma unroll
                        for (int q = 0; q < 4; q++) if (q == d) out4[q] = wv;
                    }
                    #pragma unroll
                    for (int q = 0; q < 4; q++) if (q == d) rem[q] = PG_EMPTY;
                }
                if (trail != trail0) member[c] = (int8_t)trail;
                if (npush) {
                    dist[c] = od;
                    #pragma unroll
                    for (int q = 0; q < 4; q++) if (out4[q] != PG_EMPTY) { const unsigned wv = out4[q] | ((unsigned)(npush - 1) << 30); if (small) s_pay[k4[q]] = wv; else payload[k4[q]] = wv; }
                }
            }
        }
        __syncthreads();
        PG_LAP(2)
        // ---- phase 3: ranks of the next level = exclusive scan over the pushing visits (PG_U consecutive keys per thread); its frontier and its slots
        unsigned* Fn = front[cur ^ 1];
        unsigned base = 0;
        if (tid == 0) s_nact = 0;
        for (int v0 = 0; v0 < nvis; v0 += PG_U * blockDim.x) {
            const int vb = v0 + PG_U * tid;                              // nvis is a multiple of 4 = PG_U: a thread's four keys are all inside or all outside
            uint4 w4 = make_uint4(PG_EMPTY, PG_EMPTY, PG_EMPTY, PG_EMPTY);
            if (vb < nvis) w4 = small ? *reinterpret_cast<const uint4*>(s_pay + vb) : *reinterpret_cast<const uint4*>(payload + vb);
            const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
            unsigned mine = 0;
            #pragma unroll
            for (int u = 0; u < 4; u++) mine += ws[u] != PG_EMPTY;
            unsigned tot; unsigned pos = base + pg_block_excl_scan(mine, s_wave, tot);
            #pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned w = ws[u];
                if (w == PG_EMPTY) continue;
                if (pos < (unsigned)PG_FRONT_CAP) {
                    Fn[pos] = w; if (pos < (unsigned)PG_LDS_N) s_front[cur ^ 1][pos] = w;
                    const unsigned c = w & 0xFFFFFu, pl = (w >> 20) & 0xFFu, idx = (w >> 28) & 3u, cnt = (w >> 30) + 1u;
                    if (idx == 0) slot[c] = pg_slot(level + 1, cnt, pos, pl); else slotExt[(size_t)c * (PEAC_GROW_SLOTS - 1) + idx - 1] = (pos << 8) | pl;
                } else s_err = PG_ERR_FRONTIER;
                pos++;
            }
            base += tot;
        }
        __syncthreads();
        PG_LAP(3)
#ifdef PG_PROFILE
        nactsum += nact; nsmall += small;
#endif
        if (s_err || level >= PG_MAX_LEVELS) break;
        n = (int)min(base, (unsigned)PG_FRONT_CAP); cur ^= 1; level++;
    }
#ifdef PG_PROFILE
    if (tid == 0) printf("[pg] frame %d levels %d (small %d) seeds %lld active %lld | init %.1f us  2a %.1f  2b %.1f  3 %.1f\n", f, level, nsmall, processed, nactsum, tp[0] * 0.01, tp[1] * 0.01, tp[2] * 0.01, tp[3] * 0.01);
#endif
    if (tid == 0) { status[0] = s_err ? s_err : (n > 0 ? PG_ERR_LEVELS : PG_OK); status[1] = level; status[2] = (int)min(processed, (long long)0x7fffffff); }
}

int launch_peac_grow(hipStream_t s, const PeacGrowArgs& A, int frames) {
    static_assert(PG_U == 4, "phase 3 reads four keys per thread as one 16-byte word");
    if (frames < 1 || A.W % 16 || A.H % 16 || (A.W / 16) * (A.H / 16) > PG_MAX_BLOCKS || (size_t)A.W * A.H > (1u << 20)) { sind_set_error("peac_grow: unsupported size %d x %d", A.W, A.H); return SIND_E_ARG; }
    if (frames <= 2) hipLaunchKernelGGL(k_peac_grow<1024>, dim3(frames), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL(k_peac_grow<PG_THREADS>, dim3(frames), dim3(PG_THREADS), 0, s, A);
    return SIND_OK;
}

}  // namespace sind
