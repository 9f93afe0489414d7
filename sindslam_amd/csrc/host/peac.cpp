// Host stage: agglomerative plane clustering + refinement of the SInDSLAM-modified PEAC fitter
// (reference include/PEAC/AHCPlaneFitter.hpp:186-236 run, :881-1039 initGraph, :1050-1256 ahCluster,
//  :274-400 refineDetails, :603-705 findBlockMembership, :546-594 floodFill; AHCPlaneSeg.hpp:103-134 PCA;
//  AHCParamSet.hpp:48-148 thresholds; DisjointSet.hpp).
// Split of work: the 1200 per-window second-order statistics come from the GPU (depth_kernels.hip k_peac_block_stats); the graph
// (<= 1200 nodes, priority queue, disjoint sets) is serial and stays here (part1 / part2); the pixel-level region grow along the plane
// borders (floodFill, :546-601) runs on the GPU as a level-synchronous ordered BFS (peac_kernels.hip k_peac_grow, same visiting order as
// the FIFO), with grow_host as the statement of that FIFO for inputs beyond the kernel's fixed capacities and for the CPU tests.
// Nodes live in an index pool; neighbour sets are ordered by node index (= creation order; the reference orders by heap address, which only matters for exactly tied MSEs).
// Eigen's 3x3 self-adjoint solver is replaced by a cyclic Jacobi iteration (Eigen is not available).
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <algorithm>
#include <iterator>
#include "host.hpp"
#include "peac_fit.hpp"
#include "peac_fit4.hpp"

namespace sind {
namespace {

struct Params {   // AHCParamSet.hpp:48-56 — millimetre defaults applied to metre clouds (reference quirk)
    double depthSigma = 3e-6, stdTol_init = 10, stdTol_merge = 17, z_near = 500, z_far = 6000;
    double angle_near = 10.0 * M_PI / 180.0, angle_far = 20.0 * M_PI / 180.0;
    double simMerge = std::cos(15.0 * M_PI / 180.0), simRefine = std::cos(20.0 * M_PI / 180.0);
    double mse_init(double z) const { return std::pow(depthSigma * z * z + stdTol_init, 2); }
    double mse_merge(double z) const { return std::pow(depthSigma * z * z + stdTol_merge, 2); }
    double ang_init(double z) const { const double cz = std::min(std::max(z, z_near), z_far), f = (angle_far - angle_near) / (z_far - z_near); return std::cos(f * cz + angle_near - f * z_near); }
};

struct Seg {
    double sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, syz = 0, sxz = 0; int N = 0;
    int rid = 0; double mse = 0, center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}; bool nouse = false;
    void fit() { const double m[9] = {sx, sy, sz, sxx, syy, szz, sxy, syz, sxz}; peac_fit(m, N, center, normal, mse); }
    double similarity(const Seg& o) const { return std::fabs(normal[0] * o.normal[0] + normal[1] * o.normal[1] + normal[2] * o.normal[2]); }
    double dist(const double p[3]) const { return normal[0] * (p[0] - center[0]) + normal[1] * (p[1] - center[1]) + normal[2] * (p[2] - center[2]); }
};

struct Fitter {
    const PeacInput in; Params P; const int W, H; static constexpr int bw = 16, bh = 16, minSupport = 2000, maxStep = 100000;
    std::vector<Seg> pool; std::vector<int> dsParent, dsSize; std::vector<int> extracted;
    std::vector<std::vector<int>> nbs;       // per node: neighbour node indices, ascending (the iteration order of the candidate loop); kept beside the
                                             // statistics so that a candidate merge is plain data (no allocation per candidate)
    int add_node(const Seg& s) { pool.push_back(s); nbs.emplace_back(); return (int)pool.size() - 1; }
    const float invScale;
    explicit Fitter(const PeacInput& i) : in(i), W(i.w), H(i.h), invScale(1.0f / i.depthScale) {}
    // state handed from part1 to the grow and to part2
    int Nh = 0, Nw = 0, NB = 0; std::vector<int> planes; std::map<int, int> rid2pl; std::vector<int8_t> blk8; std::vector<int> blkMap; std::vector<char> valid;
    struct Seed { uint16_t x, y; int pl; }; std::vector<Seed> seeds; std::vector<PeacGrowPlane> pc; std::vector<uint32_t> seedWords;
    struct QCmp { const std::vector<Seg>* pool; bool operator()(int a, int b) const { return (*pool)[b].mse < (*pool)[a].mse; } };
    typedef std::priority_queue<int, std::vector<int>, QCmp> Queue;

    int find(int x) { if (dsParent[x] != x) dsParent[x] = find(dsParent[x]); return dsParent[x]; }
    void unite(int x, int y) { int a = find(x), b = find(y); if (a == b) return; if (dsSize[a] < dsSize[b]) { dsParent[a] = b; dsSize[b] += dsSize[a]; } else { dsParent[b] = a; dsSize[a] += dsSize[b]; } }
    static void set_insert(std::vector<int>& v, int x) { auto it = std::lower_bound(v.begin(), v.end(), x); if (it == v.end() || *it != x) v.insert(it, x); }
    static void set_erase(std::vector<int>& v, int x) { auto it = std::lower_bound(v.begin(), v.end(), x); if (it != v.end() && *it == x) v.erase(it); }
    void link(int a, int b) { set_insert(nbs[a], b); set_insert(nbs[b], a); }
    void unlink_all(int a) { for (int nb : nbs[a]) set_erase(nbs[nb], a); nbs[a].clear(); }
    // point of the organised cloud (float arithmetic of the reference's cloud construction), computed where the region grow needs it:
    // the grow looks at ~2/3 of the pixels once, a materialised cloud would cost a pass over all of them plus 3.7 MB of traffic
    bool point(int row, int col, double p[3]) const {
        const float d = (float)in.depth[(size_t)row * W + col];
        if (d < 1e-3f) return false;
        const float z = d * invScale;
        p[0] = (double)((col - in.cx) * z / in.fx); p[1] = (double)((row - in.cy) * z / in.fy); p[2] = (double)z;
        return true;
    }

    std::vector<int> ubuf;
    int cluster(Queue& q) {
        int step = 0;
        while (!q.empty() && step <= maxStep) {
            const int p = q.top(); q.pop();
            if (pool[p].nouse) continue;
            // the candidate with the smallest MSE (the first one among equals) becomes a node -- whether or not the merge then passes the threshold, as in the
            // reference; the candidates that lose never enter the pool (node indices only order the neighbour sets, and only relatively)
            int cand = -1, cand_nb = -1; bool have = false; Seg best;
            // the candidates' fits four at a time (peac_fit4.hpp: the scalar fit's operations in the lanes of a vector register, same bits)
            int cnb[4]; PeacFitIn fin[4]; PeacFitOut fout[4]; int nc = 0;
            auto flush = [&]() {
                peac_fit4(fin, nc, fout);
                for (int k = 0; k < nc; k++) {
                    if (have && !(best.mse > fout[k].mse)) continue;
                    const Seg &a = pool[p], &b = pool[cnb[k]]; const double* m = fin[k].m;
                    best.sx = m[0]; best.sy = m[1]; best.sz = m[2]; best.sxx = m[3]; best.syy = m[4]; best.szz = m[5]; best.sxy = m[6]; best.syz = m[7]; best.sxz = m[8];
                    best.N = fin[k].N; best.rid = a.N >= b.N ? a.rid : b.rid; best.mse = fout[k].mse; best.nouse = false;
                    for (int q = 0; q < 3; q++) { best.center[q] = fout[k].center[q]; best.normal[q] = fout[k].normal[q]; }
                    have = true; cand_nb = cnb[k];
                }
                nc = 0;
            };
            for (int nb : nbs[p]) {
                if (pool[p].similarity(pool[nb]) < P.simMerge) continue;
                const Seg &a = pool[p], &b = pool[nb]; double* m = fin[nc].m;
                m[0] = a.sx + b.sx; m[1] = a.sy + b.sy; m[2] = a.sz + b.sz; m[3] = a.sxx + b.sxx; m[4] = a.syy + b.syy; m[5] = a.szz + b.szz;
                m[6] = a.sxy + b.sxy; m[7] = a.syz + b.syz; m[8] = a.sxz + b.sxz; fin[nc].N = a.N + b.N; cnb[nc] = nb;
                if (++nc == 4) flush();
            }
            if (nc) flush();
            if (have) cand = add_node(best);
            if (cand >= 0 && pool[cand].mse < P.mse_merge(pool[cand].center[2])) {
                q.push(cand);
                unite(pool[p].rid, pool[cand_nb].rid);
                std::vector<int>& u = ubuf; u.clear();
                std::set_union(nbs[p].begin(), nbs[p].end(), nbs[cand_nb].begin(), nbs[cand_nb].end(), std::back_inserter(u));
                set_erase(u, p); set_erase(u, cand_nb);
                unlink_all(p); unlink_all(cand_nb);
                for (int nb : u) nbs[nb].push_back(cand);                            // (cand is the newest node: the largest index)
                nbs[cand] = u;
                pool[p].nouse = pool[cand_nb].nouse = true;
            } else {
                if (pool[p].N >= minSupport) extracted.push_back(p);
                unlink_all(p);
            }
            ++step;
        }
        while (!q.empty()) { const int p = q.top(); q.pop(); if (pool[p].N >= minSupport) extracted.push_back(p); unlink_all(p); }
        std::sort(extracted.begin(), extracted.end(), [&](int a, int b) { return pool[b].N < pool[a].N; });
        return step;
    }

    // ---- part 1: initial graph, agglomerative clustering, plane list, block erosion, the region grow's seeds and per-plane constants
    void part1() {
        Nh = H / bh; Nw = W / bw; NB = Nh * Nw;
        dsParent.resize(NB); dsSize.assign(NB, 1); for (int i = 0; i < NB; i++) dsParent[i] = i;
        pool.reserve(NB * 3); nbs.reserve(NB * 3);
        std::vector<int> G(NB, -1);
        Queue q(QCmp{&pool});
        for (int b = 0; b < NB; b++) {
            const PeacBlockStats& S = in.blocks[b];
            if (!S.valid || S.N < 4) continue;
            Seg s; s.sx = S.sx; s.sy = S.sy; s.sz = S.sz; s.sxx = S.sxx; s.syy = S.syy; s.szz = S.szz; s.sxy = S.sxy; s.syz = S.syz; s.sxz = S.sxz; s.N = S.N; s.rid = b;
            if (S.fitted) { s.mse = S.mse; for (int k = 0; k < 3; k++) { s.center[k] = S.center[k]; s.normal[k] = S.normal[k]; } }      // fitted on the GPU (k_peac_block_fit: the same function)
            else s.fit();
            if (!(s.mse < P.mse_init(s.center[2]))) continue;
            G[b] = add_node(s);
        }
        for (int b = 0; b < NB; b++) if (G[b] >= 0) q.push(G[b]);
        for (int i = 0; i < Nh; ++i) for (int j = 1; j < Nw; j += 2) {           // row-direction links
            const int c = i * Nw + j;
            if (G[c - 1] < 0) { --j; continue; }
            if (G[c] < 0) continue;
            if (j < Nw - 1 && G[c + 1] < 0) { ++j; continue; }
            const double th = P.ang_init(pool[G[c]].center[2]);
            if ((j < Nw - 1 && pool[G[c - 1]].similarity(pool[G[c + 1]]) >= th) || (j == Nw - 1 && pool[G[c]].similarity(pool[G[c - 1]]) >= th)) { link(G[c], G[c - 1]); if (j < Nw - 1) link(G[c], G[c + 1]); }
            else --j;
        }
        for (int j = 0; j < Nw; ++j) for (int i = 1; i < Nh; i += 2) {           // column-direction links
            const int c = i * Nw + j;
            if (G[c - Nw] < 0) { --i; continue; }
            if (G[c] < 0) continue;
            if (i < Nh - 1 && G[c + Nw] < 0) { ++i; continue; }
            const double th = P.ang_init(pool[G[c]].center[2]);
            if ((i < Nh - 1 && pool[G[c - Nw]].similarity(pool[G[c + Nw]]) >= th) || (i == Nh - 1 && pool[G[c]].similarity(pool[G[c - Nw]]) >= th)) { link(G[c], G[c - Nw]); if (i < Nh - 1) link(G[c], G[c + Nw]); }
            else --i;
        }
        cluster(q);
        // ---- refineDetails: block erosion, seeds
        planes = extracted; extracted.clear();
        for (int k = 0; k < (int)planes.size(); k++) rid2pl.insert({pool[planes[k]].rid, k});
        blkMap.assign(NB, -1); valid.assign(planes.size(), 0);
        seeds.reserve(32768);
        auto seed_at = [&](int idx, int pl) { const int y = idx / W; seeds.push_back({(uint16_t)(idx - y * W), (uint16_t)y, pl}); };
        auto nb4 = [](int i, int j, int Hh, int Ww, int out[4]) { const int id = i * Ww + j; int c = 0; if (j > 0) out[c++] = id - 1; if (j < Ww - 1) out[c++] = id + 1; if (i > 0) out[c++] = id - Ww; if (i < Hh - 1) out[c++] = id + Ww; return c; };
        for (int i = 0, b = 0; i < Nh; ++i) for (int j = 0; j < Nw; ++j, ++b) {
            const int set = find(b), sz = dsSize[set] * bw * bh;
            if (sz >= minSupport) {
                int nb[4]; const int nn = nb4(i, j, Nh, Nw, nb); bool same = true;
                for (int k = 0; k < nn; k++) if (find(nb[k]) != set) { same = false; break; }
                const int pl = rid2pl[set];
                if (same) { blkMap[b] = pl; valid[pl] = 1; }
            }
            if (blkMap[b] < 0) {
                if (i > 0 && blkMap[b - Nw] >= 0) { const int s = (i * bh - 1) * W + j * bw; for (int k = 1; k < bw; ++k) seed_at(s + k, blkMap[b - Nw]); }
                if (j > 0 && blkMap[b - 1] >= 0) { const int s = (i * bh) * W + j * bw - 1; for (int k = 0; k < bh - 1; ++k) seed_at(s + k * W, blkMap[b - 1]); }
            } else {
                const int pl = blkMap[b];
                if (i > 0 && blkMap[b - Nw] != pl) { const int s = (i * bh) * W + j * bw; for (int k = 0; k < bw - 1; ++k) seed_at(s + k, pl); }
                if (j > 0 && blkMap[b - 1] != pl) { const int s = (i * bh) * W + j * bw; for (int k = 1; k < bh; ++k) seed_at(s + k * W, pl); }
            }
        }
        // per plane: the distance test's constants (normals do not change during the grow)
        const int nPl = (int)planes.size();
        pc.resize(nPl);
        for (int k = 0; k < nPl; k++) { const Seg& sg = pool[planes[k]]; for (int q = 0; q < 3; q++) { pc[k].n[q] = sg.normal[q]; pc[k].c[q] = sg.center[q]; } pc[k].thr = 9 * sg.mse + 1e-5; pc[k].pad = 0; }
    }
    // what the GPU grow takes: the block map as bytes and the initial seeds as frontier words (peac_kernels.hip: count - 1 << 30 | index << 28 | plane << 20 | pixel)
    bool pack_for_gpu() {
        const int nPl = (int)planes.size();
        if (nPl > PEAC_GROW_MAX_PLANES || (int)seeds.size() > PEAC_GROW_MAX_SEEDS0 || (size_t)W * H > (1u << 20)) return false;
        blk8.resize(NB); for (int b = 0; b < NB; b++) blk8[b] = (int8_t)blkMap[b];
        seedWords.resize(seeds.size());
        // seeds that share a pixel (rare: borders of two eroded planes meeting): per pixel, low nibble = seeds on it, high nibble = seeds seen so far.
        // (Two std::maps here cost 1.6 ms a frame -- more than the graph clustering.)
        static thread_local std::vector<uint8_t> cnt;
        if (cnt.size() < (size_t)W * H) cnt.assign((size_t)W * H, 0);
        for (const Seed& sd : seeds) { uint8_t& c = cnt[(size_t)sd.y * W + sd.x]; if ((c & 15) < 15) c++; }
        bool fits = true;
        for (size_t k = 0; k < seeds.size() && fits; k++) {
            const int c = seeds[k].y * W + seeds[k].x, n = cnt[c] & 15, idx = cnt[c] >> 4;
            if (n > PEAC_GROW_SLOTS) { fits = false; break; }
            cnt[c] += 16;
            seedWords[k] = ((uint32_t)(n - 1) << 30) | ((uint32_t)idx << 28) | ((uint32_t)seeds[k].pl << 20) | (uint32_t)c;
        }
        for (const Seed& sd : seeds) cnt[(size_t)sd.y * W + sd.x] = 0;              // leave the scratch map clean for the thread's next frame
        if (!fits) return false;
        return true;
    }
    // ---- the region grow along the plane borders (FIFO, order-defined), host statement.  member: plane index per pixel or -1 (nobody's);
    // pairSeen[a * nPl + b] = 1: a seed of plane a passed the distance test on a pixel that plane b held
    void grow_host(std::vector<int8_t>& member8, std::vector<int16_t>& member16, std::vector<uint8_t>& pairSeen) {
        const int nPl = (int)planes.size();
        // per-pixel plane index, or -1 - (number of failed tries), stops at -6; 16 bits keep the grow's working set small (at most W*H/minSupport planes)
        std::vector<int16_t> member((size_t)W * H, -1);
        for (int i = 0, b = 0; i < Nh; ++i) for (int j = 0; j < Nw; ++j, ++b) if (blkMap[b] >= 0) for (int y = i * bh; y < (i + 1) * bh; y++) std::fill_n(&member[(size_t)y * W + j * bw], bw, (int16_t)blkMap[b]);
        pairSeen.assign((size_t)nPl * nPl, 0);
        std::vector<float> distMap((size_t)W * H, std::numeric_limits<float>::max());
        auto visit = [&](int cx, int cy, int pl, const PeacGrowPlane& S) {
            const int c = cy * W + cx; int16_t& trail = member[c];
            if (trail <= -6) return;
            if (trail == pl) return;
            const int by = cy / bh, bx = cx / bw;
            if (by < Nh && bx < Nw && blkMap[by * Nw + bx] >= 0) return;
            double pt[3]; float cdist = -1;
            const bool has_pt = point(cy, cx, pt);
            if (has_pt) cdist = (float)std::fabs(S.n[0] * (pt[0] - S.c[0]) + S.n[1] * (pt[1] - S.c[1]) + S.n[2] * (pt[2] - S.c[2]));
            if (has_pt && (double)cdist * (double)cdist < S.thr) {
                if (trail >= 0) pairSeen[(size_t)pl * nPl + trail] = 1;
                float& od = distMap[c];
                if (cdist < od) { trail = (int16_t)pl; od = cdist; seeds.push_back({(uint16_t)cx, (uint16_t)cy, pl}); }
                else if (trail < 0) trail -= 1;
            } else if (trail < 0) trail -= 1;
        };
        for (size_t k = 0; k < seeds.size(); ++k) {
            if (k + 12 < seeds.size()) {      // the FIFO jumps between all growing borders: fetch the rows a later seed will look at
                const Seed f = seeds[k + 12]; const size_t c = (size_t)f.y * W + f.x;
                __builtin_prefetch(&member[c]); if (f.y > 0) __builtin_prefetch(&member[c - W]); if (f.y < H - 1) __builtin_prefetch(&member[c + W]);
                __builtin_prefetch(&in.depth[c]); __builtin_prefetch(&distMap[c]);
            }
            const Seed sd = seeds[k]; const int sx = sd.x, sy = sd.y; const PeacGrowPlane& S = pc[sd.pl];
            if (sx > 0) visit(sx - 1, sy, sd.pl, S);
            if (sx < W - 1) visit(sx + 1, sy, sd.pl, S);
            if (sy > 0) visit(sx, sy - 1, sd.pl, S);
            if (sy < H - 1) visit(sx, sy + 1, sd.pl, S);
        }
        if (nPl <= 127) { member8.resize(member.size()); for (size_t k = 0; k < member.size(); k++) member8[k] = (int8_t)std::max<int>(member[k], -1); member16.clear(); }
        else { member16.swap(member); member8.clear(); }
    }
    // ---- part 2: refine links between the planes that met during the grow, last merge, plane masks -> closed -> external contours, thickness 2
    template <class T> void part2(const T* member, const uint8_t* pairSeen, BitImg& planeContours) {
        const int nPl = (int)planes.size();
        for (int a = 0; a < nPl; a++) for (int b = a + 1; b < nPl; b++)          // the first meeting of two planes decides their link (order-free: a set of links)
            if ((pairSeen[(size_t)a * nPl + b] | pairSeen[(size_t)b * nPl + a]) && pool[planes[a]].similarity(pool[planes[b]]) >= P.simRefine) link(planes[a], planes[b]);
        Queue q2(QCmp{&pool});
        for (int k = 0; k < (int)planes.size(); k++) if (valid[k]) q2.push(planes[k]);
        cluster(q2);
        std::vector<int> plmap(planes.size(), -1); int nFinal = 0;
        for (int k = 0; k < (int)planes.size(); k++) {
            if (!valid[k]) continue;
            const int rid = pool[planes[k]].rid, root = find(rid);
            if (root == rid) { if (plmap[k] < 0) plmap[k] = nFinal++; }
            else { const int np = rid2pl[root]; if (plmap[np] < 0) plmap[k] = plmap[np] = nFinal++; else plmap[k] = plmap[np]; }
        }
        const int nOut = (int)extracted.size();
        std::vector<BitImg> masks(nOut); for (auto& m : masks) m.create(W, H);
        std::vector<int> yLo(nOut, H), yHi(nOut, -1);
        for (int y = 0; y < H; y++) {                                  // runs of equal membership -> bit ranges
            const T* m = &member[(size_t)y * W];
            for (int x = 0; x < W;) {
                const int pl = m[x], x0 = x;
                if (sizeof(T) == 1) {                                   // eight pixels a step: first byte that differs from the run's value
                    uint64_t bv; std::memset(&bv, (uint8_t)pl, 8);
                    for (;;) {
                        if (x + 8 > W) { while (x < W && m[x] == pl) x++; break; }
                        uint64_t w8; std::memcpy(&w8, &m[x], 8); w8 ^= bv;
                        if (w8) { x += __builtin_ctzll(w8) >> 3; break; }
                        x += 8;
                    }
                } else while (x < W && m[x] == pl) x++;
                if (pl < 0 || plmap[pl] < 0 || plmap[pl] >= nOut) continue;
                const int o = plmap[pl]; uint64_t* r = masks[o].row(y); const int a = x0, b = x - 1, ka = a >> 6, kb = b >> 6;
                for (int q = ka; q <= kb; q++) { uint64_t bits = ~0ull; if (q == ka) bits &= ~0ull << (a & 63); if (q == kb) bits &= (b & 63) == 63 ? ~0ull : ((1ull << ((b & 63) + 1)) - 1); r[q] |= bits; }
                yLo[o] = std::min(yLo[o], y); yHi[o] = y;
            }
        }
        const EllipseElem e3(3);
        for (int o = 0; o < nOut; o++) {
            if (yHi[o] < 0) continue;
            const BitImg c = masks[o].dilated(e3, yLo[o], yHi[o]).eroded_rows(e3, yLo[o] - 1, yHi[o] + 1);     // closing, evaluated on the mask's rows only
            std::vector<Contour> cs; find_contours(c, cs, true);
            for (const Contour& k : cs) draw_thick2(planeContours, k);
        }
    }
};
}  // namespace

PeacFitter::PeacFitter(const PeacInput& in) : f(new Fitter(in)) {}
PeacFitter::~PeacFitter() { delete static_cast<Fitter*>(f); }
#define FIT static_cast<Fitter*>(f)
void PeacFitter::part1() { FIT->part1(); gpu_ok = FIT->pack_for_gpu(); }
int PeacFitter::n_planes() const { return (int)FIT->planes.size(); }
int PeacFitter::n_blocks() const { return FIT->NB; }
const PeacGrowPlane* PeacFitter::planes() const { return FIT->pc.data(); }
const int8_t* PeacFitter::block_map() const { return FIT->blk8.data(); }
const std::vector<uint32_t>& PeacFitter::seed_words() const { return FIT->seedWords; }
void PeacFitter::grow_host(std::vector<int8_t>& m8, std::vector<int16_t>& m16, std::vector<uint8_t>& pairSeen) { FIT->grow_host(m8, m16, pairSeen); }
void PeacFitter::part2(const int8_t* m8, const int16_t* m16, const uint8_t* pairSeen, BitImg& planeContours) {
    planeContours.create(FIT->W, FIT->H);
    if (m8) FIT->part2(m8, pairSeen, planeContours); else FIT->part2(m16, pairSeen, planeContours);
}
#undef FIT

void peac_plane_contours(const PeacInput& in, BitImg& planeContours) {
    PeacFitter f(in); f.part1();
    std::vector<int8_t> m8; std::vector<int16_t> m16; std::vector<uint8_t> seen;
    f.grow_host(m8, m16, seen);
    f.part2(m8.empty() ? nullptr : m8.data(), m16.empty() ? nullptr : m16.data(), seen.data(), planeContours);
}

}  // namespace sind
