// Host stage: agglomerative plane clustering + refinement of the SInDSLAM-modified PEAC fitter
// (reference include/PEAC/AHCPlaneFitter.hpp:186-236 run, :881-1039 initGraph, :1050-1256 ahCluster,
//  :274-400 refineDetails, :603-705 findBlockMembership, :546-594 floodFill; AHCPlaneSeg.hpp:103-134 PCA;
//  AHCParamSet.hpp:48-148 thresholds; DisjointSet.hpp).
// Split of work: the 1200 per-window second-order statistics come from the GPU (depth_kernels.hip
// k_peac_block_stats); the graph (<= 1200 nodes, priority queue, disjoint sets) and the pixel-level region grow along
// plane borders are serial and stay here.  Nodes live in an index pool; neighbour sets are ordered by node index
// (= creation order; the reference orders by heap address, which only matters for exactly tied MSEs).
// Eigen's 3x3 self-adjoint solver is replaced by a cyclic Jacobi iteration (Eigen is not available).
#include <cmath>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include "host.hpp"

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

void jacobi3(const double K[3][3], double s[3], double V[3][3]) {
    double A[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A[i][j] = K[i][j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (A[p][q] == 0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
            const double c = 1 / std::sqrt(t * t + 1), sn = t * c;
            for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - sn * b; A[k][q] = sn * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - sn * b; A[q][k] = sn * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
        }
    }
    int o[3] = {0, 1, 2}; const double e[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (e[o[i]] > e[o[j]]) std::swap(o[i], o[j]);
    double T[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i][j] = V[i][o[j]];
    for (int i = 0; i < 3; i++) { s[i] = e[o[i]]; for (int j = 0; j < 3; j++) V[i][j] = T[i][j]; }
}

struct Seg {
    double sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, syz = 0, sxz = 0; int N = 0;
    int rid = 0; double mse = 0, center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}; bool nouse = false;
    std::set<int> nbs;
    void fit() {
        const double sc = 1.0 / N;
        center[0] = sx * sc; center[1] = sy * sc; center[2] = sz * sc;
        double K[3][3] = {{sxx - sx * sx * sc, sxy - sx * sy * sc, sxz - sx * sz * sc}, {0, syy - sy * sy * sc, syz - sy * sz * sc}, {0, 0, szz - sz * sz * sc}};
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double sv[3], V[3][3]; jacobi3(K, sv, V);
        const double sgn = (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) ? 1.0 : -1.0;
        normal[0] = sgn * V[0][0]; normal[1] = sgn * V[1][0]; normal[2] = sgn * V[2][0];
        mse = sv[0] * sc;
    }
    double similarity(const Seg& o) const { return std::fabs(normal[0] * o.normal[0] + normal[1] * o.normal[1] + normal[2] * o.normal[2]); }
    double dist(const double p[3]) const { return normal[0] * (p[0] - center[0]) + normal[1] * (p[1] - center[1]) + normal[2] * (p[2] - center[2]); }
};

struct Fitter {
    const PeacInput& in; Params P; const int W, H, bw = 16, bh = 16, minSupport = 2000, maxStep = 100000;
    std::vector<Seg> pool; std::vector<int> dsParent, dsSize; std::vector<int> extracted;
    explicit Fitter(const PeacInput& i) : in(i), W(i.w), H(i.h) {}
    struct QCmp { const std::vector<Seg>* pool; bool operator()(int a, int b) const { return (*pool)[b].mse < (*pool)[a].mse; } };
    typedef std::priority_queue<int, std::vector<int>, QCmp> Queue;

    int find(int x) { if (dsParent[x] != x) dsParent[x] = find(dsParent[x]); return dsParent[x]; }
    void unite(int x, int y) { int a = find(x), b = find(y); if (a == b) return; if (dsSize[a] < dsSize[b]) { dsParent[a] = b; dsSize[b] += dsSize[a]; } else { dsParent[b] = a; dsSize[a] += dsSize[b]; } }
    void link(int a, int b) { pool[a].nbs.insert(b); pool[b].nbs.insert(a); }
    void unlink_all(int a) { for (int nb : pool[a].nbs) pool[nb].nbs.erase(a); pool[a].nbs.clear(); }
    std::vector<float> cloud;     // organised cloud (x, y, z) of the frame, z = NaN-equivalent flag -1 for missing depth
    void build_cloud() {
        cloud.resize((size_t)W * H * 3);
        const float inv = 1.0f / in.depthScale;
        for (int row = 0; row < H; row++) { const uint16_t* dr = in.depth + (size_t)row * W; float* c = &cloud[(size_t)row * W * 3];
            for (int col = 0; col < W; col++) { const float d = (float)dr[col];
                if (d < 1e-3f) { c[3 * col + 2] = -1.f; continue; }
                const float z = d * inv; c[3 * col] = (col - in.cx) * z / in.fx; c[3 * col + 1] = (row - in.cy) * z / in.fy; c[3 * col + 2] = z; } }
    }
    bool point(int row, int col, double p[3]) const {
        const float* c = &cloud[((size_t)row * W + col) * 3];
        if (c[2] < 0.f) return false;
        p[0] = (double)c[0]; p[1] = (double)c[1]; p[2] = (double)c[2];
        return true;
    }

    int cluster(Queue& q) {
        int step = 0;
        while (!q.empty() && step <= maxStep) {
            const int p = q.top(); q.pop();
            if (pool[p].nouse) continue;
            int cand = -1, cand_nb = -1;
            const std::vector<int> nbv(pool[p].nbs.begin(), pool[p].nbs.end());   // pool may reallocate while candidates are appended
            for (int nb : nbv) {
                if (pool[p].similarity(pool[nb]) < P.simMerge) continue;
                Seg m; const Seg &a = pool[p], &b = pool[nb];
                m.sx = a.sx + b.sx; m.sy = a.sy + b.sy; m.sz = a.sz + b.sz; m.sxx = a.sxx + b.sxx; m.syy = a.syy + b.syy; m.szz = a.szz + b.szz;
                m.sxy = a.sxy + b.sxy; m.syz = a.syz + b.syz; m.sxz = a.sxz + b.sxz; m.N = a.N + b.N; m.rid = a.N >= b.N ? a.rid : b.rid; m.fit();
                if (cand < 0 || pool[cand].mse > m.mse) { pool.push_back(m); cand = (int)pool.size() - 1; cand_nb = nb; }
            }
            if (cand >= 0 && pool[cand].mse < P.mse_merge(pool[cand].center[2])) {
                q.push(cand);
                unite(pool[p].rid, pool[cand_nb].rid);
                std::set<int> u = pool[p].nbs; u.insert(pool[cand_nb].nbs.begin(), pool[cand_nb].nbs.end()); u.erase(p); u.erase(cand_nb);
                unlink_all(p); unlink_all(cand_nb);
                pool[cand].nbs = u; for (int nb : u) pool[nb].nbs.insert(cand);
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

    void run(BitImg& planeContours) {
        const int Nh = H / bh, Nw = W / bw, NB = Nh * Nw;
        dsParent.resize(NB); dsSize.assign(NB, 1); for (int i = 0; i < NB; i++) dsParent[i] = i;
        pool.reserve(NB * 16);
        std::vector<int> G(NB, -1);
        Queue q(QCmp{&pool});
        for (int b = 0; b < NB; b++) {
            const PeacBlockStats& S = in.blocks[b];
            if (!S.valid || S.N < 4) continue;
            Seg s; s.sx = S.sx; s.sy = S.sy; s.sz = S.sz; s.sxx = S.sxx; s.syy = S.syy; s.szz = S.szz; s.sxy = S.sxy; s.syz = S.syz; s.sxz = S.sxz; s.N = S.N; s.rid = b; s.fit();
            if (!(s.mse < P.mse_init(s.center[2]))) continue;
            pool.push_back(s); G[b] = (int)pool.size() - 1;
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
        build_cloud();
        // ---- refineDetails: block erosion, seeds, region grow, last merge
        std::vector<int> planes = extracted; extracted.clear();
        std::map<int, int> rid2pl; for (int k = 0; k < (int)planes.size(); k++) rid2pl.insert({pool[planes[k]].rid, k});
        std::vector<int> member((size_t)W * H, -1), blkMap(NB, -1); std::vector<char> valid(planes.size(), 0);
        std::vector<std::pair<int, int>> seeds;
        auto nb4 = [](int i, int j, int Hh, int Ww, int out[4]) { const int id = i * Ww + j; int c = 0; if (j > 0) out[c++] = id - 1; if (j < Ww - 1) out[c++] = id + 1; if (i > 0) out[c++] = id - Ww; if (i < Hh - 1) out[c++] = id + Ww; return c; };
        for (int i = 0, b = 0; i < Nh; ++i) for (int j = 0; j < Nw; ++j, ++b) {
            const int set = find(b), sz = dsSize[set] * bw * bh;
            if (sz >= minSupport) {
                int nb[4]; const int nn = nb4(i, j, Nh, Nw, nb); bool same = true;
                for (int k = 0; k < nn; k++) if (find(nb[k]) != set) { same = false; break; }
                const int pl = rid2pl[set];
                if (same) { blkMap[b] = pl; valid[pl] = 1; for (int y = i * bh; y < (i + 1) * bh; y++) for (int x = j * bw; x < (j + 1) * bw; x++) member[(size_t)y * W + x] = pl; }
            }
            if (blkMap[b] < 0) {
                if (i > 0 && blkMap[b - Nw] >= 0) { const int s = (i * bh - 1) * W + j * bw; for (int k = 1; k < bw; ++k) seeds.push_back({s + k, blkMap[b - Nw]}); }
                if (j > 0 && blkMap[b - 1] >= 0) { const int s = (i * bh) * W + j * bw - 1; for (int k = 0; k < bh - 1; ++k) seeds.push_back({s + k * W, blkMap[b - 1]}); }
            } else {
                const int pl = blkMap[b];
                if (i > 0 && blkMap[b - Nw] != pl) { const int s = (i * bh) * W + j * bw; for (int k = 0; k < bw - 1; ++k) seeds.push_back({s + k, pl}); }
                if (j > 0 && blkMap[b - 1] != pl) { const int s = (i * bh) * W + j * bw; for (int k = 1; k < bh; ++k) seeds.push_back({s + k * W, pl}); }
            }
        }
        std::vector<float> distMap((size_t)W * H, std::numeric_limits<float>::max());
        for (size_t k = 0; k < seeds.size(); ++k) {
            const int sIdx = seeds[k].first, sy = sIdx / W, sx = sIdx - sy * W, pl = seeds[k].second;
            const Seg& S = pool[planes[pl]];
            int nb[4]; const int nn = nb4(sy, sx, H, W, nb);
            for (int t = 0; t < nn; ++t) {
                const int c = nb[t]; int& trail = member[c];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == pl) continue;
                const int cy = c / W, cx = c - cy * W, by = cy / bh, bx = cx / bw;
                const int blk = (by < Nh && bx < Nw) ? by * Nw + bx : -1;
                if (blk >= 0 && blkMap[blk] >= 0) continue;
                double pt[3]; float cdist = -1;
                const bool has_pt = point(cy, cx, pt);
                if (has_pt) cdist = (float)std::fabs(S.dist(pt));
                if (has_pt && (double)cdist * (double)cdist < 9 * S.mse + 1e-5) {
                    if (trail >= 0) { Seg& O = pool[planes[trail]]; if (S.similarity(O) >= P.simRefine) link(planes[trail], planes[pl]); }
                    float& od = distMap[c];
                    if (cdist < od) { trail = pl; od = cdist; seeds.push_back({c, pl}); }
                    else if (trail < 0) trail -= 1;
                } else if (trail < 0) trail -= 1;
            }
        }
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
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const int pl = member[(size_t)y * W + x]; if (pl >= 0 && plmap[pl] >= 0 && plmap[pl] < nOut) masks[plmap[pl]].set(x, y); }
        const EllipseElem e3(3);
        for (const BitImg& m : masks) {
            const BitImg c = m.closed(e3);
            std::vector<Contour> cs; find_contours(c, cs, true);
            for (const Contour& k : cs) draw_thick2(planeContours, k);
        }
    }
};
}  // namespace

void peac_plane_contours(const PeacInput& in, BitImg& planeContours) {
    planeContours.create(in.w, in.h);
    Fitter f(in); f.run(planeContours);
}

}  // namespace sind
