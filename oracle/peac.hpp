// ORACLE — test infrastructure only (see cvx_core.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the vendored, SInDSLAM-modified PEAC plane fitter that DynaDetect::CalOccluded
// runs per frame (reference DynaDetect.cc:558-593 -> include/PEAC/plane_fitter_pcl.hpp:160-317 ->
// AHCPlaneFitter.hpp:186-236 run, :881-1039 initGraph, :1050-1256 ahCluster, :274-400 refineDetails,
// :603-705 findBlockMembership, :546-594 floodFill; AHCPlaneSeg.hpp:38-135 Stats, :180-330 PlaneSeg;
// AHCParamSet.hpp:48-148; DisjointSet.hpp).  Output: the plane-contour image (thickness-2 external
// contours of every extracted plane after a 3x3 CLOSE), which is all DynaDetect consumes.
//
// Deviations, both forced: (1) Eigen's SelfAdjointEigenSolver<Matrix3d> (eig33sym.hpp:46-52) is replaced by a
// cyclic Jacobi solver (Eigen is not vendored); (2) std::set<PlaneSeg*> neighbour sets are ordered by
// creation sequence instead of heap address (the reference's iteration order is address dependent; it only
// matters when two candidate merges have exactly equal MSE).
#pragma once
#include <map>
#include <memory>
#include <queue>
#include <set>
#include "cvx_shape.hpp"

namespace peac {
using namespace cvx;

struct ParamSet {   // AHCParamSet.hpp:48-56 (millimetre-designed defaults, fed with metres: SURVEY App. C-1)
    double depthSigma = 3e-6, stdTol_init = 10, stdTol_merge = 17;
    double z_near = 500, z_far = 6000, angle_near = 10.0 * M_PI / 180.0, angle_far = 20.0 * M_PI / 180.0;
    double similarityTh_merge = std::cos(15.0 * M_PI / 180.0), similarityTh_refine = std::cos(20.0 * M_PI / 180.0);
    double depthAlpha = 0.04, depthChangeTol = 0.02 * 1000;
    enum Phase { P_INIT = 0, P_MERGING = 1, P_REFINE = 2 };
    double T_mse(Phase ph, double z = 0) const { return ph == P_INIT ? std::pow(depthSigma * z * z + stdTol_init, 2) : std::pow(depthSigma * z * z + stdTol_merge, 2); }
    double T_ang(Phase ph, double z = 0) const {
        if (ph == P_INIT) { double cz = std::min(std::max(z, z_near), z_far); const double f = (angle_far - angle_near) / (z_far - z_near); return std::cos(f * cz + angle_near - f * z_near); }
        return ph == P_MERGING ? similarityTh_merge : similarityTh_refine;
    }
    double T_dz(double z) const { return depthAlpha * std::fabs(z) + depthChangeTol; }
};

// symmetric 3x3 eigen-decomposition, eigenvalues ascending, V columns = eigenvectors
inline void eig33sym(const double K[3][3], double s[3], double V[3][3]) {
    double A[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { A[i][j] = K[i][j]; V[i][j] = i == j; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag || off == 0) break;
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (A[p][q] == 0) continue;
            double theta = (A[q][q] - A[p][p]) / (2 * A[p][q]);
            double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
            double c = 1 / std::sqrt(t * t + 1), sn = t * c;
            for (int k = 0; k < 3; k++) { double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; k++) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; k++) { double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    int o[3] = {0, 1, 2}; double e[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (e[o[i]] > e[o[j]]) std::swap(o[i], o[j]);
    double Vt[3][3]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Vt[i][j] = V[i][o[j]];
    for (int i = 0; i < 3; i++) { s[i] = e[o[i]]; for (int j = 0; j < 3; j++) V[i][j] = Vt[i][j]; }
}

struct Stats {
    double sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, syz = 0, sxz = 0; int N = 0;
    void push(double x, double y, double z) { sx += x; sy += y; sz += z; sxx += x * x; syy += y * y; szz += z * z; sxy += x * y; syz += y * z; sxz += x * z; ++N; }
    static Stats sum(const Stats& a, const Stats& b) { Stats r; r.sx = a.sx + b.sx; r.sy = a.sy + b.sy; r.sz = a.sz + b.sz; r.sxx = a.sxx + b.sxx; r.syy = a.syy + b.syy; r.szz = a.szz + b.szz; r.sxy = a.sxy + b.sxy; r.syz = a.syz + b.syz; r.sxz = a.sxz + b.sxz; r.N = a.N + b.N; return r; }
    void compute(double center[3], double normal[3], double& mse, double& curvature) const {   // AHCPlaneSeg.hpp:103-134
        const double sc = 1.0 / N;
        center[0] = sx * sc; center[1] = sy * sc; center[2] = sz * sc;
        double K[3][3] = {{sxx - sx * sx * sc, sxy - sx * sy * sc, sxz - sx * sz * sc}, {0, syy - sy * sy * sc, syz - sy * sz * sc}, {0, 0, szz - sz * sz * sc}};
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double sv[3], V[3][3]; eig33sym(K, sv, V);
        double sgn = (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) ? 1.0 : -1.0;
        normal[0] = sgn * V[0][0]; normal[1] = sgn * V[1][0]; normal[2] = sgn * V[2][0];
        mse = sv[0] * sc; curvature = sv[0] / (sv[0] + sv[1] + sv[2]);
    }
};

struct PlaneSeg;
struct SeqLess { bool operator()(const PlaneSeg* a, const PlaneSeg* b) const; };
struct PlaneSeg {
    Stats stats; int rid = 0; double mse = 0, center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}, curvature = 0; int N = 0; bool nouse = false;
    long seq = 0;
    std::set<PlaneSeg*, SeqLess> nbs;
    double normalSimilarity(const PlaneSeg& p) const { return std::fabs(normal[0] * p.normal[0] + normal[1] * p.normal[1] + normal[2] * p.normal[2]); }
    double signedDist(const double pt[3]) const { return normal[0] * (pt[0] - center[0]) + normal[1] * (pt[1] - center[1]) + normal[2] * (pt[2] - center[2]); }
    void connect(PlaneSeg* p) { if (p) { nbs.insert(p); p->nbs.insert(this); } }
    void disconnectAllNbs() { for (PlaneSeg* nb : nbs) nb->nbs.erase(this); nbs.clear(); }
};
inline bool SeqLess::operator()(const PlaneSeg* a, const PlaneSeg* b) const { return a->seq < b->seq; }

struct DisjointSet {
    std::vector<int> parent, size;
    explicit DisjointSet(int n) : parent(n), size(n, 1) { for (int i = 0; i < n; i++) parent[i] = i; }
    int Find(int x) { if (parent[x] != x) parent[x] = Find(parent[x]); return parent[x]; }
    int getSetSize(int x) { return size[Find(x)]; }
    int Union(int x, int y) { int xr = Find(x), yr = Find(y); if (xr == yr) return xr;
        if (size[xr] < size[yr]) { parent[xr] = yr; size[yr] += size[xr]; return yr; } parent[yr] = xr; size[xr] += size[yr]; return xr; }
};

// organised cloud: xyz float triplets, NaN z = invalid (plane_fitter_pcl.hpp:21-39)
struct Cloud { int w, h; const float* xyz;
    bool get(int row, int col, double& x, double& y, double& z) const { const float* p = xyz + ((size_t)row * w + col) * 3; x = p[0]; y = p[1]; z = p[2]; return !std::isnan(z); } };

struct PlaneFitter {
    typedef std::shared_ptr<PlaneSeg> SP;
    struct MinMSE { bool operator()(const SP& a, const SP& b) const { return b->mse < a->mse; } };
    typedef std::priority_queue<SP, std::vector<SP>, MinMSE> Queue;
    int width = 0, height = 0, maxStep = 100000, minSupport = 2000, windowWidth = 16, windowHeight = 16;
    ParamSet params; const Cloud* points = nullptr;
    std::unique_ptr<DisjointSet> ds; std::vector<SP> extractedPlanes; ImgI membershipImg;
    std::map<int, int> rid2plid; std::vector<int> blkMap; std::vector<std::pair<int, int>> rfQueue;
    long seq = 0;

    SP make_block(int rid, int seed_row, int seed_col) {     // AHCPlaneSeg.hpp:180-262 (INIT_STRICT)
        SP p(new PlaneSeg()); p->seq = seq++; p->rid = rid;
        bool valid = true;
        for (int i = seed_row, ic = 0; ic < windowHeight && i < height && valid; ++i, ++ic)
            for (int j = seed_col, jc = 0; jc < windowWidth && j < width; ++j, ++jc) {
                double x = 0, y = 0, z = 10000, xn, yn, zn;
                if (!points->get(i, j, x, y, z)) { valid = false; break; }
                if (j + 1 < width && points->get(i, j + 1, xn, yn, zn) && std::fabs(z - zn) > params.T_dz(z)) { valid = false; break; }
                if (i + 1 < height && points->get(i + 1, j, xn, yn, zn) && std::fabs(z - zn) > params.T_dz(z)) { valid = false; break; }
                p->stats.push(x, y, z);
            }
        if (valid) { p->nouse = false; p->N = p->stats.N; } else { p->N = 0; p->stats = Stats(); p->nouse = true; }
        if (p->N < 4) p->mse = p->curvature = std::nan(""); else p->stats.compute(p->center, p->normal, p->mse, p->curvature);
        return p;
    }
    SP make_merged(const PlaneSeg& a, const PlaneSeg& b) {   // AHCPlaneSeg.hpp:270-290
        SP p(new PlaneSeg()); p->seq = seq++; p->stats = Stats::sum(a.stats, b.stats); p->nouse = false;
        p->rid = a.N >= b.N ? a.rid : b.rid; p->N = p->stats.N; p->stats.compute(p->center, p->normal, p->mse, p->curvature);
        return p;
    }
    void mergeNbsFrom(PlaneSeg& self, PlaneSeg& pa, PlaneSeg& pb) {   // AHCPlaneSeg.hpp:352-377
        ds->Union(pa.rid, pb.rid);
        self.nbs.insert(pa.nbs.begin(), pa.nbs.end()); self.nbs.insert(pb.nbs.begin(), pb.nbs.end());
        self.nbs.erase(&pa); self.nbs.erase(&pb);
        pa.disconnectAllNbs(); pb.disconnectAllNbs();
        for (PlaneSeg* nb : self.nbs) nb->nbs.insert(&self);
        pa.nouse = pb.nouse = true;
    }

    void initGraph(Queue& minQ, std::vector<SP>& keep) {       // AHCPlaneFitter.hpp:881-1039
        const int Nh = height / windowHeight, Nw = width / windowWidth;
        std::vector<PlaneSeg*> G((size_t)Nh * Nw, nullptr);
        for (int i = 0; i < Nh; ++i) for (int j = 0; j < Nw; ++j) {
            SP p = make_block(i * Nw + j, i * windowHeight, j * windowWidth);
            if (p->mse < params.T_mse(ParamSet::P_INIT, p->center[2]) && !p->nouse) { G[i * Nw + j] = p.get(); minQ.push(p); keep.push_back(p); }
        }
        for (int i = 0; i < Nh; ++i) for (int j = 1; j < Nw; j += 2) {
            const int c = i * Nw + j;
            if (G[c - 1] == 0) { --j; continue; }
            if (G[c] == 0) continue;
            if (j < Nw - 1 && G[c + 1] == 0) { ++j; continue; }
            const double th = params.T_ang(ParamSet::P_INIT, G[c]->center[2]);
            if ((j < Nw - 1 && G[c - 1]->normalSimilarity(*G[c + 1]) >= th) || (j == Nw - 1 && G[c]->normalSimilarity(*G[c - 1]) >= th)) {
                G[c]->connect(G[c - 1]); if (j < Nw - 1) G[c]->connect(G[c + 1]);
            } else --j;
        }
        for (int j = 0; j < Nw; ++j) for (int i = 1; i < Nh; i += 2) {
            const int c = i * Nw + j;
            if (G[c - Nw] == 0) { --i; continue; }
            if (G[c] == 0) continue;
            if (i < Nh - 1 && G[c + Nw] == 0) { ++i; continue; }
            const double th = params.T_ang(ParamSet::P_INIT, G[c]->center[2]);
            if ((i < Nh - 1 && G[c - Nw]->normalSimilarity(*G[c + Nw]) >= th) || (i == Nh - 1 && G[c]->normalSimilarity(*G[c - Nw]) >= th)) {
                G[c]->connect(G[c - Nw]); if (i < Nh - 1) G[c]->connect(G[c + Nw]);
            } else --i;
        }
    }

    int ahCluster(Queue& minQ) {                                 // AHCPlaneFitter.hpp:1050-1256
        int step = 0;
        while (!minQ.empty() && step <= maxStep) {
            SP p = minQ.top(); minQ.pop();
            if (p->nouse) continue;
            SP cand_merge; PlaneSeg* cand_nb = nullptr;
            for (PlaneSeg* nb : p->nbs) {
                if (p->normalSimilarity(*nb) < params.T_ang(ParamSet::P_MERGING, p->center[2])) continue;
                SP merge = make_merged(*p, *nb);
                if (!cand_merge || cand_merge->mse > merge->mse) { cand_merge = merge; cand_nb = nb; }
            }
            if (cand_merge && cand_merge->mse < params.T_mse(ParamSet::P_MERGING, cand_merge->center[2])) {
                minQ.push(cand_merge);
                mergeNbsFrom(*cand_merge, *p, *cand_nb);
                graveyard.push_back(p);     // neighbours hold raw pointers; keep merged parents alive for the frame
            } else {
                if (p->N >= minSupport) extractedPlanes.push_back(p);
                p->disconnectAllNbs();
            }
            ++step;
        }
        while (!minQ.empty()) { SP p = minQ.top(); minQ.pop(); if (p->N >= minSupport) extractedPlanes.push_back(p); p->disconnectAllNbs(); }
        std::sort(extractedPlanes.begin(), extractedPlanes.end(), [](const SP& a, const SP& b) { return b->N < a->N; });
        return step;
    }
    std::vector<SP> graveyard;

    static int nb4(int i, int j, int H, int W, int nbs[4]) { const int id = i * W + j; int c = 0; if (j > 0) nbs[c++] = id - 1; if (j < W - 1) nbs[c++] = id + 1; if (i > 0) nbs[c++] = id - W; if (i < H - 1) nbs[c++] = id + W; return c; }
    int getBlockIdx(int px, int py) const { const int Nw = width / windowWidth, Nh = height / windowHeight, by = py / windowHeight, bx = px / windowWidth; return (by < Nh && bx < Nw) ? (by * Nw + bx) : -1; }

    void findBlockMembership(std::vector<bool>& isValid) {       // AHCPlaneFitter.hpp:603-705 (ERODE_ALL_BORDER)
        rid2plid.clear();
        for (int plid = 0; plid < (int)extractedPlanes.size(); ++plid) rid2plid.insert({extractedPlanes[plid]->rid, plid});
        const int Nh = height / windowHeight, Nw = width / windowWidth, NptsPerBlk = windowHeight * windowWidth;
        membershipImg.create(width, height, 1, -1);
        blkMap.assign((size_t)Nh * Nw, 0);
        isValid.assign(extractedPlanes.size(), false);
        for (int i = 0, blkid = 0; i < Nh; ++i) for (int j = 0; j < Nw; ++j, ++blkid) {
            const int setid = ds->Find(blkid), setSize = ds->getSetSize(setid) * NptsPerBlk;
            if (setSize >= minSupport) {
                int nbs[4]; const int nN = nb4(i, j, Nh, Nw, nbs);
                bool same = true;
                for (int k = 0; k < nN; ++k) if (ds->Find(nbs[k]) != setid) { same = false; break; }
                const int plid = rid2plid[setid];
                if (same) {
                    blkMap[blkid] = plid;
                    for (int y = i * windowHeight; y < (i + 1) * windowHeight; y++) for (int x = j * windowWidth; x < (j + 1) * windowWidth; x++) membershipImg.at(y, x) = plid;
                    isValid[plid] = true;
                } else blkMap[blkid] = -1;
            } else blkMap[blkid] = -1;
            if (blkMap[blkid] < 0) {
                if (i > 0) { const int u = blkid - Nw; if (blkMap[u] >= 0) { const int up = blkMap[u], s = (i * windowHeight - 1) * width + j * windowWidth; for (int k = 1; k < windowWidth; ++k) rfQueue.push_back({s + k, up}); } }
                if (j > 0) { const int l = blkid - 1; if (blkMap[l] >= 0) { const int lp = blkMap[l], s = (i * windowHeight) * width + j * windowWidth - 1; for (int k = 0; k < windowHeight - 1; ++k) rfQueue.push_back({s + k * width, lp}); } }
            } else {
                const int plid = blkMap[blkid];
                if (i > 0) { const int u = blkid - Nw; if (blkMap[u] != plid) { const int s = (i * windowHeight) * width + j * windowWidth; for (int k = 0; k < windowWidth - 1; ++k) rfQueue.push_back({s + k, plid}); } }
                if (j > 0) { const int l = blkid - 1; if (blkMap[l] != plid) { const int s = (i * windowHeight) * width + j * windowWidth; for (int k = 1; k < windowHeight; ++k) rfQueue.push_back({s + k * width, plid}); } }
            }
        }
    }

    void floodFill() {                                           // AHCPlaneFitter.hpp:546-594
        std::vector<float> distMap((size_t)height * width, std::numeric_limits<float>::max());
        for (int k = 0; k < (int)rfQueue.size(); ++k) {
            const int sIdx = rfQueue[k].first, seedy = sIdx / width, seedx = sIdx - seedy * width, plid = rfQueue[k].second;
            const PlaneSeg& pl = *extractedPlanes[plid];
            int nbs[4]; const int nN = nb4(seedy, seedx, height, width, nbs);
            for (int it = 0; it < nN; ++it) {
                const int cIdx = nbs[it]; int& trail = membershipImg.d[cIdx];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = cIdx / width, cx = cIdx - cy * width, blkid = getBlockIdx(cx, cy);
                if (blkid >= 0 && blkMap[blkid] >= 0) continue;
                double pt[3] = {0, 0, 0}; float cdist = -1;
                if (points->get(cy, cx, pt[0], pt[1], pt[2]) && std::pow(cdist = (float)std::fabs(pl.signedDist(pt)), 2) < 9 * pl.mse + 1e-5) {
                    if (trail >= 0) { PlaneSeg& n_pl = *extractedPlanes[trail]; if (pl.normalSimilarity(n_pl) >= params.T_ang(ParamSet::P_REFINE, pl.center[2])) n_pl.connect(extractedPlanes[plid].get()); }
                    float& old_dist = distMap[cIdx];
                    if (cdist < old_dist) { trail = plid; old_dist = cdist; rfQueue.push_back({cIdx, plid}); }
                    else if (trail < 0) trail -= 1;
                } else if (trail < 0) trail -= 1;
            }
        }
    }

    // run + refineDetails: returns the plane-contour image (AHCPlaneFitter.hpp:186-236, 274-400)
    void run(const Cloud& cloud, Img8& planeContours) {
        extractedPlanes.clear(); rid2plid.clear(); blkMap.clear(); rfQueue.clear(); graveyard.clear(); seq = 0;
        points = &cloud; height = cloud.h; width = cloud.w;
        ds.reset(new DisjointSet((height / windowHeight) * (width / windowWidth)));
        Queue minQ; std::vector<SP> keep;
        initGraph(minQ, keep);
        ahCluster(minQ);
        std::vector<bool> isValid;
        findBlockMembership(isValid);
        floodFill();
        std::vector<SP> old; extractedPlanes.swap(old);
        Queue q2; for (int i = 0; i < (int)old.size(); ++i) if (isValid[i]) q2.push(old[i]);
        ahCluster(q2);
        std::vector<int> plidmap(old.size(), -1); int nFinal = 0;
        for (int i = 0; i < (int)old.size(); ++i) {
            const PlaneSeg& op = *old[i];
            if (!isValid[i]) { plidmap[i] = -1; continue; }
            int np_rid;
            if ((np_rid = ds->Find(op.rid)) == op.rid) { if (plidmap[i] < 0) plidmap[i] = nFinal++; }
            else { const int npid = rid2plid[np_rid]; if (plidmap[npid] < 0) plidmap[i] = plidmap[npid] = nFinal++; else plidmap[i] = plidmap[npid]; }
        }
        std::vector<Img8> planes(extractedPlanes.size());
        for (auto& p : planes) p.create(width, height, 1, 0);
        for (int i = 0, n = width * height; i < n; ++i) {
            int& plid = membershipImg.d[i];
            if (plid >= 0 && plidmap[plid] >= 0) { plid = plidmap[plid]; if (plid < (int)planes.size()) planes[plid].d[i] = 255; }
        }
        StructElem e3 = ellipse_elem(3);
        for (auto& one : planes) {
            Img8 closed; morph_close(one, closed, e3);
            std::vector<Contour> cs; find_contours(closed, cs, true);
            for (const Contour& c : cs) draw_contour_thick2(planeContours, c, 255);
        }
        graveyard.clear(); points = nullptr;
    }
};

}  // namespace peac
