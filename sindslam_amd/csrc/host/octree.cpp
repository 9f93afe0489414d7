// Host stage: ORBextractor::DistributeOctTree / ExtractorNode::DivideNode (reference ORBextractor.cc:481-763).
// Serial by nature (a few thousand points per level) -> stays on the host, see DESIGN.md "host stages".
// Node storage is an index pool with an intrusive doubly linked list that reproduces the reference's
// std::list order (children are pushed to the FRONT, the parent is erased in place).  The reference sorts
// (size, node address) pairs; addresses are not reproducible, so ties are broken by creation order.
#include <algorithm>
#include <cmath>
#include "../orb.hpp"

namespace sind {
namespace {
struct Node {
    std::vector<OctKp> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int prev = -1, next = -1;
    bool noMore = false, alive = false;
};
struct NodeList {
    std::vector<Node> pool; int head = -1, tail = -1, count = 0;
    int make() { pool.emplace_back(); return (int)pool.size() - 1; }
    void push_back(int n) { Node& a = pool[n]; a.alive = true; a.prev = tail; a.next = -1; if (tail >= 0) pool[tail].next = n; else head = n; tail = n; count++; }
    void push_front(int n) { Node& a = pool[n]; a.alive = true; a.next = head; a.prev = -1; if (head >= 0) pool[head].prev = n; else tail = n; head = n; count++; }
    int erase(int n) { Node& a = pool[n]; int nx = a.next; if (a.prev >= 0) pool[a.prev].next = a.next; else head = a.next; if (a.next >= 0) pool[a.next].prev = a.prev; else tail = a.prev; a.alive = false; count--; return nx; }
};
// children indices in creation order n1..n4 (-1 if the child holds no point)
void divide(NodeList& L, int parent, int child[4]) {
    for (int k = 0; k < 4; k++) child[k] = L.make();
    Node& p = L.pool[parent];      // pool may have been reallocated by make(): take the reference afterwards
    const int halfX = (int)std::ceil(static_cast<float>(p.URx - p.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.BRy - p.ULy) / 2);
    Node &n1 = L.pool[child[0]], &n2 = L.pool[child[1]], &n3 = L.pool[child[2]], &n4 = L.pool[child[3]];
    n1.ULx = p.ULx; n1.ULy = p.ULy; n1.URx = p.ULx + halfX; n1.URy = p.ULy; n1.BLx = p.ULx; n1.BLy = p.ULy + halfY; n1.BRx = p.ULx + halfX; n1.BRy = p.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = p.URx; n2.URy = p.URy; n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = p.URx; n2.BRy = p.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy; n3.BLx = p.BLx; n3.BLy = p.BLy; n3.BRx = n1.BRx; n3.BRy = p.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy; n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = p.BRx; n4.BRy = p.BRy;
    for (const OctKp& kp : p.keys) {
        if (kp.x < n1.URx) { if (kp.y < n1.BRy) n1.keys.push_back(kp); else n3.keys.push_back(kp); }
        else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    for (int k = 0; k < 4; k++) if (L.pool[child[k]].keys.size() == 1) L.pool[child[k]].noMore = true;
}
}  // namespace

void distribute_octree(const std::vector<OctKp>& in, int minX, int maxX, int minY, int maxY, int N, std::vector<OctKp>& out) {
    out.clear();
    NodeList L; L.pool.reserve(in.size() * 4 + 64);
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::vector<int> ini(nIni);
    for (int i = 0; i < nIni; i++) {
        int n = L.make(); Node& a = L.pool[n];
        a.ULx = (int)(hX * static_cast<float>(i)); a.ULy = 0; a.URx = (int)(hX * static_cast<float>(i + 1)); a.URy = 0;
        a.BLx = a.ULx; a.BLy = maxY - minY; a.BRx = a.URx; a.BRy = maxY - minY;
        L.push_back(n); ini[i] = n;
    }
    for (const OctKp& kp : in) L.pool[ini[(int)(kp.x / hX)]].keys.push_back(kp);
    for (int n = L.head; n >= 0;) {
        Node& a = L.pool[n];
        if (a.keys.size() == 1) { a.noMore = true; n = a.next; }
        else if (a.keys.empty()) n = L.erase(n);
        else n = a.next;
    }
    typedef std::pair<int, int> SP;    // (number of keys, node index == creation order)
    std::vector<SP> expand;
    auto add_children = [&](const int child[4], int* nToExpand) {
        for (int k = 0; k < 4; k++) {
            const int c = child[k];
            if (L.pool[c].keys.empty()) continue;
            L.push_front(c);
            if (L.pool[c].keys.size() > 1) { if (nToExpand) (*nToExpand)++; expand.push_back({(int)L.pool[c].keys.size(), c}); }
        }
    };
    bool finish = false;
    while (!finish) {
        int prevSize = L.count, nToExpand = 0;
        expand.clear();
        for (int n = L.head; n >= 0;) {
            if (L.pool[n].noMore) { n = L.pool[n].next; continue; }
            int child[4]; divide(L, n, child);
            add_children(child, &nToExpand);
            n = L.erase(n);
        }
        if (L.count >= N || L.count == prevSize) finish = true;
        else if (L.count + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.count;
                std::vector<SP> prev = expand; expand.clear();
                std::sort(prev.begin(), prev.end());
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    int child[4]; divide(L, prev[j].second, child);
                    add_children(child, nullptr);
                    L.erase(prev[j].second);
                    if (L.count >= N) break;
                }
                if (L.count >= N || L.count == prevSize) finish = true;
            }
        }
    }
    for (int n = L.head; n >= 0; n = L.pool[n].next) {
        const std::vector<OctKp>& k = L.pool[n].keys;
        const OctKp* best = &k[0]; float mx = best->response;
        for (size_t i = 1; i < k.size(); i++) if (k[i].response > mx) { best = &k[i]; mx = k[i].response; }
        out.push_back(*best);
    }
}

}  // namespace sind
