// Host-stage interfaces (serial graph logic that stays on the CPU by design, DESIGN.md "host stages").
#pragma once
#include <cstdint>
#include <vector>
#include "bitimg.hpp"
#include "contours.hpp"
#include "../depth.hpp"

namespace sind {

struct Pt2f { float x, y; };
bool find_homography_rho(const std::vector<Pt2f>& src, const std::vector<Pt2f>& dst, double H[9]);

// PEAC plane-contour extraction (reference include/PEAC/*, called from DynaDetect.cc:592-593).
// blocks: per 16x16 window statistics computed on the GPU (k_peac_block_stats); depth: host copy of the raw depth.
struct PeacInput { const PeacBlockStats* blocks; const uint16_t* depth; int w, h; float fx, fy, cx, cy, depthScale; };
void peac_plane_contours(const PeacInput& in, BitImg& planeContours);      // part1 + grow_host + part2
// The same in three steps, so that the region grow (AHCPlaneFitter.hpp:546-601 floodFill) can run on the GPU between the two host parts.
struct PeacGrowPlane { double n[3], c[3], thr, pad; };          // distance test of one plane: |n . (p - c)|^2 < thr
#define PEAC_GROW_MAX_PLANES 127                                 /* plane index fits the int8 membership map */
#define PEAC_GROW_MAX_SEEDS0 65536                               /* initial seeds (block-border pixels of the eroded planes) */
#define PEAC_GROW_SLOTS 4                                        /* seeds that may share a pixel within one BFS level */
class PeacFitter {
public:
    explicit PeacFitter(const PeacInput& in); ~PeacFitter();
    PeacFitter(const PeacFitter&) = delete; PeacFitter& operator=(const PeacFitter&) = delete;
    void part1();                                               // graph clustering -> planes, eroded block map, seeds, per-plane constants
    bool gpu_ok = false;                                         // the inputs fit the kernel's fixed capacities (else: grow_host)
    int n_planes() const; int n_blocks() const;
    const PeacGrowPlane* planes() const; const int8_t* block_map() const;      // block_map: plane of every eroded 16 x 16 block or -1
    const std::vector<uint32_t>& seed_words() const;            // initial frontier in FIFO order, one word per seed (see peac_kernels.hip)
    // membership per pixel (plane index or -1) as int8 when there are <= 127 planes, else int16; pairSeen [nPl][nPl]
    void grow_host(std::vector<int8_t>& member8, std::vector<int16_t>& member16, std::vector<uint8_t>& pairSeen);
    void part2(const int8_t* member8, const int16_t* member16, const uint8_t* pairSeen, BitImg& planeContours);
private:
    void* f;
};

}  // namespace sind
