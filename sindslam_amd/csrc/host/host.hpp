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
void peac_plane_contours(const PeacInput& in, BitImg& planeContours);

}  // namespace sind
