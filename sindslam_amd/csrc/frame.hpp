// Frame post-ORB stage interface (reference src/Frame.cc:143-170); kernels in frame_kernels.hip.
#pragma once
#include "common.hpp"

namespace sind {

constexpr int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64, FRAME_CELLS = FRAME_GRID_ROWS * FRAME_GRID_COLS;   // include/Frame.h:37-38

struct FrameCalib { float fx, fy, cx, cy, k1, k2, p1, p2, k3, bf, depthMapFactor; };

// all pointers device; kxy [B][cap][2], nkp [B], depth [B][H][W] u16; outputs [B][cap] (un_xy x2), grid_start [B][3073], grid_idx [B][cap]
int launch_frame_post_orb(const FrameCalib& c, const float* kxy, const int* nkp, int B, int cap, const uint16_t* depth, int W, int H, float* un_xy,
                          float* u_right, float* depth_out, int* cell, int* grid_start, int* grid_idx, float* bounds, hipStream_t s);

}  // namespace sind
