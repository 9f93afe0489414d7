// Mapping-consumer cloud generation interface (reference octomap_pub/src/pubPointCloud.cc:471-660); kernels in cloud_kernels.hip.
#pragma once
#include "common.hpp"

namespace sind {

constexpr int CLOUD_LABELS = 12, CLOUD_CHUNK = 1024;

struct CloudCam { double fx, fy, cx, cy, depthScale; };
struct CloudPose { double rel[12], twc[12]; };                    // rows 0..2 of poseRelative and of Twc, row-major
struct CloudPoint { float x, y, z; uint8_t b, g, r, a; };

struct CloudArrays {                                              // device pointers, dense per frame
    const uint8_t* bgr; const uint16_t* depth; const uint16_t* depthLast; const uint8_t* dyna; const uint8_t* dynaLast; const uint8_t* label; const CloudPose* pose;
    int* chunkCnt;        // [B][nchunks][12]  stride-2 pixels per label and chunk
    int* chunkOff;        // [B][nchunks][12]  output offset of each (chunk, label), -1 when the cluster is rejected
    int* occlusion;       // [B][12]           vecOcclusion
    int* labelCount;      // [B][12]           countNonZero(imgLabel == i), full resolution
    int* kept;            // [B][12]
    int* total;           // [B]
    CloudPoint* out;      // [B][np]
};

int launch_cloud(const CloudCam& cam, const CloudArrays& a, int W, int H, int B, hipStream_t s);
inline int cloud_grid_points(int W, int H) { return ((W + 1) / 2) * ((H + 1) / 2); }

}  // namespace sind
