// Depth-side device stage interface (host side of depth_kernels.hip).
#pragma once
#include <cmath>
#include "common.hpp"

namespace sind {

#define KM_K 12               /* numCluster = 3 x 4, reference DynaDetect.cc:46-47 */
#define KM_MAX_BLOCKS 64
#define MORPH_MAX 16

struct KmState { float ctr[KM_K][3], old[KM_K][3], base[3]; int cnt[KM_K]; int iter, done, phase, fix_k, max_k, maxCount, overflow; double eps2; unsigned long long far; };
struct MorphElem { int n, ax, ay; int j1[MORPH_MAX], j2[MORPH_MAX]; };
// moments of a 16 x 16 block + (fitted = 1: filled by k_peac_block_fit) its plane fit, so that the host's graph clustering starts from fitted nodes
struct PeacBlockStats { double sx, sy, sz, sxx, syy, szz, sxy, syz, sxz; int N, valid; double mse, center[3], normal[3]; int fitted, pad; };

MorphElem make_ellipse(int n);
// k-means chain: every launcher takes a batch of B frames (default one) with per-frame plane strides in elements
int launch_depth_half(hipStream_t s, const uint16_t* src, uint16_t* dst, int dw, int dh, int B = 1, size_t src_stride = 0, size_t dst_stride = 0);
int launch_points(hipStream_t s, const uint16_t* depth, float* px, float* py, float* pz, int w, int h, float scale, float fx, float fy, float cx, float cy, float depthScale,
                  int B = 1, size_t depth_stride = 0, size_t pt_stride = 0);
// use_prev (device, one int per frame, or nullptr): the grid labels go to the frames with use_prev == 0, the resized previous labels to the others
int launch_labels_grid(hipStream_t s, int* labels, int w, int h, int B = 1, size_t stride = 0, const int* use_prev = nullptr);
int launch_labels_resize_u8(hipStream_t s, const uint8_t* src, int* dst, int sw, int sh, int dw, int dh, int B = 1, size_t src_stride = 0, size_t dst_stride = 0, const int* use_prev = nullptr);
int launch_labels_resize_i32(hipStream_t s, const int* src, int* dst, int sw, int sh, int dw, int dh, int B = 1, size_t src_stride = 0, size_t dst_stride = 0);
// segcnt: (KM_MAX_BLOCKS * 4 + 1) * KM_K ints (per wave-segment cluster counts + the totals row) + 64 floats (the pass's 36 sequential sums); comp: 3 * n floats (per-cluster runs of every coordinate)
#define KM_SEG_WORDS ((KM_MAX_BLOCKS * 4 + 1) * KM_K + 64)
// when a level runs in the fused one-launch kernel: levels of at most `max_points` points (640 x 480: 4 800 / 19 200 / 76 800; 1280 x 720: 14 400 / 57 600; 0 = never) of batches of
// at least `min_batch` frames.  A handle copies the defaults when it is created and keeps them (the parity tests set other defaults before creating theirs).
struct KmFuse { int max_points = 81920, min_batch = 32; };
extern KmFuse g_km_fuse_default;
int launch_kmeans_level(hipStream_t s, const KmFuse& fuse, const float* px, const float* py, const float* pz, int* labels, int n, int* segcnt, float* comp, KmState* st,
                        int maxCount, double eps2, int B = 1, size_t pt_stride = 0, size_t lab_stride = 0, size_t seg_stride = 0, size_t comp_stride = 0, size_t st_stride = 0);
int debug_seqsum(hipStream_t s, const float* x_dev, int n, int* scratch_dev, float* out_host);
int launch_labels_to_u8(hipStream_t s, const int* labels, uint8_t* out, int n, int B = 1, size_t lab_stride = 0, size_t out_stride = 0);
int launch_median5(hipStream_t s, const uint16_t* src, uint16_t* dst, int w, int h, int B = 1);
int launch_max_u16(hipStream_t s, const uint16_t* src, int n, unsigned* out, int B = 1, int out_stride = 1);
int launch_grad_edge(hipStream_t s, const uint16_t* filt, const unsigned* dmax, uint8_t* edge, uint8_t* total_area, int w, int h, float depthScale, int B = 1, int dmax_stride = 1);
int launch_morph(hipStream_t s, const uint8_t* src, uint8_t* dst, int w, int h, int n, bool dilate, int B = 1);
int launch_peac_block_stats(hipStream_t s, const uint16_t* depth, int w, int h, int bw, int bh, float fx, float fy, float cx, float cy, float depthScale, PeacBlockStats* out, int B = 1);
int launch_depth_norm(hipStream_t s, const uint16_t* depth, const unsigned* dmax, uint8_t* out, int n, int B = 1, int dmax_stride = 1);
int launch_dilate_planes(hipStream_t s, const unsigned long long* src, unsigned long long* dst, int nplanes, int w, int h, int n, int* zero = nullptr, int nzero = 0);
int launch_rag_stats(hipStream_t s, const unsigned long long* planes, int C, int w, int h, int wpr, const uint8_t* occ2, const uint8_t* depthN,
                     int* overlap, int* overlapPlane, int* ljOverlap, int* ljArea, int* hist, bool already_zero = false);

}  // namespace sind
