// GPU region grow of the PEAC plane refinement (peac_kernels.hip) and its host-side batch object.
#pragma once
#include "common.hpp"
#include "host/host.hpp"

namespace sind {

#define PG_THREADS 512
#define PG_FRONT_CAP 65536            /* seeds of one BFS level (measured: <= 7.2 k at 640 x 480) */
#define PG_MAX_BLOCKS 3600            /* 16 x 16 blocks of a frame (1280 x 720) */
#define PG_MAX_LEVELS 100000
enum { PG_OK = 0, PG_ERR_FRONTIER = 1, PG_ERR_SLOTS = 2, PG_ERR_LEVELS = 3, PG_SKIPPED = 4 };

// one frame's input block (page-locked on the host, copied as it is): header, plane constants, eroded block map, initial seeds
struct PeacGrowHdr { int nPl, nSeeds, skip, depth_index; };
#define PG_OFF_PLANES 16
#define PG_OFF_BLOCKS (PG_OFF_PLANES + PEAC_GROW_MAX_PLANES * 64)
#define PG_OFF_SEEDS(nb) (PG_OFF_BLOCKS + PG_MAX_BLOCKS)
#define PG_IN_STRIDE (PG_OFF_BLOCKS + PG_MAX_BLOCKS + 4 * PEAC_GROW_MAX_SEEDS0)
static_assert(sizeof(PeacGrowPlane) == 64 && PG_OFF_BLOCKS % 16 == 0 && (PG_OFF_BLOCKS + PG_MAX_BLOCKS) % 16 == 0 && PG_IN_STRIDE % 16 == 0, "input block layout");

struct PeacGrowArgs {
    int W, H; float fx, fy, cx, cy, inv_scale; unsigned w_magic;                  // w_magic = ceil(2^32 / W): pixel / W for pixel < 2^20 is one v_mul_hi
    const uint8_t* in; size_t in_stride;                // frames' input blocks
    const uint16_t* depth_base;                         // frame f reads depth_base + hdr.depth_index * W * H
    int8_t* member; float* dist; unsigned long long* slot; unsigned *slot_ext, *frontier, *payload, *active; uint8_t* pair_seen; int* status;
};
int launch_peac_grow(hipStream_t s, const PeacGrowArgs& A, int frames);

// Device workspace for up to `cap` frames per launch.  Inputs are packed by the caller into page-locked blocks (PG_IN_STRIDE bytes per frame);
// results (membership map as int8, pair matrix, status) land in page-locked memory of the caller.
class PeacGrowBatch {
public:
    // sizes the kernel handles (whole 16 x 16 blocks, at most PG_MAX_BLOCKS of them, pixel index below 2^20); frames of any other size are grown by the
    // host statement of the same FIFO (PeacFitter::grow_host) -- callers ask before they create a workspace
    static bool supports(int W, int H) { return W > 0 && H > 0 && W % 16 == 0 && H % 16 == 0 && (W / 16) * (H / 16) <= PG_MAX_BLOCKS && (size_t)W * H <= (1u << 20); }
    int init(int W, int H, float fx, float fy, float cx, float cy, float depthScale, int cap);
    // in_h: `frames` consecutive input blocks (page-locked); member_h: frames x W*H int8, pair_h: frames x 127^2, status_h: frames x 4 ints (all page-locked).
    // Enqueues copy-in, kernel and copies back on s; the caller records / waits.
    int run(hipStream_t s, const uint8_t* in_h, const uint16_t* depth_base, int frames, int8_t* member_h, uint8_t* pair_h, int* status_h);
    int cap = 0;
private:
    int W = 0, H = 0; float fx = 0, fy = 0, cx = 0, cy = 0, inv_scale = 0;
    DevBuf<uint8_t> in_d, pair_d; DevBuf<int8_t> member_d; DevBuf<float> dist_d; DevBuf<unsigned long long> slot_d; DevBuf<unsigned> ext_d, front_d, payload_d, active_d; DevBuf<int> status_d;
};
// write one frame's input block from a fitter that finished part1 (skip = 1 when the frame does not fit the kernel's capacities: the caller grows it on the host)
void peac_grow_pack(const PeacFitter& f, int depth_index, uint8_t* block);

}  // namespace sind
