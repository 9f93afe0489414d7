// Host side of the GPU region grow (peac_kernels.hip): workspace, input packing, one enqueue per chunk of frames.
#include <cstring>
#include "peac_grow.hpp"

namespace sind {

int PeacGrowBatch::init(int W_, int H_, float fx_, float fy_, float cx_, float cy_, float depthScale, int cap_) {
    W = W_; H = H_; fx = fx_; fy = fy_; cx = cx_; cy = cy_; inv_scale = 1.0f / depthScale; cap = cap_;
    if (!supports(W, H)) { sind_set_error("PeacGrowBatch: unsupported size %d x %d", W, H); return SIND_E_ARG; }
    const size_t N = (size_t)W * H, c = (size_t)cap;
    SIND_TRY(in_d.alloc(c * PG_IN_STRIDE)); SIND_TRY(member_d.alloc(c * N)); SIND_TRY(dist_d.alloc(c * N)); SIND_TRY(slot_d.alloc(c * N)); SIND_TRY(ext_d.alloc(c * N * (PEAC_GROW_SLOTS - 1)));
    SIND_TRY(front_d.alloc(c * 2 * PG_FRONT_CAP)); SIND_TRY(payload_d.alloc(c * 4 * PG_FRONT_CAP)); SIND_TRY(active_d.alloc(c * 4 * PG_FRONT_CAP)); SIND_TRY(pair_d.alloc(c * PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES));
    SIND_TRY(status_d.alloc(c * 4));
    return SIND_OK;
}

int PeacGrowBatch::run(hipStream_t s, const uint8_t* in_h, const uint16_t* depth_base, int frames, int8_t* member_h, uint8_t* pair_h, int* status_h) {
    if (frames < 1 || frames > cap || !in_h || !depth_base || !member_h || !pair_h || !status_h) { sind_set_error("PeacGrowBatch::run: bad arguments (%d frames, capacity %d)", frames, cap); return SIND_E_ARG; }
    const size_t N = (size_t)W * H;
    HIP_TRY(hipMemcpyAsync(in_d.p, in_h, (size_t)frames * PG_IN_STRIDE, hipMemcpyHostToDevice, s));
    PeacGrowArgs A{W, H, fx, fy, cx, cy, inv_scale, (unsigned)(((1ull << 32) + (unsigned)W - 1) / (unsigned)W), in_d.p, (size_t)PG_IN_STRIDE, depth_base, member_d.p, dist_d.p, slot_d.p, ext_d.p, front_d.p, payload_d.p, active_d.p, pair_d.p, status_d.p};
    SIND_TRY(launch_peac_grow(s, A, frames));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(member_h, member_d.p, (size_t)frames * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(pair_h, pair_d.p, (size_t)frames * PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(status_h, status_d.p, (size_t)frames * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    return SIND_OK;
}

void peac_grow_pack(const PeacFitter& f, int depth_index, uint8_t* block) {
    PeacGrowHdr h{f.n_planes(), (int)f.seed_words().size(), f.gpu_ok ? 0 : 1, depth_index};
    if (h.skip) { h.nPl = 0; h.nSeeds = 0; std::memcpy(block, &h, sizeof(h)); return; }
    std::memcpy(block, &h, sizeof(h));
    std::memcpy(block + PG_OFF_PLANES, f.planes(), (size_t)h.nPl * sizeof(PeacGrowPlane));
    std::memcpy(block + PG_OFF_BLOCKS, f.block_map(), (size_t)f.n_blocks());
    std::memcpy(block + PG_OFF_SEEDS(0), f.seed_words().data(), (size_t)h.nSeeds * 4);
}

}  // namespace sind
