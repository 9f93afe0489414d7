// C ABI: Frame post-ORB steps (include/sind_hip.h, "sind_frame_*").
#include <cstring>
#include "../../include/sind_hip.h"
#include "frame.hpp"

struct sind_frame {
    int device = 0, W = 0, H = 0, maxB = 0, cap = 0; sind::FrameCalib calib{}; hipStream_t stream = nullptr;
    DevBuf<float> kxy, un, ur, dep, bounds; DevBuf<int> nkp, cell, gstart, gidx; DevBuf<uint16_t> depth;
    std::vector<float> kxy_h;
};

extern "C" {

int sind_frame_create(const sind_frame_calib* c, int width, int height, int max_batch, int cap, int device, sind_frame** out) {
    if (!c || !out || width < 1 || height < 1 || max_batch < 1 || cap < 1 || !(c->fx > 0) || !(c->fy > 0)) { sind_set_error("sind_frame_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    sind_frame* f = new sind_frame(); f->device = device; f->W = width; f->H = height; f->maxB = max_batch; f->cap = cap;
    f->calib = {c->fx, c->fy, c->cx, c->cy, c->k1, c->k2, c->p1, c->p2, c->k3, c->bf, c->depth_map_factor};
    const size_t n = (size_t)max_batch * cap;
    int r = SIND_OK;
    if ((r = f->kxy.alloc(n * 2)) || (r = f->un.alloc(n * 2)) || (r = f->ur.alloc(n)) || (r = f->dep.alloc(n)) || (r = f->bounds.alloc(4)) || (r = f->nkp.alloc(max_batch)) ||
        (r = f->cell.alloc(n)) || (r = f->gstart.alloc((size_t)max_batch * (sind::FRAME_CELLS + 1))) || (r = f->gidx.alloc(n))) { delete f; return r; }
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) { delete f; sind_set_error("sind_frame_create: stream creation failed"); return SIND_E_HIP; }
    f->kxy_h.resize(n * 2);
    *out = f; return SIND_OK;
}
int sind_frame_destroy(sind_frame* f) {
    if (!f) return SIND_OK;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    hipStream_t s = f->stream; delete f; if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}

int sind_frame_post_orb(sind_frame* f, const sind_keypoint* kps, const int* nkp, int B, const uint16_t* depth, int depth_on_device,
                        float* un_xy, float* u_right, float* depth_out, int* cell, int* grid_start, int* grid_idx, float* bounds4) {
    if (!f || !nkp || !depth || B < 1 || B > f->maxB || (!kps)) { sind_set_error("sind_frame_post_orb: bad arguments (B=%d, max %d)", B, f ? f->maxB : 0); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(f->device));
    const int cap = f->cap; const size_t n = (size_t)B * cap, fpx = (size_t)f->W * f->H;
    for (int b = 0; b < B; b++) {
        if (nkp[b] < 0 || nkp[b] > cap) { sind_set_error("sind_frame_post_orb: nkp[%d] = %d outside [0,%d]", b, nkp[b], cap); return SIND_E_CAPACITY; }
        float* o = &f->kxy_h[(size_t)b * cap * 2]; const sind_keypoint* k = kps + (size_t)b * cap;
        for (int i = 0; i < nkp[b]; i++) {
            if (!(k[i].x >= 0.f && k[i].x < (float)f->W && k[i].y >= 0.f && k[i].y < (float)f->H)) { sind_set_error("sind_frame_post_orb: keypoint %d of frame %d outside the image", i, b); return SIND_E_ARG; }
            o[2 * i] = k[i].x; o[2 * i + 1] = k[i].y;
        }
    }
    HIP_TRY(hipMemcpyAsync(f->kxy.p, f->kxy_h.data(), n * 2 * sizeof(float), hipMemcpyHostToDevice, f->stream));
    HIP_TRY(hipMemcpyAsync(f->nkp.p, nkp, (size_t)B * sizeof(int), hipMemcpyHostToDevice, f->stream));
    const uint16_t* dd = depth;
    if (!depth_on_device) { SIND_TRY(f->depth.alloc(fpx * f->maxB)); HIP_TRY(hipMemcpyAsync(f->depth.p, depth, fpx * B * sizeof(uint16_t), hipMemcpyHostToDevice, f->stream)); dd = f->depth.p; }
    SIND_TRY(sind::launch_frame_post_orb(f->calib, f->kxy.p, f->nkp.p, B, cap, dd, f->W, f->H, f->un.p, f->ur.p, f->dep.p, f->cell.p, f->gstart.p, f->gidx.p, f->bounds.p, f->stream));
    if (un_xy) HIP_TRY(hipMemcpyAsync(un_xy, f->un.p, n * 2 * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    if (u_right) HIP_TRY(hipMemcpyAsync(u_right, f->ur.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    if (depth_out) HIP_TRY(hipMemcpyAsync(depth_out, f->dep.p, n * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    if (cell) HIP_TRY(hipMemcpyAsync(cell, f->cell.p, n * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    if (grid_start) HIP_TRY(hipMemcpyAsync(grid_start, f->gstart.p, (size_t)B * (sind::FRAME_CELLS + 1) * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    if (grid_idx) HIP_TRY(hipMemcpyAsync(grid_idx, f->gidx.p, n * sizeof(int), hipMemcpyDeviceToHost, f->stream));
    if (bounds4) HIP_TRY(hipMemcpyAsync(bounds4, f->bounds.p, 4 * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream));
    return SIND_OK;
}

}  // extern "C"
