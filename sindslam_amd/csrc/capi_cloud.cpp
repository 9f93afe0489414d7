// C ABI: key-frame cloud generation of the mapping consumer (include/sind_hip.h, "sind_cloud_*").
#include <cstring>
#include "../../include/sind_hip.h"
#include "cloud.hpp"

struct sind_cloud {
    int device = 0, W = 0, H = 0, maxB = 0, np = 0, nchunks = 0; sind::CloudCam cam{}; hipStream_t stream = nullptr;
    DevBuf<uint8_t> bgr, dyna, dynaLast, label; DevBuf<uint16_t> depth, depthLast; DevBuf<sind::CloudPose> pose;
    DevBuf<int> chunkCnt, chunkOff, occ, labelCount, kept, total; DevBuf<sind::CloudPoint> out;
    std::vector<sind::CloudPose> h_pose; std::vector<int> h_total;
};

extern "C" {

int sind_cloud_create(double fx, double fy, double cx, double cy, double depth_scale, int width, int height, int max_batch, int device, sind_cloud** out) {
    if (!out || width < 2 || height < 2 || max_batch < 1 || !(fx > 0) || !(fy > 0) || !(depth_scale > 0)) { sind_set_error("sind_cloud_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    sind_cloud* c = new sind_cloud(); c->device = device; c->W = width; c->H = height; c->maxB = max_batch; c->cam = {fx, fy, cx, cy, depth_scale};
    c->np = sind::cloud_grid_points(width, height); c->nchunks = divup(c->np, sind::CLOUD_CHUNK);
    const size_t B = max_batch, px = (size_t)width * height;
    int r = SIND_OK;
    if ((r = c->bgr.alloc(B * px * 3)) || (r = c->dyna.alloc(B * px)) || (r = c->dynaLast.alloc(B * px)) || (r = c->label.alloc(B * px)) || (r = c->depth.alloc(B * px)) ||
        (r = c->depthLast.alloc(B * px)) || (r = c->pose.alloc(B)) || (r = c->chunkCnt.alloc(B * c->nchunks * sind::CLOUD_LABELS)) || (r = c->chunkOff.alloc(B * c->nchunks * sind::CLOUD_LABELS)) ||
        (r = c->occ.alloc(B * sind::CLOUD_LABELS)) || (r = c->labelCount.alloc(B * sind::CLOUD_LABELS)) || (r = c->kept.alloc(B * sind::CLOUD_LABELS)) || (r = c->total.alloc(B)) ||
        (r = c->out.alloc(B * c->np))) { delete c; return r; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; sind_set_error("sind_cloud_create: stream creation failed"); return SIND_E_HIP; }
    c->h_pose.resize(B); c->h_total.resize(B);
    *out = c; return SIND_OK;
}
int sind_cloud_destroy(sind_cloud* c) {
    if (!c) return SIND_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    hipStream_t s = c->stream; delete c; if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}
int sind_cloud_max_points(sind_cloud* c) { return c ? c->np : SIND_E_ARG; }

int sind_cloud_generate(sind_cloud* c, int B, const uint8_t* bgr, const uint16_t* depth, const uint16_t* depth_last, const uint8_t* dyna, const uint8_t* dyna_last,
                        const uint8_t* label, const double* pose_relative, const double* Twc, int inputs_on_device, sind_cloud_point* points, int cap, int* n_points,
                        int* occlusion, int* label_count, int* kept) {
    if (!c || B < 1 || B > c->maxB || !bgr || !depth || !depth_last || !dyna || !dyna_last || !label || !pose_relative || !Twc || !n_points || (points && cap < 1)) {
        sind_set_error("sind_cloud_generate: bad arguments (B=%d, max %d)", B, c ? c->maxB : 0); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream; const size_t px = (size_t)c->W * c->H;
    for (int b = 0; b < B; b++) { std::memcpy(c->h_pose[b].rel, pose_relative + 16 * b, 12 * sizeof(double)); std::memcpy(c->h_pose[b].twc, Twc + 16 * b, 12 * sizeof(double)); }
    HIP_TRY(hipMemcpyAsync(c->pose.p, c->h_pose.data(), (size_t)B * sizeof(sind::CloudPose), hipMemcpyHostToDevice, s));
    sind::CloudArrays a{};
    if (inputs_on_device) { a.bgr = bgr; a.depth = depth; a.depthLast = depth_last; a.dyna = dyna; a.dynaLast = dyna_last; a.label = label; }
    else {
        HIP_TRY(hipMemcpyAsync(c->bgr.p, bgr, B * px * 3, hipMemcpyHostToDevice, s)); HIP_TRY(hipMemcpyAsync(c->depth.p, depth, B * px * 2, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->depthLast.p, depth_last, B * px * 2, hipMemcpyHostToDevice, s)); HIP_TRY(hipMemcpyAsync(c->dyna.p, dyna, B * px, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->dynaLast.p, dyna_last, B * px, hipMemcpyHostToDevice, s)); HIP_TRY(hipMemcpyAsync(c->label.p, label, B * px, hipMemcpyHostToDevice, s));
        a.bgr = c->bgr.p; a.depth = c->depth.p; a.depthLast = c->depthLast.p; a.dyna = c->dyna.p; a.dynaLast = c->dynaLast.p; a.label = c->label.p;
    }
    a.pose = c->pose.p; a.chunkCnt = c->chunkCnt.p; a.chunkOff = c->chunkOff.p; a.occlusion = c->occ.p; a.labelCount = c->labelCount.p; a.kept = c->kept.p; a.total = c->total.p; a.out = c->out.p;
    SIND_TRY(sind::launch_cloud(c->cam, a, c->W, c->H, B, s));
    HIP_TRY(hipMemcpyAsync(c->h_total.data(), c->total.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    if (occlusion) HIP_TRY(hipMemcpyAsync(occlusion, c->occ.p, (size_t)B * sind::CLOUD_LABELS * sizeof(int), hipMemcpyDeviceToHost, s));
    if (label_count) HIP_TRY(hipMemcpyAsync(label_count, c->labelCount.p, (size_t)B * sind::CLOUD_LABELS * sizeof(int), hipMemcpyDeviceToHost, s));
    if (kept) HIP_TRY(hipMemcpyAsync(kept, c->kept.p, (size_t)B * sind::CLOUD_LABELS * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int b = 0; b < B; b++) {
        n_points[b] = c->h_total[b];
        if (!points) continue;
        if (c->h_total[b] > cap) { sind_set_error("sind_cloud_generate: frame %d has %d points, capacity %d", b, c->h_total[b], cap); return SIND_E_CAPACITY; }
        HIP_TRY(hipMemcpyAsync(points + (size_t)b * cap, c->out.p + (size_t)b * c->np, (size_t)c->h_total[b] * sizeof(sind::CloudPoint), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SIND_OK;
}

}  // extern "C"
