// C ABI: ORBextractor (include/sind_hip.h).
#include <cstdlib>
#include <cstring>
#include "../../include/sind_hip.h"
#include "orb.hpp"

struct sind_orb {
    int device = 0, nfeatures = 0, nlevels = 0, ini = 0, mn = 0; float sf = 1.2f;
    hipStream_t stream = nullptr; sind::OrbEngine eng; bool ready = false; int want_batch = 1;
    DevBuf<uint8_t> gray;
    std::vector<sind::OrbFrameResult> last;
};

static int ensure(sind_orb* o, int w, int h, int B) {
    if (o->ready && o->eng.W == w && o->eng.H == h && o->eng.maxB >= B) return SIND_OK;
    o->eng.~OrbEngine(); new (&o->eng) sind::OrbEngine();
    o->ready = false;
    SIND_TRY(o->eng.init(w, h, o->nfeatures, o->sf, o->nlevels, o->ini, o->mn, std::max(B, o->want_batch), o->stream));
    o->ready = true; return SIND_OK;
}

extern "C" {

int sind_orb_create(int nfeatures, float sf, int nlevels, int ini, int mn, int device, sind_orb** out) {
    if (!out || nfeatures < 1 || nlevels < 1 || nlevels > 16 || sf <= 1.0f) { sind_set_error("sind_orb_create: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    sind_orb* o = new sind_orb(); o->device = device; o->nfeatures = nfeatures; o->sf = sf; o->nlevels = nlevels; o->ini = ini; o->mn = mn;
    if (hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) != hipSuccess) { delete o; sind_set_error("sind_orb_create: hipStreamCreate failed"); return SIND_E_HIP; }
    *out = o; return SIND_OK;
}
int sind_orb_destroy(sind_orb* o) {
    if (!o) return SIND_OK;
    (void)hipSetDevice(o->device);
    if (o->stream) { (void)hipStreamSynchronize(o->stream); }
    hipStream_t s = o->stream; delete o; if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}
int sind_orb_reserve(sind_orb* o, int w, int h, int B) { if (!o || B < 1) return SIND_E_ARG; HIP_TRY(hipSetDevice(o->device)); o->want_batch = B; return ensure(o, w, h, B); }

int sind_orb_extract_batch(sind_orb* o, const uint8_t* gray, int w, int h, int B, const uint8_t* masks, sind_keypoint* kps, int cap, int* n, uint8_t* desc) {
    if (!o || !n) { sind_set_error("sind_orb_extract_batch: null argument"); return SIND_E_ARG; }
    if (!gray || w <= 0 || h <= 0 || B < 1) { for (int b = 0; n && b < std::max(B, 0); b++) n[b] = 0; return SIND_OK; }   // empty image: silent return
    HIP_TRY(hipSetDevice(o->device));
    SIND_TRY(ensure(o, w, h, B));
    const size_t fb = (size_t)w * h;
    SIND_TRY(o->gray.alloc(fb * B));
    HIP_TRY(hipMemcpyAsync(o->gray.p, gray, fb * B, hipMemcpyHostToDevice, o->stream));
    SIND_TRY(o->eng.extract_all(o->gray.p, B, o->last));
    for (int b = 0; b < B; b++) {
        std::vector<sind::OrbKeyPoint> k; std::vector<uint8_t> d;
        o->eng.finish(o->last[b], masks ? masks + fb * b : nullptr, w, k, d);
        n[b] = (int)k.size();
        if ((int)k.size() > cap) { sind_set_error("sind_orb_extract: %zu keypoints exceed cap %d", k.size(), cap); return SIND_E_CAPACITY; }
        if (kps) std::memcpy(kps + (size_t)b * cap, k.data(), k.size() * sizeof(sind_keypoint));
        if (desc) std::memcpy(desc + (size_t)b * cap * 32, d.data(), d.size());
    }
    return SIND_OK;
}
int sind_orb_extract(sind_orb* o, const uint8_t* gray, int w, int h, int stride, const uint8_t* mask, int mask_stride, sind_keypoint* kps, int cap, int* n, uint8_t* desc) {
    if (!o || !n) { sind_set_error("sind_orb_extract: null argument"); return SIND_E_ARG; }
    if (!gray || w <= 0 || h <= 0) { *n = 0; return SIND_OK; }
    std::vector<uint8_t> g, m;
    if (stride != w) { g.resize((size_t)w * h); for (int y = 0; y < h; y++) std::memcpy(&g[(size_t)y * w], gray + (size_t)y * stride, w); gray = g.data(); }
    if (mask && mask_stride != w) { m.resize((size_t)w * h); for (int y = 0; y < h; y++) std::memcpy(&m[(size_t)y * w], mask + (size_t)y * mask_stride, w); mask = m.data(); }
    return sind_orb_extract_batch(o, gray, w, h, 1, mask, kps, cap, n, desc);
}
int sind_orb_tables(sind_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* per_level, int* umax16) {
    if (!o) return SIND_E_ARG;
    HIP_TRY(hipSetDevice(o->device));
    if (!o->ready) SIND_TRY(ensure(o, 640, 480, 1));
    const sind::OrbEngine& e = o->eng;
    for (int i = 0; i < e.nlevels; i++) {
        if (scale) scale[i] = e.mvScaleFactor[i]; if (inv_scale) inv_scale[i] = e.mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = e.mvLevelSigma2[i]; if (inv_sigma2) inv_sigma2[i] = e.mvInvLevelSigma2[i];
        if (per_level) per_level[i] = e.mnFeaturesPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e.umax[i];
    return SIND_OK;
}
int sind_orb_pyramid(sind_orb* o, int frame, int level, uint8_t* out, int* w, int* h) {
    if (!o || !o->ready || level < 0 || level >= o->eng.nlevels || frame < 0 || frame >= o->eng.maxB) { sind_set_error("sind_orb_pyramid: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(o->device));
    const sind::OrbLevel& L = o->eng.levels[level];
    if (w) *w = L.w; if (h) *h = L.h;
    if (out) HIP_TRY(hipMemcpy(out, o->eng.slab_dev() + (size_t)frame * o->eng.slab_bytes + L.off, (size_t)(L.w + 2 * ORB_PAD) * (L.h + 2 * ORB_PAD), hipMemcpyDeviceToHost));
    return SIND_OK;
}
int sind_orb_debug_fast(sind_orb* o, int frame, int level, float* xyr, int cap) {
    if (!o || frame < 0 || frame >= (int)o->eng.dbg_fast.size() || level < 0 || level >= o->eng.nlevels) return SIND_E_ARG;
    const auto& v = o->eng.dbg_fast[frame][level];
    for (int i = 0; i < (int)v.size() && i < cap; i++) { xyr[3 * i] = v[i].x; xyr[3 * i + 1] = v[i].y; xyr[3 * i + 2] = v[i].response; }
    return (int)v.size();
}
int sind_orb_debug_selected(sind_orb* o, int frame, sind_keypoint* kps, int cap, uint8_t* desc) {
    if (!o || frame < 0 || frame >= (int)o->last.size()) return SIND_E_ARG;
    const sind::OrbFrameResult& R = o->last[frame];
    const int n = std::min((int)R.kps.size(), cap);
    if (kps) std::memcpy(kps, R.kps.data(), (size_t)n * sizeof(sind_keypoint));
    if (desc) std::memcpy(desc, R.desc.data(), (size_t)n * 32);
    return (int)R.kps.size();
}

// PNG scanline reconstruction (filters None / Sub / Up / Average / Paeth, PNG spec section 9) for the rgbd_tum_noros-shaped harness
// (sindslam_amd/harness.py inflates with zlib and calls this for the byte-serial part).  raw: h rows of (1 + stride) bytes.
int sind_png_unfilter(const uint8_t* raw, int h, int stride, int bpp, uint8_t* out) {
    if (!raw || !out || h < 1 || stride < 1 || bpp < 1) return SIND_E_ARG;
    for (int y = 0; y < h; y++) {
        const uint8_t* r = raw + (size_t)y * (stride + 1); const int ft = r[0]; r++;
        uint8_t* o = out + (size_t)y * stride; const uint8_t* up = y ? o - stride : nullptr;
        for (int x = 0; x < stride; x++) {
            const int a = x >= bpp ? o[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: sind_set_error("sind_png_unfilter: bad filter type %d in row %d", ft, y); return SIND_E_ARG;
            }
            o[x] = (uint8_t)(r[x] + pred);
        }
    }
    return SIND_OK;
}

}  // extern "C"
