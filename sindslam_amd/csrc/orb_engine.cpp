// Host orchestration of the ORB extractor (reference ORB_SLAM2/src/ORBextractor.cc): constructor tables (:410-470),
// pyramid geometry (:1166-1191), cell grid (:765-829), octree on the host (:539-763), then orientation, blur
// and BRIEF on the GPU for the survivors; mask erasure / fallback / scaling (:1063-1163) in finish().
#include <cmath>
#include <cstring>
#include <thread>
#include <pthread.h>
#include "orb.hpp"

namespace sind {

static inline int cvRoundD(double v) { return (int)std::lrint(v); }
static inline int cvRoundF(float v) { return (int)std::lrintf(v); }

int OrbEngine::init(int W_, int H_, int nf, float sf, int nl, int ini, int mn, int maxB_, hipStream_t s) {
    W = W_; H = H_; nfeatures = nf; scaleFactor = sf; nlevels = nl; iniTh = ini; minTh = mn; maxB = maxB_; stream = s;
    if (nl < 1 || nl > 16 || W < 64 || H < 64) { sind_set_error("OrbEngine: unsupported geometry"); return SIND_E_ARG; }
    mvScaleFactor.assign(nl, 1.f); mvLevelSigma2.assign(nl, 1.f); mvInvScaleFactor.assign(nl, 1.f); mvInvLevelSigma2.assign(nl, 1.f);
    for (int i = 1; i < nl; i++) { mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor); mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
    for (int i = 0; i < nl; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    mnFeaturesPerLevel.assign(nl, 0);
    const float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) { mnFeaturesPerLevel[l] = cvRoundD(nDesired); sum += mnFeaturesPerLevel[l]; nDesired *= factor; }
    mnFeaturesPerLevel[nl - 1] = std::max(nfeatures - sum, 0);
    // circular patch row extents
    const int HP = 15; umax.assign(HP + 1, 0);
    const int vmax = (int)std::floor(HP * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(HP * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt((double)HP * HP - v * v));
    for (int v = HP, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
    SIND_TRY(orb_upload_constants(umax.data()));
    // 7-tap sigma-2 Gaussian in 8.8 fixed point (OpenCV 4.2.0 fixed-point GaussianBlur for CV_8U)
    { double k[7], ksum = 0; for (int i = 0; i < 7; i++) { double x = i - 3; k[i] = std::exp(-0.5 * x * x / 4.0); ksum += k[i]; }
      for (int i = 0; i < 7; i++) taps[i] = cvRoundD(k[i] / ksum * 256.0); }
    // level geometry
    levels.clear(); slab_bytes = 0; blur_bytes = 0;
    for (int l = 0; l < nl; l++) {
        OrbLevel L; L.w = cvRoundF((float)W * mvInvScaleFactor[l]); L.h = cvRoundF((float)H * mvInvScaleFactor[l]);
        L.off = slab_bytes; L.blur_off = blur_bytes;
        slab_bytes += (size_t)(L.w + 2 * ORB_PAD) * (L.h + 2 * ORB_PAD); blur_bytes += (size_t)L.w * L.h;
        levels.push_back(L);
    }
    slab_bytes = (slab_bytes + 255) & ~(size_t)255; blur_bytes = (blur_bytes + 255) & ~(size_t)255;
    // FAST cell grid per level
    cells.clear(); level_cell_begin.clear();
    for (int l = 0; l < nl; l++) {
        level_cell_begin.push_back((int)cells.size());
        const OrbLevel& L = levels[l];
        const int minBX = ORB_PAD - 3, minBY = minBX, maxBX = L.w - ORB_PAD + 3, maxBY = L.h - ORB_PAD + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY), Wc = 30;
        const int nCols = (int)(width / Wc), nRows = (int)(height / Wc);
        if (nCols < 1 || nRows < 1) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBY + i * hCell); float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBX + j * wCell); float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                OrbCell c; c.level_off = L.off; c.pitch = L.w + 2 * ORB_PAD; c.x0 = (int)iniX + ORB_PAD; c.y0 = (int)iniY + ORB_PAD;
                c.vw = (int)maxX - (int)iniX; c.vh = (int)maxY - (int)iniY; c.shift_x = j * wCell; c.shift_y = i * hCell; c.level = l;
                if (c.vw > ORB_WIN_MAX || c.vh > ORB_WIN_MAX) { sind_set_error("OrbEngine: FAST cell %dx%d exceeds the %d px window", c.vw, c.vh, ORB_WIN_MAX); return SIND_E_ARG; }
                if (c.vw < 7 || c.vh < 7) continue;     // cv::FAST finds nothing in windows without an interior
                cells.push_back(c);
            }
        }
    }
    level_cell_begin.push_back((int)cells.size());
    const int nc = (int)cells.size();
    cell_cap = 64; for (const OrbCell& c : cells) cell_cap = std::max(cell_cap, orb_cell_bound(c.vw, c.vh));
    cell_cap = (cell_cap + 3) / 4 * 4;      // what the NMS can leave in the largest cell: no image overflows a cell
    dense_cap = nc * cell_cap;              // tight upper bound (every cell full): the dense list can never overflow
    sel_cap = nfeatures * 2 + 256;
    SIND_TRY(slab.alloc(slab_bytes * maxB)); SIND_TRY(blurred.alloc(blur_bytes * maxB)); SIND_TRY(blur_tmp.alloc(blur_bytes * maxB));
    SIND_TRY(cells_dev.alloc(nc)); SIND_TRY(levels_dev.alloc(nl));
    HIP_TRY(hipMemcpy(cells_dev.p, cells.data(), nc * sizeof(OrbCell), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(levels_dev.p, levels.data(), nl * sizeof(OrbLevel), hipMemcpyHostToDevice));
    SIND_TRY(raw.alloc((size_t)maxB * nc * cell_cap)); SIND_TRY(dense.alloc((size_t)maxB * dense_cap));
    SIND_TRY(counts.alloc((size_t)maxB * nc)); SIND_TRY(cell_offsets.alloc((size_t)maxB * nc)); SIND_TRY(frame_total.alloc(maxB)); SIND_TRY(nsel_dev.alloc(maxB));
    SIND_TRY(sel_dev.alloc((size_t)maxB * sel_cap)); SIND_TRY(angle_dev.alloc((size_t)maxB * sel_cap)); SIND_TRY(desc_dev.alloc((size_t)maxB * sel_cap * 32));
    return SIND_OK;
}

int OrbEngine::extract_all(const uint8_t* gray, int B, std::vector<OrbFrameResult>& out) {
    if (B < 1 || B > maxB) { sind_set_error("OrbEngine: batch %d outside [1,%d]", B, maxB); return SIND_E_ARG; }
    const int nc = (int)cells.size();
    // ---- pyramid: level 0 = image + border; level l = resize(level l-1 interior) + border
    SIND_TRY(launch_copy_into_slab(stream, gray, slab.p, slab_bytes, levels[0].off, W, H, B));
    SIND_TRY(launch_pad(stream, slab.p, slab_bytes, levels[0].off, W, H, B));
    for (int l = 1; l < nlevels; l++) {
        const OrbLevel& P = levels[l - 1]; const OrbLevel& L = levels[l];
        const int ppitch = P.w + 2 * ORB_PAD, lpitch = L.w + 2 * ORB_PAD;
        const uint8_t* src = slab.p + P.off + (size_t)ORB_PAD * ppitch + ORB_PAD;
        uint8_t* dst = slab.p + L.off + (size_t)ORB_PAD * lpitch + ORB_PAD;
        SIND_TRY(launch_resize_u8(stream, src, dst, P.w, P.h, L.w, L.h, B, ppitch, lpitch, slab_bytes, slab_bytes));
        SIND_TRY(launch_pad(stream, slab.p, slab_bytes, L.off, L.w, L.h, B));
    }
    // ---- cell-wise FAST + NMS, compaction
    SIND_TRY(launch_fast_cells(stream, slab.p, slab_bytes, cells_dev.p, nc, cell_cap, iniTh, minTh, raw.p, counts.p, dense.p, dense_cap, frame_total.p, cell_offsets.p, B));
    // ---- blur of every level (needed by BRIEF; independent of the keypoints) overlaps with the host octree below
    for (int l = 0; l < nlevels; l++)
        SIND_TRY(launch_blur7(stream, slab.p, slab_bytes, levels[l].off, levels[l].w, levels[l].h, taps, blur_tmp.p, blur_bytes, levels[l].blur_off, blurred.p, blur_bytes, levels[l].blur_off, B));
    std::vector<int> h_total(B), h_off((size_t)B * nc), h_cnt((size_t)B * nc);
    HIP_TRY(hipMemcpyAsync(h_total.data(), frame_total.p, B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_off.data(), cell_offsets.p, (size_t)B * nc * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_cnt.data(), counts.p, (size_t)B * nc * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    int max_total = 0;
    for (int b = 0; b < B; b++) {
        if (h_total[b] > dense_cap) { sind_set_error("OrbEngine: %d FAST keypoints exceed the dense capacity %d", h_total[b], dense_cap); return SIND_E_CAPACITY; }
        max_total = std::max(max_total, h_total[b]);
    }
    for (size_t i = 0; i < h_cnt.size(); i++) if (h_cnt[i] & 0x40000000) { sind_set_error("OrbEngine: a FAST cell overflowed %d keypoints", cell_cap); return SIND_E_CAPACITY; }
    std::vector<OrbRawKp> h_dense((size_t)B * std::max(max_total, 1));
    if (max_total > 0)
        HIP_TRY(hipMemcpy2DAsync(h_dense.data(), (size_t)max_total * sizeof(OrbRawKp), dense.p, (size_t)dense_cap * sizeof(OrbRawKp),
                                 (size_t)max_total * sizeof(OrbRawKp), B, hipMemcpyDeviceToHost, stream));
    HIP_TRY(sind_stream_wait(stream));
    // ---- host octree per frame and level
    out.assign(B, OrbFrameResult());
    dbg_fast.assign(B, std::vector<std::vector<OctKp>>(nlevels));
    std::vector<OrbSelKp> h_sel((size_t)B * sel_cap); std::vector<int> h_nsel(B, 0);
    std::vector<std::vector<OctKp>> sel_resp(B);
    int max_sel = 0;
    std::vector<int> frame_rc(B, SIND_OK);
    auto octree_frame = [&](int b) {
        const OrbRawKp* D = &h_dense[(size_t)b * std::max(max_total, 1)];
        int n = 0;
        for (int l = 0; l < nlevels; l++) {
            const int c0 = level_cell_begin[l], c1 = level_cell_begin[l + 1];
            const int beg = c0 < nc ? h_off[(size_t)b * nc + c0] : h_total[b];
            const int end = c1 < nc ? h_off[(size_t)b * nc + c1] : h_total[b];
            std::vector<OctKp>& in = dbg_fast[b][l]; in.resize(end - beg);
            for (int i = beg; i < end; i++) in[i - beg] = {(float)D[i].x, (float)D[i].y, (float)D[i].score};
            const OrbLevel& L = levels[l];
            const int minBX = ORB_PAD - 3, minBY = minBX, maxBX = L.w - ORB_PAD + 3, maxBY = L.h - ORB_PAD + 3;
            std::vector<OctKp> sel;
            if (!in.empty()) distribute_octree(in, minBX, maxBX, minBY, maxBY, mnFeaturesPerLevel[l], sel);
            for (const OctKp& k : sel) {
                if (n >= sel_cap) { frame_rc[b] = SIND_E_CAPACITY; return; }
                h_sel[(size_t)b * sel_cap + n] = {k.x + minBX, k.y + minBY, l};
                sel_resp[b].push_back(k); n++;
            }
        }
        h_nsel[b] = n;
    };
    {   // the quadtree of a frame is serial, frames are independent: spread them over host threads
        const int nth = std::max(1, std::min<int>(B, std::min<int>(12, (int)std::thread::hardware_concurrency())));
        std::vector<std::thread> th;
        for (int t = 0; t < nth; t++) th.emplace_back([&, t] { (void)pthread_setname_np(pthread_self(), "sind-octree"); for (int b = t; b < B; b += nth) octree_frame(b); });
        for (auto& t : th) t.join();
    }
    for (int b = 0; b < B; b++) { if (frame_rc[b] != SIND_OK) { sind_set_error("OrbEngine: more than %d selected keypoints", sel_cap); return frame_rc[b]; } max_sel = std::max(max_sel, h_nsel[b]); }
    // ---- orientation + descriptors on the GPU
    HIP_TRY(hipMemcpyAsync(sel_dev.p, h_sel.data(), h_sel.size() * sizeof(OrbSelKp), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(nsel_dev.p, h_nsel.data(), B * sizeof(int), hipMemcpyHostToDevice, stream));
    SIND_TRY(launch_ic_angle(stream, slab.p, slab_bytes, levels_dev.p, sel_dev.p, nsel_dev.p, sel_cap, max_sel, angle_dev.p, B));
    SIND_TRY(launch_brief(stream, blurred.p, blur_bytes, levels_dev.p, sel_dev.p, nsel_dev.p, sel_cap, max_sel, angle_dev.p, desc_dev.p, B));
    std::vector<float> h_angle((size_t)B * sel_cap); std::vector<uint8_t> h_desc((size_t)B * sel_cap * 32);
    if (max_sel > 0) {
        HIP_TRY(hipMemcpy2DAsync(h_angle.data(), (size_t)sel_cap * 4, angle_dev.p, (size_t)sel_cap * 4, (size_t)max_sel * 4, B, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpy2DAsync(h_desc.data(), (size_t)sel_cap * 32, desc_dev.p, (size_t)sel_cap * 32, (size_t)max_sel * 32, B, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(sind_stream_wait(stream));
    HIP_TRY(hipGetLastError());
    for (int b = 0; b < B; b++) {
        OrbFrameResult& R = out[b]; R.kps.resize(h_nsel[b]); R.desc.resize((size_t)h_nsel[b] * 32);
        for (int i = 0; i < h_nsel[b]; i++) {
            const OrbSelKp& s = h_sel[(size_t)b * sel_cap + i];
            OrbKeyPoint& k = R.kps[i];
            k.x = s.x; k.y = s.y; k.size = (float)(int)(31 * mvScaleFactor[s.level]); k.angle = h_angle[(size_t)b * sel_cap + i];
            k.response = sel_resp[b][i].response; k.octave = s.level; k.class_id = -1;
        }
        if (h_nsel[b]) std::memcpy(R.desc.data(), &h_desc[(size_t)b * sel_cap * 32], (size_t)h_nsel[b] * 32);
    }
    return SIND_OK;
}

void OrbEngine::finish(const OrbFrameResult& all, const uint8_t* mask, int mask_stride, std::vector<OrbKeyPoint>& kps, std::vector<uint8_t>& desc, int* fallback) const {
    std::vector<char> keep(all.kps.size(), 1);
    size_t nkeep = all.kps.size();
    if (mask) {
        nkeep = 0;
        for (size_t i = 0; i < all.kps.size(); i++) {
            const OrbKeyPoint& k = all.kps[i];
            const float scale = (float)std::pow(scaleFactor, k.octave);
            const bool dyn = mask[(size_t)(int)(k.y * scale) * mask_stride + (int)(k.x * scale)] == 255;
            keep[i] = !dyn; nkeep += !dyn;
        }
    }
    const bool fb = nkeep < 250;
    if (fallback) *fallback = fb;
    if (fb) std::fill(keep.begin(), keep.end(), 1);
    kps.clear(); desc.clear();
    for (size_t i = 0; i < all.kps.size(); i++) {
        if (!keep[i]) continue;
        OrbKeyPoint k = all.kps[i];
        if (k.octave != 0) { const float s = mvScaleFactor[k.octave]; k.x *= s; k.y *= s; }
        kps.push_back(k);
        desc.insert(desc.end(), all.desc.begin() + i * 32, all.desc.begin() + (i + 1) * 32);
    }
}

}  // namespace sind
