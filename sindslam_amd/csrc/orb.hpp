// ORB extractor engine interface (host side of orb_kernels.hip).
#pragma once
#include "common.hpp"
#include "flow.hpp"

namespace sind {

#define ORB_PAD 19            /* EDGE_THRESHOLD, reference ORBextractor.cc:74 */
#define ORB_WIN_MAX 66        /* largest FAST cell window side: a cell is ceil(extent / int(extent / 30)) <= 59 px (ORBextractor.cc:789-807), + 6 px overlap; 640 x 480 needs 40 */
/* keypoints a FAST cell can emit: the 3x3 strict-maximum NMS keeps at most one corner per 2 x 2 block of the window's interior (the window minus its 3-px margin), i.e.
   ceil((vw - 6) / 2) * ceil((vh - 6) / 2) -- 324 for the 42 x 42 windows of 640 x 480, 900 for the 66 x 66 limit.  OrbEngine sizes its per-cell records for the largest
   cell of its levels (cell_cap), so that no image can overflow one. */
static inline int orb_cell_bound(int vw, int vh) { return ((vw - 6 + 1) / 2) * ((vh - 6 + 1) / 2); }

struct OrbLevel { int w, h; size_t off, blur_off; };                  // padded level inside the slab; interior inside the blur buffer
struct OrbCell { size_t level_off; int pitch, x0, y0, vw, vh, shift_x, shift_y, level; };   // FAST window in padded coordinates
struct OrbRawKp { short x, y, score, level; };                         // cell-wise FAST output, coordinates relative to minBorder
struct OrbSelKp { float x, y; int level; };                            // octree survivors, level (ROI) coordinates
struct OrbKeyPoint { float x, y, size, angle, response; int octave, class_id; };   // cv::KeyPoint fields

int orb_upload_constants(const int umax[16]);
int launch_pad(hipStream_t s, uint8_t* slab, size_t slab_stride, size_t off, int lw, int lh, int B);
int launch_copy_into_slab(hipStream_t s, const uint8_t* gray, uint8_t* slab, size_t slab_stride, size_t off, int w, int h, int B);
int launch_fast_cells(hipStream_t s, const uint8_t* slab, size_t slab_stride, const OrbCell* cells, int ncells, int cell_cap, int iniTh, int minTh,
                      OrbRawKp* raw, int* counts, OrbRawKp* dense, int cap, int* frame_total, int* cell_offsets, int B);
int launch_ic_angle(hipStream_t s, const uint8_t* slab, size_t slab_stride, const OrbLevel* levels, const OrbSelKp* sel, const int* nsel,
                    int cap, int max_n, float* angle, int B);
int launch_blur7(hipStream_t s, const uint8_t* slab, size_t slab_stride, size_t off, int lw, int lh, const int taps[7], uint16_t* tmp,
                 size_t tmp_stride, size_t tmp_off, uint8_t* blurred, size_t bl_stride, size_t bl_off, int B);
int launch_brief(hipStream_t s, const uint8_t* blurred, size_t bl_stride, const OrbLevel* levels, const OrbSelKp* sel, const int* nsel, int cap,
                 int max_n, const float* angle, uint8_t* desc, int B);
// launch_resize_u8: declared in flow.hpp (the pyramid and the flow front share it)

// host: quadtree distribution of one level's keypoints (reference ORBextractor.cc:481-763)
struct OctKp { float x, y, response; };
void distribute_octree(const std::vector<OctKp>& in, int minX, int maxX, int minY, int maxY, int N, std::vector<OctKp>& out);

// Result of the state-free part for one frame: octree-selected keypoints of all levels with angle + descriptor.
struct OrbFrameResult {
    std::vector<OrbKeyPoint> kps;     // level coordinates (pt not yet multiplied by the level scale), octave = level
    std::vector<uint8_t> desc;        // 32 bytes per keypoint
};

class OrbEngine {
public:
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, W = 0, H = 0, maxB = 0;
    double scaleFactor = 1.2;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<OrbLevel> levels;
    std::vector<OrbCell> cells;
    std::vector<int> level_cell_begin;   // first cell index of each level (+ sentinel)
    size_t slab_bytes = 0, blur_bytes = 0;
    int dense_cap = 0, sel_cap = 0, cell_cap = 0;       // cell_cap: records per FAST cell = the NMS bound of the largest cell (orb_cell_bound)
    hipStream_t stream = nullptr;
    int init(int W, int H, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, int maxB, hipStream_t s);
    // gray: device u8 [B][H][W].  Runs pyramid + FAST + octree (host) + orientation + blur + BRIEF.  Synchronises the stream.
    int extract_all(const uint8_t* gray_dev, int B, std::vector<OrbFrameResult>& out);
    // reference operator() tail (ORBextractor.cc:1063-1163): dynamic-mask erasure, <250 fallback, level scaling.
    void finish(const OrbFrameResult& all, const uint8_t* mask_or_null, int mask_stride, std::vector<OrbKeyPoint>& kps, std::vector<uint8_t>& desc, int* fallback = nullptr) const;
    const uint8_t* slab_dev() const { return slab.p; }
    // debug access for stage-level parity tests (valid after extract_all)
    std::vector<std::vector<std::vector<OctKp>>> dbg_fast;   // [frame][level] cell-wise FAST keypoints
private:
    DevBuf<uint8_t> slab, blurred, desc_dev;
    DevBuf<uint16_t> blur_tmp;
    DevBuf<OrbCell> cells_dev; DevBuf<OrbLevel> levels_dev;
    DevBuf<OrbRawKp> raw, dense; DevBuf<int> counts, frame_total, cell_offsets, nsel_dev;
    DevBuf<OrbSelKp> sel_dev; DevBuf<float> angle_dev;
    int taps[7];
};

}  // namespace sind
