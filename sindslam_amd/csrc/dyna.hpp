// DynaDetect engine interface: state-free GPU front (gray, 0.6 resize, dense flow, up-scale) and the stateful
// per-stream tail (k-means, depth edges, PEAC, split/merge re-clustering, residual thresholds, mask fusion).
#pragma once
#include <memory>
#include <vector>
#include "common.hpp"
#include "depth.hpp"
#include "flow.hpp"
#include "peac_grow.hpp"
#include "host/host.hpp"

namespace sind {

struct DynaConfig { int W = 640, H = 480; float fx = 0, fy = 0, cx = 0, cy = 0, depthScale = 5000.f; int device = 0; };

// ---- state-free front for B frames at once (reference DynaDetect.cc:1386-1392 gray, :1033-1147 flow) ------------------
class DynaFront {
public:
    DynaConfig cfg; int fw = 0, fh = 0, maxB = 0; hipStream_t stream = nullptr;
    FlowEngine flow;
    // speculate: flow(n, n-1) -- the reference's second DeepFlow pass for large-motion pairs (DD:1121-1131) -- is solved TOGETHER with flow(n, n-2), as the second half of
    // one batch of 2 B pairs, and the large-motion test then only picks between the two results.  For a batch that leaves most of the chip idle (one camera) the second
    // half costs next to nothing and a large-motion frame no longer takes two flow latencies in a row.  Same bits: a pair's flow does not depend on its batch.  Needs 2 B <= maxB.
    bool speculate = false;
    int init(const DynaConfig& c, int maxB, hipStream_t s);
    // bgr: device u8 [n][H][W][3] -> gray [n][H][W] and grayMin [n][fh][fw] (both device, caller-owned)
    int gray_and_min(const uint8_t* bgr, int n, uint8_t* gray, uint8_t* grayMin);
    // Dense flow for B pairs.  pool: device u8 frames [*][fh][fw]; cur/prev1/prev2: HOST index arrays into the pool
    // (frame n, n-1, n-2 of each pair).  U/V: device f32 [B][H*W] full-resolution flow (negated, refined, *1/0.6).
    // large_motion (host, optional): per-pair flag of the reference's second DeepFlow pass.
    int dense_flow(const uint8_t* pool, const int* cur, const int* prev1, const int* prev2, int B, float* U, float* V, int* large_motion,
                   float* dbg_deep_u = nullptr, float* dbg_deep_v = nullptr, float* dbg_ref_u = nullptr, float* dbg_ref_v = nullptr);
private:
    DevBuf<uint8_t> g0, g1; DevBuf<float> u, v, mag, u2, v2; DevBuf<unsigned> maxbits; DevBuf<int> hist, idx_dev;
    int gather(const uint8_t* pool, const int* idx_host, int B, uint8_t* out);
};

struct DynaDebug {           // stage outputs of the last tail call (parity tests)
    double H[9] = {0}; int nPairs = 0, hist[256] = {0}; float maxError = 0, otsu = 0, triangle = 0, thr_low = 0, thr_high = 0;
    std::vector<uint8_t> maskLow, maskHigh, kmeansLabel, occ1, occ2, totalArea, gradEdge, planeContours;
    float centers[KM_K][3] = {{0}}; int nClusters = 0;
};

struct OccResult { BitImg totalArea, occ1, occ2; bool ready = false; const float* gridFlow = nullptr;
                   uint8_t* occ2_dev = nullptr; uint8_t* depthN_dev = nullptr; hipEvent_t occ2_event = nullptr; };          // optional device slots the producer fills: plane-edge mask and 8-bit normalised depth for the RAG statistics     // state-free per-frame results computed ahead of the tail: CalOccluded outputs, flow at the 10-px sample grid (host)

// state of a frame between the two host halves of CalOccluded (the PEAC region grow runs on the GPU in between)
struct OccCtx { BitImg occ, totalArea; std::vector<PtI> endPoints; std::unique_ptr<PeacFitter> fit; bool grown_on_host = false;
                std::vector<int8_t> m8; std::vector<int16_t> m16; std::vector<uint8_t> pairs; };

// ---- GPU half of CalOccluded for a chunk of frames at once (state free: depth only).  Host-side results of one frame: OccGpuOut.
struct OccGpuOut { const uint8_t* edge; const uint8_t* total; const PeacBlockStats* blocks;
                   uint8_t* occ2_stage = nullptr; hipEvent_t occ2_event = nullptr; };      // optional: page-locked N bytes for the occ2 upload and the event recorded behind it
struct OccBatch {
    DynaConfig cfg; int cap = 0;
    DevBuf<uint16_t> filt; DevBuf<uint8_t> edge, edgeTmp, total; DevBuf<unsigned> umax; DevBuf<PeacBlockStats> blocks;
    int init(const DynaConfig& c, int chunk);
    // frames at depth_dev + b * W * H; outputs (host, page-locked): edge / valid-area masks and block statistics per frame; depthN_dev (device, optional)
    int run(hipStream_t s, const uint16_t* depth_dev, int B, uint8_t* depthN_dev, uint8_t* edge_h, uint8_t* total_h, PeacBlockStats* blocks_h);
};

// ---- stateful tail of one stream (reference DynaDetect.cc:1377-1666 minus the dense flow) ---------------------------------
// Output of the flow-independent half of a frame (k-means on the depth, CalOccluded, SegAndMerge): everything the flow-dependent
// half (flow masks, fusion) needs from it.  The pipeline computes it for all frames of a step while the dense flow is on the GPU.
struct DepthStageOut { std::vector<uint8_t> label3; int maxNum = 0; BitImg totalArea; bool ready = false; };

// ---- cv::kmeans of SegByKmeans (DD:315-420) for one frame of EVERY stream at once: the streams are independent, their per-frame chains of ~70
// small kernels are not worth a launch each (measured: 55 k launches per 512-pair step, the tails phase bound by their summed durations at a
// concurrency of ~2) -- one batched chain with blockIdx = (work, frame) does the same arithmetic with 1 / B of the launches.
struct KmFrameResult { const uint8_t* label8; float centers[KM_K][3]; int counts[KM_K]; };
class KMeansBatch {
public:
    int init(const DynaConfig& c, int maxB, hipStream_t s);
    // frame b: device depth at depth_base + b * depth_stride; prev[b] = the stream's previous merged labels (host u8 [H*W]) or nullptr for the
    // 3 x 4 grid start.  Blocks until the results are on the host; they stay valid until the next call.
    int run(const uint16_t* depth_base, size_t depth_stride, int B, const uint8_t* const* prev);
    const KmFrameResult& result(int b) const { return res[b]; }
    hipStream_t stream = nullptr;
    KmFuse km_fuse = g_km_fuse_default;      // copied when the object is made; never changes afterwards
private:
    DynaConfig cfg; int W = 0, H = 0, N = 0, maxB = 0;
    DevBuf<uint16_t> dpyr[4]; DevBuf<float> px, py, pz, comp; DevBuf<int> lab[4], seg, use_prev_d; DevBuf<uint8_t> labPrev8, lab8; DevBuf<KmState> kstate;
    PinnedBuf<uint8_t> h_prev, h_lab8; PinnedBuf<KmState> h_state; PinnedBuf<int> h_use_prev;
    std::vector<KmFrameResult> res;
};

class DynaTail {
public:
    DynaConfig cfg; hipStream_t stream = nullptr; DynaDebug dbg; bool keep_debug = false;
    KmFuse km_fuse = g_km_fuse_default;      // copied when the object is made; never changes afterwards
    int piece_threads = 1;        // host threads for the per-cluster piece extraction of SegAndMerge (> 1 only when host cores idle: a single stream in the in-order mode)
    int init(const DynaConfig& c, hipStream_t s);
    ~DynaTail() { for (auto& g : kmGraph) if (g) (void)hipGraphExecDestroy(g); }
    // depth_host: H x W u16 (host); depth_dev: same on the device; U/V: device full-resolution flow of this frame.
    // dyna_out / label_out: host H x W u8 (0 invalid / 125 static / 255 dynamic ; 0 invalid, 1..n clusters).
    // depth_half: the tail object that carries the depth half's state and workspaces (nullptr = this one)
    int process(const uint16_t* depth_host, const uint16_t* depth_dev, const float* U, const float* V, uint8_t* dyna_out, uint8_t* label_out,
                const OccResult* precomputed = nullptr, DynaTail* depth_half = nullptr, const KmFrameResult* km = nullptr);
    // the k-means warm labels of the next frame (nullptr before the first frame of a stream): what KMeansBatch::run wants as prev[b]
    const uint8_t* prev_km_labels() const { return kmLabelLastAny ? kmLabelLast.data() : nullptr; }
    // process() = depth_stage() + flow_stage().  The two halves keep separate state (the k-means warm labels belong to the depth half,
    // the sample weights / previous masks to the flow half), so the depth half of the next frames may run ahead of the flow half.
    // km: this frame's k-means result when it was computed by the batched chain (else the stage runs its own)
    int depth_stage(const uint16_t* depth_host, const uint16_t* depth_dev, const OccResult* precomputed, DepthStageOut& out, const KmFrameResult* km = nullptr);
    int flow_stage(const float* U, const float* V, const DepthStageOut& d, uint8_t* dyna_out, uint8_t* label_out, const float* gridFlowPre = nullptr);
    // CalOccluded (DD:429-642) depends on the depth frame only: the pipeline runs it on this tail's stream while the dense flow
    // of the step is still on the GPU and the host cores are idle
    int compute_occluded(const uint16_t* depth_host, const uint16_t* depth_dev, OccResult& out, const OccGpuOut* pre = nullptr);
    // the same in two halves around a batched launch of the PEAC region grow (peac_grow.hpp): grow_block = the frame's page-locked input block
    int compute_occluded_p1(const uint16_t* depth_host, const uint16_t* depth_dev, OccCtx& c, const OccGpuOut* pre, uint8_t* grow_block, int depth_index);
    int compute_occluded_p2(OccCtx& c, const int8_t* member8, const uint8_t* pair_seen, const int* grow_status, OccResult& out, const OccGpuOut* pre);
    long n_grow_fallback = 0;      // frames whose region grow overflowed a capacity of the kernel and ran on the host
    void reset();
    // Inter-frame state (reference DynaDetect.h:172-178, rolled at DynaDetect.cc:1660-1664) as one flat blob, so that a sequence can
    // continue on another handle / rank exactly where this one stopped ("phase B strictly in frame order"):
    //   [dynaLast N][labelLast N][highLast N (0/255)][lastCnt 256 x i32][lastDyn 256 x i32]   flow half (sample weights, previous high mask)
    //   [kmLabelLast N][kmLabelLastAny i32]                                                  depth half (k-means warm labels)
    size_t state_bytes() const { return (size_t)4 * N + 2 * 256 * sizeof(int) + sizeof(int); }
    void save_state(uint8_t* buf, bool flow_half = true, bool depth_half = true) const;
    void load_state(const uint8_t* buf, bool flow_half = true, bool depth_half = true);
    // hash_state: leave a 128-bit fingerprint of the rolled state (host/statehash.hpp) in state_hash after every frame -- what the chunked sequence mode
    // compares at the chunk seams (sind_pipe_set_state_hashing).  It covers everything the blob carries: the blob's other fields are functions of these images.
    bool hash_state = false; uint64_t state_hash[2] = {0, 0};
    double t_stage[6] = {0, 0, 0, 0, 0, 0}; long n_frames = 0;
    double t_fine[40] = {0};       // cal_occluded: gpu+d2h, pack, endpoints, peac, contour filter, close | seg_merge: pieces, planes+h2d, rag gpu, merge    // flow masks, k-means, label prep, CalOccluded, SegAndMerge, fusion (ms, SIND_TAIL_TIMING=1)
private:
    int W = 0, H = 0, N = 0, zInvalidFrom = 65536; float invDepthScale = 0.f;
    std::vector<uint8_t> dynaLast, labelLast; BitImg highLast;       // host state images (DynaDetect.h:172-178) as seen by the flow half
    int lastCnt[256] = {0}, lastDyn[256] = {0};
    std::vector<uint8_t> kmLabelLast; bool kmLabelLastAny = false;   // imgLabelLast as seen by the depth half (k-means warm labels, DD:374-395)
    // device workspaces
    DevBuf<uint16_t> dpyr[4], filt; DevBuf<float> px, py, pz; DevBuf<int> lab[4]; DevBuf<uint8_t> lab8, labPrev8, edge, edgeTmp, total, depthN, occ2_d, magu8, low_d;
    DevBuf<int> kpart; DevBuf<float> kcomp; DevBuf<unsigned long long> planes_d; DevBuf<unsigned> umax_d; DevBuf<int> hist_d, rag_d; DevBuf<float> mag, grid_d;
    DevBuf<PeacBlockStats> blocks_d;
    // page-locked staging of everything that crosses PCIe in a tail
    PinnedBuf<float> h_grid; PinnedBuf<int> h_hist, h_rag; PinnedBuf<uint8_t> h_ab, h_lab8; PinnedBuf<KmState> h_kstate; PinnedBuf<PeacBlockStats> h_blocks;
    PinnedBuf<unsigned long long> h_planes;
    int flow_masks(const float* U, const float* V, BitImg& low, BitImg& high, const float* gridFlowPre = nullptr);
    int fuse(const BitImg& maskLow, const BitImg& maskHigh, const DepthStageOut& d, uint8_t* dyna_out, uint8_t* label_out);
    int kmeans(const uint16_t* depth_dev, std::vector<uint8_t>& label8, float centers[KM_K][3], int counts[KM_K]);
    bool hist_clean = false;      // hist_d's working block is known to be zero (left so by k_flow_thresholds)
    DevBuf<KmState> kstate; DevBuf<uint16_t> depth_fix; hipGraphExec_t kmGraph[2] = {nullptr, nullptr}; bool kmGraphBroken = false;
    int kmeans_enqueue(const uint16_t* depth0, bool prevLabels);
    int cal_occluded(const uint16_t* depth_host, const uint16_t* depth_dev, BitImg& totalArea, BitImg& occ1, BitImg& occ2, const OccGpuOut* pre = nullptr);
    int cal_occluded_p1(const uint16_t* depth_host, const uint16_t* depth_dev, OccCtx& c, const OccGpuOut* pre, uint8_t* grow_block, int depth_index);
    int cal_occluded_p2(OccCtx& c, const int8_t* member8, const uint8_t* pair_seen, const int* grow_status, BitImg& occ1, BitImg& occ2);
    int finish_occluded(OccResult& out, const OccGpuOut* pre);
    struct OwnGrow { PeacGrowBatch batch; PinnedBuf<uint8_t> in_h, pair_h; PinnedBuf<int8_t> member_h; PinnedBuf<int> status_h; };
    std::unique_ptr<OwnGrow> own_grow;         // one-frame grow workspace, created on first use (single-frame API, unbatched pipeline steps)
    int seg_and_merge(const std::vector<BitImg>& allLabels, const BitImg& occ1, const BitImg& occ2, const BitImg& labelForSegEdge,
                      const uint16_t* depth_host, const uint16_t* depth_dev, std::vector<uint8_t>& labelNew, const OccResult* pre = nullptr);
};

}  // namespace sind
