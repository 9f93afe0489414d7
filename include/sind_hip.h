/* libsind_hip — C ABI of the MI355X-native DynaDetect + ORBextractor hot path.
 *
 * The reference (qimao7213/SInDSLAM) has no FFI layer: its boundary is two C++ classes,
 *   ORB_SLAM2::DynaDetect    (ORB_SLAM2/include/DynaDetect.h:95-131, src/DynaDetect.cc:1377-1666)
 *   ORB_SLAM2::ORBextractor  (ORB_SLAM2/include/ORBextractor.h:54-88, src/ORBextractor.cc:1043-1164)
 * called from Examples/RGB-D/rgbd_tum_noros.cc:106-107,135,138 and src/Frame.cc:308.  Each entry point below names
 * the reference member it replaces.  include/DynaDetect.h and include/ORBextractor.h are header-only C++ shims with
 * the reference's class names and signatures on top of this ABI.
 *
 * Conventions: plain pointers and sizes, caller-owned memory, row strides in BYTES, return 0 on success or a negative
 * SIND_E_* code (sind_last_error() gives the text), no exceptions cross the ABI.  A handle is bound to one GPU and
 * one HIP stream and is not thread-safe; use one handle per thread.  "_dev" entry points take DEVICE pointers
 * (inputs already resident in HBM) and are asynchronous on the handle's stream until sind_*_sync().
 */
#ifndef SIND_HIP_H
#define SIND_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SIND_OK 0
#define SIND_E_ARG (-1)
#define SIND_E_HIP (-2)
#define SIND_E_ALLOC (-3)
#define SIND_E_STATE (-4)
#define SIND_E_CAPACITY (-5)

const char* sind_last_error(void);
int sind_device_count(int* count);

/* ------------------------------------------------------------------------------------------------------------
 * Dense flow stage (state free).  Replaces, for B frame pairs at once on the 0.6-scaled grid:
 *   cv::optflow::createOptFlow_DeepFlow()->calc(I_n, I_prev, flow)      DynaDetect.cc:1031,1075,1127
 *   cv::VariationalRefinement::create()->calc(I_n, I_prev, flow)        DynaDetect.cc:1133-1143
 * Images: u8 [B][fh][fw] dense.  Flow: two planes u, v, f32 [B][fh][fw] (the reference's CV_32FC2 de-interleaved).
 */
typedef struct sind_flow sind_flow;
int sind_flow_create(int fw, int fh, int max_batch, int device, sind_flow** out);
int sind_flow_destroy(sind_flow* f);
int sind_flow_levels(sind_flow* f, int* widths, int* heights, int cap);          /* returns the level count */
int sind_flow_deepflow(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v);       /* host pointers */
int sind_flow_refine(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v);         /* host, u/v in-out */
int sind_flow_varref_f32(sind_flow* f, const float* i0, const float* i1, int w, int h, int B, float* u, float* v,
                         int fixed_point_iters, int sor_iters, float alpha, float delta, float gamma, float omega); /* host, one level */
int sind_flow_deepflow_dev(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v);   /* device pointers, async */
int sind_flow_refine_dev(sind_flow* f, const uint8_t* i0, const uint8_t* i1, int B, float* u, float* v);
int sind_flow_sync(sind_flow* f);
/* build-side option (BASELINE.json config 5, "3-level flow pyramid"; no reference counterpart -- OpenCV 4.2's DeepFlow never advances its
 * maxLayers counter): n > 0 keeps only the finest n levels of the 0.95 pyramid, the flow starts from zero at the coarsest of them; 0 = all */
int sind_flow_set_max_levels(sind_flow* f, int n);
/* per handle: on != 0 (default) runs every pyramid level that is one workgroup's work (<= 4096 pixels) -- warp, coefficients, SOR, W += dW and the up-sampling, for all
 * such levels of a DeepFlow pyramid -- in ONE launch (k_coarse_chain); 0 = per-stage kernels on every level (cross-check, A/B timing).  Same bits either way. */
int sind_flow_set_coarse_chain(sind_flow* f, int on);
/* per handle: on != 0 (default) solves a tiled level whose tiles all find a compute unit of their own (few images per launch: the launch is latency-bound) with 1024-thread
 * tiles and up to 13 iterations per launch on deeper halos (k_sor_tile); 0 = the 512-thread tiles / streaming kernel at every batch size.  Same bits either way. */
int sind_flow_set_latency_tiles(sind_flow* f, int on);
/* per handle: on != 0 (default) makes the transition between two pyramid levels -- W += dW, the bilinear up-sampling / 0.95 and the next level's warp, average and temporal
 * difference -- one launch (k_level_up) instead of three; 0 = the three kernels (cross-check).  Same bits either way. */
int sind_flow_set_level_up(sind_flow* f, int on);
/* solver variant of THIS handle (every variant returns the same bits; nothing here is process-wide).  Fused register-resident SOR with 1x8 pixel strips: mode 4 = divisions
 * through a reciprocal formed on the fly (hardware estimate + one Newton step, then Markstein's correction; default: 5 iterations per launch on 64 x 64 tiles), 5 = the
 * streaming kernel on every level it fits, 6 = the one-wave pipeline on every level beyond one workgroup, 0 = one launch per colour (cross-check); lab builds also: 1 = IEEE division, 3 = reciprocals of A11 / A22 read from planes and
 * held in registers (three waves per SIMD; tiles of 256, 384 and 768 threads), 2 = 1x4 strips + reciprocal division.  fuse = iterations per launch on tiled levels
 * (default 5), 0 = a plan per level (lab builds); tile_w x tile_h = extended tile (tile_w * tile_h / 8 threads).  sind_flow_set_sor keeps the round-1 argument list
 * (tile height 48 for mode 3, 64 otherwise). */
int sind_flow_set_sor(sind_flow* f, int mode, int fuse, int tile_w);
int sind_flow_set_sor_tiled(sind_flow* f, int mode, int fuse, int tile_w, int tile_h);
/* streaming solver: at most `cap` workgroups per launch, each taking several (column strip, image) items in turn (persistent workgroups); 0 = one workgroup per item.
 * Same results.  See DESIGN.md 3.1-12 for when it pays. */
int sind_flow_set_solver_workgroups(sind_flow* f, int cap);
/* one-wave row pipelines (k_sor_wave, flow_wave.hip) for the levels and batch sizes that would otherwise go to the streaming kernel: on != 0 (default) / 0 = k_sor_stream;
 * target_items = waves a launch should have (row bands are cut until it does; 0 keeps the default), bands > 0 = exactly that many row bands (tests); on = 2 / 3 / 4 additionally selects 1 / 2 / 3 rows
 * in flight per wave (A/B timing; default 2).  Mode 6 of
 * sind_flow_set_sor_tiled runs the kernel on every level beyond one workgroup at any batch size.  Same bits either way. */
int sind_flow_set_wave_solver(sind_flow* f, int on, int target_items, int bands);
/* how k_sor_wave cuts a w x h level of B pairs (host arithmetic only, no GPU needed): out = {column strips, kept columns per strip, row bands, kept rows per band}.  A strip works on
 * 128 columns -- its kept ones plus 10 on every side that is not an image border --, a band on its kept rows plus 10 on every cut side. */
int sind_flow_wave_layout(int w, int h, int B, int target_items, int bands, int out[4]);
/* coefficient kernel: 1 = k_coef_lanes (neighbours from lanes, short correctly rounded sqrt / quotient forms; default), 2 = k_coef_lanes with the compiler's IEEE forms,
 * 0 = k_coef (neighbours from memory), 3 = variant 1 with its tiles in plain grid order over the XCDs (A/B timing: by default the tiles of a pair share an XCD's L2).  Same results. */
int sind_flow_set_coef_kernel(sind_flow* f, int variant);
int sind_lab_build(void);        /* 1: built with -DSIND_LAB (dormant solver variants and the SIND_* experiment switches of the measurement rounds), 0: the shipped drop-in */
/* HIP-event timing of everything enqueued on the handle's stream between begin and end (bench.py roofline leg) */
int sind_flow_timer_begin(sind_flow* f);
int sind_flow_timer_end(sind_flow* f, float* milliseconds);

/* ------------------------------------------------------------------------------------------------------------
 * ORBextractor.  Replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:54-88):
 *   ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)     src/ORBextractor.cc:410-470  -> sind_orb_create
 *   operator()(image, mask, keypoints, descriptors)                         src/ORBextractor.cc:1043-1164 -> sind_orb_extract
 *   GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / Get(Inverse)ScaleSigmaSquares -> sind_orb_tables
 *   public member mvImagePyramid (read by Frame::ComputeStereoMatches)                     -> sind_orb_pyramid
 * sind_keypoint mirrors cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id).
 * mask: the dilated imgDyna (255 = dynamic) or NULL; "empty image -> silent return" maps to n = 0 with SIND_OK.
 */
typedef struct sind_orb sind_orb;
typedef struct sind_keypoint { float x, y, size, angle, response; int octave, class_id; } sind_keypoint;
int sind_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device, sind_orb** out);
int sind_orb_destroy(sind_orb* o);
int sind_orb_reserve(sind_orb* o, int width, int height, int max_batch);   /* optional: pre-size the workspaces */
int sind_orb_extract(sind_orb* o, const uint8_t* gray, int width, int height, int stride, const uint8_t* mask_or_null, int mask_stride,
                     sind_keypoint* kps, int cap, int* n, uint8_t* desc /* cap x 32 */);
/* B images [B][height][width] dense (host); masks [B][height][width] or NULL; outputs [B][cap], n[B], desc [B][cap][32] */
int sind_orb_extract_batch(sind_orb* o, const uint8_t* gray, int width, int height, int B, const uint8_t* masks_or_null,
                           sind_keypoint* kps, int cap, int* n, uint8_t* desc);
int sind_orb_tables(sind_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* features_per_level, int* umax16);
/* padded pyramid level of frame `frame` of the last call (19-px REFLECT_101 border): copies (w+38)*(h+38) bytes */
int sind_orb_pyramid(sind_orb* o, int frame, int level, uint8_t* out, int* w, int* h);
/* parity-test access to the stage outputs of the last call: cell-wise FAST keypoints of one level (x, y, response triplets,
 * coordinates relative to the 16-px min border) and the octree survivors with their orientation before mask erasure */
int sind_orb_debug_fast(sind_orb* o, int frame, int level, float* xyr, int cap);
int sind_orb_debug_selected(sind_orb* o, int frame, sind_keypoint* kps, int cap, uint8_t* desc);

/* ------------------------------------------------------------------------------------------------------------
 * DynaDetect.  Replaces ORB_SLAM2::DynaDetect (include/DynaDetect.h:95-131):
 *   DynaDetect(imgLast, imgLastLast, fx, fy, cx, cy, depthScale)   DynaDetect.h:98-126         -> sind_dyna_create + sind_dyna_prime
 *   DetectDynaArea(img, imgDepth, imgDyna, imgLabel, nImg)         DynaDetect.cc:1377-1666     -> sind_dyna_detect
 * img: CV_8UC3 BGR, imgDepth: CV_16UC1 (raw units = metres * depthScale), imgDyna: CV_8UC1 0 invalid / 125 static /
 * 255 dynamic, imgLabel: CV_8UC1 0 invalid, 1..n.  The reference's imshow / waitKey / stdout side effects are dropped.
 * Like the reference class the handle carries inter-frame state and is not re-entrant.
 */
typedef struct sind_dyna sind_dyna;
int sind_dyna_create(int width, int height, float fx, float fy, float cx, float cy, float depth_scale, int device, sind_dyna** out);
int sind_dyna_destroy(sind_dyna* d);
int sind_dyna_set_flow_max_levels(sind_dyna* d, int n);      /* see sind_flow_set_max_levels */
int sind_dyna_timing(sind_dyna* d, double ms12[12], int reset); /* mean ms per detect call: upload, dense flow, wait for the depth half, flow masks + fusion, depth half, whole call;
                                                                    then the tail's stages: flow masks, k-means, label preparation, CalOccluded, SegAndMerge, fusion; returns calls */
/* summed milliseconds of the tail's sub-stages since the last reset (dyna.hpp t_fine: CalOccluded 0 gpu + d2h, 1 pack, 2 end points, 3 PEAC host parts, 4 contour filter, 5 close;
 * SegAndMerge 6 pieces, 7 alloc, 8 rag (h2d, kernels, d2h, wait), 9 merge, 10 sort + paint + pack, 11 h2d enqueue, 12 - 16 inside pieces: open, contours, masks, lianjie, centre; 20 - 23 flow masks on
 * the host: weights, sort + wait, homography, pack; 24 - 27 fusion) */
int sind_dyna_timing_fine(sind_dyna* d, double ms40[40], int reset);
int sind_dyna_set_debug(sind_dyna* d, int on);               /* on: keep the intermediate images sind_dyna_debug reports (costs four flow-sized copies per frame); default off */
int sind_dyna_set_overlap(sind_dyna* d, int on);             /* default on: the depth half of a frame (k-means, CalOccluded, SegAndMerge) runs beside its dense flow, as the reference's
                                                                 flow thread runs beside the segmentation (DynaDetect.cc:1396-1398); 0 = one after the other; same results */
int sind_dyna_prime(sind_dyna* d, const uint8_t* bgr_last, const uint8_t* bgr_lastlast, int stride);
int sind_dyna_detect(sind_dyna* d, const uint8_t* bgr, int bgr_stride, const uint16_t* depth, int depth_stride,
                     uint8_t* dyna_out, uint8_t* label_out, int n_img);
/* caller-side 15x15 elliptical dilation of imgDyna before tracking (Examples/RGB-D/rgbd_tum_noros.cc:108,138), in place */
int sind_dyna_dilate15(sind_dyna* d, uint8_t* dyna_inout);
/* parity-test access to the stage outputs of the last sind_dyna_detect call; any pointer may be NULL.
 * flow_*: f32 planes (u then v), deep/refined at the 0.6 grid, full at width x height.  thr = {maxError, otsu, triangle, low, high}.
 * info = {largeMotion, nPairs, nPieces}. */
int sind_dyna_debug(sind_dyna* d, float* flow_deep, float* flow_refined, float* flow_full, double* H9, float* thr5, int* hist256,
                    uint8_t* mask_low, uint8_t* mask_high, uint8_t* kmeans_label, float* centers36, uint8_t* occ1, uint8_t* occ2,
                    uint8_t* total_area, uint8_t* grad_edge, uint8_t* plane_contours, int* info3);

/* parity-test access to the k-means centre sums: out = the FP32 value of  acc = 0; for (i) acc += x[i]  (round to nearest even, exactly cv::kmeans'
 * centre accumulation, kmeans.cpp) computed by the wave-parallel window arithmetic of k_km_seqsum (csrc/depth_kernels.hip) */
int sind_debug_seqsum(const float* x, int n, int device, float* out);
/* k-means pyramid levels of at most n points (default 81920 = the three coarse levels at 640 x 480) of a batch of at least b frames (default 32: the batched rounds of a
 * many-stream step) run every pass in ONE launch, one workgroup per frame (k_km_level_fused); n = 0: the per-pass kernels everywhere.  Same labels and centres bit for bit
 * (tests/test_kmeans_fused_gpu.py).  They set the DEFAULTS that handles created AFTERWARDS copy; an existing handle keeps what it was created with (parity tests, A/B timing). */
int sind_debug_set_kmeans_fused_max(int n);
int sind_debug_set_kmeans_fused_min_batch(int b);
/* exhaustive check of the short forms: for every float significand and the binary exponents exp_lo..exp_hi (>= -96), out[0] = arguments whose short-form square root differs
 * from sqrtf, out[1] = quotients numer[k] / b through the reciprocal (hardware estimate + Newton step + Markstein's correction) that differ from the IEEE division */
int sind_debug_coef_math_scan(int device, int exp_lo, int exp_hi, const float numer[3], unsigned long long out[2]);
/* exhaustive check of the solver's division: for every float significand and the binary exponents exp_lo..exp_hi, out[0] = reciprocals (hardware
 * estimate + one Newton step) that differ from the correctly rounded 1 / a, out[1] = quotients through that reciprocal (Markstein) that differ
 * from the IEEE division (16 numerators per divisor), out[2] = smallest failing significand (all ones if none) */
int sind_debug_rcp_scan(int device, int exp_lo, int exp_hi, unsigned long long out[3]);

/* parity-test access to the residual-threshold kernel (Otsu + Triangle of cv::threshold and the clamping of DynaDetect.cc:1309-1367 from a 256-bin
 * histogram): hist = n blocks of 257 words (counts, then the float bits of the maximal residual), res = n blocks of 261 words (the 257 input words,
 * then lo, hi, otsu, triangle as floats; variant 0 writes only n x 4 floats, packed at the start of res).  variant 0 = serial one-thread reference
 * kernel, 1 = one-wave kernel, 2 = one-wave kernel in its production form, which clears the working histogram: the first 257 words of every result
 * block then hold the cleared words.  mu1 (optional, variants 1 / 2): n x 256 values of Otsu's running class mean, the one rounding chain with a division. */
int sind_debug_flow_thresholds(const int* hist, int n, int width, int height, int variant, int device, int* res, double* mu1);

/* parity-test access to the GPU region grow of the PEAC plane refinement (AHCPlaneFitter.hpp:546-601 floodFill; csrc/peac_kernels.hip): n depth frames
 * (host u16 [n][height][width]) -> membership map per pixel (plane index or -1) from ONE kernel launch over all frames (member_gpu) and from the host
 * statement of the same FIFO (member_host), both int8 [n][height * width]; pair_* [n][127 * 127]: which planes met (row stride = the frame's plane count);
 * status [n][4] = kernel status (0 ok, 1..3 capacity errors, 4 skipped), BFS levels, seeds processed, planes. */
int sind_debug_peac_grow(const uint16_t* depth, int n, int width, int height, float fx, float fy, float cx, float cy, float depth_scale, int device,
                         int8_t* member_gpu, int8_t* member_host, uint8_t* pair_gpu, uint8_t* pair_host, int* status);

/* parity-test access to the bit-plane dilation used by the region-adjacency stage (7x7 ellipse on 64-pixel words, cv::dilate semantics):
 * planes / out are host arrays [nplanes][height][ceil(width / 64)] of 64-bit words, bit i of word k = pixel 64 k + i. */
int sind_debug_dilate_planes(const unsigned long long* planes, int nplanes, int width, int height, int n, int device, unsigned long long* out);

/* ------------------------------------------------------------------------------------------------------------
 * Batched multi-stream pipeline: S independent camera streams x T consecutive frames per step through
 * DynaDetect + 15x15 dilation + ORBextractor, i.e. the body of the frame loop of Examples/RGB-D/rgbd_tum_noros.cc:110-170
 * for S sequences at once.  The state-free stages (gray, resize, dense flow, ORB pyramid/FAST/orientation/BRIEF) are
 * batched over all S*T frames; the stateful tail of each stream runs in frame order on its own host thread + HIP stream.
 * A stream is exactly one reference DynaDetect instance: results equal S sequential single-stream runs.
 * Layouts (dense): bgr [S][T][H][W][3] u8, depth [S][T][H][W] u16, dyna/label/mask [S][T][H][W] u8,
 * kps [S][T][cap], nkp [S][T], desc [S][T][cap][32].  "_dev" takes DEVICE input pointers (frames already in HBM);
 * outputs are always host pointers (they feed the host-side tracker).
 */
typedef struct sind_pipe sind_pipe;
typedef struct sind_pipe_config {
    int width, height; float fx, fy, cx, cy, depth_scale;
    int nfeatures; float scale_factor; int nlevels, ini_th_fast, min_th_fast;
    int orb_gray_rgb_order;      /* 1: ORB gray uses RGB2GRAY on the BGR buffer (Camera.RGB: 1, src/Tracking.cc:246-251), 0: BGR2GRAY */
    int streams, frames_per_step, device;
    int host_threads;            /* 0 = library default (2 x the CPU share of the process) */
    int flow_max_levels;         /* 0 = the reference's full DeepFlow pyramid; n > 0: finest n levels only (see sind_flow_set_max_levels) */
    int flow_slices;             /* dense-flow slices of a step that run concurrently on their own streams: 0 = by step size (default), 1..4 fixed; same results */
    int flow_opts_off;           /* A/B switches, same results: bit 0 = no k_coarse_chain (see sind_flow_set_coarse_chain), bit 1 = no k_sor_tile (see sind_flow_set_latency_tiles), bit 2 = no k_level_up (see sind_flow_set_level_up), bit 3 = k_sor_stream instead of k_sor_wave (see sind_flow_set_wave_solver), bit 4 = a round's k-means waits for the whole tails of the frame before (not only for their depth halves), bits 8.. = target_items of sind_flow_set_wave_solver; 0 = defaults */
} sind_pipe_config;
int sind_pipe_create(const sind_pipe_config* cfg, sind_pipe** out);
int sind_pipe_destroy(sind_pipe* p);
int sind_pipe_prime(sind_pipe* p, int stream, const uint8_t* bgr_last, const uint8_t* bgr_lastlast);          /* host pointers */
int sind_pipe_process(sind_pipe* p, const uint8_t* bgr, const uint16_t* depth, uint8_t* dyna, uint8_t* label, uint8_t* mask_dilated,
                      sind_keypoint* kps, int cap, int* nkp, uint8_t* desc);                                    /* host inputs */
int sind_pipe_process_dev(sind_pipe* p, const uint8_t* bgr_dev, const uint16_t* depth_dev, uint8_t* dyna, uint8_t* label,
                          uint8_t* mask_dilated, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc);          /* device inputs */
/* Software-pipelined form: phase A (GPU batch) of the submitted step overlaps with phase B (per-stream tails) of the previously
 * submitted one.  The output pointers of call i receive the results of step i-1 (*have_output = 0 on the first call);
 * sind_pipe_flush drains the last step.  Device inputs only need to stay valid until the call returns. */
int sind_pipe_submit_dev(sind_pipe* p, const uint8_t* bgr_dev, const uint16_t* depth_dev, uint8_t* dyna, uint8_t* label, uint8_t* mask_dilated,
                         sind_keypoint* kps, int cap, int* nkp, uint8_t* desc, int* have_output);
int sind_pipe_flush(sind_pipe* p, uint8_t* dyna, uint8_t* label, uint8_t* mask_dilated, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc,
                    int* have_output);
/* Schedule of the synchronous step (sind_pipe_process / _process_dev): on != 0 runs the flow-independent half of every tail (depth
 * k-means, SegAndMerge; reference DynaDetect.cc:1410-1551) underneath the dense flow instead of after it.  Results are identical;
 * default off (environment SIND_DEPTH_AHEAD=1 turns it on at create), see DESIGN.md 3.1 item 7 for the measurement.  It also applies to
 * sind_pipe_submit_dev: the depth chain of step i+1 then runs next to the flow chain of step i -- the schedule of the in-order
 * ("exact") single-sequence mode (streams = 1), whose rate is 1 / max(depth-chain, flow-chain latency per frame). */
int sind_pipe_set_depth_ahead(sind_pipe* p, int on);
/* Inter-frame state of one stream (reference DynaDetect.h:172-178: imgDynaLast, imgLabelLast, imgMaskHighErrorLast, plus the k-means
 * warm labels of DynaDetect.cc:374-395; rolled at DynaDetect.cc:1660-1664) as an opaque blob of sind_pipe_state_bytes() bytes.  Lets one
 * long sequence continue on another handle or rank exactly where this one stopped (phase A sharded, phase B strictly in
 * frame order).  get: nothing may be pending.  set: after sind_pipe_prime (which resets the state); allowed while a step submitted
 * WITHOUT depth-ahead waits for its tails -- submit (phase A), receive the predecessor's state, set, flush. */
size_t sind_pipe_state_bytes(sind_pipe* p);
int sind_pipe_get_state(sind_pipe* p, int stream, uint8_t* buf, size_t n);
int sind_pipe_set_state(sind_pipe* p, int stream, const uint8_t* buf, size_t n);
/* Chunked sequences (one long sequence cut into contiguous chunks, one per stream; driver: sindslam_amd/sequence.py).  A chunk that starts in
 * the middle of the sequence rebuilds the inter-frame state in a few warm-up frames; whether the rebuilt state IS the state the sequential loop
 * (rgbd_tum_noros.cc:110-170) would have carried there is decided by comparing fingerprints: every output of a frame is a deterministic function of the
 * input frames and the state before it, so equal states after frame q mean equal results on every later frame.
 *   sind_pipe_set_state_hashing(on): every tail leaves a 128-bit fingerprint of its rolled state (DynaDetect.cc:1660-1664) per frame.
 *   sind_pipe_get_state_hashes: the fingerprints of the step whose results were returned last, [streams][frames_per_step][2] (0, 0 = frame not processed).
 *   sind_pipe_set_active_frames(n[streams]): in the NEXT step only, the stateful tail of stream s runs for its first n[s] frames (0..frames_per_step); the
 *     state of the stream then is the state after frame n[s] - 1 -- a chunk can end, and its state be taken with sind_pipe_get_state, in the middle of a
 *     step.  Outputs of the skipped frames are not written.  The state-free work still covers all frames, and the stream's gray history ends up at the
 *     step's last frame: prime the stream again before it processes anything else.  NULL = all frames.  Not available with depth-ahead. */
int sind_pipe_set_state_hashing(sind_pipe* p, int on);
int sind_pipe_get_state_hashes(sind_pipe* p, uint64_t* out, size_t count);
int sind_pipe_set_active_frames(sind_pipe* p, const int* frames_per_stream);
/* Retained steps: the repair runs of the chunked mode re-run only the stateful 1 % of a frame.  sind_pipe_reserve_retained(n) sets buffers for n steps aside
 * (device: flow, depth, plane-edge mask, normalised depth; page-locked host: depth, sample-grid flow); sind_pipe_retain_next(tag) keeps the phase-A outputs
 * (dense flow, depth copies, ORB front results, CalOccluded results; reference DynaDetect.cc:1023-1147, :429-642, ORBextractor.cc:1043-1151) of the NEXT
 * submitted step under `tag` once its tails have run; sind_pipe_replay(tag, first, last, outputs) runs the tails (DynaDetect.cc:315-420, 653-1018, 1163-1367,
 * 1553-1664 + dilation + mask filter of the keypoints) of frames [first[s], last[s]) of that step again for every stream, from the state the stream holds now
 * (sind_pipe_set_state) -- nothing may be pending; outputs in the step layout, only the frames that ran are written; sind_pipe_release_retained(tag) hands the
 * buffers back (tag < 0: all).  Not available with depth-ahead.  The reserve is EXACTLY n sets: unused sets beyond n are freed by the call (n = 0 frees all unused ones),
 * a call that fails for lack of memory keeps the sets it had completed -- call again with a smaller n to give the surplus back. */
int sind_pipe_reserve_retained(sind_pipe* p, int steps);
int sind_pipe_retain_next(sind_pipe* p, int tag);
int sind_pipe_replay(sind_pipe* p, int tag, const int* first, const int* last, uint8_t* dyna, uint8_t* label, uint8_t* mask_dilated,
                     sind_keypoint* kps, int cap, int* nkp, uint8_t* desc);
int sind_pipe_release_retained(sind_pipe* p, int tag);
/* A ragged or replayed step in which at most n streams have frames runs them as two chains per stream -- the depth half (k-means from the previous frame's merged
 * labels, SegAndMerge) ahead of the flow half (flow masks, fusion, dilation, keypoint filter), own k-means launches, no round barrier -- instead of batched rounds
 * (default 12; 0 = always rounds).  Same results; for a handful of streams on an otherwise idle GPU a frame costs max(depth, flow) instead of a whole round (the slow
 * runners of a repair: 1.10 -> 0.70 s on the Bonn-shaped stream, profiles/r04/repair_latency.txt). */
int sind_pipe_set_chain_max_streams(sind_pipe* p, int n);
/* per-stage wall times of the last step in milliseconds: {front_gray, dense_flow, orb_front, host_upload (sind_pipe_process only, else 0), tails, total},
 * plus HIP-event statistics of the flow solver: sor_launches, sor_ms (sum of event-bracketed SOR launch groups),
 * sor_alg_bytes (algorithmic bytes those launches cover: 44 B per pixel per red+black iteration) */
int sind_pipe_stats(sind_pipe* p, double* stage_ms6, long long* sor_launches, double* sor_ms, double* sor_alg_bytes);
/* flow-solver statistics of the last step when the batch is cut into concurrent slices (one HIP stream each): launches and algorithmic
 * bytes of all slices, sum_ms = sum of the event-bracketed launch groups, union_ms = time during which at least one slice had solver
 * launches in flight (union of those intervals on a common event time base), slices = number of concurrent streams */
int sind_pipe_sor_stats(sind_pipe* p, long long* launches, double* sum_ms, double* union_ms, double* alg_bytes, int* slices);
/* since round 5 the two calls above count the launch groups of the STREAMING solver only (k_sor_stream: the kernel of slices of 80 pairs and more, the one bench.py's roofline
 * object describes); this one reports the other solver launches of the step (tiles, one-workgroup levels outside k_coarse_chain, whose solver phases are not separate launches) */
int sind_pipe_sor_other_stats(sind_pipe* p, long long* launches, double* sum_ms, double* alg_bytes);
/* last sind_pipe_submit(_dev): time the call still waited for the previous step's tails after its own phase A had finished (0 = hidden) */
int sind_pipe_tail_wait_ms(sind_pipe* p, double* ms);
/* how the handle sized its host side: {CPU share of this process (cores), pool workers, CPU tokens (max), cores the process may run on, cgroup cpu.max quota in
 * cores or -1, ranks sharing the node (LOCAL_WORLD_SIZE)}.  share = min(cores, quota) / ranks, at least 4, at most 16. */
int sind_pipe_host_info(sind_pipe* p, int* out6);
/* several handles driven concurrently on one GPU share the process's CPU share: give each its part (cores >= 1; tokens of the pool tasks and the
 * CalOccluded runners follow; the worker threads stay as created) */
int sind_pipe_set_cpu_share(sind_pipe* p, int cores);
int sind_pipe_mask_bytes(sind_pipe* p, size_t* bytes);      /* size of one step's dyna / label / mask array: streams * frames_per_step * height * width */

/* ------------------------------------------------------------------------------------------------------------
 * The multi-GPU collective of the path: frames of a sequence shard across the GPUs of a node, one process per GPU, and the per-frame
 * dynamic masks are gathered with ONE ncclAllGather (RCCL, xGMI) per pipeline step.  No other collective exists: the path shards by frame.
 * Rank 0 calls sind_comm_unique_id and passes the 128 bytes to the other ranks out of band (the application's own channel); every rank then calls
 * sind_comm_create(id, rank, world, device).  sind_pipe_gather_masks sends the `dyna` array a step returned (host memory) and receives all ranks' arrays in
 * rank order into all_dev (device memory, world * sind_pipe_mask_bytes) and, if not NULL, all_host.  RCCL is loaded at first use (dlopen): a box
 * without it still loads the library and gets SIND_E_STATE from these calls. */
typedef struct sind_comm sind_comm;
int sind_comm_unique_id(void* id128, size_t bytes);
int sind_comm_create(const void* id128, int rank, int world, int device, sind_comm** out);
int sind_comm_destroy(sind_comm* c);
int sind_comm_rank(const sind_comm* c);
int sind_comm_world(const sind_comm* c);
int sind_comm_allgather_u8(sind_comm* c, const uint8_t* local_host, size_t bytes, uint8_t* all_dev, uint8_t* all_host_or_null);
/* one hand-over along the chain of ranks as one RCCL group: `bytes` bytes of host memory to rank `to` (-1: nothing to send) and as many from rank `from` (-1: nothing) */
int sind_comm_sendrecv_u8(sind_comm* c, const uint8_t* send_host, int to, uint8_t* recv_host, int from, size_t bytes);
int sind_pipe_gather_masks(sind_pipe* p, sind_comm* c, const uint8_t* dyna_host, uint8_t* all_dev, uint8_t* all_host_or_null);
/* Where the PEAC region grow of CalOccluded (AHCPlaneFitter.hpp:546-601) runs: `quarters` of every four frames on the GPU (k_peac_grow, one compute unit for
 * a few ms per frame), the others on a host core; the results are bit-identical, the share only moves load between the GPU and the host.  -1 (default): the
 * pipeline adapts the share step by step -- towards the GPU while a step waits for host work after its dense flow is done, back towards the host while
 * no step waits (the smallest GPU share the host keeps up with).  Environment SIND_GROW_GPU=0..4 fixes it at create. */
int sind_pipe_set_grow_share(sind_pipe* p, int quarters);
int sind_pipe_get_grow_share(sind_pipe* p, int* quarters);
/* How many groups of streams run their batched k-means rounds (reference DynaDetect.cc:315-420) as independent chains at the moment: 1..4 (at most streams / 8)
 * chosen by the same controller (one more while steps wait for the host although every region grow already runs on the GPU).  Results do not depend on it. */
int sind_pipe_set_kmeans_groups(sind_pipe* p, int groups);      /* 1 .. min(4, streams / 8), or -1 = adaptive (default); takes effect with the next step */
int sind_pipe_get_kmeans_groups(sind_pipe* p, int* groups);

/* ------------------------------------------------------------------------------------------------------------
 * Frame post-ORB steps: what the reference's RGB-D Frame constructor does with the extractor's output
 * (ORB_SLAM2/src/Frame.cc:143-170) for B frames at once:
 *   UndistortKeyPoints()           src/Frame.cc:477-509  -> un_xy   [B][cap][2]  mvKeysUn[i].pt  (identity when k1 == 0)
 *   ComputeStereoFromRGBD(imDepth) src/Frame.cc:714-735  -> depth_out / u_right [B][cap]  mvDepth / mvuRight (-1 when d <= 0)
 *   ComputeImageBounds(imGray)     src/Frame.cc:511-541  -> bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY}
 *   AssignFeaturesToGrid()         src/Frame.cc:283-299  -> cell [B][cap] = x * 48 + y (-1: PosInGrid false) and mGrid as CSR:
 *                                                           grid_start [B][3073], grid_idx [B][cap] (push_back order)
 * calib: mK, mDistCoef (k1 k2 p1 p2 k3), Camera.bf, depth_map_factor = 1 / DepthMapFactor (src/Tracking.cc:262-263 converts the
 * raw u16 depth with it).  kps [B][cap] / nkp [B]: the extractor's output (host).  depth: raw u16 [B][height][width], host or
 * device pointer (depth_on_device).  Output pointers are host and may be NULL.
 */
typedef struct sind_frame sind_frame;
typedef struct sind_frame_calib { float fx, fy, cx, cy, k1, k2, p1, p2, k3, bf, depth_map_factor; } sind_frame_calib;
int sind_frame_create(const sind_frame_calib* calib, int width, int height, int max_batch, int cap, int device, sind_frame** out);
int sind_frame_destroy(sind_frame* f);
int sind_frame_post_orb(sind_frame* f, const sind_keypoint* kps, const int* nkp, int B, const uint16_t* depth, int depth_on_device,
                        float* un_xy, float* u_right, float* depth_out, int* cell, int* grid_start, int* grid_idx, float* bounds4);

/* ------------------------------------------------------------------------------------------------------------
 * Projection matcher.  Replaces, for B (CurrentFrame, LastFrame) pairs at once,
 *   int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)   src/ORBmatcher.cc:1328-1470
 * including Frame::GetFeaturesInArea (src/Frame.cc:398-451), ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1647-1665), the
 * rotation histogram and ComputeThreeMaxima (:1601-1642).  The Frame / MapPoint object graph is passed as flat arrays:
 *   last frame, per keypoint i:  x3Dw = pMP->GetWorldPos(), last_valid = (pMP && !mvbOutlier[i]), last_has_obs = pMP->Observations() > 0,
 *                                last_octave = mvKeys[i].octave, last_angle = mvKeysUn[i].angle, last_desc = pMP->GetDescriptor() (32 B)
 *   current frame, per keypoint: cur_un_xy / cur_octave / cur_angle = mvKeysUn, cur_u_right = mvuRight, cur_desc = mDescriptors rows,
 *                                grid_start / grid_idx = mGrid (as sind_frame_post_orb returns it), cur_taken = mvpMapPoints[i2] &&
 *                                Observations() > 0 on entry (NULL = all free, the TrackWithMotionModel case src/Tracking.cc:914)
 * Output: match_of_cur[i2] = index i of the last-frame keypoint whose MapPoint the reference stores in
 * CurrentFrame.mvpMapPoints[i2] (-1: none or removed by the orientation check); *nmatches = the function's return value.
 * config: fx fy cx cy bf of the current frame, bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}, scale_factors = mvScaleFactors.
 * mCheckOrientation is the matcher's constructor flag (src/ORBmatcher.cc:41); nnratio is not used by this overload.
 */
typedef struct sind_match sind_match;
typedef struct sind_match_config { float fx, fy, cx, cy, bf; float bounds[4]; float scale_factors[16]; int nlevels, cap_last, cap_cur, max_batch, device; } sind_match_config;
typedef struct sind_match_pair {
    const float* Tcw_cur; const float* Tcw_last;                     /* 4x4 row-major poses (rows 0..2 are read) */
    int n_last; const float* x3Dw; const uint8_t* last_valid; const uint8_t* last_has_obs; const int* last_octave; const float* last_angle; const uint8_t* last_desc;
    int n_cur; const float* cur_un_xy; const int* cur_octave; const float* cur_angle; const float* cur_u_right; const uint8_t* cur_desc;
    const int* grid_start; const int* grid_idx; const uint8_t* cur_taken;
    int* match_of_cur; int* nmatches;                                /* outputs (host) */
} sind_match_pair;
int sind_match_create(const sind_match_config* cfg, sind_match** out);
int sind_match_destroy(sind_match* m);
int sind_match_by_projection(sind_match* m, const sind_match_pair* pairs, int B, float th, int mono, int check_orientation);
int sind_match_last_rounds(sind_match* m);      /* resolution rounds the last call needed (see csrc/match_kernels.hip) */

/* ------------------------------------------------------------------------------------------------------------
 * Mapping consumer.  Replaces, for B key frames at once, the body of
 *   generatePointCloud(imgRGB, imgDepth, imgDepthLast, imgDynaMask, imgDynaMaskLast, imgLabel, poseRelative, Twc)
 *                                                                              octomap_pub/src/pubPointCloud.cc:471-668
 * stride-2 back-projection, re-projection depth-consistency vote per cluster (:556-607, vecOcclusion), cluster rejection
 * (vecOcclusion[i] * 9 <= 0.4 * countNonZero(imgLabel == i) keeps cluster i, :641-663) and pcl::transformPointCloud(.., Twc) (:665).
 * Images are dense [B][height][width] (bgr x3), host or device pointers (inputs_on_device); pose_relative / Twc: [B][16] row-major
 * doubles (Eigen::Matrix4d values).  points [B][cap] receives tempCloudOneFrame in the reference's order (cluster 0, then the kept
 * clusters 1..11, raster order inside a cluster; masked / out-of-range pixels are NaN points, the cloud is not dense), n_points [B];
 * occlusion / label_count / kept: [B][12], may be NULL.  The statistical outlier filter and the octree insertion that follow in the
 * ROS node (:291-309) are PCL / octomap library calls and stay with the caller.
 */
typedef struct sind_cloud sind_cloud;
typedef struct sind_cloud_point { float x, y, z; uint8_t b, g, r, a; } sind_cloud_point;       /* pcl::PointXYZRGB payload */
int sind_cloud_create(double fx, double fy, double cx, double cy, double depth_scale, int width, int height, int max_batch, int device, sind_cloud** out);
int sind_cloud_destroy(sind_cloud* c);
int sind_cloud_max_points(sind_cloud* c);                    /* ceil(width / 2) * ceil(height / 2) */
int sind_cloud_generate(sind_cloud* c, int B, const uint8_t* bgr, const uint16_t* depth, const uint16_t* depth_last, const uint8_t* dyna, const uint8_t* dyna_last,
                        const uint8_t* label, const double* pose_relative, const double* Twc, int inputs_on_device, sind_cloud_point* points, int cap, int* n_points,
                        int* occlusion, int* label_count, int* kept);

/* helper of the rgbd_tum_noros-shaped harness (sindslam_amd/harness.py): PNG scanline reconstruction, raw = h x (1 + stride) bytes */
int sind_png_unfilter(const uint8_t* raw, int h, int stride, int bytes_per_pixel, uint8_t* out);

/* ------------------------------------------------------------------------------------------------------------
 * One long sequence, sharded by frame, with results EQUAL to the sequential loop (rgbd_tum_noros.cc:110-170 is the loop it equals).
 * The sequence is cut into world x streams contiguous lock-step chunks, one per pipeline stream; chunk 0 starts like the reference loop, a later chunk starts `warmup`
 * frames early from an empty state (speculation).  After the lock-step steps every chunk seam is VERIFIED by comparing 128-bit fingerprints of the inter-frame state
 * (DynaDetect.h:165-178, rolled at DynaDetect.cc:1660-1664) and a chunk whose rebuilt state is not its predecessor's true end state is REPAIRED: the stateful tails of
 * its first frames run again from the true state (on the retained phase-A outputs, then as whole frames on a small second pipeline) until the states agree.  Between
 * ranks a round costs one all-gather of 32 bytes per chunk and, for a mismatching seam between two ranks, one send / receive of the state blob -- over RCCL
 * (sind_seq_net_rccl on a sind_comm) or TCP (sind_seq_net_tcp: ranks without a communicator between them, several ranks rehearsed on one card).
 * Everything below is C++ inside the library (csrc/host/seq.cpp): a C++ caller needs neither Python nor torch.distributed for the exact sharded mode.
 *
 * Positions: position q = frame q + 1 of the sequence (frame 0 only primes, like the reference's first frame).  Outputs are caller arrays indexed by FRAME. */
typedef struct sind_seq sind_seq;
typedef struct sind_seq_net sind_seq_net;
typedef struct sind_seq_config {
    sind_pipe_config pipe;          /* streams = chunks PER RANK; frames_per_step is set by the plan (ignored on input) */
    long long frames;               /* positions of the job = sequence length - 1 */
    int steps;                      /* > 0: exactly this many lock-step steps (bench.py); 0: as many as frames_per_step asks for */
    int frames_per_step;            /* steps == 0: a step holds at most this many frames per chunk */
    int warmup;                     /* frames a chunk after the first starts early to rebuild the inter-frame state */
    int repair_streams, repair_frames_per_step;     /* the repair pipeline (0 = min(streams, 8) x 4) */
    int retain_frames;              /* steps holding the first n owned frames of the later chunks keep their phase-A outputs for replays; -1 = every step, 0 = none */
    int verify;                     /* 0: keep the speculative results (valid masks, not identical behind some seams) */
} sind_seq_config;
/* source callbacks: device pointers of the `count` frames at these positions, laid out [count][H][W][3] / [count][H][W] (valid until the next call), and the HOST bgr
 * frame of one position (priming); return 0 */
typedef int (*sind_seq_batch_fn)(void* user, const long long* positions, int count, const uint8_t** bgr_dev, const uint16_t** depth_dev);
typedef int (*sind_seq_frame_fn)(void* user, long long position, const uint8_t** bgr_host);
typedef int (*sind_seq_hook_fn)(void* user, int index);       /* step hook: the owned frames of step `index` are out; round hook: repair round `index` has ended (called on every rank) */
int sind_seq_net_tcp(int rank, int world, const char* host_or_null /* 127.0.0.1 */, int base_port, sind_seq_net** out);      /* rank r listens on base_port + r */
int sind_seq_net_rccl(sind_comm* c, sind_seq_net** out);
int sind_seq_net_destroy(sind_seq_net* n);
int sind_seq_create(const sind_seq_config* cfg, sind_seq_net* net_or_null /* one rank */, sind_seq** out);
int sind_seq_destroy(sind_seq* q);
int sind_seq_plan(sind_seq* q, int* frames_per_step, int* steps, int* n_chunks, long long* first_last_start /* n_chunks x 3, or NULL */);
int sind_seq_set_host_source(sind_seq* q, const uint8_t* bgr, const uint16_t* depth, long long n_frames);      /* the whole sequence in host memory: [n][H][W][3] u8, [n][H][W] u16 */
int sind_seq_set_source(sind_seq* q, sind_seq_batch_fn batch, sind_seq_frame_fn frame, void* user);
/* sink: arrays of n_frames entries indexed by frame (dyna / label / mask [n][H][W]; kps [n][cap], nkp [n], desc [n][cap][32]); any may be NULL; a rank fills the frames it owns */
int sind_seq_set_outputs(sind_seq* q, long long n_frames, uint8_t* dyna, uint8_t* label, uint8_t* mask_dilated, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc);
int sind_seq_set_hooks(sind_seq* q, sind_seq_hook_fn step_hook, sind_seq_hook_fn round_hook, void* user);
int sind_seq_prime(sind_seq* q);
int sind_seq_warm(sind_seq* q, int steps);        /* untimed rehearsal after sind_seq_prime: the first `steps` steps synchronously + one step of the repair pipeline; call sind_seq_prime again */
int sind_seq_set_emit_main(sind_seq* q, int on);  /* 0: the frames of a lock-step step are not copied to the sink (the step hook reads sind_seq_step_outputs itself); repaired frames always are */
int sind_seq_submit(sind_seq* q, int step);       /* steps 0 .. steps - 1 in order; software-pipelined: the results of step - 1 are delivered */
int sind_seq_flush(sind_seq* q);                  /* delivers the last step */
int sind_seq_verify(sind_seq* q);                 /* seam verification and repairs (collective over the ranks) */
int sind_seq_run(sind_seq* q);                    /* prime + every step + flush + verify */
/* seams, mismatched seams, rounds, runners, repaired chunks, repair frames, repair steps, overridden frames, runners to chunk end, max frames to converge, replay frames,
 * replay calls, runners past replay, retained steps dropped, repair seconds, flush seconds */
int sind_seq_stats(sind_seq* q, double out16[16]);
sind_pipe* sind_seq_pipeline(sind_seq* q);        /* the main pipeline of this rank (sind_pipe_stats, sind_pipe_gather_masks, ...) */
int sind_seq_step_outputs(sind_seq* q, const uint8_t** dyna, const uint8_t** label, const uint8_t** mask_dilated);   /* the page-locked [S][T][H][W] arrays of the last delivered step */

#ifdef __cplusplus
}
#endif
#endif
