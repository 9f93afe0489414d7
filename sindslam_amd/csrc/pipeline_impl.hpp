#pragma once
// Batched multi-stream pipeline (include/sind_hip.h "sind_pipe"): the frame-loop body of the reference's
// Examples/RGB-D/rgbd_tum_noros.cc:110-170 (DetectDynaArea -> 15x15 dilate -> ORBextractor via Frame::ExtractORB2) for
// S independent streams x T frames per step.
//   phase A (state free, one batch of S*T frames on the shared HIP stream): gray, 0.6 resize, dense flow, ORB front
//   phase B (stateful, frame order inside a stream; one task per frame on a fixed worker pool, one HIP stream per worker):
//            DynaDetect tail, dilation, dynamic-mask erasure of the ORB keypoints.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <ctime>
#include <cstdio>
#include "../../include/sind_hip.h"
#include "dyna.hpp"
#include "orb.hpp"

using namespace sind;

// Fixed pool of host workers shared by the CalOccluded tasks of the step in phase A and the stateful tails of the step in phase B.
// The GPU boxes give a process a bounded CPU share (16 cores per GPU on this pool): one bounded pool instead of a thread set per
// phase keeps the runnable threads under that share, which matters most in the pipelined mode where both kinds of task coexist.
// CPU time (user + system) of the calling thread in ms, for the SIND_TAIL_TIMING report of the short-lived phase-A threads
static inline double thread_cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline std::atomic<long long> g_cpu_us_flow{0}, g_cpu_us_orb{0}, g_cpu_steps{0};

struct TaskGroup { std::mutex m; std::condition_variable cv; int left = 0; };
class WorkerPool {
public:
    void start(int n, int device, SindHostGate* gate, int spin_us = 0) {
        for (int i = 0; i < n; i++) th.emplace_back([this, i, device, gate, spin_us] {
            (void)pthread_setname_np(pthread_self(), "sind-worker");      // names show up in /proc/<pid>/task/*/comm (bench.py --thread-cpu)
            (void)hipSetDevice(device);
            t_sind_spin_us = spin_us;
            for (;;) {
                std::pair<std::function<void(int)>, TaskGroup*> job;
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return stop || !q.empty(); }); if (q.empty()) return; job = std::move(q.front()); q.pop_front(); }
                gate->acquire(); t_sind_gate = gate;          // a CPU token while the task computes (handed back inside every wait for the GPU)
                job.first(i);
                t_sind_gate = nullptr; gate->release();
                { std::lock_guard<std::mutex> lk(job.second->m); if (--job.second->left == 0) job.second->cv.notify_all(); }
            } });
    }
    void push(TaskGroup& g, std::function<void(int)> fn) {
        { std::lock_guard<std::mutex> lk(g.m); g.left++; }
        { std::lock_guard<std::mutex> lk(m); q.emplace_back(std::move(fn), &g); }
        cv.notify_one();
    }
    static void wait(TaskGroup& g) { std::unique_lock<std::mutex> lk(g.m); g.cv.wait(lk, [&g] { return g.left == 0; }); }
    int size() const { return (int)th.size(); }
    ~WorkerPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv.notify_all(); for (auto& t : th) if (t.joinable()) t.join(); }
private:
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv; std::deque<std::pair<std::function<void(int)>, TaskGroup*>> q; bool stop = false;
};

struct sind_pipe {
    sind_pipe_config c{}; DynaConfig dc; int S = 0, T = 0, fw = 0, fh = 0;
    hipStream_t stream = nullptr, orb_stream = nullptr; hipEvent_t ev_gray = nullptr, ev_depth = nullptr; std::vector<hipStream_t> worker_streams, worker_streams_lo;      // one HIP stream per pool worker, shared by the tasks it runs
    DynaFront front; std::vector<std::unique_ptr<DynaFront>> extra_fronts; std::vector<hipStream_t> extra_streams; hipEvent_t ev_pool = nullptr;      // batch slices 1.. of the dense flow (slice 0 = front)
    OrbEngine orb; std::vector<std::unique_ptr<DynaTail>> tails;
    // Depth halves (k-means warm labels + their workspaces) as objects of their own, present once depth-ahead has been switched on: the
    // depth chain of step i+1 (phase A) may then run while the flow chain of step i (phase B) is still going on the same stream of frames.
    std::vector<std::unique_ptr<DynaTail>> dtails;
    // k-means of one frame of every stream as ONE batched kernel chain (phase B then runs frame t of all streams as a round: batched k-means,
    // then the S tails of that frame on the pool); used when there are several streams and the depth half is not run ahead
    static constexpr int KM_GROUPS = 4;
    KMeansBatch kmb[KM_GROUPS]; int km_groups = 1, km_groups_max = 1, km_groups_fixed = -1; hipStream_t km_stream = nullptr, km_streams[KM_GROUPS] = {nullptr}; bool batch_km = false;
    std::vector<std::thread> round_threads; std::mutex km_stat_mu; double km_round_ms = 0; long km_rounds = 0;
    std::vector<std::unique_ptr<PinnedBuf<uint8_t>>> upload_stage;        // page-locked staging of sind_pipe_process (host-buffer entry point), two 4 MB buffers per uploading worker
    std::vector<std::unique_ptr<DynaTail>> occ_tails;     // CalOccluded workspaces, one per pool worker (state free)
    // GPU half of CalOccluded for all frames of a step, in chunks, on a stream of its own at the start of phase A (seven launches per chunk instead
    // of seven launches + a stream wait per frame); the runner tasks wait for their frame's chunk and do the host half
    OccBatch occb; hipStream_t occ_stream = nullptr; bool batch_occ = false; int occ_chunk = 64;
    // PEAC region grow of CalOccluded on the GPU, one launch per chunk of frames (peac_grow.hpp): the runner that finishes the last first-half of a chunk enqueues it
    PeacGrowBatch grow; hipStream_t grow_stream = nullptr; std::mutex grow_mu; bool grow_ok = false;       // grow_ok: the frame size fits the kernel (else every frame grows on the host)
    // Where a frame's region grow runs: grow_q of every 4 frames on the GPU (one CU for ~6 ms per frame), the others on the host (one core for ~5 ms); both give
    // the same bits, so the share only moves load.  grow_q_fixed < 0: adapted step by step (grow_adapt) -- towards the GPU while the step waits for host work
    // (CalOccluded or tails not done when the dense flow is), back towards the host while no step waits.
    // (the share starts with every grow on the GPU: measured in round 4, the controller ends there within a few steps at both sizes, and a run that starts there keeps the host two cores cooler at the same rate)
    int grow_q = 4, grow_q_fixed = -1, grow_idle_steps = 0; int cpu_share = 16; int host_info[6] = {0, 0, 0, 0, -1, 1};      // host_info: share, workers, tokens, cores usable, cgroup quota (-1 none), ranks of the node
    DevBuf<uint8_t> bgr_d, gray, gray_orb, pool; DevBuf<uint16_t> depth_d;
    // two sets of phase-A outputs: step i's phase A (GPU) overlaps with step i-1's phase B (host threads + small kernels)
    struct StepBuf {
        DevBuf<float> U, V; DevBuf<uint16_t> depth_dev; PinnedBuf<uint16_t> depth_h;       /* page-locked: the 157 MB device-to-host copy of a step must not block the enqueueing thread */ std::vector<OrbFrameResult> orb; std::vector<OccResult> occ; bool pending = false;
        DevBuf<uint8_t> occ2_dev, depthN_dev;                  // per frame: plane-edge mask and normalised depth for the tails' RAG statistics (filled by the CalOccluded tasks)
        DevBuf<float> grid_dev; PinnedBuf<float> grid_h;       // flow at the 10-px sample grid of every frame (DD:1182-1204), gathered right after the dense flow
        std::atomic<int> occ_next{0};                          // next frame for the CalOccluded runner tasks
        std::vector<OccCtx> occ_ctx; PinnedBuf<uint8_t> grow_in_h, grow_pair_h; PinnedBuf<int8_t> grow_member_h; PinnedBuf<int> grow_status_h;      // per frame: state between the halves, the grow's input block and results
        std::vector<hipEvent_t> grow_ev; std::unique_ptr<std::atomic<int>[]> grow_left, grow_state; std::atomic<int> occ_next2{0}; int grow_q = 4;      // per chunk: first halves still out, 0 = not launched / 1 = launched / < 0 = failed
        PinnedBuf<uint8_t> occ_edge_h, occ_total_h; PinnedBuf<PeacBlockStats> occ_blocks_h; std::vector<hipEvent_t> occ_ev, occ2_ev;      // batched GPU half: per-frame host results, one event per chunk; one event per frame behind its occ2 upload
        // depth half of the tails (k-means, SegAndMerge) run ahead, underneath the dense flow (synchronous steps only): per-frame results,
        // and a gate per frame that opens when both its CalOccluded result and the stream's previous depth stage are there
        bool depth_ahead = false; std::vector<DepthStageOut> dout; std::unique_ptr<std::atomic<int>[]> gate; TaskGroup depth_group;
        std::vector<int> depth_rc; std::vector<std::string> depth_err;
        TaskGroup occ_group, tail_group, km_tails[4]; int km_groups = 1, km_first[5] = {0, 0, 0, 0, 0};       /* (4 = sind_pipe::KM_GROUPS) the step's own partition of the streams */ std::vector<int> occ_rc, tail_rc, dchain_rc; std::vector<std::string> occ_err, tail_err, dchain_err;      /* dchain_*: the depth chain of a stream in two-chain mode (its flow chain writes tail_*: two workers, two slots) */
        std::vector<int> active, first; std::vector<uint64_t> state_hash;      // tails of stream s run for first[s] <= t < active[s] (empty: 0 / all T); per-frame state fingerprints [S][T][2]
        bool few_chain = false;                                        // this step runs a handful of streams as per-stream chains (see phase_b_start)
        bool two_chain = false; std::unique_ptr<std::atomic<int>[]> fgate; int fgate_n = 0;      // ... each as a depth chain running ahead of a flow chain; per frame: depth stage done + previous flow stage done
        int retain_tag = -1;                                           // >= 0: the phase-A outputs of this step are kept under this tag when its tails are done
    } sb[2];
    // Phase-A outputs of a step kept beyond the step (sind_pipe_retain_next): everything the tails read -- dense flow, depth copies, ORB front results,
    // CalOccluded results, sample-grid flow -- so that sind_pipe_replay can run the stateful tails of those frames again from another state without
    // computing the state-free 99 % of the frame again (the repair runs of the chunked sequence mode).  Buffers come from a reserve made up front.
    struct Retained { DevBuf<float> U, V, grid_dev; DevBuf<uint16_t> depth_dev; PinnedBuf<uint16_t> depth_h; PinnedBuf<float> grid_h; DevBuf<uint8_t> occ2_dev, depthN_dev;
                      std::vector<OrbFrameResult> orb; std::vector<OccResult> occ; int tag = -1; };
    std::vector<std::unique_ptr<Retained>> spare, kept; int retain_tag_next = -1;
    int cur = 0; int occ_workers = 24;
    // CPU tokens (common.hpp) for the software-pipelined steps, where CalOccluded runners and tails compete for the quota (measured: throttled periods 7 -> 2
    // of 22, +1 %); synchronous steps run ungated -- there the hand-over of tokens at every GPU wait costs more than the throttling (tails 145 -> 173 ms)
    int cpu_tokens = 15, cpu_tokens_min = 13, cpu_tokens_max = 15; bool cpu_tokens_fixed = false;
    // Optional schedule of the synchronous step: run the depth half of the tails (k-means, SegAndMerge) underneath the dense flow.
    // Parity-tested, off by default: the tails phase shrinks from ~75 to ~23 ms, but the solver loses as much to the ~13 k extra small
    // launches it then shares the GPU with (dense flow 232 -> 287 ms at high stream priority; at normal priority the chains starve).
    bool depth_ahead = false;
    std::vector<char> primed;
    // Chunked sequences (sindslam_amd/sequence.py): hashing = every tail leaves the fingerprint of its rolled state per frame (last_hash: the step whose results
    // were returned last, [S][T][2]); active_next = per-stream number of frames whose TAILS run in the next step (one step only; empty = all T)
    bool hashing = false; std::vector<uint64_t> last_hash; std::vector<int> active_next; int chain_max_streams = 12; bool split_rounds = true;      /* rounds: the next k-means waits for the depth halves of the tails only (pipeline_tails.cpp) */
    double stage_ms[6] = {0}; double tail_wait_ms = 0; double sor_ms = 0, sor_union_ms = 0, sor_bytes = 0; long long sor_launches = 0; int sor_slices = 1;      // streaming solver (k_sor_stream) launch groups of the last step
    double sor_other_ms = 0, sor_other_bytes = 0; long long sor_other_launches = 0;                             // every other solver kernel outside k_coarse_chain (tiles, one-workgroup levels)
    SindHostGate gate;           // CPU tokens of this handle's pool tasks (common.hpp)
    WorkerPool workers;          // declared last: joined first
};

static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// One step of the grow-share controller (see sind_pipe::grow_q).  host_wait_ms: how long the step waited for host work after its GPU work was done;
// step_ms: wall time of the step.  The GPU is the scarcer resource of the two (a frame's grow costs one compute unit ~6 ms against one core ~5 ms, and a
// box has 256 of the one and 16 of the other -- but the solver wants all 256), so the share settles at the SMALLEST one the host keeps up with: a quarter
// more to the GPU as soon as a step waits for the host (> 3 % of the step), a quarter back after three steps in a row without any wait (< 0.5 %).
static inline void grow_adapt(sind_pipe* p, double host_wait_ms, double step_ms) {
    if (p->grow_q_fixed >= 0 || !p->batch_occ || step_ms <= 0) return;
    if (host_wait_ms > 0.03 * step_ms) {
        if ((p->grow_q >= 4 || host_wait_ms > 0.15 * step_ms) && p->batch_km && p->km_groups_fixed < 0) p->km_groups = std::min(p->km_groups_max, p->km_groups + 1);      // every grow is on the GPU already (or the wait is long): one more k-means chain
        p->grow_q = std::min(4, p->grow_q + 1); p->grow_idle_steps = 0;
        // clearly host-bound (the step waits for the tails for > 10 % of its time): a few more tokens than cores (a token is held through short waits too: 1280 x 720 + 3 - 6 %); else one more, up to share - 1
        if (!p->cpu_tokens_fixed) p->cpu_tokens = host_wait_ms > 0.10 * step_ms ? p->cpu_tokens_max : std::max(p->cpu_tokens, std::min(p->cpu_tokens_max - 3, p->cpu_tokens + 1));
    } else if (host_wait_ms < 0.005 * step_ms) {
        if (!p->cpu_tokens_fixed) p->cpu_tokens = std::max(p->cpu_tokens_min, p->cpu_tokens - 1);
        if (++p->grow_idle_steps >= 3) {
            if (p->batch_km && p->km_groups > 1 && p->km_groups_fixed < 0) p->km_groups--; else p->grow_q = std::max(0, p->grow_q - 1);
            p->grow_idle_steps = 0;
        }
    } else p->grow_idle_steps = 0;
}

// max filter of the 0/125/255 image with the 15x15 ellipse = two binary dilations (>=125, ==255)
static inline void dilate15_codes(const uint8_t* src, int W, int H, uint8_t* dst) {
    const EllipseElem e15(15);
    BitImg hi = BitImg::from_equal(src, W, H, W, 255), any = BitImg::from_u8(src, W, H, W);
    hi = hi.dilated(e15); any = any.dilated(e15);
    any.to_u8(dst, W, 125); hi.paint_u8(dst, W, 255);
}

// Tail / CalOccluded streams are high priority: their small kernels overtake the batch stream's flow solver when both are in flight.
// (A CU partition via hipExtStreamCreateWithCUMask was measured on MI355X: every masked stream ran 3-4x slower, see DESIGN.md.)
static inline int make_stream(hipStream_t* out, bool high_priority) {
    if (high_priority) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIP_TRY(hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi)); }
    else HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return SIND_OK;
}

// ---- the parts of the pipeline (pipeline_build.cpp: handle, streams, pool; pipeline_phase_a.cpp: the state-free batch of a step; pipeline_tails.cpp: the stateful tails
// on the worker pool; pipeline_capi.cpp: the step entry points, state blobs, retained steps / replay, settings and statistics of the C ABI)
struct PipeOut { uint8_t *dyna, *label, *mask; sind_keypoint* kps; int cap; int* nkp; uint8_t* desc; };
int pipe_build(sind_pipe* p, const sind_pipe_config* cfg);
int ensure_dtails(sind_pipe* p);
DynaTail* depth_half(sind_pipe* p, int s);
int phase_a(sind_pipe* p, sind_pipe::StepBuf& sb, const uint8_t* bgr_dev, const uint16_t* depth_dev, double t[4], bool depth_ahead = false);
void depth_task(sind_pipe* p, sind_pipe::StepBuf* sb, int k, int worker);
void phase_b_start(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o);
int phase_b_finish(sind_pipe* p, sind_pipe::StepBuf& sb);
int phase_b(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o);
void swap_phase_a_outputs(sind_pipe::StepBuf& sb, sind_pipe::Retained& r);

