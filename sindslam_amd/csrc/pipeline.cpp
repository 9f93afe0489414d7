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
static double thread_cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static std::atomic<long long> g_cpu_us_flow{0}, g_cpu_us_orb{0}, g_cpu_steps{0};

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
    bool hashing = false; std::vector<uint64_t> last_hash; std::vector<int> active_next; int chain_max_streams = 12;
    double stage_ms[6] = {0}; double tail_wait_ms = 0; double sor_ms = 0, sor_union_ms = 0, sor_bytes = 0; long long sor_launches = 0; int sor_slices = 1;      // streaming solver (k_sor_stream) launch groups of the last step
    double sor_other_ms = 0, sor_other_bytes = 0; long long sor_other_launches = 0;                             // every other solver kernel outside k_coarse_chain (tiles, one-workgroup levels)
    SindHostGate gate;           // CPU tokens of this handle's pool tasks (common.hpp)
    WorkerPool workers;          // declared last: joined first
};

static double now_ms();
// One step of the grow-share controller (see sind_pipe::grow_q).  host_wait_ms: how long the step waited for host work after its GPU work was done;
// step_ms: wall time of the step.  The GPU is the scarcer resource of the two (a frame's grow costs one compute unit ~6 ms against one core ~5 ms, and a
// box has 256 of the one and 16 of the other -- but the solver wants all 256), so the share settles at the SMALLEST one the host keeps up with: a quarter
// more to the GPU as soon as a step waits for the host (> 3 % of the step), a quarter back after three steps in a row without any wait (< 0.5 %).
static void grow_adapt(sind_pipe* p, double host_wait_ms, double step_ms) {
    if (p->grow_q_fixed >= 0 || !p->batch_occ || step_ms <= 0) return;
    if (host_wait_ms > 0.03 * step_ms) {
        if ((p->grow_q >= 4 || host_wait_ms > 0.15 * step_ms) && p->batch_km && p->km_groups_fixed < 0) p->km_groups = std::min(p->km_groups_max, p->km_groups + 1);      // every grow is on the GPU already (or the wait is long): one more k-means chain
        p->grow_q = std::min(4, p->grow_q + 1); p->grow_idle_steps = 0;
        if (!p->cpu_tokens_fixed) p->cpu_tokens = host_wait_ms > 0.10 * step_ms ? p->cpu_tokens_max : std::min(p->cpu_tokens_max, p->cpu_tokens + 1);      // clearly host-bound: every core of the share at once; else one more
    } else if (host_wait_ms < 0.005 * step_ms) {
        if (!p->cpu_tokens_fixed) p->cpu_tokens = std::max(p->cpu_tokens_min, p->cpu_tokens - 1);
        if (++p->grow_idle_steps >= 3) {
            if (p->batch_km && p->km_groups > 1 && p->km_groups_fixed < 0) p->km_groups--; else p->grow_q = std::max(0, p->grow_q - 1);
            p->grow_idle_steps = 0;
        }
    } else p->grow_idle_steps = 0;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// max filter of the 0/125/255 image with the 15x15 ellipse = two binary dilations (>=125, ==255)
static void dilate15_codes(const uint8_t* src, int W, int H, uint8_t* dst) {
    const EllipseElem e15(15);
    BitImg hi = BitImg::from_equal(src, W, H, W, 255), any = BitImg::from_u8(src, W, H, W);
    hi = hi.dilated(e15); any = any.dilated(e15);
    any.to_u8(dst, W, 125); hi.paint_u8(dst, W, 255);
}

// Tail / CalOccluded streams are high priority: their small kernels overtake the batch stream's flow solver when both are in flight.
// (A CU partition via hipExtStreamCreateWithCUMask was measured on MI355X: every masked stream ran 3-4x slower, see DESIGN.md.)
static int make_stream(hipStream_t* out, bool high_priority) {
    if (high_priority) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); HIP_TRY(hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi)); }
    else HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return SIND_OK;
}

extern "C" {

int sind_pipe_destroy(sind_pipe* p);
static int ensure_dtails(sind_pipe* p);
static int pipe_build(sind_pipe* p, const sind_pipe_config* cfg) {
    p->c = *cfg; p->S = cfg->streams; p->T = cfg->frames_per_step;
    p->dc.W = cfg->width; p->dc.H = cfg->height; p->dc.fx = cfg->fx; p->dc.fy = cfg->fy; p->dc.cx = cfg->cx; p->dc.cy = cfg->cy; p->dc.depthScale = cfg->depth_scale; p->dc.device = cfg->device;
    const bool flow_hi = sind_lab_env("SIND_FLOW_PRIORITY") && atoi(sind_lab_env("SIND_FLOW_PRIORITY")) != 0;
    SIND_TRY(make_stream(&p->stream, flow_hi)); SIND_TRY(make_stream(&p->orb_stream, false)); HIP_TRY(hipEventCreateWithFlags(&p->ev_gray, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->ev_depth, hipEventDisableTiming));
    const int B = p->S * p->T; const size_t np = (size_t)cfg->width * cfg->height;
    // dense-flow slices: concurrent streams keep the GPU busy through the launch tails and the small pyramid levels of each other; small batches stay in one piece.
    // Measured (profiles/r04/split_sweep.txt, lab build): 512 pairs per step: 3 slices 1382-1417 pairs/s, 2: 1377, 4: 1354; 224 pairs (the one-GPU sequence job's step): 2 slices
    // 1192, 1: 1164, 3: 1150, 4: 1120.  Round 5 (k_coarse_chain, k_sor_tile; profiles/r05/small_step_slices.txt): slices of 12 - 50 pairs keep each other's latency-bound launches
    // company -- 24 pairs: 2 slices 750 pairs/s (1: 671), 48: 2 -> 935 (1: 607), 64: 2 -> 1042 (1: 741), 96: 3 -> 1130 (2: 991); slices of 80 and more take the streaming solver
    const int nsplit = std::max(1, std::min(sind_lab_env("SIND_FLOW_SPLIT") ? atoi(sind_lab_env("SIND_FLOW_SPLIT")) : (cfg->flow_slices > 0 ? cfg->flow_slices : B >= 320 ? 3 : B >= 160 ? 2 : B >= 80 ? 3 : B >= 24 ? 2 : 1), std::min(4, B))), Bs = (B + nsplit - 1) / nsplit;
    SIND_TRY(p->front.init(p->dc, nsplit > 1 ? std::max(Bs, 2) : B, p->stream));
    HIP_TRY(hipEventCreate(&p->ev_pool));                   // with timing: also the time base of the solver intervals
    for (int i = 1; i < nsplit; i++) {
        hipStream_t st = nullptr; SIND_TRY(make_stream(&st, flow_hi)); p->extra_streams.push_back(st);
        p->extra_fronts.emplace_back(new DynaFront()); SIND_TRY(p->extra_fronts.back()->init(p->dc, Bs, st));
    }
    p->front.flow.max_levels = std::max(0, cfg->flow_max_levels); for (auto& f : p->extra_fronts) f->flow.max_levels = p->front.flow.max_levels;
    p->front.flow.coarse_chain = !(cfg->flow_opts_off & 1); p->front.flow.latency_tiles = !(cfg->flow_opts_off & 2);
    for (auto& f : p->extra_fronts) { f->flow.coarse_chain = p->front.flow.coarse_chain; f->flow.latency_tiles = p->front.flow.latency_tiles; }
    p->fw = p->front.fw; p->fh = p->front.fh;
    SIND_TRY(p->orb.init(cfg->width, cfg->height, cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th_fast, cfg->min_th_fast, B, p->orb_stream));
    // CPU share of this process: the cores it may run on (affinity), bounded by the container's quota (cgroup v2 cpu.max: 16 cores per GPU on the MI355X
    // boxes) and divided among the ranks of the node when a launcher started several in this container (LOCAL_WORLD_SIZE: they share cores and quota) --
    // never below 4 where no quota is set (below 2 where one is), so that a rank keeps a working pool on a lease whose quota was not scaled with the GPU count.  sind_pipe_host_info reports the decision.
    int nproc = (int)std::thread::hardware_concurrency(); if (nproc <= 0) nproc = 16;
    { cpu_set_t set; CPU_ZERO(&set); if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) nproc = std::min(nproc, a); } }
    int cpu_share = nproc, quota = -1, lw = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { long long q = 0, per = 0; if (fscanf(f, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) { quota = (int)std::max<long long>(1, q / per); cpu_share = std::min(cpu_share, quota); } fclose(f); }
    // (the floor of 4 only where no quota bounds the container: with a quota the shares of all ranks must stay inside it -- 8 ranks on a 16-core quota get 2 cores each, not 4,
    // or the container burns its quota early in every period and the kernel stalls all of its threads until the period ends)
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) { lw = std::max(1, atoi(e)); if (lw > 1) cpu_share = quota > 0 ? std::max(std::min(2, cpu_share), cpu_share / lw) : std::max(std::min(4, cpu_share), cpu_share / lw); }
    cpu_share = std::min(cpu_share, 16);                                          // more host threads than this per GPU bring nothing (measured)
    p->host_info[0] = cpu_share; p->host_info[3] = nproc; p->host_info[4] = quota; p->host_info[5] = lw;
    p->cpu_share = cpu_share;
    if (const char* e = getenv("SIND_GROW_GPU")) { p->grow_q_fixed = std::max(0, std::min(4, atoi(e))); p->grow_q = p->grow_q_fixed; }
    const int nworkers = sind_lab_env("SIND_WORKERS") ? std::max(2, atoi(sind_lab_env("SIND_WORKERS"))) : cfg->host_threads > 0 ? cfg->host_threads : std::max(2, cpu_share * 2);         // default: 2x the CPU share (workers sleep while they wait for the GPU)
    // A task leaves its stream idle (every GPU section ends in a wait), so the HIP streams belong to the workers, not to the camera
    // streams: their number does not grow with S.
    p->worker_streams.resize(nworkers); p->occ_tails.resize(nworkers); p->tails.resize(p->S);
    // Two streams per worker: phase-B tasks use a high-priority stream; phase-A tasks (CalOccluded, optional depth stages) have their own
    // stream, high priority by default as well.  Measured on MI355X: at normal priority these small kernels starve behind the solver's
    // workgroups (dense flow 230 -> 221 ms, but 65-90 ms of CalOccluded / depth-stage work is then left over when the flow ends); at high
    // priority they cost the solver about what they would cost alone.  Either way the small kernels of a step are worth ~90 ms of GPU time.
    p->worker_streams_lo.resize(nworkers);
    const bool phase_a_hi = !(sind_lab_env("SIND_PHASEA_PRIORITY") && atoi(sind_lab_env("SIND_PHASEA_PRIORITY")) == 0);
    for (int w = 0; w < nworkers; w++) {
        SIND_TRY(make_stream(&p->worker_streams[w], !(sind_lab_env("SIND_TAIL_PRIORITY") && atoi(sind_lab_env("SIND_TAIL_PRIORITY")) == 0)));
        if (phase_a_hi) p->worker_streams_lo[w] = p->worker_streams[w];      // same stream (extra streams would also change how the runtime spreads the workers' streams over its hardware queues)
        else SIND_TRY(make_stream(&p->worker_streams_lo[w], false));
        p->occ_tails[w].reset(new DynaTail()); SIND_TRY(p->occ_tails[w]->init(p->dc, p->worker_streams_lo[w]));
    }
    for (int s = 0; s < p->S; s++) { p->tails[s].reset(new DynaTail()); SIND_TRY(p->tails[s]->init(p->dc, p->worker_streams[s % nworkers]));
        if (p->S == 1) p->tails[s]->piece_threads = std::max(1, std::min(4, cpu_share / 3)); }       // one stream: the tails are two serial chains, host cores idle
    // CPU tokens of the pool tasks: the flow's three launch threads, the ORB thread and the round drivers run beside them and are not gated, so share - 1 tokens
    // overshoot the quota in bursts (10-17 of 77 periods throttled) and share - 3 do not (0 periods, -1 % at 640x480 where the GPU is the bottleneck); a
    // host-bound configuration wants every core it can get.  The controller below moves between the two on the same signal as the region grow's share.
    p->cpu_tokens_max = std::max(2, cpu_share - 1); p->cpu_tokens_min = std::max(2, cpu_share - 3); p->cpu_tokens = p->cpu_tokens_min;
    if (sind_lab_env("SIND_CPU_TOKENS")) { p->cpu_tokens = std::max(1, atoi(sind_lab_env("SIND_CPU_TOKENS"))); p->cpu_tokens_fixed = true; }
    p->host_info[1] = nworkers; p->host_info[2] = p->cpu_tokens_max;
    p->occ_workers = std::max(1, std::min(nworkers, cpu_share - 2));             // CalOccluded runners: leave two cores of the share to the flow's launch threads
    if (const char* e = sind_lab_env("SIND_OCC_WORKERS")) p->occ_workers = std::max(1, std::min(atoi(e), nworkers));
    p->depth_ahead = sind_lab_env("SIND_DEPTH_AHEAD") && atoi(sind_lab_env("SIND_DEPTH_AHEAD")) != 0;
    if (p->depth_ahead) SIND_TRY(ensure_dtails(p));
    p->batch_km = p->S >= 2 && !(sind_lab_env("SIND_KM_BATCH") && atoi(sind_lab_env("SIND_KM_BATCH")) == 0);
    if (p->batch_km) {
        // Two to four groups of streams, each with its own batched k-means chain, HIP stream and round thread: a round is ~60 dependent launches and takes
        // ~20 ms next to the flow solver whatever the batch (24 or 128 frames), and while ONE batch for all streams ran, every tail worker was idle --
        // 80 of a 280 ms tail phase at 1280x720.  The groups are independent (a stream's k-means needs only its own previous frame's merged labels), so
        // one group's round overlaps the other groups' tails.  How many: more chains take more of the GPU from the flow solver (four instead of two cost
        // 5 % at 640x480, where the GPU is the bottleneck, and bring 3 % at 1280x720, where the host is), so the count follows the same signal as the
        // region grow's share (grow_adapt): one to begin with, one more when steps wait for the host although every grow already runs on the GPU.
        p->km_groups_max = std::max(1, std::min((int)sind_pipe::KM_GROUPS, p->S / 8)); p->km_groups = 1;
        for (int g = 0; g < p->km_groups_max; g++) {                          // group 0 may hold all streams, the others at most half of them
            SIND_TRY(make_stream(&p->km_streams[g], true)); SIND_TRY(p->kmb[g].init(p->dc, g == 0 ? p->S : (p->S + 1) / 2, p->km_streams[g])); }
    }
    p->batch_occ = B >= 4 && !(sind_lab_env("SIND_OCC_BATCH") && atoi(sind_lab_env("SIND_OCC_BATCH")) == 0);
    if (p->batch_occ) {
        // frames per launch of CalOccluded's GPU half and of the region grow: 128 (profiles/r04/lab_settings_sweep.txt, 512 frames per step: 32: 1295, 64: 1438-1468, 128: 1474-1510,
        // 192: 1506, 256: 1492, 512: 1499 pairs/s)
        p->occ_chunk = std::min(B, std::max(1, sind_lab_env("SIND_OCC_CHUNK") ? atoi(sind_lab_env("SIND_OCC_CHUNK")) : 128));
        SIND_TRY(make_stream(&p->occ_stream, !(sind_lab_env("SIND_OCC_PRIORITY") && atoi(sind_lab_env("SIND_OCC_PRIORITY")) == 0))); SIND_TRY(p->occb.init(p->dc, p->occ_chunk));
        const size_t nblk = (size_t)(cfg->width / 16) * (cfg->height / 16); const int nch = (B + p->occ_chunk - 1) / p->occ_chunk;
        for (int k = 0; k < 2; k++) {
            SIND_TRY(p->sb[k].occ_edge_h.alloc(np * B)); SIND_TRY(p->sb[k].occ_total_h.alloc(np * B)); SIND_TRY(p->sb[k].occ_blocks_h.alloc(nblk * B));
            p->sb[k].occ_ev.assign(nch, nullptr);
            for (int c = 0; c < nch; c++) HIP_TRY(hipEventCreateWithFlags(&p->sb[k].occ_ev[c], hipEventDisableTiming));
            p->sb[k].occ2_ev.assign(B, nullptr);
            for (int f = 0; f < B; f++) HIP_TRY(hipEventCreateWithFlags(&p->sb[k].occ2_ev[f], hipEventDisableTiming));
            SIND_TRY(p->sb[k].grow_in_h.alloc((size_t)B * PG_IN_STRIDE)); SIND_TRY(p->sb[k].grow_member_h.alloc(np * B));
            SIND_TRY(p->sb[k].grow_pair_h.alloc((size_t)B * PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES)); SIND_TRY(p->sb[k].grow_status_h.alloc((size_t)4 * B));
            p->sb[k].grow_ev.assign(nch, nullptr); p->sb[k].grow_left.reset(new std::atomic<int>[nch]); p->sb[k].grow_state.reset(new std::atomic<int>[nch]);
            for (int c = 0; c < nch; c++) HIP_TRY(hipEventCreateWithFlags(&p->sb[k].grow_ev[c], hipEventDisableTiming));
        }
        SIND_TRY(make_stream(&p->grow_stream, !(sind_lab_env("SIND_OCC_PRIORITY") && atoi(sind_lab_env("SIND_OCC_PRIORITY")) == 0)));
        p->grow_ok = PeacGrowBatch::supports(cfg->width, cfg->height);
        if (p->grow_ok) SIND_TRY(p->grow.init(cfg->width, cfg->height, cfg->fx, cfg->fy, cfg->cx, cfg->cy, cfg->depth_scale, p->occ_chunk));
    }
    p->workers.start(nworkers, cfg->device, &p->gate, p->S == 1 ? 400 : 0);        // one stream: serial chains, idle host -- poll before sleeping (common.hpp)
    SIND_TRY(p->gray.alloc(np * std::max(B, 2)));          // sind_pipe_prime converts the two priming frames through this scratch, also when S * T == 1
    SIND_TRY(p->pool.alloc((size_t)p->fw * p->fh * p->S * (p->T + 2)));
    if (cfg->orb_gray_rgb_order) SIND_TRY(p->gray_orb.alloc(np * B));
    for (int k = 0; k < 2; k++) { SIND_TRY(p->sb[k].U.alloc(np * B)); SIND_TRY(p->sb[k].V.alloc(np * B)); SIND_TRY(p->sb[k].depth_dev.alloc(np * B)); SIND_TRY(p->sb[k].depth_h.alloc(np * B)); }
    p->primed.assign(p->S, 0);
    return SIND_OK;
}
int sind_pipe_create(const sind_pipe_config* cfg, sind_pipe** out) {
    if (!cfg || !out || cfg->streams < 1 || cfg->frames_per_step < 1 || cfg->width < 64 || cfg->height < 64) { sind_set_error("sind_pipe_create: bad configuration"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(cfg->device));
    sind_pipe* p = new sind_pipe();
    const int rc = pipe_build(p, cfg);
    if (rc != SIND_OK) { const std::string keep = sind_last_error(); sind_pipe_destroy(p); sind_set_error("%s", keep.c_str()); return rc; }      // streams, events and workers created so far go with it
    *out = p; return SIND_OK;
}
int sind_pipe_destroy(sind_pipe* p) {
    if (!p) return SIND_OK;
    if (getenv("SIND_TAIL_TIMING")) {
        double t[6] = {0}; long n = 0;
        for (auto& tl : p->tails) if (tl) { for (int i = 0; i < 6; i++) t[i] += tl->t_stage[i]; n += tl->n_frames; }       // (a handle whose creation failed half-way has empty slots)
        for (auto& tl : p->dtails) if (tl) for (int i = 0; i < 6; i++) t[i] += tl->t_stage[i];              // depth halves running ahead (their frames are counted by the flow halves)
        double f[40] = {0}; for (auto& tl : p->tails) if (tl) for (int i = 0; i < 40; i++) f[i] += tl->t_fine[i];
        for (auto& tl : p->dtails) if (tl) for (int i = 0; i < 40; i++) f[i] += tl->t_fine[i];
        for (auto& tl : p->occ_tails) if (tl) for (int i = 0; i < 40; i++) f[i] += tl->t_fine[i];
        if (n) fprintf(stderr, "[sind] cal_occluded: gpu+d2h %.2f pack %.2f endpoints %.2f peac %.2f contour-filter %.2f close %.2f | seg_merge: pieces %.2f sort+paint+pack %.2f alloc %.2f h2d-enqueue %.2f rag %.2f merge %.2f\n", f[0] / n, f[1] / n, f[2] / n, f[3] / n, f[4] / n, f[5] / n, f[6] / n, f[10] / n, f[11] / n, f[7] / n, f[8] / n, f[9] / n);
        if (n) fprintf(stderr, "[sind] pieces: open %.2f contours %.2f masks %.2f lianjie %.2f centre %.2f | flow_masks host: weights %.2f sort+wait %.2f homography %.2f pack %.2f | fusion: low %.2f clusters %.2f fill %.2f out+state %.2f\n",
                       f[12] / n, f[13] / n, f[14] / n, f[15] / n, f[16] / n, f[20] / n, f[21] / n, f[22] / n, f[23] / n, f[25] / n, f[26] / n, f[27] / n, f[28] / n);
        if (p->km_rounds) fprintf(stderr, "[sind] batched k-means: %.2f ms per round of %d frames (%ld rounds)\n", p->km_round_ms / p->km_rounds, p->S / std::max(1, p->km_groups), p->km_rounds);
        if (n) fprintf(stderr, "[sind] after the tail: dilate15 %.2f output copies %.2f orb mask filter %.2f\n", f[30] / n, f[31] / n, f[32] / n);
        if (g_cpu_steps.load()) fprintf(stderr, "[sind] phase-A thread CPU per step: flow slices %.1f ms, ORB thread %.1f ms (octree threads not included)\n", g_cpu_us_flow.load() / 1e3 / g_cpu_steps.load(), g_cpu_us_orb.load() / 1e3 / g_cpu_steps.load());
        if (n) fprintf(stderr, "[sind] stream waits: %.2f ms and %.1f calls per frame (occ + tail + batch stream)\n", g_sind_wait_ns.load() / 1e6 / n, (double)g_sind_wait_calls.load() / n);
        if (n) fprintf(stderr, "[sind] tail ms/frame over %ld frames: flow_masks %.2f kmeans %.2f labels %.2f cal_occluded %.2f seg_merge %.2f fusion %.2f\n", n, t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n);
    }
    (void)hipSetDevice(p->c.device);
    (void)hipDeviceSynchronize();
    for (std::thread& t : p->round_threads) if (t.joinable()) t.join();
    std::vector<hipStream_t> ss = p->worker_streams; ss.push_back(p->stream); ss.push_back(p->km_stream); for (hipStream_t k : p->km_streams) ss.push_back(k); ss.push_back(p->occ_stream); ss.push_back(p->grow_stream);
    for (auto& b : p->sb) { for (hipEvent_t e : b.occ_ev) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : b.occ2_ev) if (e) (void)hipEventDestroy(e); for (hipEvent_t e : b.grow_ev) if (e) (void)hipEventDestroy(e); }
    for (size_t w = 0; w < p->worker_streams_lo.size(); w++) if (w >= p->worker_streams.size() || p->worker_streams_lo[w] != p->worker_streams[w]) ss.push_back(p->worker_streams_lo[w]); ss.push_back(p->orb_stream); ss.insert(ss.end(), p->extra_streams.begin(), p->extra_streams.end());
    if (p->ev_pool) (void)hipEventDestroy(p->ev_pool);
    if (p->ev_gray) (void)hipEventDestroy(p->ev_gray);
    if (p->ev_depth) (void)hipEventDestroy(p->ev_depth);
    delete p;
    for (hipStream_t s : ss) if (s) (void)hipStreamDestroy(s);
    return SIND_OK;
}
int sind_pipe_prime(sind_pipe* p, int s, const uint8_t* last, const uint8_t* lastlast) {
    if (!p || s < 0 || s >= p->S || !last || !lastlast) { sind_set_error("sind_pipe_prime: bad arguments"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    const size_t np = (size_t)p->c.width * p->c.height, fb = (size_t)p->fw * p->fh;
    SIND_TRY(p->bgr_d.alloc(np * 3 * 2));
    HIP_TRY(hipMemcpyAsync(p->bgr_d.p, lastlast, np * 3, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->bgr_d.p + np * 3, last, np * 3, hipMemcpyHostToDevice, p->stream));
    SIND_TRY(p->front.gray_and_min(p->bgr_d.p, 2, p->gray.p, p->pool.p + fb * (size_t)s * (p->T + 2)));   // slots 0 (n-2), 1 (n-1)
    HIP_TRY(sind_stream_wait(p->stream));
    p->tails[s]->reset(); if (!p->dtails.empty()) p->dtails[s]->reset(); p->primed[s] = 1;
    return SIND_OK;
}

// depth half of stream s: its own object when depth-ahead is (or was) on, else the stream's tail
static DynaTail* depth_half(sind_pipe* p, int s) { return !p->dtails.empty() ? p->dtails[s].get() : p->tails[s].get(); }
static int ensure_dtails(sind_pipe* p) {
    if (!p->dtails.empty()) return SIND_OK;
    std::vector<std::unique_ptr<DynaTail>> d(p->S); std::vector<uint8_t> st;
    for (int s = 0; s < p->S; s++) {
        d[s].reset(new DynaTail()); SIND_TRY(d[s]->init(p->dc, p->worker_streams[s % p->worker_streams.size()]));
        st.resize(p->tails[s]->state_bytes()); p->tails[s]->save_state(st.data(), false, true); d[s]->load_state(st.data(), false, true);      // the warm labels move over
        d[s]->piece_threads = p->tails[s]->piece_threads;
    }
    p->dtails.swap(d); return SIND_OK;
}

// ---- phase A of one step (state free, batched over S*T frames, shared HIP stream): fills a StepBuf
static void depth_task(sind_pipe* p, sind_pipe::StepBuf* sb, int k, int worker);
static int phase_a(sind_pipe* p, sind_pipe::StepBuf& sb, const uint8_t* bgr_dev, const uint16_t* depth_dev, double t[4], bool depth_ahead = false) {
    const int S = p->S, T = p->T, B = S * T, W = p->c.width, H = p->c.height;
    const size_t np = (size_t)W * H, fb = (size_t)p->fw * p->fh;
    t[0] = now_ms();
    SindRange range_a("sind phase A (state-free: gray, dense flow, ORB front, CalOccluded)");
    const uint8_t* gray_for_orb = p->gray.p;
    {
    SindRange range_front("sind front: gray + 0.6 resize");
    // gray for all frames, 0.6-scaled gray into the per-stream pools behind the two history slots
    SIND_TRY(launch_bgr2gray(p->stream, bgr_dev, p->gray.p, np * B, false));
    // frame t of stream s goes to pool slot s * (T + 2) + 2 + t: one launch, T frames per group, two history slots skipped between the groups
    SIND_TRY(launch_resize_u8(p->stream, p->gray.p, p->pool.p + fb * 2, W, H, p->fw, p->fh, B, W, p->fw, np, fb, T, 2));
    if (p->c.orb_gray_rgb_order) { SIND_TRY(launch_bgr2gray(p->stream, bgr_dev, p->gray_orb.p, np * B, true)); gray_for_orb = p->gray_orb.p; }
    // ORB front (pyramid, FAST, octree, orientation, blur, BRIEF) of all frames: independent of the flow, so it runs on its own HIP
    // stream and host thread underneath the dense flow instead of after it
    HIP_TRY(hipEventRecord(p->ev_gray, p->stream));
    }
    // private copies of the depth frames: device (tail kernels of this step run while the caller may reuse its buffer) and host.  They go
    // ahead of the ORB front on its stream (157 MB to the host, ~3 ms): the flow slices need not wait for them, only the CalOccluded tasks do
    HIP_TRY(hipMemcpyAsync(sb.depth_dev.p, depth_dev, np * B * sizeof(uint16_t), hipMemcpyDeviceToDevice, p->orb_stream));
    HIP_TRY(hipMemcpyAsync(sb.depth_h.data(), sb.depth_dev.p, np * B * sizeof(uint16_t), hipMemcpyDeviceToHost, p->orb_stream));
    HIP_TRY(hipEventRecord(p->ev_depth, p->orb_stream));
    int orb_rc = SIND_OK; std::string orb_err;
    std::thread orb_thread([&] {
        (void)pthread_setname_np(pthread_self(), "sind-orb");
        (void)hipSetDevice(p->c.device);
        if (hipStreamWaitEvent(p->orb_stream, p->ev_gray, 0) != hipSuccess) { orb_rc = SIND_E_HIP; orb_err = "hipStreamWaitEvent failed"; return; }
        { SindRange r("sind ORB front: pyramid, FAST, octree, orientation, BRIEF"); orb_rc = p->orb.extract_all(gray_for_orb, B, sb.orb); }
        if (orb_rc != SIND_OK) orb_err = sind_last_error();
        g_cpu_us_orb += (long long)(thread_cpu_ms() * 1e3); });
    struct OrbJoin { std::thread& t; ~OrbJoin() { if (t.joinable()) t.join(); } } orb_join{orb_thread};
    // CalOccluded of every frame (state free: depth only) on the streams' own host threads / HIP streams, concurrent with the
    // dense flow below (the host cores would otherwise idle while the GPU runs the flow solver)
    sb.occ.assign(B, OccResult());
    SIND_TRY(sb.occ2_dev.alloc(np * B)); SIND_TRY(sb.depthN_dev.alloc(np * B));
    for (int k = 0; k < B; k++) { sb.occ[k].occ2_dev = sb.occ2_dev.p + np * k; sb.occ[k].depthN_dev = sb.depthN_dev.p + np * k; }
    sb.occ_rc.assign(B, SIND_OK); sb.occ_err.assign(B, std::string());
    if (p->batch_occ) {                                   // GPU half of CalOccluded, chunk by chunk, behind the depth copies
        HIP_TRY(hipStreamWaitEvent(p->occ_stream, p->ev_depth, 0));
        const size_t nblk = (size_t)(W / 16) * (H / 16);
        for (int c0 = 0, c = 0; c0 < B; c0 += p->occ_chunk, c++) {
            const int nb = std::min(p->occ_chunk, B - c0);
            SIND_TRY(p->occb.run(p->occ_stream, sb.depth_dev.p + np * c0, nb, sb.depthN_dev.p + np * c0, sb.occ_edge_h.data() + np * c0, sb.occ_total_h.data() + np * c0,
                                 sb.occ_blocks_h.data() + nblk * c0));
            HIP_TRY(hipEventRecord(sb.occ_ev[c], p->occ_stream));
        }
    }
    struct Waiter { TaskGroup& g; ~Waiter() { WorkerPool::wait(g); } };       // no task may outlive this call's buffers on an error return
    Waiter depth_waiter{sb.depth_group}, waiter{sb.occ_group};                // destroyed in reverse order: CalOccluded runners first (they open the last gates), then the depth chains
    sb.depth_ahead = depth_ahead;
    if (depth_ahead) {
        sb.dout.assign(B, DepthStageOut()); sb.depth_rc.assign(B, SIND_OK); sb.depth_err.assign(B, std::string());
        if (!sb.gate) sb.gate.reset(new std::atomic<int>[B]);
        for (int k = 0; k < B; k++) sb.gate[k].store(k % T == 0 ? 1 : 0);       // the first frame of a stream only waits for its CalOccluded
    }
    // `occ_workers` runner tasks share the frames through a counter: CalOccluded is host-heavy (PEAC region grow), and more runnable
    // threads than the CPU quota of the box (cgroup cpu.max, 16 cores per GPU) only burn the quota early in a period and stall EVERY
    // thread of the process, the flow's launch threads included, until the period ends
    sb.occ_next.store(0); sb.occ_next2.store(0); sb.grow_q = p->grow_ok ? p->grow_q : 0;
    if (p->batch_occ) {
        sb.occ_ctx.clear(); sb.occ_ctx.resize(B);
        const int nch = (B + p->occ_chunk - 1) / p->occ_chunk;
        for (int c = 0; c < nch; c++) { sb.grow_left[c].store(std::min(p->occ_chunk, B - c * p->occ_chunk)); sb.grow_state[c].store(0); }
    }
    auto push_occ = [&] { for (int r = 0; r < std::min(p->occ_workers, B); r++) p->workers.push(sb.occ_group, [p, &sb, B, np](int w) {
        auto done = [&](int k, int rc) {          // frame k has its CalOccluded result (or its error): open the depth chain's gate
            if (rc != SIND_OK) { sb.occ_rc[k] = rc; sb.occ_err[k] = sind_last_error(); }
            if (sb.depth_ahead && sb.gate[k].fetch_add(1) == 1) { sind_pipe::StepBuf* sbp = &sb; p->workers.push(sb.depth_group, [p, sbp, k](int w2) { depth_task(p, sbp, k, w2); }); }
        };
        if (!p->batch_occ) {
            for (int k; (k = sb.occ_next.fetch_add(1)) < B;) done(k, p->occ_tails[w]->compute_occluded(sb.depth_h.data() + np * k, sb.depth_dev.p + np * k, sb.occ[k]));
            return;
        }
        const size_t nblk = (size_t)(p->c.width / 16) * (p->c.height / 16), PP = (size_t)PEAC_GROW_MAX_PLANES * PEAC_GROW_MAX_PLANES;
        auto pre_of = [&](int k) { return OccGpuOut{sb.occ_edge_h.data() + np * k, sb.occ_total_h.data() + np * k, sb.occ_blocks_h.data() + nblk * k,
                                                    sb.occ_edge_h.data() + np * k /* the edge image has been packed by then */, sb.occ2_ev[k]}; };
        // first halves: GPU stencil results of the frame's chunk -> end points, PEAC graph clustering, the grow's input block; the runner that completes a chunk
        // enqueues its region grow (one launch for the chunk's frames; the launches share one device workspace, hence the lock around the enqueue)
        for (int k; (k = sb.occ_next.fetch_add(1)) < B;) {
            SindRange r("sind CalOccluded, first half: end points, PEAC graph");
            const int ch = k / p->occ_chunk; int rc = SIND_OK;
            if (sind_event_wait(sb.occ_ev[ch]) != hipSuccess) { (void)hipGetLastError(); sind_set_error("batched CalOccluded stage failed"); rc = SIND_E_HIP; }
            else {
                const OccGpuOut pre = pre_of(k); uint8_t* block = sb.grow_in_h.p + (size_t)k * PG_IN_STRIDE;
                const bool on_gpu = ((k + 1) * sb.grow_q) / 4 > (k * sb.grow_q) / 4;          // grow_q of every four frames
                rc = p->occ_tails[w]->compute_occluded_p1(sb.depth_h.data() + np * k, sb.depth_dev.p + np * k, sb.occ_ctx[k], &pre, on_gpu ? block : nullptr, k);
                if (!on_gpu) { const PeacGrowHdr skip{0, 0, 1, k}; std::memcpy(block, &skip, sizeof(skip)); }      // grown on the host in the first half: the kernel passes it over
            }
            if (rc != SIND_OK) { sb.occ_rc[k] = rc; sb.occ_err[k] = sind_last_error(); PeacGrowHdr skip{0, 0, 1, k}; std::memcpy(sb.grow_in_h.p + (size_t)k * PG_IN_STRIDE, &skip, sizeof(skip)); }
            if (sb.grow_left[ch].fetch_sub(1) == 1) {
                const int c0 = ch * p->occ_chunk, nb = std::min(p->occ_chunk, B - c0); int lrc = SIND_OK;
                if (sb.grow_q == 0) { sb.grow_state[ch].store(2); continue; }         // every frame of the chunk grew on the host: nothing to launch (state 2)
                { std::lock_guard<std::mutex> lk(p->grow_mu);
                  lrc = p->grow.run(p->grow_stream, sb.grow_in_h.p + (size_t)c0 * PG_IN_STRIDE, sb.depth_dev.p, nb, sb.grow_member_h.p + np * c0, sb.grow_pair_h.p + PP * c0, sb.grow_status_h.p + 4 * c0);
                  if (lrc == SIND_OK && hipEventRecord(sb.grow_ev[ch], p->grow_stream) != hipSuccess) lrc = SIND_E_HIP; }
                sb.grow_state[ch].store(lrc == SIND_OK ? 1 : -1);
            }
        }
        // second halves, in frame order: wait for the chunk's grow, then PEAC's last merge, plane contours, contour filter, closing
        for (int k; (k = sb.occ_next2.fetch_add(1)) < B;) {
            SindRange r("sind CalOccluded, second half: plane contours, contour filter");
            const int ch = k / p->occ_chunk;
            // grow_state[ch] is set behind EVERY first half of the chunk (grow_left): only after this wait is frame k's occ_rc final
            if (sb.grow_state[ch].load() == 0) { SindTokenPause pause; while (sb.grow_state[ch].load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(50)); }
            int rc = sb.occ_rc[k];
            if (rc == SIND_OK) {
                const int gs = sb.grow_state[ch].load();
                if (gs < 0 || (gs == 1 && sind_event_wait(sb.grow_ev[ch]) != hipSuccess)) { (void)hipGetLastError(); sind_set_error("PEAC region grow (chunk %d) failed", ch); rc = SIND_E_HIP; }
                else { const OccGpuOut pre = pre_of(k); rc = p->occ_tails[w]->compute_occluded_p2(sb.occ_ctx[k], sb.grow_member_h.p + np * k, sb.grow_pair_h.p + PP * k, sb.grow_status_h.p + 4 * k, sb.occ[k], &pre); }
                if (rc != SIND_OK) { sb.occ_rc[k] = rc; sb.occ_err[k] = sind_last_error(); }
            }
            sb.occ_ctx[k] = OccCtx();
            done(k, SIND_OK);              // (an error of this frame has been recorded above)
        } }); };
    t[1] = now_ms();
    // dense flow for every (n, n-2) pair, second pass for large-motion pairs, refinement, up-scale
    std::vector<int> cur(B), p1(B), p2(B);
    for (int s = 0; s < S; s++) for (int tt = 0; tt < T; tt++) { const int k = s * T + tt, base = s * (T + 2) + tt; cur[k] = base + 2; p1[k] = base + 1; p2[k] = base; }
    {   // the batch may be cut into slices that run the whole flow pyramid concurrently on their own streams (SIND_FLOW_SPLIT):
        // launches of different slices overlap on the GPU, so one slice's load phase can hide under another slice's iterations
        const int nsl = 1 + (int)p->extra_fronts.size(), Bs = (B + nsl - 1) / nsl;
        const size_t gsz = (size_t)2 * ((W - 1) / 10) * ((H - 1) / 10);
        SIND_TRY(sb.grid_dev.alloc(gsz * B)); SIND_TRY(sb.grid_h.alloc(gsz * B));
        std::vector<DynaFront*> fr(1, &p->front); for (auto& f : p->extra_fronts) fr.push_back(f.get());
        HIP_TRY(hipEventRecord(p->ev_pool, p->stream));
        std::vector<int> rc(nsl, SIND_OK); std::vector<std::string> er(nsl); std::vector<std::thread> th;
        std::vector<double> slice_ms(nsl, 0.0), slice_other_ms(nsl, 0.0); std::vector<std::vector<std::pair<double, double>>> slice_iv(nsl);
        auto run = [&](int i) {
            const int b0 = i * Bs, nb = std::min(Bs, B - b0); if (nb <= 0) return;
            SindRange r("sind dense flow slice: DeepFlow, large-motion pass, refinement, up-scale");
            DynaFront& f = *fr[i]; f.flow.sor_timer.enabled = true; f.flow.sor_timer.reset();
            if (i > 0 && hipStreamWaitEvent(f.stream, p->ev_pool, 0) != hipSuccess) { rc[i] = SIND_E_HIP; er[i] = "hipStreamWaitEvent failed"; return; }
            rc[i] = f.dense_flow(p->pool.p, cur.data() + b0, p1.data() + b0, p2.data() + b0, nb, sb.U.p + np * b0, sb.V.p + np * b0, nullptr);
            if (rc[i] == SIND_OK) {       // sample grid of the slice's frames for the tails' PROSAC pairs: one launch + one copy instead of one each per frame
                rc[i] = launch_gather_grid(f.stream, sb.U.p + np * b0, sb.V.p + np * b0, sb.grid_dev.p + gsz * b0, W, H, 10, nb);
                if (rc[i] == SIND_OK && hipMemcpyAsync(sb.grid_h.p + gsz * b0, sb.grid_dev.p + gsz * b0, gsz * nb * sizeof(float), hipMemcpyDeviceToHost, f.stream) != hipSuccess) rc[i] = SIND_E_HIP;
            }
            if (rc[i] == SIND_OK && sind_stream_wait(f.stream) != hipSuccess) rc[i] = SIND_E_HIP;
            if (rc[i] != SIND_OK) er[i] = sind_last_error();
            // the slice reads its own event brackets (three hipEventElapsedTime per bracket, ~150 brackets) while the other slices still run
            else { slice_ms[i] = f.flow.sor_timer.collect_ms(0); slice_other_ms[i] = f.flow.sor_timer.collect_ms(1); f.flow.sor_timer.intervals(p->ev_pool, slice_iv[i], 0); }
        };
        for (int i = 0; i < nsl; i++) th.emplace_back([&, i] { (void)pthread_setname_np(pthread_self(), "sind-flow"); (void)hipSetDevice(p->c.device); run(i); g_cpu_us_flow += (long long)(thread_cpu_ms() * 1e3); });
        g_cpu_steps++;
        // the slices are on their way: wait for the depth copies and start the CalOccluded tasks from here
        const hipError_t depth_ok = sind_event_wait(p->ev_depth);
        if (depth_ok == hipSuccess) push_occ();
        for (auto& slice_thread : th) slice_thread.join();
        if (depth_ok != hipSuccess) { (void)hipGetLastError(); sind_set_error("copy of the depth frames failed"); return SIND_E_HIP; }
        for (int i = 0; i < nsl; i++) if (rc[i] != SIND_OK) { sind_set_error("dense flow slice %d: %s", i, er[i].c_str()); return rc[i]; }
        for (int k = 0; k < B; k++) sb.occ[k].gridFlow = sb.grid_h.p + gsz * k;
        p->sor_ms = 0; p->sor_bytes = 0; p->sor_launches = 0; p->sor_slices = nsl; p->sor_other_ms = 0; p->sor_other_bytes = 0; p->sor_other_launches = 0;
        std::vector<std::pair<double, double>> iv;
        for (int i = 0; i < nsl; i++) { p->sor_ms += slice_ms[i]; p->sor_bytes += fr[i]->flow.sor_timer.alg_bytes; p->sor_launches += fr[i]->flow.sor_timer.launches;
            p->sor_other_ms += slice_other_ms[i]; p->sor_other_bytes += fr[i]->flow.sor_timer.alg_bytes_other; p->sor_other_launches += fr[i]->flow.sor_timer.launches_other; iv.insert(iv.end(), slice_iv[i].begin(), slice_iv[i].end()); }
        // time during which at least one slice had solver launches in flight (union of the event-bracketed intervals of all slices)
        std::sort(iv.begin(), iv.end()); double un = 0, cs = 0, ce = -1;
        for (const auto& q : iv) { if (q.first > ce) { if (ce > cs) un += ce - cs; cs = q.first; ce = q.second; } else ce = std::max(ce, q.second); }
        if (ce > cs) un += ce - cs;
        p->sor_union_ms = un;
    }
    t[2] = now_ms();
    orb_thread.join();
    if (orb_rc != SIND_OK) { sind_set_error("ORB front: %s", orb_err.c_str()); return orb_rc; }
    // roll the gray history: the last two frames of every stream become slots 0, 1 (one launch; the flow grid is a multiple of 16 bytes for every
    // supported size -- width % 64 == 0 -- and the copy-per-stream path stays for anything else)
    if (fb % 16 == 0) SIND_TRY(launch_roll_history(p->stream, p->pool.p, S, T, fb));
    else for (int s = 0; s < S; s++) {
        uint8_t* base = p->pool.p + fb * (size_t)s * (T + 2);
        if (T >= 2) { HIP_TRY(hipMemcpyAsync(base, base + fb * T, fb * 2, hipMemcpyDeviceToDevice, p->stream)); }
        else { HIP_TRY(hipMemcpyAsync(base, base + fb, fb, hipMemcpyDeviceToDevice, p->stream)); HIP_TRY(hipMemcpyAsync(base + fb, base + fb * 2, fb, hipMemcpyDeviceToDevice, p->stream)); }
    }
    HIP_TRY(sind_stream_wait(p->stream));
    WorkerPool::wait(sb.occ_group);
    WorkerPool::wait(sb.depth_group);           // every chain has been started by now (a gate is opened from inside a running task of either group)
    for (int k = 0; k < B; k++) if (sb.occ_rc[k] != SIND_OK) { sind_set_error("stream %d (CalOccluded): %s", k / T, sb.occ_err[k].c_str()); return sb.occ_rc[k]; }
    if (depth_ahead) for (int k = 0; k < B; k++) if (sb.depth_rc[k] != SIND_OK) { sind_set_error("stream %d (depth stage): %s", k / T, sb.depth_err[k].c_str()); return sb.depth_rc[k]; }
    t[3] = now_ms();
    sb.active.swap(p->active_next); p->active_next.clear(); sb.first.clear();           // applies to this step only
    sb.retain_tag = p->retain_tag_next; p->retain_tag_next = -1;
    sb.state_hash.assign((size_t)2 * B, 0);
    sb.pending = true;
    return SIND_OK;
}

// ---- phase B of one step (stateful tails on the worker pool)
struct PipeOut { uint8_t *dyna, *label, *mask; sind_keypoint* kps; int cap; int* nkp; uint8_t* desc; };
// One task = one frame of one stream; it queues the stream's next frame when it is done.  Frames of a stream stay in order, and the
// pool always sees up to S runnable tasks, so the workers stay busy until the end of the phase (a task per stream left the second
// "round" of 32 streams on 24 workers half empty).
// Depth half of frame k = (stream s, frame t) of a synchronous step: runs while the dense flow is on the GPU, in frame order per stream
// (the k-means warm labels are the previous frame's merged labels).  It opens the gate of the stream's next frame when it is done.
static void depth_task(sind_pipe* p, sind_pipe::StepBuf* sb, int k, int worker) {
    for (;;) {
        const int T = p->T, s = k / T, t = k % T; const size_t np = (size_t)p->c.width * p->c.height;
        DynaTail* dt = depth_half(p, s); dt->stream = p->worker_streams_lo[worker];
        const int r = dt->depth_stage(sb->depth_h.data() + np * k, sb->depth_dev.p + np * k, &sb->occ[k], sb->dout[k]);
        if (r != SIND_OK) { sb->depth_rc[k] = r; sb->depth_err[k] = sind_last_error(); }
        if (!(t + 1 < T && sb->gate[k + 1].fetch_add(1) == 1)) return;
        if (p->S == 1) { k++; continue; }                    // one stream: the next frame's chain link right here (no hand-over to another worker)
        p->workers.push(sb->depth_group, [p, sb, k](int w) { depth_task(p, sb, k + 1, w); }); return;
    }
}
static bool tail_one(sind_pipe* p, sind_pipe::StepBuf* sb, PipeOut o, int s, int t, int worker, const KmFrameResult* km) {
    SindRange range_t("sind tail: flow masks, SegAndMerge, fusion, dilation, ORB mask filter");
    p->tails[s]->stream = p->worker_streams[worker];
    const int T = p->T, W = p->c.width, H = p->c.height; const size_t np = (size_t)W * H;
    static thread_local std::vector<uint8_t> dy, lb, dil;
    dy.resize(np); lb.resize(np); dil.resize(np);
    const int k = s * T + t;
    int r = (sb->depth_ahead || sb->two_chain) ? p->tails[s]->flow_stage(sb->U.p + np * k, sb->V.p + np * k, sb->dout[k], dy.data(), lb.data(), sb->occ[k].gridFlow)
                            : p->tails[s]->process(sb->depth_h.data() + np * k, sb->depth_dev.p + np * k, sb->U.p + np * k, sb->V.p + np * k, dy.data(), lb.data(), &sb->occ[k],
                                                   p->dtails.empty() ? nullptr : (p->dtails[s]->stream = p->worker_streams[worker], p->dtails[s].get()), km);
    if (r != SIND_OK) { sb->tail_rc[s] = r; sb->tail_err[s] = sind_last_error(); return false; }
    if (p->hashing) { sb->state_hash[2 * (size_t)k] = p->tails[s]->state_hash[0]; sb->state_hash[2 * (size_t)k + 1] = p->tails[s]->state_hash[1]; }
    double* tf = p->tails[s]->t_fine; double t0 = now_ms();
    dilate15_codes(dy.data(), W, H, dil.data());
    { const double t1 = now_ms(); tf[30] += t1 - t0; t0 = t1; }
    if (o.dyna) std::memcpy(o.dyna + np * k, dy.data(), np);
    if (o.label) std::memcpy(o.label + np * k, lb.data(), np);
    if (o.mask) std::memcpy(o.mask + np * k, dil.data(), np);
    { const double t1 = now_ms(); tf[31] += t1 - t0; t0 = t1; }
    std::vector<OrbKeyPoint> kk; std::vector<uint8_t> dd;
    p->orb.finish(sb->orb[k], dil.data(), W, kk, dd);
    { const double t1 = now_ms(); tf[32] += t1 - t0; t0 = t1; }
    if ((int)kk.size() > o.cap && (o.kps || o.desc)) { sb->tail_rc[s] = SIND_E_CAPACITY; sb->tail_err[s] = "keypoint capacity exceeded"; return false; }
    if (o.nkp) o.nkp[k] = (int)kk.size();
    if (o.kps) std::memcpy(o.kps + (size_t)k * o.cap, kk.data(), kk.size() * sizeof(sind_keypoint));
    if (o.desc) std::memcpy(o.desc + (size_t)k * o.cap * 32, dd.data(), dd.size());
    return true;
}
static void tail_task(sind_pipe* p, sind_pipe::StepBuf* sb, PipeOut o, int s, int t, int worker, const KmFrameResult* km = nullptr, bool chain = true) {
    struct Spin { int keep; explicit Spin(bool on) : keep(t_sind_spin_us) { if (on) t_sind_spin_us = 400; } ~Spin() { t_sind_spin_us = keep; } } spin(sb->few_chain);      // few chains, idle host: poll before sleeping (common.hpp)
    for (;;) {
        if (!tail_one(p, sb, o, s, t, worker, km) || !chain || t + 1 >= (sb->active.empty() ? p->T : std::min(p->T, sb->active[s]))) return;
        if (p->S == 1 || sb->few_chain) { t++; km = nullptr; continue; }      // one stream / a few chains: the next frame's chain link right here (no hand-over to another worker)
        p->workers.push(sb->tail_group, [p, sb, o, s, t](int w) { tail_task(p, sb, o, s, t + 1, w); }); return;
    }
}
// Two chains per stream for a handful of live streams (the slow runners of a repair): the depth chain -- k-means from the previous frame's merged labels, SegAndMerge; it
// needs nothing from the flow half -- runs ahead on the stream's depth-half object, the flow chain (flow masks, fusion, dilation, keypoint filter) follows frame by frame
// as soon as its frame's depth stage and the previous frame's flow stage are done.  A frame then costs max(depth, flow) instead of their sum (the in-order mode's schedule).
static void flow_chain(sind_pipe* p, sind_pipe::StepBuf* sb, PipeOut o, int s, int t, int t1, int worker) {
    struct Spin { int keep; Spin() : keep(t_sind_spin_us) { t_sind_spin_us = 400; } ~Spin() { t_sind_spin_us = keep; } } spin;
    for (;;) {
        if (!tail_one(p, sb, o, s, t, worker, nullptr) || t + 1 >= t1) return;
        if (sb->fgate[s * p->T + t + 1].fetch_add(1) != 1) return;          // the next frame's depth stage is still out: its completion starts the flow stage
        t++;
    }
}
static void depth_chain(sind_pipe* p, sind_pipe::StepBuf* sb, PipeOut o, int s, int t0, int t1, int worker) {
    struct Spin { int keep; Spin() : keep(t_sind_spin_us) { t_sind_spin_us = 400; } ~Spin() { t_sind_spin_us = keep; } } spin;
    const size_t np = (size_t)p->c.width * p->c.height;
    DynaTail* dt = p->dtails[s].get(); dt->stream = p->worker_streams_lo[worker];
    for (int t = t0; t < t1; t++) {
        const int k = s * p->T + t;
        const int r = dt->depth_stage(sb->depth_h.data() + np * k, sb->depth_dev.p + np * k, &sb->occ[k], sb->dout[k], nullptr);
        if (r != SIND_OK) { sb->dchain_rc[s] = r; sb->dchain_err[s] = sind_last_error(); return; }      // the flow chain of this stream stops at the frame before
        if (sb->fgate[k].fetch_add(1) == 1) p->workers.push(sb->tail_group, [p, sb, o, s, t, t1](int w) { flow_chain(p, sb, o, s, t, t1, w); });
    }
}
static void phase_b_start(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o) {
    const int S = p->S;
    sb.tail_rc.assign(S, SIND_OK); sb.tail_err.assign(S, std::string()); sb.dchain_rc.assign(S, SIND_OK); sb.dchain_err.assign(S, std::string());
    sind_pipe::StepBuf* sbp = &sb;
    // A step in which only a few streams have frames (the repair runs of the chunked sequence mode: the slow runners of a round, sind_pipe_replay) runs them as
    // per-stream chains with their own k-means launches instead of rounds: a round costs the batched k-means' ~60 dependent launches for every frame whatever the
    // batch, and its barrier makes every stream wait for the slowest tail -- with a handful of streams on an otherwise idle GPU the chains are about twice as fast.
    int nact = 0; for (int s = 0; s < S; s++) nact += (sb.first.empty() ? 0 : sb.first[s]) < (sb.active.empty() ? p->T : sb.active[s]);
    const bool few = (!sb.first.empty() || !sb.active.empty()) && nact <= p->chain_max_streams;
    sb.few_chain = few && S > 1; sb.two_chain = false;
    if (sb.few_chain && !sb.depth_ahead && ensure_dtails(p) == SIND_OK) {
        const int B = S * p->T;
        if (sb.fgate_n < B) { sb.fgate.reset(new std::atomic<int>[B]); sb.fgate_n = B; }
        sb.dout.assign(B, DepthStageOut()); sb.two_chain = true;
        for (int s = 0; s < S; s++) {
            const int t0 = sb.first.empty() ? 0 : sb.first[s], t1 = sb.active.empty() ? p->T : sb.active[s];
            if (t0 >= t1) continue;
            for (int t = t0; t < t1; t++) sb.fgate[s * p->T + t].store(t == t0 ? 1 : 0);
            p->workers.push(sb.tail_group, [p, sbp, o, s, t0, t1](int w) { depth_chain(p, sbp, o, s, t0, t1, w); });
        }
        return;
    }
    if (p->batch_km && !sb.depth_ahead && !few) {
        // rounds: frame t of every stream -- the batched k-means chain on its own stream, then the S tails of that frame on the pool
        sbp->km_groups = p->km_groups; for (int g = 0; g <= sbp->km_groups; g++) sbp->km_first[g] = (int)((long long)S * g / sbp->km_groups);
        for (int g = 0; g < sbp->km_groups; g++) p->round_threads.emplace_back([p, sbp, o, g] {
            (void)pthread_setname_np(pthread_self(), "sind-rounds"); (void)hipSetDevice(p->c.device);
            const size_t np = (size_t)p->c.width * p->c.height;
            const int s0 = sbp->km_first[g], ns = sbp->km_first[g + 1] - s0; std::vector<const uint8_t*> prev(ns);
            for (int t = 0; t < p->T; t++) {
                bool any = false;                                        // ragged / replayed steps: rounds in which no stream of the group has a frame are passed over
                for (int s = s0; s < s0 + ns && !any; s++) any = (sbp->first.empty() || t >= sbp->first[s]) && (sbp->active.empty() || t < sbp->active[s]);
                if (!any) continue;
                if (t > 0) WorkerPool::wait(sbp->km_tails[g]);           // this group's tails of frame t - 1 (their merged labels start this round's k-means)
                for (int s = 0; s < ns; s++) prev[s] = depth_half(p, s0 + s)->prev_km_labels();
                const double tk = now_ms();
                int rc;
                { SindRange range_km("sind round: batched k-means of frame t of one group of streams");
                  rc = p->kmb[g].run(sbp->depth_dev.p + np * ((size_t)s0 * p->T + t), np * p->T, ns, prev.data()); }
                { std::lock_guard<std::mutex> lk(p->km_stat_mu); p->km_round_ms += now_ms() - tk; p->km_rounds++; }
                if (rc != SIND_OK) { const std::string e = sind_last_error(); for (int s = s0; s < s0 + ns; s++) if (sbp->tail_rc[s] == SIND_OK) { sbp->tail_rc[s] = rc; sbp->tail_err[s] = "batched k-means: " + e; } return; }
                for (int s = s0; s < s0 + ns; s++) if (sbp->tail_rc[s] == SIND_OK && (sbp->first.empty() || t >= sbp->first[s]) && (sbp->active.empty() || t < sbp->active[s])) p->workers.push(sbp->km_tails[g], [p, sbp, o, s, t, g, s0](int w) { tail_task(p, sbp, o, s, t, w, &p->kmb[g].result(s - s0), false); });
            } });
        return;
    }
    for (int s = 0; s < S; s++) {
        const int t0 = sb.first.empty() ? 0 : sb.first[s], t1 = sb.active.empty() ? p->T : sb.active[s];
        if (t0 < t1) p->workers.push(sb.tail_group, [p, sbp, o, s, t0](int w) { tail_task(p, sbp, o, s, t0, w); });
    }
}
static void swap_phase_a_outputs(sind_pipe::StepBuf& sb, sind_pipe::Retained& r) {
    sb.U.swap(r.U); sb.V.swap(r.V); sb.grid_dev.swap(r.grid_dev); sb.depth_dev.swap(r.depth_dev); sb.depth_h.swap(r.depth_h); sb.grid_h.swap(r.grid_h);
    sb.occ2_dev.swap(r.occ2_dev); sb.depthN_dev.swap(r.depthN_dev); sb.orb.swap(r.orb); sb.occ.swap(r.occ);
}
static int phase_b_finish(sind_pipe* p, sind_pipe::StepBuf& sb) {
    for (std::thread& t : p->round_threads) if (t.joinable()) t.join();
    p->round_threads.clear();
    WorkerPool::wait(sb.tail_group); for (TaskGroup& g : sb.km_tails) WorkerPool::wait(g);
    sb.pending = false;
    p->last_hash = sb.state_hash;
    if (sb.retain_tag >= 0) {                       // keep this step's phase-A outputs: they change places with a reserve set of the same sizes
        if (p->spare.empty()) { sind_set_error("sind_pipe: no reserve left to retain step %d (sind_pipe_reserve_retained)", sb.retain_tag); sb.retain_tag = -1; return SIND_E_STATE; }
        std::unique_ptr<sind_pipe::Retained> r = std::move(p->spare.back()); p->spare.pop_back();
        swap_phase_a_outputs(sb, *r); r->tag = sb.retain_tag; sb.retain_tag = -1;
        for (OccResult& o : r->occ) o.occ2_event = nullptr;            // the uploads behind these events are long done; the events belong to the step buffer and are recorded again
        p->kept.push_back(std::move(r));
    }
    for (int s = 0; s < p->S; s++) if (sb.tail_rc[s] != SIND_OK) { sind_set_error("stream %d: %s", s, sb.tail_err[s].c_str()); return sb.tail_rc[s]; }
    for (int s = 0; s < p->S && s < (int)sb.dchain_rc.size(); s++) if (sb.dchain_rc[s] != SIND_OK) { sind_set_error("stream %d (depth chain): %s", s, sb.dchain_err[s].c_str()); return sb.dchain_rc[s]; }
    return SIND_OK;
}
static int phase_b(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o) { phase_b_start(p, sb, o); return phase_b_finish(p, sb); }

static int check_inputs(sind_pipe* p, const void* a, const void* b) {
    if (!p || !a || !b) { sind_set_error("sind_pipe: null input"); return SIND_E_ARG; }
    for (int s = 0; s < p->S; s++) if (!p->primed[s]) { sind_set_error("sind_pipe: stream %d was not primed", s); return SIND_E_STATE; }
    HIP_TRY(hipSetDevice(p->c.device));
    return SIND_OK;
}

// synchronous step: phase A then phase B
int sind_pipe_process_dev(sind_pipe* p, const uint8_t* bgr_dev, const uint16_t* depth_dev, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil,
                          sind_keypoint* kps, int cap, int* nkp, uint8_t* desc) {
    SIND_TRY(check_inputs(p, bgr_dev, depth_dev));
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_process: a submitted step is still pending, call sind_pipe_flush first"); return SIND_E_STATE; }
    p->gate.set_capacity(1 << 20);                 // no pool task is running between two calls: safe to re-base the token count
    double t[4]; const double t0 = now_ms();
    SIND_TRY(phase_a(p, p->sb[0], bgr_dev, depth_dev, t, p->depth_ahead));
    const PipeOut o{dyna, label, mask_dil, kps, cap, nkp, desc};
    SIND_TRY(phase_b(p, p->sb[0], o));
    const double t4 = now_ms();
    p->stage_ms[0] = t[1] - t[0]; p->stage_ms[1] = t[2] - t[1]; p->stage_ms[2] = t[3] - t[2]; p->stage_ms[3] = 0; p->stage_ms[4] = t4 - t[3]; p->stage_ms[5] = t4 - t0;
    grow_adapt(p, t[3] - t[2], t[3] - t0);             // synchronous step: phase A waits for CalOccluded only; its tails have the host to themselves afterwards
    return SIND_OK;
}

// pipelined step: phase A of THIS step runs while the tails of the PREVIOUS submitted step finish on the host threads.
// Outputs receive the previous step's results; *have_output tells whether there was one.
int sind_pipe_submit_dev(sind_pipe* p, const uint8_t* bgr_dev, const uint16_t* depth_dev, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil,
                         sind_keypoint* kps, int cap, int* nkp, uint8_t* desc, int* have_output) {
    SIND_TRY(check_inputs(p, bgr_dev, depth_dev));
    p->gate.set_capacity(p->cpu_tokens);
    const int prev = p->cur ^ 1;
    const bool has_prev = p->sb[prev].pending;
    if (have_output) *have_output = has_prev ? 1 : 0;
    const PipeOut o{dyna, label, mask_dil, kps, cap, nkp, desc};
    const double tb0 = now_ms(); double tb1 = tb0;
    if (has_prev) phase_b_start(p, p->sb[prev], o);            // queued ahead of this step's CalOccluded tasks
    double t[4]; const double t0 = now_ms();
    const int ra = phase_a(p, p->sb[p->cur], bgr_dev, depth_dev, t, p->depth_ahead);      // with depth-ahead the depth chain of this step runs next to the flow chain of the previous one
    int rb = SIND_OK; const double ta1 = now_ms();
    if (has_prev) { rb = phase_b_finish(p, p->sb[prev]); tb1 = now_ms(); }
    const double t4 = now_ms();
    if (rb != SIND_OK || ra != SIND_OK) {          // a failed step leaves nothing pending: the next call starts from a clean two-buffer state
        std::string keep = sind_last_error(); if (rb != SIND_OK && ra != SIND_OK) keep = "tails of the previous step failed, and so did phase A of this one: " + keep;
        p->sb[0].pending = p->sb[1].pending = false; p->cur = 0;
        sind_set_error("%s", keep.c_str());
        return rb != SIND_OK ? rb : ra;
    }
    // tail_wait_ms: how long this call still waited for the previous step's tails after its own phase A was done (0 = the tails are hidden)
    p->stage_ms[0] = t[1] - t[0]; p->stage_ms[1] = t[2] - t[1]; p->stage_ms[2] = t[3] - t[2]; p->stage_ms[3] = 0; p->tail_wait_ms = has_prev ? t4 - ta1 : 0; p->stage_ms[4] = tb1 - tb0; p->stage_ms[5] = t4 - t0;
    p->cur ^= 1;
    grow_adapt(p, (t[3] - t[2]) + p->tail_wait_ms, t4 - t0);
    return SIND_OK;
}
// drain: finish the last submitted step
int sind_pipe_flush(sind_pipe* p, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc, int* have_output) {
    if (!p) return SIND_E_ARG;
    HIP_TRY(hipSetDevice(p->c.device));
    const int prev = p->cur ^ 1;
    if (have_output) *have_output = p->sb[prev].pending ? 1 : 0;
    if (!p->sb[prev].pending) return SIND_OK;
    const PipeOut o{dyna, label, mask_dil, kps, cap, nkp, desc};
    const double t0 = now_ms();
    SIND_TRY(phase_b(p, p->sb[prev], o));
    p->stage_ms[4] = now_ms() - t0;
    return SIND_OK;
}

int sind_pipe_process(sind_pipe* p, const uint8_t* bgr, const uint16_t* depth, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil, sind_keypoint* kps,
                      int cap, int* nkp, uint8_t* desc) {
    if (!p || !bgr || !depth) { sind_set_error("sind_pipe_process: null input"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_process: a submitted step is still pending, call sind_pipe_flush first"); return SIND_E_STATE; }
    const double t_in = now_ms();
    const size_t np = (size_t)p->c.width * p->c.height, B = (size_t)p->S * p->T;
    SIND_TRY(p->bgr_d.alloc(np * 3 * B)); SIND_TRY(p->depth_d.alloc(np * B));
    // Upload from the caller's (pageable) buffers: the pool's workers copy 4 MB pieces into page-locked staging (two buffers each) and
    // queue the DMA on their own streams; 393 MB of a 256-pair step arrive in ~8 ms (stage_ms[3], "host_upload").
    struct Piece { uint8_t* dst; const uint8_t* src; size_t n; };
    std::vector<Piece> pieces; const size_t chunk = (size_t)4 << 20;
    auto cut = [&](void* d, const void* h, size_t n) { for (size_t o = 0; o < n; o += chunk) pieces.push_back({(uint8_t*)d + o, (const uint8_t*)h + o, std::min(chunk, n - o)}); };
    cut(p->bgr_d.p, bgr, np * 3 * B); cut(p->depth_d.p, depth, np * B * 2);
    const int nw = std::min<int>(p->workers.size(), 16);
    while ((int)p->upload_stage.size() < nw) { p->upload_stage.emplace_back(new PinnedBuf<uint8_t>()); SIND_TRY(p->upload_stage.back()->alloc(2 * chunk)); }
    std::atomic<size_t> next{0}; std::atomic<int> bad{0}; TaskGroup up;
    for (int k = 0; k < nw; k++) p->workers.push(up, [p, k, &pieces, &next, &bad](int w) {
        hipStream_t st = p->worker_streams[w]; uint8_t* stage = p->upload_stage[k]->p; int slot = 0;
        for (size_t i; (i = next.fetch_add(1)) < pieces.size(); slot ^= 1) {
            if (slot == 0 && sind_stream_wait(st) != hipSuccess) bad = 1;          // both staging buffers are free again
            std::memcpy(stage + slot * chunk, pieces[i].src, pieces[i].n);
            if (hipMemcpyAsync(pieces[i].dst, stage + slot * chunk, pieces[i].n, hipMemcpyHostToDevice, st) != hipSuccess) bad = 1;
        }
        if (sind_stream_wait(st) != hipSuccess) bad = 1;
    });
    WorkerPool::wait(up);
    if (bad) { (void)hipGetLastError(); sind_set_error("sind_pipe_process: host-to-device upload failed"); return SIND_E_HIP; }
    const double t_up = now_ms() - t_in;
    const int rc = sind_pipe_process_dev(p, p->bgr_d.p, p->depth_d.p, dyna, label, mask_dil, kps, cap, nkp, desc);
    p->stage_ms[3] = t_up; p->stage_ms[5] += t_up;
    return rc;
}

int sind_pipe_set_depth_ahead(sind_pipe* p, int on) {
    if (!p) return SIND_E_ARG;
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_set_depth_ahead: a submitted step is still pending"); return SIND_E_STATE; }
    HIP_TRY(hipSetDevice(p->c.device));
    if (on && !p->active_next.empty()) { sind_set_error("sind_pipe_set_depth_ahead: a ragged step is pending (sind_pipe_set_active_frames)"); return SIND_E_STATE; }
    if (on) SIND_TRY(ensure_dtails(p));
    p->depth_ahead = on != 0; return SIND_OK;
}

// ---- inter-frame state of one stream as a flat blob (DynaTail::save_state): lets a sequence continue on another handle / rank
size_t sind_pipe_state_bytes(sind_pipe* p) { return p && !p->tails.empty() ? p->tails[0]->state_bytes() : 0; }
int sind_pipe_get_state(sind_pipe* p, int s, uint8_t* buf, size_t n) {
    if (!p || s < 0 || s >= p->S || !buf || n < sind_pipe_state_bytes(p)) { sind_set_error("sind_pipe_get_state: bad arguments"); return SIND_E_ARG; }
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_get_state: a submitted step is still pending, call sind_pipe_flush first"); return SIND_E_STATE; }
    p->tails[s]->save_state(buf, true, false); depth_half(p, s)->save_state(buf, false, true);
    return SIND_OK;
}
int sind_pipe_set_state(sind_pipe* p, int s, const uint8_t* buf, size_t n) {
    if (!p || s < 0 || s >= p->S || !buf || n < sind_pipe_state_bytes(p)) { sind_set_error("sind_pipe_set_state: bad arguments"); return SIND_E_ARG; }
    if (!p->primed[s]) { sind_set_error("sind_pipe_set_state: prime the stream first (priming resets its state)"); return SIND_E_STATE; }
    // A submitted step whose tails have not run yet is fine (that is the hand-over point between ranks: phase A done, state arrives, flush),
    // unless its depth chain already ran ahead on the old warm labels
    for (int k = 0; k < 2; k++) if (p->sb[k].pending && p->sb[k].depth_ahead) { sind_set_error("sind_pipe_set_state: the depth chain of the pending step already ran (depth-ahead); set the state before submitting"); return SIND_E_STATE; }
    p->tails[s]->load_state(buf, true, false); depth_half(p, s)->load_state(buf, false, true);
    return SIND_OK;
}

// ---- chunked sequences: per-frame state fingerprints and ragged steps
int sind_pipe_set_state_hashing(sind_pipe* p, int on) {
    if (!p) { sind_set_error("sind_pipe_set_state_hashing: null handle"); return SIND_E_ARG; }
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_set_state_hashing: a submitted step is still pending"); return SIND_E_STATE; }
    p->hashing = on != 0;
    for (auto& t : p->tails) t->hash_state = p->hashing;
    return SIND_OK;
}
int sind_pipe_get_state_hashes(sind_pipe* p, uint64_t* out, size_t count) {
    if (!p || !out || count < (size_t)2 * p->S * p->T) { sind_set_error("sind_pipe_get_state_hashes: need room for 2 x streams x frames_per_step values"); return SIND_E_ARG; }
    if (!p->hashing || p->last_hash.size() != (size_t)2 * p->S * p->T) { sind_set_error("sind_pipe_get_state_hashes: state hashing is off or no step has finished yet"); return SIND_E_STATE; }
    std::memcpy(out, p->last_hash.data(), p->last_hash.size() * sizeof(uint64_t)); return SIND_OK;
}
int sind_pipe_set_active_frames(sind_pipe* p, const int* frames_per_stream) {
    if (!p) { sind_set_error("sind_pipe_set_active_frames: null handle"); return SIND_E_ARG; }
    if (!frames_per_stream) { p->active_next.clear(); return SIND_OK; }
    if (p->depth_ahead) { sind_set_error("sind_pipe_set_active_frames: not available with depth-ahead (the depth chain runs ahead of the flow chain)"); return SIND_E_STATE; }
    for (int s = 0; s < p->S; s++) if (frames_per_stream[s] < 0 || frames_per_stream[s] > p->T) { sind_set_error("sind_pipe_set_active_frames: stream %d: %d is not in 0..%d", s, frames_per_stream[s], p->T); return SIND_E_ARG; }
    p->active_next.assign(frames_per_stream, frames_per_stream + p->S);
    return SIND_OK;
}

// ---- retained steps: phase-A outputs kept for a later replay of the stateful tails
int sind_pipe_reserve_retained(sind_pipe* p, int steps) {
    if (!p || steps < 0 || steps > 64) { sind_set_error("sind_pipe_reserve_retained: 0..64 steps"); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    const size_t np = (size_t)p->c.width * p->c.height, B = (size_t)p->S * p->T, gsz = (size_t)2 * ((p->c.width - 1) / 10) * ((p->c.height - 1) / 10);
    // the reserve is `steps` sets, not "at least": spare sets beyond it are freed (their HBM and page-locked host memory go back), so that a caller whose larger request
    // failed half-way can retry with a smaller one and really get the difference back (kept sets hold results and are never freed here: release them first)
    while (!p->spare.empty() && (int)(p->spare.size() + p->kept.size()) > steps) p->spare.pop_back();
    while ((int)(p->spare.size() + p->kept.size()) < steps) {
        std::unique_ptr<sind_pipe::Retained> r(new sind_pipe::Retained());
        SIND_TRY(r->U.alloc(np * B)); SIND_TRY(r->V.alloc(np * B)); SIND_TRY(r->grid_dev.alloc(gsz * B)); SIND_TRY(r->depth_dev.alloc(np * B)); SIND_TRY(r->depth_h.alloc(np * B));
        SIND_TRY(r->grid_h.alloc(gsz * B)); SIND_TRY(r->occ2_dev.alloc(np * B)); SIND_TRY(r->depthN_dev.alloc(np * B));
        p->spare.push_back(std::move(r));
    }
    return SIND_OK;
}
int sind_pipe_retain_next(sind_pipe* p, int tag) {
    if (!p || tag < 0) { sind_set_error("sind_pipe_retain_next: tag must be >= 0"); return SIND_E_ARG; }
    if (p->depth_ahead) { sind_set_error("sind_pipe_retain_next: not available with depth-ahead"); return SIND_E_STATE; }
    for (auto& r : p->kept) if (r->tag == tag) { sind_set_error("sind_pipe_retain_next: tag %d is in use", tag); return SIND_E_STATE; }
    if (p->spare.empty()) { sind_set_error("sind_pipe_retain_next: no reserve left (sind_pipe_reserve_retained)"); return SIND_E_STATE; }
    p->retain_tag_next = tag; return SIND_OK;
}
int sind_pipe_release_retained(sind_pipe* p, int tag) {
    if (!p) return SIND_E_ARG;
    for (size_t i = 0; i < p->kept.size();) if (tag < 0 || p->kept[i]->tag == tag) { p->kept[i]->tag = -1; p->spare.push_back(std::move(p->kept[i])); p->kept.erase(p->kept.begin() + i); } else i++;
    return SIND_OK;
}
// The stateful tails of a retained step again: stream s runs frames [first[s], last[s]) of that step from whatever state it holds now (sind_pipe_set_state).
// Outputs as in sind_pipe_process ([S][T] layout, only the frames that ran are written); fingerprints through sind_pipe_get_state_hashes.
int sind_pipe_replay(sind_pipe* p, int tag, const int* first, const int* last, uint8_t* dyna, uint8_t* label, uint8_t* mask_dil, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc) {
    if (!p || !first || !last) { sind_set_error("sind_pipe_replay: null argument"); return SIND_E_ARG; }
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_replay: a submitted step is still pending, call sind_pipe_flush first"); return SIND_E_STATE; }
    if (p->depth_ahead) { sind_set_error("sind_pipe_replay: not available with depth-ahead"); return SIND_E_STATE; }
    sind_pipe::Retained* r = nullptr; for (auto& k : p->kept) if (k->tag == tag) r = k.get();
    if (!r) { sind_set_error("sind_pipe_replay: no retained step with tag %d", tag); return SIND_E_ARG; }
    for (int s = 0; s < p->S; s++) if (first[s] < 0 || last[s] > p->T || (last[s] > first[s] && !p->primed[s])) { sind_set_error("sind_pipe_replay: stream %d: bad frame range %d..%d", s, first[s], last[s]); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(p->c.device));
    p->gate.set_capacity(1 << 20);
    sind_pipe::StepBuf& sb = p->sb[0];
    swap_phase_a_outputs(sb, *r);
    sb.first.assign(first, first + p->S); sb.active.assign(last, last + p->S); sb.depth_ahead = false; sb.retain_tag = -1;
    sb.state_hash.assign((size_t)2 * p->S * p->T, 0);
    const PipeOut o{dyna, label, mask_dil, kps, cap, nkp, desc};
    const double t0 = now_ms();
    const int rc = phase_b(p, sb, o);
    p->stage_ms[4] = now_ms() - t0;
    sb.first.clear(); sb.active.clear();
    swap_phase_a_outputs(sb, *r);
    return rc;
}

int sind_pipe_set_grow_share(sind_pipe* p, int quarters) {
    if (!p || quarters > 4) { sind_set_error("sind_pipe_set_grow_share: quarters must be -1 (adaptive) or 0..4"); return SIND_E_ARG; }
    p->grow_q_fixed = quarters < 0 ? -1 : quarters; if (quarters >= 0) p->grow_q = quarters;
    return SIND_OK;
}
int sind_pipe_get_grow_share(sind_pipe* p, int* quarters) { if (!p || !quarters) return SIND_E_ARG; *quarters = p->grow_q; return SIND_OK; }
int sind_pipe_set_kmeans_groups(sind_pipe* p, int groups) {
    if (!p || groups < -1 || groups == 0 || groups > p->km_groups_max) { sind_set_error("sind_pipe_set_kmeans_groups: -1 (adaptive) or 1..%d groups for this handle", p ? p->km_groups_max : 0); return SIND_E_ARG; }
    p->km_groups_fixed = groups; if (groups > 0) p->km_groups = groups; return SIND_OK;
}
int sind_pipe_get_kmeans_groups(sind_pipe* p, int* groups) { if (!p || !groups) { sind_set_error("sind_pipe_get_kmeans_groups: null argument"); return SIND_E_ARG; } *groups = p->batch_km ? p->km_groups : 0; return SIND_OK; }
// several handles on one GPU (sindslam_amd.pipeline.PipelineGroup: the streams of a small step cut into independent pipelines whose launch chains interleave):
// each takes its part of the process's CPU share -- the tokens of its pool tasks and its CalOccluded runners follow
int sind_pipe_set_cpu_share(sind_pipe* p, int cores) {
    if (!p || cores < 1) { sind_set_error("sind_pipe_set_cpu_share: at least one core"); return SIND_E_ARG; }
    if (p->sb[0].pending || p->sb[1].pending) { sind_set_error("sind_pipe_set_cpu_share: a submitted step is still pending"); return SIND_E_STATE; }
    cores = std::min(cores, 16); p->cpu_share = cores; p->host_info[0] = cores;
    p->cpu_tokens_max = std::max(2, cores - 1); p->cpu_tokens_min = std::max(2, cores - 3); if (!p->cpu_tokens_fixed) p->cpu_tokens = p->cpu_tokens_min;
    p->host_info[2] = p->cpu_tokens_max;
    p->occ_workers = std::max(1, std::min(p->workers.size(), std::max(1, cores - 2)));
    return SIND_OK;
}
int sind_pipe_set_chain_max_streams(sind_pipe* p, int n) {
    if (!p || n < 0) { sind_set_error("sind_pipe_set_chain_max_streams: n >= 0"); return SIND_E_ARG; }
    p->chain_max_streams = n; return SIND_OK;
}
int sind_pipe_host_info(sind_pipe* p, int* out6) { if (!p || !out6) { sind_set_error("sind_pipe_host_info: null argument"); return SIND_E_ARG; } std::memcpy(out6, p->host_info, sizeof(p->host_info)); return SIND_OK; }
int sind_pipe_mask_bytes(sind_pipe* p, size_t* bytes) { if (!p || !bytes) { sind_set_error("sind_pipe_mask_bytes: null argument"); return SIND_E_ARG; } *bytes = (size_t)p->S * p->T * p->c.width * p->c.height; return SIND_OK; }
int sind_pipe_tail_wait_ms(sind_pipe* p, double* ms) { if (!p || !ms) return SIND_E_ARG; *ms = p->tail_wait_ms; return SIND_OK; }
int sind_pipe_sor_stats(sind_pipe* p, long long* launches, double* sum_ms, double* union_ms, double* alg_bytes, int* slices) {
    if (!p) return SIND_E_ARG;
    if (launches) *launches = p->sor_launches;
    if (sum_ms) *sum_ms = p->sor_ms;
    if (union_ms) *union_ms = p->sor_union_ms;
    if (alg_bytes) *alg_bytes = p->sor_bytes;
    if (slices) *slices = p->sor_slices;
    return SIND_OK;
}
int sind_pipe_sor_other_stats(sind_pipe* p, long long* launches, double* sum_ms, double* alg_bytes) {
    if (!p) return SIND_E_ARG;
    if (launches) *launches = p->sor_other_launches;
    if (sum_ms) *sum_ms = p->sor_other_ms;
    if (alg_bytes) *alg_bytes = p->sor_other_bytes;
    return SIND_OK;
}
int sind_pipe_stats(sind_pipe* p, double* stage_ms6, long long* sor_launches, double* sor_ms, double* sor_alg_bytes) {
    if (!p) return SIND_E_ARG;
    if (stage_ms6) std::memcpy(stage_ms6, p->stage_ms, sizeof(p->stage_ms));
    if (sor_launches) *sor_launches = p->sor_launches;
    if (sor_ms) *sor_ms = p->sor_ms;
    if (sor_alg_bytes) *sor_alg_bytes = p->sor_bytes;
    return SIND_OK;
}

}  // extern "C"
