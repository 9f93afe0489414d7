// sind_pipe: construction -- streams, fronts, tails, worker pool, CPU share -- destruction and priming (see pipeline_impl.hpp).
#include "pipeline_impl.hpp"

extern "C" int sind_pipe_destroy(sind_pipe* p);

int pipe_build(sind_pipe* p, const sind_pipe_config* cfg) {
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
    p->front.flow.coarse_chain = !(cfg->flow_opts_off & 1); p->front.flow.latency_tiles = !(cfg->flow_opts_off & 2); p->front.flow.level_up = !(cfg->flow_opts_off & 4); p->front.flow.solver.wave = !(cfg->flow_opts_off & 8);
    p->split_rounds = !(cfg->flow_opts_off & 16);
    if (cfg->flow_opts_off >> 8) p->front.flow.solver.wave_items = cfg->flow_opts_off >> 8;      // (experiment: bits 8.. = waves per launch the row bands of k_sor_wave are cut for)
    for (auto& f : p->extra_fronts) { f->flow.coarse_chain = p->front.flow.coarse_chain; f->flow.latency_tiles = p->front.flow.latency_tiles; f->flow.level_up = p->front.flow.level_up; f->flow.solver.wave = p->front.flow.solver.wave; f->flow.solver.wave_items = p->front.flow.solver.wave_items; }
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
    const int nworkers = sind_lab_env("SIND_WORKERS") ? std::max(2, atoi(sind_lab_env("SIND_WORKERS"))) : cfg->host_threads > 0 ? cfg->host_threads : std::max(2, cpu_share * 3);         // default: 3x the CPU share (workers sleep while they wait for the GPU: a tail waits ~6 ms of its ~11; 48 against 32 workers:
                                                                                                                                                                                        // 1280 x 720 790 - 818 -> 813 - 866 pairs/s, headline 1530 - 1577 -> 1548 - 1593, small steps unchanged, profiles/r05/pool_workers.txt)
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
    p->cpu_tokens_max = std::max(2, cpu_share + 2);      // (the controller only goes there while steps wait for the host: 1280 x 720, host-bound, 798 - 809 pairs/s with 15 tokens, 822 - 855 with 17, 836 - 857 with 20 -- a token is held
                                                              // through short waits too, so a few more tokens than cores keep the 16 cores of the quota busy: 13.6 -> 14.5 - 14.8; profiles/r05/cpu_tokens_720p.txt) p->cpu_tokens_min = std::max(2, cpu_share - 3); p->cpu_tokens = p->cpu_tokens_min;
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
DynaTail* depth_half(sind_pipe* p, int s) { return !p->dtails.empty() ? p->dtails[s].get() : p->tails[s].get(); }
int ensure_dtails(sind_pipe* p) {
    if (!p->dtails.empty()) return SIND_OK;
    std::vector<std::unique_ptr<DynaTail>> d(p->S); std::vector<uint8_t> st;
    for (int s = 0; s < p->S; s++) {
        d[s].reset(new DynaTail()); SIND_TRY(d[s]->init(p->dc, p->worker_streams[s % p->worker_streams.size()]));
        st.resize(p->tails[s]->state_bytes()); p->tails[s]->save_state(st.data(), false, true); d[s]->load_state(st.data(), false, true);      // the warm labels move over
        d[s]->piece_threads = p->tails[s]->piece_threads;
    }
    p->dtails.swap(d); return SIND_OK;
}

