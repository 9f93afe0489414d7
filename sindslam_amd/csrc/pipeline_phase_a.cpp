// sind_pipe: phase A of a step -- the state-free batch over S x T frames (see pipeline_impl.hpp).
#include "pipeline_impl.hpp"

// ---- phase A of one step (state free, batched over S*T frames, shared HIP stream): fills a StepBuf
int phase_a(sind_pipe* p, sind_pipe::StepBuf& sb, const uint8_t* bgr_dev, const uint16_t* depth_dev, double t[4], bool depth_ahead) {
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

