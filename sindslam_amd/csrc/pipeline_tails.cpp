// sind_pipe: phase B of a step -- the stateful tails on the worker pool (see pipeline_impl.hpp).
#include "pipeline_impl.hpp"

// ---- phase B of one step (stateful tails on the worker pool)
// One task = one frame of one stream; it queues the stream's next frame when it is done.  Frames of a stream stay in order, and the
// pool always sees up to S runnable tasks, so the workers stay busy until the end of the phase (a task per stream left the second
// "round" of 32 streams on 24 workers half empty).
// Depth half of frame k = (stream s, frame t) of a synchronous step: runs while the dense flow is on the GPU, in frame order per stream
// (the k-means warm labels are the previous frame's merged labels).  It opens the gate of the stream's next frame when it is done.
void depth_task(sind_pipe* p, sind_pipe::StepBuf* sb, int k, int worker) {
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
static void flow_chain(sind_pipe* p, sind_pipe::StepBuf* sb, PipeOut o, int s, int t, int t1, int worker, bool poll = true) {
    struct Spin { int keep; explicit Spin(bool on) : keep(t_sind_spin_us) { if (on) t_sind_spin_us = 400; } ~Spin() { t_sind_spin_us = keep; } } spin(poll);      // (a few chains on an idle host poll before they sleep; the rounds of a full pipeline do not)
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
void phase_b_start(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o) {
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
        // rounds: frame t of every stream -- the batched k-means chain on its own stream, then the S tails of that frame on the pool.
        // split (default): what the NEXT round's k-means waits for is only the depth half of a tail (cluster order, SegAndMerge: its merged labels start that k-means); the flow
        // half (flow masks, fusion, dilation, keypoint filter) of frame t follows on the stream's tail object as soon as its depth half and the flow half of frame t - 1 are
        // done, beside the next round.  A chunk's frames are a serial chain: this takes ~a third of a frame's tail off it.
        // Measured (profiles/r05/ab_split_rounds.txt): 5 streams x 6 frames 562 - 582 -> 665 - 676 pairs/s, 8 x 4 745 - 758 -> 816 - 826, 16 x 2 and 32 x 2 unchanged, 128 x 4 within
        // noise on the losing side -- so only while there are no more streams than pool workers (with more, other streams' tails fill the pool during a round anyway).
        const bool split = p->split_rounds && S <= (int)p->worker_streams.size() && ensure_dtails(p) == SIND_OK;
        if (split) {
            const int B = S * p->T;
            if (sb.fgate_n < B) { sb.fgate.reset(new std::atomic<int>[B]); sb.fgate_n = B; }
            sb.dout.assign(B, DepthStageOut()); sb.two_chain = true;
            for (int s = 0; s < S; s++) { const int t0 = sb.first.empty() ? 0 : sb.first[s]; for (int t = 0; t < p->T; t++) sb.fgate[s * p->T + t].store(t == t0 ? 1 : 0); }
        }
        sbp->km_groups = p->km_groups; for (int g = 0; g <= sbp->km_groups; g++) sbp->km_first[g] = (int)((long long)S * g / sbp->km_groups);
        for (int g = 0; g < sbp->km_groups; g++) p->round_threads.emplace_back([p, sbp, o, g, split] {
            (void)pthread_setname_np(pthread_self(), "sind-rounds"); (void)hipSetDevice(p->c.device);
            const size_t np = (size_t)p->c.width * p->c.height;
            const int s0 = sbp->km_first[g], ns = sbp->km_first[g + 1] - s0; std::vector<const uint8_t*> prev(ns);
            for (int t = 0; t < p->T; t++) {
                bool any = false;                                        // ragged / replayed steps: rounds in which no stream of the group has a frame are passed over
                for (int s = s0; s < s0 + ns && !any; s++) any = (sbp->first.empty() || t >= sbp->first[s]) && (sbp->active.empty() || t < sbp->active[s]);
                if (!any) continue;
                if (t > 0) WorkerPool::wait(sbp->km_tails[g]);           // this group's tails (split: their depth halves) of frame t - 1: their merged labels start this round's k-means
                for (int s = 0; s < ns; s++) prev[s] = depth_half(p, s0 + s)->prev_km_labels();
                const double tk = now_ms();
                int rc;
                { SindRange range_km("sind round: batched k-means of frame t of one group of streams");
                  rc = p->kmb[g].run(sbp->depth_dev.p + np * ((size_t)s0 * p->T + t), np * p->T, ns, prev.data()); }
                { std::lock_guard<std::mutex> lk(p->km_stat_mu); p->km_round_ms += now_ms() - tk; p->km_rounds++; }
                if (rc != SIND_OK) { const std::string e = sind_last_error(); for (int s = s0; s < s0 + ns; s++) if (sbp->tail_rc[s] == SIND_OK) { sbp->tail_rc[s] = rc; sbp->tail_err[s] = "batched k-means: " + e; } return; }
                for (int s = s0; s < s0 + ns; s++) {
                    if (!(sbp->tail_rc[s] == SIND_OK && (sbp->first.empty() || t >= sbp->first[s]) && (sbp->active.empty() || t < sbp->active[s]))) continue;
                    if (!split) { p->workers.push(sbp->km_tails[g], [p, sbp, o, s, t, g, s0](int w) { tail_task(p, sbp, o, s, t, w, &p->kmb[g].result(s - s0), false); }); continue; }
                    p->workers.push(sbp->km_tails[g], [p, sbp, o, s, t, g, s0, np](int w) {
                        const int k = s * p->T + t, t1 = sbp->active.empty() ? p->T : std::min(p->T, sbp->active[s]);
                        DynaTail* dt = p->dtails[s].get(); dt->stream = p->worker_streams[w];      // (on the chain that the next round waits for: the pool's high-priority stream)
                        const int r = dt->depth_stage(sbp->depth_h.data() + np * k, sbp->depth_dev.p + np * k, &sbp->occ[k], sbp->dout[k], &p->kmb[g].result(s - s0));
                        if (r != SIND_OK) { sbp->dchain_rc[s] = r; sbp->dchain_err[s] = sind_last_error(); return; }      // the flow halves of this stream stop at the frame before
                        if (sbp->fgate[k].fetch_add(1) == 1) p->workers.push(sbp->tail_group, [p, sbp, o, s, t, t1](int w2) { flow_chain(p, sbp, o, s, t, t1, w2, false); });
                    });
                }
            } });
        return;
    }
    for (int s = 0; s < S; s++) {
        const int t0 = sb.first.empty() ? 0 : sb.first[s], t1 = sb.active.empty() ? p->T : sb.active[s];
        if (t0 < t1) p->workers.push(sb.tail_group, [p, sbp, o, s, t0](int w) { tail_task(p, sbp, o, s, t0, w); });
    }
}
void swap_phase_a_outputs(sind_pipe::StepBuf& sb, sind_pipe::Retained& r) {
    sb.U.swap(r.U); sb.V.swap(r.V); sb.grid_dev.swap(r.grid_dev); sb.depth_dev.swap(r.depth_dev); sb.depth_h.swap(r.depth_h); sb.grid_h.swap(r.grid_h);
    sb.occ2_dev.swap(r.occ2_dev); sb.depthN_dev.swap(r.depthN_dev); sb.orb.swap(r.orb); sb.occ.swap(r.occ);
}
int phase_b_finish(sind_pipe* p, sind_pipe::StepBuf& sb) {
    for (std::thread& t : p->round_threads) if (t.joinable()) t.join();
    p->round_threads.clear();
    for (TaskGroup& g : sb.km_tails) WorkerPool::wait(g);      // (first: with split rounds a depth half queues its frame's flow half into tail_group when it is done)
    WorkerPool::wait(sb.tail_group);
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
int phase_b(sind_pipe* p, sind_pipe::StepBuf& sb, const PipeOut& o) { phase_b_start(p, sb, o); return phase_b_finish(p, sb); }

