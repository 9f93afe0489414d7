// sind_pipe: the step entry points, state blobs, state fingerprints, retained steps / replay, settings and statistics of the C ABI (see pipeline_impl.hpp).
#include "pipeline_impl.hpp"

extern "C" {

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
    p->cpu_tokens_max = std::max(2, cores + 2); p->cpu_tokens_min = std::max(2, cores - 3); if (!p->cpu_tokens_fixed) p->cpu_tokens = p->cpu_tokens_min;
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
