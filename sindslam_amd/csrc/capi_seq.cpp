// C ABI of the chunked-sequence driver (include/sind_hip.h, "One long sequence"): host/seq.cpp's SeqDriver on two sind_pipe handles (the S x T main pipeline of this
// rank's chunks and the small repair pipeline), a frame source (host arrays of the whole sequence, or callbacks that hand out device batches) and a sink (caller arrays
// indexed by frame).  A C++ caller of rgbd_tum_noros gets the exact sharded mode -- speculate, verify the chunk seams by state fingerprints, replay / repair, hand the
// seam states from rank to rank -- without Python and without torch.distributed: the exchange runs over RCCL (sind_comm) or over TCP between the hosts.
#include <cstring>
#include <memory>
#include "../../include/sind_hip.h"
#include "common.hpp"
#include "host/seq.hpp"

using namespace sind;

namespace {

struct SeqSource {
    // host arrays of the whole sequence ...
    const uint8_t* bgr = nullptr; const uint16_t* depth = nullptr; long long n_frames = 0;
    // ... or callbacks
    sind_seq_batch_fn batch = nullptr; sind_seq_frame_fn frame = nullptr; void* user = nullptr;
    int W = 0, H = 0, device = 0;
    DevBuf<uint8_t> bgr_dev; DevBuf<uint16_t> depth_dev; hipStream_t up = nullptr;
    // sequence POSITION q (0 = the first frame DetectDynaArea processes) is array frame q + 1; positions -1 and -2 (the two priming frames of the sequential loop,
    // rgbd_tum_noros.cc:103-107) are both frame 0; positions past the end repeat the last frame (lock-step padding, results unowned)
    long long index(long long q) const { return std::max<long long>(0, std::min<long long>(q + 1, n_frames - 1)); }
    int host_frame(long long q, const uint8_t** out, std::string& err) {
        if (frame) { if (frame(user, q, out) != 0 || !*out) { err = "frame source callback failed"; return SIND_E_STATE; } return SIND_OK; }
        if (!bgr) { err = "sind_seq: no frame source set"; return SIND_E_STATE; }
        *out = bgr + (size_t)index(q) * W * H * 3; return SIND_OK;
    }
    int device_batch(const long long* pos, int count, const uint8_t** b, const uint16_t** d, std::string& err) {
        if (batch) { if (batch(user, pos, count, b, d) != 0 || !*b || !*d) { err = "batch source callback failed"; return SIND_E_STATE; } return SIND_OK; }
        if (!bgr || !depth) { err = "sind_seq: no frame source set"; return SIND_E_STATE; }
        const size_t np = (size_t)W * H;
        if (bgr_dev.alloc(np * 3 * count) != SIND_OK || depth_dev.alloc(np * count) != SIND_OK) { err = sind_last_error(); return SIND_E_ALLOC; }
        if (!up && hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return SIND_E_HIP; }
        for (int k = 0; k < count; k++) {
            const size_t f = (size_t)index(pos[k]);
            if (hipMemcpyAsync(bgr_dev.p + np * 3 * k, bgr + f * np * 3, np * 3, hipMemcpyHostToDevice, up) != hipSuccess ||
                hipMemcpyAsync(depth_dev.p + np * k, depth + f * np, np * 2, hipMemcpyHostToDevice, up) != hipSuccess) { err = "upload of a frame failed"; return SIND_E_HIP; }
        }
        if (sind_stream_wait(up) != hipSuccess) { err = "upload of a step's frames failed"; return SIND_E_HIP; }
        *b = bgr_dev.p; *d = depth_dev.p; return SIND_OK;
    }
    ~SeqSource() { if (up) (void)hipStreamDestroy(up); }
};
struct SeqSink { uint8_t *dyna = nullptr, *label = nullptr, *mask = nullptr; sind_keypoint* kps = nullptr; int cap = 0; int* nkp = nullptr; uint8_t* desc = nullptr; long long n_frames = 0; };

// SeqPipe on a sind_pipe handle
class RealPipe : public SeqPipe {
public:
    sind_pipe* p = nullptr; int S_ = 0, T_ = 0, W = 0, H = 0, cap = 0; SeqSource* src = nullptr; SeqSink* sink = nullptr; std::string err_;
    PinnedBuf<uint8_t> dyna, label, mask, desc; PinnedBuf<sind_keypoint> kps; PinnedBuf<int> nkp;
    int create(const sind_pipe_config& c) {
        S_ = c.streams; T_ = c.frames_per_step; W = c.width; H = c.height; cap = 2 * c.nfeatures + 256;
        SIND_TRY(sind_pipe_create(&c, &p));
        const size_t B = (size_t)S_ * T_, np = (size_t)W * H;
        SIND_TRY(dyna.alloc(B * np)); SIND_TRY(label.alloc(B * np)); SIND_TRY(mask.alloc(B * np)); SIND_TRY(kps.alloc(B * cap)); SIND_TRY(nkp.alloc(B)); SIND_TRY(desc.alloc(B * cap * 32));
        return SIND_OK;
    }
    ~RealPipe() override { if (p) (void)sind_pipe_destroy(p); }
    int bad(int rc) { if (rc != SIND_OK) err_ = sind_last_error(); return rc; }
    int S() const override { return S_; }
    int T() const override { return T_; }
    const char* error() const override { return err_.c_str(); }
    int prime(int s, long long a, long long b) override {
        const uint8_t *fa = nullptr, *fb = nullptr;
        if (src->host_frame(a, &fa, err_) || src->host_frame(b, &fb, err_)) return -1;
        return bad(sind_pipe_prime(p, s, fa, fb));
    }
    int set_state_hashing(bool on) override { return bad(sind_pipe_set_state_hashing(p, on ? 1 : 0)); }
    int submit(const long long* pos, bool* have) override {
        const uint8_t* b = nullptr; const uint16_t* d = nullptr; int h = 0;
        if (src->device_batch(pos, S_ * T_, &b, &d, err_)) return -1;
        const int rc = bad(sind_pipe_submit_dev(p, b, d, dyna.p, label.p, mask.p, kps.p, cap, nkp.p, desc.p, &h)); *have = h != 0; return rc;
    }
    int flush(bool* have) override { int h = 0; const int rc = bad(sind_pipe_flush(p, dyna.p, label.p, mask.p, kps.p, cap, nkp.p, desc.p, &h)); *have = h != 0; return rc; }
    int process(const long long* pos, const int* active) override {
        const uint8_t* b = nullptr; const uint16_t* d = nullptr;
        if (active && bad(sind_pipe_set_active_frames(p, active))) return -1;
        if (src->device_batch(pos, S_ * T_, &b, &d, err_)) return -1;
        return bad(sind_pipe_process_dev(p, b, d, dyna.p, label.p, mask.p, kps.p, cap, nkp.p, desc.p));
    }
    int state_hashes(uint64_t* out) override { return bad(sind_pipe_get_state_hashes(p, out, (size_t)S_ * T_ * 2)); }
    size_t state_bytes() override { return sind_pipe_state_bytes(p); }
    int get_state(int s, uint8_t* blob) override { return bad(sind_pipe_get_state(p, s, blob, sind_pipe_state_bytes(p))); }
    int set_state(int s, const uint8_t* blob) override { return bad(sind_pipe_set_state(p, s, blob, sind_pipe_state_bytes(p))); }
    int reserve_retained(int n) override { return bad(sind_pipe_reserve_retained(p, n)); }
    int retain_next(int tag) override { return bad(sind_pipe_retain_next(p, tag)); }
    int release_retained(int tag) override { return bad(sind_pipe_release_retained(p, tag)); }
    int replay(int tag, const int* first, const int* last) override { return bad(sind_pipe_replay(p, tag, first, last, dyna.p, label.p, mask.p, kps.p, cap, nkp.p, desc.p)); }
    // the last runners of a repair run as two chains per stream (depth half ahead of the flow half) on separate depth-half objects: switching depth-ahead on and off
    // again creates them now, outside any timed region, instead of at the first replay
    int warm_two_chain_mode() override { const int rc = bad(sind_pipe_set_depth_ahead(p, 1)); return rc ? rc : bad(sind_pipe_set_depth_ahead(p, 0)); }
    int emit(int s, int t, long long pos) override {
        const long long f = pos + 1;
        if (!sink || f < 0 || f >= sink->n_frames) return 0;
        const size_t np = (size_t)W * H, k = (size_t)s * T_ + t;
        if (sink->dyna) std::memcpy(sink->dyna + (size_t)f * np, dyna.p + k * np, np);
        if (sink->label) std::memcpy(sink->label + (size_t)f * np, label.p + k * np, np);
        if (sink->mask) std::memcpy(sink->mask + (size_t)f * np, mask.p + k * np, np);
        if (sink->nkp) {
            const int n = std::min(nkp.p[k], sink->cap); sink->nkp[f] = n;
            if (sink->kps) std::memcpy(sink->kps + (size_t)f * sink->cap, kps.p + k * cap, (size_t)n * sizeof(sind_keypoint));
            if (sink->desc) std::memcpy(sink->desc + (size_t)f * sink->cap * 32, desc.p + k * cap * 32, (size_t)n * 32);
        }
        return 0;
    }
};

// SeqNet on RCCL: the fingerprints travel in one all-gather (staged through a device buffer), a seam's state blob in one send / receive group (sind_comm_sendrecv_u8)
class RcclNet : public SeqNet {
public:
    sind_comm* c; std::string err_; DevBuf<uint8_t> all_dev;
    explicit RcclNet(sind_comm* cc) : c(cc) {}
    int rank() const override { return sind_comm_rank(c); }
    int world() const override { return sind_comm_world(c); }
    const char* error() const override { return err_.c_str(); }
    int allgather(const void* mine, size_t bytes, void* all) override {
        if (all_dev.alloc(bytes * (size_t)world()) != SIND_OK) { err_ = sind_last_error(); return -1; }
        if (sind_comm_allgather_u8(c, (const uint8_t*)mine, bytes, all_dev.p, (uint8_t*)all) != SIND_OK) { err_ = sind_last_error(); return -1; }
        return 0;
    }
    int sendrecv(const void* send, int to, void* recv, int from, size_t bytes) override {
        if (sind_comm_sendrecv_u8(c, (const uint8_t*)send, to, (uint8_t*)recv, from, bytes) != SIND_OK) { err_ = sind_last_error(); return -1; }
        return 0;
    }
};

}  // namespace

struct sind_seq_net { std::unique_ptr<SeqNet> net; };
struct sind_seq {
    sind_seq_config cfg; SeqPlan plan; SeqSource src; SeqSink sink; RealPipe main, repair; bool has_repair = false; std::unique_ptr<SeqDriver> drv; SeqNet* net = nullptr;
    sind_seq_hook_fn step_hook = nullptr, round_hook = nullptr; void* hook_user = nullptr; int rounds_seen = 0;
};

extern "C" {

int sind_seq_net_tcp(int rank, int world, const char* host, int base_port, sind_seq_net** out) {
    if (!out) return SIND_E_ARG;
    std::string err; SeqNet* n = seq_net_tcp(rank, world, host, base_port, err);
    if (!n) { sind_set_error("sind_seq_net_tcp: %s", err.c_str()); return SIND_E_ARG; }
    *out = new sind_seq_net(); (*out)->net.reset(n); return SIND_OK;
}
int sind_seq_net_rccl(sind_comm* c, sind_seq_net** out) {
    if (!c || !out) { sind_set_error("sind_seq_net_rccl: null argument"); return SIND_E_ARG; }
    *out = new sind_seq_net(); (*out)->net.reset(new RcclNet(c)); return SIND_OK;
}
int sind_seq_net_destroy(sind_seq_net* n) { delete n; return SIND_OK; }

int sind_seq_create(const sind_seq_config* cfg, sind_seq_net* net, sind_seq** out) {
    if (!cfg || !out || cfg->frames < 1 || cfg->pipe.streams < 1 || cfg->warmup < 0 || (cfg->steps < 1 && cfg->frames_per_step < 1)) { sind_set_error("sind_seq_create: bad arguments"); return SIND_E_ARG; }
    const int world = net ? net->net->world() : 1, rank = net ? net->net->rank() : 0;
    std::unique_ptr<sind_seq> q(new sind_seq()); q->cfg = *cfg; q->net = net ? net->net.get() : nullptr;
    std::string err;
    const int n_chunks = cfg->pipe.streams * world;
    if ((cfg->steps > 0 ? seq_plan_lockstep(cfg->frames, n_chunks, cfg->steps, cfg->warmup, q->plan, err) : seq_plan_for(cfg->frames, n_chunks, cfg->frames_per_step, cfg->warmup, q->plan, err)) != 0) {
        sind_set_error("sind_seq_create: %s", err.c_str()); return SIND_E_ARG; }
    HIP_TRY(hipSetDevice(cfg->pipe.device));
    q->src.W = cfg->pipe.width; q->src.H = cfg->pipe.height; q->src.device = cfg->pipe.device;
    sind_pipe_config pc = cfg->pipe; pc.frames_per_step = q->plan.T;
    q->main.src = &q->src; q->main.sink = &q->sink;
    SIND_TRY(q->main.create(pc));
    q->has_repair = cfg->verify != 0 && q->plan.n_chunks > 1;
    if (q->has_repair) {
        sind_pipe_config rc = cfg->pipe;
        rc.streams = cfg->repair_streams > 0 ? cfg->repair_streams : std::max(1, std::min(cfg->pipe.streams, 8)); rc.frames_per_step = std::max(1, cfg->repair_frames_per_step > 0 ? cfg->repair_frames_per_step : 4);
        q->repair.src = &q->src; q->repair.sink = &q->sink;
        SIND_TRY(q->repair.create(rc));
    }
    q->drv.reset(new SeqDriver(q->plan, cfg->pipe.streams, &q->main, q->has_repair ? &q->repair : nullptr, q->net, q->has_repair ? cfg->retain_frames : 0));
    sind_seq* raw = q.get();
    q->drv->on_step = [raw](int step) { return raw->step_hook ? raw->step_hook(raw->hook_user, step) : 0; };
    q->drv->on_round = [raw]() { const int r = raw->rounds_seen++; return raw->round_hook ? raw->round_hook(raw->hook_user, r) : 0; };
    (void)rank;
    *out = q.release(); return SIND_OK;
}
int sind_seq_destroy(sind_seq* q) { if (q) { (void)hipSetDevice(q->cfg.pipe.device); delete q; } return SIND_OK; }
int sind_seq_plan(sind_seq* q, int* T, int* steps, int* n_chunks, long long* first_last_start) {
    if (!q) return SIND_E_ARG;
    if (T) *T = q->plan.T;
    if (steps) *steps = q->plan.steps;
    if (n_chunks) *n_chunks = q->plan.n_chunks;
    if (first_last_start) for (int g = 0; g < q->plan.n_chunks; g++) { first_last_start[3 * g] = q->plan.chunks[g].first; first_last_start[3 * g + 1] = q->plan.chunks[g].last; first_last_start[3 * g + 2] = q->plan.chunks[g].start; }
    return SIND_OK;
}
int sind_seq_set_host_source(sind_seq* q, const uint8_t* bgr, const uint16_t* depth, long long n_frames) {
    if (!q || !bgr || !depth || n_frames < 2) { sind_set_error("sind_seq_set_host_source: need the whole sequence (at least two frames)"); return SIND_E_ARG; }
    q->src.bgr = bgr; q->src.depth = depth; q->src.n_frames = n_frames; q->src.batch = nullptr; q->src.frame = nullptr; return SIND_OK;
}
int sind_seq_set_source(sind_seq* q, sind_seq_batch_fn batch, sind_seq_frame_fn frame, void* user) {
    if (!q || !batch || !frame) { sind_set_error("sind_seq_set_source: null callback"); return SIND_E_ARG; }
    q->src.batch = batch; q->src.frame = frame; q->src.user = user; return SIND_OK;
}
int sind_seq_set_outputs(sind_seq* q, long long n_frames, uint8_t* dyna, uint8_t* label, uint8_t* mask, sind_keypoint* kps, int cap, int* nkp, uint8_t* desc) {
    if (!q || n_frames < 0 || ((kps || desc) && (!nkp || cap < 1))) { sind_set_error("sind_seq_set_outputs: bad arguments"); return SIND_E_ARG; }
    q->sink.n_frames = n_frames; q->sink.dyna = dyna; q->sink.label = label; q->sink.mask = mask; q->sink.kps = kps; q->sink.cap = cap; q->sink.nkp = nkp; q->sink.desc = desc; return SIND_OK;
}
int sind_seq_set_hooks(sind_seq* q, sind_seq_hook_fn step_hook, sind_seq_hook_fn round_hook, void* user) { if (!q) return SIND_E_ARG; q->step_hook = step_hook; q->round_hook = round_hook; q->hook_user = user; return SIND_OK; }
static int drv_rc(sind_seq* q, int r, const char* what) { if (r) { sind_set_error("%s: %s", what, q->drv->err.c_str()); return r < 0 && r >= SIND_E_CAPACITY ? r : SIND_E_STATE; } return SIND_OK; }
int sind_seq_prime(sind_seq* q) { if (!q) return SIND_E_ARG; HIP_TRY(hipSetDevice(q->cfg.pipe.device)); return drv_rc(q, q->drv->prime(), "sind_seq_prime"); }
int sind_seq_warm(sind_seq* q, int steps) { if (!q || steps < 0) return SIND_E_ARG; HIP_TRY(hipSetDevice(q->cfg.pipe.device)); return drv_rc(q, q->drv->warm(steps), "sind_seq_warm"); }
int sind_seq_set_emit_main(sind_seq* q, int on) { if (!q) return SIND_E_ARG; q->drv->emit_main = on != 0; return SIND_OK; }
int sind_seq_submit(sind_seq* q, int step) { if (!q) return SIND_E_ARG; HIP_TRY(hipSetDevice(q->cfg.pipe.device)); return drv_rc(q, q->drv->submit(step), "sind_seq_submit"); }
int sind_seq_flush(sind_seq* q) { if (!q) return SIND_E_ARG; HIP_TRY(hipSetDevice(q->cfg.pipe.device)); return drv_rc(q, q->drv->finish_main(), "sind_seq_flush"); }
int sind_seq_verify(sind_seq* q) { if (!q) return SIND_E_ARG; HIP_TRY(hipSetDevice(q->cfg.pipe.device)); if (!q->has_repair) return SIND_OK; return drv_rc(q, q->drv->verify_and_repair(), "sind_seq_verify"); }
int sind_seq_run(sind_seq* q) {
    if (!q) return SIND_E_ARG;
    SIND_TRY(sind_seq_prime(q));
    for (int i = 0; i < q->plan.steps; i++) SIND_TRY(sind_seq_submit(q, i));
    SIND_TRY(sind_seq_flush(q));
    return sind_seq_verify(q);
}
int sind_seq_stats(sind_seq* q, double* out16) {
    if (!q || !out16) return SIND_E_ARG;
    const SeqStats& s = q->drv->stats;
    const double v[16] = {(double)s.seams, (double)s.mismatched_seams, (double)s.rounds, (double)s.runners, (double)s.repaired_chunks, (double)s.repair_frames, (double)s.repair_steps,
                          (double)s.overridden_frames, (double)s.runners_to_chunk_end, (double)s.max_frames_to_converge, (double)s.replay_frames, (double)s.replay_calls,
                          (double)s.runners_past_replay, (double)s.retained_steps_dropped, s.repair_seconds, s.flush_seconds};
    std::memcpy(out16, v, sizeof(v)); return SIND_OK;
}
sind_pipe* sind_seq_pipeline(sind_seq* q) { return q ? q->main.p : nullptr; }
int sind_seq_step_outputs(sind_seq* q, const uint8_t** dyna, const uint8_t** label, const uint8_t** mask) {
    if (!q) return SIND_E_ARG;
    if (dyna) *dyna = q->main.dyna.p;
    if (label) *label = q->main.label.p;
    if (mask) *mask = q->main.mask.p;
    return SIND_OK;
}

}  // extern "C"
