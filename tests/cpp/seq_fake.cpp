// CPU stand-in for a pipeline under the C++ chunked-sequence driver (sindslam_amd/csrc/host/seq.cpp): the same SeqPipe calls, a toy stateful "detector" instead of
// DynaDetect -- the C++ twin of tests/fake_pipeline.py.  State = one B-bit shift register per stream: frame q shifts in bit(q), so two runs of the same frames forget
// their different start states after exactly B frames; frames with q % reset_every == 0 set the register to a constant.  Outputs of a frame are a function of
// (state BEFORE the frame, q), like the real outputs.  Test infrastructure: built by tests/cpp_shim.py with plain g++, never part of the product library.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../sindslam_amd/csrc/host/seq.hpp"

using namespace sind;

namespace {
struct Toy {
    int bits, reset_every;
    static uint64_t bit(long long q) { return ((uint64_t)q * 2654435761ull >> 7) & 1ull; }
    void step(uint64_t x, long long q, int* out, uint64_t* next) const {
        *out = (int)(((unsigned __int128)x * 7 + (unsigned __int128)q * 13 + 5) % 251);          // (the Python twin computes this on unbounded integers)
        if (reset_every && q % reset_every == 0 && q > 0) { *next = bits >= 64 ? 12345ull : 12345ull % (1ull << bits); return; }
        *next = (x >> 1) | (bit(q) << (bits - 1));
    }
};
struct Sink { long long n_frames; uint8_t *dyna, *label, *mask; float* kpx; uint8_t* emitted; };

class FakePipe : public SeqPipe {
public:
    int S_, T_; Toy toy; long long frames; Sink* sink; std::string err_;
    std::vector<uint64_t> x; std::vector<long long> next_q; bool hashing = false;
    struct Res { std::vector<int> out; std::vector<uint64_t> hash; };
    Res published; std::unique_ptr<Res> pending; std::map<int, std::vector<long long>> kept; int reserve = 0, retain_tag = -1;
    long long frames_processed = 0, frames_replayed = 0;
    FakePipe(int S, int T, Toy t, long long fr, Sink* sk) : S_(S), T_(T), toy(t), frames(fr), sink(sk), x((size_t)S, 0), next_q((size_t)S, -1000000) { published.out.assign((size_t)S * T, 0); published.hash.assign((size_t)S * T * 2, 0); }
    int S() const override { return S_; }
    int T() const override { return T_; }
    const char* error() const override { return err_.c_str(); }
    int prime(int s, long long last, long long) override { x[(size_t)s] = 0; next_q[(size_t)s] = last + 1; return 0; }
    int set_state_hashing(bool on) override { hashing = on; return 0; }
    int run(const long long* pos, const int* active, Res& r) {
        if (retain_tag >= 0) { kept[retain_tag] = std::vector<long long>(pos, pos + (size_t)S_ * T_); retain_tag = -1; }
        r.out.assign((size_t)S_ * T_, 0); r.hash.assign((size_t)S_ * T_ * 2, 0);
        for (int s = 0; s < S_; s++) {
            const int act = active ? active[s] : T_;
            for (int t = 0; t < act; t++) {
                const long long q = pos[(size_t)s * T_ + t], qq = std::min(q, frames - 1);      // positions past the end repeat the last frame
                if (q < frames) {
                    if (next_q[(size_t)s] != q) { err_ = "stream " + std::to_string(s) + ": expected position " + std::to_string(next_q[(size_t)s]) + ", got " + std::to_string(q); return -1; }
                    next_q[(size_t)s] = q + 1;
                }
                int o; uint64_t nx; toy.step(x[(size_t)s], qq, &o, &nx); x[(size_t)s] = nx;
                r.out[(size_t)s * T_ + t] = o; r.hash[((size_t)s * T_ + t) * 2] = nx + 1; r.hash[((size_t)s * T_ + t) * 2 + 1] = 77;
                frames_processed++;
            }
            if (act < T_) next_q[(size_t)s] = -1000000;          // ragged step: the stream has to be primed again
        }
        return 0;
    }
    int submit(const long long* pos, bool* have) override {
        std::unique_ptr<Res> r(new Res());
        if (run(pos, nullptr, *r)) return -1;
        *have = (bool)pending;
        if (pending) published = *pending;
        pending = std::move(r);
        return 0;
    }
    int flush(bool* have) override { *have = (bool)pending; if (pending) { published = *pending; pending.reset(); } return 0; }
    int process(const long long* pos, const int* active) override { if (pending) { err_ = "process with a step pending"; return -1; } Res r; if (run(pos, active, r)) return -1; published = r; return 0; }
    int state_hashes(uint64_t* out) override { if (!hashing) { err_ = "state hashing is off"; return -1; } std::memcpy(out, published.hash.data(), published.hash.size() * 8); return 0; }
    size_t state_bytes() override { return 16; }
    int get_state(int s, uint8_t* blob) override { if (pending) { err_ = "get_state with a step pending"; return -1; } uint64_t v[2] = {x[(size_t)s], 0}; std::memcpy(blob, v, 16); return 0; }
    int set_state(int s, const uint8_t* blob) override { uint64_t v[2]; std::memcpy(v, blob, 16); x[(size_t)s] = v[0]; return 0; }
    int reserve_retained(int n) override { reserve = n; return 0; }
    int retain_next(int tag) override { if (kept.count(tag) || (int)kept.size() >= reserve) { err_ = "retain_next: no reserve or tag in use"; return -1; } retain_tag = tag; return 0; }
    int release_retained(int tag) override { if (tag < 0) kept.clear(); else kept.erase(tag); return 0; }
    int replay(int tag, const int* first, const int* last) override {
        if (pending) { err_ = "replay with a step pending"; return -1; }
        auto it = kept.find(tag); if (it == kept.end()) { err_ = "replay: unknown tag"; return -1; }
        for (int s = 0; s < S_; s++) for (int t = first[s]; t < last[s]; t++) {
            const long long q = it->second[(size_t)s * T_ + t]; int o; uint64_t nx; toy.step(x[(size_t)s], std::min(q, frames - 1), &o, &nx); x[(size_t)s] = nx;
            published.out[(size_t)s * T_ + t] = o; published.hash[((size_t)s * T_ + t) * 2] = nx + 1; published.hash[((size_t)s * T_ + t) * 2 + 1] = 77; frames_replayed++;
        }
        return 0;
    }
    int emit(int s, int t, long long pos) override {
        const long long f = pos + 1; if (f < 0 || f >= sink->n_frames) return 0;
        const int o = published.out[(size_t)s * T_ + t];
        sink->dyna[f] = (uint8_t)o; sink->label[f] = (uint8_t)(o / 2); sink->mask[f] = (uint8_t)(255 - o); sink->kpx[f] = (float)o; sink->emitted[f] = 1;
        return 0;
    }
};
}  // namespace

extern "C" {
// One rank of a toy job through the C++ driver.  Arrays of n_frames entries indexed by frame; owned[f] = 1 for the frames this rank owns.  stats: the 16 values of
// sind_seq_stats; counters: frames processed by the main pipeline, by the repair pipeline, frames replayed.  world > 1: the exchange runs over loopback TCP.
int seq_fake_run(long long n_frames, int streams, int T, int warmup, int bits, int reset_every, int world, int rank, int base_port, int repair_streams, int repair_T, int verify, int retain,
                 uint8_t* dyna, uint8_t* label, uint8_t* mask, float* kpx, uint8_t* owned, double* stats16, long long* counters3, char* errbuf, int errlen) {
    auto fail = [&](const std::string& e) { if (errbuf && errlen > 0) std::snprintf(errbuf, (size_t)errlen, "%s", e.c_str()); return -1; };
    std::string err; SeqPlan plan;
    if (seq_plan_for(n_frames - 1, streams * world, T, warmup, plan, err)) return fail(err);
    std::unique_ptr<SeqNet> net;
    if (world > 1) { net.reset(seq_net_tcp(rank, world, "127.0.0.1", base_port, err)); if (!net) return fail(err); }
    std::vector<uint8_t> emitted((size_t)n_frames, 0);
    Sink sink{n_frames, dyna, label, mask, kpx, emitted.data()};
    const Toy toy{bits, reset_every};
    FakePipe pipe(streams, plan.T, toy, n_frames - 1, &sink);
    std::unique_ptr<FakePipe> rp;
    if (verify && plan.n_chunks > 1) rp.reset(new FakePipe(repair_streams > 0 ? repair_streams : std::max(1, std::min(streams, 8)), std::max(1, repair_T), toy, n_frames - 1, &sink));
    SeqDriver drv(plan, streams, &pipe, rp.get(), net.get(), rp ? retain : 0);
    if (drv.run()) return fail(drv.err);
    for (const SeqChunk& c : drv.mine()) for (long long q = c.first; q < c.last; q++) { owned[q + 1] = 1; if (!emitted[(size_t)q + 1]) return fail("owned frame " + std::to_string(q + 1) + " was never delivered"); }
    const SeqStats& s = drv.stats;
    const double v[16] = {(double)s.seams, (double)s.mismatched_seams, (double)s.rounds, (double)s.runners, (double)s.repaired_chunks, (double)s.repair_frames, (double)s.repair_steps,
                          (double)s.overridden_frames, (double)s.runners_to_chunk_end, (double)s.max_frames_to_converge, (double)s.replay_frames, (double)s.replay_calls,
                          (double)s.runners_past_replay, (double)s.retained_steps_dropped, s.repair_seconds, s.flush_seconds};
    std::memcpy(stats16, v, sizeof(v));
    counters3[0] = pipe.frames_processed; counters3[1] = rp ? rp->frames_processed : 0; counters3[2] = pipe.frames_replayed;
    return 0;
}
int seq_fake_plan(long long frames, int n_chunks, int frames_per_step, int steps, int warmup, int* T, int* out_steps, long long* first_last_start) {
    std::string err; SeqPlan p;
    if ((steps > 0 ? seq_plan_lockstep(frames, n_chunks, steps, warmup, p, err) : seq_plan_for(frames, n_chunks, frames_per_step, warmup, p, err))) return -1;
    *T = p.T; *out_steps = p.steps;
    for (int g = 0; g < n_chunks; g++) { first_last_start[3 * g] = p.chunks[g].first; first_last_start[3 * g + 1] = p.chunks[g].last; first_last_start[3 * g + 2] = p.chunks[g].start; }
    return 0;
}
}
