// The chunked-sequence driver: plan, lock-step steps, seam verification, replay and repair runners, hand-over between ranks (see seq.hpp).  HIP-free.
#include "seq.hpp"
#include <algorithm>
#include <chrono>
#include <cstring>

namespace sind {

static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const uint64_t NO_HASH[2] = {~0ull, ~0ull};       // "no such state": never equal to a fingerprint

int seq_plan_lockstep(long long frames, int n_chunks, int steps, int warmup, SeqPlan& out, std::string& err) {
    if (frames < 1 || n_chunks < 1 || steps < 1 || warmup < 0) { err = "seq_plan_lockstep: need at least one frame, chunk and step and a non-negative warm-up"; return -1; }
    long long need = cdiv(frames + (long long)warmup * (n_chunks - 1), n_chunks);         // smallest P that covers the sequence
    if (n_chunks > 1) need = std::max<long long>(need, warmup + 1);                       // a later chunk must own something
    const long long T = cdiv(need, steps), P = (long long)steps * T;
    if (T > (1 << 20)) { err = "seq_plan_lockstep: more frames per step than a pipeline holds"; return -1; }
    out = SeqPlan(); out.frames = frames; out.n_chunks = n_chunks; out.steps = steps; out.T = (int)T; out.warmup = warmup;
    long long first = 0;
    for (int g = 0; g < n_chunks; g++) {
        const long long n = g == 0 ? P : P - warmup;
        out.chunks.push_back({std::min(first, frames), std::min(first + n, frames), first - (g ? warmup : 0)});     // start may lie before `first` even when the chunk owns nothing: it still runs
        first += n;
    }
    return 0;
}
int seq_plan_for(long long frames, int n_chunks, int frames_per_step, int warmup, SeqPlan& out, std::string& err) {
    if (frames < 1 || n_chunks < 1 || warmup < 0) { err = "seq_plan_for: bad arguments"; return -1; }
    long long need = cdiv(frames + (long long)warmup * (n_chunks - 1), n_chunks);
    if (n_chunks > 1) need = std::max<long long>(need, warmup + 1);
    return seq_plan_lockstep(frames, n_chunks, (int)cdiv(need, std::max(1, frames_per_step)), warmup, out, err);
}

SeqDriver::SeqDriver(const SeqPlan& plan, int S, SeqPipe* pipe, SeqPipe* repair, SeqNet* net, int retain_frames)
    : plan_(plan), S_(S), rank_(net ? net->rank() : 0), world_(net ? net->world() : 1), pipe_(pipe), repair_(repair), net_(net) {
    for (int s = 0; s < S; s++) mine_.push_back(plan.chunks[(size_t)rank_ * S + s]);
    // REPLAY: the steps that hold the first retain_frames owned frames of the chunks after the first (the chunks run in lock-step, so these are the same few steps for all
    // of them) keep their phase-A outputs; a runner re-runs only the stateful tails of those frames on the chunk's own stream.  How long a runner needs is a property of
    // the data (the k-means of a chunk that started from other labels can sit in another local optimum for as long as the scene stays similar), hence < 0 = every step.
    const int T = plan.T;
    long long rf = retain_frames;
    if (rf < 0) rf = std::max<long long>(1, plan.processed() - plan.warmup);
    if (rf > 0 && plan.n_chunks > 1 && repair)
        for (long long k = plan.warmup / T; k < std::min<long long>(plan.steps, (plan.warmup + rf - 1) / T + 1); k++) retained_.push_back((int)k);
    H_.assign((size_t)S * (size_t)plan.processed() * 2, 0);
}
int SeqDriver::fail(const char* what, SeqPipe* p) { err = std::string(what) + ": " + (p ? p->error() : ""); return -1; }
bool SeqDriver::is_retained(int step) const { return std::find(retained_.begin(), retained_.end(), step) != retained_.end(); }

int SeqDriver::prime() {
    for (int s = 0; s < S_; s++) if (pipe_->prime(s, mine_[s].start - 1, mine_[s].start - 2)) return fail("prime", pipe_);
    if (pipe_->set_state_hashing(true)) return fail("set_state_hashing", pipe_);
    if (repair_) {
        for (int j = 0; j < repair_->S(); j++) if (repair_->prime(j, -1, -2)) return fail("prime (repair pipeline)", repair_);
        if (repair_->set_state_hashing(true)) return fail("set_state_hashing (repair pipeline)", repair_);
    }
    if (!retained_.empty()) {
        if (pipe_->release_retained(-1)) return fail("release_retained", pipe_);
        // a retained step is ~4 KB of HBM and ~0.6 KB of page-locked host memory per pixel-frame: take what the machine gives, earliest steps first.  reserve_retained(n) means
        // exactly n sets: the retry with the smaller count gives back what the failed larger request had completed
        while (!retained_.empty()) {
            if (pipe_->reserve_retained((int)retained_.size()) == 0) break;
            if (retained_.size() == 1) return fail("reserve_retained", pipe_);
            stats.retained_steps_dropped += (long long)(retained_.size() - retained_.size() / 2);
            retained_.resize(retained_.size() / 2);
        }
        if (pipe_->warm_two_chain_mode()) return fail("warm_two_chain_mode", pipe_);
    }
    pending_ = -1;
    return 0;
}

int SeqDriver::warm(int steps) {
    const int T = plan_.T;
    std::vector<long long> pos((size_t)S_ * T);
    for (int i = 0; i < std::min(steps, plan_.steps); i++) {
        for (int s = 0; s < S_; s++) for (int t = 0; t < T; t++) pos[(size_t)s * T + t] = mine_[s].start + (long long)i * T + t;
        if (pipe_->process(pos.data(), nullptr)) return fail("process (warm-up)", pipe_);
    }
    if (repair_) {                          // one untimed step of the repair pipeline as well (first-use allocations)
        const int R = repair_->S(), Tr = repair_->T(); std::vector<long long> rp((size_t)R * Tr);
        for (int j = 0; j < R; j++) for (int t = 0; t < Tr; t++) rp[(size_t)j * Tr + t] = t;
        if (repair_->process(rp.data(), nullptr)) return fail("process (warm-up of the repair pipeline)", repair_);
    }
    return 0;
}
int SeqDriver::collect(int step) {
    const int T = plan_.T;
    std::vector<uint64_t> hh((size_t)S_ * T * 2);
    if (pipe_->state_hashes(hh.data())) return fail("state_hashes", pipe_);
    for (int s = 0; s < S_; s++) std::memcpy(H(s, (long long)step * T), &hh[(size_t)s * T * 2], (size_t)T * 2 * sizeof(uint64_t));
    if (emit_main) for (int s = 0; s < S_; s++) for (int t = 0; t < T; t++) {
        const long long q = mine_[s].start + (long long)step * T + t;
        if (mine_[s].first <= q && q < mine_[s].last) if (pipe_->emit(s, t, q)) return fail("emit", pipe_);
    }
    if (on_step) { const int r = on_step(step); if (r) { if (err.empty()) err = "on_step hook failed"; return r; } }
    return 0;
}
int SeqDriver::submit(int step) {
    if (step < 0 || step >= plan_.steps || (pending_ >= 0 && step != pending_ + 1) || (pending_ < 0 && step != 0)) { err = "SeqDriver::submit: steps go in order, 0 .. steps - 1"; return -1; }
    const int T = plan_.T;
    std::vector<long long> pos((size_t)S_ * T);
    for (int s = 0; s < S_; s++) for (int t = 0; t < T; t++) pos[(size_t)s * T + t] = mine_[s].start + (long long)step * T + t;
    if (is_retained(step) && pipe_->retain_next(step)) return fail("retain_next", pipe_);
    bool have = false;
    if (pipe_->submit(pos.data(), &have)) return fail("submit", pipe_);
    if (have) { const int r = collect(pending_); if (r) return r; }
    pending_ = step;
    return 0;
}
int SeqDriver::finish_main() {
    const double t0 = now_s(); bool have = false;
    if (pipe_->flush(&have)) return fail("flush", pipe_);
    if (have && pending_ >= 0) { const int r = collect(pending_); if (r) return r; }
    stats.flush_seconds = now_s() - t0;
    return 0;
}

void SeqDriver::hash_at(int s, long long q, uint64_t out[2]) {
    const SeqChunk& c = mine_[s];
    if (c.start <= q && q < c.start + plan_.processed()) std::memcpy(out, H(s, q - c.start), 16); else std::memcpy(out, NO_HASH, 16);
}
int SeqDriver::end_blob(int s, std::vector<uint8_t>& out) {
    auto it = end_blob_.find(s);
    if (it != end_blob_.end()) { out = it->second; return 0; }
    out.resize(pipe_->state_bytes());
    if (pipe_->get_state(s, out.data())) return fail("get_state", pipe_);
    return 0;
}

int SeqDriver::take(int s, long long q, const uint64_t hh[2], SeqPipe* from, int slot, int t, std::vector<uint64_t>& end_h, bool* done) {
    const SeqChunk& c = mine_[s]; const long long i = q - c.start;
    if (from->emit(slot, t, q)) return fail("emit", from);
    stats.overridden_frames++;
    if (std::memcmp(hh, H(s, i), 16) == 0) {                   // same state as the chain that is already there: the rest of it stands
        stats.max_frames_to_converge = std::max(stats.max_frames_to_converge, q - c.first + 1);
        stats.repaired_chunks++; *done = true; return 0;
    }
    std::memcpy(H(s, i), hh, 16);
    if (q + 1 >= c.last) {                                     // the runner IS the chunk now: new end state, the successor is verified again
        std::memcpy(&end_h[(size_t)s * 2], hh, 16);
        std::vector<uint8_t> b(from->state_bytes());
        if (from->get_state(slot, b.data())) return fail("get_state", from);
        end_blob_[s] = std::move(b);
        stats.runners_to_chunk_end++; stats.repaired_chunks++; *done = true; return 0;
    }
    *done = false; return 0;
}

// runners on the chunks' own streams over the retained steps (tails only); leaves in `starts` the runners that are still not through: {s: (next position, state)}
int SeqDriver::replay_runners(std::map<int, std::pair<long long, std::vector<uint8_t>>>& starts, std::vector<uint64_t>& end_h) {
    const int T = plan_.T, S = S_;
    for (auto& kv : starts) if (pipe_->set_state(kv.first, kv.second.second.data())) return fail("set_state", pipe_);
    std::map<int, long long> live; for (auto& kv : starts) live[kv.first] = kv.second.first;
    std::vector<int> t0(S), t1(S); std::vector<uint64_t> hh((size_t)S * T * 2);
    for (int k : retained_) {
        std::fill(t0.begin(), t0.end(), 0); std::fill(t1.begin(), t1.end(), 0); bool any = false; long long sum = 0;
        for (auto& kv : live) {
            const SeqChunk& c = mine_[kv.first];
            const long long lo = std::max(kv.second - c.start, (long long)k * T), hi = std::min(c.last - c.start, (long long)(k + 1) * T);
            if (lo < hi) {
                if (lo != kv.second - c.start) { err = "SeqDriver: a runner's frames are not consecutive"; return -1; }
                t0[kv.first] = (int)(lo - (long long)k * T); t1[kv.first] = (int)(hi - (long long)k * T); any = true; sum += hi - lo;
            }
        }
        if (!any) continue;
        if (pipe_->replay(k, t0.data(), t1.data())) return fail("replay", pipe_);
        if (pipe_->state_hashes(hh.data())) return fail("state_hashes", pipe_);
        stats.replay_calls++; stats.replay_frames += sum;
        for (auto it = live.begin(); it != live.end();) {
            const int s = it->first; const SeqChunk& c = mine_[s]; bool gone = false;
            for (int t = t0[s]; t < t1[s]; t++) {
                const long long q = c.start + (long long)k * T + t; bool done = false;
                const int r = take(s, q, &hh[((size_t)s * T + t) * 2], pipe_, s, t, end_h, &done); if (r) return r;
                if (done) { gone = true; break; }
                it->second = q + 1;
            }
            if (gone) it = live.erase(it); else ++it;
        }
    }
    std::map<int, std::pair<long long, std::vector<uint8_t>>> left;
    for (auto& kv : live) {
        std::vector<uint8_t> b(pipe_->state_bytes());
        if (pipe_->get_state(kv.first, b.data())) return fail("get_state", pipe_);
        left[kv.first] = {kv.second, std::move(b)};
    }
    stats.runners_past_replay += (long long)left.size();
    starts.swap(left);
    return 0;
}

// full re-processing on the repair pipeline: batch = {my chunk s: (first position, state before it)}, at most repair.S of them
int SeqDriver::run_runners(const std::map<int, std::pair<long long, std::vector<uint8_t>>>& batch, std::vector<uint64_t>& end_h) {
    SeqPipe* rp = repair_; const int R = rp->S(), Tr = rp->T();
    std::map<int, std::pair<int, long long>> slots; int j = 0;
    for (auto& kv : batch) {
        const long long q0 = kv.second.first;
        if (rp->prime(j, q0 - 1, q0 - 2)) return fail("prime (runner)", rp);
        if (rp->set_state(j, kv.second.second.data())) return fail("set_state (runner)", rp);
        slots[j] = {kv.first, q0}; j++;
    }
    std::vector<long long> pos((size_t)R * Tr); std::vector<int> act(R); std::vector<uint64_t> hh((size_t)R * Tr * 2);
    while (!slots.empty()) {
        long long sum = 0;
        for (int jj = 0; jj < R; jj++) {
            auto it = slots.find(jj);
            if (it != slots.end()) {
                const int s = it->second.first; const long long q = it->second.second;
                act[jj] = (int)std::min<long long>(Tr, mine_[s].last - q); sum += act[jj];
                for (int t = 0; t < Tr; t++) pos[(size_t)jj * Tr + t] = q + t;
            } else { act[jj] = 0; for (int t = 0; t < Tr; t++) pos[(size_t)jj * Tr + t] = t; }       // idle slot: any valid frames, no tail runs
        }
        if (rp->process(pos.data(), act.data())) return fail("process (runner)", rp);
        if (rp->state_hashes(hh.data())) return fail("state_hashes (runner)", rp);
        stats.repair_steps++; stats.repair_frames += sum;
        for (auto it = slots.begin(); it != slots.end();) {
            const int jj = it->first, s = it->second.first; bool gone = false;
            for (int t = 0; t < act[jj]; t++) {
                bool done = false;
                const int r = take(s, pos[(size_t)jj * Tr + t], &hh[((size_t)jj * Tr + t) * 2], rp, jj, t, end_h, &done); if (r) return r;
                if (done) { gone = true; break; }
                it->second.second = pos[(size_t)jj * Tr + t] + 1;
            }
            if (gone) it = slots.erase(it); else ++it;
        }
    }
    return 0;
}

int SeqDriver::verify_and_repair() {
    const double t_begin = now_s();
    const int S = S_, n = plan_.n_chunks;
    std::vector<uint64_t> start_h((size_t)S * 2, 0), end_h((size_t)S * 2, 0);
    for (int s = 0; s < S; s++) {
        if (rank_ * S + s > 0) hash_at(s, mine_[s].first - 1, &start_h[(size_t)s * 2]);
        hash_at(s, mine_[s].last - 1, &end_h[(size_t)s * 2]);
    }
    bool first_round = true, saved = false;
    std::vector<uint64_t> mine4((size_t)S * 4), allh((size_t)n * 4);
    for (;;) {
        for (int s = 0; s < S; s++) { std::memcpy(&mine4[(size_t)s * 4], &start_h[(size_t)s * 2], 16); std::memcpy(&mine4[(size_t)s * 4 + 2], &end_h[(size_t)s * 2], 16); }
        if (world_ == 1) allh = mine4;
        else if (net_->allgather(mine4.data(), mine4.size() * sizeof(uint64_t), allh.data())) { err = std::string("allgather of the seam fingerprints: ") + net_->error(); return -1; }
        std::vector<char> need(n, 0); long long n_need = 0;
        for (int g = 1; g < n; g++) if (plan_.chunks[g].owns() && std::memcmp(&allh[(size_t)g * 4], &allh[(size_t)(g - 1) * 4 + 2], 16) != 0) { need[g] = 1; n_need++; }
        if (first_round) { for (int g = 1; g < n; g++) if (plan_.chunks[g].owns()) stats.seams++; stats.mismatched_seams = n_need; first_round = false; }
        if (!n_need) break;
        stats.rounds++;
        if (!retained_.empty() && !saved) {      // the replay runs use the chunks' own streams: take every chunk's end state out first
            for (int s = 0; s < S; s++) if (!end_blob_.count(s)) { std::vector<uint8_t> b; const int r = end_blob(s, b); if (r) return r; end_blob_[s] = std::move(b); }
            saved = true;
        }
        // the end-state blob of the last chunk of rank r - 1 for every needed seam between ranks
        std::vector<uint8_t> got; bool have_got = false;
        if (world_ > 1) {
            const size_t sz = pipe_->state_bytes();
            std::vector<uint8_t> snd; int to = -1, from = -1;
            if (rank_ + 1 < world_ && need[(size_t)(rank_ + 1) * S]) { const int r = end_blob(S - 1, snd); if (r) return r; to = rank_ + 1; }
            if (rank_ > 0 && need[(size_t)rank_ * S]) { got.resize(sz); from = rank_ - 1; have_got = true; }
            if (to >= 0 || from >= 0)
                if (net_->sendrecv(to >= 0 ? snd.data() : nullptr, to, from >= 0 ? got.data() : nullptr, from, sz)) { err = std::string("hand-over of a seam's state: ") + net_->error(); return -1; }
        }
        std::map<int, std::pair<long long, std::vector<uint8_t>>> starts;      // runner of chunk s: (first position to process, state blob before it)
        for (int s = 0; s < S; s++) {
            if (!need[(size_t)rank_ * S + s]) continue;
            std::vector<uint8_t> blob;
            if (s == 0 && have_got) blob = got;
            else if (s > 0) { const int r = end_blob(s - 1, blob); if (r) return r; }
            else { err = "SeqDriver: a seam between ranks without its state blob"; return -1; }
            starts[s] = {mine_[s].first, std::move(blob)};
            // the state this chunk's chain now starts from (the gathered value; a local predecessor that gets a new end state in this round re-opens the seam)
            std::memcpy(&start_h[(size_t)s * 2], &allh[(size_t)(rank_ * S + s - 1) * 4 + 2], 16);
        }
        stats.runners += (long long)starts.size();
        if (!retained_.empty()) { const int r = replay_runners(starts, end_h); if (r) return r; }
        if (!starts.empty()) {
            if (!repair_) { err = "SeqDriver: a chunk needs more repair than the retained frames hold and there is no repair pipeline"; return -1; }
            std::vector<int> order; for (auto& kv : starts) order.push_back(kv.first);
            const int R = repair_->S();
            for (size_t b0 = 0; b0 < order.size(); b0 += (size_t)R) {
                std::map<int, std::pair<long long, std::vector<uint8_t>>> batch;
                for (size_t k = b0; k < std::min(order.size(), b0 + (size_t)R); k++) {
                    const int s = order[k]; long long q0 = starts[s].first; std::vector<uint8_t> blob = starts[s].second;
                    if (q0 == mine_[s].first && s > 0 && end_blob_.count(s - 1)) {
                        // a runner that has not started yet takes its LOCAL predecessor's newest end state: batches run in chunk order, so an earlier batch of this
                        // round may just have made it true (saves the round that would otherwise re-open this seam)
                        blob = end_blob_[s - 1]; std::memcpy(&start_h[(size_t)s * 2], &end_h[(size_t)(s - 1) * 2], 16);
                    }
                    batch[s] = {q0, std::move(blob)};
                }
                const int r = run_runners(batch, end_h); if (r) return r;
            }
        }
        if (on_round) { const int r = on_round(); if (r) { if (err.empty()) err = "on_round hook failed"; return r; } }
    }
    stats.repair_seconds += now_s() - t_begin;
    return 0;
}

}  // namespace sind
