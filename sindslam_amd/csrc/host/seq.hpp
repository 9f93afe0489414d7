// One long RGB-D sequence on the batched pipeline, in C++ ("frames of a TUM sequence shard naturally across the GPUs"; north_star: host code stays C++).
//
// The state-free work of DynaDetect + ORB (> 99 % of the bytes) needs only frames n, n-1, n-2 and depth n; the light stateful tail (k-means warm labels, sample
// weights, previous high mask: reference DynaDetect.h:165-178, rolled at DynaDetect.cc:1660-1664) needs frame order.  So a sequence is cut into contiguous lock-step
// CHUNKS, one per pipeline stream (S chunks per rank): chunk 0 starts at the first frame exactly like the reference loop (rgbd_tum_noros.cc:103-107, 131-139), a later
// chunk starts `warmup` frames early from an empty state (speculation).  The driver then makes every chunk the sequential result:
//   verify   the pipeline leaves a 128-bit fingerprint of the rolled state per frame; chunk g is the sequential result iff its state before its first owned frame equals
//            the end state of chunk g - 1 (chunk 0 is the sequential loop);
//   repair   otherwise a RUNNER re-runs the chunk's frames from the true state -- first only the stateful tails on the retained phase-A outputs of the chunk's own stream
//            (replay), then whole frames on a small second pipeline -- until its state equals the speculative state of the same frame (from there on the speculative
//            results ARE the sequential ones) or the chunk ends; a runner that reaches the end of its chunk changes the chunk's end state and the successor is verified again.
// With several ranks (chunk g lives on rank g / S) a round costs one all-gather of 32 bytes per chunk plus one send / receive of the state blob for a mismatching seam
// between two ranks; there is no other cross-rank dependency.
//
// This file is free of HIP: the driver talks to a pipeline through SeqPipe and to the other ranks through SeqNet.  capi_seq.cpp adapts sind_pipe_* and RCCL / loopback
// TCP; tests/cpp/seq_fake.cpp drives the same driver with a toy stateful detector on the CPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace sind {

struct SeqChunk {
    long long first, last, start;          // owned positions [first, last); first PROCESSED position (= first - warm-up for a chunk after the first)
    bool owns() const { return last > first; }
};
// A fixed-length sequence on n lock-step chunks: every chunk PROCESSES exactly steps * T frames.  Positions count the frames DetectDynaArea sees (0 = the first one =
// frame 1 of the sequence; positions -1, -2 are the priming frames).
struct SeqPlan {
    long long frames = 0; int n_chunks = 0, steps = 0, T = 0, warmup = 0; std::vector<SeqChunk> chunks;
    long long processed() const { return (long long)steps * T; }
};
// smallest T with steps * T + (n - 1) * (steps * T - warmup) >= frames; chunk 0 owns [0, P), chunk g the next P - warmup positions
int seq_plan_lockstep(long long frames, int n_chunks, int steps, int warmup, SeqPlan& out, std::string& err);
// the plan whose steps hold at most frames_per_step frames per chunk
int seq_plan_for(long long frames, int n_chunks, int frames_per_step, int warmup, SeqPlan& out, std::string& err);

// What the driver needs of a pipeline of S streams x T frames per step.  Frames are named by sequence POSITION: fetching the inputs and delivering the outputs of a
// frame are the adapter's business (it owns the source and the sink).  Every call returns 0 or a negative error code and leaves a message in error().
struct SeqPipe {
    virtual ~SeqPipe() {}
    virtual int S() const = 0;
    virtual int T() const = 0;
    virtual const char* error() const = 0;
    virtual int prime(int s, long long pos_last, long long pos_lastlast) = 0;       // stream s starts behind these two frames, with an empty state
    virtual int set_state_hashing(bool on) = 0;
    virtual int submit(const long long* pos, bool* have) = 0;      // pos [S][T]; software-pipelined: *have = the step submitted before this one is published now
    virtual int flush(bool* have) = 0;
    virtual int process(const long long* pos, const int* active) = 0;               // synchronous step; active [S] (frames of each stream that run) or null = all
    virtual int state_hashes(uint64_t* out) = 0;                   // [S][T][2] of the published step / the last replay
    virtual size_t state_bytes() = 0;
    virtual int get_state(int s, uint8_t* blob) = 0;
    virtual int set_state(int s, const uint8_t* blob) = 0;
    virtual int reserve_retained(int steps) = 0;                   // exactly `steps` sets (frees a surplus); fails when memory runs out
    virtual int retain_next(int tag) = 0;
    virtual int release_retained(int tag) = 0;
    virtual int replay(int tag, const int* first, const int* last) = 0;             // tails of frames [first[s], last[s]) of the retained step again, from the streams' states
    virtual int warm_two_chain_mode() { return 0; }                // create what the first replay would create (outside any timed region)
    virtual int emit(int s, int t, long long pos) = 0;             // frame (s, t) of the published step / last replay IS sequence position pos: hand its outputs to the sink
};

// The exchange between the ranks of one job.
struct SeqNet {
    virtual ~SeqNet() {}
    virtual int rank() const = 0;
    virtual int world() const = 0;
    virtual const char* error() const = 0;
    virtual int allgather(const void* mine, size_t bytes, void* all) = 0;           // all = world * bytes, rank order
    // one hand-over along the chain of ranks: send `bytes` to rank `to` (-1: nothing to send) and receive as many from rank `from` (-1: nothing); a rank in the middle does both
    virtual int sendrecv(const void* send, int to, void* recv, int from, size_t bytes) = 0;
};

struct SeqStats {
    long long seams = 0, mismatched_seams = 0, rounds = 0, runners = 0, repaired_chunks = 0, repair_frames = 0, repair_steps = 0, overridden_frames = 0, runners_to_chunk_end = 0,
              max_frames_to_converge = 0, replay_frames = 0, replay_calls = 0, runners_past_replay = 0, retained_steps_dropped = 0;
    double repair_seconds = 0, flush_seconds = 0;
};

class SeqDriver {
public:
    // pipe: S x plan.T; repair: R x Tr or null (no verification: the speculative results stand); net: null for one rank.
    // retain_frames: the steps holding the first retain_frames owned frames of the chunks after the first keep their phase-A outputs (replay); < 0: every step from the
    // first owned frame on; 0: runners re-process whole frames on the repair pipeline only.
    SeqDriver(const SeqPlan& plan, int S, SeqPipe* pipe, SeqPipe* repair, SeqNet* net, int retain_frames);
    int prime();
    int warm(int steps);                    // untimed rehearsal: the first `steps` steps of the job synchronously (and one step of the repair pipeline); prime() afterwards
    bool emit_main = true;                  // false: the owned frames of a lock-step step are NOT handed to the sink (the step hook reads the step's own arrays); repairs always are
    int submit(int step);                   // lock-step step `step` (0 .. steps - 1, in order); the results of step - 1 are emitted
    int finish_main();                      // drains the last step
    int verify_and_repair();
    int run() { int r = prime(); for (int i = 0; r == 0 && i < plan_.steps; i++) r = submit(i); if (r == 0) r = finish_main(); if (r == 0 && repair_) r = verify_and_repair(); return r; }
    std::function<int(int step)> on_step;   // the owned frames of `step` have been emitted (the bench's per-step mask gather)
    std::function<int()> on_round;          // end of a repair round, called on every rank (collective hook: the repaired masks travel here)
    const SeqPlan& plan() const { return plan_; }
    const std::vector<SeqChunk>& mine() const { return mine_; }
    SeqStats stats; std::string err;
private:
    SeqPlan plan_; int S_, rank_, world_; SeqPipe* pipe_; SeqPipe* repair_; SeqNet* net_;
    std::vector<SeqChunk> mine_; std::vector<int> retained_;
    std::vector<uint64_t> H_;               // [S][processed][2]: fingerprint of the state after every processed frame of my chunks (current best chain)
    std::map<int, std::vector<uint8_t>> end_blob_;      // local chunk -> end-state blob once a runner changed it (or once it was saved before the replays)
    int pending_ = -1;
    uint64_t* H(int s, long long i) { return &H_[((size_t)s * (size_t)plan_.processed() + (size_t)i) * 2]; }
    int fail(const char* what, SeqPipe* p);
    int collect(int step);
    void hash_at(int s, long long q, uint64_t out[2]);
    int end_blob(int s, std::vector<uint8_t>& out);
    bool is_retained(int step) const;
    // frame q of my chunk s has been re-processed from the true state (fingerprint hh, results in `from` at (slot, t)): emit, then decide -- *done: the runner is through
    int take(int s, long long q, const uint64_t hh[2], SeqPipe* from, int slot, int t, std::vector<uint64_t>& end_h, bool* done);
    int replay_runners(std::map<int, std::pair<long long, std::vector<uint8_t>>>& starts, std::vector<uint64_t>& end_h);
    int run_runners(const std::map<int, std::pair<long long, std::vector<uint8_t>>>& batch, std::vector<uint64_t>& end_h);
};

// loopback / LAN TCP mesh for the exchange (CPU-side rehearsal of several ranks on one card, or a cluster without RCCL): rank r listens on base_port + r
SeqNet* seq_net_tcp(int rank, int world, const char* host, int base_port, std::string& err);

}  // namespace sind
