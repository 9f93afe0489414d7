// Shared host/device helpers for libsind_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <utility>
#include <vector>

// error codes returned across the C ABI (include/sind_hip.h)
#define SIND_OK 0
#define SIND_E_ARG (-1)
#define SIND_E_HIP (-2)
#define SIND_E_ALLOC (-3)
#define SIND_E_STATE (-4)
#define SIND_E_CAPACITY (-5)

extern "C" const char* sind_last_error();
void sind_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            sind_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SIND_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)
#define SIND_TRY(expr)            \
    do {                          \
        int _r = (expr);          \
        if (_r != SIND_OK) return _r; \
    } while (0)

static inline int divup(int a, int b) { return (a + b - 1) / b; }

// The library ships as a drop-in: the A/B switches of the measurement rounds (SIND_FLOW_SPLIT, SIND_SOR_*, SIND_*_PRIORITY, ...) and the dormant solver
// variants exist only in a LAB build (make -C sindslam_amd/csrc lab  =  -DSIND_LAB; profiles/tools/*.sh need it).  Always on: SIND_TAIL_TIMING (stage
// report at destroy), SIND_GROW_GPU (fixed region-grow share).
#ifdef SIND_LAB
static inline const char* sind_lab_env(const char* name) { return getenv(name); }
#else
static inline const char* sind_lab_env(const char*) { return nullptr; }
#endif

// ROCTx ranges around the stages of a step (rocprofv3 --marker-trace shows them next to the kernels).  The marker library is looked up
// at first use (librocprofiler-sdk-roctx, else the older libroctx64); without it the calls do nothing.
void sind_range_push(const char* name);
void sind_range_pop();
struct SindRange { explicit SindRange(const char* name) { sind_range_push(name); } ~SindRange() { sind_range_pop(); } SindRange(const SindRange&) = delete; SindRange& operator=(const SindRange&) = delete; };

// CPU tokens: the GPU boxes bound a process to a CPU quota (cgroup cpu.max, 16 cores per GPU on this pool); more runnable threads than that burn the
// quota early in a 100 ms period and the kernel then stalls EVERY thread of the process -- the flow's launch threads included -- until the period ends
// (bench.py reports it as cpu_quota.throttled_periods).  Pool tasks therefore hold a token while they compute and hand it back while they wait for
// the GPU (sind_stream_wait / sind_event_wait do that themselves), so that at most `capacity` of them are runnable at any time.
#include <condition_variable>
#include <mutex>
struct SindHostGate {
    std::mutex m; std::condition_variable cv; int free_tokens = 1 << 20, capacity = 1 << 20;
    // (relative to the tokens that are out: a submit re-sizes the gate while the previous step's tails still hold theirs -- setting the free count itself
    // handed out up to twice the capacity and brought the quota throttling back)
    void set_capacity(int n) { std::lock_guard<std::mutex> lk(m); free_tokens += n - capacity; capacity = n; cv.notify_all(); }
    void acquire() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return free_tokens > 0; }); free_tokens--; }
    void release() { { std::lock_guard<std::mutex> lk(m); free_tokens++; } cv.notify_one(); }
};
// every sind_pipe owns its gate (two handles driven from different threads must not re-base each other's token count); a pool worker names the gate
// whose token it holds in t_sind_gate, so that the waits below can hand exactly that token back
extern thread_local SindHostGate* t_sind_gate;
struct SindTokenPause {        // scope in which a token holder does not need its token (a wait for the GPU)
    SindHostGate* g; SindTokenPause() : g(t_sind_gate) { if (g) { t_sind_gate = nullptr; g->release(); } }
    ~SindTokenPause() { if (g) { g->acquire(); t_sind_gate = g; } }
};

// Host wait for a stream that does not burn a core.  hipStreamSynchronize spins (measured on MI355X: CPU time == wall time), and a
// GPU box bounds the CPU share of a process (16 cores per GPU on this pool): a spinning waiter takes that share away from the host
// stages of the other streams.  Short waits stay hot (a few queries), long ones sleep between queries.
extern std::atomic<long long> g_sind_wait_ns, g_sind_wait_calls;      // statistics of sind_stream_wait (SIND_TAIL_TIMING report)
// (Tried in round 3: a per-thread hipEventBlockingSync event + hipEventSynchronize instead of the loop below -- on this ROCm it does not sleep either: host cores busy
// 11.5 -> 14.7, CPU-quota throttling in 37 of 51 periods.)  The sleep grows with the wait: a waiter that has slept 0.2 ms is behind a queue of kernels and is
// woken every 50, then 100, after ~3 ms every 250 us -- a wake-up costs 5-10 us of CPU, at 20 us a waiting thread kept a third of a core.
// A one-stream pipeline (the in-order sequence mode) is a chain of short dependent GPU steps with most of the host idle: its workers poll without sleeping for
// t_sind_spin_us first (a sleep costs 50-70 us of wake-up latency, three to four times per frame of a 3.3 ms chain).
extern thread_local int t_sind_spin_us;
static inline hipError_t sind_stream_wait(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipErrorNotReady;
    for (int i = 0; i < 8 && e == hipErrorNotReady; i++) e = hipStreamQuery(s);
    if (t_sind_spin_us > 0) while (e == hipErrorNotReady && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < t_sind_spin_us) e = hipStreamQuery(s);
    if (e == hipErrorNotReady) {
        SindTokenPause pause;
        for (int i = 0; e == hipErrorNotReady; i++) { std::this_thread::sleep_for(std::chrono::microseconds(i < 10 ? 20 : i < 20 ? 50 : i < 40 ? 100 : 250)); e = hipStreamQuery(s); }
    }
    g_sind_wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    g_sind_wait_calls.fetch_add(1, std::memory_order_relaxed);
    return e;
}

// Same for an event: used to bound how far a launching thread runs ahead of the GPU (a HIP launch into a full queue spins)
static inline hipError_t sind_event_wait(hipEvent_t ev) {
    hipError_t e = hipEventQuery(ev);
    if (e == hipErrorNotReady) { SindTokenPause pause; while (e == hipErrorNotReady) { std::this_thread::sleep_for(std::chrono::microseconds(50)); e = hipEventQuery(ev); } }
    return e;
}

// One-time set-up PER DEVICE (hipFuncSetAttribute applies to the current device only, and every create function takes a device): the first caller on a
// device runs fn under the lock, a failure is reported and not remembered, later callers take the lock-free path.
struct SindPerDeviceInit {
    std::mutex m; std::atomic<bool> done[32];
    SindPerDeviceInit() { for (auto& d : done) d.store(false); }
    template <class F> hipError_t run(F fn) {
        int d = 0; const hipError_t e0 = hipGetDevice(&d);
        if (e0 != hipSuccess) return e0;
        if (d < 0 || d >= 32) return hipErrorInvalidDevice;
        if (done[d].load(std::memory_order_acquire)) return hipSuccess;
        std::lock_guard<std::mutex> lk(m);
        if (done[d].load(std::memory_order_relaxed)) return hipSuccess;
        const hipError_t e = fn();
        if (e == hipSuccess) done[d].store(true, std::memory_order_release);
        return e;
    }
};

// simple owning device buffer
template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    int alloc(size_t count) {
        if (count <= n && p) return SIND_OK;
        release();
        if (hipMalloc((void**)&p, count * sizeof(T)) != hipSuccess) { p = nullptr; n = 0; sind_set_error("hipMalloc(%zu bytes) failed", count * sizeof(T)); return SIND_E_ALLOC; }
        n = count; return SIND_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    ~DevBuf() { release(); }
    DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
};

// page-locked host buffer: D2H / H2D on it go through the SDMA engines and never wait behind compute kernels of other streams
// (pageable copies are staged by blit kernels, which queue up behind the flow solver when tails and the next step overlap)
template <class T>
struct PinnedBuf {
    T* p = nullptr; size_t n = 0;
    int alloc(size_t count) {
        if (count <= n && p) return SIND_OK;
        release();
        if (hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault) != hipSuccess) { p = nullptr; n = 0; sind_set_error("hipHostMalloc(%zu bytes) failed", count * sizeof(T)); return SIND_E_ALLOC; }
        n = count; return SIND_OK;
    }
    void release() { if (p) { (void)hipHostFree(p); p = nullptr; n = 0; } }
    void swap(PinnedBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
    T* data() { return p; } const T* data() const { return p; }
    ~PinnedBuf() { release(); }
    PinnedBuf() = default; PinnedBuf(const PinnedBuf&) = delete; PinnedBuf& operator=(const PinnedBuf&) = delete;
};

#ifdef __HIPCC__
// OpenCV-compatible scalar helpers (device)
__device__ __forceinline__ int d_cvRound(float v) { return __float2int_rn(v); }          // round half to even
__device__ __forceinline__ int d_cvFloorf(float v) { return (int)floorf(v); }
__device__ __forceinline__ int d_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }
__device__ __forceinline__ int d_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
#endif
