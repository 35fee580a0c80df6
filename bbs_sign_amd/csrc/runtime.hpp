#pragma once
// Runtime layer of the MI355X BBS+ engine (included by every translation unit): contexts (device-resident tables), jobs
// (device-resident batches), validation in the reference's order, kernel launches on the
// context's HIP stream.  See include/bbs_sign_amd.h for the contract.
//
// Build: hipcc --offload-arch=gfx950 (product, libbbs_sign_amd.so).  The macro BBS_HOST_TWIN
// (tests/hosttwin only, never part of the product library) swaps the HIP runtime calls for
// malloc/memcpy/for-loops so that the host logic in this file can be unit-tested in a container
// without a GPU; the product build contains no CPU execution path for the stages.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <cstring>
#include <functional>
#include <iterator>
#include <map>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bbs_sign_amd.h"
#include "host_g2.hpp"
#include "codec_dev.hpp"

using namespace bbs;

// =============================================================================================
// runtime layer
// =============================================================================================
namespace rt {
#ifdef BBS_HOST_TWIN
struct Stream {};
inline int set_device(int) { return 0; }
inline int device_count() { return 1; }
inline int stream_create(Stream*) { return 0; }
inline void stream_destroy(Stream&) {}
struct QueueBudget { int total = 0, pool = 0, dedicated_cap = 0; size_t scratch = 0; };
inline QueueBudget queue_budget(int) { return QueueBudget{}; }
inline size_t size_class(size_t b) { return b; }
inline int dmalloc(void** p, size_t b) { *p = std::calloc(b ? b : 1, 1); return *p ? 0 : -1; }
inline void dfree(void* p, size_t, int) { std::free(p); }
inline int current_device() { return 0; }
inline int h2d(void* d, const void* h, size_t b, Stream&) { if (b) std::memcpy(d, h, b); return 0; }
inline int d2h(void* h, const void* d, size_t b, Stream&) { if (b) std::memcpy(h, d, b); return 0; }
inline int dmemset(void* d, int v, size_t b, Stream&) { if (b) std::memset(d, v, b); return 0; }
inline int d2d_async(void* d, const void* s, size_t b, Stream&) { if (b) std::memcpy(d, s, b); return 0; }
inline int h2d_async(void* d, const void* h, size_t b, Stream&) { if (b) std::memcpy(d, h, b); return 0; }
inline int d2h_async(void* h, const void* d, size_t b, Stream&) { if (b) std::memcpy(h, d, b); return 0; }
inline int hmalloc(void** p, size_t b) { *p = std::malloc(b ? b : 1); return *p ? 0 : -1; }
inline void hfree(void* p, size_t) { std::free(p); }
inline int sync(Stream&) { return 0; }
inline size_t mem_free_bytes() { return (size_t)64 << 20; }      // test build: the automatic window width comes out as 8
struct Event { };
inline int event_create(Event*) { return 0; }
inline void event_destroy(Event&) {}
inline int event_record(Event&, Stream&) { return 0; }
inline int stream_wait(Stream&, Event&) { return 0; }
template <class F, class A>
inline int launch(Stream&, const A& a, size_t nthreads) {
    for (size_t t = 0; t < nthreads; t++) F::run(a, t);
    return 0;
}
// a list of time stamps recorded on the stream; read back after one synchronisation
struct EventList {
    std::vector<std::chrono::steady_clock::time_point> t;
    size_t used = 0;
    explicit EventList(size_t) {}
    int record(Stream&) { t.push_back(std::chrono::steady_clock::now()); used++; return 0; }
    int finish(Stream&) { return 0; }
    float ms(size_t a, size_t b) { return std::chrono::duration<float, std::milli>(t[b] - t[a]).count(); }
};
#else
using Stream = hipStream_t;
inline int set_device(int d) { return hipSetDevice(d) == hipSuccess ? 0 : -1; }
inline int device_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
inline int stream_create(Stream* s);      // pooled, defined below
inline void stream_destroy(Stream& s);
// Device buffers, streams and events are recycled per device: a one-shot batch call allocated ~20 buffers and two
// streams and released them again, 6 ms of hipMalloc / hipFree / stream creation per 4096-proof call.  Buffers up
// to 256 MiB go back to a free list by size class (powers of two below 1 MiB, multiples of 1 MiB above), at most
// 8 GiB per device; larger ones (the window tables) are allocated and freed directly.
struct Pools {
    std::mutex mu;
    // a pooled buffer remembers WHEN it came back (`clock`): when the pool is full the buffers that have waited longest -- what
    // an earlier workload left behind -- make room, never the working set (round 5: evicting by size, smallest first, let
    // twelve 16384-item jobs' buffers sit in a full pool for ever while a later workload of smaller jobs paid hipMalloc /
    // hipFree -- a device synchronisation -- on every buffer: the issuer's lists at half their rate)
    struct Pooled { void* p; uint64_t seq; };
    uint64_t clock = 0;
    std::map<int, std::multimap<size_t, Pooled>> bufs;
    std::map<int, size_t> pooled_bytes;
    // streams: handed out in creation order (lowest creation index first), not last-freed-first.  The runtime binds a
    // stream to one of the GPU_MAX_HW_QUEUES hardware queues when it is created, round robin; a job's two streams
    // carry its two long kernels (MSM chain / pairing), and with last-in-first-out recycling some sequences of frees
    // put the pairing streams of a set of concurrent jobs on a few hardware queues (measured: 1.19 M instead of
    // 1.37 M proof_verify/s for the first eight jobs after 32 others were freed).  In creation order a set of jobs
    // created back to back gets the queue spread of freshly created streams.
    std::map<int, std::set<std::pair<uint64_t, hipStream_t>>> streams;
    std::map<hipStream_t, uint64_t> stream_index;
    uint64_t next_stream_index = 0;
    std::map<int, std::vector<hipEvent_t>> events;
    // page-locked host staging buffers (one per batch: raw inputs in, statuses out), by size class; hipHostMalloc
    // costs hundreds of microseconds, a batch call must not pay it
    std::multimap<size_t, Pooled> pinned;
    size_t pinned_bytes = 0;
    // the entry of `m` that has waited longest among the size classes other than `cls` (m.end(): there is none)
    template <class M> static typename M::iterator oldest_other(M& m, size_t cls) {
        auto best = m.end();
        for (auto it = m.begin(); it != m.end(); ++it)
            if (it->first != cls && (best == m.end() || it->second.seq < best->second.seq)) best = it;
        return best;
    }
    std::map<int, std::vector<hipEvent_t>> timing_events;      // events WITH timing (stage timers), recycled
    std::map<int, int> dedicated_made;                         // streams with a hardware queue of their own, per device
    std::map<int, int> pooled_made;                            // streams drawn from the runtime's GPU_MAX_HW_QUEUES pool, per device (alive or recycled here)
    std::set<hipStream_t> dedicated;                           // which of the recycled streams are the dedicated ones
    static Pools& get() { static Pools* p = new Pools(); return *p; }      // lives as long as the process
};
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
inline size_t size_class(size_t b) {
    if (b < 256) return 256;
    if (b <= (1u << 20)) { size_t c = 256; while (c < b) c <<= 1; return c; }
    return (b + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
}
constexpr size_t POOL_MAX_BUF = (size_t)256 << 20, POOL_MAX_TOTAL = (size_t)8 << 30;
inline int dmalloc(void** p, size_t b) {
    const size_t cls = size_class(b);
    if (cls <= POOL_MAX_BUF) {
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto& m = P.bufs[current_device()];
        auto it = m.find(cls);
        if (it != m.end()) { *p = it->second.p; m.erase(it); P.pooled_bytes[current_device()] -= cls; return 0; }
    }
    if (hipMalloc(p, cls) == hipSuccess) return 0;
    (void)hipGetLastError();      // an allocation that failed must not be reported again by the next launch's error check
    *p = nullptr;
    return -1;
}
inline void dfree(void* p, size_t b, int dev) {        // dev = the device the buffer was allocated on
    const size_t cls = size_class(b);
    std::vector<void*> evicted;
    if (cls <= POOL_MAX_BUF) {
        // as hfree: the size in use now is kept; when the pool is full, the buffers of other sizes that have waited longest make room
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto& m = P.bufs[dev];
        size_t& tot = P.pooled_bytes[dev];
        while (tot + cls > POOL_MAX_TOTAL && !m.empty()) {
            auto it = Pools::oldest_other(m, cls);
            if (it == m.end()) break;                          // only this size is cached: the pool is simply full of it
            evicted.push_back(it->second.p);
            tot -= it->first;
            m.erase(it);
        }
        if (tot + cls <= POOL_MAX_TOTAL) { m.emplace(cls, Pools::Pooled{p, ++P.clock}); tot += cls; p = nullptr; }
    }
    for (void* e : evicted) (void)hipFree(e);
    if (p) (void)hipFree(p);
}
inline int h2d(void* d, const void* h, size_t b, Stream& s) {
    if (!b) return 0;
    if (hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;   // host staging buffers are transient
}
inline int d2h(void* h, const void* d, size_t b, Stream& s) {
    if (!b) return 0;
    if (hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
inline int dmemset(void* d, int v, size_t b, Stream& s) { return (!b || hipMemsetAsync(d, v, b, s) == hipSuccess) ? 0 : -1; }
inline int d2d_async(void* d, const void* s_, size_t b, Stream& s) {
    return (!b || hipMemcpyAsync(d, s_, b, hipMemcpyDeviceToDevice, s) == hipSuccess) ? 0 : -1;
}
// asynchronous copies from / to page-locked staging memory (hmalloc): the caller keeps the host buffer alive until the
// stream has passed the copy
inline int h2d_async(void* d, const void* h, size_t b, Stream& s) {
    return (!b || hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, s) == hipSuccess) ? 0 : -1;
}
inline int d2h_async(void* h, const void* d, size_t b, Stream& s) {
    return (!b || hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, s) == hipSuccess) ? 0 : -1;
}
constexpr size_t PINNED_POOL_MAX = (size_t)1 << 30;
inline int hmalloc(void** p, size_t b) {
    const size_t cls = size_class(b);
    {
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.pinned.find(cls);
        if (it != P.pinned.end()) { *p = it->second.p; P.pinned.erase(it); P.pinned_bytes -= cls; return 0; }
    }
    return hipHostMalloc(p, cls, hipHostMallocPortable) == hipSuccess ? 0 : -1;      // portable: the pool is shared by every device of the process
}
inline void hfree(void* p, size_t b) {
    const size_t cls = size_class(b);
    std::vector<void*> evicted;
    {
        // the buffer just released is the size the caller is using NOW: it is kept, and if the pool is full, buffers of
        // other sizes (left over from an earlier workload) make room -- otherwise a serving loop whose batch shape
        // changed would pay hipHostMalloc / hipHostFree (milliseconds) on every batch from then on
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        if (cls <= PINNED_POOL_MAX) {
            while (P.pinned_bytes + cls > PINNED_POOL_MAX && !P.pinned.empty()) {
                auto it = Pools::oldest_other(P.pinned, cls);
                if (it == P.pinned.end()) break;               // only this size is cached: the pool is simply full of it
                evicted.push_back(it->second.p);
                P.pinned_bytes -= it->first;
                P.pinned.erase(it);
            }
            if (P.pinned_bytes + cls <= PINNED_POOL_MAX) { P.pinned.emplace(cls, Pools::Pooled{p, ++P.clock}); P.pinned_bytes += cls; p = nullptr; }
        }
    }
    for (void* e : evicted) (void)hipHostFree(e);
    if (p) (void)hipHostFree(p);
}
inline int sync(Stream& s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
inline size_t mem_free_bytes() { size_t f = 0, t = 0; return hipMemGetInfo(&f, &t) == hipSuccess ? f : 0; }
using Event = hipEvent_t;
// ---- hardware queues x scratch: ONE budget, checked (round 5; DESIGN.md 5 rule 6) ---------------------------------------
// The runtime reserves scratch per hardware queue for every wave slot of the chip, sized for the largest kernel frame the
// queue has run: bytes per lane x 64 lanes x wave slots (8192 on MI355X).  Measured (profiles/r03_q_*, r04_g_*): up to
// queues x bytes/lane ~ 48 k everything runs, from ~ 50 k the runtime starts reclaiming scratch between dispatches and every
// rate collapses, from ~ 56 k it ABORTS THE PROCESS (HSA_STATUS_ERROR_OUT_OF_RESOURCES).  The library therefore knows its own
// largest kernel frame -- every kernel registers itself when the library is loaded, hipFuncGetAttributes gives the frame
// -- and never lets (pooled + dedicated) hardware queues x that frame exceed SCRATCH_POOL_FRACTION of the device's memory:
// dedicated queues beyond the budget are not created (the stream comes from the pool), and if the POOL alone is over budget
// (an explicit GPU_MAX_HW_QUEUES=32) the library stops creating streams at the budget -- a job that gets no stream of its
// own shares its context's stream (JobBase::stream) or runs its side streams' work on its main stream.  Slower, never an abort.
struct KernelRegistry {
    std::mutex mu;
    std::vector<const void*> fns;
    static KernelRegistry& get() { static KernelRegistry* r = new KernelRegistry(); return *r; }      // lives as long as the process
};
inline int register_kernel(const void* f) {
    KernelRegistry& r = KernelRegistry::get();
    std::lock_guard<std::mutex> g(r.mu);
    r.fns.push_back(f);
    return (int)r.fns.size();
}
// largest private segment (scratch bytes per lane) of any kernel of this library; once, after the runtime is up
inline size_t library_max_scratch() {
    static const size_t v = []() {
        KernelRegistry& r = KernelRegistry::get();
        std::lock_guard<std::mutex> g(r.mu);
        size_t m = 0;
        for (const void* f : r.fns) {
            hipFuncAttributes a;
            if (hipFuncGetAttributes(&a, f) == hipSuccess) m = std::max(m, (size_t)a.localSizeBytes);
            else (void)hipGetLastError();
        }
        return m;
    }();
    return v;
}
constexpr double SCRATCH_POOL_FRACTION = 0.085;       // of the device's memory: 26 GB of 288 GiB = 47 k (queues x bytes per lane)
struct QueueBudget {
    int total = 0;            // hardware queues (pooled + dedicated) the scratch budget allows
    int pool = 0;             // the runtime's pool: GPU_MAX_HW_QUEUES as the environment has it, 4 when unset
    int dedicated_cap = 0;    // dedicated queues the budget leaves room for: max(0, total - pool)
    size_t scratch = 0;       // bytes per lane of the library's largest kernel frame
};
inline QueueBudget queue_budget(int dev) {
    static std::mutex mu;
    static std::map<int, QueueBudget> cache;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    QueueBudget b;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return b; }
    b.scratch = library_max_scratch();
    const size_t wave_slots = (size_t)p.multiProcessorCount * (size_t)std::max(1, p.maxThreadsPerMultiProcessor / 64);
    const double per_queue = (double)std::max<size_t>(b.scratch, 64) * 64.0 * (double)wave_slots;
    b.total = (int)std::min(64.0, std::max(1.0, (double)p.totalGlobalMem * SCRATCH_POOL_FRACTION / per_queue));
    const char* v = getenv("GPU_MAX_HW_QUEUES");
    b.pool = v && atoi(v) > 0 ? atoi(v) : 4;
    b.dedicated_cap = std::max(0, b.total - b.pool);
    cache[dev] = b;
    return b;
}
// -1: not decided yet (the environment is read at the first stream); 0: off; k > 0: up to k job streams per device
inline std::atomic<int>& dedicated_queues() { static std::atomic<int> v{-1}; return v; }
// 0: a stream; -1: the runtime failed; -2: no stream left within the queue budget (the caller shares one it has)
inline int stream_create(Stream* s) {
    Pools& P = Pools::get();
    const int dev = current_device();
    const QueueBudget qb = queue_budget(dev);           // (takes its own lock; hipFuncGetAttributes on first use)
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.streams[dev];
    if (!v.empty()) { *s = v.begin()->second; v.erase(v.begin()); return 0; }
    // Dedicated hardware queues (bbs_runtime_set_dedicated_queues / BBS_DEDICATED_QUEUES=k): the runtime gives a stream
    // created with a compute-unit mask (here: all ones, no restriction) a hardware queue of its own instead of a share of
    // the GPU_MAX_HW_QUEUES pool.  For a process whose first HIP call came BEFORE this library was loaded (any torch user):
    // the pool is then fixed at the runtime's default of 4, several jobs share a queue and a long narrow kernel blocks the
    // others (measured: 1.30 M proof_verify/s instead of 1.50 M; with dedicated queues 1.50 M whatever the pool,
    // profiles/r04_f_dedicated_queues.log).  At most min(k, what the scratch budget leaves beside the pool) streams per
    // device are created this way; further streams come from the pool.  Such streams are BLOCKING with respect to the legacy
    // default stream (the runtime offers no flags for them): off unless asked for.
    int want = dedicated_queues().load();
    if (want < 0) { const char* e = getenv("BBS_DEDICATED_QUEUES"); want = e ? atoi(e) : 0; if (want == 1) want = 12; if (want < 0) want = 0; if (want > 16) want = 16; dedicated_queues().store(want); }
    if (qb.total > 0) want = std::min(want, qb.dedicated_cap);
    bool made = false;
    if (want > 0 && P.dedicated_made[dev] < want) {
        uint32_t mask[16];
        for (auto& m : mask) m = 0xFFFFFFFFu;
        if (hipExtStreamCreateWithCUMask(s, 16, mask) == hipSuccess) { made = true; P.dedicated_made[dev]++; P.dedicated.insert(*s); }
        else (void)hipGetLastError();
    }
    if (!made) {
        // the pool alone may be over budget (an explicit GPU_MAX_HW_QUEUES): streams map onto its queues round robin, so the
        // number of pooled streams this library creates bounds the number of pooled queues it touches
        if (qb.total > 0 && qb.pool > qb.total - P.dedicated_made[dev] && P.pooled_made[dev] >= std::max(1, qb.total - P.dedicated_made[dev])) return -2;
        if (hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return -1; }
        P.pooled_made[dev]++;
    }
    P.stream_index[*s] = P.next_stream_index++;
    return 0;
}
inline void stream_destroy(Stream& s) {                  // callers synchronise the stream first
    Pools& P = Pools::get();
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.streams[current_device()];
    auto it = P.stream_index.find(s);
    if (it != P.stream_index.end() && v.size() < 256) { v.insert({it->second, s}); return; }
    if (it != P.stream_index.end()) P.stream_index.erase(it);
    if (P.dedicated.erase(s)) P.dedicated_made[current_device()]--; else P.pooled_made[current_device()]--;
    (void)hipStreamDestroy(s);
}
inline int event_create(Event* e) {
    {
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto& v = P.events[current_device()];
        if (!v.empty()) { *e = v.back(); v.pop_back(); return 0; }
    }
    return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess ? 0 : -1;
}
inline void event_destroy(Event& e) {
    Pools& P = Pools::get();
    std::lock_guard<std::mutex> g(P.mu);
    auto& v = P.events[current_device()];
    if (v.size() < 512) v.push_back(e); else (void)hipEventDestroy(e);
}
inline int event_record(Event& e, Stream& s) { return hipEventRecord(e, s) == hipSuccess ? 0 : -1; }
inline int stream_wait(Stream& s, Event& e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess ? 0 : -1; }

// stages may declare `static constexpr int WAVES_PER_EU = k;` to cap their register allocation at
// 512 / k VGPRs so that k wavefronts fit on one SIMD
template <class F, class = void> struct waves_of { static constexpr int v = 1; };
template <class F> struct waves_of<F, std::void_t<decltype(F::WAVES_PER_EU)>> { static constexpr int v = F::WAVES_PER_EU; };

template <class F, class A>
__global__ void __launch_bounds__(64, waves_of<F>::v) k_stage(A a, size_t nthreads) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t < nthreads) F::run(a, t);
}
// every kernel of the library is known when the library has been loaded (queue budget above)
template <class F, class A> struct KernelEntry { static const int token; };
template <class F, class A> const int KernelEntry<F, A>::token = register_kernel(reinterpret_cast<const void*>(&k_stage<F, A>));
template <class F, class A>
inline int launch(Stream& s, const A& a, size_t nthreads) {
    if (!nthreads) return 0;
    (void)&KernelEntry<F, A>::token;
    const unsigned blocks = (unsigned)((nthreads + 63) / 64);
    hipLaunchKernelGGL((k_stage<F, A>), dim3(blocks), dim3(64), 0, s, a, nthreads);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// HIP events recorded on the stream the kernels are launched on; read back after ONE
// synchronisation so that timing does not serialise host and device between stages
struct EventList {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    int dev = 0;
    explicit EventList(size_t cap) : ev(cap, nullptr), dev(current_device()) {
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto& v = P.timing_events[dev];
        for (auto& e : ev) {
            if (!v.empty()) { e = v.back(); v.pop_back(); }
            else (void)hipEventCreate(&e);
        }
    }
    ~EventList() {
        Pools& P = Pools::get();
        std::lock_guard<std::mutex> g(P.mu);
        auto& v = P.timing_events[dev];
        for (auto& e : ev) if (e) { if (v.size() < 4096) v.push_back(e); else (void)hipEventDestroy(e); }
    }
    int record(Stream& s) { return (used < ev.size() && hipEventRecord(ev[used++], s) == hipSuccess) ? 0 : -1; }
    int finish(Stream& s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
    float ms(size_t a, size_t b) { float m = 0; (void)hipEventElapsedTime(&m, ev[a], ev[b]); return m; }
};
#endif
}  // namespace rt

// device buffer with ownership
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) rt::dfree(p, bytes, dev); p = nullptr; bytes = 0; }
    int alloc(size_t b) { release(); bytes = b; dev = rt::current_device(); if (rt::dmalloc(&p, b)) { p = nullptr; bytes = 0; return -1; } return 0; }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(bytes, o.bytes); std::swap(dev, o.dev); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// page-locked host staging buffer with ownership
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { release(); }
    void release() { if (p) rt::hfree(p, bytes); p = nullptr; bytes = 0; }
    int alloc(size_t b) { release(); bytes = b; return rt::hmalloc(&p, b); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// no internal state (stages.hpp ST_PENDING / ST_PAIRING) may leave the library: an item still in one was never decided
inline bool statuses_final(const int8_t* st, size_t n) {
    for (size_t i = 0; i < n; i++) if (st[i] == ST_PENDING || st[i] == ST_PAIRING) return false;
    return true;
}

// =============================================================================================
// host-side packing helpers
// =============================================================================================
inline uint32_t le32(const uint8_t* b) {
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}
inline void put_le32(uint8_t* b, uint32_t v) {
    b[0] = (uint8_t)v; b[1] = (uint8_t)(v >> 8); b[2] = (uint8_t)(v >> 16); b[3] = (uint8_t)(v >> 24);
}

// SoA array of `words` 32-bit words per item.  Filled item by item into an item-major staging area (the writes of
// one item are contiguous; writing word-major directly strides by n words and misses the cache on every store), then
// transposed in 16-item tiles into the word-major layout the kernels read: soa().
struct Soa {
    std::vector<uint32_t> aos, v;
    size_t n = 0, words = 0;
    bool sealed = false;
    void init(size_t words_, size_t n_) { words = words_; n = n_; aos.assign(words_ * n_, 0u); v.clear(); sealed = false; }
    uint32_t& at(size_t w, size_t i) { return aos[i * words + w]; }
    size_t bytes() const { return aos.size() * 4; }
    const std::vector<uint32_t>& soa() {
        if (!sealed) {
            v.resize(aos.size());
            constexpr size_t T = 16;
            for (size_t i0 = 0; i0 < n; i0 += T) {
                const size_t m = std::min(T, n - i0);
                for (size_t w = 0; w < words; w++) {
                    uint32_t* dst = v.data() + w * n + i0;
                    const uint32_t* src = aos.data() + i0 * words + w;
                    for (size_t k = 0; k < m; k++) dst[k] = src[k * words];
                }
            }
            sealed = true;
        }
        return v;
    }
};

// canonical scalar (32 B LE) into SoA words [w0, w0+8); returns false if >= r
template <class P>
inline bool pack_fe(Soa& s, size_t w0, size_t i, const uint8_t* le) {
    uint32_t l[P::NC];
    for (int k = 0; k < P::NC; k++) l[k] = le32(le + 4 * k);
    for (int k = 0; k < P::NC; k++) s.at(w0 + k, i) = l[k];
    return limbs_lt_mod<P>(l);
}
template <class C>
inline bool pack_g1(Soa& s, size_t w0, size_t i, const uint8_t* xy) {
    constexpr int NC = C::FpP::NC;
    bool a = pack_fe<typename C::FpP>(s, w0, i, xy);
    bool b = pack_fe<typename C::FpP>(s, w0 + NC, i, xy + 4 * NC);
    return a && b;
}
inline void unpack_words_le(const uint32_t* v, size_t n, size_t w0, size_t i, int words, uint8_t* out) {
    for (int k = 0; k < words; k++) put_le32(out + 4 * k, v[(w0 + k) * n + i]);
}
inline void unpack_words_le(const std::vector<uint32_t>& v, size_t n, size_t w0, size_t i, int words, uint8_t* out) {
    unpack_words_le(v.data(), n, w0, i, words, out);
}

// ragged bytes -> device pool + u32 offset / len
struct BytePool {
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off, len;
    bool build(size_t n, const uint8_t* data, const uint64_t* offs) {
        off.assign(n, 0); len.assign(n, 0);
        if (!offs) { bytes.assign(4, 0); return true; }        // all empty
        const uint64_t total = offs[n] - offs[0];
        if (total > 0xF0000000ull) return false;
        for (size_t i = 0; i < n; i++) {
            if (offs[i + 1] < offs[i]) return false;
            off[i] = (uint32_t)(offs[i] - offs[0]);
            len[i] = (uint32_t)(offs[i + 1] - offs[i]);
        }
        bytes.assign(data ? data + offs[0] : nullptr, data ? data + offs[n] : nullptr);
        if (bytes.size() != total) return false;
        bytes.resize(bytes.size() + 4, 0);
        return true;
    }
};

// =============================================================================================
// context
// =============================================================================================
// jobs alive per DEVICE, whatever context they belong to: what the AUTO form of a job goes by (Ctx::latency_form_now).
// A process that drives one GPU through several contexts -- an issuer with one context per message count, a BLS12-381 and
// a BN254 engine side by side (mixed.py) -- shares the chip among all their jobs; counting per context would give every
// one of them the latency form (more instructions for a shorter path) exactly where throughput decides.
inline std::atomic<int>& device_live_jobs(int dev) {
    static std::atomic<int> live[64];
    return live[(unsigned)dev & 63u];
}
struct IJob;
struct bbs_ctx {
    int curve = 0;
    bool stage_timing = false;       // jobs created from now on record HIP events around every stage of every run
    virtual ~bbs_ctx() {}
};

template <class C>
struct Ctx : bbs_ctx {
    static constexpr int N = C::FpP::N;          // internal limbs
    static constexpr int FPB = 4 * C::FpP::NC;   // bytes of a canonical field element
    int device = 0;
    rt::Stream stream{};
    bool stream_ready = false;
    int win_bits = 8;                // width of the tables that are built
    int win_bits_requested = 0;      // bbs_ctx_set_window_bits; 0 (the default) = choose at set_generators from the free device memory
    // host state
    bool gens_set = false, pk_set = false, sk_set = false, dst_too_long = false;
    int L = 0;
    std::vector<G1Aff<C>> gens;      // Q1, H_1..H_L  (Montgomery)
    std::vector<uint8_t> api_id;
    G2Aff<C> pk{};
    uint32_t sk[8] = {0};
    // device state
    CtxConsts<C> hc{};               // host mirror
    DevBuf d_consts, d_tables, d_winbase, d_bases;
    bool consts_dirty = true;
    // batch verification (pippenger.hpp): off by default = every item gets its own pairing product
    bool batch_verify = false;
    // caller vouches that every G1 input is in the prime-order subgroup: variable-base multiplications of the
    // verification paths use the GLV split where the curve has it (g1.hpp); off by default
    bool points_in_subgroup = false;
    // latency form of a job (bbs_ctx_set_latency_mode): proof_verify's T1 as three multiplications on three lanes instead
    // of one joint chain, the two Miller loops of a pairing product on separate lane groups -- a shorter critical path for
    // more work.  0 = never, 1 = always, 2 = AUTO (default): a job gets the latency form iff at most one other job of
    // the DEVICE (any context, device_live_jobs) is alive when it is created, i.e. when it will have (most of) the chip to itself
    int latency_mode = 2;
    std::atomic<int> live_jobs{0};
    bool latency_form_now() const { return latency_mode == 1 || (latency_mode == 2 && device_live_jobs(device).load() <= 2); }
    bool fix_tree = false;           // bbs_ctx_set_fixed_base_tree: the fixed-base sums as trees of affine additions
    uint32_t rlc_seed[8] = {0};
    uint64_t rlc_counter = 0;
    std::mutex mu;                   // uploads may come from several host threads: counter and constant sync

    // secret seed of the next batch: SHA-256(context seed || counter)
    void next_rlc_seed(uint32_t* out8) {
        std::lock_guard<std::mutex> g(mu);
        Sha256 s;
        sha256_init(s);
        for (int k = 0; k < 8; k++) sha256_word(s, rlc_seed[k]);
        sha256_u64be(s, rlc_counter++);
        sha256_final(s, out8);
    }
    int set_batch_verification(int enabled, const uint8_t* seed32) {
        if (enabled) {
            uint8_t buf[32];
            if (seed32) std::memcpy(buf, seed32, 32);
            else {
                FILE* f = std::fopen("/dev/urandom", "rb");
                const size_t got = f ? std::fread(buf, 1, 32, f) : 0;
                if (f) std::fclose(f);
                if (got != 32) return BBS_E_STATE;
            }
            for (int k = 0; k < 8; k++) rlc_seed[k] = le32(buf + 4 * k);
            rlc_counter = 0;
        }
        batch_verify = enabled != 0;
        return BBS_OK;
    }

    int init(int dev) {
        device = dev;
        if (rt::set_device(dev)) return BBS_E_NO_DEVICE;
        if (const int src = rt::stream_create(&stream)) return src == -2 ? BBS_E_NO_RESOURCES : BBS_E_HIP;
        stream_ready = true;
        std::memset(&hc, 0, sizeof(hc));
        for (int j = 0; j < N; j++) { hc.p1.x.v[j] = C::K::P1X_M[j]; hc.p1.y.v[j] = C::K::P1Y_M[j]; }
        for (int k = 0; k < 3; k++) for (int m = 0; m < 6; m++) for (int c2 = 0; c2 < 2; c2++)
            for (int j = 0; j < N; j++) hc.frob[k][m][c2][j] = C::K::FROB[k][m][c2][j];
        build_schedule<C>(hc.sched);
        if (!build_line_table<C>(g2_generator<C>(), hc.tab_bp2)) return BBS_E_ARG;
        hc.tab_pk.q_is_identity = 1;
        hc.tab_pk.n_lines = 0;
        if (d_consts.alloc(sizeof(CtxConsts<C>))) return BBS_E_NOMEM;
        return BBS_OK;
    }
    ~Ctx() override {
        (void)rt::set_device(device);
        if (stream_ready) { (void)rt::sync(stream); rt::stream_destroy(stream); }
        volatile uint32_t* s = sk;                       // do not leave the secret key in freed host memory
        for (int k = 0; k < 8; k++) s[k] = 0;
    }

    int use() { return rt::set_device(device) ? BBS_E_HIP : BBS_OK; }
    size_t table_bytes() const { return d_tables.bytes + d_winbase.bytes + d_bases.bytes + d_consts.bytes; }

    // domain prefix:  Z_pad || compress(pk) || I2OSP(L,8) || compress(Q1) || compress(H_i).. || api_id
    void rebuild_hash() {
        HashCtx& h = hc.hash;
        std::memset(&h, 0, sizeof(h));
        std::vector<uint8_t> dst(api_id);
        const char* suf = "H2S_";
        dst.insert(dst.end(), suf, suf + 4);
        dst_too_long = dst.size() > 255;
        if (!dst_too_long) { std::memcpy(h.dst_h2s, dst.data(), dst.size()); h.dst_h2s_len = (uint32_t)dst.size(); }
        if (!(gens_set && pk_set)) { consts_dirty = true; return; }
        Sha256 s;
        xmd48_begin(s);
        uint8_t buf[4 * FPB];
        g2_compress<C>(pk, buf);
        sha256_bytes(s, buf, 2 * FPB);
        sha256_u64be(s, (uint64_t)L);
        for (const auto& g : gens) { g1_compress_host<C>(g, buf); sha256_bytes(s, buf, FPB); }
        sha256_bytes(s, api_id.data(), (uint32_t)api_id.size());
        for (int k = 0; k < 8; k++) h.dom_mid[k] = s.h[k];
        h.dom_mid_total = s.total - s.fill;
        h.dom_tail_len = s.fill;
        for (uint32_t k = 0; k < s.fill; k++) h.dom_tail[k] = (uint8_t)(s.w[k >> 2] >> ((3 - (k & 3)) * 8));
        consts_dirty = true;
    }

    int sync_consts() {
        std::lock_guard<std::mutex> g(mu);
        if (!consts_dirty) return BBS_OK;
        hc.tables = d_tables.as<uint32_t>();
        if (rt::h2d(d_consts.p, &hc, sizeof(hc), stream)) return BBS_E_HIP;
        consts_dirty = false;
        return BBS_OK;
    }

    int set_generators(const uint8_t* g, size_t count, const uint8_t* aid, size_t aid_len);   // op_prim.hpp

    int set_pk_internal(const G2Aff<C>& q) {
        if (!g2_on_curve<C>(q) || !g2_in_subgroup<C>(q)) return BBS_E_PUBLIC_KEY;
        if (!build_line_table<C>(q, hc.tab_pk)) return BBS_E_PUBLIC_KEY;
        pk = q;
        pk_set = true;
        rebuild_hash();
        return BBS_OK;
    }
    int set_public_key(const uint8_t* b, int is_identity) {
        G2Aff<C> q{};
        q.inf = is_identity != 0;
        if (!q.inf) {
            if (!b) return BBS_E_ARG;
            using P = typename C::FpP;
            if (!fe_from_le_bytes<P>(b, q.x.c0) || !fe_from_le_bytes<P>(b + FPB, q.x.c1) ||
                !fe_from_le_bytes<P>(b + 2 * FPB, q.y.c0) || !fe_from_le_bytes<P>(b + 3 * FPB, q.y.c1))
                return BBS_E_PUBLIC_KEY;
        } else {
            q.x = f2_zero<C>(); q.y = f2_zero<C>();
        }
        sk_set = false;
        return set_pk_internal(q);
    }
    int set_secret_key(const uint8_t* sk32) {
        if (!sk32) return BBS_E_ARG;
        uint32_t l[8];
        for (int k = 0; k < 8; k++) l[k] = le32(sk32 + 4 * k);
        if (!limbs_lt_mod<typename C::FrP>(l)) return BBS_E_ARG;
        std::memcpy(sk, l, sizeof(sk));
        int rc = set_pk_internal(g2_mul<C>(g2_generator<C>(), l));     // key_gen.rs:83-90
        if (rc) return rc;
        sk_set = true;
        return BBS_OK;
    }
    int get_public_key(uint8_t* out, int* inf) {
        if (!pk_set) return BBS_E_STATE;
        using P = typename C::FpP;
        if (inf) *inf = pk.inf ? 1 : 0;
        if (out) {
            if (pk.inf) std::memset(out, 0, 4 * FPB);
            else {
                fe_to_le_bytes<P>(pk.x.c0, out); fe_to_le_bytes<P>(pk.x.c1, out + FPB);
                fe_to_le_bytes<P>(pk.y.c0, out + 2 * FPB); fe_to_le_bytes<P>(pk.y.c1, out + 3 * FPB);
            }
        }
        return BBS_OK;
    }
};

// =============================================================================================
// jobs
// =============================================================================================
// Completion order (bbs_jobs_wait_any / bbs_job_poll).  Behind everything a run enqueues on a job's main stream the
// runtime places ONE host function (hipLaunchHostFunc); it stamps the job with the next sequence number and wakes the
// waiters -- the host thread sleeps on a condition variable, nothing polls the device.  A job may be armed more than once
// per run (the submit forms add their copies to page-locked memory behind run()): it is complete when every armed
// notification has fired.  The job's destructor synchronises its streams, so a notification never outlives its job.
struct CompletionHub {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t seq = 0;
    static CompletionHub& get() { static CompletionHub* h = new CompletionHub(); return *h; }      // lives as long as the process
};
struct bbs_job {
    size_t n = 0;
    virtual ~bbs_job() {}
    std::atomic<uint32_t> armed{0};              // notifications placed behind this job's work (by its submitting thread)
    std::atomic<uint32_t> fired{0};
    std::atomic<uint64_t> done_seq{0};           // position in the process-wide completion order (of the last notification)
    static void on_complete(void* p) {
        bbs_job* j = static_cast<bbs_job*>(p);
        CompletionHub& h = CompletionHub::get();
        std::lock_guard<std::mutex> g(h.mu);
        j->done_seq.store(++h.seq, std::memory_order_relaxed);
        j->fired.fetch_add(1, std::memory_order_release);
        h.cv.notify_all();
    }
    // A run that could not be enqueued completely (a launch failed, the notification could not be placed): the job counts as
    // FINISHED WITH AN ERROR -- bbs_job_poll says 1, bbs_jobs_wait_any hands it out at once, and bbs_job_wait / wait_any
    // return BBS_E_HIP for it instead of delivering anything.  Without this a notification that was counted but never placed
    // left the job "running" for ever (wait_any slept on it), and a re-run that failed before arming left the PREVIOUS run's
    // completion standing, i.e. stale results reported as this run's.  Cleared when the next run starts.
    std::atomic<bool> failed{false};
    void mark_failed(bool take_back_arm) {
        CompletionHub& h = CompletionHub::get();
        std::lock_guard<std::mutex> g(h.mu);
        if (take_back_arm) armed.fetch_sub(1, std::memory_order_relaxed);
        failed.store(true, std::memory_order_release);
        h.cv.notify_all();
    }
    // behind what has been enqueued on the main stream so far (the main stream joins the side streams before its last stage).
    // Counted BEFORE the host function is placed (it may fire before hipLaunchHostFunc returns: counting afterwards could lose
    // the wake-up); taken back if it could not be placed.
    int arm_completion() {
        armed.fetch_add(1, std::memory_order_relaxed);
#ifdef BBS_HOST_TWIN
        on_complete(this);
        return BBS_OK;
#else
        if (!use() && hipLaunchHostFunc(stream(), &bbs_job::on_complete, this) == hipSuccess) return BBS_OK;
        (void)hipGetLastError();
        mark_failed(true);
        return BBS_E_HIP;
#endif
    }
    bool hold_arm = false;                       // the submit functions arm once, behind their own copies (not run())
    bool ever_run() const { return armed.load(std::memory_order_relaxed) != 0 || failed.load(std::memory_order_acquire); }
    bool completed() const {
        if (failed.load(std::memory_order_acquire)) return true;
        const uint32_t a = armed.load(std::memory_order_relaxed);
        return a != 0 && fired.load(std::memory_order_acquire) == a;
    }
    // aux = k > 0: the stage runs on the job's k-th side stream (k = 1, 2), concurrently with the main-stream stages
    // that follow the fork point (the place of the first stage of that stream in the list); join: bit k - 1 set = the main
    // stream first waits for everything issued on side stream k (1: the first, 2: the second, 3: both)
    static constexpr int N_AUX = 2;
    struct Stage { const char* name; std::function<int()> launch; int aux = 0; int join = 0; };
    std::vector<Stage> stages;
    virtual int use() = 0;
    virtual rt::Stream& stream() = 0;
    virtual rt::Stream& stream_aux(int k = 1) = 0;
    virtual int fork_aux(int k = 1) = 0;         // side stream k waits for what the main stream has issued
    virtual int join_aux(int k = 1) = 0;         // main stream waits for what side stream k has issued
    virtual int sync_aux() = 0;                  // host waits for every side stream that exists
    virtual int reset() = 0;                     // restore the pre-run status so the job can run again
    virtual int fetch_status(int8_t*) = 0;
    // submit form (bbs_core_*_submit): the statuses are copied to page-locked memory behind the last stage, and
    // wait() hands them to the caller's buffer `deliver_to`
    int8_t* deliver_to = nullptr;
    virtual int enqueue_status_fetch() = 0;
    virtual int deliver() = 0;
    // sign / proof_gen: the produced records follow the statuses to page-locked memory; deliver() unpacks them into the
    // caller's buffers
    virtual int enqueue_result_fetch() { return BBS_OK; }
    virtual void set_result_targets(uint8_t*, uint8_t*, uint64_t*) {}
    virtual int set_octet_form() { return BBS_E_ARG; }    // sign / proof_gen: results leave as octet strings (before run)
    virtual size_t device_bytes() const { return 0; }     // device memory this job holds (allocation size classes)
    virtual int fetch_signatures(uint8_t*) { return BBS_E_ARG; }
    virtual int fetch_proofs(uint8_t*, uint8_t*, uint64_t*) { return BBS_E_ARG; }
    // One pass over the stages.  ev != nullptr: one event before the first stage, then a (start, stop) pair around
    // every stage, each recorded on the stream that stage is launched on (1 + 2 * stages events per pass).
    int run_recorded(rt::EventList* ev) {
        if (reset()) return BBS_E_HIP;
        if (ev && ev->record(stream())) return BBS_E_HIP;
        unsigned forked = 0;
        // FAULT INJECTION for the fail-closed tests (tests/parity_cases.py::check_fail_closed_submit): the stages named in
        // BBS_FAULT_SKIP_STAGE (comma-separated) are not launched, so the items they would have decided stay undecided and
        // every way out of the library must refuse with BBS_E_STATE.  Read at every run; unset in any real deployment.
        const char* skip = getenv("BBS_FAULT_SKIP_STAGE");
        auto skipped = [skip](const char* name) {
            if (!skip || !*skip) return false;
            const size_t len = std::strlen(name);
            for (const char* p = skip; (p = std::strstr(p, name)) != nullptr; p += len)
                if ((p == skip || p[-1] == ',') && (p[len] == 0 || p[len] == ',')) return true;
            return false;
        };
        for (auto& s : stages) {
            if (s.aux < 0 || s.aux > N_AUX) return BBS_E_STATE;
            if (skipped(s.name)) continue;
            if (s.aux && !((forked >> s.aux) & 1u)) { if (fork_aux(s.aux)) return BBS_E_HIP; forked |= 1u << s.aux; }
            for (int k = 1; k <= N_AUX; k++)
                if (((s.join >> (k - 1)) & 1) && ((forked >> k) & 1u)) { if (join_aux(k)) return BBS_E_HIP; }
            rt::Stream& st = s.aux ? stream_aux(s.aux) : stream();
            if (ev && ev->record(st)) return BBS_E_HIP;
            if (s.launch()) return BBS_E_HIP;
            if (ev && ev->record(st)) return BBS_E_HIP;
        }
        return BBS_OK;
    }
    bool timed = false;                          // set from bbs_ctx::stage_timing when the job is created
    std::unique_ptr<rt::EventList> tev;          // events of the LAST run (timed jobs)
    bool results_wanted = false;                 // submit form of sign / proof_gen: the records follow the statuses
    int run() {
        failed.store(false, std::memory_order_release);
        if (use()) { mark_failed(false); return BBS_E_HIP; }
        if (timed) tev.reset(new rt::EventList(1 + 2 * stages.size()));
        int rc = run_recorded(timed ? tev.get() : nullptr);
        // a submit-form job that is run AGAIN (bbs_job_run): what bbs_job_wait delivers must be this run's statuses and
        // records, not the first run's -- the copies to page-locked memory are enqueued behind every run
        if (!rc && deliver_to) {
            rc = enqueue_status_fetch();
            if (!rc && results_wanted) rc = enqueue_result_fetch();
        }
        if (!rc && !hold_arm) rc = arm_completion();
        else if (rc) mark_failed(false);
        return rc;
    }
    // stage durations of the last run of a timed job; call after wait()
    int stage_times(float* total_ms, float* kernel_ms, int cap, int* n_stages) {
        const int ns = (int)stages.size();
        if (n_stages) *n_stages = ns;
        if (!timed || !tev || tev->used != 1 + 2 * (size_t)ns) return BBS_E_STATE;
        if (use() || rt::sync(stream()) || sync_aux()) return BBS_E_HIP;
        if (total_ms) *total_ms = tev->ms(0, 2 * (size_t)ns);
        for (int k = 0; kernel_ms && k < ns && k < cap; k++) kernel_ms[k] = tev->ms(1 + 2 * (size_t)k, 2 + 2 * (size_t)k);
        return BBS_OK;
    }
    int wait() {
        if (use() || rt::sync(stream())) return BBS_E_HIP;
        if (failed.load(std::memory_order_acquire)) return BBS_E_HIP;     // the last run was not enqueued completely: nothing to deliver
        return deliver_to ? deliver() : BBS_OK;
    }
};

template <class C>
struct JobBase : bbs_job {
    Ctx<C>* ctx;
    DevBuf d_status, d_status0;      // per-item statuses; d_status0 = what the ingest stage decided (restored before every run)
    HostBuf h_status;                // page-locked landing area of the submit form
    // the batch as the caller handed it over: page-locked image (source of the ONE asynchronous H2D copy) and the same
    // image on the device, which the ingest stage reads and where headers / presentation headers stay.  Members of
    // this base so that they outlive the destructor's stream synchronisation.
    HostBuf h_raw;
    DevBuf d_raw;
    std::vector<std::pair<void*, size_t>> zero_on_reset;   // device arrays cleared before every run (fail closed)
    std::vector<std::unique_ptr<DevBuf>> bufs;
    // every job owns its streams: independent jobs (batches) of one context overlap on the GPU
    rt::Stream main{}, aux[bbs_job::N_AUX]{};
    rt::Event ev_fork[bbs_job::N_AUX]{}, ev_join[bbs_job::N_AUX]{};
    bool main_ready = false, aux_ready[bbs_job::N_AUX] = {false, false};
    bool aux_shared[bbs_job::N_AUX] = {false, false};      // the queue budget left no stream for this side stream: its stages run on the main stream
    bool latency_form = false;       // decided when the job is created (Ctx::latency_form_now)
    explicit JobBase(Ctx<C>* c) : ctx(c) {
        main_ready = (ctx->use() == 0) && (rt::stream_create(&main) == 0); timed = c->stage_timing;
        ctx->live_jobs.fetch_add(1);
        device_live_jobs(ctx->device).fetch_add(1);
        latency_form = ctx->latency_form_now();
    }
    ~JobBase() override {
        ctx->live_jobs.fetch_sub(1);
        device_live_jobs(ctx->device).fetch_sub(1);
        (void)ctx->use();            // streams and buffers go back to this device's pools
        for (int k = 0; k < bbs_job::N_AUX; k++)
            if (aux_ready[k]) { rt::sync(aux[k]); rt::event_destroy(ev_fork[k]); rt::event_destroy(ev_join[k]); rt::stream_destroy(aux[k]); }
        if (main_ready) { rt::sync(main); rt::stream_destroy(main); }
    }
    int use() override { return ctx->use(); }
    size_t device_bytes() const override {
        size_t t = 0;
        auto add = [&](const DevBuf& b) { if (b.p) t += rt::size_class(b.bytes); };
        add(d_status); add(d_status0); add(d_raw);
        for (const auto& b : bufs) add(*b);
        return t;
    }
    rt::Stream& stream() override { return main_ready ? main : ctx->stream; }
    int ensure_aux(int k) {                      // k = 1 .. N_AUX
        if (k < 1 || k > bbs_job::N_AUX) return -1;
        if (aux_ready[k - 1] || aux_shared[k - 1]) return 0;
        const int rc = rt::stream_create(&aux[k - 1]);
        if (rc == -2) { aux_shared[k - 1] = true; return 0; }     // over the queue budget: share the main stream (in order, no fork / join)
        if (rc) return -1;
        if (rt::event_create(&ev_fork[k - 1])) { rt::stream_destroy(aux[k - 1]); return -1; }
        if (rt::event_create(&ev_join[k - 1])) { rt::event_destroy(ev_fork[k - 1]); rt::stream_destroy(aux[k - 1]); return -1; }
        aux_ready[k - 1] = true;
        return 0;
    }
    rt::Stream& stream_aux(int k = 1) override { ensure_aux(k); return (k >= 1 && k <= bbs_job::N_AUX && aux_ready[k - 1]) ? aux[k - 1] : stream(); }
    int fork_aux(int k = 1) override {
        if (ensure_aux(k)) return -1;
        if (!aux_ready[k - 1]) return 0;
        return (rt::event_record(ev_fork[k - 1], stream()) || rt::stream_wait(aux[k - 1], ev_fork[k - 1])) ? -1 : 0;
    }
    int join_aux(int k = 1) override {
        if (ensure_aux(k)) return -1;
        if (!aux_ready[k - 1]) return 0;
        return (rt::event_record(ev_join[k - 1], aux[k - 1]) || rt::stream_wait(stream(), ev_join[k - 1])) ? -1 : 0;
    }
    int sync_aux() override {
        int rc = 0;
        for (int k = 0; k < bbs_job::N_AUX; k++) if (aux_ready[k] && rt::sync(aux[k])) rc = -1;
        return rc;
    }
    // device-to-device, asynchronous: back-to-back runs of one job never wait for the host
    int reset() override {
        if (rt::d2d_async(d_status.p, d_status0.p, n, stream())) return BBS_E_HIP;
        for (auto& z : zero_on_reset) if (rt::dmemset(z.first, 0, z.second, stream())) return BBS_E_HIP;
        return BBS_OK;
    }
    int fetch_status(int8_t* out) override {
        if (use() || rt::sync(stream())) return BBS_E_HIP;
        if (rt::d2h(out, d_status.p, n, stream())) return BBS_E_HIP;
        return statuses_final(out, n) ? BBS_OK : BBS_E_STATE;
    }
    // the stream has been synchronised: BBS_E_STATE unless every item of the last run has been decided.  The record
    // buffers of sign / proof_gen are written by the last stage of a run; before that they hold whatever the pooled
    // allocation held, which must never reach a caller.
    int require_decided() {
        std::vector<int8_t> st(n);
        if (n && rt::d2h(st.data(), d_status.p, n, stream())) return BBS_E_HIP;
        return statuses_final(st.data(), n) ? BBS_OK : BBS_E_STATE;
    }
    int enqueue_status_fetch() override {
        if (!h_status.p && h_status.alloc(n ? n : 1)) return BBS_E_NOMEM;
        return rt::d2h_async(h_status.p, d_status.p, n, stream()) ? BBS_E_HIP : BBS_OK;
    }
    int deliver() override {                       // the stream has been synchronised
        if (!h_status.p) return BBS_E_STATE;
        if (!statuses_final(h_status.template as<int8_t>(), n)) return BBS_E_STATE;
        if (n) std::memcpy(deliver_to, h_status.p, n);
        return BBS_OK;
    }
    // upload a host vector, return device pointer (owned by the job)
    template <class T>
    T* up(const std::vector<T>& v, int& rc) {
        bufs.emplace_back(new DevBuf());
        DevBuf& b = *bufs.back();
        if (b.alloc(v.size() * sizeof(T))) { rc = BBS_E_NOMEM; return nullptr; }
        if (rt::h2d(b.p, v.data(), v.size() * sizeof(T), stream())) { rc = BBS_E_HIP; return nullptr; }
        return b.as<T>();
    }
    template <class T>
    T* scratch(size_t count, int& rc) {
        bufs.emplace_back(new DevBuf());
        DevBuf& b = *bufs.back();
        if (b.alloc(count * sizeof(T))) { rc = BBS_E_NOMEM; return nullptr; }
        return b.as<T>();
    }
    // the initial statuses come from the device's ingest stage (no host validation pass, no synchronous copy)
    int finish_setup_device() {
        if (d_status.alloc(n ? n : 1) || d_status0.alloc(n ? n : 1)) return BBS_E_NOMEM;
        if (rt::dmemset(d_status.p, (uint8_t)ST_PENDING, n, stream())) return BBS_E_HIP;
        return ctx->sync_consts();
    }
    template <class T>
    int down(std::vector<T>& v, const T* dptr) {
        return rt::d2h(v.data(), dptr, v.size() * sizeof(T), stream()) ? BBS_E_HIP : BBS_OK;
    }
};


// =============================================================================================
// staging image: the batch as the caller handed it over, one page-locked buffer, one asynchronous copy
// =============================================================================================
// Ragged input arrays of one batch inside the staging image: offsets rebased to 0, sections 16-byte aligned.
// offsets of a ragged section without entries (a batch without messages): one zero that may be read at index 0
inline const uint64_t* zero_off1() { static const uint64_t z[1] = {0}; return z; }
struct RaggedIn {
    const uint64_t* off;      // n + 1 caller offsets, or nullptr (every item empty)
    const uint8_t* data;
    size_t elem;              // bytes per element
    uint64_t total = 0;       // elements
    size_t at_off = 0, at_data = 0;
    bool offsets_only = false; // the elements themselves are produced on the device (no data behind the offsets)
    // false: offsets decrease, or data missing
    bool measure(size_t n) {
        total = 0;
        if (!off) return true;
        for (size_t i = 0; i < n; i++) if (off[i + 1] < off[i]) return false;
        total = off[n] - off[0];
        if (total > ((uint64_t)1 << 36)) return false;       // no batch holds 2^36 elements: garbage offsets, and total * elem must not wrap
        return total == 0 || data != nullptr || offsets_only;
    }
    void place(size_t& cur, size_t n) {
        at_off = cur; cur += ((n + 1) * 8 + 15) & ~(size_t)15;
        at_data = cur; if (!offsets_only) cur += ((size_t)total * elem + 4 + 15) & ~(size_t)15;
    }
    void fill(uint8_t* img, size_t n) const {
        uint64_t* o = reinterpret_cast<uint64_t*>(img + at_off);
        if (!off) { std::memset(o, 0, (n + 1) * 8); return; }
        const uint64_t b = off[0];
        for (size_t i = 0; i <= n; i++) o[i] = off[i] - b;
        if (total && !offsets_only) std::memcpy(img + at_data, data + b * elem, (size_t)total * elem);
    }
};

// Fixed-size records at the start of the image + the ragged sections behind them; allocates the job's page-locked and
// device buffers, fills the host image and enqueues the ONE host-to-device copy on the job's stream.
template <class J>
inline int stage_image(J* job, size_t n, const uint8_t* records, size_t rec_bytes, std::initializer_list<RaggedIn*> sections,
                       RaggedIn* extra = nullptr, size_t extra_count = 0) {      // extra: a section ragged over extra_count entries
    size_t cur = (n * rec_bytes + 15) & ~(size_t)15;
    for (RaggedIn* s : sections) s->place(cur, n);
    if (extra) extra->place(cur, extra_count);
    if (job->h_raw.alloc(cur) || job->d_raw.alloc(cur)) return BBS_E_NOMEM;
    uint8_t* img = job->h_raw.template as<uint8_t>();
    if (n && rec_bytes) std::memcpy(img, records, n * rec_bytes);
    for (RaggedIn* s : sections) s->fill(img, n);
    if (extra) extra->fill(img, extra_count);
    return rt::h2d_async(job->d_raw.p, img, cur, job->stream()) ? BBS_E_HIP : BBS_OK;
}

// msg_to_scalars on the device (codec_dev.hpp MsgHash): the nm raw messages of a batch, placed in the staging image as
// section `mb`, become nm canonical scalars [message][8 words] -- the array the ingest stages read messages from.
// dst_too_long = 1 (and no launch): api_id || "MAP_MSG_TO_SCALAR_AS_HASH_" exceeds 255 bytes, the reference panics.
template <class C, class J>
inline uint32_t* hash_raw_messages(J* job, Ctx<C>* ctx, const RaggedIn& mb, size_t nm, MsgHashArgs& ma, int& dst_too_long, int& rc) {
    static const char SUFFIX[] = "MAP_MSG_TO_SCALAR_AS_HASH_";
    const size_t dl = ctx->api_id.size() + sizeof(SUFFIX) - 1;
    uint32_t* out = job->template scratch<uint32_t>(std::max<size_t>(nm, 1) * 8, rc);
    if (rc) return nullptr;
    dst_too_long = dl > 255 ? 1 : 0;
    if (dst_too_long || !nm) return out;
    const uint8_t* dimg = job->d_raw.template as<uint8_t>();
    std::memset(&ma, 0, sizeof(ma));
    ma.nm = nm; ma.off = reinterpret_cast<const uint64_t*>(dimg + mb.at_off); ma.bytes = dimg + mb.at_data; ma.out = out;
    std::memcpy(ma.dst, ctx->api_id.data(), ctx->api_id.size());
    std::memcpy(ma.dst + ctx->api_id.size(), SUFFIX, sizeof(SUFFIX) - 1);
    ma.dst_len = (uint32_t)dl;
    if (rt::launch<MsgHash<C>>(job->stream(), ma, nm)) rc = BBS_E_HIP;
    return out;
}

// =============================================================================================
// batch verification plumbing shared by proof_verify and verify (pippenger.hpp)
// =============================================================================================
template <class C>
struct BvState {
    RlcPrepArgs<C> prep{};
    PipArgs<C> pip{};            // host twin: the lane-per-bucket stages
    PipCoopArgs<C> coop{};       // device: the workgroup-cooperative kernel
    PipTileSumArgs<C> tsum{};
    const int8_t* batch_ok = nullptr;      // [n_checks] verdicts of the combined checks
    int n_checks = 0;
    PairArgs<C> pa_sum{};
};

#if !defined(BBS_HOST_TWIN)
namespace rt {
template <class C> struct PipKernelEntry { static const int token; };
template <class C> const int PipKernelEntry<C>::token = register_kernel(reinterpret_cast<const void*>(&k_pip_window<C>));
template <class C>
inline int launch_pip_windows(Stream& s, const PipCoopArgs<C>& a) {
    const size_t units = (size_t)a.M * a.NW * a.n_tiles;
    (void)&PipKernelEntry<C>::token;
    if (!units || !a.n) return 0;
    hipLaunchKernelGGL((k_pip_window<C>), dim3((unsigned)units), dim3(PIP_WG), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace rt
#endif

// one pairing product check as stages on the job's main (aux = 0) or second stream.  split: the two Miller loops of an
// item on separate wavefronts and the final exponentiation as its own stage (the job's latency form; always for the
// sixteen combined checks of batch verification, which are the narrow tail of their job)
template <class C, class J>
void add_pairing_stages(J* j, PairArgs<C>* pargs, int aux, const char* nm_miller, const char* nm_final, const char* nm_dist, bool force_split = false,
                        int join_first = 0,         // join_first: the (main-stream) first stage waits for the second stream
                        bool allow_split = true) {  // false: one fused kernel also in the latency form (see op_pv.hpp)
    (void)nm_miller; (void)nm_final; (void)nm_dist; (void)force_split; (void)allow_split;
    const size_t first = j->stages.size();
    struct JoinFirst { J* j; size_t first; int join; ~JoinFirst() { if (join && j->stages.size() > first) j->stages[first].join = 1; } } jf{j, first, join_first};
#ifdef BBS_HOST_TWIN
    j->stages.push_back({nm_miller, [j, pargs, aux]() { return rt::launch<PairMiller<C>>(aux ? j->stream_aux() : j->stream(), *pargs, pargs->n * 2); }, aux, 0});
    j->stages.push_back({nm_final, [j, pargs, aux]() { return rt::launch<PairFinal<C>>(aux ? j->stream_aux() : j->stream(), *pargs, pargs->n); }, aux, 0});
#else
    if ((j->latency_form && allow_split) || force_split) {
        // the two Miller loops of every item on separate wavefronts, then product + final exponentiation (stages.hpp)
        pargs->single = 0;
        j->stages.push_back({nm_miller, [j, pargs, aux]() { return rt::launch<PairMillerHalf<C>>(aux ? j->stream_aux() : j->stream(), *pargs, ((pargs->n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 128); }, aux, 0});
        j->stages.push_back({nm_final, [j, pargs, aux]() { return rt::launch<PairFinalDist<C>>(aux ? j->stream_aux() : j->stream(), *pargs, ((pargs->n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 64); }, aux, 0});
        return;
    }
#if BBS_PAIR_SPLIT2
    // both Miller loops on one six-lane group (the whole register file), then the final exponentiation two wavefronts per SIMD
    pargs->single = 1;
    j->stages.push_back({"pair_miller_both", [j, pargs, aux]() { return rt::launch<PairMillerBoth<C>>(aux ? j->stream_aux() : j->stream(), *pargs, ((pargs->n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 64); }, aux, 0});
    j->stages.push_back({nm_final, [j, pargs, aux]() { return rt::launch<PairFinalDist<C>>(aux ? j->stream_aux() : j->stream(), *pargs, ((pargs->n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 64); }, aux, 0});
#else
    pargs->single = 0;
    j->stages.push_back({nm_dist, [j, pargs, aux]() { return rt::launch<PairDist<C>>(aux ? j->stream_aux() : j->stream(), *pargs, ((pargs->n + GRP_PER_WAVE - 1) / GRP_PER_WAVE) * 64); }, aux, 0});
#endif
#endif
}

// Batch verification (pippenger.hpp), in two parts so that a job can place them on different streams:
//   add_batch_combination : RlcPrep (coefficients, the two point sets item-major), the bucket-method sums of both sets per
//                           8-bit window, and the 16 pairing products  e(sum_w pts0, pk) e(+-sum_w pts1, BP2)  side by side
//                           -- on the job's second stream (aux = 1: proof_verify, whose points are inputs) or on the main one;
//   add_batch_decision    : (main stream; joins the second) the per-item kernel `fallback` (gate: status == ST_PAIRING) with the
//                           16 verdicts: all 1 -> its lanes write Ok(true) and return, otherwise they compute the item's own product.
template <class C, class J>
int add_batch_combination(J* j, BvState<C>* bv, Ctx<C>* ctx, size_t n, const CtxConsts<C>* cc, const uint32_t* pa, const uint32_t* pb,
                          int canonical, const int8_t* gate_arr, int gate, int negate_b, int aux, bool with_prep = true) {
    constexpr int N = C::FpP::N;
    constexpr int NW = 16, M = 2;
    const size_t n_pad = (n + 3) & ~(size_t)3, nn = std::max<size_t>(n, 1);
    int rc = BBS_OK;
    RlcPrepArgs<C>& pr = bv->prep;
    uint8_t* dig = j->template scratch<uint8_t>((size_t)NW * n_pad + 4, rc);
    uint32_t* ppts = j->template scratch<uint32_t>((size_t)M * nn * 2 * N, rc);
    uint32_t* out = j->template scratch<uint32_t>((size_t)M * 2 * N * NW, rc);          // [M][2N][NW]: NW affine sums per set
    // [0 .. NW) gates of the NW combined checks (all 1), [NW .. 2 NW) their results
    std::vector<int8_t> fl(2 * NW, 0);
    for (int k = 0; k < NW; k++) fl[k] = 1;
    int8_t* flags = j->up(fl, rc);
    uint32_t* fm_sum = j->template scratch<uint32_t>((size_t)2 * 12 * N * NW, rc);
    if (rc) return rc;
    if (rt::dmemset(dig, 0, (size_t)NW * n_pad, j->stream())) return BBS_E_HIP;
    j->zero_on_reset.push_back({flags + NW, (size_t)NW});          // a combined check that never ran reads as failed
    pr.n = n; pr.n_pad = n_pad; pr.pa = pa; pr.pb = pb; pr.canonical = canonical; pr.gate_arr = gate_arr; pr.gate = gate;
    pr.dig = dig; pr.ppts = ppts;
    ctx->next_rlc_seed(pr.seed);
    bv->batch_ok = flags + NW; bv->n_checks = NW;
    PairArgs<C>& ps = bv->pa_sum;
    ps.n = NW; ps.cc = cc; ps.pa = out; ps.pb = out + (size_t)2 * N * NW; ps.negate_b = negate_b; ps.canonical = 0;
    ps.gate_arr = flags; ps.gate = 1; ps.out = flags + NW; ps.fmiller = fm_sum;
    auto strm = [j, aux]() -> rt::Stream& { return aux ? j->stream_aux() : j->stream(); };
    // with_prep = false: the caller's own stage writes the digits and the item-major points (proof_verify's PvChallengeBv)
    if (with_prep) j->stages.push_back({"rlc_prep", [j, bv, strm]() { return rt::launch<RlcPrep<C>>(strm(), bv->prep, j->n); }, aux, 0});
#ifdef BBS_HOST_TWIN
    PipArgs<C>& pp = bv->pip;
    pp.n = n; pp.n_pad = n_pad; pp.M = M; pp.NW = NW; pp.ppts = ppts; pp.dig = dig; pp.out = out;
    pp.list = j->template scratch<uint32_t>((size_t)M * NW * nn, rc);
    pp.buckets = j->template scratch<uint32_t>((size_t)3 * N * M * NW * PIP_NB, rc);
    pp.segs = j->template scratch<uint32_t>((size_t)3 * N * M * NW * (PIP_NB / PIP_SEG), rc);
    pp.wins = j->template scratch<uint32_t>((size_t)3 * N * M * NW, rc);
    if (rc) return rc;
    j->stages.push_back({"pip_buckets", [bv, strm]() { return rt::launch<PipBuckets<C>>(strm(), bv->pip, (size_t)bv->pip.M * bv->pip.NW * PIP_NB); }, aux, 0});
    j->stages.push_back({"pip_segments", [bv, strm]() { return rt::launch<PipSegments<C>>(strm(), bv->pip, (size_t)bv->pip.M * bv->pip.NW * (PIP_NB / PIP_SEG)); }, aux, 0});
    j->stages.push_back({"pip_window_sums", [bv, strm]() { return rt::launch<PipWindowSums<C>>(strm(), bv->pip, (size_t)bv->pip.M * bv->pip.NW); }, aux, 0});
#else
    PipCoopArgs<C>& co = bv->coop;
    co.n = n; co.n_pad = n_pad; co.M = M; co.NW = NW; co.n_tiles = (int)((nn + PIP_TILE - 1) / PIP_TILE); co.ppts = ppts; co.dig = dig;
    co.tile_sums = j->template scratch<uint32_t>((size_t)3 * N * M * NW * co.n_tiles, rc);
    if (rc) return rc;
    PipTileSumArgs<C>& ts = bv->tsum;
    ts.M = M; ts.NW = NW; ts.n_tiles = n ? co.n_tiles : 0; ts.shift = 0; ts.tile_sums = co.tile_sums; ts.out = out; ts.wins = nullptr;
    co.out_aff = (n && co.n_tiles == 1) ? out : nullptr;      // one tile: the kernel's thread 0 normalises the window sum itself
    j->stages.push_back({"pip_windows", [bv, strm]() { return rt::launch_pip_windows<C>(strm(), bv->coop); }, aux, 0});
    if (!co.out_aff)
        j->stages.push_back({"pip_tile_sums", [bv, strm]() { return rt::launch<PipTileSums<C>>(strm(), bv->tsum, (size_t)bv->tsum.M * bv->tsum.NW); }, aux, 0});
#endif
    add_pairing_stages<C>(j, &bv->pa_sum, aux, "rlc_pair_miller", "rlc_pair_final_exp", "rlc_pairing_6lane", true);
    return BBS_OK;
}
template <class C, class J>
void add_batch_decision(J* j, BvState<C>* bv, PairArgs<C>* fallback, int joins_aux) {
    // no kernel of its own: the per-item kernel's lanes read the 16 verdicts first -- all passed: write Ok(true) and leave
    fallback->batch_ok = bv->batch_ok; fallback->n_checks = bv->n_checks;
    add_pairing_stages<C>(j, fallback, 0, "fallback_pair_miller", "fallback_pair_final_exp", "fallback_pairing_6lane", false, joins_aux);
}
