// runtime.hip -- what every batch shares: the thread-local error text, the process-wide cache of device / pinned blocks,
// streams and events (a call of edlibAlign() is a batch of one: without the cache it pays ~30 hipMalloc / hipFree round
// trips), the helper-thread budget, device selection.  Host code only.
#include "engine.hpp"
#include <sched.h>
#include <cstdarg>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace edlib_amd {

// ------------------------------------------------------------------- errors

std::string& last_error() { static thread_local std::string s; return s; }
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    last_error() = buf;
}

// ------------------------------------------------------------ device pool

namespace {
struct Blk { void* p; unsigned long long tick; };              // tick: when the block came back (the cache evicts the oldest)
struct Pool {
    std::mutex mu;
    static const int kMaxDev = kMaxDevices;
    unsigned long long tick = 0;
    std::vector<Blk> blocks[kMaxDev][48];       // [device][log2 size class]; back() = most recently returned
    std::vector<hipStream_t> streams[kMaxDev];
    std::vector<hipEvent_t> events[kMaxDev];
    size_t cachedBytes = 0;
    std::vector<Blk> pinned[48];                // [log2 size class], host memory: device independent
    size_t cachedPinned = 0;
};
Pool& pool() { static Pool* p = new Pool; return *p; }     // leaked on purpose: no teardown-order hazards
const size_t kPoolMaxBlock = 64u << 20;                     // larger blocks go straight back to the driver
const size_t kPoolMaxCached = 1024u << 20;
const size_t kPinnedMaxBlock = 256u << 20;
const size_t kPinnedMaxCached = 512u << 20;
int size_class(size_t bytes, size_t* rounded) {
    int c = 8;                                              // 256 B minimum
    while (((size_t)1 << c) < bytes) ++c;
    *rounded = (size_t)1 << c;
    return c;
}
}  // namespace

static thread_local bool tl_quarantine = false;
void pool_quarantine(bool on) { tl_quarantine = on; }

bool pool_enabled() { return !tl_quarantine; }

void pool_trim() {
    Pool& P = pool();
    std::vector<std::pair<int, void*>> dev; std::vector<void*> pin; std::vector<std::pair<int, hipStream_t>> str;
    std::vector<std::pair<int, hipEvent_t>> ev;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (int d = 0; d < Pool::kMaxDev; ++d) {
            for (auto& v : P.blocks[d]) { for (const Blk& b : v) dev.push_back({d, b.p}); v.clear(); }
            for (hipStream_t s : P.streams[d]) str.push_back({d, s});
            P.streams[d].clear();
            for (hipEvent_t e : P.events[d]) ev.push_back({d, e});
            P.events[d].clear();
        }
        for (auto& v : P.pinned) { for (const Blk& b : v) pin.push_back(b.p); v.clear(); }
        P.cachedBytes = 0; P.cachedPinned = 0;
    }
    for (auto& b : dev) (void)hipFree(b.second);
    for (void* p : pin) (void)hipHostFree(p);
    for (auto& s : str) { DeviceGuard g(s.first); (void)hipStreamDestroy(s.second); }
    for (auto& e : ev) { DeviceGuard g(e.first); (void)hipEventDestroy(e.second); }
}

hipError_t pool_alloc(void** p, size_t bytes, size_t* granted) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (pool_enabled() && bytes <= kPoolMaxBlock && dev < Pool::kMaxDev) {
        size_t r; const int c = size_class(bytes, &r);
        {
            std::lock_guard<std::mutex> g(pool().mu);
            auto& v = pool().blocks[dev][c];
            if (!v.empty()) { *p = v.back().p; v.pop_back(); pool().cachedBytes -= r; *granted = r; return hipSuccess; }
        }
        *granted = r;
        return hipMalloc(p, r);
    }
    *granted = bytes;
    return hipMalloc(p, bytes);
}

void pool_free(void* p, size_t granted) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device; else (void)hipGetLastError();
    if (pool_enabled() && granted <= kPoolMaxBlock && dev < Pool::kMaxDev && (granted & (granted - 1)) == 0) {
        size_t r; const int c = size_class(granted, &r);
        // A full cache makes room by letting its OLDEST blocks go: the block coming back was in use a moment ago, what has sat
        // here longest belongs to batches that are gone.  (It used to refuse the block coming back: once bigger batches had
        // filled the cache, every temporary of a later batch's steps went hipMalloc -> hipFree.)
        std::vector<void*> evict;
        {
            std::lock_guard<std::mutex> g(pool().mu);
            Pool& P = pool();
            while (P.cachedBytes + r > kPoolMaxCached) {
                int bd = -1, bc = -1;
                for (int d = 0; d < Pool::kMaxDev; ++d)
                    for (int k = 0; k < 48; ++k)
                        if (!P.blocks[d][k].empty() && (bd < 0 || P.blocks[d][k].front().tick < P.blocks[bd][bc].front().tick)) { bd = d; bc = k; }
                if (bd < 0) break;
                evict.push_back(P.blocks[bd][bc].front().p);
                P.blocks[bd][bc].erase(P.blocks[bd][bc].begin());
                P.cachedBytes -= (size_t)1 << bc;
            }
            P.blocks[dev][c].push_back(Blk{p, ++P.tick}); P.cachedBytes += r;
        }
        for (void* q : evict) (void)hipFree(q);
        return;
    }
    (void)hipFree(p);
}

hipError_t pinned_alloc(void** p, size_t bytes, size_t* granted) {
    if (pool_enabled() && bytes <= kPinnedMaxBlock) {
        size_t r; const int c = size_class(bytes, &r);
        {
            std::lock_guard<std::mutex> g(pool().mu);
            auto& v = pool().pinned[c];
            if (!v.empty()) { *p = v.back().p; v.pop_back(); pool().cachedPinned -= r; *granted = r; return hipSuccess; }
        }
        *granted = r;
        return hipHostMalloc(p, r, hipHostMallocDefault);
    }
    *granted = bytes;
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

void pinned_free(void* p, size_t granted) {
    if (pool_enabled() && granted <= kPinnedMaxBlock && (granted & (granted - 1)) == 0) {
        size_t r; const int c = size_class(granted, &r);
        std::vector<void*> evict;                          // (as pool_free: the oldest blocks make room)
        {
            std::lock_guard<std::mutex> g(pool().mu);
            Pool& P = pool();
            while (P.cachedPinned + r > kPinnedMaxCached) {
                int bc = -1;
                for (int k = 0; k < 48; ++k)
                    if (!P.pinned[k].empty() && (bc < 0 || P.pinned[k].front().tick < P.pinned[bc].front().tick)) bc = k;
                if (bc < 0) break;
                evict.push_back(P.pinned[bc].front().p);
                P.pinned[bc].erase(P.pinned[bc].begin());
                P.cachedPinned -= (size_t)1 << bc;
            }
            P.pinned[c].push_back(Blk{p, ++P.tick}); P.cachedPinned += r;
        }
        for (void* q : evict) (void)hipHostFree(q);
        return;
    }
    (void)hipHostFree(p);
}

hipError_t pool_stream(hipStream_t* s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < Pool::kMaxDev) {
        std::lock_guard<std::mutex> g(pool().mu);
        auto& v = pool().streams[dev];
        if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// hands a stream of device `dev` back without touching the HIP runtime (destructors of thread-local contexts run at
// thread / process exit, when the runtime may be on its way out): the stream is cached or simply left to the process
void pool_stream_put(int dev, hipStream_t s) {
    if (dev >= 0 && dev < Pool::kMaxDev) {
        std::lock_guard<std::mutex> g(pool().mu);
        if (pool().streams[dev].size() < 16) pool().streams[dev].push_back(s);
    }
}

void pool_stream_release(hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < Pool::kMaxDev) {
        std::lock_guard<std::mutex> g(pool().mu);
        if (pool().streams[dev].size() < 16) { pool().streams[dev].push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

hipError_t pool_event(hipEvent_t* e, int* device) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *device = dev;
    if (dev < Pool::kMaxDev && pool_enabled()) {
        std::lock_guard<std::mutex> g(pool().mu);
        auto& v = pool().events[dev];
        if (!v.empty()) { *e = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipEventCreate(e);
}

void pool_event_release(hipEvent_t e, int dev) {
    if (dev >= 0 && dev < Pool::kMaxDev && pool_enabled()) {
        std::lock_guard<std::mutex> g(pool().mu);
        if (pool().events[dev].size() < 64) { pool().events[dev].push_back(e); return; }
    }
    (void)hipEventDestroy(e);
}

// Helper threads of a host fan-out (marshalling, packing): EDLIB_AMD_HOST_THREADS if set, else at most `cap` and at
// most the CPUs this process may really use (cgroup quota / affinity: the GPU boxes show 256 logical CPUs behind a
// 16-CPU quota, and 8 ranks share it).
int host_threads(int cap) {
    static const int avail = [] {
        if (const char* env = getenv("EDLIB_AMD_HOST_THREADS")) { const int v = atoi(env); if (v >= 1) return v; }
        int n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        {   // the affinity mask (taskset, container cpusets)
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && c < n) n = c; }
        }
        bool v2 = false;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {              // cgroup v2 quota
            char a[64] = {0}; long long per = 0;
            v2 = true;
            if (fscanf(f, "%63s %lld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) {
                const long long q = (atoll(a) + per / 2) / per;
                if (q >= 1 && q < n) n = (int)q;
            }
            fclose(f);
        }
        if (!v2) {                                                         // cgroup v1: cfs quota / period
            long long quota = -1, per = 0;
            if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
            if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &per) != 1) per = 0; fclose(f); }
            if (quota > 0 && per > 0) { const long long q = (quota + per / 2) / per; if (q >= 1 && q < n) n = (int)q; }
        }
        return n;
    }();
    return std::max(1, std::min(cap, avail));
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// Device of edlibAlign() and of the one-shot entry points when EDLIB_AMD_DEVICES is unset: EDLIB_AMD_DEVICE
// if given, else the calling thread's current HIP device (a host application that selected a GPU keeps it).
int default_device() {
    const int ndev = device_count();
    if (const char* env = getenv("EDLIB_AMD_DEVICE")) {
        char* e; const long d = strtol(env, &e, 10);
        if (e != env && d >= 0 && d < ndev) return (int)d;
    }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
    return (cur >= 0 && cur < ndev) ? cur : 0;
}


}  // namespace edlib_amd
