// engine.hip -- batch orchestration (host) for the MI355X edit-distance engine.
//
// The reference does, per call (edlib.cpp:146-301): transform -> Peq -> k-doubling
// scan -> (start locations) -> (path).  Here the same phases run once per BATCH:
//   phase 1  distance + end locations   reads-per-lane kernel (shared target, <= 4
//                                       target symbols, query <= 256) or
//                                       block-per-lane kernel (everything else)
//   phase 2  HW start locations         reverse SHW units on the block-per-lane kernel
//                                       (edlib.cpp:230-266)
//   phase 3  alignment path             NW units with column store + traceback kernel
//                                       (edlib.cpp:276-289, 1161-1213)
// Work is narrowed like in the reference (Ukkonen band, thresholds that grow until they hold the
// distance), but per batch and with thresholds chosen for the hardware: reads run a banded first pass
// at a small k and only the leftovers a full pass (runReads), NW pairs climb lane-ring sizes
// (solveGlobalDistances).  Thresholds only steer work: every result is a function of the full DP
// matrix, so the user's k merely filters it (SURVEY.md §7 "results are band-independent").
#include "engine.hpp"
#include "flat_results.hpp"
#include <sched.h>

#if defined(__x86_64__)
#include <immintrin.h>                 // build_tables: 16 target bytes per step through the alphabet scan (host)
#endif
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <system_error>
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
struct Pool {
    std::mutex mu;
    static const int kMaxDev = 16;
    std::vector<void*> blocks[kMaxDev][48];     // [device][log2 size class]
    std::vector<hipStream_t> streams[kMaxDev];
    std::vector<hipEvent_t> events[kMaxDev];
    size_t cachedBytes = 0;
    std::vector<void*> pinned[48];              // [log2 size class], host memory: device independent
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

static bool pool_enabled() { return !tl_quarantine; }

void pool_trim() {
    Pool& P = pool();
    std::vector<std::pair<int, void*>> dev; std::vector<void*> pin; std::vector<std::pair<int, hipStream_t>> str;
    std::vector<std::pair<int, hipEvent_t>> ev;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (int d = 0; d < Pool::kMaxDev; ++d) {
            for (auto& v : P.blocks[d]) { for (void* p : v) dev.push_back({d, p}); v.clear(); }
            for (hipStream_t s : P.streams[d]) str.push_back({d, s});
            P.streams[d].clear();
            for (hipEvent_t e : P.events[d]) ev.push_back({d, e});
            P.events[d].clear();
        }
        for (auto& v : P.pinned) { for (void* p : v) pin.push_back(p); v.clear(); }
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
            if (!v.empty()) { *p = v.back(); v.pop_back(); pool().cachedBytes -= r; *granted = r; return hipSuccess; }
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
        std::lock_guard<std::mutex> g(pool().mu);
        if (pool().cachedBytes + r <= kPoolMaxCached) { pool().blocks[dev][c].push_back(p); pool().cachedBytes += r; return; }
    }
    (void)hipFree(p);
}

hipError_t pinned_alloc(void** p, size_t bytes, size_t* granted) {
    if (pool_enabled() && bytes <= kPinnedMaxBlock) {
        size_t r; const int c = size_class(bytes, &r);
        {
            std::lock_guard<std::mutex> g(pool().mu);
            auto& v = pool().pinned[c];
            if (!v.empty()) { *p = v.back(); v.pop_back(); pool().cachedPinned -= r; *granted = r; return hipSuccess; }
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
        std::lock_guard<std::mutex> g(pool().mu);
        if (pool().cachedPinned + r <= kPinnedMaxCached) { pool().pinned[c].push_back(p); pool().cachedPinned += r; return; }
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

// ------------------------------------------------------------------- tables

// byte x byte equality matrix of EqualityDefinition (edlib.cpp:63-94); only built when there are additional
// equalities (the kernels take "no matrix" as the identity)
static void build_eq8(std::vector<uint8_t>& eq8, const EdlibEqualityPair* eqs, int neq) {
    eq8.assign(256 * 256, 0);
    for (int a = 0; a < 256; ++a) eq8[a * 256 + a] = 1;
    for (int i = 0; i < neq; ++i) {
        const int a = (uint8_t)eqs[i].first, b = (uint8_t)eqs[i].second;
        eq8[a * 256 + b] = eq8[b * 256 + a] = 1;
    }
}

// Position of the first byte at or after `from` that is not in the set described by the two nibble tables (or `n`):
// byte b is in the set iff lo[b & 15] & hi[b >> 4] != 0 (every distinct high nibble of the set owns one bit: exact for
// sets with at most 8 distinct high nibbles -- DNA, protein, text).  16 bytes per step with pshufb where the CPU has it:
// the alphabet scan of a 5 Mb target was 1-2 ms of every single edlibAlign() call against it.
#if defined(__x86_64__)
__attribute__((target("ssse3")))
static long long first_outside_ssse3(const uint8_t* p, long long from, long long n, const uint8_t* lo, const uint8_t* hi)
{
    const __m128i L = _mm_loadu_si128(reinterpret_cast<const __m128i*>(lo)), H = _mm_loadu_si128(reinterpret_cast<const __m128i*>(hi));
    const __m128i nib = _mm_set1_epi8(0x0f), zero = _mm_setzero_si128();
    long long i = from;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i));
        const __m128i a = _mm_shuffle_epi8(L, _mm_and_si128(v, nib));
        const __m128i b = _mm_shuffle_epi8(H, _mm_and_si128(_mm_srli_epi16(v, 4), nib));
        const int miss = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(a, b), zero));
        if (miss) return i + __builtin_ctz((unsigned)miss);
    }
    return i;                                                   // the tail (< 16 bytes) is the caller's
}
#endif

static void build_tables(Tables& tab, const uint8_t* targets, long long totalTargetBytes,
                         const EdlibEqualityPair* eqs, int neq) {
    memset(tab.presence, 0, sizeof tab.presence);
    memset(tab.tlut, 0, sizeof tab.tlut);
    memset(tab.idToByte, 0, sizeof tab.idToByte);
    bool seen[256] = {false};
    tab.sigmaT = 0;
    uint8_t lo[16] = {0}, hi[16] = {0}; int hiBit[16]; int hiBits = 0; bool nibbleOk = true;
    for (int h = 0; h < 16; ++h) hiBit[h] = -1;
#if defined(__x86_64__)
    static const bool haveSsse3 = __builtin_cpu_supports("ssse3");
#else
    static const bool haveSsse3 = false;
#endif
    long long i = 0;
    while (i < totalTargetBytes) {
#if defined(__x86_64__)
        if (haveSsse3 && nibbleOk && tab.sigmaT > 0) {          // skip what is already known, 16 bytes at a time
            i = first_outside_ssse3(targets, i, totalTargetBytes, lo, hi);
            if (i >= totalTargetBytes) break;
        }
#endif
        const uint8_t b = targets[i++];
        if (!seen[b]) {                                          // first appearance order (transformSequences, edlib.cpp:1417-1462)
            seen[b] = true;
            tab.tlut[b] = (uint8_t)tab.sigmaT;
            tab.idToByte[tab.sigmaT] = b;
            tab.presence[b >> 5] |= 1u << (b & 31);
            ++tab.sigmaT;
            if (hiBit[b >> 4] < 0) { if (hiBits < 8) hiBit[b >> 4] = hiBits++; else nibbleOk = false; }
            if (nibbleOk) { hi[b >> 4] = (uint8_t)(1u << hiBit[b >> 4]); lo[b & 15] |= (uint8_t)(1u << hiBit[b >> 4]); }
        }
    }
    // EqualityDefinition (edlib.cpp:63-94) on raw bytes.  The reference keeps a pair only
    // when both characters occur in that call's alphabet; a pair whose characters do not
    // occur can never be consulted, so the byte-level relation gives identical DP matrices.
    tab.eq8.clear();
    if (neq > 0) build_eq8(tab.eq8, eqs, neq);
    for (int q = 0; q < 256; ++q) {
        uint16_t mask = 0;
        for (int s = 0; s < tab.sigmaT && s < 16; ++s)
            if (tab.eq8.empty() ? (q == tab.idToByte[s]) : (tab.eq8[q * 256 + tab.idToByte[s]] != 0)) mask |= (uint16_t)(1u << s);
        tab.eqtbl[q] = mask;
    }
}

// --------------------------------------------------------- alphabetLength

// Number of distinct byte values in query (and, for non-shared batches, target):
// the alphabetLength field (edlib.cpp:162, transformSequences :1417-1462).
// One workgroup per unit: aligned 16-byte loads (the pools are padded by 16 bytes and 16-byte aligned), every byte
// marks its entry of a 256-entry table in LDS (lanes that write the same entry write the same value), and the count
// of marked entries is the answer.  (Round 1 / 2 kept a 256-bit set per thread in registers: ~25 VALU ops per byte
// behind dword loads, 2.0 ms for 100,000 pairs of 10 kb -- 1 TB/s; this form is bound by the loads.)
__global__ void __launch_bounds__(256)
alphabet_count_kernel(const uint8_t* __restrict__ qpool, const long long* __restrict__ qoff,
                      const uint8_t* __restrict__ tpool, const long long* __restrict__ toff,
                      int shared, const uint32_t* __restrict__ basePresence,
                      const int* __restrict__ unitIdx, int* __restrict__ out)
{
    __shared__ uint32_t s_seen[256];
    s_seen[threadIdx.x] = 0u;
    __syncthreads();
    const int u = unitIdx[blockIdx.x];
    auto scan = [&](const uint8_t* pool, long long lo, long long hi) {
        for (long long c = (lo >> 4) + threadIdx.x; (c << 4) < hi; c += 256) {
            const uint4 v = *reinterpret_cast<const uint4*>(pool + (c << 4));
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            const long long at = c << 4;
            if (at >= lo && at + 16 <= hi) {
#pragma unroll
                for (int k = 0; k < 16; ++k) s_seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1u;
            } else {                                                     // first / last chunk of the sequence
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (at + k >= lo && at + k < hi) s_seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1u;
            }
        }
    };
    scan(qpool, qoff[u], qoff[u + 1]);
    if (!shared) scan(tpool, toff[u], toff[u + 1]);
    __syncthreads();
    bool present = s_seen[threadIdx.x] != 0u;
    if (shared) present = present || ((basePresence[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1u);
    const int n = __syncthreads_count(present ? 1 : 0);
    if (threadIdx.x == 0) out[blockIdx.x] = n;
}

// The same count for batches of SHORT sequences: a wave per unit, four units per workgroup (a workgroup per unit has
// three idle waves and a launch of 262,144 workgroups for as many 150-base pairs: ~1 ms on the side stream, which the
// collection of a flat batch then waited for).  Each wave owns 256 bytes of the table.
__global__ void __launch_bounds__(256)
alphabet_count_short_kernel(const uint8_t* __restrict__ qpool, const long long* __restrict__ qoff,
                            const uint8_t* __restrict__ tpool, const long long* __restrict__ toff,
                            int shared, const uint32_t* __restrict__ basePresence,
                            const int* __restrict__ unitIdx, int n, int* __restrict__ out)
{
    __shared__ uint32_t s_seen[4][64];                               // per wave: 256 one-byte marks
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wv;
    if (slot >= n) return;
    uint8_t* const seen = reinterpret_cast<uint8_t*>(s_seen[wv]);
    s_seen[wv][lane] = 0u;
    const int u = unitIdx[slot];
    auto scan = [&](const uint8_t* pool, long long lo, long long hi) {
        for (long long c = (lo >> 4) + lane; (c << 4) < hi; c += 64) {
            const uint4 v = *reinterpret_cast<const uint4*>(pool + (c << 4));
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            const long long at = c << 4;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (at + k >= lo && at + k < hi) seen[(d[k >> 2] >> (8 * (k & 3))) & 0xffu] = 1;
        }
    };
    scan(qpool, qoff[u], qoff[u + 1]);
    if (!shared) scan(tpool, toff[u], toff[u + 1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t w = s_seen[wv][lane];                                   // four marks
    if (shared) {
        const uint32_t bits = (basePresence[lane >> 3] >> (4 * (lane & 7))) & 0xfu;
        w |= (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
    }
    int cnt = ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) out[slot] = cnt;
}

// overflow census of the reads path: how many slots need the exact second pass
__global__ void __launch_bounds__(256)
count_flags_kernel(const int* __restrict__ flags, int n, int* __restrict__ counter)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags[i]) atomicAdd(counter, 1);
}

// full-height pass of the reads path: a lane's threshold drops to what a scan of the target's first columns found
__global__ void __launch_bounds__(256)
seed_thresholds_kernel(int* __restrict__ kinit, const int* __restrict__ best, const int* __restrict__ cnt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cnt[i] > 0 && best[i] < kinit[i]) kinit[i] = best[i];
}

// --------------------------------------------------------------- Batch: init

Batch::~Batch() {
    DeviceGuard guard(device_);
    if (side_) { (void)hipStreamSynchronize(side_); pool_stream_release(side_); }
    if (stream_) { (void)hipStreamSynchronize(stream_); pool_stream_release(stream_); }
    wideGateRelease();           // (behind the synchronisation: a failed run may still have had a wide launch in flight)
    for (auto& p : scanEvents_) { pool_event_release(p.first, device_); pool_event_release(p.second, device_); }
}

int roundup(int x, int q) { return (x + q - 1) / q * q; }

// EDLIB_AMD_DEBUG: host wall time between named points of a run (stderr)
struct Lap {
    bool on; std::chrono::steady_clock::time_point t;
    Lap() : on(getenv("EDLIB_AMD_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
    void operator()(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[edlib_amd] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

int Batch::init(const char* queries, const long long* qoff, int n, const char* targets,
                const long long* toff, int numTargets, EdlibAlignConfig cfg, int device)
{
    if (n < 0 || (numTargets != 1 && numTargets != n)) { set_error("bad batch shape"); return 1; }
    const int ndev = device_count();
    if (ndev == 0) { set_error("no usable HIP device (this library has no CPU fallback)"); return 1; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return 1; }
    cfg_ = cfg;
    if (cfg.additionalEqualities && cfg.additionalEqualitiesLength > 0)
        eqs_.assign(cfg.additionalEqualities, cfg.additionalEqualities + cfg.additionalEqualitiesLength);
    cfg_.additionalEqualities = eqs_.empty() ? nullptr : eqs_.data();
    cfg_.additionalEqualitiesLength = (int)eqs_.size();
    device_ = device;
    n_ = n;
    shared_ = (numTargets == 1);
    qoff_.assign(qoff, qoff + n + 1);
    toff_.assign(toff, toff + numTargets + 1);
    for (int u = 0; u < n; ++u) {
        if (qoff_[u + 1] < qoff_[u] || qoff_[u + 1] - qoff_[u] > 0x7fffffffLL) { set_error("bad query offsets"); return 1; }
    }
    for (int u = 0; u < numTargets; ++u) {
        if (toff_[u + 1] < toff_[u] || toff_[u + 1] - toff_[u] > 0x7fffffffLL) { set_error("bad target offsets"); return 1; }
    }
    const long long qbytes = qoff_[n] - qoff_[0], tbytes = toff_[numTargets] - toff_[0];
    build_tables(tab_, reinterpret_cast<const uint8_t*>(targets) + toff_[0], tbytes,
                 eqs_.data(), (int)eqs_.size());

    pool_quarantine(false);
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    EDLIB_AMD_HIP(pool_stream(&stream_));
    EDLIB_AMD_HIP(evRun0_.create()); EDLIB_AMD_HIP(evRun1_.create());

    // resident inputs (pools are rebased to offset 0).  Offsets and the small tables -- and, for small batches,
    // the sequences themselves -- go up as ONE block through pinned staging with one asynchronous copy: a call
    // of edlibAlign() is a batch of one, and nine blocking hipMemcpy calls from pageable memory were most of
    // what it cost.  Large pools are copied straight from the caller's memory.
    const long long qb = qoff_[0], tb = toff_[0];
    for (auto& v : qoff_) v -= qb;
    for (auto& v : toff_) v -= tb;
    {
        const bool inlinePools = qbytes + tbytes <= (1 << 20);
        size_t at = 0;
        auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 15) & ~(size_t)15; return o; };
        const size_t oQoff = take(qoff_.size() * sizeof(long long)), oToff = take(toff_.size() * sizeof(long long));
        const size_t oTlut = take(256), oId = take(256), oEq4 = take(512), oPres = take(32);
        const size_t oQ = inlinePools ? take((size_t)qbytes + 16) : 0, oT = inlinePools ? take((size_t)tbytes + 16) : 0;
        EDLIB_AMD_HIP(d_in_.alloc(at));
        EDLIB_AMD_HIP(h_in_.alloc(at));
        uint8_t* h = h_in_.p;
        memcpy(h + oQoff, qoff_.data(), qoff_.size() * sizeof(long long));
        memcpy(h + oToff, toff_.data(), toff_.size() * sizeof(long long));
        memcpy(h + oTlut, tab_.tlut, 256); memcpy(h + oId, tab_.idToByte, 256);
        memcpy(h + oEq4, tab_.eqtbl, 512); memcpy(h + oPres, tab_.presence, 32);
        d_qoff_.alias(d_in_.p + oQoff, qoff_.size()); d_toff_.alias(d_in_.p + oToff, toff_.size());
        d_tlut_.alias(d_in_.p + oTlut, 256); d_idToByte_.alias(d_in_.p + oId, 256);
        d_eqtbl_.alias(d_in_.p + oEq4, 256); d_presence_.alias(d_in_.p + oPres, 8);
        if (inlinePools) {
            if (qbytes) memcpy(h + oQ, queries + qb, (size_t)qbytes);
            if (tbytes) memcpy(h + oT, targets + tb, (size_t)tbytes);
            memset(h + oQ + qbytes, 0, 16); memset(h + oT + tbytes, 0, 16);
            d_qpool_.alias(d_in_.p + oQ, (size_t)qbytes + 16); d_tpool_.alias(d_in_.p + oT, (size_t)tbytes + 16);
        } else {
            EDLIB_AMD_HIP(d_qpool_.alloc((size_t)qbytes + 16));
            EDLIB_AMD_HIP(d_tpool_.alloc((size_t)tbytes + 16));
            if (qbytes) EDLIB_AMD_HIP(hipMemcpy(d_qpool_.p, queries + qb, (size_t)qbytes, hipMemcpyHostToDevice));
            if (tbytes) EDLIB_AMD_HIP(hipMemcpy(d_tpool_.p, targets + tb, (size_t)tbytes, hipMemcpyHostToDevice));
        }
        EDLIB_AMD_HIP(hipMemcpyAsync(d_in_.p, h, at, hipMemcpyHostToDevice, stream_));
    }

    // classification of the units for phase 1
    const int mode = (int)cfg_.mode;
    // reads-per-lane kernels: a shared target with at most 4 distinct bytes (every mode), or up to 16 in HW mode
    // (the banded kernel keeps 8 or 16 Peq rows per word in LDS: genomes with N, soft-masked lower case, IUPAC codes)
    const int modeIn = (int)cfg.mode;
    const bool readsOk = shared_ && tlen(0) > 0 && (tab_.sigmaT <= 4 || (tab_.sigmaT <= 16 && modeIn == EDLIB_MODE_HW));
    syms_ = tab_.sigmaT <= 4 ? 4 : (tab_.sigmaT <= 8 ? 8 : 16);
    banded_ = mode == EDLIB_MODE_HW;
    // HW queries longer than kernel A's 256 rows against the shared target: piece filter + window verification
    // (long_reads.hip); EDLIB_AMD_FILTER=0 restores the groups of 12 / 16 / 24 / 32 words (up to 1024 bases) and kernel W
    static const bool filterOn = !(getenv("EDLIB_AMD_FILTER") && getenv("EDLIB_AMD_FILTER")[0] == '0');
    const bool filter = filterOn && readsOk && banded_ && modeIn == EDLIB_MODE_HW;
    const int maxReadLen = 32 * ((banded_ && modeIn == EDLIB_MODE_HW && syms_ <= 8 && !filter)
                                     ? (syms_ == 4 ? kMaxLongReadWords4 : kMaxLongReadWords) : kMaxReadWords);
    std::vector<std::vector<int>> byWords(kMaxLongReadWords4 + 1);
    for (int u = 0; u < n; ++u) {
        const int m = qlen(u), T = tlen(u);
        if (m == 0 || T == 0) emptyUnits_.push_back(u);
        else if (readsOk && m <= maxReadLen) { readUnits_.push_back(u); byWords[read_group_words(m)].push_back(u); }
        else if (filter) longUnits_.push_back(u);
        else pairUnits_.push_back(u);
    }
    stats.cells = 0;
    for (int u = 0; u < n; ++u) stats.cells += (long long)qlen(u) * tlen(u);
    {   // units whose alphabetLength the reads path does not produce
        alphaUnits_ = emptyUnits_;
        alphaUnits_.insert(alphaUnits_.end(), pairUnits_.begin(), pairUnits_.end());
        alphaUnits_.insert(alphaUnits_.end(), longUnits_.begin(), longUnits_.end());
        long long total = 0;
        for (int u : alphaUnits_) total += qlen(u) + (shared_ ? 0 : tlen(u));
        alphaOnHost_ = h_in_.p && d_qpool_.p && !d_qpool_.owned && total <= 65536;
        alphaBytes_ = total;
    }

    // reads-per-lane groups: one per query word count, slots padded to whole waves
    for (int w = 1; w <= kMaxLongReadWords4; ++w) {
        if (byWords[w].empty()) continue;
        std::unique_ptr<ReadGroup> g;
        if (makeGroup(byWords[w], w, g)) return 1;
        groups_.push_back(std::move(g));
    }
    if (initFlatPairs()) return 1;
    const int T = shared_ ? tlen(0) : 0;
    if (!groups_.empty() || !longUnits_.empty()) {
        EDLIB_AMD_HIP(d_tpk_.alloc((size_t)(T + 15) / 16 + 4));
        // the banded kernel reads whole dwords; on OUR stream: the null stream does not order with it
        EDLIB_AMD_HIP(hipMemsetAsync(d_tpk_.p, 0, d_tpk_.bytes(), stream_));
        EDLIB_AMD_HIP(d_wordSteps_.alloc(1));
        EDLIB_AMD_HIP(d_trows_.alloc(((size_t)(T + 15) / 16 + 2) * 8));
    }
    return 0;
}

// One reads-per-lane group: the units of one word count, padded to whole waves, with its resident buffers.
// oneRoundWaves > 0: the group runs on a kernel of which the chip holds that many waves without two sharing a SIMD (the
// full-height kernels of 24 / 32 words: 24 / 32 KB of LDS rows per wave) -- one launch of at most that many waves, segments as
// long as that allows (long_reads.hip, solveTallFull: the 1025th wave costs a third of the rate, every warm-up is work)
int Batch::makeGroup(const std::vector<int>& units, int w, std::unique_ptr<ReadGroup>& g, long long oneRoundWaves)
{
    const int mode = (int)cfg_.mode;
    const int T = shared_ ? tlen(0) : 0;
    g.reset(new ReadGroup);
    g->nwords = w;
    g->nslots = roundup((int)units.size(), 64);
    g->perm.assign(g->nslots, -1);
    std::copy(units.begin(), units.end(), g->perm.begin());
    const int nrblk = g->nslots / 64;
    if (mode == EDLIB_MODE_HW) {
        // enough waves to fill 256 CUs x 4 SIMDs x 8 slots many times over, segments >= 4096 columns
        // ~16 waves per resident slot: the launch ends on a thin tail (65,536 -> 131,072 waves: +1 % at 1M reads)
        const long long wantWaves = 131072;
        g->warm = 2 * 32 * w - 1;                        // 2m-1 columns
        long long S = (wantWaves + nrblk - 1) / nrblk;
        // gridDim.y limit; segments of at least 4096 columns and four warm-ups (the groups of 24 / 32 words warm up
        // over 1535 / 2047 columns: 4096-column segments were half warm-up)
        const long long maxS = std::max<long long>(1, std::min<long long>(std::min(65535, std::max(1, T / 4096)), T / (4LL * g->warm)));
        S = std::max(1LL, std::min(S, maxS));
        if (oneRoundWaves > 0 && oneRoundWaves / nrblk >= 1) S = std::min(S, oneRoundWaves / nrblk);
        g->segLen = roundup((int)((T + S - 1) / S), 16);
        g->numSegments = (T + g->segLen - 1) / g->segLen;
    } else {
        g->numSegments = 1; g->segLen = roundup(T, 16); g->warm = 0;
    }
    const size_t ns = (size_t)g->nslots, S = (size_t)g->numSegments;
    EDLIB_AMD_HIP(g->d_perm.alloc(ns));
    // (on the batch's own stream: nothing of this library runs on the null stream; perm lives as long as the group)
    EDLIB_AMD_HIP(hipMemcpyAsync(g->d_perm.p, g->perm.data(), ns * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(g->d_qlen.alloc(ns)); EDLIB_AMD_HIP(g->d_kinit.alloc(ns));
    // Small groups (a call of edlibAlign() is one slot block): the merged per-slot results live in device-visible
    // pinned host memory -- the merge / Peq / census kernels write them there, nothing is downloaded, and the
    // host reads them after the one stream synchronisation a run needs anyway.
    g->zeroCopy = ns <= 1024 && pool_enabled();
    if (g->zeroCopy) {
        EDLIB_AMD_HIP(g->hostOut.alloc((ns * 20 + ns + 1) * sizeof(int)));
        int* h = reinterpret_cast<int*>(g->hostOut.p);
        memset(h, 0, (ns * 20 + ns + 1) * sizeof(int));
        g->d_best.alias(h, ns); g->d_total.alias(h + ns, ns); g->d_alphaExtra.alias(h + 2 * ns, ns);
        g->d_flags.alias(h + 3 * ns, ns + 1); g->d_pos.alias(h + 4 * ns + 1, ns * 16);
    } else
    EDLIB_AMD_HIP(g->d_alphaExtra.alloc(ns));
    EDLIB_AMD_HIP(g->d_peq.alloc(ns * (size_t)syms_ * w));
    EDLIB_AMD_HIP(g->d_segBest.alloc(ns * S)); EDLIB_AMD_HIP(g->d_segCnt.alloc(ns * S));
    EDLIB_AMD_HIP(g->d_segPos.alloc(ns * S * 8));
    if (!g->zeroCopy) {
        EDLIB_AMD_HIP(g->d_best.alloc(ns)); EDLIB_AMD_HIP(g->d_total.alloc(ns));
        EDLIB_AMD_HIP(g->d_pos.alloc(ns * 16)); EDLIB_AMD_HIP(g->d_flags.alloc(ns + 1));
    }
    return 0;
}

// the 64 KB equality matrix goes up only when a pair kernel needs it and there are additional equalities
hipError_t Batch::uploadEq8() {
    if (tab_.eq8.empty() || d_eq8_.p) return hipSuccess;
    hipError_t e = d_eq8_.alloc(65536);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(d_eq8_.p, tab_.eq8.data(), 65536, hipMemcpyHostToDevice, stream_);
}

// ------------------------------------------------------------- scan timing

void Batch::scanTimerStart() {
    if (scanEventsUsed_ == scanEvents_.size()) {
        hipEvent_t a = nullptr, b = nullptr; int dev = 0;       // run() holds the DeviceGuard: dev == device_
        (void)pool_event(&a, &dev); (void)pool_event(&b, &dev);
        scanEvents_.push_back({a, b});
    }
    (void)hipEventRecord(scanEvents_[scanEventsUsed_].first, stream_);
}
void Batch::scanTimerStop() {
    (void)hipEventRecord(scanEvents_[scanEventsUsed_].second, stream_);
    ++scanEventsUsed_;
    ++stats.scan_launches;
}

// ------------------------------------------------- result semantics (host)

// End locations of HW / SHW from the exact best bottom-row score over the target
// columns and the complete ascending list of columns attaining it:
//  * candidates are columns scoring <= min(k, m) (HW clamps k to m, edlib.cpp:566-568;
//    SHW's best never exceeds m either);
//  * the empty target prefix (position -1, score m) takes part exactly when the
//    reference's padded last block would see it, i.e. when W = 64*ceil(m/64)-m > 0
//    (edlib.cpp:661,670,681-693; oracle-verified: m=64 all-mismatch has no -1).
void finalize_semiglobal(UnitResult& r, int kcfg, int m, int best, const int* pos, long long npos) {
    const int W = ((m + 63) / 64) * 64 - m;
    const bool kAllowsM = (kcfg < 0 || kcfg >= m);
    r.ends.clear();
    if (best < 0) {
        if (W > 0 && kAllowsM) { r.editDistance = m; r.ends.push_back(-1); r.hasEnds = true; }
        else { r.editDistance = -1; r.hasEnds = false; }
        return;
    }
    r.editDistance = best;
    r.hasEnds = true;
    if (W > 0 && best == m) r.ends.push_back(-1);
    r.ends.append(pos, (size_t)npos);
}

// a recycled record of the run before last, as a fresh one
void blank_record(UnitResult& r) {
    r.status = EDLIB_STATUS_OK; r.editDistance = -1; r.alphabetLength = 0;
    r.hasEnds = r.hasStarts = r.hasAlignment = false;
    r.ends.clear(); r.starts.clear(); r.opsView = nullptr; r.opsViewLen = 0;
}

void finalize_global(UnitResult& r, int kcfg, int mode, int T, int score) {
    if (kcfg >= 0 && score > kcfg) { r.editDistance = -1; r.hasEnds = false; return; }   // edlib.cpp:744-747, 917
    r.editDistance = score;
    if (mode == EDLIB_MODE_NW) { r.hasEnds = true; r.ends.assign(1, T - 1); }              // edlib.cpp:221-225
    else r.hasEnds = false;    // unknown mode: distance as NW, no end location
}

// ------------------------------------------------------- reads-per-lane path

// One scan launch over a group's slots (or a subset through d_slotmap), banded or not.
int Batch::scanGroup(ReadGroup& g, int mode, const int* d_slotmap, int nlanes, int kcap, const int* d_kinit,
                     int numSegments, int segLen, int warm, int* segBest, int* segCnt, int* segPos, int cap,
                     const long long* posOff, const int* posCap, bool unbanded, unsigned long long* wordSteps,
                     const uint32_t* peqDense, const int* qlenDense)
{
    ReadScanArgs a{};
    // peqDense / qlenDense (+ d_kinit): rows rebuilt for exactly the lanes of this launch, in lane order (pass 2)
    a.peq = peqDense ? peqDense : g.d_peq.p; a.tpk = d_tpk_.p; a.trows = d_trows_.p; a.targetLength = tlen(0);
    // SHW (prefix mode: row -1 is 0, 1, 2, ...): D[m][j] >= j - m, and the best score never exceeds m (the empty prefix), so
    // no column beyond 2m can tie it -- the scan stops there instead of walking the whole shared target
    if (mode == EDLIB_MODE_SHW) a.targetLength = (int)std::min<long long>(a.targetLength, 64LL * g.nwords + 1);
    a.qlen = qlenDense ? qlenDense : g.d_qlen.p; a.kinit = d_kinit; a.slotmap = d_slotmap; a.nlanes = nlanes;
    a.numSegments = numSegments; a.segLen = segLen; a.warm = warm;
    a.segBest = segBest; a.segCnt = segCnt; a.segPos = segPos; a.cap = cap;
    a.posOff = posOff; a.posCap = posCap;
    a.kcap = kcap; a.wordSteps = wordSteps ? wordSteps : d_wordSteps_.p;
    a.filter = filterScan_ ? 1 : 0;
    a.chainIn = chain_.in; a.chainOut = chain_.out; a.chainSrc = chain_.src; a.chainInLanes = chain_.inLanes;
    a.chainBlocks = chain_.blocks; a.rowBase = chain_.rowBase;
    const bool chained = chain_.in != nullptr || chain_.out != nullptr;
    static const bool dbg = getenv("EDLIB_AMD_DEBUG") != nullptr;
    if (dbg) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        fprintf(stderr, "[edlib_amd] scanGroup nwords=%d mode=%d nlanes=%d S=%d segLen=%d warm=%d cap=%d kcap=%d slotmap=%p posOff=%p\n",
                g.nwords, mode, nlanes, numSegments, segLen, warm, cap, kcap, (const void*)d_slotmap, (const void*)posOff);
    }
    scanTimerStart();
    // full-height HW scans (pass 2 over unrelated reads): scan_reads_kernel for four symbols (register-resident rows: 288 ms
    // per 1M-read step; the full-height kernel with LDS rows picked by M0 took 314 ms there, the banded kernel at full
    // height 325 ms: measured in round 2, the variants are gone), scan_reads_full_kernel above four symbols and for the
    // long word groups
    const bool longGroup = g.nwords > kMaxReadWords;                  // no plain kernel for 12 / 16 words
    // columns a lane walks: the segments' own columns (a launch may cover a prefix of the target only) and their warm-ups
    const long long colsScanned = std::min<long long>(a.targetLength, (long long)numSegments * segLen) + (long long)(numSegments - 1) * warm;
    const bool fullHeight = banded_ && mode == EDLIB_MODE_HW && unbanded && (chained || syms_ > 4 || longGroup);
    if (fullHeight) {
        EDLIB_AMD_HIP(launch_scan_reads_full(g.nwords, syms_, a, stream_));
        stats.word_steps += (long long)((nlanes + 63) / 64 * 64) * g.nwords * colsScanned;
    } else if (banded_ && mode == EDLIB_MODE_HW && (!unbanded || syms_ > 4 || longGroup)) EDLIB_AMD_HIP(launch_scan_reads_banded(g.nwords, syms_, a, stream_));
    else {
        EDLIB_AMD_HIP(launch_scan_reads(g.nwords, mode, a, stream_));
        stats.word_steps += (long long)((nlanes + 63) / 64 * 64) * g.nwords * colsScanned;
    }
    scanTimerStop();
    if (dbg) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        fprintf(stderr, "[edlib_amd] scanGroup done\n");
    }
    return 0;
}

// segmentation of a launch over `nlanes` lanes: enough waves to fill the chip, segments >= 4096 columns
void plan_segments(int nlanes, int T, int mode, int warmFull, long long wantWaves,
                          int& S, int& segLen, int& warm)
{
    S = 1; segLen = roundup(T, 16); warm = 0;
    if (mode != EDLIB_MODE_HW) return;
    const long long nrblk = ((long long)nlanes + 63) / 64;
    long long want = (wantWaves + nrblk - 1) / nrblk;
    want = std::max(1LL, std::min<long long>(want, std::min(65535, std::max(1, T / 4096))));   // gridDim.y limit
    if (warmFull > 0) want = std::max(1LL, std::min<long long>(want, std::max<long long>(1, T / (4LL * warmFull))));   // >= four warm-ups per segment
    segLen = roundup((int)((T + want - 1) / want), 16);
    S = (T + segLen - 1) / segLen;
    warm = warmFull;
}

int Batch::runReads()
{
    if (groups_.empty()) return 0;
    stats.path |= 1;
    for (auto& gp : groups_) if (runGroupScans(*gp, false)) return 1;
    for (auto& gp : groups_) if (runGroupExact(*gp)) return 1;
    return 0;
}

// The scans of one group: Peq rows, then the k-doubling levels (fullOnly: one pass at the full threshold on the
// full-height kernel -- units the piece filter handed back, whose band is the whole query).
int Batch::runGroupScans(ReadGroup& g, bool fullOnly)
{
    const int T = tlen(0);
    // unknown mode values are computed as NW (edlib.cpp:205-215)
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const bool banded = banded_ && mode == EDLIB_MODE_HW;
    const int kNoCap = 0x3fffffff;
    static const bool dbgLadder = getenv("EDLIB_AMD_DEBUG") != nullptr;
    EDLIB_AMD_HIP(launch_build_peq_reads(g.nwords, syms_, d_qpool_.p, d_qoff_.p, g.d_perm.p, g.nslots,
                                         d_eqtbl_.p, d_presence_.p, cfg_.k, g.d_peq.p, g.d_qlen.p,
                                         g.d_kinit.p, g.d_alphaExtra.p, stream_));
    // ---- pass 1: all slots; banded: threshold min(k, kFirst)
    // first threshold of the k-doubling (edlib.cpp:197-217 starts at 64): 8 up to 512 bases; the groups of 24 / 32
    // words take 12 / 16 -- at 1 % error a 1024-base read has distance ~10, and a read that fails the first level
    // pays the full 32-word height over the whole target
    const int kFirstMax = std::max(8, g.nwords / 2);
    int kFirst = kFirstMax;
    bool twoPass = !fullOnly && banded && (cfg_.k < 0 || cfg_.k > kFirst) && 32 * g.nwords > kFirst;
    std::vector<int> ladder;                    // thresholds of the banded passes between the first and the full one
    if (twoPass && g.nslots >= 16384) {
        // k-doubling only pays when most units resolve at the small threshold (pass 1 costs ~2/NWD of a
        // full scan, unresolved units then pay the full scan on top).  Probe 2048 evenly strided slots
        // first (0.2 % of the work at 1M reads) and fall back to one full-threshold pass if fewer than
        // 30 % of them resolve (e.g. noisy long-read chemistry, unrelated sequences).
        const int np = 2048;
        std::vector<int> probe(np);
        for (int i = 0; i < np; ++i) probe[i] = (int)((long long)i * g.nslots / np);
        // best score of the probe slots in `map` with thresholds capped at kc (-1: nothing <= kc)
        auto probe_scan = [&](const std::vector<int>& map, int kc, std::vector<int>& bestOut) -> int {
            const int nm = (int)map.size();
            int S2, segLen2, warm2;
            plan_segments(nm, T, mode, g.warm, 16384, S2, segLen2, warm2);
            const size_t items = (size_t)nm * S2;
            DevBuf<int> d_map, d_sb, d_sc;
            EDLIB_AMD_HIP(d_map.alloc(nm)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, map.data(), nm * sizeof(int), hipMemcpyHostToDevice, stream_));
            if (scanGroup(g, mode, d_map.p, nm, kc, g.d_kinit.p, S2, segLen2, warm2,
                          d_sb.p, d_sc.p, d_sb.p /*unused*/, 0, nullptr, nullptr)) return 1;
            std::vector<int> cnts(items), bests(items);
            EDLIB_AMD_HIP(hipMemcpyAsync(cnts.data(), d_sc.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipMemcpyAsync(bests.data(), d_sb.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            bestOut.assign(nm, -1);
            for (int i = 0; i < nm; ++i) {
                int b = 0x7fffffff;
                for (int sg = 0; sg < S2; ++sg)
                    if (cnts[(size_t)i * S2 + sg] > 0) b = std::min(b, bests[(size_t)i * S2 + sg]);
                if (b <= kc) bestOut[i] = b;
            }
            return 0;
        };
        std::vector<int> pbest;
        if (probe_scan(probe, kFirst, pbest)) return 1;
        int resolved = 0, real = 0;
        std::vector<int> hist(kFirstMax + 1, 0);                  // distances of the resolved probe reads
        std::vector<int> open;                                    // probe slots with nothing <= kFirstMax
        for (int i = 0; i < np; ++i) {
            if (g.perm[probe[i]] < 0) continue;
            ++real;
            if (pbest[i] >= 0) { ++resolved; ++hist[pbest[i]]; }
            else if (i == 0 || probe[i] != probe[i - 1]) open.push_back(probe[i]);
        }
        // The band of pass 1 is one 32-row word while the score 32 rows down stays above k + 4; against
        // unrelated sequence that score hovers around 13, so every unit of k below 8 keeps the second
        // word out more often.  Take the smallest threshold (>= 4) that still resolves 99.5 % of what 8
        // resolves: the few reads above it just join pass 2.
        if (resolved > 0) {
            int acc = 0, kq = kFirstMax;
            for (int d = 0; d <= kFirstMax; ++d) { acc += hist[d]; if (acc * 1000LL >= resolved * 995LL) { kq = d; break; } }
            kFirst = std::max(4, std::min(kFirstMax, kq));
        }
        if (real > 0 && resolved * 10 < real * 3) twoPass = false;
        // ---- the levels between the first and the full threshold (the reference doubles k: edlib.cpp:197-217).  When
        // more than a tenth of the probe is still open, the open probe reads are scanned once more with thresholds
        // capped at 64: their distances say which intermediate thresholds pay.  A level at threshold t costs every
        // read that reaches it a band of about 1 + (t - 6) / 8 words per column; it pays when what it resolves
        // would otherwise meet a taller band.  All subsets of {12, 16, 24, 32, 48, 64} are priced; reads at
        // Illumina-like error rates (leftovers = unrelated sequence) keep the two levels they always had.
        if (twoPass && (int)open.size() * 10 > real && open.size() >= 32) {
            const int kTop = std::min(64, 32 * g.nwords - 1);
            std::vector<int> obest;
            if (probe_scan(open, kTop, obest)) return 1;
            static const int cand[6] = {12, 16, 24, 32, 48, 64};
            auto words = [&](int t) { return std::min<double>(g.nwords, 1.0 + std::max(0, t - 6) / 8.0); };
            auto frac_le = [&](int t) {                           // share of the open reads with distance <= t
                size_t c = 0;
                for (int b : obest) if (b >= 0 && b <= t) ++c;
                return (double)c / (double)obest.size();
            };
            double bestCost = 1e30; int bestMask = 0;
            for (int mask = 0; mask < 64; ++mask) {
                double cost = 0.0, reach = 1.0; bool ok = true;
                for (int q = 0; q < 6; ++q) {
                    if (!((mask >> q) & 1)) continue;
                    if (cand[q] <= kFirst || cand[q] > kTop) { ok = false; break; }
                    cost += reach * words(cand[q]);
                    reach = 1.0 - frac_le(cand[q]);
                }
                if (!ok) continue;
                cost += reach * g.nwords;                         // what is left takes the full threshold
                if (cost < bestCost - 1e-9) { bestCost = cost; bestMask = mask; }
            }
            for (int q = 0; q < 6; ++q) if ((bestMask >> q) & 1) ladder.push_back(cand[q]);
            if (dbgLadder) {
                fprintf(stderr, "[edlib_amd] ladder nwords=%d kFirst=%d open=%zu/%d levels:", g.nwords, kFirst, open.size(), real);
                for (int t : ladder) fprintf(stderr, " %d(%.2f)", t, frac_le(t));
                fprintf(stderr, " full\n");
            }
        }
    }
    if (scanGroup(g, mode, nullptr, g.nslots, twoPass ? kFirst : kNoCap, g.d_kinit.p, g.numSegments, g.segLen,
                  g.warm, g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, 8, nullptr, nullptr, /*unbanded=*/fullOnly)) return 1;
    EDLIB_AMD_HIP(launch_merge_segments(g.d_segBest.p, g.d_segCnt.p, g.d_segPos.p, g.numSegments, 8,
                                        g.nslots, nullptr, 16, g.d_best.p, g.d_total.p, g.d_pos.p,
                                        g.d_flags.p, stream_));
    // ---- the next levels (k-doubling): slots with nothing <= the last threshold are rescanned with the next one,
    // the last time with their full threshold
    ladder.push_back(kNoCap);
    int kDone = kFirst;
    for (size_t lv = 0; twoPass && lv < ladder.size(); ++lv) {
        const int kcapL = ladder[lv];
        const bool last = kcapL == kNoCap;
        PinBuf totalPin;                               // pinned: the copy runs at link rate
        const int* total = g.d_total.p;
        if (!g.zeroCopy) {
            EDLIB_AMD_HIP(totalPin.alloc((size_t)g.nslots * sizeof(int)));
            EDLIB_AMD_HIP(hipMemcpyAsync(totalPin.p, g.d_total.p, (size_t)g.nslots * sizeof(int), hipMemcpyDeviceToHost, stream_));
            total = reinterpret_cast<const int*>(totalPin.p);
        }
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        std::vector<int> todo;
        for (int s = 0; s < g.nslots; ++s) {
            const int u = g.perm[s];
            if (u < 0 || total[s] > 0) continue;
            if (std::min(qlen(u), cfg_.k < 0 ? 0x3fffffff : cfg_.k) > kDone) todo.push_back(s);   // its threshold min(k, m) is above what was tried
        }
        if (todo.empty()) break;
        {
            const size_t no = todo.size();
            int S2, segLen2, warm2;
            plan_segments((int)no, T, mode, g.warm, 65536, S2, segLen2, warm2);
            const size_t items = no * (size_t)S2;
            DevBuf<int> d_map, d_sb, d_sc, d_sp;
            EDLIB_AMD_HIP(d_map.alloc(no)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
            EDLIB_AMD_HIP(d_sp.alloc(items * 8));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, todo.data(), no * sizeof(int), hipMemcpyHostToDevice, stream_));
            // What the last level leaves over is usually unrelated sequence whose band is the whole query; there the
            // plain full-height kernel (register-resident Peq rows, no band bookkeeping) is ~12 % faster per
            // column than the banded one.  256 strided leftovers tell: the banded kernel reports its band
            // height (word-steps), and a band above 85 % of the words sends the pass to the plain kernel.
            bool plain = false;
            if (last && no >= 4096) {
                const int np2 = 256;
                std::vector<int> sub(np2);
                for (int i = 0; i < np2; ++i) sub[i] = todo[(size_t)((long long)i * no / np2)];
                int S3, segLen3, warm3;
                plan_segments(np2, T, mode, g.warm, 16384, S3, segLen3, warm3);
                const size_t it3 = (size_t)np2 * S3;
                DevBuf<int> d_m3, d_b3, d_c3, d_p3; DevBuf<unsigned long long> d_ws;
                EDLIB_AMD_HIP(d_m3.alloc(np2)); EDLIB_AMD_HIP(d_b3.alloc(it3)); EDLIB_AMD_HIP(d_c3.alloc(it3));
                EDLIB_AMD_HIP(d_p3.alloc(it3 * 8)); EDLIB_AMD_HIP(d_ws.alloc(1));
                EDLIB_AMD_HIP(hipMemcpyAsync(d_m3.p, sub.data(), np2 * sizeof(int), hipMemcpyHostToDevice, stream_));
                EDLIB_AMD_HIP(hipMemsetAsync(d_ws.p, 0, sizeof(unsigned long long), stream_));
                if (scanGroup(g, mode, d_m3.p, np2, kNoCap, g.d_kinit.p, S3, segLen3, warm3,
                              d_b3.p, d_c3.p, d_p3.p, 8, nullptr, nullptr, false, d_ws.p)) return 1;
                unsigned long long ws = 0;
                EDLIB_AMD_HIP(hipMemcpyAsync(&ws, d_ws.p, sizeof ws, hipMemcpyDeviceToHost, stream_));
                EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
                const double cols = (double)np2 * ((double)T + (double)(S3 - 1) * warm3);
                plain = (double)ws >= 0.85 * g.nwords * cols;
                stats.word_steps += (long long)ws;
            }
            // The leftovers are scattered over the batch: through the slot map every lane of a wave would pull its
            // rows from a different 256-byte line (16x the bytes, once per segment: 3 GB of fetch per 1M-read step
            // in round 1).  Their rows are rebuilt in lane order instead -- the builder reads each query once.
            const size_t no64 = (no + 63) / 64 * 64;
            std::vector<int> perm2(no64, -1);
            for (size_t i = 0; i < no; ++i) perm2[i] = g.perm[todo[i]];
            DevBuf<int> d_perm2, d_qlen2, d_kinit2, d_extra2; DevBuf<uint32_t> d_peq2;
            EDLIB_AMD_HIP(d_perm2.alloc(no64)); EDLIB_AMD_HIP(d_qlen2.alloc(no64)); EDLIB_AMD_HIP(d_kinit2.alloc(no64));
            EDLIB_AMD_HIP(d_extra2.alloc(no64)); EDLIB_AMD_HIP(d_peq2.alloc(no64 * (size_t)syms_ * g.nwords));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_perm2.p, perm2.data(), no64 * sizeof(int), hipMemcpyHostToDevice, stream_));
            EDLIB_AMD_HIP(launch_build_peq_reads(g.nwords, syms_, d_qpool_.p, d_qoff_.p, d_perm2.p, (int)no64,
                                                 d_eqtbl_.p, d_presence_.p, cfg_.k, d_peq2.p, d_qlen2.p,
                                                 d_kinit2.p, d_extra2.p, stream_));
            // Every segment starts from its lane's threshold, and a lane records a position whenever its best improves: from
            // min(k, m) an unrelated read walks down ~100 improvements per segment, each a scattered 4-byte store (1.2 GB of
            // write traffic per 1M-read step in round 2).  The first columns of the target give every lane a score that
            // some column does reach; all segments start from that one (results do not depend on it: the best over
            // the whole target is at most that score, and equal scores are still recorded).
            const int seedCols = 4096;                               // (a lone wave per SIMD: 0.27 ms)
            DevBuf<int> d_b0, d_c0, d_p0;                            // (live until the synchronisation below)
            if (mode == EDLIB_MODE_HW && last && no >= 4096 && S2 > 1 && T >= 16 * seedCols) {
                EDLIB_AMD_HIP(d_b0.alloc(no)); EDLIB_AMD_HIP(d_c0.alloc(no)); EDLIB_AMD_HIP(d_p0.alloc(no * 8));
                if (scanGroup(g, mode, nullptr, (int)no, kcapL, d_kinit2.p, 1, seedCols, 0,
                              d_b0.p, d_c0.p, d_p0.p, 8, nullptr, nullptr, plain, nullptr, d_peq2.p, d_qlen2.p)) return 1;
                hipLaunchKernelGGL(seed_thresholds_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, stream_,
                                   d_kinit2.p, d_b0.p, d_c0.p, (int)no);
                EDLIB_AMD_HIP(hipGetLastError());
            }
            if (scanGroup(g, mode, nullptr, (int)no, kcapL, d_kinit2.p, S2, segLen2, warm2,
                          d_sb.p, d_sc.p, d_sp.p, 8, nullptr, nullptr, plain, nullptr, d_peq2.p, d_qlen2.p)) return 1;
            EDLIB_AMD_HIP(launch_merge_segments(d_sb.p, d_sc.p, d_sp.p, S2, 8, (int)no, d_map.p, 16,
                                                g.d_best.p, g.d_total.p, g.d_pos.p, g.d_flags.p, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));            // temporaries die here
            if (dbgLadder) fprintf(stderr, "[edlib_amd] level kcap=%d: %zu slots rescanned (plain=%d)\n", kcapL, no, (int)plain);
        }
        kDone = kcapL;
    }
    // census of slots whose end-location list did not fit (small groups: counted on the host from the pinned flags)
    if (!g.zeroCopy) {
        int* counter = g.d_flags.p + g.nslots;
        EDLIB_AMD_HIP(hipMemsetAsync(counter, 0, sizeof(int), stream_));
        hipLaunchKernelGGL(count_flags_kernel, dim3((g.nslots + 255) / 256), dim3(256), 0, stream_,
                           g.d_flags.p, g.nslots, counter);
    }
    return 0;
}

// exact second pass for the (rare) slots with more end locations than the first pass keeps.
// Their best score b is already exact, so "score <= b" selects exactly the end locations:
// (a) a counting scan over fine segments gives the number of hits of every (slot, segment),
// (b) after a prefix sum the same scan writes them to their final place.  Fine segments keep
// the pass parallel (a handful of slots still fills the chip).
int Batch::runGroupExact(ReadGroup& g)
{
    const int T = tlen(0);
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const int kNoCap = 0x3fffffff;
    const size_t ns = (size_t)g.nslots;
    g.ovfSlots.clear(); g.ovfOff.assign(1, 0);
    int novf = 0;
    if (g.zeroCopy) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        for (size_t s = 0; s < ns; ++s) novf += g.d_flags.p[s] != 0;
    }
    else {
        EDLIB_AMD_HIP(hipMemcpyAsync(&novf, g.d_flags.p + g.nslots, sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    }
    if (novf <= 0 || mode == EDLIB_MODE_NW) return 0;
    std::vector<int> flags(ns), total(ns);
    if (g.zeroCopy) memcpy(flags.data(), g.d_flags.p, ns * sizeof(int));
    else {
        EDLIB_AMD_HIP(hipMemcpyAsync(flags.data(), g.d_flags.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    }
    for (size_t s = 0; s < ns; ++s)
        if (flags[s] && g.perm[s] >= 0) g.ovfSlots.push_back((int)s);
    const size_t no = g.ovfSlots.size();
    if (!no) return 0;
    int S2, segLen2, warm2;
    plan_segments((int)no, T, mode, g.warm, 16384, S2, segLen2, warm2);
    const size_t items = no * (size_t)S2;
    DevBuf<int> d_map, d_caps, d_sb, d_sc; DevBuf<long long> d_off;
    EDLIB_AMD_HIP(d_map.alloc(no)); EDLIB_AMD_HIP(d_sb.alloc(items)); EDLIB_AMD_HIP(d_sc.alloc(items));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_map.p, g.ovfSlots.data(), no * sizeof(int), hipMemcpyHostToDevice, stream_));
    // (a) count; threshold = the exact best (d_best), so the band is as narrow as it gets
    if (scanGroup(g, mode, d_map.p, (int)no, kNoCap, g.d_best.p, S2, segLen2, warm2,
                  d_sb.p, d_sc.p, d_sb.p /*unused*/, 0, nullptr, nullptr)) return 1;
    std::vector<int> cnts(items);
    EDLIB_AMD_HIP(hipMemcpyAsync(cnts.data(), d_sc.p, items * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    std::vector<long long> offs(items);
    long long acc = 0;
    g.ovfOff.assign(no + 1, 0);
    for (size_t i = 0; i < no; ++i) {
        for (int sg = 0; sg < S2; ++sg) { offs[i * S2 + sg] = acc; acc += cnts[i * S2 + sg]; }
        g.ovfOff[i + 1] = acc;
    }
    EDLIB_AMD_HIP(d_caps.alloc(items)); EDLIB_AMD_HIP(d_off.alloc(items)); EDLIB_AMD_HIP(g.d_ovfPool.ensure((size_t)acc));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_caps.p, cnts.data(), items * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_off.p, offs.data(), items * sizeof(long long), hipMemcpyHostToDevice, stream_));
    // (b) write
    if (scanGroup(g, mode, d_map.p, (int)no, kNoCap, g.d_best.p, S2, segLen2, warm2,
                  d_sb.p, d_sc.p, g.d_ovfPool.p, 0, d_off.p, d_caps.p)) return 1;
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));                    // temporaries die here
    stats.overflow_units += (int)no;
    return 0;
}

// the shared target in the forms the reads-per-lane kernels read (once per run; the work counter of the banded kernel)
int Batch::packTarget()
{
    if (groups_.empty() && longUnits_.empty()) return 0;
    const int T = tlen(0);
    EDLIB_AMD_HIP(hipMemsetAsync(d_wordSteps_.p, 0, sizeof(unsigned long long), stream_));
    if (syms_ == 4) EDLIB_AMD_HIP(launch_pack_target_2bit(d_tpool_.p, d_tlut_.p, T, d_tpk_.p, stream_));
    if (banded_) EDLIB_AMD_HIP(launch_pack_target_rows(d_tpool_.p, d_tlut_.p, T, d_trows_.p, (int)d_trows_.n, stream_));
    return 0;
}

int Batch::collectReads(std::vector<UnitResult>& res)
{
    if (groups_.empty()) return 0;
    for (auto& gp : groups_) if (collectGroup(*gp, res)) return 1;
    readsCollected_ = true;
    return 0;
}

// D2H of one group's merged per-slot results + the result semantics of its units
int Batch::collectGroup(ReadGroup& g, std::vector<UnitResult>& res)
{
    const int T = tlen(0);
    const int mode = (cfg_.mode == EDLIB_MODE_HW || cfg_.mode == EDLIB_MODE_SHW) ? (int)cfg_.mode : (int)EDLIB_MODE_NW;
    const size_t ns = (size_t)g.nslots;
    // merged per-slot results: read in place when they already live in pinned host memory (small groups), else
    // downloaded into pinned staging (a copy into pageable memory runs at a fraction of the link rate: 64 bytes
    // per read were 20 ms per 1M reads)
    std::vector<int> ovfPos((size_t)g.ovfOff.back());
    PinBuf stage;
    const int *best, *total, *extra, *pos;
    if (g.zeroCopy) {                          // run() synchronised the stream
        best = g.d_best.p; total = g.d_total.p; extra = g.d_alphaExtra.p; pos = g.d_pos.p;
    } else {
        EDLIB_AMD_HIP(stage.alloc(ns * 19 * sizeof(int)));
        int* h = reinterpret_cast<int*>(stage.p);
        EDLIB_AMD_HIP(hipMemcpyAsync(h, g.d_best.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + ns, g.d_total.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + 2 * ns, g.d_alphaExtra.p, ns * sizeof(int), hipMemcpyDeviceToHost, stream_));
        EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * ns, g.d_pos.p, ns * 16 * sizeof(int), hipMemcpyDeviceToHost, stream_));
        best = h; total = h + ns; extra = h + 2 * ns; pos = h + 3 * ns;
    }
    if (!ovfPos.empty())
        EDLIB_AMD_HIP(hipMemcpyAsync(ovfPos.data(), g.d_ovfPool.p, ovfPos.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    size_t oi = 0;
    for (size_t s = 0; s < ns; ++s) {
        const int u = g.perm[s];
        if (u < 0) continue;
        UnitResult& r = res[u];
        r.alphabetLength = tab_.sigmaT + extra[s];
        const int m = qlen(u);
        if (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) {
            if (oi < g.ovfSlots.size() && g.ovfSlots[oi] == (int)s) {
                finalize_semiglobal(r, cfg_.k, m, best[s], ovfPos.data() + g.ovfOff[oi], g.ovfOff[oi + 1] - g.ovfOff[oi]);
                ++oi;
            } else {
                finalize_semiglobal(r, cfg_.k, m, best[s], pos + s * 16, best[s] < 0 ? 0 : total[s]);
            }
        } else {
            finalize_global(r, cfg_.k, (int)cfg_.mode, T, best[s]);
        }
    }
    return 0;
}

// ------------------------------------------------------ block-per-lane path



// Row length of the LDS-resident Peq of the ring kernels: the next power of two up to 32 blocks, a
// multiple of 32 above (bank-conflict-free lookups, scan_pairs_ring_kernel)
int peq_row_stride(long long nb) {
    if (nb > 32) return (int)std::min<long long>((nb + 31) / 32 * 32, 1 << 20);
    int s = 1; while (s < nb) s <<= 1;
    return s;
}

// the counter the ring kernels of this run add their live word-steps to (zeroed by run(), read back at its end)
unsigned long long* Batch::ringStepsCounter()
{
    if (!d_ringSteps_.p) {
        if (d_ringSteps_.alloc(1) != hipSuccess || h_ringSteps_.alloc(sizeof(unsigned long long)) != hipSuccess) return nullptr;
        if (hipMemsetAsync(d_ringSteps_.p, 0, sizeof(unsigned long long), stream_) != hipSuccess) return nullptr;
    }
    ringStepsUsed_ = true;
    return d_ringSteps_.p;
}

int Batch::solve(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units, SolveOut& out,
                 int ring, int ringH)
{
    const size_t n = units.size();
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    if (n == 0) return 0;
    if (ring == kWide && wantPath) { set_error("the wide kernel keeps no column store"); return 1; }
    stats.path |= 2;
    // chunk so that the Peq pool and (for PATH) the column store stay within a budget
    const long long peqBudget = 4LL << 30, storeBudget = 12LL << 30;
    size_t a = 0;
    while (a < n) {
        long long peqBytes = 0, storeBytes = 0;
        size_t b = a;
        while (b < n) {
            const long long nb = (units[b].qlen + 63) / 64;
            const long long pb = nb * tab_.sigmaT * 8;
            const long long sb = !wantPath ? 0 : (long long)sizeof(StoreEntry) * (ring > 0 ? ring_store_entries(ring, units[b].qlen, units[b].tlen)
                                                            : pair_store_entries(units[b].qlen, units[b].tlen));
            if (b > a && (peqBytes + pb > peqBudget || storeBytes + sb > storeBudget)) break;
            peqBytes += pb; storeBytes += sb; ++b;
        }
        if (solveChunk(mode, wantPositions, wantPath, units, a, b, out, ring, ringH)) return 1;
        a = b;
    }
    return 0;
}

int Batch::solveChunk(int mode, bool wantPositions, bool wantPath, const std::vector<UnitSpec>& units,
                      size_t ua, size_t ub, SolveOut& out, int ring, int ringH)
{
    const size_t n = ub - ua;
    Lap lap;
    PinBuf descsPin;                                   // built in pinned staging: the H2D runs at link rate
    EDLIB_AMD_HIP(descsPin.alloc(n * sizeof(PairDesc)));
    PairDesc* descs = reinterpret_cast<PairDesc*>(descsPin.p);
    std::vector<long long> opsOff(wantPath ? n + 1 : 1, 0);           // [n] = total op bytes (0 without PATH)
    long long peqWords = 0, auxInts = 0, storeEntries = 0, nbMax = 0;
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& s = units[ua + i];
        PairDesc& d = descs[i];
        const long long nb = (s.qlen + 63) / 64;
        nbMax = std::max(nbMax, nb);
        d.qoff = s.qoff; d.toff = s.toff; d.qlen = s.qlen; d.tlen = s.tlen; d.qstep = s.qstep; d.tstep = s.tstep;
        d.kinit = s.kinit; d.skip = s.skip;
        d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
        d.auxOff = auxInts; if (nb > 64 && !ring) auxInts += s.tlen;
        d.storeOff = storeEntries;
        if (wantPath) storeEntries += ring > 0 ? ring_store_entries(ring, s.qlen, s.tlen) : pair_store_entries(s.qlen, s.tlen);
        d.posCap = wantPositions ? kPosCap : 0;
        d.posOff = (long long)i * kPosCap;
        d.colOff = -1; d.bandT = (s.band && mode == EDLIB_MODE_SHW) ? -1 : 0; d.ring = ring > 0 ? ring : 0;
        // op slot of the unit, filled from the back.  An alignment has (m + T + inserts + deletes) / 2 ops, and a ring scan
        // is only walked when its distance is within kinit: (m + T + kinit) / 2 bounds the length (config 5: 1064 bytes
        // instead of 2000 per pair to bring back over PCIe)
        if (wantPath) {
            const long long full = (long long)s.qlen + s.tlen;
            opsOff[i + 1] = opsOff[i] + (ring > 0 && s.kinit >= 0 && s.kinit < full ? (full + s.kinit) / 2 + 8 : full);
        }
        // executed work: whole matrix, or one 64-block wave per column inside the band
        // executed work: the strips update every block of every column; the rings count the updates inside the band themselves
        if (!ring) stats.word_steps += 2 * nb * (long long)s.tlen;
        else if (ring == kWide) stats.word_steps += wide_word_steps(mode, s.qlen, s.tlen, d.bandT, s.kinit);
    }
    WidePlan wplan;
    if (ring == kWide && planWide(mode, descs, n, wplan)) return 1;
    // A handful of units (edlibAlign() is one): the kernels write scores, positions and op strings straight into
    // device-visible pinned host memory -- no download commands behind the launches, one stream synchronisation.
    // (Descriptors still go up with a copy: the packed rings re-read them, and every read of host memory is a PCIe
    // round trip.)  Larger chunks stage through HBM: a PCIe transaction per store does not scale.
    const long long opsTotal = wantPath ? opsOff[n] : 0;
    const bool zeroCopy = n <= 16 && opsTotal <= (64 << 10) && pool_enabled();
    PinBuf outPin;
    int* hOut3 = nullptr; int* hPos = nullptr; int* hOpsLen = nullptr; long long* hOpsOff = nullptr;
    std::shared_ptr<PinBuf> ops;
    EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords));
    EDLIB_AMD_HIP(d_aux_.ensure((size_t)auxInts));
    if (zeroCopy) {
        const size_t bytes = (3 * n + n * kPosCap + n) * sizeof(int) + (n + 1) * sizeof(long long);
        EDLIB_AMD_HIP(outPin.alloc(bytes));
        hOpsOff = reinterpret_cast<long long*>(outPin.p);
        hOut3 = reinterpret_cast<int*>(hOpsOff + n + 1); hPos = hOut3 + 3 * n; hOpsLen = hPos + n * kPosCap;
        if (wantPath) memcpy(hOpsOff, opsOff.data(), (n + 1) * sizeof(long long));
        else memset(hOpsOff, 0, (n + 1) * sizeof(long long));
        EDLIB_AMD_HIP(d_descs_.ensure(n));
        d_out3_.alias(hOut3, 3 * n); d_posPool_.alias(hPos, n * kPosCap);
        d_opsLen_.alias(hOpsLen, n); d_opsOff_.alias(hOpsOff, n + 1);
        if (wantPath && opsOff[n] > 0) {
            ops = std::make_shared<PinBuf>();
            EDLIB_AMD_HIP(ops->alloc((size_t)opsOff[n]));
            d_ops_.alias(ops->p, (size_t)opsOff[n]);
        }
    } else {
        if (!d_out3_.owned) d_out3_.release();
        if (!d_posPool_.owned) d_posPool_.release();
        if (!d_opsLen_.owned) d_opsLen_.release();
        if (!d_opsOff_.owned) d_opsOff_.release();
        if (!d_ops_.owned) d_ops_.release();
        EDLIB_AMD_HIP(d_descs_.ensure(n));
        // score / count / last of the chunk side by side: one copy brings all three back
        EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
        EDLIB_AMD_HIP(d_posPool_.ensure(n * kPosCap));
    }
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    if (wantPath) {
        EDLIB_AMD_HIP(d_store_.ensure((size_t)storeEntries));
        if (!zeroCopy) {
            EDLIB_AMD_HIP(d_ops_.ensure((size_t)opsOff[n])); EDLIB_AMD_HIP(d_opsOff_.ensure(n + 1));
            EDLIB_AMD_HIP(d_opsLen_.ensure(n));
            EDLIB_AMD_HIP(hipMemcpyAsync(d_opsOff_.p, opsOff.data(), (n + 1) * sizeof(long long), hipMemcpyHostToDevice, stream_));
        }
    }
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs, n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT,
                                         d_peq64_.p, stream_));
    lap("chunk: descs+alloc");
    PairScanArgs a{};
    a.descs = d_descs_.p; a.numUnits = (int)n; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p;
    a.peqRowStride = peq_row_stride(nbMax);
    a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    if (getenv("EDLIB_AMD_PEQFULL") && getenv("EDLIB_AMD_PEQFULL")[0] == '0') a.peqFullStride = 0;
    a.store = d_store_.p;
    a.outScore = d_outScore_.p; a.outCount = d_outCount_.p; a.outLast = d_outLast_.p; a.posPool = d_posPool_.p;
    a.colP = nullptr; a.colM = nullptr; a.colS = nullptr;
    a.wordSteps = ring > 0 ? ringStepsCounter() : nullptr;
    scanTimerStart();
    if (ring == kWide) { if (launchWide(mode, a, descs, n, wplan)) return 1; }
    else if (ring) EDLIB_AMD_HIP(launch_scan_pairs_ring(ring, mode, wantPath, a, stream_, ringH));
    else EDLIB_AMD_HIP(launch_scan_pairs(mode, wantPath, a, stream_));
    scanTimerStop();
    if (wantPath) {
        TracebackArgs tb{};
        tb.descs = d_descs_.p; tb.numUnits = (int)n; tb.score = d_outScore_.p;
        tb.store = d_store_.p;
        tb.ops = d_ops_.p; tb.opsOff = d_opsOff_.p; tb.opsLen = d_opsLen_.p;
        EDLIB_AMD_HIP(launch_traceback(tb, stream_));
    }
    if (lap.on) { EDLIB_AMD_HIP(hipStreamSynchronize(stream_)); lap("chunk: kernels"); }
    // downloads land in pinned staging (a pageable std::vector took 5 ms for the 17 MB of end positions of 262,144
    // short HW pairs); a zero-copy chunk is read where the kernels wrote it
    PinBuf stage;
    const int* score = nullptr; const int* pool = nullptr; const int* opsLen = nullptr;
    if (zeroCopy) {
        EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
        score = hOut3; pool = hPos; opsLen = hOpsLen;
        if (wantPath && ops) out.opsBufs.push_back(ops);
    } else {
        const size_t nPos = wantPositions ? n * kPosCap : 0, nLen = wantPath ? n : 0;
        EDLIB_AMD_HIP(stage.alloc((3 * n + nPos + nLen) * sizeof(int)));
        int* h = reinterpret_cast<int*>(stage.p);
        score = h; pool = h + 3 * n; opsLen = h + 3 * n + nPos;
        EDLIB_AMD_HIP(hipMemcpyAsync(h, d_out3_.p, 3 * n * sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (nPos) EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * n, d_posPool_.p, nPos * sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (nLen) {
            EDLIB_AMD_HIP(hipMemcpyAsync(h + 3 * n + nPos, d_opsLen_.p, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
            if (opsOff[n] > 0) {
                // the op slots (qlen + tlen bytes per unit, filled from the back) land in pinned staging and
                // are read from there by results(): no intermediate host copies
                ops = std::make_shared<PinBuf>();
                EDLIB_AMD_HIP(ops->alloc((size_t)opsOff[n]));
                EDLIB_AMD_HIP(hipMemcpyAsync(ops->p, d_ops_.p, (size_t)opsOff[n], hipMemcpyDeviceToHost, stream_));
                out.opsBufs.push_back(ops);
            }
        }
    }
    const int* count = score + n; const int* last = count + n;
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (ring == kWide) {
        const int w = checkWide();
        if (w == 2) return solveChunk(mode, wantPositions, wantPath, units, ua, ub, out, ring, ringH);     // (nothing of `out` was touched yet)
        if (w) return 1;
    }
    lap("chunk: kernels+D2H");

    // exact second pass for units with more end locations than kPosCap
    std::vector<int> ovf; std::vector<long long> ovfOff(1, 0); std::vector<int> ovfPos;
    if (wantPositions && mode != EDLIB_MODE_NW) {
        for (size_t i = 0; i < n; ++i)
            if (count[i] > kPosCap) { ovf.push_back((int)i); ovfOff.push_back(ovfOff.back() + count[i]); }
        if (!ovf.empty()) {
            std::vector<PairDesc> d2(ovf.size());
            for (size_t j = 0; j < ovf.size(); ++j) {
                d2[j] = descs[ovf[j]];
                d2[j].kinit = score[ovf[j]]; d2[j].posCap = count[ovf[j]]; d2[j].posOff = ovfOff[j];
            }
            DevBuf<PairDesc> dd; DevBuf<int> pool2, s2, c2, l2;
            EDLIB_AMD_HIP(dd.alloc(d2.size())); EDLIB_AMD_HIP(pool2.alloc((size_t)ovfOff.back()));
            EDLIB_AMD_HIP(s2.alloc(d2.size())); EDLIB_AMD_HIP(c2.alloc(d2.size())); EDLIB_AMD_HIP(l2.alloc(d2.size()));
            WidePlan wp2;
            if (ring == kWide && planWide(mode, d2.data(), d2.size(), wp2)) return 1;
            EDLIB_AMD_HIP(hipMemcpyAsync(dd.p, d2.data(), d2.size() * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
            PairScanArgs a2 = a;
            a2.descs = dd.p; a2.numUnits = (int)d2.size(); a2.posPool = pool2.p;
            a2.outScore = s2.p; a2.outCount = c2.p; a2.outLast = l2.p;
            scanTimerStart();
            if (ring == kWide) { if (launchWide(mode, a2, d2.data(), d2.size(), wp2)) return 1; }
            else EDLIB_AMD_HIP(launch_scan_pairs(mode, false, a2, stream_));
            scanTimerStop();
            ovfPos.resize((size_t)ovfOff.back());
            EDLIB_AMD_HIP(hipMemcpyAsync(ovfPos.data(), pool2.p, ovfPos.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            if (ring == kWide) {
                const int w = checkWide();
                if (w == 2) return solveChunk(mode, wantPositions, wantPath, units, ua, ub, out, ring, ringH);
                if (w) return 1;
            }
            stats.overflow_units += (int)ovf.size();
            for (size_t j = 0; j < ovf.size(); ++j) {
                const PairDesc& d = d2[j];
                stats.word_steps += 2LL * ((d.qlen + 63) / 64) * d.tlen;
            }
        }
    }
    size_t oj = 0;
    const bool lists = wantPositions && mode != EDLIB_MODE_NW;
    size_t w = out.posFlat.size();                                     // positions are written in place: one resize per chunk
    if (lists) {
        size_t tot = 0;
        for (size_t i = 0; i < n; ++i) if (score[i] >= 0) tot += (size_t)std::max(count[i], 0);
        out.posFlat.resize(w + tot);
    }
    int* pf = out.posFlat.data();
    for (size_t i = 0; i < n; ++i) {
        const size_t g = ua + i;
        out.score[g] = score[i]; out.count[g] = count[i]; out.last[g] = last[i];
        if (lists && score[i] >= 0) {
            const int* src = pool + i * kPosCap;
            if (oj < ovf.size() && ovf[oj] == (int)i) { src = ovfPos.data() + ovfOff[oj]; ++oj; }
            const int c = std::max(count[i], 0);
            for (int k = 0; k < c; ++k) pf[w + k] = src[k];
            w += (size_t)c;
        }
        out.posStart[g + 1] = (long long)w;
        if (wantPath && ops) {
            out.opsPtr[g] = ops->p + opsOff[i + 1] - opsLen[i];
            out.opsLen[g] = opsLen[i];
        }
    }
    lap("chunk: host gather");
    if (zeroCopy) {          // the views into this chunk's pinned block die with it
        d_out3_.release(); d_posPool_.release(); d_opsLen_.release(); d_opsOff_.release(); d_ops_.release();
        d_outScore_.release(); d_outCount_.release(); d_outLast_.release();
    }
    return 0;
}

// ------------------------------------------------------ one unit on many waves

// The strips of a unit run as a pipeline over `slots` single-wave workgroups whose hand-offs spin, so every workgroup of a
// launch has to be resident: slots * (units per launch) stays within what the device holds (wide_resident_waves).
int Batch::planWide(int mode, PairDesc* descs, size_t n, WidePlan& plan)
{
    if (wideCap_ < 0) wideCap_ = wide_resident_waves(tab_.sigmaT);
    if (wideCap_ <= 0) { set_error("wide kernel: no resident waves (occupancy query failed)"); return 1; }
    int want = 1;
    for (size_t i = 0; i < n; ++i) want = std::max(want, wide_slots_wanted(mode, descs[i].qlen, descs[i].tlen, descs[i].bandT, descs[i].kinit));
    if (const char* e = getenv("EDLIB_AMD_WIDE_SLOTS")) { if (atoi(e) > 0) want = atoi(e); }      // (tests: fewer slots than strips alive)
    // after an aborted launch of this run (workgroups not resident together, a stalled hand-off): one slot per unit -- a wave
    // then only reads granules it wrote itself and never waits, whatever else is on the device
    if (wideSerial_) want = 1;
    plan.slots = std::min(want, wideCap_);
    plan.perLaunch = (size_t)std::max(1, wideCap_ / plan.slots);
    long long words = 0, most = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i % plan.perLaunch == 0) words = 0;
        descs[i].auxOff = words;
        words += wide_stream_words(descs[i].tlen, plan.slots);
        most = std::max(most, words);
    }
    EDLIB_AMD_HIP(d_wide_.ensure((size_t)most));
    if (!d_wabort_.p) { EDLIB_AMD_HIP(d_wabort_.alloc(2)); EDLIB_AMD_HIP(h_wabort_.alloc(sizeof(unsigned))); }    // {abort word, workgroups arrived}
    EDLIB_AMD_HIP(hipMemsetAsync(d_wabort_.p, 0, 2 * sizeof(unsigned), stream_));
    return 0;
}

// The strip pipelines spin on each other, so the workgroups of a wide launch must all be resident -- which the sizing of
// ONE launch guarantees (planWide) and two launches from two host threads sharing the device would not: each could hold
// the slots the other is waiting for until the hand-off timeout.  So wide launches of a process take turns per device:
// the gate is taken before the first launch of a chunk and given back by checkWide() behind the stream synchronisation
// that follows it (or when the batch is reset / destroyed after a failure in between).  A gate, not a std::mutex: it may
// be released by another thread than the one that took it.
namespace {
struct WideGate { std::mutex m; std::condition_variable cv; bool busy = false; };
WideGate& wide_gate(int device) { static WideGate* g = new WideGate[Pool::kMaxDev]; return g[(device >= 0 && device < Pool::kMaxDev) ? device : 0]; }
}
void Batch::wideGateRelease()
{
    if (!wideGateHeld_) return;
    WideGate& g = wide_gate(device_);
    { std::lock_guard<std::mutex> l(g.m); g.busy = false; }
    g.cv.notify_one();
    wideGateHeld_ = false;
}

int Batch::launchWide(int mode, const PairScanArgs& a0, const PairDesc* hostDescs, size_t n, const WidePlan& plan)
{
    if (!wideGateHeld_) {
        WideGate& g = wide_gate(device_);
        std::unique_lock<std::mutex> l(g.m);
        g.cv.wait(l, [&] { return !g.busy; });
        g.busy = true;
        wideGateHeld_ = true;
    }
    for (size_t g0 = 0; g0 < n; g0 += plan.perLaunch) {
        const size_t g1 = std::min(n, g0 + plan.perLaunch);
        const long long words = hostDescs[g1 - 1].auxOff + wide_stream_words(hostDescs[g1 - 1].tlen, plan.slots);
        // every polled word starts at zero (tags are strip + 1): a granule of an earlier launch must never look fresh
        EDLIB_AMD_HIP(hipMemsetAsync(d_wide_.p, 0, (size_t)words * sizeof(unsigned long long), stream_));
        EDLIB_AMD_HIP(hipMemsetAsync(d_wabort_.p + 1, 0, sizeof(unsigned), stream_));     // the residency count of THIS launch
        PairScanArgs a = a0;
        a.descs = a0.descs + g0; a.numUnits = (int)(g1 - g0);
        a.outScore = a0.outScore + g0; a.outCount = a0.outCount + g0; a.outLast = a0.outLast + g0;
        a.wstream = d_wide_.p; a.wabort = d_wabort_.p;
        // (tests: the residency check of a pipelined launch waits for one workgroup more than there are -- it gives up after
        // 0.2 s as if part of the launch had not fitted the device, and the units run again with one slot each)
        a.wideExpect = (!wideSerial_ && getenv("EDLIB_AMD_WIDE_TEST_NOT_RESIDENT")) ? (unsigned)(plan.slots * (g1 - g0) + 1) : 0u;
        EDLIB_AMD_HIP(launch_scan_pairs_wide(mode, a, plan.slots, stream_));
    }
    EDLIB_AMD_HIP(hipMemcpyAsync(h_wabort_.p, d_wabort_.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    return 0;
}

// 0 = the launches since planWide() ran to their end; 2 = one of them gave up (its workgroups were not on the device
// together, or a hand-off made no progress) and the caller runs its units again, which planWide() now gives one slot each;
// 1 = that second attempt failed as well (cannot happen by construction: reported, not retried)
int Batch::checkWide()
{
    wideGateRelease();
    const unsigned code = h_wabort_.p ? *reinterpret_cast<const unsigned*>(h_wabort_.p) : 0u;
    if (code == 0u) return 0;
    ++stats.wide_retries;
    if (!wideSerial_) { wideSerial_ = true; return 2; }
    set_error(code == 2u ? "wide kernel: the workgroups of a one-slot launch did not all start"
                         : "wide kernel: a hand-off stalled inside a one-slot launch");
    return 1;
}

// alphabetLength of the units the reads path does not cover (reference transformSequences, edlib.cpp:1417-1462:
// the number of distinct bytes of query and target).  It depends on the sequences only, not on any scan, so it runs
// on a side stream next to phase 1 and is collected after it.
int Batch::alphabetLengthsBegin()
{
    alphaPending_ = false;
    if (alphaUnits_.empty() || alphaOnHost_) return 0;
    const size_t n = alphaUnits_.size();
    if (!side_) EDLIB_AMD_HIP(pool_stream(&side_));
    EDLIB_AMD_HIP(evA_.create());
    if (!d_alphaIdx_.p) {
        EDLIB_AMD_HIP(d_alphaIdx_.alloc(n)); EDLIB_AMD_HIP(d_alphaOut_.alloc(n)); EDLIB_AMD_HIP(alphaPin_.alloc(n * sizeof(int)));
        EDLIB_AMD_HIP(hipMemcpyAsync(d_alphaIdx_.p, alphaUnits_.data(), n * sizeof(int), hipMemcpyHostToDevice, side_));
    }
    // the inputs went up on stream_ (init): the side stream starts behind whatever stream_ holds now
    EDLIB_AMD_HIP(hipEventRecord(evA_.e, stream_));
    EDLIB_AMD_HIP(hipStreamWaitEvent(side_, evA_.e, 0));
    if (alphaBytes_ <= 4096LL * (long long)n)                        // short sequences: a wave per unit
        hipLaunchKernelGGL(alphabet_count_short_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, side_,
                           d_qpool_.p, d_qoff_.p, d_tpool_.p, d_toff_.p, shared_ ? 1 : 0, d_presence_.p,
                           d_alphaIdx_.p, (int)n, d_alphaOut_.p);
    else
    hipLaunchKernelGGL(alphabet_count_kernel, dim3((unsigned)n), dim3(256), 0, side_,
                       d_qpool_.p, d_qoff_.p, d_tpool_.p, d_toff_.p, shared_ ? 1 : 0, d_presence_.p,
                       d_alphaIdx_.p, d_alphaOut_.p);
    EDLIB_AMD_HIP(hipGetLastError());
    EDLIB_AMD_HIP(evB_.create());
    EDLIB_AMD_HIP(hipEventRecord(evB_.e, side_));                   // (a flat batch's collection reads d_alphaOut_ on stream_ behind this)
    EDLIB_AMD_HIP(hipMemcpyAsync(alphaPin_.p, d_alphaOut_.p, n * sizeof(int), hipMemcpyDeviceToHost, side_));
    alphaPending_ = true;
    return 0;
}

int Batch::alphabetLengthsEnd(std::vector<UnitResult>& res)
{
    if (alphaUnits_.empty()) return 0;
    if (alphaOnHost_) {
        // a handful of short sequences (edlibAlign() on a pair): counting distinct bytes on the host costs less than
        // a launch and a round trip; the sequences are still in the staging block of init()
        const uint8_t* hq = h_in_.p + (d_qpool_.p - d_in_.p);
        const uint8_t* ht = h_in_.p + (d_tpool_.p - d_in_.p);
        for (int u : alphaUnits_) {
            bool seen[256] = {false};
            int cnt = 0;
            auto add = [&](const uint8_t* p, long long len) { for (long long i = 0; i < len; ++i) if (!seen[p[i]]) { seen[p[i]] = true; ++cnt; } };
            add(hq + qoff_[u], qlen(u));
            if (!shared_) add(ht + toff_[u], tlen(u));
            else for (int b = 0; b < 256; ++b) if ((tab_.presence[b >> 5] >> (b & 31)) & 1u) { if (!seen[b]) { seen[b] = true; ++cnt; } }
            res[u].alphabetLength = cnt;
        }
        return 0;
    }
    if (!alphaPending_) return 0;
    EDLIB_AMD_HIP(hipStreamSynchronize(side_));
    alphaPending_ = false;
    const int* out = reinterpret_cast<const int*>(alphaPin_.p);
    for (size_t i = 0; i < alphaUnits_.size(); ++i) res[alphaUnits_[i]].alphabetLength = out[i];
    return 0;
}


// ------------------------------------------------------------- Hirschberg

bool needs_hirschberg(int m, int T) {
    const long long nb = (m + 63) / 64;
    return (2LL * 8 + 4) * nb * T + 8LL * T >= 1024 * 1024;              // edlib.cpp:1188-1190
}

// One level of the divide step for a set of pieces: forward scan of (query, left half) and reverse
// scan of (reversed query, reversed right half), both NW and dumped at their last column
// (edlib.cpp:1246-1260), then the split search on the device (edlib.cpp:1314-1353).
int Batch::hirschbergLevel(const std::vector<PathPiece>& big, std::vector<int>& splitRow,
                           std::vector<int>& leftScore, std::vector<int>& rightScore)
{
    const size_t np = big.size();
    // Each piece scans inside the band of the WHOLE piece with k = its distance, stopped at the half's last
    // column -- exactly the reference's two calls, edlib.cpp:1252-1260 -- on the smallest lane ring that holds
    // that band (or all its blocks); pieces no ring holds take the unbanded strips.  Both dump their last column.
    static const int rings[kNumRings + 1] = {4, 8, 16, 21, 32, 64, 0};
    // (packing only pays with enough pieces to fill the chip: a handful of long pieces runs faster one per wave)
    const bool packed = np >= 256;
    auto ring_of = [&](const PathPiece& pc) {
        const bool off = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
        if (off) return kNumRings;
        // (a few long pieces are bound by dependent steps: 0.074 us on the wide kernel's waves against 0.12 on a ring's)
        if (!packed) return (pc.m > 64 * 64 && pc.score <= kMaxBandK && pc.T < 4096) ? kNumRings - 1 : kNumRings;
        for (int g = 0; g < kNumRings; ++g) if (pc.score <= ring_max_k(rings[g])) return g;
        return (pc.m + 63) / 64 <= 64 ? kNumRings - 1 : kNumRings;
    };
    std::vector<size_t> order; order.reserve(np);
    size_t groupCount[kNumRings + 1] = {0};
    std::vector<int> groupOf(np);
    for (size_t p = 0; p < np; ++p) { groupOf[p] = ring_of(big[p]); ++groupCount[groupOf[p]]; }
    for (int g = 0; g <= kNumRings; ++g) for (size_t p = 0; p < np; ++p) if (groupOf[p] == g) order.push_back(p);
    std::vector<PairDesc> descs(2 * np);
    std::vector<int> best(np);
    // what no ring holds: the band on many waves (wide_kernels.hip)
    const bool wideOff = false;
    long long peqWords = 0, auxInts = 0, colBlocks = 0;
    for (size_t q = 0; q < np; ++q) {
        const PathPiece& pc = big[order[q]];
        const int ring = rings[groupOf[order[q]]];
        const bool wide = ring == 0 && !wideOff;
        const bool banded = ring != 0 || wide;
        const int lw = pc.T / 2, rw = pc.T - lw;                         // :1247-1248
        const long long nb = (pc.m + 63) / 64;
        best[q] = pc.score;
        for (int side = 0; side < 2; ++side) {
            PairDesc& d = descs[2 * q + side];
            d.qlen = pc.m; d.kinit = banded ? pc.score : 0; d.posCap = 0; d.posOff = 0; d.storeOff = 0;
            d.bandT = banded ? pc.T : 0; d.ring = 0; d.skip = 0;
            if (side == 0) { d.qoff = pc.qoff; d.qstep = 1; d.toff = pc.toff; d.tstep = 1; d.tlen = lw; }
            else { d.qoff = pc.qoff + pc.m - 1; d.qstep = -1; d.toff = pc.toff + pc.T - 1; d.tstep = -1; d.tlen = rw; }
            d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
            d.auxOff = auxInts; if (nb > 64 && !banded) auxInts += d.tlen;
            d.colOff = colBlocks; colBlocks += nb;
            if (!banded) stats.word_steps += 2 * nb * (long long)d.tlen;
            else if (wide) stats.word_steps += wide_word_steps(0, d.qlen, d.tlen, d.bandT, d.kinit);
        }
    }
    const size_t firstWide = np - groupCount[kNumRings];               // (the groups are laid out in ring order, this one last)
    WidePlan wplan;
    const bool anyWide = groupCount[kNumRings] > 0 && !wideOff;
    if (anyWide && planWide(0, descs.data() + 2 * firstWide, 2 * groupCount[kNumRings], wplan)) return 1;
    const size_t n = descs.size();
    DevBuf<unsigned long long> colP, colM; DevBuf<int> colS, d_best, d_out;
    EDLIB_AMD_HIP(colP.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colM.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colS.alloc((size_t)colBlocks));
    EDLIB_AMD_HIP(d_best.alloc(np)); EDLIB_AMD_HIP(d_out.alloc(3 * np));
    // blocks outside the band at the stop column: P = M = 0 and a score no sum can reach
    EDLIB_AMD_HIP(hipMemsetAsync(colP.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colM.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colS.p, 0x3f, (size_t)colBlocks * 4, stream_));
    EDLIB_AMD_HIP(d_descs_.ensure(n)); EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords)); EDLIB_AMD_HIP(d_aux_.ensure((size_t)auxInts));
    EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    EDLIB_AMD_HIP(d_posPool_.ensure(1));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_best.p, best.data(), np * sizeof(int), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT,
                                         d_peq64_.p, stream_));
    PairScanArgs a{};
    a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p;
    a.posPool = d_posPool_.p;
    {
        long long nbMax = 0;
        for (const PathPiece& pc : big) nbMax = std::max<long long>(nbMax, (pc.m + 63) / 64);
        a.peqRowStride = peq_row_stride(nbMax);
        a.peqFullStride = (int)std::min<long long>((long long)a.peqRowStride * tab_.sigmaT, 1 << 20);
    }
    a.colP = colP.p; a.colM = colM.p; a.colS = colS.p;
    a.wordSteps = ringStepsCounter();
    size_t first = 0;
    for (int g = 0; g <= kNumRings; ++g) {
        if (!groupCount[g]) continue;
        a.descs = d_descs_.p + 2 * first; a.numUnits = (int)(2 * groupCount[g]);
        a.outScore = d_outScore_.p + 2 * first; a.outCount = d_outCount_.p + 2 * first; a.outLast = d_outLast_.p + 2 * first;
        scanTimerStart();
        if (rings[g]) EDLIB_AMD_HIP(launch_scan_pairs_ring(rings[g], 0, false, a, stream_));
        else if (anyWide) { if (launchWide(0, a, descs.data() + 2 * first, 2 * groupCount[g], wplan)) return 1; }
        else EDLIB_AMD_HIP(launch_scan_pairs(EDLIB_MODE_NW, false, a, stream_));
        scanTimerStop();
        first += groupCount[g];
    }
    SplitArgs sa{};
    sa.descs = d_descs_.p; sa.numPieces = (int)np; sa.best = d_best.p;
    sa.colP = colP.p; sa.colM = colM.p; sa.colS = colS.p; sa.out = d_out.p;
    EDLIB_AMD_HIP(launch_hirschberg_split(sa, stream_));
    std::vector<int> out(3 * np);
    EDLIB_AMD_HIP(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (anyWide) {
        const int w = checkWide();
        if (w == 2) return hirschbergLevel(big, splitRow, leftScore, rightScore);
        if (w) return 1;
    }
    splitRow.resize(np); leftScore.resize(np); rightScore.resize(np);
    for (size_t q = 0; q < np; ++q) {
        const size_t p = order[q];
        splitRow[p] = out[3 * q]; leftScore[p] = out[3 * q + 1]; rightScore[p] = out[3 * q + 2];
    }
    return 0;
}

// A scan of T columns is T dependent steps however many waves share its band; the two halves of the target are independent
// of each other.  So the distance of a long unit is found like the first Hirschberg level finds its split
// (edlib.cpp:1246-1260, 1314-1353): forward scan of (query, left half) and reverse scan of (reversed query, reversed right
// half), both inside the band of the whole problem and dumped at their last column, then min over the query rows of
// L[i] + R[i+1].  Half the dependent steps; exact iff the minimum is within the threshold (cells outside the band are
// upper bounds).
int Batch::solveWideSplit(const std::vector<UnitSpec>& units, std::vector<int>& out)
{
    const size_t np = units.size();
    out.assign(4 * np, 0);
    if (np == 0) return 0;
    std::vector<PairDesc> descs(2 * np);
    long long peqWords = 0, colBlocks = 0;
    int maxRows = 1;
    for (size_t q = 0; q < np; ++q) {
        const UnitSpec& u = units[q];
        const int lw = u.tlen / 2, rw = u.tlen - lw;
        const long long nb = (u.qlen + 63) / 64;
        maxRows = std::max(maxRows, u.qlen);
        for (int side = 0; side < 2; ++side) {
            PairDesc& d = descs[2 * q + side];
            d.qlen = u.qlen; d.kinit = u.kinit; d.posCap = 0; d.posOff = 0; d.storeOff = 0; d.bandT = u.tlen; d.ring = 0; d.skip = 0;
            if (side == 0) { d.qoff = u.qoff; d.qstep = u.qstep; d.toff = u.toff; d.tstep = u.tstep; d.tlen = lw; }
            else {
                d.qoff = u.qoff + (long long)(u.qlen - 1) * u.qstep; d.qstep = -u.qstep;
                d.toff = u.toff + (long long)(u.tlen - 1) * u.tstep; d.tstep = -u.tstep; d.tlen = rw;
            }
            d.peqOff = peqWords; peqWords += nb * tab_.sigmaT;
            d.auxOff = 0;
            d.colOff = colBlocks; colBlocks += nb;
            stats.word_steps += wide_word_steps(0, d.qlen, d.tlen, d.bandT, d.kinit);
        }
    }
    WidePlan wplan;
    if (planWide(0, descs.data(), descs.size(), wplan)) return 1;
    const size_t n = descs.size();
    DevBuf<unsigned long long> colP, colM, packed; DevBuf<int> colS, d_out;
    EDLIB_AMD_HIP(colP.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colM.alloc((size_t)colBlocks)); EDLIB_AMD_HIP(colS.alloc((size_t)colBlocks));
    EDLIB_AMD_HIP(packed.alloc(np)); EDLIB_AMD_HIP(d_out.alloc(4 * np));
    // blocks outside the band at the stop column: P = M = 0 and a score no sum can reach
    EDLIB_AMD_HIP(hipMemsetAsync(colP.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colM.p, 0, (size_t)colBlocks * 8, stream_));
    EDLIB_AMD_HIP(hipMemsetAsync(colS.p, 0x3f, (size_t)colBlocks * 4, stream_));
    EDLIB_AMD_HIP(d_descs_.ensure(n)); EDLIB_AMD_HIP(d_peq64_.ensure((size_t)peqWords));
    EDLIB_AMD_HIP(d_out3_.ensure(3 * n));
    d_outScore_.alias(d_out3_.p, n); d_outCount_.alias(d_out3_.p + n, n); d_outLast_.alias(d_out3_.p + 2 * n, n);
    EDLIB_AMD_HIP(d_posPool_.ensure(1));
    EDLIB_AMD_HIP(hipMemcpyAsync(d_descs_.p, descs.data(), n * sizeof(PairDesc), hipMemcpyHostToDevice, stream_));
    EDLIB_AMD_HIP(uploadEq8());
    EDLIB_AMD_HIP(launch_build_peq_pairs(d_descs_.p, (int)n, d_qpool_.p, d_eq8_.p, d_idToByte_.p, tab_.sigmaT, d_peq64_.p, stream_));
    PairScanArgs a{};
    a.descs = d_descs_.p; a.numUnits = (int)n; a.qpool = d_qpool_.p; a.tpool = d_tpool_.p;
    a.tlut = d_tlut_.p; a.sigmaT = tab_.sigmaT; a.peq = d_peq64_.p; a.aux = d_aux_.p; a.posPool = d_posPool_.p;
    a.outScore = d_outScore_.p; a.outCount = d_outCount_.p; a.outLast = d_outLast_.p;
    a.colP = colP.p; a.colM = colM.p; a.colS = colS.p;
    scanTimerStart();
    if (launchWide(0, a, descs.data(), n, wplan)) return 1;
    scanTimerStop();
    SplitArgs sa{};
    sa.descs = d_descs_.p; sa.numPieces = (int)np; sa.best = nullptr;
    sa.colP = colP.p; sa.colM = colM.p; sa.colS = colS.p; sa.out = d_out.p;
    EDLIB_AMD_HIP(launch_split_min(sa, packed.p, maxRows, stream_));
    EDLIB_AMD_HIP(hipMemcpyAsync(out.data(), d_out.p, out.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    {
        const int w = checkWide();
        if (w == 2) return solveWideSplit(units, out);
        if (w) return 1;
    }
    return 0;
}

// Alignment paths of NW jobs of any size (reference obtainAlignment, edlib.cpp:1161-1213): pieces at
// or above the 1 MiB column-store estimate are halved Hirschberg-style, level by level across the
// whole batch, until every piece fits the traceback branch; the pieces' op strings concatenate.
int Batch::solvePaths(const std::vector<PathPiece>& jobs, std::vector<OpsOut>& opsOut, std::vector<int>& status)
{
    const size_t nj = jobs.size();
    Lap lap;
    opsOut.clear(); opsOut.resize(nj); status.assign(nj, EDLIB_STATUS_OK);
    // only jobs at or above the 1 MiB rule are ever split: the others stay a single implicit piece
    std::vector<std::vector<PathPiece>> pieces(nj);
    std::vector<size_t> bigJobs;
    for (size_t j = 0; j < nj; ++j)
        if (needs_hirschberg(jobs[j].m, jobs[j].T)) { pieces[j].push_back(jobs[j]); bigJobs.push_back(j); }
    auto npieces = [&](size_t j) { return pieces[j].empty() ? (size_t)1 : pieces[j].size(); };
    auto piece = [&](size_t j, size_t i) -> const PathPiece& { return pieces[j].empty() ? jobs[j] : pieces[j][i]; };
    for (int level = 0; level < 64 && !bigJobs.empty(); ++level) {
        std::vector<PathPiece> big; std::vector<std::pair<size_t, size_t>> where;
        for (size_t j : bigJobs) {
            if (status[j] != EDLIB_STATUS_OK) continue;
            for (size_t i = 0; i < pieces[j].size(); ++i) {
                const PathPiece& pc = pieces[j][i];
                if (pc.m > 0 && pc.T > 0 && needs_hirschberg(pc.m, pc.T)) {
                    if (pc.T < 2) { status[j] = EDLIB_STATUS_ERROR; break; }   // the reference has no answer here either
                    big.push_back(pc); where.push_back({j, i});
                }
            }
        }
        if (big.empty()) break;
        std::vector<int> row, ls, rs;
        // pieces whose two half scans already ran for the distance (solveWideSplit) bring their split along
        std::vector<PathPiece> todo; std::vector<size_t> todoAt;
        row.assign(big.size(), -2); ls.assign(big.size(), 0); rs.assign(big.size(), 0);
        for (size_t b = 0; b < big.size(); ++b) {
            const KnownSplit* ks = nullptr;
            if (level == 0)
                for (const KnownSplit& k : knownSplits_)
                    if (k.qoff == big[b].qoff && k.m == big[b].m && k.toff == big[b].toff && k.T == big[b].T && k.score == big[b].score) { ks = &k; break; }
            if (ks) { row[b] = ks->row; ls[b] = ks->left; rs[b] = ks->right; }
            else { todo.push_back(big[b]); todoAt.push_back(b); }
        }
        if (!todo.empty()) {
            std::vector<int> r2, l2, s2;
            if (hirschbergLevel(todo, r2, l2, s2)) return 1;
            for (size_t q = 0; q < todo.size(); ++q) { row[todoAt[q]] = r2[q]; ls[todoAt[q]] = l2[q]; rs[todoAt[q]] = s2[q]; }
        }
        // replace pieces back to front so the recorded indices stay valid
        for (size_t b = big.size(); b-- > 0;) {
            const size_t j = where[b].first, i = where[b].second;
            if (status[j] != EDLIB_STATUS_OK) continue;
            if (row[b] == -2) { status[j] = EDLIB_STATUS_ERROR; continue; }          // edlib.cpp:1358-1362
            const PathPiece pc = pieces[j][i];
            const int lw = pc.T / 2, ulH = row[b] + 1;                                // :1367-1370
            const PathPiece ul{pc.qoff, ulH, pc.toff, lw, ls[b]};
            const PathPiece lr{pc.qoff + ulH, pc.m - ulH, pc.toff + lw, pc.T - lw, rs[b]};
            pieces[j][i] = ul;
            pieces[j].insert(pieces[j].begin() + i + 1, lr);
        }
    }
    // leaves: trivial pieces on the host (edlib.cpp:1168-1175), the rest through store + traceback
    std::vector<UnitSpec> units;
    units.reserve(nj);
    for (size_t j = 0; j < nj; ++j) {
        if (status[j] != EDLIB_STATUS_OK) continue;
        for (size_t i = 0; i < npieces(j); ++i) {
            const PathPiece& pc = piece(j, i);
            // kinit = the piece's distance: the storing scan runs inside exactly that band (the reference's
            // second call with k = bestScore, edlib.cpp:1196-1199)
            if (pc.m > 0 && pc.T > 0) units.push_back(UnitSpec{pc.qoff, pc.m, 1, pc.toff, pc.T, 1, pc.score});
        }
    }
    lap("paths: levels+units");
    // smallest ring that holds the band (or all blocks) of each leaf; strips when none does
    std::vector<const uint8_t*> leafPtr(units.size(), nullptr); std::vector<int> leafLen(units.size(), 0);
    {
        const bool bandOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
        static const int rings[kNumRings + 1] = {4, 8, 16, 21, 32, 64, 0};
        std::vector<int> ringOfUnit(units.size(), 0);
        for (size_t u = 0; u < units.size(); ++u) {
            const int nb = (units[u].qlen + 63) / 64;
            for (int g = 0; g < kNumRings && !bandOff; ++g)
                if (nb <= rings[g] || units[u].kinit <= ring_max_k(rings[g])) { ringOfUnit[u] = rings[g]; break; }
        }
        size_t perRing[kNumRings + 1] = {0};
        for (size_t u = 0; u < units.size(); ++u) for (int g = 0; g <= kNumRings; ++g) if (ringOfUnit[u] == rings[g]) ++perRing[g];
        for (int g = 0; g <= kNumRings; ++g) {
            if (!perRing[g]) continue;
            SolveOut so;
            if (perRing[g] == units.size()) {                       // the usual case: one kind of leaf
                if (solve(EDLIB_MODE_NW, false, true, units, so, rings[g])) return 1;
                leafPtr.swap(so.opsPtr); leafLen.swap(so.opsLen);
            } else {
                std::vector<UnitSpec> sel; std::vector<size_t> who;
                sel.reserve(perRing[g]); who.reserve(perRing[g]);
                for (size_t u = 0; u < units.size(); ++u) if (ringOfUnit[u] == rings[g]) { sel.push_back(units[u]); who.push_back(u); }
                if (solve(EDLIB_MODE_NW, false, true, sel, so, rings[g])) return 1;
                for (size_t q = 0; q < sel.size(); ++q) { leafPtr[who[q]] = so.opsPtr[q]; leafLen[who[q]] = so.opsLen[q]; }
            }
            opsKeep_.insert(opsKeep_.end(), so.opsBufs.begin(), so.opsBufs.end());
        }
    }
    lap("paths: solve");
    // a job that was never split is its single leaf: hand out the view; split jobs concatenate their pieces
    std::vector<size_t> firstLeaf(nj + 1, 0);                 // leaves are listed job by job, piece by piece
    {
        size_t u = 0;
        for (size_t j = 0; j < nj; ++j) {
            firstLeaf[j] = u;
            if (status[j] != EDLIB_STATUS_OK) continue;
            for (size_t i = 0; i < npieces(j); ++i) if (piece(j, i).m > 0 && piece(j, i).T > 0) ++u;
        }
        firstLeaf[nj] = u;
    }
    for (size_t j = 0; j < nj; ++j) {
        if (status[j] != EDLIB_STATUS_OK) continue;
        OpsOut& o = opsOut[j];
        size_t u = firstLeaf[j];
        if (npieces(j) == 1 && piece(j, 0).m > 0 && piece(j, 0).T > 0) {
            o.p = leafPtr[u]; o.len = leafLen[u];
            continue;
        }
        for (size_t i = 0; i < npieces(j); ++i) {
            const PathPiece& pc = piece(j, i);
            if (pc.m == 0) o.own.insert(o.own.end(), (size_t)pc.T, (uint8_t)EDLIB_EDOP_DELETE);
            else if (pc.T == 0) o.own.insert(o.own.end(), (size_t)pc.m, (uint8_t)EDLIB_EDOP_INSERT);
            else { o.own.insert(o.own.end(), leafPtr[u], leafPtr[u] + leafLen[u]); ++u; }
        }
        o.p = o.own.data(); o.len = (int)o.own.size();
    }
    lap("paths: assemble");
    return 0;
}

// ------------------------------------------------- semi-global units on rings

// SHW / HW units of at most 4 (16) blocks share a wave 16 (4) at a time on the lane rings; longer ones take
// the strips.  Same outputs as solve().
// HW is shift-invariant (a scan that starts 2m-1 columns early from the fresh state reproduces the
// exact bottom-row scores of its own columns), and a unit of kernel W is one wave's serial walk over its target: a
// 1 kb query against a 5 Mb chromosome is 5M dependent steps (0.3 s) while 1023 SIMDs idle.  When a batch of HW units
// does not fill the chip, every unit with a long target is cut into target segments (each a unit of its own with a
// warm-up that records nothing: UnitSpec::skip) and the segments' answers are merged: minimum score, the end
// locations of the segments that attain it in order, the last of them.  Results never depend on the cut.
int Batch::solveSemiGlobal(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    if (mode == EDLIB_MODE_SHW && n > 0 && !(getenv("EDLIB_AMD_SHWBAND") && getenv("EDLIB_AMD_SHWBAND")[0] == '0')) {
        // (queries of up to four blocks sit whole on the smallest ring whatever their threshold: nothing to band)
        bool any = false;
        for (size_t i = 0; i < n && !any; ++i) any = units[i].qlen > 256;
        if (any) return solveShwBanded(wantPositions, units, out);
    }
    if (mode != EDLIB_MODE_HW || n == 0 || n >= 4096) return solveSemiGlobalUnits(mode, wantPositions, units, out);
    const long long smax = std::max<long long>(1, 8192 / (long long)n);
    std::vector<UnitSpec> sub; std::vector<int> firstSeg(n + 1, 0); std::vector<int> base;   // base: first recorded column of a segment
    bool any = false;
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& u = units[i];
        const long long segMin = std::max<long long>(4096, 8LL * u.qlen);
        const long long S = std::max<long long>(1, std::min<long long>(smax, u.tlen / segMin));
        const long long segLen = (u.tlen + S - 1) / S;
        for (long long sg = 0; sg < S; ++sg) {
            const long long c0 = sg * segLen, c1 = std::min<long long>(u.tlen, c0 + segLen);
            if (c0 >= c1) break;
            const long long cw = std::max<long long>(0, c0 - (2LL * u.qlen - 1));
            UnitSpec v = u;
            v.toff = u.toff + cw * u.tstep; v.tlen = (int)(c1 - cw); v.skip = (int)(c0 - cw);
            sub.push_back(v); base.push_back((int)cw);
        }
        firstSeg[i + 1] = (int)sub.size();
        any = any || firstSeg[i + 1] - firstSeg[i] > 1;
    }
    if (!any) return solveSemiGlobalUnits(mode, wantPositions, units, out);
    SolveOut so;
    if (solveSemiGlobalUnits(mode, wantPositions, sub, so)) return 1;
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    for (size_t i = 0; i < n; ++i) {
        int best = -1;
        for (int q = firstSeg[i]; q < firstSeg[i + 1]; ++q)
            if (so.score[q] >= 0 && (best < 0 || so.score[q] < best)) best = so.score[q];
        out.score[i] = best;
        if (best >= 0)
            for (int q = firstSeg[i]; q < firstSeg[i + 1]; ++q) {
                if (so.score[q] != best) continue;
                out.count[i] += so.count[q];
                out.last[i] = so.last[q] + base[q];
                for (long long k = so.posStart[q]; k < so.posStart[q + 1]; ++k) out.posFlat.push_back(so.posFlat[k] + base[q]);
            }
        out.posStart[i + 1] = (long long)out.posFlat.size();
    }
    return 0;
}

// SHW with a threshold: D[i][j] >= |i - j|, so a scan with threshold K only needs the diagonals [-K, K] and the first
// m + K columns (the reference's band for SHW, edlib.cpp:562, 602-630, written for a fixed k).  A unit with a real
// threshold (the reverse scans of HW start locations run with k = the distance, :253-257; calls with k >= 0) is scanned
// inside that band once; an open unit (k = -1: threshold m) climbs levels K = 256, 1024, 4096 ... like the reference
// doubles k (:197-217), the answer being exact as soon as some column scores <= K.  What it buys: the smallest ring that
// holds the BAND instead of the whole query (a 10 kb reverse scan with k = 100 on an 8-lane... here 16-lane ring, four
// units per wave, instead of five 2048-row strips), and m + K columns instead of 2 m.
int Batch::solveShwBanded(bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    out.score.assign(n, -1); out.count.assign(n, 0); out.last.assign(n, -1);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    std::vector<long long> kcur(n);
    std::vector<std::vector<int>> posOf(wantPositions ? n : 0);
    std::vector<size_t> rest;
    // a level K can only find a column when row m-1 is inside its band somewhere: T >= m - K (else the kernels would never
    // start the last block); levels that cannot are skipped, a unit whose own threshold cannot has no answer (-1)
    auto reachable = [&](const UnitSpec& u, long long K) { return (long long)u.tlen >= (long long)u.qlen - K; };
    for (size_t i = 0; i < n; ++i) {
        const UnitSpec& u = units[i];
        if (!reachable(u, std::min(u.kinit, u.qlen))) continue;
        kcur[i] = u.kinit < u.qlen ? u.kinit : 256;
        while (kcur[i] < u.kinit && !reachable(u, kcur[i])) kcur[i] *= 4;
        rest.push_back(i);
    }
    while (!rest.empty()) {
        std::vector<UnitSpec> sel; sel.reserve(rest.size());
        for (size_t i : rest) {
            UnitSpec u = units[i];
            const long long K = std::min<long long>(kcur[i], u.kinit);
            u.kinit = (int)K;
            u.band = K < u.qlen ? 1 : 0;
            u.tlen = (int)std::min<long long>(u.tlen, (long long)u.qlen + K);
            sel.push_back(u);
        }
        SolveOut so;
        if (solveSemiGlobalUnits(EDLIB_MODE_SHW, wantPositions, sel, so)) return 1;
        std::vector<size_t> again;
        for (size_t q = 0; q < sel.size(); ++q) {
            const size_t i = rest[q];
            if (so.score[q] >= 0 || sel[q].kinit >= units[i].kinit) {          // exact / the caller's own threshold found nothing
                out.score[i] = so.score[q]; out.count[i] = so.count[q]; out.last[i] = so.last[q];
                if (wantPositions) posOf[i].assign(so.posFlat.begin() + so.posStart[q], so.posFlat.begin() + so.posStart[q + 1]);
                continue;
            }
            kcur[i] = 4LL * sel[q].kinit;
            again.push_back(i);
        }
        rest.swap(again);
    }
    if (wantPositions)
        for (size_t i = 0; i < n; ++i) {
            out.posFlat.insert(out.posFlat.end(), posOf[i].begin(), posOf[i].end());
            out.posStart[i + 1] = (long long)out.posFlat.size();
        }
    return 0;
}

int Batch::solveSemiGlobalUnits(int mode, bool wantPositions, const std::vector<UnitSpec>& units, SolveOut& out)
{
    const size_t n = units.size();
    const bool ringsOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
    // units of up to 4 / 16 blocks on 4- / 16-lane rings, up to 32 / 64 blocks on 16-lane rings whose lanes hold 2 / 4
    // blocks (four units per wave, every lane busy: a 1025-base query on the strips uses 17 of a wave's 64 lanes), the
    // rest on the strips
    // more than 64 blocks: the strips as a pipeline over many waves (wide_kernels.hip) instead of one wave walking them
    // one after the other
    static const int rings[6] = {4, 16, 16, 16, 0, kWide}, ringH[6] = {1, 1, 2, 4, 1, 1};
    const int NG = 6;
    std::vector<int> grp(n, 4);
    size_t cnt[NG] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        const int nb = (units[i].qlen + 63) / 64;
        if (!ringsOff) grp[i] = nb <= 4 ? 0 : (nb <= 16 ? 1 : (nb <= 32 ? 2 : (nb <= 64 ? 3 : 4)));
        if (nb > 64) grp[i] = 5;
        // a banded SHW unit (UnitSpec::band) needs the ring that holds its band, not its query
        if (mode == EDLIB_MODE_SHW && units[i].band && !ringsOff) {
            // (the SHW band [-K, K] is 2 K + 1 rows wide, twice the NW band of the same threshold: ring_max_k / 2)
            const long long K2 = 2LL * units[i].kinit;
            if (nb > 4 && K2 <= ring_max_k(4)) grp[i] = 0;
            else if (nb > 16 && K2 <= ring_max_k(16)) grp[i] = 1;
            else if (nb > 32 && K2 <= ring_max_k(16, 2)) grp[i] = 2;
            else if (nb > 64 && K2 <= ring_max_k(16, 4)) grp[i] = 3;
        }
        ++cnt[grp[i]];
    }
    for (int g = 0; g < NG; ++g)
        if (cnt[g] == n) return solve(mode, wantPositions, false, units, out, rings[g], ringH[g]);   // the usual case: one kind
    SolveOut part[NG];
    std::vector<size_t> where(n);
    for (int g = 0; g < NG; ++g) {
        if (!cnt[g]) continue;
        std::vector<UnitSpec> sel; sel.reserve(cnt[g]);
        for (size_t i = 0; i < n; ++i) if (grp[i] == g) { where[i] = sel.size(); sel.push_back(units[i]); }
        if (solve(mode, wantPositions, false, sel, part[g], rings[g], ringH[g])) return 1;
    }
    out.score.resize(n); out.count.resize(n); out.last.resize(n);
    out.posStart.assign(n + 1, 0); out.posFlat.clear();
    out.opsPtr.assign(n, nullptr); out.opsLen.assign(n, 0); out.opsBufs.clear();
    for (size_t i = 0; i < n; ++i) {
        const SolveOut& p = part[grp[i]];
        const size_t q = where[i];
        out.score[i] = p.score[q]; out.count[i] = p.count[q]; out.last[i] = p.last[q];
        out.posFlat.insert(out.posFlat.end(), p.posFlat.begin() + p.posStart[q], p.posFlat.begin() + p.posStart[q + 1]);
        out.posStart[i + 1] = (long long)out.posFlat.size();
    }
    return 0;
}

// ------------------------------------------------------ NW distance levels

// The reference finds the NW distance by doubling k from 64 until the banded scan succeeds
// (edlib.cpp:197-217); any threshold >= the distance gives the same answer, so the levels here are the
// ring sizes of scan_pairs_ring_kernel: K = 128 on 4-lane rings (16 units per wave), 384 on 8, 896 on 16, 1216 on 21
// (three units per wave), 1920 on half waves, 3968 on whole waves, then the unbanded strips.  A unit whose blocks
// all fit a ring is exact on it for any distance (threshold max(m, T)).  A failed level is pure waste when the whole
// batch is divergent, so larger batches first measure the divergence of 64 strided units on their 1 kb prefixes
// (one small launch) and every unit starts at the level that holds its extrapolated distance.  The estimate only picks the starting level; results never depend on it.
int Batch::solveGlobalDistances(const std::vector<UnitSpec>& units, std::vector<int>& score, std::vector<OpsOut>* paths)
{
    const size_t n = units.size();
    score.assign(n, -1);
    if (paths) { paths->clear(); paths->resize(n); }
    if (n == 0) return 0;
    const bool bandOff = getenv("EDLIB_AMD_NWBAND") && getenv("EDLIB_AMD_NWBAND")[0] == '0';
    // ring levels (lanes, blocks per lane); level nl = unbanded strips.  Rings whose lanes hold 2 / 4 blocks (16 x 2: four
    // units per wave, DPP carry) were measured here in round 3 and lost: a ring computes ALL its rows every step, and
    // 16 x 2 = 2048 rows for a band that needs ~1300 is 52 % more block updates than the 21-lane ring's 1344, which the
    // cheaper step (105 against 119 SIMD cycles per block) does not pay back: config 4 24.7 ms of scans against 20.7.
    // They serve the semi-global units of 17..64 blocks instead (solveSemiGlobalUnits), where the alternative is a strip
    // that uses 17 of 64 lanes.
    static const int ringOf[kNumRings] = {4, 8, 16, 21, 32, 64}, ringH[kNumRings] = {1, 1, 1, 1, 1, 1};
    auto cap_of = [&](int l) { return ring_max_k(ringOf[l], ringH[l]); };
    auto blocks_of = [&](int l) { return ringOf[l] * ringH[l]; };
    const int nl = kNumRings;                                           // ring levels; level nl = unbanded strips
    const int kInf = 0x3fffffff;
    const int kcap = cfg_.k >= 0 ? cfg_.k : kInf;                       // answers above the caller's k are all alike
    auto blocks = [&](size_t i) { return (units[i].qlen + 63) / 64; };

    double rate = 0.0;                                                  // edits per base, median of the sample
    size_t maxBlocks = 0;
    for (size_t i = 0; i < n; ++i) maxBlocks = std::max<size_t>(maxBlocks, (size_t)blocks(i));
    // (units of at most 16 blocks climb cheap levels -- the 16-lane ring holds them whole -- and skip the probe)
    if (n >= 256 && maxBlocks > 16 && !bandOff && !getenv("EDLIB_AMD_NOPROBE")) {
        // 64 strided units, the first 1 kb of the query against the first 1 kb + 128 of the target in PREFIX mode
        // (the best end column is free: a global alignment of two equally cut prefixes would add the indel drift at
        // the cut to the count, about one edit in a hundred bases at ONT-like rates)
        const int np = 64, cut = 1024;
        std::vector<UnitSpec> probe(np);
        for (int i = 0; i < np; ++i) {
            UnitSpec u = units[(size_t)((long long)i * n / np)];
            u.qlen = std::min(u.qlen, cut); u.tlen = std::min(u.tlen, cut + 128);
            u.kinit = u.qlen;                                           // 16 blocks at most: the whole matrix on a 16-lane ring
            probe[i] = u;
        }
        SolveOut so;
        if (solve(EDLIB_MODE_SHW, false, false, probe, so, 16)) return 1;
        std::vector<double> r(np);
        for (int i = 0; i < np; ++i) r[i] = (double)std::max(so.score[i], 0) / std::max(1, probe[i].qlen);
        std::sort(r.begin(), r.end());
        rate = r[np / 2];
    }
    // First level of a unit: the smallest ring that holds all its blocks or its extrapolated distance.  The distance
    // of a unit of length L at rate r scatters like a sum of L Bernoulli trials (sigma = sqrt(r L)).  A ring of G
    // lanes costs G / 64 of a wave per unit, so trying the smaller ring first pays as long as fewer than a quarter
    // to a half of the units fail on it and move up: the estimate is the mean plus half a sigma (10 kb pairs at
    // 11.4 % sit under the 21-lane ring's 1216 -- three units per wave instead of two -- and the 7 % above it rerun).
    // (est = mean + sqrt(mean) / 2 + 8 <= cap is a bound on the mean: solved once per level, so that a unit costs a
    // multiply-add and a few compares -- the square root per unit was 2 ms of host time per 100,000 units)
    // A handful of LONG units (the reference's 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191): a level costs its
    // ~T dependent steps whether it succeeds or not (0.1 s per Mb), so each unit gets its own estimate from its first 4 kb
    // (PREFIX mode on a 16-lane ring of 4-block lanes: ~1 ms) instead of climbing.
    const bool wideLevel = paths == nullptr;                            // what follows the rings: the wide band (with the column store: the strips)
    // A handful of units (edlibAlign() on a long pair is one) are bound by DEPENDENT STEPS, not by work: a ring scan is
    // ~T steps of 0.12 us on one wave whether it succeeds or not, the wide kernel's two half scans are T / 2 steps of
    // 0.074 us on as many waves as the band is tall.  So when the strips of all units' whole matrices fit the resident
    // waves, units of 4 kb and more skip the rings: straight to two half scans, over the WHOLE matrix (no estimate, no
    // ladder, always exact) while that is at most 2e10 cells, inside a band from the unit's own first 4 kb beyond that
    // (the reference's 1 Mb Chromosome pairs, test_data/perf_tests.sh:180-191; PREFIX mode on a 16-lane ring of 4-block
    // lanes: ~1 ms against 40 ms per pass).
    std::vector<uint8_t> direct;
    std::vector<double> unitRate;
    if (wideLevel && rate == 0.0 && n <= 64 && !bandOff) {
        if (wideCap_ < 0) wideCap_ = wide_resident_waves(tab_.sigmaT);
        long long waves = 0;
        for (size_t i = 0; i < n; ++i)
            if (std::min(units[i].qlen, units[i].tlen) >= 4096)        // whole matrix: every strip is alive; a band: a few dozen
                waves += (double)units[i].qlen * (double)units[i].tlen <= 2e10 ? 2LL * ((units[i].qlen + 2047) / 2048) : 96;
        if (waves > 0 && waves <= wideCap_) {
            direct.assign(n, 0);
            for (size_t i = 0; i < n; ++i) direct[i] = std::min(units[i].qlen, units[i].tlen) >= 4096;
        }
    }
    auto whole_ok = [&](size_t i) { return (double)units[i].qlen * (double)units[i].tlen <= 2e10; };
    if (rate == 0.0 && n <= 512 && !bandOff && !getenv("EDLIB_AMD_NOPROBE")) {
        std::vector<UnitSpec> probe; std::vector<size_t> who;
        const int cut = 4096;
        for (size_t i = 0; i < n; ++i)
            if (direct.empty() ? std::min(units[i].qlen, units[i].tlen) >= 32768 : (direct[i] && !whole_ok(i))) {
                UnitSpec u = units[i];
                u.qlen = cut; u.tlen = std::min(u.tlen, cut + 512); u.kinit = cut;       // (never past the unit's own target)
                probe.push_back(u); who.push_back(i);
            }
        if (!probe.empty()) {
            SolveOut so;
            if (solve(EDLIB_MODE_SHW, false, false, probe, so, 16, 4)) return 1;
            unitRate.assign(n, 0.0);
            for (size_t q = 0; q < probe.size(); ++q) unitRate[who[q]] = (double)std::max(so.score[q], 0) / cut;
        }
    }
    auto mean_of = [&](size_t i) {
        const UnitSpec& u = units[i];
        return (unitRate.empty() ? rate : unitRate[i]) * std::min(u.qlen, u.tlen) + std::abs(u.qlen - u.tlen);
    };
    double meanCap[kNumRings + 1];
    auto mean_cap = [](double cap) { if (cap < 8) return -1.0; const double r = (-0.5 + std::sqrt(0.25 + 4.0 * (cap - 8.0))) / 2.0; return r * r; };
    for (int l = 0; l < nl; ++l) meanCap[l] = mean_cap(std::min<double>(cap_of(l), kcap));
    meanCap[nl] = mean_cap(2.0 * ring_max_k(64));
    int levelOfKcap = nl;                                               // est = kcap when the caller's k is the smaller one
    for (int l = nl - 1; l >= 0; --l) if (kcap <= cap_of(l)) levelOfKcap = l;
    auto first_level = [&](size_t i) {
        const double mean = mean_of(i);
        const int nbI = blocks(i);
        for (int l = 0; l < nl; ++l)
            if (nbI <= blocks_of(l) || mean <= meanCap[l] || l >= levelOfKcap) return l;
        // above every ring: the band on many waves.  (With the column store -- fused PATH levels -- what follows the rings is
        // the unbanded strips, nstrips times the work: the last ring is still tried while the estimate is within twice its limit.)
        if (!wideLevel) return (mean <= meanCap[nl] || kcap <= 2.0 * ring_max_k(64)) ? nl - 1 : nl;
        return nl;
    };
    std::vector<int>& lvl = lvlScratch_;
    lvl.resize(n);
    // A few units do not fill the chip at any ring size: a level then costs its ~T dependent steps on one wave
    // whether it succeeds or not (a 10 kb pair: 1.7 ms per level), so units with more blocks than a ring holds
    // go straight to whole-wave rings (K = 3968) instead of climbing.
    const bool fewUnits = n <= 512 && rate == 0.0;
    std::vector<size_t> atLevel(nl + 2, 0);
    {
        int lastQ = -1, lastT = -1, lastL = 0;                           // batches of equal shapes: one evaluation
        for (size_t i = 0; i < n; ++i) {
            if (units[i].qlen != lastQ || units[i].tlen != lastT || !direct.empty()) {
                lastQ = units[i].qlen; lastT = units[i].tlen;
                lastL = bandOff ? nl : first_level(i);
                if (!bandOff && fewUnits && lastL < nl - 1 && blocks(i) > blocks_of(lastL)) lastL = nl - 1;
                if (!direct.empty() && direct[i]) lastL = nl;
            }
            lvl[i] = lastL;
            ++atLevel[lastL];
        }
    }
    Lap lap;
    for (int l = 0; l <= nl; ++l) {
        if (atLevel[l] == 0) continue;
        if (l == nl && wideLevel) break;
        std::vector<UnitSpec>& sel = selScratch_; std::vector<size_t>& who = whoScratch_;
        sel.clear(); who.clear();
        sel.reserve(atLevel[l]); who.reserve(atLevel[l]);
        for (size_t i = 0; i < n; ++i) {
            if (lvl[i] != l) continue;
            UnitSpec u = units[i];
            if (l < nl) u.kinit = std::min(kcap, blocks(i) <= blocks_of(l) ? std::max(u.qlen, u.tlen) : cap_of(l));
            sel.push_back(u); who.push_back(i);
        }
        if (sel.empty()) continue;
        lap("nw level: select");
        SolveOut& so = soLevel_;
        const bool store = paths != nullptr && (l == nl || ringH[l] == 1);
        if (solve(EDLIB_MODE_NW, false, store, sel, so, l < nl ? ringOf[l] : 0, l < nl ? ringH[l] : 1)) return 1;
        lap("nw level: solve");
        if (store) opsKeep_.insert(opsKeep_.end(), so.opsBufs.begin(), so.opsBufs.end());
        for (size_t q = 0; q < sel.size(); ++q) {
            const size_t i = who[q];
            if (store && (l == nl || so.score[q] <= sel[q].kinit)) { (*paths)[i].p = so.opsPtr[q]; (*paths)[i].len = so.opsLen[q]; }
            if (l == nl || so.score[q] <= sel[q].kinit) score[i] = so.score[q];         // exact
            else if (sel[q].kinit >= kcap) score[i] = kInf;                              // > k: final
            else { lvl[i] = l + 1; ++atLevel[l + 1]; }                                   // next level
        }
        lap("nw level: scores");
    }
    // ---- beyond the rings: Ukkonen's band of ANY width on many waves (wide_kernels.hip).  The reference keeps doubling k
    // (edlib.cpp:197-217); a pass here costs about T dependent steps whatever its K, so the first K is generous (1.5 x the
    // estimate) and a failed pass doubles it.  K = max(m, T) is the whole matrix and always exact.
    if (wideLevel && atLevel[nl] > 0) {
        std::vector<size_t> rest;
        std::vector<long long> kcur(n, 0);
        for (size_t i = 0; i < n; ++i)
            if (lvl[i] == nl) {
                rest.push_back(i);
                const double est = mean_of(i);
                const bool dir = !direct.empty() && direct[i];
                kcur[i] = std::max<long long>(dir ? 1024 : 2LL * (ring_max_k(64) + 128), (long long)(1.5 * est + 4.0 * std::sqrt(est) + 64.0));
                if (dir && whole_ok(i)) kcur[i] = std::max(units[i].qlen, units[i].tlen);
                if (const char* e = getenv("EDLIB_AMD_WIDE_K0")) { if (atoi(e) > 0) kcur[i] = atoi(e); }     // (tests: the ladder from a small K)
            }
        // long units: two half scans that meet in the middle (solveWideSplit: half the dependent steps); a unit of one
        // target column has no two halves
        const int splitMin = direct.empty() ? 16384 : 4096;
        while (!rest.empty()) {
            std::vector<UnitSpec>& sel = selScratch_;
            sel.clear();
            std::vector<UnitSpec> halves; std::vector<size_t> whoWhole, whoHalves;
            for (size_t i : rest) {
                UnitSpec u = units[i];
                u.kinit = (int)std::min<long long>(std::min<long long>(kcap, kcur[i]), std::max(u.qlen, u.tlen));
                if (splitMin > 0 && std::min(u.qlen, u.tlen) >= splitMin && u.tlen >= 2) { halves.push_back(u); whoHalves.push_back(i); }
                else { sel.push_back(u); whoWhole.push_back(i); }
            }
            std::vector<size_t> again;
            auto settle = [&](size_t i, const UnitSpec& u, int got) -> int {
                if (got >= 0 && got <= u.kinit) score[i] = got;                                   // exact
                else if (u.kinit >= kcap) score[i] = kInf;                                          // > k: final
                else if (u.kinit >= std::max(u.qlen, u.tlen)) { set_error("wide band: no score inside the whole matrix"); return 1; }
                else { kcur[i] = 2LL * u.kinit; again.push_back(i); }
                return 0;
            };
            if (!sel.empty()) {
                SolveOut& so = soLevel_;
                if (solve(EDLIB_MODE_NW, false, false, sel, so, kWide)) return 1;
                for (size_t q = 0; q < sel.size(); ++q) if (settle(whoWhole[q], sel[q], so.score[q])) return 1;
            }
            if (!halves.empty()) {
                std::vector<int> sp;
                if (solveWideSplit(halves, sp)) return 1;
                for (size_t q = 0; q < halves.size(); ++q) {
                    const UnitSpec& u = halves[q];
                    if (sp[4 * q] <= u.kinit && u.qstep == 1 && u.tstep == 1)
                        knownSplits_.push_back(KnownSplit{u.qoff, u.qlen, u.toff, u.tlen, sp[4 * q], sp[4 * q + 1], sp[4 * q + 2], sp[4 * q + 3]});
                    if (settle(whoHalves[q], u, sp[4 * q])) return 1;
                }
            }
            lap("nw wide level");
            rest.swap(again);
        }
    }
    return 0;
}

// --------------------------------------------------------------------- run

// Every failure path of a run ends here: the device's wide gate goes back (another batch on the device -- a resident Python
// batch kept after an error, a second host thread -- would otherwise wait for it for ever), behind a synchronisation so
// that no spinning launch of this batch is still on the device when the next one takes the gate.
int Batch::run()
{
    const int rc = runImpl();
    if (rc && wideGateHeld_) {
        if (stream_) { DeviceGuard guard(device_); (void)hipStreamSynchronize(stream_); (void)hipGetLastError(); }
        wideGateRelease();
    }
    return rc;
}

int Batch::runImpl()
{
    pool_quarantine(false);
    Lap lap;
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    const long long cells = stats.cells;
    stats = EdlibAmdBatchStats{};
    stats.cells = cells;
    scanEventsUsed_ = 0;
    haveResults_ = false;
    opsKeep_.clear();            // (the previous run's views die with the reset of their records below)
    knownSplits_.clear();
    wideGateRelease();           // (a run that failed between a wide launch and its check)
    wideSerial_ = false;         // every run tries the pipelined strips first
    viewReady_ = false; cigar_[0].ready = cigar_[1].ready = false; lastRunFlat_ = false;      // (views of the previous run end here)
    opsOwned_.clear();
    // TASK_DISTANCE over reads-path units only: nothing is assembled on the host until results() asks for it, so
    // the per-unit records (160 bytes each) are not even allocated in the timed run
    const bool lazy = (cfg_.task == EDLIB_TASK_DISTANCE && pairUnits_.empty() && longUnits_.empty() && emptyUnits_.empty() && !groups_.empty()) || flatPairs_;
    pairsCollected_ = true;
    // the records of the run before last are recycled (no 160-byte-per-unit allocation + page faults per run)
    std::vector<UnitResult>& res = work_;
    bool deferReset = false;
    if (lazy) res.clear();
    else {
        const size_t keep = std::min(res.size(), (size_t)n_);
        res.resize((size_t)n_);
        // a batch of pair units only rewrites every record in its finalize loop: the recycled records are blanked there, in
        // the same pass over the 16 MB of 100,000 records, instead of in a walk of their own (0.4 ms)
        deferReset = emptyUnits_.empty() && groups_.empty() && longUnits_.empty() && !flatPairs_ && pairUnits_.size() == (size_t)n_;
        if (!deferReset) for (size_t u = 0; u < keep; ++u) blank_record(res[u]);
    }
    results_.clear();            // views of the previous run die before their staging blocks
    const int mode = (int)cfg_.mode;
    const int scanMode = (mode == EDLIB_MODE_HW || mode == EDLIB_MODE_SHW) ? mode : EDLIB_MODE_NW;
    EDLIB_AMD_HIP(hipEventRecord(evRun0_.e, stream_));
    if (d_ringSteps_.p) EDLIB_AMD_HIP(hipMemsetAsync(d_ringSteps_.p, 0, sizeof(unsigned long long), stream_));
    ringStepsUsed_ = false;
    lap("run: reset records");

    // ---- empty sequences: answered without any DP (edlib.cpp:166-184)
    for (int u : emptyUnits_) {
        UnitResult& r = res[u];
        const int m = qlen(u), T = tlen(u);
        if (mode == EDLIB_MODE_NW) { r.editDistance = std::max(m, T); r.ends.assign(1, T - 1); r.hasEnds = true; }
        else if (mode == EDLIB_MODE_SHW || mode == EDLIB_MODE_HW) { r.editDistance = m; r.ends.assign(1, -1); r.hasEnds = true; }
        else r.status = EDLIB_STATUS_ERROR;
    }
    if (alphabetLengthsBegin()) return 1;
    bool flatDone = false;
    if (flatPairs_) {                                   // ---- a flat pair batch: everything stays on the device
        bool over = false, fell = false;
        if (runPairsFlat(over, fell)) return 1;
        if (fell) {
            // something the flat layouts do not hold (more than 16 end locations with starts / paths asked for, a band
            // level that failed): this run takes the general path from the start
            res.resize((size_t)n_);
            for (size_t u = 0; u < res.size(); ++u) blank_record(res[u]);
        } else { flatDone = true; pairsCollected_ = false; lastRunFlat_ = true; }
        lap("run: flat pairs");
    }
    // ---- phase 1: distance + end locations
    if (packTarget()) return 1;
    if (runReads()) return 1;
    readsCollected_ = groups_.empty();
    // TASK_DISTANCE leaves the reads-path results in HBM until results(); LOC/PATH need them now
    lap("run: reads scans");
    if (!readsCollected_ && cfg_.task != EDLIB_TASK_DISTANCE && collectReads(res)) return 1;
    lap("run: collect reads");
    // long HW queries against the shared target: piece filter + window verification; what it hands back (low
    // complexity, thresholds beyond a quarter of the piece) joins the pair units below
    pairNow_ = pairUnits_;
    if (!longUnits_.empty()) {
        std::vector<int> fb;
        if (solveLongReads(res, fb)) return 1;
        lap("run: long reads");
        // What the filter hands back is mostly unrelated sequence: every row of every column is needed.  Up to 1024 rows
        // (512 above four target symbols) the lane-per-read full-height kernel does that at ~20 VALU ops per 64 rows and
        // column with every lane busy; longer queries take kernel W's strips.
        const int fullMax = syms_ == 4 ? 32 * kMaxLongReadWords4 : (syms_ == 8 ? 32 * kMaxLongReadWords : 0);
        std::vector<std::vector<int>> byWords(kMaxLongReadWords4 + 1);
        std::vector<int> tall;
        for (int u : fb) {
            if (qlen(u) <= fullMax) byWords[read_group_words(qlen(u))].push_back(u);
            else if (fullMax > 0) tall.push_back(u);
            else pairNow_.push_back(u);
        }
        for (int w = 1; w <= kMaxLongReadWords4; ++w) {
            if (byWords[w].empty()) continue;
            std::unique_ptr<ReadGroup> g;
            // (many short segments here: a single strip warms up over at most 2047 columns, and 7,930 waves balance themselves
            // over the SIMDs where one round of 1,014 does not -- 513 / 768 / 1024-base reads: 110 / 125 / 166 ms against
            // 129 / 144 / 172 with makeGroup's oneRoundWaves; the chained strips of solveTallFull warm up over 2m - 1 columns
            // of the WHOLE query, which is what makes one round the better plan there)
            if (makeGroup(byWords[w], w, g)) return 1;
            stats.path |= 1;
            if (runGroupScans(*g, true) || runGroupExact(*g) || collectGroup(*g, res)) return 1;
        }
        if (!tall.empty()) {                    // taller: strips of fullMax rows on the same kernel, chained through HBM
            std::vector<int> back;
            if (solveTallFull(tall, res, back)) return 1;
            pairNow_.insert(pairNow_.end(), back.begin(), back.end());
        }
        lap("run: handed back (full height)");
    }
    if (banded_ && (!groups_.empty() || !longUnits_.empty())) {   // read back with the run's final synchronisation
        EDLIB_AMD_HIP(h_wordSteps_.alloc(sizeof(unsigned long long)));
        EDLIB_AMD_HIP(hipMemcpyAsync(h_wordSteps_.p, d_wordSteps_.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
        wordStepsPending_ = true;
    }
    if (flatDone) pairNow_.clear();
    const std::vector<int>& pairUnits_ = pairNow_;          // (shadows the member: the units of THIS run's pair phase)
    if (!pairUnits_.empty()) {
        // scratch that a run needs per unit lives in the batch: a fresh 10 MB std::vector is an mmap, its page faults
        // and a munmap (45 MB of them were 5 of the 9 ms a run over 262,144 short pairs took)
        std::vector<UnitSpec>& units = pairSpecs_;
        // (the specs depend on the batch only: a run over the same pair units as the last one keeps them)
        if (pairSpecsFor_ != pairUnits_) {
            units.resize(pairUnits_.size());
            for (size_t i = 0; i < units.size(); ++i) {
                const int u = pairUnits_[i], m = qlen(u);
                // (SHW: D[m][j] >= j - m > m >= best beyond column 2m: the rest of a long target cannot matter)
                const int T = scanMode == EDLIB_MODE_SHW ? (int)std::min<long long>(tlen(u), 2LL * m + 1) : tlen(u);
                units[i] = UnitSpec{qoff_[u], m, 1, tbase(u), T, 1,
                                    (cfg_.k < 0 || cfg_.k > m) ? m : cfg_.k};
            }
            pairSpecsFor_ = pairUnits_;
        }
        SolveOut& so = soMain_;
        if (scanMode == EDLIB_MODE_NW) {
            std::vector<int>& score = scoreMain_;
            // TASK_PATH over pairs that all stay below the 1 MiB rule (edlib.cpp:1188-1190): the reference scans twice
            // (distance, then the storing scan with k = distance, :1196-1199); here a unit's first successful level
            // stores its columns and is traced back right away -- any threshold >= the distance gives the same walk
            // (every neighbour that could be "one less than here" is <= the distance, hence exact inside the band)
            bool fuse = cfg_.task == EDLIB_TASK_PATH && mode == EDLIB_MODE_NW;
            for (size_t i = 0; fuse && i < units.size(); ++i) fuse = !needs_hirschberg(units[i].qlen, units[i].tlen);
            fusedOps_.clear();
            lap("run: pair specs");
            if (solveGlobalDistances(units, score, fuse ? &fusedOps_ : nullptr)) return 1;
            lap("run: global distances");
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[pairUnits_[i]];
                if (deferReset) blank_record(r);
                finalize_global(r, cfg_.k, mode, units[i].tlen, score[i]);
            }
            if (fuse)
                for (size_t i = 0; i < units.size(); ++i) {
                    UnitResult& r = res[pairUnits_[i]];
                    if (r.editDistance >= 0 && fusedOps_[i].p) { r.opsView = fusedOps_[i].p; r.opsViewLen = fusedOps_[i].len; r.hasAlignment = true; }
                }
        } else {
            if (solveSemiGlobal(scanMode, true, units, so)) return 1;
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[pairUnits_[i]];
                if (deferReset) blank_record(r);
                finalize_semiglobal(r, cfg_.k, units[i].qlen, so.score[i],
                                    so.posFlat.data() + so.posStart[i], so.posStart[i + 1] - so.posStart[i]);
            }
        }
    }
    lap("run: finalize pairs");
    if (!flatDone && alphabetLengthsEnd(res)) return 1;      // alphabetLength for everything the reads path did not cover (flat pairs: at collection)
    lap("run: phase 1 (distance)");
    std::vector<int>& live = live_;            // non-empty units with a solution (only the later phases want them)
    live.clear();
    // (a flat batch has done its phases 2 and 3 on the device: its records do not exist yet)
    const bool laterPhases = !flatDone && (cfg_.task == EDLIB_TASK_LOC || cfg_.task == EDLIB_TASK_PATH);
    if (laterPhases)
        for (int u = 0; u < n_; ++u)
            if (qlen(u) > 0 && tlen(u) > 0 && res[u].editDistance >= 0) live.push_back(u);

    // ---- phase 2: start locations (edlib.cpp:228-272)
    if (laterPhases) {
        std::vector<UnitSpec>& units = startUnits_; std::vector<std::pair<int, int>>& where = startWhere_;   // capacity kept across runs
        units.clear(); where.clear();
        units.reserve(live.size() + live.size() / 8); where.reserve(live.size() + live.size() / 8);
        for (int u : live) {
            UnitResult& r = res[u];
            r.hasStarts = true;
            r.starts.assign(r.ends.size(), 0);
            if (mode != EDLIB_MODE_HW) continue;
            const int m = qlen(u);
            for (size_t j = 0; j < r.ends.size(); ++j) {
                const int e = r.ends[j];
                if (e == -1) continue;                                   // :237-249
                // reverse query against the reversed prefix target[0..e], prefix mode, k = distance
                // (:253-257); columns past m+distance cannot score <= distance, so the window stops there
                const long long win = std::min<long long>((long long)e + 1, (long long)m + r.editDistance);
                units.push_back(UnitSpec{qoff_[u] + m - 1, m, -1, tbase(u) + e, (int)win, -1, r.editDistance});
                where.push_back({u, (int)j});
            }
        }
        lap("starts: units");
        if (!units.empty()) {
            SolveOut so;
            if (solveSemiGlobal(EDLIB_MODE_SHW, false, units, so)) return 1;
            lap("starts: solve");
            for (size_t i = 0; i < units.size(); ++i) {
                UnitResult& r = res[where[i].first];
                // last reported position of the reverse scan (:260); -1 when only the empty prefix qualifies
                r.starts[where[i].second] = r.ends[where[i].second] - so.last[i];
            }
        }
    }
    lap("run: phase 2 (starts)");
    // ---- phase 3: alignment path of the first location (edlib.cpp:276-289, 1161-1213)
    if (cfg_.task == EDLIB_TASK_PATH) {
        std::vector<PathPiece> jobs; std::vector<int> where;
        jobs.reserve(live.size()); where.reserve(live.size());
        for (int u : live) {
            UnitResult& r = res[u];
            if (r.ends.empty() || r.hasAlignment) continue;         // (hasAlignment: traced back in phase 1)
            const int m = qlen(u);
            const int s = r.starts[0], e = r.ends[0];
            const int len = e - s + 1;
            if (len <= 0) {                                                                         // :1168-1175
                opsOwned_.emplace_back((size_t)m, (uint8_t)EDLIB_EDOP_INSERT);
                r.opsView = opsOwned_.back().data(); r.opsViewLen = m; r.hasAlignment = true; continue;
            }
            jobs.push_back(PathPiece{qoff_[u], m, tbase(u) + s, len, r.editDistance});
            where.push_back(u);
        }
        lap("paths: jobs");
        if (!jobs.empty()) {
            std::vector<OpsOut> ops; std::vector<int> st;
            if (solvePaths(jobs, ops, st)) return 1;
            for (size_t i = 0; i < jobs.size(); ++i) {
                UnitResult& r = res[where[i]];
                if (st[i] != EDLIB_STATUS_OK) { r.status = EDLIB_STATUS_ERROR; continue; }
                if (!ops[i].own.empty()) {
                    opsOwned_.emplace_back(std::move(ops[i].own));
                    r.opsView = opsOwned_.back().data(); r.opsViewLen = (int)opsOwned_.back().size();
                } else { r.opsView = ops[i].p; r.opsViewLen = ops[i].len; }
                r.hasAlignment = true;
            }
        }
    }
    lap("run: phase 3 (paths)");
    if (ringStepsUsed_) EDLIB_AMD_HIP(hipMemcpyAsync(h_ringSteps_.p, d_ringSteps_.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    EDLIB_AMD_HIP(hipEventRecord(evRun1_.e, stream_));
    EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
    if (ringStepsUsed_) stats.word_steps += (long long)*reinterpret_cast<unsigned long long*>(h_ringSteps_.p);
    float ms = 0;
    EDLIB_AMD_HIP(hipEventElapsedTime(&ms, evRun0_.e, evRun1_.e));
    stats.run_ms = ms;
    if (wordStepsPending_) { stats.word_steps += (long long)*reinterpret_cast<unsigned long long*>(h_wordSteps_.p); wordStepsPending_ = false; }
    for (size_t i = 0; i < scanEventsUsed_; ++i) {
        float t = 0;
        EDLIB_AMD_HIP(hipEventElapsedTime(&t, scanEvents_[i].first, scanEvents_[i].second));
        stats.scan_ms += t;
    }
    // algorithmic bytes: target + query + Peq + result header + end locations
    if ((!readsCollected_ && pairUnits_.empty() && longUnits_.empty() && emptyUnits_.empty()) || flatDone) {
        // everything is still resident on the device (reads path or flat pairs, TASK_DISTANCE): every unit is priced with
        // sigma = |target alphabet| and one end location -- a constant of the batch, summed once
        if (algoBase_ < 0) {
            algoBase_ = 0;
            for (int u = 0; u < n_; ++u) {
                const long long m = qlen(u);
                algoBase_ += tlen(u) + m + 8LL * (tab_.sigmaT + 1) * ((m + 63) / 64) + 16 + 4;
            }
        }
        stats.algo_bytes = algoBase_;
        algoDirty_ = false;
    } else algoDirty_ = true;                   // a walk over every record: done when somebody asks (finishStats)
    results_.swap(res);
    haveResults_ = true;
    lap("run: stats");
    return 0;
}

// algorithmic bytes of the last run: target + query + Peq + result header + end locations.  A walk
// over every record: done when the statistics are asked for, not in every run.
void Batch::finishStats()
{
    if (!algoDirty_) return;
    if (haveResults_ && ensureCollected()) return;   // (a DISTANCE batch that mixes reads-path and pair units: the read units' records are still on the device)
    algoDirty_ = false;
    const std::vector<UnitResult>& res = results_;
    stats.algo_bytes = 0;
    for (int u = 0; u < n_ && (size_t)u < res.size(); ++u) {
        const long long m = qlen(u);
        const long long sigma = res[u].alphabetLength ? res[u].alphabetLength : tab_.sigmaT;
        stats.algo_bytes += tlen(u) + m + 8LL * (sigma + 1) * ((m + 63) / 64) + 16
                            + 4LL * std::max<long long>(1, (long long)res[u].ends.size());
    }
}

// ------------------------------------------------------------ marshalling

static int* malloc_ints(const LocList& v) {
    int* p = static_cast<int*>(malloc(sizeof(int) * std::max<size_t>(v.size(), 1)));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(int));
    return p;
}

int Batch::results(EdlibAlignResult* out)
{
    // the failure contract of edlib_amd.h: on ANY failure every entry is blank with status ERROR (nothing to free)
    for (int u = 0; u < n_; ++u) {
        EdlibAlignResult& o = out[u];
        o.status = EDLIB_STATUS_ERROR; o.editDistance = -1; o.endLocations = nullptr; o.startLocations = nullptr;
        o.numLocations = 0; o.alignment = nullptr; o.alignmentLength = 0; o.alphabetLength = 0;
    }
    if (!haveResults_) { set_error("results() before a successful run()"); return 1; }
    if (ensureCollected()) return 1;
    std::atomic<int> oom(0);                      // a failed malloc: the unit reports EDLIB_STATUS_ERROR, the call fails
    auto marshal = [&](int lo, int hi) {
        for (int u = lo; u < hi; ++u) {
            const UnitResult& r = results_[u];
            EdlibAlignResult& o = out[u];
            o.status = r.status;
            o.editDistance = r.editDistance;
            o.endLocations = nullptr; o.startLocations = nullptr; o.numLocations = 0;
            o.alignment = nullptr; o.alignmentLength = 0;
            o.alphabetLength = r.alphabetLength;
            if (r.hasEnds) { o.endLocations = malloc_ints(r.ends); o.numLocations = (int)r.ends.size(); }
            if (r.hasStarts) o.startLocations = malloc_ints(r.starts);
            if (r.hasAlignment) {
                const uint8_t* src = r.opsView;
                const size_t len = (size_t)r.opsViewLen;
                o.alignment = static_cast<unsigned char*>(malloc(std::max<size_t>(len, 1)));
                if (len && o.alignment) memcpy(o.alignment, src, len);
                o.alignmentLength = (int)len;
            }
            if ((r.hasEnds && !o.endLocations) || (r.hasStarts && !o.startLocations) || (r.hasAlignment && !o.alignment)) {
                free(o.endLocations); free(o.startLocations); free(o.alignment);
                o.endLocations = nullptr; o.startLocations = nullptr; o.alignment = nullptr;
                o.numLocations = 0; o.alignmentLength = 0; o.status = EDLIB_STATUS_ERROR; o.editDistance = -1;
                oom.store(1);
            }
        }
    };
    // one malloc per array is the reference's ownership contract (edlib.h:177-205); a million of them are worth a
    // few threads (glibc arenas are per thread; free() of a block from any thread is fine)
    if (n_ >= 65536) {
        const int nthreads = host_threads(6);
        std::vector<std::thread> th;
        th.reserve(nthreads);
        int done = 0;                                 // units [0, done) are covered by a started thread
        {
            ThreadJoiner join(th);
            try {
                for (int t = 0; t < nthreads; ++t) {
                    const int hi = (int)((long long)n_ * (t + 1) / nthreads);
                    th.emplace_back(marshal, done, hi);
                    done = hi;
                }
            } catch (const std::system_error&) {}     // thread limit: this thread does the rest
        }
        if (done < n_) marshal(done, n_);
    } else marshal(0, n_);
    if (oom.load()) {                             // all or nothing: the caller gets no half-filled array to clean up
        for (int u = 0; u < n_; ++u) {
            EdlibAlignResult& o = out[u];
            free(o.endLocations); free(o.startLocations); free(o.alignment);
            o.endLocations = nullptr; o.startLocations = nullptr; o.alignment = nullptr;
            o.numLocations = 0; o.alignmentLength = 0; o.status = EDLIB_STATUS_ERROR; o.editDistance = -1;
        }
        set_error("out of host memory while marshalling results");
        return 1;
    }
    return 0;
}

// The caller-facing arrays of a general batch, from its records (a flat batch makes them on the device: engine_flat.hip).
int Batch::buildHostView()
{
    if (viewReady_) return 0;
    if (ensureCollected()) return 1;
    const size_t n = (size_t)n_;
    long long nloc = 0, naln = 0;
    bool anyStarts = false;
    for (size_t u = 0; u < n; ++u) {
        const UnitResult& r = results_[u];
        nloc += r.hasEnds ? (long long)r.ends.size() : 0;
        naln += r.hasAlignment ? (long long)r.opsViewLen : 0;
        anyStarts = anyStarts || r.hasStarts;
    }
    viewInts_.assign(4 * n + 2 * (size_t)nloc + 2, 0); viewOffs_.assign(2 * (n + 1), 0); viewOps_.assign((size_t)naln + 1, 0);
    int* st = viewInts_.data(); int* ed = st + n; int* nl = ed + n; int* al = nl + n; int* ends = al + n; int* starts = ends + nloc;
    long long* lo = viewOffs_.data(); long long* ao = lo + n + 1;
    long long li = 0, ai = 0;
    for (size_t u = 0; u < n; ++u) {
        const UnitResult& r = results_[u];
        st[u] = r.status; ed[u] = r.editDistance; al[u] = r.alphabetLength;
        const size_t c = r.hasEnds ? r.ends.size() : 0;
        nl[u] = (int)c; lo[u] = li; ao[u] = ai;
        if (c) memcpy(ends + li, r.ends.data(), c * sizeof(int));
        for (size_t i = 0; i < c; ++i) starts[li + (long long)i] = r.hasStarts ? r.starts[i] : -1;
        li += (long long)c;
        if (r.hasAlignment && r.opsViewLen > 0) { memcpy(viewOps_.data() + ai, r.opsView, (size_t)r.opsViewLen); ai += r.opsViewLen; }
    }
    lo[n] = li; ao[n] = ai;
    view_ = EdlibAmdResultsView{};
    view_.numUnits = n_; view_.status = st; view_.editDistance = ed; view_.numLocations = nl; view_.alphabetLength = al;
    view_.locOffsets = lo; view_.endLocations = ends; view_.startLocations = anyStarts ? starts : nullptr;
    view_.alnOffsets = ao; view_.alignment = cfg_.task == EDLIB_TASK_PATH ? viewOps_.data() : nullptr;
    viewAlnDev_ = nullptr; viewAlnOffDev_ = nullptr;
    viewReady_ = true;
    return 0;
}

int Batch::resultsView(EdlibAmdResultsView* out)
{
    if (!haveResults_) { set_error("results before a successful run()"); return 1; }
    DeviceGuard guard(device_);
    EDLIB_AMD_HIP(guard.status);
    if (!viewReady_ && (lastRunFlat_ ? buildFlatView() : buildHostView())) return 1;
    if (out) *out = view_;
    return 0;
}

// Flat form of results() as malloc'd copies of the view (edlibAmdBatchResultsFlat).
int Batch::resultsFlat(int* status, int* editDistance, int* numLocations, int* alphabetLength,
                       long long* locOffsets, int** endLocations, int** startLocations,
                       long long* alnOffsets, unsigned char** alignment)
{
    if (endLocations) *endLocations = nullptr;
    if (startLocations) *startLocations = nullptr;
    if (alignment) *alignment = nullptr;
    EdlibAmdResultsView v;
    if (resultsView(&v)) return 1;
    const size_t n = (size_t)n_;
    if (status) memcpy(status, v.status, n * sizeof(int));
    if (editDistance) memcpy(editDistance, v.editDistance, n * sizeof(int));
    if (numLocations) memcpy(numLocations, v.numLocations, n * sizeof(int));
    if (alphabetLength) memcpy(alphabetLength, v.alphabetLength, n * sizeof(int));
    if (locOffsets) memcpy(locOffsets, v.locOffsets, (n + 1) * sizeof(long long));
    if (alnOffsets) memcpy(alnOffsets, v.alnOffsets, (n + 1) * sizeof(long long));
    const long long nloc = v.locOffsets[n], naln = v.alnOffsets[n];
    int* ends = endLocations ? static_cast<int*>(malloc(sizeof(int) * (size_t)std::max<long long>(nloc, 1))) : nullptr;
    int* starts = (startLocations && v.startLocations) ? static_cast<int*>(malloc(sizeof(int) * (size_t)std::max<long long>(nloc, 1))) : nullptr;
    unsigned char* aln = alignment ? static_cast<unsigned char*>(malloc((size_t)std::max<long long>(naln, 1))) : nullptr;
    if ((endLocations && !ends) || (startLocations && v.startLocations && !starts) || (alignment && !aln)) {
        free(ends); free(starts); free(aln); set_error("out of memory"); return 1;
    }
    if (ends && nloc) memcpy(ends, v.endLocations, (size_t)nloc * sizeof(int));
    if (starts && nloc) memcpy(starts, v.startLocations, (size_t)nloc * sizeof(int));
    if (aln && naln && v.alignment) memcpy(aln, v.alignment, (size_t)naln);
    if (endLocations) *endLocations = ends;
    if (startLocations) *startLocations = starts;
    if (alignment) *alignment = aln;
    return 0;
}

// edlibAlignmentToCigar (edlib.cpp:303-350) over every op string of the last run.  A flat batch's op bytes are dense on the
// device: lengths, a prefix sum and the strings are three launches there (flat_results.hip) and one block comes back;
// other batches run-length encode their view on the host.
int Batch::cigarView(int format, const char** chars, const long long** offsets)
{
    if (format != EDLIB_CIGAR_STANDARD && format != EDLIB_CIGAR_EXTENDED) { set_error("unknown CIGAR format"); return 1; }
    EdlibAmdResultsView v;
    if (resultsView(&v)) return 1;
    CigarOut& c = cigar_[format == EDLIB_CIGAR_STANDARD ? 1 : 0];
    const size_t n = (size_t)n_;
    if (!c.ready) {
        DeviceGuard guard(device_);
        EDLIB_AMD_HIP(guard.status);
        if (viewAlnDev_ && v.alignment) {
            const size_t nblocks = (n + 255) / 256;
            EDLIB_AMD_HIP(d_cigWork_.ensure(3 * n + nblocks + 4));
            long long* cigLen = d_cigWork_.p; long long* cigRel = cigLen + n; long long* blockTot = cigRel + n;
            long long* totals = blockTot + nblocks; long long* cigOff = totals + 2;
            const int standard = format == EDLIB_CIGAR_STANDARD ? 1 : 0;
            EDLIB_AMD_HIP(launch_cigars(viewAlnDev_, viewAlnOffDev_, n_, standard, cigLen, cigRel, blockTot, totals, nullptr, cigOff, 0, stream_));
            if (c.offs.n < (n + 1) * sizeof(long long)) EDLIB_AMD_HIP(c.offs.alloc((n + 1) * sizeof(long long)));
            long long total = 0;
            EDLIB_AMD_HIP(hipMemcpyAsync(c.offs.p, totals, sizeof(long long), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            total = *reinterpret_cast<const long long*>(c.offs.p);
            if (total < (long long)n) { set_error("CIGAR: bad total"); return 1; }
            EDLIB_AMD_HIP(d_cigChars_.ensure((size_t)total));
            if (c.chars.n < (size_t)total) EDLIB_AMD_HIP(c.chars.alloc((size_t)total));
            EDLIB_AMD_HIP(launch_cigars(viewAlnDev_, viewAlnOffDev_, n_, standard, cigLen, cigRel, blockTot, totals, d_cigChars_.p, cigOff, 1, stream_));
            EDLIB_AMD_HIP(hipMemcpyAsync(c.chars.p, d_cigChars_.p, (size_t)total, hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipMemcpyAsync(c.offs.p, cigOff, (n + 1) * sizeof(long long), hipMemcpyDeviceToHost, stream_));
            EDLIB_AMD_HIP(hipStreamSynchronize(stream_));
            c.p = reinterpret_cast<const char*>(c.chars.p); c.off = reinterpret_cast<const long long*>(c.offs.p);
        } else {
            static const char ext[4] = {'=', 'I', 'D', 'X'}, stdc[4] = {'M', 'I', 'D', 'M'};
            const char* letters = format == EDLIB_CIGAR_STANDARD ? stdc : ext;
            c.hostChars.clear(); c.hostOffs.assign(n + 1, 0);
            for (size_t u = 0; u < n; ++u) {
                c.hostOffs[u] = (long long)c.hostChars.size();
                const long long a0 = v.alnOffsets[u], a1 = v.alnOffsets[u + 1];
                long long i = a0;
                while (v.alignment && i < a1) {
                    const unsigned char op = v.alignment[i];
                    if (op > 3) { set_error("CIGAR: invalid op code"); return 1; }
                    const char ch = letters[op];
                    long long run = 0;
                    while (i < a1 && v.alignment[i] <= 3 && letters[v.alignment[i]] == ch) { ++run; ++i; }
                    char buf[24];
                    const int w = snprintf(buf, sizeof buf, "%lld", run);
                    c.hostChars.insert(c.hostChars.end(), buf, buf + w);
                    c.hostChars.push_back(ch);
                }
                c.hostChars.push_back('\0');
            }
            c.hostOffs[n] = (long long)c.hostChars.size();
            c.p = c.hostChars.data(); c.off = c.hostOffs.data();
        }
        c.ready = true;
    }
    if (chars) *chars = c.p;
    if (offsets) *offsets = c.off;
    return 0;
}

// ----------------------------------------------------------------- one pair

int align_one(const char* q, int qn, const char* t, int tn, EdlibAlignConfig cfg, EdlibAlignResult* out)
{
    const long long qoff[2] = {0, qn}, toff[2] = {0, tn};
    Lap lap;
    int rc;
    {
        Batch b;
        if (b.init(q, qoff, 1, t, toff, 1, cfg, default_device())) return 1;
        lap("one: init");
        if (b.run()) return 1;
        lap("one: run");
        rc = b.results(out);
        lap("one: results");
    }
    lap("one: destroy");
    return rc;
}

}  // namespace edlib_amd
