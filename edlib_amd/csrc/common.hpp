// common.hpp -- error plumbing and small RAII helpers shared by the host side.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <thread>
#include <vector>

namespace edlib_amd {

// Thread-local text of the last failure (edlibAmdLastError()).
std::string& last_error();
void set_error(const char* fmt, ...);

#define EDLIB_AMD_HIP(expr)                                                            \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            ::edlib_amd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                   __FILE__, __LINE__);                                \
            ::edlib_amd::pool_quarantine(true);                                        \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// Process-wide cache of device allocations (power-of-two size classes, per device).  edlibAlign()
// is a batch of one: without the cache every call pays ~30 hipMalloc/hipFree round trips.
hipError_t pool_alloc(void** p, size_t bytes, size_t* granted);
void pool_free(void* p, size_t granted);
// Pinned host staging blocks (D2H of op strings and result tables at full link rate, no page faults);
// cached like the device blocks.  Never handed to the caller.
hipError_t pinned_alloc(void** p, size_t bytes, size_t* granted);
void pinned_free(void* p, size_t granted);
// After a failed HIP call the early return drops local buffers while work on the batch's stream may still be in
// flight; until the next entry point clears the flag this thread's frees go straight to hipFree / hipHostFree
// (which wait for the device) instead of into the cache, so no other batch can be handed a block that is still in use.
void pool_quarantine(bool on);
bool pool_enabled();                              // false while this thread is quarantined
// Releases every cached device / pinned block and idle stream (edlibAmdTrim()).
void pool_trim();
hipError_t pool_stream(hipStream_t* s);
void pool_stream_release(hipStream_t s);
void pool_stream_put(int device, hipStream_t s);      // no HIP call: for destructors that may run at process exit

// Device allocation that frees itself.  Never holds host-visible result memory:
// everything handed to the caller is libc malloc() (reference ownership rules,
// edlib.h:177-205).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t granted = 0;      // bytes of the underlying pool block
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    bool owned = true;       // false: a view into another DevBuf's block (alias())
    ~DevBuf() { release(); }
    void release() { if (p && owned) pool_free(p, granted); p = nullptr; n = 0; granted = 0; owned = true; }
    // view of `count` elements inside a block somebody else owns (and outlives this view)
    void alias(void* ptr, size_t count) { release(); p = static_cast<T*>(ptr); n = count; owned = false; }
    hipError_t alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        n = count;
        owned = true;
        return pool_alloc(reinterpret_cast<void**>(&p), count * sizeof(T), &granted);
    }
    // grow-only; a view (alias()) is never "enough": its block belongs to somebody else and may be gone
    hipError_t ensure(size_t count) { return (count <= n && p && owned) ? hipSuccess : alloc(count); }
    size_t bytes() const { return n * sizeof(T); }
};

// Pinned host block that frees itself (back into the cache).
struct PinBuf {
    uint8_t* p = nullptr;
    size_t n = 0, granted = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { release(); }
    void release() { if (p) { pinned_free(p, granted); p = nullptr; n = 0; granted = 0; } }
    hipError_t alloc(size_t bytes) {
        release();
        if (bytes == 0) bytes = 1;
        n = bytes;
        return pinned_alloc(reinterpret_cast<void**>(&p), bytes, &granted);
    }
};

// Makes `device` current for the scope and restores the caller's device afterwards: a host application with
// its own HIP context (PyTorch on cuda:3 calling the drop-in binding) must not find its device switched.
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != device) status = hipSetDevice(device);
        else prev = -1;                                   // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// events are cached like streams (a batch of one pays for every hipEventCreate / hipEventDestroy it makes)
// An event belongs to the device that was current when it was created: the cache is keyed on that device, not on
// whatever device is current when the event is handed back (a Batch's members die after its DeviceGuard).
hipError_t pool_event(hipEvent_t* e, int* device);
void pool_event_release(hipEvent_t e, int device);

struct Event {
    hipEvent_t e = nullptr;
    int device = -1;
    ~Event() { if (e) pool_event_release(e, device); }
    hipError_t create() { return e ? hipSuccess : pool_event(&e, &device); }
};

// Joins the threads of a fan-out on every exit path: a std::thread that is still joinable when its vector unwinds
// (thread creation failed part-way: EAGAIN) would be std::terminate before guarded() sees the exception.
struct ThreadJoiner {
    std::vector<std::thread>& th;
    explicit ThreadJoiner(std::vector<std::thread>& t) : th(t) {}
    ~ThreadJoiner() { for (auto& x : th) if (x.joinable()) x.join(); }
};

}  // namespace edlib_amd
