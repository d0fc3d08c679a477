// lds_check.hpp -- kernels that address LDS from 0 (M0 row offsets of the banded reads kernels, the 512-byte aligned target
// rings of the lane rings, the Peq slice of the wide kernel) rely on their LDS object being the first and only STATIC one.
// That is a property of the compiled kernel, so it is checked on the HOST, once per kernel at its first launch, from the
// static LDS bytes the code object declares: a violation is hipErrorInvalidValue at the launch (-> EDLIB_STATUS_ERROR) and
// never a device trap (a trap is a queue exception that aborts the host process; SURVEY.md 5: errors must be statuses).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

// (a runtime that reports the group segment rounded up to its allocation granule still passes: what must not pass is a
// second static object, which no rounding of the expected size explains)
inline bool lds_static_size_ok(size_t reported, size_t expect)
{
    if (reported == expect) return true;
    for (size_t g = 16; g <= 1024; g *= 2)
        if (reported == (expect + g - 1) / g * g) return true;
    return false;
}

#define EDLIB_AMD_CHECK_STATIC_LDS(kernel, expectBytes)                                                              \
    do {                                                                                                             \
        static std::atomic<int> ok_{-1};             /* (launches come from several host threads) */                 \
        int v_ = ok_.load(std::memory_order_relaxed);                                                                \
        if (v_ < 0) {                                                                                                \
            hipFuncAttributes at_;                                                                                   \
            v_ = (hipFuncGetAttributes(&at_, reinterpret_cast<const void*>(kernel)) == hipSuccess &&                \
                  lds_static_size_ok(at_.sharedSizeBytes, (size_t)(expectBytes))) ? 1 : 0;                           \
            ok_.store(v_, std::memory_order_relaxed);                                                                \
        }                                                                                                            \
        if (!v_) return hipErrorInvalidValue;                                                                        \
    } while (0)
