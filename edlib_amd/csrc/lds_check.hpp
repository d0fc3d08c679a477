// lds_check.hpp -- kernels that address LDS from 0 (M0 row offsets of the banded reads kernels, the 512-byte aligned target
// rings of the lane rings, the Peq slice of the wide kernel) rely on their LDS object being the first and only STATIC one.
// That is a property of the compiled kernel, so it is checked on the HOST, once per kernel at its first launch, from the
// static LDS bytes the code object declares: a violation is hipErrorInvalidValue at the launch (-> EDLIB_STATUS_ERROR) and
// never a device trap (a trap is a queue exception that aborts the host process; SURVEY.md 5: errors must be statuses).
#pragma once
#include <hip/hip_runtime.h>

#define EDLIB_AMD_CHECK_STATIC_LDS(kernel, expectBytes)                                                              \
    do {                                                                                                             \
        static int ok_ = -1;                                                                                         \
        if (ok_ < 0) {                                                                                               \
            hipFuncAttributes at_;                                                                                   \
            ok_ = (hipFuncGetAttributes(&at_, reinterpret_cast<const void*>(kernel)) == hipSuccess &&               \
                   at_.sharedSizeBytes == (size_t)(expectBytes)) ? 1 : 0;                                            \
        }                                                                                                            \
        if (!ok_) return hipErrorInvalidValue;                                                                       \
    } while (0)
