// reads_kernels_long.hip -- the groups of 10 / 12 / 14 / 16 / 24 / 32 words of the reads-per-lane family (reads of 257..1024 bases):
// scan_reads_full_kernel (what the long HW reads handed back by the piece filter run on, whole or as chained strips) and
// scan_reads_banded_kernel (round 2's routing, EDLIB_AMD_FILTER=0; the exact end-location pass of those groups).  Its own
// translation unit: these instantiations are most of the library's compile time.
#include "reads_scan.hpp"

namespace edlib_amd {

template <int S, bool CHAIN>
static hipError_t launch_full_long_s(int nwords, const ReadScanArgs& a, hipStream_t stream)
{
    dim3 grid((a.nlanes + 63) / 64, a.numSegments), block(64);
    switch (nwords) {
        // reads of 257..512 bases: targets of up to 8 symbols (16 would need 64 KB of LDS rows per wave)
        case 10: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<10, S, CHAIN>), 10 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<10, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 12: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<12, S, CHAIN>), 12 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<12, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 14: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<14, S, CHAIN>), 14 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<14, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 16: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<16, S, CHAIN>), 16 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<16, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        // 513..1024 bases: four-symbol targets (24 / 32 KB of LDS rows per wave)
        case 24: if constexpr (S == 4) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<24, S, CHAIN>), 24 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<24, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 32: if constexpr (S == 4) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_full_kernel<32, S, CHAIN>), 32 * S * 256); hipLaunchKernelGGL((scan_reads_full_kernel<32, S, CHAIN>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_scan_reads_full_long(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream)
{
    const bool chain = a.chainIn != nullptr || a.chainOut != nullptr;
    switch (syms) {
        case 4: return chain ? launch_full_long_s<4, true>(nwords, a, stream) : launch_full_long_s<4, false>(nwords, a, stream);
        case 8: return chain ? launch_full_long_s<8, true>(nwords, a, stream) : launch_full_long_s<8, false>(nwords, a, stream);
    }
    return hipErrorInvalidValue;
}

template <int S>
static hipError_t launch_banded_long_s(int nwords, const ReadScanArgs& a, hipStream_t stream)
{
    dim3 grid((a.nlanes + 63) / 64, a.numSegments), block(64);
    switch (nwords) {
        case 10: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<10, S>), 10 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<10, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 12: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<12, S>), 12 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<12, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 14: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<14, S>), 14 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<14, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 16: if constexpr (S <= 8) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<16, S>), 16 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<16, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 24: if constexpr (S == 4) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<24, S>), 24 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<24, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        case 32: if constexpr (S == 4) { EDLIB_AMD_CHECK_STATIC_LDS((scan_reads_banded_kernel<32, S>), 32 * S * 256); hipLaunchKernelGGL((scan_reads_banded_kernel<32, S>), grid, block, 0, stream, a); break; } else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_scan_reads_banded_long(int nwords, int syms, const ReadScanArgs& a, hipStream_t stream)
{
    switch (syms) {
        case 4: return launch_banded_long_s<4>(nwords, a, stream);
        case 8: return launch_banded_long_s<8>(nwords, a, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace edlib_amd
