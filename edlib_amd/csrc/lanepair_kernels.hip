// lanepair_kernels.hip -- the lane-per-pair NW distance scan of big pair batches: the pack kernel and the
// 48-word scan (thresholds up to 1504).  The other windows build in lanepair_kernels42.hip and lanepair_kernels24.hip: a scan
// kernel is a ladder of one unrolled loop per window height, and each is best allocated on its own (lanepair.hpp).
#define LANEPAIR_WINDOWS 8
#include "lanepair_kernels.hpp"

namespace edlib_amd {

hipError_t launch_lanepair_scan42(const lanepair::ScanArgs& a, hipStream_t s);
hipError_t launch_lanepair_scan_small(const lanepair::ScanArgs& a, int W, hipStream_t s);

hipError_t launch_lanepair_pack(const lanepair::PackArgs& a, hipStream_t s) { return lanepair::launch_pack(a, s); }

hipError_t launch_lanepair_scan(const lanepair::ScanArgs& a, int W, hipStream_t s)
{
    if (W == 16 || W == 24) return launch_lanepair_scan_small(a, W, s);
    if (W == 42) return launch_lanepair_scan42(a, s);
    return lanepair::launch_scan(a, W, s);
}

}  // namespace edlib_amd
