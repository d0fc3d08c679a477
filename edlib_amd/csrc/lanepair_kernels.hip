// lanepair_kernels.hip -- the lane-per-pair NW distance scan of big pair batches: the pack kernel and the
// 48-word scan (thresholds up to 1504).  The 24-word scan builds in lanepair_kernels24.hip: a scan kernel is a ladder of
// one unrolled loop per window height, minutes of compile time each.
#define LANEPAIR_NO_W24 1
#include "lanepair_kernels.hpp"

namespace edlib_amd {

hipError_t launch_lanepair_scan24(const lanepair::ScanArgs& a, hipStream_t s);

hipError_t launch_lanepair_pack(const lanepair::PackArgs& a, hipStream_t s) { return lanepair::launch_pack(a, s); }

hipError_t launch_lanepair_scan(const lanepair::ScanArgs& a, int W, hipStream_t s)
{
    if (W == 24) return launch_lanepair_scan24(a, s);
    return lanepair::launch_scan(a, W, s);
}

}  // namespace edlib_amd
