// lanepair_kernels24.hip -- the 16- and 24-word instantiations of the lane-per-pair scan (thresholds up to 480 / 736; four /
// three waves per SIMD).
#define LANEPAIR_WINDOWS 3
#define LANEPAIR_NO_PACK 1
#include "lanepair_kernels.hpp"

namespace edlib_amd {
hipError_t launch_lanepair_scan_small(const lanepair::ScanArgs& a, int W, hipStream_t s) { return lanepair::launch_scan(a, W, s); }
}  // namespace edlib_amd
