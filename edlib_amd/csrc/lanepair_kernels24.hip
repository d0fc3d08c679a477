// lanepair_kernels24.hip -- the 24-word instantiation of the lane-per-pair scan (thresholds up to 736; 4 waves per SIMD):
// its own translation unit so that it builds beside the 48-word one.
#define LANEPAIR_NO_W48 1
#define LANEPAIR_NO_PACK 1
#include "lanepair_kernels.hpp"

namespace edlib_amd {
hipError_t launch_lanepair_scan24(const lanepair::ScanArgs& a, hipStream_t s) { return lanepair::launch_scan(a, 24, s); }
}  // namespace edlib_amd
