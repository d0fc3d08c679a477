// lanepair_kernels42.hip -- the 42-word instantiation of the lane-per-pair scan (thresholds up to 1312: BASELINE config 4).
#define LANEPAIR_WINDOWS 4
#define LANEPAIR_NO_PACK 1
#include "lanepair_kernels.hpp"

namespace edlib_amd {
hipError_t launch_lanepair_scan42(const lanepair::ScanArgs& a, hipStream_t s) { return lanepair::launch_scan(a, 42, s); }
}  // namespace edlib_amd
