// explicit instantiation: the k_fixed_msm launches named below are compiled in this translation unit only
// (fixed_launch.hpp)
#define BPP_FIXED_LAUNCH_DEFINITIONS 1
#include "fixed_launch.hpp"
namespace bpp {
BPP_FIXED_LAUNCH_INSTANTIATE(Bls12381, 1)
}
