// explicit instantiation: MsmImpl<Bls12381> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "codec.hpp"
#include "impl_msm.hpp"
namespace bpp {
template struct MsmImpl<Bls12381>;
template struct CodecImpl<Bls12381>;
}
