// explicit instantiation: VerifyImpl<Bls12381> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Bls12381>;
}
