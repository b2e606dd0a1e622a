// explicit instantiation: VerifyImpl<Bls12381> (its kernels are compiled in this translation unit only)
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Bls12381>;
}
