// explicit instantiation: VerifyImpl<Secp256k1> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Secp256k1>;
}
