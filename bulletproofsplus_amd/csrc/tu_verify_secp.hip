// explicit instantiation: VerifyImpl<Secp256k1> (its kernels are compiled in this translation unit only)
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Secp256k1>;
}
