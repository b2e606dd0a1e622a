// explicit instantiation: VerifyImpl<Ed25519> (its kernels are compiled in this translation unit only)
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Ed25519>;
}
