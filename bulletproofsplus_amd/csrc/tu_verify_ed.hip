// explicit instantiation: VerifyImpl<Ed25519> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "impl_verify.hpp"
namespace bpp {
template struct VerifyImpl<Ed25519>;
}
