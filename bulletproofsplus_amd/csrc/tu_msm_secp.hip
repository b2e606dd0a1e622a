// explicit instantiation: MsmImpl<Secp256k1> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "codec.hpp"
#include "impl_msm.hpp"
namespace bpp {
template struct MsmImpl<Secp256k1>;
template struct CodecImpl<Secp256k1>;
}
