// explicit instantiation: MsmImpl<Ed25519> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "codec.hpp"
#include "impl_msm.hpp"
namespace bpp {
template struct MsmImpl<Ed25519>;
template struct CodecImpl<Ed25519>;
}
