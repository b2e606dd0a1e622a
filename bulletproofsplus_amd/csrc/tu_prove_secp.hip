// explicit instantiation: ProveImpl<Secp256k1> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "impl_prove.hpp"
namespace bpp {
template struct ProveImpl<Secp256k1>;
}
