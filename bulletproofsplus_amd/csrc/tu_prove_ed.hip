// explicit instantiation: ProveImpl<Ed25519> (its kernels are compiled in this translation unit only)
#define BPP_IMPL_DEFINITIONS 1
#include "impl_prove.hpp"
namespace bpp {
template struct ProveImpl<Ed25519>;
}
