// explicit instantiation: ProveImpl<Ed25519> (its kernels are compiled in this translation unit only)
#include "impl_prove.hpp"
namespace bpp {
template struct ProveImpl<Ed25519>;
}
