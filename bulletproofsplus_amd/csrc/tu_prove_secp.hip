// explicit instantiation: ProveImpl<Secp256k1> (its kernels are compiled in this translation unit only)
#include "impl_prove.hpp"
namespace bpp {
template struct ProveImpl<Secp256k1>;
}
