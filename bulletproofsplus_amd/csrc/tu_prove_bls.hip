// explicit instantiation: ProveImpl<Bls12381> (its kernels are compiled in this translation unit only)
#include "impl_prove.hpp"
namespace bpp {
template struct ProveImpl<Bls12381>;
}
