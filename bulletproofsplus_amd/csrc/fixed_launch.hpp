// fixed_launch.hpp -- the launch of k_fixed_msm (kernels.hpp) behind a function template that is DEFINED in its own
// translation units (tu_fixed_*.hip): the kernel is the largest piece of device code of the library (~90 000
// instructions per instantiation, three roles per curve), and compiled inside tu_verify_*.hip it alone decided the
// wall time of a build.  Everybody else sees the declaration and the `extern template`s below.
#pragma once
#include <algorithm>

#include "kernels.hpp"

namespace bpp {

// dynamic LDS of a k_fixed_msm launch: the gather ring, reused by the block reduction of ROLE 1
template <class C>
constexpr unsigned fixed_lds() {
    return std::max<unsigned>(fixed_lds_bytes<C>(), FIXED_BLOCK * jac_words<C>() * 4);
}

// grid_blocks blocks of FIXED_BLOCK threads on `st`; arguments as k_fixed_msm's
template <class C, int ROLE>
void launch_fixed_msm(unsigned grid_blocks, hipStream_t st, VerifyShape s, const uint32_t* scalars, const uint32_t* table,
                      uint32_t* partials, uint32_t per, uint32_t horner_blocks, const uint32_t* wsum, uint32_t* var_out,
                      size_t horner_count, uint32_t horner_tree, VpSel sel);

#ifdef BPP_FIXED_LAUNCH_DEFINITIONS
template <class C, int ROLE>
void launch_fixed_msm(unsigned grid_blocks, hipStream_t st, VerifyShape s, const uint32_t* scalars, const uint32_t* table,
                      uint32_t* partials, uint32_t per, uint32_t horner_blocks, const uint32_t* wsum, uint32_t* var_out,
                      size_t horner_count, uint32_t horner_tree, VpSel sel) {
    hipLaunchKernelGGL((k_fixed_msm<C, ROLE>), dim3(grid_blocks), dim3(FIXED_BLOCK), fixed_lds<C>(), st, s, scalars, table,
                       partials, per, horner_blocks, wsum, var_out, horner_count, horner_tree, sel);
}
#endif

#define BPP_FIXED_LAUNCH_EXTERN(C, ROLE)                                                                                     \
    extern template void launch_fixed_msm<C, ROLE>(unsigned, hipStream_t, VerifyShape, const uint32_t*, const uint32_t*,      \
                                                   uint32_t*, uint32_t, uint32_t, const uint32_t*, uint32_t*, size_t,         \
                                                   uint32_t, VpSel);
#define BPP_FIXED_LAUNCH_INSTANTIATE(C, ROLE)                                                                                \
    template void launch_fixed_msm<C, ROLE>(unsigned, hipStream_t, VerifyShape, const uint32_t*, const uint32_t*, uint32_t*,   \
                                            uint32_t, uint32_t, const uint32_t*, uint32_t*, size_t, uint32_t, VpSel);
#ifndef BPP_FIXED_LAUNCH_DEFINITIONS
BPP_FIXED_LAUNCH_EXTERN(Bls12381, 0)
BPP_FIXED_LAUNCH_EXTERN(Bls12381, 1)
BPP_FIXED_LAUNCH_EXTERN(Bls12381, 2)
BPP_FIXED_LAUNCH_EXTERN(Secp256k1, 0)
BPP_FIXED_LAUNCH_EXTERN(Secp256k1, 1)
BPP_FIXED_LAUNCH_EXTERN(Secp256k1, 2)
BPP_FIXED_LAUNCH_EXTERN(Ed25519, 0)
BPP_FIXED_LAUNCH_EXTERN(Ed25519, 1)
BPP_FIXED_LAUNCH_EXTERN(Ed25519, 2)
#endif

}  // namespace bpp
