// combined.hpp -- kernels of the COMBINED batch check ("final multiscalar check", SURVEY.md 8e mode B).
//
// Not a reference code path: the reference verifies one proof at a time (src/range/mod.rs:57-78) and has no
// batch API.  For a batch of proofs p = 1..B with verification MulVecs  M_p = sum_t s_{p,t} * P_{p,t}
// (each must be the identity), this mode checks the single equation   sum_p w_p * M_p == identity
// with 128-bit weights w_p that the proofs' author must not be able to predict: either supplied by the caller
// (count x 16 bytes, e.g. drawn from a transcript over the whole batch) or expanded on the device from a 256-bit
// secret key, w_p = SHA-256(key || "bppw" || global proof index)[0..16) -- a PRF whose domain is the GLOBAL index,
// so ranks sharing one key never reuse a weight:
//   * the 2mn+2 fixed generators are shared by every proof, so their terms collapse to ONE fixed-base
//     MulVec with scalars S_f = sum_p w_p * s_{p,f}               (k_comb_fixed, then k_fixed_msm, count 1)
//   * the proof-carried points form ONE variable-base MulVec of B * (3+2k+m) terms with scalars
//     w_p * s_{p,v} (k_comb_var_scalars).  It runs through the per-proof Straus kernels of the verifier
//     (k_var_digits / k_var_tables / k_var_windows): the window sums of every proof (var_wsums) are added ACROSS proofs,
//     window by window (k_comb_window_fold), and ONE Horner lane -- riding in the fixed-generator launch --
//     finishes the sum.  (A 300 k-point bucket MSM took 17 ms per 8192 proofs here; this takes ~5 ms + the Horner.)
// All-valid batches always pass; a batch with an invalid proof fails except with probability ~2^-128 PROVIDED the
// weights were unpredictable when the proofs were made (a fixed or guessable key gives no such bound: two invalid
// proofs can be built to cancel), and the caller then falls back to the per-proof path (bpp_verifier_run) for
// exact verdicts.  The check assumes proof points in the prime-order subgroup (the serialized-proof path checks
// it, container.hpp); a small-order component can vanish under an unlucky weight.
// Across GPUs each rank produces one partial -- its jacobian sum plus a validity word (set when any of its
// proofs carried an invalid point) -- they are exchanged once and summed / OR-ed (k_comb_sum_partials): the
// "single reduce over xGMI" of the north star.
#pragma once
#include "kernels.hpp"
#include "sha256.hpp"

namespace bpp {

// words of one partial: the jacobian image followed by 4 words [invalid-point flag, 0, 0, 0]
template <class C>
constexpr int partial_words() {
    return jac_words<C>() + 4;
}

// weights[p]: Montgomery form, packed 8 words.  raw != null: w_p = raw[p] (16 bytes, little-endian);
// else w_p = SHA-256(key[32] || "bppw" || (index_base + p) as u64 LE)[0..16) read little-endian.  0 -> 1.
struct WeightKey {
    uint32_t w[8];   // the 32 key bytes as little-endian words
};
template <class C>
__global__ void __launch_bounds__(256) k_comb_weights(WeightKey key, uint64_t index_base,
                                                      const uint32_t* __restrict__ raw, uint32_t* __restrict__ weights,
                                                      size_t count) {
    using P = typename C::Fr;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (raw) {
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = raw[p * 4 + i];
    } else {
        Sha256 s;
        sha256_init(s);
#pragma unroll
        for (int i = 0; i < 8; i++) sha256_word_le(s, key.w[i]);
        sha256_word_le(s, 0x77707062u);   // "bppw"
        const uint64_t idx = index_base + p;
        sha256_word_le(s, (uint32_t)idx);
        sha256_word_le(s, (uint32_t)(idx >> 32));
        uint32_t dg[8];
        sha256_final(s, dg);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t be = dg[i];
            w[i] = (be >> 24) | ((be >> 8) & 0xff00u) | ((be << 8) & 0xff0000u) | (be << 24);
        }
    }
    if ((w[0] | w[1] | w[2] | w[3]) == 0) w[0] = 1;
    Fe<P> x = fe_from_canonical<P>(w);
    uint32_t o[8];
    fe_store(x, o);
    st_words<8>(weights + p * 8, o);
}

// out[item] = canonical(w_p * s[p][var_term_index(v)]),  item = p * NV + v
template <class C>
__global__ void __launch_bounds__(256) k_comb_var_scalars(VerifyShape s, const uint32_t* __restrict__ scalars,
                                                          const uint32_t* __restrict__ weights,
                                                          uint32_t* __restrict__ out, size_t items) {
    using P = typename C::Fr;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const size_t p = item / s.NV;
    const uint32_t v = (uint32_t)(item % s.NV);
    uint32_t w[8];
    ld_words<8>(scalars + (p * s.N + var_term_index(s, v)) * 8, w);
    Fe<P> x = fe_from_canonical<P>(w);
    ld_words<8>(weights + p * 8, w);
    x = fe_mul(x, fe_load<P>(w));
    fe_to_canonical(x, w);
    st_words<8>(out + item * 8, w);
}

// one block per fixed generator f: out[fixed_term_index(f)] = canonical(sum_p w_p * s[p][fixed_term_index(f)])
template <class C>
__global__ void __launch_bounds__(256) k_comb_fixed(VerifyShape s, const uint32_t* __restrict__ scalars,
                                                    const uint32_t* __restrict__ weights, size_t count,
                                                    uint32_t* __restrict__ out) {
    using P = typename C::Fr;
    using F = Fe<P>;
    __shared__ F red[256];
    const uint32_t f = blockIdx.x, t = threadIdx.x;
    const uint32_t idx = fixed_term_index(s, f);
    F acc = F::zero();
    for (size_t p = t; p < count; p += blockDim.x) {
        uint32_t w[8];
        ld_words<8>(scalars + (p * s.N + idx) * 8, w);
        F x = fe_from_canonical<P>(w);
        ld_words<8>(weights + p * 8, w);
        acc = fe_add(acc, fe_mul(x, fe_load<P>(w)));
    }
    red[t] = acc;
    __syncthreads();
    for (uint32_t h = blockDim.x >> 1; h >= 1; h >>= 1) {
        if (t < h) red[t] = fe_add(red[t], red[t + h]);
        __syncthreads();
    }
    if (t == 0) {
        uint32_t w[8];
        fe_to_canonical(red[0], w);
        st_words<8>(out + (size_t)idx * 8, w);
    }
}

// out[g][j] = sum over the proofs p of group g (p = g * group + t, t < group, p < count) of in[p][j]
// (jacobian window sums, `per` of them per proof); one lane per (g, j)
template <class C>
__global__ void __launch_bounds__(64) k_comb_window_fold(const uint32_t* __restrict__ in, size_t count, uint32_t group,
                                                         uint32_t* __restrict__ out, size_t n_out, uint32_t per) {
    constexpr int JW = jac_words<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const size_t g = i / per;
    const uint32_t j = (uint32_t)(i % per);
    Jac<C> acc = jac_inf<C>();
    for (uint32_t t = 0; t < group; t++) {
        const size_t p = g * group + t;
        if (p >= count) break;
        acc = jac_add(acc, jac_ldg<C>(in + (p * per + j) * JW));
    }
    jac_stg<C>(out + i * JW, acc);
}

// ---- grouped check: one weighted check per group of `group` neighbouring proofs ---------------------------------
// A batch whose proofs are (nearly) all valid gets its per-proof verdicts at the combined check's price: group g's
// proofs p = g group .. are checked together, sum_p w_p M_p == identity, as ONE virtual proof of the batch verifier
// (its scalars: rows[g], below; its proof-point window sums: the group's, added by k_comb_window_fold), and only the
// proofs of a group that FAILS go through the exact per-proof path afterwards.  The verdicts are the exact path's unless a
// group of proofs, not all valid, passes its weighted check: probability ~2^-128 per group under the same conditions as
// the combined check above (unpredictable weights, proof points in the prime-order subgroup).

// rows[g][fixed_term_index(f)] = canonical(sum_{p in group g} w_p * s[p][fixed_term_index(f)]): one lane per (g, f), f
// fastest, so that a wave reads 64 neighbouring scalars of one proof at a time and the weight is wave-uniform
template <class C>
__global__ void __launch_bounds__(64) k_comb_fixed_grouped(VerifyShape s, const uint32_t* __restrict__ scalars,
                                                           const uint32_t* __restrict__ weights, size_t count,
                                                           uint32_t group, uint32_t* __restrict__ rows) {
    using P = typename C::Fr;
    using F = Fe<P>;
    const uint32_t bpg = (s.NF + 63) / 64;
    const size_t g = blockIdx.x / bpg;
    const uint32_t f = (blockIdx.x % bpg) * 64 + threadIdx.x;
    if (f >= s.NF) return;
    const uint32_t idx = fixed_term_index(s, f);
    F acc = F::zero();
    for (uint32_t t = 0; t < group; t++) {
        const size_t p = g * group + t;
        if (p >= count) break;
        uint32_t w[8];
        ld_words<8>(scalars + (p * s.N + idx) * 8, w);
        const F x = fe_from_canonical<P>(w);
        ld_words<8>(weights + p * 8, w);
        acc = fe_add(acc, fe_mul(x, fe_load<P>(w)));
    }
    uint32_t w[8];
    fe_to_canonical(acc, w);
    st_words<8>(rows + (g * s.N + idx) * 8, w);
}

// gbad[g] = any proof of group g carried an invalid point
static __global__ void __launch_bounds__(256) k_comb_group_bad(const uint32_t* __restrict__ bad, size_t count, uint32_t group,
                                                        uint32_t* __restrict__ gbad, size_t groups) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    uint32_t any = 0;
    for (uint32_t t = 0; t < group; t++) {
        const size_t p = g * group + t;
        if (p < count) any |= bad[p];
    }
    gbad[g] = any ? 1u : 0u;
}

// verdicts[p] = its group's verdict (0 = every proof of the group is valid; 1 = stands until the exact pass overwrites it)
static __global__ void __launch_bounds__(256) k_comb_group_spread(const uint32_t* __restrict__ gok, uint32_t group,
                                                           uint32_t* __restrict__ verdicts, size_t count) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < count) verdicts[p] = gok[p / group] ? 1u : 0u;
}

// out[i][0..row) = in[list[i]][0..row): one block per row
static __global__ void k_comb_gather_rows(const uint32_t* __restrict__ in, const uint32_t* __restrict__ list, uint32_t row,
                                   uint32_t* __restrict__ out) {
    const uint32_t* src = in + (size_t)list[blockIdx.x] * row;
    uint32_t* dst = out + (size_t)blockIdx.x * row;
    for (uint32_t t = threadIdx.x; t < row; t += blockDim.x) dst[t] = src[t];
}

// out[list[i]] = in[i]
static __global__ void __launch_bounds__(256) k_comb_scatter_words(const uint32_t* __restrict__ in, const uint32_t* __restrict__ list,
                                                            uint32_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[list[i]] = in[i];
}

// verdict of one rank's share: ok = partial is the identity and no proof carried an invalid point
template <class C>
__global__ void __launch_bounds__(256) k_comb_verdict(const uint32_t* __restrict__ partial,
                                                      const uint32_t* __restrict__ bad, size_t count,
                                                      uint32_t* __restrict__ ok) {
    int anybad = 0;
    for (size_t p = threadIdx.x; p < count; p += blockDim.x) anybad |= (bad[p] != 0);
    anybad = __syncthreads_or(anybad);
    if (threadIdx.x == 0) {
        Jac<C> acc = jac_ldg<C>(partial + threadIdx.x);
        ok[0] = (jac_is_identity_class(acc) && !anybad) ? 0u : 1u;
        uint32_t* flag = const_cast<uint32_t*>(partial) + jac_words<C>();   // travels with the partial to the other ranks
        flag[0] = anybad ? 1u : 0u;
        flag[1] = flag[2] = flag[3] = 0u;
    }
}

// sum of n partials, `stride` words apart -> verdict (and the sum itself, optional).  with_flags: every partial is
// followed by its validity word (the cross-rank form); the verdict then also requires every flag to be clear.
template <class C>
__global__ void __launch_bounds__(64) k_comb_sum_partials(const uint32_t* __restrict__ partials, uint32_t n, uint32_t stride,
                                                          uint32_t with_flags, uint32_t* __restrict__ ok,
                                                          uint32_t* __restrict__ out) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t lane_zero;  // keeps the wave-uniform chain on the vector unit (a uniform address would move the arithmetic to the scalar unit)
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    partials += lane_zero;
    Jac<C> acc = jac_inf<C>();
    uint32_t anybad = 0;
    for (uint32_t t = 0; t < n; t++) {
        acc = jac_add(acc, jac_ldg<C>(partials + (size_t)t * stride));
        if (with_flags) anybad |= partials[(size_t)t * stride + JW];
    }
    ok[0] = (jac_is_identity_class(acc) && !anybad) ? 0u : 1u;
    if (out) jac_stg<C>(out, acc);
}

}  // namespace bpp
