// batch_affine.hpp -- the fixed-generator part of the verification MulVec with the first level of additions
// done in AFFINE coordinates under one shared inversion per thread (Montgomery's trick).
//
// k_fixed_msm (kernels.hpp) adds every gathered table entry to an XYZZ accumulator: 8M + 2S per entry.  Here the
// entries of two neighbouring windows of the same generator are first added to each other in affine
// coordinates -- lambda = (y1 - y0) / (x1 - x0), 2M + 1S once the inverse is known -- and only the sum goes into
// the accumulator.  The inverses of one thread's whole chain of pairs cost 3M per pair plus ONE field
// inversion per thread (fe_inv: safegcd, ~40 M), so two entries cost ~(6.4 + 10) M instead of 20 M.
//
//   k_ba_forward   walks the thread's pairs in order, multiplies the denominators into a running product and
//                  streams the prefix products to HBM ([pair][16-byte chunk][thread]: coalesced); ends with the
//                  inversion of the total.  Only x coordinates are gathered (the y's only when x0 == x1).
//   k_ba_backward  walks the same pairs in reverse: inverse of pair i = running inverse * prefix[i-1], affine
//                  sum -> HBM, same coalesced layout.
//   k_ba_accumulate  mixed addition of the thread's sums into an XYZZ accumulator, block reduction -> partials
//                  (the layout k_fixed_msm produces, so k_finalize is shared).
// Three kernels rather than one: each inner loop stays within the 64 KB instruction cache (a fused
// affine-add + XYZZ-add body is ~96 KB of code) and within 256 VGPRs without scratch.
//
// Both kernels see every case the reference's point addition distinguishes (macros.rs:42-146): a zero digit
// (no entry), equal entries (-> tangent, denominator 2y), opposite entries (-> nothing to add).  They are not
// rare here: the reference's generators are small multiples of one point (publickey.rs:31,38), so entries of
// different windows coincide regularly.  The denominator of such a pair is 1 (or 2y), never 0, so the running
// product stays invertible.
//
// HBM traffic per pair: 2 x-gathers + 2 full gathers (4 x 96 B lines touched) + 48 B prefix written and read:
// ~5x the bytes of k_fixed_msm, on a path that used 5 % of the HBM bandwidth -- bytes bought for ALU work.
#pragma once
#include "kernels.hpp"

namespace bpp {

template <class C>
struct has_batch_affine {
    static constexpr bool value = true;
};
template <>
struct has_batch_affine<Ed25519> {   // extended Edwards coordinates: the unified addition is already 9M, kept as is
    static constexpr bool value = false;
};

#ifndef BPP_BA_FWD_WAVES
#define BPP_BA_FWD_WAVES 4
#endif

// canonical scalar of generator f + the signed-digit bias (VerifyShape::bias), 10 words
__device__ __forceinline__ void ba_load_scalar(const VerifyShape& s, const uint32_t* __restrict__ sc, uint32_t f,
                                               uint32_t* w) {
    ld_words<8>(sc + (size_t)fixed_term_index(s, f) * 8, w);
    w[8] = 0;
    w[9] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int t = 0; t < 10; t++) {
        uint64_t x = (uint64_t)w[t] + s.bias[t] + carry;
        w[t] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
}
// lowest window -> signed digit; the value moves down by one window
__device__ __forceinline__ int32_t ba_take_low(const VerifyShape& s, uint32_t* w) {
    const int32_t dg = (int32_t)(w[0] & ((1u << s.c) - 1u)) - (int32_t)s.half;
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = (w[t] >> s.c) | (w[t + 1] << (32 - s.c));
    w[9] >>= s.c;
    return dg;
}
// moves window W-1 to the top bits of the 320-bit value, so that ba_take_top can read windows downwards
__device__ __forceinline__ void ba_align_top(const VerifyShape& s, uint32_t* w) {
    const uint32_t amt = 320u - s.c * s.W;   // c W >= 258, so amt < 64
    if (amt >= 32) {
#pragma unroll
        for (int t = 9; t >= 1; t--) w[t] = w[t - 1];
        w[0] = 0;
    }
    const uint32_t bs = amt & 31u;
    if (bs) {
#pragma unroll
        for (int t = 9; t >= 1; t--) w[t] = (w[t] << bs) | (w[t - 1] >> (32 - bs));
        w[0] <<= bs;
    }
}
__device__ __forceinline__ int32_t ba_take_top(const VerifyShape& s, uint32_t* w) {
    const int32_t dg = (int32_t)(w[9] >> (32 - s.c)) - (int32_t)s.half;
#pragma unroll
    for (int t = 9; t >= 1; t--) w[t] = (w[t] << s.c) | (w[t - 1] >> (32 - s.c));
    w[0] <<= s.c;
    return dg;
}
// table entry of digit dg != 0 of window j of generator f (the caller applies the sign)
template <class C>
__device__ __forceinline__ const uint32_t* ba_entry(const VerifyShape& s, const uint32_t* __restrict__ table, uint32_t f,
                                                    uint32_t j, int32_t dg) {
    const uint32_t mag = dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg;
#ifdef BPP_DBG_LOCAL_GATHER   // timing experiment only: every gather lands in a 48 MB window (wrong results)
    return table + (((size_t)(f & 15u) * s.W + j) * s.half + (mag - 1)) * 2 * C::Fp::N;
#endif
    return table + (((size_t)f * s.W + j) * s.half + (mag - 1)) * 2 * C::Fp::N;
}
// element `slot` of thread `gtid` in a [slot][chunk][thread] array of packed field elements
template <class P>
__device__ __forceinline__ void ba_put(uint32_t* __restrict__ base, size_t slot, size_t nthreads, size_t gtid,
                                       const Fe<P>& v) {
    constexpr int N = P::N;
    uint32_t w[N];
    fe_store(v, w);
    uint4* q = reinterpret_cast<uint4*>(base);
#pragma unroll
    for (int ch = 0; ch < N / 4; ch++)
        q[(slot * (N / 4) + ch) * nthreads + gtid] = make_uint4(w[4 * ch], w[4 * ch + 1], w[4 * ch + 2], w[4 * ch + 3]);
}
template <int N>
__device__ __forceinline__ void ba_get_raw(const uint32_t* __restrict__ base, size_t slot, size_t nthreads, size_t gtid,
                                           uint32_t* w) {
    const uint4* q = reinterpret_cast<const uint4*>(base);
#pragma unroll
    for (int ch = 0; ch < N / 4; ch++) {
        const uint4 v = q[(slot * (N / 4) + ch) * nthreads + gtid];
        w[4 * ch] = v.x;
        w[4 * ch + 1] = v.y;
        w[4 * ch + 2] = v.z;
        w[4 * ch + 3] = v.w;
    }
}

// Denominator of the affine addition P0 + P1 of two table entries (signs applied to the y's):
//   x0 != x1 -> x1 - x0 ;  P0 == P1 -> 2 y0 ;  P0 == -P1 -> 1 (no sum).  Never zero.
// ys are only fetched when the x's agree.
template <class C>
__device__ __forceinline__ Fe<typename C::Fp> ba_denominator(const Fe<typename C::Fp>& x0, const Fe<typename C::Fp>& x1,
                                                             const uint32_t* __restrict__ e0, bool n0,
                                                             const uint32_t* __restrict__ e1, bool n1) {
    using F = Fe<typename C::Fp>;
    constexpr int N = C::Fp::N;
    const F dx = fe_sub(x1, x0);
    if (!dx.is_zero()) return dx;
    uint32_t r[N];
    ld_words<N>(e0 + N, r);
    F y0 = fe_load<typename C::Fp>(r);
    ld_words<N>(e1 + N, r);
    F y1 = fe_load<typename C::Fp>(r);
    if (n0) y0 = fe_neg(y0);
    if (n1) y1 = fe_neg(y1);
    if (y0 == y1) return fe_dbl(y0);
    return F::one();
}

// pairs per generator: windows (0,1), (2,3), ...; with an odd W the top window is a pair without a partner
__device__ __forceinline__ uint32_t ba_pairs_per_generator(const VerifyShape& s) { return (s.W + 1) >> 1; }

template <class C>
__global__ void __launch_bounds__(FIXED_BLOCK, BPP_BA_FWD_WAVES)
    k_ba_forward(VerifyShape s, const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ table,
                 uint32_t* __restrict__ prefix, uint32_t* __restrict__ inv_out, uint32_t per) {
    using P = typename C::Fp;
    using F = Fe<P>;
    constexpr int N = P::N;
    const size_t b = blockIdx.x / per;
    const uint32_t part = blockIdx.x % per;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const uint32_t* sc = scalars + b * (size_t)s.N * 8;
    const uint32_t PQ = ba_pairs_per_generator(s);
    const uint32_t stride = per * blockDim.x;
    uint32_t f = part * blockDim.x + threadIdx.x;
    uint32_t q = PQ;   // forces the first scalar load
    uint32_t w[10];
    // one pair in flight: its x gathers are issued before the previous pair's product is computed
    bool have = false;
    const uint32_t *e0 = nullptr, *e1 = nullptr;
    bool n0 = false, n1 = false;
    uint32_t rx0[N], rx1[N];
    auto fetch = [&]() __attribute__((always_inline)) {
        have = false;
        if (q == PQ) {
            if (f >= s.NF) return;
            ba_load_scalar(s, sc, f, w);
            q = 0;
        }
        const int32_t d0 = ba_take_low(s, w);
        const int32_t d1 = 2 * q + 1 < s.W ? ba_take_low(s, w) : 0;
        e0 = d0 ? ba_entry<C>(s, table, f, 2 * q, d0) : nullptr;
        e1 = d1 ? ba_entry<C>(s, table, f, 2 * q + 1, d1) : nullptr;
        n0 = d0 < 0;
        n1 = d1 < 0;
        if (e0 && e1) {
            ld_words<N>(e0, rx0);
            ld_words<N>(e1, rx1);
        }
        if (++q == PQ) f += stride;
        have = true;
    };
    F run = F::one();
    size_t slot = 0;
    fetch();
    while (have) {
        const uint32_t *c0 = e0, *c1 = e1;
        const bool cn0 = n0, cn1 = n1;
        F x0, x1;
        if (c0 && c1) {
            x0 = fe_load<P>(rx0);
            x1 = fe_load<P>(rx1);
        }
        fetch();
        if (c0 && c1) run = fe_mul(run, ba_denominator<C>(x0, x1, c0, cn0, c1, cn1));
#ifndef BPP_DBG_NO_STORE
        ba_put<P>(prefix, slot, nthreads, gtid, run);
#endif
        slot++;
    }
    ba_put<P>(inv_out, 0, nthreads, gtid, fe_inv(run));
}

// reverse walk: one affine sum per pair -> sums[slot] ([slot][chunk][thread], x | y; x = y = 0: nothing to add)
template <class C>
__global__ void __launch_bounds__(FIXED_BLOCK, BPP_FIXED_WAVES)
    k_ba_backward(VerifyShape s, const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ table,
                  const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ inv_in,
                  uint32_t* __restrict__ sums, uint32_t per) {
    using P = typename C::Fp;
    using F = Fe<P>;
    constexpr int N = P::N;
    const size_t b = blockIdx.x / per;
    const uint32_t part = blockIdx.x % per;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const uint32_t* sc = scalars + b * (size_t)s.N * 8;
    const uint32_t PQ = ba_pairs_per_generator(s);
    const uint32_t stride = per * blockDim.x;
    const uint32_t f0 = part * blockDim.x + threadIdx.x;
    const uint32_t G = f0 < s.NF ? (s.NF - 1 - f0) / stride + 1 : 0;   // generators of this thread
    F inv;
    {
        uint32_t r[N];
        ba_get_raw<N>(inv_in, 0, nthreads, gtid, r);
        inv = fe_load<P>(r);
    }
    int32_t g = (int32_t)G - 1;   // generator iteration being consumed, downwards
    int32_t q = -1;               // pair within the generator, downwards; -1 forces the next scalar load
    size_t slot = (size_t)G * PQ;  // pairs not yet issued
    uint32_t f = 0;
    uint32_t w[10];
    bool have = false;
    const uint32_t *e0 = nullptr, *e1 = nullptr;
    bool n0 = false, n1 = false, first = false;
    uint32_t r0[2 * N], r1[2 * N], rp[N];
    auto fetch = [&]() __attribute__((always_inline)) {
        have = false;
        if (q < 0) {
            if (g < 0) return;
            f = f0 + (uint32_t)g * stride;
            ba_load_scalar(s, sc, f, w);
            ba_align_top(s, w);
            q = (int32_t)PQ - 1;
            g--;
        }
        const int32_t d1 = 2 * (uint32_t)q + 1 < s.W ? ba_take_top(s, w) : 0;
        const int32_t d0 = ba_take_top(s, w);
        e0 = d0 ? ba_entry<C>(s, table, f, 2 * (uint32_t)q, d0) : nullptr;
        e1 = d1 ? ba_entry<C>(s, table, f, 2 * (uint32_t)q + 1, d1) : nullptr;
        n0 = d0 < 0;
        n1 = d1 < 0;
        if (e0) ld_words<2 * N>(e0, r0);
        if (e1) ld_words<2 * N>(e1, r1);
        slot--;
        first = slot == 0;
        if (!first && e0 && e1) ba_get_raw<N>(prefix, slot - 1, nthreads, gtid, rp);
        q--;
        have = true;
    };
    fetch();
    while (have) {
        const uint32_t *c0 = e0, *c1 = e1;
        const bool cfirst = first;
        const size_t cslot = slot;
        Aff<C> p0, p1;
        F pm;
        if (c0) {
            p0 = aff_load<C>(r0);
            if (n0) p0 = aff_neg(p0);
        }
        if (c1) {
            p1 = aff_load<C>(r1);
            if (n1) p1 = aff_neg(p1);
        }
        if (c0 && c1) pm = cfirst ? F::one() : fe_load<P>(rp);
        fetch();
        Aff<C> sum = aff_inf<C>();
        if (c0 && c1) {
            F den = fe_sub(p1.x, p0.x), num;
            bool ok = true;
            if (!den.is_zero()) {
                num = fe_sub(p1.y, p0.y);
            } else if (p0.y == p1.y) {
                den = fe_dbl(p0.y);
                const F xx = fe_sqr(p0.x);
                num = fe_add(fe_dbl(xx), xx);
            } else {
                ok = false;   // opposite entries: denominator 1 in the chain, nothing to add
            }
            if (ok) {
                const F inv_i = fe_mul(inv, pm);
                inv = fe_mul(inv, den);
                const F lam = fe_mul(num, inv_i);
                sum.x = fe_sub(fe_sub(fe_sqr(lam), p0.x), p1.x);
                sum.y = fe_sub(fe_mul(lam, fe_sub(p0.x, sum.x)), p0.y);
            }
        } else if (c0) {
            sum = p0;
        } else if (c1) {
            sum = p1;
        }
        ba_put<P>(sums, 2 * cslot, nthreads, gtid, sum.x);
        ba_put<P>(sums, 2 * cslot + 1, nthreads, gtid, sum.y);
    }
}

// XYZZ accumulation of a thread's affine sums (coalesced reads), block reduction -> partials: the layout
// k_fixed_msm produces, so k_finalize is shared.
template <class C>
__global__ void __launch_bounds__(FIXED_BLOCK, BPP_FIXED_WAVES)
    k_ba_accumulate(VerifyShape s, const uint32_t* __restrict__ sums, uint32_t* __restrict__ partials, uint32_t per) {
    using P = typename C::Fp;
    constexpr int N = P::N;
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t part = blockIdx.x % per;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const uint32_t stride = per * blockDim.x;
    const uint32_t f0 = part * blockDim.x + threadIdx.x;
    const uint32_t G = f0 < s.NF ? (s.NF - 1 - f0) / stride + 1 : 0;
    const size_t n = (size_t)G * ba_pairs_per_generator(s);
    Xyzz<C> acc = xyzz_inf<C>();
    uint32_t raw[2 * N];
    if (n) {
        ba_get_raw<N>(sums, 0, nthreads, gtid, raw);
        ba_get_raw<N>(sums, 1, nthreads, gtid, raw + N);
    }
    for (size_t i = 0; i < n; i++) {
        const Aff<C> cur = aff_load<C>(raw);
        if (i + 1 < n) {
            ba_get_raw<N>(sums, 2 * (i + 1), nthreads, gtid, raw);
            ba_get_raw<N>(sums, 2 * (i + 1) + 1, nthreads, gtid, raw + N);
        }
        acc = xyzz_madd(acc, cur);
    }
    Jac<C> tot = block_reduce_jac<C>(xyzz_to_jac(acc), lds);
    if (threadIdx.x == 0) jac_stg<C>(partials + (size_t)blockIdx.x * JW, tot);
}

}  // namespace bpp
