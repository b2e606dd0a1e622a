// field.hpp -- Montgomery prime-field arithmetic for gfx950 (and the host side of the same library).
//
// One template serves every field of the engine (constants_gen.h): BLS12-381 Fp / Fr and
// secp256k1 Fp / Fr.  On the hot path it replaces what the reference obtains from mcl's Fp/Fr
// (reference src/bls12_381/building_block/scalar/prime_field_elem.rs:51-248 -> mcl_rust) and from
// num-bigint (reference src/secp256k1/building_block/field/prime_field_elem.rs:251-281, :339-392).
//
// Representation (chosen for CDNA4, not translated from anything):
//   * in REGISTERS an element is NL unsaturated 30-bit limbs (13 for the 381-bit field, 9 for the
//     255/256-bit fields), normalised (every limb < 2^30), in Montgomery form with R = 2^(30*NL), and
//     LAZILY reduced: the value lies in [0, 2p), not [0, p).  With 4p < R a Montgomery product of two such
//     values is < p (4p/R + 1) < 1.01 p, so fe_mul / fe_sqr need no final conditional subtraction; add and
//     sub keep the invariant with one conditional +-2p.  Zero tests accept 0 and p; equality, canonical
//     output and the packed memory image reduce fully.
//   * in MEMORY (HBM tables, proofs, wire) it is N packed 32-bit words (12 / 8) of the fully reduced value.
// Why 30-bit limbs: the only wide multiplier on gfx950 is v_mad_u64_u32 (32x32+64 -> 64).  It has a
// carry-OUT but no carry-IN, and on gfx90a+/gfx950 a VALU carry written to VCC/SGPR needs two wait
// states before a VALU instruction may read it, so a saturated 32-bit-limb schedule costs >= 3 issue
// slots per limb product.  With 30-bit limbs a 64-bit column accumulator absorbs all NL <= 13
// products of a column (13 * 2^60 < 2^64) with no carry handling at all: one v_mad_u64_u32 per limb
// product, plus one shift + one mask per column.  No MFMA: there is no carry chain in the matrix pipe.
#pragma once
#include <stdint.h>
#include <type_traits>

#include "constants_gen.h"

#if defined(__HIPCC__)
#define BPP_HD __host__ __device__ __forceinline__
// cold, bulky helpers (the hash): a real call on the device keeps kernels that use them many times at a sane size
#define BPP_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define BPP_HD inline
#define BPP_HD_NOINLINE inline
#endif

namespace bpp {

constexpr int LIMB_BITS = 30;
constexpr uint32_t LIMB_MASK = (1u << LIMB_BITS) - 1u;

template <class P>
struct Fe {
    static constexpr int NL = P::NL;
    uint32_t l[NL];

    BPP_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = 0;
        return r;
    }
    BPP_HD static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = P::ONE[i];
        return r;
    }
    BPP_HD static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = P::R2[i];
        return r;
    }
    // value == 0 mod p, for a value in [0, 2p): the limbs spell 0 or p
    BPP_HD bool is_zero() const {
        uint32_t o = 0, q = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            o |= l[i];
            q |= l[i] ^ P::MOD[i];
        }
        return o == 0 || q == 0;
    }
    BPP_HD bool operator==(const Fe& b) const;   // (a - b) == 0 mod p, defined after fe_sub
    BPP_HD bool operator!=(const Fe& b) const { return !(*this == b); }
};

// a in [0, 2M), limbs normalised -> a - M if a >= M   (M = p or 2p, given as limbs)
template <class P>
BPP_HD void fe_cond_sub(Fe<P>& a, const uint32_t* M) {
    constexpr int NL = P::NL;
    uint32_t d[NL];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)M[i] + borrow;
        d[i] = (uint32_t)t & LIMB_MASK;
        borrow = t >> LIMB_BITS;  // 0 or -1
    }
    const bool take = (borrow == 0);
#pragma unroll
    for (int i = 0; i < NL; i++) a.l[i] = take ? d[i] : a.l[i];
}
// [0, 2p) -> [0, p)
template <class P>
BPP_HD void fe_cond_sub_p(Fe<P>& a) {
    fe_cond_sub<P>(a, P::MOD);
}

// a, b in [0, 2p) -> a + b mod p in [0, 2p)
template <class P>
BPP_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    constexpr int NL = P::NL;
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t t = a.l[i] + b.l[i] + c;
        r.l[i] = t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    // a + b < 4p < 2^(30 NL): no carry out of the top limb
    fe_cond_sub<P>(r, P::MOD2);
    return r;
}

// a, b in [0, 2p) -> a - b mod p in [0, 2p)
template <class P>
BPP_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    constexpr int NL = P::NL;
    Fe<P> r;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + borrow;
        r.l[i] = (uint32_t)t & LIMB_MASK;
        borrow = t >> LIMB_BITS;
    }
    const uint32_t mask = (uint32_t)borrow;  // all ones when a < b: add 2p
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t t = r.l[i] + (P::MOD2[i] & mask) + c;
        r.l[i] = t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    return r;
}

template <class P>
BPP_HD bool Fe<P>::operator==(const Fe<P>& b) const {
    return fe_sub(*this, b).is_zero();
}

template <class P>
BPP_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub(Fe<P>::zero(), a);
}

template <class P>
BPP_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add(a, a);
}

// ---- lazy ("nr" = not reduced) additions for straight-line formulas ---------------------------------------
// The Montgomery radix leaves room above p (R / p = P::HEADROOM >= 630 for every field here): a product of
// a < alpha p and b < beta p reduces to < p (1 + alpha beta / HEADROOM), and the column accumulators of
// fe_mul only need NORMALISED limbs (< 2^30), not a reduced value.  So inside a formula a sum or difference
// may skip the conditional subtraction altogether -- one carry pass, 3-5 instructions per limb instead of ~10
// -- as long as the caller keeps track of the multiple of p each value stays below (<= 8 p here, so that
// every product of two such values stays far below HEADROOM p^2).  Callers state their bounds in comments.

// a + b (no reduction); the caller guarantees a + b < 2^(30 NL)
template <class P>
BPP_HD Fe<P> fe_add_nr(const Fe<P>& a, const Fe<P>& b) {
    constexpr int NL = P::NL;
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t t = a.l[i] + b.l[i] + c;
        r.l[i] = t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    return r;
}

// a - b + K p (no reduction); the caller guarantees b <= K p, so the value is in [0, a + K p]
template <int K, class P>
BPP_HD Fe<P> fe_sub_nr(const Fe<P>& a, const Fe<P>& b) {
    static_assert(K >= 0 && K <= 8, "MODK holds 0..8 p");
    constexpr int NL = P::NL;
    Fe<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)P::MODK[K][i] + c;
        r.l[i] = (uint32_t)t & LIMB_MASK;
        c = t >> LIMB_BITS;   // arithmetic: -1, 0 or 1
    }
    return r;
}

// (neg ? -a : a) - b + K p (no reduction); the caller guarantees a + b <= K p
template <int K, class P>
BPP_HD Fe<P> fe_csub_nr(const Fe<P>& a, bool neg, const Fe<P>& b) {
    static_assert(K >= 0 && K <= 8, "MODK holds 0..8 p");
    constexpr int NL = P::NL;
    const int32_t s = neg ? -1 : 0;
    Fe<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int32_t t = (((int32_t)a.l[i] ^ s) - s) - (int32_t)b.l[i] + (int32_t)P::MODK[K][i] + c;
        r.l[i] = (uint32_t)t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    return r;
}

// a + 2 b (no reduction); the caller guarantees a + 2 b < 2^(30 NL)
template <class P>
BPP_HD Fe<P> fe_add_dbl_nr(const Fe<P>& a, const Fe<P>& b) {
    constexpr int NL = P::NL;
    Fe<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t t = a.l[i] + (b.l[i] << 1) + c;   // < 3 * 2^30 + 3
        r.l[i] = t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    return r;
}

// value == 0 mod p for a normalised value in [0, (KMAX + 1) p): compares with 0, p, .., KMAX p (the slow,
// exact test -- lazy formulas reach it only on their exceptional paths)
template <int KMAX, class P>
BPP_HD bool fe_is_zero_mod(const Fe<P>& a) {
    static_assert(KMAX >= 0 && KMAX <= 8, "MODK holds 0..8 p");
    bool z = false;
#pragma unroll
    for (int k = 0; k <= KMAX; k++) {
        uint32_t q = 0;
#pragma unroll
        for (int i = 0; i < P::NL; i++) q |= a.l[i] ^ P::MODK[k][i];
        z = z || q == 0;
    }
    return z;
}

// normalised value < K p ?  (bound checks of the lazy formulas in host-side tests)
template <int K, class P>
BPP_HD bool fe_below_kp(const Fe<P>& a) {
    static_assert(K >= 0 && K <= 8, "MODK holds 0..8 p");
    for (int i = P::NL - 1; i >= 0; i--) {
        if (a.l[i] >> LIMB_BITS) return false;   // not normalised
        if (a.l[i] != P::MODK[K][i]) return a.l[i] < P::MODK[K][i];
    }
    return false;
}

// ---- the multiplier ------------------------------------------------------------------------------------------
// Column-wise (product scanning) with ONE 64-bit running accumulator: column k sums at most NL products < 2^60
// plus a carry < 2^35, which cannot overflow; per column one mask (the limb) and one 64-bit shift (the carry).
//
// What the hardware wants (measured on MI355X, tools/ubench.hip, profiles/ubench_r02.json): one wave issues a
// v_mad_u64_u32 every ~9.6 clocks whether or not it depends on the previous one, so instruction-level
// parallelism inside a wave buys nothing for this instruction -- the SIMD's second wave fills the pipe -- and
// every other instruction costs its full issue slot (v_mad + v_lshrrev_b64 back to back run at the SUM of their
// costs: nothing hides in the multiplier's shadow).  Left to itself the compiler reassociates the column sums to
// shorten the dependency chain: every column starts from zero and the previous column's carry is merged with
// an extra 64-bit add (52 v_lshl_add_u64 and ~25 v_mov per product).  An empty asm on the accumulator at the
// end of each column stops the reassociation there: the carry stays the addend of the column's first
// multiply-add, one chain, no merge instructions.  (Writing the multiply-adds themselves as inline asm is worse:
// the compiler must assume the gfx940 dst-forwarding hazard after every asm and pads each with an s_nop.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BPP_NO_CHAIN_BARRIER)
#define BPP_CHAIN_BARRIER(acc) asm volatile("" : "+v"(acc))
// the constant 1 in a register the compiler cannot see through: `acc += x * one` is then emitted as ONE
// v_mad_u64_u32 instead of a zero-extension (v_mov) plus a 64-bit add
__device__ __forceinline__ uint32_t opaque_one() {
    uint32_t one;
    asm("v_mov_b32 %0, 1" : "=v"(one));
    return one;
}
// the same barrier over the accumulator AND the reduction multipliers found so far, in one statement: all of them
// then look equally "late" to the reassociation, which keeps the carry as the seed of the next column's chain
// (otherwise the products of the early multipliers are summed on their own and merged with a 64-bit add)
template <int CNT>
__device__ __forceinline__ void chain_barrier_m(uint64_t& acc, uint32_t* m) {
    static_assert(CNT >= 0 && CNT <= 13, "operand limit of one asm statement");
    if constexpr (CNT == 0) asm volatile("" : "+v"(acc));
    if constexpr (CNT == 1) asm volatile("" : "+v"(acc), "+v"(m[0]));
    if constexpr (CNT == 2) asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]));
    if constexpr (CNT == 3) asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]));
    if constexpr (CNT == 4) asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
    if constexpr (CNT == 5) asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]));
    if constexpr (CNT == 6)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]));
    if constexpr (CNT == 7)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]));
    if constexpr (CNT == 8)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]));
    if constexpr (CNT == 9)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]), "+v"(m[8]));
    if constexpr (CNT == 10)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]), "+v"(m[8]), "+v"(m[9]));
    if constexpr (CNT == 11)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]));
    if constexpr (CNT == 12)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]));
    if constexpr (CNT == 13)
        asm volatile("" : "+v"(acc), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]),
                          "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]), "+v"(m[12]));
}
// ... and over both accumulators and ALL limbs of one (or two) operands of the plain products.  LLVM's reassociation
// orders the terms of a column sum by how late they are defined and seeds the chain with the EARLIEST: the operands
// of a * b are old values, so left alone the products are summed on their own from zero and the carry joins with a
// 64-bit add.  After this statement the operand limbs are "younger" than the carries, and every column is one chain
// of multiply-adds that starts from its carry.  No instruction is emitted; the operand must be a copy the caller no
// longer needs in its old form (a value still live elsewhere would cost a register move per limb).
#define BPP_L9(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8])
#define BPP_L13(x) BPP_L9(x), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12])
// one statement for everything (two statements in a row cost a wait state each: the hazard recogniser assumes the
// worst of an inline asm's outputs): both accumulators, the CNT reduction multipliers the next column reads, and the
// operand limbs given as the variadic part
#define BPP_PIN_CASES(...)                                                                                                   \
    if constexpr (CNT == 0) asm volatile("" : "+v"(accB), "+v"(accA), __VA_ARGS__);                                         \
    if constexpr (CNT == 1) asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), __VA_ARGS__);                             \
    if constexpr (CNT == 2) asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), __VA_ARGS__);                 \
    if constexpr (CNT == 3) asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), __VA_ARGS__);     \
    if constexpr (CNT == 4)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), __VA_ARGS__);             \
    if constexpr (CNT == 5)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), __VA_ARGS__); \
    if constexpr (CNT == 6)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          __VA_ARGS__);                                                                                      \
    if constexpr (CNT == 7)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), __VA_ARGS__);                                                                          \
    if constexpr (CNT == 8)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), __VA_ARGS__);                                                              \
    if constexpr (CNT == 9)                                                                                                  \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), __VA_ARGS__);                                                  \
    if constexpr (CNT == 10)                                                                                                 \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), __VA_ARGS__);                                      \
    if constexpr (CNT == 11)                                                                                                 \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), __VA_ARGS__);                         \
    if constexpr (CNT == 12)                                                                                                 \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]), __VA_ARGS__);            \
    if constexpr (CNT == 13)                                                                                                 \
        asm volatile("" : "+v"(accB), "+v"(accA), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]),   \
                          "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]), "+v"(m[12]), __VA_ARGS__);
template <int NL, int CNT>
__device__ __forceinline__ void chain_barrier_ops(uint64_t& accB, uint64_t& accA, uint32_t* m, uint32_t* x) {
    static_assert((NL == 9 || NL == 13) && CNT >= 0 && CNT <= 13, "limb counts of the fields in use");
    if constexpr (NL == 9) {
        BPP_PIN_CASES(BPP_L9(x))
    } else {
        BPP_PIN_CASES(BPP_L13(x))
    }
}
template <int NL, int CNT>
__device__ __forceinline__ void chain_barrier_ops(uint64_t& accB, uint64_t& accA, uint32_t* m, uint32_t* x, uint32_t* y) {
    static_assert((NL == 9 || NL == 13) && CNT >= 0 && CNT <= 13, "limb counts of the fields in use");
    if constexpr (NL == 9) {
        BPP_PIN_CASES(BPP_L9(x), BPP_L9(y))
    } else {
        BPP_PIN_CASES(BPP_L13(x), BPP_L13(y))
    }
}
#else
#define BPP_CHAIN_BARRIER(acc) ((void)0)
BPP_HD uint32_t opaque_one() { return 1u; }
template <int CNT>
BPP_HD void chain_barrier_m(uint64_t&, uint32_t*) {}
template <int NL, int CNT>
BPP_HD void chain_barrier_ops(uint64_t&, uint64_t&, uint32_t*, uint32_t*) {}
template <int NL, int CNT>
BPP_HD void chain_barrier_ops(uint64_t&, uint64_t&, uint32_t*, uint32_t*, uint32_t*) {}
#endif

// Montgomery reduction of a 2*NL-limb product T (limbs < 2^31): returns T * R^-1 mod p.
// Column k sums at most NL products < 2^60, a limb < 2^31 and a carry < 2^35: no overflow.
template <class P>
BPP_HD Fe<P> fe_mont_reduce(const uint32_t* T) {
    constexpr int NL = P::NL;
    uint32_t m[NL];
    Fe<P> r;
    uint64_t acc = 0;
    const uint32_t one = opaque_one();
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc += (uint64_t)T[k] * one;
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::MOD[k - i];
        m[k] = ((uint32_t)acc * P::INV) & LIMB_MASK;
        acc += (uint64_t)m[k] * P::MOD[0];
        acc >>= LIMB_BITS;
        BPP_CHAIN_BARRIER(acc);
    }
#pragma unroll
    for (int k = NL; k < 2 * NL; k++) {
        acc += (uint64_t)T[k] * one;
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)m[i] * P::MOD[k - i];
        r.l[k - NL] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
        BPP_CHAIN_BARRIER(acc);
    }
    // (T + m p) / R < T / R + p: far below 2p for every caller (T < HEADROOM / 8 p^2)  =>  acc == 0 here
    return r;
}

// Plain product a*b as 2 NL normalised limbs (NL^2 v_mad_u64_u32)
template <class P>
BPP_HD void fe_mul_wide(const Fe<P>& a, const Fe<P>& b, uint32_t* T) {
    constexpr int NL = P::NL;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        T[k] = (uint32_t)acc & LIMB_MASK;
        acc >>= LIMB_BITS;
        BPP_CHAIN_BARRIER(acc);
    }
    T[2 * NL - 1] = (uint32_t)acc;
}

// ---- fused product + Montgomery reduction ------------------------------------------------------------------
// fe_mul / fe_sqr / fe_mul_add run the product and its reduction in ONE pass over the 2 NL columns.  Column k
// receives na(k) = min(k, 2NL-2-k) + 1 plain products (W times that for the two-product form) and nm(k) =
// (k < NL ? k + 1 : 2NL-1-k) products m_i p_j of the reduction.  Where W na + nm <= 15 -- the outer columns --
// they all fit one 64-bit accumulator (15 * 2^60 + carry < 2^64), so the column needs no intermediate limb
// T[k] at all: no mask, no second shift, no re-entry of T[k] into the reduction chain.  Only the middle
// columns (k = 7..17 of 26 for the 13-limb field, k = 7..9 of 18 for the 9-limb fields) would overflow; there
// the plain products run in a second accumulator whose limb enters the reduction chain as T * 1 (one
// multiply-add).  Per 13-limb product: 338 + 11 v_mad_u64_u32, 37 masks, 37 64-bit shifts, 13 v_mul_lo_u32,
// against 338 v_mad + 60 v_lshl_add_u64 + 48 shifts + 51 masks + 25 v_mov + 13 v_mul_lo of the two-pass form
// as the compiler scheduled it.
//   col(k, acc): adds the plain products of column k (k <= 2NL-2) to acc.
// acc += a * b, and acc += a * k for a limb k of the modulus.
// Tried and not kept (-DBPP_ASM_MAD): every multiply-add as inline asm pins the chain order completely (no
// reassociation, no 64-bit merge adds) but the compiler pads almost every asm statement with an s_nop (it must assume
// the gfx940 dst-forwarding hazard), and on MI355X those nops cost what the merges cost: fe_mul 80.3 vs 80.6 G/s, lazy
// XYZZ addition 7.41 vs 7.66 G/s on the same box (tools/ubench.hip).
// Kept for the sparse moduli only (P::SPARSE_MOD: secp256k1's 2^256 - 2^32 - 977 and 2^255 - 19, whose 30-bit limbs are
// mostly all-ones): there the compiler strength-reduces m * 0x3fffffff into shifts, subtractions and 64-bit adds that
// cost more issue slots than the multiply-add they replace; the asm form keeps the v_mad_u64_u32 (+5 % on the
// secp256k1 addition).
BPP_HD void fe_mad(uint64_t& acc, uint32_t a, uint32_t b) {
#if defined(BPP_ASM_MAD) && defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
    acc += (uint64_t)a * b;
#endif
}
template <class P>
BPP_HD void fe_mad_k(uint64_t& acc, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(BPP_ASM_MAD)
    constexpr bool use_asm = true;
#else
    constexpr bool use_asm = P::SPARSE_MOD;
#endif
    if constexpr (use_asm) {
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
        return;
    }
#endif
    acc += (uint64_t)a * k;
}

template <class P, int W, int K, class ColFn, class PinFn>
BPP_HD void fe_fused_column(ColFn& col, PinFn& pin, uint64_t& accA, uint64_t& accB, uint32_t* m, Fe<P>& r, uint32_t one) {
    constexpr int NL = P::NL;
    constexpr int na = K <= 2 * NL - 2 ? (K < 2 * NL - 2 - K ? K : 2 * NL - 2 - K) + 1 : 0;
    constexpr int nm = K < NL ? K + 1 : 2 * NL - 1 - K;
    constexpr bool fused = W * na + nm <= 15;
    // the column before this one: was it split?  (its plain-product chain then holds a carry for this column)
    constexpr int nap = K >= 1 ? ((K - 1) < 2 * NL - 2 - (K - 1) ? (K - 1) : 2 * NL - 2 - (K - 1)) + 1 : 0;
    constexpr int nmp = K >= 1 ? ((K - 1) < NL ? K : 2 * NL - K) : 0;
    constexpr bool prev_split = K >= 1 && !(W * nap + nmp <= 15);
    if constexpr (!fused) {
        if constexpr (!prev_split) accA = 0;
        col(K, accA);
        const uint32_t T = (uint32_t)accA & LIMB_MASK;
        accA >>= LIMB_BITS;
        fe_mad(accB, T, one);
    } else {
        if constexpr (prev_split) accB += accA;
        if constexpr (na > 0) col(K, accB);
    }
    if constexpr (K < NL) {
#pragma unroll
        for (int i = 0; i < K; i++) fe_mad_k<P>(accB, m[i], P::MOD[K - i]);
        m[K] = ((uint32_t)accB * P::INV) & LIMB_MASK;
        fe_mad_k<P>(accB, m[K], P::MOD[0]);
    } else {
#pragma unroll
        for (int i = K - NL + 1; i < NL; i++) fe_mad_k<P>(accB, m[i], P::MOD[K - i]);
        r.l[K - NL] = (uint32_t)accB & LIMB_MASK;
    }
    accB >>= LIMB_BITS;
    if constexpr (K + 1 < 2 * NL) {
        // the multipliers column K + 1 reads: m[lo .. hi)
        constexpr int lo = K + 1 < NL ? 0 : K + 1 - NL + 1;
        constexpr int hi = K + 1 < NL ? K + 1 : NL;
        pin(accB, accA, m + lo, std::integral_constant<int, (hi > lo ? hi - lo : 0)>{});
        fe_fused_column<P, W, K + 1>(col, pin, accA, accB, m, r, one);
    }
}

template <class P, int W, class ColFn, class PinFn>
BPP_HD Fe<P> fe_fused_reduce(ColFn&& col, PinFn&& pin) {
    uint32_t m[P::NL];
    Fe<P> r;
    uint64_t accB = 0, accA = 0;   // reduction chain (and the fused columns); plain-product chain of the middle columns
    fe_fused_column<P, W, 0>(col, pin, accA, accB, m, r, opaque_one());
    // (T + m p) / R < T / R + p: far below 2p for every caller (T < HEADROOM / 8 p^2)  =>  accB == 0 here
    return r;
}

// Montgomery product a*b*R^-1 mod p.  For a < alpha p, b < beta p the result is < p (1 + alpha beta / HEADROOM).
// fe_mul_io: `a` is read AND handed back (same value, re-defined: see chain_barrier_ops) -- for an operand that is
// needed again afterwards, so that its later readers use the limbs as the multiplication left them and no register
// copies are made to keep the old ones alive.  fe_mul takes the operand by value-semantics instead: free when `a` dies
// here (pass the operand that dies first as `a`), NL register moves otherwise.
template <class P>
BPP_HD Fe<P> fe_mul_io(Fe<P>& a, const Fe<P>& b) {
    constexpr int NL = P::NL;
    uint32_t* x = a.l;
    return fe_fused_reduce<P, 1>(
        [&](int k, uint64_t& acc) {
#pragma unroll
            for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) fe_mad(acc, x[i], b.l[k - i]);
        },
        [&](uint64_t& accB, uint64_t& accA, uint32_t* mm, auto cnt) { chain_barrier_ops<NL, decltype(cnt)::value>(accB, accA, mm, x); });
}
template <class P>
BPP_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> x = a;
    return fe_mul_io(x, b);
}

// (a*b + c*d) * R^-1 mod p with ONE Montgomery reduction: < p (1 + (alpha beta + gamma delta) / HEADROOM).
// Saves NL^2 + NL of the 3 NL^2 + NL multiplier operations of two separate products.
template <class P>
BPP_HD Fe<P> fe_mul_add(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
    constexpr int NL = P::NL;
    uint32_t x[NL], y[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        x[i] = a.l[i];
        y[i] = c.l[i];
    }
    return fe_fused_reduce<P, 2>(
        [&](int k, uint64_t& acc) {
#pragma unroll
            for (int i = (k < NL ? 0 : k - NL + 1); i <= (k < NL ? k : NL - 1); i++) {
                fe_mad(acc, x[i], b.l[k - i]);
                fe_mad(acc, y[i], d.l[k - i]);
            }
        },
        [&](uint64_t& accB, uint64_t& accA, uint32_t* mm, auto cnt) {
            chain_barrier_ops<NL, decltype(cnt)::value>(accB, accA, mm, x, y);
        });
}

// Montgomery square: the NL(NL-1)/2 cross products are computed once, against a pre-doubled copy of the operand
// (2 a_j < 2^31: a cross product counts as two plain ones in the column bound, which is what it replaces):
// 91 v_mad_u64_u32 instead of 169 for the product part.
template <class P>
BPP_HD Fe<P> fe_sqr_io(Fe<P>& a) {
    constexpr int NL = P::NL;
    uint32_t a2[NL];
    uint32_t* x = a.l;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    return fe_fused_reduce<P, 1>(
        [&](int k, uint64_t& acc) {
            // pairs i < j, i + j = k
#pragma unroll
            for (int i = (k < NL ? 0 : k - NL + 1); 2 * i < k; i++) fe_mad(acc, x[i], a2[k - i]);
            if ((k & 1) == 0) fe_mad(acc, x[k / 2], x[k / 2]);
        },
        [&](uint64_t& accB, uint64_t& accA, uint32_t* mm, auto cnt) { chain_barrier_ops<NL, decltype(cnt)::value>(accB, accA, mm, x); });
}
template <class P>
BPP_HD Fe<P> fe_sqr(const Fe<P>& a) {
    Fe<P> x = a;
    return fe_sqr_io(x);
}

// ---- memory / wire formats -----------------------------------------------------------------------

// N packed 32-bit words (value < 2^(32N), expected < p) -> NL 30-bit limbs.  Pure bit shuffling.
template <class P>
BPP_HD Fe<P> fe_unpack(const uint32_t* w) {
    constexpr int NL = P::NL, N = P::N;
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LIMB_BITS * i;
        const int wi = bit >> 5, s = bit & 31;
        uint32_t v = (wi < N) ? (w[wi] >> s) : 0u;
        if (s > 32 - LIMB_BITS && wi + 1 < N) v |= w[wi + 1] << (32 - s);
        r.l[i] = v & LIMB_MASK;
    }
    return r;
}

// NL 30-bit limbs (value < 2^(32N)) -> N packed 32-bit words
template <class P>
BPP_HD void fe_pack(const Fe<P>& a, uint32_t* w) {
    constexpr int NL = P::NL, N = P::N;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const int bit = 32 * j;
        const int li = bit / LIMB_BITS, o = bit % LIMB_BITS;
        uint32_t v = a.l[li] >> o;
        if (li + 1 < NL) v |= a.l[li + 1] << (LIMB_BITS - o);
        if (2 * LIMB_BITS - o < 32 && li + 2 < NL) v |= a.l[li + 2] << (2 * LIMB_BITS - o);
        w[j] = v;
    }
}

// is the N-word canonical value < p ?
template <class P>
BPP_HD bool words_lt_mod(const uint32_t* w) {
    // compare from the top word down
    for (int i = P::N - 1; i >= 0; i--) {
        if (w[i] < P::MODW[i]) return true;
        if (w[i] > P::MODW[i]) return false;
    }
    return false;
}

// canonical words (any value < 2^(32N); values >= p are reduced, as PrimeFieldElem::new does for
// BigUint inputs, reference src/secp256k1/building_block/field/prime_field_elem.rs:236-246) -> Montgomery
template <class P>
BPP_HD Fe<P> fe_from_canonical(const uint32_t* w) {
    Fe<P> t = fe_unpack<P>(w);
    // value < 2^(32N) < 2^(30 NL): T = t * R2 as a plain product, then one Montgomery reduction.
    // fe_mul requires a normalised operand but not a reduced one: (t*R2 + m p)/R < p * (t/R + 1) < 2p.
    return fe_mul(t, Fe<P>::r2());
}

template <class P>
BPP_HD void fe_to_canonical(const Fe<P>& a, uint32_t* w) {
    uint32_t T[2 * P::NL];
#pragma unroll
    for (int i = 0; i < P::NL; i++) T[i] = a.l[i];
#pragma unroll
    for (int i = P::NL; i < 2 * P::NL; i++) T[i] = 0;
    Fe<P> t = fe_mont_reduce<P>(T);   // < a / R + p < 2p
    fe_cond_sub_p(t);
    fe_pack(t, w);
}

// Montgomery-form element <-> its packed memory image (no arithmetic, only repacking)
template <class P>
BPP_HD Fe<P> fe_load(const uint32_t* w) {
    return fe_unpack<P>(w);
}
template <class P>
BPP_HD void fe_store(const Fe<P>& a, uint32_t* w) {
    Fe<P> t = a;
    fe_cond_sub_p(t);   // the memory image is the fully reduced value (2p need not fit in N words)
    fe_pack(t, w);
}

template <class P>
BPP_HD Fe<P> fe_from_u32(uint32_t x) {
    Fe<P> t = Fe<P>::zero();
    t.l[0] = x & LIMB_MASK;
    t.l[1] = x >> LIMB_BITS;
    return fe_mul(t, Fe<P>::r2());
}

// PrimeFieldElem::new(i32): negative n -> p - |n|
// (reference src/bls12_381/building_block/scalar/prime_field_elem.rs:191-195, Fr::set_int)
template <class P>
BPP_HD Fe<P> fe_from_i32(int32_t n) {
    if (n >= 0) return fe_from_u32<P>((uint32_t)n);
    return fe_neg(fe_from_u32<P>((uint32_t)(-(int64_t)n)));
}

// a^(p-2) by square-and-multiply over the bits of p-2 (a = 0 -> 0): ~1.5 BITS Montgomery products.  Kept
// as the independent cross-check of fe_inv (tests/host/field_host_test.cpp, tools/ubench.hip).
template <class P>
BPP_HD Fe<P> fe_inv_fermat(const Fe<P>& a) {
    Fe<P> acc = Fe<P>::one();
    for (int i = P::BITS - 1; i >= 0; i--) {
        acc = fe_sqr(acc);
        if ((P::PM2[i >> 5] >> (i & 31)) & 1u) acc = fe_mul(acc, a);
    }
    return acc;
}

// Modular inverse by the Bernstein-Yang "safegcd" division steps (half-delta variant), arranged for this
// library's 30-bit limbs: 30 divsteps are run on the low words of (f, g) alone and collected in a 2x2
// transition matrix t = [[u, v], [q, r]] (entries < 2^30 in magnitude); t is then applied to the full-width
// (f, g) -- an exact division by 2^30 -- and to the Bezout pair (d, e) modulo p.  P::INV_BATCHES batches make
// g = 0 and f = +-1 for every input, so there is no data-dependent control flow: all 64 lanes of a wave run
// the same instructions.  Cost on gfx950: ~30 batches x (600 32-bit ops + 130 v_mad_i64_i32) for the 381-bit
// field, i.e. about 26 Montgomery products instead of the ~570 of Fermat's little theorem.
// Same field element as the reference's Fr::inv (mcl) / ext-Euclid safe_inv
// (field/prime_field_elem.rs:339-392); a = 0 -> 0.
namespace safegcd {

// value = sum v[i] 2^(30 i); limbs 0..NL-2 in [0, 2^30), top limb signed
template <int NL>
struct S30 {
    int32_t v[NL];
};

struct Trans {
    int32_t u, v, q, r;
};

// 30 half-delta divsteps on the low words.  zeta = -(delta + 1/2).
BPP_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, Trans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);     // all ones iff delta > 0
        const uint32_t c2 = 0u - (g & 1u);        // all ones iff g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v) or their negation
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                                 // swap: delta > 0 and g odd
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}

// signed 32 x 32 -> 64 product.  The limb operand is passed through an empty asm so that the compiler forgets
// it is a masked (non-negative) value: otherwise it widens signed x unsigned to a 64 x 32 multiply
// (v_mad_u64_u32 + v_mul_lo_u32) instead of one v_mad_i64_i32.
BPP_HD int64_t mul32(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(b));
#endif
    return (int64_t)a * (int64_t)b;
}

// (f, g) <- t (f, g) / 2^30   (the division is exact)
template <int NL>
BPP_HD void update_fg(S30<NL>& f, S30<NL>& g, const Trans& t) {
    int64_t cf = mul32(t.u, f.v[0]) + mul32(t.v, g.v[0]);
    int64_t cg = mul32(t.q, f.v[0]) + mul32(t.r, g.v[0]);
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        const int32_t fi = f.v[i], gi = g.v[i];
        cf += mul32(t.u, fi) + mul32(t.v, gi);
        cg += mul32(t.q, fi) + mul32(t.r, gi);
        f.v[i - 1] = (int32_t)((uint32_t)cf & LIMB_MASK);
        g.v[i - 1] = (int32_t)((uint32_t)cg & LIMB_MASK);
        cf >>= 30;
        cg >>= 30;
    }
    f.v[NL - 1] = (int32_t)cf;
    g.v[NL - 1] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod p, both kept in (-2p, p): a multiple of p is added first to make the
// combination non-negative-ish, then the multiple that clears the low 30 bits.
template <class P>
BPP_HD void update_de(S30<P::NL>& d, S30<P::NL>& e, const Trans& t) {
    constexpr int NL = P::NL;
    const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = mul32(t.u, d.v[0]) + mul32(t.v, e.v[0]);
    int64_t ce = mul32(t.q, d.v[0]) + mul32(t.r, e.v[0]);
    md -= (int32_t)((P::PINV * (uint32_t)cd + (uint32_t)md) & LIMB_MASK);
    me -= (int32_t)((P::PINV * (uint32_t)ce + (uint32_t)me) & LIMB_MASK);
    cd += mul32((int32_t)P::MOD[0], md);
    ce += mul32((int32_t)P::MOD[0], me);
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < NL; i++) {
        const int32_t di = d.v[i], ei = e.v[i];
        cd += mul32(t.u, di) + mul32(t.v, ei) + mul32((int32_t)P::MOD[i], md);
        ce += mul32(t.q, di) + mul32(t.r, ei) + mul32((int32_t)P::MOD[i], me);
        d.v[i - 1] = (int32_t)((uint32_t)cd & LIMB_MASK);
        e.v[i - 1] = (int32_t)((uint32_t)ce & LIMB_MASK);
        cd >>= 30;
        ce >>= 30;
    }
    d.v[NL - 1] = (int32_t)cd;
    e.v[NL - 1] = (int32_t)ce;
}

// r in (-2p, p) -> [0, p), negated first when `neg` (f ended at -1)
template <class P>
BPP_HD void normalize(S30<P::NL>& r, int32_t neg) {
    constexpr int NL = P::NL;
    int32_t add = r.v[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t x = r.v[i] + ((int32_t)P::MOD[i] & add);
        r.v[i] = (x ^ neg) - neg;
    }
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= (int32_t)LIMB_MASK;
    }
    add = r.v[NL - 1] >> 31;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] += (int32_t)P::MOD[i] & add;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= (int32_t)LIMB_MASK;
    }
}

}  // namespace safegcd

// x^-1 mod p for a plain (non-Montgomery) residue given as normalised limbs of a value in [0, p)
template <class P>
BPP_HD Fe<P> fe_inv_plain(const Fe<P>& x) {
    constexpr int NL = P::NL;
    safegcd::S30<NL> d, e, f, g;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        d.v[i] = 0;
        e.v[i] = i == 0 ? 1 : 0;
        f.v[i] = (int32_t)P::MOD[i];
        g.v[i] = (int32_t)x.l[i];
    }
    int32_t zeta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int it = 0; it < P::INV_BATCHES; it++) {
        safegcd::Trans t;
        zeta = safegcd::divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        safegcd::update_de<P>(d, e, t);
        safegcd::update_fg<NL>(f, g, t);
    }
    safegcd::normalize<P>(d, f.v[NL - 1] >> 31);
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = (uint32_t)d.v[i];
    return r;
}

// Montgomery form in, Montgomery form out: (aR)^-1 = a^-1 R^-1, times R^3 / R = a^-1 R.
template <class P>
BPP_HD Fe<P> fe_inv(const Fe<P>& a) {
    Fe<P> t = a;
    fe_cond_sub_p(t);
    Fe<P> r3;
#pragma unroll
    for (int i = 0; i < P::NL; i++) r3.l[i] = P::R3[i];
    return fe_mul(fe_inv_plain(t), r3);
}

// a^n for a small exponent (reference src/util.rs:39-52 scalar_exp_vartime)
template <class P>
BPP_HD Fe<P> fe_pow_u64(const Fe<P>& a, uint64_t n) {
    Fe<P> result = Fe<P>::one(), aux = a;
    while (n > 0) {
        if (n & 1) result = fe_mul(result, aux);
        n >>= 1;
        aux = fe_sqr(aux);
    }
    return result;
}

}  // namespace bpp
