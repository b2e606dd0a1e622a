// ec.hpp -- short-Weierstrass (a = 0) group law for BLS12-381 G1 and secp256k1, host + gfx950.
//
// Replaces, on the hot path, the reference's `Point` (mcl G1: add/sub/mul/neg/zero/is_zero/eq,
// reference src/bls12_381/building_block/point/point.rs:34-117) and the secp256k1 `AffinePoint`
// (case analysis of reference src/secp256k1/building_block/macros.rs:42-146).
//
// Every addition here is COMPLETE for the cases the reference distinguishes: inf + Q, P + inf,
// P + (-P) -> inf, P + P -> doubling.  This matters on the parity vectors: the reference's generators
// are small multiples of g (G_4 = H_2 = 15 g, reference src/publickey.rs:31,38), so equal and opposite
// points do meet inside an MSM.
//
// Memory images (all little-endian 32-bit words, Fp in Montgomery form, see field.hpp):
//   affine  : x | y                (2 N words)   x = y = 0 encodes the point at infinity
//             (not on y^2 = x^3 + b for b != 0, so the encoding is unambiguous)
//   jacobian: X | Y | Z            (3 N words)   Z = 0 encodes infinity
//   wire    : x | y | inf          (2 N + 2 words = (2 L + 1) u64), canonical (non-Montgomery) --
//             the C-ABI format of include/bpp_amd.h
#pragma once
#include "field.hpp"

namespace bpp {

struct Bls12381 {
    using Fp = BlsFp;
    using Fr = BlsFr;
    using K = Bls12381Consts;
    static constexpr int ID = 0;
};
struct Secp256k1 {
    using Fp = SecpFp;
    using Fr = SecpFr;
    using K = Secp256k1Consts;
    static constexpr int ID = 1;
};

// 32-bit words of the packed projective ("jacobian") image of a point: 3 coordinates for the
// short-Weierstrass curves
template <class C>
constexpr int jac_words() {
    return 3 * C::Fp::N;
}

template <class C>
struct Aff {
    Fe<typename C::Fp> x, y;
    BPP_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class C>
struct Jac {
    Fe<typename C::Fp> X, Y, Z;
    BPP_HD bool is_inf() const { return Z.is_zero(); }
};

template <class C>
BPP_HD Aff<C> aff_inf() {
    Aff<C> r;
    r.x = Fe<typename C::Fp>::zero();
    r.y = Fe<typename C::Fp>::zero();
    return r;
}
template <class C>
BPP_HD Jac<C> jac_inf() {
    Jac<C> r;
    r.X = Fe<typename C::Fp>::one();
    r.Y = Fe<typename C::Fp>::one();
    r.Z = Fe<typename C::Fp>::zero();
    return r;
}
template <class C>
BPP_HD Jac<C> jac_from_aff(const Aff<C>& p) {
    if (p.is_inf()) return jac_inf<C>();
    Jac<C> r;
    r.X = p.x;
    r.Y = p.y;
    r.Z = Fe<typename C::Fp>::one();
    return r;
}
template <class C>
BPP_HD Aff<C> aff_generator() {
    Aff<C> g;
#pragma unroll
    for (int i = 0; i < C::Fp::NL; i++) {
        g.x.l[i] = C::K::GX[i];
        g.y.l[i] = C::K::GY[i];
    }
    return g;
}
template <class C>
BPP_HD Aff<C> aff_neg(const Aff<C>& p) {
    Aff<C> r;
    r.x = p.x;
    r.y = fe_neg(p.y);  // -0 = 0 keeps the infinity encoding
    return r;
}
template <class C>
BPP_HD Jac<C> jac_neg(const Jac<C>& p) {
    Jac<C> r = p;
    r.Y = fe_neg(p.Y);
    return r;
}

// y^2 == x^3 + b
template <class C>
BPP_HD bool aff_on_curve(const Aff<C>& p) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf()) return true;
    F b;
#pragma unroll
    for (int i = 0; i < C::Fp::NL; i++) b.l[i] = C::K::B[i];
    F lhs = fe_sqr(p.y);
    F rhs = fe_add(fe_mul(fe_sqr(p.x), p.x), b);
    return lhs == rhs;
}

// dbl-2009-l (a = 0): 2M + 5S.  Y = 0 has no points on these prime-order curves.
template <class C>
BPP_HD Jac<C> jac_dbl(const Jac<C>& p) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf()) return p;
    F A = fe_sqr(p.X);
    F B = fe_sqr(p.Y);
    F Cc = fe_sqr(B);
    F t = fe_sqr(fe_add(p.X, B));
    t = fe_sub(fe_sub(t, A), Cc);
    F D = fe_dbl(t);
    F E = fe_add(fe_dbl(A), A);
    F Fq = fe_sqr(E);
    Jac<C> r;
    r.X = fe_sub(Fq, fe_dbl(D));
    F C8 = fe_dbl(fe_dbl(fe_dbl(Cc)));
    r.Y = fe_sub(fe_mul(E, fe_sub(D, r.X)), C8);
    r.Z = fe_dbl(fe_mul(p.Y, p.Z));
    return r;
}

// doubling of an affine point (Z = 1): 1M + 5S
template <class C>
BPP_HD Jac<C> aff_dbl(const Aff<C>& p) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf()) return jac_inf<C>();
    F A = fe_sqr(p.x);
    F B = fe_sqr(p.y);
    F Cc = fe_sqr(B);
    F t = fe_sqr(fe_add(p.x, B));
    t = fe_sub(fe_sub(t, A), Cc);
    F D = fe_dbl(t);
    F E = fe_add(fe_dbl(A), A);
    F Fq = fe_sqr(E);
    Jac<C> r;
    r.X = fe_sub(Fq, fe_dbl(D));
    F C8 = fe_dbl(fe_dbl(fe_dbl(Cc)));
    r.Y = fe_sub(fe_mul(E, fe_sub(D, r.X)), C8);
    r.Z = fe_dbl(p.y);
    return r;
}

// add-2007-bl with the reference's case analysis (macros.rs:42-146): 11M + 5S
template <class C>
BPP_HD Jac<C> jac_add(const Jac<C>& p, const Jac<C>& q) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = fe_sqr(p.Z);
    F Z2Z2 = fe_sqr(q.Z);
    F U1 = fe_mul(p.X, Z2Z2);
    F U2 = fe_mul(q.X, Z1Z1);
    F S1 = fe_mul(fe_mul(p.Y, q.Z), Z2Z2);
    F S2 = fe_mul(fe_mul(q.Y, p.Z), Z1Z1);
    F H = fe_sub(U2, U1);
    F rr = fe_sub(S2, S1);
    if (H.is_zero()) {
        if (rr.is_zero()) return jac_dbl(p);  // same point
        return jac_inf<C>();                   // vertical line
    }
    rr = fe_dbl(rr);
    F I = fe_sqr(fe_dbl(H));
    F J = fe_mul(H, I);
    F V = fe_mul(U1, I);
    Jac<C> r;
    r.X = fe_sub(fe_sub(fe_sqr(rr), J), fe_dbl(V));
    r.Y = fe_mul_add(rr, fe_sub(V, r.X), fe_neg(fe_dbl(S1)), J);   // rr (V - X3) - 2 S1 J, one reduction
    F zz = fe_sqr(fe_add(p.Z, q.Z));
    r.Z = fe_mul(fe_sub(fe_sub(zz, Z1Z1), Z2Z2), H);
    return r;
}

// mixed addition, madd-2007-bl (Z2 = 1): 7M + 4S, same case analysis
template <class C>
BPP_HD Jac<C> jac_madd(const Jac<C>& p, const Aff<C>& q) {
    using F = Fe<typename C::Fp>;
    if (q.is_inf()) return p;
    if (p.is_inf()) return jac_from_aff(q);
    F Z1Z1 = fe_sqr(p.Z);
    F U2 = fe_mul(q.x, Z1Z1);
    F S2 = fe_mul(fe_mul(q.y, p.Z), Z1Z1);
    F H = fe_sub(U2, p.X);
    F rr = fe_sub(S2, p.Y);
    if (H.is_zero()) {
        if (rr.is_zero()) return aff_dbl(q);
        return jac_inf<C>();
    }
    rr = fe_dbl(rr);
    F HH = fe_sqr(H);
    F I = fe_dbl(fe_dbl(HH));
    F J = fe_mul(H, I);
    F V = fe_mul(p.X, I);
    Jac<C> r;
    r.X = fe_sub(fe_sub(fe_sqr(rr), J), fe_dbl(V));
    r.Y = fe_mul_add(rr, fe_sub(V, r.X), fe_neg(fe_dbl(p.Y)), J);   // rr (V - X3) - 2 Y1 J, one reduction
    F zh = fe_sqr(fe_add(p.Z, H));
    r.Z = fe_sub(fe_sub(zh, Z1Z1), HH);
    return r;
}

template <class C>
BPP_HD Aff<C> jac_to_aff(const Jac<C>& p) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf()) return aff_inf<C>();
    F zi = fe_inv(p.Z);
    F zi2 = fe_sqr(zi);
    Aff<C> r;
    r.x = fe_mul(p.X, zi2);
    r.y = fe_mul(p.Y, fe_mul(zi2, zi));
    return r;
}

// affine image of (X, Y, Z) given zi = Z^-1 (shared-inversion normalisation of several points)
template <class C>
BPP_HD Aff<C> jac_scale_to_aff(const Jac<C>& p, const Fe<typename C::Fp>& zi) {
    const Fe<typename C::Fp> zi2 = fe_sqr(zi);
    Aff<C> r;
    r.x = fe_mul(p.X, zi2);
    r.y = fe_mul(p.Y, fe_mul(zi2, zi));
    return r;
}

// ---- XYZZ accumulator (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 encodes infinity) ----------------------
// Used where a running sum only ever receives affine points (the fixed-generator MSM): the mixed addition
// costs 8M + 2S (madd-2008-s) against 7M + 4S for jacobian madd-2007-bl, with about half the add/sub work.
template <class C>
struct Xyzz {
    Fe<typename C::Fp> X, Y, ZZ, ZZZ;
    BPP_HD bool is_inf() const { return ZZ.is_zero(); }
};
template <class C>
BPP_HD Xyzz<C> xyzz_inf() {
    Xyzz<C> r;
    r.X = Fe<typename C::Fp>::one();
    r.Y = Fe<typename C::Fp>::one();
    r.ZZ = Fe<typename C::Fp>::zero();
    r.ZZZ = Fe<typename C::Fp>::zero();
    return r;
}
// 2 * (affine q), mdbl-2008-s-1 with a = 0
template <class C>
BPP_HD Xyzz<C> xyzz_dbl_aff(const Aff<C>& q) {
    using F = Fe<typename C::Fp>;
    if (q.is_inf()) return xyzz_inf<C>();
    F U = fe_dbl(q.y);
    F V = fe_sqr(U);
    F W = fe_mul(U, V);
    F S = fe_mul(q.x, V);
    F xx = fe_sqr(q.x);
    F M = fe_add(fe_dbl(xx), xx);
    Xyzz<C> r;
    r.X = fe_sub(fe_sqr(M), fe_dbl(S));
    r.Y = fe_mul_add(M, fe_sub(S, r.X), fe_neg(W), q.y);   // M (S - X3) - W y, one reduction
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}
// acc + affine q, complete for the reference's cases (inf, equal, opposite)
template <class C>
BPP_HD Xyzz<C> xyzz_madd(const Xyzz<C>& p, const Aff<C>& q) {
    using F = Fe<typename C::Fp>;
    if (q.is_inf()) return p;
    if (p.is_inf()) {
        Xyzz<C> r;
        r.X = q.x;
        r.Y = q.y;
        r.ZZ = F::one();
        r.ZZZ = F::one();
        return r;
    }
    F U2 = fe_mul(q.x, p.ZZ);
    F S2 = fe_mul(q.y, p.ZZZ);
    F Pp = fe_sub(U2, p.X);
    F R = fe_sub(S2, p.Y);
    if (Pp.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_aff(q);
        return xyzz_inf<C>();
    }
    F PP = fe_sqr(Pp);
    F PPP = fe_mul(Pp, PP);
    F Q = fe_mul(p.X, PP);
    Xyzz<C> r;
    r.X = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    r.Y = fe_mul_add(R, fe_sub(Q, r.X), fe_neg(p.Y), PPP);   // R (Q - X3) - Y1 PPP, one reduction
    r.ZZ = fe_mul(p.ZZ, PP);
    r.ZZZ = fe_mul(p.ZZZ, PPP);
    return r;
}
// The same addition for the MSM inner loops, acc += (neg ? -q : q), with LAZY field additions (field.hpp,
// fe_*_nr): no sum or difference inside the formula is reduced, the sign of q is folded into the one subtraction
// that consumes q.y, and the exceptional cases are detected AFTER the fact from the new ZZ (ZZ3 = ZZ * Pp^2
// vanishes iff the x-coordinates agree), so the common path carries one zero test instead of five.
// q: canonical affine coordinates (x, y < p; x = y = 0 is infinity), as the tables and proof points hold them.
// Invariants of the accumulator between calls (multiples of p, checked on the host by tests/host/lazy_host_test.cpp --
// random walks and an accumulator built AT the stated bounds -- run by tests/test_host_arith_cpu.py): X < 6p, Y <= 2p, ZZ, ZZZ < 1.1p; every product below has alpha * beta <= 40,
// far under HEADROOM = R / p >= 630, so every Montgomery product comes out < 1.07 p.
template <class C>
BPP_HD void xyzz_madd_lazy(Xyzz<C>& p, const Aff<C>& q, bool neg) {
    using P = typename C::Fp;
    using F = Fe<P>;
    uint32_t qz = 0;
#pragma unroll
    for (int i = 0; i < P::NL; i++) qz |= q.x.l[i] | q.y.l[i];
    if (qz == 0) return;   // q is the point at infinity
    if (p.is_inf()) {
        p.X = q.x;
        p.Y = neg ? fe_sub_nr<1>(F::zero(), q.y) : q.y;   // p - y <= p
        p.ZZ = F::one();
        p.ZZZ = F::one();
        return;
    }
    // first operands: the one that dies in the product where there is one; the others (the table entry, which the rare
    // doubling branch reads again, Pp and R) go through the _io forms (field.hpp) and are read back re-defined
    F qx = q.x, qy = q.y;
    const F U2 = fe_mul_io(qx, p.ZZ);               // < 1.01p
    const F S2 = fe_mul_io(qy, p.ZZZ);              // < 1.01p
    F Pp = fe_sub_nr<6>(U2, p.X);                   // U2 - X1 + 6p          in (0, 7.1p)
    F R = fe_csub_nr<4>(S2, neg, p.Y);              // +-S2 - Y1 + 4p        in (0.9p, 5.1p)
    const F PP = fe_sqr_io(Pp);                     // 7.1^2 / 630           < 1.09p
    const F PPP = fe_mul(Pp, PP);                   // < 1.02p
    const F Q = fe_mul(p.X, PP);                    // < 1.02p
    const F u = fe_add_dbl_nr(PPP, Q);              // PPP + 2Q              < 3.1p
    const F X3 = fe_sub_nr<4>(fe_sqr_io(R), u);     // R^2 - PPP - 2Q + 4p   in (0.9p, 5.1p)
    const F T = fe_sub_nr<6>(Q, X3);                // Q - X3 + 6p           in (0.9p, 7.1p)
    const F nY = fe_sub_nr<2>(F::zero(), p.Y);      // 2p - Y1               in [0, 2p]
    const F ZZ3 = fe_mul(p.ZZ, PP);
#if defined(BPP_LAZY_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    BPP_LAZY_CHECK(fe_below_kp<2>(U2) && fe_below_kp<2>(S2) && fe_below_kp<8>(Pp) && fe_below_kp<6>(R) &&
                   fe_below_kp<2>(PP) && fe_below_kp<2>(PPP) && fe_below_kp<2>(Q) && fe_below_kp<4>(u) &&
                   fe_below_kp<6>(X3) && fe_below_kp<8>(T) && fe_below_kp<3>(nY) && fe_below_kp<2>(ZZ3));
#endif
    if (ZZ3.is_zero()) {   // equal x: P + P or P + (-P)   (rare; the reference's case analysis, macros.rs:42-146)
        if (fe_is_zero_mod<5>(R)) {
            Aff<C> qq;
            qq.x = qx;
            qq.y = neg ? fe_sub_nr<1>(F::zero(), qy) : qy;
            p = xyzz_dbl_aff(qq);
        } else {
            p = xyzz_inf<C>();
        }
        return;
    }
    p.Y = fe_mul_add(R, T, nY, PPP);                // R (Q - X3) - Y1 PPP: (5.1 * 7.1 + 2 * 1.02) / 630 -> < 1.07p
    p.X = X3;
    p.ZZZ = fe_mul(p.ZZZ, PPP);
    p.ZZ = ZZ3;
}

// jacobian image with Z = ZZ: (X * ZZ, Y * ZZZ, ZZ)
template <class C>
BPP_HD Jac<C> xyzz_to_jac(const Xyzz<C>& p) {
    if (p.is_inf()) return jac_inf<C>();
    Jac<C> r;
    r.X = fe_mul(p.X, p.ZZ);
    r.Y = fe_mul(p.Y, p.ZZZ);
    r.Z = p.ZZ;
    return r;
}

// The verdict test of the verification MulVec: is the sum the identity of the group the PROOF lives in?  For the
// Weierstrass curves that is the point at infinity; the Edwards instantiation overrides it with the identity of
// ristretto255's quotient group (ristretto.hpp).
template <class C>
BPP_HD bool jac_is_identity_class(const Jac<C>& p) {
    return p.is_inf();
}

// Membership of the prime-order subgroup for points that arrive SERIALIZED (container.hpp / codec.hpp; the in-memory
// API takes points that are valid by construction, as the reference's mcl / BigUint values are).
//   BLS12-381 G1 (cofactor 0x396c8c005555e1568c00aaab0000aaab): the endomorphism phi(x, y) = (beta x, y) acts on G1 as
//   multiplication by -z^2 (z = -0xd201000000010000, r = z^4 - z^2 + 1), and a curve point lies in G1 iff
//   phi(P) = -[z^2] P (Scott, "A note on group membership tests for G1, G2 and GT on BLS pairing-friendly curves",
//   eprint 2021/1130): two multiplications by the sparse 64-bit |z| instead of one by the 255-bit r.  Checked against
//   [r] P == O on random, torsion and mixed points by tests/test_container_cpu.py.
//   secp256k1 has cofactor 1: every curve point is in the group.
template <class C>
BPP_HD bool aff_in_prime_subgroup(const Aff<C>& p) {
    if constexpr (C::ID == 0) {
        using F = Fe<typename C::Fp>;
        if (p.is_inf()) return true;
        Jac<C> t = jac_inf<C>();
        for (int i = 63; i >= 0; i--) {   // t <- [|z|] P: the base is affine, every addition a mixed one
            t = jac_dbl(t);
            if ((C::K::ZABS >> i) & 1ull) t = jac_madd(t, p);
        }
        {   // t <- [|z|] t
            const Jac<C> base = t;
            Jac<C> acc = jac_inf<C>();
            for (int i = 63; i >= 0; i--) {
                acc = jac_dbl(acc);
                if ((C::K::ZABS >> i) & 1ull) acc = jac_add(acc, base);
            }
            t = acc;
        }
        // phi(P) + [z^2] P == O  <=>  [z^2] P == (beta x, -y)
        if (t.is_inf()) return false;
        F beta;
#pragma unroll
        for (int i = 0; i < C::Fp::NL; i++) beta.l[i] = C::K::BETA[i];
        const F zz = fe_sqr(t.Z);
        if (fe_mul(fe_mul(beta, p.x), zz) != t.X) return false;
        return fe_mul(fe_neg(p.y), fe_mul(zz, t.Z)) == t.Y;
    } else {
        (void)p;
        return true;
    }
}

// GLV split of a BLS12-381 scalar for the curve's endomorphism: k = k1 + k2 z^2 with k1 = k mod z^2, k2 = k div z^2,
// both < 2^128 (z^2 has 128 bits, r / z^2 < 2^128), and [z^2] P = -phi(P) = (beta x, -y) on G1.
// k: 8 words, any value < 2^256 with k / z^2 < 2^128 (every canonical scalar).  The quotient starts from the Barrett
// estimate floor(floor(k / 2^128) MU / 2^128), MU = floor(2^256 / z^2): at most 2 short (tools/gen_constants.py checks
// the bound on 200 000 scalars; tests/host/glv_host_test.cpp checks the split against 128-bit integer arithmetic).
template <class C>
BPP_HD void glv_split(const uint32_t* k, uint32_t* k1, uint32_t* k2) {
    using K = typename C::K;
    uint32_t q[4];
    {   // (k[4..8) * MU[0..5)) >> 128: 4 x 5 words, column sums in 64 bits + a carry word
        uint64_t lo = 0;
        uint32_t hi = 0;
        uint32_t prod[9];
#pragma unroll
        for (int col = 0; col < 9; col++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int jx = col - i;
                if (jx < 0 || jx > 4) continue;
                const uint64_t pr = (uint64_t)k[4 + i] * K::GLV_MU[jx];
                lo += pr;
                hi += lo < pr ? 1u : 0u;
            }
            prod[col] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = prod[4 + i];
    }
    // rem = k - q z^2, low 160 bits (the true remainder is < 3 z^2 < 2^130)
    uint32_t rem[5];
    {
        uint32_t qz[5];
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int col = 0; col < 5; col++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int jx = col - i;
                if (jx < 0 || jx > 3) continue;
                const uint64_t pr = (uint64_t)q[i] * K::ZSQW[jx];
                lo += pr;
                hi += lo < pr ? 1u : 0u;
            }
            qz[col] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        uint32_t borrow = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const uint64_t d = (uint64_t)k[t] - qz[t] - borrow;
            rem[t] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
    }
    for (int it = 0; it < 3; it++) {   // at most two corrections
        uint32_t t5[5];
        uint32_t borrow = 0;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const uint64_t d = (uint64_t)rem[t] - (t < 4 ? K::ZSQW[t] : 0u) - borrow;
            t5[t] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
        if (!borrow) {   // rem >= z^2
#pragma unroll
            for (int t = 0; t < 5; t++) rem[t] = t5[t];
            uint32_t c = 1;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t x = (uint64_t)q[t] + c;
                q[t] = (uint32_t)x;
                c = (uint32_t)(x >> 32);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        k1[t] = rem[t];
        k2[t] = q[t];
    }
}

// Which curves split scalars with an endomorphism: BLS12-381 G1 ([z^2] P = (beta x, -y), above) and secp256k1
// ([lambda] P = (beta x, y)); the Edwards instantiation has none.
template <class C>
constexpr bool curve_has_glv() {
    return C::ID == 0 || C::ID == 1;
}
// the y-coordinate of the endomorphism image changes sign (BLS12-381: the image under [z^2] is -phi(P))
template <class C>
constexpr bool glv_image_negates_y() {
    return C::ID == 0;
}

// k = (+-k1) + (+-k2) * mu for the curve's endomorphism eigenvalue mu (z^2 on BLS12-381 G1, lambda on secp256k1):
// magnitudes k1, k2 < 2^128 (4 words each) and their signs.  k: 8 canonical words, < r.
// secp256k1 (Gallant-Lambert-Vanstone with the lattice basis libsecp256k1 documents; constants and bounds checked in
// tools/gen_constants.py, the function itself against Python integers by tests/host/glv_host_test.cpp):
//   c1 = round(k g1 / 2^384), c2 = round(k g2 / 2^384);  k2 = c1 (-b1) + c2 (-b2), k1 = k - k2 lambda  (mod n);
//   a half above n / 2 stands for its negative.
template <class C>
BPP_HD void glv_split_signed(const uint32_t* k, uint32_t* k1, uint32_t* k2, bool& neg1, bool& neg2) {
    if constexpr (C::ID == 0) {
        glv_split<C>(k, k1, k2);
        neg1 = false;
        neg2 = false;
    } else {
        using K = typename C::K;
        using P = typename C::Fr;
        using F = Fe<P>;
        static_assert(C::ID == 1, "secp256k1");
        auto mulshift384 = [&](const uint32_t* g, uint32_t* c) {   // c[0..8) = round(k g / 2^384) (< 2^128)
            uint64_t lo = 0;
            uint32_t hi = 0;
            uint32_t prod[16];
#pragma unroll
            for (int col = 0; col < 16; col++) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int jx = col - i;
                    if (jx < 0 || jx > 7) continue;
                    const uint64_t pr = (uint64_t)k[i] * g[jx];
                    lo += pr;
                    hi += lo < pr ? 1u : 0u;
                }
                prod[col] = (uint32_t)lo;
                lo = (lo >> 32) | ((uint64_t)hi << 32);
                hi = 0;
            }
            uint32_t carry = prod[11] >> 31;   // bit 383: round to nearest
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint64_t x = (uint64_t)(t < 4 ? prod[12 + t] : 0u) + carry;
                c[t] = (uint32_t)x;
                carry = (uint32_t)(x >> 32);
            }
        };
        auto konst = [](const uint32_t* limbs) {
            F r;
#pragma unroll
            for (int i = 0; i < P::NL; i++) r.l[i] = limbs[i];
            return r;
        };
        uint32_t c1[8], c2[8];
        mulshift384(K::GLV_G1W, c1);
        mulshift384(K::GLV_G2W, c2);
        const F f2 = fe_add(fe_mul(fe_from_canonical<P>(c1), konst(K::GLV_MB1)), fe_mul(fe_from_canonical<P>(c2), konst(K::GLV_MB2)));
        const F f1 = fe_sub(fe_from_canonical<P>(k), fe_mul(f2, konst(K::GLV_LAMBDA)));
        auto take = [](const F& f, uint32_t* out, bool& neg) {
            uint32_t w[8];
            fe_to_canonical(f, w);
            bool above = false;   // w > (n - 1) / 2 ?
            for (int t = 7; t >= 0; t--) {
                if (w[t] != P::HALFW[t]) {
                    above = w[t] > P::HALFW[t];
                    break;
                }
            }
            neg = above;
            if (above) {   // n - w
                uint32_t borrow = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint64_t d = (uint64_t)P::MODW[t] - w[t] - borrow;
                    w[t] = (uint32_t)d;
                    borrow = (uint32_t)(d >> 63);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) out[t] = w[t];
        };
        take(f1, k1, neg1);
        take(f2, k2, neg2);
    }
}

// Same group element?  (cross-multiplied comparison, no inversion)
template <class C>
BPP_HD bool jac_eq(const Jac<C>& p, const Jac<C>& q) {
    using F = Fe<typename C::Fp>;
    if (p.is_inf() || q.is_inf()) return p.is_inf() && q.is_inf();
    F Z1Z1 = fe_sqr(p.Z), Z2Z2 = fe_sqr(q.Z);
    if (fe_mul(p.X, Z2Z2) != fe_mul(q.X, Z1Z1)) return false;
    return fe_mul(fe_mul(p.Y, q.Z), Z2Z2) == fe_mul(fe_mul(q.Y, p.Z), Z1Z1);
}

// ---- memory images ---------------------------------------------------------------------------------
template <class C>
BPP_HD Aff<C> aff_load(const uint32_t* w) {
    Aff<C> r;
    r.x = fe_load<typename C::Fp>(w);
    r.y = fe_load<typename C::Fp>(w + C::Fp::N);
    return r;
}
template <class C>
BPP_HD void aff_store(const Aff<C>& p, uint32_t* w) {
    fe_store(p.x, w);
    fe_store(p.y, w + C::Fp::N);
}
template <class C>
BPP_HD Jac<C> jac_load(const uint32_t* w) {
    Jac<C> r;
    r.X = fe_load<typename C::Fp>(w);
    r.Y = fe_load<typename C::Fp>(w + C::Fp::N);
    r.Z = fe_load<typename C::Fp>(w + 2 * C::Fp::N);
    return r;
}
template <class C>
BPP_HD void jac_store(const Jac<C>& p, uint32_t* w) {
    fe_store(p.X, w);
    fe_store(p.Y, w + C::Fp::N);
    fe_store(p.Z, w + 2 * C::Fp::N);
}

// wire (canonical x | y | inf:u64) -> affine.  Returns false when a coordinate is >= p or the point is
// not on the curve (the reference has no such check: mcl/BigUint values are valid by construction).
template <class C>
BPP_HD bool aff_from_wire(const uint32_t* w, Aff<C>& out) {
    constexpr int N = C::Fp::N;
    if (w[2 * N] | w[2 * N + 1]) {
        out = aff_inf<C>();
        return true;
    }
    if (!words_lt_mod<typename C::Fp>(w) || !words_lt_mod<typename C::Fp>(w + N)) return false;
    out.x = fe_from_canonical<typename C::Fp>(w);
    out.y = fe_from_canonical<typename C::Fp>(w + N);
    if (out.is_inf()) return false;  // (0,0) is not a curve point
    return aff_on_curve(out);
}
template <class C>
BPP_HD void aff_to_wire(const Aff<C>& p, uint32_t* w) {
    constexpr int N = C::Fp::N;
    if (p.is_inf()) {
#pragma unroll
        for (int i = 0; i < 2 * N + 2; i++) w[i] = 0;
        w[2 * N] = 1;
        return;
    }
    fe_to_canonical(p.x, w);
    fe_to_canonical(p.y, w + N);
    w[2 * N] = 0;
    w[2 * N + 1] = 0;
}

// scalar * point, MSB-first double-and-add over `nbits` bits of the canonical scalar words.
// Same group element as the reference's LSB-first loop (macros.rs:9-27) / mcl's G1::mul.
template <class C>
BPP_HD Jac<C> aff_mul_words(const Aff<C>& p, const uint32_t* k, int nwords) {
    Jac<C> acc = jac_inf<C>();
    int top = nwords * 32 - 1;
    while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1u)) top--;
    for (int i = top; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = jac_madd(acc, p);
    }
    return acc;
}

}  // namespace bpp
