// ed25519.hpp -- third instantiation of the engine's group layer: the twisted Edwards curve
// -x^2 + y^2 = 1 + d x^2 y^2 over F_(2^255 - 19) (edwards25519, the curve under Ristretto255), restricted to
// its prime-order subgroup (every point the protocol forms is a multiple of the base point).
//
// PARITY STATUS: the reference has NO curve25519 / Ristretto backend (only a stale README
// example and comment remnants, src/weighted_inner_product_proof.rs:21), so nothing here can be pinned by
// reference code or vectors -- "parity unpinned".  It is pinned by group-law identities, RFC 8032's base
// point, and the dlog-shadow / big-integer restatement of the protocol over this group (oracle/pyref.py).
//
// Representation: affine (x, y) with the identity (0, 1); projective = extended coordinates (X, Y, Z, T),
// x = X/Z, y = Y/Z, T = XY/Z.  The unified addition (add-2008-hwcd-3, a = -1, d non-square) is complete:
// no special cases for equal / opposite / identity operands.  The templates of ec.hpp are specialised under
// the same names (Aff, Jac, Xyzz, jac_add, ...), so every kernel compiles unchanged for this curve; "Jac"
// and "Xyzz" both mean "extended" here.
#pragma once
#include "ec.hpp"

namespace bpp {

struct Ed25519 {
    using Fp = EdFp;
    using Fr = EdFr;
    using K = Ed25519Consts;
    static constexpr int ID = 2;
};

template <>
constexpr int jac_words<Ed25519>() {
    return 4 * EdFp::N;
}

template <>
struct Aff<Ed25519> {
    Fe<EdFp> x, y;
    BPP_HD bool is_inf() const { return x.is_zero() && y == Fe<EdFp>::one(); }
};
template <>
struct Jac<Ed25519> {
    Fe<EdFp> X, Y, Z, T;
    BPP_HD bool is_inf() const { return X.is_zero() && Y == Z; }
};
template <>
struct Xyzz<Ed25519> {
    Jac<Ed25519> e;
    BPP_HD bool is_inf() const { return e.is_inf(); }
};

namespace ed {
using F = Fe<EdFp>;
BPP_HD F konst(const uint32_t* c) {
    F r;
#pragma unroll
    for (int i = 0; i < EdFp::NL; i++) r.l[i] = c[i];
    return r;
}
BPP_HD F select(bool c, const F& a, const F& b) {
    F r;
#pragma unroll
    for (int i = 0; i < EdFp::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
}  // namespace ed

template <>
BPP_HD Aff<Ed25519> aff_inf<Ed25519>() {
    Aff<Ed25519> r;
    r.x = ed::F::zero();
    r.y = ed::F::one();
    return r;
}
template <>
BPP_HD Jac<Ed25519> jac_inf<Ed25519>() {
    Jac<Ed25519> r;
    r.X = ed::F::zero();
    r.Y = ed::F::one();
    r.Z = ed::F::one();
    r.T = ed::F::zero();
    return r;
}
template <>
BPP_HD Xyzz<Ed25519> xyzz_inf<Ed25519>() {
    Xyzz<Ed25519> r;
    r.e = jac_inf<Ed25519>();
    return r;
}
template <>
BPP_HD Aff<Ed25519> aff_generator<Ed25519>() {
    Aff<Ed25519> g;
    g.x = ed::konst(Ed25519Consts::GX);
    g.y = ed::konst(Ed25519Consts::GY);
    return g;
}

BPP_HD Jac<Ed25519> jac_from_aff(const Aff<Ed25519>& p) {
    Jac<Ed25519> r;
    r.X = p.x;
    r.Y = p.y;
    r.Z = ed::F::one();
    r.T = fe_mul(p.x, p.y);
    return r;
}
BPP_HD Aff<Ed25519> aff_neg(const Aff<Ed25519>& p) {
    Aff<Ed25519> r;
    r.x = fe_neg(p.x);
    r.y = p.y;
    return r;
}
BPP_HD Jac<Ed25519> jac_neg(const Jac<Ed25519>& p) {
    Jac<Ed25519> r = p;
    r.X = fe_neg(p.X);
    r.T = fe_neg(p.T);
    return r;
}
// -x^2 + y^2 == 1 + d x^2 y^2
BPP_HD bool aff_on_curve(const Aff<Ed25519>& p) {
    const ed::F xx = fe_sqr(p.x), yy = fe_sqr(p.y);
    const ed::F lhs = fe_sub(yy, xx);
    const ed::F rhs = fe_add(ed::F::one(), fe_mul(ed::konst(Ed25519Consts::D), fe_mul(xx, yy)));
    return lhs == rhs;
}
// dbl-2008-hwcd (a = -1): 4M + 4S
BPP_HD Jac<Ed25519> jac_dbl(const Jac<Ed25519>& p) {
    const ed::F A = fe_sqr(p.X), B = fe_sqr(p.Y), C = fe_dbl(fe_sqr(p.Z));
    const ed::F D = fe_neg(A);
    const ed::F E = fe_sub(fe_sub(fe_sqr(fe_add(p.X, p.Y)), A), B);
    const ed::F G = fe_add(D, B), Fq = fe_sub(G, C), H = fe_sub(D, B);
    Jac<Ed25519> r;
    r.X = fe_mul(E, Fq);
    r.Y = fe_mul(G, H);
    r.T = fe_mul(E, H);
    r.Z = fe_mul(Fq, G);
    return r;
}
BPP_HD Jac<Ed25519> aff_dbl(const Aff<Ed25519>& p) { return jac_dbl(jac_from_aff(p)); }

// add-2008-hwcd-3 (a = -1), unified and complete: 9M
BPP_HD Jac<Ed25519> jac_add(const Jac<Ed25519>& p, const Jac<Ed25519>& q) {
    const ed::F A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const ed::F B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const ed::F C = fe_mul(fe_mul(p.T, ed::konst(Ed25519Consts::D2)), q.T);
    const ed::F D = fe_dbl(fe_mul(p.Z, q.Z));
    const ed::F E = fe_sub(B, A), Fq = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    Jac<Ed25519> r;
    r.X = fe_mul(E, Fq);
    r.Y = fe_mul(G, H);
    r.T = fe_mul(E, H);
    r.Z = fe_mul(Fq, G);
    return r;
}
// mixed addition with an affine point (Z2 = 1, T2 = x2 y2): 9M, complete
BPP_HD Jac<Ed25519> jac_madd(const Jac<Ed25519>& p, const Aff<Ed25519>& q) {
    const ed::F A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.y, q.x));
    const ed::F B = fe_mul(fe_add(p.Y, p.X), fe_add(q.y, q.x));
    const ed::F C = fe_mul(fe_mul(p.T, ed::konst(Ed25519Consts::D2)), fe_mul(q.x, q.y));
    const ed::F D = fe_dbl(p.Z);
    const ed::F E = fe_sub(B, A), Fq = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    Jac<Ed25519> r;
    r.X = fe_mul(E, Fq);
    r.Y = fe_mul(G, H);
    r.T = fe_mul(E, H);
    r.Z = fe_mul(Fq, G);
    return r;
}
BPP_HD Aff<Ed25519> jac_to_aff(const Jac<Ed25519>& p) {
    const ed::F zi = fe_inv(p.Z);
    Aff<Ed25519> r;
    r.x = fe_mul(p.X, zi);
    r.y = fe_mul(p.Y, zi);
    return r;
}
BPP_HD Aff<Ed25519> jac_scale_to_aff(const Jac<Ed25519>& p, const ed::F& zi) {
    Aff<Ed25519> r;
    r.x = fe_mul(p.X, zi);
    r.y = fe_mul(p.Y, zi);
    return r;
}
BPP_HD bool jac_eq(const Jac<Ed25519>& p, const Jac<Ed25519>& q) {
    return fe_mul(p.X, q.Z) == fe_mul(q.X, p.Z) && fe_mul(p.Y, q.Z) == fe_mul(q.Y, p.Z);
}
BPP_HD Xyzz<Ed25519> xyzz_dbl_aff(const Aff<Ed25519>& q) {
    Xyzz<Ed25519> r;
    r.e = aff_dbl(q);
    return r;
}
BPP_HD Xyzz<Ed25519> xyzz_madd(const Xyzz<Ed25519>& p, const Aff<Ed25519>& q) {
    Xyzz<Ed25519> r;
    r.e = jac_madd(p.e, q);
    return r;
}
// acc += (neg ? -q : q): the unified addition above (complete: no exceptional cases to defer) with its eight sums and
// differences left UNREDUCED -- R / p = 2^15 for this field, so nothing below comes near the bound of a Montgomery
// product -- and the sign of q folded in without a negation: -q = (-x, y) makes y - x and y + x change places and flips
// the sign of x y, i.e. F = D - C and G = D + C change places.  Inputs: coordinates of p below 1.01 p (products),
// q canonical.  Multiples of p each value stays below are in the comments; every product has alpha * beta <= 14.
BPP_HD void xyzz_madd_lazy(Xyzz<Ed25519>& p, const Aff<Ed25519>& q, bool neg) {
    using F = ed::F;
    Jac<Ed25519>& e = p.e;
    const F qm = fe_sub_nr<1>(q.y, q.x);       // y2 - x2 + p            (0, 2p)
    const F qp = fe_add_nr(q.y, q.x);          //                        < 2p
    const F a1 = fe_sub_nr<2>(e.Y, e.X);       // Y1 - X1 + 2p           (0.9p, 3.1p)
    const F b1 = fe_add_nr(e.Y, e.X);          //                        < 2.1p
    const F A = fe_mul(a1, ed::select(neg, qp, qm));
    const F B = fe_mul(b1, ed::select(neg, qm, qp));
    const F C = fe_mul(fe_mul(e.T, ed::konst(Ed25519Consts::D2)), fe_mul(q.x, q.y));   // +- C: the sign is in F / G below
    const F D = fe_add_nr(e.Z, e.Z);           // 2 Z1                   < 2.1p
    F E = fe_sub_nr<2>(B, A);                  // B - A + 2p             < 3.1p
    const F H = fe_add_nr(B, A);               //                        < 2.1p
    const F Fm = fe_sub_nr<2>(D, C);           // D - C + 2p             < 4.2p
    const F Gp = fe_add_nr(D, C);              //                        < 3.2p
    F Fq = ed::select(neg, Gp, Fm);
    F G = ed::select(neg, Fm, Gp);
    e.X = fe_mul_io(E, Fq);                    // E and G are read again: handed back re-defined (field.hpp)
    e.T = fe_mul(E, H);
    e.Y = fe_mul_io(G, H);
    e.Z = fe_mul(Fq, G);
}
BPP_HD Jac<Ed25519> xyzz_to_jac(const Xyzz<Ed25519>& p) { return p.e; }

// ---- memory images: affine x | y (2N words) ; extended X | Y | Z | T (4N words) --------------------------
template <>
BPP_HD Jac<Ed25519> jac_load<Ed25519>(const uint32_t* w) {
    constexpr int N = EdFp::N;
    Jac<Ed25519> r;
    r.X = fe_load<EdFp>(w);
    r.Y = fe_load<EdFp>(w + N);
    r.Z = fe_load<EdFp>(w + 2 * N);
    r.T = fe_load<EdFp>(w + 3 * N);
    return r;
}
BPP_HD void jac_store(const Jac<Ed25519>& p, uint32_t* w) {
    constexpr int N = EdFp::N;
    fe_store(p.X, w);
    fe_store(p.Y, w + N);
    fe_store(p.Z, w + 2 * N);
    fe_store(p.T, w + 3 * N);
}

// wire (canonical x | y | inf:u64) -> affine.  inf = 1 is the identity (0, 1).
template <>
BPP_HD bool aff_from_wire<Ed25519>(const uint32_t* w, Aff<Ed25519>& out) {
    constexpr int N = EdFp::N;
    if (w[2 * N] | w[2 * N + 1]) {
        out = aff_inf<Ed25519>();
        return true;
    }
    if (!words_lt_mod<EdFp>(w) || !words_lt_mod<EdFp>(w + N)) return false;
    out.x = fe_from_canonical<EdFp>(w);
    out.y = fe_from_canonical<EdFp>(w + N);
    return aff_on_curve(out);
}

}  // namespace bpp
