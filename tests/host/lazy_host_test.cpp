// Host-side check of the lazy (unreduced) XYZZ mixed addition of csrc/ec.hpp against the eager one
// (compiled with g++, no GPU needed): random walks of acc += +-q over multiples of the generator, with the
// exceptional cases forced (q = infinity, acc = infinity, P + P, P + (-P)), the bound of every intermediate
// asserted inside the formula (BPP_LAZY_CHECK) and the accumulator invariants asserted after every step.
// Prints "ok <steps>" per curve; exits non-zero on the first mismatch.
#include <cstdio>
#include <cstdlib>
static int g_bound_failures = 0;
#define BPP_LAZY_CHECK(cond) do { if (!(cond)) { g_bound_failures++; fprintf(stderr, "bound violated: %s\n", #cond); } } while (0)
#include "../../bulletproofsplus_amd/csrc/ec.hpp"
using namespace bpp;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t z = rng_state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// canonical affine image (x, y < p), as the tables hold it
template <class C> static Aff<C> canon(const Aff<C>& a) {
    Aff<C> r = a;
    fe_cond_sub_p(r.x);
    fe_cond_sub_p(r.y);
    return r;
}

template <class C> static int run(const char* name, int steps) {
    const Aff<C> g = aff_generator<C>();
    constexpr int NP = 24;
    Aff<C> pts[NP];
    for (int i = 0; i < NP; i++) {
        uint32_t k[8];
        for (int t = 0; t < 8; t++) k[t] = (uint32_t)rnd();
        k[7] &= 0x0fffffffu;
        pts[i] = canon(jac_to_aff(aff_mul_words(g, k, 8)));
    }
    Xyzz<C> lazy = xyzz_inf<C>(), eager = xyzz_inf<C>();
    int extremes = 0;
    Aff<C> last = pts[0];
    bool last_neg = false;
    for (int s = 0; s < steps; s++) {
        Aff<C> q = pts[rnd() % NP];
        bool neg = rnd() & 1;
        const unsigned what = (unsigned)(rnd() % 16);
        if (what == 0) q = aff_inf<C>();                         // q = infinity
        if (what == 1 && !eager.is_inf()) {                      // P + P: q = the affine image of acc
            q = canon(jac_to_aff(xyzz_to_jac(eager)));
            neg = false;
        }
        if (what == 2 && !eager.is_inf()) {                      // P + (-P)
            q = canon(jac_to_aff(xyzz_to_jac(eager)));
            neg = true;
        }
        if (what == 3) { q = last; neg = last_neg; }             // the same point twice in a row
        if (what == 4) { q = last; neg = !last_neg; }            // ... and its opposite
        last = q;
        last_neg = neg;
        if (s % 8 == 5 && !eager.is_inf()) {
            // the accumulator AT the stated upper bounds of the invariant (ec.hpp xyzz_madd_lazy): the same group element
            // with X = x + 5p (< 6p), Y = y + p (< 2p) and, where the canonical value is small enough, ZZ / ZZZ + p (< 1.1p)
            using P = typename C::Fp;
            using F = Fe<P>;
            auto canon_fe = [](F v) { fe_cond_sub_p(v); fe_cond_sub_p(v); return v; };
            auto small = [](const F& v) { return (uint64_t)v.l[P::NL - 1] * 10 + 10 < (uint64_t)P::MOD[P::NL - 1]; };
            Xyzz<C> hi;
            hi.X = fe_sub_nr<5>(canon_fe(eager.X), F::zero());
            hi.Y = fe_sub_nr<1>(canon_fe(eager.Y), F::zero());
            const F zz = canon_fe(eager.ZZ), zzz = canon_fe(eager.ZZZ);
            hi.ZZ = small(zz) && !zz.is_zero() ? fe_sub_nr<1>(zz, F::zero()) : zz;
            hi.ZZZ = small(zzz) && !zzz.is_zero() ? fe_sub_nr<1>(zzz, F::zero()) : zzz;
            lazy = hi;
            extremes++;
        }
        xyzz_madd_lazy(lazy, q, neg);
        eager = xyzz_madd(eager, neg ? canon(aff_neg(q)) : q);
        if (!jac_eq(xyzz_to_jac(lazy), xyzz_to_jac(eager))) {
            fprintf(stderr, "%s: mismatch at step %d (case %u)\n", name, s, what);
            return 1;
        }
        using P = typename C::Fp;
        if (!lazy.is_inf() && !(fe_below_kp<6>(lazy.X) && (fe_below_kp<2>(lazy.Y) || lazy.Y == Fe<P>::zero()) &&
                                fe_below_kp<2>(lazy.ZZ) && fe_below_kp<2>(lazy.ZZZ))) {
            fprintf(stderr, "%s: accumulator invariant violated at step %d\n", name, s);
            return 1;
        }
    }
    if (g_bound_failures) return 1;
    printf("ok %s %d (%d steps from an accumulator at its upper bounds)\n", name, steps, extremes);
    return 0;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4000;
    int rc = run<Bls12381>("bls12_381", steps);
    rc |= run<Secp256k1>("secp256k1", steps);
    return rc;
}
