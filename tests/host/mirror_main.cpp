// The reference's demo driver (src/main.rs:6-56) written against the C++ mirror (include/bpp_amd.hpp):
// n = 64, m = 2, v = {2, 5}, gamma = {3, 7}; prove, verify == Ok(()); then tampering -> VerificationError.
// Prints r', s', delta' (hex) so that the caller can compare them with the golden fixture.
#include <cstdio>
#include "../../include/bpp_amd.hpp"
using namespace bpp;

static void print_scalar(const char* name, const PrimeFieldElem& x) {
    printf("%s=%016llx%016llx%016llx%016llx\n", name, (unsigned long long)x.e[3], (unsigned long long)x.e[2],
           (unsigned long long)x.e[1], (unsigned long long)x.e[0]);
}

int main() {
    Arith::init();
    const size_t n = 64, m = 2;
    PublicKey pk = PublicKey::create(n * m);
    RangeProver prover;
    prover.commit(pk, 2, PrimeFieldElem(3));
    prover.commit(pk, 5, PrimeFieldElem(7));
    RangeProof proof = RangeProof::prove(pk, n, prover);
    print_scalar("r_prime", proof.proof.r_prime);
    print_scalar("s_prime", proof.proof.s_prime);
    print_scalar("d_prime", proof.proof.d_prime);
    auto result = proof.verify(pk, n, prover.commitment_vec);
    printf("verify=%s\n", result ? "Err(VerificationError)" : "Ok(())");
    RangeProof bad = proof;
    bad.proof.d_prime.e[0] ^= 1;
    RangeVerifier verifier;                        // the README's calling convention (README.md:47-55)
    verifier.allocate(prover.commitment_vec);
    auto r2 = bad.verify(pk, n, verifier);
    printf("tampered=%s\n", r2 ? "Err(VerificationError)" : "Ok(())");
    // MulVec panics on a length mismatch (mulvec.rs:23-25)
    MulVec mv;
    mv.add_scalar(PrimeFieldElem(1));
    bool threw = false;
    try { mv.calculate(); } catch (const std::logic_error&) { threw = true; }
    printf("mulvec_mismatch=%s\n", threw ? "panic" : "no-panic");
    // g * 2 == h, via MulVec
    MulVec mv2;
    mv2.add_scalar(PrimeFieldElem(2));
    mv2.add_point(pk.g);
    printf("two_g_is_h=%d\n", (int)(mv2.calculate() == pk.h));
    return (!result && r2 && threw) ? 0 : 1;
}
