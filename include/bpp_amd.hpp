// bpp_amd.hpp -- C++ host-side mirror of the reference crate's API for the hot path, over the C ABI of
// bpp_amd.h (header only; link with -lbpp_amd).  The reference is Rust; no Rust toolchain exists in the
// build image, so the host side above the C ABI is C++ with the reference's names, argument meaning and
// error behaviour (paths relative to /root/reference/src):
//
//   bpp::Arith::init()                         bls12_381/building_block/arith.rs:6-19
//   bpp::PrimeFieldElem                        bls12_381/building_block/scalar/prime_field_elem.rs:13-15 (as data)
//   bpp::Point                                 bls12_381/building_block/point/point.rs:12 (as data, wire format)
//   bpp::MulVec                                bls12_381/building_block/mulvec.rs:7-53
//   bpp::PublicKey{g,h,G_vec,H_vec}            publickey.rs:13-52
//   bpp::RangeProver{v_vec,gamma_vec,commitment_vec}   range/prover.rs:13-42
//   bpp::RangeVerifier{commitment_vec}.allocate()       README.md:47-48 (no code in the reference)
//   bpp::WeightedInnerProductProof             weighted_inner_product_proof.rs:25-33
//   bpp::RangeProof{A, proof}::prove / verify  range/mod.rs:25-78
//   bpp::ProofError::VerificationError         errors.rs:14-50
//
// Where the reference panics (assert!/panic!) these classes throw std::logic_error; `verify` returns
// Result-like `std::optional<ProofError>` (nullopt = Ok(())).  Every operation runs HIP kernels through
// libbpp_amd.so; nothing here computes on the CPU beyond (de)serialisation.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "bpp_amd.h"

namespace bpp {

enum class ProofError { VerificationError };

// One process-wide context per curve, as the reference's global Once-guarded init.
class Arith {
  public:
    static void init(int curve_id = BPP_BLS12_381_G1, int device = 0) {
        std::lock_guard<std::mutex> lk(mu());
        if (ctx_ref()) return;
        bpp_ctx* c = nullptr;
        if (bpp_init(curve_id, device, &c) != BPP_OK)
            throw std::runtime_error(std::string("Initializing bpp_amd failed: ") + bpp_last_error());  // arith.rs:13-15
        ctx_ref() = c;
        curve_ref() = curve_id;
    }
    static bpp_ctx* ctx() {
        if (!ctx_ref()) init();
        return ctx_ref();
    }
    static int curve() { return curve_ref(); }
    static size_t point_words() { return (size_t)bpp_point_words(curve_ref()); }

  private:
    static bpp_ctx*& ctx_ref() {
        static bpp_ctx* c = nullptr;
        return c;
    }
    static int& curve_ref() {
        static int c = BPP_BLS12_381_G1;
        return c;
    }
    static std::mutex& mu() {
        static std::mutex m;
        return m;
    }
};

// canonical value < r as 4 little-endian u64 limbs
struct PrimeFieldElem {
    std::array<uint64_t, 4> e{};
    PrimeFieldElem() = default;
    // PrimeFieldElem::new(i32) for n >= 0 (negative values need the group order: use from_limbs)
    explicit PrimeFieldElem(uint32_t n) { e[0] = n; }
    static PrimeFieldElem from_limbs(const uint64_t* l) {
        PrimeFieldElem x;
        std::memcpy(x.e.data(), l, 32);
        return x;
    }
    bool operator==(const PrimeFieldElem& o) const { return e == o.e; }
};

// affine wire point: x | y | inf  (2L+1 u64 words)
struct Point {
    std::vector<uint64_t> w;
    Point() : w(Arith::point_words(), 0) { w.back() = 1; }   // Point::zero()
    explicit Point(const uint64_t* src) : w(src, src + Arith::point_words()) {}
    static Point zero() { return Point(); }
    bool is_zero() const { return w.back() != 0; }
    bool operator==(const Point& o) const { return w == o.w; }
};

class MulVec {
  public:
    void add_scalar(const PrimeFieldElem& s) { scalars_.insert(scalars_.end(), s.e.begin(), s.e.end()); n_s_++; }
    void add_scalars(const std::vector<PrimeFieldElem>& ss) { for (auto& s : ss) add_scalar(s); }
    void add_point(const Point& p) { points_.insert(points_.end(), p.w.begin(), p.w.end()); n_p_++; }
    void add_points(const std::vector<Point>& ps) { for (auto& p : ps) add_point(p); }
    Point calculate() const {
        if (n_s_ != n_p_) throw std::logic_error("mulvec: lengths of scalars and points must match");  // mulvec.rs:23-25
        Point out;
        int rc = bpp_msm(Arith::ctx(), scalars_.data(), points_.data(), n_s_, out.w.data());
        if (rc != BPP_OK) throw std::runtime_error(std::string("bpp_msm: ") + bpp_last_error());
        return out;
    }

  private:
    std::vector<uint64_t> scalars_, points_;
    size_t n_s_ = 0, n_p_ = 0;
};

struct PublicKey {
    Point g, h;
    std::vector<Point> G_vec, H_vec;
    static PublicKey create(size_t length) {   // PublicKey::new(length), publickey.rs:21-48
        const size_t pw = Arith::point_words();
        std::vector<uint64_t> gh(2 * pw), G(length * pw + 1), H(length * pw + 1);
        if (bpp_pk_new(Arith::ctx(), length, gh.data(), G.data(), H.data()) != BPP_OK)
            throw std::runtime_error(std::string("bpp_pk_new: ") + bpp_last_error());
        PublicKey pk;
        pk.g = Point(gh.data());
        pk.h = Point(gh.data() + pw);
        for (size_t i = 0; i < length; i++) {
            pk.G_vec.emplace_back(G.data() + i * pw);
            pk.H_vec.emplace_back(H.data() + i * pw);
        }
        return pk;
    }
    Point commitment(const PrimeFieldElem& v, const PrimeFieldElem& gamma) const {   // publickey.rs:50-52
        MulVec mv;
        mv.add_scalar(v);
        mv.add_scalar(gamma);
        mv.add_point(g);
        mv.add_point(h);
        return mv.calculate();
    }
    std::vector<uint64_t> gh_wire() const {
        std::vector<uint64_t> o(g.w);
        o.insert(o.end(), h.w.begin(), h.w.end());
        return o;
    }
    static std::vector<uint64_t> flat(const std::vector<Point>& v) {
        std::vector<uint64_t> o;
        for (auto& p : v) o.insert(o.end(), p.w.begin(), p.w.end());
        return o;
    }
};

struct RangeProver {
    std::vector<uint64_t> v_vec;
    std::vector<PrimeFieldElem> gamma_vec;
    std::vector<Point> commitment_vec;
    void commit(const PublicKey& pk, uint64_t v, const PrimeFieldElem& gamma) {   // range/prover.rs:28-42
        Point out;
        auto gh = pk.gh_wire();
        if (bpp_commit(Arith::ctx(), gh.data(), v, gamma.e.data(), out.w.data()) != BPP_OK)
            throw std::runtime_error(std::string("bpp_commit: ") + bpp_last_error());
        v_vec.push_back(v);
        gamma_vec.push_back(gamma);
        commitment_vec.push_back(out);
    }
};

// Verifier-side holder of the commitments.  It exists only in the reference's (stale) README
// (README.md:47-55: RangeVerifier::new(), allocate(&prover.commitment_vec), proof.verify(.., &verifier));
// the reference's code takes the commitment slice directly (range/mod.rs:57-62).  Both forms are offered.
struct RangeVerifier {
    std::vector<Point> commitment_vec;
    void allocate(const std::vector<Point>& commitments) { commitment_vec = commitments; }
};

struct WeightedInnerProductProof {
    std::vector<Point> L_vec, R_vec;
    Point A, B;
    PrimeFieldElem r_prime, s_prime, d_prime;
};

struct RangeProof {
    Point A;
    WeightedInnerProductProof proof;

    static RangeProof prove(const PublicKey& pk, size_t n, const RangeProver& prover) {   // range/mod.rs:31-55
        const size_t m = prover.v_vec.size(), mn = n * m, pw = Arith::point_words();
        if (m == 0 || (mn & (mn - 1))) throw std::logic_error("n * m must be a power of two");        // wip.rs:67
        if (pk.G_vec.size() != mn || pk.H_vec.size() != mn)
            throw std::logic_error("assertion failed: pk.G_vec.len() == n * m");                       // range/mod.rs:90-91,252-253
        size_t k = 0;
        while (((size_t)1 << k) < mn) k++;
        auto gh = pk.gh_wire(), G = PublicKey::flat(pk.G_vec), H = PublicKey::flat(pk.H_vec);
        auto V = PublicKey::flat(prover.commitment_vec);
        std::vector<uint64_t> gm;
        for (auto& g : prover.gamma_vec) gm.insert(gm.end(), g.e.begin(), g.e.end());
        std::vector<uint64_t> pts((3 + 2 * k) * pw), sc(12);
        if (bpp_range_prove(Arith::ctx(), gh.data(), G.data(), H.data(), n, m, prover.v_vec.data(), gm.data(), V.data(),
                            pts.data(), sc.data()) != BPP_OK)
            throw std::runtime_error(std::string("bpp_range_prove: ") + bpp_last_error());
        RangeProof rp;
        rp.A = Point(pts.data());
        rp.proof.A = Point(pts.data() + pw);
        rp.proof.B = Point(pts.data() + 2 * pw);
        for (size_t i = 0; i < k; i++) {
            rp.proof.L_vec.emplace_back(pts.data() + (3 + i) * pw);
            rp.proof.R_vec.emplace_back(pts.data() + (3 + k + i) * pw);
        }
        rp.proof.r_prime = PrimeFieldElem::from_limbs(sc.data());
        rp.proof.s_prime = PrimeFieldElem::from_limbs(sc.data() + 4);
        rp.proof.d_prime = PrimeFieldElem::from_limbs(sc.data() + 8);
        return rp;
    }

    // Ok(()) -> std::nullopt ; Err(ProofError::VerificationError) -> the error   (range/mod.rs:57-78)
    std::optional<ProofError> verify(const PublicKey& pk, size_t n, const std::vector<Point>& commitment_vec) const {
        const size_t m = commitment_vec.size(), k = proof.L_vec.size();
        auto gh = pk.gh_wire(), G = PublicKey::flat(pk.G_vec), H = PublicKey::flat(pk.H_vec);
        auto V = PublicKey::flat(commitment_vec);
        std::vector<uint64_t> pts(A.w);
        pts.insert(pts.end(), proof.A.w.begin(), proof.A.w.end());
        pts.insert(pts.end(), proof.B.w.begin(), proof.B.w.end());
        for (auto& p : proof.L_vec) pts.insert(pts.end(), p.w.begin(), p.w.end());
        for (auto& p : proof.R_vec) pts.insert(pts.end(), p.w.begin(), p.w.end());
        uint64_t sc[12];
        std::memcpy(sc, proof.r_prime.e.data(), 32);
        std::memcpy(sc + 4, proof.s_prime.e.data(), 32);
        std::memcpy(sc + 8, proof.d_prime.e.data(), 32);
        int rc = bpp_range_verify(Arith::ctx(), gh.data(), G.data(), H.data(), n, m, pts.data(), k, sc, V.data());
        if (rc == BPP_OK) return std::nullopt;
        if (rc == BPP_VERIFICATION_ERROR) return ProofError::VerificationError;
        throw std::runtime_error(std::string("bpp_range_verify: ") + bpp_last_error());
    }
    std::optional<ProofError> verify(const PublicKey& pk, size_t n, const RangeVerifier& verifier) const {
        return verify(pk, n, verifier.commitment_vec);
    }
};

}  // namespace bpp
