// prover.hpp -- RangeProof::prove on the MI355X: host-side scalar bookkeeping in C++ (the reference's is
// Rust), every group operation on the GPU.
//
// Follows reference src/range/mod.rs:80-187 (prove_single), :240-403 (prove_multiple) and
// src/weighted_inner_product_proof.rs:36-227 (WeightedInnerProductProof::prove) with the hard-coded
// "transcript" constants of SURVEY.md 3.4.  Every MulVec of the reference becomes a device MulVec
// (k_msm_naive_partial), the per-element fold of wip.rs:147-164 becomes one k_fold_points launch per round.
// Values the reference computes but never reads are not computed: A_hat (range/mod.rs:153,:343 -- it only
// feeds `P`) and P itself (wip.rs:137-142).
#pragma once
#include "impl_msm.hpp"

namespace bpp {

// One WIP folding round on the generator vectors (wip.rs:151-163), in place:
//   G[i] <- sG1 * G[i] + sG2 * G[i + nh]      (MulVec [e^-1, y^-nh e] . [G1_i, G2_i])
//   H[i] <- sH1 * H[i] + sH2 * H[i + nh]      (MulVec [e, e^-1] . [H1_i, H2_i])
// threads [0, nh) fold G, threads [nh, 2 nh) fold H.  sc = [sG1, sG2, sH1, sH2] canonical.
template <class C>
__global__ void __launch_bounds__(64) k_fold_points(uint32_t* __restrict__ G, uint32_t* __restrict__ H, uint32_t nh,
                              const uint32_t* __restrict__ sc) {
    constexpr int N = C::Fp::N;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * nh) return;
    const bool isH = t >= nh;
    const uint32_t i = isH ? t - nh : t;
    uint32_t* V = isH ? H : G;
    uint32_t k1[8], k2[8];
    ld_words<8>(sc + (isH ? 16 : 0), k1);
    ld_words<8>(sc + (isH ? 24 : 8), k2);
    Aff<C> p1 = aff_ldg<C>(V + (size_t)i * 2 * N);
    Aff<C> p2 = aff_ldg<C>(V + (size_t)(i + nh) * 2 * N);
    Jac<C> r = jac_add(aff_mul_words(p1, k1, 8), aff_mul_words(p2, k2, 8));
    aff_stg<C>(V + (size_t)i * 2 * N, jac_to_aff(r));
}

template <class C>
struct HostFr {
    using P = typename C::Fr;
    using F = Fe<P>;
    static F from_i32(int32_t n) { return fe_from_i32<P>(n); }
    static F from_words(const uint32_t* w) {
        uint32_t t[8];
        std::memcpy(t, w, 32);
        reduce_scalar_words<C>(t);
        return fe_from_canonical<P>(t);
    }
    static void to_words(const F& a, uint32_t* w) { fe_to_canonical(a, w); }
};

template <class C>
struct ProveImpl {
static int wip_fold_round(uint64_t* a, uint64_t* b, uint64_t* G, uint64_t* H, size_t len, const uint64_t* y_nhat,
                          const uint64_t* e);
static int range_prove(const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                     const uint64_t* v, const uint64_t* gamma, const uint64_t* V, uint64_t* out_points,
                     uint64_t* out_scalars, std::string& err);
};

// ---- definitions: compiled only by the translation unit that instantiates the struct (tu_*.hip defines
// BPP_IMPL_DEFINITIONS); capi.hip sees the declarations above and the `extern template` below, so it does not
// compile the kernels a second time ----
#ifdef BPP_IMPL_DEFINITIONS
template <class C>
int ProveImpl<C>::range_prove(const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                 const uint64_t* v, const uint64_t* gamma, const uint64_t* V, uint64_t* out_points,
                 uint64_t* out_scalars, std::string& err) {
using P = typename C::Fr;
using F = Fe<P>;
using HF = HostFr<C>;
constexpr int N = C::Fp::N;
constexpr int PW = (2 * N + 2) / 2;
const size_t mn = n * m;
if (n == 0 || m == 0 || n > 64 || !is_pow2(mn)) {
    err = "n*m must be a power of two, n <= 64";
    return BPP_E_ARG;
}
size_t k = 0;
while (((size_t)1 << k) < mn) k++;
hipStream_t st = nullptr;
auto E = [&](int rc) {
    if (rc) err = g_err;
    return rc;
};

// ---- device-resident points: [g, h] | G (mn) | H (mn) | V (m)
DevBuf d_gh, d_G, d_H, d_V;
int rc;
if ((rc = E(upload_points<C>(gh, 2, d_gh, st)))) return rc;
if ((rc = E(upload_points<C>(G, mn, d_G, st)))) return rc;
if ((rc = E(upload_points<C>(H, mn, d_H, st)))) return rc;
if ((rc = E(upload_points<C>(V, m, d_V, st)))) return rc;
const size_t PB = 2 * N * 4;  // bytes per affm point

// ---- constants
const bool single = (m == 1);
const F alpha = HF::from_i32(single ? 7 : 33);        // range/mod.rs:94 / :256
const F y = HF::from_i32(single ? 7 : 12);            // :109 / :278
const F z = HF::from_i32(single ? 7 : 23);            // :110 / :279
const F one = F::one();
const F minus_one = fe_neg(one);
const F z_sqr = fe_sqr(z);

// ---- A = alpha * h + sum_i (bit_i ? G_i : -H_i)        range/mod.rs:97-106 / :259-277
// as one device MulVec: [alpha | (1 on G_i where bit) | (-1 on H_i where !bit)]
std::vector<uint8_t> bits(mn);
{
    std::vector<uint32_t> sc((1 + 2 * mn) * 8, 0);
    HF::to_words(alpha, sc.data());
    uint32_t w_one[8], w_m1[8];
    HF::to_words(one, w_one);
    HF::to_words(minus_one, w_m1);
    for (size_t i = 0; i < mn; i++) {
        const size_t i1 = i % n, i2 = i / n;
        bits[i] = (uint8_t)((v[i2] >> i1) & 1);
        if (bits[i]) std::memcpy(sc.data() + (1 + i) * 8, w_one, 32);
        else std::memcpy(sc.data() + (1 + mn + i) * 8, w_m1, 32);
    }
    DevBuf dsc, dpts;
    if (hipMalloc(&dsc.p, sc.size() * 4) != hipSuccess || hipMalloc(&dpts.p, (1 + 2 * mn) * PB) != hipSuccess) {
        err = "hipMalloc failed";
        return BPP_E_NOMEM;
    }
    (void)hipMemcpyAsync(dsc.p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(dpts.p, static_cast<uint8_t*>(d_gh.p) + PB, PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(static_cast<uint8_t*>(dpts.p) + PB, d_G.p, mn * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(static_cast<uint8_t*>(dpts.p) + (1 + mn) * PB, d_H.p, mn * PB, hipMemcpyDeviceToDevice, st);
    std::vector<uint64_t> off = {0, 1 + 2 * mn};
    if ((rc = E(MsmImpl<C>::msm_batch_dev(dsc.u32(), dpts.u32(), off, out_points, st)))) return rc;
}

// ---- scalar vectors                                     range/mod.rs:113-172 / :283-376
std::vector<F> p2(n), py(mn), H_exp(mn), a_vec(mn), b_vec(mn);
{
    F cur = one;
    const F two = HF::from_i32(2);
    for (size_t i = 0; i < n; i++) {
        p2[i] = cur;
        cur = fe_mul(cur, two);
    }
    cur = y;
    for (size_t i = 0; i < mn; i++) {
        py[i] = cur;
        cur = fe_mul(cur, y);
    }
}
const F y_mn1 = fe_pow_u64(y, (uint64_t)mn + 1);
F alpha_hat;
if (single) {
    for (size_t i = 0; i < n; i++) H_exp[i] = fe_add(fe_mul(p2[i], py[n - 1 - i]), z);
    F gm = HF::from_words(reinterpret_cast<const uint32_t*>(gamma));
    alpha_hat = fe_add(alpha, fe_mul(gm, y_mn1));                           // :172
} else {
    std::vector<F> pz(m);
    F cur = z_sqr;
    for (size_t j = 0; j < m; j++) {
        pz[j] = cur;
        cur = fe_mul(cur, z_sqr);
    }
    for (size_t i = 0; i < mn; i++) {
        const F d = fe_mul(p2[i % n], pz[i / n]);
        H_exp[i] = fe_add(fe_mul(d, py[mn - 1 - i]), z);                    // :298-302
    }
    F pzg = F::zero();
    for (size_t j = 0; j < m; j++)
        pzg = fe_add(pzg, fe_mul(pz[j], HF::from_words(reinterpret_cast<const uint32_t*>(gamma) + 8 * j)));
    alpha_hat = fe_add(alpha, fe_mul(pzg, y_mn1));                          // :376
}
{
    const F nz = fe_neg(z), one_minus_z = fe_sub(one, z);
    for (size_t i = 0; i < mn; i++) {
        a_vec[i] = bits[i] ? one_minus_z : nz;                              // :159-162 / :351-355
        b_vec[i] = bits[i] ? H_exp[i] : fe_sub(H_exp[i], one);              // :164-170 / :357-364
    }
}

// ---- WeightedInnerProductProof::prove                     wip.rs:36-227
std::vector<F> a = a_vec, b = b_vec, pw = py;
F alpha_w = alpha_hat;
DevBuf d_pts, d_sc, d_fold;
const size_t maxterms = 2 * (mn + 2);
if (hipMalloc(&d_pts.p, maxterms * PB) != hipSuccess || hipMalloc(&d_sc.p, maxterms * 32) != hipSuccess ||
    hipMalloc(&d_fold.p, 4 * 32) != hipSuccess) {
    err = "hipMalloc failed";
    return BPP_E_NOMEM;
}
uint8_t* pG = static_cast<uint8_t*>(d_G.p);
uint8_t* pH = static_cast<uint8_t*>(d_H.p);
uint8_t* pts = static_cast<uint8_t*>(d_pts.p);
std::vector<uint64_t> LR(2 * PW);
size_t nn = mn, round = 0;
const F d_L = HF::from_i32(4), d_R = HF::from_i32(5);                       // wip.rs:94-95
while (nn != 1) {
    nn /= 2;
    const size_t T = 2 * nn + 2;
    F c_L = F::zero(), c_R = F::zero();
    for (size_t i = 0; i < nn; i++) {                                        // util.rs:117-127
        c_L = fe_add(c_L, fe_mul(fe_mul(a[i], b[nn + i]), pw[i]));
        c_R = fe_add(c_R, fe_mul(fe_mul(a[nn + i], b[i]), pw[nn + i]));
    }
    const F y_nhat = pw[nn - 1];
    const F y_nhat_inv = fe_inv(y_nhat);
    std::vector<uint32_t> sc(2 * T * 8);
    for (size_t i = 0; i < nn; i++) {
        HF::to_words(fe_mul(y_nhat_inv, a[i]), sc.data() + i * 8);               // G2_exp  :101
        HF::to_words(b[nn + i], sc.data() + (nn + i) * 8);                       // b2
        HF::to_words(fe_mul(y_nhat, a[nn + i]), sc.data() + (T + i) * 8);        // G1_exp  :100
        HF::to_words(b[i], sc.data() + (T + nn + i) * 8);                        // b1
    }
    HF::to_words(c_L, sc.data() + (2 * nn) * 8);
    HF::to_words(d_L, sc.data() + (2 * nn + 1) * 8);
    HF::to_words(c_R, sc.data() + (T + 2 * nn) * 8);
    HF::to_words(d_R, sc.data() + (T + 2 * nn + 1) * 8);
    (void)hipMemcpyAsync(d_sc.p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st);
    // L points [G2 | H1 | g | h]  (wip.rs:109-112) ; R points [G1 | H2 | g | h]  (:121-124)
    (void)hipMemcpyAsync(pts, pG + nn * PB, nn * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + nn * PB, pH, nn * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + 2 * nn * PB, d_gh.p, 2 * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + T * PB, pG, nn * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + (T + nn) * PB, pH + nn * PB, nn * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + (T + 2 * nn) * PB, d_gh.p, 2 * PB, hipMemcpyDeviceToDevice, st);
    std::vector<uint64_t> off = {0, T, 2 * T};
    if ((rc = E(MsmImpl<C>::msm_batch_dev(d_sc.u32(), d_pts.u32(), off, LR.data(), st)))) return rc;
    std::memcpy(out_points + (3 + round) * PW, LR.data(), PW * 8);
    std::memcpy(out_points + (3 + k + round) * PW, LR.data() + PW, PW * 8);
    round++;
    // challenge and folding                                 wip.rs:131-171
    const F e = HF::from_i32(7);
    const F e_inv = fe_inv(e);
    const F e_sqr = fe_sqr(e), e_sqr_inv = fe_sqr(e_inv);
    const F y_nhat_e_inv = fe_mul(y_nhat, e_inv);
    const F y_nhat_inv_e = fe_mul(y_nhat_inv, e);
    for (size_t i = 0; i < nn; i++) {
        a[i] = fe_add(fe_mul(a[i], e), fe_mul(a[nn + i], y_nhat_e_inv));
        b[i] = fe_add(fe_mul(b[i], e_inv), fe_mul(b[nn + i], e));
    }
    uint32_t fsc[32];
    HF::to_words(e_inv, fsc);             // G1 scalar
    HF::to_words(y_nhat_inv_e, fsc + 8);  // G2 scalar
    HF::to_words(e, fsc + 16);            // H1 scalar
    HF::to_words(e_inv, fsc + 24);        // H2 scalar
    if (hipMemcpyAsync(d_fold.p, fsc, sizeof fsc, hipMemcpyHostToDevice, st) != hipSuccess) {
        err = "hipMemcpyAsync failed";
        return BPP_E_HIP;
    }
    (void)hipStreamSynchronize(st);  // fsc is a stack buffer
    hipLaunchKernelGGL(k_fold_points<C>, dim3(cdiv(2 * nn, 64)), dim3(64), 0, st, d_G.u32(), d_H.u32(),
                       (uint32_t)nn, d_fold.u32());
    if (hipGetLastError() != hipSuccess) {
        err = "k_fold_points launch failed";
        return BPP_E_HIP;
    }
    alpha_w = fe_add(alpha_w, fe_add(fe_mul(e_sqr, d_L), fe_mul(e_sqr_inv, d_R)));
}
// ---- final A, B, r', s', delta'                            wip.rs:175-216
const F r = HF::from_i32(33), s = HF::from_i32(44), delta = HF::from_i32(88), eta = HF::from_i32(123);
const F rcbsca = fe_add(fe_mul(fe_mul(r, pw[0]), b[0]), fe_mul(fe_mul(s, pw[0]), a[0]));
const F rcs = fe_mul(fe_mul(r, pw[0]), s);
{
    uint32_t sc[6 * 8];
    HF::to_words(r, sc);
    HF::to_words(s, sc + 8);
    HF::to_words(rcbsca, sc + 16);
    HF::to_words(delta, sc + 24);
    HF::to_words(rcs, sc + 32);
    HF::to_words(eta, sc + 40);
    (void)hipMemcpyAsync(d_sc.p, sc, sizeof sc, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(pts, pG, PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + PB, pH, PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + 2 * PB, d_gh.p, 2 * PB, hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(pts + 4 * PB, d_gh.p, 2 * PB, hipMemcpyDeviceToDevice, st);
    std::vector<uint64_t> off = {0, 4, 6};
    std::vector<uint64_t> AB(2 * PW);
    if ((rc = E(MsmImpl<C>::msm_batch_dev(d_sc.u32(), d_pts.u32(), off, AB.data(), st)))) return rc;
    std::memcpy(out_points + PW, AB.data(), 2 * PW * 8);
}
const F e = HF::from_i32(99);
const F r_prime = fe_add(r, fe_mul(a[0], e));
const F s_prime = fe_add(s, fe_mul(b[0], e));
const F d_prime = fe_add(fe_add(eta, fe_mul(delta, e)), fe_mul(fe_mul(alpha_w, e), e));
uint32_t* os = reinterpret_cast<uint32_t*>(out_scalars);
HF::to_words(r_prime, os);
HF::to_words(s_prime, os + 8);
HF::to_words(d_prime, os + 16);
return BPP_OK;
}

// One folding round of WeightedInnerProductProof::prove as a seam of its own (reference wip.rs:147-164), host buffers,
// in place on the first len / 2 entries:
//   a[i] <- a[i] e + a[n' + i] y^n' e^-1        b[i] <- b[i] e^-1 + b[n' + i] e
//   G[i] <- e^-1 G[i] + y^-n' e G[n' + i]       H[i] <- e H[i] + e^-1 H[n' + i]                        n' = len / 2
template <class C>
int ProveImpl<C>::wip_fold_round(uint64_t* a, uint64_t* b, uint64_t* G, uint64_t* H, size_t len, const uint64_t* y_nhat,
                                 const uint64_t* e_in) {
    using P = typename C::Fr;
    using F = Fe<P>;
    using HF = HostFr<C>;
    constexpr int N = C::Fp::N;
    constexpr int PW = (2 * N + 2) / 2;
    if (len < 2 || (len & (len - 1))) return fail(BPP_E_ARG, "len must be a power of two >= 2");
    const size_t nn = len / 2;
    const F e = HF::from_words(reinterpret_cast<const uint32_t*>(e_in));
    const F yh = HF::from_words(reinterpret_cast<const uint32_t*>(y_nhat));
    if (e.is_zero() || yh.is_zero()) return fail(BPP_E_ARG, "the challenge and y^n' must be invertible");
    const F e_inv = fe_inv(e), yh_inv = fe_inv(yh);
    const F yh_e_inv = fe_mul(yh, e_inv), yh_inv_e = fe_mul(yh_inv, e);
    uint32_t* aw = reinterpret_cast<uint32_t*>(a);
    uint32_t* bw = reinterpret_cast<uint32_t*>(b);
    for (size_t i = 0; i < nn; i++) {
        const F a1 = HF::from_words(aw + i * 8), a2 = HF::from_words(aw + (nn + i) * 8);
        const F b1 = HF::from_words(bw + i * 8), b2 = HF::from_words(bw + (nn + i) * 8);
        HF::to_words(fe_add(fe_mul(a1, e), fe_mul(a2, yh_e_inv)), aw + i * 8);
        HF::to_words(fe_add(fe_mul(b1, e_inv), fe_mul(b2, e)), bw + i * 8);
    }
    DevBuf dG, dH, dsc, dw;
    int rc = upload_points<C>(G, len, dG, nullptr);
    if (rc) return rc;
    rc = upload_points<C>(H, len, dH, nullptr);
    if (rc) return rc;
    uint32_t fsc[32];
    HF::to_words(e_inv, fsc);
    HF::to_words(yh_inv_e, fsc + 8);
    HF::to_words(e, fsc + 16);
    HF::to_words(e_inv, fsc + 24);
    HIPCHK(dsc.alloc(sizeof fsc));
    HIPCHK(hipMemcpy(dsc.p, fsc, sizeof fsc, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fold_points<C>, dim3(cdiv(2 * nn, 64)), dim3(64), 0, nullptr, dG.u32(), dH.u32(), (uint32_t)nn,
                       dsc.u32());
    HIPCHK(hipGetLastError());
    HIPCHK(dw.alloc(nn * (2 * N + 2) * 4));
    for (int which = 0; which < 2; which++) {
        hipLaunchKernelGGL(k_points_to_wire<C>, dim3(cdiv(nn, 64)), dim3(64), 0, nullptr, which ? dH.u32() : dG.u32(), dw.u32(), nn);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(which ? H : G, dw.p, nn * PW * 8, hipMemcpyDeviceToHost));
    }
    return BPP_OK;
}

#endif  // BPP_IMPL_DEFINITIONS


extern template struct ProveImpl<Bls12381>;
extern template struct ProveImpl<Secp256k1>;
extern template struct ProveImpl<Ed25519>;

}  // namespace bpp
