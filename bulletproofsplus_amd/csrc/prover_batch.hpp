// prover_batch.hpp -- RangeProof::prove for a BATCH of provers, device resident.
//
// Reference: src/range/mod.rs:80-187 / :240-403 (A, a_vec, b_vec, alpha_hat) and
// src/weighted_inner_product_proof.rs:36-227 (k folding rounds, final A, B, r', s', delta').
//
// MI355X-first formulation.  The reference folds the generator vectors every round
// (G1[i] <- e^-1 G1[i] + y^-n' e G2[i], wip.rs:151-163: 4 n' scalar multiplications per round) and then
// forms L, R as MulVecs over the folded points.  Here the POINTS are never folded: every folded generator is
// a known linear combination of the original generators,
//     G^(t)[i] = sum_{j = i mod n_t} cG_j^(t) G_j ,   cG_j^(t) = prod_{r<t} (bit_r(j) ? y^-n'_r e_r : e_r^-1)
//     H^(t)[i] = sum_{j = i mod n_t} cH_j^(t) H_j ,   cH_j^(t) = prod_{r<t} (bit_r(j) ? e_r^-1      : e_r)
// (bit_r(j) = the r-th most significant bit of j), so every L_t, R_t, the final A and B, and the range
// proof's A are MulVecs over the ORIGINAL fixed generators g, h, G_vec, H_vec with per-proof scalars.  The
// rounds therefore fold only scalars (Fr vectors a, b, cG, cH), and all 2k+3 MulVecs of a proof go through
// the verifier's fixed-base window tables (k_fixed_msm) as "virtual proofs" -- no doublings anywhere.
// The group elements are the same as the reference's, so the proof is bit-identical.
// A_hat (range/mod.rs:153,:343) and P (wip.rs:137-142) are dead values in the reference and not computed.
//
// Per-proof device state (Fr elements packed, 8 words): a[mn], b[mn], cG[mn], cH[mn], pwy[mn] = y^(i+1),
// consts[...]; virtual-proof scalar arrays vps[(2k+3+m)][N][8] in the verifier's MulVec layout
// (fixed_term_index), canonical.
#pragma once
#include "kernels.hpp"

namespace bpp {

// hard-coded prover "randomness" of the reference, canonical small integers
struct ProverConsts {
    uint32_t alpha;            // range/mod.rs:94 (7, m == 1) / :256 (33, m > 1)
    uint32_t d_L, d_R;         // wip.rs:94-95
    uint32_t r, s, delta, eta; // wip.rs:175-178
};

// The reference's blinding values are literals, so its proofs hide nothing (with alpha, r, s, delta, eta, d_L, d_R known,
// r', s', delta' give away the folded a, b and a linear combination of the gammas).  A caller that wants hiding proofs
// supplies them per proof -- `blind`: [count][5 + 2k] canonical scalars [alpha, r, s, delta, eta, d_L[0..k), d_R[0..k)] --
// either as a buffer of its own or expanded from a 32-byte secret key by k_pb_blind.  blind == nullptr: the literals.
__host__ __device__ inline uint32_t pb_blind_elems(uint32_t k) { return 5 + 2 * k; }
enum { PB_BL_ALPHA = 0, PB_BL_R = 1, PB_BL_S = 2, PB_BL_DELTA = 3, PB_BL_ETA = 4, PB_BL_DL = 5 };
template <class P>
__device__ __forceinline__ Fe<P> pb_blind(uint32_t literal, const uint32_t* __restrict__ blind, size_t p, uint32_t k,
                                          uint32_t slot) {
    if (!blind) return fe_from_u32<P>(literal);
    uint32_t w[8];
    ld_words<8>(blind + (p * pb_blind_elems(k) + slot) * 8, w);
    return fe_from_canonical<P>(w);
}
// blind[p][slot] = (c0 + 2^256 c1) mod r,  c_h = SHA-256(key[32] || "bppb" || (index_base + p) as u64 LE || slot as u32 LE
// || h as u32 LE) read as little-endian 256-bit integers; zero is replaced by one.  One lane per (proof, slot).
struct BlindKey {
    uint32_t w[8];   // the 32 key bytes as little-endian words
};
template <class C>
__global__ void __launch_bounds__(64) k_pb_blind(BlindKey key, uint64_t index_base, uint32_t k,
                                                 uint32_t* __restrict__ blind, size_t count) {
    using P = typename C::Fr;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ne = pb_blind_elems(k);
    if (i >= count * ne) return;
    const uint64_t idx = index_base + i / ne;
    const uint32_t slot = (uint32_t)(i % ne);
    uint32_t c[16];
    for (uint32_t h = 0; h < 2; h++) {
        Sha256 s;
        sha256_init(s);
#pragma unroll
        for (int t = 0; t < 8; t++) sha256_word_le(s, key.w[t]);
        sha256_word_le(s, 0x62707062u);   // "bppb"
        sha256_word_le(s, (uint32_t)idx);
        sha256_word_le(s, (uint32_t)(idx >> 32));
        sha256_word_le(s, slot);
        sha256_word_le(s, h);
        uint32_t dg[8];
        sha256_final(s, dg);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint32_t be = dg[t];
            c[8 * h + t] = (be >> 24) | ((be >> 8) & 0xff00u) | ((be << 8) & 0xff0000u) | (be << 24);
        }
    }
    uint32_t w128[8] = {0, 0, 0, 0, 1, 0, 0, 0};
    const Fe<P> f128 = fe_from_canonical<P>(w128);
    const Fe<P> f256 = fe_mul(f128, f128);
    Fe<P> x = fe_add(fe_from_canonical<P>(c), fe_mul(fe_from_canonical<P>(c + 8), f256));
    if (x.is_zero()) x = Fe<P>::one();
    uint32_t o[8];
    fe_to_canonical(x, o);
    st_words<8>(blind + i * 8, o);
}

// layout of the per-proof constants block (Fr elements, Montgomery, packed 8 words each)
//   [0] yinv  [1] alpha_w  [2] y  [3] z  [4] e_final  [5 .. 5+k) e_t  [5+k .. 5+2k) e_t^-1
__host__ __device__ inline uint32_t pb_consts_elems(uint32_t k) { return 5 + 2 * k; }
// virtual proofs of one real proof: 0 = range A ; 1+2t = L_t ; 2+2t = R_t ; 2k+1 = wip.A ; 2k+2 = wip.B ;
// 2k+3+j = commitment V_j
__host__ __device__ inline uint32_t pb_num_vps(uint32_t k, uint32_t m) { return 2 * k + 3 + m; }

template <class P>
__device__ __forceinline__ Fe<P> pb_ld(const uint32_t* p) {
    uint32_t w[8];
    ld_words<8>(p, w);
    return fe_load<P>(w);
}
template <class P>
__device__ __forceinline__ void pb_st(uint32_t* p, const Fe<P>& x) {
    uint32_t w[8];
    fe_store(x, w);
    st_words<8>(p, w);
}
template <class P>
__device__ __forceinline__ void pb_st_canon(uint32_t* p, const Fe<P>& x) {
    uint32_t w[8];
    fe_to_canonical(x, w);
    st_words<8>(p, w);
}

// phases of the per-proof kernels.  In the reference's constants mode every challenge is known up front and each
// kernel runs whole (PB_ALL).  Under the Fiat-Shamir transcript (transcript.hpp) a challenge exists only after the
// points it binds have been computed, so the same kernels run in two halves around the MulVecs and the hashing:
//   k_pb_init : PB_PRE  = what needs no challenge (the scalars of A and of the commitments V_j)
//               PB_POST = what needs y, z (vectors a, b, y-powers, alpha_hat)
//   k_pb_round: PB_PRE  = the scalars of L_t, R_t ;  PB_POST = the fold with e_t
//   k_pb_final: PB_PRE  = the scalars of wip.A, wip.B ;  PB_POST = r', s', delta' with e
enum { PB_ALL = 0, PB_PRE = 1, PB_POST = 2 };

// One block (256 threads) per proof.  values: [count][m] u64 ; gammas: [count][m][8] canonical ;
// challenges: [y, z, e, e_1..e_k] (shared when ch_stride == 0).  fs: the round challenges are not known yet (only
// y and z are read; e_t and e_t^-1 are filled in round by round by k_pb_fs_round).
template <class C>
__global__ void __launch_bounds__(256) k_pb_init(VerifyShape s, ProverConsts pc, const uint32_t* __restrict__ blind, uint32_t phase, uint32_t fs,
                                                 const uint64_t* __restrict__ values,
                                                 const uint32_t* __restrict__ gammas,
                                                 const uint32_t* __restrict__ challenges, uint32_t ch_stride,
                                                 uint32_t* __restrict__ st_a, uint32_t* __restrict__ st_b,
                                                 uint32_t* __restrict__ st_cG, uint32_t* __restrict__ st_cH,
                                                 uint32_t* __restrict__ st_pwy, uint32_t* __restrict__ st_consts,
                                                 uint32_t* __restrict__ vps) {
    using P = typename C::Fr;
    using F = Fe<P>;
    __shared__ F sh_ypw[VS_MAXK + 2];
    __shared__ F sh_pz[VS_MAXM];
    __shared__ F sh_p2[VS_MAXN];
    __shared__ F sh_z, sh_ymn1;
    const uint32_t tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const uint32_t k = s.k, mn = s.mn, n = s.n, m = s.m;
    const uint32_t* ch = challenges + (size_t)ch_stride * p;
    uint32_t* consts = st_consts + p * (size_t)pb_consts_elems(k) * 8;
    const uint32_t nvp = pb_num_vps(k, m);
    uint32_t* vp0 = vps + p * (size_t)nvp * s.N * 8;

    if (phase != PB_POST) {
        // zero the virtual-proof arrays of this proof
        uint4* q = reinterpret_cast<uint4*>(vp0);
        const size_t n16 = (size_t)nvp * s.N * 2;
        for (size_t t = tid; t < n16; t += blockDim.x) q[t] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        // virtual proof 0: range A = alpha h + sum (bit ? G_i : -H_i)          (range/mod.rs:94-107 / :256-277)
        // The -1 on H_i is stored as +1: k_fixed_msm negates the H_i table entries of this virtual proof instead
        // (VpSel, kernels.hpp) -- one subtraction in place of the 15 additions of the full-width scalar r - 1.
        const F one_ = F::one();
        for (uint32_t i = tid; i < mn; i += blockDim.x) {
            const uint32_t i1 = i % n, i2 = i / n;
            const uint32_t bit = (uint32_t)((values[p * m + i2] >> i1) & 1ull);
            pb_st_canon<P>(vp0 + (size_t)fixed_term_index(s, 2 + (bit ? 0u : mn) + i) * 8, one_);
        }
        if (tid == 0) {
            pb_st_canon<P>(vp0 + (size_t)fixed_term_index(s, 1) * 8, pb_blind<P>(pc.alpha, blind, p, k, PB_BL_ALPHA));
            for (uint32_t j = 0; j < m; j++) {
                // commitment V_j = new(v as i32) g + gamma h            (range/prover.rs:34-40)
                uint32_t w[8];
                ld_words<8>(gammas + (p * m + j) * 8, w);
                const F g = fe_from_canonical<P>(w);
                uint32_t* vpV = vp0 + (size_t)(2 * k + 3 + j) * s.N * 8;
                const int32_t vi = (int32_t)(uint32_t)values[p * m + j];
                pb_st_canon<P>(vpV + (size_t)fixed_term_index(s, 0) * 8, fe_from_i32<P>(vi));
                pb_st_canon<P>(vpV + (size_t)fixed_term_index(s, 1) * 8, g);
            }
        }
        if (phase == PB_PRE) return;
        __syncthreads();
    }
    if (tid == 0) {
        // batched inversion of [y, e_1..e_k] (one fe_inv), power table y^(2^b)
        uint32_t w[8];
        ld_words<8>(ch, w);
        const F y = fe_from_canonical<P>(w);
        ld_words<8>(ch + 8, w);
        const F z = fe_from_canonical<P>(w);
        pb_st<P>(consts + 2 * 8, y);
        pb_st<P>(consts + 3 * 8, z);
        if (fs) {
            pb_st<P>(consts + 0, fe_inv(y));  // y^-1 ; e, e_t, e_t^-1 arrive later (k_pb_fs_round / k_pb_fs_final)
        } else {
            ld_words<8>(ch + 16, w);
            const F ef = fe_from_canonical<P>(w);
            pb_st<P>(consts + 4 * 8, ef);
            F acc = y;  // prefix products kept in the e^-1 slots
            for (uint32_t t = 0; t < k; t++) {
                ld_words<8>(ch + (3 + t) * 8, w);
                const F e = fe_from_canonical<P>(w);
                pb_st<P>(consts + (5 + t) * 8, e);
                pb_st<P>(consts + (5 + k + t) * 8, acc);   // prefix before e_t
                acc = fe_mul(acc, e);
            }
            F inv = fe_inv(acc);
            for (uint32_t t = k; t-- > 0;) {
                const F e = pb_ld<P>(consts + (5 + t) * 8);
                const F pre = pb_ld<P>(consts + (5 + k + t) * 8);
                pb_st<P>(consts + (5 + k + t) * 8, fe_mul(inv, pre));  // e_t^-1
                inv = fe_mul(inv, e);
            }
            pb_st<P>(consts + 0, inv);  // y^-1
        }
        F yy = y;
        for (uint32_t bnum = 0; bnum <= k + 1; bnum++) {
            sh_ypw[bnum] = yy;
            yy = fe_sqr(yy);
        }
        sh_z = z;
        sh_ymn1 = fe_mul(sh_ypw[k], y);  // y^(mn+1)
        const F zsq = fe_sqr(z);
        F cur = m == 1 ? F::one() : zsq;
        for (uint32_t j = 0; j < m; j++) {
            sh_pz[j] = cur;
            cur = fe_mul(cur, zsq);
        }
    } else if (tid >= 64 && tid < 64 + n) {
        const uint32_t t = tid - 64;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        w[t >> 5] = 1u << (t & 31);
        sh_p2[t] = fe_from_canonical<P>(w);
    }
    __syncthreads();

    const F one = F::one();
    const F z = sh_z;
    const F nz = fe_neg(z), one_minus_z = fe_sub(one, z);
    // pwy[i] = y^(i+1)
    for (uint32_t i = tid; i < mn; i += blockDim.x) {
        F yp = one;
        const uint32_t e1 = i + 1;
        for (uint32_t bnum = 0; bnum <= k; bnum++)
            if ((e1 >> bnum) & 1u) yp = fe_mul(yp, sh_ypw[bnum]);
        pb_st<P>(st_pwy + (p * mn + i) * 8, yp);
    }
    __syncthreads();
    for (uint32_t i = tid; i < mn; i += blockDim.x) {
        const uint32_t i1 = i % n, i2 = i / n;
        const uint32_t bit = (uint32_t)((values[p * m + i2] >> i1) & 1ull);
        // H_exp[i] = d_i y^(mn-i) + z      (range/mod.rs:125-129 / :298-302)
        F d = sh_p2[i1];
        if (m != 1) d = fe_mul(d, sh_pz[i2]);
        const F ypow = pb_ld<P>(st_pwy + (p * mn + (mn - 1 - i)) * 8);
        const F hexp = fe_add(fe_mul(d, ypow), z);
        pb_st<P>(st_a + (p * mn + i) * 8, bit ? one_minus_z : nz);                 // :159-162 / :351-355
        pb_st<P>(st_b + (p * mn + i) * 8, bit ? hexp : fe_sub(hexp, one));         // :164-170 / :357-364
        pb_st<P>(st_cG + (p * mn + i) * 8, one);
        pb_st<P>(st_cH + (p * mn + i) * 8, one);
    }
    if (tid == 0) {
        const F alpha = pb_blind<P>(pc.alpha, blind, p, k, PB_BL_ALPHA);
        // alpha_hat = alpha + y^(mn+1) * sum_j pz_j gamma_j        (:172 / :366-376)
        F acc = F::zero();
        for (uint32_t j = 0; j < m; j++) {
            uint32_t w[8];
            ld_words<8>(gammas + (p * m + j) * 8, w);
            const F g = fe_from_canonical<P>(w);
            acc = fe_add(acc, fe_mul(sh_pz[j], g));
        }
        pb_st<P>(consts + 1 * 8, fe_add(alpha, fe_mul(acc, sh_ymn1)));
    }
}

// One folding round t (wip.rs:79-172) for every proof: emits the scalar arrays of L_t and R_t, then folds
// a, b, cG, cH and alpha.  One block (256 threads) per proof.
template <class C>
__global__ void __launch_bounds__(256) k_pb_round(VerifyShape s, ProverConsts pc, const uint32_t* __restrict__ blind, uint32_t t, uint32_t phase,
                                                  uint32_t* __restrict__ st_a, uint32_t* __restrict__ st_b,
                                                  uint32_t* __restrict__ st_cG, uint32_t* __restrict__ st_cH,
                                                  const uint32_t* __restrict__ st_pwy,
                                                  uint32_t* __restrict__ st_consts, uint32_t* __restrict__ vps) {
    using P = typename C::Fr;
    using F = Fe<P>;
    __shared__ F redL[256], redR[256];
    const uint32_t tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const uint32_t k = s.k, mn = s.mn;
    const uint32_t nt = mn >> t, nh = nt >> 1;
    uint32_t* a = st_a + p * (size_t)mn * 8;
    uint32_t* b = st_b + p * (size_t)mn * 8;
    uint32_t* cG = st_cG + p * (size_t)mn * 8;
    uint32_t* cH = st_cH + p * (size_t)mn * 8;
    const uint32_t* pwy = st_pwy + p * (size_t)mn * 8;
    uint32_t* consts = st_consts + p * (size_t)pb_consts_elems(k) * 8;
    const uint32_t nvp = pb_num_vps(k, s.m);
    uint32_t* vpL = vps + (p * nvp + 1 + 2 * t) * (size_t)s.N * 8;
    uint32_t* vpR = vpL + (size_t)s.N * 8;

    const F yh = pb_ld<P>(pwy + (size_t)(nh - 1) * 8);   // y^n'            (wip.rs:98)
    // y^-n' = (y^-1)^(n'), n' a power of two
    F yhinv = pb_ld<P>(consts + 0);
    for (uint32_t q = 1; q < nh; q <<= 1) yhinv = fe_sqr(yhinv);
    if (phase != PB_POST) {
    // c_L = sum a1 b2 y1 ; c_R = sum a2 b1 y2          (wip.rs:90-91, util.rs:117-127)
    F cl = F::zero(), cr = F::zero();
    for (uint32_t i = tid; i < nh; i += blockDim.x) {
        const F a1 = pb_ld<P>(a + (size_t)i * 8), a2 = pb_ld<P>(a + (size_t)(nh + i) * 8);
        const F b1 = pb_ld<P>(b + (size_t)i * 8), b2 = pb_ld<P>(b + (size_t)(nh + i) * 8);
        cl = fe_add(cl, fe_mul(fe_mul(a1, b2), pb_ld<P>(pwy + (size_t)i * 8)));
        cr = fe_add(cr, fe_mul(fe_mul(a2, b1), pb_ld<P>(pwy + (size_t)(nh + i) * 8)));
    }
    redL[tid] = cl;
    redR[tid] = cr;
    __syncthreads();
    for (uint32_t h = blockDim.x >> 1; h >= 1; h >>= 1) {
        if (tid < h) {
            redL[tid] = fe_add(redL[tid], redL[tid + h]);
            redR[tid] = fe_add(redR[tid], redR[tid + h]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        pb_st_canon<P>(vpL + (size_t)fixed_term_index(s, 0) * 8, redL[0]);
        pb_st_canon<P>(vpL + (size_t)fixed_term_index(s, 1) * 8, pb_blind<P>(pc.d_L, blind, p, k, PB_BL_DL + t));
        pb_st_canon<P>(vpR + (size_t)fixed_term_index(s, 0) * 8, redR[0]);
        pb_st_canon<P>(vpR + (size_t)fixed_term_index(s, 1) * 8, pb_blind<P>(pc.d_R, blind, p, k, PB_BL_DL + k + t));
    }
    // scalars of L_t and R_t over the ORIGINAL generators        (wip.rs:100-125)
    for (uint32_t j = tid; j < mn; j += blockDim.x) {
        const uint32_t i = j & (nt - 1);
        const F cg = pb_ld<P>(cG + (size_t)j * 8), chh = pb_ld<P>(cH + (size_t)j * 8);
        if (i >= nh) {
            // G_j feeds G2[i-n'] (L, scalar y^-n' a1) ; H_j feeds H2[i-n'] (R, scalar b1)
            const F a1 = pb_ld<P>(a + (size_t)(i - nh) * 8), b1 = pb_ld<P>(b + (size_t)(i - nh) * 8);
            pb_st_canon<P>(vpL + (size_t)fixed_term_index(s, 2 + j) * 8, fe_mul(fe_mul(yhinv, a1), cg));
            pb_st_canon<P>(vpR + (size_t)fixed_term_index(s, 2 + mn + j) * 8, fe_mul(b1, chh));
        } else {
            // G_j feeds G1[i] (R, scalar y^n' a2) ; H_j feeds H1[i] (L, scalar b2)
            const F a2 = pb_ld<P>(a + (size_t)(nh + i) * 8), b2 = pb_ld<P>(b + (size_t)(nh + i) * 8);
            pb_st_canon<P>(vpR + (size_t)fixed_term_index(s, 2 + j) * 8, fe_mul(fe_mul(yh, a2), cg));
            pb_st_canon<P>(vpL + (size_t)fixed_term_index(s, 2 + mn + j) * 8, fe_mul(b2, chh));
        }
    }
    if (phase == PB_PRE) return;
    __syncthreads();
    }   // phase != PB_POST
    const F e = pb_ld<P>(consts + (size_t)(5 + t) * 8);
    const F einv = pb_ld<P>(consts + (size_t)(5 + k + t) * 8);
    if (tid == 0) {
        // alpha += e^2 d_L + e^-2 d_R                                (wip.rs:171)
        F al = pb_ld<P>(consts + 1 * 8);
        al = fe_add(al, fe_add(fe_mul(fe_sqr(e), pb_blind<P>(pc.d_L, blind, p, k, PB_BL_DL + t)),
                           fe_mul(fe_sqr(einv), pb_blind<P>(pc.d_R, blind, p, k, PB_BL_DL + k + t))));
        pb_st<P>(consts + 1 * 8, al);
    }
    // fold the coefficient products (wip.rs:151-163 expressed on scalars)
    const F fG_hi = fe_mul(yhinv, e);
    for (uint32_t j = tid; j < mn; j += blockDim.x) {
        const bool hi = (j & (nt - 1)) >= nh;
        pb_st<P>(cG + (size_t)j * 8, fe_mul(pb_ld<P>(cG + (size_t)j * 8), hi ? fG_hi : einv));
        pb_st<P>(cH + (size_t)j * 8, fe_mul(pb_ld<P>(cH + (size_t)j * 8), hi ? einv : e));
    }
    // fold a and b                                                   (wip.rs:148-149)
    const F yh_einv = fe_mul(yh, einv);
    for (uint32_t i = tid; i < nh; i += blockDim.x) {
        const F a1 = pb_ld<P>(a + (size_t)i * 8), a2 = pb_ld<P>(a + (size_t)(nh + i) * 8);
        const F b1 = pb_ld<P>(b + (size_t)i * 8), b2 = pb_ld<P>(b + (size_t)(nh + i) * 8);
        pb_st<P>(a + (size_t)i * 8, fe_add(fe_mul(a1, e), fe_mul(a2, yh_einv)));
        pb_st<P>(b + (size_t)i * 8, fe_add(fe_mul(b1, einv), fe_mul(b2, e)));
    }
}

// After the k rounds: scalars of wip.A and wip.B, and r', s', delta'   (wip.rs:175-216)
template <class C>
__global__ void __launch_bounds__(256) k_pb_final(VerifyShape s, ProverConsts pc, const uint32_t* __restrict__ blind, uint32_t phase,
                                                  const uint32_t* __restrict__ st_a,
                                                  const uint32_t* __restrict__ st_b,
                                                  const uint32_t* __restrict__ st_cG,
                                                  const uint32_t* __restrict__ st_cH,
                                                  const uint32_t* __restrict__ st_consts,
                                                  uint32_t* __restrict__ vps, uint32_t* __restrict__ out_scalars) {
    using P = typename C::Fr;
    using F = Fe<P>;
    const uint32_t tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const uint32_t k = s.k, mn = s.mn;
    const uint32_t* consts = st_consts + p * (size_t)pb_consts_elems(k) * 8;
    const uint32_t nvp = pb_num_vps(k, s.m);
    uint32_t* vpA = vps + (p * nvp + 2 * k + 1) * (size_t)s.N * 8;
    uint32_t* vpB = vpA + (size_t)s.N * 8;
    const F r = pb_blind<P>(pc.r, blind, p, k, PB_BL_R), sc = pb_blind<P>(pc.s, blind, p, k, PB_BL_S);
    if (phase != PB_POST)
        for (uint32_t j = tid; j < mn; j += blockDim.x) {
            pb_st_canon<P>(vpA + (size_t)fixed_term_index(s, 2 + j) * 8, fe_mul(r, pb_ld<P>(st_cG + (p * mn + j) * 8)));
            pb_st_canon<P>(vpA + (size_t)fixed_term_index(s, 2 + mn + j) * 8, fe_mul(sc, pb_ld<P>(st_cH + (p * mn + j) * 8)));
        }
    if (tid == 0) {
        const F y = pb_ld<P>(consts + 2 * 8), alpha = pb_ld<P>(consts + 1 * 8);
        const F a0 = pb_ld<P>(st_a + p * (size_t)mn * 8), b0 = pb_ld<P>(st_b + p * (size_t)mn * 8);
        const F delta = pb_blind<P>(pc.delta, blind, p, k, PB_BL_DELTA), eta = pb_blind<P>(pc.eta, blind, p, k, PB_BL_ETA);
        const F ry = fe_mul(r, y);
        if (phase != PB_POST) {
            const F rcbsca = fe_add(fe_mul(ry, b0), fe_mul(fe_mul(sc, y), a0));
            pb_st_canon<P>(vpA + (size_t)fixed_term_index(s, 0) * 8, rcbsca);
            pb_st_canon<P>(vpA + (size_t)fixed_term_index(s, 1) * 8, delta);
            pb_st_canon<P>(vpB + (size_t)fixed_term_index(s, 0) * 8, fe_mul(ry, sc));
            pb_st_canon<P>(vpB + (size_t)fixed_term_index(s, 1) * 8, eta);
        }
        if (phase == PB_PRE) return;
        const F ef = pb_ld<P>(consts + 4 * 8);
        uint32_t* o = out_scalars + p * 24;
        pb_st_canon<P>(o, fe_add(r, fe_mul(a0, ef)));
        pb_st_canon<P>(o + 8, fe_add(sc, fe_mul(b0, ef)));
        pb_st_canon<P>(o + 16, fe_add(fe_add(eta, fe_mul(delta, ef)), fe_mul(fe_mul(alpha, ef), ef)));
    }
}

// One lane per virtual proof of the launch (sel.cnt per real proof, starting at virtual proof sel.first): sums its
// `per` jacobian partials, converts to affine, writes the wire point at its place in the proof record:
// out_points[p][3+2k] = [A, wip.A, wip.B, L.., R..] and out_V[p][m].
template <class C>
__global__ void __launch_bounds__(64) k_pb_collect(VerifyShape s, VpSel sel, const uint32_t* __restrict__ partials,
                                                   uint32_t per, uint32_t* __restrict__ out_points,
                                                   uint32_t* __restrict__ out_V, size_t nvp_total) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    constexpr int WW = 2 * N + 2;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nvp_total) return;
    const size_t p = g / sel.cnt;
    const uint32_t v = sel.first + (uint32_t)(g % sel.cnt);
    Jac<C> acc = jac_inf<C>();
    for (uint32_t t = 0; t < per; t++) acc = jac_add(acc, jac_ldg<C>(partials + (g * per + t) * JW));
    uint32_t w[WW];
    aff_to_wire(jac_to_aff(acc), w);
    uint32_t* dst;
    const uint32_t k = s.k;
    if (v == 0) dst = out_points + (p * (3 + 2 * k) + 0) * WW;
    else if (v == 2 * k + 1) dst = out_points + (p * (3 + 2 * k) + 1) * WW;
    else if (v == 2 * k + 2) dst = out_points + (p * (3 + 2 * k) + 2) * WW;
    else if (v <= 2 * k) {
        const uint32_t t = (v - 1) >> 1;
        const bool isR = ((v - 1) & 1u) != 0;
        dst = out_points + (p * (3 + 2 * k) + 3 + (isR ? k : 0) + t) * WW;
    } else {
        dst = out_V + (p * s.m + (v - (2 * k + 3))) * WW;
    }
#pragma unroll
    for (int t = 0; t < WW; t++) dst[t] = w[t];
}

// ---- Fiat-Shamir steps of the batched prover (transcript.hpp), one lane per proof -------------------------------
// tr_st: [count][8] running transcript states ; ch: [count][3 + k][8] the challenge block [y, z, e, e_1..e_k]
// (canonical) -- the same block the verifier derives (bpp_verifier_derive_challenges).

// after A and the commitments exist: V_0.., A -> y, z ; then the argument's separator
template <class C>
__global__ void __launch_bounds__(64) k_pb_fs_yz(VerifyShape s, TranscriptState st0, const uint32_t* __restrict__ out_points,
                                                 const uint32_t* __restrict__ out_V, uint32_t* __restrict__ tr_st,
                                                 uint32_t* __restrict__ ch, size_t count) {
    using P = typename C::Fr;
    constexpr uint32_t WW = 2 * C::Fp::N + 2;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    Transcript t;
    for (int i = 0; i < 8; i++) t.st[i] = st0.st[i];
    for (uint32_t j = 0; j < s.m; j++) tr_append_point<C>(t, tr_tag('V'), out_V + (p * s.m + j) * WW);
    tr_append_point<C>(t, tr_tag('A'), out_points + p * (size_t)(3 + 2 * s.k) * WW);
    uint32_t w[8];
    uint32_t* c = ch + p * (size_t)(3 + s.k) * 8;
    fe_to_canonical(tr_challenge<P>(t, tr_tag('y')), w);
    for (int i = 0; i < 8; i++) c[i] = w[i];
    fe_to_canonical(tr_challenge<P>(t, tr_tag('z')), w);
    for (int i = 0; i < 8; i++) c[8 + i] = w[i];
    const uint32_t dsep[2] = {tr_tag('w', 'i', 'p', 'p'), tr_tag(' ', 'v', '1', 0)};
    tr_append_words(t, tr_tag('d', 's', 'e', 'p'), dsep, 2);
    tr_append_u64(t, tr_tag('n'), s.mn);
    for (int i = 0; i < 8; i++) tr_st[p * 8 + i] = t.st[i];
}

// after L_t, R_t exist: e_t and its inverse into the proof's constants block
template <class C>
__global__ void __launch_bounds__(64) k_pb_fs_round(VerifyShape s, uint32_t t, const uint32_t* __restrict__ out_points,
                                                    uint32_t* __restrict__ tr_st, uint32_t* __restrict__ ch,
                                                    uint32_t* __restrict__ st_consts, size_t count) {
    using P = typename C::Fr;
    constexpr uint32_t WW = 2 * C::Fp::N + 2;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const uint32_t k = s.k;
    Transcript tr;
    for (int i = 0; i < 8; i++) tr.st[i] = tr_st[p * 8 + i];
    const uint32_t* rec = out_points + p * (size_t)(3 + 2 * k) * WW;
    tr_append_point<C>(tr, tr_tag('L'), rec + (size_t)(3 + t) * WW);
    tr_append_point<C>(tr, tr_tag('R'), rec + (size_t)(3 + k + t) * WW);
    const Fe<P> e = tr_challenge<P>(tr, tr_tag('e'));
    for (int i = 0; i < 8; i++) tr_st[p * 8 + i] = tr.st[i];
    uint32_t w[8];
    fe_to_canonical(e, w);
    for (int i = 0; i < 8; i++) ch[(p * (3 + k) + 3 + t) * 8 + i] = w[i];
    uint32_t* consts = st_consts + p * (size_t)pb_consts_elems(k) * 8;
    pb_st<P>(consts + (size_t)(5 + t) * 8, e);
    pb_st<P>(consts + (size_t)(5 + k + t) * 8, fe_inv(e));
}

// after wip.A, wip.B exist: the final challenge e
template <class C>
__global__ void __launch_bounds__(64) k_pb_fs_final(VerifyShape s, const uint32_t* __restrict__ out_points,
                                                    uint32_t* __restrict__ tr_st, uint32_t* __restrict__ ch,
                                                    uint32_t* __restrict__ st_consts, size_t count) {
    using P = typename C::Fr;
    constexpr uint32_t WW = 2 * C::Fp::N + 2;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const uint32_t k = s.k;
    Transcript tr;
    for (int i = 0; i < 8; i++) tr.st[i] = tr_st[p * 8 + i];
    const uint32_t* rec = out_points + p * (size_t)(3 + 2 * k) * WW;
    tr_append_point<C>(tr, tr_tag('w', 'A'), rec + (size_t)1 * WW);
    tr_append_point<C>(tr, tr_tag('w', 'B'), rec + (size_t)2 * WW);
    const Fe<P> e = tr_challenge<P>(tr, tr_tag('e'));
    uint32_t w[8];
    fe_to_canonical(e, w);
    for (int i = 0; i < 8; i++) ch[(p * (3 + k) + 2) * 8 + i] = w[i];
    pb_st<P>(st_consts + p * (size_t)pb_consts_elems(k) * 8 + 4 * 8, e);
}

}  // namespace bpp
