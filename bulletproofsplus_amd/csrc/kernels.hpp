// kernels.hpp -- HIP kernels of the Bulletproofs+ verify / MSM hot path for gfx950 (MI355X).
//
// Data layout in HBM (32-bit words, see ec.hpp for the point images):
//   wire points      : canonical x | y | inf            (2N+2 words)   -- what the C ABI hands over
//   "affm" points    : Montgomery x | y                 (2N words)     -- pk, proof points, tables
//   jacobian partials: Montgomery X | Y | Z             (3N words)
//   scalars          : canonical, 8 words (4 x u64)
//   window table     : entry[f * per_f + j * half + (d - 1)] = d * 2^(c j) * F_f   (affm); windows 0..W-2 hold
//                      d = 1..half (signed digits, half = 2^(c-1)), the top window W-1 holds d = 1..top (its digit
//                      is the unsigned remainder of the scalar, so no window is spent on a recoding carry);
//                      per_f = (W-1) half + top; F = [g, h, G_0.., H_0..] the verifier's fixed generators.
//
// Kernel inventory (each names the reference call site it serves; paths relative to /root/reference/src):
//   k_points_from_wire   wire -> affm (+ on-curve check)
//   k_scalar_mul         n independent Point * scalar          point/point.rs:69-85, publickey.rs:21-48
//   k_msm_naive_partial  MulVec::calculate, one scalar-mul per thread + block reduce   mulvec.rs:20-33
//   k_jac_reduce         sums jacobian partials of one MulVec -> wire point
//   k_vs_prepare/expand  all verifier scalars of one proof      wip.rs:330-382, range/mod.rs:417-477, :198-226,
//                                                               wip.rs:254-295
//   k_fixed_msm          the 2mn+2 fixed-generator terms of the final MulVec via window tables (XYZZ sums, LDS-DMA
//                        gather ring, one partial per thread); its leading blocks run the Horner lanes of the
//                        proof-point MulVec                    range/mod.rs:480-503 / wip.rs:297-320
//   k_partials_fold      dense sums of the per-thread partials
//   k_var_digits/tables/windows  the 3+2k+m proof-dependent terms of the same MulVec (Straus per proof)
//   k_finalize           sum of partials, is_zero -> verdict    range/mod.rs:505-509, wip.rs:323-327
//   k_tbl_bases/k_tbl_fill  builds the window tables (setup, like PublicKey::new); affine_chain is the shared
//                        "chain of mixed additions + one inversion" of k_tbl_fill and k_var_tables
#pragma once
#include <hip/hip_runtime.h>
#include "ec.hpp"
#include "ed25519.hpp"
#include "ristretto.hpp"
#include "transcript.hpp"

namespace bpp {

// launch geometry (threads per block) of the heavy kernels; the __launch_bounds__ below let the register
// allocator use the whole 512-entry VGPR file at that occupancy instead of the 1024-thread default (128)
constexpr unsigned MSM_BLOCK = 64;
constexpr unsigned VS_PB = 8;                 // proofs per block of k_vs_expand, 64 lanes each
constexpr unsigned VS_BLOCK = VS_PB * 64;
constexpr unsigned FIXED_BLOCK = 128;
constexpr unsigned VAR_BLOCK = 64;
// Proof-point MSM: signed 4-bit windows.  Plain: 65 windows of a 260-bit value (scalar + bias).  BLS12-381 and secp256k1
// split every scalar with the curve's endomorphism first (GLV, ec.hpp glv_split_signed: k = +-k1 +- k2 mu, the image of P
// under [mu] is (beta x, -+y); k_var_digits), so a point contributes TWO 128-bit halves and there are 33 windows -- the
// same number of mixed additions (33 x 2 against 65), but half the ~256 sequential doublings of the Horner stage, which
// is what a small batch waits for.  edwards25519 has no such endomorphism and keeps 65 windows.
template <class C>
constexpr bool var_glv() {
    return curve_has_glv<C>();
}
template <class C>
constexpr uint32_t var_windows() {
    return var_glv<C>() ? 33u : 65u;
}
// window sums k_var_windows writes per proof: [half][window] (one lane each, so that a lone proof does not wait for
// 2 x NV sequential additions); the Horner stage adds the two halves of a window as it reads them
template <class C>
constexpr uint32_t var_wsums() {
    return var_windows<C>() * (var_glv<C>() ? 2u : 1u);
}
// ... and, for a launch so small that only latency counts (a lone proof), the proof's points are dealt to VAR_GROUPS
// lanes per (half, window) as well: a third of the additions in a row, two more partial sums for the Horner wave to add
constexpr uint32_t VAR_GROUPS = 3;
template <class C>
constexpr uint32_t var_wsums_max() {
    return var_wsums<C>() * VAR_GROUPS;
}
constexpr uint32_t VAR_DIGIT_STRIDE = 80;  // bytes reserved per item in the digit buffer (16-byte multiple): 65, or 2 x 33
constexpr uint32_t VAR_MULTIPLES = 8;      // table entries per proof point: 1P .. 8P
// minimum waves per SIMD the register allocator must leave room for (512 VGPRs / waves)
#ifndef BPP_FIXED_WAVES
#define BPP_FIXED_WAVES 2
#endif
#ifndef BPP_VAR_WAVES
#define BPP_VAR_WAVES 2
#endif
#ifndef BPP_VAR_TABLES_WAVES
#define BPP_VAR_TABLES_WAVES 2
#endif

// ---- small helpers -----------------------------------------------------------------------------------

// Zero-fill as a kernel instead of hipMemsetAsync: inside a captured HIP graph (ROCm 7.2) the memset node of a fill of
// more than ~1 KB wrote garbage from its second replay on (tests/test_gpu_round2.py::test_verifier_run_is_graph_capturable
// found every proof of a 300-proof batch "invalid" that way); a kernel node replays as it was captured.
static __global__ void __launch_bounds__(256) k_zero_words(uint32_t* __restrict__ p, size_t words) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) p[i] = 0;
}
inline hipError_t zero_words_async(void* p, size_t bytes, hipStream_t st) {
    const size_t words = (bytes + 3) / 4;
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t*>(p), words);
    return hipGetLastError();
}

template <int NW>
__device__ __forceinline__ void ld_words(const uint32_t* __restrict__ p, uint32_t* dst) {
    static_assert(NW % 4 == 0, "16-byte granules");
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        uint4 v = q[i];
        dst[4 * i] = v.x;
        dst[4 * i + 1] = v.y;
        dst[4 * i + 2] = v.z;
        dst[4 * i + 3] = v.w;
    }
}
template <int NW>
__device__ __forceinline__ void st_words(uint32_t* __restrict__ p, const uint32_t* src) {
    static_assert(NW % 4 == 0, "16-byte granules");
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) q[i] = make_uint4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
}

template <class C>
__device__ __forceinline__ Aff<C> aff_ldg(const uint32_t* __restrict__ p) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    uint32_t w[2 * N];
    ld_words<2 * N>(p, w);
    return aff_load<C>(w);
}
template <class C>
__device__ __forceinline__ void aff_stg(uint32_t* __restrict__ p, const Aff<C>& a) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    uint32_t w[2 * N];
    aff_store(a, w);
    st_words<2 * N>(p, w);
}
template <class C>
__device__ __forceinline__ Jac<C> jac_ldg(const uint32_t* __restrict__ p) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    uint32_t w[JW];
    ld_words<JW>(p, w);
    return jac_load<C>(w);
}
template <class C>
__device__ __forceinline__ void jac_stg(uint32_t* __restrict__ p, const Jac<C>& a) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    uint32_t w[JW];
    jac_store(a, w);
    st_words<JW>(p, w);
}

// Sum of the jacobian accumulators of a thread block.  `lds` holds blockDim.x * 3N words.
// On return thread 0 holds the block sum.  blockDim.x must be a power of two.
template <class C>
__device__ __forceinline__ Jac<C> block_reduce_jac(Jac<C> acc, uint32_t* lds) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const int tid = threadIdx.x;
    for (int half = blockDim.x >> 1; half >= 1; half >>= 1) {
        if (tid >= half && tid < 2 * half) jac_store(acc, lds + (size_t)tid * JW);
        __syncthreads();
        if (tid < half) acc = jac_add(acc, jac_load<C>(lds + (size_t)(tid + half) * JW));
        __syncthreads();
    }
    return acc;
}

// wave-wide exchange of a whole struct of 32-bit words (ds_bpermute per word; no LDS memory is touched)
template <class T>
__device__ __forceinline__ T wave_shfl(const T& v, int src_lane) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    struct Words {
        uint32_t w[sizeof(T) / 4];
    };
    Words a = __builtin_bit_cast(Words, v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) a.w[i] = (uint32_t)__shfl((int)a.w[i], src_lane, 64);
    return __builtin_bit_cast(T, a);
}

// sum over groups of `span` lanes (a power of two <= 64) of a wave (butterfly: every lane ends with its group's total)
template <class C>
__device__ __forceinline__ Jac<C> wave_sum_jac(Jac<C> x, int span = 64) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int d = span >> 1; d >= 1; d >>= 1) {
        const Jac<C> o = wave_shfl(x, lane ^ d);
        x = jac_add(x, o);
    }
    return x;
}

// A doubling shared by THREE lanes (lanes 3g, 3g+1, 3g+2 of a wave hold the same jacobian point; every lane of the
// wave calls this; all three return 2 p).  dbl-2009-l is 2M + 5S in a row for one lane -- ~10 us on the 13-limb field,
// and the Horner tails (the verifier's proof-point MulVec, the bucket MulVec) are chains of 110..130 of them that
// nothing else can hide.  Its seven products have a dependency depth of three, so three lanes finish in three
// product-times: level 1  X^2 | Y^2 | Y Z ;  level 2  (3 X^2)^2 | (Y^2)^2 | (X + Y^2)^2 ;  level 3  E (D - X3), the
// same in all three.  Operands travel by ds_bpermute (78 words per doubling).  Z = 0 stays Z = 0 (infinity), whatever
// X and Y turn into.  Short-Weierstrass a = 0 curves only.
template <class C>
__device__ __forceinline__ Jac<C> jac_dbl_tri(const Jac<C>& p) {
    using F = Fe<typename C::Fp>;
    static_assert(C::ID != 2, "short-Weierstrass curves");
    const int lane = threadIdx.x & 63;
    const int role = lane % 3, g0 = lane - role;
    auto pick = [](bool c, const F& a, const F& b) {
        F r;
#pragma unroll
        for (int i = 0; i < C::Fp::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
        return r;
    };
    const F m1 = fe_mul(pick(role == 0, p.X, p.Y), pick(role == 0, p.X, pick(role == 1, p.Y, p.Z)));   // A | B | Y Z
    const F B = wave_shfl(m1, (g0 + 1) & 63);
    const F m2 = fe_sqr(pick(role == 0, fe_add(fe_dbl(m1), m1), pick(role == 1, m1, fe_add(p.X, B))));   // F | C | t
    const F A = wave_shfl(m1, g0), Fq = wave_shfl(m2, g0), Cc = wave_shfl(m2, (g0 + 1) & 63);
    const F t = wave_shfl(m2, (g0 + 2) & 63), YZ = wave_shfl(m1, (g0 + 2) & 63);
    const F D = fe_dbl(fe_sub(fe_sub(t, A), Cc));
    const F E = fe_add(fe_dbl(A), A);
    Jac<C> r;
    r.X = fe_sub(Fq, fe_dbl(D));
    r.Y = fe_sub(fe_mul(E, fe_sub(D, r.X)), fe_dbl(fe_dbl(fe_dbl(Cc))));
    r.Z = fe_dbl(YZ);
    return r;
}

// The same idea for the Edwards instantiation, FOUR lanes per doubling (lanes 4g..4g+3 hold the same extended point; every
// lane of the wave calls this): dbl-2008-hwcd is 4S then 4M with nothing but additions between them, so four lanes finish
// in two product-times instead of eight:  X^2 | Y^2 | Z^2 | (X+Y)^2 ,  then  E F | G H | E H | F G.  72 words travel by
// ds_bpermute per doubling.  edwards25519 has no endomorphism, so its tails are twice as long as the other curves'
// (~250 doublings): this is what shortens them.
__device__ __forceinline__ Jac<Ed25519> ed_dbl_quad(const Jac<Ed25519>& p) {
    using F = ed::F;
    const int lane = threadIdx.x & 63;
    const int role = lane & 3, g0 = lane - role;
    const F s1 = fe_sqr(ed::select(role == 0, p.X, ed::select(role == 1, p.Y, ed::select(role == 2, p.Z, fe_add(p.X, p.Y)))));
    const F A = wave_shfl(s1, g0), B = wave_shfl(s1, g0 + 1), Zs = wave_shfl(s1, g0 + 2), XY = wave_shfl(s1, g0 + 3);
    const F C = fe_dbl(Zs), D = fe_neg(A);
    const F E = fe_sub(fe_sub(XY, A), B), G = fe_add(D, B), Fq = fe_sub(G, C), H = fe_sub(D, B);
    const F m2 = fe_mul(ed::select((role & 1) == 0, E, ed::select(role == 1, G, Fq)),
                        ed::select(role == 0, Fq, ed::select(role == 3, G, H)));   // E F | G H | E H | F G
    Jac<Ed25519> r;
    r.X = wave_shfl(m2, g0);
    r.Y = wave_shfl(m2, g0 + 1);
    r.T = wave_shfl(m2, g0 + 2);
    r.Z = wave_shfl(m2, g0 + 3);
    return r;
}
// lanes per shared doubling of a curve, and the shared doubling itself
template <class C>
constexpr uint32_t dbl_lanes() {
    return C::ID == 2 ? 4u : 3u;
}
template <class C>
__device__ __forceinline__ Jac<C> jac_dbl_shared(const Jac<C>& p) {
    if constexpr (C::ID == 2)
        return ed_dbl_quad(p);
    else
        return jac_dbl_tri<C>(p);
}

// ---- wire <-> device images --------------------------------------------------------------------------

// wire points -> affm.  per_group > 0: bad[i / per_group] is set when point i is invalid (coordinate
// >= p or not on the curve; with check_subgroup also: outside the prime-order subgroup, ec.hpp
// aff_in_prime_subgroup); invalid points are replaced by infinity.
template <class C>
__global__ void __launch_bounds__(128) k_points_from_wire(const uint32_t* __restrict__ wire, uint32_t* __restrict__ affm,
                                   uint32_t* __restrict__ bad, size_t n, uint32_t per_group, uint32_t check_subgroup = 0) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[2 * N + 2];
#pragma unroll
    for (int t = 0; t < 2 * N + 2; t++) w[t] = wire[i * (2 * N + 2) + t];
    Aff<C> p;
    bool ok = aff_from_wire<C>(w, p);
    if (ok && check_subgroup) ok = aff_in_prime_subgroup<C>(p);
    if (!ok) {
        p = aff_inf<C>();
        if (bad) atomicOr(&bad[per_group ? i / per_group : 0], 1u);
    }
    aff_stg<C>(affm + i * 2 * N, p);
}

template <class C>
__global__ void __launch_bounds__(64) k_points_to_wire(const uint32_t* __restrict__ affm, uint32_t* __restrict__ wire, size_t n) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<C> p = aff_ldg<C>(affm + i * 2 * N);
    uint32_t w[2 * N + 2];
    aff_to_wire(p, w);
#pragma unroll
    for (int t = 0; t < 2 * N + 2; t++) wire[i * (2 * N + 2) + t] = w[t];
}

// ---- scalar multiplication / naive MulVec ------------------------------------------------------------

// out[i] = scalars[i] * points[i]  (affm in, affm out).  point_stride = 0 broadcasts points[0].
template <class C>
__global__ void __launch_bounds__(64) k_scalar_mul(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points,
                             size_t point_stride, uint32_t* __restrict__ out, size_t n) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    ld_words<8>(scalars + i * 8, k);
    Aff<C> p = aff_ldg<C>(points + i * point_stride);
    Jac<C> r = aff_mul_words(p, k, 8);
    aff_stg<C>(out + i * 2 * N, jac_to_aff(r));
}

// MulVec::calculate, data-parallel restatement of reference mulvec.rs:28-31: every thread performs the
// scalar multiplications of its terms, a block reduces them; blockIdx.y selects the MulVec of a batch.
// offsets[c] .. offsets[c+1] delimit MulVec c.  partials: [count][gridDim.x] jacobians.
template <class C>
__global__ void __launch_bounds__(MSM_BLOCK) k_msm_naive_partial(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points,
                                    const uint64_t* __restrict__ offsets, uint32_t* __restrict__ partials) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const size_t c = blockIdx.y;
    const size_t beg = offsets[c], end = offsets[c + 1];
    Jac<C> acc = jac_inf<C>();
    for (size_t i = beg + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end;
         i += (size_t)gridDim.x * blockDim.x) {
        uint32_t k[8];
        ld_words<8>(scalars + i * 8, k);
        Aff<C> p = aff_ldg<C>(points + i * 2 * N);
        acc = jac_add(acc, aff_mul_words(p, k, 8));
    }
    acc = block_reduce_jac<C>(acc, lds);
    if (threadIdx.x == 0) jac_stg<C>(partials + (c * gridDim.x + blockIdx.x) * JW, acc);
}

// one thread per MulVec: sums its `per` jacobian partials, writes the wire point (affine, canonical)
template <class C>
__global__ void __launch_bounds__(64) k_jac_reduce(const uint32_t* __restrict__ partials, uint32_t per, uint32_t* __restrict__ wire_out,
                             size_t count) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    Jac<C> acc = jac_inf<C>();
    for (uint32_t t = 0; t < per; t++) acc = jac_add(acc, jac_ldg<C>(partials + (c * per + t) * JW));
    uint32_t w[2 * N + 2];
    aff_to_wire(jac_to_aff(acc), w);
#pragma unroll
    for (int t = 0; t < 2 * N + 2; t++) wire_out[c * (2 * N + 2) + t] = w[t];
}

// ---- verifier scalars --------------------------------------------------------------------------------

struct VerifyShape {
    uint32_t n, m, mn, k;      // bits per value, values, n*m, log2(mn)
    uint32_t N;                // MulVec length 2mn + 2k + m + 5
    uint32_t NF, NV;           // fixed terms 2mn + 2, proof-dependent terms 3 + 2k + m
    uint32_t c, W, half;       // window bits, windows, 2^(c-1)
    uint32_t top;              // entries of the top window (its digit is unsigned: 0..top)
    uint32_t per_f;            // table entries per generator: (W-1) * half + top
    uint32_t bias[10];         // sum_{j < W-1} half * 2^(c j) as 32-bit words (signed-digit recoding bias)
};

// index of fixed generator f (0 = g, 1 = h, 2.. = G_i, 2+mn.. = H_i) in the MulVec
__host__ __device__ __forceinline__ uint32_t fixed_term_index(const VerifyShape& s, uint32_t f) {
    return f < 2 ? 3 + f : 5 + 2 * s.k + (f - 2);
}
// index, in the MulVec, of proof-dependent point v of a proof record [A, wip.A, wip.B, L.., R.., V..].
// The head of the MulVec is [A, wip.A, wip.B] for m > 1 (range/mod.rs:492-494) but [wip.B, wip.A, A]
// for m == 1 (wip.rs:309-311).
__host__ __device__ __forceinline__ uint32_t var_term_index(const VerifyShape& s, uint32_t v) {
    if (v < 3) return s.m == 1 ? 2 - v : v;
    if (v < 3 + 2 * s.k) return 5 + (v - 3);
    return 5 + 2 * s.k + 2 * s.mn + (v - 3 - 2 * s.k);
}

// util.rs:54-71 / :81-98 including the n == 1 quirk (returns 1 for both types)
template <class P>
__device__ Fe<P> sum_of_powers(const Fe<P>& x, uint32_t n, bool type2) {
    if (n == 0) return Fe<P>::zero();
    if (n == 1) return Fe<P>::one();
    Fe<P> result = type2 ? fe_add(x, fe_sqr(x)) : fe_add(Fe<P>::one(), x);
    Fe<P> factor = x;
    uint32_t mm = n;
    while (mm > 2) {
        factor = fe_sqr(factor);
        result = fe_add(result, fe_mul(factor, result));
        mm >>= 1;
    }
    return result;
}

#define VS_MAXK 20
#define VS_MAXM 64
#define VS_MAXN 64
#define VS_MAXCH 64   // indices per lane = mn / 64 (<= n)

// per-proof scratch of the verifier-scalars kernels: a block of `prep` in HBM (k_vs_prepare) that k_vs_expand copies
// into dynamic LDS (sizes depend on k, m, mn/64)
template <class F>
struct VsShared {
    F* chsq;      // [k]    e_j^2
    F* chinvsq;   // [k]    e_j^-2
    F* ypw;       // [k+1]  y^(2^b)
    F* yipw;      // [k+1]  y^-(2^b)
    F* pz;        // [m]    (z^2)^(j+1)   (m > 1)
    F* sy_lo;     // [CH]   prod_{low bits of l set} e^2  *  y^-l
    F* sc_lo;     // [CH]   prod_{low bits of l unset} e^2
    F* t_lo;      // [CH]   (2 y^-1)^l
    F* tmp;       // [k+2]  prefix products of the batched inversion, then the e_j^-1
    F* c;         // [8]    kGa, kHa, yinv, cG, zc, hmul, einv
    __host__ __device__ static uint32_t elems(uint32_t k, uint32_t m, uint32_t CH) {
        return 2 * k + 2 * (k + 1) + m + 3 * CH + (k + 2) + 8;
    }
    __device__ VsShared(F* base, uint32_t k, uint32_t m, uint32_t CH) {
        chsq = base;
        chinvsq = chsq + k;
        ypw = chinvsq + k;
        yipw = ypw + k + 1;
        pz = yipw + k + 1;
        sy_lo = pz + m;
        sc_lo = sy_lo + CH;
        t_lo = sc_lo + CH;
        tmp = t_lo + CH;
        c = tmp + k + 2;
    }
    __device__ F& kGa() { return c[0]; }
    __device__ F& kHa() { return c[1]; }
    __device__ F& yinv() { return c[2]; }
    __device__ F& cG() { return c[3]; }
    __device__ F& zc() { return c[4]; }
    __device__ F& hmul() { return c[5]; }
    __device__ F& einv() { return c[6]; }
};

// dynamic LDS bytes of one k_vs_expand block (Fr elements are NL 32-bit words each)
template <class C>
inline size_t vs_lds_bytes(const VerifyShape& s) {
    using F = Fe<typename C::Fr>;
    const uint32_t CH = s.mn >= 64 ? s.mn / 64 : 1;
    return ((size_t)VS_MAXN + (size_t)VS_PB * VsShared<F>::elems(s.k, s.m, CH)) * sizeof(F);
}

// The verifier's scalars in two kernels.  Writes the N MulVec scalars of each proof (canonical, 8 words each) in
// the reference's MulVec order:
//   m > 1 (range/mod.rs:481-490): [1, e^-1, e^-2, g_exp, h_exp, e_i^2 (k), e_i^-2 (k), G_exp (mn), H_exp (mn), V_exp (m)]
//   m = 1 (wip.rs:298-307)      : [1, e,    e^2,  g_exp, h_exp, e_i^2 e^2,  e_i^-2 e^2, G_exp (n),  H_exp (n),  V_exp (1)]
// proof_scalars: [r', s', delta'] per proof; challenges: [y, z, e, e_1..e_k] (per proof when ch_stride != 0, shared
// otherwise).
//   k_vs_prepare (ONE LANE PER PROOF): what is serial per proof -- the inversion of y, e, e_1..e_k with ONE safegcd call
//     (Montgomery's trick for the rest), the per-proof constants and head scalars (~150 dependent Fr products), the
//     power tables y^(2^b), y^-(2^b) -- into a per-proof block of `prep`.  Every lane of every wave works on its own
//     proof; in round 1 this ran on one or two lanes of a 512-thread block while the other 500 waited, and the
//     kernel was latency bound (1.4 ms per 8192 proofs against ~0.3 ms of arithmetic).
//   k_vs_expand (VS_PB proofs per block, 64 lanes per proof, mn/64 consecutive indices each): copies the proofs'
//     blocks into LDS, builds three small per-proof tables, then three multiplications per index:
//   G_exp[i] = cG - [kG allinv prod_{hi bits} e^2 y^-(i0+1)] * sy_lo[l]                (range/mod.rs:456-459)
//   H_exp[i] = [pz y^(mn-i0) 2^(i0%n) cH] * t_lo[l] + z cH - [kH allinv prod_{hi unset} e^2] * sc_lo[l]   (:461-465)
// with i = i0 + l, using s_vec[i] = allinv * prod_{bit b of i set} e^2_{k-1-b} (wip.rs:372-380 unrolled).
template <class C>
inline size_t vs_prep_bytes(const VerifyShape& s) {
    using F = Fe<typename C::Fr>;
    const uint32_t CH = s.mn >= 64 ? s.mn / 64 : 1;
    return (size_t)VsShared<F>::elems(s.k, s.m, CH) * sizeof(F);
}

template <class C>
__global__ void __launch_bounds__(64) k_vs_prepare(VerifyShape s, const uint32_t* __restrict__ proof_scalars,
                                                   const uint32_t* __restrict__ challenges, uint32_t ch_stride,
                                                   uint32_t* __restrict__ prep, uint32_t* __restrict__ out, size_t count) {
    using P = typename C::Fr;
    using F = Fe<P>;
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    const uint32_t k = s.k, mn = s.mn;
    const uint32_t CH = mn >= 64 ? mn / 64 : 1;
    const uint32_t per_proof = VsShared<F>::elems(k, s.m, CH);
    VsShared<F> sh(reinterpret_cast<F*>(prep) + b * per_proof, k, s.m, CH);
    const uint32_t* ch = challenges + (size_t)ch_stride * b;
    // ---- inverses of [y, e, e_1..e_k] ------------------------------------------------------------------
    {
        // order: x_0 = y, x_1 = e, x_{2+j} = e_j ; zero entries are skipped (their "inverse" stays 0)
        F acc = F::one();
        for (uint32_t j = 0; j < k + 2; j++) {
            uint32_t w[8];
            ld_words<8>(ch + (j == 0 ? 0 : (j == 1 ? 2 : 1 + j)) * 8, w);
            F x = fe_from_canonical<P>(w);
            if (!x.is_zero()) acc = fe_mul(acc, x);
            sh.tmp[j] = acc;
        }
        F inv = fe_inv(acc);
        for (uint32_t j = k + 2; j-- > 0;) {
            uint32_t w[8];
            ld_words<8>(ch + (j == 0 ? 0 : (j == 1 ? 2 : 1 + j)) * 8, w);
            F x = fe_from_canonical<P>(w);
            F xi = F::zero();
            if (!x.is_zero()) {
                xi = j ? fe_mul(inv, sh.tmp[j - 1]) : inv;
                inv = fe_mul(inv, x);
            }
            if (j == 0) sh.yinv() = xi;
            else if (j == 1) sh.einv() = xi;
            else {
                sh.chsq[j - 2] = fe_sqr(x);
                sh.chinvsq[j - 2] = fe_sqr(xi);
                sh.tmp[j] = xi;                        // e_{j-2}^-1 (slot j is no longer needed as a prefix)
            }
        }
    }
    // ---- per-proof constants and head scalars ------------------------------------------------------------
    {
        uint32_t* o = out + b * (size_t)s.N * 8;
        uint32_t w[8];
        ld_words<8>(ch + 0, w);
        const F y = fe_from_canonical<P>(w);
        ld_words<8>(ch + 8, w);
        const F z = fe_from_canonical<P>(w);
        ld_words<8>(ch + 16, w);
        const F e = fe_from_canonical<P>(w);
        ld_words<8>(proof_scalars + b * 24, w);
        const F rp = fe_from_canonical<P>(w);
        ld_words<8>(proof_scalars + b * 24 + 8, w);
        const F sp = fe_from_canonical<P>(w);
        ld_words<8>(proof_scalars + b * 24 + 16, w);
        const F dp = fe_from_canonical<P>(w);
        const F einv = sh.einv();
        F allinv = F::one();                        // batch_invert's product of the e_j^-1
        for (uint32_t j = 0; j < k; j++) allinv = fe_mul(allinv, sh.tmp[j + 2]);
        const F zsq = fe_sqr(z);
        F head1, head2, g_exp, h_exp, lr_mul, kG, kH, cH;
        uint32_t wv[8];
        if (s.m == 1) {
            // range/mod.rs:198-226 + wip.rs:254-295
            const F esq = fe_sqr(e);
            head1 = e;
            head2 = esq;
            sh.cG() = fe_mul(fe_neg(z), esq);
            kG = fe_mul(fe_mul(rp, e), y);
            kH = fe_mul(sp, e);
            cH = esq;
            const F y_n1 = fe_pow_u64(y, (uint64_t)s.n + 1);
            F gc = F::zero();  // sum_{i<n} y^{i+1}
            {
                F cur = y;
                for (uint32_t i = 0; i < s.n; i++) {
                    gc = fe_add(gc, cur);
                    cur = fe_mul(cur, y);
                }
            }
            gc = fe_mul(gc, fe_sub(z, zsq));
            const F two = fe_from_u32<P>(2);
            const F t = fe_sub(fe_pow_u64(two, s.n), F::one());
            gc = fe_sub(gc, fe_mul(fe_mul(t, y_n1), z));
            g_exp = fe_add(fe_mul(fe_mul(fe_neg(rp), y), sp), fe_mul(gc, esq));
            h_exp = fe_neg(dp);
            lr_mul = esq;
            sh.pz[0] = F::one();
            fe_to_canonical(fe_mul(y_n1, esq), wv);  // V_exp
            st_words<8>(o + (size_t)(5 + 2 * k + 2 * mn) * 8, wv);
        } else {
            // range/mod.rs:417-477
            const F einv2 = fe_sqr(einv);  // == (e^2)^-1
            head1 = einv;
            head2 = einv2;
            sh.cG() = fe_neg(z);
            kG = fe_mul(fe_mul(rp, einv), y);
            kH = fe_mul(sp, einv);
            cH = F::one();
            const F y_mn1 = fe_pow_u64(y, (uint64_t)mn + 1);
            const F sum_y = sum_of_powers<P>(y, mn, true);
            const F sum_2 = sum_of_powers<P>(fe_from_u32<P>(2), s.n, false);
            const F sum_z = sum_of_powers<P>(zsq, s.m, true);
            const F t1 = fe_mul(fe_mul(fe_mul(fe_neg(rp), sp), y), einv2);
            const F t2 = fe_sub(fe_mul(sum_y, fe_sub(z, zsq)), fe_mul(fe_mul(fe_mul(y_mn1, z), sum_2), sum_z));
            g_exp = fe_add(t1, t2);
            h_exp = fe_mul(fe_neg(dp), einv2);
            lr_mul = F::one();
            F cur = zsq;
            for (uint32_t j = 0; j < s.m; j++) {  // power_of_z and V_exp
                sh.pz[j] = cur;
                fe_to_canonical(fe_mul(cur, y_mn1), wv);
                st_words<8>(o + (size_t)(5 + 2 * k + 2 * mn + j) * 8, wv);
                cur = fe_mul(cur, zsq);
            }
        }
        sh.kGa() = fe_mul(kG, allinv);
        sh.kHa() = fe_mul(kH, allinv);
        sh.hmul() = cH;
        sh.zc() = fe_mul(z, cH);
        fe_to_canonical(F::one(), wv);
        st_words<8>(o + 0, wv);
        fe_to_canonical(head1, wv);
        st_words<8>(o + 8, wv);
        fe_to_canonical(head2, wv);
        st_words<8>(o + 16, wv);
        fe_to_canonical(g_exp, wv);
        st_words<8>(o + 24, wv);
        fe_to_canonical(h_exp, wv);
        st_words<8>(o + 32, wv);
        for (uint32_t j = 0; j < k; j++) {
            fe_to_canonical(fe_mul(sh.chsq[j], lr_mul), wv);
            st_words<8>(o + (size_t)(5 + j) * 8, wv);
            fe_to_canonical(fe_mul(sh.chinvsq[j], lr_mul), wv);
            st_words<8>(o + (size_t)(5 + k + j) * 8, wv);
        }
        // power tables y^(2^b), y^-(2^b)
        F yy = y;
        F yi = sh.yinv();
        for (uint32_t bnum = 0; bnum <= k; bnum++) {
            sh.ypw[bnum] = yy;
            sh.yipw[bnum] = yi;
            yy = fe_sqr(yy);
            yi = fe_sqr(yi);
        }
    }
}

template <class C>
__global__ void __launch_bounds__(VS_BLOCK) k_vs_expand(VerifyShape s, const uint32_t* __restrict__ prep,
                                                        uint32_t* __restrict__ out, size_t count) {
    using P = typename C::Fr;
    using F = Fe<P>;
    extern __shared__ __align__(16) uint32_t lds_raw[];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = s.k, mn = s.mn;
    const uint32_t CH = mn >= 64 ? mn / 64 : 1;      // indices per lane (power of two, <= n)
    uint32_t cb = 0;
    while ((1u << cb) < CH) cb++;
    F* const lds_f = reinterpret_cast<F*>(lds_raw);
    F* const sh_p2 = lds_f;                           // [VS_MAXN] 2^t, shared by the block's proofs
    const uint32_t per_proof = VsShared<F>::elems(k, s.m, CH);
    auto proof_lds = [&](uint32_t slot) { return VsShared<F>(lds_f + VS_MAXN + (size_t)slot * per_proof, k, s.m, CH); };

    // ---- the proofs' prepared blocks: global -> LDS (word by word, coalesced); 2^t table ---------------------
    {
        const size_t first = (size_t)blockIdx.x * VS_PB;
        const size_t nproofs = count - first < VS_PB ? count - first : VS_PB;
        const size_t words = nproofs * per_proof * (sizeof(F) / 4);
        const uint32_t* src = prep + first * per_proof * (sizeof(F) / 4);
        uint32_t* dst = lds_raw + VS_MAXN * (sizeof(F) / 4);
        for (size_t t = tid; t < words; t += blockDim.x) dst[t] = src[t];
        if (tid < s.n && tid < VS_MAXN) {
            uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            w[tid >> 5] = 1u << (tid & 31);
            sh_p2[tid] = fe_from_canonical<P>(w);
        }
    }
    __syncthreads();

    // ---- low-bit tables, lane l < CH of each proof -----------------------------------------------------
    const uint32_t q = tid / 64, lane = tid % 64;
    const size_t b = (size_t)blockIdx.x * VS_PB + q;
    const bool active = b < count;
    VsShared<F> sh = proof_lds(q);
    if (active && lane < CH) {
        F slo = F::one(), sclo = F::one(), yl = F::one();
        for (uint32_t bnum = 0; bnum < cb; bnum++) {
            const F u = sh.chsq[k - 1 - bnum];
            if ((lane >> bnum) & 1u) {
                slo = fe_mul(slo, u);
                yl = fe_mul(yl, sh.yipw[bnum]);
            } else {
                sclo = fe_mul(sclo, u);
            }
        }
        sh.sy_lo[lane] = fe_mul(slo, yl);                // prod e^2 * y^-l
        sh.sc_lo[lane] = sclo;
        sh.t_lo[lane] = fe_mul(sh_p2[lane], yl);         // 2^l * y^-l   (l < CH <= n)
    }
    __syncthreads();

    // ---- three multiplications per index -----------------------------------------------------------------
    if (!active) return;
    const uint32_t i0 = lane * CH;
    if (i0 >= mn) return;
    uint32_t* o = out + b * (size_t)s.N * 8;
    F a_hi = sh.kGa(), c_hi = sh.kHa();
    for (uint32_t bnum = cb; bnum < k; bnum++) {
        const F u = sh.chsq[k - 1 - bnum];
        if ((i0 >> bnum) & 1u) a_hi = fe_mul(a_hi, u);
        else c_hi = fe_mul(c_hi, u);
    }
    F yip = F::one(), yp = F::one();  // y^-(i0+1), y^(mn-i0)
    {
        const uint32_t e1 = i0 + 1, e2 = mn - i0;
        for (uint32_t bnum = 0; bnum <= k; bnum++) {
            if ((e1 >> bnum) & 1u) yip = fe_mul(yip, sh.yipw[bnum]);
            if ((e2 >> bnum) & 1u) yp = fe_mul(yp, sh.ypw[bnum]);
        }
    }
    a_hi = fe_mul(a_hi, yip);
    F b_hi = fe_mul(fe_mul(yp, sh_p2[i0 % s.n]), sh.hmul());
    if (s.m != 1) b_hi = fe_mul(b_hi, sh.pz[i0 / s.n]);
    const F cG = sh.cG(), zc = sh.zc();
    for (uint32_t l = 0; l < CH; l++) {
        const uint32_t i = i0 + l;
        const F ge = fe_sub(cG, fe_mul(a_hi, sh.sy_lo[l]));
        const F he = fe_sub(fe_add(fe_mul(b_hi, sh.t_lo[l]), zc), fe_mul(c_hi, sh.sc_lo[l]));
        uint32_t wv[8];
        fe_to_canonical(ge, wv);
        st_words<8>(o + (size_t)(5 + 2 * k + i) * 8, wv);
        fe_to_canonical(he, wv);
        st_words<8>(o + (size_t)(5 + 2 * k + mn + i) * 8, wv);
    }
}

// ---- Fiat-Shamir challenges (transcript.hpp) --------------------------------------------------------------
struct TranscriptState {
    uint32_t st[8];
};
// one lane per proof: record -> [y, z, e, e_1..e_k]
template <class C>
__global__ void __launch_bounds__(64) k_transcript_challenges(VerifyShape s, TranscriptState st0,
                                                              const uint32_t* __restrict__ records,
                                                              uint32_t* __restrict__ challenges, size_t count) {
    constexpr uint32_t WW = 2 * C::Fp::N + 2;
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    tr_verifier_challenges<C>(st0.st, records + b * (size_t)s.NV * WW, s.k, s.m, s.mn, challenges + b * (size_t)(3 + s.k) * 8);
}

// ---- window tables -------------------------------------------------------------------------------------

// J, J + step, J + 2 step, ... (count points) as AFFINE points in out[0 .. count-1]: a chain of mixed additions
// in projective coordinates, then ONE inversion (safegcd) of the product of all Z's and a backward pass
// (Montgomery's trick) -- ~19 field multiplications per point instead of an inversion each.  While the chain runs,
// X | Y are parked in the output slot itself and Z with the running product in scratch[0 .. 2 count) (N words
// each).  A Weierstrass point outside the prime-order subgroup (BLS12-381 G1 has cofactor 3 * 11^2 * ...) can
// reach infinity inside the chain: such a point is parked as x = y = 0 with Z = 1, so the product stays
// invertible and the scaled entry is the infinity encoding.  (Edwards: Z is never 0.)
// `step` is given as a pointer (global memory) and re-read at every link, and the running product of the Z's lives in
// `scratch` between links: both would otherwise sit in registers (39 of them for BLS12-381) across the mixed addition,
// and with them the function does not fit the 256 registers of two waves per SIMD.
template <class C>
__device__ __forceinline__ void affine_chain(Jac<C> J, const uint32_t* __restrict__ step_ptr, uint32_t count,
                                             uint32_t* __restrict__ out, uint32_t* __restrict__ scratch) {
    using P = typename C::Fp;
    using F = Fe<P>;
    constexpr int N = P::N;
    uint32_t w[N];
    for (uint32_t k = 0; k < count; k++) {
        if (k > 0) J = jac_madd(J, aff_ldg<C>(step_ptr));
        const bool at_inf = C::ID != 2 && J.is_inf();
        const F z = at_inf ? F::one() : J.Z;
        F run = z;
        if (k > 0) {
            ld_words<N>(scratch + (size_t)(count + k - 1) * N, w);
            run = fe_mul(fe_load<P>(w), z);
        }
        fe_store(at_inf ? F::zero() : J.X, w);
        st_words<N>(out + (size_t)k * 2 * N, w);
        fe_store(at_inf ? F::zero() : J.Y, w);
        st_words<N>(out + (size_t)k * 2 * N + N, w);
        fe_store(z, w);
        st_words<N>(scratch + (size_t)k * N, w);
        fe_store(run, w);
        st_words<N>(scratch + (size_t)(count + k) * N, w);
    }
    ld_words<N>(scratch + (size_t)(2 * count - 1) * N, w);
    F inv = fe_inv(fe_load<P>(w));   // (Z_0 ... Z_{count-1})^-1
    for (uint32_t k = count; k-- > 0;) {
        F zi = inv;
        if (k > 0) {
            ld_words<N>(scratch + (size_t)(count + k - 1) * N, w);   // product of the Z's before k
            zi = fe_mul(inv, fe_load<P>(w));
            ld_words<N>(scratch + (size_t)k * N, w);
            inv = fe_mul(inv, fe_load<P>(w));
        }
        // only X and Y are read by jac_scale_to_aff; the running point J must NOT be named here, or its 3 N registers
        // stay alive across the inversion (that alone was most of this function's scratch spills)
        Jac<C> q;
        ld_words<N>(out + (size_t)k * 2 * N, w);
        q.X = fe_load<P>(w);
        ld_words<N>(out + (size_t)k * 2 * N + N, w);
        q.Y = fe_load<P>(w);
        q.Z = q.Y;
        if constexpr (C::ID == 2) q.T = q.Y;
        aff_stg<C>(out + (size_t)k * 2 * N, jac_scale_to_aff(q, zi));
    }
}

// thread f: base_{f,j} = 2^(c j) * F_f for every window j, written as entry d = 1
template <class C>
__global__ void __launch_bounds__(64) k_tbl_bases(VerifyShape s, const uint32_t* __restrict__ fixed_pts, uint32_t* __restrict__ table) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= s.NF) return;
    Aff<C> p = aff_ldg<C>(fixed_pts + (size_t)f * 2 * N);
    for (uint32_t j = 0; j < s.W; j++) {
        aff_stg<C>(table + ((size_t)f * s.per_f + (size_t)j * s.half) * 2 * N, p);
        if (j + 1 < s.W) {
            Jac<C> q = jac_from_aff(p);
            for (uint32_t t = 0; t < s.c; t++) q = jac_dbl(q);
            p = jac_to_aff(q);
        }
    }
}

// one thread per RUN of TBL_RUN consecutive entries of one window: d0 * base by double-and-add, then
// (d0 + 1) * base, ... by affine_chain.  scratch: [thread][2 * TBL_RUN] field elements.
constexpr uint32_t TBL_RUN = 32;
__host__ __device__ __forceinline__ uint32_t tbl_runs_per_generator(const VerifyShape& s) {
    return (s.W - 1) * ((s.half + TBL_RUN - 1) / TBL_RUN) + (s.top + TBL_RUN - 1) / TBL_RUN;
}
template <class C>
__global__ void __launch_bounds__(64, BPP_VAR_TABLES_WAVES) k_tbl_fill(VerifyShape s, uint32_t* __restrict__ table, uint32_t* __restrict__ scratch,
                                                    uint32_t f_begin, uint32_t f_end) {
    constexpr int N = C::Fp::N;
    const uint32_t runs_low = (s.half + TBL_RUN - 1) / TBL_RUN;
    const uint32_t runs_f = tbl_runs_per_generator(s);
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(f_end - f_begin) * runs_f) return;
    const uint32_t f = f_begin + (uint32_t)(idx / runs_f);
    const uint32_t rem = (uint32_t)(idx % runs_f);
    const uint32_t j = min(rem / runs_low, s.W - 1);   // the top window has its own number of runs
    const uint32_t r = rem - j * runs_low;
    const uint32_t cnt = j + 1 < s.W ? s.half : s.top;
    const uint32_t d0 = r * TBL_RUN + 1;
    const uint32_t n = min(TBL_RUN, cnt - (d0 - 1));
    uint32_t* win = table + ((size_t)f * s.per_f + (size_t)j * s.half) * 2 * N;
    const Aff<C> base = aff_ldg<C>(win);   // entry d = 1, written by k_tbl_bases
    Jac<C> J;
    if (d0 == 1) {
        J = jac_from_aff(base);
    } else {
        uint32_t kw[1] = {d0};
        J = aff_mul_words(base, kw, 1);
    }
    affine_chain<C>(J, win, n, win + (size_t)(d0 - 1) * 2 * N, scratch + idx * 2 * TBL_RUN * N);
}

// ---- the verification MulVec ---------------------------------------------------------------------------

// Last stage of the proof-point MSM (see k_var_tables / k_var_windows below), one lane per proof:
// out[b] = sum_j 16^j * wsum[b][j] by Horner's rule -- 256 doublings that can only run one after the other.
// Alone on the chip this is 128 waves of pure latency, so it does not get a launch of its own: the first
// `horner_blocks` blocks of k_fixed_msm's grid run it, beside the blocks that do the fixed-generator sums
// (one lane per proof; for small batches a block per proof: var_horner_wave2 / var_horner_wave_ed).
// window sum j of proof b.  SPLIT (the layout k_var_windows writes for the tree Horner): S_j = (half 0) + (half 1)
template <class C, bool SPLIT>
__device__ __forceinline__ Jac<C> var_wsum_ld(const uint32_t* __restrict__ wsum, size_t b, uint32_t j, uint32_t groups = 1) {
    constexpr int JW = jac_words<C>();
    constexpr uint32_t NW = var_windows<C>();
    const uint32_t parts = SPLIT ? (var_wsums<C>() / NW) * groups : 1u;   // partial sums of window j: [group][half]
    const uint32_t* W = wsum + b * parts * NW * JW;
    Jac<C> sj = jac_ldg<C>(W + (size_t)j * JW);
    for (uint32_t t = 1; t < parts; t++) sj = jac_add(sj, jac_ldg<C>(W + (size_t)(t * NW + j) * JW));
    return sj;
}

template <class C>
__device__ __forceinline__ void var_horner_lane(const uint32_t* __restrict__ wsum, uint32_t* __restrict__ out, size_t b) {
    constexpr int JW = jac_words<C>();
    constexpr uint32_t NW = var_windows<C>();
    Jac<C> acc = var_wsum_ld<C, false>(wsum, b, NW - 1);
    for (int j = (int)NW - 2; j >= 0; j--) {
        if (!acc.is_inf()) {
            acc = jac_dbl(acc);
            acc = jac_dbl(acc);
            acc = jac_dbl(acc);
            acc = jac_dbl(acc);
        }
        acc = jac_add(acc, var_wsum_ld<C, false>(wsum, b, (uint32_t)j));
    }
    jac_stg<C>(out + b * JW, acc);
}

// The wave-per-proof form for the curves with 33 windows (GLV), as a BLOCK of two waves: three lanes per window share
// every doubling (jac_dbl_tri above) -- window j doubles its sum 4 j times, all windows at once, so the critical path is
// the top window's 128 doublings at three product-times each instead of seven -- then the windows' lanes are compacted,
// a butterfly adds them and the second wave's sum joins through LDS.  Every lane of the 128-thread block calls it.
template <class C>
__device__ __forceinline__ void var_horner_wave2(const uint32_t* __restrict__ wsum, uint32_t* __restrict__ out, size_t b,
                                                 uint32_t* lds, uint32_t groups) {
    constexpr uint32_t NW = var_windows<C>(), PER = 21;   // windows per wave (63 lanes)
    static_assert(NW <= 2 * PER, "two waves hold the windows");
    constexpr int JW = jac_words<C>();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t q = lane / 3, j = wave * PER + q;
    const bool has = q < PER && j < NW;
    Jac<C> R = has ? var_wsum_ld<C, true>(wsum, b, j, groups) : jac_inf<C>();
    const uint32_t times = has ? 4 * j : 0u;
    const uint32_t maxt = 4 * (wave == 0 ? PER - 1 : NW - 1);   // wave-uniform
    for (uint32_t t = 0; t < maxt; t++) {
        const Jac<C> d = jac_dbl_tri<C>(R);
        if (t < times) R = d;
    }
    R = wave_shfl(R, (int)((3 * lane) & 63u));
    if (lane >= PER || wave * PER + lane >= NW) R = jac_inf<C>();
    R = wave_sum_jac<C>(R, 32);
    if (wave == 1 && lane == 0) jac_store(R, lds);
    __syncthreads();
    if (wave == 0 && lane == 0) jac_stg<C>(out + b * JW, jac_add(R, jac_load<C>(lds)));
}

// The wave-per-proof form for the 65 windows of the Edwards instantiation: sixteen lane-quads, quad q runs Horner's rule
// over windows q, q + 16, q + 32, q + 48 (64 doublings between them; quad 0 also takes window 64 on top), then 4 q more
// doublings, every doubling shared by the quad's four lanes (ed_dbl_quad) -- at most 256 doublings of two product-times
// each on the critical path instead of 256 of eight -- and a butterfly adds the sixteen quads.  One wave.
__device__ __forceinline__ void var_horner_wave_ed(const uint32_t* __restrict__ wsum, uint32_t* __restrict__ out, size_t b,
                                                   uint32_t groups) {
    using C = Ed25519;
    constexpr uint32_t NW = var_windows<C>();   // 65
    static_assert(NW == 65, "sixteen quads x four windows + one");
    constexpr int JW = jac_words<C>();
    const uint32_t lane = threadIdx.x & 63u, q = lane >> 2;
    Jac<C> acc = jac_inf<C>();
    if (q == 0) {   // (a quad only ever reads its own four lanes, so quads may diverge from one another)
        acc = var_wsum_ld<C, true>(wsum, b, 64, groups);
        for (int t = 0; t < 64; t++) acc = ed_dbl_quad(acc);
    }
    for (int w = 3; w >= 0; w--) {
        acc = jac_add(acc, var_wsum_ld<C, true>(wsum, b, q + 16 * (uint32_t)w, groups));
        const int times = w > 0 ? 64 : 60;   // after the last window: 4 q doublings, the wave runs the maximum (60)
        for (int t = 0; t < times; t++) {
            const Jac<C> d = ed_dbl_quad(acc);
            if (w > 0 || (uint32_t)t < 4 * q) acc = d;
        }
    }
    acc = wave_shfl(acc, (int)((4 * lane) & 63u));   // quad q's sum to lane q
    if (lane >= 16) acc = jac_inf<C>();
    acc = wave_sum_jac<C>(acc, 16);
    if (lane == 0) jac_stg<C>(out + b * JW, acc);
}

// Between the two: EIGHT lanes per proof (eight proofs per wave), for batches that are too large for a wave per proof
// but whose fixed-generator work is over before a one-lane chain would be (4 096 proofs of (64,1): the chain of 128
// doublings + 33 additions was 2.2 of the pass's 3.7 ms).  Lane g runs Horner over its own eighth of the windows, then
// three tree levels combine the eight partials: the same 128 (256) doublings on the critical path, 7 (11) additions.
// Unsplit layout of the window sums (one per window).  Every lane of the wave must call it (`active`: has a proof).
template <class C>
__device__ __forceinline__ void var_horner_group(const uint32_t* __restrict__ wsum, uint32_t* __restrict__ out, size_t b,
                                                 bool active, uint32_t* lds_wave) {
    constexpr uint32_t NW = var_windows<C>(), L = NW - 1, G = 8, CW = L / G;   // CW windows per lane + the carry window
    static_assert(L % G == 0, "windows split evenly over the eight lanes");
    constexpr int JW = jac_words<C>();
    const uint32_t lane = threadIdx.x & 63u, g = lane & (G - 1);
    Jac<C> acc = jac_inf<C>();
    if (active) {
        const int lo = (int)(g * CW);
        const int top = g == G - 1 ? (int)L : lo + (int)CW - 1;
        acc = var_wsum_ld<C, false>(wsum, b, (uint32_t)top);
        for (int j = top - 1; j >= lo; j--) {
            if (!acc.is_inf()) {
                acc = jac_dbl(acc);
                acc = jac_dbl(acc);
                acc = jac_dbl(acc);
                acc = jac_dbl(acc);
            }
            acc = jac_add(acc, var_wsum_ld<C, false>(wsum, b, (uint32_t)j));
        }
    }
    for (uint32_t stride = 1; stride < G; stride <<= 1) {
        jac_store(acc, lds_wave + (size_t)lane * JW);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active && (g & (2 * stride - 1)) == 0) {
            Jac<C> hi = jac_load<C>(lds_wave + (size_t)(lane + stride) * JW);
            if (!hi.is_inf())
                for (uint32_t t = 0; t < 4 * CW * stride; t++) hi = jac_dbl(hi);
            acc = jac_add(acc, hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (active && g == 0) jac_stg<C>(out + b * JW, acc);
}

// LDS-DMA (gfx950 global_load_lds_dwordx4): every lane copies 16 bytes from ITS OWN global address to
// lds_byte_addr + lane * 16 (lds_byte_addr wave-uniform).  No VGPR destination, and -- being inline asm -- not
// part of the compiler's s_waitcnt bookkeeping: the caller counts completions itself (vmcnt retires in order).
__device__ __forceinline__ void glds16(const uint32_t* gsrc, uint32_t lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // M0 is scalar: make the uniformity explicit
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}

// table entries in flight per lane, and the LDS they land in: per wave FIXED_RING slots of 2N/4 pieces of 1 KiB
// (64 lanes x 16 B) + 2 pieces for the next generator's scalar
#ifndef BPP_FIXED_RING
#define BPP_FIXED_RING 2
#endif
constexpr int FIXED_RING = BPP_FIXED_RING;
template <class C>
constexpr unsigned fixed_lds_bytes() {
    return (FIXED_BLOCK / 64) * (FIXED_RING * (2 * C::Fp::N / 4) + 2) * 1024;
}

// Which scalar arrays a k_fixed_msm launch runs over.  The verifier's arrays are one per proof (nvp = 1, first = 0,
// cnt = 1).  The batched prover keeps nvp "virtual proofs" per real proof (its L_t, R_t, A, B, commitments: each a
// MulVec over the same fixed generators) and launches either all of them or, under the Fiat-Shamir transcript, the
// cnt consecutive ones a step has just produced: flat index b of the launch -> array (b / cnt) * nvp + first + b % cnt.
//
// prover = 1 tells the kernel which virtual proof is which (prover_batch.hpp: 0 = range A, 1 + 2t / 2 + 2t = L_t / R_t,
// 2k+1, 2k+2 = wip.A, wip.B, then the commitments), because most of them are sparse over the fixed generators in a way
// that is known up front, and a lane that walks a zero scalar idles through 15 steps beside its busy neighbours:
//   * L_t and R_t each use HALF of the G_j and the complementary half of the H_j (bit k-1-t of j decides,
//     k_pb_round): their mn + 2 terms are enumerated densely, slot x -> generator by inserting that bit;
//   * a commitment has two terms (g, h);
//   * range A has scalars +1 (on G_j) or -1 (on H_j): -1 is a full-width scalar, 15 mixed additions for what is one
//     subtraction -- k_pb_init stores +1 there and this kernel negates the table entry instead.
struct VpSel {
    uint32_t nvp, first, cnt, prover;
};

// Fixed-generator part: for proof b, sum_f scalar_f * F_f through the window tables.
// scalar + bias -> W windows -> signed digits in [-half, half) (top window: unsigned) -> one table gather and
// one mixed addition per (generator, window).  No doublings, no buckets, no scatter: the 288 GB of HBM pay for
// that.  partials: ROLE 0, 2: [count][per][blockDim.x] jacobians (one per thread, summed by k_partials_fold);
// ROLE 1: [count][per] (block sums: the combined check's single MulVec, whose handful of blocks is summed by one
// lane).  ROLE also separates the launches in profiles: 0 = the batch verifier's hot path, 1 = combined check,
// 2 = batched prover.
//
// Gathers: a wave spends ~20 us on one mixed addition (7 300 instructions at two waves per SIMD), far longer than
// a random 96-byte HBM read, so latency is not the issue -- registers are: 24 VGPRs of landing space for the
// next entry and 8 for the next scalar sat on top of the ~200 the addition needs.  So the entries do not land in
// registers at all: they are LDS-DMA'd into a ring of FIXED_RING slots per lane, issued FIXED_RING additions
// ahead, and read from LDS (ds_read_b128) at the moment of use (229 -> 200 VGPRs, no vmcnt wait placed by the
// compiler inside the loop; measured -2 % on the launch).  All lanes of a block walk the same
// (generator, window) sequence in lockstep -- a lane without an entry (zero digit, or past its last generator)
// DMAs a dummy line -- so every step issues exactly 2N/4 DMA instructions per wave and a counted
// s_waitcnt vmcnt((FIXED_RING-1) * 2N/4) is all the synchronisation the ring needs.  The next generator's
// scalar travels the same way (2 pieces), one generator ahead.
// waves per SIMD the register allocator must leave room for: two for the 13-limb field (201 VGPRs), three for the 9-limb
// fields (152 / 164 VGPRs fit the 170 of a three-wave budget; tools/kernel_resources.py prints the current numbers)
template <class C>
constexpr int fixed_waves() {
    return C::Fp::NL > 9 ? BPP_FIXED_WAVES : 3;
}
// horner_tree (the form of the Horner stage in the leading `horner_blocks` blocks; chosen on the host, impl_verify.hpp):
//   0  one lane per proof (large batches)            2  eight lanes per proof (mid-size batches)
//   1  one wave per proof, window sums split by half 3  as 1, with the proof's points in VAR_GROUPS groups, and every
//                                                        block of the launch sums its own partials (lone batches)
template <class C, int ROLE = 0>
__global__ void __launch_bounds__(FIXED_BLOCK, fixed_waves<C>()) k_fixed_msm(VerifyShape s, const uint32_t* __restrict__ scalars,
                            const uint32_t* __restrict__ table, uint32_t* __restrict__ partials, uint32_t per,
                            uint32_t horner_blocks, const uint32_t* __restrict__ wsum, uint32_t* __restrict__ var_out,
                            size_t horner_count, uint32_t horner_tree, VpSel sel) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    constexpr int CH = 2 * N / 4;                              // 16-byte pieces of a table entry
    constexpr int WAVE_WORDS = (FIXED_RING * CH + 2) * 256;    // LDS words of one wave's ring + scalar buffer
    extern __shared__ __align__(16) uint32_t lds[];
    if (blockIdx.x < horner_blocks) {   // block-uniform: the Horner stage of the proof-point MSM
        if (horner_tree == 2) {   // mid-size batches: eight lanes per proof, blockDim.x / 8 proofs per block
            const size_t b = (size_t)blockIdx.x * (blockDim.x / 8) + threadIdx.x / 8;
            var_horner_group<C>(wsum, var_out, b, b < horner_count, lds + (size_t)(threadIdx.x >> 6) * 64 * JW);
        } else if (horner_tree) {   // (1 or 3) small batches: one wave per proof, ONE proof per block (the block's second wave leaves):
                             // two tree waves in one block slowed each other down (5.8 ms for 2 proofs against 4.3 ms
                             // for one); a block per proof spreads the chains over the CUs
            const size_t b = blockIdx.x;
            if constexpr (var_glv<C>()) {   // 33 windows: both waves, three lanes per window (var_horner_wave2)
                if (b < horner_count) var_horner_wave2<C>(wsum, var_out, b, lds, horner_tree == 3 ? VAR_GROUPS : 1u);
            } else {   // 65 windows (edwards25519): one wave, a lane-quad per four windows (var_horner_wave_ed)
                if (b < horner_count && threadIdx.x < 64) var_horner_wave_ed(wsum, var_out, b, horner_tree == 3 ? VAR_GROUPS : 1u);
            }
        } else {             // one lane per proof
            const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (lane < horner_count) var_horner_lane<C>(wsum, var_out, lane);
        }
        return;
    }
    // flat grid after the Horner blocks: block = proof * per + part   (`per` blocks share one proof's generators)
#ifdef BPP_XCD_REMAP   // tuning builds only (DESIGN "XCDs"): the `per` blocks of a proof on ONE XCD -- workgroups go to the 8 XCDs round-robin
    uint32_t bid = blockIdx.x - horner_blocks;
    {
        const uint32_t nb = gridDim.x - horner_blocks;
        if (nb % 8 == 0 && horner_blocks % 8 == 0) bid = (bid % 8) * (nb / 8) + bid / 8;
    }
#else
    const uint32_t bid = blockIdx.x - horner_blocks;
#endif
    const size_t b = bid / per;
    const uint32_t part = bid % per;
    const uint32_t* sc = scalars + ((b / sel.cnt) * sel.nvp + sel.first + b % sel.cnt) * (size_t)s.N * 8;
    const uint32_t mask = (1u << s.c) - 1u;
    const uint32_t stride = per * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* ring = lds + (threadIdx.x >> 6) * WAVE_WORDS;    // [slot][piece][lane][4 words]
    uint32_t* sbuf = ring + FIXED_RING * CH * 256;             // [2][lane][4 words]
    const uint32_t ring_addr = (uint32_t)reinterpret_cast<uintptr_t>(ring);   // LDS byte address (low 32 bits)
    const uint32_t sbuf_addr = ring_addr + FIXED_RING * CH * 1024;
    const uint32_t f0 = part * blockDim.x + threadIdx.x;   // the lane's index among the proof's `stride` lanes
    const uint32_t first = part * blockDim.x;
    // dense enumeration of the generators this MulVec can have non-zero scalars on (block-uniform; see VpSel)
    uint32_t NFc = s.NF, cmode = 0, cbit = 0, neg_from = 0xffffffffu;
    if (sel.prover) {
        const uint32_t vp = sel.first + (uint32_t)(b % sel.cnt);
        if (vp == 0) {
            neg_from = 2 + s.mn;
        } else if (vp <= 2 * s.k) {
            cmode = 1 + ((vp - 1) & 1u);          // 1: L_t, 2: R_t
            cbit = s.k - 1 - (vp - 1) / 2;
            NFc = 2 + s.mn;
        } else if (vp >= 2 * s.k + 3) {
            NFc = 2;
        }
    }
    auto gen_of = [&](uint32_t x) -> uint32_t {
        if (cmode == 0 || x < 2) return x;
        uint32_t u = x - 2;
        const uint32_t hm = s.mn >> 1;
        const bool isH = u >= hm;
        if (isH) u -= hm;
        const uint32_t bitv = ((cmode == 1) != isH) ? 1u : 0u;   // L: G_j with the bit set, H_j with it clear; R: the rest
        const uint32_t j = ((u >> cbit) << (cbit + 1)) | (bitv << cbit) | (u & ((1u << cbit) - 1u));
        return 2 + (isH ? s.mn : 0u) + j;
    };
    // Every lane takes G whole generators (f0, f0 + stride, ...).  The NF mod stride generators that are left
    // (g and h at n=64, m=16: 2050 = 16 * 128 + 2) are not given to two lanes as a 17th generator -- their waves
    // would run 6 % longer than the rest -- but spread window by window over all lanes in E extra steps.
    const bool spread = NFc >= stride;
    const uint32_t G = spread ? NFc / stride : (first < NFc ? 1u : 0u);
    const uint32_t T = G * s.W;                                    // whole-generator steps, the same for every lane
    const uint32_t LW = spread ? (NFc - G * stride) * s.W : 0u;    // left-over (generator, window) entries of the proof
    const uint32_t TT = T + (LW + stride - 1) / stride;            // + extra steps
    Xyzz<C> acc = xyzz_inf<C>();  // the running sum only ever receives affine points: 8M + 2S per addition

    auto dma_scalar = [&](uint32_t g) {
        const uint32_t x = f0 + g * stride;
        const uint32_t* src = x < NFc ? sc + (size_t)fixed_term_index(s, gen_of(x)) * 8 : sc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous scalar has been read out of sbuf
        glds16(src, sbuf_addr);
        glds16(src + 4, sbuf_addr + 1024);
    };
    uint32_t w[10];
    auto take_scalar = [&]() {   // sbuf -> w = scalar + bias K (K = sum_{j < W-1} half * 2^(c j))
        const uint4* q = reinterpret_cast<const uint4*>(sbuf);
        const uint4 lo = q[lane], hi = q[64 + lane];
        const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t carry = 0;
#pragma unroll
        for (int t = 0; t < 10; t++) {
            uint64_t x = (uint64_t)(t < 8 ? v[t] : 0u) + s.bias[t] + carry;
            w[t] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
    };
    // issue side: runs FIXED_RING steps ahead of the additions
    uint32_t ti = 0, gi = 0, ji = 0;   // step, generator iteration, window -- block-uniform
    uint32_t vbits = 0, nbits = 0;     // per ring slot: this lane has an entry there / it must be negated
    auto issue = [&]() {
        const uint32_t slot = ti % FIXED_RING;
        const uint32_t* src = table;   // dummy line for lanes without an entry
        uint32_t valid = 0, neg = 0;
        if (ti < T) {
            if (ji == 0 && gi > 0) {
                take_scalar();
                if (gi + 1 < G) dma_scalar(gi + 1);
            }
            const uint32_t x = f0 + gi * stride;
            // windows below the top: signed digit; top window: what is left of the value, unsigned (<= top)
            const int32_t dg = ji + 1 < s.W ? (int32_t)(w[0] & mask) - (int32_t)s.half : (int32_t)w[0];
#pragma unroll
            for (int t = 0; t < 9; t++) w[t] = (w[t] >> s.c) | (w[t + 1] << (32 - s.c));
            w[9] >>= s.c;
            if (x < NFc && dg != 0) {
                const uint32_t f = gen_of(x);
                const uint32_t mag = dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg;
                src = table + ((size_t)f * s.per_f + (size_t)ji * s.half + (mag - 1)) * 2 * N;
                valid = 1;
                neg = (dg < 0 ? 1u : 0u) ^ (f >= neg_from ? 1u : 0u);
            }
        } else if (ti < TT) {
            const uint32_t x = (ti - T) * stride + f0;   // this lane's left-over entry, if any
            if (x < LW) {
                const uint32_t l = x / s.W, jx = x - l * s.W;
                const uint32_t f = gen_of(G * stride + l);
                uint32_t we[10];
                ld_words<8>(sc + (size_t)fixed_term_index(s, f) * 8, we);   // an ordinary load: once per block
                we[8] = 0;
                we[9] = 0;
                uint32_t carry = 0;
#pragma unroll
                for (int t = 0; t < 10; t++) {
                    uint64_t v = (uint64_t)we[t] + s.bias[t] + carry;
                    we[t] = (uint32_t)v;
                    carry = (uint32_t)(v >> 32);
                }
                for (uint32_t r = 0; r < jx; r++) {
#pragma unroll
                    for (int t = 0; t < 9; t++) we[t] = (we[t] >> s.c) | (we[t + 1] << (32 - s.c));
                    we[9] >>= s.c;
                }
                const int32_t dg = jx + 1 < s.W ? (int32_t)(we[0] & mask) - (int32_t)s.half : (int32_t)we[0];
                if (dg != 0) {
                    const uint32_t mag = dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg;
                    src = table + ((size_t)f * s.per_f + (size_t)jx * s.half + (mag - 1)) * 2 * N;
                    valid = 1;
                    neg = (dg < 0 ? 1u : 0u) ^ (f >= neg_from ? 1u : 0u);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's previous entry has been read out
#pragma unroll
        for (int k = 0; k < CH; k++) glds16(src + 4 * k, ring_addr + (slot * CH + k) * 1024);
        vbits = (vbits & ~(1u << slot)) | (valid << slot);
        nbits = (nbits & ~(1u << slot)) | (neg << slot);
        ti++;
        if (++ji == s.W) {
            ji = 0;
            gi++;
        }
    };
    if (G) {
        dma_scalar(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        take_scalar();
        if (G > 1) dma_scalar(1);
    }
    for (int d = 0; d < FIXED_RING; d++) issue();
    for (uint32_t t = 0; t < TT; t++) {
        const uint32_t slot = t % FIXED_RING;
        // everything but the newest FIXED_RING - 1 steps' DMAs has landed: step t's entry is in LDS
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FIXED_RING - 1) * CH) : "memory");
        uint32_t raw[2 * N];
        const uint4* q = reinterpret_cast<const uint4*>(ring + slot * CH * 256);
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const uint4 v = q[k * 64 + lane];
            raw[4 * k] = v.x;
            raw[4 * k + 1] = v.y;
            raw[4 * k + 2] = v.z;
            raw[4 * k + 3] = v.w;
        }
        const bool valid = (vbits >> slot) & 1u, neg = (nbits >> slot) & 1u;
        const Aff<C> cur = aff_load<C>(raw);
        issue();   // step t + FIXED_RING goes into the slot just read
        if (valid) xyzz_madd_lazy(acc, cur, neg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dummy DMAs have landed: LDS may be reused
    if (ROLE != 1 && horner_tree != 3) {
        // one partial per THREAD: a tree reduction here would run 7 jacobian additions with most lanes idle
        // (~4.5 % of the block's time); k_partials_fold sums them with every lane busy.  (Not for the small batches of
        // the wave-tree mode: there the chip is mostly idle, the block's own tree runs beside the Horner chain that
        // everything waits for anyway, and k_finalize_tree then has bpp_ partials per proof to add instead of 16 bpp_:
        // horner_tree == 3, chosen while all the blocks of the launch are resident at once.)
        jac_stg<C>(partials + ((size_t)bid * blockDim.x + threadIdx.x) * JW, xyzz_to_jac(acc));
    } else {
        __syncthreads();
        Jac<C> sum = block_reduce_jac<C>(xyzz_to_jac(acc), lds);
        if (threadIdx.x == 0) jac_stg<C>(partials + (size_t)bid * JW, sum);
    }
}

// out[i] = sum of in[i * group .. i * group + group - 1]   (jacobian partials, one lane per output)
template <class C>
__global__ void __launch_bounds__(64) k_partials_fold(const uint32_t* __restrict__ in, uint32_t group,
                                                      uint32_t* __restrict__ out, size_t n_out) {
    constexpr int JW = jac_words<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const uint32_t* src = in + i * group * JW;
    Jac<C> acc = jac_ldg<C>(src);
    for (uint32_t t = 1; t < group; t++) acc = jac_add(acc, jac_ldg<C>(src + (size_t)t * JW));
    jac_stg<C>(out + i * JW, acc);
}

// ---- proof-dependent part: the 3 + 2k + m points carried by each proof / its commitments ---------------
// Straus per proof, with the ~260 doublings paid once per proof and every addition a MIXED one:
//   k_var_digits   one lane per (proof, point): the signed 4-bit digits of (scalar + 0x88..8), one byte each -- 65, or
//                  2 x 33 after the GLV split of the scalar (BLS12-381)
//   k_var_tables   one lane per (proof, point): the multiples 1P..8P as AFFINE points -- a chain of mixed
//                  additions, then one inversion (safegcd) of the product of the seven Z's
//   k_var_windows  one lane per (proof, window): sum over the proof's points of +-T[point][|digit|] in an XYZZ
//                  accumulator (8M + 2S each, no bucket reduction)
//   var_horner_lane / _group / _wave  Horner over the window sums (4 doublings per step), run by the
//                  leading blocks of k_fixed_msm's grid

// flat = 0: `scalars` is the verifier's [proof][N] array (the item's scalar sits at var_term_index);
// flat = 1: `scalars` holds one scalar per item (the combined check's w_p * s_{p,v})
template <class C>
__global__ void __launch_bounds__(256) k_var_digits(VerifyShape s, const uint32_t* __restrict__ scalars,
                                                    uint8_t* __restrict__ digits, size_t items, uint32_t flat) {
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const size_t b = item / s.NV;
    const uint32_t v = (uint32_t)(item % s.NV);
    uint32_t k[8];
    ld_words<8>(scalars + (flat ? item : b * s.N + var_term_index(s, v)) * 8, k);
    uint32_t out[VAR_DIGIT_STRIDE / 4];
#pragma unroll
    for (int t = 0; t < (int)(VAR_DIGIT_STRIDE / 4); t++) out[t] = 0;
    // digit j of a value = nibble j of (value + 0x88..8) minus 8, stored biased (the nibble itself): 0..15, 8 means 0
    if constexpr (var_glv<C>()) {
        uint32_t rem[4], q[4];
        bool nh[2];
        glv_split_signed<C>(k, rem, q, nh[0], nh[1]);   // k = +-rem +- q mu, both < 2^128 (ec.hpp)
        // 33 biased nibbles of each half: k1 at byte 0.., k2 at byte 33..; a negative half stores the negated digits
        // (16 - nibble: the digit -(nibble - 8), still |digit| <= 8)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t w[5];
            uint32_t carry = 0;
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const uint64_t x = (uint64_t)(t < 4 ? (h ? q[t] : rem[t]) : 0u) + (t < 4 ? 0x88888888u : 0x8u) + carry;
                w[t] = (uint32_t)x;
                carry = (uint32_t)(x >> 32);
            }
#pragma unroll
            for (int j = 0; j < 33; j++) {
                uint32_t nib = (w[j >> 3] >> ((j & 7) * 4)) & 15u;
                if (nh[h]) nib = 16u - nib;
                const int byte = h * 33 + j;
                out[byte >> 2] |= nib << ((byte & 3) * 8);
            }
        }
    } else {
        uint32_t w[9];
#pragma unroll
        for (int t = 0; t < 8; t++) w[t] = k[t];
        w[8] = 0;
        uint32_t carry = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const uint32_t kw = t < 8 ? 0x88888888u : 0x8u;
            uint64_t x = (uint64_t)w[t] + kw + carry;
            w[t] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
#pragma unroll
        for (int j = 0; j < (int)var_windows<C>(); j++) {
            const uint32_t nib = (w[j >> 3] >> ((j & 7) * 4)) & 15u;
            out[j >> 2] |= nib << ((j & 3) * 8);
        }
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(digits + item * VAR_DIGIT_STRIDE);
    st_words<VAR_DIGIT_STRIDE / 4>(dst, out);
}

// lane = (proof, point).  tables: [lane][8] affm.  scratch: [lane][14] field elements (Z_k and their prefix
// products, k = 2..8).
template <class C>
__global__ void __launch_bounds__(VAR_BLOCK, BPP_VAR_TABLES_WAVES) k_var_tables(const uint32_t* __restrict__ proof_pts,
                                                                        uint32_t* __restrict__ tables,
                                                                        uint32_t* __restrict__ scratch, size_t lanes) {
    using P = typename C::Fp;
    using F = Fe<P>;
    constexpr int N = P::N;
    constexpr int M = (int)VAR_MULTIPLES;
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    uint32_t* T = tables + lane * M * 2 * N;
    uint32_t* S = scratch + lane * 2 * (M - 1) * N;
    const Aff<C> p = aff_ldg<C>(proof_pts + lane * 2 * N);
    aff_stg<C>(T, p);
    if (p.is_inf()) {   // an invalid point was replaced by infinity (k_points_from_wire): every multiple is infinity
        for (int k = 1; k < M; k++) aff_stg<C>(T + (size_t)k * 2 * N, p);
        return;
    }
    affine_chain<C>(aff_dbl(p), T, M - 1, T + 2 * N, S);   // 2P .. 8P (T[0] = P, stored above)
}

// split = 1: lane = (proof b, group g of the points, half h, window j), wsum[lane] = sum_v sign * T[b][v][|digit| - 1]
//            over the group's points (half 1: of (beta x, -y))
//            -- the layout of the wave-tree Horner (small batches, the combined check), where nobody should wait for
//            2 NV additions in a row;
// split = 0: lane = (proof b, window j) adds both halves of every point itself, wsum holds var_windows sums per proof --
//            the layout of the one-lane-per-proof Horner, which then has one addition per window on its chain, not two.
template <class C>
__global__ void __launch_bounds__(VAR_BLOCK, BPP_VAR_WAVES) k_var_windows(VerifyShape s, const uint8_t* __restrict__ digits,
                                                                         const uint32_t* __restrict__ tables,
                                                                         uint32_t* __restrict__ wsum, size_t lanes,
                                                                         uint32_t split, uint32_t groups) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    constexpr uint32_t NW = var_windows<C>();
    constexpr uint32_t H = var_glv<C>() ? 2u : 1u;   // scalar halves per point (GLV: k1 on P, k2 on (beta x, -y))
    const uint32_t HL = split ? 1u : H;              // halves this lane adds
    // split: lane = ((b * groups + g) * H + h) * NW + j ; else lane = b * NW + j (groups == 1)
    const uint32_t per_proof = NW * (H / HL) * groups;
    const size_t b = lane / per_proof;
    const uint32_t g = (uint32_t)(lane % per_proof) / (NW * (H / HL));
    const uint32_t h0 = split ? (uint32_t)(lane / NW) % H : 0u;
    const uint32_t j = (uint32_t)(lane % NW);
    const uint32_t gs = (s.NV + groups - 1) / groups;                  // points per group
    const uint32_t v0 = g * gs, v1 = min(s.NV, v0 + gs);              // this lane's points
    const uint8_t* dg = digits + b * s.NV * VAR_DIGIT_STRIDE + j;   // half h's digits sit h * NW bytes further
    const uint32_t* T = tables + b * s.NV * VAR_MULTIPLES * 2 * N;
    Xyzz<C> acc = xyzz_inf<C>();
    Fe<typename C::Fp> beta;
    if constexpr (var_glv<C>()) {
#pragma unroll
        for (int i = 0; i < C::Fp::NL; i++) beta.l[i] = C::K::BETA[i];
    }
    // item u = (point u / HL, half h0 + u % HL); one entry in flight: the gather of item u + 1 is issued before the
    // addition of item u
    const uint32_t items = (v1 > v0 ? v1 - v0 : 0u) * HL;
    auto digit_of = [&](uint32_t u) -> int32_t {
        return (int32_t)dg[(size_t)(v0 + u / HL) * VAR_DIGIT_STRIDE + (h0 + u % HL) * NW] - 8;
    };
    auto entry_of = [&](uint32_t u, int32_t d) -> const uint32_t* {
        return T + ((size_t)(v0 + u / HL) * VAR_MULTIPLES + (d < 0 ? -d : d) - 1) * 2 * N;
    };
    uint32_t raw[2 * N];
    int32_t d_next = items ? digit_of(0) : 0;
    if (d_next) ld_words<2 * N>(entry_of(0, d_next), raw);
    for (uint32_t u = 0; u < items; u++) {
        const int32_t d = d_next;
        Aff<C> cur;
        if (d) cur = aff_load<C>(raw);
        if (u + 1 < items) {
            d_next = digit_of(u + 1);
            if (d_next) ld_words<2 * N>(entry_of(u + 1, d_next), raw);
        }
        bool neg = d < 0;
        if constexpr (var_glv<C>()) {
            if (d && ((h0 + u % HL) & 1u)) {   // [mu] T = (beta x, -+y); infinity (x = y = 0) stays infinity
                cur.x = fe_mul(cur.x, beta);
                fe_cond_sub_p(cur.x);
                if (glv_image_negates_y<C>()) neg = !neg;
            }
        }
        if (d) xyzz_madd_lazy(acc, cur, neg);
    }
    jac_stg<C>(wsum + lane * JW, xyzz_to_jac(acc));
}

// expected = fixed part + proof part ; verdict = expected.is_zero() ? Ok : VerificationError
// (range/mod.rs:503-509, wip.rs:320-327).  One LANE per proof adds its few folded partials (`per` fixed ones and
// `nv` from the proof points) one after the other: a wave per proof with a tree reduction spent six levels of
// mostly idle lanes on five points.  A proof with an invalid point is rejected.
template <class C>
__global__ void __launch_bounds__(64) k_finalize(const uint32_t* __restrict__ fixed_partials, uint32_t per,
                                                 const uint32_t* __restrict__ var_partials, uint32_t nv,
                                                 const uint32_t* __restrict__ bad, uint32_t* __restrict__ ok,
                                                 uint32_t* __restrict__ wire_result, size_t count) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    Jac<C> acc = jac_inf<C>();
    for (uint32_t t = 0; t < per; t++) acc = jac_add(acc, jac_ldg<C>(fixed_partials + (b * per + t) * JW));
    for (uint32_t t = 0; t < nv; t++) acc = jac_add(acc, jac_ldg<C>(var_partials + (b * nv + t) * JW));
    ok[b] = (jac_is_identity_class(acc) && !bad[b]) ? 0u : 1u;
    if (wire_result) {
        uint32_t w[2 * N + 2];
        aff_to_wire(jac_to_aff(acc), w);
#pragma unroll
        for (int t = 0; t < 2 * N + 2; t++) wire_result[b * (2 * N + 2) + t] = w[t];
    }
}

// The same for a batch too small to fill the chip (count <= HORNER_TREE_MAX): one 64-lane block per proof over the
// proof's `per` partials (the first fold pass's output) -- strided serial sums, then an LDS tree: ceil(per / 64) + 6
// additions deep instead of the ~25 of the remaining fold passes + k_finalize, which a lone proof would wait for.
template <class C>
__global__ void __launch_bounds__(64) k_finalize_tree(const uint32_t* __restrict__ fixed_partials, uint32_t per,
                                                      const uint32_t* __restrict__ var_partials,
                                                      const uint32_t* __restrict__ bad, uint32_t* __restrict__ ok,
                                                      uint32_t* __restrict__ wire_result, size_t count) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    __shared__ __align__(16) uint32_t lds[64 * JW];
    const size_t b = blockIdx.x;
    if (b >= count) return;
    Jac<C> acc = jac_inf<C>();
    for (uint32_t t = threadIdx.x; t < per; t += 64) acc = jac_add(acc, jac_ldg<C>(fixed_partials + (b * per + t) * JW));
    acc = block_reduce_jac<C>(acc, lds);
    if (threadIdx.x != 0) return;
    acc = jac_add(acc, jac_ldg<C>(var_partials + b * JW));
    ok[b] = (jac_is_identity_class(acc) && !bad[b]) ? 0u : 1u;
    if (wire_result) {
        uint32_t w[2 * N + 2];
        aff_to_wire(jac_to_aff(acc), w);
#pragma unroll
        for (int t = 0; t < 2 * N + 2; t++) wire_result[b * (2 * N + 2) + t] = w[t];
    }
}

}  // namespace bpp
