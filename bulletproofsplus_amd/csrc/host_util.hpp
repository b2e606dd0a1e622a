// host_util.hpp -- host-side helpers shared by capi.hip and prover.hpp (device buffers, error state,
// curve dispatch, scalar canonicalisation, the MulVec launcher).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bpp_amd.h"
#include "kernels.hpp"

struct bpp_ctx {
    int curve;
    int device;
    // per-stage timing of bpp_msm_device (bpp_msm_set_profiling): MSM_SLOTS passes x (stages + 1) events
    bool msm_profiling = false;
    size_t msm_passes = 0;
    std::vector<hipEvent_t> msm_events;
    uint32_t msm_shape[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // of the last bpp_msm_device call: n, items, W, q, nwide, nbuckets, L, c
    // the small-table verifiers bpp_range_verify keeps per public key (capi.hip, struct VerifyCache); owned by the
    // context that bpp_init returned -- the copies inside verifiers never touch it
    void* verify_cache = nullptr;
    bool verify_cache_off = false;
};   // (the events are destroyed by bpp_destroy: verifiers keep a COPY of their context)
constexpr size_t BPP_MSM_SLOTS = 16;

// Entry points of the per-curve implementation structs (impl_*.hpp, codec.hpp).  They are defined in class, hence
// implicitly inline, and `extern template` does not stop the compiler from instantiating an inline function in order
// to inline it: without this attribute capi.hip compiled every kernel of every curve a second time (8 minutes, 12 MB).
#define BPP_NOINL __attribute__((noinline))

// ---- error handling ------------------------------------------------------------------------------------
inline thread_local std::string g_err;
inline int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(BPP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)


namespace bpp {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        bytes = n;
        return hipMalloc(&p, n ? n : 16);
    }
    uint32_t* u32() const { return static_cast<uint32_t*>(p); }
};

template <class F>
inline int dispatch(int curve, F&& f) {
    switch (curve) {
        case BPP_BLS12_381_G1: return f(Bls12381{});
        case BPP_SECP256K1: return f(Secp256k1{});
        case BPP_ED25519: return f(Ed25519{});
        default: return fail(BPP_E_ARG, "unknown curve id");
    }
}

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// scalar (4 x u64 = 8 words) reduced mod r on the host: PrimeFieldElem values are always < r
template <class C>
void reduce_scalar_words(uint32_t* w) {
    using P = typename C::Fr;
    for (int iter = 0; iter < 4; iter++) {
        if (words_lt_mod<P>(w)) return;
        uint64_t borrow = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)w[i] - P::MODW[i] - borrow;
            w[i] = (uint32_t)t;
            borrow = (t >> 32) & 1u;
        }
    }
}

// PrimeFieldElem::new(i32) as canonical words (prime_field_elem.rs:191-195)
template <class C>
void scalar_from_i32(int32_t n, uint32_t* w) {
    using P = typename C::Fr;
    for (int i = 0; i < 8; i++) w[i] = 0;
    if (n >= 0) {
        w[0] = (uint32_t)n;
        return;
    }
    uint64_t mag = (uint64_t)(-(int64_t)n);
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t sub = (i == 0) ? (mag & 0xffffffffu) : (i == 1 ? (mag >> 32) : 0);
        uint64_t t = (uint64_t)P::MODW[i] - sub - borrow;
        w[i] = (uint32_t)t;
        borrow = (t >> 32) & 1u;
    }
}

template <class C>
constexpr int wire_words() {
    return 2 * C::Fp::N + 2;
}

// Upload wire points, convert to affm on the device.  Returns BPP_E_POINT when any is invalid.
template <class C>
int upload_points(const uint64_t* wire, size_t n, DevBuf& affm, hipStream_t st) {
    constexpr int N = C::Fp::N;
    DevBuf dw, bad;
    HIPCHK(dw.alloc(n * wire_words<C>() * 4));
    HIPCHK(bad.alloc(4));
    HIPCHK(affm.alloc(n * 2 * N * 4));
    HIPCHK(zero_words_async(bad.p, 4, st));
    if (n) {
        HIPCHK(hipMemcpyAsync(dw.p, wire, n * wire_words<C>() * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_points_from_wire<C>, dim3(cdiv(n, 128)), dim3(128), 0, st, dw.u32(), affm.u32(),
                           bad.u32(), n, 0u);
        HIPCHK(hipGetLastError());
    }
    uint32_t hbad = 0;
    HIPCHK(hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (hbad) return fail(BPP_E_POINT, "point not on curve / coordinate out of range");
    return BPP_OK;
}

template <class C>
int upload_scalars(const uint64_t* sc, size_t n, DevBuf& d, hipStream_t st) {
    std::vector<uint32_t> h(n * 8 + 8);
    if (n) std::memcpy(h.data(), sc, n * 32);
    for (size_t i = 0; i < n; i++) reduce_scalar_words<C>(h.data() + i * 8);
    HIPCHK(d.alloc(n * 32));
    if (n) HIPCHK(hipMemcpyAsync(d.p, h.data(), n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    return BPP_OK;
}

template <class C>
inline void store_generator(uint32_t* affm_words) {
    Aff<C> g = aff_generator<C>();
    aff_store(g, affm_words);
}

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

// fr_modw: the scalar-field modulus as 8 little-endian 32-bit words, fr_bits its bit length.
inline int make_shape(size_t n, size_t m, int c, const uint32_t* fr_modw, int fr_bits, VerifyShape& s) {
    const size_t mn = n * m;
    if (n == 0 || m == 0 || !is_pow2(mn)) return fail(BPP_E_ARG, "n*m must be a power of two");
    if (n > VS_MAXN || m > VS_MAXM) return fail(BPP_E_ARG, "n or m exceeds the supported maximum (64)");
    uint32_t k = 0;
    while (((size_t)1 << k) < mn) k++;
    if (k > VS_MAXK) return fail(BPP_E_ARG, "n*m too large");
    if (c < 2 || c > 20) return fail(BPP_E_ARG, "window_bits must be in [2, 20]");
    s.n = (uint32_t)n;
    s.m = (uint32_t)m;
    s.mn = (uint32_t)mn;
    s.k = k;
    s.N = (uint32_t)(2 * mn + 2 * k + m + 5);
    s.NF = (uint32_t)(2 * mn + 2);
    s.NV = (uint32_t)(3 + 2 * k + m);
    s.c = (uint32_t)c;
    // W - 1 signed windows below bit c (W-1) < fr_bits, then one unsigned top window for the rest of the value
    s.W = (uint32_t)((fr_bits - 1) / c + 1);
    s.half = 1u << (c - 1);
    for (int t = 0; t < 10; t++) s.bias[t] = 0;
    for (uint32_t j = 0; j + 1 < s.W; j++) {
        const uint32_t bit = s.c * j + (s.c - 1);
        s.bias[bit >> 5] |= 1u << (bit & 31);
    }
    // largest top digit: (r - 1 + bias) >> c (W - 1)
    uint32_t v[10];
    uint32_t carry = 0;
    for (int t = 0; t < 10; t++) {
        uint64_t x = (uint64_t)(t < 8 ? fr_modw[t] : 0u) + s.bias[t] + carry;
        v[t] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    uint32_t borrow = 1;   // - 1
    for (int t = 0; t < 10 && borrow; t++) {
        borrow = v[t] == 0 ? 1u : 0u;
        v[t] -= 1u;
    }
    const uint32_t sh = s.c * (s.W - 1);
    uint64_t top = 0;
    for (int t = 9; t >= 0; t--) {
        const int lo = 32 * t - (int)sh;   // bit position of word t after the shift
        if (lo >= 32 && v[t]) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
        if (lo > -32 && lo < 32) top |= lo >= 0 ? (uint64_t)v[t] << lo : (uint64_t)(v[t] >> (-lo));
    }
    if (top == 0 || top >= ((uint64_t)1 << 31)) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
    s.top = (uint32_t)top;
    const uint64_t per_f = (uint64_t)(s.W - 1) * s.half + s.top;
    if (per_f >> 32) return fail(BPP_E_ARG, "window table too large");
    s.per_f = (uint32_t)per_f;
    return BPP_OK;
}

// the two verifier-scalars kernels (kernels.hpp): d_prep holds count * vs_prep_bytes<C>(s) bytes
template <class C>
inline int launch_verify_scalars(const VerifyShape& s, const uint32_t* d_proof_scalars, const uint32_t* d_challenges,
                                 uint32_t ch_stride, uint32_t* d_out, size_t count, uint32_t* d_prep, hipStream_t st) {
    static bool lds_opted_in = false;   // above the default dynamic-LDS limit: opt in once (160 KB per CU on gfx950)
    if (vs_lds_bytes<C>(s) > 64 * 1024 && !lds_opted_in) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vs_expand<C>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)vs_lds_bytes<C>(s)));
        lds_opted_in = true;
    }
    hipLaunchKernelGGL(k_vs_prepare<C>, dim3(cdiv(count, 64)), dim3(64), 0, st, s, d_proof_scalars, d_challenges, ch_stride,
                       d_prep, d_out, count);
    hipLaunchKernelGGL(k_vs_expand<C>, dim3(cdiv(count, VS_PB)), dim3(VS_BLOCK), vs_lds_bytes<C>(s), st, s, d_prep, d_out,
                       count);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

// default challenges = the reference's hard-coded "transcript" (SURVEY.md 3.4):
// [y, z, e, e_1..e_k] = m == 1 ? [7, 7, 99, 7..] : [12, 23, 99, 7..]
inline void default_challenges(const VerifyShape& s, std::vector<uint32_t>& w) {
    w.assign((size_t)(3 + s.k) * 8, 0);
    w[0] = s.m == 1 ? 7 : 12;   // range/mod.rs:198 / :417
    w[8] = s.m == 1 ? 7 : 23;   // range/mod.rs:199 / :418
    w[16] = 99;                 // wip.rs:369
    for (uint32_t j = 0; j < s.k; j++) w[(size_t)(3 + j) * 8] = 7;  // wip.rs:353
}

}  // namespace bpp

