// codec.hpp -- standard COMPRESSED point encodings <-> the C ABI's wire points, on the device.
//
// A data format next to the hot path.  The reference has no serialization: only the
// commented-out size() functions (src/range/mod.rs:512-517, src/weighted_inner_product_proof.rs:384-397), which
// assume compressed points and 32-byte scalars, and the reserved ProofError::FormatError (src/errors.rs:20).
// So nothing here can be pinned by reference code -- PARITY UNPINNED; it is pinned by the public standard
// vectors of the encodings (generator encodings) and by oracle/pyref.py's big-integer restatement.
//
//   BLS12-381 G1 : 48 bytes, x big-endian; byte 0 bit 7 = compressed (always 1), bit 6 = infinity, bit 5 = y is
//                  the lexicographically larger root (y > (p-1)/2)           [ZCash / IETF pairing-friendly curves]
//   secp256k1    : 33 bytes, SEC1: 0x02 (y even) / 0x03 (y odd) || x big-endian; infinity = 33 zero bytes (SEC1's
//                  one-byte 0x00, padded to the fixed width)
//   edwards25519 : 32 bytes, ristretto255 (RFC 9496; ristretto.hpp) -- the encoding IS the prime-order group: every
//                  accepted string is an element, the 4-torsion is quotiented away
// Decompression costs one square root per point: both Weierstrass base fields have p = 3 mod 4, so
// y = (x^3 + b)^((p+1)/4), ~1.2 BITS Montgomery products, checked by squaring.
// check_subgroup (decompression): also reject curve points outside the prime-order subgroup -- BLS12-381 G1 has
// cofactor 0x396c8c00..aaab, so "on the curve" is not "in the group" (ec.hpp aff_in_prime_subgroup); a no-op for
// secp256k1 (cofactor 1) and ristretto255.
#pragma once
#include "host_util.hpp"

namespace bpp {

template <class C>
struct has_codec {
    static constexpr bool value = true;
};

template <class C>
constexpr int compressed_bytes() {
    return C::ID == 0 ? 48 : (C::ID == 1 ? 33 : 32);
}

// Version 2 of the container carries UNCOMPRESSED points: no square root at decode time (a third of the decoder's
// arithmetic) for 48 / 32 more bytes per point.  BLS12-381 G1: 96 bytes, x | y big-endian, byte 0 bit 7 = 0 (not
// compressed), bit 6 = infinity (then everything else zero), bit 5 = 0 (ZCash / IETF uncompressed form); secp256k1: SEC1
// 0x04 | x | y, 65 bytes, infinity = 65 zero bytes.  ristretto255 has no such form: version 1 only.
template <class C>
constexpr int uncompressed_bytes() {
    return C::ID == 0 ? 96 : (C::ID == 1 ? 65 : 0);
}
template <class C>
__host__ __device__ constexpr int container_point_bytes(uint32_t version) {
    return version == 2 ? uncompressed_bytes<C>() : compressed_bytes<C>();
}

// a^((p+1)/4): the square root of a quadratic residue when p = 3 mod 4; the caller checks the result by squaring.
// Fixed 2-bit windows over the public constant P::SQRTW, most significant first, with the three table entries a, a^2,
// a^3 in REGISTERS: the exponent is the same in every lane, so the window digit is wave-uniform and the entry is picked
// by uniform branches.  (A 4-bit table -- 14 + 95 products instead of 2 + 142 for BLS12-381 -- is a local array that
// the digit indexes: it lived in scratch memory and made the decoder run at 73 % of the multiplier's rate.)
template <class P>
BPP_HD Fe<P> fe_sqrt_3mod4(const Fe<P>& a) {
    const Fe<P> a2 = fe_sqr(a);
    const Fe<P> a3 = fe_mul(a2, a);
    Fe<P> acc = Fe<P>::one();
    bool started = false;
    for (int w = P::N * 16 - 1; w >= 0; w--) {
        const uint32_t d = (P::SQRTW[w >> 4] >> ((w & 15) * 2)) & 3u;
        if (started) {
            acc = fe_sqr(acc);
            acc = fe_sqr(acc);
        }
        if (d) {
            if (!started) {
                acc = d == 1 ? a : (d == 2 ? a2 : a3);
                started = true;
            } else if (d == 1) {
                acc = fe_mul(acc, a);
            } else if (d == 2) {
                acc = fe_mul(acc, a2);
            } else {
                acc = fe_mul(acc, a3);
            }
        }
    }
    return acc;
}

// canonical words > (p - 1) / 2 ?
template <class P>
BPP_HD bool words_gt_half(const uint32_t* w) {
    for (int i = P::N - 1; i >= 0; i--) {
        if (w[i] != P::HALFW[i]) return w[i] > P::HALFW[i];
    }
    return false;
}

// one thread per point: wire (canonical x | y | inf) -> compressed bytes
template <class C>
__global__ void __launch_bounds__(128) k_points_compress(const uint32_t* __restrict__ wire, uint8_t* __restrict__ out, size_t n) {
    using P = typename C::Fp;
    constexpr int N = P::N;
    constexpr int CB = compressed_bytes<C>();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = wire + i * (2 * N + 2);
    uint8_t* o = out + i * CB;
    const bool inf = (w[2 * N] | w[2 * N + 1]) != 0;
    if constexpr (C::ID == 2) {
        Aff<C> a = aff_inf<C>();
        if (!inf) {
            a.x = fe_from_canonical<P>(w);
            a.y = fe_from_canonical<P>(w + N);
        }
        rist_encode(jac_from_aff(a), o);
    } else if (C::ID == 0) {
        // x big-endian over 48 bytes; flags in the three top bits (x < p < 2^381 leaves them free)
        for (int b = 0; b < 48; b++) {
            const int k = 47 - b;   // byte k of the little-endian value
            o[b] = inf ? 0 : (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
        }
        uint8_t flags = 0x80;
        if (inf) flags |= 0x40;
        else if (words_gt_half<P>(w + N)) flags |= 0x20;
        o[0] |= flags;
    } else {
        if (inf) {
            for (int b = 0; b < 33; b++) o[b] = 0;
        } else {
            o[0] = (w[N] & 1u) ? 0x03 : 0x02;
            for (int b = 0; b < 32; b++) {
                const int k = 31 - b;
                o[1 + b] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            }
        }
    }
}

// compressed bytes -> wire point at w (2N + 2 words).  false: malformed (bad flags, x >= p, x not on the curve, not a
// ristretto255 encoding; with check_subgroup also: outside the prime-order subgroup) -- w then holds infinity, so that
// a consumer that ignores the verdict still sees a valid wire point.
template <class C, bool CHECK_SUBGROUP>
__device__ bool point_decompress(const uint8_t* __restrict__ s, uint32_t* __restrict__ w) {
    constexpr bool check_subgroup = CHECK_SUBGROUP;
    using P = typename C::Fp;
    using F = Fe<P>;
    constexpr int N = P::N;
    auto fail_point = [&]() {
        for (int t = 0; t < 2 * N + 2; t++) w[t] = 0;
        w[2 * N] = 1;
        return false;
    };
    if constexpr (C::ID == 2) {
        Aff<C> a;
        if (!rist_decode(s, a)) return fail_point();
        fe_to_canonical(a.x, w);
        fe_to_canonical(a.y, w + N);
        w[2 * N] = 0;
        w[2 * N + 1] = 0;
        return true;
    } else {
        uint32_t x[N];
        for (int t = 0; t < N; t++) x[t] = 0;
        bool inf = false, want_flag = false;
        if (C::ID == 0) {
            const uint8_t f = s[0];
            if (!(f & 0x80)) return fail_point();   // uncompressed form is not accepted here
            inf = (f & 0x40) != 0;
            want_flag = (f & 0x20) != 0;
            for (int b = 0; b < 48; b++) {
                const int k = 47 - b;
                const uint32_t v = b == 0 ? (uint32_t)(s[0] & 0x1f) : (uint32_t)s[b];
                x[k >> 2] |= v << (8 * (k & 3));
            }
            if (inf) {
                bool zero = !want_flag;
                for (int t = 0; t < N; t++) zero = zero && x[t] == 0;
                if (!zero) return fail_point();   // infinity must be 0xc0 00 .. 00
            }
        } else {
            const uint8_t f = s[0];
            bool all_zero = true;
            for (int b = 0; b < 33; b++) all_zero = all_zero && s[b] == 0;
            if (all_zero) {
                inf = true;
            } else {
                if (f != 0x02 && f != 0x03) return fail_point();
                want_flag = f == 0x03;   // y odd
                for (int b = 0; b < 32; b++) {
                    const int k = 31 - b;
                    x[k >> 2] |= (uint32_t)s[1 + b] << (8 * (k & 3));
                }
            }
        }
        if (inf) {
            for (int t = 0; t < 2 * N + 2; t++) w[t] = 0;
            w[2 * N] = 1;
            return true;
        }
        if (!words_lt_mod<P>(x)) return fail_point();
        const F xm = fe_from_canonical<P>(x);
        F b;
#pragma unroll
        for (int t = 0; t < P::NL; t++) b.l[t] = C::K::B[t];
        const F rhs = fe_add(fe_mul(fe_sqr(xm), xm), b);
        F y = fe_sqrt_3mod4(rhs);
        if (fe_sqr(y) != rhs) return fail_point();   // x^3 + b is not a square: no such point
        uint32_t yw[N];
        fe_to_canonical(y, yw);
        const bool flag = C::ID == 0 ? words_gt_half<P>(yw) : (yw[0] & 1u) != 0;
        if (flag != want_flag) {
            y = fe_neg(y);
            fe_to_canonical(y, yw);
        }
        if constexpr (check_subgroup) {
            Aff<C> a;
            a.x = xm;
            a.y = y;
            if (!aff_in_prime_subgroup(a)) return fail_point();
        }
        // y = 0 cannot carry the "larger" / "odd" flag; it does not occur on these curves (no point of order 2)
        for (int t = 0; t < N; t++) {
            w[t] = x[t];
            w[N + t] = yw[t];
        }
        w[2 * N] = 0;
        w[2 * N + 1] = 0;
        return true;
    }
}

// uncompressed bytes -> wire point at w.  false: malformed (flags / prefix, a coordinate >= p, not on the curve) -- w then
// holds infinity.  Membership of the prime-order subgroup is k_records_subgroup's business, as for version 1.
template <class C>
__device__ bool point_uncompressed_read(const uint8_t* __restrict__ s, uint32_t* __restrict__ w) {
    using P = typename C::Fp;
    constexpr int N = P::N;
    auto fail_point = [&]() {
        for (int t = 0; t < 2 * N + 2; t++) w[t] = 0;
        w[2 * N] = 1;
        return false;
    };
    if constexpr (C::ID == 2) {
        return fail_point();
    } else {
        constexpr int FB = N * 4;                       // bytes per coordinate
        constexpr int OFF = C::ID == 0 ? 0 : 1;         // SEC1 prefix byte
        uint32_t x[N], y[N];
        for (int t = 0; t < N; t++) x[t] = y[t] = 0;
        bool inf = false;
        if (C::ID == 0) {
            const uint8_t f = s[0];
            if (f & 0xA0) return fail_point();          // compressed form / sign bit have no place here
            inf = (f & 0x40) != 0;
        } else {
            bool all_zero = true;
            for (int b = 0; b < 65; b++) all_zero = all_zero && s[b] == 0;
            inf = all_zero;
            if (!inf && s[0] != 0x04) return fail_point();
        }
        for (int b = 0; b < FB; b++) {
            const int k = FB - 1 - b;
            const uint32_t vx = (C::ID == 0 && b == 0) ? (uint32_t)(s[0] & 0x1f) : (uint32_t)s[OFF + b];
            x[k >> 2] |= vx << (8 * (k & 3));
            y[k >> 2] |= (uint32_t)s[OFF + FB + b] << (8 * (k & 3));
        }
        if (inf) {
            bool zero = true;
            for (int t = 0; t < N; t++) zero = zero && x[t] == 0 && y[t] == 0;
            if (!zero) return fail_point();             // infinity is 0x40 00 .. 00 / 65 zero bytes
            for (int t = 0; t < 2 * N + 2; t++) w[t] = 0;
            w[2 * N] = 1;
            return true;
        }
        for (int t = 0; t < N; t++) {
            w[t] = x[t];
            w[N + t] = y[t];
        }
        w[2 * N] = 0;
        w[2 * N + 1] = 0;
        Aff<C> a;
        if (!aff_from_wire<C>(w, a)) return fail_point();   // coordinates < p, on the curve
        return true;
    }
}

// one thread per point: compressed bytes -> wire; ok[i] = 0 valid, 1 malformed (see point_decompress)
template <class C>
__global__ void __launch_bounds__(64, 2) k_points_decompress(const uint8_t* __restrict__ in, uint32_t* __restrict__ wire,
                                                          uint32_t* __restrict__ ok, size_t n, uint32_t check_subgroup) {
    constexpr int N = C::Fp::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool good = check_subgroup ? point_decompress<C, true>(in + i * compressed_bytes<C>(), wire + i * (2 * N + 2))
                                     : point_decompress<C, false>(in + i * compressed_bytes<C>(), wire + i * (2 * N + 2));
    ok[i] = good ? 0u : 1u;
}

// ---- the proof container on the device (layout: include/bpp_amd.h "serialized proofs") ---------------------------
constexpr uint32_t CONTAINER_HDR = 12;   // "BPP+" | version | curve | n | m | k | 3 reserved zero bytes
template <class C>
__host__ __device__ constexpr size_t container_bytes(uint32_t k, uint32_t version = 1) {
    return CONTAINER_HDR + (size_t)(3 + 2 * k) * container_point_bytes<C>(version) + 96;
}

// One lane per point of every proof's verification record [A, wip.A, wip.B, L.., R.., V_0..V_{m-1}]: the 3 + 2k points
// of the container and the m commitments, decompressed WITH the subgroup check straight into the layout
// bpp_verifier_run reads.  The lane of a proof's first point also checks the header and the canonicity of r', s',
// delta' and copies them out.  status[p] (zeroed by the caller) becomes non-zero when anything of proof p is rejected.
template <class C>
__global__ void __launch_bounds__(64, 2) k_container_decode(VerifyShape s, const uint8_t* __restrict__ proofs,
                                                         const uint8_t* __restrict__ commitments,
                                                         uint32_t* __restrict__ records, uint32_t* __restrict__ scalars,
                                                         uint32_t* __restrict__ status, size_t count, uint32_t version) {
    constexpr int N = C::Fp::N;
    const int CB = container_point_bytes<C>(version);   // version 2: uncompressed points (and commitments)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * s.NV) return;
    const size_t p = i / s.NV;
    const uint32_t t = (uint32_t)(i - p * s.NV);
    const uint32_t npp = 3 + 2 * s.k;
    const uint8_t* rec = proofs + p * container_bytes<C>(s.k, version);
    const uint8_t* src = t < npp ? rec + CONTAINER_HDR + (size_t)t * CB : commitments + (p * s.m + (t - npp)) * CB;
    // the G1 membership test is a kernel of its own (k_records_subgroup below): two chains of 64 doublings beside the
    // square root's window table in one kernel cost 256 VGPRs and scratch
    bool good = version == 2 ? point_uncompressed_read<C>(src, records + i * (2 * N + 2))
                             : point_decompress<C, false>(src, records + i * (2 * N + 2));
    if (t == 0) {
        const uint8_t hdr[CONTAINER_HDR] = {'B', 'P', 'P', '+', (uint8_t)version, (uint8_t)C::ID, (uint8_t)s.n, (uint8_t)s.m, (uint8_t)s.k, 0, 0, 0};
        for (uint32_t b = 0; b < CONTAINER_HDR; b++) good = good && rec[b] == hdr[b];
        const uint8_t* sc = rec + CONTAINER_HDR + (size_t)npp * CB;
        for (int e = 0; e < 3; e++) {
            uint32_t w[8];
            for (int q = 0; q < 8; q++)
                w[q] = (uint32_t)sc[32 * e + 4 * q] | ((uint32_t)sc[32 * e + 4 * q + 1] << 8) |
                       ((uint32_t)sc[32 * e + 4 * q + 2] << 16) | ((uint32_t)sc[32 * e + 4 * q + 3] << 24);
            good = good && words_lt_mod<typename C::Fr>(w);   // one encoding per scalar
            for (int q = 0; q < 8; q++) scalars[(p * 3 + e) * 8 + q] = w[q];
        }
    }
    if (!good) atomicOr(status + p, 1u);
}

// One lane per decoded record point: outside the prime-order subgroup (BLS12-381 G1: csrc/ec.hpp aff_in_prime_subgroup) =>
// the proof's status word is raised and the point replaced by infinity.  A no-op launch on the other curves.
template <class C>
__global__ void __launch_bounds__(64, 2) k_records_subgroup(uint32_t* __restrict__ records, uint32_t* __restrict__ status,
                                                            uint32_t per_proof, size_t npoints) {
    using P = typename C::Fp;
    constexpr int N = P::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoints) return;
    uint32_t* w = records + i * (2 * N + 2);
    if (w[2 * N] | w[2 * N + 1]) return;   // infinity
    Aff<C> a;
    a.x = fe_from_canonical<P>(w);
    a.y = fe_from_canonical<P>(w + N);
    if (aff_in_prime_subgroup(a)) return;
    for (int t = 0; t < 2 * N + 2; t++) w[t] = t == 2 * N ? 1u : 0u;
    atomicOr(status + i / per_proof, 1u);
}

// ok[p] = BPP_FORMAT_ERROR where the decoder rejected proof p: ProofError::FormatError takes precedence over the
// MulVec verdict
template <class C>
__global__ void __launch_bounds__(256) k_container_status(const uint32_t* __restrict__ status, uint32_t* __restrict__ ok,
                                                          size_t count) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < count && status[p]) ok[p] = BPP_FORMAT_ERROR;
}

template <class C>
struct CodecImpl {
    static constexpr int N = C::Fp::N;
    static constexpr int WW = 2 * N + 2;

    static int compress(const uint64_t* points, size_t n, uint8_t* out);

    // device to device: `d_in` n x CB bytes, `d_wire` n wire points, `d_ok` n words
    static int decompress_device(const uint8_t* d_in, size_t n, uint64_t* d_wire, uint32_t* d_ok, hipStream_t st,
                                 bool check_subgroup = false);

    static int decompress(const uint8_t* in, size_t n, uint64_t* out_points, uint32_t* out_ok);
};

// ---- definitions: compiled only by the translation unit that instantiates the struct (tu_*.hip defines
// BPP_IMPL_DEFINITIONS); capi.hip sees the declarations above and the `extern template` below, so it does not
// compile the kernels a second time ----
#ifdef BPP_IMPL_DEFINITIONS
template <class C>
int CodecImpl<C>::compress(const uint64_t* points, size_t n, uint8_t* out) {
    if constexpr (!has_codec<C>::value) {
        return fail(BPP_E_ARG, "compressed encoding is not offered for this curve");
    } else {
        if (n == 0) return BPP_OK;
        constexpr int CB = compressed_bytes<C>();
        DevBuf dw, db;
        HIPCHK(dw.alloc(n * WW * 4));
        HIPCHK(db.alloc(n * CB));
        HIPCHK(hipMemcpy(dw.p, points, n * WW * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_points_compress<C>, dim3(cdiv(n, 128)), dim3(128), 0, nullptr, dw.u32(),
                           static_cast<uint8_t*>(db.p), n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(out, db.p, n * CB, hipMemcpyDeviceToHost));
        return BPP_OK;
    }
}

template <class C>
int CodecImpl<C>::decompress_device(const uint8_t* d_in, size_t n, uint64_t* d_wire, uint32_t* d_ok, hipStream_t st,
                                    bool check_subgroup) {
    if constexpr (!has_codec<C>::value) {
        return fail(BPP_E_ARG, "compressed encoding is not offered for this curve");
    } else {
        if (n == 0) return BPP_OK;
        hipLaunchKernelGGL(k_points_decompress<C>, dim3(cdiv(n, 64)), dim3(64), 0, st, d_in,
                           reinterpret_cast<uint32_t*>(d_wire), d_ok, n, check_subgroup ? 1u : 0u);
        HIPCHK(hipGetLastError());
        return BPP_OK;
    }
}

template <class C>
int CodecImpl<C>::decompress(const uint8_t* in, size_t n, uint64_t* out_points, uint32_t* out_ok) {
    if constexpr (!has_codec<C>::value) {
        return fail(BPP_E_ARG, "compressed encoding is not offered for this curve");
    } else {
        if (n == 0) return BPP_OK;
        constexpr int CB = compressed_bytes<C>();
        DevBuf dw, db, dk;
        HIPCHK(dw.alloc(n * WW * 4));
        HIPCHK(db.alloc(n * CB));
        HIPCHK(dk.alloc(n * 4));
        HIPCHK(hipMemcpy(db.p, in, n * CB, hipMemcpyHostToDevice));
        int rc = decompress_device(static_cast<const uint8_t*>(db.p), n, static_cast<uint64_t*>(dw.p), dk.u32(), nullptr);
        if (rc) return rc;
        HIPCHK(hipMemcpy(out_points, dw.p, n * WW * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out_ok, dk.p, n * 4, hipMemcpyDeviceToHost));
        return BPP_OK;
    }
}

#endif  // BPP_IMPL_DEFINITIONS


extern template struct CodecImpl<Bls12381>;
extern template struct CodecImpl<Secp256k1>;
extern template struct CodecImpl<Ed25519>;

}  // namespace bpp
