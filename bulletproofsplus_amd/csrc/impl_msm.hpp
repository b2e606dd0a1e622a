// impl_msm.hpp -- host orchestration of the MulVec / scalar-multiplication kernels, one instantiation per
// curve (tu_msm_*.hip).  Serves bpp_msm, bpp_msm_batch, bpp_scalar_mul_batch, bpp_pk_new, bpp_commit,
// bpp_range_verify (single proof, no tables) and the device unit-test hooks.
#pragma once
#include "hash_to_group.hpp"
#include "host_util.hpp"
#include "pippenger.hpp"

namespace bpp {

template <class P>
__global__ void __launch_bounds__(64) k_dbg_field(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wa[P::N], wb[P::N], wr[P::N];
    for (int t = 0; t < P::N; t++) {
        wa[t] = a[i * P::N + t];
        wb[t] = b[i * P::N + t];
    }
    Fe<P> x = fe_from_canonical<P>(wa), y = fe_from_canonical<P>(wb), r;
    switch (op) {
        case 0: r = fe_mul(x, y); break;
        case 1: r = fe_add(x, y); break;
        case 2: r = fe_sub(x, y); break;
        case 3: r = fe_inv(x); break;
        case 4: r = fe_sqr(x); break;
        default: r = fe_neg(x); break;
    }
    fe_to_canonical(r, wr);
    for (int t = 0; t < P::N; t++) out[i * P::N + t] = wr[t];
}

template <class C>
__global__ void __launch_bounds__(64) k_dbg_point(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    constexpr int WW = 2 * C::Fp::N + 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wa[WW], wb[WW], wr[WW];
    for (int t = 0; t < WW; t++) {
        wa[t] = a[i * WW + t];
        wb[t] = b[i * WW + t];
    }
    Aff<C> p, q;
    aff_from_wire<C>(wa, p);
    aff_from_wire<C>(wb, q);
    Jac<C> r;
    switch (op) {
        case 0: r = jac_add(jac_from_aff(p), jac_from_aff(q)); break;
        case 1: r = jac_madd(jac_from_aff(p), q); break;
        case 2: r = jac_dbl(jac_from_aff(p)); break;
        case 3: r = jac_madd(jac_dbl(jac_from_aff(p)), q); break;  // non-trivial Z
        case 4: r = jac_add(jac_dbl(jac_from_aff(p)), jac_dbl(jac_from_aff(q))); break;
        default: {  // XYZZ accumulator: inf + p + q + p
            Xyzz<C> acc = xyzz_inf<C>();
            acc = xyzz_madd(acc, p);
            acc = xyzz_madd(acc, q);
            acc = xyzz_madd(acc, p);
            r = xyzz_to_jac(acc);
            break;
        }
    }
    aff_to_wire(jac_to_aff(r), wr);
    for (int t = 0; t < WW; t++) out[i * WW + t] = wr[t];
}

template <class C>
struct MsmImpl {
    static constexpr int N = C::Fp::N;
    static constexpr int JW = jac_words<C>();
    static constexpr int WW = 2 * N + 2;  // 32-bit words per wire point
    static constexpr int PW = WW / 2;     // 64-bit words per wire point

    // `count` MulVecs over device-resident affm points / canonical scalars -> wire points on the host
    static int msm_batch_dev(const uint32_t* d_scalars, const uint32_t* d_points, const std::vector<uint64_t>& offsets,
                             uint64_t* out, hipStream_t st);

    // one large MulVec through the bucket method (pippenger.hpp), every buffer in HBM, asynchronous on `st`:
    // d_scalars n canonical scalars, d_wire_points n wire points, d_out_wire one wire point, d_status (may be null) one
    // word: 1 when a point was invalid.  window_bits = 0 picks the width from n.
    static constexpr size_t PIPPENGER_MIN_N = 4096;
    static size_t msm_workspace_bytes(size_t n, int window_bits);
    static int msm_device(const uint32_t* d_scalars, const uint32_t* d_wire_points, size_t n, int window_bits,
                          uint32_t* d_out_wire, uint32_t* d_status, void* d_ws, size_t ws_bytes, hipStream_t st,
                          bpp_ctx* ctx = nullptr);

    // the same from host pointers (uploads, runs, downloads); explicit window width (tests sweep it)
    static int msm_pippenger(const uint64_t* scalars, const uint64_t* points, size_t n, int window_bits, uint64_t* out);

    static int msm_batch(const uint64_t* scalars, const uint64_t* points, const uint32_t* lens, size_t count,
                         uint64_t* out);

    static int scalar_mul_batch(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t* out);

    // PublicKey::new (publickey.rs:21-48)
    static int pk_new(size_t length, uint64_t* out_gh, uint64_t* out_G, uint64_t* out_H);
    static int pk_hashed(const uint8_t* label, size_t label_len, size_t length, uint64_t* out_gh, uint64_t* out_G,
                         uint64_t* out_H);

    // RangeProver::commit (prover.rs:28-42)
    static int commit(const uint64_t* gh, uint64_t v, const uint64_t* gamma, uint64_t* out);

    // RangeProof::verify for one proof, without window tables: verifier scalars on the device, then the
    // MulVec exactly as the reference assembles it (range/mod.rs:480-509 / wip.rs:297-327).
    static int range_verify_single(const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                                   const uint64_t* proof_points, size_t k, const uint64_t* proof_scalars,
                                   const uint64_t* V);

    // field: 0 = base field, 1 = scalar field
    static int debug_field_op(int field, int op, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out);

    static int debug_point_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
};

// ---- definitions: compiled only by the translation unit that instantiates the struct (tu_*.hip defines
// BPP_IMPL_DEFINITIONS); capi.hip sees the declarations above and the `extern template` below, so it does not
// compile the kernels a second time ----
#ifdef BPP_IMPL_DEFINITIONS
template <class C>
int MsmImpl<C>::msm_batch_dev(const uint32_t* d_scalars, const uint32_t* d_points, const std::vector<uint64_t>& offsets,
                         uint64_t* out, hipStream_t st) {
    const size_t count = offsets.size() - 1;
    if (count == 0) return BPP_OK;
    size_t maxlen = 0;
    for (size_t c = 0; c < count; c++) maxlen = std::max<size_t>(maxlen, offsets[c + 1] - offsets[c]);
    const unsigned block = MSM_BLOCK;
    unsigned gx = std::max(1u, std::min(cdiv(maxlen, block), 1024u));
    DevBuf doff, dpart, dout;
    HIPCHK(doff.alloc(offsets.size() * 8));
    HIPCHK(dpart.alloc(count * gx * JW * 4));
    HIPCHK(dout.alloc(count * WW * 4));
    HIPCHK(hipMemcpyAsync(doff.p, offsets.data(), offsets.size() * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_msm_naive_partial<C>, dim3(gx, (unsigned)count), dim3(block), block * JW * 4, st,
                       d_scalars, d_points, static_cast<const uint64_t*>(doff.p), dpart.u32());
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_jac_reduce<C>, dim3(cdiv(count, 64)), dim3(64), 0, st, dpart.u32(), gx, dout.u32(), count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, count * WW * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return BPP_OK;
}

template <class C>
size_t MsmImpl<C>::msm_workspace_bytes(size_t n, int window_bits) {
    if (n == 0) return 256;
    if (n >= ((size_t)1 << 28)) return 0;
    PipShape ps;
    if (pip_shape_for<C>(n, window_bits ? window_bits : pip_pick_c<C>(n), ps)) return 0;
    return pip_workspace<C>(ps).total;
}

template <class C>
int MsmImpl<C>::msm_device(const uint32_t* d_scalars, const uint32_t* d_wire_points, size_t n, int window_bits,
                           uint32_t* d_out_wire, uint32_t* d_status, void* d_ws, size_t ws_bytes, hipStream_t st,
                           bpp_ctx* ctx) {
    if (n >= ((size_t)1 << 28)) return fail(BPP_E_ARG, "n too large");
    if (n == 0) {   // Point::zero()
        if (d_status) HIPCHK(zero_words_async(d_status, 4, st));
        hipLaunchKernelGGL(k_pip_zero_point<C>, dim3(1), dim3(64), 0, st, d_out_wire);
        HIPCHK(hipGetLastError());
        return BPP_OK;
    }
    PipShape ps;
    int rc = pip_shape_for<C>(n, window_bits ? window_bits : pip_pick_c<C>(n), ps);
    if (rc) return rc;
    if (ws_bytes < pip_workspace<C>(ps).total) return fail(BPP_E_ARG, "workspace too small (bpp_msm_workspace_bytes)");
    hipEvent_t* ev = nullptr;
    if (ctx) {
        const uint32_t sh[8] = {ps.n, ps.items, ps.W, ps.q, ps.nwide, ps.nbuckets, ps.L, ps.c};
        std::memcpy(ctx->msm_shape, sh, sizeof sh);
        if (ctx->msm_profiling) ev = ctx->msm_events.data() + (ctx->msm_passes++ % BPP_MSM_SLOTS) * (PIP_STAGES + 1);
    }
    HIPCHK(pip_launch<C>(ps, d_scalars, d_wire_points, static_cast<uint8_t*>(d_ws), d_out_wire, d_status, st, ev));
    return BPP_OK;
}

template <class C>
int MsmImpl<C>::msm_pippenger(const uint64_t* scalars, const uint64_t* points, size_t n, int window_bits, uint64_t* out) {
    if (window_bits && (window_bits < 2 || window_bits > 16)) return fail(BPP_E_ARG, "window_bits must be in [2, 16]");
    if (n == 0) {
        std::memset(out, 0, WW * 4);
        out[PW - 1] = 1;
        return BPP_OK;
    }
    const size_t wsb = msm_workspace_bytes(n, window_bits);
    if (wsb == 0) return fail(BPP_E_ARG, "n too large");
    DevBuf dsc, dpt, ws, dout, dst;
    HIPCHK(dsc.alloc(n * 32));
    HIPCHK(dpt.alloc(n * WW * 4));
    HIPCHK(ws.alloc(wsb));
    HIPCHK(dout.alloc(WW * 4));
    HIPCHK(dst.alloc(4));
    HIPCHK(hipMemcpy(dsc.p, scalars, n * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dpt.p, points, n * WW * 4, hipMemcpyHostToDevice));
    int rc = msm_device(dsc.u32(), dpt.u32(), n, window_bits, dout.u32(), dst.u32(), ws.p, wsb, nullptr);
    if (rc) return rc;
    uint32_t bad = 0;
    HIPCHK(hipMemcpy(out, dout.p, WW * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&bad, dst.p, 4, hipMemcpyDeviceToHost));
    if (bad) return fail(BPP_E_POINT, "point not on curve / coordinate out of range");
    return BPP_OK;
}

template <class C>
int MsmImpl<C>::msm_batch(const uint64_t* scalars, const uint64_t* points, const uint32_t* lens, size_t count,
                     uint64_t* out) {
    std::vector<uint64_t> off(count + 1, 0);
    for (size_t c = 0; c < count; c++) off[c + 1] = off[c] + lens[c];
    const size_t total = off[count];
    if (total && (!scalars || !points)) return fail(BPP_E_ARG, "null scalars/points");
    if (count == 1 && total >= PIPPENGER_MIN_N) return msm_pippenger(scalars, points, total, 0, out);
    DevBuf dsc, dpt;
    int rc = upload_scalars<C>(scalars, total, dsc, nullptr);
    if (rc) return rc;
    rc = upload_points<C>(points, total, dpt, nullptr);
    if (rc) return rc;
    return msm_batch_dev(dsc.u32(), dpt.u32(), off, out, nullptr);
}

template <class C>
int MsmImpl<C>::scalar_mul_batch(const uint64_t* scalars, const uint64_t* points, size_t n, uint64_t* out) {
    if (n == 0) return BPP_OK;
    DevBuf dsc, dpt, dres, dw;
    int rc = upload_scalars<C>(scalars, n, dsc, nullptr);
    if (rc) return rc;
    rc = upload_points<C>(points, n, dpt, nullptr);
    if (rc) return rc;
    HIPCHK(dres.alloc(n * 2 * N * 4));
    HIPCHK(dw.alloc(n * WW * 4));
    hipLaunchKernelGGL(k_scalar_mul<C>, dim3(cdiv(n, 64)), dim3(64), 0, nullptr, dsc.u32(), dpt.u32(),
                       (size_t)(2 * N), dres.u32(), n);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_points_to_wire<C>, dim3(cdiv(n, 64)), dim3(64), 0, nullptr, dres.u32(), dw.u32(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dw.p, n * WW * 4, hipMemcpyDeviceToHost));
    return BPP_OK;
}

template <class C>
int MsmImpl<C>::pk_new(size_t length, uint64_t* out_gh, uint64_t* out_G, uint64_t* out_H) {
    const size_t total = 2 + 2 * length;
    // scalars: [1, 2, 3(i+1).., 5(i+1)..] with Rust's `i as i32` wrap (publickey.rs:29-39)
    std::vector<uint32_t> sc(total * 8);
    scalar_from_i32<C>(1, sc.data());
    scalar_from_i32<C>(2, sc.data() + 8);
    for (size_t i = 0; i < length; i++) {
        const uint32_t ip1 = (uint32_t)i + 1u;
        scalar_from_i32<C>((int32_t)(ip1 * 3u), sc.data() + (2 + i) * 8);
        scalar_from_i32<C>((int32_t)(ip1 * 5u), sc.data() + (2 + length + i) * 8);
    }
    uint32_t g[2 * N];
    Aff<C> gen = aff_generator<C>();
    aff_store(gen, g);
    DevBuf dsc, dg, dres, dw;
    HIPCHK(dsc.alloc(total * 32));
    HIPCHK(dg.alloc(sizeof g));
    HIPCHK(dres.alloc(total * 2 * N * 4));
    HIPCHK(dw.alloc(total * WW * 4));
    HIPCHK(hipMemcpy(dsc.p, sc.data(), total * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dg.p, g, sizeof g, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scalar_mul<C>, dim3(cdiv(total, 64)), dim3(64), 0, nullptr, dsc.u32(), dg.u32(),
                       (size_t)0, dres.u32(), total);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_points_to_wire<C>, dim3(cdiv(total, 64)), dim3(64), 0, nullptr, dres.u32(), dw.u32(),
                       total);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> hw(total * WW);
    HIPCHK(hipMemcpy(hw.data(), dw.p, hw.size() * 4, hipMemcpyDeviceToHost));
    const size_t pb = WW * 4;
    std::memcpy(out_gh, hw.data(), 2 * pb);
    if (length) {
        std::memcpy(out_G, hw.data() + 2 * WW, length * pb);
        std::memcpy(out_H, hw.data() + (2 + length) * WW, length * pb);
    }
    return BPP_OK;
}

// g = the base point; h, G_i, H_i hashed to the group (hash_to_group.hpp): nobody knows their discrete logarithms
template <class C>
int MsmImpl<C>::pk_hashed(const uint8_t* label, size_t label_len, size_t length, uint64_t* out_gh, uint64_t* out_G,
                          uint64_t* out_H) {
    const size_t total = 1 + 2 * length;
    const H2gSeed seed = h2g_seed(C::ID, label, label_len);
    DevBuf dw;
    HIPCHK(dw.alloc(total * WW * 4));
    hipLaunchKernelGGL(k_hash_to_group<C>, dim3(cdiv(total, 64)), dim3(64), 0, nullptr, seed, (uint32_t)length, dw.u32());
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> hw(total * WW);
    HIPCHK(hipMemcpy(hw.data(), dw.p, hw.size() * 4, hipMemcpyDeviceToHost));
    uint32_t gw[WW];
    aff_to_wire(aff_generator<C>(), gw);
    const size_t pb = WW * 4;
    std::memcpy(out_gh, gw, pb);
    std::memcpy(reinterpret_cast<uint8_t*>(out_gh) + pb, hw.data(), pb);
    if (length) {
        std::memcpy(out_G, hw.data() + WW, length * pb);
        std::memcpy(out_H, hw.data() + (1 + length) * WW, length * pb);
    }
    return BPP_OK;
}

template <class C>
int MsmImpl<C>::commit(const uint64_t* gh, uint64_t v, const uint64_t* gamma, uint64_t* out) {
    uint32_t sc[16];
    scalar_from_i32<C>((int32_t)(uint32_t)v, sc);  // `v as i32`, prover.rs:37
    std::memcpy(sc + 8, gamma, 32);
    const uint32_t len = 2;
    return msm_batch(reinterpret_cast<const uint64_t*>(sc), gh, &len, 1, out);
}

template <class C>
int MsmImpl<C>::range_verify_single(const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                               const uint64_t* proof_points, size_t k, const uint64_t* proof_scalars,
                               const uint64_t* V) {
    VerifyShape s;
    int rc = make_shape(n, m, 8, C::Fr::MODW, C::Fr::BITS, s);
    if (rc) return rc;
    if (k != s.k) return BPP_VERIFICATION_ERROR;  // wip.rs:335-337
    std::vector<uint64_t> pts((size_t)s.N * PW);
    auto put = [&](size_t idx, const uint64_t* src, size_t cnt) {
        std::memcpy(pts.data() + idx * PW, src, cnt * PW * 8);
    };
    const uint64_t* pA = proof_points;
    const uint64_t* pWA = proof_points + PW;
    const uint64_t* pWB = proof_points + 2 * PW;
    if (m == 1) {  // wip.rs:309-311
        put(0, pWB, 1);
        put(1, pWA, 1);
        put(2, pA, 1);
    } else {  // range/mod.rs:492-494
        put(0, pA, 1);
        put(1, pWA, 1);
        put(2, pWB, 1);
    }
    put(3, gh, 2);
    put(5, proof_points + 3 * PW, 2 * k);
    put(5 + 2 * k, G, s.mn);
    put(5 + 2 * k + s.mn, H, s.mn);
    put(5 + 2 * k + 2 * s.mn, V, m);
    DevBuf dpt, dps, dch, dsc;
    rc = upload_points<C>(pts.data(), s.N, dpt, nullptr);
    if (rc == BPP_E_POINT) return BPP_VERIFICATION_ERROR;
    if (rc) return rc;
    rc = upload_scalars<C>(proof_scalars, 3, dps, nullptr);
    if (rc) return rc;
    std::vector<uint32_t> ch;
    default_challenges(s, ch);
    HIPCHK(dch.alloc(ch.size() * 4));
    HIPCHK(hipMemcpy(dch.p, ch.data(), ch.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(dsc.alloc((size_t)s.N * 32));
    DevBuf dprep;
    HIPCHK(dprep.alloc(vs_prep_bytes<C>(s)));
    rc = launch_verify_scalars<C>(s, dps.u32(), dch.u32(), 0u, dsc.u32(), (size_t)1, dprep.u32(), nullptr);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> off = {0, s.N};
    std::vector<uint64_t> res(PW);
    rc = msm_batch_dev(dsc.u32(), dpt.u32(), off, res.data(), nullptr);
    if (rc) return rc;
    bool identity = res[PW - 1] != 0;
    if constexpr (C::ID == 2) {
        // the identity of ristretto255's quotient group: x = 0 or y = 0 (ristretto.hpp ed_is_identity_class)
        bool x0 = true, y0 = true;
        for (int i = 0; i < PW / 2; i++) {
            x0 = x0 && res[i] == 0;
            y0 = y0 && res[PW / 2 + i] == 0;
        }
        identity = identity || x0 || y0;
    }
    return identity ? BPP_OK : BPP_VERIFICATION_ERROR;
}

template <class C>
int MsmImpl<C>::debug_field_op(int field, int op, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
    auto run = [&](auto pv) -> int {
        using P = decltype(pv);
        DevBuf da, db, dout;
        const size_t bytes = n * P::N * 4;
        HIPCHK(da.alloc(bytes));
        HIPCHK(db.alloc(bytes));
        HIPCHK(dout.alloc(bytes));
        HIPCHK(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_dbg_field<P>, dim3(cdiv(n, 64)), dim3(64), 0, nullptr, op, da.u32(), db.u32(),
                           dout.u32(), n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
        return BPP_OK;
    };
    if (field == 0) return run(typename C::Fp{});
    return run(typename C::Fr{});
}

template <class C>
int MsmImpl<C>::debug_point_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    DevBuf da, db, dout;
    const size_t bytes = n * WW * 4;
    HIPCHK(da.alloc(bytes));
    HIPCHK(db.alloc(bytes));
    HIPCHK(dout.alloc(bytes));
    HIPCHK(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg_point<C>, dim3(cdiv(n, 64)), dim3(64), 0, nullptr, op, da.u32(), db.u32(), dout.u32(),
                       n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return BPP_OK;
}

#endif  // BPP_IMPL_DEFINITIONS


extern template struct MsmImpl<Bls12381>;
extern template struct MsmImpl<Secp256k1>;
extern template struct MsmImpl<Ed25519>;

}  // namespace bpp
