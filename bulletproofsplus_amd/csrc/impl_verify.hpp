// impl_verify.hpp -- the batch verifier (bpp_verifier_*): window tables in HBM + one pass of the hot path
// over a device-resident batch.  One instantiation per curve (tu_verify_*.hip).
#pragma once
#include <mutex>

#include "codec.hpp"
#include "combined.hpp"
#include "fixed_launch.hpp"
#include "host_util.hpp"
#include "pippenger.hpp"
#include "prover_batch.hpp"

// stages of one pass, in launch order (bpp_verifier_profile reports one duration per stage)
enum { BPP_STAGE_FROM_WIRE = 0, BPP_STAGE_SCALARS, BPP_STAGE_FIXED_MSM, BPP_STAGE_VAR_MSM, BPP_STAGE_FINALIZE,
       BPP_NUM_STAGES };
constexpr int BPP_PROFILE_SLOTS = 64;  // passes remembered by the event ring

struct bpp_verifier {
    bpp_ctx ctx;
    bpp::VerifyShape s;
    bpp::DevBuf table;       // window tables
    bpp::DevBuf challenges;  // default challenges
    bpp::TranscriptState tr0;   // transcript state after the domain, curve, (n, m) and generator digest
    size_t table_bytes = 0;
    // bpp_verifier_set_subgroup_check: wire points outside the prime-order subgroup count as invalid points (off by
    // default: the in-memory API takes points that are in the group by construction, as the reference's mcl values are)
    bool check_subgroup = false;
    // optional per-stage HIP-event timing: one (begin, end) event pair per stage and remembered pass
    bool profiling = false;
    std::vector<hipEvent_t> events;  // BPP_PROFILE_SLOTS x BPP_NUM_STAGES x 2
    size_t passes_recorded = 0;
    unsigned last_blocks_per_proof = 0;
    // side stream of the lone-batch path (run(): the proof-point tables are built beside the verifier scalars), created
    // on first use
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::mutex aux_mu;   // the fork .. join of one pass is enqueued as a whole (passes of several host threads share the events)
    ~bpp_verifier() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (aux) (void)hipStreamDestroy(aux);
    }
};

namespace bpp {

inline unsigned blocks_per_proof(const VerifyShape& s, size_t count) {
    const unsigned maxb = cdiv(s.NF, FIXED_BLOCK);
    // small batches: aim at ~2^18 resident threads; at least one block, at most one generator per thread
    size_t tpp = ((size_t)1 << 18) / (count ? count : 1);
    tpp = std::max<size_t>(FIXED_BLOCK, std::min<size_t>(tpp, s.NF));
    const unsigned b_lat = cdiv(tpp, FIXED_BLOCK);
    // large batches: the chip holds 1024 blocks at a time, and a launch of only a few such rounds ends with a
    // long, mostly idle tail (8 rounds of 5 ms blocks: the last 3.5 ms ran 64 blocks).  Aim at >= 32 rounds,
    // but keep at least four generators per lane so that a block's prologue stays amortised.
    const unsigned b_thr = std::min<unsigned>(cdiv((size_t)32768, count ? count : 1),
                                              std::max<unsigned>(1, s.NF / (FIXED_BLOCK * 4)));
#ifdef BPP_FORCE_PER   // tuning builds only (A/B of the launch geometry on one box)
    if (count >= 1024) return std::max(1u, std::min<unsigned>(BPP_FORCE_PER, maxb));
#endif
    return std::max(1u, std::min(std::max(b_lat, b_thr), maxb));
}

// The wave-per-proof (tree) Horner serves the batches too small to fill the chip: one block per proof, so that the
// chains spread over the CUs (with two tree waves per block they slowed each other down: 5.8 ms for 2 proofs against
// 4.3 ms for one).  Above HORNER_TREE_MAX proofs the one-lane-per-proof form rides beside the fixed-generator blocks.
#ifndef BPP_HORNER_TREE_MAX
#define BPP_HORNER_TREE_MAX 256
#endif
constexpr size_t HORNER_TREE_MAX = BPP_HORNER_TREE_MAX;
constexpr unsigned FOLD_GROUP = 8;    // thread partials summed by one lane of k_partials_fold (first pass)
constexpr unsigned FOLD_GROUP2 = 4;   // ... and of the second pass

struct WsLayout {
    size_t pts, bad, scalars, prep, fthread, fpart, fpart2, fpart3, vpart, vdig, vwsum, vtbl, vscr, total;
};

template <class C>
struct VerifyImpl {
    static constexpr int N = C::Fp::N;
    static constexpr int JW = jac_words<C>();
    static constexpr int WW = 2 * N + 2;
    static constexpr int PW = WW / 2;

    static WsLayout ws_layout(const VerifyShape& s, size_t count) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        WsLayout w;
        size_t o = 0;
        w.pts = o;
        o += al(count * s.NV * 2 * N * 4);
        w.bad = o;
        o += al(count * 4);
        w.scalars = o;
        o += al(count * (size_t)s.N * 32);
        w.prep = o;
        o += al(count * vs_prep_bytes<C>(s));                      // per-proof constants of the verifier-scalars kernels
        w.fthread = o;
        o += al(count * blocks_per_proof(s, count) * FIXED_BLOCK * JW * 4);                // one partial per thread
        w.fpart = o;
        o += al(count * blocks_per_proof(s, count) * (FIXED_BLOCK / FOLD_GROUP) * JW * 4);  // folded 8 to 1
        w.fpart2 = o;
        o += al(count * blocks_per_proof(s, count) * (FIXED_BLOCK / FOLD_GROUP / FOLD_GROUP2) * JW * 4);  // then 4 to 1
        w.fpart3 = o;
        o += al(count * (FIXED_BLOCK / FOLD_GROUP / FOLD_GROUP2) * JW * 4);   // then the blocks of a proof: 4 per proof
        w.vpart = o;
        o += al(count * JW * 4);                                   // one jacobian per proof
        w.vdig = o;
        o += al(count * s.NV * VAR_DIGIT_STRIDE);                  // digit bytes per proof point (65, or 2 x 33)
        w.vwsum = o;
        o += al(count * var_wsums_max<C>() * JW * 4);                 // window sums
        w.vtbl = o;
        o += al(count * s.NV * VAR_MULTIPLES * 2 * N * 4);         // 1P..8P of every proof point, affine
        w.vscr = o;
        o += al(count * s.NV * 2 * (VAR_MULTIPLES - 1) * N * 4);   // Z's and their prefix products while normalising
        w.total = o;
        return w;
    }

    static int create(const bpp_ctx& ctx, const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                      int window_bits, bpp_verifier** out);

    static int run(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                   const uint64_t* d_challenges, uint32_t* d_ok, void* d_workspace, size_t workspace_bytes,
                   uint64_t* d_out_scalars, uint64_t* d_out_result, hipStream_t st);

    // form of the Horner stage for a pass over `count` proofs (k_fixed_msm's horner_tree: 0, 1 or 2)
    static uint32_t horner_form(const VerifyShape& s, size_t count) {
        const bool small_job = (double)count * ((double)s.NF * s.W / 7.0e9 + 9.2e-8) < 2.0e-3;
        return count <= HORNER_TREE_MAX ? 1u : (small_job ? 2u : 0u);
    }
    // The proof points' tables (k_var_tables) need the points but not the scalars: fork_tables builds them on the verifier's
    // side stream, beside whatever the caller enqueues next on `st` (the scalar kernels); join_tables makes `st` wait for them
    // (before k_var_windows).  The fork .. join of one pass is enqueued under the verifier's mutex: the side stream and its
    // two events are shared by the passes of every stream and host thread.
    static int fork_tables(bpp_verifier* v, hipStream_t st, const uint32_t* w_pts, uint32_t* w_vt, uint32_t* w_vscr, size_t items,
                           std::unique_lock<std::mutex>& lock) {
        lock = std::unique_lock<std::mutex>(v->aux_mu);
        if (!v->aux) {
            HIPCHK(hipStreamCreateWithFlags(&v->aux, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&v->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&v->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(v->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(v->aux, v->ev_fork, 0));
        hipLaunchKernelGGL(k_var_tables<C>, dim3(cdiv(items, VAR_BLOCK)), dim3(VAR_BLOCK), 0, v->aux, w_pts, w_vt, w_vscr, items);
        HIPCHK(hipEventRecord(v->ev_join, v->aux));
        return BPP_OK;
    }
    static int join_tables(bpp_verifier* v, hipStream_t st, std::unique_lock<std::mutex>& lock) {
        HIPCHK(hipStreamWaitEvent(st, v->ev_join, 0));
        lock.unlock();
        return BPP_OK;
    }
    static int finish(bpp_verifier* v, uint8_t* ws, const WsLayout& L, size_t count, const uint32_t* w_sc,
                      const uint32_t* w_vw, const uint32_t* w_bad, uint32_t* d_ok, uint32_t* d_out_result, uint32_t tree,
                      bool lone, hipStream_t st, hipEvent_t* ev);

    static int derive_challenges(bpp_verifier* v, const uint64_t* d_points, size_t count, uint64_t* d_challenges,
                                 hipStream_t st);

    // ---- RangeProof::verify over SERIALIZED proofs resident in HBM (codec.hpp: the container) ---------------
    // workspace = decoded records | scalars | decoder status | challenges (transcript mode) | run()'s workspace
    struct SerLayout {
        size_t records, scalars, status, challenges, run, total;
    };
    // group != 0: the verification behind the decoder is the grouped check (run_grouped) with groups of that size
    static SerLayout ser_layout(const VerifyShape& s, size_t count, uint32_t group = 0) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        SerLayout w;
        size_t o = 0;
        w.records = o;
        o += al(count * s.NV * WW * 4);
        w.scalars = o;
        o += al(count * 96);
        w.status = o;
        o += al(count * 4);
        w.challenges = o;
        o += al(count * (size_t)(3 + s.k) * 32);
        w.run = o;
        o += group ? group_layout(s, count, group).total : ws_layout(s, count).total;
        w.total = o;
        return w;
    }
    // the grouped check's arguments when it stands behind the decoder (run_serialized)
    struct GroupedArgs {
        const uint8_t* weight_key;
        uint64_t index_base;
        const uint64_t* d_weights;
        uint32_t group;
        uint64_t* h_stats;
    };
    // d_proofs: count x container_bytes ; d_commitments: count x m compressed points ; d_ok: 0 Ok / 1 VerificationError /
    // 2 FormatError per proof.  Everything on `st`, no host synchronisation.
    // grouped != null: verdicts through run_grouped (synchronises `st`)
    static int run_serialized(bpp_verifier* v, const uint8_t* d_proofs, const uint8_t* d_commitments, size_t count,
                              bool transcript, uint32_t* d_ok, void* d_workspace, size_t workspace_bytes, hipStream_t st,
                              uint32_t version = 1, const GroupedArgs* grouped = nullptr);

    // ---- combined batch check (combined.hpp) ------------------------------------------------------------
    struct CombLayout {
        size_t pts, bad, scalars, prep, weights, comb_sc, fpart, var_sc, vdig, vtbl, vscr, vwsum, vfold, total;
        unsigned fixed_blocks;
    };
    static constexpr uint32_t COMB_FOLD_GROUP = 4;    // proofs whose window sums one lane of k_comb_window_fold adds
    static CombLayout comb_layout(const VerifyShape& s, size_t count) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        CombLayout w;
        const size_t items = count * s.NV;
        w.fixed_blocks = blocks_per_proof(s, 1);
        size_t o = 0;
        w.pts = o;
        o += al(items * 2 * N * 4);
        w.bad = o;
        o += al(count * 4);
        w.scalars = o;
        o += al(count * (size_t)s.N * 32);
        w.prep = o;
        o += al(count * vs_prep_bytes<C>(s));
        w.weights = o;
        o += al(count * 32);
        w.comb_sc = o;
        o += al((size_t)s.N * 32);
        w.fpart = o;
        o += al((size_t)(w.fixed_blocks + 1) * JW * 4);            // fixed-generator block sums + the Horner result
        w.var_sc = o;
        o += al(items * 32);
        w.vdig = o;
        o += al(items * VAR_DIGIT_STRIDE);
        w.vtbl = o;
        o += al(items * VAR_MULTIPLES * 2 * N * 4);
        w.vscr = o;
        o += al(items * 2 * (VAR_MULTIPLES - 1) * N * 4);
        w.vwsum = o;
        o += al(count * var_wsums<C>() * JW * 4);
        w.vfold = o;
        o += al((size_t)cdiv(count, COMB_FOLD_GROUP) * var_wsums<C>() * JW * 4);
        w.total = o;
        return w;
    }

    // d_out_partial: one jacobian (3N words, opaque to the caller) = this batch's weighted sum
    static int run_combined(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                            const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                            const uint64_t* d_weights, uint32_t* d_out_partial, uint32_t* d_ok, void* d_workspace,
                            size_t workspace_bytes, hipStream_t st);

    // ---- grouped check (combined.hpp): per-proof verdicts from one weighted check per GROUP of proofs ------
    struct GroupLayout {
        size_t pts, bad, scalars, prep, weights, var_sc, vdig, vtbl, vscr, vwsum, vfold,   // per proof, as the combined check's
            grows, gbad, gok, tail,                                                         // per group
            list, x_pts, x_sc, x_ch, x_ok, x_run,                                           // the exact second pass
            total;
        size_t groups, slice;
    };
    static constexpr size_t GROUP_EXACT_SLICE = 2048;   // proofs of failing groups re-verified per exact pass
    static GroupLayout group_layout(const VerifyShape& s, size_t count, uint32_t group) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        GroupLayout w;
        const size_t items = count * s.NV;
        w.groups = cdiv(count, group);
        w.slice = std::min<size_t>(std::max<size_t>(count, 1), GROUP_EXACT_SLICE);
        size_t o = 0;
        w.pts = o;
        o += al(items * 2 * N * 4);
        w.bad = o;
        o += al(count * 4);
        w.scalars = o;
        o += al(count * (size_t)s.N * 32);
        w.prep = o;
        o += al(count * vs_prep_bytes<C>(s));
        w.weights = o;
        o += al(count * 32);
        w.var_sc = o;
        o += al(items * 32);
        w.vdig = o;
        o += al(items * VAR_DIGIT_STRIDE);
        w.vtbl = o;
        o += al(items * VAR_MULTIPLES * 2 * N * 4);
        w.vscr = o;
        o += al(items * 2 * (VAR_MULTIPLES - 1) * N * 4);
        w.vwsum = o;
        o += al(count * var_wsums<C>() * JW * 4);
        w.vfold = o;
        o += al((size_t)cdiv(count, 2) * var_wsums<C>() * JW * 4);
        w.grows = o;
        o += al(w.groups * (size_t)s.N * 32);
        w.gbad = o;
        o += al(w.groups * 4);
        w.gok = o;
        o += al(w.groups * 4);
        w.tail = o;
        o += ws_layout(s, w.groups).total;
        w.list = o;
        o += al(w.slice * 4);
        w.x_pts = o;
        o += al(w.slice * s.NV * WW * 4);
        w.x_sc = o;
        o += al(w.slice * 96);
        w.x_ch = o;
        o += al(w.slice * (size_t)(3 + s.k) * 32);
        w.x_ok = o;
        o += al(w.slice * 4);
        w.x_run = o;
        // the exact pass runs over however many proofs the failing groups hold, and a SMALLER batch can need a LARGER
        // workspace (more blocks per proof, blocks_per_proof): room for the worst count up to the slice
        size_t xrun = 0;
        for (size_t c = 1; c <= w.slice; c++) xrun = std::max(xrun, ws_layout(s, c).total);
        o += xrun;
        w.total = o;
        return w;
    }
    // d_out_verdicts: count words, 0 = Ok / 1 = VerificationError, as bpp_verifier_run writes them.  h_stats (host, may be
    // null): [groups that failed their weighted check, proofs re-verified by the exact pass].  Synchronises `st`.
    static int run_grouped(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                           const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                           const uint64_t* d_weights, uint32_t group, uint32_t* d_out_verdicts, uint64_t* h_stats,
                           void* d_workspace, size_t workspace_bytes, hipStream_t st);
    // the same in two calls, so that ONE host thread can keep several batches in flight (a stream and a workspace each):
    // grouped_begin only enqueues pass 1; grouped_finish (same buffers, same stream) synchronises, reads the groups' verdicts
    // and runs pass 2.  run_grouped = begin + finish.
    static int grouped_begin(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                             const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                             const uint64_t* d_weights, uint32_t group, uint32_t* d_out_verdicts, void* d_workspace,
                             size_t workspace_bytes, hipStream_t st);
    static int grouped_finish(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                              const uint64_t* d_challenges, uint32_t group, uint32_t* d_out_verdicts, uint64_t* h_stats,
                              void* d_workspace, size_t workspace_bytes, hipStream_t st);

    // ---- batched prover (prover_batch.hpp) -------------------------------------------------------------
    // Device-resident form: values, gammas, outputs and workspace are device buffers, nothing touches the host
    // and nothing synchronises.  The batch is processed in chunks that reuse one workspace.
    struct ProveLayout {
        size_t a, b, cG, cH, pwy, con, vps, part, part1, part2, vout, trst, ch, blind, total;
        size_t chunk;
        unsigned per;
    };
    static size_t prove_chunk(const VerifyShape& s, size_t count) {
        const uint32_t nvp = pb_num_vps(s.k, s.m);
        const size_t chunk_max = std::max<size_t>(1, std::min<size_t>(2048, ((size_t)12 << 30) / ((size_t)nvp * s.N * 32)));
        return std::min(chunk_max, std::max<size_t>(count, 1));
    }
    static ProveLayout prove_layout(const VerifyShape& s, size_t count) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        ProveLayout w;
        const uint32_t nvp = pb_num_vps(s.k, s.m);
        w.chunk = prove_chunk(s, count);
        const size_t nv_total = w.chunk * nvp;
        w.per = blocks_per_proof(s, nv_total);
        const size_t vec = w.chunk * (size_t)s.mn * 32;
        size_t o = 0;
        w.a = o;
        o += al(vec);
        w.b = o;
        o += al(vec);
        w.cG = o;
        o += al(vec);
        w.cH = o;
        o += al(vec);
        w.pwy = o;
        o += al(vec);
        w.con = o;
        o += al(w.chunk * (size_t)pb_consts_elems(s.k) * 32);
        w.vps = o;
        o += al(nv_total * (size_t)s.N * 32);
        w.part = o;
        o += al(nv_total * w.per * FIXED_BLOCK * JW * 4);                                  // one partial per thread
        w.part1 = o;
        o += al(nv_total * w.per * (FIXED_BLOCK / FOLD_GROUP) * JW * 4);                   // folded 8 to 1
        w.part2 = o;
        o += al(nv_total * w.per * (FIXED_BLOCK / FOLD_GROUP / FOLD_GROUP2) * JW * 4);     // then 4 to 1
        w.vout = o;
        o += al(w.chunk * (size_t)s.m * WW * 4);   // the commitments of a chunk when the caller does not want them
        w.trst = o;
        o += al(w.chunk * 32);                      // transcript states (Fiat-Shamir mode)
        w.ch = o;
        o += al(w.chunk * (size_t)(3 + s.k) * 32);  // ... and the challenge blocks when the caller does not want them
        w.blind = o;
        o += al(w.chunk * (size_t)pb_blind_elems(s.k) * 32);   // blinding scalars expanded from the caller's key
        w.total = o;
        return w;
    }
    // d_values: count x m u64 ; d_gammas: count x m scalars ; d_out_points: count x (3 + 2k) wire points ;
    // d_out_scalars: count x 3 scalars ; d_out_V: count x m wire points (may be null).
    // fs = false: the reference's constant challenges, every MulVec of the batch in ONE k_fixed_msm launch.
    // fs = true : challenges from the transcript (transcript.hpp): A and the commitments first, then y, z; each round's
    //             L_t, R_t before e_t; wip.A, wip.B before e -- 3 + k smaller launches and the hashing steps between
    //             them.  d_out_challenges (count x (3 + k) scalars, may be null) receives [y, z, e, e_1..e_k].
    // Blinding (alpha, r, s, delta, eta, d_L[t], d_R[t] per proof): d_blinding (count x (5 + 2k) canonical scalars), or
    // blind_key (32 bytes, host) expanded on the device with the global proof index index_base + p (k_pb_blind), or --
    // both null -- the reference's literals (range/mod.rs:94,256; wip.rs:94-95,175-178), whose proofs hide nothing.
    static int prove_batch_device(bpp_verifier* v, const uint64_t* d_values, const uint64_t* d_gammas, size_t count,
                                  uint64_t* d_out_points, uint64_t* d_out_scalars, uint64_t* d_out_V, bool fs,
                                  uint64_t* d_out_challenges, void* d_workspace, size_t workspace_bytes, hipStream_t st,
                                  const uint8_t* blind_key = nullptr, uint64_t index_base = 0,
                                  const uint64_t* d_blinding = nullptr);

    // host buffers in, host buffers out
    static int prove_batch(bpp_verifier* v, const uint64_t* values, const uint64_t* gammas, size_t count,
                           uint64_t* out_points, uint64_t* out_scalars, uint64_t* out_V, bool fs,
                           const uint8_t* blind_key = nullptr, uint64_t index_base = 0);

    // d_partials: n partials as bpp_verifier_run_combined wrote them (jacobian + validity word each)
    static int sum_partials(const uint32_t* d_partials, size_t n, uint32_t* d_ok, hipStream_t st);
};

// ---- definitions: compiled only by the translation unit that instantiates the struct (tu_*.hip defines
// BPP_IMPL_DEFINITIONS); capi.hip sees the declarations above and the `extern template` below, so it does not
// compile the kernels a second time ----
#ifdef BPP_IMPL_DEFINITIONS
template <class C>
int VerifyImpl<C>::create(const bpp_ctx& ctx, const uint64_t* gh, const uint64_t* G, const uint64_t* H, size_t n, size_t m,
                  int window_bits, bpp_verifier** out) {
    VerifyShape s;
    int rc = make_shape(n, m, window_bits, C::Fr::MODW, C::Fr::BITS, s);
    if (rc) return rc;
    std::vector<uint64_t> fixed((size_t)s.NF * PW);
    std::memcpy(fixed.data(), gh, 2 * PW * 8);
    std::memcpy(fixed.data() + 2 * PW, G, (size_t)s.mn * PW * 8);
    std::memcpy(fixed.data() + (size_t)(2 + s.mn) * PW, H, (size_t)s.mn * PW * 8);
    DevBuf dfixed;
    rc = upload_points<C>(fixed.data(), s.NF, dfixed, nullptr);
    if (rc) return rc;
    bpp_verifier* v = new bpp_verifier();
    v->ctx = ctx;
    v->s = s;
    const size_t entries = (size_t)s.NF * s.per_f;
    v->table_bytes = entries * 2 * N * 4;
    hipError_t e = v->table.alloc(v->table_bytes);
    if (e != hipSuccess) {
        delete v;
        return fail(BPP_E_NOMEM, std::string("window table allocation failed: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_tbl_bases<C>, dim3(cdiv(s.NF, 64)), dim3(64), 0, nullptr, s, dfixed.u32(), v->table.u32());
    // fill in slabs of generators: one thread per run of TBL_RUN entries, 2 * TBL_RUN field elements of scratch each
    const uint32_t runs_f = tbl_runs_per_generator(s);
    const uint32_t slab = (uint32_t)std::max<size_t>(1, ((size_t)1 << 21) / runs_f);
    DevBuf tbl_scratch;
    e = tbl_scratch.alloc((size_t)std::min<uint32_t>(slab, s.NF) * runs_f * 2 * TBL_RUN * N * 4);
    if (e != hipSuccess) {
        delete v;
        return fail(BPP_E_NOMEM, std::string("table scratch allocation failed: ") + hipGetErrorString(e));
    }
    for (uint32_t f0 = 0; f0 < s.NF; f0 += slab) {
        const uint32_t f1 = std::min<uint32_t>(s.NF, f0 + slab);
        const size_t total = (size_t)(f1 - f0) * runs_f;
        hipLaunchKernelGGL(k_tbl_fill<C>, dim3(cdiv(total, 64)), dim3(64), 0, nullptr, s, v->table.u32(),
                           tbl_scratch.u32(), f0, f1);
    }
    tr_initial_state<C>(s.n, s.m, reinterpret_cast<const uint32_t*>(fixed.data()), s.NF, v->tr0.st);
    std::vector<uint32_t> ch;
    default_challenges(s, ch);
    e = v->challenges.alloc(ch.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(v->challenges.p, ch.data(), ch.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        delete v;
        return fail(BPP_E_HIP, std::string("table build failed: ") + hipGetErrorString(e));
    }
    *out = v;
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::run(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
               const uint64_t* d_challenges, uint32_t* d_ok, void* d_workspace, size_t workspace_bytes,
               uint64_t* d_out_scalars, uint64_t* d_out_result, hipStream_t st) {
    const VerifyShape& s = v->s;
    const WsLayout L = ws_layout(s, count);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    uint32_t* w_pts = reinterpret_cast<uint32_t*>(ws + L.pts);
    uint32_t* w_bad = reinterpret_cast<uint32_t*>(ws + L.bad);
    uint32_t* w_sc = d_out_scalars ? reinterpret_cast<uint32_t*>(d_out_scalars)
                                   : reinterpret_cast<uint32_t*>(ws + L.scalars);
    uint32_t* w_vt = reinterpret_cast<uint32_t*>(ws + L.vtbl);
    const unsigned bpp_ = blocks_per_proof(s, count);
    const size_t npts = count * s.NV;
    hipEvent_t* ev = nullptr;   // ev[2 * stage], ev[2 * stage + 1]
    if (v->profiling) {
        ev = v->events.data() + (v->passes_recorded % BPP_PROFILE_SLOTS) * (BPP_NUM_STAGES * 2);
        v->passes_recorded++;
    }
    auto mark = [&](int idx, hipStream_t s_) { return ev ? hipEventRecord(ev[idx], s_) : hipSuccess; };
    v->last_blocks_per_proof = bpp_;
    HIPCHK(zero_words_async(w_bad, count * 4, st));
    HIPCHK(mark(2 * BPP_STAGE_FROM_WIRE, st));
    hipLaunchKernelGGL(k_points_from_wire<C>, dim3(cdiv(npts, 128)), dim3(128), 0, st,
                       reinterpret_cast<const uint32_t*>(d_points), w_pts, w_bad, npts, s.NV, v->check_subgroup ? 1u : 0u);
    HIPCHK(mark(2 * BPP_STAGE_FROM_WIRE + 1, st));
    // The tables of the proof points need the points only, not the scalars -- a chain of seven additions and an inversion
    // that nothing else waits for yet -- so they are built on a side stream beside the scalar kernels and join before the
    // window sums.  For a lone batch both are latency bound; for a large one k_vs_prepare is (one lane per proof: 128 waves
    // for 8 192 proofs, 0.2 ms with the chip nearly empty) and the tables fill what it leaves.
#ifndef BPP_SIDE_TABLES_ALWAYS
#define BPP_SIDE_TABLES_ALWAYS 1
#endif
    const bool side_tables = BPP_SIDE_TABLES_ALWAYS || (count * blocks_per_proof(s, count) <= 1024 && count <= HORNER_TREE_MAX);
    std::unique_lock<std::mutex> aux_lock;
    if (side_tables) {
        int rc_f = fork_tables(v, st, w_pts, w_vt, reinterpret_cast<uint32_t*>(ws + L.vscr), npts, aux_lock);
        if (rc_f) return rc_f;
    }
    const uint32_t* ch = d_challenges ? reinterpret_cast<const uint32_t*>(d_challenges) : v->challenges.u32();
    const uint32_t ch_stride = d_challenges ? (3 + s.k) * 8 : 0;
    HIPCHK(mark(2 * BPP_STAGE_SCALARS, st));
    {
        int rc_vs = launch_verify_scalars<C>(s, reinterpret_cast<const uint32_t*>(d_scalars), ch, ch_stride, w_sc, count,
                                             reinterpret_cast<uint32_t*>(ws + L.prep), st);
        if (rc_vs) return rc_vs;
    }
    HIPCHK(mark(2 * BPP_STAGE_SCALARS + 1, st));
    // proof-point MSM: digits, per-point tables, window sums (all arithmetic bound, so they simply run in
    // sequence); its latency-bound Horner stage rides in the first blocks of the fixed-generator launch
    uint8_t* w_vd = ws + L.vdig;
    uint32_t* w_vw = reinterpret_cast<uint32_t*>(ws + L.vwsum);
    // Horner stage: one lane per proof, or -- while the waves are there to spare -- one wave per proof (tree); the tree
    // reads the window sums split by scalar half (k_var_windows)
    // (mode 1, window sums split by scalar half); in between, eight lanes per proof (mode 2) while the launch would
    // otherwise wait for the one-lane chains: the chain is ~2 ms, the eight-lane form costs ~0.09 us of chip time per
    // proof on top of the fixed-generator work (~7 G mixed additions/s) -- measured on (64,1): better at 4 096 proofs,
    // worse at 8 192
    const uint32_t tree = horner_form(s, count);
    // a launch whose blocks are all resident at once (<= 1024): only latency counts -- its blocks also sum their own
    // partials (mode 3 of k_fixed_msm) and the points of a proof are dealt to VAR_GROUPS lanes per window
    const bool lone = tree == 1 && count * bpp_ <= 1024;
    const uint32_t vgroups = lone ? VAR_GROUPS : 1u;
    const size_t vlanes = count * (tree == 1 ? var_wsums<C>() * vgroups : var_windows<C>());
    HIPCHK(mark(2 * BPP_STAGE_VAR_MSM, st));
    hipLaunchKernelGGL(k_var_digits<C>, dim3(cdiv(npts, 256)), dim3(256), 0, st, s, w_sc, w_vd, npts, 0u);
    if (side_tables) {
        int rc_j = join_tables(v, st, aux_lock);
        if (rc_j) return rc_j;
    } else
        hipLaunchKernelGGL(k_var_tables<C>, dim3(cdiv(npts, VAR_BLOCK)), dim3(VAR_BLOCK), 0, st, w_pts, w_vt,
                           reinterpret_cast<uint32_t*>(ws + L.vscr), npts);
    hipLaunchKernelGGL(k_var_windows<C>, dim3(cdiv(vlanes, VAR_BLOCK)), dim3(VAR_BLOCK), 0, st, s, w_vd, w_vt, w_vw,
                       vlanes, tree == 1 ? 1u : 0u, vgroups);
    HIPCHK(mark(2 * BPP_STAGE_VAR_MSM + 1, st));
    return finish(v, ws, L, count, w_sc, w_vw, w_bad, d_ok, reinterpret_cast<uint32_t*>(d_out_result), tree, lone, st, ev);
}

// The rest of a pass, from the scalars [count][N] and the proof points' window sums: the fixed-generator MulVec with the
// Horner stage in its leading blocks, the folds of its partials, the verdicts.  ws / L: a workspace laid out for `count`.
template <class C>
int VerifyImpl<C>::finish(bpp_verifier* v, uint8_t* ws, const WsLayout& L, size_t count, const uint32_t* w_sc,
                          const uint32_t* w_vw, const uint32_t* w_bad, uint32_t* d_ok, uint32_t* d_out_result, uint32_t tree,
                          bool lone, hipStream_t st, hipEvent_t* ev) {
    const VerifyShape& s = v->s;
    const unsigned bpp_ = blocks_per_proof(s, count);
    uint32_t* w_fp = reinterpret_cast<uint32_t*>(ws + L.fpart);
    uint32_t* w_vp = reinterpret_cast<uint32_t*>(ws + L.vpart);
    auto mark = [&](int idx, hipStream_t s_) { return ev ? hipEventRecord(ev[idx], s_) : hipSuccess; };
    HIPCHK(mark(2 * BPP_STAGE_FIXED_MSM, st));
    const unsigned hb = tree == 1 ? (unsigned)count : cdiv(count, tree == 2 ? FIXED_BLOCK / 8 : FIXED_BLOCK);
    uint32_t* w_ft = reinterpret_cast<uint32_t*>(ws + L.fthread);
    launch_fixed_msm<C, 0>((unsigned)(hb + count * bpp_), st, s, w_sc, v->table.u32(), w_ft, bpp_, hb, w_vw, w_vp, count,
                           lone ? 3u : tree, VpSel{1u, 0u, 1u, 0u});
    HIPCHK(mark(2 * BPP_STAGE_FIXED_MSM + 1, st));
    HIPCHK(mark(2 * BPP_STAGE_FINALIZE, st));
    // 128 per-thread partials per block -> 16 -> 4 (-> 4 per proof), every lane of the fold kernels busy;
    // k_finalize adds the rest
    const unsigned folded = bpp_ * (FIXED_BLOCK / FOLD_GROUP);
    const unsigned folded2 = folded / FOLD_GROUP2;
    uint32_t* w_fp2 = reinterpret_cast<uint32_t*>(ws + L.fpart2);
    if (tree == 1) {   // a small batch waits for latency, not throughput: one block per proof finishes the sum as a tree,
                       // over the per-block partials k_fixed_msm left (lone) or over the first fold pass's output
        if (!lone)
            hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(count * folded, 64)), dim3(64), 0, st, w_ft, FOLD_GROUP, w_fp,
                               count * folded);
        hipLaunchKernelGGL(k_finalize_tree<C>, dim3((unsigned)count), dim3(64), 0, st, lone ? w_ft : w_fp,
                           lone ? bpp_ : folded, w_vp, w_bad, d_ok, d_out_result, count);
        HIPCHK(mark(2 * BPP_STAGE_FINALIZE + 1, st));
        HIPCHK(hipGetLastError());
        return BPP_OK;
    }
    hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(count * folded, 64)), dim3(64), 0, st, w_ft, FOLD_GROUP, w_fp,
                       count * folded);
    hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(count * folded2, 64)), dim3(64), 0, st, w_fp, FOLD_GROUP2, w_fp2,
                       count * folded2);
    const uint32_t* w_last = w_fp2;
    unsigned last = folded2;
    if (bpp_ > 1) {   // several blocks per proof: one more pass, so that k_finalize always sees 4 partials
        uint32_t* w_fp3 = reinterpret_cast<uint32_t*>(ws + L.fpart3);
        last = folded2 / bpp_;
        hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(count * last, 64)), dim3(64), 0, st, w_fp2, bpp_, w_fp3,
                           count * last);
        w_last = w_fp3;
    }
    hipLaunchKernelGGL(k_finalize<C>, dim3(cdiv(count, 64)), dim3(64), 0, st, w_last, last, w_vp, 1u, w_bad, d_ok,
                       d_out_result, count);
    HIPCHK(mark(2 * BPP_STAGE_FINALIZE + 1, st));
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::run_serialized(bpp_verifier* v, const uint8_t* d_proofs, const uint8_t* d_commitments, size_t count,
                                  bool transcript, uint32_t* d_ok, void* d_workspace, size_t workspace_bytes,
                                  hipStream_t st, uint32_t version, const GroupedArgs* grouped) {
    const VerifyShape& s = v->s;
    if (s.n > 255 || s.m > 255) return fail(BPP_E_ARG, "the container holds n, m <= 255");
    if (version != 1 && !(version == 2 && uncompressed_bytes<C>() != 0))
        return fail(BPP_E_ARG, "container version 2 (uncompressed points) is not offered for this curve");
    const SerLayout L = ser_layout(s, count, grouped ? grouped->group : 0u);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    uint32_t* w_rec = reinterpret_cast<uint32_t*>(ws + L.records);
    uint32_t* w_sc = reinterpret_cast<uint32_t*>(ws + L.scalars);
    uint32_t* w_st = reinterpret_cast<uint32_t*>(ws + L.status);
    uint64_t* w_ch = reinterpret_cast<uint64_t*>(ws + L.challenges);
    HIPCHK(zero_words_async(w_st, count * 4, st));
    hipLaunchKernelGGL(k_container_decode<C>, dim3(cdiv(count * s.NV, 64)), dim3(64), 0, st, s, d_proofs, d_commitments,
                       w_rec, w_sc, w_st, count, version);
    if constexpr (C::ID == 0)   // cofactor > 1: membership of the prime-order subgroup
        hipLaunchKernelGGL(k_records_subgroup<C>, dim3(cdiv(count * s.NV, 64)), dim3(64), 0, st, w_rec, w_st, s.NV,
                           count * s.NV);
    HIPCHK(hipGetLastError());
    if (transcript) {
        int rc = derive_challenges(v, reinterpret_cast<const uint64_t*>(w_rec), count, w_ch, st);
        if (rc) return rc;
    }
    int rc = grouped ? run_grouped(v, reinterpret_cast<const uint64_t*>(w_rec), reinterpret_cast<const uint64_t*>(w_sc), count,
                                   transcript ? w_ch : nullptr, grouped->weight_key, grouped->index_base, grouped->d_weights,
                                   grouped->group, d_ok, grouped->h_stats, ws + L.run, workspace_bytes - L.run, st)
                     : run(v, reinterpret_cast<const uint64_t*>(w_rec), reinterpret_cast<const uint64_t*>(w_sc), count,
                           transcript ? w_ch : nullptr, d_ok, ws + L.run, workspace_bytes - L.run, nullptr, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_container_status<C>, dim3(cdiv(count, 256)), dim3(256), 0, st, w_st, d_ok, count);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::derive_challenges(bpp_verifier* v, const uint64_t* d_points, size_t count, uint64_t* d_challenges,
                             hipStream_t st) {
    hipLaunchKernelGGL(k_transcript_challenges<C>, dim3(cdiv(count, 64)), dim3(64), 0, st, v->s, v->tr0,
                       reinterpret_cast<const uint32_t*>(d_points), reinterpret_cast<uint32_t*>(d_challenges), count);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::run_combined(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                        const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                        const uint64_t* d_weights, uint32_t* d_out_partial, uint32_t* d_ok, void* d_workspace,
                        size_t workspace_bytes, hipStream_t st) {
    const VerifyShape& s = v->s;
    const CombLayout L = comb_layout(s, count);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    if (count * s.NV >= ((size_t)1 << 30)) return fail(BPP_E_ARG, "count too large");
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    uint32_t* w_pts = reinterpret_cast<uint32_t*>(ws + L.pts);
    uint32_t* w_bad = reinterpret_cast<uint32_t*>(ws + L.bad);
    uint32_t* w_sc = reinterpret_cast<uint32_t*>(ws + L.scalars);
    uint32_t* w_wt = reinterpret_cast<uint32_t*>(ws + L.weights);
    uint32_t* w_cs = reinterpret_cast<uint32_t*>(ws + L.comb_sc);
    uint32_t* w_fp = reinterpret_cast<uint32_t*>(ws + L.fpart);
    uint32_t* w_vs = reinterpret_cast<uint32_t*>(ws + L.var_sc);
    uint8_t* w_vd = ws + L.vdig;
    uint32_t* w_vt = reinterpret_cast<uint32_t*>(ws + L.vtbl);
    uint32_t* w_vscr = reinterpret_cast<uint32_t*>(ws + L.vscr);
    uint32_t* w_vw = reinterpret_cast<uint32_t*>(ws + L.vwsum);
    uint32_t* w_vf = reinterpret_cast<uint32_t*>(ws + L.vfold);
    const size_t items = count * s.NV;
    HIPCHK(zero_words_async(w_bad, count * 4, st));
    HIPCHK(zero_words_async(w_cs, (size_t)s.N * 32, st));
    hipLaunchKernelGGL(k_points_from_wire<C>, dim3(cdiv(items, 128)), dim3(128), 0, st,
                       reinterpret_cast<const uint32_t*>(d_points), w_pts, w_bad, items, s.NV, v->check_subgroup ? 1u : 0u);
    std::unique_lock<std::mutex> aux_lock;
    {
        int rc_f = fork_tables(v, st, w_pts, w_vt, w_vscr, items, aux_lock);
        if (rc_f) return rc_f;
    }
    const uint32_t* ch = d_challenges ? reinterpret_cast<const uint32_t*>(d_challenges) : v->challenges.u32();
    const uint32_t ch_stride = d_challenges ? (3 + s.k) * 8 : 0;
    {
        int rc_vs = launch_verify_scalars<C>(s, reinterpret_cast<const uint32_t*>(d_scalars), ch, ch_stride, w_sc, count,
                                             reinterpret_cast<uint32_t*>(ws + L.prep), st);
        if (rc_vs) return rc_vs;
    }
    WeightKey wk;
    for (int i = 0; i < 8; i++)
        wk.w[i] = d_weights ? 0u
                            : (uint32_t)weight_key[4 * i] | ((uint32_t)weight_key[4 * i + 1] << 8) |
                                  ((uint32_t)weight_key[4 * i + 2] << 16) | ((uint32_t)weight_key[4 * i + 3] << 24);
    hipLaunchKernelGGL(k_comb_weights<C>, dim3(cdiv(count, 256)), dim3(256), 0, st, wk, index_base,
                       reinterpret_cast<const uint32_t*>(d_weights), w_wt, count);
    hipLaunchKernelGGL(k_comb_fixed<C>, dim3(s.NF), dim3(256), 0, st, s, w_sc, w_wt, count, w_cs);
    // proof-carried points: weighted scalars -> per-proof Straus window sums -> summed across proofs per window
    hipLaunchKernelGGL(k_comb_var_scalars<C>, dim3(cdiv(items, 256)), dim3(256), 0, st, s, w_sc, w_wt, w_vs, items);
    hipLaunchKernelGGL(k_var_digits<C>, dim3(cdiv(items, 256)), dim3(256), 0, st, s, w_vs, w_vd, items, 1u);
    {
        int rc_j = join_tables(v, st, aux_lock);
        if (rc_j) return rc_j;
    }
    const size_t vlanes = count * var_wsums<C>();
    hipLaunchKernelGGL(k_var_windows<C>, dim3(cdiv(vlanes, VAR_BLOCK)), dim3(VAR_BLOCK), 0, st, s, w_vd, w_vt, w_vw,
                       vlanes, 1u, 1u);
    uint32_t* cur = w_vw;
    uint32_t* nxt = w_vf;
    for (size_t nrem = count; nrem > 1;) {
        const size_t groups = cdiv(nrem, COMB_FOLD_GROUP);
        hipLaunchKernelGGL(k_comb_window_fold<C>, dim3(cdiv(groups * var_wsums<C>(), 64)), dim3(64), 0, st, cur, nrem,
                           COMB_FOLD_GROUP, nxt, groups * var_wsums<C>(), var_wsums<C>());
        std::swap(cur, nxt);
        nrem = groups;
    }
    // the collapsed fixed-generator MulVec (one "virtual proof") with the Horner lane over the 65 sums in its
    // leading block; the Horner result lands behind the block sums
    launch_fixed_msm<C, 1>(1 + L.fixed_blocks, st, s, w_cs, v->table.u32(), w_fp, L.fixed_blocks, 1u, cur,
                           w_fp + (size_t)L.fixed_blocks * JW, (size_t)1, 1u, VpSel{1u, 0u, 1u, 0u});
    hipLaunchKernelGGL(k_comb_sum_partials<C>, dim3(1), dim3(64), 0, st, w_fp, L.fixed_blocks + 1, (uint32_t)JW, 0u, d_ok,
                       d_out_partial);
    hipLaunchKernelGGL(k_comb_verdict<C>, dim3(1), dim3(256), 0, st, d_out_partial, w_bad, count, d_ok);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::run_grouped(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                               const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                               const uint64_t* d_weights, uint32_t group, uint32_t* d_out_verdicts, uint64_t* h_stats,
                               void* d_workspace, size_t workspace_bytes, hipStream_t st) {
    int rc = grouped_begin(v, d_points, d_scalars, count, d_challenges, weight_key, index_base, d_weights, group, d_out_verdicts,
                           d_workspace, workspace_bytes, st);
    if (rc) return rc;
    return grouped_finish(v, d_points, d_scalars, count, d_challenges, group, d_out_verdicts, h_stats, d_workspace,
                          workspace_bytes, st);
}

// pass 1 (only enqueues): one weighted check per group, through the batch verifier's own last stages at count = G
template <class C>
int VerifyImpl<C>::grouped_begin(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                                 const uint64_t* d_challenges, const uint8_t* weight_key, uint64_t index_base,
                                 const uint64_t* d_weights, uint32_t group, uint32_t* d_out_verdicts, void* d_workspace,
                                 size_t workspace_bytes, hipStream_t st) {
    const VerifyShape& s = v->s;
    if (group < 2 || (group & (group - 1))) return fail(BPP_E_ARG, "group must be a power of two, at least 2");
    const GroupLayout L = group_layout(s, count, group);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    if (count * s.NV >= ((size_t)1 << 30)) return fail(BPP_E_ARG, "count too large");
    if (count == 0) return BPP_OK;
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    auto W = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    uint32_t *w_pts = W(L.pts), *w_bad = W(L.bad), *w_sc = W(L.scalars), *w_wt = W(L.weights), *w_vs = W(L.var_sc);
    uint32_t *w_vt = W(L.vtbl), *w_vw = W(L.vwsum), *w_vf = W(L.vfold), *w_rows = W(L.grows), *w_gbad = W(L.gbad);
    uint32_t* w_gok = W(L.gok);
    uint8_t* w_vd = ws + L.vdig;
    const size_t items = count * s.NV, G = L.groups;
    HIPCHK(zero_words_async(w_bad, count * 4, st));
    hipLaunchKernelGGL(k_points_from_wire<C>, dim3(cdiv(items, 128)), dim3(128), 0, st,
                       reinterpret_cast<const uint32_t*>(d_points), w_pts, w_bad, items, s.NV, v->check_subgroup ? 1u : 0u);
    std::unique_lock<std::mutex> aux_lock;
    {
        int rc_f = fork_tables(v, st, w_pts, w_vt, W(L.vscr), items, aux_lock);
        if (rc_f) return rc_f;
    }
    const uint32_t* ch = d_challenges ? reinterpret_cast<const uint32_t*>(d_challenges) : v->challenges.u32();
    const uint32_t ch_stride = d_challenges ? (3 + s.k) * 8 : 0;
    {
        int rc_vs = launch_verify_scalars<C>(s, reinterpret_cast<const uint32_t*>(d_scalars), ch, ch_stride, w_sc, count,
                                             W(L.prep), st);
        if (rc_vs) return rc_vs;
    }
    WeightKey wk;
    for (int i = 0; i < 8; i++)
        wk.w[i] = d_weights ? 0u
                            : (uint32_t)weight_key[4 * i] | ((uint32_t)weight_key[4 * i + 1] << 8) |
                                  ((uint32_t)weight_key[4 * i + 2] << 16) | ((uint32_t)weight_key[4 * i + 3] << 24);
    hipLaunchKernelGGL(k_comb_weights<C>, dim3(cdiv(count, 256)), dim3(256), 0, st, wk, index_base,
                       reinterpret_cast<const uint32_t*>(d_weights), w_wt, count);
    hipLaunchKernelGGL(k_comb_fixed_grouped<C>, dim3((unsigned)(G * cdiv(s.NF, 64))), dim3(64), 0, st, s, w_sc, w_wt, count,
                       group, w_rows);
    hipLaunchKernelGGL(k_comb_group_bad, dim3(cdiv(G, 256)), dim3(256), 0, st, w_bad, count, group, w_gbad, G);
    const uint32_t tree = horner_form(s, G);
    const uint32_t per = tree == 1 ? var_wsums<C>() : var_windows<C>();   // window sums per proof, in the layout the Horner form reads
    hipLaunchKernelGGL(k_comb_var_scalars<C>, dim3(cdiv(items, 256)), dim3(256), 0, st, s, w_sc, w_wt, w_vs, items);
    hipLaunchKernelGGL(k_var_digits<C>, dim3(cdiv(items, 256)), dim3(256), 0, st, s, w_vs, w_vd, items, 1u);
    {
        int rc_j = join_tables(v, st, aux_lock);
        if (rc_j) return rc_j;
    }
    const size_t vlanes = count * per;
    hipLaunchKernelGGL(k_var_windows<C>, dim3(cdiv(vlanes, VAR_BLOCK)), dim3(VAR_BLOCK), 0, st, s, w_vd, w_vt, w_vw,
                       vlanes, tree == 1 ? 1u : 0u, 1u);
    uint32_t* cur = w_vw;
    uint32_t* nxt = w_vf;
    size_t nrem = count;
    for (uint32_t left = group; left > 1;) {   // the proofs of a group are neighbours: 4 (at last 2) to 1 per level
        const uint32_t step = left >= 4 ? 4u : 2u;
        const size_t outn = cdiv(nrem, step);
        hipLaunchKernelGGL(k_comb_window_fold<C>, dim3(cdiv(outn * per, 64)), dim3(64), 0, st, cur, nrem, step, nxt, outn * per,
                           per);
        std::swap(cur, nxt);
        nrem = outn;
        left /= step;
    }
    {
        const WsLayout T = ws_layout(s, G);
        int rc = finish(v, ws + L.tail, T, G, w_rows, cur, w_gbad, w_gok, nullptr, tree, false, st, nullptr);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_comb_group_spread, dim3(cdiv(count, 256)), dim3(256), 0, st, w_gok, group, d_out_verdicts, count);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

// the groups' verdicts come back to the host (synchronises `st`); pass 2: the proofs of the groups that failed, exactly,
// through the per-proof path.  Same buffers and stream as the grouped_begin it completes.
template <class C>
int VerifyImpl<C>::grouped_finish(bpp_verifier* v, const uint64_t* d_points, const uint64_t* d_scalars, size_t count,
                                  const uint64_t* d_challenges, uint32_t group, uint32_t* d_out_verdicts, uint64_t* h_stats,
                                  void* d_workspace, size_t workspace_bytes, hipStream_t st) {
    const VerifyShape& s = v->s;
    if (group < 2 || (group & (group - 1))) return fail(BPP_E_ARG, "group must be a power of two, at least 2");
    const GroupLayout L = group_layout(s, count, group);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    if (h_stats) h_stats[0] = h_stats[1] = 0;
    if (count == 0) return BPP_OK;
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    auto W = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    const size_t G = L.groups;
    std::vector<uint32_t> gok(G);
    HIPCHK(hipMemcpyAsync(gok.data(), W(L.gok), G * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> list;
    size_t failed = 0;
    for (size_t g = 0; g < G; g++)
        if (gok[g]) {
            failed++;
            for (size_t p = g * group; p < std::min(count, (g + 1) * (size_t)group); p++) list.push_back((uint32_t)p);
        }
    if (h_stats) {
        h_stats[0] = failed;
        h_stats[1] = list.size();
    }
    const uint32_t row_pts = s.NV * WW, row_sc = 24, row_ch = (3 + s.k) * 8;
    for (size_t lo = 0; lo < list.size(); lo += L.slice) {
        const size_t cnt = std::min(L.slice, list.size() - lo);
        HIPCHK(hipMemcpyAsync(W(L.list), list.data() + lo, cnt * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_comb_gather_rows, dim3((unsigned)cnt), dim3(256), 0, st,
                           reinterpret_cast<const uint32_t*>(d_points), W(L.list), row_pts, W(L.x_pts));
        hipLaunchKernelGGL(k_comb_gather_rows, dim3((unsigned)cnt), dim3(64), 0, st,
                           reinterpret_cast<const uint32_t*>(d_scalars), W(L.list), row_sc, W(L.x_sc));
        if (d_challenges)
            hipLaunchKernelGGL(k_comb_gather_rows, dim3((unsigned)cnt), dim3(64), 0, st,
                               reinterpret_cast<const uint32_t*>(d_challenges), W(L.list), row_ch, W(L.x_ch));
        int rc = run(v, reinterpret_cast<const uint64_t*>(ws + L.x_pts), reinterpret_cast<const uint64_t*>(ws + L.x_sc), cnt,
                     d_challenges ? reinterpret_cast<const uint64_t*>(ws + L.x_ch) : nullptr, W(L.x_ok), ws + L.x_run,
                     workspace_bytes - L.x_run, nullptr, nullptr, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_comb_scatter_words, dim3(cdiv(cnt, 256)), dim3(256), 0, st, W(L.x_ok), W(L.list), d_out_verdicts,
                           cnt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));   // `list` slices are staged from pageable memory
    }
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::prove_batch_device(bpp_verifier* v, const uint64_t* d_values, const uint64_t* d_gammas, size_t count,
                              uint64_t* d_out_points, uint64_t* d_out_scalars, uint64_t* d_out_V, bool fs,
                              uint64_t* d_out_challenges, void* d_workspace, size_t workspace_bytes, hipStream_t st,
                              const uint8_t* blind_key, uint64_t index_base, const uint64_t* d_blinding) {
    const VerifyShape& s = v->s;
    const uint32_t k = s.k, m = s.m;
    const uint32_t nvp = pb_num_vps(k, m);
    BlindKey bk;
    for (int i = 0; i < 8; i++)
        bk.w[i] = blind_key ? (uint32_t)blind_key[4 * i] | ((uint32_t)blind_key[4 * i + 1] << 8) |
                                  ((uint32_t)blind_key[4 * i + 2] << 16) | ((uint32_t)blind_key[4 * i + 3] << 24)
                            : 0u;
    const ProveLayout L = prove_layout(s, count);
    if (workspace_bytes < L.total) return fail(BPP_E_ARG, "workspace too small");
    ProverConsts pc;
    pc.alpha = m == 1 ? 7 : 33;   // range/mod.rs:94 / :256
    pc.d_L = 4;                   // wip.rs:94
    pc.d_R = 5;                   // wip.rs:95
    pc.r = 33;                    // wip.rs:175
    pc.s = 44;
    pc.delta = 88;
    pc.eta = 123;
    uint8_t* ws = static_cast<uint8_t*>(d_workspace);
    auto W = [&](size_t off) { return reinterpret_cast<uint32_t*>(ws + off); };
    for (size_t base = 0; base < count; base += L.chunk) {
        const size_t cnt = std::min(L.chunk, count - base);
        uint32_t* o_pts = reinterpret_cast<uint32_t*>(d_out_points) + base * (size_t)(3 + 2 * k) * WW;
        uint32_t* o_sc = reinterpret_cast<uint32_t*>(d_out_scalars) + base * 24;
        uint32_t* o_V = d_out_V ? reinterpret_cast<uint32_t*>(d_out_V) + base * (size_t)m * WW : W(L.vout);
        const uint64_t* vals = d_values + base * m;
        const uint32_t* gams = reinterpret_cast<const uint32_t*>(d_gammas) + base * (size_t)m * 8;
        const uint32_t* blind = nullptr;   // this chunk's blinding scalars
        if (d_blinding) {
            blind = reinterpret_cast<const uint32_t*>(d_blinding) + base * (size_t)pb_blind_elems(k) * 8;
        } else if (blind_key) {
            hipLaunchKernelGGL(k_pb_blind<C>, dim3(cdiv(cnt * pb_blind_elems(k), 64)), dim3(64), 0, st, bk, index_base + base, k,
                               W(L.blind), cnt);
            blind = W(L.blind);
        }
        // one MulVec launch over `sel` of every proof's virtual proofs, then their wire points into the records
        auto msm = [&](VpSel sel) {
            const size_t nv = cnt * sel.cnt;
            // never more blocks per virtual proof than the workspace was sized for
            const unsigned per = std::min(L.per, blocks_per_proof(s, nv));
            launch_fixed_msm<C, 2>((unsigned)(nv * per), st, s, W(L.vps), v->table.u32(), W(L.part), per, 0u,
                                   (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)0, 0u, sel);
            // per-thread partials -> 16 -> 4 per block with every lane busy (as the verifier does); k_pb_collect adds
            // the 4 * per that are left of each MulVec
            const size_t f1 = nv * per * (FIXED_BLOCK / FOLD_GROUP), f2 = f1 / FOLD_GROUP2;
            hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(f1, 64)), dim3(64), 0, st, W(L.part), FOLD_GROUP, W(L.part1), f1);
            hipLaunchKernelGGL(k_partials_fold<C>, dim3(cdiv(f2, 64)), dim3(64), 0, st, W(L.part1), FOLD_GROUP2, W(L.part2), f2);
            hipLaunchKernelGGL(k_pb_collect<C>, dim3(cdiv(nv, 64)), dim3(64), 0, st, s, sel, W(L.part2),
                               per * (FIXED_BLOCK / FOLD_GROUP / FOLD_GROUP2), o_pts, o_V, nv);
        };
        if (!fs) {
            hipLaunchKernelGGL(k_pb_init<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_ALL, 0u, vals, gams,
                               v->challenges.u32(), 0u, W(L.a), W(L.b), W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
            for (uint32_t t = 0; t < k; t++)
                hipLaunchKernelGGL(k_pb_round<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, t, (uint32_t)PB_ALL, W(L.a),
                                   W(L.b), W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
            hipLaunchKernelGGL(k_pb_final<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_ALL, W(L.a), W(L.b),
                               W(L.cG), W(L.cH), W(L.con), W(L.vps), o_sc);
            msm(VpSel{nvp, 0u, nvp, 1u});
            continue;
        }
        uint32_t* o_ch = d_out_challenges ? reinterpret_cast<uint32_t*>(d_out_challenges) + base * (size_t)(3 + k) * 8
                                          : W(L.ch);
        const uint32_t chs = (3 + k) * 8;
        const unsigned lanes = cdiv(cnt, 64);
        hipLaunchKernelGGL(k_pb_init<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_PRE, 1u, vals, gams, o_ch,
                           chs, W(L.a), W(L.b), W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
        msm(VpSel{nvp, 0u, 1u, 1u});              // A
        msm(VpSel{nvp, 2 * k + 3, m, 1u});        // V_0 .. V_{m-1}
        hipLaunchKernelGGL(k_pb_fs_yz<C>, dim3(lanes), dim3(64), 0, st, s, v->tr0, o_pts, o_V, W(L.trst), o_ch, cnt);
        hipLaunchKernelGGL(k_pb_init<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_POST, 1u, vals, gams, o_ch,
                           chs, W(L.a), W(L.b), W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
        for (uint32_t t = 0; t < k; t++) {
            hipLaunchKernelGGL(k_pb_round<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, t, (uint32_t)PB_PRE, W(L.a), W(L.b),
                               W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
            msm(VpSel{nvp, 1 + 2 * t, 2u, 1u});   // L_t, R_t
            hipLaunchKernelGGL(k_pb_fs_round<C>, dim3(lanes), dim3(64), 0, st, s, t, o_pts, W(L.trst), o_ch, W(L.con), cnt);
            hipLaunchKernelGGL(k_pb_round<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, t, (uint32_t)PB_POST, W(L.a),
                               W(L.b), W(L.cG), W(L.cH), W(L.pwy), W(L.con), W(L.vps));
        }
        hipLaunchKernelGGL(k_pb_final<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_PRE, W(L.a), W(L.b),
                           W(L.cG), W(L.cH), W(L.con), W(L.vps), o_sc);
        msm(VpSel{nvp, 2 * k + 1, 2u, 1u});       // wip.A, wip.B
        hipLaunchKernelGGL(k_pb_fs_final<C>, dim3(lanes), dim3(64), 0, st, s, o_pts, W(L.trst), o_ch, W(L.con), cnt);
        hipLaunchKernelGGL(k_pb_final<C>, dim3((unsigned)cnt), dim3(256), 0, st, s, pc, blind, (uint32_t)PB_POST, W(L.a), W(L.b),
                           W(L.cG), W(L.cH), W(L.con), W(L.vps), o_sc);
    }
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::prove_batch(bpp_verifier* v, const uint64_t* values, const uint64_t* gammas, size_t count,
                       uint64_t* out_points, uint64_t* out_scalars, uint64_t* out_V, bool fs, const uint8_t* blind_key,
                       uint64_t index_base) {
    const VerifyShape& s = v->s;
    const uint32_t k = s.k, m = s.m;
    hipStream_t st = nullptr;
    const ProveLayout L = prove_layout(s, count);
    DevBuf d_val, d_gam, d_pts, d_V, d_sc, d_ws;
    HIPCHK(d_val.alloc(count * m * 8));
    HIPCHK(hipMemcpyAsync(d_val.p, values, count * m * 8, hipMemcpyHostToDevice, st));
    int rc = upload_scalars<C>(gammas, count * m, d_gam, st);
    if (rc) return rc;
    HIPCHK(d_pts.alloc(count * (size_t)(3 + 2 * k) * WW * 4));
    HIPCHK(d_V.alloc(count * (size_t)m * WW * 4));
    HIPCHK(d_sc.alloc(count * 3 * 32));
    HIPCHK(d_ws.alloc(L.total));
    rc = prove_batch_device(v, static_cast<const uint64_t*>(d_val.p), static_cast<const uint64_t*>(d_gam.p), count,
                            static_cast<uint64_t*>(d_pts.p), static_cast<uint64_t*>(d_sc.p),
                            static_cast<uint64_t*>(d_V.p), fs, nullptr, d_ws.p, L.total, st, blind_key, index_base, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out_points, d_pts.p, count * (size_t)(3 + 2 * k) * WW * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(out_scalars, d_sc.p, count * 96, hipMemcpyDeviceToHost, st));
    if (out_V) HIPCHK(hipMemcpyAsync(out_V, d_V.p, count * (size_t)m * WW * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return BPP_OK;
}

template <class C>
int VerifyImpl<C>::sum_partials(const uint32_t* d_partials, size_t n, uint32_t* d_ok, hipStream_t st) {
    hipLaunchKernelGGL(k_comb_sum_partials<C>, dim3(1), dim3(64), 0, st, d_partials, (uint32_t)n,
                       (uint32_t)partial_words<C>(), 1u, d_ok, (uint32_t*)nullptr);
    HIPCHK(hipGetLastError());
    return BPP_OK;
}

#endif  // BPP_IMPL_DEFINITIONS


extern template struct VerifyImpl<Bls12381>;
extern template struct VerifyImpl<Secp256k1>;
extern template struct VerifyImpl<Ed25519>;

}  // namespace bpp
