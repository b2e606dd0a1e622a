/*
 * bpp_amd.h -- C ABI of the MI355X-native Bulletproofs+ engine (libbpp_amd.so).
 *
 * This is the drop-in boundary for the reference crate's hot path (gogoex/BulletProofsPlus): every
 * entry point below names the reference interface it replaces (paths relative to /root/reference).
 * The reference has no FFI of its own; its only ABI crossing is Rust -> libmcl inside mcl_rust.
 * Here the crossing moves up to the MulVec / RangeProof level: host -> extern "C" -> HIP.
 * INTEGRATION.md shows the Rust `extern "C"` block and the replacement bodies of
 * `MulVec::calculate`, `RangeProof::{prove,verify}`, `PublicKey::new`, `RangeProver::commit`.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every buffer; nothing borrowed outlives a call.
 *   - return value: 0 = Ok(()), 1 = Err(ProofError::VerificationError)
 *     (reference src/errors.rs:14-50; the only variant the path constructs, range/mod.rs:508,
 *     weighted_inner_product_proof.rs:326,336), negative = usage / runtime error (BPP_E_*), where the
 *     reference would panic (mulvec.rs:23-25, range/mod.rs:90-91,252-253, wip.rs:60-67).
 *     Nothing unwinds, nothing is printed.
 *   - scalar: 4 x uint64_t little-endian limbs, canonical (non-Montgomery) value; values >= r are
 *     reduced mod r on entry.
 *   - point : (2*L + 1) x uint64_t = affine x (L limbs LE) | y (L limbs LE) | infinity flag (0/1),
 *     canonical coordinates; L = 6 for BLS12-381 G1, 4 for secp256k1 (bpp_point_words()).
 *   - there is no CPU fallback: every call runs HIP kernels on the context's device and fails with
 *     BPP_E_HIP if the device is unusable.
 *   - a context is bound to one device and one thread at a time; contexts are independent.
 */
#ifndef BPP_AMD_H
#define BPP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* curve ids.  BLS12-381 G1 is what the reference's range proof is wired to (src/range/mod.rs:10-15);
 * secp256k1 is its second, in-tree backend (src/secp256k1/building_block/). */
#define BPP_BLS12_381_G1 0
#define BPP_SECP256K1 1
/* edwards25519 (the curve under Ristretto255), prime-order subgroup, points as affine Edwards (x, y), L = 4.
 * The reference has NO such backend (only a stale README example): parity unpinned, see csrc/ed25519.hpp. */
#define BPP_ED25519 2

#define BPP_OK 0
#define BPP_VERIFICATION_ERROR 1
#define BPP_FORMAT_ERROR 2  /* ProofError::FormatError (src/errors.rs:20): per-proof status of the serialized-proof path */
#define BPP_E_ARG (-1)      /* bad argument (null pointer, unknown curve, n*m not a power of two...) */
#define BPP_E_HIP (-2)      /* HIP runtime error; bpp_last_error() has the text */
#define BPP_E_LENGTH (-3)   /* "mulvec: lengths of scalars and points must match" and friends */
#define BPP_E_POINT (-4)    /* a point is not on the curve / coordinate >= p */
#define BPP_E_NOMEM (-5)

typedef struct bpp_ctx bpp_ctx;
typedef struct bpp_verifier bpp_verifier;

/* Arith::init (src/bls12_381/building_block/arith.rs:6-19): one-time library/device initialisation.
 * `device` is the HIP device ordinal. */
int bpp_init(int curve_id, int device, bpp_ctx **out_ctx);
void bpp_destroy(bpp_ctx *ctx);
const char *bpp_last_error(void);

/* words (uint64_t) per wire point for a curve: 2*L + 1 */
int bpp_point_words(int curve_id);

/* MulVec::calculate (src/bls12_381/building_block/mulvec.rs:20-33; secp256k1 twin
 * src/secp256k1/building_block/secp256k1/util.rs:22-36): out = sum_i scalars[i] * points[i].
 * n == 0 gives the point at infinity (Point::zero()).  Host pointers. */
int bpp_msm(bpp_ctx *ctx, const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t *out);

/* The same MulVec through the bucket (Pippenger) pipeline regardless of n (bpp_msm switches to it by
 * itself from n = 4096 up); window_bits in [2, 16], 0 = chosen from n.  Host pointers. */
int bpp_msm_pippenger(bpp_ctx *ctx, const uint64_t *scalars, const uint64_t *points, size_t n, int window_bits,
                      uint64_t *out);

/* MulVec::calculate (mulvec.rs:20-33) with EVERY buffer in HBM, asynchronous on `stream` (a hipStream_t; NULL = default
 * stream): nothing is copied, nothing synchronises -- the seam for callers whose scalars and points already live on
 * the device, and what bench.py's `msm` leg times.  The bucket (Pippenger) pipeline of csrc/pippenger.hpp at any n:
 * signed c-bit windows with an unsigned top window, on BLS12-381 after the GLV split of every scalar (two 128-bit halves
 * per point), counting sort per window, one lane per bucket with an LDS-DMA gather ring, bucket sums reduced per tile
 * of 64 S buckets by one wave (running sums per lane, suffix scan and butterfly across the wave by shuffles).
 *   d_scalars : n scalars (4 x u64, canonical; values >= r are reduced on the device)
 *   d_points  : n wire points
 *   d_out     : one wire point (affine, canonical) -- the sum, bit for bit what bpp_msm returns
 *   d_status  : one uint32_t (may be NULL): 0, or 1 when some point was not on the curve / had a coordinate >= p (the
 *               host-pointer calls report BPP_E_POINT; here the point counts as infinity and the flag is raised)
 *   d_workspace: bpp_msm_workspace_bytes(ctx, n, window_bits) bytes
 * window_bits in [2, 16], 0 = chosen from n.  n == 0 gives Point::zero(). */
size_t bpp_msm_workspace_bytes(bpp_ctx *ctx, size_t n, int window_bits);
int bpp_msm_device(bpp_ctx *ctx, const uint64_t *d_scalars, const uint64_t *d_points, size_t n, int window_bits,
                   uint64_t *d_out, uint32_t *d_status, void *d_workspace, size_t workspace_bytes, void *stream);

/* Per-stage timing of bpp_msm_device with HIP events recorded on the caller's stream (stages: 0 sort = points from
 * wire, digits + histogram, scans, scatter; 1 bucket sums, chunk by chunk [dominant: k_pip_chunks]; 2 fold of the
 * segments into buckets; 3 bucket reduction per tile and per window; 4 doublings + final sum + affine point).
 * bpp_msm_profile averages over the calls recorded since profiling was switched on (at most 16): out_stage_ms[5];
 * out_shape (may be NULL) receives the geometry of the last call: n, items, windows, narrow window bits, wide
 * windows, buckets, entries per chunk, requested window bits. */
int bpp_msm_set_profiling(bpp_ctx *ctx, int on);
int bpp_msm_profile(bpp_ctx *ctx, float *out_stage_ms, size_t *out_passes, uint32_t *out_shape);

/* `count` independent MulVecs in one launch: MulVec c has lens[c] terms starting at offset
 * sum(lens[0..c)).  out: count points.  (The fold of src/weighted_inner_product_proof.rs:151-163 is
 * 2 n' MulVecs of length 2.)  Host pointers. */
int bpp_msm_batch(bpp_ctx *ctx, const uint64_t *scalars, const uint64_t *points, const uint32_t *lens,
                  size_t count, uint64_t *out);

/* Point * PrimeFieldElem for n independent pairs (src/bls12_381/building_block/point/point.rs:69-85).
 * Host pointers. */
int bpp_scalar_mul_batch(bpp_ctx *ctx, const uint64_t *scalars, const uint64_t *points, size_t n,
                         uint64_t *out);

/* PublicKey::new(length) (src/publickey.rs:21-48): g = base point, h = 2g, G_i = 3(i+1) g,
 * H_i = 5(i+1) g.  out_gh: 2 points [g, h]; out_G, out_H: `length` points each. */
int bpp_pk_new(bpp_ctx *ctx, size_t length, uint64_t *out_gh, uint64_t *out_G, uint64_t *out_H);

/* The production counterpart of PublicKey::new: g = the base point, h, G_i, H_i hashed to the group from `label`
 * (csrc/hash_to_group.hpp), so that no discrete-log relation between the generators is known.  The reference only has
 * the test generators above (its own comment at src/publickey.rs:23-39).  Not a standard hash-to-curve suite: try-and-
 * increment (+ cofactor clearing on BLS12-381 G1) for the Weierstrass curves, RFC 9496 element derivation for
 * ristretto255.  PARITY UNPINNED; restated in oracle/pyref.py. */
int bpp_pk_hashed(bpp_ctx *ctx, const uint8_t *label, size_t label_len, size_t length, uint64_t *out_gh,
                  uint64_t *out_G, uint64_t *out_H);

/* RangeProver::commit / PublicKey::commitment (src/range/prover.rs:28-42, src/publickey.rs:50-52):
 * out = g * new(v as i32) + h * gamma.  The `v as i32` truncation of prover.rs:37 is kept. */
int bpp_commit(bpp_ctx *ctx, const uint64_t *gh, uint64_t v, const uint64_t *gamma, uint64_t *out);

/* RangeProof::prove (src/range/mod.rs:31-55 -> prove_single :80-187 / prove_multiple :240-403, and
 * WeightedInnerProductProof::prove, src/weighted_inner_product_proof.rs:36-227).
 * pk = (gh, G, H) with n*m generators each; v[m], gamma[m] (scalars), V[m] commitments.
 * out_points : 3 + 2k points  [A, wip.A, wip.B, L_0..L_{k-1}, R_0..R_{k-1}],  k = log2(n*m)
 * out_scalars: 3 scalars      [r', s', delta'] */
int bpp_range_prove(bpp_ctx *ctx, const uint64_t *gh, const uint64_t *G, const uint64_t *H, size_t n,
                    size_t m, const uint64_t *v, const uint64_t *gamma, const uint64_t *V,
                    uint64_t *out_points, uint64_t *out_scalars);

/* One folding round of WeightedInnerProductProof::prove as a seam of its own (src/weighted_inner_product_proof.rs:147-164),
 * in place on the first n' = len / 2 entries of each array (host pointers; a, b: len scalars; G, H: len points):
 *   a[i] = a[i] e + a[n'+i] y^n' e^-1 ;  b[i] = b[i] e^-1 + b[n'+i] e ;
 *   G[i] = MulVec[e^-1, y^-n' e].[G[i], G[n'+i]] ;  H[i] = MulVec[e, e^-1].[H[i], H[n'+i]]
 * y_nhat = y^n' (wip.rs:98), e = the round's challenge (wip.rs:131).  bpp_range_prove runs this per round on resident
 * vectors; the batched prover (bpp_range_prove_batch) never folds points at all (csrc/prover_batch.hpp). */
int bpp_wip_fold_round(bpp_ctx *ctx, uint64_t *a, uint64_t *b, uint64_t *G, uint64_t *H, size_t len,
                       const uint64_t *y_nhat, const uint64_t *e);

/* RangeProof::verify (src/range/mod.rs:57-78 -> verify_single :189-238 + wip verify
 * src/weighted_inner_product_proof.rs:238-328, or verify_multiple :405-510).
 * proof_points as written by bpp_range_prove with k rounds.  Returns 0 / 1 / negative.
 * The reference takes the public key with every call and pays the whole naive MulVec each time; so does the FIRST call
 * here with a given key (data-parallel naive MulVec, no setup).  When the same key comes back the context builds a
 * verifier with narrow window tables for it (c = 8: milliseconds to build, < 1 GB at n m = 1024) and that and later
 * calls run the batch verifier's pass at count = 1.  A cached entry is found by hash and confirmed by comparing the
 * key bytes; at most four keys, least recently used out; bpp_set_verify_cache(ctx, 0) switches the mechanism off and
 * frees it.  Verdicts do not depend on which path ran (tests/test_gpu_round3.py). */
int bpp_set_verify_cache(bpp_ctx *ctx, int on);
int bpp_range_verify(bpp_ctx *ctx, const uint64_t *gh, const uint64_t *G, const uint64_t *H, size_t n,
                     size_t m, const uint64_t *proof_points, size_t k, const uint64_t *proof_scalars,
                     const uint64_t *V);

/* ---- batch verifier: the north-star path -----------------------------------------------------
 * A verifier holds the public key in HBM together with the fixed-base window tables built from it
 * (see DESIGN.md), for one (n, m).  Proof batches are DEVICE buffers (e.g. torch CUDA tensors'
 * data_ptr()), so a step of the hot path touches no host memory:
 *   d_points  : count x (3 + 2k + m) wire points   [A, wip.A, wip.B, L_0.., R_0.., V_0..V_{m-1}]
 *   d_scalars : count x 3 scalars                  [r', s', delta']
 *   d_ok      : count x uint32_t                   0 = Ok(()), 1 = Err(VerificationError)
 * The reference has no batch API (src/lib.rs:11-13); each entry of d_ok is exactly the verdict
 * RangeProof::verify would return for that proof.
 * window_bits in [2, 20] trades HBM for arithmetic: a b-bit scalar costs floor((b-1)/c) + 1 table additions per
 * generator and the tables hold about (2mn + 2) x b/c x 2^(c-1) affine points (n=64, m=16 on BLS12-381:
 * 12.7 GB at c = 13, 103 GB at c = 16, 204 GB at c = 17).  BPP_E_NOMEM (-5) if they do not fit. */
int bpp_verifier_create(bpp_ctx *ctx, const uint64_t *gh, const uint64_t *G, const uint64_t *H, size_t n,
                        size_t m, int window_bits, bpp_verifier **out);
void bpp_verifier_destroy(bpp_verifier *v);
/* bytes of device workspace bpp_verifier_run needs for `count` proofs */
size_t bpp_verifier_workspace_bytes(const bpp_verifier *v, size_t count);
/* number of MulVec terms per proof, N = 2mn + 2k + m + 5 */
size_t bpp_verifier_msm_len(const bpp_verifier *v);
/* bytes of HBM held by the window tables */
size_t bpp_verifier_table_bytes(const bpp_verifier *v);

/* One pass of the hot path over a resident batch, asynchronous on `stream` (a hipStream_t; NULL =
 * default stream).  d_challenges is NULL (the reference's hard-coded "transcript") or
 * count x (3 + k) scalars [y, z, e, e_1..e_k] per proof.
 * Optional debug / parity outputs (NULL to skip):
 *   d_out_scalars: count x N scalars -- the MulVec scalars in the reference's MulVec order
 *                  (range/mod.rs:481-490 for m > 1, wip.rs:298-307 for m == 1)
 *   d_out_result : count x wire point -- the MulVec result ("expected", range/mod.rs:503)
 * The call only enqueues work on `stream` (kernels, and for a batch small enough to be latency bound an event fork/join
 * with a side stream of the verifier): after one eager call it can be captured into a HIP graph and replayed
 * (tests/test_gpu_round2.py::test_verifier_run_is_graph_capturable).
 * A verifier may be used by several host threads and on several streams at once, each pass with a workspace and a verdict
 * buffer of its own (the tables are read-only); the stage profiling is the exception: one thread while it is on.
 * Points are elements of the prime-order group (what the prover, mcl, or the decoder with its subgroup check produce).
 * On BLS12-381 the proof-carried points are multiplied through G1's endomorphism (GLV, as mcl itself does): for points
 * of G1 the result is sum s_i P_i bit for bit; a curve point OUTSIDE G1 is still processed deterministically and its
 * proof judged by the same equation, but the result point is then not the full-curve sum. */
int bpp_verifier_run(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                     const uint64_t *d_challenges, uint32_t *d_ok, void *d_workspace,
                     size_t workspace_bytes, uint64_t *d_out_scalars, uint64_t *d_out_result,
                     void *stream);
/* The same pass captured ONCE into a HIP graph and replayed: a small batch is a chain of a dozen launches and a stream
 * fork/join that a replay submits in one call.  bpp_verifier_graph_capture runs the pass once eagerly (argument checks;
 * what a pass creates lazily must exist before a capture), captures it on a stream of its own and instantiates the graph;
 * the device pointers and `count` are baked in: the caller refreshes the CONTENTS of d_points / d_scalars / d_challenges
 * between replays and reads d_ok after them.  bpp_graph_launch only enqueues (hipGraphLaunch on `stream`).  The verifier must
 * outlive its graphs; stage profiling must be off while capturing. */
typedef struct bpp_graph bpp_graph;
int bpp_verifier_graph_capture(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                               const uint64_t *d_challenges, uint32_t *d_ok, void *d_workspace, size_t workspace_bytes,
                               bpp_graph **out);
int bpp_graph_launch(bpp_graph *g, void *stream);
void bpp_graph_destroy(bpp_graph *g);

/* RangeProof::prove for `count` independent provers that share (pk, n, m) -- and RangeProver::commit for
 * their values -- in one device-resident pass (csrc/prover_batch.hpp): the folding rounds of
 * src/weighted_inner_product_proof.rs:79-172 fold only scalars, every L, R, A, B is a MulVec over the
 * original generators through the engine's window tables.  Output is bit-identical to bpp_range_prove.
 *   v: count x m uint64_t ; gamma: count x m scalars
 *   out_points : count x (3 + 2k) points [A, wip.A, wip.B, L.., R..] ; out_scalars: count x 3 scalars
 *   out_V      : count x m commitments (may be NULL).  Host pointers. */
int bpp_range_prove_batch(bpp_verifier *engine, const uint64_t *v, const uint64_t *gamma, size_t count,
                          uint64_t *out_points, uint64_t *out_scalars, uint64_t *out_V);

/* The same pass with every buffer in HBM (what bench.py's `prove` leg times): d_v count x m uint64_t, d_gamma
 * count x m scalars, outputs as above (d_out_V may be NULL), asynchronous on `stream`; no host memory is touched
 * and nothing synchronises.  The batch is processed in chunks that reuse d_workspace
 * (bpp_prover_workspace_bytes(engine, count) bytes). */
size_t bpp_prover_workspace_bytes(const bpp_verifier *engine, size_t count);
int bpp_range_prove_batch_device(bpp_verifier *engine, const uint64_t *d_v, const uint64_t *d_gamma, size_t count,
                                 uint64_t *d_out_points, uint64_t *d_out_scalars, uint64_t *d_out_V,
                                 void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- combined batch check ("final multiscalar check") -- an engine mode, NOT a reference code path ----
 * One random linear combination of the batch's verification MulVecs, sum_p w_p * M_p == identity, with 128-bit
 * weights w_p (csrc/combined.hpp): the fixed generators collapse to one fixed-base MulVec, the proof-carried
 * points run through the verifier's Straus kernels and are summed window by window.  An all-valid batch always
 * passes.  A batch holding an invalid proof fails except with probability ~2^-128 -- PROVIDED the weights could not
 * be predicted by whoever made the proofs; with a fixed or guessable key two invalid proofs can be made to cancel.
 * So the weights come from the caller, one of
 *   d_weights  : count x 16 bytes on the device (little-endian 128-bit values), e.g. from a transcript over the
 *                whole batch; or
 *   weight_key : 32 secret bytes (host pointer) from the OS CSPRNG, fresh per call or per verifier; the device
 *                expands w_p = SHA-256(key || "bppw" || (index_base + p) as u64)[0..16).  index_base is the GLOBAL index
 *                of this call's first proof, so ranks that share a key never share a weight.
 * The caller falls back to bpp_verifier_run for the exact per-proof verdicts of the reference when the check fails.
 * It assumes proof points of the prime-order subgroup (bpp_proofs_decode checks that for serialized proofs).
 *   d_out_partial: bpp_verifier_partial_bytes() bytes -- this call's weighted sum (opaque jacobian image) followed by
 *                  a validity word (non-zero when a proof of this call carried an invalid point); ranks exchange
 *                  these once (RCCL all-gather) and bpp_verifier_sum_partials adds the sums and ORs the words
 *   d_ok         : one uint32_t, 0 iff the partial is the identity and every proof point was valid */
size_t bpp_verifier_partial_bytes(const bpp_verifier *v);
size_t bpp_verifier_combined_workspace_bytes(const bpp_verifier *v, size_t count);
int bpp_verifier_run_combined(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                              const uint64_t *d_challenges, const uint8_t *weight_key, uint64_t index_base,
                              const uint64_t *d_weights, void *d_out_partial, uint32_t *d_ok, void *d_workspace,
                              size_t workspace_bytes, void *stream);
/* d_partials: n partials of bpp_verifier_partial_bytes() bytes each; d_ok = 0 iff their sum is the identity and no
 * rank reported an invalid point */
int bpp_verifier_sum_partials(bpp_verifier *v, const void *d_partials, size_t n, uint32_t *d_ok, void *stream);

/* ---- grouped check -- per-proof verdicts at (nearly) the combined check's price; an engine mode, NOT a reference path ----
 * The reference verifies one proof at a time (src/range/mod.rs:57-78).  A service that needs the verdict of EVERY proof
 * but expects nearly all of them to be valid does not have to pay the per-proof MulVec: neighbouring proofs are checked
 * in groups of `group` (a power of two >= 2; 32 is a good default), sum_{p in group} w_p * M_p == identity, each group as
 * ONE virtual proof of the batch verifier's last stages, and only the proofs of a group that fails go through the exact
 * per-proof path (bpp_verifier_run) afterwards.  Weights, their key and index_base as for bpp_verifier_run_combined, and
 * the same conditions: a group holding an invalid proof passes with probability ~2^-128 if the weights were unpredictable
 * and the proof points lie in the prime-order subgroup (bpp_verifier_set_subgroup_check / the serialized path).
 *   d_out_verdicts : count x uint32_t, 0 = Ok / 1 = VerificationError -- the vector bpp_verifier_run writes
 *   stats          : HOST pointer, may be NULL: [groups that failed, proofs re-verified exactly]
 * The call synchronises `stream` (the list of failing groups comes back to the host between the two passes), so it
 * cannot be captured into a graph. */
size_t bpp_verifier_grouped_workspace_bytes(const bpp_verifier *v, size_t count, uint32_t group);
int bpp_verifier_run_grouped(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                             const uint64_t *d_challenges, const uint8_t *weight_key, uint64_t index_base,
                             const uint64_t *d_weights, uint32_t group, uint32_t *d_out_verdicts, uint64_t *stats,
                             void *d_workspace, size_t workspace_bytes, void *stream);

/* The same in two calls, so that ONE host thread can keep several batches in flight (a stream, a workspace and a verdict
 * buffer per batch): bpp_verifier_grouped_begin only enqueues the weighted checks of the groups; bpp_verifier_grouped_finish,
 * given the same buffers, count, group and stream, synchronises that stream, reads the groups' verdicts and re-verifies the
 * proofs of the failing ones.  begin(A), begin(B), finish(A), begin(C), finish(B), ... hides the latency-bound parts of one
 * batch behind the other (+13 % on (64,16) x 8192).  bpp_verifier_run_grouped = begin + finish. */
int bpp_verifier_grouped_begin(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                               const uint64_t *d_challenges, const uint8_t *weight_key, uint64_t index_base,
                               const uint64_t *d_weights, uint32_t group, uint32_t *d_out_verdicts, void *d_workspace,
                               size_t workspace_bytes, void *stream);
int bpp_verifier_grouped_finish(bpp_verifier *v, const uint64_t *d_points, const uint64_t *d_scalars, size_t count,
                                const uint64_t *d_challenges, uint32_t group, uint32_t *d_out_verdicts, uint64_t *stats,
                                void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- Fiat-Shamir transcript (csrc/transcript.hpp) -- what the reference's constants stand in for --------
 * The reference has no transcript (every challenge is a literal, src/range/mod.rs:278-279,
 * :417-418, src/weighted_inner_product_proof.rs:131, :211, :353, :369; the intended labels survive as a comment at
 * src/weighted_inner_product_proof.rs:339-348).  PARITY UNPINNED: pinned by oracle/pyref.py and the C oracle.
 * bpp_verifier_derive_challenges hashes each proof record of a resident batch (SHA-256, one lane per proof) into the
 * block [y, z, e, e_1..e_k] that bpp_verifier_run / bpp_verifier_run_combined accept as d_challenges:
 *   d_points     : count x (3 + 2k + m) wire points, as for bpp_verifier_run
 *   d_challenges : count x (3 + k) scalars (out)
 * bpp_range_prove_batch_fs is the prover under the same transcript (round t + 1 waits for L_t, R_t); outputs as
 * bpp_range_prove_batch_device.  An infinity enters the transcript as its canonical image (zero coordinates, flag = 1)
 * whatever non-zero flag word the caller wrote: one byte string per group element. */
int bpp_verifier_derive_challenges(bpp_verifier *v, const uint64_t *d_points, size_t count, uint64_t *d_challenges,
                                   void *stream);
/* Blinding.  The reference's blinding values are literals (alpha = 7 / 33, range/mod.rs:94,256; d_L = 4, d_R = 5,
 * wip.rs:94-95; r, s, delta, eta = 33, 44, 88, 123, wip.rs:175-178), so a proof made with them is sound but NOT hiding:
 * with those known, r', s', delta' give away the folded a, b and a linear combination of the gammas (for m = 1: gamma
 * itself, hence v).  That is kept, bit for bit, in the reference-parity calls (bpp_range_prove, bpp_range_prove_batch*).
 * Under the transcript the caller supplies the blinding: per proof alpha, r, s, delta, eta and d_L[t], d_R[t] for each of
 * the k rounds, one of
 *   d_blinding : count x (5 + 2k) canonical scalars on the device [alpha, r, s, delta, eta, d_L[0..k), d_R[0..k)]; or
 *   blind_key  : 32 secret bytes (host pointer) from the OS CSPRNG; the device expands slot j of proof p as
 *                (c0 + 2^256 c1) mod r, c_h = SHA-256(key || "bppb" || (index_base + p) as u64 LE || j as u32 LE || h as
 *                u32 LE) read little-endian (zero -> one).  index_base is the GLOBAL index of this call's first proof: a
 *                key must never meet the same index twice.
 * Both NULL: the literals above (for parity tests against the oracle's transcript-mode prover; such proofs leak).
 * host buffers, as bpp_range_prove_batch */
int bpp_range_prove_batch_fs(bpp_verifier *engine, const uint64_t *v, const uint64_t *gamma, size_t count,
                             const uint8_t *blind_key, uint64_t index_base, uint64_t *out_points, uint64_t *out_scalars,
                             uint64_t *out_V);
/* device buffers, as bpp_range_prove_batch_device; d_out_challenges: count x (3 + k) scalars [y, z, e, e_1..e_k] the
 * prover drew (may be NULL) */
int bpp_range_prove_batch_fs_device(bpp_verifier *engine, const uint64_t *d_v, const uint64_t *d_gamma, size_t count,
                                    const uint8_t *blind_key, uint64_t index_base, const uint64_t *d_blinding,
                                    uint64_t *d_out_points, uint64_t *d_out_scalars, uint64_t *d_out_V,
                                    uint64_t *d_out_challenges, void *d_workspace, size_t workspace_bytes, void *stream);

/* Points of unknown origin.  bpp_verifier_run takes wire points that are elements of the prime-order group by
 * construction (what the prover, mcl, or the decoders with their subgroup check produce), and on BLS12-381 evaluates the
 * proof-carried points through G1's endomorphism: for a curve point OUTSIDE G1 the MulVec is then not the full-curve sum
 * (on the order-3 point T = (0, 2) the engine forms (k1 - k2) T where the definition gives (k1 + k2 z^2) T), so a proof
 * whose R_0 was replaced by R_0 + T is ACCEPTED by the raw call where a full-curve evaluation rejects it
 * (tests/test_gpu_round3.py pins both outcomes).  on != 0 makes every wire point of a pass go through the membership
 * test of the decoders (csrc/ec.hpp aff_in_prime_subgroup; two multiplications by |z| per point, about +20 % on a (64,16)
 * pass); a point outside the group then counts as an invalid point and its proof gets verdict 1.  Off by default.  A
 * no-op on secp256k1 (cofactor 1); the edwards25519 instantiation works in ristretto255's quotient group instead: its
 * verdict test accepts a sum in E[4] (csrc/ristretto.hpp), i.e. points that differ by 4-torsion are THE SAME element
 * for every ed25519 entry point, raw wire points included -- the transcript hashes their ristretto255 encoding. */
int bpp_verifier_set_subgroup_check(bpp_verifier *v, int on);

/* Per-stage timing with HIP events recorded on the caller's stream around the kernels of a pass
 * (stages: 0 wire->Montgomery, 1 verifier scalars, 2 fixed-generator MSM [dominant; its first blocks also run
 * the Horner stage of the proof-point MSM], 3 proof-point MSM: digits, per-point tables, window sums,
 * 4 finalize).  The stages run back to back on one stream.  bpp_verifier_profile averages over the passes
 * recorded since profiling was switched on (at most 64): out_stage_ms[5]. */
int bpp_verifier_set_profiling(bpp_verifier *v, int on);
int bpp_verifier_profile(bpp_verifier *v, float *out_stage_ms, size_t *out_passes, unsigned *out_blocks_per_proof);

/* Host-pointer convenience over bpp_verifier_run (allocates, copies, synchronises). */
int bpp_range_verify_batch(bpp_verifier *v, const uint64_t *points, const uint64_t *scalars, size_t count,
                           uint32_t *out_ok);

/* ---- compressed point encodings: a data format next to the path ---------------------------------
 * The reference has no serialization; its commented-out size() functions (src/range/mod.rs:512-517,
 * src/weighted_inner_product_proof.rs:384-397) assume compressed points + 32-byte scalars and
 * src/errors.rs:20 reserves ProofError::FormatError for it.  PARITY UNPINNED by the reference; pinned by the
 * encodings' public generator vectors and oracle/pyref.py.
 *   BLS12-381 G1: 48 bytes, x big-endian, byte 0 bit 7 = compressed, bit 6 = infinity, bit 5 = y > (p-1)/2
 *   secp256k1   : 33 bytes, SEC1 02/03 || x big-endian; infinity = 33 zero bytes
 *   edwards25519: 32 bytes, ristretto255 (RFC 9496): the encoding of the prime-order quotient group, csrc/ristretto.hpp  Decompression runs on the device (one square root per point);
 * out_ok[i] = 0 valid, 1 malformed (flags, x >= p, x not on the curve) -- such a point is returned as infinity. */
size_t bpp_point_compressed_bytes(int curve_id);
int bpp_points_compress(bpp_ctx *ctx, const uint64_t *points, size_t n, uint8_t *out);
int bpp_points_decompress(bpp_ctx *ctx, const uint8_t *in, size_t n, uint64_t *out_points, uint32_t *out_ok);
/* device buffers, asynchronous on `stream`: feeds bpp_verifier_run's d_points without touching the host.
 * check_subgroup != 0: a curve point outside the prime-order subgroup is malformed too (BLS12-381 G1; what the container
 * decoder always does) */
int bpp_points_decompress_device(bpp_ctx *ctx, const void *d_in, size_t n, uint64_t *d_points, uint32_t *d_ok,
                                 int check_subgroup, void *stream);
/* bpp_range_verify_batch over serialized proofs: records = count x (3 + 2k + m) compressed points in the order
 * of d_points above, scalars = count x 3 (4 x u64 each).  out_ok[p] = 0 Ok / 1 VerificationError / 2 FormatError: a
 * malformed point encoding, a point outside the prime-order subgroup or a scalar >= the group order is a FormatError. */
int bpp_range_verify_batch_compressed(bpp_verifier *v, const uint8_t *records, const uint64_t *scalars, size_t count,
                                      uint32_t *out_ok);

/* ---- serialized proofs: the container -------------------------------------------------------------------
 * The reference never serializes a proof; its commented-out size() functions (src/range/mod.rs:512-517,
 * src/weighted_inner_product_proof.rs:384-397) count compressed points and 32-byte scalars, and src/errors.rs:20
 * reserves ProofError::FormatError for a deserializer.  PARITY UNPINNED; pinned by oracle/pyref.py's restatement.
 * One proof = bpp_proof_bytes(curve, n, m) bytes:
 *   "BPP+" | version = 1 | curve id | n | m | k = log2(n m) | 0 0 0            (12 bytes)
 *   A, wip.A, wip.B, L_0..L_{k-1}, R_0..R_{k-1}   compressed points (48 / 33 / 32 bytes each: BLS12-381 G1 ZCash form,
 *                                                 SEC1, ristretto255)
 *   r', s', delta'                                 32-byte little-endian scalars, canonical (< group order)
 * Decoding rejects with BPP_FORMAT_ERROR (2): a wrong header, a malformed point encoding, a point off the curve, a point
 * OUTSIDE THE PRIME-ORDER SUBGROUP (BLS12-381 G1 has a 126-bit cofactor: checked with the curve's endomorphism,
 * csrc/ec.hpp aff_in_prime_subgroup), a non-canonical scalar.  Host pointers. */
size_t bpp_proof_bytes(int curve_id, size_t n, size_t m);
int bpp_proofs_encode(bpp_ctx *ctx, size_t n, size_t m, const uint64_t *points, const uint64_t *scalars, size_t count,
                      uint8_t *out);
/* Container VERSION 2 (round 3): the same layout with UNCOMPRESSED points -- BLS12-381 G1 96 bytes (x | y big-endian, byte 0
 * bit 7 = 0, bit 6 = infinity, bit 5 = 0), secp256k1 SEC1 0x04 | x | y (65 bytes; infinity = 65 zero bytes); not offered
 * for ristretto255.  Decoding then needs no square root (a third of the decoder's arithmetic: 7.6 -> 5.4 ms per 8 192
 * (64,16) proofs) for 48 / 32 more bytes per point; every other check is the same (header with version = 2, coordinates
 * < p, on the curve, in the prime-order subgroup, canonical scalars).  The verify entry points take it with
 * BPP_SER_UNCOMPRESSED, and then expect the commitments uncompressed too (bpp_points_uncompressed). */
size_t bpp_point_uncompressed_bytes(int curve_id);
int bpp_points_uncompressed(bpp_ctx *ctx, const uint64_t *points, size_t n, uint8_t *out);
size_t bpp_proof_bytes_version(int curve_id, size_t n, size_t m, int version);
int bpp_proofs_encode_version(bpp_ctx *ctx, size_t n, size_t m, int version, const uint64_t *points, const uint64_t *scalars,
                              size_t count, uint8_t *out);
/* out_points: count x (3 + 2k) wire points (infinity where an encoding was rejected); out_status: 0 / BPP_FORMAT_ERROR.
 * Reads version 1 containers (a version 2 header is a FormatError here: those are consumed by the verify entry points). */
int bpp_proofs_decode(bpp_ctx *ctx, size_t n, size_t m, const uint8_t *in, size_t count, uint64_t *out_points,
                      uint64_t *out_scalars, uint32_t *out_status);
/* RangeProof::verify for `count` serialized proofs: proofs count x bpp_proof_bytes, commitments count x m compressed
 * points.  flags: BPP_SER_TRANSCRIPT (1): challenges from the Fiat-Shamir transcript instead of the reference's
 * constants; BPP_SER_UNCOMPRESSED (2): container version 2, proofs count x bpp_proof_bytes_version(.., 2) and the
 * commitments count x m uncompressed points.  out_ok[p] = 0 Ok / 1 VerificationError / 2 FormatError. */
#define BPP_SER_TRANSCRIPT 1
#define BPP_SER_UNCOMPRESSED 2
int bpp_range_verify_batch_serialized(bpp_verifier *v, const uint8_t *proofs, const uint8_t *commitments, size_t count,
                                      int flags, uint32_t *out_ok);
/* The same with every buffer in HBM, asynchronous on `stream`, no host synchronisation: what a service that receives
 * proofs off the wire calls after one copy.  One kernel decodes the containers and the commitments (header, point
 * encodings with the subgroup check, scalar canonicity) straight into bpp_verifier_run's record layout inside the
 * workspace; d_ok[p] = 0 / 1 / 2 as above.  d_workspace: bpp_verifier_serialized_workspace_bytes(v, count) bytes. */
size_t bpp_verifier_serialized_workspace_bytes(const bpp_verifier *v, size_t count);
int bpp_range_verify_batch_serialized_device(bpp_verifier *v, const void *d_proofs, const void *d_commitments, size_t count,
                                             int flags, uint32_t *d_ok, void *d_workspace, size_t workspace_bytes,
                                             void *stream);
/* ... and with the grouped check (above) behind the decoder: the same status vector (0 / 1 / 2 per proof) at the grouped
 * check's price when the batch is (nearly) all valid.  weight_key: 32 fresh secret bytes (host); index_base, group, stats as
 * for bpp_verifier_run_grouped; the decoder's subgroup check provides the prime-order points the weighted check assumes.
 * Synchronises `stream`.  d_workspace: bpp_verifier_serialized_grouped_workspace_bytes(v, count, group) bytes. */
size_t bpp_verifier_serialized_grouped_workspace_bytes(const bpp_verifier *v, size_t count, uint32_t group);
int bpp_range_verify_batch_serialized_grouped_device(bpp_verifier *v, const void *d_proofs, const void *d_commitments,
                                                     size_t count, int flags, const uint8_t *weight_key, uint64_t index_base,
                                                     uint32_t group, uint32_t *d_ok, uint64_t *stats, void *d_workspace,
                                                     size_t workspace_bytes, void *stream);

/* name of the kernel that dominates bpp_verifier_run (for profilers) and its launch geometry */
const char *bpp_verifier_dominant_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* BPP_AMD_H */
