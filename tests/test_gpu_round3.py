"""-m gpu: round-3 behaviour through the C ABI.
  * blinding supplied by the caller to the transcript-mode prover (include/bpp_amd.h "Blinding") == the C oracle's
    transcript-mode prover with the same blinding values, bit for bit; the reference's literals
    (src/range/mod.rs:94,256; src/weighted_inner_product_proof.rs:94-95,175-178) stay the default of the parity calls
  * points outside BLS12-381's G1: what the raw wire call does with R_0 + T (T of order 3), pinned; the opt-in subgroup
    check (bpp_verifier_set_subgroup_check) rejects it, like the reference's full-curve sum
    (src/bls12_381/building_block/point/point.rs:69-85 = mcl G1::mul) would
  * edwards25519: a proof point shifted by 4-torsion is the same ristretto255 element
  * an infinity with a non-canonical flag word hashes like the canonical one"""

import numpy as np
import pytest

import oracle as O
import pyref as P
from gpu_util import need_gpu, run_verifier_device

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname,cid,n,vals,gams", [
    ("bls12_381", 0, 8, [200, 5], [3, 7]),
    ("secp256k1", 1, 8, [1, 254], [11, 2]),
    ("bls12_381", 0, 8, [77], [9]),
])
def test_blinded_transcript_prover_matches_oracle(cname, cid, n, vals, gams):
    torch = need_gpu()
    import bulletproofsplus_amd as B
    m = len(vals)
    a = B.Arith.init(cname)
    r = P.CURVES[cname]["r"]
    opk = O.PublicKey(cid, n * m)
    pk = B.PublicKey.from_points(a, opk.gh, opk.G, opk.H)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    k = bv.k
    key = bytes(range(32))
    base = 1000
    vals2 = [vals, [(v * 3 + 1) % (1 << n) for v in vals], vals]
    gams2 = [gams, [g + 1 for g in gams], gams]
    pts, scs, V = bv.prove_batch(vals2, gams2, transcript=True, blind_key=key, index_base=base)
    O.set_transcript(True)
    try:
        for i in range(3):
            O.set_blinding(O.blinding_from_key(key, base + i, k, r))
            opts, osc, oV = O.range_prove(opk, n, vals2[i], gams2[i])
            assert np.array_equal(pts[i], opts) and np.array_equal(scs[i], osc) and np.array_equal(V[i], oV), i
            rc = O.range_verify(opk, n, m, opts, osc, oV)
            assert (rc[0] if isinstance(rc, tuple) else rc) == 0
    finally:
        O.set_blinding(None)
        O.set_transcript(False)
    # the same values under two proof indices: different blinding, different proofs, same commitments
    assert not np.array_equal(pts[0], pts[2]) and not np.array_equal(scs[0], scs[2]) and np.array_equal(V[0], V[2])
    # ... and the literals when no key is given (the oracle's transcript-mode prover as it was)
    pts_l, scs_l, _ = bv.prove_batch(vals2[:1], gams2[:1], transcript=True)
    O.set_transcript(True)
    try:
        opts, osc, _ = O.range_prove(opk, n, vals, gams)
    finally:
        O.set_transcript(False)
    assert np.array_equal(pts_l[0], opts) and np.array_equal(scs_l[0], osc)
    # the device verifier accepts the blinded proofs under the transcript
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1))
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_ch = torch.zeros((3, 3 + k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), 3, d_ch.data_ptr())
    torch.cuda.synchronize()
    ch = d_ch.cpu().numpy().view(np.uint64)
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False, challenges=ch)
    assert ok.tolist() == [0, 0, 0]
    bv.close()


def test_point_outside_g1_pinned():
    """(8,1) on BLS12-381, R_0 replaced by R_0 + T with T = (0, 2) of order 3.  The MulVec scalar of R_j is e_j^-2 e^2
    (wip.rs:303-304) = k with k mod 3 = 1 but (k mod z^2) - (k div z^2) = 0 mod 3: the full-curve sum is k T != O
    (reject), the endomorphism evaluation gives (k1 - k2) T = O (accept).  The raw call accepts -- documented in
    include/bpp_amd.h -- and the subgroup check rejects."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 1
    a = B.Arith.init("bls12_381")
    r = P.BLS12_381["r"]
    z2 = 0xd201000000010000 ** 2
    kR = pow(49, -1, r) * 9801 % r
    assert kR % 3 == 1 and (kR % z2 - kR // z2) % 3 == 0          # the disagreement this test is about
    opk = O.PublicKey(0, n * m)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    k = bv.k
    opts, osc, oV = O.range_prove(opk, n, [200], [3])
    T = O.point_to_wire(0, (0, 2))
    assert O.on_curve(0, T) and a.is_zero(O.point_add(0, O.point_add(0, T, T), T))
    bad = opts.copy()
    bad[3 + k] = O.point_add(0, opts[3 + k], T)
    assert O.on_curve(0, bad[3 + k])
    rc = O.range_verify(opk, n, m, bad, osc, oV)
    assert (rc[0] if isinstance(rc, tuple) else rc) == 1            # the definition (full-curve sum): reject
    recs = np.stack([np.concatenate([opts, oV]), np.concatenate([bad, oV])])
    scs = np.stack([osc, osc])
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False)
    assert ok.tolist() == [0, 0]                                    # raw wire call: the endomorphism evaluation accepts
    bv.set_subgroup_check(True)
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False)
    assert ok.tolist() == [0, 1]                                    # with the membership test: an invalid point
    bv.set_subgroup_check(False)
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False)
    assert ok.tolist() == [0, 0]
    bv.close()


def test_ed25519_four_torsion_is_the_same_element():
    """edwards25519 entry points work in ristretto255's quotient group: A + T4 (T4 of order 4) is the same element as A for
    the raw wire call too -- verdict 0 -- and the transcript (ristretto255 encodings) yields the same challenges."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    a = B.Arith.init("ed25519")
    G = P.EdwardsGroup(P.ED25519)
    p = P.ED25519["p"]
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    pts, scs, V = bv.prove_batch([[200, 5]], [[3, 7]])
    i4 = pow(2, (p - 1) // 4, p)
    T4 = (i4, 0)
    assert G.on_curve(T4) and not G.is_zero(G.mul(T4, 2)) and G.is_zero(G.mul(T4, 4))
    A = O.wire_to_point(2, pts[0, 0])
    shifted = pts.copy()
    shifted[0, 0] = O.point_to_wire(2, G.add(A, T4))
    recs = np.concatenate([np.concatenate([pts, V], axis=1), np.concatenate([shifted, V], axis=1)])
    ok, _, _ = run_verifier_device(torch, bv, recs, np.concatenate([scs, scs]), want_scalars=False, want_result=False)
    assert ok.tolist() == [0, 0]
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(np.ascontiguousarray(recs).view(np.int64)).to(dev)
    d_ch = torch.zeros((2, 3 + bv.k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), 2, d_ch.data_ptr())
    torch.cuda.synchronize()
    ch = d_ch.cpu().numpy()
    assert np.array_equal(ch[0], ch[1])
    bv.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("secp256k1", 1)])
def test_transcript_hashes_canonical_infinity(cname, cid):
    """aff_from_wire reads any non-zero flag word as infinity and ignores x, y: the transcript hashes the canonical image,
    so the challenges do not depend on those bytes"""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    a = B.Arith.init(cname)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    pts, scs, V = bv.prove_batch([[200, 5]], [[3, 7]])
    rec = np.concatenate([pts, V], axis=1)[0]
    r1, r2 = rec.copy(), rec.copy()
    r1[4] = a.zero_point()                     # canonical infinity in place of L_1
    r2[4] = rec[5]                             # some coordinates ...
    r2[4, 2 * a.L] = 5                         # ... under a non-canonical flag word
    recs = np.ascontiguousarray(np.stack([r1, r2]))
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_ch = torch.zeros((2, 3 + bv.k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), 2, d_ch.data_ptr())
    torch.cuda.synchronize()
    ch = d_ch.cpu().numpy()
    assert np.array_equal(ch[0], ch[1])
    bv.close()


@pytest.mark.parametrize("cname,cid,n,vals,gams", [
    ("bls12_381", 0, 8, [200, 5], [3, 7]),
    ("bls12_381", 0, 32, [31], [7]),          # C1 of BASELINE.json: n = 32, m = 1
    ("secp256k1", 1, 8, [1, 254], [11, 2]),
])
def test_range_verify_cached_and_uncached_agree_with_oracle(cname, cid, n, vals, gams):
    """RangeProof::verify without a verifier object (src/range/mod.rs:57-78): the first call with a key runs the naive
    MulVec, the second builds that key's small window tables, later ones use them -- every call returns the oracle's
    verdict for valid / tampered / wrong-key input, and so does the path with the cache switched off."""
    need_gpu()
    import bulletproofsplus_amd as B
    m = len(vals)
    a = B.Arith(cname)           # a context of its own: the cache belongs to the context
    opk = O.PublicKey(cid, n * m)
    pk = B.PublicKey.new(a, n * m)
    pr = B.RangeProver.new()
    for v, g in zip(vals, gams):
        pr.commit(pk, v, g)
    proof = B.RangeProof.prove(pk, n, pr)
    opts, osc, oV = O.range_prove(opk, n, vals, gams)
    assert np.array_equal(proof.points_wire(), opts) and np.array_equal(proof.scalars_wire(), osc)
    bad = B.RangeProof.from_wire(opts, osc)
    bad.proof.s_prime = bad.proof.s_prime.copy()
    bad.proof.s_prime[0] ^= np.uint64(2)
    moved = B.RangeProof.from_wire(opts, osc)
    moved.proof.L_vec = moved.proof.L_vec.copy()
    moved.proof.L_vec[0] = opts[1]
    G2 = pk.G_vec.copy()
    G2[[0, 1]] = G2[[1, 0]]
    pk2 = B.PublicKey.from_points(a, pk.gh, G2, pk.H_vec)     # a different key: the same generators in another order

    def verdict(p, key):
        try:
            p.verify(key, n, pr.commitment_vec)
            return 0
        except B.VerificationError:
            return 1

    opk2 = O.PublicKey(cid, n * m)
    opk2.G = G2

    def oracle_verdict(p, swapped):
        return O.range_verify(opk2 if swapped else opk, n, m, p.points_wire(), p.scalars_wire(), oV)

    seq = [(proof, pk, False), (proof, pk, False), (bad, pk, False), (moved, pk, False), (proof, pk2, True), (proof, pk2, True),
           (bad, pk2, True), (proof, pk, False), (bad, pk, False)]
    got = [verdict(p, key) for p, key, _ in seq]
    exp = [oracle_verdict(p, sw) for p, _, sw in seq]
    assert got == exp == [0, 0, 1, 1, 1, 1, 1, 0, 1], (got, exp)
    a.set_verify_cache(False)
    assert [verdict(p, key) for p, key, _ in seq] == exp
    a.set_verify_cache(True)
    assert [verdict(p, key) for p, key, _ in seq] == exp


def test_two_verifiers_coresident_interleaved():
    """A service with two shapes: the C2 (64,16) and C3 (64,1) verifiers side by side in HBM, batches interleaved on one
    stream -- each batch gets exactly its own verdicts (one tampered proof each), checked against the oracle for the
    tampered proofs."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init("bls12_381")
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    shapes = {"c2": (64, 16, 96, 12), "c3": (64, 1, 200, 12)}
    eng, bufs = {}, {}
    for name, (n, m, batch, c) in shapes.items():
        pk = B.PublicKey.new(a, n * m)
        bv = B.BatchVerifier(pk, n, m, window_bits=c)
        vals = [[(7 * d + 3 * j + 1) % (1 << 20) for j in range(m)] for d in range(8)]
        gams = [[d + j + 2 for j in range(m)] for d in range(8)]
        pts, sc, V = bv.prove_batch(vals, gams)
        ix = np.arange(batch) % 8
        recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1)[ix])
        scs = np.ascontiguousarray(sc[ix])
        bad = batch // 3
        scs[bad, 2, 0] ^= np.uint64(1)
        opk = O.PublicKey(0, n * m)
        assert O.range_verify(opk, n, m, recs[bad, :pts.shape[1]], scs[bad], recs[bad, pts.shape[1]:]) == 1
        assert O.range_verify(opk, n, m, recs[0, :pts.shape[1]], scs[0], recs[0, pts.shape[1]:]) == 0
        wsb = bv.workspace_bytes(batch)
        bufs[name] = (torch.from_numpy(recs.view(np.int64)).to(dev), torch.from_numpy(scs.view(np.int64)).to(dev),
                      torch.full((batch,), 7, dtype=torch.int32, device=dev), torch.empty(wsb, dtype=torch.uint8, device=dev), wsb,
                      batch, bad)
        eng[name] = bv
    for _ in range(3):
        for name in ("c2", "c3", "c3", "c2"):
            p, s_, ok, ws, wsb, batch, bad = bufs[name]
            eng[name].run_device(p.data_ptr(), s_.data_ptr(), batch, ok.data_ptr(), ws.data_ptr(), wsb, st)
    torch.cuda.synchronize()
    for name in ("c2", "c3"):
        ok, batch, bad = bufs[name][2].cpu().numpy(), bufs[name][5], bufs[name][6]
        want = np.zeros(batch, dtype=ok.dtype)
        want[bad] = 1
        assert np.array_equal(ok, want), name
        eng[name].close()


@pytest.mark.parametrize("count", [1, 5, 300])
def test_native_pass_graph_replays(count):
    """bpp_verifier_graph_capture / bpp_graph_launch: the pass captured by the library itself (no PyTorch capture in between);
    replays follow the contents of the captured buffers, on the caller's stream."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    a = B.Arith.init("bls12_381")
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    vals = [[(11 * i + j) % 256 for j in range(m)] for i in range(count)]
    gams = [[1 + i + 2 * j for j in range(m)] for i in range(count)]
    pts, scs, V = bv.prove_batch(vals, gams)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1))
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(np.ascontiguousarray(scs).view(np.int64)).to(dev)
    d_ok = torch.full((count,), 7, dtype=torch.int32, device=dev)
    wsb = bv.workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    g = bv.graph_capture(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb)
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(3):
        d_ok.fill_(7)
        g.launch(stream)
        torch.cuda.synchronize()
        assert d_ok.cpu().numpy().tolist() == [0] * count
    bad = scs.copy()
    bad[count // 2, 1, 0] ^= np.uint64(2)
    d_sc.copy_(torch.from_numpy(np.ascontiguousarray(bad).view(np.int64)))
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    g.launch(s2.cuda_stream)
    s2.synchronize()
    assert d_ok.cpu().numpy().tolist() == [1 if i == count // 2 else 0 for i in range(count)]
    bv.set_profiling(True)
    with pytest.raises(B.BppError):
        bv.graph_capture(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb)
    bv.set_profiling(False)
    g.close()
    bv.close()
