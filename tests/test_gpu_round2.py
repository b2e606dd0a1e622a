"""-m gpu: the configurations and geometries the round-1 review found untested.

 * C3 of BASELINE.json: 4096 independent (n=64, m=1) proofs at the bench's window width -- valid + tampered, the
   verifier's MulVec scalars and result point of a sample against the oracle, verdicts for all.
 * the bench geometry of C2: (64,16) at c = 17 with 8192 proofs (4 blocks per proof): a tampered subset must give
   exactly the expected verdict vector, and the MulVec result point of two proofs must equal the oracle's.
 * the Fiat-Shamir transcript on the device against its hashlib restatement (oracle/pyref.FsTranscript), and
   verification under derived challenges.
 * combined check: caller-supplied weights, the PRF's global index, the validity word that travels with a partial.
All through the C ABI; the oracle is the checker."""

import hashlib

import numpy as np
import pytest

import oracle as O
import pyref as P
from gpu_util import need_gpu, run_verifier_device, run_combined_device

pytestmark = pytest.mark.gpu


def _values(seed, m):
    vals = [((0x9E3779B97F4A7C15 * (j + 1 + seed)) & 0xFFFFFFFFFFFFFFFF) % (1 << 31) for j in range(m)]
    gams = [j + 3 + seed for j in range(m)]
    return vals, gams


def _prove_batch(bv, count, m, seed0=0, nbits=64):
    vals, gams = [], []
    for d in range(count):
        v, g = _values(seed0 + 17 * d, m)
        vals.append([x % (1 << nbits) for x in v])      # in range for an n-bit proof
        gams.append(g)
    pts, scs, V = bv.prove_batch(vals, gams)
    return np.ascontiguousarray(np.concatenate([pts, V], axis=1)), np.ascontiguousarray(scs), vals, gams


def _tamper(recs, scs, which):
    """bit flips in r' / s' / delta' and exchanged A points, cycling over the four kinds"""
    rec_t, sc_t = recs.copy(), scs.copy()
    B = recs.shape[0]
    for j, i in enumerate(which):
        kind = j % 4
        if kind < 3:
            sc_t[i, kind, 0] ^= np.uint64(1 << (j % 60))
        else:
            # the A of another proof.  With the reference's generators (small multiples of g, publickey.rs:23-39) two
            # different values can give the same A, so look for one that really differs
            src = next(s for s in range(i + 1, i + B) if not np.array_equal(recs[s % B, 0], recs[i, 0])) % B
            rec_t[i, 0] = recs[src, 0]
    return rec_t, sc_t


def test_c3_batch_4096_n64_m1_at_bench_window():
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m, count, c = 64, 1, 4096, 16
    a = B.Arith.init("bls12_381")
    opk = O.PublicKey(O.BLS12_381, n * m)
    pk = B.PublicKey.new(a, n * m)
    assert np.array_equal(pk.G_vec, opk.G) and np.array_equal(pk.H_vec, opk.H)
    bv = B.BatchVerifier(pk, n, m, window_bits=c)
    assert bv.msm_len == 146
    recs, scs, vals, gams = _prove_batch(bv, count, m)
    k = bv.k
    # the GPU-proved proofs are the oracle's proofs (sample), and all verify
    for i in (0, 1, 2047, 4095):
        opts, osc, oV = O.range_prove(opk, n, vals[i], gams[i])
        assert np.array_equal(recs[i, :3 + 2 * k], opts) and np.array_equal(recs[i, 3 + 2 * k:], oV)
        assert np.array_equal(scs[i], osc)
    ok, vsc, res = run_verifier_device(torch, bv, recs, scs)
    assert ok.tolist() == [0] * count
    assert all(int(r[2 * a.L]) == 1 for r in res)                       # every MulVec result is the identity
    for i in (0, 777, 4095):                                             # scalars in the reference's MulVec order
        rc, esc, eres = O.range_verify(opk, n, m, recs[i, :3 + 2 * k], scs[i], recs[i, 3 + 2 * k:], want_scalars=True,
                                       want_result=True)
        assert rc == 0 and np.array_equal(vsc[i], esc) and np.array_equal(res[i], eres)
    # tampered subset: exact verdict vector, and the same non-identity result point as the oracle
    rs = np.random.RandomState(3)
    which = np.sort(rs.choice(count, size=97, replace=False))
    rec_t, sc_t = _tamper(recs, scs, which)
    ok, vsc, res = run_verifier_device(torch, bv, rec_t, sc_t)
    want = np.zeros(count, dtype=np.uint32)
    want[which] = 1
    assert np.array_equal(ok, want)
    for i in which[:6]:
        rc, esc, eres = O.range_verify(opk, n, m, rec_t[i, :3 + 2 * k], sc_t[i], rec_t[i, 3 + 2 * k:], want_scalars=True,
                                       want_result=True)
        assert rc == 1 and np.array_equal(vsc[i], esc) and np.array_equal(res[i], eres)
    bv.close()


def test_c2_bench_geometry_c17_8192_proofs():
    """(64,16), window 17, 8192 distinct proofs: the launch geometry bench.py times (4 blocks per proof)."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m, count, c = 64, 16, 8192, 17
    a = B.Arith.init("bls12_381")
    opk = O.PublicKey(O.BLS12_381, n * m)
    pk = B.PublicKey.new(a, n * m)
    torch.cuda.empty_cache()
    free_before, _ = torch.cuda.mem_get_info()
    try:
        bv = B.BatchVerifier(pk, n, m, window_bits=c)
    except B.BppError as e:
        # the one test of the bench geometry: a box that HAS the memory (204 GB of tables + workspace) and still says
        # NOMEM is a failure, not a skip
        if e.code == -5 and free_before < 225 * (1 << 30):
            pytest.skip("c = 17 tables need 204 GB of free HBM; this box has %.0f GB free" % (free_before / 2**30))
        raise
    recs, scs, vals, gams = _prove_batch(bv, count, m, seed0=5)
    k = bv.k
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_ok = torch.full((count,), 7, dtype=torch.int32, device=dev)
    wsb = bv.workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    d_res = torch.zeros((count, a.PW), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(sc):
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(dev)
        d_ok.fill_(7)
        bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb, stream,
                      d_out_result=d_res.data_ptr())
        torch.cuda.synchronize()
        return d_ok.cpu().numpy().astype(np.uint32), d_res.cpu().numpy().view(np.uint64)

    bv.set_profiling(True)
    ok, res = run(scs)
    _, _, bpp_ = bv.profile()
    bv.set_profiling(False)
    assert bpp_ == 4                                                    # the geometry of the bench
    assert ok.tolist() == [0] * count
    which = np.sort(np.random.RandomState(11).choice(count, size=64, replace=False))
    sc_t = scs.copy()
    for j, i in enumerate(which):
        sc_t[i, j % 3, (j // 3) % 4] ^= np.uint64(1 << (j % 63))
    ok, res = run(sc_t)
    want = np.zeros(count, dtype=np.uint32)
    want[which] = 1
    assert np.array_equal(ok, want)
    # result points of one valid and one tampered proof against the oracle (0.6 s of CPU each)
    good = int(np.setdiff1d(np.arange(count), which)[5])
    for i, exp_rc in ((good, 0), (int(which[3]), 1)):
        rc, _, eres = O.range_verify(opk, n, m, recs[i, :3 + 2 * k], sc_t[i], recs[i, 3 + 2 * k:], want_result=True)
        assert rc == exp_rc and np.array_equal(res[i], eres)
    bv.close()


@pytest.mark.parametrize("cname,cid,n,vals", [("secp256k1", 1, 8, [200, 5]), ("bls12_381", 0, 4, [9, 3])])
def test_transcript_on_device(cname, cid, n, vals):
    """derive_challenges == the hashlib restatement; a proof made under the transcript verifies under the derived
    challenges and is rejected under the reference's constants, and vice versa."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    m = len(vals)
    G = P.make_group(cname, False)
    ppk = P.PublicKey(G, n * m)
    fs = P.FsTranscript(P.CURVES[cname], cid, n, m, ppk)
    gam = [3 + j for j in range(m)]
    P.Transcript.fs = fs
    try:
        _, prover, proof_fs = P.prove_case(cname, n, vals, gam, shadow=False)
        ch_fs = fs.verifier_challenges(proof_fs, prover.commitment_vec)
    finally:
        P.Transcript.fs = None
    _, prover_c, proof_c = P.prove_case(cname, n, vals, gam, shadow=False)
    ch_c = fs.verifier_challenges(proof_c, prover_c.commitment_vec)

    def record(proof, prover):
        w = proof.proof
        pts = O.points_to_wire(cid, [proof.A, w.A, w.B] + list(w.L_vec) + list(w.R_vec) + list(prover.commitment_vec))
        return pts, O.scalars_to_wire([w.r_prime, w.s_prime, w.d_prime])

    a = B.Arith.init(cname)
    opk = O.PublicKey(cid, n * m)
    pk = B.PublicKey.from_points(a, opk.gh, opk.G, opk.H)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    r_fs, s_fs = record(proof_fs, prover)
    r_c, s_c = record(proof_c, prover_c)
    recs = np.stack([r_fs, r_c])
    scs = np.stack([s_fs, s_c])
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(np.ascontiguousarray(recs).view(np.int64)).to(dev)
    k = bv.k
    d_ch = torch.zeros((2, 3 + k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), 2, d_ch.data_ptr())
    torch.cuda.synchronize()
    got = d_ch.cpu().numpy().view(np.uint64)
    for row, ch in ((0, ch_fs), (1, ch_c)):
        exp = [ch["y"], ch["z"], ch["e"]] + ch["e_rounds"]
        assert [O.wire_to_scalars(got[row, i:i + 1])[0] for i in range(3 + k)] == exp
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False, challenges=got)
    assert ok.tolist() == [0, 1]            # the transcript's proof passes, the constants' proof fails
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False)
    assert ok.tolist() == [1, 0]            # and under the reference's constants it is the other way round
    bv.close()


def test_combined_check_weights_and_validity_word():
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    a = B.Arith.init("bls12_381")
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=6)
    recs, scs, _, _ = _prove_batch(bv, 8, m, nbits=n)
    bad = scs.copy()
    bad[5, 1, 0] ^= np.uint64(8)
    # caller-supplied 128-bit weights
    w = np.random.RandomState(1).randint(1, 2**62, size=(8, 2)).astype(np.uint64)
    assert run_combined_device(torch, bv, recs, scs, weights=w)[0] == 0
    assert run_combined_device(torch, bv, recs, bad, weights=w)[0] == 1
    # the PRF's domain is the GLOBAL proof index: the shards of one batch under one key see the weights of the whole
    dev = torch.device("cuda:0")
    d_ok = torch.full((1,), 7, dtype=torch.int32, device=dev)

    def summed(parts):
        d_parts = torch.from_numpy(np.concatenate(parts)).to(dev)
        bv.sum_partials_device(d_parts.data_ptr(), len(parts), d_ok.data_ptr())
        torch.cuda.synchronize()
        return int(d_ok.item())

    for sc, exp in ((scs, 0), (bad, 1)):
        parts = [run_combined_device(torch, bv, recs[lo:hi], sc[lo:hi], seed=4, index_base=lo)[1] for lo, hi in ((0, 3), (3, 8))]
        assert summed(parts) == exp
    # weights that repeat across shards (same key, index_base 0 for both) still detect a single bad proof
    parts = [run_combined_device(torch, bv, recs[lo:hi], bad[lo:hi], seed=4, index_base=0)[1] for lo, hi in ((0, 3), (3, 8))]
    assert summed(parts) == 1
    # the validity word travels with the partial: a shard whose proof carries an off-curve point reports it locally
    # AND through the cross-rank sum, whatever its jacobian sum happens to be
    off = recs[:3].copy()
    off[1, 2, 0] ^= np.uint64(1)                        # wip.B of proof 1: x coordinate off the curve
    okf, part_bad = run_combined_device(torch, bv, off, scs[:3], seed=4, index_base=0)
    assert okf == 1
    flag_word = bv.partial_bytes() // 4 - 4
    assert part_bad.view(np.uint32)[flag_word] == 1
    part_ok = run_combined_device(torch, bv, recs[3:], scs[3:], seed=4, index_base=3)[1]
    assert part_ok.view(np.uint32)[flag_word] == 0
    part_clean = run_combined_device(torch, bv, recs[:3], scs[:3], seed=4, index_base=0)[1]
    assert summed([part_clean, part_ok]) == 0
    forged = part_clean.copy()
    forged.view(np.uint32)[flag_word] = 1             # identity sums, but the word says a point was invalid
    assert summed([forged, part_ok]) == 1
    bv.close()


def test_length_mismatches_raise():
    """the reference panics on these (bounds-checked slices, mulvec.rs:23-25); the mirror raises"""
    need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init("secp256k1")
    pk = B.PublicKey.new(a, 16)
    short = B.PublicKey.from_points(a, pk.gh, pk.G_vec[:8], pk.H_vec[:8])
    with pytest.raises(AssertionError):
        B.BatchVerifier(short, 8, 2, window_bits=4)
    pr = B.RangeProver.new()
    for v, g in ((200, 3), (5, 7)):
        pr.commit(pk, v, g)
    proof = B.RangeProof.prove(pk, 8, pr)
    with pytest.raises(AssertionError):
        proof.verify(short, 8, pr.commitment_vec)
    bv = B.BatchVerifier(pk, 8, 2, window_bits=4)
    rec = B.proof_record(proof, pr.commitment_vec)
    with pytest.raises(RuntimeError):
        bv.verify_wire(np.stack([rec, rec]), proof.scalars_wire()[None])
    bv.close()


@pytest.mark.parametrize("cname,cid,n,vals,gams,c", [
    ("secp256k1", 1, 8, [200, 5], [3, 7], 5),
    ("bls12_381", 0, 8, [77], [9], 4),
    ("bls12_381", 0, 4, [9, 3, 15, 0], [1, 2, 3, 4], 6),
])
def test_fs_prover_small_bit_exact(cname, cid, n, vals, gams, c):
    """The batched prover under the Fiat-Shamir transcript == the C oracle's transcript-mode prover, bit for bit, and
    the challenges it drew == the ones the verifier derives from the proof."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    m = len(vals)
    a = B.Arith.init(cname)
    opk = O.PublicKey(cid, n * m)
    pk = B.PublicKey.from_points(a, opk.gh, opk.G, opk.H)
    bv = B.BatchVerifier(pk, n, m, window_bits=c)
    O.set_transcript(True)
    try:
        opts, osc, oV = O.range_prove(opk, n, vals, gams)
        rc, _, _, och = O.range_verify(opk, n, m, opts, osc, oV, want_challenges=True)
        assert rc == 0
    finally:
        O.set_transcript(False)
    vals2 = [[(v + 1) % (1 << n) for v in vals], vals]           # a second, different proof in the batch
    gams2 = [[g + 5 for g in gams], gams]
    pts, scs, V = bv.prove_batch(vals2, gams2, transcript=True)
    assert np.array_equal(pts[1], opts) and np.array_equal(scs[1], osc) and np.array_equal(V[1], oV)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1))
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_ch = torch.zeros((2, 3 + bv.k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), 2, d_ch.data_ptr())
    torch.cuda.synchronize()
    ch = d_ch.cpu().numpy().view(np.uint64)
    assert np.array_equal(ch[1], och)
    ok, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False, challenges=ch)
    assert ok.tolist() == [0, 0]
    bad = scs.copy()
    bad[0, 2, 0] ^= np.uint64(1)
    ok, _, _ = run_verifier_device(torch, bv, recs, bad, want_scalars=False, want_result=False, challenges=ch)
    assert ok.tolist() == [1, 0]
    # the constants-mode prover of the same engine is untouched by the transcript code path
    cpts, csc, cV = O.range_prove(opk, n, vals, gams)
    pts_c, scs_c, V_c = bv.prove_batch([vals], [gams])
    assert np.array_equal(pts_c[0], cpts) and np.array_equal(scs_c[0], csc) and np.array_equal(V_c[0], cV)
    bv.close()


def test_fs_prover_reference_size_64x16():
    """(64,16) under the transcript: device-resident prover (challenge block returned) -> the oracle's transcript-mode
    verifier accepts proof 0 and derives the same challenges; the device verifier accepts all, rejects a tampered one."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m, count = 64, 16, 6
    a = B.Arith.init("bls12_381")
    opk = O.PublicKey(O.BLS12_381, n * m)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=10)
    k = bv.k
    dev = torch.device("cuda:0")
    vals = np.array([_values(3 + 17 * d, m)[0] for d in range(count)], dtype=np.uint64)
    gam = np.zeros((count, m, 4), dtype=np.uint64)
    for d in range(count):
        gam[d, :, 0] = np.array(_values(3 + 17 * d, m)[1], dtype=np.uint64)
    d_v = torch.from_numpy(vals.view(np.int64)).to(dev)
    d_g = torch.from_numpy(gam.view(np.int64)).to(dev)
    d_po = torch.zeros((count, 3 + 2 * k, a.PW), dtype=torch.int64, device=dev)
    d_ps = torch.zeros((count, 3, 4), dtype=torch.int64, device=dev)
    d_pV = torch.zeros((count, m, a.PW), dtype=torch.int64, device=dev)
    d_ch = torch.zeros((count, 3 + k, 4), dtype=torch.int64, device=dev)
    wsb = bv.prover_workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bv.prove_batch_device(d_v.data_ptr(), d_g.data_ptr(), count, d_po.data_ptr(), d_ps.data_ptr(), d_pV.data_ptr(),
                          d_ws.data_ptr(), wsb, transcript=True, d_out_challenges=d_ch.data_ptr())
    torch.cuda.synchronize()
    pts = d_po.cpu().numpy().view(np.uint64)
    scs = d_ps.cpu().numpy().view(np.uint64)
    V = d_pV.cpu().numpy().view(np.uint64)
    ch = d_ch.cpu().numpy().view(np.uint64)
    O.set_transcript(True)
    try:
        rc, _, _, och = O.range_verify(opk, n, m, pts[0], scs[0], V[0], want_challenges=True)
    finally:
        O.set_transcript(False)
    assert rc == 0 and np.array_equal(ch[0], och)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1))
    d_rec = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_ch2 = torch.zeros_like(d_ch)
    bv.derive_challenges_device(d_rec.data_ptr(), count, d_ch2.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_ch2.cpu().numpy().view(np.uint64), ch)
    bad = scs.copy()
    bad[4, 0, 0] ^= np.uint64(2)
    ok, _, _ = run_verifier_device(torch, bv, recs, bad, want_scalars=False, want_result=False, challenges=ch)
    assert ok.tolist() == [0, 0, 0, 0, 1, 0]
    bv.close()


@pytest.mark.parametrize("cname", ["bls12_381", "secp256k1"])
def test_wip_fold_round_seam(cname):
    """bpp_wip_fold_round == one round of the big-integer restatement's fold (wip.rs:147-164)."""
    need_gpu()
    import bulletproofsplus_amd as B
    cid = O.CURVE_IDS[cname]
    c = P.CURVES[cname]
    r = c["r"]
    G_ = P.make_group(cname, False)
    a = B.Arith.init(cname)
    n = 8
    pk = P.PublicKey(G_, n)
    rs = np.random.RandomState(4)
    av = [int(x) * 0x1234567 % r for x in rs.randint(1, 2**31, size=n)]
    bv_ = [int(x) * 0x7654321 % r for x in rs.randint(1, 2**31, size=n)]
    y, e = 12, 7
    y_nhat = pow(y, n // 2, r)
    e_inv, yi = pow(e, -1, r), pow(y_nhat, -1, r)
    h = n // 2
    exp_a = [(av[i] * e + av[h + i] * y_nhat * e_inv) % r for i in range(h)]
    exp_b = [(bv_[i] * e_inv + bv_[h + i] * e) % r for i in range(h)]
    exp_G = [G_.add(G_.mul(pk.G_vec[i], e_inv), G_.mul(pk.G_vec[h + i], yi * e % r)) for i in range(h)]
    exp_H = [G_.add(G_.mul(pk.H_vec[i], e), G_.mul(pk.H_vec[h + i], e_inv)) for i in range(h)]
    fa, fb, fG, fH = B.wip_fold_round(a, av, bv_, O.points_to_wire(cid, list(pk.G_vec)), O.points_to_wire(cid, list(pk.H_vec)),
                                      y_nhat, e)
    assert O.wire_to_scalars(fa) == exp_a and O.wire_to_scalars(fb) == exp_b
    assert O.wire_to_points(cid, fG) == exp_G and O.wire_to_points(cid, fH) == exp_H
    with pytest.raises(B.BppError):
        B.wip_fold_round(a, av[:6], bv_[:6], O.points_to_wire(cid, list(pk.G_vec[:6])), O.points_to_wire(cid, list(pk.H_vec[:6])),
                         y_nhat, e)


@pytest.mark.parametrize("count", [1, 5, 300])
def test_verifier_run_is_graph_capturable(count):
    """bpp_verifier_run only enqueues work (memsets, launches, and for a lone batch an event fork/join with its side
    stream): after one eager call it can be captured into a HIP graph; a replay gives the verdicts of the eager pass,
    including after the inputs in the captured buffers change."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    a = B.Arith.init("bls12_381")
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    recs, scs, _, _ = _prove_batch(bv, count, m, nbits=n)
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(scs.view(np.int64)).to(dev)
    d_ok = torch.full((count,), 7, dtype=torch.int32, device=dev)
    wsb = bv.workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()          # the uploads above ran on the default stream; `s` does not wait for it by itself
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb, s.cuda_stream)
        s.synchronize()
    assert d_ok.cpu().numpy().tolist() == [0] * count
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb,
                      torch.cuda.current_stream().cuda_stream)
    d_ok.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    assert d_ok.cpu().numpy().tolist() == [0] * count
    bad = scs.copy()
    bad[count // 2, 1, 0] ^= np.uint64(2)
    d_sc.copy_(torch.from_numpy(bad.view(np.int64)))
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert d_ok.cpu().numpy().tolist() == [1 if i == count // 2 else 0 for i in range(count)]
    del g
    bv.close()
