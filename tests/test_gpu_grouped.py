"""The grouped check (include/bpp_amd.h "grouped check", csrc/combined.hpp): per-proof verdicts from one weighted check
per group of neighbouring proofs plus an exact pass over the groups that fail.  An engine mode, not a reference path
(the reference verifies one proof at a time, src/range/mod.rs:57-78): the bar is the verdict vector of the exact per-proof
path (bpp_verifier_run), itself checked against the oracle elsewhere, on valid, tampered and malformed input."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import oracle as O  # noqa: E402
from gpu_util import need_gpu, run_verifier_device, run_grouped_device  # noqa: E402

pytestmark = pytest.mark.gpu
CID = O.CURVE_IDS


def _engine(B, cname, n, m, c):
    """-> (oracle key or None, engine).  The C oracle has no edwards25519: there the proofs come from the device prover."""
    cid = CID[cname]
    a = B.Arith.init(cid)
    if cname == "ed25519":
        return None, B.BatchVerifier(B.PublicKey.new(a, n * m), n, m, window_bits=c)
    opk = O.PublicKey(cid, n * m)
    pk = B.PublicKey.from_points(a, opk.gh, opk.G, opk.H)
    return opk, B.BatchVerifier(pk, n, m, window_bits=c)


def _proofs(bv, opk, n, vals, gams, distinct):
    vs = [[(v * (t + 3) + t) % (1 << n) for v in vals] for t in range(distinct)]
    gs = [[g + 5 * t for g in gams] for t in range(distinct)]
    if opk is None:
        pts, sc, V = bv.prove_batch(vs, gs)
        return np.concatenate([pts, V], axis=1), sc
    out = []
    for t in range(distinct):
        pts, sc, V = O.range_prove(opk, n, vs[t], gs[t])
        assert O.range_verify(opk, n, len(vals), pts, sc, V) == 0
        out.append((np.concatenate([pts, V]), sc))
    return np.stack([g[0] for g in out]), np.stack([g[1] for g in out])


@pytest.mark.parametrize("cname,n,vals,gams,c", [
    ("bls12_381", 8, [200, 5], [3, 7], 5),
    ("secp256k1", 8, [77], [9], 6),
    ("ed25519", 8, [1, 255], [4, 6], 5),
])
def test_grouped_verdicts_equal_the_exact_path(cname, n, vals, gams, c):
    torch = need_gpu()
    import bulletproofsplus_amd as B
    opk, bv = _engine(B, cname, n, len(vals), c)
    base_r, base_s = _proofs(bv, opk, n, vals, gams, 7)
    count = 21                                    # ragged: the last group of 4 / 8 / 16 is short
    recs = np.stack([base_r[i % 7] for i in range(count)])
    scs = np.stack([base_s[i % 7] for i in range(count)])
    for group in (2, 4, 8, 16, 32):
        ok, failed, redone = run_grouped_device(torch, bv, recs, scs, group, seed=group)
        assert ok.tolist() == [0] * count and (failed, redone) == (0, 0)
    cases = [[0], [20], [5, 6], [3, 4], [0, 7, 8, 19, 20], list(range(count))]
    for victims in cases:
        bad = scs.copy()
        for t, v in enumerate(victims):
            bad[v, t % 3, 0] ^= np.uint64(1 + t)   # r', s' or delta'
        exact, _, _ = run_verifier_device(torch, bv, recs, bad, want_scalars=False, want_result=False)
        assert exact.tolist() == [1 if i in victims else 0 for i in range(count)]
        for group in (2, 4, 16):
            ok, failed, redone = run_grouped_device(torch, bv, recs, bad, group, seed=7 * group + len(victims))
            assert ok.tolist() == exact.tolist(), (victims, group)
            groups_hit = {v // group for v in victims}
            assert failed == len(groups_hit)
            assert redone == sum(min(count, (g + 1) * group) - g * group for g in groups_hit)
    # a proof point swapped for another, and a point that is not on the curve (an invalid point fails its group whatever
    # the weights)
    r2 = recs.copy()
    r2[9, 3] = recs[9, 4]
    r2[17, 0, 0] ^= np.uint64(1)
    exact, _, _ = run_verifier_device(torch, bv, r2, scs, want_scalars=False, want_result=False)
    assert exact.tolist() == [1 if i in (9, 17) else 0 for i in range(count)]
    for group in (4, 8):
        ok, failed, redone = run_grouped_device(torch, bv, r2, scs, group, seed=3)
        assert ok.tolist() == exact.tolist() and failed == 2
    # an empty batch, and bad group sizes
    ok, failed, redone = run_grouped_device(torch, bv, recs[:0], scs[:0], 4)
    assert ok.tolist() == [] and (failed, redone) == (0, 0)
    for group in (0, 1, 3, 12):
        with pytest.raises(Exception):
            run_grouped_device(torch, bv, recs, scs, group)
    bv.close()


def test_grouped_every_horner_form_and_exact_slices():
    """More groups than the tree Horner serves (G > 256: eight lanes per group; then one lane per group), and more failing
    proofs than one slice of the exact pass (2 048): the verdicts still equal the exact path's."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, vals, gams = 8, [200], [3]
    opk, bv = _engine(B, "bls12_381", n, 1, 5)
    base_r, base_s = _proofs(bv, opk, n, vals, gams, 5)
    rng = np.random.RandomState(5)
    for count, group, nbad in ((1200, 2, 9), (20480, 2, 40), (6000, 2, 2500)):
        idx = rng.randint(0, 5, size=count)
        recs, scs = base_r[idx], base_s[idx].copy()
        victims = sorted(rng.choice(count, size=nbad, replace=False).tolist())
        for v in victims:
            scs[v, 2, 0] ^= np.uint64(2)
        exact, _, _ = run_verifier_device(torch, bv, recs, scs, want_scalars=False, want_result=False)
        assert int(exact.sum()) == nbad and all(exact[v] == 1 for v in victims)
        ok, failed, redone = run_grouped_device(torch, bv, recs, scs, group, seed=count)
        assert np.array_equal(ok, exact)
        assert failed == len({v // group for v in victims}) and redone == failed * group
    bv.close()


def test_grouped_under_transcript_challenges():
    """d_challenges (Fiat-Shamir mode): proofs made by the batched prover under the transcript, challenges derived on the
    device, verdicts as the exact path's."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m = 8, 2
    opk, bv = _engine(B, "secp256k1", n, m, 5)
    count = 12
    vals = [[(17 * i + j) % 256 for j in range(m)] for i in range(count)]
    gams = [[5 + i + j for j in range(m)] for i in range(count)]
    pts, sc, V = bv.prove_batch(vals, gams, transcript=True)
    recs = np.concatenate([pts, V], axis=1)
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(np.ascontiguousarray(recs).view(np.int64)).to(dev)
    d_ch = torch.zeros((count, 3 + bv.k, 4), dtype=torch.int64, device=dev)
    bv.derive_challenges_device(d_pts.data_ptr(), count, d_ch.data_ptr())
    torch.cuda.synchronize()
    ch = d_ch.cpu().numpy().view(np.uint64)
    bad = sc.copy()
    bad[7, 0, 1] ^= np.uint64(4)
    for scalars, expect in ((sc, [0] * count), (bad, [1 if i == 7 else 0 for i in range(count)])):
        exact, _, _ = run_verifier_device(torch, bv, recs, scalars, want_scalars=False, want_result=False, challenges=ch)
        assert exact.tolist() == expect
        ok, failed, redone = run_grouped_device(torch, bv, recs, scalars, 4, challenges=ch)
        assert ok.tolist() == expect and failed == sum(expect)
    # without the challenges the transcript's proofs fail under the reference's constants: every group fails, every proof
    # is re-verified, every verdict is 1
    ok, failed, redone = run_grouped_device(torch, bv, recs, sc, 4)
    assert ok.tolist() == [1] * count and (failed, redone) == (3, 12)
    bv.close()


@pytest.mark.parametrize("cname,transcript", [("bls12_381", False), ("bls12_381", True), ("ed25519", True)])
def test_grouped_behind_the_decoder(cname, transcript):
    """bpp_range_verify_batch_serialized_grouped_device: containers + compressed commitments in, status words 0 / 1 / 2 out,
    equal to the per-proof serialized path's on valid, tampered and malformed input."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m, count = 8, 2, 19
    a = B.Arith.init(cname)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    vals = [[(13 * i + j) % 256 for j in range(m)] for i in range(count)]
    gams = [[2 + i + 3 * j for j in range(m)] for i in range(count)]
    pts, scs, V = bv.prove_batch(vals, gams, transcript=transcript)
    blobs = B.encode_proofs(a, n, m, pts, scs)
    comm = B.compress_points(a, V.reshape(-1, a.PW)).reshape(count, m, -1)
    dev = torch.device("cuda:0")
    d_cm = torch.from_numpy(np.ascontiguousarray(comm)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    cb = B.compressed_bytes(a)
    sc0 = blobs.shape[1] - 96

    def both(bl, group):
        exact = bv.verify_serialized(bl, comm, transcript=transcript)
        d_bl = torch.from_numpy(np.ascontiguousarray(bl)).to(dev)
        d_ok = torch.full((count,), 7, dtype=torch.int32, device=dev)
        wsb = bv.serialized_grouped_workspace_bytes(count, group)
        d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        stats = bv.verify_serialized_grouped_device(d_bl.data_ptr(), d_cm.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb,
                                                    None, 0, group, stream, transcript=transcript)
        torch.cuda.synchronize()
        return exact.tolist(), d_ok.cpu().numpy().astype(np.uint32).tolist(), stats

    for group in (4, 8):
        exact, got, stats = both(blobs, group)
        assert exact == [0] * count and got == exact and stats == (0, 0)
    bad = blobs.copy()
    bad[1, sc0 + 3] ^= 4             # r': VerificationError
    bad[6, 4] = 9                    # version byte: FormatError
    bad[7, sc0 + 64 + 1] ^= 1        # delta'
    bad[12, 12 + 2 * cb + 5] ^= 0x10  # wip.B: another point or no point at all
    bad[18, sc0 + 32] ^= 1           # s' of the last proof (short group)
    for group in (2, 4, 16):
        exact, got, stats = both(bad, group)
        assert got == exact, (group, exact, got)
        assert exact[1] == 1 and exact[6] == 2 and exact[7] == 1 and exact[12] in (1, 2) and exact[18] == 1
        assert sum(1 for x in exact if x) == 5
        assert 1 <= stats[0] <= 5
    bv.close()


def test_grouped_exact_pass_fits_whatever_the_failing_count():
    """A smaller exact pass can need a LARGER workspace than a full slice (more blocks per proof for fewer proofs): the
    grouped workspace must hold the worst count.  (64,1) x 4096 with 47 failing groups = 1 504 proofs re-verified hit it."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    n, m, count, group = 64, 1, 4096, 32
    a = B.Arith.init("bls12_381")
    bv = B.BatchVerifier(B.PublicKey.new(a, n * m), n, m, window_bits=6)
    vals = [[(0x1234567 * (i + 1)) % (1 << 31)] for i in range(8)]   # below 2^31: the reference's `v as i32` (range/mod.rs)
    gams = [[3 + i] for i in range(8)]
    pts, scs, V = bv.prove_batch(vals, gams)
    recs8 = np.concatenate([pts, V], axis=1)
    idx = np.arange(count) % 8
    recs, sc = np.ascontiguousarray(recs8[idx]), np.ascontiguousarray(scs[idx])
    for ngroups in (47, 1, 64):
        bad = sc.copy()
        victims = [g * group + (7 * g) % group for g in range(0, 2 * ngroups, 2)]
        for v in victims:
            bad[v, 1, 0] ^= np.uint64(1)
        ok, failed, redone = run_grouped_device(torch, bv, recs, bad, group, seed=ngroups)
        want = np.zeros(count, dtype=np.uint32)
        want[victims] = 1
        assert np.array_equal(ok, want) and (failed, redone) == (ngroups, ngroups * group)
    bv.close()


def test_grouped_begin_finish_two_batches_in_flight():
    """bpp_verifier_grouped_begin / _finish: two batches in flight from one host thread (a stream, a workspace and a verdict
    buffer each), one of them tampered; and the lone-batch path of the exact pass from two host threads at once."""
    torch = need_gpu()
    import threading
    import hashlib
    import bulletproofsplus_amd as B
    n, vals, gams = 8, [200, 5], [3, 7]
    opk, bv = _engine(B, "bls12_381", n, 2, 5)
    base_r, base_s = _proofs(bv, opk, n, vals, gams, 6)
    count, group = 40, 4
    recs = np.ascontiguousarray(base_r[np.arange(count) % 6])
    good = np.ascontiguousarray(base_s[np.arange(count) % 6])
    bad = good.copy()
    for v in (3, 17, 18, 39):
        bad[v, 2, 0] ^= np.uint64(8)
    want_bad = [1 if i in (3, 17, 18, 39) else 0 for i in range(count)]
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = [torch.from_numpy(x.view(np.int64)).to(dev) for x in (good, bad)]
    wsb = bv.grouped_workspace_bytes(count, group)
    streams = [torch.cuda.Stream() for _ in range(2)]
    wss = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]
    oks = [torch.full((count,), 7, dtype=torch.int32, device=dev) for _ in range(2)]
    key = hashlib.sha256(b"two in flight").digest()
    torch.cuda.synchronize()
    stats = [None, None]
    for rep in range(3):
        for q in range(2):
            bv.grouped_begin_device(d_pts.data_ptr(), d_sc[q].data_ptr(), count, key, 0, oks[q].data_ptr(), wss[q].data_ptr(), wsb,
                                    group=group, stream=streams[q].cuda_stream)
        for q in range(2):
            stats[q] = bv.grouped_finish_device(d_pts.data_ptr(), d_sc[q].data_ptr(), count, oks[q].data_ptr(), wss[q].data_ptr(),
                                                wsb, group=group, stream=streams[q].cuda_stream)
        torch.cuda.synchronize()
        assert oks[0].cpu().numpy().tolist() == [0] * count and stats[0] == (0, 0)
        assert oks[1].cpu().numpy().tolist() == want_bad and stats[1] == (3, 12)
    # two host threads, each running whole grouped checks whose exact pass is a lone batch (side stream, shared events)
    errs = []

    def worker(q):
        try:
            for _ in range(6):
                bv.run_grouped_device(d_pts.data_ptr(), d_sc[1].data_ptr(), count, key, 0, oks[q].data_ptr(), wss[q].data_ptr(), wsb,
                                      group=group, stream=streams[q].cuda_stream)
                streams[q].synchronize()
                if oks[q].cpu().numpy().tolist() != want_bad:
                    errs.append(q)
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(q,)) for q in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errs == []
    bv.close()
