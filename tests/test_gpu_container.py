"""-m gpu: serialized proofs (the container), ristretto255 and the G1 subgroup check on the device against their
restatements in oracle/pyref.py.  No reference counterpart (no serialization, no Ristretto in the reference): parity
unpinned; pinned by the standard generator encodings and the restatements."""

import hashlib

import numpy as np
import pytest

import oracle as O
import pyref as P
from gpu_util import need_gpu

pytestmark = pytest.mark.gpu


def test_ristretto255_codec_on_device():
    need_gpu()
    import bulletproofsplus_amd as B
    R = P.Ristretto255
    G = P.EdwardsGroup(P.ED25519)
    a = B.Arith.init("ed25519")
    assert B.compressed_bytes(a) == 32
    Bp = G.base()
    ks = [0, 1, 2, 3, 7, 1000003, P.ED25519["r"] - 1]
    pts = [G.mul(Bp, k) if k else None for k in ks]
    T4 = (R.SQRT_M1, 0)
    pts += [G.add(pts[3], T4), G.add(pts[4], (0, R.P - 1))]          # other representatives of 3B and 7B
    wire = O.points_to_wire(O.ED25519, pts)
    enc = B.compress_points(a, wire)
    assert bytes(enc[0]) == bytes(32)
    assert bytes(enc[1]).hex() == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"     # RFC 9496 A.1
    assert bytes(enc[2]).hex() == "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"
    for i, pt in enumerate(pts):
        assert bytes(enc[i]) == R.encode(pt)
    assert bytes(enc[7]) == bytes(enc[3]) and bytes(enc[8]) == bytes(enc[4])
    dec, ok = B.decompress_points(a, enc)
    assert ok.tolist() == [0] * len(pts)
    for i, pt in enumerate(pts):
        got = O.wire_to_point(O.ED25519, dec[i])
        exp = R.decode(R.encode(pt))
        assert (got if got is not None else (0, 1)) == exp
    bad = np.stack([np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8),            # negative s
                    np.frombuffer(R.P.to_bytes(32, "little"), dtype=np.uint8),           # non-canonical
                    np.frombuffer(hashlib.sha256(b"x").digest(), dtype=np.uint8)])
    _, okb = B.decompress_points(a, bad)
    assert okb.tolist()[:2] == [1, 1] and okb.tolist()[2] == (0 if R.decode(hashlib.sha256(b"x").digest()) else 1)


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("secp256k1", 1), ("ed25519", 2)])
def test_container_encode_decode_verify(cname, cid):
    need_gpu()
    import bulletproofsplus_amd as B
    c = P.CURVES[cname]
    G = P.make_group(cname, False)
    n, m = 4, 2
    a = B.Arith.init(cname)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=4)
    vals = [[9, 3], [15, 0], [1, 2]]
    gams = [[5, 6], [7, 8], [9, 10]]
    pts, scs, V = bv.prove_batch(vals, gams)
    blobs = B.encode_proofs(a, n, m, pts, scs)
    assert blobs.shape == (3, B.proof_bytes(a, n, m))
    # the device encoder == the restatement's, proof by proof
    for i in range(3):
        _, prover, proof = P.prove_case(cname, n, vals[i], gams[i], shadow=False)
        assert bytes(blobs[i]) == P.encode_proof(c, n, m, proof)
        assert P.decode_proof(c, G, n, m, bytes(blobs[i])) is not None
    dp, ds, st = B.decode_proofs(a, n, m, blobs)
    assert st.tolist() == [0, 0, 0] and np.array_equal(ds, scs)
    if cname != "ed25519":
        assert np.array_equal(dp, pts)
    comm = B.compress_points(a, V.reshape(-1, a.PW)).reshape(3, m, -1)
    assert bv.verify_serialized(blobs, comm).tolist() == [0, 0, 0]
    # rejections: header, scalar range, point encoding, and a tampered scalar (VerificationError, not FormatError)
    bad = blobs.copy()
    bad[0, 4] = 2                                                            # version
    r_le = np.frombuffer(c["r"].to_bytes(32, "little"), dtype=np.uint8)
    bad[1, -32:] = r_le                                                      # delta' = r: not canonical
    bad[2, -64] ^= 1                                                         # s' off by one: parses, fails the MulVec
    assert bv.verify_serialized(bad, comm).tolist() == [2, 2, 1]
    _, _, st = B.decode_proofs(a, n, m, bad)
    assert st.tolist() == [2, 2, 0]
    bad = blobs.copy()
    cb = B.compressed_bytes(a)
    if cname == "bls12_381":
        bad[0, 12:12 + cb] = np.frombuffer(P.compress_point(c, (0, 2)), dtype=np.uint8)   # on the curve, order 3: outside G1
        bad[1, 12] &= 0x7F                                                                # "uncompressed" flag
    elif cname == "secp256k1":
        bad[0, 12] = 4
        bad[1, 13:13 + 32] = 0xFF                                                         # x >= p
    else:
        bad[0, 12] |= 1                                                                   # negative s
        bad[1, 12:12 + cb] = np.frombuffer(R_P_BYTES, dtype=np.uint8)
    assert bv.verify_serialized(bad, comm).tolist() == [2, 2, 0]
    # a commitment that does not parse is a FormatError of its proof
    cbad = comm.copy()
    cbad[2, 1] = 0xFF
    assert bv.verify_serialized(blobs, cbad).tolist() == [0, 0, 2]
    bv.close()


R_P_BYTES = ((1 << 255) - 19).to_bytes(32, "little")


def test_bls12_381_subgroup_check_on_device():
    """decode with the subgroup check: G1 points pass; curve points outside G1 (random, pure torsion, G1 + torsion) are
    rejected exactly when [r] P != O."""
    need_gpu()
    import random
    import bulletproofsplus_amd as B
    c = P.BLS12_381
    p, r = c["p"], c["r"]
    G = P.WeierstrassGroup(c)
    a = B.Arith.init("bls12_381")
    rng = random.Random(9)
    g = G.base()
    pts, exp = [], []
    for _ in range(4):
        while True:
            x = rng.randrange(p)
            rhs = (x ** 3 + 4) % p
            y = pow(rhs, (p + 1) // 4, p)
            if y * y % p == rhs:
                break
        Pt = (x, y)
        T = G.mul(Pt, r)
        for cand in (Pt, T, G.add(G.mul(g, rng.randrange(r)), T), G.mul(g, rng.randrange(r))):
            if cand is not None:
                pts.append(cand)
                exp.append(0 if G.is_zero(G.mul(cand, r)) else 1)
    pts.append((0, 2))
    exp.append(1)
    assert 0 in exp and 1 in exp
    # through the container decoder (the path that applies the check): put each point in the A slot of a valid proof
    n, m = 4, 2
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=4)
    ppts, pscs, _ = bv.prove_batch([[9, 3]], [[5, 6]])
    blob = B.encode_proofs(a, n, m, ppts, pscs)[0]
    blobs = np.stack([blob] * len(pts))
    for i, pt in enumerate(pts):
        blobs[i, 12:60] = np.frombuffer(P.compress_point(c, pt), dtype=np.uint8)
    _, _, st = B.decode_proofs(a, n, m, blobs)
    assert st.tolist() == [2 * e for e in exp]
    # the plain codec (no subgroup check) accepts every curve point
    _, ok = B.decompress_points(a, np.stack([np.frombuffer(P.compress_point(c, pt), dtype=np.uint8) for pt in pts]))
    assert ok.tolist() == [0] * len(pts)
    bv.close()


@pytest.mark.parametrize("cname", ["bls12_381", "secp256k1", "ed25519"])
def test_hashed_generators(cname):
    """PublicKey.hashed: == the restatement; on the curve, in the prime-order group, pairwise distinct; label-separated;
    and a range proof over them proves and verifies (constants and transcript mode)."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    c = P.CURVES[cname]
    cid = O.CURVE_IDS[cname]
    G = P.make_group(cname, False)
    a = B.Arith.init(cname)
    n, m = 4, 2
    pk = B.PublicKey.hashed(a, n * m, b"test generators")
    assert O.wire_to_point(cid, pk.gh[0]) == G.base()
    exp = [P.hash_to_group(c, G, b"test generators", "h", 0)] + \
          [P.hash_to_group(c, G, b"test generators", "G", i) for i in range(n * m)] + \
          [P.hash_to_group(c, G, b"test generators", "H", i) for i in range(n * m)]
    got = [O.wire_to_point(cid, w) for w in [pk.gh[1]] + list(pk.G_vec) + list(pk.H_vec)]
    assert got == exp
    assert len(set(got)) == len(got) and all(G.on_curve(q) for q in got)
    for q in got[:3]:
        L = G.mul(q, c["r"])
        assert L is None or (cname == "ed25519" and (L[0] == 0 or L[1] == 0))
    pk2 = B.PublicKey.hashed(a, n * m, b"another label")
    assert not np.array_equal(pk2.G_vec, pk.G_vec)
    bv = B.BatchVerifier(pk, n, m, window_bits=4)
    for fs in (False, True):
        pts, scs, V = bv.prove_batch([[9, 3], [15, 1]], [[5, 6], [7, 8]], transcript=fs)
        blobs = B.encode_proofs(a, n, m, pts, scs)
        comm = B.compress_points(a, V.reshape(-1, a.PW)).reshape(2, m, -1)
        assert bv.verify_serialized(blobs, comm, transcript=fs).tolist() == [0, 0]
        assert bv.verify_serialized(blobs, comm, transcript=not fs).tolist() == [1, 1]
    bv.close()


@pytest.mark.parametrize("cname", ["bls12_381", "secp256k1", "ed25519"])
def test_serialized_device_path(cname):
    """bpp_range_verify_batch_serialized_device: containers and compressed commitments resident in HBM -> status words,
    no host round trip.  70 proofs (a ragged last wave of the decoder), rejections of every kind at known indices, both
    challenge modes; the host-pointer entry point must give the same vector."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    c = P.CURVES[cname]
    a = B.Arith.init(cname)
    n, m, cnt = 8, 2, 70
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    vals = [[(7 * i + 1) % 256, (3 * i) % 256] for i in range(cnt)]
    gams = [[i + 1, 2 * i + 5] for i in range(cnt)]
    dev = torch.device("cuda:0")
    for fs in (False, True):
        pts, scs, V = bv.prove_batch(vals, gams, transcript=fs)
        blobs = B.encode_proofs(a, n, m, pts, scs)
        comm = B.compress_points(a, V.reshape(-1, a.PW)).reshape(cnt, m, -1)
        exp = [0] * cnt
        blobs[3, 0] = ord("X")                   # magic
        exp[3] = 2
        blobs[17, 9] = 1                         # reserved byte
        exp[17] = 2
        blobs[40, -96:-64] = np.frombuffer(c["r"].to_bytes(32, "little"), dtype=np.uint8)   # r' = group order
        exp[40] = 2
        blobs[41, -96] ^= 1                      # r' off by one: parses, fails the MulVec
        exp[41] = 1
        blobs[69, 12 + B.compressed_bytes(a)] ^= 0x55 if cname != "secp256k1" else 0x04   # wip.A: garbage encoding
        comm[5, 1], comm[6, 1] = comm[6, 1].copy(), comm[5, 1].copy()   # valid points, wrong proofs
        exp[5] = exp[6] = 1
        d_pr = torch.from_numpy(blobs).to(dev)
        d_cm = torch.from_numpy(comm.copy()).to(dev)
        d_ok = torch.full((cnt,), 9, dtype=torch.int32, device=dev)
        wsb = bv.serialized_workspace_bytes(cnt)
        d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        bv.verify_serialized_device(d_pr.data_ptr(), d_cm.data_ptr(), cnt, d_ok.data_ptr(), d_ws.data_ptr(), wsb,
                                    transcript=fs)
        torch.cuda.synchronize()
        got = d_ok.cpu().numpy().tolist()
        host = bv.verify_serialized(blobs, comm, transcript=fs).tolist()
        assert got == host
        # index 69: a flipped byte of an encoding is rejected by the decoder or, if it happens to decode, by the MulVec
        assert got[69] in (1, 2)
        exp[69] = got[69]
        assert got == exp
        with pytest.raises(B.BppError):
            bv.verify_serialized_device(d_pr.data_ptr(), d_cm.data_ptr(), cnt, d_ok.data_ptr(), d_ws.data_ptr(), wsb - 1)
    bv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("secp256k1", 1)])
def test_container_version_2_on_device(cname, cid):
    """Container version 2 (uncompressed points): the engine's encoder == the restatement's, the device decode + verify gives
    the statuses of the restatement for valid proofs and for every kind of rejection, in both challenge modes; the same
    proofs through version 1 give the same verdicts."""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    c = P.CURVES[cname]
    G = P.make_group(cname, False)
    n, m, cnt = 4, 2, 9
    a = B.Arith.init(cname)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=4)
    vals = [[(5 * i + 1) % 16, (3 * i) % 16] for i in range(cnt)]
    gams = [[i + 1, 2 * i + 5] for i in range(cnt)]
    ub = B.uncompressed_bytes(a)
    assert ub == {"bls12_381": 96, "secp256k1": 65}[cname]
    dev = torch.device("cuda:0")
    for fs in (False, True):
        pts, scs, V = bv.prove_batch(vals, gams, transcript=fs)
        blobs = B.encode_proofs(a, n, m, pts, scs, version=2)
        assert blobs.shape == (cnt, B.proof_bytes(a, n, m, 2)) and B.proof_bytes(a, n, m, 2) == 12 + 9 * ub + 96
        comm = B.uncompressed_points(a, V.reshape(-1, a.PW)).reshape(cnt, m, ub)
        if not fs:
            for i in range(3):
                _, prover, proof = P.prove_case(cname, n, vals[i], gams[i], shadow=False)
                assert bytes(blobs[i]) == P.encode_proof(c, n, m, proof, version=2)
                assert bytes(comm[i, 0]) == P.uncompressed_point(c, prover.commitment_vec[0])
        assert bv.verify_serialized(blobs, comm, transcript=fs, uncompressed=True).tolist() == [0] * cnt
        # version 1 of the same proofs: the same verdicts
        b1 = B.encode_proofs(a, n, m, pts, scs)
        c1 = B.compress_points(a, V.reshape(-1, a.PW)).reshape(cnt, m, -1)
        assert bv.verify_serialized(b1, c1, transcript=fs).tolist() == [0] * cnt
        bad = blobs.copy()
        exp = [0] * cnt
        bad[0, 4] = 1                                   # version byte says 1
        exp[0] = 2
        bad[1, 12 + ub - 1] ^= 1                        # y of A off by one: off the curve
        exp[1] = 2
        bad[2, 12 + ub] = (bad[2, 12 + ub] | 0x80) if cname == "bls12_381" else 0x02   # wip.A: compressed flag / prefix
        exp[2] = 2
        bad[3, -96] ^= 1                                # r' off by one: parses, fails the MulVec
        exp[3] = 1
        bad[4, -32:] = np.frombuffer(c["r"].to_bytes(32, "little"), dtype=np.uint8)      # delta' = r
        exp[4] = 2
        if cname == "bls12_381":
            bad[5, 12:12 + ub] = np.frombuffer(P.uncompressed_point(c, (0, 2)), dtype=np.uint8)   # outside G1
            exp[5] = 2
        cbad = comm.copy()
        cbad[6, 1, -1] ^= 1                             # a commitment off the curve
        exp[6] = 2
        for i in range(cnt):
            dec = P.decode_proof(c, G, n, m, bytes(bad[i]), version=2)
            assert (dec is None) == (exp[i] == 2 and i != 6), i
        assert bv.verify_serialized(bad, cbad, transcript=fs, uncompressed=True).tolist() == exp
        d_pr = torch.from_numpy(bad).to(dev)
        d_cm = torch.from_numpy(cbad).to(dev)
        d_ok = torch.full((cnt,), 9, dtype=torch.int32, device=dev)
        wsb = bv.serialized_workspace_bytes(cnt)
        d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        bv.verify_serialized_device(d_pr.data_ptr(), d_cm.data_ptr(), cnt, d_ok.data_ptr(), d_ws.data_ptr(), wsb, transcript=fs,
                                    uncompressed=True)
        torch.cuda.synchronize()
        assert d_ok.cpu().numpy().tolist() == exp
    # not offered for ristretto255
    a_e = B.Arith.init("ed25519")
    assert B.uncompressed_bytes(a_e) == 0 and B.proof_bytes(a_e, n, m, 2) == 0
    bv.close()


@pytest.mark.parametrize("cname,cid", [("bls12_381", 0), ("secp256k1", 1), ("ed25519", 2)])
def test_container_mutations_statuses_match_restatement(cname, cid):
    """Random byte mutations of valid containers (1-3 bytes each, anywhere: header, point encodings, scalars): the device
    decoder's FormatError verdict equals the restatement's (pyref.decode_proof is None), container by container; a mutated
    container that still parses fails verification (status 1); nothing else changes its neighbours' statuses."""
    need_gpu()
    import random
    import bulletproofsplus_amd as B
    c = P.CURVES[cname]
    G = P.make_group(cname, False)
    n, m, count = 4, 2, 160
    a = B.Arith.init(cname)
    bv = B.BatchVerifier(B.PublicKey.new(a, n * m), n, m, window_bits=4)
    base = 8
    vals = [[(5 * i + j) % 16 for j in range(m)] for i in range(base)]
    gams = [[2 + i + j for j in range(m)] for i in range(base)]
    pts, scs, V = bv.prove_batch(vals, gams)
    blobs8 = B.encode_proofs(a, n, m, pts, scs)
    comm8 = B.compress_points(a, V.reshape(-1, a.PW)).reshape(base, m, -1)
    idx = np.arange(count) % base
    blobs, comm = np.ascontiguousarray(blobs8[idx]), np.ascontiguousarray(comm8[idx])
    assert bv.verify_serialized(blobs, comm).tolist() == [0] * count
    rng = random.Random(1234 + cid)
    mutated = blobs.copy()
    touched = set()
    for i in range(count):
        if i % 4 == 3:
            continue                       # every fourth container stays valid
        touched.add(i)
        for _ in range(rng.randint(1, 3)):
            pos = rng.randrange(blobs.shape[1])
            mutated[i, pos] ^= 1 << rng.randrange(8)
        if np.array_equal(mutated[i], blobs[i]):          # two flips cancelled
            mutated[i, -1] ^= 1
    got = bv.verify_serialized(mutated, comm).tolist()
    n_format = 0
    for i in range(count):
        if i not in touched:
            assert got[i] == 0, i
            continue
        parsed = P.decode_proof(c, G, n, m, bytes(mutated[i]))
        if parsed is None:
            n_format += 1
            assert got[i] == 2, (i, got[i])
        else:
            assert got[i] == 1, (i, got[i])
    assert 10 < n_format < len(touched)        # both kinds occurred
    bv.close()
