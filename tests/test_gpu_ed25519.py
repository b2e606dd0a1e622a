"""-m gpu: the engine's third instantiation, edwards25519 ("Ristretto-class", csrc/ed25519.hpp).

PARITY UNPINNED: the reference has no curve25519 / Ristretto backend, so there is no
reference code, test or vector for any result below.  The checker is the big-integer restatement of the
reference's protocol code run over an Edwards group backend (oracle/pyref.py: EdwardsGroup, RFC 8032 base
point) plus the group-independent dlog-shadow scalars."""

import random

import numpy as np
import pytest

import oracle as O
import pyref as P
from gpu_util import need_gpu, run_verifier_device, run_combined_device

pytestmark = pytest.mark.gpu

ED = O.ED25519


def words(x, n):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def test_ed25519_field_and_group_primitives():
    need_gpu()
    import bulletproofsplus_amd as B
    from bulletproofsplus_amd import _lib
    a = B.Arith.init("ed25519")
    c = P.ED25519
    rnd = random.Random(8)
    for field, mod in ((0, c["p"]), (1, c["r"])):
        vals = [0, 1, 2, mod - 1, mod - 2, (1 << 30) - 1, 1 << 30, (1 << 255) - 1, mod, mod + 18] + \
               [rnd.randrange(mod) for _ in range(246)]
        xs = [rnd.choice(vals) for _ in range(1024)]
        ys = [rnd.choice(vals) for _ in range(1024)]
        A = np.array([words(x, 8) for x in xs], dtype=np.uint32)
        Bv = np.array([words(y, 8) for y in ys], dtype=np.uint32)
        for op, fn in ((0, lambda x, y: x * y % mod), (1, lambda x, y: (x + y) % mod), (2, lambda x, y: (x - y) % mod),
                       (4, lambda x, y: x * x % mod), (5, lambda x, y: -x % mod)):
            out = np.zeros_like(A)
            assert _lib.lib().bpp_debug_field_op(a.handle, field, op, A.ctypes.data, Bv.ctypes.data, len(xs), out.ctypes.data) == 0
            got = [sum(int(w) << (32 * i) for i, w in enumerate(row)) for row in out]
            assert got == [fn(x % mod, y % mod) for x, y in zip(xs, ys)], (field, op)
        out = np.zeros_like(A[:32])
        assert _lib.lib().bpp_debug_field_op(a.handle, field, 3, A[:32].ctypes.data, Bv[:32].ctypes.data, 32, out.ctypes.data) == 0
        got = [sum(int(w) << (32 * i) for i, w in enumerate(row)) for row in out]
        assert got == [pow(x % mod, -1, mod) if x % mod else 0 for x in xs[:32]]
    G = P.EdwardsGroup(c)
    g = G.base()
    ks = [1, 2, 3, 5, 15, 15, G.r - 1, G.r - 15] + [rnd.randrange(1, G.r) for _ in range(12)]
    pts = [G.mul(g, k) for k in ks] + [None]
    pairs = [(p, q) for p in pts[:10] + [None] for q in pts[:10] + [None]] + list(zip(pts[8:], reversed(pts[8:])))
    Aw = O.points_to_wire(ED, [p for p, _ in pairs])
    Bw = O.points_to_wire(ED, [q for _, q in pairs])
    dbl = lambda p: G.add(p, p)
    for op, fn in {0: lambda p, q: G.add(p, q), 1: lambda p, q: G.add(p, q), 2: lambda p, q: dbl(p),
                   3: lambda p, q: G.add(dbl(p), q), 4: lambda p, q: G.add(dbl(p), dbl(q)),
                   5: lambda p, q: G.add(G.add(p, q), p)}.items():
        out = np.zeros_like(Aw)
        assert _lib.lib().bpp_debug_point_op(a.handle, op, Aw.ctypes.data, Bw.ctypes.data, len(pairs), out.ctypes.data) == 0
        assert O.wire_to_points(ED, out) == [fn(p, q) for p, q in pairs], op
    # scalar multiplication incl. k = 0, k = l (the group order) and the identity
    gw = O.point_to_wire(ED, g)
    ks2 = [0, 1, 2, G.r - 1, G.r, 0xDEADBEEF, rnd.randrange(G.r)]
    out = a.scalar_mul(ks2, np.stack([gw] * len(ks2)))
    assert O.wire_to_points(ED, out) == [G.mul(g, k % G.r) for k in ks2]
    # an off-curve point is rejected
    bad = gw.copy()
    bad[0] ^= 1
    with pytest.raises(B.BppError):
        B.msm_batch(a, [3], bad[None], [1])


def test_ed25519_pk_mulvec_pippenger():
    need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init("ed25519")
    G = P.EdwardsGroup(P.ED25519)
    pk = B.PublicKey.new(a, 40)
    ppk = P.PublicKey(G, 40)
    assert O.wire_to_points(ED, pk.G_vec) == ppk.G_vec and O.wire_to_points(ED, pk.H_vec) == ppk.H_vec
    assert O.wire_to_points(ED, pk.gh) == [ppk.g, ppk.h]
    pr = B.RangeProver.new()
    pr.commit(pk, 2**40 + 3, G.r - 5)
    assert O.wire_to_point(ED, pr.commitment_vec[0]) == ppk.commitment(P.Fr(G.r).new(P._i32(2**40 + 3)), G.r - 5)
    rnd = random.Random(4)
    base = np.concatenate([pk.gh, pk.G_vec, pk.H_vec, O.points_to_wire(ED, [None])])
    n = 200
    idx = [rnd.randrange(base.shape[0]) for _ in range(n)]
    scs = [rnd.randrange(G.r) for _ in range(n)]
    scs[:6] = [0, 1, G.r - 1, 5, G.r - 5, 2]
    idx[4] = idx[3]
    pts = base[idx]
    mv = P.MulVec(G)
    mv.add_scalars(scs)
    mv.add_points(O.wire_to_points(ED, pts))
    exp = O.point_to_wire(ED, mv.calculate())
    assert np.array_equal(B.msm_batch(a, scs, pts, [n])[0], exp)
    for c in (0, 3, 8, 13):
        assert np.array_equal(B.msm_pippenger(a, scs, pts, c), exp), c
    # G_4 = H_2 = 15 g collide; sum with opposite scalars is the identity
    mv2 = B.MulVec(a)
    mv2.add_scalars([7, G.r - 7])
    mv2.add_points([pk.G_vec[4], pk.H_vec[2]])
    assert a.is_zero(mv2.calculate())


@pytest.mark.parametrize("n,vals,gams,c", [(8, [200, 5], [3, 7], 5), (8, [77], [9], 4), (8, [300, 5], [3, 7], 6)])
def test_ed25519_prove_verify_match_bigint(n, vals, gams, c):
    torch = need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init("ed25519")
    m = len(vals)
    ppk, ppr, pproof = P.prove_case("ed25519", n, vals, gams, shadow=False)
    exp_pts = [pproof.A, pproof.proof.A, pproof.proof.B] + pproof.proof.L_vec + pproof.proof.R_vec
    exp_sc = [pproof.proof.r_prime, pproof.proof.s_prime, pproof.proof.d_prime]
    pmv = pproof.verify_mulvec(ppk, n, ppr.commitment_vec)
    exp_ok = pproof.verify(ppk, n, ppr.commitment_vec)
    pk = B.PublicKey.new(a, n * m)
    pr = B.RangeProver.new()
    for v, g in zip(vals, gams):
        pr.commit(pk, v, g)
    assert O.wire_to_points(ED, np.stack(pr.commitment_vec)) == ppr.commitment_vec
    proof = B.RangeProof.prove(pk, n, pr)                      # host-driven single-proof prover
    assert O.wire_to_points(ED, proof.points_wire()) == exp_pts
    assert O.wire_to_scalars(proof.scalars_wire()) == exp_sc
    if exp_ok:
        assert proof.verify(pk, n, pr.commitment_vec) is None
    else:
        with pytest.raises(B.VerificationError):
            proof.verify(pk, n, pr.commitment_vec)
    eng = B.BatchVerifier(pk, n, m, window_bits=c)
    bpts, bsc, bV = eng.prove_batch([vals, vals], [gams, gams])  # batched device prover
    assert np.array_equal(bpts[0], proof.points_wire()) and np.array_equal(bsc[1], proof.scalars_wire())
    rec = np.concatenate([bpts[0], bV[0]])
    bad = bsc[0].copy()
    bad[1, 0] ^= 1
    ok, got_sc, got_res = run_verifier_device(torch, eng, np.stack([rec, rec]), np.stack([bsc[0], bad]))
    assert ok.tolist() == [0 if exp_ok else 1, 1]
    assert O.wire_to_scalars(got_sc[0]) == pmv.scalars                      # reference MulVec order
    assert (O.wire_to_point(ED, got_res[0]) is None) == exp_ok
    assert O.wire_to_point(ED, got_res[0]) == pmv.calculate()
    assert run_combined_device(torch, eng, np.stack([rec, rec]), np.stack([bsc[0], bsc[0]]), 9)[0] == (0 if exp_ok else 1)
    assert run_combined_device(torch, eng, np.stack([rec, rec]), np.stack([bsc[0], bad]), 9)[0] == 1


def test_ed25519_full_size_round_trip_and_shadow_scalars():
    """(64,16): proof scalars == dlog-shadow known answers over the group order l; batch round trip."""
    need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init("ed25519")
    pk = B.PublicKey.new(a, 1024)
    eng = B.BatchVerifier(pk, 64, 16, window_bits=10)
    rnd = np.random.RandomState(2)
    vals = rnd.randint(0, 2**31 - 1, size=(48, 16)).astype(np.uint64)
    vals[0] = 31
    gams = [[7] * 16] + [[int(x) for x in row] for row in rnd.randint(1, 2**62, size=(47, 16))]
    pts, sc, V = eng.prove_batch(vals, gams)
    _, _, sproof = P.prove_case("ed25519", 64, [31] * 16, [7] * 16, shadow=True)
    assert O.wire_to_scalars(sc[0]) == [sproof.proof.r_prime, sproof.proof.s_prime, sproof.proof.d_prime]
    G = P.EdwardsGroup(P.ED25519)
    assert O.wire_to_point(ED, pts[0, 2]) == G.mul(G.base(), sproof.proof.B)      # point == dlog * g
    assert O.wire_to_point(ED, pts[0, 3]) == G.mul(G.base(), sproof.proof.L_vec[0])
    recs = np.concatenate([pts, V], axis=1)
    assert eng.verify_wire(recs, sc).tolist() == [0] * 48
    sc2 = sc.copy()
    sc2[17, 2, 0] ^= 4
    recs2 = recs.copy()
    recs2[30, [5, 15]] = recs2[30, [15, 5]]
    assert eng.verify_wire(recs2, sc2).tolist() == [1 if i in (17, 30) else 0 for i in range(48)]
