"""CPU: the checker side of the serialized-proof path -- the container restatement (pyref.encode_proof / decode_proof),
ristretto255 (pyref.Ristretto255) and the BLS12-381 G1 subgroup test the engine uses (endomorphism shortcut) against
its definition ([r] P == O).  None of this has a reference counterpart (no serialization, no Ristretto: SURVEY.md facts
1 and 3): parity unpinned; the public vectors used are the standard generator encodings."""

import hashlib
import random

import pyref as P


def test_ristretto255_standard_vectors_and_laws():
    R = P.Ristretto255
    G = P.EdwardsGroup(P.ED25519)
    B = G.base()
    # RFC 9496 A.1: the encodings of 0 B, 1 B, 2 B
    assert R.encode(None) == bytes(32)
    assert R.encode(B).hex() == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
    assert R.encode(G.mul(B, 2)).hex() == "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"
    p = R.P
    T2, T4 = (0, p - 1), (R.SQRT_M1, 0)
    for k in (3, 8, 1000003):
        Pk = G.mul(B, k)
        e = R.encode(Pk)
        for T in (T2, T4, G.neg(T4)):                     # the encoding does not see the 4-torsion
            assert R.encode(G.add(Pk, T)) == e
        q = R.decode(e)
        assert q is not None and R.equal(q, Pk) and R.encode(q) == e and G.on_curve(q)
    assert R.decode((1).to_bytes(32, "little")) is None   # negative s
    assert R.decode(p.to_bytes(32, "little")) is None     # non-canonical field element
    for j in range(3):                                    # element derivation: on the curve, in the even subgroup
        Q = R.from_uniform_bytes(hashlib.sha512(b"bpp %d" % j).digest(), G)
        assert G.on_curve(Q) and R.equal(R.decode(R.encode(Q)), Q)
        L = G.mul(Q, P.ED25519["r"])
        assert L is None or L[0] == 0 or L[1] == 0


def test_bls12_381_subgroup_shortcut_equals_definition():
    c = P.BLS12_381
    p, r = c["p"], c["r"]
    G = P.WeierstrassGroup(c)
    z = 0xd201000000010000
    h = (z + 1) ** 2 // 3
    beta = 0x5f19672fdf76ce51ba69c6076a0f77eaddb3a93be6f89688de17d813620a00022e01fffffffefffe
    assert pow(beta, 3, p) == 1 and beta != 1 and r == z ** 4 - z ** 2 + 1

    def shortcut(Pt):       # phi(P) + [z^2] P == O  (csrc/ec.hpp aff_in_prime_subgroup)
        if Pt is None:
            return True
        return G.is_zero(G.add((beta * Pt[0] % p, Pt[1]), G.mul(G.mul(Pt, z), z)))

    rng = random.Random(5)
    g = G.base()
    assert shortcut(g) and shortcut(G.mul(g, 0x1234567890abcdef1234567890abcdef))
    assert not shortcut((0, 2))                            # the order-3 point of the GPU tests
    for _ in range(6):
        while True:
            x = rng.randrange(p)
            rhs = (x ** 3 + 4) % p
            y = pow(rhs, (p + 1) // 4, p)
            if y * y % p == rhs:
                break
        Pt = (x, y)
        assert shortcut(Pt) == G.is_zero(G.mul(Pt, r))
        T = G.mul(Pt, r)                                   # pure torsion
        assert G.is_zero(T) or not shortcut(T)
        M = G.add(G.mul(g, rng.randrange(r)), T)           # G1 + torsion
        assert shortcut(M) == G.is_zero(G.mul(M, r))
        for q in (3, 11, 10177, 859267, 52437899):         # every prime that divides the cofactor
            assert h % q == 0
            Tq = G.mul(Pt, r * (h // q))
            assert G.is_zero(Tq) or not shortcut(Tq)


def test_container_round_trip_and_rejections():
    for cname in ("secp256k1", "bls12_381", "ed25519"):
        c = P.CURVES[cname]
        G = P.make_group(cname, False)
        n, vals = 4, [9, 3]
        pk, prover, proof = P.prove_case(cname, n, vals, [5, 6], shadow=False)
        blob = P.encode_proof(c, n, 2, proof)
        k = 3
        cb = {"bls12_381": 48, "secp256k1": 33, "ed25519": 32}[cname]
        assert len(blob) == 12 + (3 + 2 * k) * cb + 96
        back = P.decode_proof(c, G, n, 2, blob)
        assert back is not None and back.verify(pk, n, prover.commitment_vec)
        if cname != "ed25519":
            assert back.A == proof.A and back.proof.L_vec == proof.proof.L_vec
        for pos, val in ((0, 0x43), (4, 2), (5, 7), (6, 5), (9, 1)):          # magic, version, curve, n, reserved
            bad = bytearray(blob)
            bad[pos] = val
            assert P.decode_proof(c, G, n, 2, bytes(bad)) is None
        bad = bytearray(blob)                                                  # scalar >= group order
        bad[-32:] = (c["r"]).to_bytes(32, "little")
        assert P.decode_proof(c, G, n, 2, bytes(bad)) is None
        assert P.decode_proof(c, G, n, 2, blob[:-1]) is None
    # BLS12-381: a point on the curve but outside G1 is a FormatError
    c = P.BLS12_381
    G = P.make_group("bls12_381", False)
    pk, prover, proof = P.prove_case("bls12_381", 4, [9, 3], [5, 6], shadow=False)
    blob = bytearray(P.encode_proof(c, 4, 2, proof))
    blob[12:60] = P.compress_point(c, (0, 2))
    assert P.decode_proof(c, G, 4, 2, bytes(blob)) is None


def test_container_version_2_uncompressed_points():
    """container version 2 (uncompressed points; include/bpp_amd.h): round trip, and every rejection of version 1 plus the
    ones of the point form -- compressed flag set, a coordinate >= p, a point off the curve, a point outside G1"""
    for cname, ub in (("secp256k1", 65), ("bls12_381", 96)):
        c = P.CURVES[cname]
        G = P.make_group(cname, False)
        n, vals = 4, [9, 3]
        pk, prover, proof = P.prove_case(cname, n, vals, [5, 6], shadow=False)
        blob = P.encode_proof(c, n, 2, proof, version=2)
        k = 3
        assert len(blob) == 12 + (3 + 2 * k) * ub + 96 and blob[4] == 2
        back = P.decode_proof(c, G, n, 2, blob, version=2)
        assert back is not None and back.verify(pk, n, prover.commitment_vec)
        assert back.A == proof.A and back.proof.L_vec == proof.proof.L_vec and back.proof.R_vec == proof.proof.R_vec
        assert P.decode_proof(c, G, n, 2, blob, version=1) is None                      # a version 2 blob is not version 1
        assert P.decode_proof(c, G, n, 2, P.encode_proof(c, n, 2, proof), version=2) is None
        bad = bytearray(blob)
        bad[12 + ub - 1] ^= 1                                                            # y of A off by one: not on the curve
        assert P.decode_proof(c, G, n, 2, bytes(bad), version=2) is None
        bad = bytearray(blob)
        bad[12] = 0x80 | bad[12] if cname == "bls12_381" else 0x02                        # compressed flag / prefix
        assert P.decode_proof(c, G, n, 2, bytes(bad), version=2) is None
        bad = bytearray(blob)
        off = 12 if cname == "bls12_381" else 13
        bad[off:off + (48 if cname == "bls12_381" else 32)] = (c["p"]).to_bytes(48 if cname == "bls12_381" else 32, "big")   # x = p
        assert P.decode_proof(c, G, n, 2, bytes(bad), version=2) is None
        bad = bytearray(blob)
        bad[-32:] = (c["r"]).to_bytes(32, "little")
        assert P.decode_proof(c, G, n, 2, bytes(bad), version=2) is None
        inf = P.uncompressed_point(c, None)
        assert P.parse_uncompressed_point(c, inf) == (True, None)
    c = P.BLS12_381
    G = P.make_group("bls12_381", False)
    pk, prover, proof = P.prove_case("bls12_381", 4, [9, 3], [5, 6], shadow=False)
    blob = bytearray(P.encode_proof(c, 4, 2, proof, version=2))
    blob[12:108] = P.uncompressed_point(c, (0, 2))                                       # order 3: on the curve, outside G1
    assert P.decode_proof(c, G, 4, 2, bytes(blob), version=2) is None
