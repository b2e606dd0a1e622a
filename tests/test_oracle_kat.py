"""The C oracle against the reference's own known-answer vectors (tests/golden/secp256k1_kat.json,
transcribed from reference src/secp256k1/building_block/secp256k1/affine_point.rs:146-341 and
field/prime_field_elem.rs:642-658,:855-865) and against the BLS12-381 generator literal
(reference src/bls12_381/building_block/point/point.rs:16)."""

import random

import numpy as np
import pytest

import oracle as O
import pyref as P

S = O.SECP256K1
B = O.BLS12_381


def pt(x, y):
    return (int(x, 16), int(y, 16))


def test_secp256k1_small_multiples(golden):
    kat = golden("secp256k1_kat.json")
    g = O.generator(S)
    for k, (x, y) in enumerate(kat["g_multiples"], start=1):
        assert O.wire_to_point(S, O.point_mul(S, g, k)) == pt(x, y)      # affine_point.rs:245-254
    two = kat["two_g_decimal"]
    assert O.wire_to_point(S, O.point_add(S, g, g)) == (int(two[0]), int(two[1]))   # :141-150


def test_secp256k1_scalar_mul_kats(golden):
    kat = golden("secp256k1_kat.json")
    g = O.generator(S)
    for k, x, y in kat["scalar_mul"]:                                    # affine_point.rs:263-319
        assert O.wire_to_point(S, O.point_mul(S, g, int(k, 16))) == pt(x, y)


def test_secp256k1_add_cases(golden):
    kat = golden("secp256k1_kat.json")
    gs = [None] + [O.point_to_wire(S, pt(x, y)) for x, y in kat["g_multiples"]]
    for a, b, c in kat["add_cases"]:                                     # affine_point.rs:343-358
        assert np.array_equal(O.point_add(S, gs[a], gs[b]), gs[c])
    la = kat["large_add"]
    r = O.point_add(S, O.point_to_wire(S, pt(*la["a"])), O.point_to_wire(S, pt(*la["b"])))
    assert O.wire_to_point(S, r) == pt(*la["c"])                         # :360-366


@pytest.mark.parametrize("curve", [S, B])
def test_special_case_adds(curve):
    # affine_point.rs:152-193: vertical line, inf+P, P+inf, inf+inf
    g = O.generator(curve)
    inf = O.point_to_wire(curve, None)
    assert O.wire_to_point(curve, O.point_add(curve, g, O.point_neg(curve, g))) is None
    assert np.array_equal(O.point_add(curve, g, inf), g)
    assert np.array_equal(O.point_add(curve, inf, g), g)
    assert O.wire_to_point(curve, O.point_add(curve, inf, inf)) is None
    # bls12_381 point.rs:126-185 identities: 1g=g, 2g=g+g, 3g=g+g+g, g-g=0
    assert np.array_equal(O.point_mul(curve, g, 1), g)
    g2 = O.point_add(curve, g, g)
    assert np.array_equal(O.point_mul(curve, g, 2), g2)
    assert np.array_equal(O.point_mul(curve, g, 3), O.point_add(curve, g2, g))
    assert O.wire_to_point(curve, O.point_mul(curve, g, 0)) is None
    assert O.on_curve(curve, g2)


def test_secp256k1_field_kats(golden):
    f = golden("secp256k1_kat.json")["field"]
    m = f["mul_mod_n"]
    assert O.field_op(S, 1, "mul", int(m["a"]), int(m["b"])) == int(m["expect"])
    i = f["inv_mod_p"]
    assert O.field_op(S, 0, "inv", int(i["a"])) == int(i["expect"])


def test_bls_generator_literal(golden):
    gen = golden("bls12_381_generator.json")
    assert O.wire_to_point(B, O.generator(B)) == (int(gen["x_decimal"]), int(gen["y_decimal"]))
    assert O.on_curve(B, O.generator(B))
    # generator has order r
    r = int(gen["r_hex"], 16)
    gm1 = O.point_mul(B, O.generator(B), r - 1)
    assert np.array_equal(gm1, O.point_neg(B, O.generator(B)))


def test_bls_fr_small_kats():
    # reference bls12_381/building_block/scalar/prime_field_elem.rs:256-347 (tiny-integer KATs)
    r = P.BLS12_381["r"]
    assert O.field_op(B, 1, "sub", 9, 2) == 7
    assert O.field_op(B, 1, "mul", 2, 5) == 10
    assert O.field_op(B, 1, "mul", 10, O.field_op(B, 1, "inv", 2)) == 5
    assert O.field_op(B, 1, "mul", 5, O.field_op(B, 1, "inv", 5)) == 1
    assert O.field_op(B, 1, "mul", 9, 9) == 81
    assert O.fr_from_i32(B, -1) == r - 1 and O.fr_from_i32(B, 0) == 0 and O.fr_from_i32(B, 7) == 7
    assert O.fr_from_i32(B, -2**31) == r - 2**31


@pytest.mark.parametrize("cname,curve", [("secp256k1", S), ("bls12_381", B)])
def test_field_ops_random(cname, curve):
    c = P.CURVES[cname]
    rnd = random.Random(7)
    for which, mod in ((0, c["p"]), (1, c["r"])):
        edge = [0, 1, 2, mod - 1, mod - 2, (1 << 64) - 1, 1 << 64, (1 << 128) + 5]
        vals = edge + [rnd.randrange(mod) for _ in range(40)]
        for a in vals:
            for b in vals[:12]:
                assert O.field_op(curve, which, "mul", a, b) == a * b % mod
                assert O.field_op(curve, which, "add", a, b) == (a + b) % mod
                assert O.field_op(curve, which, "sub", a, b) == (a - b) % mod
            if a:
                assert O.field_op(curve, which, "inv", a) == pow(a, -1, mod)


@pytest.mark.parametrize("cname,curve", [("secp256k1", S), ("bls12_381", B)])
def test_msm_matches_bigint(cname, curve):
    G = P.WeierstrassGroup(P.CURVES[cname])
    rnd = random.Random(11)
    g = G.base()
    pts = [G.mul(g, rnd.randrange(1, 1000)) for _ in range(6)] + [None, g, G.neg(g)]
    scs = [rnd.randrange(G.r) for _ in range(6)] + [5, 3, 3]
    mv = P.MulVec(G)
    mv.add_scalars(scs)
    mv.add_points(pts)
    exp = mv.calculate()
    got = O.msm(curve, O.scalars_to_wire(scs), O.points_to_wire(curve, pts))
    assert O.wire_to_point(curve, got) == exp
    # empty MulVec -> zero ; length mismatch -> "panic"
    assert O.wire_to_point(curve, O.msm(curve, np.zeros((0, 4), np.uint64),
                                        np.zeros((0, O.point_words(curve)), np.uint64))) is None
    with pytest.raises(AssertionError):
        O.msm(curve, O.scalars_to_wire([1, 2]), O.points_to_wire(curve, [g]))
