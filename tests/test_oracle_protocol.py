"""The C oracle's range-proof / WIP restatement against (a) the dlog-shadow known answers of
SURVEY.md section 8c, (b) complete small proofs from the independent big-integer implementation
(oracle/pyref.py), (c) the reference's only end-to-end check, src/main.rs:10-56 (verify == Ok)."""

import numpy as np
import pytest

import oracle as O
import pyref as P

CID = O.CURVE_IDS


def hexpt(curve, h):
    return None if h is None else (int(h[0], 16), int(h[1], 16))


def test_shadow_matches_survey_values(golden):
    ka = golden("shadow_known_answers.json")
    c = [x for x in ka if x["curve"] == "bls12_381"]
    main = next(x for x in c if x["n"] == 64 and x["m"] == 2)
    assert main["r_prime"] == "462cedc3fea60e21bbbbeba18354723e8ee5c30fa238203e66c6116c0958a680"
    assert main["dlog_L"][0] == "66de33444aa2674d37082775bc5397c89f159f73aa3be2a338b31112cd531866"
    assert main["dlog_R"][0] == "612893d7e89558350dfb182fee84e740e35eb58d15132b6dd044b4ff33fc0eb0"
    assert main["dlog_A"] == "73eda753299d7d483339d80809a1d80553bda402fffe5bfefffffffeffff6333"
    assert [x["msm_len"] for x in c[:7]] == [277, 80, 146, 2089, 147, 279, 1055]
    for x in ka:
        # recompute with pyref: the fixture is reproducible
        pk, pr, proof = P.prove_case(x["curve"], x["n"], x["values"], x["gammas"], shadow=True)
        assert "%064x" % proof.proof.r_prime == x["r_prime"]
        assert proof.verify(pk, x["n"], pr.commitment_vec) == x["verify_ok"]
    assert [x["verify_ok"] for x in c] == [True] * 9 + [False]


@pytest.mark.parametrize("idx", range(8))
def test_oracle_small_proofs_match_bigint(golden, idx):
    case = golden("protocol_small.json")[idx]
    curve = CID[case["curve"]]
    n, m = case["n"], case["m"]
    pk = O.PublicKey(curve, n * m)
    pts, sc, V = O.range_prove(pk, n, case["values"], case["gammas"])
    exp_pts = [hexpt(curve, case[k]) for k in ("A", "wipA", "wipB")] + \
              [hexpt(curve, h) for h in case["L"]] + [hexpt(curve, h) for h in case["R"]]
    assert O.wire_to_points(curve, pts) == exp_pts
    assert ["%064x" % s for s in O.wire_to_scalars(sc)] == [case["r_prime"], case["s_prime"], case["d_prime"]]
    assert O.wire_to_points(curve, V) == [hexpt(curve, h) for h in case["V"]]
    rc, vsc, res = O.range_verify(pk, n, m, pts, sc, V, want_scalars=True, want_result=True)
    assert ["%064x" % s for s in O.wire_to_scalars(vsc)] == case["verify_scalars"]
    assert (rc == 0) == case["verify_ok"]
    assert (O.wire_to_point(curve, res) is None) == case["verify_ok"]


def test_oracle_main_rs_case(golden):
    # reference src/main.rs:10-56: n=64, m=2, v={2,5}, gamma={3,7} => verify Ok
    full = golden("protocol_full_bls12_381.json")[0]
    ka = next(x for x in golden("shadow_known_answers.json")
              if x["curve"] == "bls12_381" and x["n"] == 64 and x["m"] == 2)
    curve = O.BLS12_381
    pk = O.PublicKey(curve, 128)
    pts = O.points_to_wire(curve, [hexpt(curve, h) for h in full["points"]])
    V = O.points_to_wire(curve, [hexpt(curve, h) for h in full["V"]])
    sc = O.scalars_to_wire([int(full[k], 16) for k in ("r_prime", "s_prime", "d_prime")])
    assert full["r_prime"] == ka["r_prime"] and full["d_prime"] == ka["d_prime"]
    # points == dlog * g (spot check two of them; make_golden.py checked all)
    g = O.generator(curve)
    assert np.array_equal(O.point_mul(curve, g, int(ka["dlog_wipB"], 16)), pts[2])
    assert np.array_equal(O.point_mul(curve, g, int(ka["dlog_L"][0], 16)), pts[3])
    rc, vsc, _ = O.range_verify(pk, 64, 2, pts, sc, V, want_scalars=True, skip_msm=True)
    vs = ["%064x" % s for s in O.wire_to_scalars(vsc)]
    assert vs[:8] == ka["vs_first8"] and vs[-4:] == ka["vs_last4"] and len(vs) == 277
    assert O.range_verify(pk, 64, 2, pts, sc, V) == 0
    # tampering => VerificationError
    sc2 = sc.copy()
    sc2[2, 0] ^= 1
    assert O.range_verify(pk, 64, 2, pts, sc2, V) == 1
    # wrong number of rounds => VerificationError branch of wip.rs:335-337
    assert O.range_verify(pk, 64, 2, np.concatenate([pts[:3], pts[3:9], pts[10:16]]), sc, V) == 1


def test_oracle_prove_c1(golden):
    # config C1: n=32, m=1 prove + verify on the CPU path
    full = golden("protocol_full_bls12_381.json")[1]
    curve = O.BLS12_381
    pk = O.PublicKey(curve, 32)
    pts, sc, V = O.range_prove(pk, 32, [31], [7])
    assert O.wire_to_points(curve, pts) == [hexpt(curve, h) for h in full["points"]]
    assert "%064x" % O.wire_to_scalars(sc)[0] == full["r_prime"]
    assert O.range_verify(pk, 32, 1, pts, sc, V) == 0


def test_oracle_pippenger_mulvec_equals_naive():
    """The bucket-method MulVec of the oracle (bench.py's CPU-Pippenger baseline only) returns the same result point
    and verdict as the reference-semantics naive MulVec, for valid and tampered proofs, several window widths."""
    import oracle as O
    for curve in (O.BLS12_381, O.SECP256K1):
        pk = O.PublicKey(curve, 16)
        pts, sc, V = O.range_prove(pk, 8, [200, 5], [3, 7])
        bad = sc.copy()
        bad[1, 0] ^= 4
        for scal, want in ((sc, 0), (bad, 1)):
            rc0, _, res0 = O.range_verify(pk, 8, 2, pts, scal, V, want_result=True)
            assert rc0 == want
            for w in (2, 5, 8, 11):
                rc, _, res = O.range_verify(pk, 8, 2, pts, scal, V, want_result=True, pippenger_window=w)
                assert rc == want and np.array_equal(res, res0), (curve, w)
