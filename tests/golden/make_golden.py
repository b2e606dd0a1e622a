#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/.  Run from the repo root:

    python tests/golden/make_golden.py

Sources of the vectors
----------------------
* ``secp256k1_kat.json``    -- DATA transcribed from the reference's own unit tests
  (/root/reference/src/secp256k1/building_block/secp256k1/affine_point.rs:146-149 (2G),
  :231-242 (k*G, k=1..10), :271-297 (five 256-bit scalar-mult KATs), :324-341 (large add),
  :360-367 (add cases); field/prime_field_elem.rs:642-658 (product mod n), :855-865 (inverse mod p)).
  Only the numbers are kept; each was re-checked here with Python big integers.
* ``bls12_381_generator.json`` -- the decimal literal of reference
  src/bls12_381/building_block/point/point.rs:16 (checked on-curve here).
* ``shadow_known_answers.json`` -- dlog-shadow protocol answers (oracle/pyref.py, ShadowGroup):
  for each case the proof scalars r', s', d' and the discrete logs of every proof point.  The
  BLS12-381 (64,2), (32,1), (64,16) entries reproduce the values recorded in SURVEY.md section 8c.
* ``protocol_small.json``   -- complete proofs over the REAL curves for small (n, m), produced by the
  big-integer implementation in oracle/pyref.py (WeierstrassGroup) -- an implementation independent
  of the C oracle.  Includes the final verification MulVec scalars.
* ``protocol_full_bls12_381.json`` -- complete BLS12-381 proofs for the reference-sized cases
  (main.rs (64,2); (32,1); (64,16)), produced by the C oracle and cross-checked point by point
  against the shadow (point == dlog * g).

The reference itself cannot be executed here (no Rust toolchain, mcl_rust absent), so none of these
files is an output of the reference binary; see DESIGN.md "Oracle and parity status".
"""

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import pyref as P  # noqa: E402


def hx(x, nbytes=32):
    return "%0*x" % (2 * nbytes, x)


def pt_hex(P_, nbytes):
    return None if P_ is None else [hx(P_[0], nbytes), hx(P_[1], nbytes)]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)


# ------------------------------------------------------------------------------------------
def secp256k1_kat():
    G = P.WeierstrassGroup(P.SECP256K1)
    g_multiples = [  # affine_point.rs:231-242
        ("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798", "483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8"),
        ("C6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5", "1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A"),
        ("F9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9", "388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672"),
        ("E493DBF1C10D80F3581E4904930B1404CC6C13900EE0758474FA94ABE8C4CD13", "51ED993EA0D455B75642E2098EA51448D967AE33BFBDFE40CFE97BDC47739922"),
        ("2F8BDE4D1A07209355B4A7250A5C5128E88B84BDDC619AB7CBA8D569B240EFE4", "D8AC222636E5E3D6D4DBA9DDA6C9C426F788271BAB0D6840DCA87D3AA6AC62D6"),
        ("FFF97BD5755EEEA420453A14355235D382F6472F8568A18B2F057A1460297556", "AE12777AACFBB620F3BE96017F45C560DE80F0F6518FE4A03C870C36B075F297"),
        ("5CBDF0646E5DB4EAA398F365F2EA7A0E3D419B7E0330E39CE92BDDEDCAC4F9BC", "6AEBCA40BA255960A3178D6D861A54DBA813D0B813FDE7B5A5082628087264DA"),
        ("2F01E5E15CCA351DAFF3843FB70F3C2F0A1BDD05E5AF888A67784EF3E10A2A01", "5C4DA8A741539949293D082A132D13B4C2E213D6BA5B7617B5DA2CB76CBDE904"),
        ("ACD484E2F0C7F65309AD178A9F559ABDE09796974C57E714C35F110DFC27CCBE", "CC338921B0A7D9FD64380971763B61E9ADD888A4375F8E0F05CC262AC64F9C37"),
        ("A0434D9E47F3C86235477C7B1AE6AE5D3442D49B1943C2B752A68E2A47E247C7", "893ABA425419BC27A3B6C7E693A24C696F794C2ED877A1593CBEE53B037368D7"),
    ]
    scalar_mul = [  # affine_point.rs:271-297
        ("AA5E28D6A97A2479A65527F7290311A3624D4CC0FA1578598EE3C2613BF99522", "34F9460F0E4F08393D192B3C5133A6BA099AA0AD9FD54EBCCFACDFA239FF49C6", "0B71EA9BD730FD8923F6D25A7A91E7DD7728A960686CB5A901BB419E0F2CA232"),
        ("7E2B897B8CEBC6361663AD410835639826D590F393D90A9538881735256DFAE3", "D74BF844B0862475103D96A611CF2D898447E288D34B360BC885CB8CE7C00575", "131C670D414C4546B88AC3FF664611B1C38CEB1C21D76369D7A7A0969D61D97D"),
        ("6461E6DF0FE7DFD05329F41BF771B86578143D4DD1F7866FB4CA7E97C5FA945D", "E8AECC370AEDD953483719A116711963CE201AC3EB21D3F3257BB48668C6A72F", "C25CAF2F0EBA1DDB2F0F3F47866299EF907867B7D27E95B3873BF98397B24EE1"),
        ("376A3A2CDCD12581EFFF13EE4AD44C4044B8A0524C42422A7E1E181E4DEECCEC", "14890E61FCD4B0BD92E5B36C81372CA6FED471EF3AA60A3E415EE4FE987DABA1", "297B858D9F752AB42D3BCA67EE0EB6DCD1C2B7B0DBE23397E66ADC272263F982"),
        ("1B22644A7BE026548810C378D0B2994EEFA6D2B9881803CB02CEFF865287D1B9", "F73C65EAD01C5126F28F442D087689BFA08E12763E0CEC1D35B01751FD735ED3", "F449A8376906482A84ED01479BD18882B919C140D638307F0C0934BA12590BDE"),
    ]
    large_add = dict(  # affine_point.rs:324-341 ; l1 + l2 = l3
        a=("A6B594B38FB3E77C6EDF78161FADE2041F4E09FD8497DB776E546C41567FEB3C", "71444009192228730CD8237A490FEBA2AFE3D27D7CC1136BC97E439D13330D55"),
        b=("00000000000000000000003B78CE563F89A0ED9414F5AA28AD0D96D6795F9C63", "3F3979BF72AE8202983DC989AEC7F2FF2ED91BDD69CE02FC0700CA100E59DDF3"),
        c=("E24CE4BEEE294AA6350FAA67512B99D388693AE4E7F53D19882A6EA169FC1CE1", "8B71E83545FC2B5872589F99D948C03108D36797C4DE363EBD3FF6A9E1A95B10"),
    )
    add_cases = [[1, 2, 3], [2, 2, 4], [2, 6, 8], [3, 4, 7], [5, 1, 6], [5, 2, 7], [8, 1, 9], [9, 1, 10]]
    two_g_dec = (  # affine_point.rs:146-149
        "89565891926547004231252920425935692360644145829622209833684329913297188986597",
        "12158399299693830322967808612713398636155367887041628176798871954788371653930",
    )
    field = dict(
        # field/prime_field_elem.rs:642-658 : 1234 * rhs mod n
        mul_mod_n=dict(a="1234",
                       b="63954422509139660694275478881573291931659433822585593108077818434106113196321",
                       expect="65344605666012213284100148944976995885360063020612010813911848313075706616617"),
        # field/prime_field_elem.rs:855-865 : inverse of 1112121212121 mod p
        inv_mod_p=dict(a="1112121212121",
                       expect="52624297956533532283067125375510330718705195823487497799082320305224600546911"),
    )
    # re-check every transcribed number
    g = G.base()
    for k, (x, y) in enumerate(g_multiples, start=1):
        assert G.mul(g, k) == (int(x, 16), int(y, 16))
    for k, x, y in scalar_mul:
        assert G.mul(g, int(k, 16)) == (int(x, 16), int(y, 16))
    la = {k: (int(v[0], 16), int(v[1], 16)) for k, v in large_add.items()}
    assert G.add(la["a"], la["b"]) == la["c"]
    assert G.add(g, g) == (int(two_g_dec[0]), int(two_g_dec[1]))
    n_, p_ = P.SECP256K1["r"], P.SECP256K1["p"]
    assert int(field["mul_mod_n"]["a"]) * int(field["mul_mod_n"]["b"]) % n_ == int(field["mul_mod_n"]["expect"])
    assert pow(int(field["inv_mod_p"]["a"]), -1, p_) == int(field["inv_mod_p"]["expect"])
    dump("secp256k1_kat.json", dict(
        source="reference unit tests (see make_golden.py docstring for file:line)",
        g_multiples=[list(t) for t in g_multiples], scalar_mul=[list(t) for t in scalar_mul],
        large_add={k: list(v) for k, v in large_add.items()}, add_cases=add_cases,
        two_g_decimal=list(two_g_dec), field=field))


def bls_generator():
    c = P.BLS12_381
    x = "3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507"
    y = "1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569"
    assert int(x) == c["gx"] and int(y) == c["gy"]
    assert P.WeierstrassGroup(c).on_curve((int(x), int(y)))
    dump("bls12_381_generator.json", dict(
        source="reference src/bls12_381/building_block/point/point.rs:16 (decimal literal)",
        x_decimal=x, y_decimal=y, p_hex=hx(c["p"], 48), r_hex=hx(c["r"], 32), b=4))


SHADOW_CASES = [
    # (n, values, gammas)
    (64, [2, 5], [3, 7]),          # reference src/main.rs:10-56
    (32, [31], [7]),               # config C1
    (64, [31], [7]),
    (64, [31] * 16, [7] * 16),     # config C2 / C4
    (32, [31] * 2, [7] * 2), (32, [31] * 4, [7] * 4), (64, [31] * 8, [7] * 8),
    (8, [200, 5], [3, 7]), (8, [77], [9]),
    (8, [300, 5], [3, 7]),         # out of range -> must fail verification
]


def shadow_known_answers():
    out = []
    for cname in ("bls12_381", "secp256k1"):
        for n, vals, gams in SHADOW_CASES:
            pk, pr, proof = P.prove_case(cname, n, vals, gams, shadow=True)
            w = proof.proof
            mv = proof.verify_mulvec(pk, n, pr.commitment_vec)
            ok = proof.verify(pk, n, pr.commitment_vec)
            out.append(dict(
                curve=cname, n=n, m=len(vals), values=vals, gammas=gams, verify_ok=ok,
                msm_len=len(mv.scalars),
                r_prime=hx(w.r_prime), s_prime=hx(w.s_prime), d_prime=hx(w.d_prime),
                dlog_A=hx(proof.A), dlog_wipA=hx(w.A), dlog_wipB=hx(w.B),
                dlog_L=[hx(x) for x in w.L_vec], dlog_R=[hx(x) for x in w.R_vec],
                dlog_V=[hx(x) for x in pr.commitment_vec],
                # a few verification scalars as spot checks (full list would be large)
                vs_first8=[hx(x) for x in mv.scalars[:8]],
                vs_last4=[hx(x) for x in mv.scalars[-4:]],
                vs_sum=hx(sum(mv.scalars) % pk.G.r),
            ))
    # values recorded in SURVEY.md section 8c
    c0 = out[0]
    assert c0["r_prime"] == "462cedc3fea60e21bbbbeba18354723e8ee5c30fa238203e66c6116c0958a680"
    assert c0["s_prime"] == "106936f2950fbeea308b86448e75b3cb55d7db144d61c9459cd52c38057a1741"
    assert c0["d_prime"] == "2a131ec85c442f5cdfc0bca009895e204e2d44bb194041b7b6f191836452353b"
    assert int(c0["dlog_wipB"], 16) == 0x4506
    assert out[1]["r_prime"] == "73eda753299d7d483339d80809a1d80553bda402ffa6806798ac6509df761957"
    assert out[3]["d_prime"] == "68040e5fce14cf500dcae5149da688298e4e932cffb4be82864c42e09d4a0e3f"
    dump("shadow_known_answers.json", out)


def protocol_small():
    out = []
    for cname in ("bls12_381", "secp256k1"):
        nb = P.CURVES[cname]["fp_bytes"]
        for n, vals, gams in [(8, [200, 5], [3, 7]), (8, [77], [9]), (4, [9, 3, 15, 0], [1, 2, 3, 4]),
                              (8, [300, 5], [3, 7])]:
            pk, pr, proof = P.prove_case(cname, n, vals, gams, shadow=False)
            w = proof.proof
            mv = proof.verify_mulvec(pk, n, pr.commitment_vec)
            ok = proof.verify(pk, n, pr.commitment_vec)
            out.append(dict(
                curve=cname, n=n, m=len(vals), values=vals, gammas=gams, verify_ok=ok,
                A=pt_hex(proof.A, nb), wipA=pt_hex(w.A, nb), wipB=pt_hex(w.B, nb),
                L=[pt_hex(x, nb) for x in w.L_vec], R=[pt_hex(x, nb) for x in w.R_vec],
                V=[pt_hex(x, nb) for x in pr.commitment_vec],
                r_prime=hx(w.r_prime), s_prime=hx(w.s_prime), d_prime=hx(w.d_prime),
                verify_scalars=[hx(x) for x in mv.scalars],
            ))
    dump("protocol_small.json", out)


def protocol_full_bls():
    import numpy as np
    import oracle as O
    cid = O.BLS12_381
    nb = 48
    Gpy = P.WeierstrassGroup(P.BLS12_381)
    out = []
    for n, vals, gams in [(64, [2, 5], [3, 7]), (32, [31], [7]), (64, [31] * 16, [7] * 16)]:
        m = len(vals)
        opk = O.PublicKey(cid, n * m)
        pts, sc, V = O.range_prove(opk, n, vals, gams)
        rc = O.range_verify(opk, n, m, pts, sc, V)
        assert rc == 0
        # cross-check every point against the shadow: point == dlog * g
        _, spr, sproof = P.prove_case("bls12_381", n, vals, gams, shadow=True)
        dl = [sproof.A, sproof.proof.A, sproof.proof.B] + sproof.proof.L_vec + sproof.proof.R_vec
        ptl = O.wire_to_points(cid, pts)
        for d, pt in zip(dl, ptl):
            assert Gpy.mul(Gpy.base(), d) == pt
        for d, pt in zip(spr.commitment_vec, O.wire_to_points(cid, V)):
            assert Gpy.mul(Gpy.base(), d) == pt
        scl = O.wire_to_scalars(sc)
        assert scl == [sproof.proof.r_prime, sproof.proof.s_prime, sproof.proof.d_prime]
        out.append(dict(curve="bls12_381", n=n, m=m, values=vals, gammas=gams, verify_ok=True,
                        points=[pt_hex(x, nb) for x in ptl],
                        points_order="A, wip.A, wip.B, L_0..L_{k-1}, R_0..R_{k-1}",
                        V=[pt_hex(x, nb) for x in O.wire_to_points(cid, V)],
                        r_prime=hx(scl[0]), s_prime=hx(scl[1]), d_prime=hx(scl[2])))
        print("full case", n, m, "done")
    dump("protocol_full_bls12_381.json", out)


if __name__ == "__main__":
    secp256k1_kat()
    bls_generator()
    shadow_known_answers()
    protocol_small()
    protocol_full_bls()
