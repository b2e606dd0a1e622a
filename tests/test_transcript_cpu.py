"""CPU: the engine's SHA-256 / HMAC (csrc/sha256.hpp) against the reference's own known answers, and the
Fiat-Shamir transcript (csrc/transcript.hpp) against its hashlib restatement (oracle/pyref.FsTranscript), through
the host build of the same headers (tests/host/transcript_host_test.cpp).  The reference has no transcript (every
challenge is a literal): the protocol side is parity-unpinned and pinned only by this restatement."""

import hashlib
import os
import struct
import subprocess
import tempfile

import pytest

import pyref as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# BPP_HOST_SANITIZE=1: host builds under ASan + UBSan (see tests/test_host_arith_cpu.py)
SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("BPP_HOST_SANITIZE") else []


@pytest.fixture(scope="module")
def harness():
    exe = os.path.join(tempfile.gettempdir(), "bpp_transcript_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17"] + SANITIZE + ["-o", exe, os.path.join(ROOT, "tests", "host", "transcript_host_test.cpp")])
    return exe


def test_sha256_reference_kats(harness, golden):
    kat = golden("sha256_kat.json")
    for v in kat["sha256"]:
        rep = v.get("repeat", 1)
        got = subprocess.check_output([harness, "sha", v["msg_hex"], str(rep)]).decode().strip()
        assert got == v["digest"]
        assert hashlib.sha256(bytes.fromhex(v["msg_hex"]) * rep).hexdigest() == v["digest"]    # the checker agrees too
        assert subprocess.check_output([harness, "sha", got]).decode().strip() == v["digest2"]


def test_hmac_reference_kats(harness, golden):
    for v in golden("sha256_kat.json")["hmac_sha256"]:
        key = bytes.fromhex(v["key_hex"]) if "key_hex" in v else v["key_ascii"].encode()
        msg = bytes.fromhex(v["msg_hex"]) if "msg_hex" in v else v["msg_ascii"].encode()
        got = subprocess.check_output([harness, "hmac", key.hex(), msg.hex()]).decode().strip()
        assert got == v["digest"]


def _words(b):
    return list(struct.unpack("<%dI" % (len(b) // 4), b))


@pytest.mark.parametrize("curve,cid,n,vals", [("secp256k1", 1, 8, [200, 5]), ("bls12_381", 0, 4, [9]),
                                              ("secp256k1", 1, 4, [3, 7, 1, 15]), ("ed25519", 2, 4, [9, 3])])
def test_transcript_matches_hashlib_restatement(harness, curve, cid, n, vals):
    m = len(vals)
    G = P.make_group(curve, False)
    pk = P.PublicKey(G, n * m)
    fs = P.FsTranscript(P.CURVES[curve], cid, n, m, pk)
    P.Transcript.fs = fs
    try:
        pk, prover, proof = P.prove_case(curve, n, vals, [3 + j for j in range(m)], shadow=False)
        assert proof.verify(pk, n, prover.commitment_vec)          # a proof made under the transcript verifies under it
        ch = fs.verifier_challenges(proof, prover.commitment_vec)
    finally:
        P.Transcript.fs = None
    assert not proof.verify(pk, n, prover.commitment_vec)          # ... and not under the reference's constants
    w = proof.proof
    k = len(w.L_vec)
    rec = [proof.A, w.A, w.B] + list(w.L_vec) + list(w.R_vec) + list(prover.commitment_vec)
    recb = b"".join(fs.wire_bytes(pt) for pt in rec)
    pkb = b"".join(fs.wire_bytes(pt) for pt in [pk.g, pk.h] + list(pk.G_vec) + list(pk.H_vec))
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(struct.pack("<6I", cid, n, m, k, len(pkb) // 4, len(recb) // 4) + pkb + recb)
        path = f.name
    try:
        out = subprocess.check_output([harness, "tr", path]).decode().split()
    finally:
        os.unlink(path)
    words = [int(x, 16) for x in out]
    st0 = b"".join(struct.pack(">I", x) for x in words[:8])
    assert st0 == fs.st0
    got = []
    for i in range(3 + k):
        ws = words[8 + 8 * i: 16 + 8 * i]
        got.append(sum(x << (32 * j) for j, x in enumerate(ws)))
    assert got == [ch["y"], ch["z"], ch["e"]] + ch["e_rounds"]


@pytest.mark.parametrize("cname,cid,n,vals", [("secp256k1", 1, 8, [200, 5]), ("bls12_381", 0, 4, [9, 3]), ("bls12_381", 0, 8, [77])])
def test_c_oracle_transcript_mode_matches_pyref(cname, cid, n, vals):
    """Two independent restatements of the transcript (C with its own SHA-256, Python with hashlib) make the same proof
    and derive the same challenges; a transcript proof fails under the reference's constants and vice versa."""
    import numpy as np
    import oracle as O
    assert O.sha256(b"abc") == hashlib.sha256(b"abc").digest()
    m = len(vals)
    gam = [3 + j for j in range(m)]
    opk = O.PublicKey(cid, n * m)
    cpts, csc, cV = O.range_prove(opk, n, vals, gam)                      # the reference's constants
    O.set_transcript(True)
    try:
        pts, sc, V = O.range_prove(opk, n, vals, gam)
        rc, _, _, ch = O.range_verify(opk, n, m, pts, sc, V, want_challenges=True)
        assert rc == 0
        assert O.range_verify(opk, n, m, cpts, csc, cV) == 1               # constants' proof under the transcript
        bad = sc.copy()
        bad[0, 0] ^= np.uint64(1)
        assert O.range_verify(opk, n, m, pts, bad, V) == 1
    finally:
        O.set_transcript(False)
    assert O.range_verify(opk, n, m, pts, sc, V) == 1                     # transcript's proof under the constants
    assert O.range_verify(opk, n, m, cpts, csc, cV) == 0
    G = P.make_group(cname, False)
    fs = P.FsTranscript(P.CURVES[cname], cid, n, m, P.PublicKey(G, n * m))
    P.Transcript.fs = fs
    try:
        _, prover, proof = P.prove_case(cname, n, vals, gam, shadow=False)
        chp = fs.verifier_challenges(proof, prover.commitment_vec)
    finally:
        P.Transcript.fs = None
    w = proof.proof
    assert np.array_equal(O.points_to_wire(cid, [proof.A, w.A, w.B] + list(w.L_vec) + list(w.R_vec)), pts)
    assert O.wire_to_scalars(sc) == [w.r_prime, w.s_prime, w.d_prime]
    assert O.wire_to_scalars(ch) == [chp["y"], chp["z"], chp["e"]] + chp["e_rounds"]
