"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the C oracle (oracle/bpp_oracle.c).

Import rules: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the timed CPU baseline.  The product package
``bulletproofsplus_amd`` never does.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbpp_oracle.so")

BLS12_381 = 0
SECP256K1 = 1
ED25519 = 2   # wire-format helpers only: the C oracle has no Edwards backend (pyref.EdwardsGroup is the checker)
CURVE_IDS = {"bls12_381": BLS12_381, "secp256k1": SECP256K1, "ed25519": ED25519}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bpp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbpp_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_set_compute_dead.restype = None
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def fp_limbs(curve: int) -> int:
    return 4 if curve == ED25519 else lib().orc_fp_limbs(curve)


def point_words(curve: int) -> int:
    return 9 if curve == ED25519 else lib().orc_point_words(curve)


# ---- int <-> wire helpers ------------------------------------------------------------------

def int_to_limbs(x: int, n: int) -> np.ndarray:
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    v = 0
    for i, w in enumerate(np.asarray(a, dtype=np.uint64).tolist()):
        v |= int(w) << (64 * i)
    return v


def scalars_to_wire(xs) -> np.ndarray:
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = int_to_limbs(x, 4)
    return out


def wire_to_scalars(a) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [limbs_to_int(r) for r in a]


def point_to_wire(curve: int, P) -> np.ndarray:
    """P is None (infinity) or an (x, y) tuple of ints."""
    L = fp_limbs(curve)
    w = np.zeros(2 * L + 1, dtype=np.uint64)
    if P is None:
        w[2 * L] = 1
    else:
        w[:L] = int_to_limbs(P[0], L)
        w[L:2 * L] = int_to_limbs(P[1], L)
    return w


def points_to_wire(curve: int, Ps) -> np.ndarray:
    return np.stack([point_to_wire(curve, P) for P in Ps]) if len(Ps) else \
        np.zeros((0, point_words(curve)), dtype=np.uint64)


def wire_to_point(curve: int, w):
    L = fp_limbs(curve)
    w = np.asarray(w, dtype=np.uint64).reshape(-1)
    if int(w[2 * L]):
        return None
    return (limbs_to_int(w[:L]), limbs_to_int(w[L:2 * L]))


def wire_to_points(curve: int, a) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, point_words(curve))
    return [wire_to_point(curve, r) for r in a]


# ---- primitives ----------------------------------------------------------------------------

def field_op(curve: int, which: int, op: str, a: int, b: int | None = None) -> int:
    L = fp_limbs(curve) if which == 0 else 4
    out = np.zeros(L, dtype=np.uint64)
    A = int_to_limbs(a, L)
    if op == "inv":
        rc = lib().orc_field_inv(curve, which, _p(A), _p(out))
    else:
        B = int_to_limbs(b, L)
        rc = getattr(lib(), "orc_field_" + op)(curve, which, _p(A), _p(B), _p(out))
    assert rc == 0
    return limbs_to_int(out)


def fr_from_i32(curve: int, n: int) -> int:
    out = np.zeros(4, dtype=np.uint64)
    assert lib().orc_fr_from_i32(curve, ctypes.c_int32(n), _p(out)) == 0
    return limbs_to_int(out)


def generator(curve: int):
    out = np.zeros(point_words(curve), dtype=np.uint64)
    assert lib().orc_generator(curve, _p(out)) == 0
    return out


def on_curve(curve: int, w) -> bool:
    w = np.ascontiguousarray(w, dtype=np.uint64)
    return lib().orc_point_on_curve(curve, _p(w)) == 1


def point_add(curve: int, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(point_words(curve), dtype=np.uint64)
    assert lib().orc_point_add(curve, _p(a), _p(b), _p(out)) == 0
    return out


def point_neg(curve: int, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros(point_words(curve), dtype=np.uint64)
    assert lib().orc_point_neg(curve, _p(a), _p(out)) == 0
    return out


def point_mul(curve: int, a, k: int):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    K = int_to_limbs(k, 4)
    out = np.zeros(point_words(curve), dtype=np.uint64)
    assert lib().orc_point_mul(curve, _p(a), _p(K), _p(out)) == 0
    return out


def msm(curve: int, scalars: np.ndarray, points: np.ndarray):
    """MulVec::calculate (naive).  scalars (n,4) u64, points (n,PW) u64 -> (PW,) u64."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, point_words(curve))
    assert scalars.shape[0] == points.shape[0], "mulvec: lengths of scalars and points must match"
    out = np.zeros(point_words(curve), dtype=np.uint64)
    assert lib().orc_msm(curve, _p(scalars), _p(points), ctypes.c_size_t(scalars.shape[0]), _p(out)) == 0
    return out


# ---- protocol ------------------------------------------------------------------------------

class PublicKey:
    """PublicKey::new(length) (reference src/publickey.rs:21-48) in wire format."""

    def __init__(self, curve: int, length: int):
        PW = point_words(curve)
        self.curve = curve
        self.length = length
        self.gh = np.zeros((2, PW), dtype=np.uint64)
        self.G = np.zeros((max(length, 1), PW), dtype=np.uint64)
        self.H = np.zeros((max(length, 1), PW), dtype=np.uint64)
        assert lib().orc_pk_new(curve, ctypes.c_size_t(length), _p(self.gh), _p(self.G), _p(self.H)) == 0
        self.G = self.G[:length]
        self.H = self.H[:length]


def commit(pk: PublicKey, v: int, gamma: int):
    out = np.zeros(point_words(pk.curve), dtype=np.uint64)
    g = int_to_limbs(gamma, 4)
    assert lib().orc_commit(pk.curve, _p(pk.gh), ctypes.c_uint64(v), _p(g), _p(out)) == 0
    return out


def range_prove(pk: PublicKey, n: int, values, gammas, V=None, faithful_timing=False):
    """RangeProof::prove.  Returns (points (3+2k, PW), scalars (3,4), V (m,PW))."""
    curve = pk.curve
    m = len(values)
    mn = n * m
    k = mn.bit_length() - 1
    PW = point_words(curve)
    if V is None:
        V = np.stack([commit(pk, v, g) for v, g in zip(values, gammas)])
    V = np.ascontiguousarray(V, dtype=np.uint64)
    v = np.array(values, dtype=np.uint64)
    gm = scalars_to_wire(gammas)
    pts = np.zeros((3 + 2 * k, PW), dtype=np.uint64)
    sc = np.zeros((3, 4), dtype=np.uint64)
    lib().orc_set_compute_dead(1 if faithful_timing else 0)
    G = np.ascontiguousarray(pk.G)
    H = np.ascontiguousarray(pk.H)
    rc = lib().orc_range_prove(curve, _p(pk.gh), _p(G), _p(H), ctypes.c_size_t(n), ctypes.c_size_t(m),
                               _p(v), _p(gm), _p(V), _p(pts), _p(sc))
    lib().orc_set_compute_dead(0)
    assert rc == 0, rc
    return pts, sc, V


def range_verify(pk: PublicKey, n: int, m: int, proof_points, proof_scalars, V,
                 want_scalars=False, want_result=False, skip_msm=False, pippenger_window=0, want_challenges=False):
    """RangeProof::verify.  Returns rc (0 Ok / 1 VerificationError) or a tuple with the extras.
    pippenger_window in 2..16: the final MulVec by the bucket method instead of the reference's naive loop
    (same point; bench.py's "CPU-Pippenger" baseline)."""
    curve = pk.curve
    PW = point_words(curve)
    proof_points = np.ascontiguousarray(proof_points, dtype=np.uint64).reshape(-1, PW)
    k = (proof_points.shape[0] - 3) // 2
    proof_scalars = np.ascontiguousarray(proof_scalars, dtype=np.uint64).reshape(3, 4)
    V = np.ascontiguousarray(V, dtype=np.uint64).reshape(m, PW)
    N = 2 * n * m + 2 * k + m + 5
    sc = np.zeros((N, 4), dtype=np.uint64) if want_scalars else None
    res = np.zeros(PW, dtype=np.uint64) if want_result else None
    G = np.ascontiguousarray(pk.G)
    H = np.ascontiguousarray(pk.H)
    ch = np.zeros((3 + k, 4), dtype=np.uint64) if want_challenges else None
    rc = lib().orc_range_verify_fs(curve, _p(pk.gh), _p(G), _p(H), ctypes.c_size_t(n), ctypes.c_size_t(m),
                                   _p(proof_points), ctypes.c_size_t(k), _p(proof_scalars), _p(V),
                                   _p(sc) if sc is not None else None,
                                   _p(res) if res is not None else None,
                                   1 if skip_msm else (int(pippenger_window) if 2 <= int(pippenger_window) <= 16 else 0),
                                   _p(ch) if ch is not None else None)
    if want_challenges:
        return rc, sc, res, ch
    if not want_scalars and not want_result:
        return rc
    return rc, sc, res


def set_transcript(on: bool):
    """Transcript mode of the C oracle (csrc/transcript.hpp restated; the reference has none): prove and verify draw
    their challenges from the SHA-256 transcript instead of the reference's literals.  Global; reset after use."""
    lib().orc_set_transcript(1 if on else 0)


def set_blinding(values):
    """Blinding of the oracle's prover: None = the reference's literals, else 5 + 2k ints
    [alpha, r, s, delta, eta, d_L[0..k), d_R[0..k)] (include/bpp_amd.h "Blinding").  Global; reset after use."""
    if values is None:
        lib().orc_set_blinding(None, 0)
        return
    k = (len(values) - 5) // 2
    assert len(values) == 5 + 2 * k
    w = scalars_to_wire(list(values))
    lib().orc_set_blinding(_p(w), k)


def blinding_from_key(key: bytes, index: int, k: int, r: int):
    """the key expansion of csrc/prover_batch.hpp k_pb_blind, restated with hashlib: slot j of proof `index` is
    (c0 + 2^256 c1) mod r with c_h = SHA-256(key || "bppb" || index u64 LE || j u32 LE || h u32 LE) little-endian"""
    import hashlib
    out = []
    for j in range(5 + 2 * k):
        c = [int.from_bytes(hashlib.sha256(key + b"bppb" + index.to_bytes(8, "little") + j.to_bytes(4, "little") +
                                           h.to_bytes(4, "little")).digest(), "little") for h in (0, 1)]
        v = (c[0] + (c[1] << 256)) % r
        out.append(v or 1)
    return out


def sha256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().orc_sha256(data, ctypes.c_size_t(len(data)), out)
    return out.raw
