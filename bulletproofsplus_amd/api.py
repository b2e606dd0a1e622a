"""Host-side mirror of the reference crate's public API for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (paths relative to /root/reference/src):

  Arith.init()                        bls12_381/building_block/arith.rs:6-19
  MulVec                              bls12_381/building_block/mulvec.rs:7-53
  PublicKey(length) / .commitment     publickey.rs:13-52
  RangeProver / .commit               range/prover.rs:13-42
  RangeProof.prove / .verify          range/mod.rs:25-78
  WeightedInnerProductProof (fields)  weighted_inner_product_proof.rs:25-33
  ProofError                          errors.rs:14-50 (only VerificationError is ever constructed)

plus ``BatchVerifier``, the device-resident batch path the reference does not have.

Data is numpy ``uint64`` in the wire format of include/bpp_amd.h: scalars (..., 4), points (..., 2L+1).
Python ints are accepted for scalars.  Every call runs HIP kernels; nothing is computed on the CPU
beyond (de)serialisation.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import BLS12_381_G1, SECP256K1, ED25519, CURVE_IDS, BppError, check  # noqa: F401


class ProofError(Exception):
    """reference src/errors.rs:14-50"""


class VerificationError(ProofError):
    """ProofError::VerificationError"""


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def scalar_to_wire(x) -> np.ndarray:
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(4)
    x = int(x)
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def scalars_to_wire(xs) -> np.ndarray:
    if isinstance(xs, np.ndarray) and xs.dtype == np.uint64:
        return np.ascontiguousarray(xs).reshape(-1, 4)
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = scalar_to_wire(x)
    return out


def wire_to_int(a) -> int:
    v = 0
    for i, w in enumerate(np.asarray(a, dtype=np.uint64).reshape(-1).tolist()):
        v |= int(w) << (64 * i)
    return v


class Arith:
    """One context per (curve, device); ``Arith.init()`` mirrors the reference's global one-time init."""

    _ctxs = {}

    def __init__(self, curve=BLS12_381_G1, device=0):
        if isinstance(curve, str):
            curve = CURVE_IDS[curve]
        self.curve = curve
        self.device = device
        self.L = _lib.FP_LIMBS[curve]
        self.PW = 2 * self.L + 1
        h = ctypes.c_void_p()
        check(_lib.lib().bpp_init(curve, device, ctypes.byref(h)), "bpp_init")
        self.handle = h

    @classmethod
    def init(cls, curve=BLS12_381_G1, device=0) -> "Arith":
        if isinstance(curve, str):
            curve = CURVE_IDS[curve]
        key = (curve, device)
        if key not in cls._ctxs:
            cls._ctxs[key] = cls(curve, device)
        return cls._ctxs[key]

    def set_verify_cache(self, on: bool):
        """the per-key small-table verifiers behind RangeProof.verify (include/bpp_amd.h, bpp_range_verify); on by default"""
        check(_lib.lib().bpp_set_verify_cache(self.handle, 1 if on else 0), "bpp_set_verify_cache")

    # Point::zero()
    def zero_point(self) -> np.ndarray:
        z = np.zeros(self.PW, dtype=np.uint64)
        z[2 * self.L] = 1
        return z

    def is_zero(self, p) -> bool:
        return int(np.asarray(p).reshape(-1)[2 * self.L]) != 0

    # Point * PrimeFieldElem, n independent pairs
    def scalar_mul(self, scalars, points) -> np.ndarray:
        sc = scalars_to_wire(scalars)
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.PW)
        if sc.shape[0] != pts.shape[0]:
            raise ValueError("scalar_mul: lengths must match")
        out = np.zeros((sc.shape[0], self.PW), dtype=np.uint64)
        check(_lib.lib().bpp_scalar_mul_batch(self.handle, _ptr(sc), _ptr(pts), sc.shape[0], _ptr(out)),
              "bpp_scalar_mul_batch")
        return out


class MulVec:
    """reference bls12_381/building_block/mulvec.rs: append scalars, append points, calculate()."""

    def __init__(self, arith: Arith):
        self.arith = arith
        self.scalars = []
        self.points = []

    def add_scalar(self, s):
        self.scalars.append(scalar_to_wire(s))

    def add_scalars(self, ss):
        for s in ss:
            self.add_scalar(s)

    def add_point(self, p):
        self.points.append(np.ascontiguousarray(p, dtype=np.uint64).reshape(self.arith.PW))

    def add_points(self, ps):
        for p in np.asarray(ps, dtype=np.uint64).reshape(-1, self.arith.PW):
            self.add_point(p)

    def calculate(self) -> np.ndarray:
        if len(self.scalars) != len(self.points):
            # the reference panics here (mulvec.rs:23-25)
            raise RuntimeError("mulvec: lengths of scalars and points must match")
        n = len(self.scalars)
        sc = np.stack(self.scalars) if n else np.zeros((0, 4), np.uint64)
        pts = np.stack(self.points) if n else np.zeros((0, self.arith.PW), np.uint64)
        out = np.zeros(self.arith.PW, dtype=np.uint64)
        check(_lib.lib().bpp_msm(self.arith.handle, _ptr(sc), _ptr(pts), n, _ptr(out)), "bpp_msm")
        return out


def msm_pippenger(arith: Arith, scalars, points, window_bits: int = 0) -> np.ndarray:
    """MulVec::calculate through the bucket-method pipeline (any n; window_bits 0 = chosen from n)."""
    sc = scalars_to_wire(scalars)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, arith.PW)
    if sc.shape[0] != pts.shape[0]:
        raise RuntimeError("mulvec: lengths of scalars and points must match")
    out = np.zeros(arith.PW, dtype=np.uint64)
    check(_lib.lib().bpp_msm_pippenger(arith.handle, _ptr(sc), _ptr(pts), sc.shape[0], window_bits, _ptr(out)),
          "bpp_msm_pippenger")
    return out


def msm_workspace_bytes(arith: Arith, n: int, window_bits: int = 0) -> int:
    return _lib.lib().bpp_msm_workspace_bytes(arith.handle, n, window_bits)


def msm_device(arith: Arith, d_scalars: int, d_points: int, n: int, d_out: int, d_workspace: int, workspace_bytes: int,
               window_bits: int = 0, d_status: int = 0, stream: int = 0):
    """MulVec::calculate with every buffer in HBM (raw device pointers, e.g. torch tensors' data_ptr()), asynchronous
    on `stream`: d_scalars (n, 4) u64, d_points (n, PW) u64 wire points, d_out one wire point, d_status one u32."""
    check(_lib.lib().bpp_msm_device(arith.handle, d_scalars or None, d_points or None, n, window_bits, d_out,
                                    d_status or None, d_workspace, workspace_bytes, stream or None), "bpp_msm_device")


MSM_STAGES = ("sort", "chunks", "fold", "reduce", "final")


def msm_set_profiling(arith: Arith, on: bool):
    check(_lib.lib().bpp_msm_set_profiling(arith.handle, 1 if on else 0), "bpp_msm_set_profiling")


def msm_profile(arith: Arith):
    """-> ({stage: mean ms}, passes, shape dict of the last bpp_msm_device call), HIP events on the launch stream"""
    ms = (ctypes.c_float * 5)()
    passes = ctypes.c_size_t()
    sh = (ctypes.c_uint32 * 8)()
    check(_lib.lib().bpp_msm_profile(arith.handle, ms, ctypes.byref(passes), sh), "bpp_msm_profile")
    keys = ("n", "items", "windows", "narrow_bits", "wide_windows", "buckets", "chunk_entries", "window_bits")
    return {k: float(ms[i]) for i, k in enumerate(MSM_STAGES)}, passes.value, {k: int(sh[i]) for i, k in enumerate(keys)}


def msm_batch(arith: Arith, scalars, points, lens) -> np.ndarray:
    """`len(lens)` independent MulVecs in one launch."""
    sc = scalars_to_wire(scalars)
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, arith.PW)
    ln = np.ascontiguousarray(lens, dtype=np.uint32)
    if int(ln.sum()) != sc.shape[0] or sc.shape[0] != pts.shape[0]:
        raise RuntimeError("mulvec: lengths of scalars and points must match")
    out = np.zeros((len(ln), arith.PW), dtype=np.uint64)
    check(_lib.lib().bpp_msm_batch(arith.handle, _ptr(sc), _ptr(pts), _ptr(ln), len(ln), _ptr(out)), "bpp_msm_batch")
    return out


def wip_fold_round(arith: Arith, a, b, G, H, y_nhat, e):
    """One folding round of WeightedInnerProductProof::prove (wip.rs:147-164).  a, b: (len, 4) scalars; G, H: (len, PW)
    points.  Returns the folded (a, b, G, H) of length len / 2."""
    aw = np.ascontiguousarray(scalars_to_wire(a)).copy()
    bw = np.ascontiguousarray(scalars_to_wire(b)).copy()
    Gw = np.ascontiguousarray(G, dtype=np.uint64).reshape(-1, arith.PW).copy()
    Hw = np.ascontiguousarray(H, dtype=np.uint64).reshape(-1, arith.PW).copy()
    n = aw.shape[0]
    if not (bw.shape[0] == Gw.shape[0] == Hw.shape[0] == n):
        raise AssertionError("wip fold: vector lengths must match")          # wip.rs:60-67 asserts
    check(_lib.lib().bpp_wip_fold_round(arith.handle, _ptr(aw), _ptr(bw), _ptr(Gw), _ptr(Hw), n,
                                        _ptr(scalar_to_wire(y_nhat)), _ptr(scalar_to_wire(e))), "bpp_wip_fold_round")
    h = n // 2
    return aw[:h], bw[:h], Gw[:h], Hw[:h]


def compressed_bytes(arith: Arith) -> int:
    """bytes of one compressed point (48 BLS12-381 G1, 33 secp256k1 SEC1; 0 = not offered)"""
    return _lib.lib().bpp_point_compressed_bytes(arith.curve)


def compress_points(arith: Arith, points) -> np.ndarray:
    """wire points (n, PW) u64 -> (n, compressed_bytes) u8.  No reference counterpart (include/bpp_amd.h)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, arith.PW)
    out = np.zeros((pts.shape[0], compressed_bytes(arith)), dtype=np.uint8)
    check(_lib.lib().bpp_points_compress(arith.handle, _ptr(pts), pts.shape[0], _ptr(out)), "bpp_points_compress")
    return out


def decompress_points(arith: Arith, data):
    """(n, compressed_bytes) u8 -> (wire points (n, PW) u64, ok (n,) u32: 0 valid / 1 malformed -> infinity)"""
    cb = compressed_bytes(arith)
    raw = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, cb)
    pts = np.zeros((raw.shape[0], arith.PW), dtype=np.uint64)
    ok = np.zeros(raw.shape[0], dtype=np.uint32)
    check(_lib.lib().bpp_points_decompress(arith.handle, _ptr(raw), raw.shape[0], _ptr(pts), _ptr(ok)),
          "bpp_points_decompress")
    return pts, ok


def proof_bytes(arith: Arith, n: int, m: int, version: int = 1) -> int:
    """bytes of one serialized proof (include/bpp_amd.h, "the container"); version 2 = uncompressed points"""
    return _lib.lib().bpp_proof_bytes_version(arith.curve, n, m, version)


def uncompressed_bytes(arith: Arith) -> int:
    """bytes of one uncompressed point (96 BLS12-381 G1, 65 secp256k1 SEC1; 0 = not offered)"""
    return _lib.lib().bpp_point_uncompressed_bytes(arith.curve)


def uncompressed_points(arith: Arith, points) -> np.ndarray:
    """wire points (n, PW) u64 -> (n, uncompressed_bytes) u8: the point encoding of container version 2"""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, arith.PW)
    out = np.zeros((pts.shape[0], uncompressed_bytes(arith)), dtype=np.uint8)
    check(_lib.lib().bpp_points_uncompressed(arith.handle, _ptr(pts), pts.shape[0], _ptr(out)), "bpp_points_uncompressed")
    return out


def encode_proofs(arith: Arith, n: int, m: int, points, scalars, version: int = 1) -> np.ndarray:
    """points (count, 3+2k, PW), scalars (count, 3, 4) -> (count, proof_bytes) u8.  No reference counterpart."""
    k = (n * m).bit_length() - 1
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3 + 2 * k, arith.PW)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 3, 4)
    if sc.shape[0] != pts.shape[0]:
        raise RuntimeError("encode_proofs: one scalar triple per proof")
    out = np.zeros((pts.shape[0], proof_bytes(arith, n, m, version)), dtype=np.uint8)
    check(_lib.lib().bpp_proofs_encode_version(arith.handle, n, m, version, _ptr(pts), _ptr(sc), pts.shape[0], _ptr(out)),
          "bpp_proofs_encode_version")
    return out


def decode_proofs(arith: Arith, n: int, m: int, data):
    """(count, proof_bytes) u8 -> (points, scalars, status); status 0 valid / 2 FormatError"""
    k = (n * m).bit_length() - 1
    raw = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, proof_bytes(arith, n, m))
    count = raw.shape[0]
    pts = np.zeros((count, 3 + 2 * k, arith.PW), dtype=np.uint64)
    sc = np.zeros((count, 3, 4), dtype=np.uint64)
    st = np.zeros(count, dtype=np.uint32)
    check(_lib.lib().bpp_proofs_decode(arith.handle, n, m, _ptr(raw), count, _ptr(pts), _ptr(sc), _ptr(st)), "bpp_proofs_decode")
    return pts, sc, st


class FormatError(ProofError):
    """ProofError::FormatError (reference src/errors.rs:20): a serialized proof that does not parse"""


class PublicKey:
    """reference publickey.rs:13-52.  Fields g, h, G_vec, H_vec as in the reference."""

    def __init__(self, arith: Arith, length: int):
        self.arith = arith
        PW = arith.PW
        self.gh = np.zeros((2, PW), dtype=np.uint64)
        self.G_vec = np.zeros((max(length, 1), PW), dtype=np.uint64)
        self.H_vec = np.zeros((max(length, 1), PW), dtype=np.uint64)
        check(_lib.lib().bpp_pk_new(arith.handle, length, _ptr(self.gh), _ptr(self.G_vec), _ptr(self.H_vec)),
              "bpp_pk_new")
        self.G_vec = self.G_vec[:length]
        self.H_vec = self.H_vec[:length]

    @classmethod
    def new(cls, arith: Arith, length: int) -> "PublicKey":
        return cls(arith, length)

    @classmethod
    def hashed(cls, arith: Arith, length: int, label: bytes = b"") -> "PublicKey":
        """Generators hashed to the group from `label` (g stays the base point): what a deployment uses instead of the
        reference's test generators.  No reference counterpart (include/bpp_amd.h, bpp_pk_hashed)."""
        self = cls.__new__(cls)
        self.arith = arith
        PW = arith.PW
        self.gh = np.zeros((2, PW), dtype=np.uint64)
        G = np.zeros((max(length, 1), PW), dtype=np.uint64)
        H = np.zeros((max(length, 1), PW), dtype=np.uint64)
        check(_lib.lib().bpp_pk_hashed(arith.handle, bytes(label), len(label), length, _ptr(self.gh), _ptr(G), _ptr(H)),
              "bpp_pk_hashed")
        self.G_vec, self.H_vec = G[:length], H[:length]
        return self

    @classmethod
    def from_points(cls, arith: Arith, gh, G_vec, H_vec) -> "PublicKey":
        """An arbitrary generator set (the reference only has `new`; used for the 'hard' distribution)."""
        self = cls.__new__(cls)
        self.arith = arith
        self.gh = np.ascontiguousarray(gh, dtype=np.uint64).reshape(2, arith.PW)
        self.G_vec = np.ascontiguousarray(G_vec, dtype=np.uint64).reshape(-1, arith.PW)
        self.H_vec = np.ascontiguousarray(H_vec, dtype=np.uint64).reshape(-1, arith.PW)
        return self

    @property
    def g(self):
        return self.gh[0]

    @property
    def h(self):
        return self.gh[1]

    def commitment(self, v, gamma) -> np.ndarray:
        """g * v + h * gamma (publickey.rs:50-52); v, gamma scalars."""
        mv = MulVec(self.arith)
        mv.add_scalar(v)
        mv.add_scalar(gamma)
        mv.add_point(self.g)
        mv.add_point(self.h)
        return mv.calculate()


class RangeProver:
    """reference range/prover.rs:13-42."""

    def __init__(self):
        self.v_vec = []
        self.gamma_vec = []
        self.commitment_vec = []

    @classmethod
    def new(cls):
        return cls()

    def commit(self, pk: PublicKey, v: int, gamma):
        g = scalar_to_wire(gamma)
        out = np.zeros(pk.arith.PW, dtype=np.uint64)
        check(_lib.lib().bpp_commit(pk.arith.handle, _ptr(pk.gh), ctypes.c_uint64(v), _ptr(g), _ptr(out)), "bpp_commit")
        self.v_vec.append(int(v))
        self.gamma_vec.append(g)
        self.commitment_vec.append(out)


class RangeVerifier:
    """The verifier-side holder of the commitments.  It exists only in the reference's (stale) README
    (README.md:47-55: ``RangeVerifier::new()``, ``allocate(&prover.commitment_vec)``, ``proof.verify(.., &verifier)``);
    the code takes the commitment slice directly (range/mod.rs:57-62).  ``RangeProof.verify`` accepts either."""

    def __init__(self):
        self.commitment_vec = []

    @classmethod
    def new(cls):
        return cls()

    def allocate(self, commitment_vec):
        self.commitment_vec = [np.ascontiguousarray(c, dtype=np.uint64) for c in commitment_vec]


class WeightedInnerProductProof:
    """Field holder, reference weighted_inner_product_proof.rs:25-33."""

    def __init__(self, L_vec, R_vec, A, B, r_prime, s_prime, d_prime):
        self.L_vec, self.R_vec, self.A, self.B = L_vec, R_vec, A, B
        self.r_prime, self.s_prime, self.d_prime = r_prime, s_prime, d_prime


class RangeProof:
    """reference range/mod.rs:25-78: struct RangeProof { A, proof }."""

    def __init__(self, A, proof: WeightedInnerProductProof):
        self.A = A
        self.proof = proof

    # wire record used by the C ABI: points [A, wip.A, wip.B, L.., R..], scalars [r', s', d']
    def points_wire(self) -> np.ndarray:
        p = self.proof
        return np.concatenate([np.stack([self.A, p.A, p.B]), np.asarray(p.L_vec), np.asarray(p.R_vec)]).astype(np.uint64)

    def scalars_wire(self) -> np.ndarray:
        p = self.proof
        return np.stack([scalar_to_wire(p.r_prime), scalar_to_wire(p.s_prime), scalar_to_wire(p.d_prime)])

    @classmethod
    def from_wire(cls, points, scalars) -> "RangeProof":
        points = np.asarray(points, dtype=np.uint64)
        k = (points.shape[0] - 3) // 2
        sc = np.asarray(scalars, dtype=np.uint64).reshape(3, 4)
        return cls(points[0], WeightedInnerProductProof(points[3:3 + k], points[3 + k:3 + 2 * k], points[1], points[2],
                                                        sc[0], sc[1], sc[2]))

    @classmethod
    def prove(cls, pk: PublicKey, n: int, prover: RangeProver) -> "RangeProof":
        a = pk.arith
        m = len(prover.v_vec)
        mn = n * m
        if m == 0 or mn & (mn - 1):
            raise AssertionError("n * m must be a power of two")      # wip.rs:67 assert
        if len(pk.G_vec) != mn or len(pk.H_vec) != mn:
            raise AssertionError("pk must hold n*m generators")       # range/mod.rs:90-91,252-253 assert_eq
        k = mn.bit_length() - 1
        v = np.array(prover.v_vec, dtype=np.uint64)
        gm = np.stack(prover.gamma_vec)
        V = np.stack(prover.commitment_vec)
        pts = np.zeros((3 + 2 * k, a.PW), dtype=np.uint64)
        sc = np.zeros((3, 4), dtype=np.uint64)
        G = np.ascontiguousarray(pk.G_vec)
        H = np.ascontiguousarray(pk.H_vec)
        check(_lib.lib().bpp_range_prove(a.handle, _ptr(pk.gh), _ptr(G), _ptr(H), n, m, _ptr(v), _ptr(gm), _ptr(V),
                                         _ptr(pts), _ptr(sc)), "bpp_range_prove")
        return cls.from_wire(pts, sc)

    def verify(self, pk: PublicKey, n: int, commitment_vec) -> None:
        """Returns None for Ok(()); raises VerificationError for Err(ProofError::VerificationError)."""
        a = pk.arith
        if isinstance(commitment_vec, RangeVerifier):   # the README's calling convention
            commitment_vec = commitment_vec.commitment_vec
        V = np.ascontiguousarray(np.asarray(commitment_vec, dtype=np.uint64).reshape(-1, a.PW))
        m = V.shape[0]
        if len(pk.G_vec) != n * m or len(pk.H_vec) != n * m:
            # the reference indexes pk.G_vec / H_vec with bounds-checked slices and panics (mulvec.rs:23-25)
            raise AssertionError("pk must hold n*m generators")
        pts = np.ascontiguousarray(self.points_wire())
        sc = np.ascontiguousarray(self.scalars_wire())
        k = (pts.shape[0] - 3) // 2
        G = np.ascontiguousarray(pk.G_vec)
        H = np.ascontiguousarray(pk.H_vec)
        rc = check(_lib.lib().bpp_range_verify(a.handle, _ptr(pk.gh), _ptr(G), _ptr(H), n, m, _ptr(pts), k, _ptr(sc),
                                               _ptr(V)), "bpp_range_verify")
        if rc != 0:
            raise VerificationError("VerificationError")


class BatchVerifier:
    """Device-resident batch verification of independent proofs for one (pk, n, m).

    ``verify_wire`` takes host arrays; ``run_device`` takes raw device pointers (e.g. torch tensors'
    ``data_ptr()``) and is the call the benchmark times."""

    def __init__(self, pk: PublicKey, n: int, m: int, window_bits: int = 13):
        self.arith = pk.arith
        self.n, self.m = n, m
        if len(pk.G_vec) != n * m or len(pk.H_vec) != n * m:
            raise AssertionError("pk must hold n*m generators")       # range/mod.rs:90-91,252-253 assert_eq
        h = ctypes.c_void_p()
        G = np.ascontiguousarray(pk.G_vec)
        H = np.ascontiguousarray(pk.H_vec)
        check(_lib.lib().bpp_verifier_create(pk.arith.handle, _ptr(pk.gh), _ptr(G), _ptr(H), n, m, window_bits,
                                             ctypes.byref(h)), "bpp_verifier_create")
        self.handle = h
        self.msm_len = _lib.lib().bpp_verifier_msm_len(h)
        self.table_bytes = _lib.lib().bpp_verifier_table_bytes(h)
        mn = n * m
        self.k = mn.bit_length() - 1
        self.points_per_proof = 3 + 2 * self.k + m

    def close(self):
        if self.handle:
            _lib.lib().bpp_verifier_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_bytes(self, count: int) -> int:
        return _lib.lib().bpp_verifier_workspace_bytes(self.handle, count)

    def verify_wire(self, points, scalars) -> np.ndarray:
        """points (count, 3+2k+m, PW) [A, wip.A, wip.B, L.., R.., V..], scalars (count, 3, 4) -> ok (count,) u32"""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, self.points_per_proof, self.arith.PW)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 3, 4)
        count = pts.shape[0]
        if sc.shape[0] != count:
            raise RuntimeError("verify_wire: one scalar triple per proof record")
        ok = np.zeros(count, dtype=np.uint32)
        check(_lib.lib().bpp_range_verify_batch(self.handle, _ptr(pts), _ptr(sc), count, _ptr(ok)),
              "bpp_range_verify_batch")
        return ok

    def verify_compressed(self, records, scalars) -> np.ndarray:
        """records (count, 3+2k+m, compressed_bytes) u8 in the order of verify_wire, scalars (count, 3, 4) -> ok"""
        cb = compressed_bytes(self.arith)
        rec = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, self.points_per_proof, cb)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 3, 4)
        count = rec.shape[0]
        if sc.shape[0] != count:
            raise RuntimeError("verify_compressed: one scalar triple per proof record")
        ok = np.zeros(count, dtype=np.uint32)
        check(_lib.lib().bpp_range_verify_batch_compressed(self.handle, _ptr(rec), _ptr(sc), count, _ptr(ok)),
              "bpp_range_verify_batch_compressed")
        return ok

    def verify_serialized(self, proofs, commitments, transcript: bool = False, uncompressed: bool = False) -> np.ndarray:
        """proofs (count, proof_bytes) u8, commitments (count, m, compressed_bytes) u8 -> status (count,) u32:
        0 Ok / 1 VerificationError / 2 FormatError.  uncompressed: container version 2 (and uncompressed commitments)"""
        pb = proof_bytes(self.arith, self.n, self.m, 2 if uncompressed else 1)
        raw = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(-1, pb)
        cm = np.ascontiguousarray(commitments, dtype=np.uint8).reshape(
            -1, self.m, uncompressed_bytes(self.arith) if uncompressed else compressed_bytes(self.arith))
        if cm.shape[0] != raw.shape[0]:
            raise RuntimeError("verify_serialized: m commitments per proof")
        ok = np.zeros(raw.shape[0], dtype=np.uint32)
        check(_lib.lib().bpp_range_verify_batch_serialized(self.handle, _ptr(raw), _ptr(cm), raw.shape[0],
                                                           (1 if transcript else 0) | (2 if uncompressed else 0), _ptr(ok)),
              "bpp_range_verify_batch_serialized")
        return ok

    def serialized_workspace_bytes(self, count: int) -> int:
        return _lib.lib().bpp_verifier_serialized_workspace_bytes(self.handle, count)

    def verify_serialized_device(self, d_proofs: int, d_commitments: int, count: int, d_ok: int, d_workspace: int,
                                 workspace_bytes: int, stream: int = 0, transcript: bool = False, uncompressed: bool = False):
        """verify_serialized with every buffer in HBM (raw device pointers), asynchronous on `stream`: containers and
        compressed (version 2: uncompressed) commitments in, per-proof status words (0 / 1 / 2) out"""
        check(_lib.lib().bpp_range_verify_batch_serialized_device(self.handle, d_proofs, d_commitments, count,
                                                                  (1 if transcript else 0) | (2 if uncompressed else 0), d_ok, d_workspace,
                                                                  workspace_bytes, stream or None),
              "bpp_range_verify_batch_serialized_device")

    def serialized_grouped_workspace_bytes(self, count: int, group: int = 32) -> int:
        return _lib.lib().bpp_verifier_serialized_grouped_workspace_bytes(self.handle, count, group)

    def verify_serialized_grouped_device(self, d_proofs: int, d_commitments: int, count: int, d_ok: int, d_workspace: int,
                                         workspace_bytes: int, weight_key: bytes = None, index_base: int = 0, group: int = 32,
                                         stream: int = 0, transcript: bool = False, uncompressed: bool = False):
        """verify_serialized_device with the grouped check behind the decoder (include/bpp_amd.h): the same status words at the
        grouped check's price when (nearly) every proof is valid.  weight_key: 32 secret bytes, None = os.urandom(32).
        Synchronises the stream.  -> (groups that failed, proofs re-verified exactly)"""
        if weight_key is None:
            import os
            weight_key = os.urandom(32)
        key = bytes(weight_key)
        if len(key) != 32:
            raise ValueError("weight_key must be 32 bytes")
        stats = (ctypes.c_uint64 * 2)()
        check(_lib.lib().bpp_range_verify_batch_serialized_grouped_device(
            self.handle, d_proofs, d_commitments, count, (1 if transcript else 0) | (2 if uncompressed else 0), key,
            ctypes.c_uint64(index_base), group, d_ok, stats, d_workspace, workspace_bytes, stream or None),
            "bpp_range_verify_batch_serialized_grouped_device")
        return int(stats[0]), int(stats[1])

    def run_device(self, d_points: int, d_scalars: int, count: int, d_ok: int, d_workspace: int, workspace_bytes: int,
                   stream: int = 0, d_challenges: int = 0, d_out_scalars: int = 0, d_out_result: int = 0):
        check(_lib.lib().bpp_verifier_run(self.handle, d_points, d_scalars, count, d_challenges or None, d_ok,
                                          d_workspace, workspace_bytes, d_out_scalars or None, d_out_result or None,
                                          stream or None), "bpp_verifier_run")


class PassGraph:
    """One pass of the batch verifier captured into a HIP graph (bpp_verifier_graph_capture): launch() replays it over the
    buffers it was captured with.  Keep the verifier alive while its graphs are."""

    def __init__(self, handle):
        self.handle = handle

    def launch(self, stream: int = 0):
        check(_lib.lib().bpp_graph_launch(self.handle, stream or None), "bpp_graph_launch")

    def close(self):
        if self.handle:
            _lib.lib().bpp_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _verifier_graph_capture(self, d_points: int, d_scalars: int, count: int, d_ok: int, d_workspace: int, workspace_bytes: int,
                            d_challenges: int = 0) -> PassGraph:
    h = ctypes.c_void_p()
    check(_lib.lib().bpp_verifier_graph_capture(self.handle, d_points, d_scalars, count, d_challenges or None, d_ok, d_workspace,
                                                workspace_bytes, ctypes.byref(h)), "bpp_verifier_graph_capture")
    return PassGraph(h)


STAGES = ("from_wire", "verify_scalars", "fixed_msm", "var_msm", "finalize")


def _verifier_set_profiling(self, on: bool):
    check(_lib.lib().bpp_verifier_set_profiling(self.handle, 1 if on else 0), "bpp_verifier_set_profiling")


def _verifier_profile(self):
    """-> ({stage: mean ms}, passes, blocks_per_proof of k_fixed_msm), HIP events on the launch stream"""
    ms = (ctypes.c_float * 5)()
    passes = ctypes.c_size_t()
    bpp_ = ctypes.c_uint()
    check(_lib.lib().bpp_verifier_profile(self.handle, ms, ctypes.byref(passes), ctypes.byref(bpp_)),
          "bpp_verifier_profile")
    return {k: float(ms[i]) for i, k in enumerate(STAGES)}, passes.value, bpp_.value


def _verifier_partial_bytes(self) -> int:
    return _lib.lib().bpp_verifier_partial_bytes(self.handle)


def _verifier_combined_workspace_bytes(self, count: int) -> int:
    return _lib.lib().bpp_verifier_combined_workspace_bytes(self.handle, count)


def _verifier_run_combined_device(self, d_points: int, d_scalars: int, count: int, weight_key, index_base: int,
                                  d_out_partial: int, d_ok: int, d_workspace: int, workspace_bytes: int, stream: int = 0,
                                  d_challenges: int = 0, d_weights: int = 0):
    """Combined batch check (NOT the reference's per-proof semantics, see include/bpp_amd.h): one weighted
    sum of the batch's verification MulVecs.  d_ok[0] == 0 iff it is the identity.
    weight_key: 32 secret bytes (None = drawn here from os.urandom, fresh per call) expanded on the device by a
    SHA-256 PRF over the global proof index index_base + p; or d_weights: count x 16 bytes on the device."""
    if d_weights:
        key = None
    else:
        if weight_key is None:
            import os
            weight_key = os.urandom(32)
        key = bytes(weight_key)
        if len(key) != 32:
            raise ValueError("weight_key must be 32 bytes")
    check(_lib.lib().bpp_verifier_run_combined(self.handle, d_points, d_scalars, count, d_challenges or None, key,
                                               ctypes.c_uint64(index_base), d_weights or None, d_out_partial, d_ok,
                                               d_workspace, workspace_bytes, stream or None),
          "bpp_verifier_run_combined")


def _verifier_grouped_workspace_bytes(self, count: int, group: int = 32) -> int:
    return _lib.lib().bpp_verifier_grouped_workspace_bytes(self.handle, count, group)


def _verifier_run_grouped_device(self, d_points: int, d_scalars: int, count: int, weight_key, index_base: int,
                                 d_out_verdicts: int, d_workspace: int, workspace_bytes: int, group: int = 32,
                                 stream: int = 0, d_challenges: int = 0, d_weights: int = 0):
    """Per-proof verdicts (the vector run_device writes) from one weighted check per group of `group` neighbouring proofs
    and an exact pass over the proofs of the groups that fail (include/bpp_amd.h "grouped check"; an engine mode, not a
    reference path).  weight_key / d_weights as for run_combined_device.  Synchronises the stream.
    Returns (groups that failed, proofs re-verified exactly)."""
    if d_weights:
        key = None
    else:
        if weight_key is None:
            import os
            weight_key = os.urandom(32)
        key = bytes(weight_key)
        if len(key) != 32:
            raise ValueError("weight_key must be 32 bytes")
    stats = (ctypes.c_uint64 * 2)()
    check(_lib.lib().bpp_verifier_run_grouped(self.handle, d_points, d_scalars, count, d_challenges or None, key,
                                              ctypes.c_uint64(index_base), d_weights or None, group, d_out_verdicts, stats,
                                              d_workspace, workspace_bytes, stream or None),
          "bpp_verifier_run_grouped")
    return int(stats[0]), int(stats[1])


def _verifier_grouped_begin_device(self, d_points: int, d_scalars: int, count: int, weight_key, index_base: int,
                                   d_out_verdicts: int, d_workspace: int, workspace_bytes: int, group: int = 32, stream: int = 0,
                                   d_challenges: int = 0, d_weights: int = 0):
    """First half of run_grouped_device: enqueues the weighted checks of the groups and returns (nothing synchronises)."""
    key = None
    if not d_weights:
        if weight_key is None:
            import os
            weight_key = os.urandom(32)
        key = bytes(weight_key)
        if len(key) != 32:
            raise ValueError("weight_key must be 32 bytes")
    check(_lib.lib().bpp_verifier_grouped_begin(self.handle, d_points, d_scalars, count, d_challenges or None, key,
                                                ctypes.c_uint64(index_base), d_weights or None, group, d_out_verdicts,
                                                d_workspace, workspace_bytes, stream or None), "bpp_verifier_grouped_begin")


def _verifier_grouped_finish_device(self, d_points: int, d_scalars: int, count: int, d_out_verdicts: int, d_workspace: int,
                                    workspace_bytes: int, group: int = 32, stream: int = 0, d_challenges: int = 0):
    """Second half: same buffers, count, group and stream as the begin it completes; synchronises the stream.
    -> (groups that failed, proofs re-verified exactly)"""
    stats = (ctypes.c_uint64 * 2)()
    check(_lib.lib().bpp_verifier_grouped_finish(self.handle, d_points, d_scalars, count, d_challenges or None, group,
                                                 d_out_verdicts, stats, d_workspace, workspace_bytes, stream or None),
          "bpp_verifier_grouped_finish")
    return int(stats[0]), int(stats[1])


def _verifier_derive_challenges_device(self, d_points: int, count: int, d_challenges: int, stream: int = 0):
    """Fiat-Shamir challenges [y, z, e, e_1..e_k] of every proof record of a resident batch (csrc/transcript.hpp),
    in the layout run_device takes as d_challenges.  The reference has no transcript: parity unpinned."""
    check(_lib.lib().bpp_verifier_derive_challenges(self.handle, d_points, count, d_challenges, stream or None),
          "bpp_verifier_derive_challenges")


def _verifier_sum_partials_device(self, d_partials: int, n: int, d_ok: int, stream: int = 0):
    check(_lib.lib().bpp_verifier_sum_partials(self.handle, d_partials, n, d_ok, stream or None),
          "bpp_verifier_sum_partials")


def _engine_prove_batch(self, values, gammas, transcript: bool = False, blind_key: bytes = None, index_base: int = 0):
    """RangeProof::prove + RangeProver::commit for `count` provers sharing this engine's (pk, n, m).
    transcript=True: challenges from the Fiat-Shamir transcript (csrc/transcript.hpp) instead of the reference's
    constants -- not a reference code path, parity unpinned.  blind_key (32 secret bytes, transcript mode only): the
    blinding values come from this key (include/bpp_amd.h "Blinding"); None = the reference's literals, which hide nothing.
    values: (count, m) ints < 2^64 ; gammas: (count, m) scalars (ints or (count, m, 4) uint64).
    Returns (points (count, 3+2k, PW), scalars (count, 3, 4), V (count, m, PW)) in wire format --
    bit-identical to RangeProof.prove / RangeProver.commit one by one."""
    vals = np.ascontiguousarray(np.asarray(values, dtype=np.uint64).reshape(-1, self.m))
    count = vals.shape[0]
    if isinstance(gammas, np.ndarray) and gammas.dtype == np.uint64 and gammas.ndim == 3:
        gm = np.ascontiguousarray(gammas)
    else:
        gm = np.zeros((count, self.m, 4), dtype=np.uint64)
        for i, row in enumerate(gammas):
            for j, g in enumerate(row):
                gm[i, j] = scalar_to_wire(g)
    PW = self.arith.PW
    pts = np.zeros((count, 3 + 2 * self.k, PW), dtype=np.uint64)
    sc = np.zeros((count, 3, 4), dtype=np.uint64)
    V = np.zeros((count, self.m, PW), dtype=np.uint64)
    if blind_key is not None and (not transcript or len(blind_key) != 32):
        raise ValueError("blind_key: 32 bytes, transcript mode only")
    if transcript:
        check(_lib.lib().bpp_range_prove_batch_fs(self.handle, _ptr(vals), _ptr(gm), count, blind_key, index_base, _ptr(pts),
                                                  _ptr(sc), _ptr(V)), "bpp_range_prove_batch_fs")
    else:
        check(_lib.lib().bpp_range_prove_batch(self.handle, _ptr(vals), _ptr(gm), count, _ptr(pts), _ptr(sc), _ptr(V)),
              "bpp_range_prove_batch")
    return pts, sc, V


def _engine_prover_workspace_bytes(self, count: int) -> int:
    return _lib.lib().bpp_prover_workspace_bytes(self.handle, count)


def _engine_prove_batch_device(self, d_values: int, d_gammas: int, count: int, d_out_points: int, d_out_scalars: int,
                               d_out_V: int, d_workspace: int, workspace_bytes: int, stream: int = 0,
                               transcript: bool = False, d_out_challenges: int = 0, blind_key: bytes = None,
                               index_base: int = 0, d_blinding: int = 0):
    """prove_batch with every buffer in HBM (raw device pointers), asynchronous on `stream`.  Transcript mode: blinding
    from blind_key (32 bytes) / d_blinding (count x (5 + 2k) scalars on the device), else the reference's literals."""
    if transcript:
        check(_lib.lib().bpp_range_prove_batch_fs_device(self.handle, d_values, d_gammas, count, blind_key, index_base,
                                                         d_blinding or None, d_out_points,
                                                         d_out_scalars, d_out_V or None, d_out_challenges or None,
                                                         d_workspace, workspace_bytes, stream or None),
              "bpp_range_prove_batch_fs_device")
        return
    check(_lib.lib().bpp_range_prove_batch_device(self.handle, d_values, d_gammas, count, d_out_points, d_out_scalars,
                                                  d_out_V or None, d_workspace, workspace_bytes, stream or None),
          "bpp_range_prove_batch_device")


BatchVerifier.prove_batch = _engine_prove_batch
BatchVerifier.prover_workspace_bytes = _engine_prover_workspace_bytes
BatchVerifier.prove_batch_device = _engine_prove_batch_device
BatchVerifier.partial_bytes = _verifier_partial_bytes
BatchVerifier.combined_workspace_bytes = _verifier_combined_workspace_bytes
BatchVerifier.run_combined_device = _verifier_run_combined_device
BatchVerifier.grouped_workspace_bytes = _verifier_grouped_workspace_bytes
BatchVerifier.run_grouped_device = _verifier_run_grouped_device
BatchVerifier.grouped_begin_device = _verifier_grouped_begin_device
BatchVerifier.grouped_finish_device = _verifier_grouped_finish_device
BatchVerifier.derive_challenges_device = _verifier_derive_challenges_device
BatchVerifier.sum_partials_device = _verifier_sum_partials_device
def _verifier_set_subgroup_check(self, on: bool):
    """wire points outside the prime-order subgroup count as invalid points (include/bpp_amd.h); off by default"""
    check(_lib.lib().bpp_verifier_set_subgroup_check(self.handle, 1 if on else 0), "bpp_verifier_set_subgroup_check")


BatchVerifier.set_subgroup_check = _verifier_set_subgroup_check
BatchVerifier.graph_capture = _verifier_graph_capture
BatchVerifier.set_profiling = _verifier_set_profiling
BatchVerifier.profile = _verifier_profile


def proof_record(proof: RangeProof, commitment_vec) -> np.ndarray:
    """[A, wip.A, wip.B, L.., R.., V..] -- the per-proof point record of the batch verifier."""
    return np.concatenate([proof.points_wire(), np.asarray(commitment_vec, dtype=np.uint64)])
