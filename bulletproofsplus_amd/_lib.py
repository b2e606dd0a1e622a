"""ctypes binding of libbpp_amd.so (the C ABI of include/bpp_amd.h).

There is no fallback of any kind: if the shared library is missing or a HIP call fails, the import or
the call raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C bulletproofsplus_amd/csrc``.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BPP_AMD_LIB selects an alternative build of the same library (A/B experiments); default is the in-tree one
LIB_PATH = os.environ.get("BPP_AMD_LIB") or os.path.join(_HERE, "libbpp_amd.so")

BLS12_381_G1 = 0
SECP256K1 = 1
ED25519 = 2   # edwards25519 / "Ristretto-class": no reference counterpart (parity unpinned, csrc/ed25519.hpp)
CURVE_IDS = {"bls12_381": BLS12_381_G1, "secp256k1": SECP256K1, "ed25519": ED25519}
FP_LIMBS = {BLS12_381_G1: 6, SECP256K1: 4, ED25519: 4}

OK = 0
VERIFICATION_ERROR = 1
FORMAT_ERROR = 2   # ProofError::FormatError (reference src/errors.rs:20)

# every symbol include/bpp_amd.h declares (tests check that the library exports all of them)
EXPORTS = [
    "bpp_init", "bpp_destroy", "bpp_last_error", "bpp_point_words", "bpp_msm", "bpp_msm_batch", "bpp_msm_pippenger",
    "bpp_msm_workspace_bytes", "bpp_msm_device", "bpp_msm_set_profiling", "bpp_msm_profile",
    "bpp_scalar_mul_batch", "bpp_pk_new", "bpp_pk_hashed", "bpp_commit", "bpp_range_prove", "bpp_range_prove_batch", "bpp_range_verify", "bpp_set_verify_cache", "bpp_wip_fold_round",
    "bpp_prover_workspace_bytes", "bpp_range_prove_batch_device",
    "bpp_verifier_create", "bpp_verifier_destroy", "bpp_verifier_workspace_bytes", "bpp_verifier_msm_len",
    "bpp_verifier_table_bytes", "bpp_verifier_run", "bpp_verifier_graph_capture", "bpp_graph_launch", "bpp_graph_destroy", "bpp_range_verify_batch", "bpp_verifier_dominant_kernel",
    "bpp_verifier_set_profiling", "bpp_verifier_profile", "bpp_verifier_set_subgroup_check", "bpp_verifier_partial_bytes",
    "bpp_verifier_combined_workspace_bytes", "bpp_verifier_run_combined", "bpp_verifier_sum_partials",
    "bpp_verifier_grouped_workspace_bytes", "bpp_verifier_run_grouped", "bpp_verifier_grouped_begin", "bpp_verifier_grouped_finish",
    "bpp_verifier_derive_challenges", "bpp_range_prove_batch_fs", "bpp_range_prove_batch_fs_device",
    "bpp_point_compressed_bytes", "bpp_points_compress", "bpp_points_decompress", "bpp_points_decompress_device",
    "bpp_range_verify_batch_compressed", "bpp_proof_bytes", "bpp_proofs_encode", "bpp_proofs_decode", "bpp_point_uncompressed_bytes", "bpp_points_uncompressed",
    "bpp_proof_bytes_version", "bpp_proofs_encode_version",
    "bpp_range_verify_batch_serialized", "bpp_verifier_serialized_workspace_bytes",
    "bpp_range_verify_batch_serialized_device",
    "bpp_verifier_serialized_grouped_workspace_bytes", "bpp_range_verify_batch_serialized_grouped_device",
]


class BppError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().bpp_last_error().decode(errors="replace") if _lib is not None else ""
        super().__init__("%s failed with code %d: %s" % (where, code, msg))


_lib = None


def lib():
    """Loads libbpp_amd.so.  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libbpp_amd.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C bulletproofsplus_amd/csrc -j8` (the engine has no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, i32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        L.bpp_init.argtypes = [i32, i32, ctypes.POINTER(vp)]
        L.bpp_destroy.argtypes = [vp]
        L.bpp_destroy.restype = None
        L.bpp_last_error.restype = ctypes.c_char_p
        L.bpp_point_words.argtypes = [i32]
        L.bpp_msm.argtypes = [vp, vp, vp, sz, vp]
        L.bpp_msm_batch.argtypes = [vp, vp, vp, vp, sz, vp]
        L.bpp_msm_pippenger.argtypes = [vp, vp, vp, sz, i32, vp]
        L.bpp_msm_workspace_bytes.argtypes = [vp, sz, i32]
        L.bpp_msm_workspace_bytes.restype = sz
        L.bpp_msm_device.argtypes = [vp, vp, vp, sz, i32, vp, vp, vp, sz, vp]
        L.bpp_msm_set_profiling.argtypes = [vp, i32]
        L.bpp_msm_profile.argtypes = [vp, vp, vp, vp]
        L.bpp_scalar_mul_batch.argtypes = [vp, vp, vp, sz, vp]
        L.bpp_pk_new.argtypes = [vp, sz, vp, vp, vp]
        L.bpp_pk_hashed.argtypes = [vp, ctypes.c_char_p, sz, sz, vp, vp, vp]
        L.bpp_commit.argtypes = [vp, vp, u64, vp, vp]
        L.bpp_range_prove.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp, vp, vp, vp]
        L.bpp_wip_fold_round.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
        L.bpp_set_verify_cache.argtypes = [vp, i32]
        L.bpp_range_verify.argtypes = [vp, vp, vp, vp, sz, sz, vp, sz, vp, vp]
        L.bpp_range_prove_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp]
        L.bpp_prover_workspace_bytes.argtypes = [vp, sz]
        L.bpp_prover_workspace_bytes.restype = sz
        L.bpp_range_prove_batch_device.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, sz, vp]
        L.bpp_verifier_create.argtypes = [vp, vp, vp, vp, sz, sz, i32, ctypes.POINTER(vp)]
        L.bpp_verifier_destroy.argtypes = [vp]
        L.bpp_verifier_destroy.restype = None
        L.bpp_verifier_workspace_bytes.argtypes = [vp, sz]
        L.bpp_verifier_workspace_bytes.restype = sz
        L.bpp_verifier_msm_len.argtypes = [vp]
        L.bpp_verifier_msm_len.restype = sz
        L.bpp_verifier_table_bytes.argtypes = [vp]
        L.bpp_verifier_table_bytes.restype = sz
        L.bpp_verifier_run.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, vp]
        L.bpp_range_verify_batch.argtypes = [vp, vp, vp, sz, vp]
        L.bpp_verifier_graph_capture.argtypes = [vp, vp, vp, sz, vp, vp, vp, sz, ctypes.POINTER(vp)]
        L.bpp_graph_launch.argtypes = [vp, vp]
        L.bpp_graph_destroy.argtypes = [vp]
        L.bpp_graph_destroy.restype = None
        L.bpp_verifier_dominant_kernel.restype = ctypes.c_char_p
        L.bpp_verifier_set_profiling.argtypes = [vp, i32]
        L.bpp_verifier_set_subgroup_check.argtypes = [vp, i32]
        L.bpp_verifier_profile.argtypes = [vp, vp, vp, vp]
        L.bpp_verifier_partial_bytes.argtypes = [vp]
        L.bpp_verifier_partial_bytes.restype = sz
        L.bpp_verifier_combined_workspace_bytes.argtypes = [vp, sz]
        L.bpp_verifier_combined_workspace_bytes.restype = sz
        L.bpp_verifier_run_combined.argtypes = [vp, vp, vp, sz, vp, ctypes.c_char_p, u64, vp, vp, vp, vp, sz, vp]
        L.bpp_verifier_grouped_workspace_bytes.argtypes = [vp, sz, ctypes.c_uint32]
        L.bpp_verifier_grouped_workspace_bytes.restype = sz
        L.bpp_verifier_run_grouped.argtypes = [vp, vp, vp, sz, vp, ctypes.c_char_p, u64, vp, ctypes.c_uint32, vp, vp, vp, sz, vp]
        L.bpp_verifier_grouped_begin.argtypes = [vp, vp, vp, sz, vp, ctypes.c_char_p, u64, vp, ctypes.c_uint32, vp, vp, sz, vp]
        L.bpp_verifier_grouped_finish.argtypes = [vp, vp, vp, sz, vp, ctypes.c_uint32, vp, vp, vp, sz, vp]
        L.bpp_verifier_derive_challenges.argtypes = [vp, vp, sz, vp, vp]
        L.bpp_range_prove_batch_fs.argtypes = [vp, vp, vp, sz, ctypes.c_char_p, u64, vp, vp, vp]
        L.bpp_range_prove_batch_fs_device.argtypes = [vp, vp, vp, sz, ctypes.c_char_p, u64, vp, vp, vp, vp, vp, vp, sz, vp]
        L.bpp_verifier_sum_partials.argtypes = [vp, vp, sz, vp, vp]
        L.bpp_point_compressed_bytes.argtypes = [i32]
        L.bpp_point_compressed_bytes.restype = sz
        L.bpp_points_compress.argtypes = [vp, vp, sz, vp]
        L.bpp_points_decompress.argtypes = [vp, vp, sz, vp, vp]
        L.bpp_points_decompress_device.argtypes = [vp, vp, sz, vp, vp, i32, vp]
        L.bpp_range_verify_batch_compressed.argtypes = [vp, vp, vp, sz, vp]
        L.bpp_proof_bytes.argtypes = [i32, sz, sz]
        L.bpp_proof_bytes.restype = sz
        L.bpp_proofs_encode.argtypes = [vp, sz, sz, vp, vp, sz, vp]
        L.bpp_point_uncompressed_bytes.argtypes = [i32]
        L.bpp_point_uncompressed_bytes.restype = sz
        L.bpp_points_uncompressed.argtypes = [vp, vp, sz, vp]
        L.bpp_proof_bytes_version.argtypes = [i32, sz, sz, i32]
        L.bpp_proof_bytes_version.restype = sz
        L.bpp_proofs_encode_version.argtypes = [vp, sz, sz, i32, vp, vp, sz, vp]
        L.bpp_proofs_decode.argtypes = [vp, sz, sz, vp, sz, vp, vp, vp]
        L.bpp_range_verify_batch_serialized.argtypes = [vp, vp, vp, sz, i32, vp]
        L.bpp_verifier_serialized_workspace_bytes.argtypes = [vp, sz]
        L.bpp_verifier_serialized_workspace_bytes.restype = sz
        L.bpp_range_verify_batch_serialized_device.argtypes = [vp, vp, vp, sz, i32, vp, vp, sz, vp]
        L.bpp_verifier_serialized_grouped_workspace_bytes.argtypes = [vp, sz, ctypes.c_uint32]
        L.bpp_verifier_serialized_grouped_workspace_bytes.restype = sz
        L.bpp_range_verify_batch_serialized_grouped_device.argtypes = [vp, vp, vp, sz, i32, ctypes.c_char_p, u64, ctypes.c_uint32,
                                                                       vp, vp, vp, sz, vp]
        L.bpp_debug_field_op.argtypes = [vp, i32, i32, vp, vp, sz, vp]
        L.bpp_debug_point_op.argtypes = [vp, i32, vp, vp, sz, vp]
        _lib = L
    return _lib


def check(rc, where):
    if rc < 0:
        raise BppError(rc, where)
    return rc
