"""bulletproofsplus_amd -- MI355X-native engine for the MSM + weighted-inner-product hot path of
gogoex/BulletProofsPlus.  HIP kernels behind a C ABI (include/bpp_amd.h); this package
is the host-side mirror of the reference's API for that path.  No CPU fallback."""

from .api import (Arith, BatchVerifier, MulVec, ProofError, PublicKey, RangeProof, RangeProver, RangeVerifier,  # noqa: F401
                  VerificationError, WeightedInnerProductProof, compress_points, compressed_bytes,
                  decompress_points, uncompressed_bytes, uncompressed_points, msm_batch, msm_pippenger, msm_device, msm_workspace_bytes, msm_set_profiling, msm_profile, proof_record, FormatError, proof_bytes, encode_proofs,
                  decode_proofs, wip_fold_round)
from ._lib import BLS12_381_G1, SECP256K1, ED25519, CURVE_IDS, BppError  # noqa: F401
