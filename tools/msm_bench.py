#!/usr/bin/env python3
"""Times MulVec::calculate through the bucket-method pipeline (csrc/pippenger.hpp) on large variable-base
inputs, device resident.  usage: python tools/msm_bench.py [--curve bls12_381] [--log2n 16 18 20]
Points are distinct multiples of g (produced on the GPU), scalars are SplitMix64-derived full-width values;
the result is checked against the known discrete logs: sum_i s_i (k_i g) == (sum_i s_i k_i) g."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bulletproofsplus_amd as B  # noqa: E402

R = {"bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
     "secp256k1": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
     "ed25519": (1 << 252) + 27742317777372353535851937790883648493}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--log2n", type=int, nargs="+", default=[14, 16, 18, 20])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    a = B.Arith.init(args.curve)
    r = R[args.curve]
    pk = B.PublicKey.new(a, 0)
    g = pk.gh[0]
    rng = np.random.RandomState(1)
    out = []
    for lg in args.log2n:
        n = 1 << lg
        ks = rng.randint(1, 2**62, size=n).astype(np.uint64)
        kw = np.zeros((n, 4), dtype=np.uint64)
        kw[:, 0] = ks
        pts = a.scalar_mul(kw, np.broadcast_to(g, (n, a.PW)).copy())
        sc = rng.randint(0, 2**63 - 1, size=(n, 4)).astype(np.uint64) * 2 + rng.randint(0, 2, size=(n, 4)).astype(np.uint64)
        sc[:, 3] >>= np.uint64(3)        # < 2^253: below every curve's group order
        best = None
        for _ in range(args.reps):
            t0 = time.perf_counter()
            res = B.msm_pippenger(a, sc, pts, 0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        tot = 0
        for i in range(n):
            s = int(sc[i, 0]) | int(sc[i, 1]) << 64 | int(sc[i, 2]) << 128 | int(sc[i, 3]) << 192
            tot = (tot + s * int(ks[i])) % r
        exp = a.scalar_mul([tot], g[None])[0]
        assert np.array_equal(res, exp), "MSM result mismatch at n=2^%d" % lg
        out.append({"log2n": lg, "seconds_incl_pcie": best, "points_per_s": n / best})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
