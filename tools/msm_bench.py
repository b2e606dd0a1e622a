#!/usr/bin/env python3
"""Times MulVec::calculate through the device-resident bucket pipeline (bpp_msm_device, csrc/pippenger.hpp) on large
variable-base inputs: scalars and points resident in HBM, HIP events on the launch stream.
usage: python tools/msm_bench.py [--curve bls12_381] [--log2n 16 18 20 22] [--windows 0 13 14] [--reps 5] [--check]
Points are distinct multiples of g (produced on the GPU: a 64-bit base set expanded by additions would be cheaper, but
k_scalar_mul is fast enough), scalars are uniformly random below 2^252; --check verifies the result against the known
discrete logs: sum_i s_i (k_i g) == (sum_i s_i k_i) g."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bulletproofsplus_amd as B  # noqa: E402

R = {"bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
     "secp256k1": 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
     "ed25519": (1 << 252) + 27742317777372353535851937790883648493}


def make_inputs(a, n, seed=1):
    """-> (scalars (n,4) u64, points (n,PW) u64, ks (n,) u64 with points[i] = ks[i] g)"""
    g = B.PublicKey.new(a, 0).gh[0]
    rng = np.random.RandomState(seed)
    ks = rng.randint(1, 2**62, size=n).astype(np.uint64)
    kw = np.zeros((n, 4), dtype=np.uint64)
    kw[:, 0] = ks
    pts = np.zeros((n, a.PW), dtype=np.uint64)
    step = 1 << 18
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        pts[lo:hi] = a.scalar_mul(kw[lo:hi], np.broadcast_to(g, (hi - lo, a.PW)).copy())
    sc = rng.randint(0, 2**63 - 1, size=(n, 4)).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=(n, 4)).astype(np.uint64)
    sc[:, 3] >>= np.uint64(4)
    return sc, pts, ks, g


def expected(a, sc, ks, g, r):
    # sum s_i k_i mod r with 64-bit limbs of s against the 62-bit k: python big ints over numpy object arrays
    s = sc[:, 0].astype(object) + (sc[:, 1].astype(object) << 64) + (sc[:, 2].astype(object) << 128) + (sc[:, 3].astype(object) << 192)
    tot = int((s * ks.astype(object)).sum() % r)
    w = np.array([[(tot >> (64 * t)) & 0xFFFFFFFFFFFFFFFF for t in range(4)]], dtype=np.uint64)
    return a.scalar_mul(w, g[None])[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--log2n", type=int, nargs="+", default=[16, 18, 20])
    ap.add_argument("--windows", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--equal-scalars", action="store_true",
                    help="the adversarial input of a bucket method: every scalar the same, so each window has ONE bucket holding all points")
    args = ap.parse_args()
    a = B.Arith.init(args.curve)
    dev = torch.device("cuda:0")
    for lg in args.log2n:
        n = 1 << lg
        sc, pts, ks, g = make_inputs(a, n)
        if args.equal_scalars:
            sc = np.broadcast_to(sc[3], sc.shape).copy()
        d_sc = torch.from_numpy(sc.view(np.int64)).to(dev)
        d_pt = torch.from_numpy(pts.view(np.int64)).to(dev)
        d_out = torch.zeros(a.PW, dtype=torch.int64, device=dev)
        exp = expected(a, sc, ks, g, R[args.curve]) if args.check else None
        for c in args.windows:
            wsb = B.msm_workspace_bytes(a, n, c)
            d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def run():
                B.msm_device(a, d_sc.data_ptr(), d_pt.data_ptr(), n, d_out.data_ptr(), d_ws.data_ptr(), wsb,
                             window_bits=c, stream=st)
            run()
            torch.cuda.synchronize()
            if exp is not None:
                assert np.array_equal(d_out.cpu().numpy().view(np.uint64), exp), "MSM result mismatch at n=2^%d c=%d" % (lg, c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print(json.dumps({"curve": args.curve, "log2n": lg, "window_bits": c, "ms": round(ms, 4),
                              "points_per_s": round(n / ms * 1e3), "workspace_MB": round(wsb / 1e6, 1),
                              "checked": exp is not None, "equal_scalars": bool(args.equal_scalars)}), flush=True)
            del d_ws


if __name__ == "__main__":
    main()
