#!/usr/bin/env python3
"""Does alternating the passes of the batch verifier between two streams (two workspaces) raise the sustained rate?
The tail of one pass's k_fixed_msm launch could take the next pass's scalar / proof-point kernels.
usage: python tools/pipeline_probe.py [--batch 8192] [--window 17] [--steps 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bulletproofsplus_amd as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--window", type=int, default=17)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--m", type=int, default=16)
    args = ap.parse_args()
    n, m, Bsz = args.n, args.m, args.batch
    a = B.Arith.init("bls12_381")
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=args.window)
    D = min(Bsz, 256)
    vals = [[(7919 * (d + 1) + j) % (1 << 31) for j in range(m)] for d in range(D)]
    gams = [[3 + d + j for j in range(m)] for d in range(D)]
    pts, scs, V = bv.prove_batch(vals, gams)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1)[np.arange(Bsz) % D])
    scs = np.ascontiguousarray(scs[np.arange(Bsz) % D])
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(scs.view(np.int64)).to(dev)
    wsb = bv.workspace_bytes(Bsz)
    out = {}
    for lanes in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        wss = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(lanes)]
        oks = [torch.full((Bsz,), 7, dtype=torch.int32, device=dev) for _ in range(lanes)]
        torch.cuda.synchronize()

        def step(i):
            k = i % lanes
            bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), Bsz, oks[k].data_ptr(), wss[k].data_ptr(), wsb, streams[k].cuda_stream)
        for i in range(lanes):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(int(o.sum().item()) == 0 for o in oks)
        out["streams=%d" % lanes] = {"ms_per_step": dt / args.steps * 1e3, "verifies_per_s": Bsz * args.steps / dt}
        del wss, oks
    print(json.dumps(out))


if __name__ == "__main__":
    main()
