#!/usr/bin/env python3
"""Grouped check from one host thread vs two (each with its own stream, workspace and verdict buffer): how much of a pass
is latency that a second pass in flight would hide?  usage: python tools/grouped_pipeline_probe.py [--steps 40]"""
import argparse
import json
import os
import sys
import threading
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
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--group", type=int, default=32)
    args = ap.parse_args()
    n, m, Bsz = 64, 16, args.batch
    a = B.Arith.init("bls12_381")
    bv = B.BatchVerifier(B.PublicKey.new(a, n * m), n, m, window_bits=args.window)
    D = 256
    vals = [[(7919 * (d + 1) + j) % (1 << 31) for j in range(m)] for d in range(D)]
    gams = [[3 + d + j for j in range(m)] for d in range(D)]
    pts, scs, V = bv.prove_batch(vals, gams)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1)[np.arange(Bsz) % D])
    scs = np.ascontiguousarray(scs[np.arange(Bsz) % D])
    dev = torch.device("cuda:0")
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(scs.view(np.int64)).to(dev)
    wsb = bv.grouped_workspace_bytes(Bsz, args.group)
    key = os.urandom(32)
    out = {}
    for threads in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(threads)]
        wss = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(threads)]
        oks = [torch.full((Bsz,), 7, dtype=torch.int32, device=dev) for _ in range(threads)]
        torch.cuda.synchronize()

        def worker(k, reps):
            for _ in range(reps):
                bv.run_grouped_device(d_pts.data_ptr(), d_sc.data_ptr(), Bsz, key, 0, oks[k].data_ptr(), wss[k].data_ptr(), wsb,
                                      group=args.group, stream=streams[k].cuda_stream)
        for k in range(threads):
            worker(k, 1)
        torch.cuda.synchronize()
        per = args.steps // threads
        ts = [threading.Thread(target=worker, args=(k, per)) for k in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert all(int(o.sum().item()) == 0 for o in oks)
        out["threads=%d" % threads] = {"ms_per_pass": dt / (per * threads) * 1e3, "verifies_per_s": Bsz * per * threads / dt}
        del wss, oks
    print(json.dumps(out))


if __name__ == "__main__":
    main()
