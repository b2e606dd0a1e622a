#!/usr/bin/env python3
"""Footprint as a measured knob: verifies/s, table bytes and build seconds of the batch verifier per window width.
usage: python tools/window_sweep.py [--out profiles/r03_window_sweep.json]
(64,16) x 8192 proofs at c = 13..17 and (64,1) x 4096 at c = 12..16, BLS12-381, per-proof verdicts (mode A); the proofs
are 256 distinct GPU-proved ones tiled over the batch (the pass does not depend on their being distinct; the headline's
8192 distinct proofs are bench.py's business).  Also: two verifiers of different shapes co-resident in HBM."""
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


def values(seed, m):
    vs = [((0x9E3779B97F4A7C15 * (j + 1 + seed)) & 0xFFFFFFFFFFFFFFFF) % (1 << 31) for j in range(m)]
    return vs, [j + 3 + seed for j in range(m)]


def run_shape(a, n, m, batch, widths, steps, dev):
    pk = B.PublicKey.new(a, n * m)
    out = []
    recs = scs = None
    for c in widths:
        t0 = time.perf_counter()
        try:
            bv = B.BatchVerifier(pk, n, m, window_bits=c)
        except B.BppError as e:
            out.append({"window_bits": c, "error": "code %d (tables do not fit)" % e.code})
            continue
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        if recs is None:
            D = 256
            vals, gams = zip(*[values(d * 17, m) for d in range(D)])
            pts, sc, V = bv.prove_batch(list(vals), list(gams))
            ix = np.arange(batch) % D
            recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1)[ix])
            scs = np.ascontiguousarray(sc[ix])
        d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
        d_sc = torch.from_numpy(scs.view(np.int64)).to(dev)
        d_ok = torch.full((batch,), 7, dtype=torch.int32, device=dev)
        wsb = bv.workspace_bytes(batch)
        d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), batch, d_ok.data_ptr(), d_ws.data_ptr(), wsb, st)
        torch.cuda.synchronize()
        assert int(d_ok.sum().item()) == 0
        bv.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), batch, d_ok.data_ptr(), d_ws.data_ptr(), wsb, st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stage, _, bpp_ = bv.profile()
        out.append({"window_bits": c, "verifies_per_s": batch * steps / dt, "ms_per_step": dt / steps * 1e3,
                    "table_bytes": bv.table_bytes, "table_GB": round(bv.table_bytes / 1e9, 2), "build_s": round(build_s, 3),
                    "fixed_msm_ms": round(stage["fixed_msm"], 3), "blocks_per_proof": bpp_})
        print(json.dumps({"n": n, "m": m, **out[-1]}), flush=True)
        bv.close()
        del d_ws, d_pts, d_sc
        torch.cuda.empty_cache()
    return out


def coresident(a, dev, steps):
    """C2 (64,16) and C3 (64,1) verifiers at c = 16 side by side (103 + 6 GB of tables), interleaved batches"""
    res = {}
    sh = {"c2": (64, 16, 8192), "c3": (64, 1, 4096)}
    bvs, bufs = {}, {}
    for name, (n, m, batch) in sh.items():
        pk = B.PublicKey.new(a, n * m)
        bv = B.BatchVerifier(pk, n, m, window_bits=16)
        vals, gams = zip(*[values(d * 17, m) for d in range(256)])
        pts, sc, V = bv.prove_batch(list(vals), list(gams))
        ix = np.arange(batch) % 256
        recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1)[ix])
        scs = np.ascontiguousarray(sc[ix])
        scs[batch // 3, 1, 0] ^= np.uint64(8)          # one tampered proof per batch
        wsb = bv.workspace_bytes(batch)
        bufs[name] = (torch.from_numpy(recs.view(np.int64)).to(dev), torch.from_numpy(scs.view(np.int64)).to(dev),
                      torch.full((batch,), 7, dtype=torch.int32, device=dev), torch.empty(wsb, dtype=torch.uint8, device=dev), wsb, batch)
        bvs[name] = bv
    st = torch.cuda.current_stream().cuda_stream

    def both():
        for name in ("c2", "c3"):
            p, s, ok, ws, wsb, batch = bufs[name]
            bvs[name].run_device(p.data_ptr(), s.data_ptr(), batch, ok.data_ptr(), ws.data_ptr(), wsb, st)
    both()
    torch.cuda.synchronize()
    for name in ("c2", "c3"):
        ok = bufs[name][2].cpu().numpy()
        batch = bufs[name][5]
        assert ok[batch // 3] == 1 and int(ok.sum()) == 1, name
    t0 = time.perf_counter()
    for _ in range(steps):
        both()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"tables_GB": round(sum(bv.table_bytes for bv in bvs.values()) / 1e9, 2), "ms_per_interleaved_pair": dt / steps * 1e3,
           "verifies_per_s": (8192 + 4096) * steps / dt, "verdicts_exact": True}
    for bv in bvs.values():
        bv.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    a = B.Arith.init("bls12_381")
    total = torch.cuda.get_device_properties(dev).total_memory
    res = {"hbm_bytes": total,
           "c2_n64_m16_batch8192": run_shape(a, 64, 16, 8192, [13, 14, 15, 16, 17], args.steps, dev),
           "c3_n64_m1_batch4096": run_shape(a, 64, 1, 4096, [12, 13, 14, 15, 16], args.steps, dev),
           "coresident_c2_c3_at_c16": coresident(a, dev, args.steps)}
    for rows in (res["c2_n64_m16_batch8192"], res["c3_n64_m1_batch4096"]):
        for r in rows:
            if "table_bytes" in r:
                r["table_frac_of_hbm"] = round(r["table_bytes"] / total, 4)
    text = json.dumps(res, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
