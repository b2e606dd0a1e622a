"""CPU, world_size = 2, gloo: the N > 1 path of the batch verifier -- proof-index sharding, the single
all-reduce of failure counts, and verdict gathering.  The per-proof verdicts here come from the oracle
(this test has no GPU); on the GPU box the same functions are driven by bench.py over RCCL."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, verdicts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from bulletproofsplus_amd.sharding import shard_bounds, batch_verdict, gather_verdicts
    dist.init_process_group("gloo", rank=rank, world_size=world)
    count = len(verdicts)
    lo, hi = shard_bounds(count, world, rank)
    local = torch.tensor(verdicts[lo:hi], dtype=torch.int32)
    total, ok = batch_verdict(local, dist)
    full = gather_verdicts(local, count, dist)
    q.put((rank, lo, hi, total, ok, full.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(verdicts, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, verdicts, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_shard_bounds_cover_and_balance():
    from bulletproofsplus_amd.sharding import shard_bounds
    for count in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(count, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(65536, 8, 3) == (24576, 32768)     # config C5: 8192 proofs per GPU
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_two_rank_verdict_exchange_with_oracle_verdicts():
    # real verdicts from the oracle for a small mixed batch (valid / tampered), n=8, m=2, secp256k1
    import oracle as O
    pk = O.PublicKey(O.SECP256K1, 16)
    pts, sc, V = O.range_prove(pk, 8, [200, 5], [3, 7])
    verdicts = []
    for i in range(7):
        s2 = sc.copy()
        if i in (2, 5):
            s2[i % 3, 0] ^= 1
        verdicts.append(O.range_verify(pk, 8, 2, pts, s2, V))
    assert verdicts == [0, 0, 1, 0, 0, 1, 0]
    res = _run(verdicts)
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    for _, _, _, total, ok, full in res:
        assert total == 2 and ok is False and full == verdicts
    res = _run([0] * 6)
    assert all(r[3] == 0 and r[4] is True for r in res)
