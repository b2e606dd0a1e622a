"""-m gpu: the N > 1 path on ONE GPU -- two ranks, both on cuda:0, collectives over gloo (the one-GPU box cannot run
RCCL across devices; the driver's 8-GPU run uses the same code with backend nccl).

 * bench.py launched as `python bench.py --gpus 2` (it spawns its own ranks) in the C5 shape, small batch: the line
   must report n_gpus == ranks_seen == 2, the per-rank tamper check and both exchange modes (all-reduce of failure
   counts; all-gather of combined-check partials) must pass inside it.
 * two ranks through the product API: rank 1 holds a tampered proof -- mode A must name exactly that proof, mode B
   (gather + sum of partials under one shared key) must reject; with all proofs valid both must accept."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_spawns_two_ranks_c5_shape():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"BPP_BENCH_BACKEND": "gloo", "BPP_BENCH_DEVICE": "0"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c5", "--batch", "512", "--window", "10",
           "--steps", "2", "--warmup", "1", "--combined-steps", "2", "--hard-steps", "0", "--other-curves-steps", "0",
           "--prove-steps", "1", "--tampered", "9"]
    out = subprocess.check_output(cmd, env=env, timeout=900)
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["config"]["launcher"] == "bench.py spawned the ranks" and line["config"]["backend"] == "gloo"
    assert line["metric"] == "aggregated range-proof verifies/sec (n=64,m=1)"
    assert line["value"] > 0 and line["tamper_check"]["verdicts_exact"] is True
    assert line["combined_check"]["value"] > 0 and line["prove"]["value"] > 0
    assert line["cpu_baseline"] is None          # rank 0 at N = 1 only


def test_rccl_initialises_and_runs_the_bench_collectives():
    """One rank, backend nccl (= RCCL): the process group, the all-reduce of the verdict count and of `ranks_seen`, and the
    all-gather of the combined-check partials all go through RCCL on the real device.  (Two RCCL ranks cannot share one
    GPU; the N > 1 control flow is the gloo test above, the N > 1 RCCL run is the driver's.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BPP_BENCH_BACKEND")}
    env.update({"BPP_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(_free_port())})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--batch", "512", "--window", "10",
           "--steps", "2", "--warmup", "1", "--combined-steps", "2", "--hard-steps", "0", "--other-curves-steps", "0",
           "--prove-steps", "0", "--serialized-steps", "0", "--latency-steps", "0", "--cpu-seconds", "0", "--tampered", "5"]
    out = subprocess.check_output(cmd, env=env, timeout=900)
    line = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["config"]["backend"] == "nccl"
    assert line["value"] > 0 and line["tamper_check"]["verdicts_exact"] is True and line["combined_check"]["value"] > 0


def _worker(rank, world, port, tamper, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bulletproofsplus_amd as B
    from bulletproofsplus_amd.sharding import shard_bounds, batch_verdict, gather_verdicts
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    n, m, total = 8, 2, 10
    a = B.Arith.init("bls12_381", 0)
    pk = B.PublicKey.new(a, n * m)
    bv = B.BatchVerifier(pk, n, m, window_bits=5)
    lo, hi = shard_bounds(total, world, rank)
    vals = [[(37 * p + j) % 256 for j in range(m)] for p in range(lo, hi)]
    gams = [[p + j + 1 for j in range(m)] for p in range(lo, hi)]
    pts, scs, V = bv.prove_batch(vals, gams)
    recs = np.ascontiguousarray(np.concatenate([pts, V], axis=1))
    scs = np.ascontiguousarray(scs)
    if tamper is not None and lo <= tamper < hi:
        scs[tamper - lo, 2, 0] ^= np.uint64(2)
    cnt = hi - lo
    d_pts = torch.from_numpy(recs.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(scs.view(np.int64)).to(dev)
    d_ok = torch.full((cnt,), 7, dtype=torch.int32, device=dev)
    wsb = bv.workspace_bytes(cnt)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    # mode A: per-proof verdicts, one all-reduce of the failure count
    bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), cnt, d_ok.data_ptr(), d_ws.data_ptr(), wsb)
    torch.cuda.synchronize()
    local = d_ok.cpu()
    fails, all_ok = batch_verdict(local, dist)
    full = gather_verdicts(local, total, dist).tolist()
    # mode B: one shared secret key, weights indexed by the GLOBAL proof index, one all-gather of the partials
    key = torch.zeros(32, dtype=torch.uint8)
    if rank == 0:
        key = torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8).clone()
    dist.broadcast(key, src=0)
    pbytes = bv.partial_bytes()
    d_part = torch.zeros(pbytes, dtype=torch.uint8, device=dev)
    d_cok = torch.full((1,), 7, dtype=torch.int32, device=dev)
    cwsb = bv.combined_workspace_bytes(cnt)
    d_cws = torch.empty(cwsb, dtype=torch.uint8, device=dev)
    bv.run_combined_device(d_pts.data_ptr(), d_sc.data_ptr(), cnt, bytes(key.tolist()), lo, d_part.data_ptr(), d_cok.data_ptr(),
                           d_cws.data_ptr(), cwsb)
    torch.cuda.synchronize()
    local_comb = int(d_cok.item())
    h_all = torch.zeros(world * pbytes, dtype=torch.uint8)
    dist.all_gather_into_tensor(h_all, d_part.cpu())
    d_all = h_all.to(dev)
    bv.sum_partials_device(d_all.data_ptr(), world, d_cok.data_ptr())
    torch.cuda.synchronize()
    q.put((rank, lo, hi, fails, all_ok, full, local_comb, int(d_cok.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tamper", [None, 7])
def test_two_ranks_mode_a_and_mode_b(tamper):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, tamper, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = [0] * 10
    if tamper is not None:
        exp[tamper] = 1
    for rank, lo, hi, fails, all_ok, full, local_comb, global_comb in res:
        assert fails == sum(exp) and all_ok == (tamper is None) and full == exp
        assert global_comb == (0 if tamper is None else 1)            # every rank reaches the same combined verdict
        assert local_comb == (1 if tamper is not None and lo <= tamper < hi else 0)
