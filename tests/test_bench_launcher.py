"""CPU: bench.py's multi-GPU control flow without a GPU (`--dry-run`: the same launcher, rendezvous, barriers, failure-count
all-reduce, partial all-gather and rank-0 JSON line, over gloo, with the verification pass left out).
 * `python bench.py --gpus 2` with no WORLD_SIZE in the environment must spawn the two ranks itself;
 * under `python -m torch.distributed.run` (the driver's launch line) it must act as a rank.
The real 2-rank rehearsal on a GPU is tests/test_gpu_multirank.py."""

import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_gpus_flag_spawns_its_own_ranks():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2",
                                   "--config", "c5"], env=_env(), timeout=300)
    line = _json_line(out)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["metric"] == "aggregated range-proof verifies/sec (n=64,m=1)"


def test_single_process_default():
    line = _json_line(subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=_env(),
                                              timeout=300))
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


def test_under_torch_distributed_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"]
    line = _json_line(subprocess.check_output(cmd, env=_env(), timeout=300, stderr=subprocess.DEVNULL))
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
