"""Helpers shared by the -m gpu tests: the product is driven through the C ABI (bulletproofsplus_amd),
the oracle (oracle/) is the checker."""

import numpy as np
import pytest


def need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def hexpt(h):
    return None if h is None else (int(h[0], 16), int(h[1], 16))


def run_verifier_device(torch, bv, records, scalars, want_scalars=True, want_result=True, challenges=None):
    """records (count, NV, PW) u64, scalars (count, 3, 4) u64 -> (ok, out_scalars, out_result) numpy"""
    dev = torch.device("cuda:0")
    count = records.shape[0]
    PW = bv.arith.PW
    d_pts = torch.from_numpy(np.ascontiguousarray(records).view(np.int64)).to(dev)
    d_sc = torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).to(dev)
    d_ok = torch.full((count,), 7, dtype=torch.int32, device=dev)
    wsb = bv.workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    d_os = torch.zeros((count, bv.msm_len, 4), dtype=torch.int64, device=dev) if want_scalars else None
    d_or = torch.zeros((count, PW), dtype=torch.int64, device=dev) if want_result else None
    d_ch = None
    if challenges is not None:
        d_ch = torch.from_numpy(np.ascontiguousarray(challenges).view(np.int64)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    bv.run_device(d_pts.data_ptr(), d_sc.data_ptr(), count, d_ok.data_ptr(), d_ws.data_ptr(), wsb, stream,
                  d_challenges=d_ch.data_ptr() if d_ch is not None else 0,
                  d_out_scalars=d_os.data_ptr() if d_os is not None else 0,
                  d_out_result=d_or.data_ptr() if d_or is not None else 0)
    torch.cuda.synchronize()
    ok = d_ok.cpu().numpy().astype(np.uint32)
    os_ = d_os.cpu().numpy().view(np.uint64) if d_os is not None else None
    or_ = d_or.cpu().numpy().view(np.uint64) if d_or is not None else None
    return ok, os_, or_


def run_combined_device(torch, bv, records, scalars, seed=12345, index_base=0, weights=None):
    """Combined batch check on device -> (ok_flag, partial bytes as numpy uint8).  The 32-byte weight key is derived
    from `seed` (tests want repeatable weights; production callers pass os.urandom(32)); weights: (count, 2) uint64
    to supply the 128-bit weights directly."""
    import hashlib
    dev = torch.device("cuda:0")
    count = records.shape[0]
    d_pts = torch.from_numpy(np.ascontiguousarray(records).view(np.int64)).to(dev)
    d_sc = torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).to(dev)
    d_ok = torch.full((1,), 7, dtype=torch.int32, device=dev)
    d_part = torch.zeros(bv.partial_bytes(), dtype=torch.uint8, device=dev)
    wsb = bv.combined_workspace_bytes(count)
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    key = hashlib.sha256(b"test weight key %d" % seed).digest()
    d_w = None
    if weights is not None:
        d_w = torch.from_numpy(np.ascontiguousarray(weights, dtype=np.uint64).view(np.int64)).to(dev)
    bv.run_combined_device(d_pts.data_ptr(), d_sc.data_ptr(), count, key, index_base, d_part.data_ptr(), d_ok.data_ptr(),
                           d_ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream,
                           d_weights=d_w.data_ptr() if d_w is not None else 0)
    torch.cuda.synchronize()
    return int(d_ok.item()), d_part.cpu().numpy()


def run_grouped_device(torch, bv, records, scalars, group, seed=12345, index_base=0, challenges=None):
    """Grouped check on device -> (verdicts (count,) u32, groups failed, proofs re-verified exactly).  Weight key from
    `seed` (repeatable; production callers pass os.urandom(32))."""
    import hashlib
    dev = torch.device("cuda:0")
    count = records.shape[0]
    d_pts = torch.from_numpy(np.ascontiguousarray(records).view(np.int64)).to(dev)
    d_sc = torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).to(dev)
    d_ok = torch.full((max(count, 1),), 7, dtype=torch.int32, device=dev)
    wsb = bv.grouped_workspace_bytes(count, group)
    d_ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    key = hashlib.sha256(b"test weight key %d" % seed).digest()
    d_ch = None
    if challenges is not None:
        d_ch = torch.from_numpy(np.ascontiguousarray(challenges).view(np.int64)).to(dev)
    failed, redone = bv.run_grouped_device(d_pts.data_ptr(), d_sc.data_ptr(), count, key, index_base, d_ok.data_ptr(),
                                           d_ws.data_ptr(), wsb, group=group,
                                           stream=torch.cuda.current_stream().cuda_stream,
                                           d_challenges=d_ch.data_ptr() if d_ch is not None else 0)
    torch.cuda.synchronize()
    return d_ok.cpu().numpy().astype(np.uint32)[:count], failed, redone
