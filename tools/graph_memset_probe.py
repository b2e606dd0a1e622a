#!/usr/bin/env python3
"""The graph-memset finding of round 2, reproduced in isolation under the SAME capture the failing test used
(torch.cuda.graph: PyTorch's capture of a side stream with its private pool), since raw HIP capture shows no fault
(tools/graph_memset_probe.hip: every size and offset zeroes correctly on every replay).
Per case: dirty a sub-range of a torch allocation with 0xff, replay a captured graph {hipMemsetAsync(sub-range, 0, bytes);
count the non-zero bytes}, four replays; prints one JSON line per case."""
import ctypes
import json

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def case(nbytes, offset, eager_first):
    dev = torch.device("cuda:0")
    buf = torch.empty(1 << 22, dtype=torch.uint8, device=dev)
    sub = buf[offset:offset + nbytes]
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    if eager_first:
        with torch.cuda.stream(s):
            assert hip.hipMemsetAsync(sub.data_ptr(), 0, nbytes, s.cuda_stream) == 0
            s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rc = hip.hipMemsetAsync(sub.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        cnt.copy_(torch.count_nonzero(sub).reshape(1))
    seen = []
    for _ in range(4):
        sub.fill_(255)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        seen.append(int(cnt.item()))
    print(json.dumps({"bytes": nbytes, "offset": offset, "eager_call_first": eager_first, "memset_rc": rc,
                      "nonzero_after_replay": seen}), flush=True)


for nb in (256, 1024, 1028, 1200, 4096, 65536, 1200000):
    for off in (0, 256, 4352):
        for eager in (False, True):
            case(nb, off, eager)
