import sys, os, time, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import bulletproofsplus_amd as B
from bulletproofsplus_amd import _lib
a = B.Arith.init("bls12_381")
pk = B.PublicKey.new(a, 1024)
n = 8192 * 55
pts = np.tile(np.concatenate([pk.gh, pk.G_vec[:53]]), (8192, 1))
enc = B.compress_points(a, pts)
dev = torch.device("cuda:0")
d_in = torch.from_numpy(enc).to(dev)
d_out = torch.zeros((n, a.PW), dtype=torch.int64, device=dev)
d_ok = torch.zeros(n, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
def run():
    rc = L.bpp_points_decompress_device(a.handle, d_in.data_ptr(), n, d_out.data_ptr(), d_ok.data_ptr(), 0, st)
    assert rc == 0
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert int(d_ok.sum().item()) == 0
assert np.array_equal(d_out.cpu().numpy().view(np.uint64), pts)
print("decompress %d points: %.3f ms (%.1f M points/s)" % (n, dt * 1e3, n / dt / 1e6))
