"""CPU: host builds (g++) of the device headers' arithmetic that has no oracle counterpart because it is an ENGINE
optimisation, not reference behaviour: the lazy (unreduced) XYZZ mixed addition against the eager one with every
intermediate bound asserted (tests/host/lazy_host_test.cpp), and the GLV scalar split of the BLS12-381 proof-point MSM
(tests/host/glv_host_test.cpp) against Python integers."""

import os
import random
import subprocess
import tempfile

import pyref as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(name, opt="-O1"):
    exe = os.path.join(tempfile.gettempdir(), "bpp_" + name)
    subprocess.check_call(["g++", opt, "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", name + ".cpp")])
    return exe


def test_lazy_mixed_addition_bounds_and_values():
    out = subprocess.check_output([_build("lazy_host_test")]).decode()
    assert "ok bls12_381" in out and "ok secp256k1" in out


def test_edwards_lazy_mixed_addition():
    out = subprocess.check_output([_build("ed_lazy_host_test")]).decode()
    assert "ok ed25519" in out


def test_glv_split_matches_integers():
    exe = _build("glv_host_test", "-O2")
    r = P.BLS12_381["r"]
    z2 = 0xd201000000010000 ** 2
    rng = random.Random(11)
    ks = [0, 1, z2 - 1, z2, z2 + 1, 2 * z2 - 1, 2 * z2, r - 1, r - 2, (r // z2) * z2, (r // z2) * z2 - 1, (1 << 128) - 1, 1 << 128,
          (1 << 255) - 1 if (1 << 255) - 1 < r else r - 3]
    ks += [rng.randrange(r) for _ in range(3000)]
    ks += [rng.randrange(1 << 128) * z2 + d for d in (0, 1, z2 - 1) for _ in range(200) if True]
    ks = [k for k in ks if k < r]
    for off in range(0, len(ks), 500):
        chunk = ks[off:off + 500]
        out = subprocess.check_output([exe] + ["%064x" % k for k in chunk]).decode().split("\n")
        for k, line in zip(chunk, out):
            k1, k2 = (int(x, 16) for x in line.split())
            assert (k1, k2) == (k % z2, k // z2), hex(k)
            assert k1 < (1 << 128) and k2 < (1 << 128)
