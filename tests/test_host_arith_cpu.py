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


# BPP_HOST_SANITIZE=1: the same host builds under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool offers no
# sanitizer, so the device headers' arithmetic gets its sanitizer run here, on the CPU build)
SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("BPP_HOST_SANITIZE") else []


def _build(name, opt="-O1"):
    exe = os.path.join(tempfile.gettempdir(), "bpp_" + name + ("_san" if SANITIZE else ""))
    subprocess.check_call(["g++", opt, "-std=c++17"] + SANITIZE + ["-o", exe, os.path.join(ROOT, "tests", "host", name + ".cpp")])
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


def test_glv_split_secp256k1_signed():
    """k == (+-k1) + (+-k2) lambda mod n with both magnitudes below 2^128 -- random scalars and the edges (0, 1, n - 1,
    lambda, n - lambda, the middle of the range, values that make either half negative)"""
    exe = _build("glv_host_test", "-O2")
    n = P.SECP256K1["r"]
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    assert (lam * lam + lam + 1) % n == 0
    rng = random.Random(12)
    ks = [0, 1, 2, n - 1, n - 2, lam, n - lam, lam + 1, lam - 1, (n - 1) // 2, (n + 1) // 2, (1 << 128) - 1, 1 << 128, (1 << 255),
          (1 << 256) - 1 - ((1 << 256) - n) - 5]
    ks += [rng.randrange(n) for _ in range(4000)]
    ks += [(a + b * lam) % n for a in (1, -1, (1 << 127) - 1, -(1 << 127) + 1) for b in (1, -1, (1 << 127) - 1, -(1 << 127) + 1)]
    ks = [k % n for k in ks]
    seen_neg = [0, 0]
    for off in range(0, len(ks), 500):
        chunk = ks[off:off + 500]
        out = subprocess.check_output([exe, "secp"] + ["%064x" % k for k in chunk]).decode().split("\n")
        for k, line in zip(chunk, out):
            s1, k1, s2, k2 = line.split()
            k1, k2 = int(k1, 16), int(k2, 16)
            v1 = -k1 if s1 == "1" else k1
            v2 = -k2 if s2 == "1" else k2
            assert (v1 + v2 * lam - k) % n == 0, hex(k)
            assert k1 < (1 << 128) and k2 < (1 << 128), hex(k)
            seen_neg[0] += s1 == "1"
            seen_neg[1] += s2 == "1"
    assert seen_neg[0] > 100 and seen_neg[1] > 100
