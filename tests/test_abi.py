"""CPU: the C-ABI library loads and exports every symbol include/bpp_amd.h declares (no compute calls)."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# BPP_HOST_SANITIZE=1: host builds under ASan + UBSan (see tests/test_host_arith_cpu.py)
SANITIZE = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if os.environ.get("BPP_HOST_SANITIZE") else []


def test_library_exports_every_declared_symbol():
    from bulletproofsplus_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "bpp_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bpp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for s in declared:
        assert hasattr(L, s), s


def test_point_words_and_usage_errors_without_gpu():
    from bulletproofsplus_amd import _lib
    L = _lib.lib()
    assert L.bpp_point_words(0) == 13 and L.bpp_point_words(1) == 9 and L.bpp_point_words(7) < 0
    # null arguments are usage errors, not crashes
    assert L.bpp_msm(None, None, None, 0, None) < 0
    assert L.bpp_verifier_workspace_bytes(None, 10) == 0


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "bulletproofsplus_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"import\s+(oracle|pyref)|from\s+(oracle|pyref)|libbpp_oracle|bpp_oracle|orc_", src), \
                    os.path.join(dirpath, f)


def test_host_field_arithmetic_matches_bigints():
    """csrc/field.hpp compiled for the host (g++) against Python big integers."""
    import random
    import subprocess
    import tempfile
    import pyref as P
    exe = os.path.join(tempfile.gettempdir(), "bpp_field_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17"] + SANITIZE + ["-o", exe, os.path.join(ROOT, "tests", "host", "field_host_test.cpp")])
    mods = {"blsfp": P.BLS12_381["p"], "blsfr": P.BLS12_381["r"], "secpfp": P.SECP256K1["p"], "secpfr": P.SECP256K1["r"],
            "edfp": P.ED25519["p"], "edfr": P.ED25519["r"]}
    rnd = random.Random(3)
    lines, exp = [], []
    for f, m in mods.items():
        vals = [0, 1, 2, m - 1, m - 2, (1 << 64) - 1, (1 << 30) - 1, 1 << 30, (1 << 60) + 1] + [rnd.randrange(m) for _ in range(20)]
        for a in vals:
            for b in vals[:5] + vals[-2:]:
                for op, fn in (("mul", lambda a, b: a * b % m), ("add", lambda a, b: (a + b) % m), ("sub", lambda a, b: (a - b) % m),
                               ("muladd", lambda a, b: -a * b % m)):
                    lines.append("%s %s %x %x" % (f, op, a, b)); exp.append(fn(a, b))
            lines.append("%s sqr %x" % (f, a)); exp.append(a * a % m)
            lines.append("%s neg %x" % (f, a)); exp.append(-a % m)
            lines.append("%s tocanon %x" % (f, a)); exp.append(a)
            # safegcd inversion (0 -> 0) and the Fermat cross-check
            lines.append("%s inv %x" % (f, a)); exp.append(pow(a, -1, m) if a else 0)
            lines.append("%s invf %x" % (f, a)); exp.append(pow(a, -1, m) if a else 0)
        # inversion stress: powers of two, values around them, many random residues
        stress = [1 << k for k in range(0, m.bit_length() - 1, 7)] + [(1 << k) - 1 for k in range(2, m.bit_length() - 1, 11)]
        stress += [m - (1 << k) for k in range(0, m.bit_length() - 2, 13)] + [rnd.randrange(m) for _ in range(300)]
        for a in stress:
            lines.append("%s inv %x" % (f, a % m)); exp.append(pow(a % m, -1, m) if a % m else 0)
        top = (1 << (384 if f == "blsfp" else 256)) - 1
        lines.append("%s mul %x 3" % (f, top)); exp.append(top * 3 % m)
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split()
    assert [int(o, 16) for o in out] == exp


def test_generated_constants_are_in_sync():
    """csrc/constants_gen.h is what tools/gen_constants.py generates from the field moduli (Montgomery constants,
    p^-1 mod 2^30, number of divstep batches, curve constants)."""
    import subprocess
    import sys
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "bpp_constants_check.h")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_constants.py"), out], stdout=subprocess.DEVNULL)
    with open(out) as f, open(os.path.join(ROOT, "bulletproofsplus_amd", "csrc", "constants_gen.h")) as g:
        assert f.read() == g.read()
