"""CPU: the source-only Rust binding (bindings/rust) stays in step with the C ABI.  north_star asks for Rust host code;
the image has no Rust toolchain, so nothing here compiles Rust -- instead the prototypes of include/bpp_amd.h and the
`extern "C"` block of bindings/rust/src/ffi.rs are parsed INDEPENDENTLY and compared (name, arity, pointer-ness,
constness, scalar type), the generator must reproduce the committed ffi.rs, and every ffi call in src/lib.rs must name
an existing function with the arity it is declared with.  Mirrors reference src/lib.rs:11-13 / build.rs:1-3."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as G  # noqa: E402

RUST_SCALAR = {"c_int": "int", "usize": "size_t", "u64": "uint64_t", "u32": "uint32_t", "u8": "uint8_t", "c_uint": "unsigned",
               "f32": "float", "c_char": "char", "c_void": "void", "BppCtx": "bpp_ctx", "BppVerifier": "bpp_verifier", "BppGraph": "bpp_graph"}


def rust_prototypes(text):
    """-> {name: (ret (base, const, stars) or None, [(base, const, stars)])} from the extern "C" block"""
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}

    def ty(t):
        t = t.strip()
        stars, const = 0, False
        first = True
        while t.startswith("*"):
            m = re.match(r"\*(const|mut)\s+", t)
            # the OUTERMOST qualifier of the Rust type is the innermost C pointee's: `*mut *mut T` / `*const T`
            if first:
                first = False
            inner_const = m.group(1) == "const"
            t = t[m.end():]
            stars += 1
            last_const = inner_const
        if stars:
            const = last_const if stars == 1 else False
            # for multi-level pointers only the innermost level's constness is compared; the header has none that are const
        return RUST_SCALAR[t], const, stars
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        name, params, ret = m.group(1), m.group(2).strip(), m.group(3)
        ps = []
        if params:
            for p in params.split(","):
                ps.append(ty(p.split(":", 1)[1]))
        out[name] = (ty(ret) if ret else None, ps)
    return out


def test_ffi_rs_matches_the_header():
    header = open(os.path.join(ROOT, "include", "bpp_amd.h")).read()
    ffi = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
    cps = G.c_prototypes(header)
    rps = rust_prototypes(ffi)
    assert len(cps) >= 50 and {n for n, _, _ in cps} == set(rps), set(rps) ^ {n for n, _, _ in cps}
    for name, (rb, rc, rs), ps in cps:
        rret, rparams = rps[name]
        if rb == "void" and rs == 0:
            assert rret is None, name
        else:
            assert rret == (rb, rc if rs == 1 else False, rs), (name, rret, (rb, rc, rs))
        assert len(ps) == len(rparams), (name, "arity")
        for (pn, b, c, s), (rb2, rc2, rs2) in zip(ps, rparams):
            assert (b, s) == (rb2, rs2), (name, pn, "type / pointer depth")
            if s == 1:
                assert c == rc2, (name, pn, "constness")
    # every exported symbol the Python binding knows is declared in both
    from bulletproofsplus_amd import _lib
    assert set(_lib.EXPORTS) <= set(rps)


def test_generator_reproduces_committed_ffi_rs():
    assert G.generate() == open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read(), \
        "include/bpp_amd.h changed: run python tools/gen_rust_ffi.py"


def test_lib_rs_calls_exist_with_the_declared_arity():
    lib = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    rps = rust_prototypes(open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read())
    calls = list(re.finditer(r"ffi::(bpp_\w+)\(", lib))
    assert len(calls) >= 7
    for m in calls:
        name = m.group(1)
        assert name in rps, name
        # argument list: up to the matching parenthesis
        depth, i = 1, m.end()
        args, cur = [], ""
        while depth:
            ch = lib[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        assert len(args) == len(rps[name][1]), (name, len(args), len(rps[name][1]))
    # the reference's public types and entry points are all there (src/lib.rs:11-13, README.md:47-55)
    for item in ("pub struct PublicKey", "pub struct RangeProver", "pub struct RangeProof", "pub struct RangeVerifier",
                 "pub struct MulVec", "pub fn prove(pk: &PublicKey, n: usize, prover: &RangeProver) -> RangeProof",
                 "pub fn verify(&self, pk: &PublicKey, n: usize, commitment_vec: &[Point]) -> Result<(), ProofError>",
                 "pub fn commit(&mut self, pk: &PublicKey, v: u64, gamma: PrimeFieldElem)", "pub fn calculate(&self) -> Point"):
        assert item in lib, item
    assert "rustc-link-lib=dylib=bpp_amd" in open(os.path.join(ROOT, "bindings", "rust", "build.rs")).read()
