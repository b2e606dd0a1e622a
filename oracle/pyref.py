"""TEST INFRASTRUCTURE ONLY -- big-integer restatement of the reference's range-proof path.

This module is part of ``oracle/``: it may be imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, as the checker.
The product (``bulletproofsplus_amd``) never imports it.

What it is
----------
A pure-Python (arbitrary precision ``int``) restatement of the reference crate's protocol
code, written against an abstract *group backend* so that the same protocol code runs over

* ``ShadowGroup``  -- the "dlog shadow": a point is represented by its discrete log w.r.t. the
  base point, so ``add -> + mod r``, ``scalar-mul -> * mod r``, ``MulVec -> dot product``.
  Possible because the reference's ``PublicKey::new`` makes every generator a *known* multiple
  of g (reference ``src/publickey.rs:21-48``).  Gives group-independent known answers.
* ``WeierstrassGroup`` -- real affine short-Weierstrass (a = 0) arithmetic for BLS12-381 G1 and
  secp256k1, following the case analysis of reference
  ``src/secp256k1/building_block/macros.rs:34-152`` (add) and ``:1-32`` (LSB-first
  double-and-add scalar multiplication).

It is used (1) to pin the C oracle (``oracle/bpp_oracle.c``) and (2) to generate the golden
fixtures under ``tests/golden/`` (``tests/golden/make_golden.py``).

Parity status: the reference's BLS12-381 arithmetic is the third-party ``mcl_rust``
(herumi/mcl, un-pinned HEAD, absent from /root/reference), so BLS12-381 *point coordinates*
are "parity unpinned" by the reference's own tests; they are pinned by BLS12-381 G1 being a
standard curve plus the generator literal at reference
``src/bls12_381/building_block/point/point.rs:16``.  secp256k1 primitives are pinned by the
reference's known-answer tests (``affine_point.rs:231-341``).  Protocol scalars are pinned by
the dlog-shadow known answers recorded in SURVEY.md section 8c.

Reference files restated (line numbers cite /root/reference/src):
  util.rs:29-127, publickey.rs:21-52, range/prover.rs:20-42, range/mod.rs:31-510,
  weighted_inner_product_proof.rs:36-382, bls12_381/building_block/mulvec.rs:20-53,
  bls12_381/building_block/scalar/prime_field_elem.rs:191-248.
"""

from __future__ import annotations

# --------------------------------------------------------------------------------------
# Curve parameters
# --------------------------------------------------------------------------------------

BLS12_381 = dict(
    name="bls12_381",
    # base field
    p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    # group order (= scalar field Fr)
    r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    b=4,
    # generator: decimal literal at reference bls12_381/building_block/point/point.rs:16
    gx=3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
    gy=1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569,
    fp_bytes=48,
)

SECP256K1 = dict(
    name="secp256k1",
    # reference secp256k1/building_block/secp256k1/secp256k1.rs:22,26,47-48
    p=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F,
    r=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    b=7,
    gx=0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    gy=0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
    fp_bytes=32,
)

_ED_P = (1 << 255) - 19
ED25519 = dict(
    name="ed25519",
    # edwards25519 (RFC 8032).  NOT a backend of the reference: used only to check the
    # engine's third instantiation; parity unpinned.
    p=_ED_P,
    r=(1 << 252) + 27742317777372353535851937790883648493,
    d=(-121665 * pow(121666, -1, _ED_P)) % _ED_P,
    gx=15112221349535400772501151409588531511454012693041857206046113283949847762202,
    gy=4 * pow(5, -1, _ED_P) % _ED_P,
    fp_bytes=32,
)

CURVES = {"bls12_381": BLS12_381, "secp256k1": SECP256K1, "ed25519": ED25519}


# --------------------------------------------------------------------------------------
# Group backends.  A "point" is an opaque value; INF is backend specific.
# --------------------------------------------------------------------------------------

class ShadowGroup:
    """Points are discrete logs mod r (reference publickey.rs:23-39 makes this possible)."""

    def __init__(self, r: int):
        self.r = r

    def base(self):
        return 1

    def zero(self):
        return 0

    def is_zero(self, P):
        return P % self.r == 0

    def add(self, P, Q):
        return (P + Q) % self.r

    def neg(self, P):
        return (-P) % self.r

    def mul(self, P, k):
        return (P * k) % self.r

    def eq(self, P, Q):
        return (P - Q) % self.r == 0


class WeierstrassGroup:
    """y^2 = x^3 + b over F_p, affine, points are (x, y) tuples or None for infinity."""

    def __init__(self, curve: dict):
        self.c = curve
        self.p = curve["p"]
        self.r = curve["r"]
        self.b = curve["b"]

    def base(self):
        return (self.c["gx"], self.c["gy"])

    def zero(self):
        return None

    def is_zero(self, P):
        return P is None

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (y * y - x * x * x - self.b) % self.p == 0

    def neg(self, P):
        if P is None:
            return None
        return (P[0], (-P[1]) % self.p)

    def add(self, P, Q):
        # case order of reference macros.rs:42-146
        p = self.p
        if P is None and Q is None:
            return None
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2 and y1 != y2:
            return None
        if x1 == x2 and y1 == y2:
            if x1 == 0 or y1 == 0:  # macros.rs:53-55 (quirk kept; unreachable on these curves)
                return None
            m = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
            x3 = (m * m - 2 * x1) % p
            y3 = (m * (x1 - x3) - y1) % p
            return (x3, y3)
        m = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (m * m - x1 - x2) % p
        y3 = (-(m * (x3 - x1) + y1)) % p
        return (x3, y3)

    def mul(self, P, k):
        # LSB-first double-and-add, reference macros.rs:9-27.  k is NOT reduced: the
        # reference's secp256k1 tests pass base-field elements as scalars (affine_point.rs:258).
        res = None
        q = P
        while k:
            if k & 1:
                res = self.add(res, q)
            q = self.add(q, q)
            k >>= 1
        return res

    def eq(self, P, Q):
        return P == Q


class EdwardsGroup:
    """-x^2 + y^2 = 1 + d x^2 y^2 over F_p (a = -1), affine; the identity (0, 1) is represented as None so
    that the wire format (inf flag) matches the Weierstrass backends."""

    def __init__(self, curve: dict):
        self.c = curve
        self.p = curve["p"]
        self.r = curve["r"]
        self.d = curve["d"]

    def base(self):
        return (self.c["gx"], self.c["gy"])

    def zero(self):
        return None

    def is_zero(self, P):
        return P is None

    def is_identity_class(self, P):
        """identity of ristretto255's quotient group: a point of E[4] (x = 0 or y = 0); on prime-order inputs this is
        the exact identity test (csrc/ristretto.hpp ed_is_identity_class)"""
        return P is None or P[0] == 0 or P[1] == 0

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (-x * x + y * y - 1 - self.d * x * x * y * y) % self.p == 0

    def neg(self, P):
        if P is None:
            return None
        return ((-P[0]) % self.p, P[1])

    def add(self, P, Q):
        p, d = self.p, self.d
        x1, y1 = (0, 1) if P is None else P
        x2, y2 = (0, 1) if Q is None else Q
        t = d * x1 * x2 * y1 * y2 % p
        x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, p) % p
        y3 = (y1 * y2 + x1 * x2) * pow(1 - t, -1, p) % p
        return None if (x3 == 0 and y3 == 1) else (x3, y3)

    def mul(self, P, k):
        res = None
        q = P
        while k:
            if k & 1:
                res = self.add(res, q)
            q = self.add(q, q)
            k >>= 1
        return res

    def eq(self, P, Q):
        return P == Q


# --------------------------------------------------------------------------------------
# Scalar helpers (reference src/util.rs)
# --------------------------------------------------------------------------------------

class Fr:
    """Scalar-field helper bound to an order r.  All values are ints in [0, r)."""

    def __init__(self, r: int):
        self.r = r

    def new(self, n: int) -> int:
        # PrimeFieldElem::new(i32) -> Fr::set_int; negative n maps to r-|n|
        # (reference bls12_381/building_block/scalar/prime_field_elem.rs:191-195)
        return n % self.r

    def inv(self, x: int) -> int:
        return pow(x, -1, self.r)

    def batch_invert(self, xs):
        # prime_field_elem.rs:239-248: returns (prod of inverses, [inverses])
        prod = 1
        inv = []
        for x in xs:
            ix = self.inv(x)
            inv.append(ix)
            prod = prod * ix % self.r
        return prod, inv

    def exp_iter_type1(self, x, n):
        # util.rs:29-32 : 1, x, x^2, ...
        out, cur = [], 1
        for _ in range(n):
            out.append(cur)
            cur = cur * x % self.r
        return out

    def exp_iter_type2(self, x, n):
        # util.rs:34-37 : x, x^2, ...
        out, cur = [], x % self.r
        for _ in range(n):
            out.append(cur)
            cur = cur * x % self.r
        return out

    def scalar_exp_vartime(self, x, n):
        # util.rs:39-52
        result, aux = 1, x % self.r
        while n > 0:
            if n & 1:
                result = result * aux % self.r
            n >>= 1
            aux = aux * aux % self.r
        return result

    def sum_of_powers_type1(self, x, n):
        # util.rs:54-79
        r = self.r
        if n & (n - 1) or n == 0:
            if n == 0:
                # is_power_of_two(0) is false in Rust -> slow path -> empty sum
                return 0
            return sum(self.exp_iter_type1(x, n)) % r
        if n == 1:
            return 1
        m = n
        result = (1 + x) % r
        factor = x % r
        while m > 2:
            factor = factor * factor % r
            result = (result + factor * result) % r
            m //= 2
        return result

    def sum_of_powers_type2(self, x, n):
        # util.rs:81-106
        r = self.r
        if n & (n - 1) or n == 0:
            if n == 0:
                return 0
            return sum(self.exp_iter_type2(x, n)) % r
        if n == 1:
            return 1  # util.rs:86 returns new(n) = 1 (NOT x) -- quirk kept
        m = n
        result = (x + x * x) % r
        factor = x % r
        while m > 2:
            factor = factor * factor % r
            result = (result + factor * result) % r
            m //= 2
        return result

    def weighted_inner_product(self, a, b, c):
        # util.rs:117-127
        out = 0
        for ai, bi, ci in zip(a, b, c):
            out = (out + ai * bi * ci) % self.r
        return out


# --------------------------------------------------------------------------------------
# MulVec (reference bls12_381/building_block/mulvec.rs:20-53)
# --------------------------------------------------------------------------------------

class MulVec:
    def __init__(self, G):
        self.G = G
        self.scalars = []
        self.points = []

    def add_scalar(self, s):
        self.scalars.append(s)

    def add_scalars(self, ss):
        self.scalars.extend(ss)

    def add_point(self, p):
        self.points.append(p)

    def add_points(self, ps):
        self.points.extend(ps)

    def calculate(self):
        if len(self.scalars) != len(self.points):
            raise RuntimeError("mulvec: lengths of scalars and points must match")
        G = self.G
        acc = G.zero()
        for s, p in zip(self.scalars, self.points):
            acc = G.add(acc, G.mul(p, s))
        return acc


# --------------------------------------------------------------------------------------
# PublicKey / RangeProver (reference publickey.rs, range/prover.rs)
# --------------------------------------------------------------------------------------

def _i32(v: int) -> int:
    """Rust ``as i32`` truncation (two's complement wrap)."""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


class PublicKey:
    def __init__(self, G, length: int):
        # publickey.rs:21-48
        F = Fr(G.r)
        g = G.base()
        self.G = G
        self.g = g
        self.h = G.mul(g, F.new(2))
        self.G_vec = [G.mul(g, F.new(_i32((i + 1) * 3))) for i in range(length)]
        self.H_vec = [G.mul(g, F.new(_i32((i + 1) * 5))) for i in range(length)]

    def commitment(self, v, gamma):
        # publickey.rs:50-52
        G = self.G
        return G.add(G.mul(self.g, v), G.mul(self.h, gamma))


class RangeProver:
    def __init__(self):
        self.v_vec = []
        self.gamma_vec = []
        self.commitment_vec = []

    def commit(self, pk: PublicKey, v: int, gamma: int):
        # range/prover.rs:28-42 ; note the `v as i32` truncation at :37
        F = Fr(pk.G.r)
        self.v_vec.append(v)
        self.gamma_vec.append(gamma)
        self.commitment_vec.append(pk.commitment(F.new(_i32(v)), gamma))


# --------------------------------------------------------------------------------------
# Hard-coded "transcript" constants
# --------------------------------------------------------------------------------------

class Transcript:
    """The reference has no Fiat-Shamir transcript; these are its literals."""

    ALPHA_SINGLE = 7      # range/mod.rs:94
    ALPHA_MULTI = 33      # range/mod.rs:256
    Y_SINGLE, Z_SINGLE = 7, 7      # range/mod.rs:109-110, :198-199
    Y_MULTI, Z_MULTI = 12, 23      # range/mod.rs:278-279, :417-418
    D_L, D_R = 4, 5       # wip.rs:94-95
    E_ROUND = 7           # wip.rs:131, :353 ; tests may set a list (one challenge per round)
    R, S, DELTA, ETA = 33, 44, 88, 123   # wip.rs:175-178
    E_FINAL = 99          # wip.rs:211, :369

    # Fiat-Shamir mode (csrc/transcript.hpp; the reference has none): `fs` holds an FsTranscript factory, `run` the
    # running transcript of the proof being made, `cached` the challenges a verifier derived from a proof.
    fs = None
    run = None
    cached = None

    @classmethod
    def e_round(cls, i):
        if cls.cached is not None:
            return cls.cached["e_rounds"][i]
        return cls.E_ROUND[i] if isinstance(cls.E_ROUND, (list, tuple)) else cls.E_ROUND

    @classmethod
    def e_final(cls):
        return cls.cached["e"] if cls.cached is not None else cls.E_FINAL

    @classmethod
    def yz(cls, m):
        if cls.cached is not None:
            return cls.cached["y"], cls.cached["z"]
        return (cls.Y_SINGLE, cls.Z_SINGLE) if m == 1 else (cls.Y_MULTI, cls.Z_MULTI)


class FsTranscript:
    """The SHA-256 transcript of bulletproofsplus_amd/csrc/transcript.hpp, restated with hashlib (TEST ORACLE for it).
    st0 = SHA-256(domain || curve id, n, m as u32 LE || SHA-256(pk wire bytes)); append / challenge as documented
    there.  Points enter as the (2L+1) x u64 little-endian wire words of include/bpp_amd.h."""

    DOMAIN = b"BulletproofsPlus-AMD transcript v1\0\0"

    def __init__(self, curve: dict, curve_id: int, n: int, m: int, pk):
        import hashlib
        self.h = hashlib.sha256
        self.curve = curve
        self.L = curve["fp_bytes"] // 8
        self.r = curve["r"]
        pkb = b"".join(self.point_bytes(P) for P in [pk.g, pk.h] + list(pk.G_vec) + list(pk.H_vec))
        hdr = self.DOMAIN + curve_id.to_bytes(4, "little") + n.to_bytes(4, "little") + m.to_bytes(4, "little")
        self.st0 = self.h(hdr + self.h(pkb).digest()).digest()
        self.st = self.st0

    def start(self):
        self.st = self.st0
        return self

    def point_bytes(self, P) -> bytes:
        """canonical bytes of a point in the transcript: wire words for the Weierstrass curves, the ristretto255
        encoding for the Edwards instantiation (an element there has four affine representatives)"""
        if self.curve["name"] == "ed25519":
            return Ristretto255.encode(P)
        nb = 8 * self.L
        if P is None:
            return bytes(2 * nb) + (1).to_bytes(8, "little")
        return P[0].to_bytes(nb, "little") + P[1].to_bytes(nb, "little") + bytes(8)

    def wire_bytes(self, P) -> bytes:
        """the C ABI's wire image of a point (what a proof record holds), whatever the transcript hashes"""
        nb = 8 * self.L
        if P is None:
            return bytes(2 * nb) + (1).to_bytes(8, "little")
        return P[0].to_bytes(nb, "little") + P[1].to_bytes(nb, "little") + bytes(8)

    def append(self, tag: bytes, data: bytes):
        self.st = self.h(self.st + tag.ljust(4, b"\0") + len(data).to_bytes(4, "little") + data).digest()

    def append_point(self, tag: bytes, P):
        self.append(tag, self.point_bytes(P))

    def challenge(self, tag: bytes) -> int:
        t4 = tag.ljust(4, b"\0")
        c0 = self.h(self.st + t4 + (0x80000000).to_bytes(4, "little")).digest()
        c1 = self.h(self.st + t4 + (0x80000001).to_bytes(4, "little")).digest()
        self.st = self.h(self.st + t4 + (0x80000002).to_bytes(4, "little")).digest()
        x = (int.from_bytes(c0, "little") + (int.from_bytes(c1, "little") << 256)) % self.r
        return x or 1

    # ---- the protocol's sequence -----------------------------------------------------------
    def yz(self, V_list, A):
        for V in V_list:
            self.append_point(b"V", V)
        self.append_point(b"A", A)
        return self.challenge(b"y"), self.challenge(b"z")

    def wip_start(self, mn: int):
        self.append(b"dsep", b"wipp v1\0")
        self.append(b"n", mn.to_bytes(8, "little"))

    def round(self, L, R) -> int:
        self.append_point(b"L", L)
        self.append_point(b"R", R)
        return self.challenge(b"e")

    def final(self, A, B) -> int:
        self.append_point(b"wA", A)
        self.append_point(b"wB", B)
        return self.challenge(b"e")

    def verifier_challenges(self, proof, commitment_vec):
        """[y, z, e, e_1..e_k] for a RangeProof, as bpp_verifier_derive_challenges computes them"""
        self.start()
        y, z = self.yz(list(commitment_vec), proof.A)
        w = proof.proof
        self.wip_start(1 << len(w.L_vec))
        es = [self.round(L, R) for L, R in zip(w.L_vec, w.R_vec)]
        e = self.final(w.A, w.B)
        return dict(y=y, z=z, e=e, e_rounds=es)


# --------------------------------------------------------------------------------------
# Weighted inner product proof (reference weighted_inner_product_proof.rs)
# --------------------------------------------------------------------------------------

class WeightedInnerProductProof:
    def __init__(self, L_vec, R_vec, A, B, r_prime, s_prime, d_prime):
        self.L_vec, self.R_vec, self.A, self.B = L_vec, R_vec, A, B
        self.r_prime, self.s_prime, self.d_prime = r_prime, s_prime, d_prime

    @staticmethod
    def prove(pk: PublicKey, a_vec, b_vec, power_of_y_vec, gamma, commitment, trace=None):
        # wip.rs:36-227
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        T = Transcript
        G = list(pk.G_vec)
        H = list(pk.H_vec)
        a = list(a_vec)
        b = list(b_vec)
        pw = list(power_of_y_vec)
        alpha = gamma
        n = len(G)
        assert len(H) == n and len(a) == n and len(b) == n and len(pw) == n
        assert n & (n - 1) == 0 and n > 0
        if T.run is not None:
            T.run.wip_start(n)
        L_vec, R_vec = [], []
        while n != 1:
            n //= 2
            a1, a2 = a[:n], a[n:]
            b1, b2 = b[:n], b[n:]
            y1, y2 = pw[:n], pw[n:]
            G1, G2 = G[:n], G[n:]
            H1, H2 = H[:n], H[n:]
            c_L = F.weighted_inner_product(a1, b2, y1)
            c_R = F.weighted_inner_product(a2, b1, y2)
            d_L, d_R = F.new(T.D_L), F.new(T.D_R)
            y_nhat = y1[n - 1]
            y_nhat_inv = F.inv(y_nhat)
            G1_exp = [y_nhat * x % r for x in a2]
            G2_exp = [y_nhat_inv * x % r for x in a1]

            mv = MulVec(G_)
            mv.add_scalars(G2_exp); mv.add_scalars(b2); mv.add_scalar(c_L); mv.add_scalar(d_L)
            mv.add_points(G2); mv.add_points(H1); mv.add_point(pk.g); mv.add_point(pk.h)
            L = mv.calculate()
            mv = MulVec(G_)
            mv.add_scalars(G1_exp); mv.add_scalars(b1); mv.add_scalar(c_R); mv.add_scalar(d_R)
            mv.add_points(G1); mv.add_points(H2); mv.add_point(pk.g); mv.add_point(pk.h)
            R = mv.calculate()
            L_vec.append(L)
            R_vec.append(R)

            e = T.run.round(L, R) if T.run is not None else F.new(T.e_round(len(L_vec) - 1))
            e_inv = F.inv(e)
            e_sqr = e * e % r
            e_sqr_inv = e_inv * e_inv % r
            # wip.rs:137-142 accumulates P, which is never read again: skipped.
            y_nhat_e_inv = y_nhat * e_inv % r
            y_nhat_inv_e = y_nhat_inv * e % r
            na, nb, nG, nH = [], [], [], []
            for i in range(n):
                na.append((a1[i] * e + a2[i] * y_nhat_e_inv) % r)
                nb.append((b1[i] * e_inv + b2[i] * e) % r)
                mv = MulVec(G_)
                mv.add_scalar(e_inv); mv.add_scalar(y_nhat_inv_e)
                mv.add_point(G1[i]); mv.add_point(G2[i])
                nG.append(mv.calculate())
                mv = MulVec(G_)
                mv.add_scalar(e); mv.add_scalar(e_inv)
                mv.add_point(H1[i]); mv.add_point(H2[i])
                nH.append(mv.calculate())
            a, b, pw, G, H = na, nb, y1, nG, nH
            alpha = (alpha + e_sqr * d_L + e_sqr_inv * d_R) % r
            if trace is not None:
                trace.append(dict(n=n, L=L, R=R, a=list(a), b=list(b), G=list(G), H=list(H),
                                  alpha=alpha))

        rr, s, delta, eta = F.new(T.R), F.new(T.S), F.new(T.DELTA), F.new(T.ETA)
        rcbsca = (rr * pw[0] * b[0] + s * pw[0] * a[0]) % r
        rcs = rr * pw[0] * s % r
        mv = MulVec(G_)
        mv.add_scalar(rr); mv.add_scalar(s); mv.add_scalar(rcbsca); mv.add_scalar(delta)
        mv.add_point(G[0]); mv.add_point(H[0]); mv.add_point(pk.g); mv.add_point(pk.h)
        A = mv.calculate()
        mv = MulVec(G_)
        mv.add_scalar(rcs); mv.add_scalar(eta)
        mv.add_point(pk.g); mv.add_point(pk.h)
        B = mv.calculate()
        e = T.run.final(A, B) if T.run is not None else F.new(T.E_FINAL)
        r_prime = (rr + a[0] * e) % r
        s_prime = (s + b[0] * e) % r
        d_prime = (eta + delta * e + alpha * e * e) % r
        return WeightedInnerProductProof(L_vec, R_vec, A, B, r_prime, s_prime, d_prime)

    def verification_scalars(self, n: int, G_):
        # wip.rs:330-382 ; returns None for the VerificationError branch at :335-337
        F = Fr(G_.r)
        r = G_.r
        logn = len(self.L_vec)
        if n != (1 << logn):
            return None
        challenges = [F.new(Transcript.e_round(i)) for i in range(logn)]
        allinv, challenges_inv = F.batch_invert(challenges)
        challenges_sqr = [c * c % r for c in challenges]
        challenges_inv_sqr = [c * c % r for c in challenges_inv]
        e = F.new(Transcript.e_final())
        s_vec = [allinv]
        for i in range(1, n):
            log_i = i.bit_length() - 1
            k = 1 << log_i
            u = challenges_sqr[(logn - 1) - log_i]
            s_vec.append(s_vec[i - k] * u % r)
        return challenges_sqr, challenges_inv_sqr, s_vec, e

    def verify_mulvec(self, pk: PublicKey, power_of_y_vec, G_exp_c, H_exp_c, g_exp_c, V_exp_c,
                      A_prime, V):
        """wip.rs:238-320: returns the assembled MulVec (or None on the error branch)."""
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        logn = len(self.L_vec)
        n = 1 << logn
        y = power_of_y_vec[0]
        vs = self.verification_scalars(n, G_)
        if vs is None:
            return None
        challenges_sqr, challenges_inv_sqr, s_vec, e = vs
        s_rev = s_vec[::-1]
        e_sqr = e * e % r
        r_prime_e_y = self.r_prime * e * y % r
        s_prime_e = self.s_prime * e % r
        Ls_exp = [c * e_sqr % r for c in challenges_sqr]
        Rs_exp = [c * e_sqr % r for c in challenges_inv_sqr]
        y_inv_pows = F.exp_iter_type2(F.inv(y), n)
        G_exp = [(-(s_vec[i]) * y_inv_pows[i] * r_prime_e_y + G_exp_c[i] * e_sqr) % r
                 for i in range(n)]
        H_exp = [(-(s_rev[i]) * s_prime_e + H_exp_c[i] * e_sqr) % r for i in range(n)]
        g_exp = (-self.r_prime * y * self.s_prime + g_exp_c * e_sqr) % r
        h_exp = (-self.d_prime) % r
        V_exp = [v * e_sqr % r for v in V_exp_c]
        mv = MulVec(G_)
        mv.add_scalar(F.new(1)); mv.add_scalar(e); mv.add_scalar(e_sqr)
        mv.add_scalar(g_exp); mv.add_scalar(h_exp)
        mv.add_scalars(Ls_exp); mv.add_scalars(Rs_exp)
        mv.add_scalars(G_exp); mv.add_scalars(H_exp); mv.add_scalars(V_exp)
        mv.add_point(self.B); mv.add_point(self.A); mv.add_point(A_prime)
        mv.add_point(pk.g); mv.add_point(pk.h)
        mv.add_points(self.L_vec); mv.add_points(self.R_vec)
        mv.add_points(pk.G_vec); mv.add_points(pk.H_vec); mv.add_points(list(V))
        return mv


# --------------------------------------------------------------------------------------
# RangeProof (reference range/mod.rs)
# --------------------------------------------------------------------------------------

class RangeProof:
    def __init__(self, A, proof: WeightedInnerProductProof):
        self.A = A
        self.proof = proof

    # ---- prove -----------------------------------------------------------------------
    @staticmethod
    def prove(pk: PublicKey, n: int, prover: RangeProver, trace=None):
        # range/mod.rs:31-55
        m = len(prover.v_vec)
        Transcript.cached = None
        Transcript.run = Transcript.fs.start() if Transcript.fs is not None else None
        if m == 1:
            return RangeProof._prove_single(pk, n, prover.v_vec[0], prover.gamma_vec[0],
                                            prover.commitment_vec[0], trace)
        return RangeProof._prove_multiple(pk, n, m, prover.v_vec, prover.gamma_vec,
                                          prover.commitment_vec, trace)

    @staticmethod
    def _prove_single(pk, n, v, gamma, commitment, trace=None):
        # range/mod.rs:80-187
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        T = Transcript
        assert len(pk.G_vec) == n and len(pk.H_vec) == n
        alpha = F.new(T.ALPHA_SINGLE)
        v_bits = []
        A = G_.mul(pk.h, alpha)
        for i in range(n):
            bit = (v >> i) & 1 if i < 64 else 0
            v_bits.append(bit)
            pt = pk.G_vec[i] if bit else G_.neg(pk.H_vec[i])
            A = G_.add(A, pt)
        if T.run is not None:
            y, z = T.run.yz([commitment], A)
        else:
            y, z = F.new(T.Y_SINGLE), F.new(T.Z_SINGLE)
        one, two = 1, 2
        power_of_two = F.exp_iter_type1(2, n)
        power_of_y = F.exp_iter_type2(y, n)
        power_of_y_rev = power_of_y[::-1]
        G_vec_sum = G_.zero()
        for p in pk.G_vec:
            G_vec_sum = G_.add(G_vec_sum, p)
        G_vec_sum_exp = (-z) % r
        H_exp = [(power_of_two[i] * power_of_y_rev[i] + z) % r for i in range(n)]
        V_exp = F.scalar_exp_vartime(y, n + 1)
        g_exp = sum(power_of_y) % r
        g_exp = g_exp * (z - z * z) % r
        g_exp = (g_exp - (F.scalar_exp_vartime(two, n) - one) * V_exp * z) % r
        mv = MulVec(G_)
        mv.add_scalar(F.new(1)); mv.add_scalar(G_vec_sum_exp); mv.add_scalars(H_exp)
        mv.add_scalar(g_exp); mv.add_scalar(V_exp)
        mv.add_point(A); mv.add_point(G_vec_sum); mv.add_points(pk.H_vec)
        mv.add_point(pk.g); mv.add_point(commitment)
        A_hat = mv.calculate()
        nz = (-z) % r
        one_minus_z = (one - z) % r
        a_vec = [one_minus_z if bbit else nz for bbit in v_bits]
        b_vec = [H_exp[i] if v_bits[i] else (H_exp[i] - one) % r for i in range(n)]
        alpha_hat = (alpha + gamma * V_exp) % r
        if trace is not None:
            trace.append(dict(stage="range", A=A, A_hat=A_hat, a_vec=a_vec, b_vec=b_vec,
                              alpha_hat=alpha_hat))
        proof = WeightedInnerProductProof.prove(pk, a_vec, b_vec, power_of_y, alpha_hat, A_hat,
                                                trace)
        return RangeProof(A, proof)

    @staticmethod
    def _prove_multiple(pk, n, m, v, gamma_vec, commitment_vec, trace=None):
        # range/mod.rs:240-403
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        T = Transcript
        mn = n * m
        assert len(pk.G_vec) == mn and len(pk.H_vec) == mn
        alpha = F.new(T.ALPHA_MULTI)
        v_bits = []
        A = G_.mul(pk.h, alpha)
        for i in range(mn):
            index1, index2 = i % n, i // n
            bit = (v[index2] >> index1) & 1 if index1 < 64 else 0
            v_bits.append(bit)
            pt = pk.G_vec[i] if bit else G_.neg(pk.H_vec[i])
            A = G_.add(A, pt)
        if T.run is not None:
            y, z = T.run.yz(list(commitment_vec), A)
        else:
            y, z = F.new(T.Y_MULTI), F.new(T.Z_MULTI)
        power_of_two = F.exp_iter_type1(2, n)
        power_of_y = F.exp_iter_type2(y, mn)
        power_of_y_rev = power_of_y[::-1]
        z_sqr = z * z % r
        power_of_z = F.exp_iter_type2(z_sqr, m)
        d = [e2 * ez % r for ez in power_of_z for e2 in power_of_two]
        G_vec_sum_exp = (-z) % r
        H_exp = [(d[i] * power_of_y_rev[i] + z) % r for i in range(mn)]
        y_mn1 = F.scalar_exp_vartime(y, mn + 1)
        V_exp = [pz * y_mn1 % r for pz in power_of_z]
        g_exp = sum(power_of_y) % r
        g_exp = g_exp * (z - z_sqr) % r
        d_sum = sum(d) % r
        g_exp = (g_exp - d_sum * y_mn1 * z) % r
        G_vec_sum = G_.zero()
        for p in pk.G_vec:
            G_vec_sum = G_.add(G_vec_sum, p)
        mv = MulVec(G_)
        mv.add_scalar(F.new(1)); mv.add_scalar(G_vec_sum_exp); mv.add_scalars(H_exp)
        mv.add_scalar(g_exp); mv.add_scalars(V_exp)
        mv.add_point(A); mv.add_point(G_vec_sum); mv.add_points(pk.H_vec)
        mv.add_point(pk.g); mv.add_points(list(commitment_vec))
        A_hat = mv.calculate()
        one = 1
        nz = (-z) % r
        one_minus_z = (one - z) % r
        a_vec = [one_minus_z if bbit else nz for bbit in v_bits]
        b_vec = [H_exp[i] if v_bits[i] else (H_exp[i] - one) % r for i in range(mn)]
        pzg = 0
        for pz, gm in zip(power_of_z, gamma_vec):
            pzg = (pzg + pz * gm) % r
        alpha_hat = (alpha + pzg * y_mn1) % r
        if trace is not None:
            trace.append(dict(stage="range", A=A, A_hat=A_hat, a_vec=a_vec, b_vec=b_vec,
                              alpha_hat=alpha_hat))
        proof = WeightedInnerProductProof.prove(pk, a_vec, b_vec, power_of_y, alpha_hat, A_hat,
                                                trace)
        return RangeProof(A, proof)

    # ---- verify ----------------------------------------------------------------------
    def verify_mulvec(self, pk: PublicKey, n: int, commitment_vec):
        """Returns the final MulVec of range/mod.rs:480-501 (m>1) or wip.rs:297-318 (m==1),
        or None when verification_scalars takes its error branch."""
        m = len(commitment_vec)
        Transcript.run = None
        Transcript.cached = (Transcript.fs.verifier_challenges(self, commitment_vec)
                             if Transcript.fs is not None else None)
        try:
            if m == 1:
                return self._verify_single_mv(pk, n, commitment_vec[0])
            return self._verify_multiple_mv(pk, n, m, commitment_vec)
        finally:
            Transcript.cached = None

    def verify(self, pk: PublicKey, n: int, commitment_vec) -> bool:
        # range/mod.rs:57-78 ; True = Ok(()), False = Err(VerificationError)
        mv = self.verify_mulvec(pk, n, commitment_vec)
        if mv is None:
            return False
        return getattr(pk.G, "is_identity_class", pk.G.is_zero)(mv.calculate())

    def _verify_single_mv(self, pk, n, commitment):
        # range/mod.rs:189-238
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        T = Transcript
        y, z = (F.new(c) for c in T.yz(1))
        one, two = 1, 2
        power_of_two = F.exp_iter_type1(2, n)
        power_of_y = F.exp_iter_type2(y, n)
        power_of_y_rev = power_of_y[::-1]
        G_exp = [(-z) % r] * n
        H_exp = [(power_of_two[i] * power_of_y_rev[i] + z) % r for i in range(n)]
        V_exp = F.scalar_exp_vartime(y, n + 1)
        g_exp = sum(power_of_y) % r
        g_exp = g_exp * (z - z * z) % r
        g_exp = (g_exp - (F.scalar_exp_vartime(two, n) - one) * V_exp * z) % r
        return self.proof.verify_mulvec(pk, power_of_y, G_exp, H_exp, g_exp, [V_exp], self.A,
                                        [commitment])

    def _verify_multiple_mv(self, pk, n, m, commitment_vec):
        # range/mod.rs:405-510
        G_ = pk.G
        F = Fr(G_.r)
        r = G_.r
        T = Transcript
        mn = n * m
        y, z = (F.new(c) for c in T.yz(m))
        minus_z = (-z) % r
        z_sqr = z * z % r
        power_of_two = F.exp_iter_type1(2, n)
        power_of_y = F.exp_iter_type2(y, mn + 1)
        y_mn1 = power_of_y.pop()
        power_of_y_rev = power_of_y[::-1]
        power_of_z = F.exp_iter_type2(z_sqr, m)
        concat_z_and_2 = [e2 * ez % r for ez in power_of_z for e2 in power_of_two]
        vs = self.proof.verification_scalars(mn, G_)
        if vs is None:
            return None
        challenges_sqr, challenges_inv_sqr, s_vec, e = vs
        s_rev = s_vec[::-1]
        e_inv = F.inv(e)
        e_sqr = e * e % r
        e_sqr_inv = F.inv(e_sqr)
        r_prime, s_prime, d_prime = self.proof.r_prime, self.proof.s_prime, self.proof.d_prime
        r_prime_e_inv_y = r_prime * e_inv * y % r
        s_prime_e_inv = s_prime * e_inv % r
        y_inv_pows = F.exp_iter_type2(F.inv(y), mn)
        G_exp = [(minus_z - s_vec[i] * y_inv_pows[i] * r_prime_e_inv_y) % r for i in range(mn)]
        H_exp = [(-s_prime_e_inv * s_rev[i] + (concat_z_and_2[i] * power_of_y_rev[i] + z)) % r
                 for i in range(mn)]
        sum_y = F.sum_of_powers_type2(y, mn)
        sum_2 = F.sum_of_powers_type1(F.new(2), n)
        sum_z = F.sum_of_powers_type2(z_sqr, m)
        g_exp = (-r_prime * s_prime * y * e_sqr_inv
                 + (sum_y * (z - z_sqr) - y_mn1 * z * sum_2 * sum_z)) % r
        h_exp = (-d_prime * e_sqr_inv) % r
        V_exp = [pz * y_mn1 % r for pz in power_of_z]
        mv = MulVec(G_)
        mv.add_scalar(F.new(1)); mv.add_scalar(e_inv); mv.add_scalar(e_sqr_inv)
        mv.add_scalar(g_exp); mv.add_scalar(h_exp)
        mv.add_scalars(challenges_sqr); mv.add_scalars(challenges_inv_sqr)
        mv.add_scalars(G_exp); mv.add_scalars(H_exp); mv.add_scalars(V_exp)
        mv.add_point(self.A); mv.add_point(self.proof.A); mv.add_point(self.proof.B)
        mv.add_point(pk.g); mv.add_point(pk.h)
        mv.add_points(self.proof.L_vec); mv.add_points(self.proof.R_vec)
        mv.add_points(pk.G_vec); mv.add_points(pk.H_vec); mv.add_points(list(commitment_vec))
        return mv


# --------------------------------------------------------------------------------------
# Convenience
# --------------------------------------------------------------------------------------

# ---- compressed point encodings (TEST ORACLE for csrc/codec.hpp; no reference counterpart: the reference has no
# serialization, only the commented-out size() fns of range/mod.rs:512-517 and wip.rs:384-397) ------------------
# BLS12-381 G1: 48 bytes, x big-endian, byte 0 bit 7 = compressed, bit 6 = infinity, bit 5 = y > (p-1)/2
# secp256k1   : SEC1, 33 bytes, 02/03 || x big-endian; infinity = 33 zero bytes (fixed-width variant)
def compress_point(curve: dict, P) -> bytes:
    p = curve["p"]
    if curve["name"] == "bls12_381":
        if P is None:
            return bytes([0xC0]) + bytes(47)
        x, y = P
        b = bytearray(x.to_bytes(48, "big"))
        b[0] |= 0x80 | (0x20 if y > (p - 1) // 2 else 0)
        return bytes(b)
    if curve["name"] == "secp256k1":
        if P is None:
            return bytes(33)
        x, y = P
        return bytes([2 + (y & 1)]) + x.to_bytes(32, "big")
    if curve["name"] == "ed25519":
        return Ristretto255.encode(P)
    raise ValueError("no compressed encoding for " + curve["name"])


def decompress_point(curve: dict, data: bytes):
    """-> (ok, point): ok False for a malformed encoding (flags, x >= p, x not on the curve)"""
    if curve["name"] == "ed25519":
        q = Ristretto255.decode(data)
        return (q is not None), q
    p, b = curve["p"], curve["b"]
    if curve["name"] == "bls12_381":
        assert len(data) == 48
        f = data[0]
        if not f & 0x80:
            return False, None
        x = int.from_bytes(bytes([f & 0x1F]) + data[1:], "big")
        if f & 0x40:
            return (x == 0 and not f & 0x20), None
        want = bool(f & 0x20)
    elif curve["name"] == "secp256k1":
        assert len(data) == 33
        if data == bytes(33):
            return True, None
        if data[0] not in (2, 3):
            return False, None
        x = int.from_bytes(data[1:], "big")
        want = data[0] == 3
    else:
        raise ValueError("no compressed encoding for " + curve["name"])
    if x >= p:
        return False, None
    rhs = (x * x * x + b) % p
    y = pow(rhs, (p + 1) // 4, p)      # p = 3 mod 4 for both fields
    if y * y % p != rhs:
        return False, None
    flag = (y > (p - 1) // 2) if curve["name"] == "bls12_381" else bool(y & 1)
    if flag != want:
        y = p - y
    return True, (x, y)


def make_group(curve_name: str, shadow: bool):
    c = CURVES[curve_name]
    if shadow:
        return ShadowGroup(c["r"])
    return EdwardsGroup(c) if curve_name == "ed25519" else WeierstrassGroup(c)


def prove_case(curve_name: str, n: int, values, gammas, shadow=True, trace=None):
    """Runs PublicKey::new(n*m), commit x m, prove.  Returns (pk, prover, proof)."""
    G = make_group(curve_name, shadow)
    pk = PublicKey(G, n * len(values))
    prover = RangeProver()
    for v, gm in zip(values, gammas):
        prover.commit(pk, v, gm % G.r)
    proof = RangeProof.prove(pk, n, prover, trace)
    return pk, prover, proof


# --------------------------------------------------------------------------------------
# ristretto255 (RFC 9496 section 4) -- TEST ORACLE for bulletproofsplus_amd/csrc/ristretto.hpp.  The reference has no
# Ristretto code; this follows the RFC's pseudocode over Python integers.
# --------------------------------------------------------------------------------------
class Ristretto255:
    P = (1 << 255) - 19
    D = (-121665 * pow(121666, -1, P)) % P
    SQRT_M1 = 19681161376707505956807079304988542015446066515923890162744021073123829784752
    SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
    INVSQRT_A_MINUS_D = 54469307008909316920995813868745141605393597292927456921205312896311721017578
    ONE_MINUS_D_SQ = (1 - D * D) % P
    D_MINUS_ONE_SQ = (D - 1) ** 2 % P
    BASE_ENCODING = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")

    @classmethod
    def is_neg(cls, x):
        return (x % cls.P) & 1

    @classmethod
    def ct_abs(cls, x):
        x %= cls.P
        return (-x) % cls.P if x & 1 else x

    @classmethod
    def sqrt_ratio_m1(cls, u, v):
        p = cls.P
        v3 = v * v * v % p
        v7 = v3 * v3 * v % p
        r = u * v3 * pow(u * v7, (p - 5) // 8, p) % p
        check = v * r * r % p
        correct = check == u % p
        flipped = check == (-u) % p
        flipped_i = check == (-u) * cls.SQRT_M1 % p
        if flipped or flipped_i:
            r = cls.SQRT_M1 * r % p
        return (correct or flipped), cls.ct_abs(r)

    @classmethod
    def decode(cls, b: bytes):
        """-> (x, y) representative or None"""
        p = cls.P
        s = int.from_bytes(b, "little")
        if s >= p or s & 1:
            return None
        ss = s * s % p
        u1, u2 = (1 - ss) % p, (1 + ss) % p
        u2s = u2 * u2 % p
        v = (-(cls.D * u1 * u1) - u2s) % p
        ok, inv = cls.sqrt_ratio_m1(1, v * u2s % p)
        den_x = inv * u2 % p
        den_y = inv * den_x * v % p
        x = cls.ct_abs(2 * s * den_x)
        y = u1 * den_y % p
        t = x * y % p
        if not ok or t & 1 or y == 0:
            return None
        return (x, y)

    @classmethod
    def encode(cls, P_) -> bytes:
        """affine (x, y) of the even subgroup (None = the identity (0, 1)) -> 32 bytes"""
        p = cls.P
        x0, y0 = (0, 1) if P_ is None else P_
        z0, t0 = 1, x0 * y0 % p
        u1 = (z0 + y0) * (z0 - y0) % p
        u2 = x0 * y0 % p
        _, inv = cls.sqrt_ratio_m1(1, u1 * u2 * u2 % p)
        den1, den2 = inv * u1 % p, inv * u2 % p
        z_inv = den1 * den2 * t0 % p
        ix0, iy0 = x0 * cls.SQRT_M1 % p, y0 * cls.SQRT_M1 % p
        ench = den1 * cls.INVSQRT_A_MINUS_D % p
        rotate = cls.is_neg(t0 * z_inv)
        x, y, den_inv = (iy0, ix0, ench) if rotate else (x0, y0, den2)
        if cls.is_neg(x * z_inv):
            y = (-y) % p
        return cls.ct_abs(den_inv * (z0 - y)).to_bytes(32, "little")

    @classmethod
    def equal(cls, A, B):
        p = cls.P
        A = (0, 1) if A is None else A
        B = (0, 1) if B is None else B
        return (A[0] * B[1] - A[1] * B[0]) % p == 0 or (A[1] * B[1] - A[0] * B[0]) % p == 0

    @classmethod
    def map(cls, t):
        """MAP of RFC 9496 4.3.4 -> extended (X, Y, Z, T)"""
        p = cls.P
        r = cls.SQRT_M1 * t * t % p
        u = (r + 1) * cls.ONE_MINUS_D_SQ % p
        v = (-1 - r * cls.D) * (r + cls.D) % p
        ok, s = cls.sqrt_ratio_m1(u, v)
        s_prime = (-cls.ct_abs(s * t)) % p
        s = s if ok else s_prime
        c = (p - 1) if ok else r
        N = (c * (r - 1) * cls.D_MINUS_ONE_SQ - v) % p
        w0 = 2 * s * v % p
        w1 = N * cls.SQRT_AD_MINUS_ONE % p
        w2, w3 = (1 - s * s) % p, (1 + s * s) % p
        return (w0 * w3 % p, w2 * w1 % p, w1 * w3 % p, w0 * w2 % p)

    @classmethod
    def from_uniform_bytes(cls, b: bytes, G):
        """64 bytes -> affine point (G: EdwardsGroup for the addition)"""
        p = cls.P
        out = []
        for h in range(2):
            t = int.from_bytes(b[32 * h:32 * h + 32], "little") & ((1 << 255) - 1)
            X, Y, Z, _ = cls.map(t % p)
            zi = pow(Z, -1, p)
            out.append((X * zi % p, Y * zi % p))
        return G.add(out[0], out[1])


# --------------------------------------------------------------------------------------
# Serialized proofs -- TEST ORACLE for the container of include/bpp_amd.h (bpp_proofs_encode / decode).  The reference
# has no serialization (commented-out size() fns only, range/mod.rs:512-517, wip.rs:384-397): parity unpinned.
#   "BPP+" | version 1 | curve id | n | m | k | 0 0 0 | (3 + 2k) compressed points | r', s', delta' (32 bytes LE each)
# --------------------------------------------------------------------------------------
CURVE_IDS = {"bls12_381": 0, "secp256k1": 1, "ed25519": 2}


def uncompressed_point(curve: dict, P) -> bytes:
    """container version 2: BLS12-381 G1 96 bytes x | y big-endian (byte 0 bit 6 = infinity), secp256k1 SEC1 04 | x | y
    (infinity = 65 zero bytes)"""
    if curve["name"] == "bls12_381":
        if P is None:
            return bytes([0x40]) + bytes(95)
        return P[0].to_bytes(48, "big") + P[1].to_bytes(48, "big")
    if curve["name"] == "secp256k1":
        if P is None:
            return bytes(65)
        return b"\x04" + P[0].to_bytes(32, "big") + P[1].to_bytes(32, "big")
    raise ValueError("no uncompressed encoding for " + curve["name"])


def parse_uncompressed_point(curve: dict, data: bytes):
    """-> (ok, point): ok False for a malformed encoding (flags / prefix, a coordinate >= p, not on the curve)"""
    p, b = curve["p"], curve["b"]
    if curve["name"] == "bls12_381":
        assert len(data) == 96
        f = data[0]
        if f & 0xA0:
            return False, None
        x = int.from_bytes(bytes([f & 0x1F]) + data[1:48], "big")
        y = int.from_bytes(data[48:], "big")
        if f & 0x40:
            return (x == 0 and y == 0), None
    else:
        assert len(data) == 65
        if data == bytes(65):
            return True, None
        if data[0] != 4:
            return False, None
        x, y = int.from_bytes(data[1:33], "big"), int.from_bytes(data[33:], "big")
    if x >= p or y >= p or (x == 0 and y == 0) or (y * y - x * x * x - b) % p:
        return False, None
    return True, (x, y)


def encode_proof(curve: dict, n: int, m: int, proof, version: int = 1) -> bytes:
    w = proof.proof
    k = len(w.L_vec)
    out = b"BPP+" + bytes([version, CURVE_IDS[curve["name"]], n, m, k, 0, 0, 0])
    for P in [proof.A, w.A, w.B] + list(w.L_vec) + list(w.R_vec):
        out += compress_point(curve, P) if version == 1 else uncompressed_point(curve, P)
    for x in (w.r_prime, w.s_prime, w.d_prime):
        out += int(x).to_bytes(32, "little")
    return out


def point_in_prime_subgroup(curve: dict, G, P) -> bool:
    """the DEFINITION ([r] P == O), independent of the endomorphism shortcut the engine uses for BLS12-381 G1"""
    if P is None or curve["name"] != "bls12_381":
        return True
    return G.is_zero(G.mul(P, curve["r"]))


def decode_proof(curve: dict, G, n: int, m: int, data: bytes, version: int = 1):
    """-> RangeProof, or None for ProofError::FormatError"""
    mn = n * m
    k = mn.bit_length() - 1
    cb = ({"bls12_381": 48, "secp256k1": 33, "ed25519": 32} if version == 1 else {"bls12_381": 96, "secp256k1": 65})[curve["name"]]
    if len(data) != 12 + (3 + 2 * k) * cb + 96:
        return None
    if data[:12] != b"BPP+" + bytes([version, CURVE_IDS[curve["name"]], n, m, k, 0, 0, 0]):
        return None
    pts = []
    for i in range(3 + 2 * k):
        chunk = data[12 + i * cb: 12 + (i + 1) * cb]
        ok, P = decompress_point(curve, chunk) if version == 1 else parse_uncompressed_point(curve, chunk)
        if not ok or not point_in_prime_subgroup(curve, G, P):
            return None
        pts.append(P)
    off = 12 + (3 + 2 * k) * cb
    sc = [int.from_bytes(data[off + 32 * t: off + 32 * t + 32], "little") for t in range(3)]
    if any(x >= curve["r"] for x in sc):
        return None
    return RangeProof(pts[0], WeightedInnerProductProof(pts[3:3 + k], pts[3 + k:3 + 2 * k], pts[1], pts[2], *sc))


# --------------------------------------------------------------------------------------
# Hashed generators -- TEST ORACLE for bulletproofsplus_amd/csrc/hash_to_group.hpp (no reference counterpart: the
# reference's PublicKey::new only has test generators, publickey.rs:23-39)
# --------------------------------------------------------------------------------------
def hash_to_group(curve: dict, G, label: bytes, kind: str, idx: int):
    import hashlib
    cid = CURVE_IDS[curve["name"]]
    seed = hashlib.sha256(b"BulletproofsPlus-AMD generators v1\0\0" + cid.to_bytes(4, "little") + label).digest()

    def H(ctr, half):
        return hashlib.sha256(seed + b"bppg" + ord(kind).to_bytes(4, "little") + idx.to_bytes(4, "little") +
                              ctr.to_bytes(4, "little") + half.to_bytes(4, "little")).digest()

    if curve["name"] == "ed25519":
        return Ristretto255.from_uniform_bytes(H(0, 0) + H(0, 1), G)
    p, b = curve["p"], curve["b"]
    ctr = 0
    while True:
        x = (int.from_bytes(H(ctr, 0), "little") + (int.from_bytes(H(ctr, 1), "little") << 256)) % p
        rhs = (x * x * x + b) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs:
            if (y & 1) != (H(ctr, 2)[0] & 1):
                y = p - y
            P = (x, y)
            if curve["name"] == "bls12_381":
                P = G.mul(P, 0xd201000000010001)
                if P is None:
                    ctr += 1
                    continue
            return P
        ctr += 1
