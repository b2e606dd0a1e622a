"""-m gpu: the device-resident general MulVec (bpp_msm_device, csrc/pippenger.hpp) -- MulVec::calculate
(reference src/bls12_381/building_block/mulvec.rs:20-33, secp256k1 twin secp256k1/util.rs:22-36) at sizes far beyond
what the reference itself forms.  Checked (i) bit for bit against the C oracle's naive MulVec at a size it finishes in
seconds, at every window width; (ii) at N = 2^16 through the size-independent property the reference's generators
offer: every point is a known multiple k_i g (publickey.rs:23-39 builds its generators the same way), so
sum_i s_i (k_i g) == (sum_i s_i k_i mod r) g; plus the hard cases of a bucket method: all scalars equal (one heavy
bucket per window), zero scalars, points at infinity, P and -P in one bucket, scalars >= r, an off-curve point."""

import random

import numpy as np
import pytest

import oracle as O
import pyref as P
from gpu_util import need_gpu

pytestmark = pytest.mark.gpu

CURVES = [("bls12_381", 0), ("secp256k1", 1), ("ed25519", 2)]


def to_words(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for t in range(4):
            out[i, t] = (int(v) >> (64 * t)) & 0xFFFFFFFFFFFFFFFF
    return out


def msm_dev(torch, B, a, scalars, points, c=0, want_status=False):
    """scalars (n, 4) u64, points (n, PW) u64 numpy -> wire point numpy, through bpp_msm_device"""
    dev = torch.device("cuda:0")
    n = scalars.shape[0]
    d_sc = torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).to(dev) if n else None
    d_pt = torch.from_numpy(np.ascontiguousarray(points).view(np.int64)).to(dev) if n else None
    d_out = torch.full((a.PW,), -1, dtype=torch.int64, device=dev)
    d_st = torch.full((1,), 7, dtype=torch.int32, device=dev)
    wsb = B.msm_workspace_bytes(a, n, c)
    assert wsb > 0
    d_ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    B.msm_device(a, d_sc.data_ptr() if n else 0, d_pt.data_ptr() if n else 0, n, d_out.data_ptr(), d_ws.data_ptr(), wsb,
                 window_bits=c, d_status=d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(np.uint64)
    return (res, int(d_st.item())) if want_status else res


def oracle_msm(cname, cid, scalar_ints, pts):
    """the checker: the C oracle's naive MulVec (Weierstrass curves) / the big-integer restatement (edwards25519)"""
    if cid != 2:
        return O.msm(cid, to_words(scalar_ints), pts)
    G = P.EdwardsGroup(P.ED25519)
    mv = P.MulVec(G)
    mv.add_scalars([s % G.r for s in scalar_ints])
    mv.add_points(O.wire_to_points(cid, pts))
    return O.point_to_wire(cid, mv.calculate())


def neg_wire(cname, a, pt):
    """-P of a wire point: (x, p - y) on the Weierstrass curves, (p - x, y) on edwards25519"""
    p = P.CURVES[cname]["p"]
    L = a.L
    out = pt.copy()
    lo, hi = (0, L) if cname == "ed25519" else (L, 2 * L)
    v = sum(int(pt[lo + t]) << (64 * t) for t in range(L))
    v = (p - v) % p
    for t in range(L):
        out[lo + t] = (v >> (64 * t)) & 0xFFFFFFFFFFFFFFFF
    return out


@pytest.mark.parametrize("cname,cid", CURVES)
def test_msm_device_matches_oracle_every_window(cname, cid):
    """n = 300 with collisions, infinity, zero / tiny / huge scalars == the oracle's naive MulVec, window widths 2..16"""
    torch = need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init(cid)
    r = P.CURVES[cname]["r"]
    rnd = random.Random(1234 + cid)
    pk = B.PublicKey.new(a, 100)          # checked against the oracle in test_gpu_primitives / test_gpu_ed25519
    base = np.concatenate([pk.gh, pk.G_vec, pk.H_vec, O.points_to_wire(cid, [None])])
    n = 300
    pts = base[[rnd.randrange(base.shape[0]) for _ in range(n)]].copy()
    scs = [rnd.randrange(r) for _ in range(n)]
    scs[:8] = [0, 1, 2, r - 1, r - 2, 5, r - 5, (1 << 255) % r]
    pts[6] = pts[5]
    pts[11] = neg_wire(cname, a, pts[10])     # P and -P ...
    scs[11] = scs[10]                          # ... with the same scalar: the same bucket in every window
    sw = to_words(scs)
    exp = oracle_msm(cname, cid, scs, pts)
    for c in (0, 2, 3, 5, 7, 8, 11, 13, 14, 16):
        got, st = msm_dev(torch, B, a, sw, pts, c, want_status=True)
        assert st == 0
        assert np.array_equal(got, exp), c
    # scalars >= r are reduced on the device (PrimeFieldElem values are always < r)
    big = sw.copy()
    big[0] = to_words([scs[0] + r])[0]
    big[1] = to_words([(1 << 256) - 1])[0]
    exp2 = oracle_msm(cname, cid, [scs[0] % r, ((1 << 256) - 1) % r] + scs[2:], pts)
    assert np.array_equal(msm_dev(torch, B, a, big, pts, 0), exp2)
    # empty and single-term MulVecs
    assert a.is_zero(msm_dev(torch, B, a, np.zeros((0, 4), np.uint64), np.zeros((0, a.PW), np.uint64)))
    assert np.array_equal(msm_dev(torch, B, a, to_words([7]), pts[9:10], 4), oracle_msm(cname, cid, [7], pts[9:10]))
    # an off-curve point raises the status word (and counts as infinity)
    bad = pts.copy()
    bad[20, 0] ^= 1
    got, st = msm_dev(torch, B, a, sw, bad, 0, want_status=True)
    assert st == 1
    bad[20] = O.points_to_wire(cid, [None])[0]
    assert np.array_equal(got, oracle_msm(cname, cid, scs, bad))
    # ... and the host-pointer call reports it as an error
    bad[20] = pts[20]
    bad[20, 0] ^= 1
    with pytest.raises(B.BppError):
        B.msm_pippenger(a, sw, bad, 0)


@pytest.mark.parametrize("cname,cid", CURVES)
def test_msm_device_2_16_dlog_identity_and_hard_cases(cname, cid):
    torch = need_gpu()
    import bulletproofsplus_amd as B
    a = B.Arith.init(cid)
    r = P.CURVES[cname]["r"]
    g = B.PublicKey.new(a, 0).gh[0]
    n = 1 << 16
    rng = np.random.RandomState(99 + cid)
    ks = rng.randint(1, 2**62, size=n).astype(np.uint64)          # distinct with overwhelming probability
    kw = np.zeros((n, 4), dtype=np.uint64)
    kw[:, 0] = ks
    pts = a.scalar_mul(kw, np.broadcast_to(g, (n, a.PW)).copy())
    sc = rng.randint(0, 2**63 - 1, size=(n, 4)).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=(n, 4)).astype(np.uint64)
    sc[:, 3] >>= np.uint64(4)                                      # < 2^252: below every curve's group order

    def expect(scalars, mult, cnt=n):
        tot = 0
        for i in range(cnt):
            s = int(scalars[i, 0]) | int(scalars[i, 1]) << 64 | int(scalars[i, 2]) << 128 | int(scalars[i, 3]) << 192
            tot = (tot + s * int(mult[i])) % r
        return a.scalar_mul(to_words([tot]), g[None])[0]

    ki = [int(k) for k in ks]
    # full-width scalars, distinct points, default and two explicit window widths
    e0 = expect(sc, ki)
    for c in (0, 12, 16):
        assert np.array_equal(msm_dev(torch, B, a, sc, pts, c), e0), c
    # all scalars equal: every window has ONE bucket holding all 65 536 points (the heavy-bucket path)
    s1 = np.broadcast_to(sc[7], (n, 4)).copy()
    assert np.array_equal(msm_dev(torch, B, a, s1, pts, 0), expect(s1, ki))
    # half of the scalars zero, a quarter of the points at infinity
    s2 = sc.copy()
    s2[::2] = 0
    p2 = pts.copy()
    inf = a.zero_point()
    p2[1::4] = inf
    k2 = [0 if (i % 4 == 1) else ki[i] for i in range(n)]
    assert np.array_equal(msm_dev(torch, B, a, s2, p2, 0), expect(s2, k2))
    # P and -P with the same scalar land in one bucket with opposite signs... (pairs 2i, 2i+1): everything cancels
    p3 = pts.copy()
    s3 = sc.copy()
    for i in range(0, 4096, 2):
        p3[i + 1] = neg_wire(cname, a, p3[i])
        s3[i + 1] = s3[i]
    k3 = list(ki)
    for i in range(0, 4096, 2):
        k3[i + 1] = r - ki[i]
    assert np.array_equal(msm_dev(torch, B, a, s3, p3, 0), expect(s3, k3))
    # ... and the same point twice in one bucket (a doubling inside the bucket sum)
    p4 = pts.copy()
    s4 = sc.copy()
    p4[1:4096:2] = p4[0:4096:2]
    s4[1:4096:2] = s4[0:4096:2]
    k4 = list(ki)
    for i in range(0, 4096, 2):
        k4[i + 1] = ki[i]
    assert np.array_equal(msm_dev(torch, B, a, s4, p4, 0), expect(s4, k4))
    # the host-pointer MulVec takes the same path from n = 4096 up
    assert np.array_equal(B.msm_batch(a, sc[:5000], pts[:5000], [5000])[0], expect(sc, ki, 5000))
