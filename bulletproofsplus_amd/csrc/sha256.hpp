// sha256.hpp -- SHA-256 (FIPS 180-4) and HMAC-SHA-256 (RFC 2104) for gfx950 and the host side of the library.
//
// The reference carries hand-written SHA-256 / SHA-512 / HMAC with known-answer tests
// (src/secp256k1/building_block/hasher/sha256.rs:37-89 and :96-144, hmac.rs:6-48 and :56-87) but never calls
// them: they are the remains of a planned Fiat-Shamir transcript (merlin is listed in Cargo.toml:16 and never
// imported; every challenge is a constant).  Here the hash does have callers: the transcript
// that replaces those constants (csrc/transcript.hpp) and the weights of the combined batch check
// (csrc/combined.hpp).  Pinned by the reference's own KATs, transcribed as data into tests/golden/sha256_kat.json
// (checked on the host build and on the device).
//
// One lane hashes one message: the state is 8 + 16 registers and a block costs 64 rounds of ~20 integer
// instructions, all full-rate VALU -- at 128 bytes of input per mixed addition's worth of time the hash is never
// the bottleneck next to the group arithmetic, so there is no cross-lane cleverness here.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "field.hpp"

namespace bpp {

struct Sha256 {
    uint32_t h[8];
    uint32_t w[16];     // the pending block, big-endian words
    uint32_t fill;      // bytes of the pending block that are filled (0..63)
    uint64_t total;     // bytes absorbed so far
};

namespace sha {

BPP_HD uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const uint32_t K256[64] = {
#else
static const uint32_t K256[64] = {
#endif
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// one compression: s.h <- s.h + F(s.h, s.w); s.w is consumed (used as the rolling message schedule).
// Not inlined on the device: the transcript calls it from dozens of places.
BPP_HD_NOINLINE void compress(Sha256& s) {
    uint32_t a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], hh = s.h[7];
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = s.w[i];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        if (t >= 16) {
            const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
            const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
        }
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K256[t] + w[t & 15];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    s.h[0] += a;
    s.h[1] += b;
    s.h[2] += c;
    s.h[3] += d;
    s.h[4] += e;
    s.h[5] += f;
    s.h[6] += g;
    s.h[7] += hh;
}

}  // namespace sha

BPP_HD void sha256_init(Sha256& s) {
    s.h[0] = 0x6a09e667;
    s.h[1] = 0xbb67ae85;
    s.h[2] = 0x3c6ef372;
    s.h[3] = 0xa54ff53a;
    s.h[4] = 0x510e527f;
    s.h[5] = 0x9b05688c;
    s.h[6] = 0x1f83d9ab;
    s.h[7] = 0x5be0cd19;
#pragma unroll
    for (int i = 0; i < 16; i++) s.w[i] = 0;
    s.fill = 0;
    s.total = 0;
}

BPP_HD void sha256_byte(Sha256& s, uint8_t x) {
    const uint32_t wi = s.fill >> 2, sh = 24 - 8 * (s.fill & 3);
    // dynamic word index: a small select chain keeps s.w in registers on the device
    uint32_t v = (uint32_t)x << sh;
#pragma unroll
    for (int i = 0; i < 16; i++)
        if ((uint32_t)i == wi) s.w[i] |= v;
    s.fill++;
    s.total++;
    if (s.fill == 64) {
        sha::compress(s);
#pragma unroll
        for (int i = 0; i < 16; i++) s.w[i] = 0;
        s.fill = 0;
    }
}

BPP_HD void sha256_update(Sha256& s, const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) sha256_byte(s, p[i]);
}

// absorbs one 32-bit word as 4 bytes, little-endian (the byte order of the wire format's u64 limbs); when the
// pending block is word aligned -- every caller here keeps it so -- this is one store instead of four byte steps
BPP_HD void sha256_word_le(Sha256& s, uint32_t x) {
    if ((s.fill & 3) == 0) {
        const uint32_t be = (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
        const uint32_t wi = s.fill >> 2;
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((uint32_t)i == wi) s.w[i] = be;
        s.fill += 4;
        s.total += 4;
        if (s.fill == 64) {
            sha::compress(s);
#pragma unroll
            for (int i = 0; i < 16; i++) s.w[i] = 0;
            s.fill = 0;
        }
    } else {
        sha256_byte(s, (uint8_t)x);
        sha256_byte(s, (uint8_t)(x >> 8));
        sha256_byte(s, (uint8_t)(x >> 16));
        sha256_byte(s, (uint8_t)(x >> 24));
    }
}

// digest as 8 big-endian words (out[0] holds the first four bytes of the digest)
BPP_HD_NOINLINE void sha256_final(Sha256& s, uint32_t out[8]) {
    const uint64_t bits = s.total * 8;
    sha256_byte(s, 0x80);
    while (s.fill != 56) sha256_byte(s, 0);
    s.w[14] = (uint32_t)(bits >> 32);
    s.w[15] = (uint32_t)bits;
    sha::compress(s);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = s.h[i];
}

BPP_HD void sha256_digest_bytes(const uint32_t dg[8], uint8_t out[32]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(dg[i] >> 24);
        out[4 * i + 1] = (uint8_t)(dg[i] >> 16);
        out[4 * i + 2] = (uint8_t)(dg[i] >> 8);
        out[4 * i + 3] = (uint8_t)dg[i];
    }
}

// HMAC-SHA-256 (RFC 2104; reference src/secp256k1/building_block/hasher/hmac.rs:6-48): keys longer than the
// block are hashed first
BPP_HD void hmac_sha256(const uint8_t* key, size_t klen, const uint8_t* msg, size_t mlen, uint32_t out[8]) {
    uint8_t k0[64];
    for (int i = 0; i < 64; i++) k0[i] = 0;
    if (klen > 64) {
        Sha256 s;
        sha256_init(s);
        sha256_update(s, key, klen);
        uint32_t d[8];
        sha256_final(s, d);
        sha256_digest_bytes(d, k0);
    } else {
        for (size_t i = 0; i < klen; i++) k0[i] = key[i];
    }
    Sha256 in;
    sha256_init(in);
    for (int i = 0; i < 64; i++) sha256_byte(in, k0[i] ^ 0x36);
    sha256_update(in, msg, mlen);
    uint32_t di[8];
    sha256_final(in, di);
    uint8_t dib[32];
    sha256_digest_bytes(di, dib);
    Sha256 o;
    sha256_init(o);
    for (int i = 0; i < 64; i++) sha256_byte(o, k0[i] ^ 0x5c);
    sha256_update(o, dib, 32);
    sha256_final(o, out);
}

}  // namespace bpp
