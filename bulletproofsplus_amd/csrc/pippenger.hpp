// pippenger.hpp -- bucket-method MulVec for LARGE variable-base inputs on gfx950, device resident.
//
// The reference's MulVec::calculate (src/bls12_381/building_block/mulvec.rs:20-33; secp256k1 twin
// src/secp256k1/building_block/secp256k1/util.rs:22-36) is one scalar multiplication per term.  For the sizes the
// reference itself produces (N <= 2 089) the data-parallel restatement in kernels.hpp (k_msm_naive_partial) already
// finishes in one scalar-mult latency; this file is the path for N in the thousands and up: bpp_msm_device /
// bpp_msm on big inputs.  Every buffer lives in HBM; nothing here synchronises with the host.
//
// Scalars.  On BLS12-381 every scalar is first split with G1's endomorphism (GLV, ec.hpp glv_split: k = k1 + k2 z^2,
// both halves < 2^128, [z^2] P = (beta x, -y)): an input point becomes TWO items (P with k1, phi(P) with k2) of 128-bit
// sub-scalars -- the same number of bucket additions, but half the windows, so half the buckets to reduce and half
// the ~255 sequential doublings of the Horner tail.  The other curves run one item of Fr::BITS bits per point.
// A sub-scalar is cut into W = floor((bits - 1) / c) + 1 windows: signed c-bit digits (digit j of value + bias,
// minus 2^(c-1)) below, and an UNSIGNED top window that takes what is left of the value, so no window is spent on the
// carry of the recoding (the top window has `top` buckets instead of 2^(c-1); flat bucket index j * half + b).
//
// Pipeline (one stream, no host round trips):
//   k_pip_points   wire points -> Montgomery affine (+ on-curve check) and, with GLV, the endomorphism image
//   k_pip_digits   per point: reduce mod r, split, W digits per item -> key (bucket, sign); bucket histogram with
//                  returning atomics, the returned value is the item's slot inside its bucket
//   k_pip_scan     per window: exclusive prefix sum of the histogram (LDS, one block per window)
//   k_pip_scatter  per (window, item): sorted[offset[bucket] + slot] = item | sign
//   k_pip_buckets  per (window, bucket): XYZZ running sum of its points, gathered through the LDS-DMA ring of
//                  k_fixed_msm; buckets with more than `heavy` points are deferred to
//   k_pip_heavy / k_pip_heavy_fold   which split one bucket over PIP_SPLIT x 128 lanes
//   k_pip_tiles    per tile of 64 S consecutive buckets, ONE WAVE: every lane runs the running sums over its S
//                  buckets (sum B_b and sum (b - b0 + 1) B_b), then the lanes' partials are combined inside the
//                  wave with shuffles: a suffix scan of the lane totals (ds_bpermute of the limbs, 6 steps) gives
//                  sum_l l * run_l, a butterfly reduces the rest.  Out: (A_t, T_t) per tile.
//   k_pip_windows  per window: R_j = sum_t A_t + (64 S) sum_t t T_t  (lane-local double-and-add by the tile number,
//                  wave butterfly)
//   k_pip_final    Horner over the W window sums (one wave, a tree over the windows), then the affine wire point
#pragma once
#include <algorithm>

#include "kernels.hpp"

namespace bpp {

template <class C>
constexpr bool pip_glv() {
    return C::ID == 0;
}

struct PipShape {
    uint32_t n;                // input points
    uint32_t items;            // (point, sub-scalar) pairs: n, or 2 n with the endomorphism split
    uint32_t glv;
    uint32_t c, W, half;       // window bits, windows, buckets of an ordinary window 2^(c-1)
    uint32_t top;              // buckets of the top window (its digit is unsigned: 1..top)
    uint32_t nbuckets;         // (W - 1) half + top
    uint32_t S, TS;            // buckets per lane / per tile (64 S) in k_pip_tiles
    uint32_t tiles_lo, tiles_top, ntiles;
    uint32_t heavy;            // a bucket with more points than this is split over many lanes
    uint32_t bias[10];         // sum_{j < W-1} half * 2^(c j)
};

// max_words: the largest sub-scalar value (8 words); bits: its bit length
inline int pip_shape(size_t n, int c, bool glv, const uint32_t* max_words, int bits, PipShape& s) {
    if (c < 2 || c > 16) return fail(BPP_E_ARG, "window_bits must be in [2, 16]");
    s.n = (uint32_t)n;
    s.glv = glv ? 1u : 0u;
    s.items = (uint32_t)(glv ? 2 * n : n);
    s.c = (uint32_t)c;
    s.W = (uint32_t)((bits - 1) / c + 1);
    s.half = 1u << (c - 1);
    for (int t = 0; t < 10; t++) s.bias[t] = 0;
    for (uint32_t j = 0; j + 1 < s.W; j++) {
        const uint32_t bit = s.c * j + (s.c - 1);
        s.bias[bit >> 5] |= 1u << (bit & 31);
    }
    uint32_t v[10];
    uint32_t carry = 0;
    for (int t = 0; t < 10; t++) {
        uint64_t x = (uint64_t)(t < 8 ? max_words[t] : 0u) + s.bias[t] + carry;
        v[t] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    const uint32_t sh = s.c * (s.W - 1);
    uint64_t top = 0;
    for (int t = 9; t >= 0; t--) {
        const int lo = 32 * t - (int)sh;
        if (lo >= 32 && v[t]) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
        if (lo > -32 && lo < 32) top |= lo >= 0 ? (uint64_t)v[t] << lo : (uint64_t)(v[t] >> (-lo));
    }
    if (top == 0 || top > ((uint64_t)1 << 17)) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
    s.top = (uint32_t)top;
    s.nbuckets = (s.W - 1) * s.half + s.top;
    // tiles: enough of them to spread a window over the chip, few enough that k_pip_windows stays short
    uint32_t S = s.half / 4096;
    S = S < 1 ? 1 : (S > 8 ? 8 : S);
    s.S = S;
    s.TS = 64 * S;
    s.tiles_lo = (s.half + s.TS - 1) / s.TS;
    s.tiles_top = (s.top + s.TS - 1) / s.TS;
    s.ntiles = (s.W - 1) * s.tiles_lo + s.tiles_top;
    // heavy = far above the load of a uniformly filled window
    s.heavy = (uint32_t)std::max<size_t>(96, 6 * ((size_t)s.items / s.half + 1));
    return BPP_OK;
}

template <class C>
inline int pip_shape_for(size_t n, int c, PipShape& s) {
    if (pip_glv<C>()) {
        const uint32_t mx[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0};
        return pip_shape(n, c, true, mx, 128, s);
    }
    uint32_t mx[8];
    for (int t = 0; t < 8; t++) mx[t] = C::Fr::MODW[t];   // r - 1 would do; r is as good a bound
    return pip_shape(n, c, false, mx, C::Fr::BITS, s);
}

// window width for n points.  Large inputs: about 2^7..2^8 items per bucket (the bucket additions dominate, the
// reduction of (W-1) 2^(c-1) buckets stays a few per cent).  Small inputs are latency bound -- the chain is the
// per-lane bucket sum, then the tile / window reduction, then ~c (W-1) doublings -- and want MORE, shorter buckets.
template <class C>
inline int pip_pick_c(size_t n) {
    const size_t items = pip_glv<C>() ? 2 * n : n;
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= items) lg++;
    int c = lg - 5;
    if (lg <= 18) c = lg - 3;
    return c < 7 ? 7 : (c > 16 ? 16 : c);
}

constexpr uint32_t PIP_EMPTY = 0xffffffffu;
constexpr uint32_t PIP_SPLIT = 16;

// wave-wide exchange of a whole struct of 32-bit words (ds_bpermute per word; no LDS memory is touched)
template <class T>
__device__ __forceinline__ T wave_shfl(const T& v, int src_lane) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    struct Words {
        uint32_t w[sizeof(T) / 4];
    };
    Words a = __builtin_bit_cast(Words, v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) a.w[i] = (uint32_t)__shfl((int)a.w[i], src_lane, 64);
    return __builtin_bit_cast(T, a);
}

// sum over the 64 lanes of a wave (butterfly: every lane ends with the total)
template <class C>
__device__ __forceinline__ Jac<C> wave_sum_jac(Jac<C> x) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        const Jac<C> o = wave_shfl(x, lane ^ d);
        x = jac_add(x, o);
    }
    return x;
}

// inclusive SUFFIX sums over the lanes of a wave: lane l ends with x_l + x_{l+1} + ... + x_63
template <class C>
__device__ __forceinline__ Jac<C> wave_suffix_jac(Jac<C> x) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
        const Jac<C> o = wave_shfl(x, (lane + d) & 63);
        if (lane + d < 64) x = jac_add(x, o);
    }
    return x;
}

// ---- points -----------------------------------------------------------------------------------------------
// wire -> affm; items n..2n-1 (GLV): (beta x, -y), the image under [z^2].  bad[0] |= 1 for an invalid point
// (replaced by infinity).
template <class C>
__global__ void __launch_bounds__(128) k_pip_points(PipShape s, const uint32_t* __restrict__ wire,
                                                    uint32_t* __restrict__ affm, uint32_t* __restrict__ bad) {
    constexpr int N = C::Fp::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t w[2 * N + 2];
#pragma unroll
    for (int t = 0; t < 2 * N + 2; t++) w[t] = wire[i * (2 * N + 2) + t];
    Aff<C> p;
    if (!aff_from_wire<C>(w, p)) {
        p = aff_inf<C>();
        atomicOr(bad, 1u);
    }
    aff_stg<C>(affm + i * 2 * N, p);
    if constexpr (pip_glv<C>()) {
        Aff<C> q = p;
        if (!p.is_inf()) {
            Fe<typename C::Fp> beta;
#pragma unroll
            for (int t = 0; t < C::Fp::NL; t++) beta.l[t] = C::K::BETA[t];
            q.x = fe_mul(beta, p.x);
            q.y = fe_neg(p.y);
        }
        aff_stg<C>(affm + ((size_t)s.n + i) * 2 * N, q);
    }
}

// ---- digits, histogram ------------------------------------------------------------------------------------
// keys / slots: [W][items]; counts: [nbuckets] (zeroed by the caller)
template <class C>
__global__ void __launch_bounds__(256) k_pip_digits(PipShape s, const uint32_t* __restrict__ scalars,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ slots,
                                                    uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t k[8];
    ld_words<8>(scalars + (size_t)i * 8, k);
    for (int it = 0; it < 16 && !words_lt_mod<typename C::Fr>(k); it++) {   // PrimeFieldElem values are < r
        uint32_t borrow = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const uint64_t d = (uint64_t)k[t] - C::Fr::MODW[t] - borrow;
            k[t] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
        }
    }
    uint32_t sub[2][8];
    if constexpr (pip_glv<C>()) {
        uint32_t k1[4], k2[4];
        glv_split<C>(k, k1, k2);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            sub[0][t] = t < 4 ? k1[t] : 0u;
            sub[1][t] = t < 4 ? k2[t] : 0u;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) sub[0][t] = k[t];
    }
    const uint32_t mask = (1u << s.c) - 1u;
    const int halves = pip_glv<C>() ? 2 : 1;
    for (int h = 0; h < halves; h++) {
        const uint32_t e = i + (uint32_t)h * s.n;
        uint32_t w[10];
        uint32_t carry = 0;
#pragma unroll
        for (int t = 0; t < 10; t++) {
            const uint64_t x = (uint64_t)(t < 8 ? sub[h][t] : 0u) + s.bias[t] + carry;
            w[t] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
        for (uint32_t j = 0; j < s.W; j++) {
            const int32_t dg = j + 1 < s.W ? (int32_t)(w[0] & mask) - (int32_t)s.half : (int32_t)w[0];
#pragma unroll
            for (int t = 0; t < 9; t++) w[t] = (w[t] >> s.c) | (w[t + 1] << (32 - s.c));
            w[9] >>= s.c;
            uint32_t key = PIP_EMPTY, slot = 0;
            if (dg != 0) {
                const uint32_t b = (dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg) - 1;
                slot = atomicAdd(&counts[(size_t)j * s.half + b], 1u);
                key = (b << 1) | (dg < 0 ? 1u : 0u);
            }
            keys[(size_t)j * s.items + e] = key;
            slots[(size_t)j * s.items + e] = slot;
        }
    }
}

// one block per window: offsets[j][b] = exclusive prefix sum of counts[j][.]
template <class C>
__global__ void __launch_bounds__(1024) k_pip_scan(PipShape s, const uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ offsets) {
    __shared__ uint32_t part[1024];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t nb = j + 1 < s.W ? s.half : s.top;
    const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(nb, t * per), hi = min(lo + per, nb);
    const uint32_t* cj = counts + (size_t)j * s.half;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += cj[b];
    part[t] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the per-thread sums
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    uint32_t* oj = offsets + (size_t)j * s.half;
    for (uint32_t b = lo; b < hi; b++) {
        oj[b] = run;
        run += cj[b];
    }
}

// sorted: [W][items] (only the first sum(counts[j]) entries of each row are written)
template <class C>
__global__ void __launch_bounds__(256) k_pip_scatter(PipShape s, const uint32_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ slots,
                                                     const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= s.items) return;
    const uint32_t key = keys[(size_t)j * s.items + i];
    if (key == PIP_EMPTY) return;
    const uint32_t b = key >> 1;
    sorted[(size_t)j * s.items + offsets[(size_t)j * s.half + b] + slots[(size_t)j * s.items + i]] = (i << 1) | (key & 1u);
}

// ---- bucket sums --------------------------------------------------------------------------------------------
// One lane per (window, bucket): XYZZ running sum of the bucket's points; the jacobian goes to buckets[flat].
// The points are gathered by LDS-DMA (glds16, kernels.hpp) into a two-deep per-lane ring, one addition ahead, exactly
// as k_fixed_msm gathers its table entries: all lanes of a wave step together up to the wave's longest bucket, a lane
// that has run out DMAs a dummy line, and a counted s_waitcnt vmcnt is all the synchronisation the ring needs.
// Buckets holding more than s.heavy points are not summed by one lane: they go to a list and are split over
// PIP_SPLIT blocks of 128 lanes each (k_pip_heavy), then folded back (k_pip_heavy_fold).
constexpr unsigned PIP_BLOCK = 128;
template <class C>
constexpr unsigned pip_ring_bytes() {
    return (PIP_BLOCK / 64) * (2 * (2 * C::Fp::N / 4)) * 1024;
}
template <class C>
__global__ void __launch_bounds__(PIP_BLOCK, fixed_waves<C>()) k_pip_buckets(PipShape s, const uint32_t* __restrict__ points,
                                                        const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ counts,
                                                        uint32_t* __restrict__ buckets,
                                                        uint32_t* __restrict__ heavy_list,
                                                        uint32_t* __restrict__ heavy_count) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    constexpr int CH = 2 * N / 4;                 // 16-byte pieces of a point
    constexpr int WAVE_WORDS = 2 * CH * 256;      // LDS words of one wave's ring
    extern __shared__ __align__(16) uint32_t lds[];
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = gid < (size_t)s.nbuckets;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* ring = lds + (threadIdx.x >> 6) * WAVE_WORDS;
    const uint32_t ring_addr = (uint32_t)reinterpret_cast<uintptr_t>(ring);
    uint32_t cnt = 0;
    const uint32_t* row = sorted;
    if (live) {
        const uint32_t j = min((uint32_t)(gid / s.half), s.W - 1);
        cnt = counts[gid];
        row = sorted + (size_t)j * s.items + offsets[gid];
        if (cnt > s.heavy) {
            heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)gid;
            cnt = 0;
        }
    }
    // the wave's longest bucket (wave-uniform loop bound)
    uint32_t maxc = cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxc = max(maxc, (uint32_t)__shfl_xor((int)maxc, d, 64));
    maxc = __builtin_amdgcn_readfirstlane(maxc);
    uint32_t nbits = 0;
    uint32_t e_next = cnt ? row[0] : 0u;   // the sorted entry of the NEXT step to issue: its load rides under an addition
    auto issue = [&](uint32_t t) {
        const uint32_t slot = t & 1u;
        const uint32_t* src = points;   // dummy line
        uint32_t neg = 0;
        if (t < cnt) {
            src = points + (size_t)(e_next >> 1) * 2 * N;
            neg = e_next & 1u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's previous point has been read out
#pragma unroll
        for (int k = 0; k < CH; k++) glds16(src + 4 * k, ring_addr + (slot * CH + k) * 1024);
        nbits = (nbits & ~(1u << slot)) | (neg << slot);
        e_next = row[t + 1 < cnt ? t + 1 : 0];   // every lane loads (a dummy when it has run out): one VMEM op per step
    };
    Xyzz<C> acc = xyzz_inf<C>();
    issue(0);
    issue(1);
    for (uint32_t t = 0; t < maxc; t++) {
        const uint32_t slot = t & 1u;
        // in flight, oldest first: step t's DMAs, step t+1's DMAs, the e_next load -- all but the last CH + 1 have landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH + 1) : "memory");
        uint32_t raw[2 * N];
        const uint4* q = reinterpret_cast<const uint4*>(ring + slot * CH * 256);
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const uint4 v = q[k * 64 + lane];
            raw[4 * k] = v.x;
            raw[4 * k + 1] = v.y;
            raw[4 * k + 2] = v.z;
            raw[4 * k + 3] = v.w;
        }
        const bool neg = (nbits >> slot) & 1u;
        const Aff<C> cur = aff_load<C>(raw);
        issue(t + 2);
        if (t < cnt) xyzz_madd_lazy(acc, cur, neg);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (live && !(counts[gid] > s.heavy)) jac_stg<C>(buckets + gid * JW, xyzz_to_jac(acc));
}

// grid (any, PIP_SPLIT): block (h, part) sums part `part` of heavy bucket heavy_list[h] with 128 lanes and an
// LDS tree; heavy_parts[h][part] receives the jacobian
template <class C>
__global__ void __launch_bounds__(128, 2) k_pip_heavy(PipShape s, const uint32_t* __restrict__ points,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ offsets,
                                                      const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ heavy_list,
                                                      const uint32_t* __restrict__ heavy_count,
                                                      uint32_t* __restrict__ heavy_parts) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t gid = heavy_list[h];
        const uint32_t j = min(gid / s.half, s.W - 1);
        const uint32_t beg = offsets[gid], cnt = counts[gid];
        const uint32_t per = (cnt + PIP_SPLIT - 1) / PIP_SPLIT;
        const uint32_t lo = min(cnt, blockIdx.y * per), hi = min(cnt, lo + per);
        const uint32_t* row = sorted + (size_t)j * s.items + beg;
        Xyzz<C> acc = xyzz_inf<C>();
        for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
            const uint32_t e = row[t];
            const Aff<C> q = aff_ldg<C>(points + (size_t)(e >> 1) * 2 * N);
            xyzz_madd_lazy(acc, q, (e & 1u) != 0);
        }
        Jac<C> sum = block_reduce_jac<C>(xyzz_to_jac(acc), lds);
        if (threadIdx.x == 0) jac_stg<C>(heavy_parts + ((size_t)h * PIP_SPLIT + blockIdx.y) * JW, sum);
        __syncthreads();
    }
}

// one block of PIP_SPLIT lanes per heavy bucket: bucket = sum of its PIP_SPLIT parts (LDS tree)
template <class C>
__global__ void __launch_bounds__(PIP_SPLIT) k_pip_heavy_fold(const uint32_t* __restrict__ heavy_list,
                                                              const uint32_t* __restrict__ heavy_count,
                                                              const uint32_t* __restrict__ heavy_parts,
                                                              uint32_t* __restrict__ buckets) {
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        Jac<C> acc = jac_ldg<C>(heavy_parts + ((size_t)h * PIP_SPLIT + threadIdx.x) * JW);
        acc = block_reduce_jac<C>(acc, lds);
        if (threadIdx.x == 0) jac_stg<C>(buckets + (size_t)heavy_list[h] * JW, acc);
        __syncthreads();
    }
}

// ---- bucket reduction: sum_b (b + 1) B_b per window -----------------------------------------------------------
// One WAVE per tile of TS = 64 S consecutive buckets of one window (tile t: window j = min(t / tiles_lo, W-1), tile u
// of that window, buckets [u TS, (u+1) TS) of its nb).  Lane l owns buckets b0 + [0, S): descending running sums
// give run_l = sum B_b and acc_l = sum (b - b0 + 1) B_b.  Across the wave
//   T = sum_l run_l ,  A = sum_l [acc_l + (l S) run_l] = sum_l acc_l + S * sum_{l >= 1} suffix_l(run)
// with the suffix sums and the final sums exchanged by wave shuffles.  tile_out[t] = (A, T).
template <class C>
__global__ void __launch_bounds__(64) k_pip_tiles(PipShape s, const uint32_t* __restrict__ buckets,
                                                  uint32_t* __restrict__ tile_out) {
    constexpr int JW = jac_words<C>();
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t j = min(t / s.tiles_lo, s.W - 1);
    const uint32_t u = t - j * s.tiles_lo;
    const uint32_t nb = j + 1 < s.W ? s.half : s.top;
    const uint32_t b0 = min(nb, u * s.TS + lane * s.S), b1 = min(nb, b0 + s.S);
    const uint32_t* bj = buckets + (size_t)j * s.half * JW;
    Jac<C> run = jac_inf<C>(), acc = jac_inf<C>();
    for (uint32_t b = b1; b-- > b0;) {
        run = jac_add(run, jac_ldg<C>(bj + (size_t)b * JW));
        acc = jac_add(acc, run);
    }
    const Jac<C> suf = wave_suffix_jac<C>(run);
    Jac<C> y = lane >= 1 ? suf : jac_inf<C>();
    for (uint32_t d = 1; d < s.S; d <<= 1) y = jac_dbl(y);
    const Jac<C> A = wave_sum_jac<C>(jac_add(acc, y));
    if (lane == 0) {
        jac_stg<C>(tile_out + (size_t)t * 2 * JW, A);
        jac_stg<C>(tile_out + ((size_t)t * 2 + 1) * JW, suf);
    }
}

// one wave per window: R_j = sum_u [A_u + (u TS) T_u]
template <class C>
__global__ void __launch_bounds__(64) k_pip_windows(PipShape s, const uint32_t* __restrict__ tile_out,
                                                    uint32_t* __restrict__ window_sums) {
    constexpr int JW = jac_words<C>();
    const uint32_t j = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t nt = j + 1 < s.W ? s.tiles_lo : s.tiles_top;
    const uint32_t* tj = tile_out + (size_t)j * s.tiles_lo * 2 * JW;
    Jac<C> asum = jac_inf<C>(), x = jac_inf<C>();
    for (uint32_t u = lane; u < nt; u += 64) {
        asum = jac_add(asum, jac_ldg<C>(tj + (size_t)u * 2 * JW));
        const Jac<C> T = jac_ldg<C>(tj + ((size_t)u * 2 + 1) * JW);
        Jac<C> m = jac_inf<C>();   // u * T
        for (int bit = 31 - __builtin_clz(u | 1u); bit >= 0; bit--) {
            m = jac_dbl(m);
            if ((u >> bit) & 1u) m = jac_add(m, T);
        }
        x = jac_add(x, m);
    }
    for (uint32_t d = 1; d < s.TS; d <<= 1) x = jac_dbl(x);
    const Jac<C> R = wave_sum_jac<C>(jac_add(asum, x));
    if (lane == 0) jac_stg<C>(window_sums + (size_t)j * JW, R);
}

// Horner over the windows; the result is ADDED to `extra` partials (may be none) and written as one jacobian to
// out_jac (may be null) and as the affine wire point to out_wire (may be null).  One wave, as a binary tree over the
// windows: lane j starts with R_j; at level l the lanes whose index is a multiple of 2^(l+1) take the partial of lane
// j + 2^l through LDS, double it c * 2^l times and add it.  The ~c (W - 1) doublings of the top window still form one
// chain, but the W additions of Horner's rule shrink to log2 W on the critical path (the same form as var_horner_wave
// in kernels.hpp).
template <class C>
__global__ void __launch_bounds__(64) k_pip_final(PipShape s, const uint32_t* __restrict__ window_sums,
                                                  const uint32_t* __restrict__ extra, uint32_t n_extra,
                                                  uint32_t* __restrict__ out_jac, uint32_t* __restrict__ out_wire) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    __shared__ __align__(16) uint32_t lds[64 * JW];
    if (blockIdx.x != 0) return;
    const uint32_t j = threadIdx.x & 63u;
    // G consecutive windows per lane (G = 1 unless an explicit narrow window gives W > 64), folded serially first
    const uint32_t G = (s.W + 63) / 64, L = (s.W + G - 1) / G;
    Jac<C> acc = jac_inf<C>();
    for (uint32_t g = G; g-- > 0;) {
        const uint32_t w = j * G + g;
        if (w >= s.W) continue;
        if (!acc.is_inf())
            for (uint32_t t = 0; t < s.c; t++) acc = jac_dbl(acc);
        acc = jac_add(acc, jac_ldg<C>(window_sums + (size_t)w * JW));
    }
    for (uint32_t stride = 1; stride < L; stride <<= 1) {
        jac_store(acc, lds + (size_t)j * JW);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((j & (2 * stride - 1)) == 0 && j + stride < L) {
            Jac<C> hi = jac_load<C>(lds + (size_t)(j + stride) * JW);
            if (!hi.is_inf())
                for (uint32_t t = 0; t < s.c * G * stride; t++) hi = jac_dbl(hi);
            acc = jac_add(acc, hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (j != 0) return;
    for (uint32_t t = 0; t < n_extra; t++) acc = jac_add(acc, jac_ldg<C>(extra + (size_t)t * JW));
    if (out_jac) jac_stg<C>(out_jac, acc);
    if (out_wire) {
        uint32_t w[2 * N + 2];
        aff_to_wire(jac_to_aff(acc), w);
#pragma unroll
        for (int t = 0; t < 2 * N + 2; t++) out_wire[t] = w[t];
    }
}

// the result of an empty MulVec: Point::zero()
template <class C>
__global__ void __launch_bounds__(64) k_pip_zero_point(uint32_t* __restrict__ out_wire) {
    constexpr int N = C::Fp::N;
    if (threadIdx.x < 2 * N + 2) out_wire[threadIdx.x] = threadIdx.x == 2 * N ? 1u : 0u;
}

struct PipWorkspace {
    size_t points, keys, slots, sorted, counts, offsets, buckets, tiles, wsums, hlist, hcount, hparts, bad, total;
    size_t max_heavy;
};
template <class C>
inline PipWorkspace pip_workspace(const PipShape& s) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    PipWorkspace w;
    size_t o = 0;
    w.points = o;
    o += al(((size_t)s.items + 1) * 2 * N * 4);
    w.keys = o;
    o += al((size_t)s.W * s.items * 4);
    w.slots = o;
    o += al((size_t)s.W * s.items * 4);
    w.sorted = o;
    o += al((size_t)s.W * s.items * 4 + 16);
    w.counts = o;
    o += al((size_t)s.nbuckets * 4);
    w.offsets = o;
    o += al((size_t)s.nbuckets * 4);
    w.buckets = o;
    o += al((size_t)s.nbuckets * JW * 4);
    w.tiles = o;
    o += al((size_t)s.ntiles * 2 * JW * 4);
    w.wsums = o;
    o += al((size_t)s.W * JW * 4);
    // a heavy bucket holds > s.heavy of the W * items sorted entries
    w.max_heavy = std::min<size_t>((size_t)s.nbuckets, (size_t)s.W * s.items / s.heavy + 1);
    w.hlist = o;
    o += al(w.max_heavy * 4);
    w.hcount = o;
    w.bad = o + 4;   // private status word, right behind the heavy-bucket counter
    o += al(8);
    w.hparts = o;
    o += al(w.max_heavy * PIP_SPLIT * JW * 4);
    w.total = o;
    return w;
}

// Enqueues the whole pipeline on `st`.  d_wire_points: n wire points; d_scalars: n canonical scalars (values >= r are
// reduced).  d_out_wire: one wire point; d_status (may be null): 0, or 1 when a point was not on the curve (it counts
// as infinity).  n >= 1.
template <class C>
inline hipError_t pip_launch(const PipShape& s, const uint32_t* d_scalars, const uint32_t* d_wire_points, uint8_t* d_ws,
                             uint32_t* d_out_wire, uint32_t* d_status, hipStream_t st) {
    constexpr int JW = jac_words<C>();
    const PipWorkspace w = pip_workspace<C>(s);
    uint32_t* points = reinterpret_cast<uint32_t*>(d_ws + w.points);
    uint32_t* keys = reinterpret_cast<uint32_t*>(d_ws + w.keys);
    uint32_t* slots = reinterpret_cast<uint32_t*>(d_ws + w.slots);
    uint32_t* sorted = reinterpret_cast<uint32_t*>(d_ws + w.sorted);
    uint32_t* counts = reinterpret_cast<uint32_t*>(d_ws + w.counts);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(d_ws + w.offsets);
    uint32_t* buckets = reinterpret_cast<uint32_t*>(d_ws + w.buckets);
    uint32_t* tiles = reinterpret_cast<uint32_t*>(d_ws + w.tiles);
    uint32_t* wsums = reinterpret_cast<uint32_t*>(d_ws + w.wsums);
    uint32_t* hlist = reinterpret_cast<uint32_t*>(d_ws + w.hlist);
    uint32_t* hcount = reinterpret_cast<uint32_t*>(d_ws + w.hcount);
    uint32_t* hparts = reinterpret_cast<uint32_t*>(d_ws + w.hparts);
    uint32_t* bad = d_status ? d_status : reinterpret_cast<uint32_t*>(d_ws + w.bad);
    hipError_t e = zero_words_async(counts, (size_t)s.nbuckets * 4, st);
    if (e != hipSuccess) return e;
    e = zero_words_async(hcount, 8, st);   // hcount and the private `bad` word behind it
    if (e != hipSuccess) return e;
    if (d_status) {
        e = zero_words_async(d_status, 4, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_pip_points<C>, dim3((s.n + 127) / 128), dim3(128), 0, st, s, d_wire_points, points, bad);
    hipLaunchKernelGGL(k_pip_digits<C>, dim3((s.n + 255) / 256), dim3(256), 0, st, s, d_scalars, keys, slots, counts);
    hipLaunchKernelGGL(k_pip_scan<C>, dim3(s.W), dim3(1024), 0, st, s, counts, offsets);
    hipLaunchKernelGGL(k_pip_scatter<C>, dim3((s.items + 255) / 256, s.W), dim3(256), 0, st, s, keys, slots, offsets, sorted);
    hipLaunchKernelGGL(k_pip_buckets<C>, dim3((unsigned)(((size_t)s.nbuckets + PIP_BLOCK - 1) / PIP_BLOCK)), dim3(PIP_BLOCK),
                       pip_ring_bytes<C>(), st, s, points, sorted, offsets, counts, buckets, hlist, hcount);
    // heavy buckets are few (none at all for uniformly distributed digits): a small grid that strides over
    // the list -- an oversized grid of immediately-exiting blocks costs milliseconds
    const unsigned hgrid = (unsigned)std::min<size_t>(w.max_heavy, 64);
    hipLaunchKernelGGL(k_pip_heavy<C>, dim3(hgrid, PIP_SPLIT), dim3(128), 128 * JW * 4, st, s, points, sorted,
                       offsets, counts, hlist, hcount, hparts);
    hipLaunchKernelGGL(k_pip_heavy_fold<C>, dim3(hgrid), dim3(PIP_SPLIT), PIP_SPLIT * JW * 4, st, hlist, hcount, hparts,
                       buckets);
    hipLaunchKernelGGL(k_pip_tiles<C>, dim3(s.ntiles), dim3(64), 0, st, s, buckets, tiles);
    hipLaunchKernelGGL(k_pip_windows<C>, dim3(s.W), dim3(64), 0, st, s, tiles, wsums);
    hipLaunchKernelGGL(k_pip_final<C>, dim3(1), dim3(64), 0, st, s, wsums, (const uint32_t*)nullptr, 0u,
                       (uint32_t*)nullptr, d_out_wire);
    return hipGetLastError();
}

}  // namespace bpp
