// pippenger.hpp -- bucket-method MulVec for LARGE variable-base inputs on gfx950, device resident.
//
// The reference's MulVec::calculate (src/bls12_381/building_block/mulvec.rs:20-33; secp256k1 twin
// src/secp256k1/building_block/secp256k1/util.rs:22-36) is one scalar multiplication per term.  For the sizes the
// reference itself produces (N <= 2 089) the data-parallel restatement in kernels.hpp (k_msm_naive_partial) already
// finishes in one scalar-mult latency; this file is the path for N in the thousands and up: bpp_msm_device /
// bpp_msm on big inputs.  Every buffer lives in HBM; nothing here synchronises with the host.
//
// Scalars.  On BLS12-381 and secp256k1 every scalar is first split with the curve's endomorphism (GLV, ec.hpp
// glv_split_signed: k = +-k1 +- k2 mu, both halves < 2^128, [mu] P = (beta x, -+y)): an input point becomes TWO items
// (P with k1, its image with k2) of 128-bit sub-scalars -- the same number of bucket additions, but half the windows, so
// half the buckets to reduce and half the ~255 sequential doublings of the tail.  edwards25519 has no endomorphism and
// runs one item of Fr::BITS bits per point.
// Windows of MIXED width: W = ceil(bits / c) windows share the bits as evenly as they divide -- the low `nwide` windows
// are q + 1 bits wide, the rest q (<= c) -- so that every window has about the same number of buckets and no window is
// left with a handful of bits (a 2-bit top window would put a quarter of all points into each of its buckets).  All
// windows but the top one hold SIGNED digits (digit of value + bias, minus half the range: 2^(w-1) buckets), the top
// window takes what is left of the value UNSIGNED (`top` buckets), so no window is spent on the carry of the recoding.
//
// The unit of work of the bucket sums is not a bucket but a CHUNK: L consecutive entries of a window's sorted item
// list, whatever buckets they belong to.  Every lane performs exactly L mixed additions -- no lane waits for a
// neighbour's longer bucket, the launch has as many equal units as it takes to fill the chip several times over, and a
// bucket holding half of all points (equal scalars) is simply many chunks.  A chunk emits one partial sum per bucket
// SEGMENT it touches; a bucket's sum is the sum of its segments (k_pip_fold).
//
// Pipeline (one stream, no host round trips):
//   k_pip_points   wire points -> Montgomery affine (+ on-curve check) and, with GLV, the endomorphism image; the
//                  point's scalar reduced mod r and split, once
//   k_pip_count / k_pip_cscan / k_pip_place / k_pip_binsort   per window a counting sort of the items by bucket, in
//                  two levels (coarse bins of 32..256 buckets, then the buckets of a bin) with every per-entry atomic in LDS;
//                  out: sorted[j] = items in bucket order, counts / offsets per bucket
//   k_pip_segments per window (one block, LDS scan): per bucket the number of chunks it touches and the prefix sums of
//                  that (segment bases); per chunk its first bucket
//   k_pip_chunks   per (window, chunk): XYZZ running sums over the chunk's entries, gathered through the LDS-DMA ring
//                  of k_fixed_msm (kernels.hpp), flushed as a jacobian at every bucket boundary and at the chunk's end
//   k_pip_fold     per bucket: sum of its segments (buckets spread over more than PIP_FOLD_MAX chunks: one wave each,
//                  k_pip_fold_heavy, strided sums + a shuffle butterfly)
//   k_pip_tiles    per tile of 64 S consecutive buckets, ONE WAVE: every lane runs the running sums over its S
//                  buckets (sum B_b and sum (b - b0 + 1) B_b), then the lanes' partials are combined inside the
//                  wave with shuffles: a suffix scan of the lane totals (ds_bpermute of the limbs, 6 steps) gives
//                  sum_l l * run_l, a butterfly reduces the rest.  Out: (A_t, T_t) per tile.
//   k_pip_windows  per window: R_j = sum_t A_t + (64 S) sum_t t T_t  (lane-local double-and-add by the tile number,
//                  wave butterfly)
//   k_pip_final    window j doubles R_j off(j) times, all windows at once, 3 / 4 lanes sharing every doubling (the
//                  critical path is the top window's off(W-1) doublings and log2 W additions of a butterfly), then the
//                  affine wire point
#pragma once
#include <algorithm>

#include "kernels.hpp"

namespace bpp {

template <class C>
constexpr bool pip_glv() {
    return curve_has_glv<C>();
}

struct PipShape {
    uint32_t n;                // input points
    uint32_t items;            // (point, sub-scalar) pairs: n, or 2 n with the endomorphism split
    uint32_t glv;
    uint32_t c, W;             // requested (maximum) window bits, windows
    uint32_t q, nwide;         // windows j < nwide are q + 1 bits wide, the others q
    uint32_t top;              // buckets of the top window (its digit is unsigned: 1..top)
    uint32_t nbuckets;         // all windows
    uint32_t nbmax;            // buckets of the largest window
    uint32_t S, TS;            // buckets per lane / per tile (64 S) in k_pip_tiles
    uint32_t tw, tn, ntiles;   // tiles of a wide / narrow signed window, all tiles
    uint32_t L, cpw, capseg;   // entries per chunk, chunks per window, segment slots per window
    uint32_t fb;               // log2 of the buckets per coarse bin of the sort (5..8)
    uint32_t fl;               // lanes per bucket in k_pip_fold (1, 2, 4 or 8)
    uint32_t istride;          // entries per row of `sorted`: items rounded up to 4 (rows start 16-byte aligned for the entry DMA)
    uint32_t bias[10];         // sum over the signed windows of half their range at their offset

    __host__ __device__ uint32_t width(uint32_t j) const { return q + (j < nwide ? 1u : 0u); }
    __host__ __device__ uint32_t off(uint32_t j) const { return j * q + (j < nwide ? j : nwide); }
    __host__ __device__ uint32_t nb(uint32_t j) const { return j + 1 == W ? top : 1u << (width(j) - 1); }
    __host__ __device__ uint32_t bbase(uint32_t j) const {
        return j < nwide ? j << q : (nwide << q) + ((j - nwide) << (q - 1));
    }
    __host__ __device__ uint32_t tiles(uint32_t j) const { return (nb(j) + TS - 1) / TS; }
    __host__ __device__ uint32_t tbase(uint32_t j) const { return j < nwide ? j * tw : nwide * tw + (j - nwide) * tn; }
    __host__ __device__ uint32_t window_of_tile(uint32_t t) const {
        if (t < nwide * tw) return t / tw;
        const uint32_t r = (t - nwide * tw) / tn;
        return nwide + (r < W - 1 - nwide ? r : W - 1 - nwide);
    }
};

// max_words: the largest sub-scalar value (8 words); bits: its bit length
inline int pip_shape(size_t n, int c, bool glv, const uint32_t* max_words, int bits, PipShape& s) {
    if (c < 2 || c > 16) return fail(BPP_E_ARG, "window_bits must be in [2, 16]");
    if (n >= ((size_t)1 << 28)) return fail(BPP_E_ARG, "MulVec too long for one call");   // a sorted entry is item << 1 | sign, bit 31 a flag
    s.n = (uint32_t)n;
    s.glv = glv ? 1u : 0u;
    s.items = (uint32_t)(glv ? 2 * n : n);
    s.istride = (s.items + 3u) & ~3u;
    s.c = (uint32_t)c;
    s.W = (uint32_t)((bits + c - 1) / c);
    s.q = (uint32_t)bits / s.W;
    s.nwide = (uint32_t)bits - s.q * s.W;
    for (int t = 0; t < 10; t++) s.bias[t] = 0;
    for (uint32_t j = 0; j + 1 < s.W; j++) {
        const uint32_t bit = s.off(j) + s.width(j) - 1;
        s.bias[bit >> 5] |= 1u << (bit & 31);
    }
    uint32_t v[10];
    uint32_t carry = 0;
    for (int t = 0; t < 10; t++) {
        uint64_t x = (uint64_t)(t < 8 ? max_words[t] : 0u) + s.bias[t] + carry;
        v[t] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    const uint32_t sh = s.off(s.W - 1);
    uint64_t top = 0;
    for (int t = 9; t >= 0; t--) {
        const int lo = 32 * t - (int)sh;
        if (lo >= 32 && v[t]) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
        if (lo > -32 && lo < 32) top |= lo >= 0 ? (uint64_t)v[t] << lo : (uint64_t)(v[t] >> (-lo));
    }
    if (top == 0 || top > ((uint64_t)1 << 17)) return fail(BPP_E_ARG, "window_bits too small for this scalar field");
    s.top = (uint32_t)top;
    s.nbuckets = s.bbase(s.W - 1) + s.top;
    s.nbmax = s.top;
    for (uint32_t j = 0; j + 1 < s.W; j++) s.nbmax = std::max(s.nbmax, s.nb(j));
    // tiles: enough of them to spread a window over the chip, few enough that k_pip_windows stays short
    uint32_t S = (1u << s.q) / 8192;
    S = S < 1 ? 1 : (S > 8 ? 8 : S);
    s.S = S;
    s.TS = 64 * S;
    s.tw = ((1u << s.q) + s.TS - 1) / s.TS;
    s.tn = ((1u << (s.q - (s.q ? 1 : 0))) + s.TS - 1) / s.TS;
    s.ntiles = s.tbase(s.W - 1) + s.tiles(s.W - 1);
    // chunks: 64 entries each for large inputs; shorter ones while the launch would not fill the chip (2^17 lanes)
    const size_t entries = (size_t)s.W * s.items;
    uint32_t L = 64;
    while (L > 8 && entries / L < ((size_t)1 << 18)) L >>= 1;
    s.L = L;
    // coarse bins of the sort: about 1024 of them (one block of k_pip_binsort each), 32..256 buckets wide
    uint32_t fb = 8;
    while (fb > 5 && (s.nbuckets >> fb) < 1024) fb--;
    s.fb = fb;
    s.cpw = (s.items + L - 1) / L;
    s.capseg = s.cpw + s.nbmax;
    // k_pip_fold: a bucket touches about (its entries / L) + 1 chunks; while the launch stays within one residency of the
    // chip, several lanes share a bucket's segments (strided sums, then a butterfly inside the lane group)
    const size_t nseg = entries / ((size_t)s.nbuckets * L) + 1;
    uint32_t fl = 1;
    while (fl < 8 && 2 * fl <= nseg && (size_t)s.nbuckets * 2 * fl <= ((size_t)1 << 18)) fl <<= 1;
    s.fl = fl;
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
// reduction of the buckets stays a few per cent).  Small inputs are latency bound -- the chain is a chunk, the
// fold / tile / window reduction, then the doublings of the top window -- and want MORE, shorter buckets.
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
constexpr uint32_t PIP_FOLD_MAX = 32;   // a bucket spread over more chunks than this is folded by a whole wave

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
// wire -> affm; items n..2n-1 (GLV): the image under the endomorphism, (beta x, -y) / (beta x, y).  bad[0] |= 1 for an invalid point
// (replaced by infinity).
// ... and the point's scalar, reduced mod r and split ONCE (the secp256k1 split is two 256 x 256-bit products and three
// products mod n: recomputing it per window in the sort cost more than the sort): split[i] = k1 (4 words) | k2 (4 words) |
// sign bits (bit 0: k1 negative, bit 1: k2 negative) | 3 spare words; without an endomorphism: the reduced scalar (8 words).
constexpr uint32_t PIP_SPLIT_WORDS = 12;
template <class C>
__global__ void __launch_bounds__(128) k_pip_points(PipShape s, const uint32_t* __restrict__ wire,
                                                    const uint32_t* __restrict__ scalars, uint32_t* __restrict__ affm,
                                                    uint32_t* __restrict__ split, uint32_t* __restrict__ bad) {
    constexpr int N = C::Fp::N;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    {
        uint32_t k[8];
        ld_words<8>(scalars + i * 8, k);
        for (int it = 0; it < 16 && !words_lt_mod<typename C::Fr>(k); it++) {   // PrimeFieldElem values are < r
            uint32_t borrow = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint64_t d = (uint64_t)k[t] - C::Fr::MODW[t] - borrow;
                k[t] = (uint32_t)d;
                borrow = (uint32_t)(d >> 63);
            }
        }
        uint32_t o[PIP_SPLIT_WORDS];
#pragma unroll
        for (int t = 0; t < (int)PIP_SPLIT_WORDS; t++) o[t] = t < 8 ? k[t] : 0u;
        if constexpr (pip_glv<C>()) {
            uint32_t k1[4], k2[4];
            bool n1, n2;
            glv_split_signed<C>(k, k1, k2, n1, n2);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                o[t] = k1[t];
                o[4 + t] = k2[t];
            }
            o[8] = (n1 ? 1u : 0u) | (n2 ? 2u : 0u);
        }
        st_words<PIP_SPLIT_WORDS>(split + i * PIP_SPLIT_WORDS, o);
    }
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
            if (glv_image_negates_y<C>()) q.y = fe_neg(p.y);
        }
        aff_stg<C>(affm + ((size_t)s.n + i) * 2 * N, q);
    }
}

// ---- digits and the sort by bucket -----------------------------------------------------------------------------
// A counting sort per window in two levels, with every per-entry atomic in LDS.  (Global atomics execute at the memory
// side on this chip: one returning atomic per (item, window) -- 67 M of them at N = 2^22 -- ran at the chip-wide
// atomic rate and cost a third of the whole MulVec.)
//   level 1, coarse bins of 2^fb consecutive buckets (fb = 5..8, chosen so that there are about a thousand bins):
//     k_pip_count   per (window, block of PIP_CHUNK points): digit -> coarse bin, LDS histogram, ONE global add per
//                   (block, coarse bin)
//     k_pip_cscan   offsets of the coarse bins inside their window's row; wtotal
//     k_pip_place   the same walk again: LDS rank inside the block, ONE global add per (block, coarse bin) reserves the
//                   block's run inside the bin, records (item | sign, bucket mod 512) go to their bin
//   level 2, k_pip_binsort: one block per (window, coarse bin): LDS histogram over the bin's <= 256 buckets, LDS scan,
//                   counts[] and offsets[] of those buckets, then the records are placed in bucket order in `sorted`
//   k_pip_segments  per window (one block, LDS scan): per bucket the number of chunks it touches and the prefix sums of
//                   that (segment bases); per chunk its first bucket
constexpr uint32_t PIP_FINE_MAX = 256;      // buckets per coarse bin: 2^fb, fb in [5, 8] (PipShape::fb)
constexpr uint32_t PIP_CHUNK = 2048;        // points per block of k_pip_count / k_pip_place (256 threads x 8)
constexpr uint32_t PIP_IPT = PIP_CHUNK / 256;
constexpr uint32_t PIP_MAXCOARSE = 2052;    // coarse bins of the largest window: (2^16 + 1) / 32 + 1 = 2049

__host__ __device__ inline uint32_t pip_ncoarse(const PipShape& s, uint32_t j) { return (s.nb(j) + (1u << s.fb) - 1) >> s.fb; }
// coarse bins are numbered window by window; cbase(j) = bins of the windows below j (W + 1 entries fit a kernel argument
// badly for W up to 128, so it is recomputed: windows come in at most three sizes)
__host__ __device__ inline uint32_t pip_cbase(const PipShape& s, uint32_t j) {
    const uint32_t fine = 1u << s.fb;
    const uint32_t cw = ((1u << s.q) + fine - 1) >> s.fb, cn = ((1u << (s.q - 1)) + fine - 1) >> s.fb;
    return j < s.nwide ? j * cw : s.nwide * cw + (j - s.nwide) * cn;
}
__host__ __device__ inline uint32_t pip_ncoarse_total(const PipShape& s) { return pip_cbase(s, s.W - 1) + pip_ncoarse(s, s.W - 1); }

// the sub-scalars of point i as k_pip_points left them: w[h] = value + bias (10 words), neg[h]
template <class C>
__device__ __forceinline__ void pip_subscalars(const PipShape& s, const uint32_t* __restrict__ split, uint32_t i,
                                               uint32_t w[2][10], bool neg[2]) {
    uint32_t k[PIP_SPLIT_WORDS];
    ld_words<PIP_SPLIT_WORDS>(split + (size_t)i * PIP_SPLIT_WORDS, k);
    neg[0] = pip_glv<C>() && (k[8] & 1u);
    neg[1] = pip_glv<C>() && (k[8] & 2u);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t carry = 0;
#pragma unroll
        for (int t = 0; t < 10; t++) {
            uint32_t v;
            if constexpr (pip_glv<C>())
                v = t < 4 ? k[4 * h + t] : 0u;
            else
                v = (h == 0 && t < 8) ? k[t] : 0u;
            const uint64_t x = (uint64_t)v + s.bias[t] + carry;
            w[h][t] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
    }
}
// digit of window j of (value + bias): signed below the top window, unsigned in it
__device__ __forceinline__ int32_t pip_digit(const PipShape& s, const uint32_t w[10], uint32_t j) {
    const uint32_t o = s.off(j), wi = o >> 5, sh = o & 31u;
    uint32_t v = w[wi] >> sh;
    if (sh && wi + 1 < 10) v |= w[wi + 1] << (32 - sh);
    if (j + 1 == s.W) return (int32_t)v;   // everything that is left of the value (< 2^18)
    const uint32_t wd = s.width(j);
    return (int32_t)(v & ((1u << wd) - 1u)) - (int32_t)(1u << (wd - 1));
}

// grid (blocks of PIP_CHUNK points, W).  ccount: [coarse bins of all windows], zeroed by the caller.
template <class C>
__global__ void __launch_bounds__(256) k_pip_count(PipShape s, const uint32_t* __restrict__ split,
                                                   uint32_t* __restrict__ ccount) {
    __shared__ uint32_t hist[PIP_MAXCOARSE];
    const uint32_t j = blockIdx.y, t = threadIdx.x;
    const uint32_t nc = pip_ncoarse(s, j);
    for (uint32_t c = t; c < nc; c += blockDim.x) hist[c] = 0;
    __syncthreads();
    constexpr int halves = pip_glv<C>() ? 2 : 1;
    for (uint32_t q = 0; q < PIP_IPT; q++) {
        const uint32_t i = blockIdx.x * PIP_CHUNK + q * 256 + t;
        if (i >= s.n) break;
        uint32_t w[2][10];
        bool neg[2];
        pip_subscalars<C>(s, split, i, w, neg);
#pragma unroll
        for (int h = 0; h < halves; h++) {
            const int32_t dg = pip_digit(s, w[h], j);
            if (dg != 0) atomicAdd(&hist[((dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg) - 1) >> s.fb], 1u);
        }
    }
    __syncthreads();
    uint32_t* cj = ccount + pip_cbase(s, j);
    for (uint32_t c = t; c < nc; c += blockDim.x)
        if (hist[c]) atomicAdd(&cj[c], hist[c]);
}

// one block: cstart[bin] = exclusive prefix of ccount inside the bin's window; wtotal[j]; ccursor zeroed
static __global__ void __launch_bounds__(256) k_pip_cscan(PipShape s, const uint32_t* __restrict__ ccount,
                                                          uint32_t* __restrict__ cstart, uint32_t* __restrict__ ccursor,
                                                          uint32_t* __restrict__ wtotal) {
    for (uint32_t j = threadIdx.x; j < s.W; j += blockDim.x) {
        const uint32_t cb = pip_cbase(s, j), nc = pip_ncoarse(s, j);
        uint32_t run = 0;
        for (uint32_t c = 0; c < nc; c++) {
            cstart[cb + c] = run;
            ccursor[cb + c] = 0;
            run += ccount[cb + c];
        }
        wtotal[j] = run;
    }
}

// the same walk as k_pip_count: rec_item / rec_low: [W][items], grouped by coarse bin
template <class C>
__global__ void __launch_bounds__(256) k_pip_place(PipShape s, const uint32_t* __restrict__ split,
                                                   const uint32_t* __restrict__ cstart, uint32_t* __restrict__ ccursor,
                                                   uint32_t* __restrict__ rec_item, uint16_t* __restrict__ rec_low) {
    __shared__ uint32_t hist[PIP_MAXCOARSE];
    const uint32_t j = blockIdx.y, t = threadIdx.x;
    const uint32_t nc = pip_ncoarse(s, j), cb = pip_cbase(s, j);
    for (uint32_t c = t; c < nc; c += blockDim.x) hist[c] = 0;
    __syncthreads();
    constexpr int halves = pip_glv<C>() ? 2 : 1;
    // per entry of this thread: (coarse bin, rank inside the block's share of it) ; item | sign ; bucket mod 512
    uint32_t where[PIP_IPT * halves], what[PIP_IPT * halves], lowb[PIP_IPT * halves];
#pragma unroll
    for (uint32_t q = 0; q < PIP_IPT; q++) {
        const uint32_t i = blockIdx.x * PIP_CHUNK + q * 256 + t;
        uint32_t w[2][10];
        bool neg[2] = {false, false};
        if (i < s.n) pip_subscalars<C>(s, split, i, w, neg);
#pragma unroll
        for (int h = 0; h < halves; h++) {
            where[q * halves + h] = 0xffffffffu;
            what[q * halves + h] = 0;
            lowb[q * halves + h] = 0;
            if (i < s.n) {
                const int32_t dg = pip_digit(s, w[h], j);
                if (dg != 0) {
                    const uint32_t b = (dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg) - 1;
                    const uint32_t c = b >> s.fb;
                    const uint32_t rank = atomicAdd(&hist[c], 1u);      // < 2 PIP_CHUNK = 2^12
                    where[q * halves + h] = (c << 16) | rank;
                    what[q * halves + h] = ((i + (uint32_t)h * s.n) << 1) | (((dg < 0) != neg[h]) ? 1u : 0u);
                    lowb[q * halves + h] = b & ((1u << s.fb) - 1u);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t c = t; c < nc; c += blockDim.x) hist[c] = hist[c] ? atomicAdd(&ccursor[cb + c], hist[c]) : 0u;   // the block's base in bin c
    __syncthreads();
    uint32_t* ri = rec_item + (size_t)j * s.items;
    uint16_t* rl = rec_low + (size_t)j * s.items;
#pragma unroll
    for (uint32_t q = 0; q < PIP_IPT * halves; q++) {
        const uint32_t wh = where[q];
        if (wh == 0xffffffffu) continue;
        const uint32_t c = wh >> 16, rank = wh & 0xffffu;
        const uint32_t pos = cstart[cb + c] + hist[c] + rank;
        ri[pos] = what[q];
        rl[pos] = (uint16_t)lowb[q];
    }
}

// one block per (window, coarse bin): the bin's records -> bucket order.  counts / offsets: flat per bucket.
// A sorted entry is  item << 1 | sign  with PIP_LAST set on the last entry of every bucket: k_pip_chunks then knows where a
// bucket ends without looking anything up.
constexpr uint32_t PIP_LAST = 0x80000000u;
template <class C>
__global__ void __launch_bounds__(PIP_FINE_MAX) k_pip_binsort(PipShape s, const uint32_t* __restrict__ ccount,
                                                          const uint32_t* __restrict__ cstart,
                                                          const uint32_t* __restrict__ rec_item,
                                                          const uint16_t* __restrict__ rec_low, uint32_t* __restrict__ counts,
                                                          uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[PIP_FINE_MAX], pre[PIP_FINE_MAX];
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    // window of the bin
    const uint32_t fine = 1u << s.fb;
    const uint32_t cw = ((1u << s.q) + fine - 1) >> s.fb, cn = ((1u << (s.q - 1)) + fine - 1) >> s.fb;
    uint32_t j;
    if (bin < s.nwide * cw) {
        j = bin / cw;
    } else {
        const uint32_t r = (bin - s.nwide * cw) / cn;
        j = s.nwide + (r < s.W - 1 - s.nwide ? r : s.W - 1 - s.nwide);
    }
    const uint32_t c = bin - pip_cbase(s, j);
    const uint32_t nb = s.nb(j), b0 = c << s.fb;
    const uint32_t nfine = min(fine, nb - b0);
    const uint32_t lo = cstart[bin], n = ccount[bin];
    const uint32_t* ri = rec_item + (size_t)j * s.items + lo;
    const uint16_t* rl = rec_low + (size_t)j * s.items + lo;
    cnt[t] = 0;
    __syncthreads();
    // four records per thread per step: the loads of a step are in flight together
    for (uint32_t r = t; r < n; r += 4 * blockDim.x) {
        uint32_t lw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) lw[u] = r + u * blockDim.x < n ? rl[r + u * blockDim.x] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (lw[u] != 0xffffffffu) atomicAdd(&cnt[lw[u]], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt (Hillis-Steele)
    const uint32_t mine = cnt[t];
    pre[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < PIP_FINE_MAX; d <<= 1) {
        const uint32_t v = t >= d ? pre[t - d] : 0;
        __syncthreads();
        pre[t] += v;
        __syncthreads();
    }
    const uint32_t ex = pre[t] - mine;
    if (t < nfine) {
        counts[s.bbase(j) + b0 + t] = mine;
        offsets[s.bbase(j) + b0 + t] = lo + ex;
    }
    __syncthreads();
    cnt[t] = ex;   // cursors
    __syncthreads();
    uint32_t* so = sorted + (size_t)j * s.istride + lo;
    for (uint32_t r = t; r < n; r += 4 * blockDim.x) {
        uint32_t lw[4], it[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = r + u * blockDim.x < n;
            lw[u] = in ? rl[r + u * blockDim.x] : 0xffffffffu;
            it[u] = in ? ri[r + u * blockDim.x] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (lw[u] != 0xffffffffu) {   // bit 31: the last entry of its bucket (pre[] = one past it)
                const uint32_t pos = atomicAdd(&cnt[lw[u]], 1u);
                so[pos] = it[u] | (pos + 1 == pre[lw[u]] ? PIP_LAST : 0u);
            }
    }
}

// One block per window, LDS scan over the window's buckets (counts / offsets come from k_pip_binsort):
//   segbase[b]  = exclusive prefix sum of the number of chunks each bucket touches (its segments)
//   chunk_first[j][k] = the bucket that entry k L belongs to
template <class C>
__global__ void __launch_bounds__(1024) k_pip_segments(PipShape s, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ offsets, uint32_t* __restrict__ segbase,
                                                       uint32_t* __restrict__ chunk_first) {
    __shared__ uint32_t part[1024];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t nb = s.nb(j), bb = s.bbase(j);
    const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(nb, t * per), hi = min(lo + per, nb);
    uint32_t nseg = 0;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t cnt = counts[bb + b], o = offsets[bb + b];
        if (cnt) {
            nseg += (o + cnt - 1) / s.L - o / s.L + 1;
            for (uint32_t k = (o + s.L - 1) / s.L; k * s.L < o + cnt; k++) chunk_first[(size_t)j * s.cpw + k] = b;
        }
    }
    part[t] = nseg;
    __syncthreads();
    for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t srun = part[t] - nseg;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t cnt = counts[bb + b], o = offsets[bb + b];
        segbase[bb + b] = srun;
        if (cnt) srun += (o + cnt - 1) / s.L - o / s.L + 1;
    }
}

// ---- bucket sums, chunk by chunk ------------------------------------------------------------------------------
// One lane per (window, chunk of L sorted entries): an XYZZ running sum over the chunk, written into the next segment slot
// whenever a bucket ends (PIP_LAST of the entry) and at the chunk's end.  The slots of a window are numbered in the order the
// segments appear -- bucket by bucket, and inside a bucket chunk by chunk: slot(b, k) = segbase[b] + (k - first chunk of b) --
// so after a bucket's last entry the lane's next segment is simply the next slot, and nothing is looked up inside the loop.
// A segment is the RAW accumulator (four coordinates, unpacked 30-bit limbs, not reduced): converting it to the jacobian
// form costs two products, and inside this loop a whole wave pays them whenever ONE of its lanes meets a bucket boundary
// (a quarter to two thirds of the steps); k_pip_fold pays them once per segment, every lane busy.
// The points are gathered by LDS-DMA (glds16, kernels.hpp) into a two-deep per-lane ring, one addition ahead, exactly as
// k_fixed_msm gathers its table entries: all lanes of a wave step together, a lane past its last entry DMAs a dummy line,
// and a counted s_waitcnt vmcnt is all the synchronisation the ring needs (a flush's stores in flight only make the count
// more conservative).  The sorted entries themselves travel the same way, four per lane per DMA, one batch ahead.
constexpr unsigned PIP_BLOCK = 128;
template <class C>
constexpr unsigned pip_ring_bytes() {
    return (PIP_BLOCK / 64) * (2 * (2 * C::Fp::N / 4) + 2) * 1024;   // two point slots + two slots of four sorted entries per lane
}
// words of a segment: the accumulator as it is in registers
template <class C>
constexpr int seg_words() {
    return (int)(sizeof(Xyzz<C>) / 4);
}
template <class C>
__device__ __forceinline__ void seg_stg(uint32_t* __restrict__ p, const Xyzz<C>& a) {
    struct Raw {
        uint32_t w[seg_words<C>()];
    };
    const Raw r = __builtin_bit_cast(Raw, a);
    st_words<seg_words<C>()>(p, r.w);
}
template <class C>
__device__ __forceinline__ Jac<C> seg_ldg(const uint32_t* __restrict__ p) {
    struct Raw {
        uint32_t w[seg_words<C>()];
    };
    Raw r;
    ld_words<seg_words<C>()>(p, r.w);
    return xyzz_to_jac(__builtin_bit_cast(Xyzz<C>, r));
}
template <class C>
__global__ void __launch_bounds__(PIP_BLOCK, fixed_waves<C>()) k_pip_chunks(PipShape s, const uint32_t* __restrict__ points,
                                                        const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ segbase,
                                                        const uint32_t* __restrict__ chunk_first,
                                                        const uint32_t* __restrict__ wtotal,
                                                        uint32_t* __restrict__ segs) {
    constexpr int N = C::Fp::N;
    constexpr int SW = seg_words<C>();
    static_assert(SW % 4 == 0, "segments are written in 16-byte pieces");
    constexpr int CH = 2 * N / 4;                 // 16-byte pieces of a point
    constexpr int WAVE_WORDS = (2 * CH + 2) * 256;   // LDS words of one wave: the point ring + two batches of sorted entries
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t bpw = (s.cpw + PIP_BLOCK - 1) / PIP_BLOCK;   // blocks never straddle windows
    const uint32_t j = blockIdx.x / bpw;
    const uint32_t k0 = (blockIdx.x - j * bpw) * PIP_BLOCK;
    const uint32_t total = wtotal[j];
    if ((uint64_t)k0 * s.L >= total) return;      // block-uniform: nothing left of this window
    const uint32_t k = k0 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* ring = lds + (threadIdx.x >> 6) * WAVE_WORDS;
    uint32_t* ebuf = ring + 2 * CH * 256;                        // [2][lane][4 entries]
    const uint32_t ring_addr = (uint32_t)reinterpret_cast<uintptr_t>(ring);
    const uint32_t ebuf_addr = ring_addr + 2 * CH * 1024;
    const uint32_t* row = sorted + (size_t)j * s.istride;
    const uint32_t pos0 = (uint64_t)k * s.L < total ? k * s.L : total;
    const uint32_t stop = min(pos0 + s.L, total);           // pos0 >= stop: a lane without entries
    uint32_t* seg = segs;
    if (pos0 < stop) {
        const uint32_t b = s.bbase(j) + chunk_first[(size_t)j * s.cpw + k];
        seg = segs + ((size_t)j * s.capseg + segbase[b] + (k - offsets[b] / s.L)) * SW;
    }
    // batch q = this lane's entries 4 q .. 4 q + 3 (16 bytes, 16-byte aligned: pos0 is a multiple of L >= 8 and rows start
    // aligned) -> entry slot q & 1.  A batch is requested four steps before its first entry is needed.
    auto dma_entries = [&](uint32_t q) {
        const uint32_t p = pos0 + 4 * q;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's previous batch has been read out
        glds16(row + (p < stop ? p : 0), ebuf_addr + (q & 1u) * 1024);
    };
    uint32_t nbits = 0, lbits = 0;   // sign / last-of-bucket flag of the entries in the two ring slots
    auto issue = [&](uint32_t t) {
        const uint32_t slot = t & 1u;
        const uint32_t p = pos0 + t;
        if ((t & 3u) == 0) dma_entries((t >> 2) + 1);
        const uint32_t e = ebuf[((t >> 2) & 1u) * 256 + lane * 4 + (t & 3u)];
        const uint32_t* src = points;   // dummy line
        uint32_t neg = 0, last = 0;
        if (p < stop) {
            src = points + (size_t)((e & ~PIP_LAST) >> 1) * 2 * N;
            neg = e & 1u;
            last = e >> 31;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's previous point has been read out
#pragma unroll
        for (int kk = 0; kk < CH; kk++) glds16(src + 4 * kk, ring_addr + (slot * CH + kk) * 1024);
        nbits = (nbits & ~(1u << slot)) | (neg << slot);
        lbits = (lbits & ~(1u << slot)) | (last << slot);
    };
    dma_entries(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Xyzz<C> acc = xyzz_inf<C>();
    issue(0);
    issue(1);
    for (uint32_t t = 0; t < s.L; t++) {
        const uint32_t slot = t & 1u;
        // everything but the newest CH VMEM operations has completed: step t's DMAs were issued before step t+1's
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
        uint32_t raw[2 * N];
        const uint4* q = reinterpret_cast<const uint4*>(ring + slot * CH * 256);
#pragma unroll
        for (int kk = 0; kk < CH; kk++) {
            const uint4 v = q[kk * 64 + lane];
            raw[4 * kk] = v.x;
            raw[4 * kk + 1] = v.y;
            raw[4 * kk + 2] = v.z;
            raw[4 * kk + 3] = v.w;
        }
        const bool neg = (nbits >> slot) & 1u;
        const bool last = (lbits >> slot) & 1u;
        const Aff<C> cur = aff_load<C>(raw);
        issue(t + 2);
        const uint32_t p = pos0 + t;
        if (p < stop) {
            xyzz_madd_lazy(acc, cur, neg);
            if (last || p + 1 == stop) {   // the bucket's last entry, or the chunk's: on to the next slot
                seg_stg<C>(seg, acc);
                seg += SW;
                acc = xyzz_inf<C>();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// s.fl lanes per bucket (consecutive lanes of one wave): buckets[flat] = sum of the bucket's segments (infinity for an
// empty bucket).  Lane `sub` of the group adds segments sub, sub + fl, ..; a butterfly inside the group adds the lanes.
template <class C>
__global__ void __launch_bounds__(128) k_pip_fold(PipShape s, const uint32_t* __restrict__ offsets,
                                                  const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ segbase, const uint32_t* __restrict__ segs,
                                                  uint32_t* __restrict__ buckets, uint32_t* __restrict__ heavy_list,
                                                  uint32_t* __restrict__ heavy_count) {
    constexpr int JW = jac_words<C>();
    const uint32_t th = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t gid = th / s.fl, sub = th % s.fl;
    const bool live = gid < s.nbuckets;
    const uint32_t cnt = live ? counts[gid] : 0u;
    Jac<C> acc = jac_inf<C>();
    bool heavy = false;
    if (cnt) {
        const uint32_t o = offsets[gid];
        const uint32_t nseg = (o + cnt - 1) / s.L - o / s.L + 1;
        if (nseg > PIP_FOLD_MAX) {
            heavy = true;
            if (sub == 0) heavy_list[atomicAdd(heavy_count, 1u)] = gid;
        } else {
            // window of the bucket: the wide windows come first
            const uint32_t wide = s.nwide << s.q;
            const uint32_t j = gid < wide ? gid >> s.q : min(s.nwide + ((gid - wide) >> (s.q - 1)), s.W - 1);
            const uint32_t* sp = segs + ((size_t)j * s.capseg + segbase[gid]) * seg_words<C>();
            for (uint32_t t = sub; t < nseg; t += s.fl) acc = jac_add(acc, seg_ldg<C>(sp + (size_t)t * seg_words<C>()));
        }
    }
    if (s.fl > 1) acc = wave_sum_jac<C>(acc, (int)s.fl);   // every lane of the wave takes part
    if (live && sub == 0 && !heavy) jac_stg<C>(buckets + (size_t)gid * JW, acc);
}

// one wave per bucket of the heavy list (grid-stride): lanes sum every 64th segment, a butterfly adds the lanes
template <class C>
__global__ void __launch_bounds__(64) k_pip_fold_heavy(PipShape s, const uint32_t* __restrict__ offsets,
                                                       const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ segbase,
                                                       const uint32_t* __restrict__ segs, uint32_t* __restrict__ buckets,
                                                       const uint32_t* __restrict__ heavy_list,
                                                       const uint32_t* __restrict__ heavy_count) {
    constexpr int JW = jac_words<C>();
    const uint32_t nheavy = *heavy_count, lane = threadIdx.x & 63u;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        const uint32_t gid = heavy_list[h];
        const uint32_t o = offsets[gid], cnt = counts[gid];
        const uint32_t nseg = (o + cnt - 1) / s.L - o / s.L + 1;
        const uint32_t wide = s.nwide << s.q;
        const uint32_t j = gid < wide ? gid >> s.q : min(s.nwide + ((gid - wide) >> (s.q - 1)), s.W - 1);
        const uint32_t* sp = segs + ((size_t)j * s.capseg + segbase[gid]) * seg_words<C>();
        Jac<C> acc = jac_inf<C>();
        for (uint32_t t = lane; t < nseg; t += 64) acc = jac_add(acc, seg_ldg<C>(sp + (size_t)t * seg_words<C>()));
        acc = wave_sum_jac<C>(acc);
        if (lane == 0) jac_stg<C>(buckets + (size_t)gid * JW, acc);
    }
}

// ---- bucket reduction: sum_b (b + 1) B_b per window -----------------------------------------------------------
// One WAVE per tile of TS = 64 S consecutive buckets of one window (tile t: window j, tile u of that window, buckets
// [u TS, (u+1) TS) of its nb).  Lane l owns buckets b0 + [0, S): descending running sums give run_l = sum B_b and
// acc_l = sum (b - b0 + 1) B_b.  Across the wave
//   T = sum_l run_l ,  A = sum_l [acc_l + (l S) run_l] = sum_l acc_l + S * sum_{l >= 1} suffix_l(run)
// with the suffix sums and the final sums exchanged by wave shuffles.  tile_out[t] = (A, T).
template <class C>
__global__ void __launch_bounds__(64) k_pip_tiles(PipShape s, const uint32_t* __restrict__ buckets,
                                                  uint32_t* __restrict__ tile_out) {
    constexpr int JW = jac_words<C>();
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t j = s.window_of_tile(t);
    const uint32_t u = t - s.tbase(j);
    const uint32_t nb = s.nb(j);
    const uint32_t b0 = min(nb, u * s.TS + lane * s.S), b1 = min(nb, b0 + s.S);
    const uint32_t* bj = buckets + (size_t)s.bbase(j) * JW;
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
    const uint32_t nt = s.tiles(j);
    const uint32_t* tj = tile_out + (size_t)s.tbase(j) * 2 * JW;
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
    uint32_t span = 1;
    while (span < nt && span < 64) span <<= 1;
    const Jac<C> R = wave_sum_jac<C>(jac_add(asum, x), (int)span);
    if (lane == 0) jac_stg<C>(window_sums + (size_t)j * JW, R);
}

// sum_j 2^off(j) R_j.  A block of two waves; window j doubles its sum off(j) times, all windows at once, and every
// doubling is shared by the 3 (Weierstrass: jac_dbl_tri) or 4 (Edwards: ed_dbl_quad) lanes the window has been given
// (kernels.hpp) -- the critical path is the top window's off(W-1) doublings at three / two product-times each -- then the
// windows' lanes are compacted, a butterfly adds them and the second wave's sum joins through LDS.  More windows than the
// two waves hold (an explicit narrow width): one lane per window, windows lane, lane + 64, .. by Horner's rule.  The result
// goes to out_jac (may be null) and, as the affine wire point, to out_wire (may be null).
template <class C>
__global__ void __launch_bounds__(128) k_pip_final(PipShape s, const uint32_t* __restrict__ window_sums,
                                                   uint32_t* __restrict__ out_jac, uint32_t* __restrict__ out_wire) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    constexpr uint32_t LPW = dbl_lanes<C>(), PER = 64 / LPW;   // lanes per window, windows per wave
    __shared__ __align__(16) uint32_t lds[JW];
    if (blockIdx.x != 0) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    Jac<C> acc = jac_inf<C>();
    if (s.W <= 2 * PER) {
        const uint32_t q = lane / LPW, j = wave * PER + q;
        const bool has = q < PER && j < s.W;
        acc = has ? jac_ldg<C>(window_sums + (size_t)j * JW) : jac_inf<C>();
        const uint32_t times = has ? s.off(j) : 0u;
        const uint32_t last = min(s.W, (wave + 1) * PER);                   // one past this wave's top window
        const uint32_t maxt = last > wave * PER ? s.off(last - 1) : 0u;     // wave-uniform
        for (uint32_t t = 0; t < maxt; t++) {
            const Jac<C> d = jac_dbl_shared<C>(acc);
            if (t < times) acc = d;
        }
        acc = wave_shfl(acc, (int)((LPW * lane) & 63u));
        if (lane >= PER || wave * PER + lane >= s.W) acc = jac_inf<C>();
        acc = wave_sum_jac<C>(acc, 32);
        if (wave == 1 && lane == 0) jac_store(acc, lds);
        __syncthreads();
        if (threadIdx.x != 0) return;
        acc = jac_add(acc, jac_load<C>(lds));
    } else {
        if (wave != 0) return;
        uint32_t at = 0;   // acc is in units of 2^at
        for (uint32_t hi = ((s.W - 1 - lane) / 64) * 64 + lane; lane < s.W; hi -= 64) {   // windows lane + 64 i, highest first
            if (!acc.is_inf())
                for (uint32_t t = s.off(hi); t < at; t++) acc = jac_dbl(acc);
            at = s.off(hi);
            acc = jac_add(acc, jac_ldg<C>(window_sums + (size_t)hi * JW));
            if (hi < 64) break;
        }
        if (!acc.is_inf())
            for (uint32_t t = 0; t < at; t++) acc = jac_dbl(acc);
        acc = wave_sum_jac<C>(acc, 64);
        if (lane != 0) return;
    }
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
    size_t points, split, rec_item, rec_low, sorted, ccount, cstart, ccursor, counts, offsets, segbase, chunk_first, wtotal, segs,
        buckets, tiles, wsums, hlist, hcount, bad, total;
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
    w.split = o;
    o += al((size_t)s.n * PIP_SPLIT_WORDS * 4);
    w.rec_item = o;
    o += al((size_t)s.W * s.items * 4);
    w.rec_low = o;
    o += al((size_t)s.W * s.items * 2);
    w.sorted = o;
    o += al((size_t)s.W * s.istride * 4 + 64);
    const size_t ncb = pip_ncoarse_total(s);
    w.ccount = o;
    o += al(ncb * 4);
    w.cstart = o;
    o += al(ncb * 4);
    w.ccursor = o;
    o += al(ncb * 4);
    w.counts = o;
    o += al((size_t)s.nbuckets * 4);
    w.offsets = o;
    o += al((size_t)s.nbuckets * 4);
    w.segbase = o;
    o += al((size_t)s.nbuckets * 4);
    w.chunk_first = o;
    o += al((size_t)s.W * s.cpw * 4);
    w.wtotal = o;
    o += al((size_t)s.W * 4);
    w.segs = o;
    o += al((size_t)s.W * s.capseg * seg_words<C>() * 4);
    w.buckets = o;
    o += al((size_t)s.nbuckets * JW * 4);
    w.tiles = o;
    o += al((size_t)s.ntiles * 2 * JW * 4);
    w.wsums = o;
    o += al((size_t)s.W * JW * 4);
    w.hlist = o;
    o += al((size_t)s.nbuckets * 4);
    w.hcount = o;
    w.bad = o + 4;   // private status word, right behind the heavy-bucket counter
    o += al(8);
    w.total = o;
    return w;
}

// Enqueues the whole pipeline on `st`.  d_wire_points: n wire points; d_scalars: n canonical scalars (values >= r are
// reduced).  d_out_wire: one wire point; d_status (may be null): 0, or 1 when a point was not on the curve (it counts
// as infinity).  n >= 1.  ev (may be null): PIP_STAGES + 1 events recorded around the stage groups
// [points, digits, scan, scatter | chunks | fold | tiles, windows | final].
constexpr int PIP_STAGES = 5;
template <class C>
inline hipError_t pip_launch(const PipShape& s, const uint32_t* d_scalars, const uint32_t* d_wire_points, uint8_t* d_ws,
                             uint32_t* d_out_wire, uint32_t* d_status, hipStream_t st, hipEvent_t* ev = nullptr) {
    auto mark = [&](int i) {
        if (ev) (void)hipEventRecord(ev[i], st);
    };
    const PipWorkspace w = pip_workspace<C>(s);
    auto at = [&](size_t off) { return reinterpret_cast<uint32_t*>(d_ws + off); };
    uint32_t *points = at(w.points), *split = at(w.split), *rec_item = at(w.rec_item), *sorted = at(w.sorted);
    uint16_t* rec_low = reinterpret_cast<uint16_t*>(d_ws + w.rec_low);
    uint32_t *ccount = at(w.ccount), *cstart = at(w.cstart), *ccursor = at(w.ccursor);
    uint32_t *counts = at(w.counts), *offsets = at(w.offsets), *segbase = at(w.segbase), *chunk_first = at(w.chunk_first);
    uint32_t *wtotal = at(w.wtotal), *segs = at(w.segs), *buckets = at(w.buckets), *tiles = at(w.tiles);
    uint32_t *wsums = at(w.wsums), *hlist = at(w.hlist), *hcount = at(w.hcount);
    uint32_t* bad = d_status ? d_status : at(w.bad);
    const uint32_t ncb = pip_ncoarse_total(s);
    hipError_t e = zero_words_async(ccount, (size_t)ncb * 4, st);
    if (e != hipSuccess) return e;
    e = zero_words_async(hcount, 8, st);   // hcount and the private `bad` word behind it
    if (e != hipSuccess) return e;
    if (d_status) {
        e = zero_words_async(d_status, 4, st);
        if (e != hipSuccess) return e;
    }
    mark(0);
    hipLaunchKernelGGL(k_pip_points<C>, dim3((s.n + 127) / 128), dim3(128), 0, st, s, d_wire_points, d_scalars, points, split, bad);
    const dim3 cgrid((s.n + PIP_CHUNK - 1) / PIP_CHUNK, s.W);
    hipLaunchKernelGGL(k_pip_count<C>, cgrid, dim3(256), 0, st, s, split, ccount);
    hipLaunchKernelGGL(k_pip_cscan, dim3(1), dim3(256), 0, st, s, ccount, cstart, ccursor, wtotal);
    hipLaunchKernelGGL(k_pip_place<C>, cgrid, dim3(256), 0, st, s, split, cstart, ccursor, rec_item, rec_low);
    hipLaunchKernelGGL(k_pip_binsort<C>, dim3(ncb), dim3(PIP_FINE_MAX), 0, st, s, ccount, cstart, rec_item, rec_low, counts, offsets,
                       sorted);
    hipLaunchKernelGGL(k_pip_segments<C>, dim3(s.W), dim3(1024), 0, st, s, counts, offsets, segbase, chunk_first);
    mark(1);
    const uint32_t bpw = (s.cpw + PIP_BLOCK - 1) / PIP_BLOCK;
    hipLaunchKernelGGL(k_pip_chunks<C>, dim3(s.W * bpw), dim3(PIP_BLOCK), pip_ring_bytes<C>(), st, s, points, sorted,
                       offsets, segbase, chunk_first, wtotal, segs);
    mark(2);
    hipLaunchKernelGGL(k_pip_fold<C>, dim3((unsigned)(((size_t)s.nbuckets * s.fl + 127) / 128)), dim3(128), 0, st, s, offsets, counts, segbase, segs,
                       buckets, hlist, hcount);
    // buckets spread over many chunks are few (none at all for uniformly distributed digits): a small grid that strides
    // over the list
    hipLaunchKernelGGL(k_pip_fold_heavy<C>, dim3(std::min<uint32_t>(s.nbuckets, 256u)), dim3(64), 0, st, s, offsets, counts,
                       segbase, segs, buckets, hlist, hcount);
    mark(3);
    hipLaunchKernelGGL(k_pip_tiles<C>, dim3(s.ntiles), dim3(64), 0, st, s, buckets, tiles);
    hipLaunchKernelGGL(k_pip_windows<C>, dim3(s.W), dim3(64), 0, st, s, tiles, wsums);
    mark(4);
    hipLaunchKernelGGL(k_pip_final<C>, dim3(1), dim3(128), 0, st, s, wsums, (uint32_t*)nullptr, d_out_wire);
    mark(5);
    return hipGetLastError();
}

}  // namespace bpp
