// pippenger.hpp -- bucket-method MulVec for LARGE variable-base inputs on gfx950.
//
// The reference's MulVec::calculate (src/bls12_381/building_block/mulvec.rs:20-33) is one scalar
// multiplication per term.  For the sizes the reference itself produces (N <= 2 089) the data-parallel
// restatement in kernels.hpp (k_msm_naive_partial) already finishes in one scalar-mult latency; this file
// is the path for N in the tens of thousands and up (bpp_msm on big inputs, and the combined batch check
// of impl_verify.hpp, which is one MulVec over every proof-carried point of a batch).
//
// Signed c-bit windows (digit j of scalar + bias, minus 2^(c-1)), W = ceil(258 / c) windows, 2^(c-1)
// buckets per window.  Pipeline (all on one stream, no host round trips):
//   k_pip_digits   per point: W digits -> key (bucket, sign); bucket histogram with returning atomics, the
//                  returned value is the point's slot inside its bucket
//   k_pip_scan     per window: exclusive prefix sum of the histogram (LDS, one block per window)
//   k_pip_scatter  per (window, point): sorted[offset[bucket] + slot] = point index | sign
//   k_pip_buckets  per (window, bucket): XYZZ running sum of its points (gathered from HBM); buckets with
//                  more than PIP_HEAVY points are deferred to
//   k_pip_heavy / k_pip_heavy_fold   which split one bucket over PIP_SPLIT x 128 lanes
//   k_pip_windows  per window: sum_k (k+1) * B_k  by per-thread running sums over bucket segments, a small
//                  scalar multiplication for the segment offset, and an LDS tree over the block
//   k_pip_final    Horner over the W window sums (one wave, a tree over the windows), written as a jacobian
#pragma once
#include <algorithm>

#include "kernels.hpp"

namespace bpp {

struct PipShape {
    uint32_t n;                // points
    uint32_t c, W, half;       // window bits, windows, buckets per window 2^(c-1)
    uint32_t heavy;            // a bucket with more points than this is split over many lanes
    uint32_t bias[10];         // sum_j half * 2^(c j)
};

inline PipShape pip_shape(size_t n, int c) {
    PipShape s;
    s.n = (uint32_t)n;
    s.c = (uint32_t)c;
    s.W = (258 + c - 1) / c;
    s.half = 1u << (c - 1);
    // heavy = far above the load of a uniformly filled window (n / half points per bucket)
    s.heavy = (uint32_t)std::max<size_t>(96, 6 * (n / s.half + 1));
    for (int t = 0; t < 10; t++) s.bias[t] = 0;
    for (uint32_t j = 0; j < s.W; j++) {
        const uint32_t bit = s.c * j + (s.c - 1);
        s.bias[bit >> 5] |= 1u << (bit & 31);
    }
    return s;
}

// window width for n points: about 16 points per bucket, within [7, 16]
inline int pip_pick_c(size_t n) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg - 3;
    return c < 7 ? 7 : (c > 16 ? 16 : c);
}

constexpr uint32_t PIP_EMPTY = 0xffffffffu;

// keys / slots: [W][n]; counts: [W][half] (zeroed by the caller).  (Templated on the curve only to get
// vague linkage: this header is included by several translation units.)
template <class C>
__global__ void __launch_bounds__(256) k_pip_digits(PipShape s, const uint32_t* __restrict__ scalars,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ slots,
                                                    uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    uint32_t w[10];
    ld_words<8>(scalars + (size_t)i * 8, w);
    w[8] = 0;
    w[9] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int t = 0; t < 10; t++) {
        uint64_t x = (uint64_t)w[t] + s.bias[t] + carry;
        w[t] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
    const uint32_t mask = (1u << s.c) - 1u;
    for (uint32_t j = 0; j < s.W; j++) {
        const int32_t dg = (int32_t)(w[0] & mask) - (int32_t)s.half;
#pragma unroll
        for (int t = 0; t < 9; t++) w[t] = (w[t] >> s.c) | (w[t + 1] << (32 - s.c));
        w[9] >>= s.c;
        uint32_t key = PIP_EMPTY, slot = 0;
        if (dg != 0) {
            const uint32_t b = (dg < 0 ? (uint32_t)(-dg) : (uint32_t)dg) - 1;
            slot = atomicAdd(&counts[(size_t)j * s.half + b], 1u);
            key = (b << 1) | (dg < 0 ? 1u : 0u);
        }
        keys[(size_t)j * s.n + i] = key;
        slots[(size_t)j * s.n + i] = slot;
    }
}

// one block per window: offsets[j][b] = exclusive prefix sum of counts[j][.]
template <class C>
__global__ void __launch_bounds__(1024) k_pip_scan(PipShape s, const uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ offsets) {
    __shared__ uint32_t part[1024];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (s.half + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = t * per, hi = min(lo + per, s.half);
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

// sorted: [W][n] (only the first sum(counts[j]) entries of each row are written)
template <class C>
__global__ void __launch_bounds__(256) k_pip_scatter(PipShape s, const uint32_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ slots,
                                                     const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ sorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= s.n) return;
    const uint32_t key = keys[(size_t)j * s.n + i];
    if (key == PIP_EMPTY) return;
    const uint32_t b = key >> 1;
    sorted[(size_t)j * s.n + offsets[(size_t)j * s.half + b] + slots[(size_t)j * s.n + i]] = (i << 1) | (key & 1u);
}

// Buckets holding more than PIP_HEAVY points are not summed by one lane: they go to a list and are split
// over PIP_SPLIT blocks of 128 lanes each (k_pip_heavy), then folded back (k_pip_heavy_fold).  This is not a
// corner case: the top window of a scalar < 2^255 holds only a carry bit, so half of all points can land
// in ONE bucket there.
constexpr uint32_t PIP_HEAVY = 96;
constexpr uint32_t PIP_WIN_BLOCK = 256;   // lanes per window in k_pip_windows (latency bound; at 512 lanes the 256-register budget spilled 140 B)
constexpr uint32_t PIP_SPLIT = 16;

// one thread per (window, bucket): bucket sum as a jacobian in buckets[j][b]
template <class C>
__global__ void __launch_bounds__(128, 2) k_pip_buckets(PipShape s, const uint32_t* __restrict__ points,
                                                        const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint32_t* __restrict__ counts,
                                                        uint32_t* __restrict__ buckets,
                                                        uint32_t* __restrict__ heavy_list,
                                                        uint32_t* __restrict__ heavy_count) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)s.W * s.half) return;
    const uint32_t j = (uint32_t)(gid / s.half);
    const uint32_t beg = offsets[gid], cnt = counts[gid];
    if (cnt > s.heavy) {
        heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)gid;
        return;
    }
    const uint32_t* row = sorted + (size_t)j * s.n;
    Xyzz<C> acc = xyzz_inf<C>();
    for (uint32_t t = 0; t < cnt; t++) {
        const uint32_t e = row[beg + t];
        const Aff<C> q = aff_ldg<C>(points + (size_t)(e >> 1) * 2 * N);
        xyzz_madd_lazy(acc, q, (e & 1u) != 0);
    }
    jac_stg<C>(buckets + gid * JW, xyzz_to_jac(acc));
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
        const uint32_t j = gid / s.half;
        const uint32_t beg = offsets[gid], cnt = counts[gid];
        const uint32_t per = (cnt + PIP_SPLIT - 1) / PIP_SPLIT;
        const uint32_t lo = min(cnt, blockIdx.y * per), hi = min(cnt, lo + per);
        const uint32_t* row = sorted + (size_t)j * s.n + beg;
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
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        Jac<C> acc = jac_ldg<C>(heavy_parts + ((size_t)h * PIP_SPLIT + threadIdx.x) * JW);
        acc = block_reduce_jac<C>(acc, lds);
        if (threadIdx.x == 0) jac_stg<C>(buckets + (size_t)heavy_list[h] * JW, acc);
        __syncthreads();
    }
    (void)N;
}

// one block per window: R_j = sum_b (b + 1) * B_b.  Thread t owns the segment [t*S, (t+1)*S) with
// S = half / blockDim.x (>= 1): descending running sums give sum (b - lo + 1) B_b and sum B_b, the
// segment offset lo is applied with a short double-and-add, an LDS tree adds the threads.
template <class C>
__global__ void __launch_bounds__(PIP_WIN_BLOCK) k_pip_windows(PipShape s, const uint32_t* __restrict__ buckets,
                                                        uint32_t* __restrict__ window_sums) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t S = max(1u, s.half / blockDim.x);
    const uint32_t lo = t * S;
    Jac<C> total = jac_inf<C>();
    if (lo < s.half) {
        const uint32_t hi = min(lo + S, s.half);
        const uint32_t* bj = buckets + (size_t)j * s.half * JW;
        Jac<C> run = jac_inf<C>(), acc = jac_inf<C>();
        for (uint32_t b = hi; b-- > lo;) {
            run = jac_add(run, jac_ldg<C>(bj + (size_t)b * JW));
            acc = jac_add(acc, run);
        }
        // + lo * run
        Jac<C> off = jac_inf<C>();
        for (int bit = 31; bit >= 0; bit--) {
            off = jac_dbl(off);
            if ((lo >> bit) & 1u) off = jac_add(off, run);
        }
        total = jac_add(acc, off);
    }
    total = block_reduce_jac<C>(total, lds);
    if (t == 0) jac_stg<C>(window_sums + (size_t)j * JW, total);
}

// Horner over the windows; the result is ADDED to `extra` partials (may be none) and written as one
// jacobian to out.  One wave, as a binary tree over the windows (W <= 37 for the c in [7, 16] pip_pick_c chooses): lane j starts with R_j;
// at level l the lanes whose index is a multiple of 2^(l+1) take the partial of lane j + 2^l through LDS, double it
// c * 2^l times and add it.  The ~c (W - 1) doublings of the top window still form one chain, but the W additions
// of Horner's rule shrink to 6 on the critical path (the same form as var_horner_wave in kernels.hpp).
template <class C>
__global__ void __launch_bounds__(64) k_pip_final(PipShape s, const uint32_t* __restrict__ window_sums,
                                                  const uint32_t* __restrict__ extra, uint32_t n_extra,
                                                  uint32_t* __restrict__ out) {
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
    jac_stg<C>(out, acc);
}

struct PipWorkspace {
    size_t keys, slots, sorted, counts, offsets, buckets, wsums, hlist, hcount, hparts, total;
    size_t max_heavy;
};
template <class C>
inline PipWorkspace pip_workspace(const PipShape& s) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    PipWorkspace w;
    size_t o = 0;
    w.keys = o;
    o += al((size_t)s.W * s.n * 4);
    w.slots = o;
    o += al((size_t)s.W * s.n * 4);
    w.sorted = o;
    o += al((size_t)s.W * s.n * 4);
    w.counts = o;
    o += al((size_t)s.W * s.half * 4);
    w.offsets = o;
    o += al((size_t)s.W * s.half * 4);
    w.buckets = o;
    o += al((size_t)s.W * s.half * JW * 4);
    w.wsums = o;
    o += al((size_t)s.W * JW * 4);
    // a heavy bucket holds > s.heavy of the W * n sorted entries
    w.max_heavy = std::min<size_t>((size_t)s.W * s.half, (size_t)s.W * s.n / s.heavy + 1);
    w.hlist = o;
    o += al(w.max_heavy * 4);
    w.hcount = o;
    o += al(4);
    w.hparts = o;
    o += al(w.max_heavy * PIP_SPLIT * JW * 4);
    w.total = o;
    return w;
}

// Enqueues the whole pipeline on `st`.  d_out: one jacobian (3N words).  d_extra: n_extra jacobians added in.
template <class C>
inline hipError_t pip_launch(const PipShape& s, const uint32_t* d_scalars, const uint32_t* d_points, uint8_t* d_ws,
                             const uint32_t* d_extra, uint32_t n_extra, uint32_t* d_out, hipStream_t st) {
    constexpr int N = C::Fp::N;
    constexpr int JW = jac_words<C>();
    const PipWorkspace w = pip_workspace<C>(s);
    uint32_t* keys = reinterpret_cast<uint32_t*>(d_ws + w.keys);
    uint32_t* slots = reinterpret_cast<uint32_t*>(d_ws + w.slots);
    uint32_t* sorted = reinterpret_cast<uint32_t*>(d_ws + w.sorted);
    uint32_t* counts = reinterpret_cast<uint32_t*>(d_ws + w.counts);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(d_ws + w.offsets);
    uint32_t* buckets = reinterpret_cast<uint32_t*>(d_ws + w.buckets);
    uint32_t* wsums = reinterpret_cast<uint32_t*>(d_ws + w.wsums);
    uint32_t* hlist = reinterpret_cast<uint32_t*>(d_ws + w.hlist);
    uint32_t* hcount = reinterpret_cast<uint32_t*>(d_ws + w.hcount);
    uint32_t* hparts = reinterpret_cast<uint32_t*>(d_ws + w.hparts);
    hipError_t e = zero_words_async(counts, (size_t)s.W * s.half * 4, st);
    if (e != hipSuccess) return e;
    e = zero_words_async(hcount, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pip_digits<C>, dim3((s.n + 255) / 256), dim3(256), 0, st, s, d_scalars, keys, slots, counts);
    hipLaunchKernelGGL(k_pip_scan<C>, dim3(s.W), dim3(1024), 0, st, s, counts, offsets);
    hipLaunchKernelGGL(k_pip_scatter<C>, dim3((s.n + 255) / 256, s.W), dim3(256), 0, st, s, keys, slots, offsets, sorted);
    const size_t nb = (size_t)s.W * s.half;
    hipLaunchKernelGGL(k_pip_buckets<C>, dim3((unsigned)((nb + 127) / 128)), dim3(128), 0, st, s, d_points, sorted,
                       offsets, counts, buckets, hlist, hcount);
    // heavy buckets are few (none at all for uniformly distributed digits): a small grid that strides over
    // the list -- an oversized grid of immediately-exiting blocks costs milliseconds
    const unsigned hgrid = (unsigned)std::min<size_t>(w.max_heavy, 64);
    hipLaunchKernelGGL(k_pip_heavy<C>, dim3(hgrid, PIP_SPLIT), dim3(128), 128 * JW * 4, st, s, d_points, sorted,
                       offsets, counts, hlist, hcount, hparts);
    hipLaunchKernelGGL(k_pip_heavy_fold<C>, dim3(hgrid), dim3(PIP_SPLIT), PIP_SPLIT * JW * 4, st, hlist, hcount, hparts,
                       buckets);
    // 512 lanes x 144 B of LDS exceed the 64 KB default for dynamic LDS: opt in (160 KB per CU on gfx950)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pip_windows<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(PIP_WIN_BLOCK * JW * 4));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pip_windows<C>, dim3(s.W), dim3(PIP_WIN_BLOCK), PIP_WIN_BLOCK * JW * 4, st, s, buckets, wsums);
    hipLaunchKernelGGL(k_pip_final<C>, dim3(1), dim3(64), 0, st, s, wsums, d_extra, n_extra, d_out);
    return hipGetLastError();
}

}  // namespace bpp
