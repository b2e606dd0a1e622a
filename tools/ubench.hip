// ubench.hip -- MI355X micro-benchmarks behind DESIGN.md's integer-ALU roofline:
//   (1) issue rate of the VALU instructions the field arithmetic is made of (v_mad_u64_u32 first)
//   (2) throughput of fe_mul / fe_sqr / fe_add and of the group operations built from them.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/ubench tools/ubench.hip
// Prints one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bulletproofsplus_amd/csrc/ec.hpp"
#include "../bulletproofsplus_amd/csrc/ed25519.hpp"
using namespace bpp;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

// kind: 0 v_mad_u64_u32, 1 v_lshl_add_u64, 2 v_and_b32, 3 v_mul_lo_u32, 4 v_fma_f64, 5 v_mad_u32_u24,
//       6 v_lshrrev_b64, 7 v_mul_hi_u32, 8 v_add_u32, 9 v_add3_u32
template <int KIND>
__global__ void __launch_bounds__(256) k_inst(uint64_t* out, uint32_t seed) {
    uint64_t a[CHAINS], b[CHAINS];
    uint32_t x = seed + threadIdx.x, y = seed * 3 + 7;
    double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c] = seed + c; b[c] = seed * 5 + c; d[c] = 1.0 + c; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (KIND == 0) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
            if (KIND == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]));
            if (KIND == 2) { uint32_t t = (uint32_t)a[c]; asm volatile("v_and_b32 %0, %0, %1" : "+v"(t) : "v"(x)); a[c] = t; }
            if (KIND == 3) { uint32_t t = (uint32_t)a[c]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(x)); a[c] = t; }
            if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
            if (KIND == 5) { uint32_t t = (uint32_t)a[c]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(t) : "v"(x)); a[c] = t; }
            if (KIND == 6) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[c]));
            if (KIND == 7) { uint32_t t = (uint32_t)a[c]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(x)); a[c] = t; }
            if (KIND == 8) { uint32_t t = (uint32_t)a[c]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(x)); a[c] = t; }
            if (KIND == 9) { uint32_t t = (uint32_t)a[c]; asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(t) : "v"(x)); a[c] = t; }
            // mixes: does a cheap instruction issue in the shadow of the multiplier?  (ops counted = the mads only)
            if (KIND == 10) { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
                              uint32_t t = (uint32_t)d[c]; asm volatile("v_and_b32 %0, %0, %1" : "+v"(t) : "v"(x)); d[c] = t; }
            if (KIND == 11) { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
                              asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(b[c])); }
            if (KIND == 12) { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
                              asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(b[c]) : "v"(b[(c + 1) % CHAINS])); }
            if (KIND == 13) { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
                              uint32_t t = (uint32_t)d[c], u = (uint32_t)b[c];
                              asm volatile("v_and_b32 %0, %0, %1" : "+v"(t) : "v"(x)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(x));
                              d[c] = t; b[c] = u; }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c] + b[c] + (uint64_t)d[c];
    if (s == 0x1234567) out[0] = s;
}

// kind: 0 fe_mul, 1 fe_sqr, 2 fe_add, 3 fe_sub, 4 fe_inv (safegcd), 5 fe_inv_fermat
template <class P, int KIND>
__global__ void __launch_bounds__(256) k_field(uint32_t* out, uint32_t seed, int iters) {
    Fe<P> a = fe_from_u32<P>(seed + threadIdx.x + 1), b = fe_from_u32<P>(seed * 7 + blockIdx.x + 3 * threadIdx.x + 3);   // both per-lane: a wave-uniform operand would be computed on the scalar unit
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { a = fe_mul(a, b); b = fe_mul(b, a); }
        if (KIND == 1) { a = fe_sqr(a); b = fe_sqr(b); }
        if (KIND == 2) { a = fe_add(a, b); b = fe_add(b, a); }
        if (KIND == 3) { a = fe_sub(a, b); b = fe_sub(b, a); }
        if (KIND == 4) { a = fe_inv(a); b = fe_inv(b); }
        if (KIND == 5) { a = fe_inv_fermat(a); b = fe_inv_fermat(b); }
    }
    if (a.l[0] == 0x3fffffff && b.l[1] == 0x12345) out[0] = a.l[2];
}

// kind: 0 jac_madd, 1 jac_dbl, 2 jac_add, 3 xyzz_madd (the accumulation step of k_fixed_msm)
template <class C, int KIND>
__global__ void __launch_bounds__(128) k_group(uint32_t* out, uint32_t seed, int iters) {
    Aff<C> g = aff_generator<C>();
    uint32_t kw[1] = {seed + threadIdx.x + 2};
    Jac<C> acc = aff_mul_words(g, kw, 1);
    Jac<C> q = jac_dbl(acc);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) acc = jac_madd(acc, g);
        if (KIND == 1) acc = jac_dbl(acc);
        if (KIND == 2) acc = jac_add(acc, q);
    }
    if (acc.X.l[0] == 0x3fffffff && acc.Y.l[1] == 0x12345) out[0] = acc.Z.l[2];
}

// the accumulation step of k_fixed_msm: XYZZ running sum += affine point
template <class C>
__global__ void __launch_bounds__(128, 2) k_xyzz(uint32_t* out, uint32_t seed, int iters) {
    Aff<C> g = aff_generator<C>();
    uint32_t kw[1] = {seed + threadIdx.x + 2};
    Aff<C> p = jac_to_aff(aff_mul_words(g, kw, 1));
    Xyzz<C> xa = xyzz_madd(xyzz_dbl_aff(g), g);
    for (int it = 0; it < iters; it++) xa = xyzz_madd(xa, p);
    if (xa.X.l[0] == 0x3fffffff && xa.Y.l[1] == 0x12345) out[0] = xa.ZZ.l[2];
}

// the same step through the lazy formula (ec.hpp xyzz_madd_lazy): what k_fixed_msm runs
template <class C>
__global__ void __launch_bounds__(128, 2) k_xyzz_lazy(uint32_t* out, uint32_t seed, int iters) {
    Aff<C> g = aff_generator<C>();
    uint32_t kw[1] = {seed + threadIdx.x + 2};
    Aff<C> p = jac_to_aff(aff_mul_words(g, kw, 1));
    fe_cond_sub_p(p.x);
    fe_cond_sub_p(p.y);
    Xyzz<C> xa = xyzz_madd(xyzz_dbl_aff(g), g);
    for (int it = 0; it < iters; it++) xyzz_madd_lazy(xa, p, (it & 1) != 0);
    const Jac<C> r = xyzz_to_jac(xa);
    if (r.X.l[0] == 0x3fffffff && r.Y.l[1] == 0x12345) out[0] = r.Z.l[2];
}

// one dependent v_mad_u64_u32 chain per wave, NCH independent chains: issue-to-issue latency of the multiplier
template <int NCH>
__global__ void __launch_bounds__(64) k_mad_lat(uint64_t* out, uint32_t seed) {
    uint64_t a[NCH];
    uint32_t x = seed + threadIdx.x, y = seed * 3 + 7;
#pragma unroll
    for (int c = 0; c < NCH; c++) a[c] = seed + c;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < NCH; c++)
                asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "s10", "s11");
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) s += a[c];
    if (s == 0x1234567) out[0] = s;
}

template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint64_t* dout;
    CK(hipMalloc(&dout, 4096));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n", prop.name, cus, prop.clockRate / 1000);
    const char* names[] = {"v_mad_u64_u32", "v_lshl_add_u64", "v_and_b32", "v_mul_lo_u32", "v_fma_f64", "v_mad_u32_u24",
                           "v_lshrrev_b64", "v_mul_hi_u32", "v_add_u32", "v_add3_u32", "mad_with_and_1to1", "mad_with_lshrrev_b64_1to1",
                           "mad_with_lshl_add_u64_1to1", "mad_with_and_add_1to2"};
    // 8 waves per SIMD: 256-thread blocks, 8 per CU
    const int grid = cus * 8;
    auto inst = [&](int kind, auto kern) {
        double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dout, 12345u); }, 5);
        double ops = (double)grid * 256 * ITERS * CHAINS;
        double per_cu_clk = ops / (ms * 1e-3) / cus / (prop.clockRate * 1e3);
        printf(" \"%s\": {\"Gops\": %.1f, \"lanes_per_clk_per_cu\": %.2f},\n", names[kind], ops / ms * 1e-6, per_cu_clk);
    };
    inst(0, k_inst<0>); inst(1, k_inst<1>); inst(2, k_inst<2>); inst(3, k_inst<3>); inst(4, k_inst<4>);
    inst(5, k_inst<5>); inst(6, k_inst<6>); inst(7, k_inst<7>); inst(8, k_inst<8>); inst(9, k_inst<9>);
    inst(10, k_inst<10>); inst(11, k_inst<11>); inst(12, k_inst<12>); inst(13, k_inst<13>);
    // dependent multiplier chains, ONE wave per SIMD (4 waves of 64 threads per CU): cycles between dependent issues
    auto lat = [&](const char* name, auto kern, int nch) {
        const int g = cus * 4;
        double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(64), 0, 0, dout, 12345u); }, 3);
        double per_wave = (double)ITERS * 8 * nch;                      // mads issued by one wave
        printf(" \"%s\": {\"clk_per_mad_one_wave_per_simd\": %.2f},\n", name, ms * 1e-3 * prop.clockRate * 1e3 / per_wave);
    };
    lat("mad_chain_1", k_mad_lat<1>, 1); lat("mad_chain_2", k_mad_lat<2>, 2); lat("mad_chain_4", k_mad_lat<4>, 4);

    uint32_t* o32 = reinterpret_cast<uint32_t*>(dout);
    auto fld = [&](const char* name, auto kern, int iters, int block, int blocks_per_cu) {
        const int g = cus * blocks_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(block), 0, 0, o32, 99u, iters); }, 3);
        double ops = (double)g * block * iters * 2;
        printf(" \"%s\": {\"Gops\": %.2f},\n", name, ops / ms * 1e-6);
    };
    fld("fe_mul_blsfp", k_field<BlsFp, 0>, 2000, 256, 8);
    fld("fe_sqr_blsfp", k_field<BlsFp, 1>, 2000, 256, 8);
    fld("fe_add_blsfp", k_field<BlsFp, 2>, 4000, 256, 8);
    fld("fe_sub_blsfp", k_field<BlsFp, 3>, 4000, 256, 8);
    fld("fe_mul_blsfr", k_field<BlsFr, 0>, 2000, 256, 8);
    fld("fe_sqr_blsfr", k_field<BlsFr, 1>, 2000, 256, 8);
    fld("fe_mul_secpfp", k_field<SecpFp, 0>, 2000, 256, 8);
    fld("fe_inv_blsfp", k_field<BlsFp, 4>, 40, 256, 8);
    fld("fe_inv_fermat_blsfp", k_field<BlsFp, 5>, 4, 256, 8);
    fld("fe_inv_blsfr", k_field<BlsFr, 4>, 40, 256, 8);
    fld("fe_inv_secpfp", k_field<SecpFp, 4>, 40, 256, 8);
    auto grp = [&](const char* name, auto kern, int iters, int blocks_per_cu) {
        const int g = cus * blocks_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(128), 0, 0, o32, 5u, iters); }, 3);
        double ops = (double)g * 128 * iters;
        printf(" \"%s\": {\"Gops\": %.3f},\n", name, ops / ms * 1e-6);
    };
    grp("jac_madd_bls", k_group<Bls12381, 0>, 300, 8);
    grp("jac_dbl_bls", k_group<Bls12381, 1>, 300, 8);
    grp("jac_add_bls", k_group<Bls12381, 2>, 300, 8);
    grp("xyzz_madd_bls", k_xyzz<Bls12381>, 300, 8);
    grp("xyzz_madd_secp", k_xyzz<Secp256k1>, 300, 8);
    grp("xyzz_madd_lazy_bls", k_xyzz_lazy<Bls12381>, 300, 8);
    grp("xyzz_madd_lazy_secp", k_xyzz_lazy<Secp256k1>, 300, 8);
    grp("xyzz_madd_lazy_ed", k_xyzz_lazy<Ed25519>, 300, 8);   // the unified extended-coordinate addition of k_fixed_msm<Ed25519>
    grp("jac_madd_secp", k_group<Secp256k1, 0>, 300, 8);
    grp("jac_dbl_secp", k_group<Secp256k1, 1>, 300, 8);
    printf(" \"end\": 0}\n");
    return 0;
}
