// graph_memset_probe.hip -- what exactly goes wrong with hipMemsetAsync inside a captured HIP graph on this ROCm.
// Round 2 found that bpp_verifier_run, captured into a graph, returned "invalid point" for every proof of a 300-proof
// batch from the SECOND replay on, and replaced its hipMemsetAsync calls by a fill kernel (kernels.hpp zero_words_async).
// The only memset of that pass was the per-proof invalid-point flags: count x 4 = 1200 bytes at offset `bad` of the
// caller's workspace.  This probe reproduces the pattern in isolation: a sub-range of a larger allocation is dirtied by a
// kernel, then a captured graph { memset(sub-range, 0, bytes) ; kernel that counts the non-zero bytes } is replayed four
// times (dirtying in between), for several sizes, offsets and both capture modes.  Output: one JSON line per case with the
// non-zero byte count the check kernel saw in each replay (0 = the memset node did its job).
// build + run: hipcc -O2 --offload-arch=gfx950 -o /tmp/gmp tools/graph_memset_probe.hip && /tmp/gmp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_dirty(unsigned char* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0xff;
}
__global__ void k_count(const unsigned char* p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i]) atomicAdd(out, 1u);
}

int main() {
    unsigned char* buf;
    unsigned* cnt;
    const size_t total = 1 << 22;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&cnt, 4 * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t sizes[] = {256, 1024, 1028, 1200, 4096, 65536, 1200000};
    const size_t offs[] = {0, 256, 4096 + 256};
    for (size_t bytes : sizes)
        for (size_t off : offs)
            for (int variant = 0; variant < 8; variant++) {   // bit 0: hipGraphDestroy right after instantiation (as PyTorch does); bit 1: replay on the NULL stream (torch's current stream)
                const int early_destroy = variant & 1, null_stream = (variant >> 1) & 1, churn = variant >> 2;
                hipStream_t ls = null_stream ? nullptr : st;
                unsigned char* p = buf + off;
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                CK(hipMemsetAsync(p, 0, bytes, st));
                CK(hipMemsetAsync(cnt, 0, 4, st));
                hipLaunchKernelGGL(k_count, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, st, p, bytes, cnt);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                if (early_destroy) CK(hipGraphDestroy(g));
                unsigned seen[4];
                for (int r = 0; r < 4; r++) {
                    hipLaunchKernelGGL(k_dirty, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ls, p, bytes);
                    CK(hipStreamSynchronize(ls));
                    if (churn) {   // unrelated work between replays: other launches (their kernel arguments), host and device allocations
                        for (int c = 0; c < 64; c++)
                            hipLaunchKernelGGL(k_dirty, dim3(4), dim3(256), 0, c & 1 ? st : nullptr, buf + (1 << 21) + 4096 * c, (size_t)(1000 + c));
                        void* tmp[8];
                        for (int c = 0; c < 8; c++) CK(hipMalloc(&tmp[c], 1 << (12 + c)));
                        for (int c = 0; c < 8; c++) CK(hipFree(tmp[c]));
                        std::vector<unsigned char> junk(1 << 20, 0xa5);
                        CK(hipMemcpy(buf + (3 << 20), junk.data(), junk.size(), hipMemcpyHostToDevice));
                        CK(hipDeviceSynchronize());
                    }
                    CK(hipGraphLaunch(ge, ls));
                    CK(hipStreamSynchronize(ls));
                    CK(hipMemcpy(&seen[r], cnt, 4, hipMemcpyDeviceToHost));
                }
                printf("{\"bytes\": %zu, \"offset\": %zu, \"graph_destroyed_before_replay\": %d, \"replay_on_null_stream\": %d, \"churn_between_replays\": %d, \"nonzero_after_replay\": [%u, %u, %u, %u]}\n",
                       bytes, off, early_destroy, null_stream, churn, seen[0], seen[1], seen[2], seen[3]);
                CK(hipGraphExecDestroy(ge));
                if (!early_destroy) CK(hipGraphDestroy(g));
            }
    return 0;
}
