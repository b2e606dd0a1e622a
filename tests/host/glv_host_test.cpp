// Host-side check of the GLV scalar split of csrc/ec.hpp (glv_split, the device's k_var_digits runs the same code),
// compiled with g++: reads scalars (64 hex digits each) from the command line, prints "k1 k2" (32 hex digits each) per
// scalar.  tests/test_host_arith_cpu.py compares with Python integers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../bulletproofsplus_amd/csrc/ec.hpp"
using namespace bpp;

// argv[1] == "secp": the signed split of secp256k1 (glv_split_signed): prints "s1 k1 s2 k2" (signs as 0 / 1)
int main(int argc, char** argv) {
    const bool secp = argc > 1 && strcmp(argv[1], "secp") == 0;
    for (int a = secp ? 2 : 1; a < argc; a++) {
        const char* h = argv[a];
        if (strlen(h) != 64) return 2;
        uint32_t k[8], k1[4], k2[4];
        for (int w = 0; w < 8; w++) {
            char buf[9];
            memcpy(buf, h + 8 * (7 - w), 8);
            buf[8] = 0;
            k[w] = (uint32_t)strtoul(buf, nullptr, 16);
        }
        if (secp) {
            bool n1, n2;
            glv_split_signed<Secp256k1>(k, k1, k2, n1, n2);
            printf("%d %08x%08x%08x%08x %d %08x%08x%08x%08x\n", n1 ? 1 : 0, k1[3], k1[2], k1[1], k1[0], n2 ? 1 : 0, k2[3], k2[2], k2[1],
                   k2[0]);
            continue;
        }
        glv_split<Bls12381>(k, k1, k2);
        printf("%08x%08x%08x%08x %08x%08x%08x%08x\n", k1[3], k1[2], k1[1], k1[0], k2[3], k2[2], k2[1], k2[0]);
    }
    return 0;
}
