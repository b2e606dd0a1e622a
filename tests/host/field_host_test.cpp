// Host-side harness for csrc/field.hpp and csrc/ec.hpp (compiled with g++, no GPU needed).
// Reads lines:  <field> <op> <hex a> [<hex b>]   and prints the canonical hex result.
// Fields: blsfp blsfr secpfp secpfr edfp edfr.  Ops: mul sqr add sub neg inv invf(Fermat) tocanon(roundtrip) pow5
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include <sstream>
#include "../../bulletproofsplus_amd/csrc/field.hpp"
using namespace bpp;

template <class P> static void parse_hex(const std::string& h, uint32_t* w) {
    for (int i = 0; i < P::N; i++) w[i] = 0;
    int n = (int)h.size();
    for (int i = 0; i < n; i++) {
        char c = h[n - 1 - i];
        uint32_t d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
        if (i / 8 < P::N) w[i / 8] |= d << (4 * (i % 8));
    }
}
template <class P> static void print_hex(const uint32_t* w) {
    for (int i = P::N - 1; i >= 0; i--) printf("%08x", w[i]);
    printf("\n");
}
template <class P> static void run(const std::string& op, const std::string& ha, const std::string& hb) {
    uint32_t wa[P::N], wb[P::N], wr[P::N];
    parse_hex<P>(ha, wa); parse_hex<P>(hb.empty() ? "0" : hb, wb);
    Fe<P> a = fe_from_canonical<P>(wa), b = fe_from_canonical<P>(wb), r;
    if (op == "mul") r = fe_mul(a, b);
    else if (op == "sqr") r = fe_sqr(a);
    else if (op == "add") r = fe_add(a, b);
    else if (op == "sub") r = fe_sub(a, b);
    else if (op == "neg") r = fe_neg(a);
    else if (op == "muladd") r = fe_mul_add(a, b, fe_neg(b), fe_add(a, a));   // a b - 2 a b = -a b, at the 2p bound
    else if (op == "inv") r = fe_inv(a);
    else if (op == "invf") r = fe_inv_fermat(a);
    else if (op == "pow5") r = fe_pow_u64(a, 5);
    else if (op == "tocanon") { uint32_t m[P::N]; fe_store(a, m); r = fe_load<P>(m); }
    else { printf("bad op\n"); return; }
    fe_to_canonical(r, wr);
    print_hex<P>(wr);
}
int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string f, op, a, b;
        is >> f >> op >> a >> b;
        if (f == "blsfp") run<BlsFp>(op, a, b);
        else if (f == "blsfr") run<BlsFr>(op, a, b);
        else if (f == "secpfp") run<SecpFp>(op, a, b);
        else if (f == "secpfr") run<SecpFr>(op, a, b);
        else if (f == "edfp") run<EdFp>(op, a, b);
        else if (f == "edfr") run<EdFr>(op, a, b);
    }
    return 0;
}
