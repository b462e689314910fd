// Host-side check of the bit-layout helpers of csrc/gml_bits.h (compiled by tests/test_bit_layouts.py with g++).
#include "../../graphicalmodellearning.jl_amd/csrc/gml_bits.h"
#include <cstdio>
#include <cstdlib>
using namespace gml;
static uint32_t rnd(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
int main() {
    uint64_t seed = 1;
    for (int s = 0; s < 64; ++s)
        if (vq_sample(vq_pos(s)) != s || vq_pos(vq_sample(s)) != s) { printf("vq_pos/vq_sample not inverse at %d\n", s); return 1; }
    // Xtb dword h' of a step: bit 4t + e + 8b <-> operand position 32t + 16h' + 4e + b, which holds sample vq_sample(position);
    // the natural word holds sample 32h' + j at bit j
    for (int rep = 0; rep < 1000; ++rep) {
        const uint32_t x = rnd(seed);
        const uint32_t y = xtb_from_natural(x);
        for (int hp = 0; hp < 2; ++hp)
            for (int t = 0; t < 2; ++t)
                for (int e = 0; e < 4; ++e)
                    for (int b = 0; b < 4; ++b) {
                        const int pos = 32 * t + 16 * hp + 4 * e + b, smp = vq_sample(pos);
                        if ((smp >> 5) != hp) { printf("position %d of half %d holds sample %d\n", pos, hp, smp); return 1; }
                        if (((y >> (4 * t + e + 8 * b)) & 1u) != ((x >> (smp & 31)) & 1u)) { printf("xtb_from_natural wrong\n"); return 1; }
                    }
    }
    // Xb: bit j of dword h <-> column xb_col(j, h); together the two halves cover the 64 columns once
    int seen[64] = {0};
    for (int h = 0; h < 2; ++h)
        for (int j = 0; j < 32; ++j) ++seen[xb_col(j, h)];
    for (int c = 0; c < 64; ++c)
        if (seen[c] != 1) { printf("xb_col does not cover column %d once\n", c); return 1; }
    // fragment dword e' of K-half t = (v >> (4t + e')) & 0x01010101: byte b <-> column 32t + 16h + 4e' + b
    for (int h = 0; h < 2; ++h)
        for (int t = 0; t < 2; ++t)
            for (int e = 0; e < 4; ++e)
                for (int b = 0; b < 4; ++b)
                    if (xb_col(4 * t + e + 8 * b, h) != 32 * t + 16 * h + 4 * e + b) { printf("xb_col mismatch\n"); return 1; }
    for (int rep = 0; rep < 200; ++rep) {
        uint32_t a[32], o[32];
        for (int i = 0; i < 32; ++i) o[i] = a[i] = rnd(seed);
        transpose32(a);
        for (int s = 0; s < 32; ++s)
            for (int j = 0; j < 32; ++j)
                if (((a[s] >> j) & 1u) != ((o[j] >> s) & 1u)) { printf("transpose32 wrong at (%d,%d)\n", s, j); return 1; }
    }
    printf("ok\n");
    return 0;
}
