// Bit-level layout helpers shared by the device kernels and by the host-side unit test
// (tests/test_bit_layouts.py compiles this header with g++): no HIP types in here.
//
// The +-1 design matrix lives in HBM as ONE BIT per entry (set <=> -1).  Three images:
//   Sb   spin-major sign bits of the n spins, natural order: word w of row i holds samples 32w .. 32w+31,
//        bit j <-> sample 32w + j.  Everything else is derived from it (a statistic of key S is the XOR
//        of the rows of its spins: prod of +-1 = parity of the sign bits).
//   Xb   forward operand (sample-major), dword (k, kt, h): bit e + 8b <-> column 64kt + 32(e>>2) + 16h +
//        4(e&3) + b, so that dword e' of the MFMA fragment for K-half t is (v >> (4t + e')) & 0x01010101.
//   Xtb  backward operand (feature-major), dword (c, kt, h): bit e + 8b <-> operand position
//        32(e>>2) + 16h + 4(e&3) + b of the 64-sample step kt, which holds sample 64kt + vq_sample(position)
//        (the sample order of the Vq limb images written by the forward epilogue).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GML_HD __host__ __device__ inline
#else
#define GML_HD inline
#endif

namespace gml {

// Sample order of a 64-sample step inside the Vq limb images (and of Xtb / Mb / Hq): the forward epilogue's lane
// (node, half h) owns the samples 32 i + 8 g + 4 h + j (i < 2, g < 4, j < 4) and stores them at byte
// 32 h + 16 i + 4 g + j, so that its 32 bytes per limb are contiguous.
GML_HD int vq_pos(int s) { return ((s >> 2) & 1) * 32 + (s >> 5) * 16 + ((s >> 3) & 3) * 4 + (s & 3); }
GML_HD int vq_sample(int p) { return ((p >> 4) & 1) * 32 + ((p >> 2) & 3) * 8 + (p >> 5) * 4 + (p & 3); }

// Xtb dword from the natural-order word of the same 32 samples (word 2kt + h' of the statistic's bit row):
// dword h' covers the samples 64kt + 32h' + (8e + 4t + b) at bit 4t + e + 8b  (e < 4, t < 2, b < 4), i.e. the
// 2-bit fields e and b of the bit index are exchanged: two delta swaps.
GML_HD uint32_t xtb_from_natural(uint32_t x) {
    uint32_t t = ((x >> 14) ^ x) & 0x0000CCCCu; // index bit 4 <-> index bit 1
    x ^= t ^ (t << 14);
    t = ((x >> 7) ^ x) & 0x00AA00AAu; // index bit 3 <-> index bit 0
    x ^= t ^ (t << 7);
    return x;
}

// column offset (within the 64-column step) of bit j of the Xb dword of half h
GML_HD int xb_col(int j, int h) {
    const int e = j & 7, b = j >> 3;
    return 32 * (e >> 2) + 16 * h + 4 * (e & 3) + b;
}

// 32 x 32 bit transpose in place: afterwards bit j of a[s] = bit s of the old a[j]
GML_HD void transpose32(uint32_t (&a)[32]) {
    uint32_t m = 0x0000FFFFu;
    for (int j = 16; j != 0; j >>= 1, m ^= (m << j)) {
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j;
            a[k + j] ^= t;
        }
    }
}

} // namespace gml
