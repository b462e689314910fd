// Int8-limb path, part 4: working-set Hessians on the int8 matrix cores (overview: gml_i8.h).
#include "gml_i8.h"
#include "gml_solver.h"
#include <algorithm>
#include <string>
#include <type_traits>

namespace gml {

// ------------------------------------------------------------------------------------------
// Working-set Hessian on the int8 matrix cores.
//   H_r[i][j] = sum_k h_rk x_ki x_kj,  x = +-1 = 1 - 2b  (b = 1 where x = -1)
//             = S - 2 T_ii - 2 T_jj + 4 T_ij,   T_ij = sum_k h_rk b_ki b_kj,  S = sum_k h_rk,
// computed per base-256 digit plane h_l of the (non-negative) weight as (mask_i & h_l) * b_j with the byte masks
// 0x00 / 0xFF made from the bits: exact integer GEMMs.
//
// The weight enters as HL = 2 digit planes: the 31-bit magnitude the V planes hold, cut to the 15 bits below the row's
// largest (unbiased: a dither is added before the shift).  These Hessians run over a sub-sample of the configurations and
// are secant-corrected (gml_solver.cpp), so their own error is percent-level; the weights' 2^-15 is far inside it also
// when a block uses every configuration (a Newton matrix, not a result: the gradient is what fixes the optimum).  Rounds
// 1-4 carried all four planes, i.e. twice the matrix work and twice the weight bytes.
//
// Sub-sampled Newton: the sum runs over `Kh` configurations taken as every `kstride`-th block of 512 (block cb of
// the compact index <-> samples [512 cb kstride, +512)): spread over the whole histogram, whose rows are usually
// sorted, instead of its first rows.
// ------------------------------------------------------------------------------------------
constexpr int HL = 2;

// bits by which a row's weights are shifted down so that its largest fits the 15 bits of two balanced digits.  exp forms: mm = the row's largest |V| in the unit
// of the planes, as its last pass recorded it.  RPLE: the passes do not record it, and need not: tau comes from the bound 2 w_max,
// which the weights of a well-classified configuration reach, and h = 4 w sig (1 - sig) <= w_max = half the planes' range (2^30).
__device__ __forceinline__ int hw_shift(unsigned mm, int form) {
    if (form == 2) return 16;
    int sh = 0;
    while ((mm >> sh) + 1u > 32639u) ++sh; // (the dither adds less than one unit after the shift; 32 639 = two balanced digits' largest)
    return sh;
}

// Hessian weights of the active rows as limb planes over the compact index, in the sample order of the bit images
// (vq_pos within each 64): RISE / logRISE h = |V|; RPLE h = 2a(1 - a/(2w)), a = |V|
__global__ __launch_bounds__(256) void k_make_hw(const int8_t *__restrict__ Vq, const unsigned *__restrict__ Sb,
                                                 const double *__restrict__ w, const double *__restrict__ tau,
                                                 const unsigned *__restrict__ mmax /* by slot: largest |V| of the planes, in their unit */,
                                                 const int *__restrict__ rowcol /* row -> node */,
                                                 const int *__restrict__ vslot /* row -> slot of its V planes */,
                                                 const int *__restrict__ mt /* rows with mt[r] = 0 are skipped */, int64_t Kp,
                                                 int64_t Hpitch, int64_t kstride, int64_t Kh, int form, int8_t *__restrict__ Hq,
                                                 long long *__restrict__ hS, int vlbt, int vpl0, double vscale) {
    // A thread owns 4 consecutive bytes of a 64-sample row piece: the V image and the weight planes share the byte order
    // vq_pos() within a piece, and 4 consecutive positions are 4 consecutive samples, so the four limbs of V come in as four
    // dwords and the two of the weight leave as two.
    const int r = blockIdx.y;
    if (mt[r] == 0) return;
    const int u = rowcol[r], vs = vslot[r];
    const int sh = hw_shift(mmax[vs], form);
    const unsigned dmask = (1u << sh) - 1u;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; // (piece, dword)
    const int64_t jc = (t >> 4) * 64;                            // compact index of the piece's first sample
    const int p4 = (int)(t & 15) * 4;                            // byte position within the piece
    const int tile = r >> 5, rl = r & 31;
    long long sm = 0;
    if (jc < Kh) {
        const int64_t kc = (jc >> 9) * kstride * 512 + (jc & 511); // the configuration the piece starts at
        // samples of the positions p4 .. p4 + 3: s = s0 .. s0 + 3 (inverse of vq_pos)
        const int s0 = (((p4 >> 5) & 1) << 2) | (((p4 >> 2) & 3) << 3) | (((p4 >> 4) & 1) << 5);
        unsigned q[4] = {0u, 0u, 0u, 0u};
        unsigned sg = 0;
        if (kc < Kp) {
            const int8_t *vq = Vq + vq_off(vs, vpl0, kc, Kp, vlbt) + p4; // (vq_pos(0) = 0: the piece's first byte)
#pragma unroll
            for (int l = 0; l < 4; ++l) q[l] = *reinterpret_cast<const unsigned *>(vq + l * 32 * 64);
            const int64_t k0 = kc + s0;
            sg = (Sb[(int64_t)u * (Kp >> 5) + (k0 >> 5)] >> (k0 & 31)) & 15u; // s_u^k = 1 - 2 bit
        }
        unsigned dgw[HL] = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qv = (int)(int8_t)(q[0] >> (8 * e)) + 256 * ((int)(int8_t)(q[1] >> (8 * e)) + 256 * ((int)(int8_t)(q[2] >> (8 * e)) + 256 * (int)(int8_t)(q[3] >> (8 * e))));
            int mag = ((sg >> e) & 1u) ? qv : -qv; // V = -w exp(-E) s: |V| = -q s >= 0
            if (mag < 0) mag = 0; // (cannot happen: the top four planes of a 6-plane image are V / 65536 tau rounded to nearest, same sign or 0)
            if (form == 2) {
                const double tt = tau[vs] * vscale, a = (double)mag * tt, wk = kc < Kp ? w[kc + s0 + e] : 0.0;
                const double hv = wk > 0 ? rint(2.0 * a * (1.0 - a / (2.0 * wk)) / tt) : 0.0;
                mag = hv > 0.0 ? (hv < 4294967040.0 ? (int)(unsigned)hv : -1) : 0; // (as unsigned below: up to 2 |V| < 2^32)
            }
            // 15 bits below the row's largest, unbiased: a fixed function of (node, sample) in [0, 2^sh) is added before the shift
            const unsigned dth = ((((unsigned)(kc + s0 + e) * 0x9E3779B1u) ^ ((unsigned)u * 0x85EBCA6Bu)) >> 9) & dmask;
            unsigned long long hq = ((unsigned long long)(unsigned)mag + dth) >> sh;
            unsigned h2 = hq > 32639ull ? 32639u : (unsigned)hq; // (balanced digits: the high one must stay <= 127)
            sm += h2;
            const int lo = (int)((h2 + 128u) & 255u) - 128, hi = ((int)h2 - lo) >> 8;
            dgw[0] |= ((unsigned)lo & 0xffu) << (8 * e);
            dgw[1] |= ((unsigned)hi & 0xffu) << (8 * e);
        }
        int8_t *hq = Hq + ((int64_t)tile * (HL * 32) + rl) * Hpitch + jc + p4;
#pragma unroll
        for (int l = 0; l < HL; ++l) *reinterpret_cast<unsigned *>(hq + (int64_t)l * 32 * Hpitch) = dgw[l];
    }
    // S = sum of the weights
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    __shared__ long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&hS[r]), (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}

// Row-major twin of Xtb for the gathered-row DMA of the Hessian kernel: Mb [Qp][Kp/64][2 h] dwords, same dword format
// and sample order as Xtb (rows at and beyond Qfp -- the constant column and the padding -- hold zero bits: x = +1).
__global__ __launch_bounds__(256) void k_build_mb(const unsigned *__restrict__ Xtb, int64_t nkk, unsigned *__restrict__ Mb) {
    const int64_t kt = (int64_t)blockIdx.y * 256 + threadIdx.x, c = blockIdx.x;
    if (kt >= nkk) return;
    const uint2 v = *reinterpret_cast<const uint2 *>(Xtb + ((((c >> 7) * nkk + kt) * 128) + (c & 127)) * 2);
    *reinterpret_cast<uint2 *>(Mb + (c * nkk + kt) * 2) = v;
}

constexpr int kHessSmall = 4; // working sets of up to this many 32-entry tiles take the one-workgroup kernel, larger ones the 2 x 4 blocks

// Working sets of up to 128 entries (and the 128-entry preconditioner tiles of the matrix-free rows): ONE workgroup holds the whole
// lower triangle of a row's block -- up to ten 32 x 32 tile products -- over one chunk of the compact index.  Wave (l, t) multiplies
// weight limb l over K-half t of every 64-sample step.  Per group of 8 steps (512 samples) the workgroup DMAs the 128 gathered rows
// x 64 B of bits and HL x 512 B of weight limbs into a 3-stage ring; inside a group there is no barrier: every lane reads the dword
// of bits of its entry of each tile (ds_read_b32), expands the B fragment in registers (shift + and) and derives the A fragment from
// it -- the 0x00 / 0xFF mask of the same bits, (f << 8) - f, and the weight digits: A and B share lane <-> entry.  At the end the
// four waves' partial sums meet in LDS and leave as ONE int64 atomic per entry.
// (Rounds 1-4: 2 x 2 tile blocks, three workgroups per 4-tile row each expanding its operands cooperatively through LDS with a barrier
// per step, every wave adding its own limb: at 128 rows x 32 768 configurations 233 us, of which 136 the skeleton without MFMAs and
// atomics and 86 the atomics -- profiles/r5_ab_hess_ablation.txt.)
__global__ __launch_bounds__(256, 2) void k_hess_bits_small(const unsigned *__restrict__ Mb, const int8_t *__restrict__ Hq,
                                                            const int *__restrict__ F, const int *__restrict__ mt,
                                                            const long long *__restrict__ hoff, int cap, int64_t Kh, int64_t Kp,
                                                            int64_t Hpitch, int64_t kchunk /* multiple of 512 */, int64_t kstride,
                                                            long long *__restrict__ H64, int z0, int R0, const int *__restrict__ tF,
                                                            const int *__restrict__ trow, int tT) {
    constexpr int NR = 128, STAGE = NR * 64 + HL * 512, NSG = 3;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int r = blockIdx.z + z0; // block: a row's working set (r < R0), or tile r - R0 of the matrix-free rows' preconditioner
    const int m = mt[r];
    if (m == 0 || m > kHessSmall) return;
    const int64_t kb = (int64_t)blockIdx.x * kchunk; // compact index
    if (kb >= Kh) return;
    const int64_t ke = (kb + kchunk < Kh) ? kb + kchunk : Kh;
    const int ngrp = (int)((ke - kb + 511) / 512);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, h = lane >> 5;
    const int wl = wave >> 1, wt = wave & 1; // this wave's weight limb and K-half
    const int wr = r < R0 ? r : trow[r - R0]; // the row whose weights this block uses
    const int tile = wr >> 5, rl = wr & 31;
    const int *Fr = r < R0 ? F + (int64_t)r * cap : tF + (int64_t)(r - R0) * tT;
    const int mrows = m * 32;
    const int64_t nkk = Kp >> 6;

    // DMA sources: row pieces wave and wave + 4 (16 rows x 64 B each; per group + 64 kstride B: 8 steps x 8 B of every kstride-th
    // block), and (wave 0) the weight piece (HL x 512 B; per group + 512 B, compact)
    const int8_t *src[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + 4 * j) * 16 + (lane >> 2);
        const int fr = row < mrows ? row : 0;
        const int slot = (lane & 3) ^ ((row >> 2) & 3); // swizzle on the source (LDS side is linear)
        src[j] = reinterpret_cast<const int8_t *>(Mb) + ((int64_t)Fr[fr] * nkk + (kb >> 9) * kstride * 8) * 8 + slot * 16;
    }
    src[2] = Hq + ((int64_t)tile * (HL * 32) + (lane >> 5) * 32 + rl) * Hpitch + kb + (lane & 31) * 16;
    const int64_t adv = 64 * kstride;
    auto issue = [&](int g) {
        int8_t *sb = lds + (g % NSG) * STAGE;
        __builtin_amdgcn_global_load_lds((gptr_t)(src[0] + (int64_t)g * adv), (lptr_t)(sb + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src[1] + (int64_t)g * adv), (lptr_t)(sb + (wave + 4) * 1024), 16, 0, 0);
        if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(src[2] + (int64_t)g * 512), (lptr_t)(sb + 8 * 1024), 16, 0, 0);
    };
    v16i acc[10]; // tile pair (i, j <= i) at i (i + 1) / 2 + j
#pragma unroll
    for (int q = 0; q < 10; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0;

    const int sw = (lr >> 2) & 3;
    const int boff = lr * 64 + (h << 2), moff = NR * 64 + wl * 512 + (2 * wt + h) * 16;
    issue(0);
    if (ngrp > 1) issue(1);
    for (int g = 0; g < ngrp; ++g) {
        // this wave's pieces of stage g have landed (stage g + 1 may still be in flight), then every wave's
        if (g + 1 < ngrp) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < ngrp) issue(g + 2);
        const int8_t *st = lds + (g % NSG) * STAGE;
        const int left = (int)((ke - kb - (int64_t)g * 512 + 63) / 64);
        const int nsteps = left < 8 ? left : 8;
        for (int ks = 0; ks < nsteps; ++ks) {
            // logical 16-byte slot ks >> 1 of the row, dword (ks & 1) * 2 + h: the lane's 32 samples of the step, 16 per K-half
            const int so = ((((ks >> 1) ^ sw)) << 4) + ((ks & 1) << 3);
            const v4i mg = *reinterpret_cast<const v4i *>(st + moff + ks * 64);
            v4i fa[4], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < m) {
                    const unsigned vB = *reinterpret_cast<const unsigned *>(st + j * 2048 + boff + so);
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const unsigned f = (vB >> (4 * wt + dd)) & 0x01010101u;
                        fb[j][dd] = (int)f;
                        fa[j][dd] = (int)(((f << 8) - f) & (unsigned)mg[dd]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < m) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) acc[i * (i + 1) / 2 + j] = MFMA_I8(fa[i], fb[j], acc[i * (i + 1) / 2 + j]);
                }
            }
        }
    }
    // the four waves' sums of a tile pair meet in LDS (the ring is free now): value = (t0 + t1 of limb 0) + 256 (t0 + t1 of limb 1)
    long long *Hr = H64 + hoff[r];
    const int hp = 32 * m;
    int *red = reinterpret_cast<int *>(lds); // [wave][e][lane]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < m) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[i * (i + 1) / 2 + j][e];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = tid + 256 * q, e = idx >> 6, ln = idx & 63;
                    const long long v = ((long long)red[(0 * 16 + e) * 64 + ln] + (long long)red[(1 * 16 + e) * 64 + ln]) +
                                        256ll * ((long long)red[(2 * 16 + e) * 64 + ln] + (long long)red[(3 * 16 + e) * 64 + ln]);
                    const int ii = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), jj = j * 32 + (ln & 31);
                    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&Hr[(int64_t)ii * hp + jj]), (unsigned long long)v);
                }
            }
        }
    }
}

// Blocked kernel for the larger working sets (5 .. 16 tiles): a workgroup computes the tile block (rows 2a, 2a+1) x (columns BT b ..
// BT b + BT - 1) of one row's working-set matrix (needed iff BT b <= 2a + 1: lower triangle) over one chunk of the compact index.
// Per group of 8 steps (512 samples) it DMAs the 64 + 32 BT gathered rows x 64 B of bits and HL x 512 B of weight limbs into a
// 3-stage ring.  Per step the four waves first expand the operands cooperatively into LDS, in MFMA fragment layout --
// wave w expands B tile w (0/1 bytes, both K-halves; w < BT) and A-mask fragment (i = w >> 1, t = w & 1) (0x00/0xFF
// bytes) -- then wave (l, t) = (w >> 1, w & 1) runs the 2 x BT tile block for weight limb l over K-half t:
//   acc[i][j] += (mask_i & h_l) * b_j  =  sum_k h_lk b_ik b_jk        (one barrier per step)
template <int BT>
__global__ __launch_bounds__(256, 2) void k_hess_bits_blk(const unsigned *__restrict__ Mb, const int8_t *__restrict__ Hq,
                                                          const int *__restrict__ F, const int *__restrict__ mt,
                                                          const long long *__restrict__ hoff, int cap, int64_t Kh,
                                                          int64_t Kp, int64_t Hpitch, int64_t kchunk /* multiple of 512 */,
                                                          int64_t kstride, long long *__restrict__ H64, int z0, int R0,
                                                          const int *__restrict__ tF, const int *__restrict__ trow, int tT) {
    constexpr int AR = 64, BR = 32 * BT, RP = (AR + BR) / 16; // row pieces
    constexpr int STAGE = (AR + BR) * 64 + HL * 512, NPIECE = STAGE / 1024, NSG = 3;
    constexpr int NPJ = (NPIECE + 3) / 4;                     // pieces of the waves that carry one more
    constexpr int EBUF = (4 + 2 * BT) * 1024;                 // expanded operands of one step
    static_assert(NPIECE == RP + 1, "one weight piece: HL x 512 B");
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *eb = lds + NSG * STAGE;
    const int r = blockIdx.z + z0; // block: a row's working set (r < R0), or tile r - R0 of the matrix-free rows' preconditioner
    const int m = mt[r];
    if (m <= kHessSmall) return; // (the small working sets: k_hess_bits_small)
    // decode the block index: a = tile-row pair, b = group of BT tile columns, needed iff BT b <= 2a+1
    int a = 0, b = blockIdx.y;
    for (;;) {
        const int nb = (2 * a + 1) / BT + 1;
        if (b < nb) break;
        b -= nb;
        ++a;
        if (2 * a >= m) return;
    }
    if (2 * a >= m) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, h = lane >> 5;
    const int wl = wave >> 1, wt = wave & 1; // this wave's weight limb and K-half
    const int64_t kb = (int64_t)blockIdx.x * kchunk; // compact index
    if (kb >= Kh) return;
    const int64_t ke = (kb + kchunk < Kh) ? kb + kchunk : Kh;
    const int ngrp = (int)((ke - kb + 511) / 512);
    const int wr = r < R0 ? r : trow[r - R0]; // the row whose weights this block uses
    const int tile = wr >> 5, rl = wr & 31;
    const int *Fr = r < R0 ? F + (int64_t)r * cap : tF + (int64_t)(r - R0) * tT;
    const int mrows = m * 32;
    const int64_t nkk = Kp >> 6;

    // DMA sources of this wave's pieces (wave + 4 j); per group: + 64 kstride B (bits: 8 steps x 8 B of every
    // kstride-th block) resp. + 512 B (limb bytes, compact)
    const bool extra = wave < NPIECE - 4 * (NPJ - 1);
    const int8_t *src[NPJ];
    int64_t adv[NPJ];
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int pc = wave + 4 * j;
        if (pc < RP) {
            const int row = pc * 16 + (lane >> 2); // A rows then B rows
            int fr = row < AR ? 2 * a * 32 + row : BT * b * 32 + (row - AR);
            if (fr >= mrows) fr = 0;
            const int slot = (lane & 3) ^ ((row >> 2) & 3); // swizzle on the source (LDS side is linear)
            src[j] = reinterpret_cast<const int8_t *>(Mb) + ((int64_t)Fr[fr] * nkk + (kb >> 9) * kstride * 8) * 8 + slot * 16;
            adv[j] = 64 * kstride;
        } else {
            src[j] = Hq + ((int64_t)tile * (HL * 32) + (lane >> 5) * 32 + rl) * Hpitch + kb + (lane & 31) * 16;
            adv[j] = 512;
        }
    }
    auto issue = [&](int g) {
        int8_t *sb = lds + (g % NSG) * STAGE;
#pragma unroll
        for (int j = 0; j < NPJ - 1; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)g * adv[j]), (lptr_t)(sb + (wave + 4 * j) * 1024), 16, 0, 0);
        if (extra)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[NPJ - 1] + (int64_t)g * adv[NPJ - 1]), (lptr_t)(sb + (wave + 4 * (NPJ - 1)) * 1024), 16, 0, 0);
    };
    v16i acc[2][BT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < BT; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0;

    // LDS offsets (stage-relative) of the dwords this lane expands: its row of B tile `wave` and of A tile wave >> 1
    const int rowB = AR + (wave < BT ? wave : 0) * 32 + lr, rowA = (wave >> 1) * 32 + lr;
    const int swB = (rowB >> 2) & 3, swA = (rowA >> 2) & 3;
    issue(0);
    if (ngrp > 1) issue(1);
    for (int g = 0; g < ngrp; ++g) {
        // this wave's pieces of stage g have landed (stage g + 1 may still be in flight), then every wave's
        if (g + 1 < ngrp) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPJ - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < ngrp) issue(g + 2);
        const int8_t *st = lds + (g % NSG) * STAGE;
        const int left = (int)((ke - kb - (int64_t)g * 512 + 63) / 64);
        const int nsteps = left < 8 ? left : 8;
        for (int ks = 0; ks < nsteps; ++ks) {
            int8_t *e = eb + ((g * 8 + ks) & 1) * EBUF;
            // cooperative expansion of step ks: logical 16-byte slot ks >> 1 of the row, dword (ks & 1) * 2 + h
            {
                const unsigned vA = *reinterpret_cast<const unsigned *>(st + rowA * 64 + ((((ks >> 1) ^ swA)) << 4) + (((ks & 1) * 2 + h) << 2));
                v4i fm;
#pragma unroll
                for (int d = 0; d < 4; ++d) fm[d] = (int)(((vA >> (4 * (wave & 1) + d)) & 0x01010101u) * 0xFFu);
                *reinterpret_cast<v4i *>(e + wave * 1024 + lane * 16) = fm; // fragment (i = wave >> 1, t = wave & 1)
                if (wave < BT) {
                    const unsigned vB = *reinterpret_cast<const unsigned *>(st + rowB * 64 + ((((ks >> 1) ^ swB)) << 4) + (((ks & 1) * 2 + h) << 2));
                    v4i f0, f1;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        f0[d] = (int)((vB >> d) & 0x01010101u);
                        f1[d] = (int)((vB >> (4 + d)) & 0x01010101u);
                    }
                    *reinterpret_cast<v4i *>(e + 4096 + (wave * 2 + 0) * 1024 + lane * 16) = f0;
                    *reinterpret_cast<v4i *>(e + 4096 + (wave * 2 + 1) * 1024 + lane * 16) = f1;
                }
            }
            __syncthreads();
            {
                const v4i mg = *reinterpret_cast<const v4i *>(st + (AR + BR) * 64 + wl * 512 + ks * 64 + (2 * wt + h) * 16);
                v4i fa[2], fb[BT];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const v4i *>(e + (i * 2 + wt) * 1024 + lane * 16) & mg;
#pragma unroll
                for (int jn = 0; jn < BT; ++jn) fb[jn] = *reinterpret_cast<const v4i *>(e + 4096 + (jn * 2 + wt) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < BT; ++jn) acc[i][jn] = MFMA_I8(fa[i], fb[jn], acc[i][jn]);
            }
        }
    }
    long long *Hr = H64 + hoff[r];
    const int hp = 32 * m;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < BT; ++jn) {
            const int ti = 2 * a + i, tj = BT * b + jn;
            if (ti < m && tj <= ti) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ii = ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, jj = tj * 32 + lr;
                    const long long v = ((long long)acc[i][jn][e]) * (1ll << (8 * wl));
                    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&Hr[(int64_t)ii * hp + jj]), (unsigned long long)v);
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_hess_i8_fin(const long long *__restrict__ H64, const long long *__restrict__ hS,
                                                     const double *__restrict__ tau, const unsigned *__restrict__ mmax, int form,
                                                     const int *__restrict__ vslot, const int *__restrict__ mt,
                                                     const long long *__restrict__ hoff, double *__restrict__ H, int y0, int R0,
                                                     const int *__restrict__ trow, double vscale) {
    const int r = blockIdx.y + y0;
    const int wr = r < R0 ? r : trow[r - R0];
    const int m = mt[r] * 32;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * m) return;
    const int i = idx / m, j = idx % m;
    if ((j >> 5) > (i >> 5)) return;
    const long long *Hr = H64 + hoff[r];
    const long long T = Hr[(int64_t)i * m + j], Ti = Hr[(int64_t)i * m + i], Tj = Hr[(int64_t)j * m + j];
    const int vs = vslot[wr];
    H[hoff[r] + (int64_t)i * m + j] = ldexp(tau[vs] * vscale, hw_shift(mmax[vs], form)) * (double)(hS[wr] - 2 * Ti - 2 * Tj + 4 * T);
}

// Largest number of configurations one int8 Hessian call can use (pitch of its weight planes): all of them.
int64_t i8_hess_kmax(const DevProblem &d) { return d.Kp; }

// k-split of a Hessian launch: chunks of whole 512-sample groups, `target` workgroups in all
static void hess_split(int64_t Kh, int64_t wg, int target, int *ns_out, int64_t *kc_out) {
    const int maxsplit = (int)(Kh / 1024) > 0 ? (int)(Kh / 1024) : 1;
    int ns = (int)((target + wg - 1) / wg);
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    int64_t kc = (Kh + ns - 1) / ns;
    kc = (kc + 511) / 512 * 512;
    if (kc > ((int64_t)1 << 23)) kc = (int64_t)1 << 23; // i32 sums per workgroup: |sum| <= 128 kc
    *ns_out = (int)((Kh + kc - 1) / kc);
    *kc_out = kc;
}

// the small working sets (and small preconditioner tiles): one workgroup per (block, chunk)
static void launch_hess_small(const I8Ws *w, const DevProblem &d, const int *dF, const int *dMt, const long long *dHoff, int R, int cap, int64_t Kh,
                              int64_t kstride, int64_t nblocks_active, hipStream_t st, const HessTiles &tl) {
    // (counted on the blocks that exist: late in a solve a handful of rows remain, each with all K configurations)
    int ns = 1;
    int64_t kc = 0;
    hess_split(Kh, nblocks_active > 0 ? nblocks_active : 1, g_tune[GML_TUNE_HESS_WGS] > 0 ? (int)g_tune[GML_TUNE_HESS_WGS] : 1024, &ns, &kc);
    constexpr int shmem = 3 * (128 * 64 + HL * 512);
    const int64_t nv = R + tl.n;
    for (int64_t z0 = 0; z0 < nv; z0 += 8192)
        hipLaunchKernelGGL(k_hess_bits_small, dim3((unsigned)ns, 1u, (unsigned)std::min<int64_t>(8192, nv - z0)), dim3(256), shmem, st, w->Mb, w->Hq, dF,
                           dMt, dHoff, cap, Kh, d.Kp, w->hKh, kc, kstride, w->H64, (int)z0, R, tl.F, tl.wrow, tl.T);
}

template <int BT>
static void launch_hess_blk(const I8Ws *w, const DevProblem &d, const int *dF, const int *dMt, const long long *dHoff, int R, int cap,
                            int maxm, int64_t Kh, int64_t kstride, int64_t nrows_active /* blocks of this size class */, hipStream_t st,
                            const HessTiles &tl) {
    // blocks (a, b) with BT b <= 2a + 1 for a < ceil(maxm / 2)
    int nblk = 0;
    for (int a = 0; 2 * a < maxm; ++a) nblk += (2 * a + 1) / BT + 1;
    // k-split so that the grid fills the chip (~4096 workgroups), in chunks of whole 512-sample groups
    // (counted on the rows that have a working set: late in a solve a handful of rows remain, each with all K configurations,
    // and sized on R they would get a few long workgroups each)
    int ns = 1;
    int64_t kc = 0;
    hess_split(Kh, (int64_t)(nrows_active > 0 ? nrows_active : 1) * nblk, 4096, &ns, &kc);
    constexpr int shmem = 3 * ((64 + 32 * BT) * 64 + HL * 512) + 2 * (4 + 2 * BT) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hess_bits_blk<BT>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    const int64_t nv = R + tl.n;
    // (grid z is limited to 65 535; slices of 8 192 blocks so that the slicing is exercised by config 5's 28 k tiles, not only
    // by problems ten times its size)
    for (int64_t z0 = 0; z0 < nv; z0 += 8192)
        hipLaunchKernelGGL(k_hess_bits_blk<BT>, dim3((unsigned)ns, (unsigned)nblk, (unsigned)std::min<int64_t>(8192, nv - z0)), dim3(256), shmem,
                           st, w->Mb, w->Hq, dF, dMt, dHoff, cap, Kh, d.Kp, w->hKh, kc, kstride, w->H64, (int)z0, R, tl.F, tl.wrow, tl.T);
}

// Working-set Hessians of the rows 0..R-1 of the caller's arrays (mt[r] = 0: skip) from the int8 limb planes their
// last objective passes left in the slots vslot[r], over Kh configurations (a multiple of 512; block cb of the
// compact index = samples [512 cb kstride, +512)).  Returns GML_EUNSUPPORTED when a working set exceeds 512 entries
// (the solver handles larger ones matrix-free).
int i8_hessian(void *wsp, const DevProblem &d, const int *dRowcol /* row -> node */, const int *dVslot /* row -> slot */, const int *dF,
               const int *dMt, const int *hMt, const long long *dHoff, int64_t htotal, int R, int cap, int form, int64_t Kh,
               int64_t kstride, double *dH, hipStream_t st, std::string *err, const HessTiles *tiles) {
    const HessTiles tl = tiles ? *tiles : HessTiles{};
    const int *dFlag = tl.n > 0 ? tl.hflag : dMt; // rows whose weights are needed
    I8Ws *w = static_cast<I8Ws *>(wsp);
    if (!w) {
        if (err) *err = "no int8 pass has run on this handle";
        return GML_EINVAL;
    }
    int maxm = 0, maxsmall = 0;
    int64_t nsmall = 0, nlarge = 0;
    for (int r = 0; r < R; ++r) {
        maxm = hMt[r] > maxm ? hMt[r] : maxm;
        if (hMt[r] <= kHessSmall) maxsmall = hMt[r] > maxsmall ? hMt[r] : maxsmall;
        if (hMt[r] > kHessSmall) ++nlarge;
        else if (hMt[r] > 0) ++nsmall;
    }
    if (maxm > 16) return GML_EUNSUPPORTED;
    const bool tiles_small = tl.T / 32 <= kHessSmall;
    if (tl.n > 0) { // the tiles: tl.n blocks of T / 32 <= 4 tiles each
        maxm = std::max(maxm, tl.T / 32);
        if (tiles_small) {
            maxsmall = std::max(maxsmall, tl.T / 32);
            nsmall += tl.n;
        } else {
            nlarge += tl.n;
        }
    }
    const int64_t pitch = d.Kp, Rp = (R + 31) / 32 * 32;
    if (Kh > pitch) Kh = pitch;
    if (w->hKh != pitch || w->hrows < Rp) {
        if (w->Hq) (void)dev_free(w->Hq);
        w->Hq = nullptr;
        I8CHK(dev_malloc(&w->Hq, (size_t)Rp * HL * pitch));
        w->hKh = pitch;
        w->hrows = Rp;
    }
    if (!w->Mb) {
        I8CHK(dev_malloc(&w->Mb, (size_t)d.Qp * (d.Kp / 8)));
        I8CHK(hipMemsetAsync(w->Mb, 0, (size_t)d.Qp * (d.Kp / 8), st));
        hipLaunchKernelGGL(k_build_mb, dim3((unsigned)d.Qfp, (unsigned)((d.Kp / 64 + 255) / 256)), dim3(256), 0, st, d.Xtb, d.Kp / 64, w->Mb);
    }
    const int64_t need = htotal + Rp; // the integer blocks, then the rows' weight sums: cleared by one fill
    if (need > w->hcap_elems) {
        if (w->H64) (void)dev_free(w->H64);
        w->H64 = nullptr;
        I8CHK(dev_malloc(&w->H64, sizeof(long long) * need));
        w->hcap_elems = need;
    }
    I8CHK(hipMemsetAsync(w->H64, 0, sizeof(long long) * need, st));
    long long *hS = w->H64 + htotal;
    hipLaunchKernelGGL(k_make_hw, dim3((unsigned)((Kh / 4 + 255) / 256), (unsigned)R), dim3(256), 0, st, w->Vq, d.Sb, d.w, w->sc[0].tau, w->sc[0].mmax,
                       dRowcol, dVslot, dFlag, d.Kp, pitch, kstride, Kh, form, w->Hq, hS, w->LBT, w->vpl0(), w->vscale());
    HessTiles rows_only = tl; // (the tiles are all of one size class: the other launch covers the rows' own blocks only)
    rows_only.n = 0;
    if (maxsmall > 0) launch_hess_small(w, d, dF, dMt, dHoff, R, cap, Kh, kstride, nsmall, st, tiles_small ? tl : rows_only);
    if (maxm > kHessSmall) launch_hess_blk<4>(w, d, dF, dMt, dHoff, R, cap, maxm, Kh, kstride, nlarge, st, tiles_small ? rows_only : tl);
    hipLaunchKernelGGL(k_hess_i8_fin, dim3((unsigned)((maxm * 32 * maxm * 32 + 255) / 256), (unsigned)R), dim3(256), 0, st, w->H64, hS,
                       w->sc[0].tau, w->sc[0].mmax, form, dVslot, dMt, dHoff, dH, 0, R, tl.wrow, w->vscale());
    const int tm = tl.T / 32;
    for (int64_t y0 = 0; y0 < tl.n; y0 += 8192)
        hipLaunchKernelGGL(k_hess_i8_fin, dim3((unsigned)((tm * 32 * tm * 32 + 255) / 256), (unsigned)std::min<int64_t>(8192, tl.n - y0)), dim3(256),
                           0, st, w->H64, hS, w->sc[0].tau, w->sc[0].mmax, form, dVslot, dMt, dHoff, dH, (int)(R + y0), R, tl.wrow, w->vscale());
    I8CHK(hipGetLastError());
    return GML_OK;
}


} // namespace gml
