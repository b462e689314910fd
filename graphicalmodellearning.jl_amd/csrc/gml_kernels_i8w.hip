// FP64-grade fixed-point pass of the learn() hot path on the int8 matrix cores (gfx950, v_mfma_i32_32x32x32_i8): precision
// "i8w".  Same idea as the i8x pass (gml_i8.h) -- the statistics are +-1 (GraphicalModelLearning.jl:162, :107), so the two
// contractions of the objective/gradient pass (:196, :205-207) are exact integer GEMMs once the real operand is written in
// balanced base-256 digits -- carried to the width of the reference's Float64 arithmetic:
//   Theta[r][c] = sigma_r * q,  q an integer of 54 bits in 7 digit planes (sigma_r a power of two: the entries within a factor
//                 two of the row's largest are represented exactly, the others to 2^-55 of it -- the rounding a Float64
//                 accumulation of E = sum_c theta_c x_c commits on every term);
//   V[r][k]     = tau_r * v,    v an integer of 47 bits in 6 digit planes, rounded with a dither (f and the gradient then carry
//                 ~0.4 sqrt(K) 2^-47 of the largest weight: 1e-15 relative at K = 1e6, the order of Float64 summation error);
//   exp         in FP64 (Cody-Waite reduction, 2^(j/64) table, degree-5 polynomial: 1e-16 relative).
// Results are deterministic and independent of tiling, split-K order and GPU count, like those of the i8x pass.
//
// Register budget.  7 planes x 2 sample tiles x 16 accumulators do not fit the 256 registers of a wave at two workgroups per
// CU, so the forward kernel sweeps the columns TWICE inside one workgroup: sweep A multiplies the planes 0..3 (128
// accumulators), folds them into 32 FP64 partial energies per lane (64 registers), sweep B multiplies the planes 4..6 (96
// accumulators).  The LDS-DMA ring runs through both sweeps without a restart; the sample bits are loaded twice (2 KB per step
// against 8 / 6 KB of digit planes).  V is kept as two halves of 3 planes, V / tau = lo + 2^24 hi: the backward GEMM runs as two
// launches of the 3-plane form of k_bwd_i8 and every integer sum (per half) stays far inside 64 bits.
#include "gml_i8.h"
#include <string>
#include <type_traits>

namespace gml {

namespace {

constexpr int LFA = 4, LFB = 3;          // digit planes of Theta per sweep
constexpr int BRT = 32 * LFW;            // rows of a Tq image
constexpr int PA = 2 + 2 * LFA, PB = 2 + 2 * LFB; // 1-KB pieces of a 64-column step: bits + digit-plane rows
constexpr int STEPW = PA * 1024, DSW = 2, STAGEW = DSW * STEPW, NSW = 3, RINGW = NSW * STAGEW;
constexpr int NLA = DSW * PA / 4, NLB = DSW * PB / 4; // DMA instructions per wave and stage: 5, 4
static_assert(NLA == NLB + 1, "issue() drops the last load in sweep B");

__device__ __forceinline__ double flip_if(double v, int mneg /* 0 or -1 */) {
    return __hiloint2double(__double2hiint(v) + (mneg << 31), __double2loint(v));
}

// 6 balanced base-256 digits of the 48-bit two's complement integer held in the mantissa of yr = v + 1.5 * 2^52 (or the scaled
// form): (v + C) ^ C; dl = digits 0..3, the low half of dh = digits 4, 5
__device__ __forceinline__ void digits6(double yr, unsigned &dl, unsigned &dh) {
    unsigned long long v = ((unsigned long long)(unsigned)__double2hiint(yr) << 32) | (unsigned)__double2loint(yr);
    v += 0x0000808080808080ull;
    v ^= 0x0000808080808080ull;
    dl = (unsigned)v;
    dh = (unsigned)(v >> 32);
}

// One LDS-DMA instruction with the address split the way the hardware takes it: a wave-uniform 64-bit base in scalar registers,
// a 32-bit per-lane offset, the (wave-uniform) LDS destination in M0.  (Through __builtin_amdgcn_global_load_lds the compiler
// folds the lane offset into loop-invariant 64-bit VECTOR pointers, one pair per instruction of the stage: 18 registers this
// kernel does not have.)
__device__ __forceinline__ void dma16(const int8_t *ubase, int voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(ubase), "s"(lds_addr) : "memory"); // (M0 is reserved: the compiler never keeps a value in it)
}

} // namespace

// ------------------------------------------------------------------------------------------
// forward: energies by two sweeps of C[k][m] = sum_c b[k][c] * Tq[m][c] (b = [x = -1] from the bit image), then the pointwise
// epilogue   E = s sigma (C0 - 2 sum_l 256^l C_l),  V = -w exp(-E) s  (RISE / logRISE),  -2 w s / (1 + exp(2E))  (RPLE),
// V -> 6 balanced digits -> the planes of the wave's Vq image.  Workgroup = 4 waves along the samples: 256 samples x one
// 32-node tile; block mapping, ring and fragment handling as in k_fwd_i8.
// ------------------------------------------------------------------------------------------
template <int FORM /* 0: exp forms (RISE, logRISE), 2: RPLE */, bool WANTF, bool WIDE /* more than 32768 statistics columns */, bool UNIW,
          bool COARSE /* sweep A only: Theta from its top four planes (30 bits), V in three planes (dithered 23 bits: planes 3..5, plane 2
                         zero) -- the cheap form of the pass for iterates far from the optimum (exp forms) */>
__global__ __launch_bounds__(256, 2) void k_fwd_i8w(
    const unsigned *__restrict__ Xb, const unsigned *__restrict__ Sb, const int8_t *__restrict__ Tq, const int *__restrict__ rowcol,
    const int *__restrict__ groups, int ngroups, const double *__restrict__ w, const double *__restrict__ sigma,
    const long long *__restrict__ qconst, const long long *__restrict__ qconst2, const double *__restrict__ invtau, int64_t Kp, int ntiles_k,
    int nk_all /* 64-column steps of a sweep over all columns (0: every row of Theta is zero) */,
    double wuni, int64_t Kreal, int8_t *__restrict__ Vq, long long *__restrict__ csum, long long *__restrict__ csum2,
    long long *__restrict__ asum, long long *__restrict__ asum2, double *__restrict__ fsum, unsigned *__restrict__ mmax,
    // column compaction (gml_i8_pack.hip: k_col_union): steps of each tile's compact image (-1: all columns), the images, bytes per tile,
    // and the steps between two tiles' Tq images (= Qfp / 64 whatever is swept)
    const int *__restrict__ cnk, const int8_t *__restrict__ Xc, int64_t xc_tile, int nk_tq) {
    constexpr int WM = 2; // 32-sample MFMA tiles per wave
    extern __shared__ __attribute__((aligned(16))) int8_t lds[]; // ring, then the exp (and log) tables
    double *etab = reinterpret_cast<double *>(lds + RINGW);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    if (tid < 64) {
        const double v = exp2((double)tid / 64.0);
        // exp forms: 2^(j/64) with j << 14 taken off the high word (the exponent of 2^(n >> 6), n = 64 q + j, goes on as n << 14)
        etab[tid] = FORM == 0 ? __hiloint2double(__double2hiint(v) - (tid << 14), __double2loint(v)) : v;
    }
    if (FORM == 2 && tid < 64) { // log table for RPLE: c_j = 1 + (j + 1/2)/64 -> 1/c_j, log c_j
        const double cj = 1.0 + ((double)tid + 0.5) / 64.0;
        etab[64 + tid] = 1.0 / cj;
        etab[128 + tid] = log(cj);
    }
    __syncthreads();

    // XCD-aware L2 blocking, as in k_fwd_i8: XCD x owns the sample tiles st = 8 i + x and sweeps them inside groups of TG node tiles
    constexpr int TG = 8;
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int ntk8 = (ntiles_k + 7) >> 3;
    const int nfull = ngroups / TG, per_full = ntk8 * TG;
    int st, gi;
    if (bi < nfull * per_full) {
        const int rem = bi % per_full;
        st = (rem / TG) * 8 + xcd;
        gi = (bi / per_full) * TG + rem % TG;
    } else {
        const int lastn = ngroups - nfull * TG, rem = bi - nfull * per_full;
        st = (rem / lastn) * 8 + xcd;
        gi = nfull * TG + rem % lastn;
    }
    if (st >= ntiles_k) return;
    const int64_t k0 = (int64_t)st * 256;
    if (k0 >= Kp) return;
    const int mytile = groups[gi];
    // the columns this tile sweeps: all of them, or its compact list (the image then has the tile's own step count in its strides)
    int nk = nk_all;
    const int8_t *xbase = reinterpret_cast<const int8_t *>(Xb);
    if (cnk) {
        const int ck = cnk[mytile];
        if (ck >= 0) {
            nk = ck;
            xbase = Xc + (int64_t)mytile * xc_tile;
        }
    }

    // DMA plan: a ring stage holds DSW = 2 consecutive 64-column steps of ONE sweep; its 2 PA (2 PB) 1-KB pieces are dealt to the
    // four waves, 5 (4) each.  Piece pc of a step: pc < 2 the two 128-sample pieces of bits, else 16 rows of the sweep's digit
    // planes.  Addresses = a wave-uniform part (piece, step) + one of two per-lane offsets: the XOR swizzle of lds_off() is
    // applied to the SOURCE (the LDS side of the DMA is linear) and depends on the lane only (16-row pieces: (row >> 2) & 3 =
    // (lane >> 4) & 3).
    const int voffX = lane * 16;
    const int voffT = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const int8_t *const gX = xbase + (int64_t)(2 * st) * nk * 1024;
    const int8_t *const gT = Tq + (int64_t)mytile * nk_tq * BRT * 64;
    const int nst = (nk + DSW - 1) / DSW; // ring stages per sweep; global stage gs < nst: sweep A, else sweep B
    // per DMA instruction of this wave (wave-uniform, scalar registers): source of step 0, bytes per step, step within the
    // stage, destination within the stage
    const int8_t *baseA[NLA], *baseB[NLB];
    int advA[NLA], advB[NLB], subA[NLA], subB[NLB], dstA[NLA], dstB[NLB];
    bool bitsA[NLA], bitsB[NLB];
    auto plan = [&](int sp, int pps, int row0, const int8_t *&base, int &adv, int &sub, int &dst, bool &bits) {
        sub = sp / pps;
        const int pc = sp - sub * pps;
        dst = sub * STEPW + pc * 1024;
        bits = pc < 2;
        base = bits ? gX + (int64_t)pc * nk * 1024 : gT + (row0 + (pc - 2) * 16) * 64;
        adv = bits ? 1024 : BRT * 64;
    };
#pragma unroll
    for (int j = 0; j < NLA; ++j) plan(wave + 4 * j, PA, 32 * LFB, baseA[j], advA[j], subA[j], dstA[j], bitsA[j]); // planes 3..6
#pragma unroll
    for (int j = 0; j < NLB; ++j) plan(wave + 4 * j, PB, 0, baseB[j], advB[j], subB[j], dstB[j], bitsB[j]);        // planes 0..2
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int8_t *)lds;
    auto issue = [&](int gs) {
        const unsigned stage_base = lds0 + (gs % NSW) * STAGEW;
        if (gs < nst) {
#pragma unroll
            for (int j = 0; j < NLA; ++j) {
                int kt = DSW * gs + subA[j];
                kt = kt < nk ? kt : nk - 1; // (a step beyond the last one: the last one again, so that every stage counts the same loads)
                dma16(baseA[j] + (int64_t)(kt * advA[j]), bitsA[j] ? voffX : voffT, stage_base + dstA[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NLB; ++j) {
                int kt = DSW * (gs - nst) + subB[j];
                kt = kt < nk ? kt : nk - 1;
                dma16(baseB[j] + (int64_t)(kt * advB[j]), bitsB[j] ? voffX : voffT, stage_base + dstB[j]);
            }
        }
    };

    // what the fold between the sweeps needs is fetched now, so that the latency hides under the GEMM; the inputs of the pointwise
    // arithmetic (sign bits, 1 / tau) are fetched behind sweep B -- registers are what this kernel is short of, and the
    // co-resident workgroup covers the wait
    const int r = mytile * 32 + lr;
    const int rc = rowcol[r];
    const bool active = rc >= 0;
    const double sg = active ? sigma[r] : 0.0;
    // C0 = sum_c q_c + q_const: the energy of the all-(+1) configuration / sigma.  C0 = c_lo + 2^24 c_hi with 0 <= c_lo < 2^24: both
    // halves, and everything combined with them below, are exact in FP64.  The coarse form has its own constant: the sum of the
    // numbers the top four planes spell (q / 2^24 rounded to nearest, entry by entry).
    const long long qc = active ? (COARSE ? qconst2[r] : qconst[r]) : 0;
    const double c_lo = COARSE ? 0.0 : (double)(unsigned)(qc & 0xffffffll), c_hi = COARSE ? (double)qc : (double)(qc >> 24);
    const double sgT = sg * 16777216.0, us0 = c_hi * sgT, m2sT = -2.0 * sgT;
#ifdef ABL_EARLY_INPUTS
    unsigned sgn[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) sgn[i] = active ? (Sb[(int64_t)rc * (Kp >> 5) + ((k0 + wave * 64) >> 5) + i] >> (4 * h)) : 0u;
    const double it = active ? invtau[r] : 0.0;
#endif

#ifdef ABL_TIMING
    unsigned long long tst[6];
    tst[0] = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(1);
    const int ntot = COARSE ? nst : 2 * nst; // (the ring of a coarse pass ends with sweep A)
#pragma unroll
    for (int s = 0; s < NSW - 1; ++s)
        if (s < ntot) issue(s);
    // one ring stage of GEMM work on NPL digit planes
    auto gemm_stage = [&](int gs, int ks, auto first, auto &acc) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int NPL = sizeof(acc[0]) / sizeof(acc[0][0]);
        // the next stage may stay in flight: its loads are the last ones this wave issued
        if (gs + 1 < ntot) {
            if (gs + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (gs + NSW - 1 < ntot) issue(gs + NSW - 1);
#pragma unroll
        for (int sub = 0; sub < DSW; ++sub) {
            if (sub > 0 && DSW * ks + sub >= nk) break; // (an odd number of steps: the last stage is half full)
            const int8_t *cur = lds + (gs % NSW) * STAGEW + sub * STEPW;
            unsigned vb[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = wave * 64 + i * 32 + lr;
                vb[i] = *reinterpret_cast<const unsigned *>(cur + (row >> 7) * 1024 + (((row & 127) * 2 + h) << 2));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4i fa[WM], fb[NPL];
#pragma unroll
                for (int l = 0; l < NPL; ++l) fb[l] = *reinterpret_cast<const v4i *>(cur + 2048 + lds_off(l * 32 + lr, 2 * t + h));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][e] = (int)((vb[i] >> (4 * t + e)) & 0x01010101u);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int l = 0; l < NPL; ++l) {
                        if (FIRST && sub == 0 && t == 0) acc[i][l] = MFMA_I8(fa[i], fb[l], ((v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}));
                        else acc[i][l] = MFMA_I8(fa[i], fb[l], acc[i][l]);
                    }
            }
        }
    };

    // ---- sweep A: digit planes 3..6, folded into us = sigma 2^24 (c_hi - 2 a_hi), a_hi = sum_{l>=3} 256^(l-3) C_l (exact: an integer
    // below 2^53 times a power of two)
    double us[WM][16];
    {
        v16i acc[WM][LFA];
        if (nk > 0) { // Qfp >= 64; nk = 0: every row of Theta is zero (the caller says so): all sums are 0, nothing is loaded
            gemm_stage(0, 0, std::true_type{}, acc);
            for (int ks = 1; ks < nst; ++ks) gemm_stage(ks, ks, std::false_type{}, acc);
        } else {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int l = 0; l < LFA; ++l)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;
        }
#ifdef ABL_TIMING
        tst[1] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                double ahi;
                if (WIDE) {
                    ahi = (double)acc[i][3][e];
#pragma unroll
                    for (int l = 2; l >= 0; --l) ahi = fma(ahi, 256.0, (double)acc[i][l][e]);
                } else { // |acc_l| <= 128 Qfp <= 2^22: pairs in int32
                    const int p0 = acc[i][0][e] + (acc[i][1][e] << 8), p1 = acc[i][2][e] + (acc[i][3][e] << 8);
                    ahi = fma((double)p1, 65536.0, (double)p0);
                }
                us[i][e] = fma(ahi, m2sT, us0);
                // (pinned here: left alone, the compiler sinks the whole fold behind sweep B and keeps the 128 accumulators of
                // sweep A alive under the 96 of sweep B)
                asm volatile("" : "+v"(us[i][e]));
            }
    }
#ifdef ABL_TIMING
    tst[2] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (!COARSE) {
        // ---- sweep B: digit planes 0..2; E / s = sigma (c_lo - 2 a_lo) + us with ONE rounding (c_lo - 2 a_lo is an exact integer).
        // Done for all 32 elements of the lane at once: the 96 accumulators and the 64 registers of `us` become 64 registers of
        // energies before the pointwise arithmetic starts.
        v16i acc[WM][LFB];
        if (nk > 0) {
            gemm_stage(nst, 0, std::true_type{}, acc);
#ifndef ABL_ONESWEEP
            for (int ks = 1; ks < nst; ++ks) gemm_stage(nst + ks, ks, std::false_type{}, acc);
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        } else {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int l = 0; l < LFB; ++l)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                double alo;
                if (WIDE) {
                    alo = fma((double)acc[i][2][e], 256.0, (double)acc[i][1][e]);
                    alo = fma(alo, 256.0, (double)acc[i][0][e]);
                } else {
                    alo = fma((double)acc[i][2][e], 65536.0, (double)(acc[i][0][e] + (acc[i][1][e] << 8)));
                }
                us[i][e] = fma(fma(alo, -2.0, c_lo), sg, us[i][e]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef ABL_TIMING
    tst[3] = __builtin_amdgcn_s_memrealtime();
#endif

#ifndef ABL_EARLY_INPUTS
    unsigned sgn[WM]; // the node's sign bits for this wave's 64 samples, shifted so that bit 8g + j is this lane's sample 8g + 4h + j
#pragma unroll
    for (int i = 0; i < WM; ++i) sgn[i] = active ? (Sb[(int64_t)rc * (Kp >> 5) + ((k0 + wave * 64) >> 5) + i] >> (4 * h)) : 0u;
    const double it = active ? invtau[r] : 0.0;
#endif
    const int64_t left = Kreal - (k0 + wave * 64 + 4 * h);
    const int nreal = left > 64 ? 64 : (left < 0 ? 0 : (int)left);

    // ---- epilogue ----------------------------------------------------------------------------
    // lane <-> node row (lr), register e <-> sample (e&3) + 8*(e>>2) + 4*h within the 32-sample tile; the Vq image stores a
    // step's samples in the order vq_pos() (gml_bits.h): this lane's 16 samples of tile i are 16 contiguous bytes per plane.
    int8_t *vimg = Vq + vq_off(mytile * 32 + lr, 0, k0 + wave * 64, Kp, LBW) + h * 32;
    const int64_t kw = k0 + wave * 64; // first sample of this wave
    // 2^32 w / tau; a coarse pass rounds V to multiples of 2^24 tau (the same dither, one step up)
    const double wscale = COARSE ? 256.0 : 4294967296.0; // 2^32 / 2^24
    const double wk32 = wscale * (wuni * it);
    const unsigned dh0 = (unsigned)rc * 0x85EBCA6Bu + (unsigned)(kw + 4 * h) * 0x9E3779B9u; // dither: see k_fwd_i8
    constexpr double MAGIC = 6755399441055744.0, MAGIC32 = 6755399441055744.0 * 4294967296.0;
    constexpr unsigned GOLD = 0x9E3779B9u;
    const int wleft = (int)((Kreal - kw) < 64 ? (Kreal - kw) : 64); // wave-uniform: real samples among this wave's 64
    int csl[LBW] = {0, 0, 0, 0, 0, 0};
    unsigned long long as64 = 0;
    int ymax_hi = 0; // high word of the largest 2^32 (|V| / tau + dither): non-negative doubles order like their bit patterns
    double fp = 0.0;

    // digits of 4 consecutive samples -> one dword per plane, and the plane sums
    auto pack4 = [&](const unsigned (&dl)[4], const unsigned (&dhh)[4], v4i (&pl)[LBW], int slot) {
        if (COARSE) { // the three digits of the 23-bit value go to the planes 3..5; plane 2 reads zero (the consumers of the top four)
#pragma unroll
            for (int lb = 0; lb < 3; ++lb) {
                const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                const unsigned pk = __builtin_amdgcn_perm(__builtin_amdgcn_perm(dl[3], dl[2], sel), __builtin_amdgcn_perm(dl[1], dl[0], sel), 0x05040100u);
                pl[3 + lb][slot] = (int)pk;
                csl[3 + lb] = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csl[3 + lb], false);
            }
            pl[2][slot] = 0;
            return;
        }
#pragma unroll
        for (int lb = 0; lb < LBW; ++lb) {
            const unsigned bsel = (unsigned)(lb & 3);
            const unsigned sel = ((4u + bsel) << 8) | bsel;
            const unsigned t01 = lb < 4 ? __builtin_amdgcn_perm(dl[1], dl[0], sel) : __builtin_amdgcn_perm(dhh[1], dhh[0], sel);
            const unsigned t23 = lb < 4 ? __builtin_amdgcn_perm(dl[3], dl[2], sel) : __builtin_amdgcn_perm(dhh[3], dhh[2], sel);
            const unsigned pk = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
            pl[lb][slot] = (int)pk;
            csl[lb] = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csl[lb], false);
        }
    };

    if constexpr (FORM == 0) {
        // Exp forms: the arithmetic is laid out in layers of 8 independent instructions (two 4-sample groups), fenced by
        // sched_barriers, as in k_fwd_i8 -- a wave in its epilogue then issues back to back instead of waiting out the latency
        // of each dependent FP64 instruction.
#define SB __builtin_amdgcn_sched_barrier(0)
#ifdef I8W_LINE_STORES
        v4i plk[WM][LBW]; // the digits of both sample tiles, until the wave's image leaves in whole lines (below)
#endif
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const unsigned nsg = ~sgn[i]; // bit 8g + j set <=> s = +1
            v4i pl[LBW];
#pragma unroll
            for (int hg = 0; hg < 2; ++hg) {
                double Ea[8], tm[8], x[8], tj0[8], yy[8], pp[8], wk[8];
                int mneg[8], nn[8];
                // A: the energies, the sign bits, the weights
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pos = 8 * (2 * hg + (q >> 2)) + (q & 3);
                    Ea[q] = us[i][8 * hg + q];
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mneg[q]) : "v"(nsg), "n"(pos)); // -1 iff s = +1
                    if (!UNIW) wk[q] = w[kw + i * 32 + 8 * (2 * hg + (q >> 2)) + 4 * h + (q & 3)];
                }
                SB;
#ifdef ABL_NOEPI
#pragma unroll
                for (int q = 0; q < 8; ++q) yy[q] = fma(fabs(Ea[q]), 1.0e21, 2.0e23);
                if (false) {
#endif
                // B: x = -s E, n = rint(64 x / ln2), r = x - n ln2 / 64 (two-part constant)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[q] = flip_if(Ea[q], mneg[q]);
                    tm[q] = fma(x[q], 92.33248261689366, MAGIC);
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    nn[q] = __double2loint(tm[q]);
                    tm[q] = tm[q] - MAGIC;
                    tj0[q] = etab[nn[q] & 63];
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tm[q], -0.01083042469326756, x[q]);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tm[q], -2.9815858269852933e-12, x[q]);
                SB;
                // C: expm1(r) in FP64, |r| <= ln2/128: degree 5 leaves 4e-17
#pragma unroll
                for (int q = 0; q < 8; ++q) pp[q] = fma(x[q], 8.3333333333333332e-03, 4.1666666666666664e-02);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) pp[q] = fma(pp[q], x[q], 1.6666666666666666e-01);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) pp[q] = fma(pp[q], x[q], 0.5);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) pp[q] = fma(pp[q], x[q], 1.0);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) pp[q] = pp[q] * x[q];
                SB;
                // D: 2^32 (w / tau exp(-E) + dither)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3);
                    tj0[q] = __hiloint2double((int)((unsigned)__double2hiint(tj0[q]) + ((unsigned)nn[q] << 14)), __double2loint(tj0[q]));
                    yy[q] = (double)(int)(dh0 + (unsigned)idx * GOLD);
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tj0[q], pp[q], tj0[q]); // exp(-E)
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) yy[q] = fma(UNIW ? wk32 : wscale * (wk[q] * it), x[q], yy[q]);
#ifdef ABL_NOEPI
                }
#endif
                if (UNIW && wleft < 64) { // the last sample tile: padding samples carry no weight
                    asm volatile("; padding samples" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3) >= nreal) yy[q] = 0.0;
                }
                SB;
                // E: sign, rounding to the 48-bit integer, 6 balanced digits, 4 samples x 6 planes byte transpose
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    ymax_hi = max(ymax_hi, __double2hiint(yy[q]));
                    if (WANTF) {
                        const double ya = yy[q] + MAGIC32; // |V| / tau >= 0 rounded: its integer sits in the low 48 bits
                        as64 += (((unsigned long long)((unsigned)__double2hiint(ya) & 0xffffu)) << 32) | (unsigned)__double2loint(ya);
                    }
                    x[q] = flip_if(yy[q], mneg[q]) + MAGIC32;
                }
                SB;
#undef SB
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    unsigned dl[4], dhh[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) digits6(x[4 * gg + j], dl[j], dhh[j]);
                    pack4(dl, dhh, pl, 2 * hg + gg);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef I8W_LINE_STORES
#ifdef ABL_NOSTORE
            if (active && pl[0][0] == 0x12345678 && pl[5][3] == 0x1234567) {
#else
            if (active) {
#endif
                // (every lane stores its 16 bytes per plane and tile straight from registers: an instruction covers 32 rows x 16 B, a
                // quarter of each 64-byte row, and the write-combining of L2 puts the lines together -- WRITE_SIZE reads 8.8 GB per
                // launch for 6.1 GB of planes.  The line-wide form below (-DI8W_LINE_STORES) removes that inflation and is 0.8 %
                // SLOWER: profiles/r5_ab_i8w_line_stores.txt -- the partial writes cost no time, the LDS round trip does.)
#pragma unroll
                for (int lb = COARSE ? 2 : 0; lb < LBW; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
            }
#else
#pragma unroll
            for (int lb = COARSE ? 2 : 0; lb < LBW; ++lb) plk[i][lb] = pl[lb];
#endif
        }
#ifdef I8W_LINE_STORES
        // (A/B variant, not the default -- see above.)  The wave's 64 samples x 32 rows x 6 planes are ONE contiguous 12-KB image of
        // Vq.  It leaves through LDS -- the two ring
        // stages the last GEMM step no longer reads are free: every wave has passed that step's barrier -- so that each store
        // instruction writes 1 KB = eight whole 128-byte lines: the lanes put their 16-byte pieces where the image has them
        // (16-byte slots XOR-swizzled by (row >> 2) & 3: conflict-free on both sides), then read the image back linearly.  Three
        // planes at a time (6 KB per wave).  Rows that are not part of this pass keep what they hold (a re-run of some rows of a
        // tile must not touch the planes of the others): their lines are stored partially.
        {
            const int lastslot = (ntot > 0 ? ntot - 1 : 0) % NSW;
            int8_t *stg = lds + ((lastslot + 1 + (wave >> 1)) % NSW) * STAGEW + (wave & 1) * 6144;
            const unsigned amask = (unsigned)__ballot(active); // bit lr (lanes 0..31)
            int8_t *img = Vq + vq_off(mytile * 32, 0, kw, Kp, LBW);
            constexpr int G0 = COARSE ? 2 : 0, NG = COARSE ? 2 : 3; // plane groups [G0, G0 + NG), [G0 + NG, LBW)
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const int p0 = G0 + grp * NG;
#pragma unroll
                for (int lbl = 0; lbl < NG; ++lbl)
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        const int row = lbl * 32 + lr, sl = 2 * h + i;
                        *reinterpret_cast<v4i *>(stg + row * 64 + ((sl ^ ((row >> 2) & 3)) << 4)) = plk[i][p0 + lbl];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (one wave: its LDS accesses execute in order)
#pragma unroll
                for (int j = 0; j < 2 * NG; ++j) {
                    const int o = j * 1024 + lane * 16, row = o >> 6, sl = (o >> 4) & 3;
                    const v4i dv = *reinterpret_cast<const v4i *>(stg + row * 64 + ((sl ^ ((row >> 2) & 3)) << 4));
#ifdef ABL_NOSTORE
                    if (dv[0] == 0x12345678 && dv[3] == 0x1234567 && ((amask >> (row & 31)) & 1u))
#else
                    if ((amask >> (row & 31)) & 1u)
#endif
                        *reinterpret_cast<v4i *>(img + p0 * 2048 + o) = dv;
                }
                asm volatile("" ::: "memory");
            }
        }
#endif
    } else { // RPLE (:317): f = w log(1 + exp(-2E)), V = -2 w s / (1 + exp(2E)), E = s Ea
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            v4i pl[LBW];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t kk = kw + i * 32 + 8 * g + 4 * h;
                unsigned dl[4], dhh[4];
                asm volatile("" : "+v"(fp)); // gate each 4-sample group on the previous one (register pressure)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * g + j;
                    const double Ea = us[i][e];
                    const double dith = (double)(int)(dh0 + (unsigned)(i * 32 + 8 * g + j) * GOLD) * 2.3283064365386963e-10; // [-1/2, 1/2)
                    const bool neg = ((sgn[i] >> (8 * g + j)) & 1u) != 0; // s_u^k = -1
                    const double wk0 = UNIW ? (i * 32 + 8 * g + j < nreal ? wuni : 0.0) : w[kk + j];
                    const double E2 = neg ? -2.0 * Ea : 2.0 * Ea;
                    const double u = exp_tab(-fabs(E2), etab); // in (0, 1]
                    const double opu = 1.0 + u;
                    double rcp = __builtin_amdgcn_rcp(opu); // 1 / (1 + u), two Newton steps
                    rcp = fma(fma(-opu, rcp, 1.0), rcp, rcp);
                    rcp = fma(fma(-opu, rcp, 1.0), rcp, rcp);
                    const double sig = E2 >= 0.0 ? u * rcp : rcp; // 1 / (1 + exp(2E))
                    const double y = fma(2.0 * wk0 * it, sig, dith); // |V| / tau + dither
                    digits6((neg ? y : -y) + MAGIC, dl[j], dhh[j]);
                    // log(1 + u), 1 + u in (1, 2]: table of log c_j on 64 intervals + log1p of the residual
                    int jt = (int)(u * 64.0);
                    jt = jt > 63 ? 63 : jt;
                    const double r1 = fma(opu, etab[64 + jt], -1.0); // |r1| <= 1/128
                    double lp = fma(r1, 1.0 / 7.0, -1.0 / 6.0);
                    lp = fma(lp, r1, 0.2);
                    lp = fma(lp, r1, -0.25);
                    lp = fma(lp, r1, 1.0 / 3.0);
                    lp = fma(lp, r1, -0.5);
                    lp = fma(lp, r1, 1.0);
                    const double l1p = fma(lp, r1, etab[128 + jt]);
                    fp += wk0 * ((E2 < 0.0 ? -E2 : 0.0) + l1p);
                }
                pack4(dl, dhh, pl, g);
            }
            if (active) {
#pragma unroll
                for (int lb = 0; lb < LBW; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
            }
        }
    }
#ifdef ABL_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tst[4] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && wave == 0 && active) { // 100 MHz ticks: start, end of sweep A, of the fold, of sweep B, of the epilogue (stores landed)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(Vq + vq_off(mytile * 32, 0, k0, Kp, LBW));
        for (int q = 0; q < 5; ++q) o[q] = tst[q];
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) /* HW_ID */;
    }
#endif
    // per-slot sums: sum_k V (two halves), sum_k |V| (objective-only passes), max_k |V| >> 16
    long long cs_lo = (long long)csl[0] + 256ll * csl[1] + 65536ll * csl[2];
    long long cs_hi = (long long)csl[3] + 256ll * csl[4] + 65536ll * csl[5];
    cs_lo += __shfl_xor(cs_lo, 32);
    cs_hi += __shfl_xor(cs_hi, 32);
    if (active && h == 0) {
        if (!COARSE) atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs_lo);
        atomicAdd(reinterpret_cast<unsigned long long *>(&csum2[r]), (unsigned long long)cs_hi);
    }
    if (FORM == 0) {
        if (WANTF) {
            as64 += __shfl_xor(as64, 32);
            if (active && h == 0) {
                atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), as64 & 0xffffffffull);
                atomicAdd(reinterpret_cast<unsigned long long *>(&asum2[r]), as64 >> 32);
            }
        }
        ymax_hi = max(ymax_hi, __shfl_xor(ymax_hi, 32));
        // the high word + 1 bounds 2^32 max(|V| / tau + dither) from above (to 2^-20 relative); in units of 2^16 tau
        const double ymax = __hiloint2double(ymax_hi + 1, 0);
        // (coarse: ymax bounds 2^32 |V| / (2^24 tau); reported in the same unit, rounded up to the next multiple of 2^24 tau)
        const unsigned mxu = COARSE ? (((unsigned)fmin(ymax * 2.3283064365386963e-10 /* 2^-32 */, 8388606.0) + 1u) << 8)
                                    : (unsigned)fmin(ymax * 3.5527136788005009e-15 /* 2^-48 */, 4294967295.0);
        if (active && h == 0) atomicMax(&mmax[r], mxu);
    } else {
        fp += __shfl_xor(fp, 32);
        if (active && h == 0) unsafeAtomicAdd(&fsum[r], fp);
    }
}

// G[row][c] = tau_r (C_lo - 2 S_lo + 2^24 (C_hi - 2 S_hi)),  S_half = sum_l 256^l Gacc_l over the half's three planes (x = 1 - 2b),
// C_half = sum_k of the half's digits; G[row][cconst] = tau_r (C_lo + 2^24 C_hi).  f: RISE / logRISE from the gradient's own
// column (with the gradient) or from sum |V| (objective-only passes); RPLE keeps the forward kernel's FP64 sum.
__global__ __launch_bounds__(256) void k_finalize_i8w(const int32_t *__restrict__ Gacc, const double *__restrict__ tau,
                                                      const long long *__restrict__ csum, const long long *__restrict__ csum2,
                                                      const long long *__restrict__ asum, const long long *__restrict__ asum2,
                                                      const int *__restrict__ srow, const int *__restrict__ rowcol, int slot0, int64_t Qp,
                                                      int64_t Qfp, int64_t Qf, int64_t cconst, int form, int want_grad,
                                                      double *__restrict__ G, double *__restrict__ f, int nplanes, int64_t plane_stride,
                                                      const unsigned *__restrict__ mmax, SlotResult *__restrict__ res,
                                                      int coarse /* only the high half carries the values: multiples of 2^24 tau */) {
    const int r = slot0 + blockIdx.y;
    if (rowcol[r] < 0) return;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double t = tau[r];
    const int tile = r >> 5, rl = r & 31;
    auto gcol = [&](int64_t col) -> double {
        long long s[2] = {0, 0};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (coarse && half == 0) continue;
#pragma unroll
            for (int l = 2; l >= 0; --l) {
                long long a = 0;
                for (int pl = 0; pl < nplanes; ++pl) a += (long long)Gacc[pl * plane_stride + ((int64_t)(tile * LBW + 3 * half + l) * 32 + rl) * Qfp + col];
                s[half] = s[half] * 256 + a;
            }
        }
        return fma((double)(csum2[r] - 2 * s[1]), 16777216.0, coarse ? 0.0 : (double)(csum[r] - 2 * s[0]));
    };
    if (c == 0) {
        double fv = f ? f[r] : 0.0; // RPLE: the forward kernel's FP64 sum
        if (form != 2) {
            if (want_grad) fv = -t * gcol(rowcol[r]); // f = sum_k w exp(-E) = -sum_k V_k s_k = -G[r][u]
            else fv = t * (coarse ? 16777216.0 : 1.0) * fma((double)asum2[r], 4294967296.0, (double)asum[r]);
            f[r] = fv;
        }
        if (res) res[r] = SlotResult{fv, t, mmax[r], 0u};
    }
    if (!want_grad || c >= Qp) return;
    double v = 0.0;
    if (c < Qf) v = t * gcol(c);
    else if (c == cconst) v = t * fma((double)csum2[r], 16777216.0, coarse ? 0.0 : (double)csum[r]);
    G[(int64_t)srow[r] * Qp + c] = v;
}

void launch_finalize_i8w(const int32_t *Gacc, const SlotScalars &sc, const int *srow, const int *rowcol, int slot0, int ns, int64_t Qp,
                         int64_t Qfp, int64_t Qf, int64_t cconst, int form, int want_grad, double *G, double *f, int nplanes,
                         int64_t plane_stride, SlotResult *res, bool coarse, hipStream_t st) {
    hipLaunchKernelGGL(k_finalize_i8w, dim3((unsigned)((Qp + 255) / 256), (unsigned)ns), dim3(256), 0, st, Gacc, sc.tau, sc.csum, sc.csum2,
                       sc.asum, sc.asum2, srow, rowcol, slot0, Qp, Qfp, Qf, cconst, form, want_grad, G, f, nplanes, plane_stride, sc.mmax, res,
                       coarse ? 1 : 0);
}

template <int FORM, bool WANTF, bool WIDE, bool UNIW, bool COARSE>
static void launch_fwd_w5(const FwdWArgs &a) {
    constexpr int shmem = RINGW + 512 + 1024; // ring + exp, log tables
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd_i8w<FORM, WANTF, WIDE, UNIW, COARSE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem); // per device: set on every launch
    const DevProblem &d = *a.d;
    const int ntk = (int)(d.Kp / 256);
    const int grid = ((ntk + 7) / 8) * 8 * a.ngroups; // one workgroup per (sample tile, node tile)
    hipLaunchKernelGGL((k_fwd_i8w<FORM, WANTF, WIDE, UNIW, COARSE>), dim3(grid), dim3(256), shmem, a.st, d.Xb, d.Sb, a.Tq, a.rowcol, a.groups,
                       a.ngroups, d.w, a.sc->sigma, a.sc->qconst, a.sc->qconst2, a.sc->invtau, d.Kp, ntk, a.zero_theta ? 0 : (int)(d.Qfp / 64), d.wuni, d.K, a.Vq,
                       a.sc->csum, a.sc->csum2, a.sc->asum, a.sc->asum2, a.F, a.sc->mmax, a.cc ? a.cc->cnk : nullptr, a.cc ? a.cc->Xc : nullptr,
                       a.cc ? a.cc->xc_tile : 0, (int)(d.Qfp / 64));
}

template <int FORM, bool WANTF, bool WIDE, bool UNIW>
static void launch_fwd_w4(const FwdWArgs &a) {
    if constexpr (FORM == 0) {
        if (a.coarse) return launch_fwd_w5<FORM, WANTF, WIDE, UNIW, true>(a);
    }
    launch_fwd_w5<FORM, WANTF, WIDE, UNIW, false>(a);
}

template <int FORM, bool WANTF, bool WIDE>
static void launch_fwd_w3(const FwdWArgs &a) {
    if (a.d->wuni > 0.0) launch_fwd_w4<FORM, WANTF, WIDE, true>(a);
    else launch_fwd_w4<FORM, WANTF, WIDE, false>(a);
}

template <int FORM, bool WANTF>
static void launch_fwd_w2(const FwdWArgs &a) {
    if (a.d->Qfp > 32768) launch_fwd_w3<FORM, WANTF, true>(a);
    else launch_fwd_w3<FORM, WANTF, false>(a);
}

void launch_fwd_i8w(const FwdWArgs &a) {
    if (a.form == 2) launch_fwd_w2<2, true>(a);
    else if (a.want_f) launch_fwd_w2<0, true>(a);
    else launch_fwd_w2<0, false>(a);
}

} // namespace gml
