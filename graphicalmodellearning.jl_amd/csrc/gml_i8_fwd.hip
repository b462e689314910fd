// Int8-limb path, part 2: the forward kernel of the 38/31-bit pass "i8x" and of the Hessian-vector forms (overview: gml_i8.h).
#include "gml_i8.h"
#include <algorithm>
#include <string>
#include <type_traits>

namespace gml {

// V / tau of one element of the exp forms: -s rint(w/tau exp(-s E) + dither), E = s Ea.  Ea to 3e-10 relative before
// the rounding.  FP64 range reduction with one FMA (the product t * ln2/64 is not rounded inside an FMA), FP32
// polynomial for expm1 of the reduced argument, table of 2^(j/64), exponent added as an integer, and the final rounding
// to an integer through the 1.5 * 2^52 trick, after adding a dither in [-1/2, 1/2) that is a fixed function of
// (node, sample): the rounding is then "stochastic" -- still deterministic and within one unit, but uncorrelated across
// samples.  Round-to-nearest is coherent whenever a sparse theta row leaves only a few distinct energies (thousands of
// samples share each rounding error), which made the realised error of f and grad approach the K * tau / 2 worst case
// instead of ~ sqrt(K) * tau.
// The epilogue is bound by the number of vector instructions, so this is written for few of them:
//   * sb (bit 0: s = +1) flips the sign of Ea going in (x = -s E) and of the result coming out by adding sb << 31 to
//     the high word -- round-half-even is symmetric, so rounding -y gives minus the rounding of y;
//   * everything from the weight on is scaled by 2^32 (wk32 = 2^32 w / tau; exact): the dither is then the hash itself,
//     converted int -> double, and the integer is read off below 1.5 * 2^84;
//   * the table holds 2^(j/64) with j << 14 taken off the high word: the exponent of 2^(n >> 6), n = 64 q + j, goes on
//     as n << 14 (= (q << 20) + (j << 14)) in one shift-add.
__device__ __forceinline__ int vq_exp(double Ea, unsigned sb, double wk32, unsigned dh, const double *__restrict__ tabb) {
    const double MAGIC = 6755399441055744.0;                  // 1.5 * 2^52
    const double MAGIC32 = 6755399441055744.0 * 4294967296.0; // 1.5 * 2^84: rounds to multiples of 2^32
    const int flip = (int)(sb << 31);
    const double x = __hiloint2double(__double2hiint(Ea) + flip, __double2loint(Ea)); // -s E
    const double tm = fma(x, 92.33248261689366, MAGIC);                               // 64/ln2
    const int n = __double2loint(tm);
    const double t = tm - MAGIC;
    const double r = fma(t, -0.010830424696249145, x); // ln2/64
    const float rf = (float)r;
    float d = fmaf(rf, 4.1666668e-02f, 1.6666667e-01f);
    d = fmaf(d, rf, 0.5f);
    d = fmaf(d, rf, 1.0f);
    d = d * rf; // expm1(r)
    const double tj0 = tabb[n & 63];
    const double tj = __hiloint2double((int)((unsigned)__double2hiint(tj0) + ((unsigned)n << 14)), __double2loint(tj0)); // 2^(n/64)
    const double res = fma(tj, (double)d, tj);
    const double y = fma(wk32, res, (double)(int)dh); // 2^32 (|V| / tau + dither)
    const double ys = __hiloint2double(__double2hiint(y) + flip, __double2loint(y));
    return __double2loint(ys + MAGIC32);
}

// ------------------------------------------------------------------------------------------
// forward: C[k][m] = sum_c b[k][c] * Tq[m][c] on i8 MFMA (b = [x = -1] from the bit image), then the
// pointwise epilogue
//   E = s * sigma_r * (q0 + S - 2 sum_l 256^l C_l),  V = -w_k exp(-E) s  (RISE / logRISE),
//   V -> LB balanced limbs -> Vq planes (via an LDS transpose so that global stores are 16 B).
// Workgroup = 4 waves along the samples: 256 samples x one 32-node tile x LF limb planes.
// Stage image of the 4-deep LDS-DMA ring: 2 KB of bits (two 128-sample pieces) + the (tile, kt) image
// of Tq.  The A fragments never touch LDS as bytes: each lane expands its dword of bits in registers.
// ------------------------------------------------------------------------------------------
template <int LF, int FORM /* 0: exp forms (RISE, logRISE), 2: RPLE; Hessian-vector products: 3 (exp forms), 4 (RPLE) */,
          bool WANTF,
          bool WIDE /* more than 32768 statistics columns: |acc_l| <= 128 Qfp no longer leaves room for the int32 pairing */,
          bool COARSE /* exp forms: V rounded to multiples of 2^8 tau (dithered 23 bits: planes 1..3, plane 0 zero), for the cheap early
                         passes of a solve (with LF = 4 and a 3-plane backward launch) */,
          bool UNIW /* every real sample has the weight wuni (all counts equal): no weight loads.  A template parameter, not
                       a run-time test: a branch per element would put each of the epilogue's 32 dependent chains (range
                       reduction -> table read -> polynomial -> rounding) into its own basic block and serialise them */>
__global__ __launch_bounds__(256, 2) void k_fwd_i8(
    const unsigned *__restrict__ Xb, const unsigned *__restrict__ Sb, const int8_t *__restrict__ Tq,
    const int *__restrict__ rowcol, const int *__restrict__ groups, int ngroups, const double *__restrict__ w,
    const double *__restrict__ sigma, const long long *__restrict__ qconst, const double *__restrict__ invtau,
    int64_t Kp, int ntiles_k, int nk_all /* 64-column steps of a sweep over all columns (0: every row of Theta is zero) */,
    double wuni /* > 0: every real sample has this weight */,
    int64_t Kreal, int8_t *__restrict__ Vq, long long *__restrict__ csum, long long *__restrict__ asum,
    double *__restrict__ fsum, unsigned *__restrict__ mmax,
    // Hessian-vector forms only: the limb planes of V written by the rows' last objective pass, the slot that holds
    // them for each slot of this pass, and their scales
    const int8_t *__restrict__ Vsrc, const int *__restrict__ vmap, const double *__restrict__ tauV,
    int vsrc_lbt /* planes of a source image */, int vsrc_pl0 /* first of the 4 planes read */, double vsrc_scale /* their unit / tauV */,
    // sub-sampled passes (Hessian-vector products over a part of the configurations): compact sample tile t stands for the
    // tile (t / part_tiles) * chunk_tiles + t % part_tiles -- the first part_tiles tiles of every split-K chunk of the
    // backward kernel.  chunk_tiles == part_tiles: every configuration.
    int chunk_tiles, int part_tiles,
    // column compaction of objective passes (gml_i8_pack.hip: k_col_union): steps of each tile's compact image (-1: all columns), the
    // images, bytes per tile, and the steps between two tiles' Tq images (= Qfp / 64 whatever is swept)
    const int *__restrict__ cnk, const int8_t *__restrict__ Xc, int64_t xc_tile, int nk_tq) {
    constexpr int WM = 2;                 // 32-sample MFMA tiles per wave
    constexpr bool HV = FORM >= 3;
    constexpr int BR = 32 * LF;           // rows of the Tq image
    constexpr int NPIECE = 2 + BR / 16, NP = (NPIECE + 3) / 4;
    // Ring stages hold DS consecutive 64-column steps: one barrier per DS steps (the waves of a workgroup then re-align
    // half as often, and the LDS reads of a stage's second step issue under the MFMAs of its first).
    constexpr int DS = 2;
    constexpr int STEP = NPIECE * 1024, STAGE = DS * STEP, NS = 3;
    constexpr int RING = NS * STAGE;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[]; // ring, then the exp (and log) tables
    double *etab = reinterpret_cast<double *>(lds + RING);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    if (tid < 64) {
        const double v = exp2((double)tid / 64.0);
        // exp forms: the table of vq_exp(), j << 14 taken off the high word
        etab[tid] = FORM == 0 ? __hiloint2double(__double2hiint(v) - (tid << 14), __double2loint(v)) : v;
    }
    if (FORM == 2 && tid < 64) { // log table for RPLE: c_j = 1 + (j + 1/2)/64 -> 1/c_j, log c_j
        const double cj = 1.0 + ((double)tid + 0.5) / 64.0;
        etab[64 + tid] = 1.0 / cj;
        etab[128 + tid] = log(cj);
    }
    __syncthreads(); // tables visible to every wave (the ring uses raw s_barrier without an LDS wait)

    // XCD-aware L2 blocking.  Blocks b and b+8 share an XCD (round-robin dispatch); XCD x owns the
    // sample tiles st = 8*i + x.  Within an XCD: groups of TG node tiles (outer), sample tiles
    // (middle), the TG node tiles (inner): Tq of the group stays resident in the XCD's L2 over the
    // sweep and each bit piece is fetched once per node-tile group.  The last group holds ngroups % TG tiles; the grid
    // has no idle workgroups beyond the sample tiles that pad ntiles_k to a multiple of 8 (a node-sharded rank runs few
    // node tiles: half of its launch would otherwise be workgroups that start only to exit).
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
    if (chunk_tiles != part_tiles) st = (st / part_tiles) * chunk_tiles + st % part_tiles;
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

    // per-lane source of each 1-KB piece this wave loads, and its advance per 64-column step
    const int8_t *src[NP];
    int adv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        int pc = wave + 4 * j;
        if (pc >= NPIECE) pc = NPIECE - 1; // duplicate piece: keeps the per-wave vmcnt count uniform
        if (pc < 2) {
            src[j] = xbase + ((int64_t)(2 * st + pc) * nk) * 1024 + lane * 16;
            adv[j] = 1024;
        } else {
            const int row = (pc - 2) * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3); // XOR swizzle applied to the source (LDS side is linear)
            src[j] = Tq + ((int64_t)mytile * nk_tq * BR + row) * 64 + slot * 16;
            adv[j] = BR * 64;
        }
    }
    const int nst = (nk + DS - 1) / DS; // ring stages of this tile
    auto issue = [&](int ks) { // stage ks = steps DS ks .. DS ks + DS - 1 (a step beyond the last one: the last one again, so
                               // that every stage counts the same number of loads for the vmcnt waits)
        int8_t *stage_base = lds + (ks % NS) * STAGE;
#pragma unroll
        for (int sub = 0; sub < DS; ++sub) {
            int kt = DS * ks + sub;
            kt = kt < nk ? kt : nk - 1;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int pc = wave + 4 * j;
                if (pc >= NPIECE) pc = NPIECE - 1;
                __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)kt * adv[j]), (lptr_t)(stage_base + sub * STEP + pc * 1024), 16, 0, 0);
            }
        }
    };

    v16i acc[WM][LF]; // first written by the peeled step 0 below (C operand = the constant 0: no clearing moves)

    // the epilogue's per-lane inputs are fetched now, so that their latency hides under the GEMM
    const int r = mytile * 32 + lr;
    const int rc = rowcol[r];
    const bool active = rc >= 0;
    // the node's sign bits for this wave's 64 samples (word i <-> MFMA tile i), shifted so that bit 8g + j is this
    // lane's sample 8g + 4h + j of the tile
    unsigned sgn[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) sgn[i] = active ? (Sb[(int64_t)rc * (Kp >> 5) + ((k0 + wave * 64) >> 5) + i] >> (4 * h)) : 0u;
    // samples at or beyond Kreal are padding (they carry no weight): this lane's element (i, g, j) sits 32 i + 8 g + j
    // samples after its first one, k0 + 64 wave + 4 h
    const int64_t left = Kreal - (k0 + wave * 64 + 4 * h);
    const int nreal = left > 64 ? 64 : (left < 0 ? 0 : (int)left);
    const double sg = active ? sigma[r] : 0.0;
    const double q0 = active ? (double)qconst[r] : 0.0;
    const double it = active ? invtau[r] : 0.0;

    // Two workgroups share a CU, one wave of each per SIMD.  The wave that is in its GEMM gets the issue priority over
    // the one that is in its epilogue: the matrix pipe is the scarcer resource (-3 % forward time, interleaved A/B).
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) issue(s);
    auto gemm_stage = [&](int ks, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        ring_wait_ahead<DS * NP>(nst - 1 - ks > NS - 2 ? NS - 2 : nst - 1 - ks); // NS - 2 later stages may still be in flight
        if (ks + NS - 1 < nst) issue(ks + NS - 1);
#pragma unroll
        for (int sub = 0; sub < DS; ++sub) {
            if (sub > 0 && DS * ks + sub >= nk) break; // (an odd number of steps: the last stage is half full)
            const int8_t *cur = lds + (ks % NS) * STAGE + sub * STEP;
            unsigned vb[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = wave * 64 + i * 32 + lr;
                vb[i] = *reinterpret_cast<const unsigned *>(cur + (row >> 7) * 1024 + (((row & 127) * 2 + h) << 2));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4i fa[WM], fb[LF];
#pragma unroll
                for (int l = 0; l < LF; ++l)
                    fb[l] = *reinterpret_cast<const v4i *>(cur + 2048 + lds_off(l * 32 + lr, 2 * t + h));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][e] = (int)((vb[i] >> (4 * t + e)) & 0x01010101u);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int l = 0; l < LF; ++l) {
                        if (FIRST && sub == 0 && t == 0) acc[i][l] = MFMA_I8(fa[i], fb[l], ((v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}));
                        else acc[i][l] = MFMA_I8(fa[i], fb[l], acc[i][l]);
                    }
            }
        }
    };
    if (nk > 0) { // Qfp >= 64; nk = 0: every row of Theta is zero (the caller says so), the sums are
        gemm_stage(0, std::true_type{});
        for (int ks = 1; ks < nst; ++ks) gemm_stage(ks, std::false_type{});
    } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int l = 0; l < LF; ++l)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- epilogue ----------------------------------------------------------------------------
    // lane <-> node row (lr), register e <-> sample (e&3) + 8*(e>>2) + 4*h within the 32-sample tile.  The
    // Vq image stores a step's samples in the order vq_pos() (gml_dev.h), in which this lane's 16 samples of
    // tile i are 16 contiguous bytes per limb: no LDS transpose, two 16-byte stores per limb.
    const int form = FORM;
    int8_t *vimg = Vq + vq_off(mytile * 32 + lr, 0, k0 + wave * 64, Kp) + h * 32; // row (limb 0, lr) of the wave's image
    const int8_t *vsrc = nullptr; // Hessian-vector forms: the same bytes of the row's V image
    double tvh = 0.0;
    if (HV && active) {
        const int vs = vmap[r];
        vsrc = Vsrc + vq_off(vs, vsrc_pl0, k0 + wave * 64, Kp, vsrc_lbt) + h * 32;
        tvh = tauV[vs] * vsrc_scale;
    }
    long long cs = 0, as = 0;
    double fp = 0.0;
    int mx = 0;
    const int64_t kw = k0 + wave * 64; // first sample of this wave
    constexpr double WSCALE = COARSE ? 16777216.0 : 4294967296.0; // 2^32 w / tau (vq_exp), 2^24 for the coarse form
    const double sgq0 = sg * q0, wk32 = WSCALE * (wuni * it);
    double sg2 = -2.0 * sg;
    // dither of the V rounding: golden-ratio (Weyl) sequence in the global sample index, offset per node --
    // independent of tiling, node sharding and compaction, so results stay bit-identical across GPU counts
    const unsigned dh0 = (unsigned)rc * 0x85EBCA6Bu + (unsigned)(kw + 4 * h) * 0x9E3779B9u;
    if constexpr (FORM == 0) {
        // Exp forms (RISE, logRISE): the arithmetic of vq_exp(), laid out in STAGES over 8 elements at a time (two 4-sample
        // groups).  Every stage is 8 independent copies of a short chain, fenced by sched_barriers: a wave in its epilogue
        // then issues back to back instead of waiting out the 16-20 cycle latency of each dependent FP64 instruction (the
        // element-at-a-time form left the scheduler, at 200+ live registers, emitting each element's chain serially).
        // Fewer instructions per element as well: the sign is applied to the rounded magnitude in integers (one bit-field
        // extract serves both sign flips), sum_k V comes from dot4 over the packed digit planes, max|V| from the unsigned
        // magnitudes, and padding samples are masked in a branch only the last sample tile takes.
        constexpr double MAGIC = 6755399441055744.0, MAGIC32 = 6755399441055744.0 * 4294967296.0;
        constexpr unsigned GOLD = 0x9E3779B9u, CB = 0x80808080u;
        const int wleft = (int)((Kreal - kw) < 64 ? (Kreal - kw) : 64); // wave-uniform: real samples among this wave's 64
        int csl[LB] = {0, 0, 0, 0};
        unsigned mxu = 0;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const unsigned nsg = ~sgn[i]; // bit 8g + j set <=> s = +1
            v4i pl[LB];
#pragma unroll
            for (int hg = 0; hg < 2; ++hg) {
                // Layers of 8 independent instructions each, fenced (SB): whatever order the scheduler picks inside a layer, a
                // result is not needed before 8 issue slots later.
#define SB __builtin_amdgcn_sched_barrier(0)
                double a[8], Ea[8], wk[8], tm[8], x[8], tj0[8], yy[8];
                int mneg[8], nn[8];
                float rf[8], dd[8];
                unsigned mag[8];
                // A: exact recombination of the limb planes (pairs in int32, then FP64; all planes through FP64 when WIDE)
                if (WIDE || LF != 5) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = 8 * hg + q;
                        if (WIDE) {
                            a[q] = (double)acc[i][LF - 1][e];
#pragma unroll
                            for (int l = LF - 2; l >= 0; --l) a[q] = fma(a[q], 256.0, (double)acc[i][l][e]);
                        } else if (LF == 4) {
                            const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                            const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                            a[q] = fma((double)mid, 65536.0, (double)lo);
                        } else {
                            const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                            a[q] = fma((double)acc[i][2][e], 65536.0, (double)lo);
                        }
                    }
                    SB;
                } else { // LF == 5, the production form, layer by layer
                    int lo[8], mid[8];
                    double c4[8], cm[8], cl[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = 8 * hg + q;
                        lo[q] = acc[i][0][e] + (acc[i][1][e] << 8);
                        mid[q] = acc[i][2][e] + (acc[i][3][e] << 8);
                        c4[q] = (double)acc[i][4][e];
                    }
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        cm[q] = (double)mid[q];
                        cl[q] = (double)lo[q];
                    }
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = fma(c4[q], 65536.0, cm[q]);
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = fma(a[q], 65536.0, cl[q]);
                    SB;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pos = 8 * (2 * hg + (q >> 2)) + (q & 3);
                    Ea[q] = fma(a[q], sg2, sgq0);
                    // -1 iff s = +1 (v_bfe_i32 spelled out: the generic lowering is a shift pair, and the compiler then
                    // re-derives the two sign flips below from the shifted word with an and + an arithmetic shift each)
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mneg[q]) : "v"(nsg), "n"(pos));
                    if (!UNIW) wk[q] = w[kw + i * 32 + 8 * (2 * hg + (q >> 2)) + 4 * h + (q & 3)];
                }
                SB;
                // B: x = -s E, range reduction n = rint(64 x / ln2), r = x - n ln2 / 64
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[q] = __hiloint2double(__double2hiint(Ea[q]) + (mneg[q] << 31), __double2loint(Ea[q]));
                    tm[q] = fma(x[q], 92.33248261689366, MAGIC);
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    nn[q] = __double2loint(tm[q]);
                    tm[q] = tm[q] - MAGIC;
                    tj0[q] = etab[nn[q] & 63]; // 2^(j/64) (exponent bits of j << 14 taken off: see the table's construction)
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tm[q], -0.010830424696249145, x[q]); // r
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) rf[q] = (float)x[q];
                SB;
                // C: expm1(r) in FP32
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(rf[q], 4.1666668e-02f, 1.6666667e-01f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(dd[q], rf[q], 0.5f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(dd[q], rf[q], 1.0f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = dd[q] * rf[q];
                SB;
                // D: 2^32 (w / tau exp(-E) + dither), rounded to an integer magnitude
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3);
                    tj0[q] = __hiloint2double((int)((unsigned)__double2hiint(tj0[q]) + ((unsigned)nn[q] << 14)), __double2loint(tj0[q]));
                    x[q] = (double)dd[q];
                    yy[q] = (double)(int)(dh0 + (unsigned)idx * GOLD); // the dither
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tj0[q], x[q], tj0[q]); // exp(-E)
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) yy[q] = fma(UNIW ? wk32 : WSCALE * (wk[q] * it), x[q], yy[q]);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) mag[q] = (unsigned)__double2loint(yy[q] + MAGIC32); // >= 0: y > -2^31
                if (UNIW && wleft < 64) { // the last sample tile: padding samples carry no weight
                    asm volatile("; padding samples" ::: "memory"); // (keeps this a branch: as selects it costs every tile 2 instructions per element)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3) >= nreal) mag[q] = 0u;
                }
                SB;
#undef SB
                // E: sign, 4 balanced base-256 digits per sample, 4 samples x 4 limbs byte transpose
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    unsigned dj[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = 4 * gg + j;
                        mxu = mag[q] > mxu ? mag[q] : mxu;
                        if (WANTF) as += COARSE ? (long long)mag[q] << 8 : (long long)mag[q];
                        // V / tau = -s |V| / tau = (mag ^ m) - m, then the 4 balanced digits (v + CB) ^ CB: one v_xad
                        unsigned tq;
                        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(tq) : "v"(mag[q]), "v"(mneg[q]), "v"(CB - (unsigned)mneg[q]));
                        dj[j] = COARSE ? (tq ^ CB) << 8 : tq ^ CB; // (coarse: the three digits of the 23-bit value in the planes 1..3)
                    }
#pragma unroll
                    for (int lb = 0; lb < LB; ++lb) {
                        const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                        const unsigned t01 = __builtin_amdgcn_perm(dj[1], dj[0], sel);
                        const unsigned t23 = __builtin_amdgcn_perm(dj[3], dj[2], sel);
                        const unsigned pk = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                        pl[lb][2 * hg + gg] = (int)pk;
                        csl[lb] = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csl[lb], false); // sum of the 4 digits
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (active) {
#pragma unroll
                for (int lb = 0; lb < LB; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
            }
        }
        cs = (long long)csl[0] + 256ll * csl[1] + 65536ll * csl[2] + 16777216ll * csl[3];
        cs += __shfl_xor(cs, 32);
        as += __shfl_xor(as, 32);
        const unsigned mo = (unsigned)__shfl_xor((int)mxu, 32);
        mxu = mo > mxu ? mo : mxu;
        if (COARSE) mxu = (mxu + 1u) << 8; // in units of tau, rounded up to the next multiple of 2^8 tau
        if (active && h == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs);
            if (WANTF) atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), (unsigned long long)as);
            atomicMax(&mmax[r], mxu);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        v4i pl[LB], pv[LB];
        if (HV) {
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) pv[lb] = active ? *reinterpret_cast<const v4i *>(vsrc + lb * 32 * 64 + i * 16) : (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t kk = kw + i * 32 + 8 * g + 4 * h;
            unsigned dj[4], dv[4];
            if (HV) { // the 4 balanced digits of V of each of the group's 4 samples (inverse of the transpose below)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sel = ((4u + j) << 8) | (unsigned)j;
                    const unsigned t01 = __builtin_amdgcn_perm((unsigned)pv[1][g], (unsigned)pv[0][g], sel);
                    const unsigned t23 = __builtin_amdgcn_perm((unsigned)pv[3][g], (unsigned)pv[2][g], sel);
                    dv[j] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                }
            }
            if (FORM == 2) { // RPLE: gate each 4-sample group on the previous one (its longer arithmetic otherwise
                             // interleaves across groups and spills); pure arithmetic floats across sched_barriers
                asm volatile("" : "+v"(sg2), "+v"(fp));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * g + j;
                // exact recombination of the limb planes: pairs in int32 (|acc_l| <= 128 Qfp <= 2^22, so
                // |acc_l + 256 acc_{l+1}| < 2^31), then FP64; beyond 32768 columns every plane goes through FP64
                // (|a| < 2^53 always: a is the integer sum_c q_c b_c with |q_c| <= 2^38)
                double a;
                if (WIDE) {
                    a = (double)acc[i][LF - 1][e];
#pragma unroll
                    for (int l = LF - 2; l >= 0; --l) a = fma(a, 256.0, (double)acc[i][l][e]);
                } else if (LF == 5) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                    a = fma((double)acc[i][4][e], 65536.0, (double)mid);
                    a = fma(a, 65536.0, (double)lo);
                } else if (LF == 4) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                    a = fma((double)mid, 65536.0, (double)lo);
                } else if (LF == 3) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    a = fma((double)acc[i][2][e], 65536.0, (double)lo);
                } else { // 2 limbs: directions of the Hessian-vector passes
                    a = (double)(acc[i][0][e] + (acc[i][1][e] << 8));
                }
                const double Ea = fma(a, sg2, sgq0);                // |E| pre-sign: sigma * (q0 + S - 2 A)
                const double dith = (double)(int)(dh0 + (unsigned)(i * 32 + 8 * g + j) * 0x9E3779B9u) * 2.3283064365386963e-10; // [-1/2, 1/2)
                const bool neg = ((sgn[i] >> (8 * g + j)) & 1u) != 0; // s_u^k = -1
                int vq;
                if (HV) {
                    // Hessian-vector product: u_k = h_k (x_k . p), h_k the curvature weight of the row's iterate:
                    // |V_k| for the exp forms, 2a(1 - a/(2w)) with a = |V_k| for RPLE.  In units of tau_V * emax:
                    // u = (h_k / tau_V) * (x_k . p) / emax, |.| <= 2^31 (Ea = x_k . p, it = 1 / emax)
                    const int qv = (int)((dv[j] ^ 0x80808080u) - 0x80808080u);
                    double hh = (double)(qv < 0 ? -qv : qv);
                    if (FORM == 4) {
                        const double wk0 = UNIW ? (i * 32 + 8 * g + j < nreal ? wuni : 0.0) : w[kk + j];
                        hh = wk0 > 0.0 ? 2.0 * hh * (1.0 - hh * tvh / (2.0 * wk0)) : 0.0;
                    }
                    vq = __double2loint(fma(hh, Ea * it, dith) + 6755399441055744.0);
                } else if (FORM == 2) { // RPLE (:317): f = w log(1 + exp(-2E)), V = -2 w s / (1 + exp(2E)), E = s * Ea
                    const double wk0 = UNIW ? (i * 32 + 8 * g + j < nreal ? wuni : 0.0) : w[kk + j];
                    const double E2 = neg ? -2.0 * Ea : 2.0 * Ea;
                    const double u = exp_tab(-fabs(E2), etab); // in (0, 1]
                    const double opu = 1.0 + u;
                    double rc = __builtin_amdgcn_rcp(opu); // 1 / (1 + u), two Newton steps
                    rc = fma(fma(-opu, rc, 1.0), rc, rc);
                    rc = fma(fma(-opu, rc, 1.0), rc, rc);
                    const double sig = E2 >= 0.0 ? u * rc : rc; // 1 / (1 + exp(2E))
                    const int mag = __double2loint(fma(2.0 * wk0 * it, sig, dith) + 6755399441055744.0);
                    vq = neg ? mag : -mag;
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
                } else { // RISE (:196,:204) / logRISE Z (:279): V = -w exp(-E) s
                    const unsigned sb = ~sgn[i] >> (8 * g + j); // bit 0: s = +1
                    const unsigned dh = dh0 + (unsigned)(i * 32 + 8 * g + j) * 0x9E3779B9u;
                    if (UNIW) {
                        vq = vq_exp(Ea, sb, wk32, dh, etab);
                        if (i * 32 + 8 * g + j >= nreal) vq = 0; // padding samples carry no weight
                    } else {
                        vq = vq_exp(Ea, sb, 4294967296.0 * (w[kk + j] * it), dh, etab);
                    }
                    const int nvq = -vq, mag = vq > nvq ? vq : nvq;
                    mx = mag > mx ? mag : mx;
                    if (WANTF) as += mag;
                }
                cs += vq;
                dj[j] = ((unsigned)vq + 0x80808080u) ^ 0x80808080u; // 4 balanced base-256 digits
            }
            // 4 samples x 4 limbs byte transpose -> one dword per limb plane
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                const unsigned t01 = __builtin_amdgcn_perm(dj[1], dj[0], sel);
                const unsigned t23 = __builtin_amdgcn_perm(dj[3], dj[2], sel);
                pl[lb][g] = (int)__builtin_amdgcn_perm(t23, t01, 0x05040100u);
            }
        }
        if (active) {
#pragma unroll
            // (plain stores: the two 16-byte halves of a 64-byte row come from two instructions and merge in L2; as
            // non-temporal stores they reach HBM separately -- 8.2 GB written instead of 5.1 -- for 2 % less time)
            for (int lb = 0; lb < LB; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
        }
    }
    cs += __shfl_xor(cs, 32);
    as += __shfl_xor(as, 32);
    if (active && h == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs);
        if (WANTF) atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), (unsigned long long)as);
    }
    if (form == 0) {
        const int mo = __shfl_xor(mx, 32);
        mx = mo > mx ? mo : mx;
        if (active && h == 0) atomicMax(&mmax[r], (unsigned)mx);
    }
    if (form == 2) {
        fp += __shfl_xor(fp, 32);
        if (active && h == 0) unsafeAtomicAdd(&fsum[r], fp);
    }
}

template <int LF, int FORM, bool WANTF, bool WIDE, bool UNIW, bool COARSE>
static void launch_fwd5(const FwdLaunch &a) {
    constexpr int STAGE = 2 * (2 + 2 * LF) * 1024; // two 64-column steps per ring stage, three stages
    constexpr int shmem = 3 * STAGE + 512 + 1024;   // ring + exp, log tables
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd_i8<LF, FORM, WANTF, WIDE, COARSE, UNIW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem); // per device: set on every launch
    const DevProblem &d = *a.d;
    const int ntk = a.ntk;
    const int grid = ((ntk + 7) / 8) * 8 * a.ngroups; // one workgroup per (sample tile, node tile); see the kernel's block mapping
    hipLaunchKernelGGL((k_fwd_i8<LF, FORM, WANTF, WIDE, COARSE, UNIW>), dim3(grid), dim3(256), shmem, a.st, d.Xb, d.Sb, a.w->Tq, a.rowcol, a.groups,
                       a.ngroups, d.w, a.sc->sigma, a.sc->qconst, a.sc->invtau, d.Kp, ntk, a.zero_theta ? 0 : (int)(d.Qfp / 64), d.wuni, d.K, a.Vout,
                       a.sc->csum, a.sc->asum, a.F, a.sc->mmax, a.w->Vq, a.vmap, a.w->sc[0].tau, a.w->LBT, a.w->vpl0(), a.w->vscale(),
                       a.chunk_tiles, a.part_tiles, a.cc ? a.cc->cnk : nullptr, a.cc ? a.cc->Xc : nullptr, a.cc ? a.cc->xc_tile : 0, (int)(d.Qfp / 64));
}

template <int LF, int FORM, bool WANTF, bool WIDE, bool UNIW>
static void launch_fwd4(const FwdLaunch &a) {
    if constexpr (LF == 4 && FORM == 0) {
        if (a.coarse) return launch_fwd5<LF, FORM, WANTF, WIDE, UNIW, true>(a);
    }
    launch_fwd5<LF, FORM, WANTF, WIDE, UNIW, false>(a);
}

template <int LF, int FORM, bool WANTF, bool WIDE>
static void launch_fwd3(const FwdLaunch &a) {
    if (a.d->wuni > 0.0) launch_fwd4<LF, FORM, WANTF, WIDE, true>(a);
    else launch_fwd4<LF, FORM, WANTF, WIDE, false>(a);
}

template <int LF, int FORM, bool WANTF>
static void launch_fwd2(const FwdLaunch &a) {
    if (a.d->Qfp > 32768) launch_fwd3<LF, FORM, WANTF, true>(a);
    else launch_fwd3<LF, FORM, WANTF, false>(a);
}

template <int LF>
static void launch_fwd(const FwdLaunch &a, int form, bool wantf, int hv) {
    if (hv) {
        if (form == 2) launch_fwd2<LF, 4, false>(a);
        else launch_fwd2<LF, 3, false>(a);
    } else if (form == 2) launch_fwd2<LF, 2, true>(a);
    else if (wantf) launch_fwd2<LF, 0, true>(a);
    else launch_fwd2<LF, 0, false>(a);
}


void launch_fwd_i8(const FwdLaunch &fl, int LF, int form, bool wantf, int hv) {
    switch (LF) {
    case 2: // (Hessian-vector forms only)
        if (form == 2) launch_fwd2<2, 4, false>(fl);
        else launch_fwd2<2, 3, false>(fl);
        break;
    case 3: launch_fwd<3>(fl, form, wantf, hv); break;
    case 4: launch_fwd<4>(fl, form, wantf, hv); break;
    default: launch_fwd<5>(fl, form, wantf, hv);
    }
}

} // namespace gml
