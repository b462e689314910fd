// Int8-limb path, part 3: the backward (gradient) GEMM, the zeroing of a pass's accumulators and the finalisation of the i8x pass
// (overview: gml_i8.h).
#include "gml_i8.h"
#include <algorithm>
#include <string>
#include <type_traits>

namespace gml {

// ------------------------------------------------------------------------------------------
// backward: Gacc[m][c] += sum_k Vq[m][k] * b[k][c]  (i32, split-K with integer atomics), b = [x = -1]
// from the feature-major bit image; the gradient is tau * (sum_k V - 2 sum_l 256^l Gacc_l).
// Workgroup tile: 2 node tiles (256 rows of Vq: 4 limbs x 32 nodes each) x 256 columns,
// 8 waves as 2 (M) x 4 (N), each 128 x 64; 4-deep LDS-DMA ring of 18-KB stages (two 8-KB Vq images +
// 2 KB of bits), 2 waves/SIMD.  All (tile, column-tile) blocks of one k-chunk run on one XCD so that
// the chunk's slabs of Vq and of the bit image are fetched from HBM once and shared through that XCD's L2.
// ------------------------------------------------------------------------------------------
template <int TM /* node tiles per workgroup: 1 (4 waves, two workgroups per CU) or 2 (8 waves) */,
          int NL /* limb planes of Vq multiplied: 4; 2 for the products of a 2-limb Hessian-vector pass; 3 for one half of the
                    6 planes of the i8w pass (TM = 1); 6 for all of them in one launch (TM = 1: wave tile 192 x 64) */>
__global__ __launch_bounds__(256 * TM, 2) void k_bwd_i8(
    const int8_t *__restrict__ Vq, const unsigned *__restrict__ Xtb, const int *__restrict__ groups, int ngroups_t,
    int nNt, int64_t Qfp, int64_t Kp, int64_t kchunk, int nsplit, int32_t *__restrict__ Gacc,
    int chunks_per_plane /* split-K chunks that share one set of i32 accumulators (<= 2^24 configurations: |sum| < 2^31) */,
    int64_t plane_stride /* elements between the accumulator sets */,
    int64_t kpart /* configurations of every chunk that take part (== kchunk: all; less: sub-sampled Hessian-vector products) */,
    int lbt /* limb planes of a Vq image (and of the accumulator rows of a node tile) */, int pl0 /* first plane multiplied */) {
    constexpr int NW = 4 * TM;
    constexpr int AP = NL == 6 ? 12 : 8 * TM; // 1-KB pieces (16 rows x 64 B) of limb rows in a stage
    constexpr int AR = AP * 16, NPIECE = AP + 2, STAGE = NPIECE * 1024, NS = 4;
    constexpr int WMT = NL, WNT = 2; // wave tile 128 (192, 96, 64) x 64: MFMA tile i <-> limb plane pl0 + i of the node tile
    static_assert(NL == 4 || ((NL == 2 || NL == 3 || NL == 6) && TM == 1), "the 2-, 3- and 6-limb forms exist for 4-wave workgroups");
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    const int wm = wave % TM, wn = wave / TM; // wn in 0..3
    const int T = ngroups_t * nNt; // ngroups_t = number of TM-groups of node tiles
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int chunk = (bi / T) * 8 + xcd, ti = bi % T; // all tiles of one k-chunk on one XCD
    if (chunk >= nsplit) return;
    Gacc += (int64_t)(chunk / chunks_per_plane) * plane_stride;
    const int gi = ti / nNt, nt = ti % nNt;
    int tiles[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) tiles[t] = groups[gi * TM + t]; // -1: padding (computed on tile 0, not stored)
    const int64_t kb = (int64_t)chunk * kchunk;
    const int64_t ke = (kb + kpart < Kp) ? kb + kpart : Kp;
    const int64_t n0 = (int64_t)nt * 256, nkk = Kp >> 6, kt0 = kb >> 6;

    // 8*TM + 2 pieces over 4*TM waves: waves 0 and 1 load three (the third is a piece of bits), the others two.
    // NL = 2: only the four pieces of limb planes 0 and 1 (rows 0..63 of the image) + the bits: two per wave.
    // NL = 3: the six pieces of three planes + the bits: two per wave.
    // NL = 6: the twelve pieces of six planes + the bits: waves 0 and 1 load four, the others three.
    constexpr int NJ = NL == 6 ? 4 : 3;               // loads of the waves that carry one more
    const bool three = (NL == 4 || NL == 6) && wave < 2; // ... which are these
    const int8_t *src[NJ];
    int adv[NJ], dst[NJ];
    const int img = lbt * 2048; // bytes of a Vq image
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int pc = wave + NW * j; // piece of the full stage image: 0 .. AP - 1 rows of Vq, then the bits
        if (NL == 2) pc = j == 0 ? wave : 8 * TM + (wave & 1);
        if (NL == 3) pc = wave + 4 * j < 6 ? wave + 4 * j : 8 * TM + ((wave + 4 * j - 6) & 1);
        dst[j] = pc * 1024;
        if (pc < AP) {
            int tl = tiles[NL == 6 ? 0 : pc >> 3];
            if (tl < 0) tl = tiles[0];
            const int row = (NL == 6 ? pc : (pc & 7)) * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3);
            src[j] = Vq + ((int64_t)tl * nkk + kt0) * img + (pl0 * 32 + row) * 64 + slot * 16;
            adv[j] = img;
        } else {
            const int pb = pc < NPIECE ? pc - AP : 0;
            src[j] = reinterpret_cast<const int8_t *>(Xtb) + ((int64_t)(2 * nt + pb) * nkk + kt0) * 1024 + lane * 16;
            adv[j] = 1024;
        }
    }
    auto issue = [&](int kt) {
        int8_t *stage_base = lds + (kt & (NS - 1)) * STAGE;
#pragma unroll
        for (int j = 0; j < NJ - 1; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)kt * adv[j]), (lptr_t)(stage_base + dst[j]), 16, 0, 0);
        if (three) __builtin_amdgcn_global_load_lds((gptr_t)(src[NJ - 1] + (int64_t)kt * adv[NJ - 1]), (lptr_t)(stage_base + dst[NJ - 1]), 16, 0, 0);
    };
    v16i acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0;

    const int nk = (int)((ke - kb) / 64);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (three) ring_wait_ahead<NJ>(nk - 1 - kt);
        else ring_wait_ahead<NJ - 1>(nk - 1 - kt);
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        const int8_t *cur = lds + (kt & (NS - 1)) * STAGE;
        unsigned vb[WNT];
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn) {
            const int cw = wn * 64 + jn * 32 + lr;
            vb[jn] = *reinterpret_cast<const unsigned *>(cur + AR * 64 + (cw >> 7) * 1024 + (((cw & 127) * 2 + h) << 2));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            v4i fa[WMT], fb[WNT];
#pragma unroll
            for (int i = 0; i < WMT; ++i)
                fa[i] = *reinterpret_cast<const v4i *>(cur + lds_off(wm * 128 + i * 32 + lr, 2 * t + h));
#pragma unroll
            for (int jn = 0; jn < WNT; ++jn)
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[jn][e] = (int)((vb[jn] >> (4 * t + e)) & 0x01010101u);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int jn = 0; jn < WNT; ++jn) acc[i][jn] = MFMA_I8(fa[i], fb[jn], acc[i][jn]);
        }
    }
    // C layout: column (lane&31) <-> column c, register e <-> Vq row (e&3)+8*(e>>2)+4*h
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn) {
            const int64_t c = n0 + wn * 64 + jn * 32 + lr;
            const int grow = wm * 128 + i * 32; // first row of this MFMA tile within the workgroup tile
            const int tl = tiles[NL == 6 ? 0 : grow >> 7];
            if (c < Qfp && tl >= 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mrow = (NL == 6 ? grow : (grow & 127)) + (e & 3) + 8 * (e >> 2) + 4 * h;
                    atomicAdd(&Gacc[((int64_t)tl * lbt * 32 + pl0 * 32 + mrow) * Qfp + c], acc[i][jn][e]);
                }
            }
        }
}

// G[row][c] = tau_r * (csum[r] - 2 sum_l 256^l Gacc[(t*4+l)*32+rl][c])  (x = 1 - 2b);  G[row][cconst] = tau_r * csum[r];
// f[r] = tau_r * asum[r]  (= sum_k w_k exp(-E) for RISE / logRISE; RPLE keeps its FP64 sum).  r = slot, row = srow[r].
__global__ __launch_bounds__(256) void k_finalize_i8(const int32_t *__restrict__ Gacc, const double *__restrict__ tau,
                                                     const long long *__restrict__ csum,
                                                     const long long *__restrict__ asum, const int *__restrict__ srow,
                                                     const int *__restrict__ rowcol, int slot0, int64_t Qp, int64_t Qfp, int64_t Qf,
                                                     int64_t cconst, int form, int want_grad, int hv,
                                                     double *__restrict__ G, double *__restrict__ f, int nplanes,
                                                     int64_t plane_stride, const unsigned *__restrict__ mmax,
                                                     SlotResult *__restrict__ res) {
    const int r = slot0 + blockIdx.y;
    if (rowcol[r] < 0) return;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double t = tau[r];
    if (c == 0) {
        double fv = f ? f[r] : 0.0; // RPLE: the forward kernel's FP64 sum
        if (form != 2 && !hv) {
            if (want_grad) { // f = sum_k w exp(-E) = -sum_k V_k s_k = -G[r][u] (u = the node's own, masked, column)
                const int tile = r >> 5, rl = r & 31, u = rowcol[r];
                long long s = 0;
#pragma unroll
                for (int l = LB - 1; l >= 0; --l) {
                    long long a = 0;
                    for (int pl = 0; pl < nplanes; ++pl) a += (long long)Gacc[pl * plane_stride + ((int64_t)(tile * LB + l) * 32 + rl) * Qfp + u];
                    s = s * 256 + a;
                }
                fv = -t * (double)(csum[r] - 2 * s);
            } else {
                fv = t * (double)asum[r];
            }
            f[r] = fv;
        }
        if (res) res[r] = SlotResult{fv, t, mmax[r], 0u};
    }
    if (!want_grad || c >= Qp) return;
    double v = 0.0;
    if (c < Qf) {
        const int tile = r >> 5, rl = r & 31;
        long long s = 0;
#pragma unroll
        for (int l = LB - 1; l >= 0; --l) {
            long long a = 0;
            for (int pl = 0; pl < nplanes; ++pl) a += (long long)Gacc[pl * plane_stride + ((int64_t)(tile * LB + l) * 32 + rl) * Qfp + c];
            s = s * 256 + a;
        }
        v = t * (double)(csum[r] - 2 * s);
    } else if (c == cconst) {
        v = t * (double)csum[r];
    }
    G[(int64_t)srow[r] * Qp + c] = v;
}


// one launch zeroes every accumulator of a pass over the slots [slot0, slot0 + ns): slot sums, maxima, f, and the i32
// gradient planes of those slots' tiles
__global__ __launch_bounds__(256) void k_zero_pass(long long *__restrict__ csum, long long *__restrict__ asum,
                                                   long long *__restrict__ csum2, long long *__restrict__ asum2,
                                                   unsigned *__restrict__ mmax, double *__restrict__ f, const int *__restrict__ rowcol,
                                                   int slot0, int ns, v4i *__restrict__ gacc, int64_t ngacc, int nplanes,
                                                   int64_t plane_stride4) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    // (only the slots this pass evaluates: a re-run of some rows of a tile leaves the others' planes in place, and with them the
    // scalars that describe those planes -- the working-set Hessians read tau and the largest |V| of a row's LAST pass)
    if (i0 < ns && rowcol[slot0 + i0] >= 0) {
        csum[slot0 + i0] = 0;
        asum[slot0 + i0] = 0;
        csum2[slot0 + i0] = 0;
        asum2[slot0 + i0] = 0;
        mmax[slot0 + i0] = 0;
        if (f) f[slot0 + i0] = 0.0;
    }
    const v4i z = {0, 0, 0, 0};
    for (int pl = 0; pl < nplanes; ++pl)
        for (int64_t i = i0; i < ngacc; i += stride) gacc[pl * plane_stride4 + i] = z;
}


void launch_zero_pass(const SlotScalars &sc, double *F, const int *rowcol, int slot0, int ns, int32_t *gacc0, int64_t ngacc4, int nplanes,
                      int64_t plane_stride4, hipStream_t st) {
    hipLaunchKernelGGL(k_zero_pass, dim3(1024), dim3(256), 0, st, sc.csum, sc.asum, sc.csum2, sc.asum2, sc.mmax, F, rowcol, slot0, ns,
                       reinterpret_cast<v4i *>(gacc0), ngacc4, nplanes, plane_stride4);
}

void launch_bwd_i8(int NL, const int8_t *Vin, const DevProblem &d, const int *groups, int ngt, int nNt, int64_t kchunk, int nsplit, int32_t *Gacc,
                   int cpp, int64_t plane_stride, int64_t kpart, int lbt, int pl0, hipStream_t st) {
    constexpr int TM = 1; // node tiles per backward workgroup (the 8-wave form with two, TM = 2, measured slower)
    const int T = ngt * nNt;
    const int grid = ((nsplit + 7) / 8) * 8 * T;
    const int shmem = 4 * ((NL == 6 ? 12 : 8 * TM) + 2) * 1024;
#define BWD(NLV)                                                                                                                            \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd_i8<1, NLV>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);     \
        hipLaunchKernelGGL((k_bwd_i8<1, NLV>), dim3(grid), dim3(256), shmem, st, Vin, d.Xtb, groups, ngt, nNt, d.Qfp, d.Kp, kchunk, nsplit,  \
                           Gacc, cpp, plane_stride, kpart, lbt, pl0);                                                                        \
    } while (0)
    if (NL == 2) BWD(2);
    else if (NL == 3) BWD(3);
    else if (NL == 6) BWD(6);
    else BWD(4);
#undef BWD
}

void launch_finalize_i8(const int32_t *Gacc, const SlotScalars &sc, const int *srow, const int *rowcol, int slot0, int ns, const DevProblem &d,
                        int form, int want_grad, int hv, double *G, double *F, int nplanes, int64_t plane_stride, SlotResult *res, hipStream_t st) {
    hipLaunchKernelGGL(k_finalize_i8, dim3((unsigned)((d.Qp + 255) / 256), (unsigned)ns), dim3(256), 0, st, Gacc, sc.tau, sc.csum, sc.asum, srow,
                       rowcol, slot0, d.Qp, d.Qfp, d.Qf, d.cconst, form, want_grad, hv, G, F, nplanes, plane_stride, sc.mmax, res);
}

} // namespace gml
