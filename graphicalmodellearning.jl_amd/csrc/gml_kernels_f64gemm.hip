// The two GEMM-shaped kernels of the FP64 device path (precision "f64": the reference's own arithmetic, instruction for
// instruction, on v_mfma_f64_16x16x4_f64):
//   energies  A[r][k] = sum_c Theta[r][c] x[k][c]   + the pointwise epilogue      (GraphicalModelLearning.jl:162, :170, :196)
//   gradient  G[r][c] = sum_k V[r][k] x[k][c]                                      (:205-207)
// Round-4 form.  The +-1 operand comes from the BIT images the int8 kernels use (Xb sample-major, Xtb feature-major: one dword per
// lane and 32-deep block, expanded to +-1.0 by two integer instructions per element) -- the two 1-GB byte images of rounds 1-3
// are gone from these kernels.  The waves of a workgroup share the FP64 operand and split the +-1 side: 4 waves = 32 rows x 512
// samples (forward) / 8 waves = 32 rows x 1024 columns (backward), wave tile 32 x 128 = 16 MFMA tiles, 128 accumulator registers.  With
// every column of a row group in ONE workgroup the backward kernel reads V from HBM once per pass (8.2 GB at the headline size;
// the round-1 kernel re-read it once per 256-column block: 61 GB fetched per launch), and a 32-deep block carries 128 MFMAs of
// 64 cycles each per wave against 16 + 8 loads: the kernels need no LDS staging, and a barrier only now and then.
#include "../../include/gml.h"
#include "gml_dev.h"
#include <algorithm>

namespace gml {

typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

__device__ __forceinline__ void ld8(const double *p, double (&d)[8]) {
    const double4 *q = reinterpret_cast<const double4 *>(p);
    const double4 a = q[0], b = q[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
// bit `sh` of w set (spin -1) -> -1.0, else +1.0
__device__ __forceinline__ double pm1(unsigned w, int sh) { return __hiloint2double((int)(((w >> sh) << 31) | 0x3FF00000u), 0); }

constexpr int NT = 8; // 16-wide MFMA tiles of the +-1 side per wave (wave tile 32 x 128)

} // namespace

// ------------------------------------------------------------------------------------------
// forward.  grid = (Kp / 512, listed row groups); workgroup = 4 waves x 128 samples of one 32-row group.
// Lane (li = lane & 15, q = lane >> 4) feeds MFMA step s of a 32-column block with the element at contraction index 8 q + s
// of both operands: Theta[row li][.. + 8 q + s] (8 consecutive doubles) and, from the lane's dword (sample li, step kt,
// half h = q >> 1) of the Xb image, the bit of column 64 kt + 32 g + 8 q + s: bit 4 g + 2 (q & 1) + (s >> 2) + 8 (s & 3)
// (gml_bits.h: xb_col).  The constant statistic (column cconst, x = 1) is added in the epilogue.
// ------------------------------------------------------------------------------------------
// Forward workgroups are 4 waves (32 rows x 512 samples), two per CU: the epilogue of one -- 64 exp per lane and the V stores, with
// no matrix work -- then runs beside the other's GEMM loop (8-wave workgroups, one per CU: 36.7 ms; 4-wave: 32.7 ms).
#ifndef F64_FWD_WAVES
#define F64_FWD_WAVES 4
#endif
__global__ __launch_bounds__(64 * F64_FWD_WAVES, 8 / F64_FWD_WAVES) void k_fwd_f64(const double *__restrict__ Theta, const unsigned *__restrict__ Xb,
                                                    const unsigned *__restrict__ Sb, const int *__restrict__ rowcol,
                                                    const int *__restrict__ groups, const double *__restrict__ w, int64_t Qp,
                                                    int64_t cconst, int64_t Kp, int nk, int form, double *__restrict__ V,
                                                    double *__restrict__ fsum) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, q = lane >> 4;
    const int grp = groups[blockIdx.y]; // 32-row group of this workgroup (-1: padding of the list)
    if (grp < 0) return;
    const int r0 = grp * 32;
    const int64_t k0 = (int64_t)blockIdx.x * (128 * F64_FWD_WAVES) + wave * 128; // this wave's 128 samples: one 128-sample piece of Xb
    if (k0 >= Kp) return;

    v4d acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (v4d){0, 0, 0, 0};

    const double *arow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) arow[mi] = Theta + (int64_t)(r0 + 16 * mi + li) * Qp + 8 * q;
    // dword (sample k0 + 16 ni + li, step kt, half q >> 1): piece (k0 >> 7), 256 dwords per step
    const unsigned *bp = Xb + ((k0 >> 7) * nk * 128 + li) * 2 + (q >> 1);
    const int qs = 2 * (q & 1);

    for (int kt = 0; kt < nk; ++kt) {
        unsigned bw[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bw[ni] = bp[(int64_t)kt * 256 + ni * 32] >> qs;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            double a[2][8];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) ld8(arow[mi] + 64 * kt + 32 * g, a[mi]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                double bd[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) bd[ni] = pm1(bw[ni], 4 * g + (s >> 2) + 8 * (s & 3));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = MFMA_F64(a[mi][s], bd[ni], acc[mi][ni]);
            }
        }
    }

    // epilogue: lane holds C[r = r0 + 16 mi + q + 4 j][k = k0 + 16 ni + li]
    const int64_t wpr = Kp >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + 16 * mi + q + 4 * j;
            const int rc = rowcol[r];
            const double tc = rc >= 0 ? Theta[(int64_t)r * Qp + cconst] : 0.0;
            double fpart = 0.0;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int64_t k = k0 + 16 * ni + li;
                if (rc >= 0) {
                    const double wk = w[k];
                    const double s = ((Sb[(int64_t)rc * wpr + (k >> 5)] >> (k & 31)) & 1u) ? -1.0 : 1.0; // the node's own spin
                    const double E = s * (acc[mi][ni][j] + tc);
                    double val;
                    if (form >= 4) {
                        // Hessian-vector forward (forms 4..6; Theta holds the direction p): U_k = h_k (acc + tc), acc + tc = sum_j p_j x_kj
                        // over the raw columns -- the backward GEMM contracts U with the same raw columns, which gives H p since s^2 = 1.
                        // The curvature weight h_k comes from the V the objective pass at theta left in this row: RISE / logRISE Z:
                        // V = -w e^-E s, h = w e^-E = -V s; RPLE: V = -2 w sigma(-2E) s, h = 4 w sigma(-2E) sigma(2E).
                        const double vold = V[(int64_t)r * Kp + k];
                        double h = -vold * s;
                        if (form == 6) {
                            const double sg = wk > 0.0 ? h / (2.0 * wk) : 0.0; // sigma(-2E)
                            h = 2.0 * h * (1.0 - sg);
                        }
                        val = h * (E * s); // (E s = acc + tc)
                        V[(int64_t)r * Kp + k] = val;
                        continue;
                    }
                    if (form == 2) { // RPLE (:317)
                        const double t = -2.0 * E;
                        const double sp = t > 0 ? t + log1p(exp(-t)) : log1p(exp(t));
                        const double sg = 1.0 / (1.0 + exp(2.0 * E));
                        fpart += wk * sp;
                        val = -2.0 * wk * sg * s; // d/dE of w log(1+exp(-2E)), times s
                    } else { // RISE (:196) / logRISE Z (:279)
                        const double e = wk * exp(-E);
                        fpart += e;
                        val = -e * s; // partial_obj (:204) times the node's sign
                    }
                    V[(int64_t)r * Kp + k] = val; // (inactive rows keep their previous V)
                }
            }
            // reduce over the 16 lanes sharing this row (li = 0..15)
            fpart += __shfl_xor(fpart, 1);
            fpart += __shfl_xor(fpart, 2);
            fpart += __shfl_xor(fpart, 4);
            fpart += __shfl_xor(fpart, 8);
            if (li == 0 && rc >= 0) unsafeAtomicAdd(&fsum[r], fpart);
        }
    }
}

void launch_fwd_f64(const DevProblem &P, const double *Theta, const int *rowcol, const int *groups, int ngroups4, int form, double *V,
                    double *fsum, hipStream_t st) {
    dim3 grid((unsigned)((P.Kp + 128 * F64_FWD_WAVES - 1) / (128 * F64_FWD_WAVES)), (unsigned)ngroups4);
    hipLaunchKernelGGL(k_fwd_f64, grid, dim3(64 * F64_FWD_WAVES), 0, st, Theta, P.Xb, P.Sb, rowcol, groups, P.w, P.Qp, P.cconst, P.Kp, (int)(P.Qfp / 64), form,
                       V, fsum);
}

// ------------------------------------------------------------------------------------------
// backward.  grid = (ceil(Qfp / 1024), row groups, split-K chunks); workgroup = 8 waves x 128 columns of one 32-row group over
// one chunk of the samples; partial sums are combined with f64 atomics.  Lane (li, q) feeds step s of a 32-sample block with
// V[row li][k + 8 q + s] and, from the lane's dword (column li, block) of the Xtb image, the bit of sample k + 8 q + s:
// bit q + 4 (s >> 2) + 8 (s & 3) (gml_bits.h: xtb_from_natural).  The constant statistic (x = 1 on every real sample; V is 0 on
// the padding ones) is the plain row sum of V: wave 0 of the first column block adds it up on the side.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void k_bwd_f64(const double *__restrict__ V, const unsigned *__restrict__ Xtb,
                                                    const int *__restrict__ groups, int64_t Qp, int64_t Qf, int64_t Qc /* columns of the image */,
                                                    int64_t cconst, int64_t Kp, int64_t kchunk, double *__restrict__ G) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, q = lane >> 4;
    const int r0 = groups[blockIdx.y] * 32;
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + wave * 128;
    if (c0 >= Qf) return;
    const int64_t kb = (int64_t)blockIdx.z * kchunk;
    const int64_t ke = (kb + kchunk < Kp) ? kb + kchunk : Kp;
    const int64_t nkk = Kp >> 6;
    const bool sums = blockIdx.x == 0 && wave == 0;

    v4d acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (v4d){0, 0, 0, 0};
    double rs[2] = {0.0, 0.0};

    const double *arow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) arow[mi] = V + (int64_t)(r0 + 16 * mi + li) * Kp + 8 * q;
    // dword (column c, 64-sample step kt, half h) of the image: pieces [Qc / 128][Kp / 64][128 columns][2]
    const unsigned *bp[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        int64_t c = c0 + 16 * ni + li;
        if (c >= Qc) c = Qc - 1; // (beyond the image: any column, the result is not stored)
        bp[ni] = Xtb + ((c >> 7) * nkk * 128 + (c & 127)) * 2;
    }

    for (int64_t k = kb; k < ke; k += 32) {
        // the 8 waves read the same V: kept within 32 blocks of each other (one barrier per 4096 MFMAs of a wave), the seven
        // followers hit in L2 (-1.7 % on the kernel; every 8 blocks: +0.5 %)
        if ((((k - kb) >> 5) & 31) == 31) __builtin_amdgcn_s_barrier();
        const int64_t boff = (k >> 6) * 256 + ((k >> 5) & 1);
        unsigned bw[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bw[ni] = bp[ni][boff] >> q;
        double a[2][8];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) ld8(arow[mi] + k, a[mi]);
        if (sums) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int s = 0; s < 8; ++s) rs[mi] += a[mi][s];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double bd[NT];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) bd[ni] = pm1(bw[ni], 4 * (s >> 2) + 8 * (s & 3));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = MFMA_F64(a[mi][s], bd[ni], acc[mi][ni]);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int64_t c = c0 + 16 * ni + li;
            if (c < Qf) {
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(&G[(int64_t)(r0 + 16 * mi + q + 4 * j) * Qp + c], acc[mi][ni][j]);
            }
        }
    if (sums) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            double v = rs[mi];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) unsafeAtomicAdd(&G[(int64_t)(r0 + 16 * mi + li) * Qp + cconst], v);
        }
    }
}

void launch_bwd_f64(const DevProblem &P, const double *V, const int *groups, int ngroups, double *G, hipStream_t st) {
    const unsigned gx = (unsigned)((P.Qf + 1023) / 1024 > 0 ? (P.Qf + 1023) / 1024 : 1), gy = (unsigned)ngroups;
    // one 8-wave workgroup per CU at a time: about two rounds of workgroups over the 256 CUs, in whole 1024-sample chunks
    int64_t nsplit = (512 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t maxsplit = P.Kp / 2048 > 0 ? P.Kp / 2048 : 1;
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    int64_t kchunk = (P.Kp + nsplit - 1) / nsplit;
    kchunk = (kchunk + 63) / 64 * 64;
    nsplit = (P.Kp + kchunk - 1) / kchunk;
    dim3 grid(gx, gy, (unsigned)nsplit);
    const int64_t Qc = (P.Qfp + 255) / 256 * 256;
    hipLaunchKernelGGL(k_bwd_f64, grid, dim3(512), 0, st, V, P.Xtb, groups, P.Qp, P.Qf, Qc, P.cconst, P.Kp, kchunk, G);
}

} // namespace gml
