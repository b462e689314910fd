// FP64 device path of the learn() hot path: packing kernels, the working-set Hessian on FP64 MFMA, the batched Newton /
// preconditioner solves.  (The two GEMM-shaped kernels -- energies + pointwise, gradient -- live in gml_kernels_f64gemm.hip.)
// gfx950 only.
//
// Math restated from /root/reference/src/GraphicalModelLearning.jl:
//   energy   E_rk = s_u^k * sum_c Theta[r][c] * X[k][c]          (:162 + :170 inner sum)
//   RISE     f = sum_k w_k exp(-E)          partial_k = -w_k exp(-E)            (:196, :204)
//   logRISE  Z = sum_k w_k exp(-E) (f = log Z, done on the host side)           (:279)
//   RPLE     f = sum_k w_k log(1+exp(-2E))                                      (:317)
//   gradient g[c] = sum_k stat[k,c] * partial_k                                 (:205-207)
// All three kernels are "NT" products C[i][j] = sum_t A[i][t] * B[j][t] on
// v_mfma_f64_16x16x4_f64 with the +-1 operand converted from int8 in registers; operands go
// straight from global memory / L2 to VGPRs (one f64 MFMA takes 64 cycles per SIMD, which
// leaves ample time for the few loads per step), no LDS.
#include "../../include/gml.h"
#include "gml_dev.h"
#include <algorithm>

namespace gml {

typedef double v4d __attribute__((ext_vector_type(4)));

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_i8(const int8_t *__restrict__ src, int64_t rows,
                                                      int64_t cols, int64_t ld_src,
                                                      int8_t *__restrict__ dst, int64_t ld_dst) {
    __shared__ int8_t tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[r * ld_src + c] : (int8_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int64_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[c * ld_dst + r] = tile[tx][i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_convert_hist(const T *__restrict__ H, int64_t K, int64_t n, int64_t ld, int col_major,
                                                      double *__restrict__ counts, int8_t *__restrict__ spins,
                                                      long long *__restrict__ bad) {
    // one thread per element, the fastest-varying source index on threadIdx: both reads and writes coalesce
    const int64_t total = K * (n + 1), stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        int64_t k, j;
        if (col_major) {
            j = e / K;
            k = e - j * K;
        } else {
            k = e / (n + 1);
            j = e - k * (n + 1);
        }
        const double v = (double)H[col_major ? k + j * ld : k * ld + j];
        if (j == 0) {
            counts[k] = v;
        } else {
            int8_t sv = 0;
            if (v == 1.0) sv = 1;
            else if (v == -1.0) sv = -1;
            else atomicMin(reinterpret_cast<unsigned long long *>(bad), (unsigned long long)k);
            spins[col_major ? (j - 1) * K + k : k * n + (j - 1)] = sv;
        }
    }
}

void launch_convert_hist(const void *H, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, double *counts, int8_t *spins,
                         long long *bad, hipStream_t st) {
    const dim3 grid(8192), block(256);
    switch (dtype) {
    case GML_I8: hipLaunchKernelGGL(k_convert_hist<int8_t>, grid, block, 0, st, (const int8_t *)H, K, n, ld, col_major, counts, spins, bad); break;
    case GML_I32: hipLaunchKernelGGL(k_convert_hist<int32_t>, grid, block, 0, st, (const int32_t *)H, K, n, ld, col_major, counts, spins, bad); break;
    case GML_I64: hipLaunchKernelGGL(k_convert_hist<long long>, grid, block, 0, st, (const long long *)H, K, n, ld, col_major, counts, spins, bad); break;
    default: hipLaunchKernelGGL(k_convert_hist<double>, grid, block, 0, st, (const double *)H, K, n, ld, col_major, counts, spins, bad);
    }
}

__global__ __launch_bounds__(256) void k_check_pm1(const int8_t *__restrict__ S, int64_t total, int64_t n,
                                                   long long *__restrict__ bad) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int8_t v = S[i];
        if (v != 1 && v != -1) {
            // first offender: bad starts at -1, so compare in the unsigned order (-1 = largest)
            atomicMin(reinterpret_cast<unsigned long long *>(bad), (unsigned long long)(i / n));
        }
    }
}

void launch_check_pm1(const int8_t *S, int64_t K, int64_t n, long long *bad, hipStream_t st) {
    hipLaunchKernelGGL(k_check_pm1, dim3(4096), dim3(256), 0, st, S, K * n, n, bad);
}

void launch_transpose_i8(const int8_t *src, int64_t rows, int64_t cols, int64_t ld_src, int8_t *dst,
                         int64_t ld_dst, hipStream_t st) {
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(k_transpose_i8, grid, dim3(256), 0, st, src, rows, cols, ld_src, dst, ld_dst);
}

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8d(const double *p, double (&d)[8]) {
    const double4 *q = reinterpret_cast<const double4 *>(p);
    double4 a = q[0], b = q[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}
__device__ __forceinline__ void load8b(const int8_t *p, double (&d)[8]) {
    // 8 consecutive int8 -> 8 doubles
    uint2 u = *reinterpret_cast<const uint2 *>(p);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        d[s] = (double)(int)(int8_t)((u.x >> (8 * s)) & 0xff);
        d[4 + s] = (double)(int)(int8_t)((u.y >> (8 * s)) & 0xff);
    }
}

// ------------------------------------------------------------------------------------------
// working-set Hessian: H_r[i][j] += sum_k h_rk * Xt[F_ri][k] * Xt[F_rj][k]  for the lower
// triangular 32x32 tiles of row r's working set.  h_rk is the second-derivative weight:
//   RISE / logRISE(Z): h = w_k exp(-E) = V*s ;   RPLE: h = 4 w sg (1-sg), sg = (V*s)/(2w).
// One wave per (row, tile pair, k-split); grid = (ceil(nsplit/4), maxpairs, R).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hess_f64(const double *__restrict__ V,
                                                  const int8_t *__restrict__ Xt,
                                                  const double *__restrict__ w,
                                                  const int *__restrict__ rowcol,
                                                  const int *__restrict__ F, const int *__restrict__ mt,
                                                  const long long *__restrict__ hoff, int cap, int64_t Kp,
                                                  int64_t Kh, int64_t kchunk, int64_t kstride, int nsplit, int form,
                                                  double *__restrict__ H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, q = lane >> 4;
    const int r = blockIdx.z;
    const int m = mt[r];
    const int pair = blockIdx.y;
    if (pair >= m * (m + 1) / 2) return;
    const int ks = blockIdx.x * 4 + wave;
    if (ks >= nsplit) return;
    int ti = (int)((sqrtf(8.0f * pair + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= pair) ++ti;
    while (ti * (ti + 1) / 2 > pair) --ti;
    const int tj = pair - ti * (ti + 1) / 2;
    const int64_t kb = (int64_t)ks * kchunk;
    // Kh configurations of a compact index whose block cb of 512 stands for the samples [512 cb kstride, +512)
    const int64_t ke = (kb + kchunk < Kh) ? kb + kchunk : Kh;
    const int rc = rowcol[r];
    if (rc < 0) return;

    const int *Fr = F + (int64_t)r * cap;
    const int8_t *arow[2], *brow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) arow[mi] = Xt + (int64_t)Fr[ti * 32 + 16 * mi + li] * Kp + 8 * q;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) brow[ni] = Xt + (int64_t)Fr[tj * 32 + 16 * ni + li] * Kp + 8 * q;
    const double *vrow = V + (int64_t)r * Kp + 8 * q;
    const int8_t *srow = Xt + (int64_t)rc * Kp + 8 * q;
    const double *wrow = w + 8 * q;

    v4d acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (v4d){0, 0, 0, 0};

    for (int64_t tc = kb; tc < ke; tc += 32) {
        const int64_t t0 = (tc >> 9) * kstride * 512 + (tc & 511);
        double h[8], sg[8], a[2][8], b[2][8];
        load8d(vrow + t0, h);
        load8b(srow + t0, sg);
#pragma unroll
        for (int s = 0; s < 8; ++s) h[s] *= -sg[s]; // |V| = w |phi'|
        if (form == 2) {
            double wk[8];
            load8d(wrow + t0, wk);
#pragma unroll
            for (int s = 0; s < 8; ++s) h[s] = wk[s] > 0 ? 2.0 * h[s] * (1.0 - h[s] / (2.0 * wk[s])) : 0.0;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) load8b(arow[mi] + t0, a[mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) load8b(brow[ni] + t0, b[ni]);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = MFMA_F64(a[mi][s] * h[s], b[ni][s], acc[mi][ni]);
    }
    double *Hr = H + hoff[r]; // row r's block: (32 m) x (32 m), pitch 32 m
    const int hp = 32 * m;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = ti * 32 + 16 * mi + q + 4 * j;
                const int jj = tj * 32 + 16 * ni + li;
                unsafeAtomicAdd(&Hr[(int64_t)i * hp + jj], acc[mi][ni][j]);
            }
}

void launch_hess_f64(const DevProblem &P, const double *V,
                     const int *rowcol, const int *F, const int *mt, const long long *hoff, int R, int cap, int form,
                     int64_t Kh, int64_t kstride, double *H, hipStream_t st) {
    const int tiles = cap / 32;
    const int maxpairs = tiles * (tiles + 1) / 2;
    int64_t nsplit = (8192 + (int64_t)R * maxpairs - 1) / ((int64_t)R * maxpairs);
    const int64_t maxsplit = Kh / 512 > 0 ? Kh / 512 : 1;
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    int64_t kchunk = (Kh + nsplit - 1) / nsplit;
    kchunk = (kchunk + 31) / 32 * 32;
    nsplit = (Kh + kchunk - 1) / kchunk;
    dim3 grid((unsigned)((nsplit + 3) / 4), (unsigned)maxpairs, (unsigned)R);
    hipLaunchKernelGGL(k_hess_f64, grid, dim3(256), 0, st, V, P.Xt, P.w, rowcol, F, mt, hoff, cap, P.Kp, Kh, kchunk,
                       kstride, (int)nsplit, form, H);
}


// ------------------------------------------------------------------------------------------
// Inverse of a preconditioner tile (see gml_solver.hip, Newton-CG): A = sc * H_t - s2 g g^T on the tile's first m entries,
// Cholesky A = L L^T in LDS (left-looking, one thread per row), L^-1 into the upper triangle (thread j owns column j of
// L^-1, kept as row j above the diagonal), A^-1 = L^-T L^-1 written over the tile as a full symmetric matrix.
// ------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(256) void k_tile_inverse(double *__restrict__ H, const long long *__restrict__ hoff, const int *__restrict__ vm,
                                                      const int *__restrict__ wrow, const double *__restrict__ s1, double s2,
                                                      const double *__restrict__ gV) {
    constexpr int LP = T + 1;
    const int64_t v = blockIdx.x;
    const int m = vm[v], tid = threadIdx.x;
    if (m == 0) return;
    double *A = H + hoff[v];
    const double sc = s1[wrow[v]];
    extern __shared__ double sm[]; // L [T][T + 1] | gg [T] | dinv [T] | adiag [T]
    double *L = sm, *gg = sm + T * LP, *dinv = gg + T, *adiag = dinv + T;
    __shared__ double red[4], pivot;
    __shared__ int bad;
    if (tid < T) gg[tid] = (s2 != 0.0 && tid < m) ? gV[v * T + tid] : 0.0;
    __syncthreads();
    double dmax = 0.0;
    if (tid < m) {
        adiag[tid] = sc * A[(int64_t)tid * T + tid] - s2 * gg[tid] * gg[tid];
        dmax = fabs(adiag[tid]);
    }
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
    if ((tid & 63) == 0) red[tid >> 6] = dmax;
    __syncthreads();
    dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    double ridge = 0.0;
    bool ok = false;
    for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
        for (int idx = tid; idx < m * T; idx += 256) {
            const int i = idx / T, j = idx % T;
            if (j <= i) L[i * LP + j] = sc * A[(int64_t)i * T + j] - s2 * gg[i] * gg[j] + (i == j ? ridge : 0.0);
        }
        if (tid == 0) bad = 0;
        __syncthreads();
        for (int c = 0; c < m; ++c) {
            double x = 0.0;
            if (tid >= c && tid < m) {
                double x0 = L[tid * LP + c], x1 = 0.0, x2 = 0.0, x3 = 0.0;
                int k = 0;
                for (; k + 3 < c; k += 4) {
                    x0 = fma(-L[tid * LP + k], L[c * LP + k], x0);
                    x1 = fma(-L[tid * LP + k + 1], L[c * LP + k + 1], x1);
                    x2 = fma(-L[tid * LP + k + 2], L[c * LP + k + 2], x2);
                    x3 = fma(-L[tid * LP + k + 3], L[c * LP + k + 3], x3);
                }
                for (; k < c; ++k) x0 = fma(-L[tid * LP + k], L[c * LP + k], x0);
                x = (x0 + x1) + (x2 + x3);
                if (tid == c) pivot = x;
            }
            __syncthreads();
            const double piv = pivot;
            if (!(piv > 1e-12 * dmax) || !isfinite(piv)) { // (uniform: every thread reads the same pivot)
                if (tid == 0) bad = 1;
                break;
            }
            const double dgc = sqrt(piv);
            if (tid == c) {
                L[c * LP + c] = dgc;
                dinv[c] = 1.0 / dgc;
            } else if (tid > c && tid < m) {
                L[tid * LP + c] = x / dgc;
            }
            __syncthreads();
        }
        __syncthreads();
        ok = !bad;
        __syncthreads();
        if (!ok) ridge = ridge == 0.0 ? 1e-10 * fmax(dmax, 1e-300) : ridge * 100.0;
    }
    if (!ok) { // not positive definite (duplicate statistics): the diagonal
        for (int idx = tid; idx < m * T; idx += 256) {
            const int i = idx / T, j = idx % T;
            if (j < m) A[(int64_t)i * T + j] = i == j ? 1.0 / fmax(adiag[i], 1e-300) : 0.0;
        }
        return;
    }
    // column j of L^-1 by forward substitution, stored as row j right of the diagonal: U[j][i] = (L^-1)[i][j], i > j
    if (tid < m) {
        const int j = tid;
        const double dj = dinv[j];
        for (int i = j + 1; i < m; ++i) {
            double s0 = L[i * LP + j] * dj, s1 = 0.0;
            int k = j + 1;
            for (; k + 1 < i; k += 2) {
                s0 = fma(L[i * LP + k], L[j * LP + k], s0);
                s1 = fma(L[i * LP + k + 1], L[j * LP + k + 1], s1);
            }
            if (k < i) s0 = fma(L[i * LP + k], L[j * LP + k], s0);
            L[j * LP + i] = -(s0 + s1) * dinv[i];
        }
    }
    __syncthreads();
    // A^-1[i][j] = sum_{k >= i} (L^-1)[k][i] (L^-1)[k][j],  j <= i
    for (int idx = tid; idx < m * m; idx += 256) {
        const int i = idx / m, j = idx % m;
        if (j > i) continue;
        double a0 = dinv[i] * (i == j ? dinv[i] : L[j * LP + i]), a1 = 0.0;
        int k = i + 1;
        for (; k + 1 < m; k += 2) {
            a0 = fma(L[i * LP + k], L[j * LP + k], a0);
            a1 = fma(L[i * LP + k + 1], L[j * LP + k + 1], a1);
        }
        if (k < m) a0 = fma(L[i * LP + k], L[j * LP + k], a0);
        const double a = a0 + a1;
        A[(int64_t)i * T + j] = a;
        A[(int64_t)j * T + i] = a;
    }
}

void launch_tile_inverse(int T, double *H, const long long *hoff, const int *vm, const int *wrow, const double *s1, double s2, const double *gV,
                         int64_t ntiles, hipStream_t st) {
    if (ntiles <= 0) return;
    const size_t lds = sizeof(double) * ((size_t)T * (T + 1) + 3 * T);
    if (T == 64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_inverse<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_tile_inverse<64>, dim3((unsigned)ntiles), dim3(256), lds, st, H, hoff, vm, wrow, s1, s2, gV);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_inverse<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_tile_inverse<128>, dim3((unsigned)ntiles), dim3(256), lds, st, H, hoff, vm, wrow, s1, s2, gV);
    }
}


// value of lane l (wave-uniform index) of a double: two v_readlane_b32, the result in scalar registers.  (__shfl goes through
// ds_bpermute: an LDS round trip per value -- the 496 of a 32 x 32 diagonal block were most of the batched Cholesky kernels' time.)
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------
// Batched Newton solve on the device: for every row r, A d = -pg with
//   A = s1[r] * H_r  -  s2 * gF gF^T        (s2 = 1 for logRISE: Hess log Z = Hess Z / Z - g g^T)
// H_r is row r's ragged block (lower 32x32 tiles valid, pitch hp); Sdiag[r] = A[m-1][m-1] (the constant column, the Hessian's
// diagonal scale).  Blocked right-looking Cholesky (round 3).
//
// One workgroup per row.  A = s1 H - s2 g g^T is read from the lower triangle of the block and never written, so a ridge
// restart starts from it again; the factor goes to the strict upper triangle as U[k][i] = L[i][k] (row k = column k of L,
// contiguous in i), its diagonal to LDS.  Per panel of 32 columns:
//   (1) the 32 x 32 diagonal block -> LDS, factored by one wave, which also forward-solves the panel of the right-hand side;
//   (2) every row below the panel solves its 32 entries against that block (one thread per row, the entries in registers,
//       the block read from LDS as broadcasts) and leaves them in LDS (Lp) and in U;
//   (3) the trailing matrix is updated from Lp alone, in 4 x 4 register tiles spread over the 256 threads -- first panel:
//       read from the lower triangle, written to the upper; later panels: updated in place in the upper -- and so is the
//       rest of the right-hand side: the forward substitution costs no pass of its own.
// The back substitution walks the panels in reverse: 32 dot products of rows of U with the solved tail (coalesced, one wave
// per 8 rows), the diagonal block back into LDS, one wave for the 32 x 32 triangular solve.
// Why: rounds 1-2 had a left-looking panel Cholesky (every panel re-read all previous ones through L2, two barriers per 32 x 32
// block, the diagonal block factored by 496 dependent LDS inner products) and a one-wave LDS path for small blocks (m dependent
// column steps): 0.32 ms at 100 entries, 0.44 ms at 190, 0.68 ms at 250 for 128 rows (scripts/gpu_newton_ubench.py); this
// one: 0.13 / 0.28 / 0.45 ms, 0.03 ms at 30 entries.  What is left is latency: three dependent global round trips and one
// wave's 32 pivots per panel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_newton_chol(double *__restrict__ H, const long long *__restrict__ hoff, const int *__restrict__ mt,
                                                     const int *__restrict__ msz, const double *__restrict__ s1, double s2,
                                                     const double *__restrict__ gF, const double *__restrict__ pgF, int cap,
                                                     double *__restrict__ dout, double *__restrict__ Sdiag, int mcap /* >= every msz */,
                                                     // orthant faces (below): the working sets' columns, the iterates and their column
                                                     // kinds, the share of the predicted decrease that triggers a re-solve, the number
                                                     // of re-solves; F = NULL: none
                                                     const int *__restrict__ F, const double *__restrict__ X, const uint8_t *__restrict__ kind,
                                                     int64_t Qp, double share, int rounds,
                                                     // entries fixed from the start (tests): fix != 0 keeps the step dfix
                                                     const uint8_t *__restrict__ fix, const double *__restrict__ dfix,
                                                     int mlo /* blocks of up to mlo entries are another kernel's */) {
    constexpr int PW = 32, LP = PW + 1;
    const int r = blockIdx.x;
    const int m = msz[r];
    if (m == 0 || m <= mlo) return;
    const int hp = 32 * mt[r];
    double *A = H + hoff[r];
    const double sc = s1[r];
    const double *g = gF + (int64_t)r * cap, *pg = pgF + (int64_t)r * cap;
    extern __shared__ double sm[]; // dgw | dg | y | gg | fx | dfx [mcap each] | Ld [32][33] | tmp [32] | Lp [mcap - 32 (>= 32)][33]
    double *dgw = sm, *dg = sm + mcap, *y = sm + 2 * mcap, *gg = sm + 3 * mcap, *fx = sm + 4 * mcap, *dfx = sm + 5 * mcap, *Ld = sm + 6 * mcap,
           *tmp = Ld + PW * LP, *Lp = tmp + PW;
    __shared__ int bad;
    __shared__ double red[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool masked = fix != nullptr || F != nullptr; // some entries may be fixed
    for (int i = tid; i < m; i += 256) {
        gg[i] = s2 != 0.0 ? g[i] : 0.0;
        fx[i] = fix && fix[(int64_t)r * cap + i] ? 1.0 : 0.0;
        dfx[i] = fix ? dfix[(int64_t)r * cap + i] : 0.0;
    }
    __syncthreads();
    auto a_orig = [&](int i, int j) { return sc * A[(int64_t)i * hp + j] - s2 * gg[i] * gg[j]; }; // i >= j: the matrix itself
    auto a_low = [&](int i, int j) { // ... with the fixed entries decoupled (unit diagonal)
        if (masked && (fx[i] != 0.0 || fx[j] != 0.0)) return i == j ? 1.0 : 0.0;
        return a_orig(i, j);
    };
    double dmax = 0;
    for (int i = tid; i < m; i += 256) dmax = fmax(dmax, fabs(a_low(i, i)));
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
    if (lane == 0) red[wave] = dmax;
    __syncthreads();
    dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (tid == 0) Sdiag[r] = a_orig(m - 1, m - 1);
    for (int face = 0;; ++face) { // (re-solves on an orthant face, see the end of the loop)
    double ridge = 0.0;
    bool ok = false;
    for (int attempt = 0; attempt < 10 && !ok; ++attempt) {
        for (int i = tid; i < m; i += 256) {
            dgw[i] = a_low(i, i) + (fx[i] != 0.0 ? 0.0 : ridge);
            double v = -pg[i];
            if (masked) {
                if (fx[i] != 0.0) {
                    v = dfx[i];
                } else {
                    for (int j = 0; j < m; ++j)
                        if (fx[j] != 0.0 && dfx[j] != 0.0) v -= (i >= j ? a_orig(i, j) : a_orig(j, i)) * dfx[j];
                }
            }
            y[i] = v;
        }
        if (tid == 0) bad = 0;
        __syncthreads();
        for (int c0 = 0; c0 < m; c0 += PW) {
            const int pw = m - c0 < PW ? m - c0 : PW;
            const bool first = c0 == 0;
            // (1) diagonal block -> Ld (lower part; the working diagonal from dgw)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q, c = e >> 5, k = e & 31;
                double v = 0.0;
                if (c < pw && k < c) v = first ? a_low(c0 + c, c0 + k) : A[(int64_t)(c0 + k) * hp + c0 + c];
                else if (c < pw && k == c) v = dgw[c0 + c];
                Ld[c * LP + k] = v;
            }
            __syncthreads();
            if (wave == 0) {
                // lane t holds row t of the block in registers; right-looking: column c is scaled, then every lane updates
                // the rest of its row with L[k][c] shuffled in from lane k (independent FMAs: the dependent chain is the 32
                // pivots, not the 496 inner products an LDS-resident left-looking sweep serialises)
                const int t = lane & 31;
                double v[PW];
#pragma unroll
                for (int k = 0; k < PW; ++k) v[k] = Ld[t * LP + k];
                bool fail = false;
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    if (c < pw && !fail) {
                        const double piv = readlane_f64(v[c], c);
                        if (!(piv > 1e-300 * dmax) || !isfinite(piv)) {
                            fail = true;
                        } else {
                            // 1 / sqrt(piv): the hardware estimate + two Newton steps (a square root and a division in FP64
                            // are ~40 instructions each, and the 32 pivots are this kernel's one sequential chain)
                            double inv = __builtin_amdgcn_rsq(piv);
                            inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                            inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                            const double dgc = piv * inv;
                            v[c] = t == c ? dgc : v[c] * inv;
                            const double ltc = t > c ? v[c] : 0.0;
#pragma unroll
                            for (int k = c + 1; k < PW; ++k) v[k] = fma(-ltc, readlane_f64(v[c], k), v[k]);
                        }
                    }
                }
                if (fail) {
                    if (lane == 0) bad = 1;
                } else {
                    if (lane < pw) {
#pragma unroll
                        for (int k = 0; k < PW; ++k) Ld[t * LP + k] = k <= t ? v[k] : 0.0;
                    }
                    double dgt = 1.0;
#pragma unroll
                    for (int k = 0; k < PW; ++k)
                        if (k == t) dgt = v[k];
                    if (lane < pw) dg[c0 + t] = dgt;
                    const double invd = 1.0 / dgt;
                    tmp[t] = invd; // 1 / L_tt of this block, for the panel solve
                    // forward substitution of the panel's right-hand side: lanes = rows of the block
                    double yc = t < pw ? y[c0 + t] : 0.0;
#pragma unroll
                    for (int k = 0; k < PW; ++k) {
                        if (k < pw) {
                            const double yk = readlane_f64(yc, k) * readlane_f64(invd, k);
                            if (t == k) yc = yk;
                            else if (t > k) yc = fma(-v[k], yk, yc);
                        }
                    }
                    if (lane < pw) y[c0 + t] = yc;
                }
            }
            __syncthreads();
            if (!bad) { // the factored block's strict lower part -> U (the back substitution reads it from there)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q, c = e >> 5, k = e & 31;
                    if (c < pw && k < c) A[(int64_t)(c0 + k) * hp + c0 + c] = Ld[c * LP + k];
                }
            }
            if (bad) break;
            const int r0 = c0 + pw, nt = m - r0; // trailing rows
            if (nt <= 0) break;
            // (2) panel solve: row i = r0 + a, a = tid, tid + 256
            for (int a = tid; a < nt; a += 256) {
                const int i = r0 + a;
                double x[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) x[c] = c < pw ? (first ? a_low(i, c0 + c) : A[(int64_t)(c0 + c) * hp + i]) : 0.0;
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    if (c < pw) {
                        double v = x[c];
#pragma unroll
                        for (int k = 0; k < c; ++k) v = fma(-x[k], Ld[c * LP + k], v);
                        x[c] = v * tmp[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    Lp[a * LP + c] = x[c];
                    if (c < pw) A[(int64_t)(c0 + c) * hp + i] = x[c];
                }
            }
            __syncthreads();
            // (3) trailing update in 4 x 4 tiles (ib >= jb), and the rest of the right-hand side
            const int nb = (nt + 3) >> 2, ntile = nb * (nb + 1) / 2;
            for (int tile = tid; tile < ntile; tile += 256) {
                int ib = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
                while ((ib + 1) * (ib + 2) / 2 <= tile) ++ib;
                while (ib * (ib + 1) / 2 > tile) --ib;
                const int jb = tile - ib * (ib + 1) / 2;
                double acc[4][4], old[4][4]; // (the entries to be updated are requested first: their latency hides under the products)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc[a][b] = 0.0;
                        const int ia = 4 * ib + a, jbb = 4 * jb + b;
                        old[a][b] = (ia < nt && jbb < ia) ? (first ? a_low(r0 + ia, r0 + jbb) : A[(int64_t)(r0 + jbb) * hp + r0 + ia]) : 0.0;
                    }
                const double *li = Lp + (4 * ib) * LP, *lj = Lp + (4 * jb) * LP;
                for (int c = 0; c < pw; ++c) {
                    double va[4], vb[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        va[a] = li[a * LP + c];
                        vb[a] = lj[a * LP + c];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = fma(va[a], vb[b], acc[a][b]);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int ia = 4 * ib + a, jbb = 4 * jb + b;
                        if (ia >= nt || jbb > ia) continue;
                        const int i = r0 + ia, j = r0 + jbb;
                        if (i == j) dgw[i] -= acc[a][b];
                        else A[(int64_t)j * hp + i] = old[a][b] - acc[a][b];
                    }
            }
            for (int a = tid; a < nt; a += 256) {
                double v = 0.0;
                for (int c = 0; c < pw; ++c) v = fma(Lp[a * LP + c], y[c0 + c], v);
                y[r0 + a] -= v;
            }
            __syncthreads();
        }
        __syncthreads();
        ok = !bad;
        __syncthreads();
        if (!ok) ridge = ridge == 0.0 ? 1e-12 * fmax(dmax, 1e-300) : ridge * 100.0;
    }
    if (!ok) {
        for (int i = tid; i < m; i += 256) dout[(int64_t)r * cap + i] = 0.0;
        return;
    }
    // back substitution L^T d = y, panels in reverse
    for (int c0 = (m - 1) / PW * PW; c0 >= 0; c0 -= PW) {
        const int pw = m - c0 < PW ? m - c0 : PW, r0 = c0 + pw;
        // tmp[c] = sum_{i >= r0} L[i][c0 + c] d_i = sum_i U[c0 + c][i] y[i]
        double ldv[4]; // (requested before the dot products: one round trip for both)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, c = e >> 5, k = e & 31;
            ldv[q] = (c < pw && k < c) ? A[(int64_t)(c0 + k) * hp + c0 + c] : 0.0;
        }
        for (int c = wave; c < pw; c += 4) {
            double v = 0.0;
            for (int i = r0 + lane; i < m; i += 64) v = fma(A[(int64_t)(c0 + c) * hp + i], y[i], v);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) tmp[c] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            Ld[(e >> 5) * LP + (e & 31)] = ldv[q];
        }
        __syncthreads();
        if (wave == 0) {
            const int c = lane & 31;
            double yc = c < pw ? y[c0 + c] - tmp[c] : 0.0;
            for (int k = pw - 1; k >= 0; --k) { // d_k = y_k / L_kk; y_c -= L[c0 + k][c0 + c] d_k for c < k
                const double dk = readlane_f64(yc, k) / dg[c0 + k];
                if (c == k) yc = dk;
                else if (c < k) yc = fma(-Ld[k * LP + c], dk, yc);
            }
            if (lane < pw) y[c0 + c] = yc;
        }
        __syncthreads();
    }
    // Orthant faces.  The line search projects the step onto the orthant of the iterate: an entry at zero may only move
    // against its pseudo-gradient, a non-zero one not past zero.  Entries whose step leaves that face are fixed where the
    // projection would put them (at zero: d = 0 resp. -x) and, when they carry more than `share` of the predicted decrease,
    // the others are solved again -- with correlated statistics the unconstrained solution is full of moves that cancel each
    // other, and clipping one of a pair leaves the other uncompensated (the matrix-free rows do the same: k_pcg_faces).
    if (!F || face >= rounds) break;
    int nf = 0;
    double mass = 0, total = 0;
    for (int a = tid; a < m; a += 256) {
        if (fx[a] != 0.0) continue;
        const int c = F[(int64_t)r * cap + a];
        const double x = X[(int64_t)r * Qp + c], dc = y[a], pv = pg[a];
        total += fabs(pv * dc);
        if (kind[(int64_t)r * Qp + c] != 2) continue;
        if (x == 0.0 ? dc * pv > 0.0 : (x + dc) * x < 0.0) {
            const double fixed = x == 0.0 ? 0.0 : -x;
            mass += fabs(pv * (dc - fixed));
            fx[a] = 2.0; // candidate
            dfx[a] = fixed;
            ++nf;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nf += __shfl_xor(nf, o);
        mass += __shfl_xor(mass, o);
        total += __shfl_xor(total, o);
    }
    __syncthreads();
    if (lane == 0) {
        redi[wave] = nf;
        red[wave] = mass;
    }
    __syncthreads();
    nf = redi[0] + redi[1] + redi[2] + redi[3];
    mass = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = total;
    __syncthreads();
    total = red[0] + red[1] + red[2] + red[3];
    const bool again = nf > 0 && mass > share * total;
    for (int a = tid; a < m; a += 256)
        if (fx[a] == 2.0) fx[a] = again ? 1.0 : 0.0;
    __syncthreads();
    if (!again) break;
    } // face
    for (int i = tid; i < m; i += 256) dout[(int64_t)r * cap + i] = y[i];
}

// ------------------------------------------------------------------------------------------
// The same solve for blocks of up to 128 entries -- the working sets of the pairwise configurations -- with the whole matrix
// in LDS (round 5).  k_newton_chol above keeps the block in global memory and pays three dependent L2 round trips per panel:
// 0.13 ms for 128 rows of 100 entries, one launch per Newton iteration of a solve whose other kernels had shrunk to less
// (profiles/r5_shard128_kernel_stats.csv).  Here A = s1 H - s2 g g^T (lower triangle, masked) is loaded once into W [mcap][mcap + 1]
// and factored in place: the diagonal block by one wave in registers (as above), the panel below it one thread per row, the
// trailing matrix by v_mfma_f64_16x16x4_f64 tiles straight from W (lane (li, q) feeds A[row li][k q] and B[k q][col li], i.e. the
// same access for both operands of L21 L21^T), the back substitution from W.  A ridge restart or an orthant-face re-solve reloads W
// from the global block, which is never written.  Same arithmetic order inside a panel as above; results agree to rounding.
// ------------------------------------------------------------------------------------------
typedef double v4d_c __attribute__((ext_vector_type(4)));
// sum over the 256 threads of a workgroup through 8 x 32 doubles of LDS (no shuffles); every thread gets the result
__device__ __forceinline__ double wg_sum(double v, double *buf /* [256] */) {
    __syncthreads();
    buf[threadIdx.x] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < 64) {
        s = buf[threadIdx.x] + buf[threadIdx.x + 64] + buf[threadIdx.x + 128] + buf[threadIdx.x + 192];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    }
    __syncthreads();
    if (threadIdx.x == 0) buf[0] = s;
    __syncthreads();
    return buf[0];
}
__global__ __launch_bounds__(256) void k_newton_chol_lds(const double *__restrict__ H, const long long *__restrict__ hoff, const int *__restrict__ mt,
                                                         const int *__restrict__ msz, const double *__restrict__ s1, double s2,
                                                         const double *__restrict__ gF, const double *__restrict__ pgF, int cap,
                                                         double *__restrict__ dout, double *__restrict__ Sdiag, int mcap /* multiple of 32, <= 128, >= every msz handled here */,
                                                         const int *__restrict__ F, const double *__restrict__ X, const uint8_t *__restrict__ kind,
                                                         int64_t Qp, double share, int rounds, const uint8_t *__restrict__ fix,
                                                         const double *__restrict__ dfix,
                                                         // BFGS correction of the (sub-sampled) block by the row's last secant pairs,
                                                         // applied to the matrix in LDS (npairs = NULL: none)
                                                         const double *__restrict__ secS, const double *__restrict__ secY,
                                                         const int *__restrict__ npairs, int64_t pair_stride) {
    constexpr int PW = 32;
    const int r = blockIdx.x;
    const int m = msz[r];
    if (m == 0 || m > mcap) return;
    const int hp = 32 * mt[r], LDW = mcap + 1, mp = (m + 15) & ~15;
    const double *A = H + hoff[r];
    const double sc = s1[r];
    const double *g = gF + (int64_t)r * cap, *pg = pgF + (int64_t)r * cap;
    extern __shared__ double sm[]; // W [mcap][mcap + 1] | idg | y | gg | fx | dfx [mcap each]
    double *W = sm, *idg = W + mcap * LDW, *y = idg + mcap, *gg = y + mcap, *fx = gg + mcap, *dfx = fx + mcap;
    __shared__ double part8[8 * 32];
    __shared__ int bad;
    __shared__ double red[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool masked = fix != nullptr || F != nullptr; // some entries may be fixed
#ifdef CHOL_TIMING
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = __builtin_amdgcn_s_memrealtime(), tn;
#define TMARK(i) do { tn = __builtin_amdgcn_s_memrealtime(); tm[i] += tn - tq; tq = tn; } while (0)
#else
#define TMARK(i)
#endif
    // one global round trip: the lower triangle of s1 H straight into W, eight independent loads per thread and turn (one element
    // per turn, each waiting for its own load, took 28 of the 99 us of a 100-entry solve); the rank-one term, the masks and the ridge
    // are applied in LDS
    for (int i = tid; i < m; i += 256) gg[i] = s2 != 0.0 ? g[i] : 0.0;
    auto load_raw = [&]() { // W <- sc * H on the lower triangle of the first mp rows (rows m .. mp - 1, the MFMA tile padding: zero)
        const int total = mp * mp;
        for (int base = 0; base < total; base += 256 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + tid + 256 * u, i = idx / mp, j = idx - i * mp;
                v[u] = (idx < total && j <= i && i < m) ? A[(int64_t)i * hp + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + tid + 256 * u, i = idx / mp, j = idx - i * mp;
                if (idx < total && j <= i) W[i * LDW + j] = sc * v[u];
            }
        }
    };
    // B <- B + y y^T / (y.s) - (B s)(B s)^T / (s.B s) for the row's pairs, oldest first, on B = s1 H - s2 g g^T (the pairs come from
    // exact gradients, so they describe the true Hessian along the last steps: gml_solver.hip, k_secant, which keeps them and
    // applies them itself to the larger blocks).  W holds s1 H; the rank-one term rides along in B s only.
    const int np = npairs ? npairs[r] : 0;
    auto secant = [&]() {
        double *vs = fx, *vy = dfx; // (free here: the masks are set up after the correction) -- s and y of the pair; y doubles as B s below
        for (int l = 0; l < np; ++l) {
            __syncthreads();
            for (int i = tid; i < m; i += 256) {
                vs[i] = secS[l * pair_stride + (int64_t)r * cap + i];
                vy[i] = secY[l * pair_stride + (int64_t)r * cap + i];
            }
            __syncthreads();
            double gs = 0.0;
            if (s2 != 0.0) {
                for (int i = tid; i < m; i += 256) gs += gg[i] * vs[i];
                gs = wg_sum(gs, part8);
            }
            double bs = 0.0, sAs = 0.0, ysl = 0.0;
            if (tid < m) {
                const int i = tid;
                for (int j = 0; j <= i; ++j) bs = fma(W[i * LDW + j], vs[j], bs);
                for (int j = i + 1; j < m; ++j) bs = fma(W[j * LDW + i], vs[j], bs);
                bs -= s2 * gg[i] * gs;
                sAs = vs[i] * bs;
                ysl = vs[i] * vy[i];
            }
            sAs = wg_sum(sAs, part8);
            ysl = wg_sum(ysl, part8);
            if (!(sAs > 0.0 && ysl > 0.0)) continue; // (uniform)
            if (tid < m) y[tid] = bs; // (y: the right-hand side's place, not set yet)
            __syncthreads();
            const double ia = 1.0 / sAs, iy = 1.0 / ysl;
            for (int idx = tid; idx < m * m; idx += 256) {
                const int i = idx / m, j = idx - i * m;
                if (j <= i) W[i * LDW + j] += vy[i] * vy[j] * iy - y[i] * y[j] * ia;
            }
        }
        __syncthreads();
    };
    load_raw();
    __syncthreads();
    if (np > 0) secant();
    for (int i = tid; i < m; i += 256) { // (the masks, after the correction borrowed their place)
        fx[i] = fix && fix[(int64_t)r * cap + i] ? 1.0 : 0.0;
        dfx[i] = fix ? dfix[(int64_t)r * cap + i] : 0.0;
    }
    __syncthreads();
    double dmax = 0;
    for (int i = tid; i < m; i += 256) dmax = fmax(dmax, masked && fx[i] != 0.0 ? 1.0 : fabs(W[i * LDW + i] - s2 * gg[i] * gg[i]));
    for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o));
    if (lane == 0) red[wave] = dmax;
    __syncthreads();
    dmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (tid == 0) Sdiag[r] = W[(m - 1) * LDW + m - 1] - s2 * gg[m - 1] * gg[m - 1];
    bool fresh = true;              // W holds the raw block
    bool anyfixed = fix != nullptr; // some entries are decoupled (test hook, or a face re-solve below)
    for (int face = 0;; ++face) { // (re-solves on an orthant face, see the end of the loop)
    double ridge = 0.0;
    bool ok = false;
    for (int attempt = 0; attempt < 10 && !ok; ++attempt) {
        if (!fresh) {
            __syncthreads();
            load_raw();
            __syncthreads();
            if (np > 0) { // (the correction borrows the masks' place: saved around it)
                double f0 = 0.0, d0 = 0.0;
                if (tid < m) {
                    f0 = fx[tid];
                    d0 = dfx[tid];
                }
                secant();
                if (tid < m) {
                    fx[tid] = f0;
                    dfx[tid] = d0;
                }
                __syncthreads();
            }
        }
        fresh = false;
        // the right-hand side first, from the matrix as it stands in W -- s1 H with the secant correction, the SAME matrix that is
        // factored below -- before the masks decouple the fixed entries: a free row's share of the fixed steps is A_corr[i][j] dfx[j]
        // (the global block, a_orig, lacks the correction: both sides of a face re-solve must see one matrix)
        for (int i = tid; i < m; i += 256) {
            double v = -pg[i];
            if (anyfixed) {
                if (fx[i] != 0.0) {
                    v = dfx[i];
                } else {
                    for (int j = 0; j < m; ++j)
                        if (fx[j] != 0.0 && dfx[j] != 0.0)
                            v -= ((i >= j ? W[i * LDW + j] : W[j * LDW + i]) - s2 * gg[i] * gg[j]) * dfx[j];
                }
            }
            y[i] = v;
        }
        __syncthreads();
        // the (masked) matrix: rank-one term, fixed entries decoupled (unit diagonal), ridge on the free diagonal
        if (s2 != 0.0 || anyfixed || ridge != 0.0) {
            for (int idx = tid; idx < m * m; idx += 256) {
                const int i = idx / m, j = idx - i * m;
                if (j > i) continue;
                double v = W[i * LDW + j] - s2 * gg[i] * gg[j];
                if (anyfixed && (fx[i] != 0.0 || fx[j] != 0.0)) v = i == j ? 1.0 : 0.0;
                else if (i == j) v += ridge;
                W[i * LDW + j] = v;
            }
        }
        if (tid == 0) bad = 0;
        __syncthreads();
        TMARK(0);
        for (int c0 = 0; c0 < m; c0 += PW) {
            const int pw = m - c0 < PW ? m - c0 : PW;
            if (wave == 0) {
                // (1) diagonal block: lane t holds row t in registers; right-looking: column c is scaled, then every lane updates the
                // rest of its row with L[k][c] shuffled in from lane k (the dependent chain is the 32 pivots)
                const int t = lane & 31;
                double v[PW];
#pragma unroll
                for (int k = 0; k < PW; ++k) v[k] = (t < pw && k <= t) ? W[(c0 + t) * LDW + c0 + k] : 0.0;
                // (straight-line code: a pivot that fails -- or lies beyond the block -- is replaced by 1 and remembered, so that no
                // control flow separates the columns and the scheduler can start a pivot's chain under the previous column's updates)
                bool fail = false;
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    double piv = readlane_f64(v[c], c);
                    const bool live = c < pw, good = piv > 1e-300 * dmax && isfinite(piv);
                    fail |= live && !good;
                    piv = live && good ? piv : 1.0;
                    // 1 / sqrt(piv): the hardware estimate + two Newton steps
                    double inv = __builtin_amdgcn_rsq(piv);
                    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                    inv = inv * fma(-0.5 * piv * inv, inv, 1.5);
                    const double dgc = piv * inv;
                    v[c] = t == c ? dgc : v[c] * inv;
                    const double ltc = t > c ? v[c] : 0.0;
#pragma unroll
                    for (int k = c + 1; k < PW; ++k) v[k] = fma(-ltc, readlane_f64(v[c], k), v[k]);
                }
                if (fail) {
                    if (lane == 0) bad = 1;
                } else {
                    if (lane < pw) {
#pragma unroll
                        for (int k = 0; k < PW; ++k)
                            if (k <= t) W[(c0 + t) * LDW + c0 + k] = v[k];
                    }
                    double dgt = 1.0;
#pragma unroll
                    for (int k = 0; k < PW; ++k)
                        if (k == t) dgt = v[k];
                    const double invd = 1.0 / dgt;
                    if (lane < pw) idg[c0 + t] = invd; // 1 / L_tt
                    // forward substitution of the panel's right-hand side: lanes = rows of the block
                    double yc = t < pw ? y[c0 + t] : 0.0;
#pragma unroll
                    for (int k = 0; k < PW; ++k) {
                        if (k < pw) {
                            const double yk = readlane_f64(yc, k) * readlane_f64(invd, k);
                            if (t == k) yc = yk;
                            else if (t > k) yc = fma(-v[k], yk, yc);
                        }
                    }
                    if (lane < pw) y[c0 + t] = yc;
                }
            }
            __syncthreads();
            TMARK(1);
            if (bad) break;
            const int r0 = c0 + pw, nt = m - r0; // trailing rows
            if (nt <= 0) break;
            // (2) panel solve: row i = r0 + a, one thread per row (pw = 32 here: only the last panel is narrower, and it has no rows below)
            for (int a = tid; a < nt; a += 256) {
                double *wr = W + (r0 + a) * LDW + c0;
                double x[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) x[c] = wr[c];
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    double v = x[c];
                    const double *lc = W + (c0 + c) * LDW + c0;
#pragma unroll
                    for (int k = 0; k < c; ++k) v = fma(-x[k], lc[k], v);
                    x[c] = v * idg[c0 + c];
                }
#pragma unroll
                for (int c = 0; c < PW; ++c) wr[c] = x[c];
            }
            __syncthreads();
            TMARK(2);
            // (3) trailing update W22 -= L21 L21^T in 16 x 16 MFMA tiles (I >= J) over the waves, and the rest of the right-hand side
            {
                const int li = lane & 15, q = lane >> 4;
                const int T = (nt + 15) >> 4, ntile = T * (T + 1) / 2;
                for (int tile = wave; tile < ntile; tile += 4) {
                    int I = 0, rem = tile;
                    while (rem > I) {
                        rem -= I + 1;
                        ++I;
                    }
                    const int J = rem;
                    const double *pa = W + (r0 + 16 * I + li) * LDW + c0 + q, *pb = W + (r0 + 16 * J + li) * LDW + c0 + q;
                    v4d_c acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < PW / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = r0 + 16 * I + q + 4 * j, col = r0 + 16 * J + li;
                        if (row < mp && col <= row) W[row * LDW + col] -= acc[j];
                    }
                }
                for (int a = tid; a < nt; a += 256) {
                    const double *wr = W + (r0 + a) * LDW + c0;
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < PW; ++c) v = fma(wr[c], y[c0 + c], v);
                    y[r0 + a] -= v;
                }
            }
            __syncthreads();
            TMARK(3);
        }
        __syncthreads();
        ok = !bad;
        __syncthreads();
        if (!ok) ridge = ridge == 0.0 ? 1e-12 * fmax(dmax, 1e-300) : ridge * 100.0;
    }
    if (!ok) {
        for (int i = tid; i < m; i += 256) dout[(int64_t)r * cap + i] = 0.0;
        return;
    }
    // back substitution L^T d = y, panels in reverse
    for (int c0 = (m - 1) / PW * PW; c0 >= 0; c0 -= PW) {
        const int pw = m - c0 < PW ? m - c0 : PW, r0 = c0 + pw;
        // tmp[c] = sum_{i >= r0} L[i][c0 + c] d_i: lanes over the columns, the rows split over the eight half-waves, partial sums
        // through LDS (no shuffles: a xor-reduction is six ds_bpermute round trips per column)
        {
            const int c = lane & 31, part = tid >> 5;
            double v = 0.0;
            if (c < pw)
                for (int i = r0 + part; i < m; i += 8) v = fma(W[i * LDW + c0 + c], y[i], v);
            part8[part * 32 + c] = v;
        }
        __syncthreads();
        if (wave == 0) {
            const int c = lane & 31;
            double yc = 0.0;
            if (c < pw) {
                double ts = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) ts += part8[q * 32 + c];
                yc = y[c0 + c] - ts;
            }
            double lc[PW], iv = c < pw ? idg[c0 + c] : 0.0; // column c of the block, below the diagonal: L[c0 + k][c0 + c], k > c
#pragma unroll
            for (int k = 0; k < PW; ++k) lc[k] = (k < pw && c < k) ? W[(c0 + k) * LDW + c0 + c] : 0.0;
#pragma unroll
            for (int k = PW - 1; k >= 0; --k) { // d_k = y_k / L_kk; y_c -= L[c0 + k][c0 + c] d_k for c < k
                if (k < pw) {
                    const double dk = readlane_f64(yc, k) * readlane_f64(iv, k);
                    if (c == k) yc = dk;
                    else yc = fma(-lc[k], dk, yc);
                }
            }
            if (lane < pw) y[c0 + c] = yc;
        }
        __syncthreads();
    }
    TMARK(4);
    // Orthant faces (as in k_newton_chol): entries whose step leaves the face of the iterate are fixed where the projection of the
    // line search would put them and, when they carry more than `share` of the predicted decrease, the others are solved again
    if (!F || face >= rounds) break;
    int nf = 0;
    double mass = 0, total = 0;
    for (int a = tid; a < m; a += 256) {
        if (fx[a] != 0.0) continue;
        const int c = F[(int64_t)r * cap + a];
        const double x = X[(int64_t)r * Qp + c], dc = y[a], pv = pg[a];
        total += fabs(pv * dc);
        if (kind[(int64_t)r * Qp + c] != 2) continue;
        if (x == 0.0 ? dc * pv > 0.0 : (x + dc) * x < 0.0) {
            const double fixed = x == 0.0 ? 0.0 : -x;
            mass += fabs(pv * (dc - fixed));
            fx[a] = 2.0; // candidate
            dfx[a] = fixed;
            ++nf;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nf += __shfl_xor(nf, o);
        mass += __shfl_xor(mass, o);
        total += __shfl_xor(total, o);
    }
    __syncthreads();
    if (lane == 0) {
        redi[wave] = nf;
        red[wave] = mass;
    }
    __syncthreads();
    nf = redi[0] + redi[1] + redi[2] + redi[3];
    mass = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = total;
    __syncthreads();
    total = red[0] + red[1] + red[2] + red[3];
    const bool again = nf > 0 && mass > share * total;
    for (int a = tid; a < m; a += 256)
        if (fx[a] == 2.0) fx[a] = again ? 1.0 : 0.0;
    __syncthreads();
    if (!again) break;
    anyfixed = true;
    } // face
    for (int i = tid; i < m; i += 256) dout[(int64_t)r * cap + i] = y[i];
#ifdef CHOL_TIMING
    TMARK(5);
    if (tid == 0 && cap >= 508)
        for (int i = 0; i < 6; ++i) dout[(int64_t)r * cap + 500 + i] = (double)tm[i] * 0.01; // us
#endif
#undef TMARK
}

void launch_newton_solve(double *H, const long long *hoff, const int *mt, const int *msz, const double *s1, double s2,
                         const double *gF, const double *pgF, int R, int cap, double *dout, double *Sdiag, hipStream_t st, int maxm,
                         const NewtonFaces *faces, const uint8_t *fix, const double *dfix, const SecantPairs *pairs) {
    // maxm: largest block of this call, as far as the host knows it (0 = unknown): sizes the LDS of a workgroup
    int mcap = maxm <= 0 || maxm > cap ? cap : maxm;
    mcap = (mcap + 31) / 32 * 32;
    const NewtonFaces nf = faces ? *faces : NewtonFaces{};
    // blocks of up to 128 entries: the matrix in LDS (k_newton_chol_lds); larger ones: the blocked kernel on the global block
    const int msmall = mcap < kCholLds ? mcap : kCholLds;
    const SecantPairs sp = pairs ? *pairs : SecantPairs{};
    {
        const size_t lds = sizeof(double) * ((size_t)msmall * (msmall + 1) + 5 * (size_t)msmall + 32);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_newton_chol_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_newton_chol_lds, dim3((unsigned)R), dim3(256), lds, st, H, hoff, mt, msz, s1, s2, gF, pgF, cap, dout, Sdiag, msmall, nf.F,
                           nf.X, nf.kind, nf.Qp, nf.share, nf.rounds, fix, dfix, sp.S, sp.Y, sp.npairs, sp.stride);
    }
    if (mcap > kCholLds) {
        const size_t lds = sizeof(double) * ((size_t)6 * mcap + 32 * 33 + 32 + (size_t)std::max(mcap - 32, 32) * 33);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_newton_chol), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_newton_chol, dim3((unsigned)R), dim3(256), lds, st, H, hoff, mt, msz, s1, s2, gF, pgF, cap, dout, Sdiag, mcap, nf.F, nf.X,
                           nf.kind, nf.Qp, nf.share, nf.rounds, fix, dfix, kCholLds);
    }
}

} // namespace gml
