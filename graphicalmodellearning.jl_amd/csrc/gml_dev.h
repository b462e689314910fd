// Internal device-side interface of libgml_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gml {

// Device-resident problem data.  Layout (all padded, padding is zero):
//   Xt [Qp][Kp] int8   feature-major design matrix: Xt[c][k] = prod_{i in key_c} s_i^k
//   Xb                 the same matrix sample-major, one bit per entry (set <=> -1), in the piece
//                      layout of k_pack_bits (gml_kernels_i8.hip): the forward operand of the int8 path
//   Xtb                the bits of Xt in the piece layout of k_pack_bits_t: the backward operand
//   Xs [Kp][Qp] int8   sample-major bytes; built on first use by the FP64 path only
//   w  [Kp]     f64    c_k / M   (samples[k,1]/num_samples, GraphicalModelLearning.jl:170)
// Columns 0..n-1 are the single spins, then pairs (i<j) in lexicographic order, ... (multi-body,
// :94-108); column `cconst` is the empty key (constant 1: the node's field).  Node u's sign
// s_u^k is row u of Xt.  Column Qp-1 is always a zero (padding) column.
struct DevProblem {
    int64_t K, Kp, n;
    int64_t Qf, Qfp;  // statistic columns [0,Qf), zero padded to Qfp (multiple of 64)
    int64_t cconst;   // column of the constant statistic (= Qfp)
    int64_t Qp;       // row pitch of Xs / number of rows of Xt (= Qfp + 64)
    int8_t *Xs, *Xt;
    unsigned *Xb, *Xtb; // bit images: sample-major (forward operand) and feature-major (backward operand)
    double *w;
    double wmax;      // max_k w_k
    double wuni;      // the common weight when all K samples weigh the same (counts all equal), else 0
};

// ---- packing -----------------------------------------------------------------------------
void launch_transpose_i8(const int8_t *src, int64_t rows, int64_t cols, int64_t ld_src,
                         int8_t *dst, int64_t ld_dst, hipStream_t st);
void launch_pack_bits(const DevProblem &d, hipStream_t st); // Xt -> Xb, Xtb
// *bad = smallest configuration index holding an entry that is not +-1 (unchanged if there is none; init -1)
// Histogram matrix (K x (1+n), element type `dtype` of gml.h, leading dimension ld) -> counts [K] and +-1 int8 spins:
// column-major input gives spin-major output [n][K], row-major input sample-major [K][n].  *bad as in launch_check_pm1.
void launch_convert_hist(const void *H, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, double *counts, int8_t *spins,
                         long long *bad, hipStream_t st);
void launch_check_pm1(const int8_t *S, int64_t K, int64_t n, long long *bad, hipStream_t st);
int64_t xtb_bytes(const DevProblem &d);
void launch_expand_features(const int8_t *St, int64_t n, int64_t K, int64_t Kp,
                            const int32_t *keys, int order, int64_t Q, int8_t *Xt,
                            hipStream_t st);

// Byte offset of limb l of V[r][k] in the int8 limb image Vq of the exact fixed-point pass:
// images [node tile r/32][k/64] of [4 limbs x 32 rows][64 B], contiguous (8 KB each).  Within a row the
// 64 samples of the step are PERMUTED: the forward epilogue's lane (node, half h) owns the samples
// 32 i + 8 g + 4 h + j (i < 2, g < 4, j < 4) and stores them at byte 32 h + 16 i + 4 g + j, so that its
// 32 bytes per limb are contiguous and go out as two 16-byte stores without an LDS transpose.  The
// feature-major bit image (k_pack_bits_t) uses the same order, so the backward GEMM contracts
// position against position.
__host__ __device__ inline int vq_pos(int s) { return ((s >> 2) & 1) * 32 + (s >> 5) * 16 + ((s >> 3) & 3) * 4 + (s & 3); }
__host__ __device__ inline int vq_sample(int p) { return ((p >> 4) & 1) * 32 + ((p >> 2) & 3) * 8 + (p >> 5) * 4 + (p & 3); }
__host__ __device__ inline int64_t vq_off(int64_t r, int l, int64_t k, int64_t Kp) {
    return ((((r >> 5) * (Kp >> 6) + (k >> 6)) * 4 + l) * 32 + (r & 31)) * 64 + vq_pos((int)(k & 63));
}

// ---- FP64 path -----------------------------------------------------------------------------
// Theta [Rp][Qp] (internal column layout, masked slots zero), rowcol[r] = u (row of Xt
// holding node u's sign) or -1 for padding rows.  Rp multiple of 32.
// V [Rp][Kp]: V[r][k] = d f_r / d E_rk * s_rk = -w_k exp(-E) s (RISE: `partial_obj` of :204 times s)
// fsum [Rp]: sum_k w_k phi(E_rk)  (must be zeroed by the caller)
// groups: ids of the 32-row groups to evaluate (padded with -1 to a multiple of 4).
void launch_fwd_f64(const DevProblem &P, const double *Theta, const int *rowcol, const int *groups,
                    int ngroups4, int form, double *V, double *fsum, hipStream_t st);
// G [Rp][Qp] += sum_k V[r][k] * Xt[c][k] = the gradient (:205-207); must be zeroed by the caller
void launch_bwd_f64(const DevProblem &P, const double *V, const int *groups, int ngroups, double *G,
                    hipStream_t st);
// H [Rp][cap][cap] += sum_k h_rk Xt[F_ri][k] Xt[F_rj][k], lower-triangular 32x32 tiles only.
// F [Rp][cap] column ids (padding = Qp-1), mt[r] = number of 32-tiles used by row r.
void launch_hess_f64(const DevProblem &P, const double *V,
                     const int *rowcol, const int *F, const int *mt, const long long *hoff, int R, int cap, int form,
                     int64_t Kh, double *H, hipStream_t st);
// H is ragged: row r's block starts at hoff[r] and is (32 mt[r]) x (32 mt[r]) with that pitch.
// Kh (multiple of 32, <= Kp): the Hessian is accumulated over the first Kh configurations only
// (sub-sampled Newton: the gradient stays exact, so only the convergence rate is affected).

// Exact sampling of one block (connected component) of a model given as terms (bit masks over the block's
// spins + weights): energies of its 2^sb states, CDF, N draws written into S [N][n] (sample-major, +-1) at
// the block's spin columns.
void launch_block_sampler(const unsigned *dmasks, const double *dwts, int nt, int sb, const int *dmembers, int64_t N, int64_t n,
                          unsigned long long seed, int block, double *den, double *dcdf, int8_t *dS, hipStream_t st);

// Glauber dynamics (N independent chains, `sweeps` sweeps) on incidence lists; St [n][Np] spin-major.
void launch_glauber(const int *dioff, const double *diw, const int *dooff, const int *doth, int64_t n, int64_t N, int64_t Np,
                    int sweeps, unsigned long long seed, int8_t *dSt, hipStream_t st);

// Batched Newton solve on the ragged Hessian blocks: A = s1[r]*H_r - s2*gF gF^T, A d = -pgF, in place
// (Cholesky, ridge restart).  gF/pgF/dout are R x cap; Sdiag[r] = A[m-1][m-1].
void launch_newton_solve(double *H, const long long *hoff, const int *mt, const int *msz, const double *s1, double s2,
                         const double *gF, const double *pgF, int R, int cap, double *dout, double *Sdiag, hipStream_t st);

} // namespace gml
