// Internal device-side interface of libgml_hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "gml_bits.h"

namespace gml {

// Device memory of the library (gml_alloc.cpp): freed blocks of >= 1 MB are kept per device and size for the next handle of
// the same shape (a hipMalloc of a multi-GB block after a hipFree sporadically takes a second).
hipError_t dev_malloc_bytes(void **out, size_t bytes);
hipError_t dev_free(void *p);        // synchronises the block's device before the block can be handed out again (as hipFree does)
hipError_t dev_free_synced(void *p); // the caller has synchronised the device (or every stream that touched the block) itself
hipError_t dev_mem_info(size_t *free_bytes, size_t *total_bytes); // free = the driver's + the cache's
size_t dev_trim_cache();
template <typename T> inline hipError_t dev_malloc(T **out, size_t bytes) { return dev_malloc_bytes(reinterpret_cast<void **>(out), bytes); }

// Device-resident problem data (all padded, padding is zero).  The samples are stored as ONE BIT per entry:
//   Sb  [n][Kp/32]     sign bits of the spins, spin-major, natural order (bit j of word w <-> sample 32w + j; set <=> -1)
//   Xb                 forward operand of the int8 path: the design matrix sample-major, in the piece layout of
//                      k_build_xb (gml_i8_pack.hip)
//   Xtb                backward operand: the same matrix feature-major, piece layout of k_build_xtb
//   keys [Qf][ko]      spins of every statistic column (-1 = unused slot): column c = prod of its spins = XOR of sign bits
//   w   [Kp]     f64   c_k / M   (samples[k,1]/num_samples, GraphicalModelLearning.jl:170)
//   Xt [Qp][Kp] int8   feature-major byte image, built on first use by the FP64 path only (rows gathered by its Hessian kernel)
// Columns 0..n-1 are the single spins, then pairs (i<j) in lexicographic order, ... (multi-body, :94-108); column
// `cconst` is the empty key (constant 1: the node's field).  Column Qp-1 is always a zero (padding) column.
struct DevProblem {
    int64_t K, Kp, n;
    int64_t Qf, Qfp;  // statistic columns [0,Qf), zero padded to Qfp (multiple of 64)
    int64_t cconst;   // column of the constant statistic (= Qfp)
    int64_t Qp;       // row pitch of the parameter arrays / number of rows of Xt (= Qfp + 64)
    unsigned *Sb;
    unsigned *Xb, *Xtb;
    int32_t *keys;
    int ko;
    int8_t *Xt;
    double *w;
    double wmax;      // max_k w_k
    double wuni;      // the common weight when all K samples weigh the same (counts all equal), else 0
};

// ---- packing -----------------------------------------------------------------------------
void launch_transpose_i8(const int8_t *src, int64_t rows, int64_t cols, int64_t ld_src,
                         int8_t *dst, int64_t ld_dst, hipStream_t st);
// Sb from +-1 bytes: sample-major S [K][n] (ld ignored) or spin-major S [n][ld]
void launch_spin_bits(const int8_t *S, bool spin_major, int64_t K, int64_t n, int64_t ld, int64_t Kp, unsigned *Sb, hipStream_t st);
void launch_pack_bits(const DevProblem &d, hipStream_t st);      // Sb, keys -> Xb, Xtb
void launch_unpack_spins(const DevProblem &d, int64_t k0, int64_t kk, int8_t *out /* [kk][n] */, hipStream_t st);
void launch_expand_xt(const DevProblem &d, int8_t *Xt, hipStream_t st); // Sb, keys -> rows [0, Qf) of the byte image
// *bad = smallest configuration index holding an entry that is not +-1 (unchanged if there is none; init -1)
// Histogram matrix (K x (1+n), element type `dtype` of gml.h, leading dimension ld) -> counts [K] and +-1 int8 spins:
// column-major input gives spin-major output [n][K], row-major input sample-major [K][n].  *bad as in launch_check_pm1.
void launch_convert_hist(const void *H, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, double *counts, int8_t *spins,
                         long long *bad, hipStream_t st);
void launch_check_pm1(const int8_t *S, int64_t K, int64_t n, long long *bad, hipStream_t st);
int64_t xtb_bytes(const DevProblem &d);
// device-pointer operator calls (gml_oplayout.hip): the caller's rows in the reference's parameter order <-> the internal columns
void launch_ref_to_internal(const double *theta, int64_t ld, int64_t R, int64_t P, int64_t Qp, const int *rowcol, int64_t cconst,
                            const int32_t *cols, double *X, int *bad, hipStream_t st);
void launch_internal_to_ref(const double *G, const double *F, int64_t R, int64_t Qp, int64_t P, int64_t ld, const int *rowcol, const uint8_t *sel,
                            int64_t cconst, const int32_t *cols, int logz, double *f, double *g, hipStream_t st);
void launch_hv_to_ref(const double *Hv, const double *Gz, const double *F, int64_t R, int64_t Qp, int64_t P, int64_t ld, const int *rowcol,
                      int64_t cconst, const int32_t *cols, int logz, const double *vec, double *hv, hipStream_t st);

// Byte offset of limb l of V[r][k] in the int8 limb image Vq of the fixed-point pass: images [node tile r/32][k/64]
// of [lbt limbs x 32 rows][64 B], contiguous (lbt = 4: 8 KB each, the i8x pass; 6: 12 KB, the i8w pass); within a row the
// 64 samples of the step are stored in the order vq_pos() (gml_bits.h), which the feature-major bit image shares, so the
// backward GEMM contracts position against position.
__host__ __device__ inline int64_t vq_off(int64_t r, int l, int64_t k, int64_t Kp, int lbt = 4) {
    return ((((r >> 5) * (Kp >> 6) + (k >> 6)) * lbt + l) * 32 + (r & 31)) * 64 + vq_pos((int)(k & 63));
}

// ---- int8-limb path --------------------------------------------------------------------------
// One pass of the fixed-point operator (gml_i8_pass.hip).  The rows to evaluate are listed in consecutive SLOTS
// (32 slots = one MFMA node tile; slot0, slot1 multiples of 32), so the tiles that run are full whatever subset of the
// caller's rows is active; all per-slot arrays (and the limb planes of V the Hessians are built from) live in the
// workspace, indexed by slot.
struct SlotResult { // per-slot scalars of a pass, packed for one download
    double f, tau;      // f per slot (RISE f, logRISE Z, RPLE f); scale of the V planes
    unsigned mmax, pad; // largest |V| / tau seen (dynamic-range check)
};
struct I8Pass {
    const double *theta;  // device, [rows][Qp]: the parameter rows (internal column layout, masked slots zero)
    const int *srow;      // device [slots]: slot -> row of theta / G (unused slots: anything)
    const int *rowcol;    // device [slots]: slot -> node u (whose sign row is the node's spin), -1 = unused slot
    const int *groups;    // device: the slot tiles to run (slot / 32), padded with -1 to a multiple of 4
    int ngroups;          // tiles listed (without the padding)
    int slot0, slot1;     // slot range the listed tiles lie in: its accumulators are zeroed, its rows quantised and finalised
    int form;             // GML_RISE / GML_LOGRISE (Z) / GML_RPLE
    bool want_grad;       // false: objective only (no backward GEMM)
    double *F;            // device [slots]: f per slot (RISE f, logRISE Z, RPLE f)
    double *G;            // device, [rows][Qp]: gradient rows (row srow[slot]); may be NULL when !want_grad
    const double *tauovr; // device [slots] or NULL: per-slot scale of V imposed by the caller (0 = from the bound)
    bool zero_theta;      // the caller guarantees that every listed row of theta is zero (a solve's first pass at x = 0): all energies are 0
                          // and the forward kernel skips its sweeps over the columns -- the same bits, without the GEMM
    const double *tauovr_lnrow; // device [rows of theta] or NULL: tauovr[slot] is multiplied by exp(tauovr_lnrow[srow[slot]]) (a trial
                          //    point's distance from the iterate the scale was measured at, left on the device by the kernel that formed it)
    int hv;               // 1: Hessian-vector product -- theta rows are directions p, G receives sum_k h_k (x_k.p) x_k with the
                          //    curvature weights h_k of the objective pass that last ran in slot vmap[slot]
                          //    2: the same with the products h_k (x_k.p) carried in 2 backward limbs (15 bits) instead of 4
    const int *vmap;      // hv: device [slots]
    int ksub;             // hv: the products run over ~1/ksub of the configurations (i8_split_plan); 0, 1 = all of them
    int64_t kchunk, kpart; // hv: a split plan fixed by the caller (one operator for all the steps of a CG solve); 0 = plan here
    SlotResult *res;      // device [slots] or NULL: {f, tau, mmax} of every slot of the pass, written by its last kernel
    int lf;               // forward limb planes (3, 4, 5; 0 = the default, 5; 2 for Hessian-vector directions): 8 lf - 2 significant bits of theta
    bool wide;            // the FP64-grade pass (precision i8w): theta in 7 limb planes (54 bits), V in 6 (47 bits), FP64 exp;
                          // the workspace then holds 6-plane V images, of which Hessians and Hessian-vector passes read the top 4
    bool compact;         // objective passes: the forward GEMM of a node tile sweeps only the columns on which one of the tile's rows is
                          // non-zero (l1-sparse iterates: a few dozen of thousands of columns) -- the compact column list and the bit
                          // image of those columns are built on the device in front of the pass; the integer sums, hence every result
                          // bit, are those of the sweep over all columns.  Tiles whose union exceeds a quarter of the columns run dense
    bool coarse;          // the cheap form of an objective pass (exp forms) for iterates far from the optimum.  wide: the top four
                          // planes of theta (30 bits), V in three planes (dithered 23 bits): one forward sweep and one backward launch
                          // instead of two; SlotResult.tau stays the unit of the 47-bit planes, the values are multiples of 2^24 tau.
                          // i8x: theta in 4 limb planes (30 bits), V in the planes 1..3 (23 bits, multiples of 2^8 tau), a 3-plane
                          // backward launch
};
void i8_split_plan(const DevProblem &d, int ngroups, int ksub, int64_t *kchunk, int64_t *kpart, int *nsplit);
int i8_pass(void **ws, const DevProblem &d, int64_t slot_capacity, const I8Pass &a, hipStream_t st, hipEvent_t *ev /* [3] or NULL */,
            std::string *err);
// Unit of SlotResult.mmax in multiples of SlotResult.tau, and the largest |V| / tau the planes hold, for a pass of the given width:
// (mmax + 1) * i8_mmax_unit * tau bounds max_k |V_k|; a caller-imposed scale is bound / i8_vdiv.
inline double i8_mmax_unit(bool wide) { return wide ? 65536.0 : 1.0; }
inline double i8_vdiv(bool wide) { return wide ? 1.400e14 : 2130000000.0; }
// unit of the values of a coarse pass, in multiples of SlotResult.tau (the noise of f and grad scales with it)
inline double i8_coarse_unit(bool wide) { return wide ? 16777216.0 : 256.0; }
void i8_free(void *ws);
void i8_compact_table(void *ws, const int **cnk, int *csteps);
// per-slot results of the last pass of the given kind (device pointers)
void i8_slot_results(void *ws, int hv, const double **tau, const unsigned **mmax);
// the limb planes of V of the workspace (device pointer, bytes): the kernel-timing experiments read per-workgroup timestamps from it
void i8_vq_buffer(void *ws, const int8_t **vq, int64_t *bytes, const DevProblem &d);
// Hessian-vector products of a few matrix-free rows over their working sets only, bit-identical to an i8_pass with hv = 2 and two
// forward limbs (gml_hv_sparse.hip).  rows / node / vslot [nrows]: local row, its spin, the slot of its V planes; nw, t0 (indexed by
// local row): size and first tile of the row's list of working-set columns in FV (tiles of T entries); wcap >= every listed nw,
// <= 65536; buf: i8_hv_sparse_bytes(d, nrows, wcap) bytes of device scratch.  Exp forms (RISE, logRISE), statistics of <= 2 spins.
size_t i8_hv_sparse_bytes(const DevProblem &d, int nrows, int64_t wcap);
int i8_hv_sparse(void *ws, const DevProblem &d, int nrows, const int *rows, const int *node, const int *vslot, const long long *t0, const int *nw,
                 const int *FV, int T, int64_t wcap, const double *P, double *Hout, int64_t kchunk, int64_t kpart, void *buf, hipStream_t st,
                 std::string *err);
// Extra blocks of a Hessian call: the preconditioner tiles of the matrix-free rows.  Block R + t (t < n) is the T x T Hessian of
// the T columns F[t T ..] under the weights of row wrow[t]; the caller's mt / hoff arrays cover R + n blocks (mt = T / 32 for a
// tile).  hflag [R]: rows whose weights are needed (those with a working set, and those with tiles).
struct HessTiles {
    int64_t n = 0;
    int T = 0;
    const int *F = nullptr, *wrow = nullptr, *hflag = nullptr;
};
int i8_hessian(void *ws, const DevProblem &d, const int *dRowcol, const int *dVslot, const int *dF, const int *dMt, const int *hMt,
               const long long *dHoff, int64_t htotal, int R, int cap, int form, int64_t Kh, int64_t kstride, double *dH,
               hipStream_t st, std::string *err, const HessTiles *tiles = nullptr);

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
                     int64_t Kh, int64_t kstride, double *H, hipStream_t st);
// H is ragged: row r's block starts at hoff[r] and is (32 mt[r]) x (32 mt[r]) with that pitch.
// Kh (multiple of 512, <= Kp), kstride: the Hessian is accumulated over Kh configurations, every kstride-th block
// of 512 (sub-sampled Newton: the gradient stays exact, so only the convergence rate is affected).

// Exact sampling of one block (connected component) of a model given as terms (bit masks over the block's
// spins + weights): energies of its 2^sb states, CDF, N draws written into S [N][n] (sample-major, +-1) at
// the block's spin columns.
void launch_block_sampler(const unsigned *dmasks, const double *dwts, int nt, int sb, const int *dmembers, int64_t N, int64_t n,
                          unsigned long long seed, int block, double *den, double *dcdf, int8_t *dS, hipStream_t st);

// Histogramming on the device (gml_dedupe.hip): distinct configurations of N samples held as +-1 bytes (n <= 64, N < 2^31) as
// ascending 64-bit keys (bit i set <=> spin i is -1) with their multiplicities; both arrays are device memory owned by the caller.
int dedupe_samples(const int8_t *dS, bool spin_major, int64_t ld, int64_t N, int64_t n, hipStream_t st, unsigned long long **dkeys_out,
                   int **dcounts_out, int64_t *K_out, std::string *err);
void launch_bits_from_keys(const unsigned long long *dkeys, int64_t K, int64_t n, int64_t Kp, unsigned *Sb, hipStream_t st);

// Glauber dynamics (N independent chains, `sweeps` sweeps) on incidence lists; St [n][Np] spin-major.
void launch_glauber(const int *dioff, const double *diw, const int *dooff, const int *doth, int64_t n, int64_t N, int64_t Np,
                    int sweeps, unsigned long long seed, int8_t *dSt, hipStream_t st);

// Batched Newton solve on the ragged Hessian blocks: A = s1[r]*H_r - s2*gF gF^T, A d = -pgF, in place
// (Cholesky, ridge restart).  gF/pgF/dout are R x cap; Sdiag[r] = A[m-1][m-1].
// faces: the orthant-face re-solves inside the kernel (working-set columns F [R][cap], iterates X [R][Qp] and their column kinds,
// the share of the predicted decrease that triggers a re-solve, the number of re-solves); fix / dfix [R][cap]: entries fixed
// from the start (tests).
struct NewtonFaces {
    const int *F = nullptr;
    const double *X = nullptr;
    const uint8_t *kind = nullptr;
    int64_t Qp = 0;
    double share = 0.05;
    int rounds = 0;
};
// the rows' last secant pairs (k_secant keeps them): pair l of row r at S / Y + l * stride + r * cap, npairs[r] of them.  Blocks of up
// to 128 entries take the BFGS correction inside the solve kernel, on the matrix it holds in LDS
struct SecantPairs {
    const double *S = nullptr, *Y = nullptr;
    const int *npairs = nullptr;
    int64_t stride = 0;
};
constexpr int kCholLds = 128; // blocks of up to this many entries are solved with the matrix in LDS
void launch_newton_solve(double *H, const long long *hoff, const int *mt, const int *msz, const double *s1, double s2,
                         const double *gF, const double *pgF, int R, int cap, double *dout, double *Sdiag, hipStream_t st, int maxm = 0,
                         const NewtonFaces *faces = nullptr, const uint8_t *fix = nullptr, const double *dfix = nullptr,
                         const SecantPairs *pairs = nullptr);
// Preconditioner tiles (T = 64 or 128): tile t holds the lower triangle of its T x T Hessian block at H + hoff[t] (pitch T);
// A = s1[wrow[t]] * H_t - s2 * g g^T on its first vm[t] entries is replaced, in place, by its inverse (full symmetric matrix);
// a block that is not positive definite even with a ridge becomes its inverse diagonal.
void launch_tile_inverse(int T, double *H, const long long *hoff, const int *vm, const int *wrow, const double *s1, double s2, const double *gV,
                         int64_t ntiles, hipStream_t st);

} // namespace gml
