// Interface between the host solver (gml_solver.cpp) and its device kernels (gml_solver.hip).
#pragma once
#include "gml_dev.h"

#include <atomic>

namespace gml {

struct SelectOut {
    double l1;     // lambda * sum |x_c| over the penalised columns
    double worst;  // KKT residual: max |pseudo-gradient|
    double worstW; // the same over the current support
    int m;         // working-set size (Cholesky row), or -|W| for a matrix-free (Newton-CG) row
    int nsupp, nviol;
    int pad;
};
// (test hook, gml_testhooks.cpp) the Hessian-vector products of few live rows go entry by entry when the sum of their working-set
// sizes is below this multiple of (statistics columns x node tiles of the GEMM pass); < 0: never
extern double g_hv_sparse_ratio;
extern std::atomic<long long> g_hv_sparse_calls; // products taken entry by entry so far (all solves of the process)
// experiment knobs (gml_test_tune; 0 = the built-in rule): what scripts/gpu_tune_*.py sweep, not an interface
enum { GML_TUNE_KH_BASE = 0, GML_TUNE_FACE_ROUNDS = 1 /* value - 1 = rounds */, GML_TUNE_HESS_WGS = 2, GML_TUNE_DUAL_STREAMS = 3 /* 1: passes on a low-priority stream, the rest on a high-priority one; 2: two streams of equal priority */, GML_TUNE_NO_ZERO_SHORTCUT = 4 /* 1: the first pass runs its GEMM like any other */, GML_TUNE_NO_ZEROCOPY = 5 /* 1: results and row lists through device arrays and copies, as on handles too large for the pinned arena */, GML_TUNE_NO_COMPACT = 6 /* 1: the forward GEMMs of objective passes sweep all columns, always (A/B of the column compaction) */, GML_TUNE_BENCH_COMPACT = 7 /* 1: the timing hooks gml_bench_pass* compact too (they sweep all columns by default: the bench headline prices the dense contraction) */, GML_TUNE_SOLVER_COMPACT = 8 /* 1: gml_learn's own passes try the compaction whatever the column count.  By default they do from 4 096 statistics columns on (multi-body problems, thousands of spins: config 5 at c = 1.2 3.5 -> 2.2 s): measured (profiles/r6_compact_ab.txt) -- with the 1 024 columns of the headline problem a node's optimum keeps ~100 noise-level coefficients, a tile's 32 rows cover every column after the third pass: no gain, 1 - 5 % overhead */, GML_NTUNE = 16 };
extern double g_tune[GML_NTUNE];

// the rows' lists of working-set columns (k_cg_tiles): row r's |W| = nw[r] columns, in column order, at FV + t0[r] * T
struct WList {
    const int *FV;
    const long long *t0;
    const int *nw;
    int T;
};

struct CgState {
    double rs, rs0, pHp, rz;
};
struct FaceOut {
    int nfixed, pad; // coordinates of W whose step left the orthant face (fixed at zero, removed from W)
    double mass;     // their share sum |pg_c (d_c - fixed_c)| of ...
    double total;    // ... sum |pg_c d_c| over W
};
struct TrialOut {
    double dd;    // pg . (xt - x)
    double stepn; // |xt - x|_1
    double l1t;   // lambda * sum |xt_c|
    double back;  // F'(xt; x - xt) (k_back)
};

void launch_kind(const DevProblem &d, int order, const int *dnode, int R, uint8_t *kind, hipStream_t st);
void launch_scale_rows(const int *drows, int nrows, const double *dscale, int64_t Qp, double *G, hipStream_t st);
void launch_copy_rows(const int *drows, int nrows, int64_t Qp, const double *s0, double *d0, const double *s1, double *d1, hipStream_t st);
void launch_scale_slots_inv(const int *srow, const int *rowcol, int slot0, int ns, const SlotResult *res, int64_t Qp, double *G, hipStream_t st);
void launch_scale_rows_inv(const int *drows, int nrows, const double *fs, int64_t Qp, double *G, hipStream_t st);
void launch_rows_to_reference(const double *X, int64_t R, int64_t Qp, int64_t P, int64_t node0, int64_t cconst, const int32_t *cols, double *out,
                              hipStream_t st);
void launch_select(const int *drows, int nrows, const double *X, const double *G, const uint8_t *kind, int64_t Qp, double lambda,
                   int max_add, int capW, int capP, double viol_frac, double *PG, int *F, double *gF, double *pgF, SelectOut *out, double *best,
                   double *Xbest, hipStream_t st);
void launch_scatter_dir(const int *drows, int nrows, const int *F, const double *dsol, const int *msz, int capP, int64_t Qp, double *D,
                        hipStream_t st);
void launch_secant(const int *drows, int nrows, const int *F, const int *msz, int cap, const double *X, int64_t Qp, const double *gF, double *H,
                   const long long *hoff, const int *mt, const double *s1, double s2, const double *ynoise, int *Fprev, int *mprev, double *xprev,
                   double *gprev, double *S, double *Y, int *npairs, int64_t pair_stride, int apply_above, hipStream_t st);
void launch_trial(const int *drows, int nrows, const double *X, const double *D, const double *PG, const uint8_t *kind, int64_t Qp,
                  double lambda, const double *alpha, double *Xt, TrialOut *out, double *stepn, hipStream_t st);
void launch_back(const int *drows, int nrows, const double *X, const double *Xt, const double *Gt, const uint8_t *kind, int64_t Qp,
                 double lambda, TrialOut *out, hipStream_t st);
void launch_cg_tiles(const int *drows, int nrows, const double *X, const double *PG, const double *G, const uint8_t *kind, int64_t Qp, int T,
                     const long long *t0, int *FV, double *gV, hipStream_t st);
void launch_tile_apply(int T, const double *Minv, const int *FV, const int *vm, const int *wrow, const int *live, int64_t ntiles, int64_t Qp,
                       const double *Rv, double *Zv, hipStream_t st);
void launch_pcg_init(const int *drows, int nrows, const double *X, const double *PG, const uint8_t *kind, int64_t Qp, double *D, double *Rv,
                     double *Zv, double *Pv, uint8_t *Wm, CgState *cg, hipStream_t st);
void launch_pcg_dir(const int *drows, int nrows, int64_t Qp, const uint8_t *Wm, const double *Rv, const double *Zv, double *Pv, int first,
                    CgState *cg, const WList &wl, hipStream_t st);
void launch_pcg_step(const int *drows, int nrows, const double *G, const uint8_t *Wm, int64_t Qp, const double *s1, double s2, double *Hp,
                     double *D, double *Rv, const double *Pv, CgState *cg, const WList &wl, hipStream_t st);
void launch_pcg_faces(const int *drows, int nrows, const double *X, const double *PG, const uint8_t *kind, int64_t Qp, double *D, uint8_t *Wm,
                      FaceOut *out, hipStream_t st);
void launch_pcg_resid(const int *drows, int nrows, const double *PG, const double *G, int64_t Qp, const double *s1, double s2, const double *Hd,
                      const double *D, const uint8_t *Wm, double *Rv, CgState *cg, hipStream_t st);

} // namespace gml
