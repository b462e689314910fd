// Interface between the host solver (gml_solver.cpp) and its device kernels (gml_solver.hip).
#pragma once
#include "gml_dev.h"

namespace gml {

struct SelectOut {
    double l1;     // lambda * sum |x_c| over the penalised columns
    double worst;  // KKT residual: max |pseudo-gradient|
    double worstW; // the same over the current support
    int m;         // working-set size (Cholesky row), or -|W| for a matrix-free (Newton-CG) row
    int nsupp, nviol;
    int pad;       // matrix-free rows: size of the preconditioner block S gathered in F (0 otherwise)
};
struct CgState {
    double rs, rs0, pHp, rz;
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
void launch_trial(const int *drows, int nrows, const double *X, const double *D, const double *PG, const uint8_t *kind, int64_t Qp,
                  double lambda, const double *alpha, double *Xt, TrialOut *out, hipStream_t st);
void launch_back(const int *drows, int nrows, const double *X, const double *Xt, const double *Gt, const uint8_t *kind, int64_t Qp,
                 double lambda, TrialOut *out, hipStream_t st);
void launch_pcg_init(const int *drows, int nrows, const double *X, const double *PG, const uint8_t *kind, int64_t Qp, const int *F,
                     const int *ms, int capP, double *D, double *Rv, double *nrS, CgState *cg, hipStream_t st);
void launch_pcg_dir(const int *drows, int nrows, int64_t Qp, const int *F, const int *ms, int capP, const double *zS, const double *dinv,
                    const double *Rv, double *Zv, double *Pv, int first, CgState *cg, hipStream_t st);
void launch_pcg_step(const int *drows, int nrows, const double *X, const double *PG, const double *G, const uint8_t *kind, int64_t Qp,
                     const double *s1, double s2, const int *F, const int *ms, int capP, double *Hp, double *D, double *Rv, const double *Pv,
                     double *nrS, CgState *cg, hipStream_t st);

} // namespace gml
