// Internal host-side interface of libgml_hip shared by gml_host.cpp (handles, packing, operator) and gml_solver.cpp
// (the l1 solver behind gml_learn).  Not part of the C ABI.
#pragma once
#include "../../include/gml.h"
#include "gml_dev.h"

#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

int gml_fail(int code, const char *fmt, ...);
#define fail gml_fail
#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

double gml_now_s();
void gml_parallel_for(int64_t n, const std::function<void(int64_t)> &fn); // persistent host worker pool
inline int64_t gml_round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------
// problem handle
// ------------------------------------------------------------------------------------------
struct gml_problem {
    int device = 0;
    hipStream_t st = nullptr;
    int64_t n = 0, K = 0, P = 0, node0 = 0, node1 = 0;
    int order = 2;
    double M = 0;
    gml::DevProblem d{};
    std::vector<int32_t> gkeys; // [Q][ko] subsets of spins (feature keys), -1 padded
    int ko = 1;
    std::vector<int64_t> qoff; // qoff[q] = first column of the size-q subsets
    std::vector<double> wblk;    // wblk[j] = sum of w over the configurations [512 j, 512 j + 512)
    // workspace of the host-pointer operator calls (sized for ws_rows rows)
    int64_t ws_rows = 0;
    double *dTheta = nullptr, *dV = nullptr, *dG = nullptr, *dF = nullptr;
    int64_t dVrows = 0; // rows of the FP64 path's V (gml_ensure_f64)
    int *hCtl = nullptr; // pinned twin of the control block: srow | rowcol | groups
    int *dSrow = nullptr, *dRowcol = nullptr, *dGroups = nullptr;
    double *hTh = nullptr, *hG = nullptr, *hF = nullptr; // pinned staging (ws_rows x Qp, ws_rows)
    // int8-limb path workspace (gml_i8_pass.hip, allocated lazily)
    void *i8ws = nullptr;
    // pinned staging arena of the solver's small control / scalar transfers (gml_solver.cpp), allocated on first use
    char *stage = nullptr;
    size_t stage_bytes = 0;
    // device-pointer operator calls (gml_operator.cpp: objgrad_dev / hessvec_dev): parameter-slot -> column table of the node list
    // of the last multi-body call (kept: an external solver calls with the same nodes every iteration), flag + row mask, and the
    // gradient rows a logRISE Hessian-vector call keeps across its second pass
    std::vector<int64_t> opNodes;
    int32_t *opCols = nullptr;
    int *opFlag = nullptr;
    uint8_t *opSel = nullptr;
    int64_t opSelCap = 0;
    double *opG2 = nullptr;
    int64_t opG2rows = 0;
    // how long building the handle took (gml_problem_ingest_times): host packing, uploads not hidden by it, bit images, total
    double t_ingest[6] = {0, 0, 0, 0, 0, 0};
};

// reference parameter vector of node u <-> internal column layout (pairwise :162, multi-body :94-104)
struct NodeLayout {
    std::vector<int32_t> cols; // reference slot j -> internal column
};
void gml_build_layout(const gml_problem *p, int64_t u, NodeLayout &L);

// GML_PREC_AUTO -> an int8-limb precision, decided on the whole problem, not on the rows of one call or one shard (the arithmetic does
// not depend on the GPU count); -1: unknown value.  tol: the KKT tolerance of a solve, 0 for operator calls.
//  * operator calls -- gml_objgrad_batch is what an external solver registers in place of the reference's Float64 obj / grad pair --
//    and solves below 2e-10 take the FP64-grade limbs (the 38/31-bit pass would stall at the noise floor of its gradient and finish on
//    the FP64-MFMA path; the 54/47-bit pass gets there directly);
//  * small problems (samples x parameters x spins <= 2^28: the reference's own fixtures, the README example) take them too, at any
//    tolerance: every kernel is launch-bound there, the wide pass costs the same, and it converges in as few iterations as Float64.
//    (Rounds 1-4 sent these to the FP64-MFMA path, which round 5's int8 direction phase left 2-10x behind: README example 1.19 ms
//    against 0.63, n = 64 / K = 2e4 13.1 ms against 1.3 -- profiles/r5_auto_probe.txt.  GML_PREC_F64 stays available by name.)
//  * everything else: the 38/31-bit pass.
inline int gml_resolve_precision(const gml_problem *p, int precision, double tol = 0.0) {
    if (precision == GML_PREC_AUTO) {
        if (p->d.Qfp > ((int64_t)1 << 21)) return GML_PREC_I8X; // (i8w: up to 2^21 statistics columns)
        const bool small = (double)p->K * (double)p->P * (double)p->n <= 268435456.0;
        return (tol <= 0.0 || tol < 2e-10 || small) ? GML_PREC_I8W : GML_PREC_I8X;
    }
    return precision == GML_PREC_F64 || precision == GML_PREC_I8X || precision == GML_PREC_I8W ? precision : -1;
}
// the int8-limb precisions (fixed point on the matrix cores: slots, limb planes, tracked scales)
inline bool gml_is_i8(int precision) { return precision == GML_PREC_I8X || precision == GML_PREC_I8W; }

// handles for several node ranges / devices from one host histogram: packed once, the bits copied to every device
int gml_create_parts(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, int order,
                     const std::vector<std::pair<int64_t, int64_t>> &ranges, const std::vector<int> &devices,
                     std::vector<gml_problem *> &parts);

// shared between the host files (gml_host.cpp: core; gml_ingest.cpp; gml_sampled.cpp; gml_operator.cpp)
int64_t gml_binom(int64_t n, int64_t k);
bool gml_next_comb(std::vector<int> &idx, int64_t n);
void gml_node_cols(const gml_problem *p, int64_t u, std::vector<int32_t> &cols);
int gml_check_create_args(int64_t K, int64_t n, int order, int64_t node0, int64_t node1, int device);
gml_problem *gml_new_problem(int64_t K, int64_t n, double M, int order, int64_t node0, int64_t node1, int device);
int gml_create_from_device_bytes(gml_problem *p, int8_t *dbytes, bool spin_major, int64_t ld, const double *counts, gml_problem **out,
                                 bool dedupe = false);

int gml_ensure_ws(gml_problem *p, int64_t rows);
int gml_ensure_f64(gml_problem *p, int64_t vrows); // byte images + V [vrows][Kp] of the FP64 path
