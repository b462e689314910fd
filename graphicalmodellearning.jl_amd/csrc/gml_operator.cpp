// libgml_hip, host side: the operator boundary -- gml_objgrad_batch / gml_hessvec_batch (GraphicalModelLearning.jl:191-208,
// :221-233) and the timing hooks of the benchmark: orchestration of the device passes for host-pointer callers.
#include "gml_internal.h"
#include "gml_solver.h"
#include "gml_pack.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace gml;

static double now_s() { return gml_now_s(); }
// operator calls compact the forward GEMM (sparse rows are what an l1 solver evaluates); the timing hooks sweep all columns unless asked
static bool op_compact() { return g_tune[GML_TUNE_NO_COMPACT] == 0; }
static bool bench_compact() { return g_tune[GML_TUNE_NO_COMPACT] == 0 && g_tune[GML_TUNE_BENCH_COMPACT] > 0; }
static void parallel_for(int64_t n, const std::function<void(int64_t)> &fn) { gml_parallel_for(n, fn); }
static int64_t round_up(int64_t a, int64_t b) { return gml_round_up(a, b); }
static void build_layout(const gml_problem *p, int64_t u, NodeLayout &L) { gml_build_layout(p, u, L); }

// ------------------------------------------------------------------------------------------
// device pass orchestration for the host-pointer operator calls (gml_objgrad_batch, gml_bench_pass*)
// ------------------------------------------------------------------------------------------
int gml_ensure_ws(gml_problem *p, int64_t rows) {
    const int64_t Rp = round_up(rows, 32);
    if (Rp <= p->ws_rows) return GML_OK;
    void *ptrs[] = {p->dTheta, p->dV, p->dG, p->dF, p->dSrow};
    for (void *q : ptrs)
        if (q) (void)dev_free(q);
    void *hptrs[] = {p->hTh, p->hG, p->hF, p->hCtl};
    for (void *q : hptrs)
        if (q) (void)hipHostFree(q);
    p->hTh = p->hG = p->hF = nullptr;
    p->hCtl = nullptr;
    p->dTheta = p->dV = p->dG = p->dF = nullptr;
    p->dSrow = p->dRowcol = p->dGroups = nullptr;
    p->ws_rows = 0;
    size_t freeb = 0, totalb = 0;
    HIPCHK(dev_mem_info(&freeb, &totalb));
    const double need = 2.0 * Rp * p->d.Qp * 8.0;
    if (need > 0.9 * (double)freeb)
        return fail(GML_ENOMEM, "workspace of %.1f GB for %lld rows does not fit in %.1f GB free HBM", need / 1e9,
                    (long long)Rp, freeb / 1e9);
    HIPCHK(dev_malloc(&p->dTheta, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(dev_malloc(&p->dG, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(dev_malloc(&p->dF, sizeof(double) * Rp));
    // control block: srow [Rp] | rowcol [Rp] | active tiles, padded with -1 [Rp/32 + 4]; one pinned twin, one upload per pass
    const int64_t nctl = 2 * Rp + Rp / 32 + 4;
    HIPCHK(dev_malloc(&p->dSrow, sizeof(int) * nctl));
    p->dRowcol = p->dSrow + Rp;
    p->dGroups = p->dRowcol + Rp;
    HIPCHK(hipHostMalloc(&p->hCtl, sizeof(int) * nctl));
    HIPCHK(hipHostMalloc(&p->hTh, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipHostMalloc(&p->hG, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipHostMalloc(&p->hF, sizeof(double) * Rp));
    HIPCHK(hipMemsetAsync(p->dTheta, 0, sizeof(double) * Rp * p->d.Qp, p->st));
    p->ws_rows = Rp;
    return GML_OK;
}
static int ensure_ws(gml_problem *p, int64_t rows) { return gml_ensure_ws(p, rows); }

// What only the FP64 path needs: the two byte images of the design matrix and V [vrows][Kp].
int gml_ensure_f64(gml_problem *p, int64_t vrows) {
    DevProblem &d = p->d;
    size_t freeb = 0, totalb = 0;
    if (!d.Xt) { // feature-major byte image: the rows the FP64 Hessian kernel gathers (the GEMM kernels read the bit images)
        HIPCHK(dev_mem_info(&freeb, &totalb));
        if ((double)d.Kp * d.Qp > 0.9 * (double)freeb)
            return fail(GML_EUNSUPPORTED, "the FP64 path needs a %.1f GB byte image of the design matrix: use precision i8x or i8w",
                        (double)d.Kp * d.Qp / 1e9);
        HIPCHK(dev_malloc(&d.Xt, (size_t)d.Kp * d.Qp));
        HIPCHK(hipMemsetAsync(d.Xt, 0, (size_t)d.Kp * d.Qp, p->st));
        launch_expand_xt(d, d.Xt, p->st);
        HIPCHK(hipMemsetAsync(d.Xt + d.cconst * d.Kp, 1, (size_t)p->K, p->st)); // the constant statistic
    }
    vrows = round_up(vrows, 32);
    if (!p->dV || p->dVrows < vrows) {
        if (p->dV) (void)dev_free(p->dV);
        p->dV = nullptr;
        p->dVrows = 0;
        HIPCHK(dev_mem_info(&freeb, &totalb));
        if ((double)vrows * d.Kp * 8.0 > 0.9 * (double)freeb)
            return fail(GML_ENOMEM, "FP64 workspace of %.1f GB does not fit: use precision i8x", (double)vrows * d.Kp * 8.0 / 1e9);
        HIPCHK(dev_malloc(&p->dV, sizeof(double) * vrows * d.Kp));
        HIPCHK(hipMemsetAsync(p->dV, 0, sizeof(double) * vrows * d.Kp, p->st));
        p->dVrows = vrows;
    }
    return GML_OK;
}

static thread_local bool t_bench_call = false; // device_pass on behalf of gml_bench_pass (its compaction follows the bench knob)
struct RowSet {
    int64_t R = 0;
    std::vector<int64_t> node; // node id per row
};

// One device pass over the rows flagged in `act` (size R).  theta: R x Qp host, internal
// layout.  Writes f[r], and g (R x Qp) when want_grad, for the active rows only.
static int device_pass(gml_problem *p, const RowSet &rs, const std::vector<uint8_t> &act, const double *theta,
                       int form, int precision, bool want_grad, double *f, double *g, gml_stats *stats,
                       float *ms /* [2]: fwd, bwd or NULL */ = nullptr,
                       const std::vector<double> *tau_ovr = nullptr /* Rp per-row tau of the rescaled re-run below */,
                       int depth = 0, bool f64_fallback = false /* rows the fixed point cannot hold go to the FP64 path (precision auto) */) {
    const int64_t R = rs.R, Qp = p->d.Qp;
    const int64_t Rp = round_up(R, 32);
    int rc = ensure_ws(p, R);
    if (rc) return rc;
    std::vector<int> rowcol((size_t)Rp, -1), groups;
    int64_t nact = 0;
    for (int64_t r = 0; r < R; ++r)
        if (act[r]) {
            rowcol[r] = (int)rs.node[r]; // the node whose sign bits the row uses
            ++nact;
        }
    if (nact == 0) return GML_OK;
    for (int64_t gidx = 0; gidx < Rp / 32; ++gidx) {
        bool any = false;
        for (int64_t r = gidx * 32; r < std::min(R, (gidx + 1) * 32); ++r) any |= (act[r] != 0);
        if (any) groups.push_back((int)gidx);
    }
    const double t0 = now_s();
    hipStream_t st = p->st;
    // one contiguous upload covering the active groups, through the pinned staging buffer
    const int64_t ra = (int64_t)groups.front() * 32, rb = std::min(R, (int64_t)groups.back() * 32 + 32);
    parallel_for((rb - ra + 31) / 32, [&](int64_t b) {
        const int64_t r0 = ra + b * 32, r1 = std::min(rb, r0 + 32);
        std::memcpy(p->hTh + r0 * Qp, theta + r0 * Qp, sizeof(double) * (r1 - r0) * Qp);
    });
    HIPCHK(hipMemcpyAsync(p->dTheta + ra * Qp, p->hTh + ra * Qp, sizeof(double) * (rb - ra) * Qp, hipMemcpyHostToDevice, st));
    // control block (slot = row here: identity map, row -> node, active tiles) through its pinned twin: one upload
    std::vector<int> gpad = groups;
    while (gpad.size() % 4) gpad.push_back(-1);
    const int64_t W = p->ws_rows;
    for (int64_t r = 0; r < Rp; ++r) p->hCtl[r] = (int)r;
    std::memcpy(p->hCtl + W, rowcol.data(), sizeof(int) * Rp);
    std::memcpy(p->hCtl + 2 * W, gpad.data(), sizeof(int) * gpad.size());
    HIPCHK(hipMemcpyAsync(p->dSrow, p->hCtl, sizeof(int) * (2 * W + gpad.size()), hipMemcpyHostToDevice, st));
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    if (ms)
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    double *dOvr = nullptr;
    const bool wide = precision == GML_PREC_I8W;
    const bool compact_ok = t_bench_call ? bench_compact() : op_compact();
    if (gml_is_i8(precision)) {
        if (tau_ovr) {
            HIPCHK(dev_malloc(&dOvr, sizeof(double) * Rp));
            HIPCHK(hipMemcpyAsync(dOvr, tau_ovr->data(), sizeof(double) * Rp, hipMemcpyHostToDevice, st));
        }
        std::string err;
        gml::I8Pass a{};
        a.theta = p->dTheta;
        a.srow = p->dSrow;
        a.rowcol = p->dRowcol;
        a.groups = p->dGroups;
        a.ngroups = (int)groups.size();
        a.slot0 = 0;
        a.slot1 = (int)Rp;
        a.form = form;
        a.want_grad = want_grad;
        a.F = p->dF;
        a.G = p->dG;
        a.tauovr = dOvr;
        a.wide = wide;
        a.compact = compact_ok;
        rc = gml::i8_pass(&p->i8ws, p->d, W, a, st, ms ? ev : nullptr, &err);
        if (rc) {
            if (dOvr) (void)dev_free(dOvr);
            return fail(rc, "%s", err.c_str());
        }
    } else {
        rc = gml_ensure_f64(p, p->ws_rows);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(p->dF, 0, sizeof(double) * Rp, st));
        if (want_grad) HIPCHK(hipMemsetAsync(p->dG, 0, sizeof(double) * Rp * Qp, st));
        if (ms) HIPCHK(hipEventRecord(ev[0], st));
        launch_fwd_f64(p->d, p->dTheta, p->dRowcol, p->dGroups, (int)gpad.size(), form, p->dV, p->dF, st);
        if (ms) HIPCHK(hipEventRecord(ev[1], st));
        if (want_grad) launch_bwd_f64(p->d, p->dV, p->dGroups, (int)groups.size(), p->dG, st);
        if (ms) HIPCHK(hipEventRecord(ev[2], st));
    }
    HIPCHK(hipGetLastError());
    double *fh = p->hF;
    HIPCHK(hipMemcpyAsync(fh, p->dF, sizeof(double) * Rp, hipMemcpyDeviceToHost, st));
    if (want_grad)
        HIPCHK(hipMemcpyAsync(p->hG + ra * Qp, p->dG + ra * Qp, sizeof(double) * (rb - ra) * Qp, hipMemcpyDeviceToHost, st));
    std::vector<double> tauh;
    std::vector<unsigned> mmaxh;
    const bool i8exp = gml_is_i8(precision) && form != GML_RPLE;
    if (i8exp) {
        const double *tau = nullptr;
        const unsigned *mm = nullptr;
        gml::i8_slot_results(p->i8ws, 0, &tau, &mm);
        tauh.resize((size_t)Rp);
        mmaxh.resize((size_t)Rp);
        HIPCHK(hipMemcpyAsync(tauh.data(), tau, sizeof(double) * Rp, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(mmaxh.data(), mm, sizeof(unsigned) * Rp, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (dOvr) (void)dev_free(dOvr);
    if (ms) {
        HIPCHK(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
        HIPCHK(hipEventElapsedTime(&ms[1], ev[1], ev[2]));
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    for (int64_t r = 0; r < R; ++r)
        if (act[r]) f[r] = fh[r];
    if (want_grad)
        parallel_for((int64_t)groups.size(), [&](int64_t a) {
            for (int i = 0; i < 32; ++i) {
                const int64_t r = (int64_t)groups[a] * 32 + i;
                if (r < R && act[r]) std::memcpy(g + r * Qp, p->hG + r * Qp, sizeof(double) * Qp);
            }
        });
    if (stats) {
        stats->t_pass += now_s() - t0;
        stats->node_evals += nact;
        if (want_grad) ++stats->passes;
        else ++stats->forward_passes;
    }
    if (i8exp) {
        // Dynamic range of the fixed-point V: tau_r was derived from the bound w_max exp(sum_j |theta_rj|).  When
        // the largest |V_rk| actually seen is more than 8 bits (i8w: 4 bits) below that bound (dense theta), re-run the row with
        // tau_r taken from it: (mmax + 1) tau bounds every |V_rk| rigorously, so the re-run cannot overflow.
        std::vector<uint8_t> again((size_t)R, 0);
        std::vector<double> ovr((size_t)Rp, 0.0);
        int64_t nagain = 0;
        // (the FP64-grade pass is stricter: it is re-run as soon as four of its 47 bits would go unused, so that its error stays
        // at 2^-43 of the largest weight whatever the bound was)
        const unsigned mm_min = wide ? (1u << 27) : (1u << 23);
        for (int64_t r = 0; r < R; ++r)
            if (act[r] && mmaxh[r] < mm_min) {
                again[r] = 1;
                ovr[r] = ((double)mmaxh[r] + 1.0) * gml::i8_mmax_unit(wide) * tauh[r] * (1.0 + 1e-12) / gml::i8_vdiv(wide);
                ++nagain;
            }
        if (nagain > 0) {
            if (depth >= 6) {
                // Six rescalings did not bring the row's largest weight into the planes: energies spread over hundreds of units
                // (|theta|_1 in the hundreds -- a far trial point of an external solver).  The reference's Float64 operator
                // (GraphicalModelLearning.jl:191-197) returns a number there, so `auto` -- its stand-in -- evaluates these rows on
                // the FP64 path; a caller who named an int8-limb precision gets the error.
                if (f64_fallback) return device_pass(p, rs, again, theta, form, GML_PREC_F64, want_grad, f, g, stats);
                return fail(GML_EUNSUPPORTED, "precision %s: the weights exp(-E) of a row underflow its fixed-point range; use precision f64 (or auto)",
                            wide ? "i8w" : "i8x");
            }
            return device_pass(p, rs, again, theta, form, precision, want_grad, f, g, stats, nullptr, &ovr, depth + 1, f64_fallback);
        }
    }
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// The same pass for callers whose rows live in HBM (device pointers for theta / f / g): the reference-order rows are
// scattered into the internal layout by a kernel, the pass runs, a kernel gathers the results back -- no staging, no PCIe.
// The only host round trip is the one the dynamic-range check needs (two small per-row arrays).
// ------------------------------------------------------------------------------------------
static bool is_device_ptr(const void *q) {
    if (!q) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, q) == hipSuccess) return attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return false;
}

// slot -> column table of a multi-body node list on the device (NULL for pairwise: closed form in the kernels)
static int dev_cols(gml_problem *p, int64_t nrows, const int64_t *nodes, const int32_t **out) {
    *out = nullptr;
    if (p->order == 2) return GML_OK;
    if (p->opCols && (int64_t)p->opNodes.size() == nrows && std::equal(nodes, nodes + nrows, p->opNodes.begin())) {
        *out = p->opCols;
        return GML_OK;
    }
    if (p->opCols) (void)dev_free(p->opCols);
    p->opCols = nullptr;
    p->opNodes.clear();
    const int64_t P = p->P;
    std::vector<int32_t> cols((size_t)nrows * P);
    parallel_for(nrows, [&](int64_t r) {
        NodeLayout L;
        build_layout(p, nodes[r], L);
        std::memcpy(cols.data() + (size_t)r * P, L.cols.data(), sizeof(int32_t) * P);
    });
    HIPCHK(dev_malloc(&p->opCols, sizeof(int32_t) * (size_t)nrows * P));
    HIPCHK(hipMemcpyAsync(p->opCols, cols.data(), sizeof(int32_t) * (size_t)nrows * P, hipMemcpyHostToDevice, p->st));
    HIPCHK(hipStreamSynchronize(p->st)); // cols is a local
    p->opNodes.assign(nodes, nodes + nrows);
    *out = p->opCols;
    return GML_OK;
}

static int dev_scratch(gml_problem *p, int64_t Rp) {
    if (!p->opFlag) HIPCHK(dev_malloc(&p->opFlag, sizeof(int) * 4));
    if (p->opSelCap < Rp) {
        if (p->opSel) (void)dev_free(p->opSel);
        p->opSel = nullptr;
        p->opSelCap = 0;
        HIPCHK(dev_malloc(&p->opSel, (size_t)Rp));
        p->opSelCap = Rp;
    }
    return GML_OK;
}

// control block of a pass over the rows flagged in act (slot = row): srow | rowcol | active tiles; returns the tiles
static int upload_ctl(gml_problem *p, int64_t R, const int64_t *nodes, const std::vector<uint8_t> &act, std::vector<int> &groups, int *npad_out) {
    const int64_t Rp = round_up(R, 32), W = p->ws_rows;
    groups.clear();
    for (int64_t r = 0; r < Rp; ++r) {
        p->hCtl[r] = (int)r;
        p->hCtl[W + r] = (r < R && act[r]) ? (int)nodes[r] : -1;
    }
    for (int64_t gidx = 0; gidx < Rp / 32; ++gidx) {
        bool any = false;
        for (int64_t r = gidx * 32; r < std::min(R, (gidx + 1) * 32); ++r) any |= (act[r] != 0);
        if (any) groups.push_back((int)gidx);
    }
    int npad = 0;
    for (size_t g = 0; g < groups.size() || (npad % 4); ++g, ++npad) p->hCtl[2 * W + g] = g < groups.size() ? groups[g] : -1;
    *npad_out = npad;
    HIPCHK(hipMemcpyAsync(p->dSrow, p->hCtl, sizeof(int) * (2 * W + npad), hipMemcpyHostToDevice, p->st));
    return GML_OK;
}

// dtheta [R][ld] in, df [R] / dg [R][ldg] out (either may be NULL); raw: f and g as the pass leaves them (logRISE: Z and grad Z, no
// log / division)
static int pass_dev(gml_problem *p, int64_t R, const int64_t *nodes, const std::vector<uint8_t> &act, const double *dtheta, int64_t ld,
                    int form, int precision, bool want_grad, double *df, double *dg, int64_t ldg, const int32_t *dcols,
                    const std::vector<double> *tau_ovr, int depth, bool f64_fallback, bool raw = false) {
    const int64_t Qp = p->d.Qp, P = p->P, Rp = round_up(R, 32);
    int rc = ensure_ws(p, R);
    if (rc) return rc;
    rc = dev_scratch(p, Rp);
    if (rc) return rc;
    hipStream_t st = p->st;
    std::vector<int> groups;
    int npad = 0;
    rc = upload_ctl(p, R, nodes, act, groups, &npad);
    if (rc) return rc;
    if (groups.empty()) return GML_OK;
    const int64_t W = p->ws_rows;
    if (depth == 0) { // (a re-run finds the rows where the first pass put them)
        HIPCHK(hipMemsetAsync(p->opFlag, 0, sizeof(int) * 4, st));
        HIPCHK(hipMemsetAsync(p->dTheta, 0, sizeof(double) * Rp * Qp, st));
        launch_ref_to_internal(dtheta, ld, R, P, Qp, p->dRowcol, p->d.cconst, dcols, p->dTheta, p->opFlag, st);
    }
    double *dOvr = nullptr;
    const bool wide = precision == GML_PREC_I8W;
    const bool compact_ok = t_bench_call ? bench_compact() : op_compact();
    if (gml_is_i8(precision)) {
        if (tau_ovr) {
            HIPCHK(dev_malloc(&dOvr, sizeof(double) * Rp));
            HIPCHK(hipMemcpyAsync(dOvr, tau_ovr->data(), sizeof(double) * Rp, hipMemcpyHostToDevice, st));
        }
        std::string err;
        gml::I8Pass a{};
        a.theta = p->dTheta;
        a.srow = p->dSrow;
        a.rowcol = p->dRowcol;
        a.groups = p->dGroups;
        a.ngroups = (int)groups.size();
        a.slot0 = 0;
        a.slot1 = (int)Rp;
        a.form = form;
        a.want_grad = want_grad;
        a.F = p->dF;
        a.G = p->dG;
        a.tauovr = dOvr;
        a.wide = wide;
        a.compact = compact_ok;
        rc = gml::i8_pass(&p->i8ws, p->d, W, a, st, nullptr, &err);
        if (rc) {
            if (dOvr) (void)dev_free(dOvr);
            return fail(rc, "%s", err.c_str());
        }
    } else {
        rc = gml_ensure_f64(p, p->ws_rows);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(p->dF, 0, sizeof(double) * Rp, st));
        if (want_grad) HIPCHK(hipMemsetAsync(p->dG, 0, sizeof(double) * Rp * Qp, st));
        launch_fwd_f64(p->d, p->dTheta, p->dRowcol, p->dGroups, npad, form, p->dV, p->dF, st);
        if (want_grad) launch_bwd_f64(p->d, p->dV, p->dGroups, (int)groups.size(), p->dG, st);
    }
    HIPCHK(hipGetLastError());
    std::vector<double> tauh;
    std::vector<unsigned> mmaxh;
    int bad[4] = {0, 0, 0, 0};
    const bool i8exp = gml_is_i8(precision) && form != GML_RPLE;
    if (i8exp) {
        const double *tau = nullptr;
        const unsigned *mm = nullptr;
        gml::i8_slot_results(p->i8ws, 0, &tau, &mm);
        tauh.resize((size_t)Rp);
        mmaxh.resize((size_t)Rp);
        HIPCHK(hipMemcpyAsync(tauh.data(), tau, sizeof(double) * Rp, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(mmaxh.data(), mm, sizeof(unsigned) * Rp, hipMemcpyDeviceToHost, st));
    }
    if (depth == 0) HIPCHK(hipMemcpyAsync(bad, p->opFlag, sizeof bad, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (dOvr) (void)dev_free(dOvr);
    if (bad[0]) return fail(GML_EINVAL, "theta contains a non-finite value");
    // dynamic range of the fixed-point V: as device_pass
    std::vector<uint8_t> again((size_t)R, 0), sel((size_t)Rp, 0);
    std::vector<double> ovr((size_t)Rp, 0.0);
    int64_t nagain = 0;
    if (i8exp) {
        const unsigned mm_min = wide ? (1u << 27) : (1u << 23);
        for (int64_t r = 0; r < R; ++r)
            if (act[r] && mmaxh[r] < mm_min) {
                again[r] = 1;
                ovr[r] = ((double)mmaxh[r] + 1.0) * gml::i8_mmax_unit(wide) * tauh[r] * (1.0 + 1e-12) / gml::i8_vdiv(wide);
                ++nagain;
            }
    }
    for (int64_t r = 0; r < R; ++r) sel[r] = act[r] && !again[r];
    // the rows this pass settled go back to the caller's arrays (a re-run of the others overwrites the workspace rows of its tiles)
    HIPCHK(hipMemcpyAsync(p->opSel, sel.data(), (size_t)Rp, hipMemcpyHostToDevice, st));
    launch_internal_to_ref(p->dG, p->dF, R, Qp, P, ldg, p->dRowcol, p->opSel, p->d.cconst, dcols, (form == GML_LOGRISE && !raw) ? 1 : 0, df,
                           want_grad ? dg : nullptr, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st)); // (sel is a local; and the call returns with its outputs written)
    if (nagain > 0) {
        if (depth >= 6) {
            if (f64_fallback) return pass_dev(p, R, nodes, again, dtheta, ld, form, GML_PREC_F64, want_grad, df, dg, ldg, dcols, nullptr, 0, false, raw);
            return fail(GML_EUNSUPPORTED, "precision %s: the weights exp(-E) of a row underflow its fixed-point range; use precision f64 (or auto)",
                        wide ? "i8w" : "i8x");
        }
        return pass_dev(p, R, nodes, again, dtheta, ld, form, precision, want_grad, df, dg, ldg, dcols, &ovr, depth + 1, f64_fallback, raw);
    }
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// gml_objgrad_batch: the operator (:191-208, :221-233)
// ------------------------------------------------------------------------------------------
extern "C" int gml_objgrad_batch(gml_problem *p, int formulation, int precision, int64_t nrows,
                                 const int64_t *nodes, const double *theta, int64_t ld, double *f, double *g) {
    if (!p || !nodes || !theta || !f) return fail(GML_EINVAL, "NULL argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (nrows <= 0) return fail(GML_EINVAL, "nrows must be positive");
    if (ld < p->P) return fail(GML_EINVAL, "ld %lld smaller than the %lld parameters per node", (long long)ld, (long long)p->P);
    const bool asked_auto = precision == GML_PREC_AUTO;
    {
        const int asked = precision;
        precision = gml_resolve_precision(p, asked);
        if (precision < 0) return fail(GML_EINVAL, "unknown precision %d", asked);
    }
    for (int64_t r = 0; r < nrows; ++r)
        if (nodes[r] < 0 || nodes[r] >= p->n) return fail(GML_EINVAL, "node id %lld out of range", (long long)nodes[r]);
    HIPCHK(hipSetDevice(p->device));
    const int64_t Qp = p->d.Qp, P = p->P;
    {
        // rows resident in HBM: theta, f and g (if given) are device pointers -- all of them or none
        const bool td = is_device_ptr(theta), fd = is_device_ptr(f), gd = g ? is_device_ptr(g) : td;
        if (td != fd || td != gd) return fail(GML_EINVAL, "theta, f and g must be all host or all device pointers");
        if (td) {
            const int32_t *dcols = nullptr;
            int rcd = dev_cols(p, nrows, nodes, &dcols);
            if (rcd) return rcd;
            std::vector<uint8_t> act((size_t)nrows, 1);
            return pass_dev(p, nrows, nodes, act, theta, ld, formulation, precision, g != nullptr, f, g, ld, dcols, nullptr, 0, asked_auto);
        }
    }
    RowSet rs;
    rs.R = nrows;
    rs.node.assign(nodes, nodes + nrows);
    std::vector<NodeLayout> lay((size_t)nrows);
    std::vector<double> Th((size_t)nrows * Qp, 0.0), Gi(g ? (size_t)nrows * Qp : 0);
    std::vector<uint8_t> badrow((size_t)nrows, 0);
    parallel_for(nrows, [&](int64_t r) {
        build_layout(p, nodes[r], lay[r]);
        for (int64_t j = 0; j < P; ++j) {
            const double v = theta[r * ld + j];
            if (!std::isfinite(v)) badrow[r] = 1;
            Th[(size_t)r * Qp + lay[r].cols[j]] = v;
        }
    });
    for (int64_t r = 0; r < nrows; ++r)
        if (badrow[r]) return fail(GML_EINVAL, "theta of row %lld contains a non-finite value", (long long)r);
    std::vector<uint8_t> act((size_t)nrows, 1);
    std::vector<double> fv((size_t)nrows);
    int rc = device_pass(p, rs, act, Th.data(), formulation, precision, g != nullptr, fv.data(), Gi.data(), nullptr, nullptr, nullptr, 0, asked_auto);
    if (rc) return rc;
    parallel_for(nrows, [&](int64_t r) {
        double z = fv[r];
        if (formulation == GML_LOGRISE) f[r] = std::log(z);
        else f[r] = z;
        if (g)
            for (int64_t j = 0; j < P; ++j) {
                double v = Gi[(size_t)r * Qp + lay[r].cols[j]];
                if (formulation == GML_LOGRISE) v /= z;
                g[r * ld + j] = v;
            }
    });
    return GML_OK;
}

// gml_hessvec_batch for rows resident in HBM: the objective pass at theta leaves the curvature weights in the slots, the directions are
// scattered into the internal layout, the Hessian-vector pass runs, a kernel writes hv (logRISE: with the rank-one correction)
static int hv_second_pass(gml_problem *p, int64_t Rp, int ngroups, int npad, int formulation, int precision);
static int hessvec_dev(gml_problem *p, int form, int precision, int64_t nrows, const int64_t *nodes, const double *dtheta, const double *dvec,
                       int64_t ld, double *dhv) {
    const int64_t Qp = p->d.Qp, P = p->P, Rp = round_up(nrows, 32);
    const int32_t *dcols = nullptr;
    int rc = dev_cols(p, nrows, nodes, &dcols);
    if (rc) return rc;
    double *G2 = nullptr, *F2 = nullptr;
    if (form == GML_LOGRISE) { // Z and grad Z of every row, in the reference order, kept across the second pass
        if (p->opG2rows < nrows) {
            if (p->opG2) (void)dev_free(p->opG2);
            p->opG2 = nullptr;
            p->opG2rows = 0;
            HIPCHK(dev_malloc(&p->opG2, sizeof(double) * ((size_t)nrows * P + (size_t)Rp)));
            p->opG2rows = nrows;
        }
        G2 = p->opG2;
        F2 = p->opG2 + (size_t)p->opG2rows * P;
    }
    std::vector<uint8_t> act((size_t)nrows, 1);
    rc = pass_dev(p, nrows, nodes, act, dtheta, ld, form, precision, true, F2, G2, P, dcols, nullptr, 0, false, true);
    if (rc) return rc;
    hipStream_t st = p->st;
    std::vector<int> groups;
    int npad = 0;
    rc = upload_ctl(p, nrows, nodes, act, groups, &npad);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(p->opFlag, 0, sizeof(int) * 4, st));
    HIPCHK(hipMemsetAsync(p->dTheta, 0, sizeof(double) * Rp * Qp, st));
    launch_ref_to_internal(dvec, ld, nrows, P, Qp, p->dRowcol, p->d.cconst, dcols, p->dTheta, p->opFlag, st);
    rc = hv_second_pass(p, Rp, (int)groups.size(), npad, form, precision);
    if (rc) return rc;
    launch_hv_to_ref(p->dG, G2, F2, nrows, Qp, P, ld, p->dRowcol, p->d.cconst, dcols, form == GML_LOGRISE ? 1 : 0, dvec, dhv, st);
    HIPCHK(hipGetLastError());
    int bad[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(bad, p->opFlag, sizeof bad, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (bad[0]) return fail(GML_EINVAL, "vec contains a non-finite value");
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// gml_hessvec_batch: curvature operator, H_u(theta) v for many nodes at once
// ------------------------------------------------------------------------------------------
extern "C" int gml_hessvec_batch(gml_problem *p, int formulation, int64_t nrows, const int64_t *nodes, const double *theta,
                                 const double *vec, int64_t ld, double *hv) {
    return gml_hessvec_batch_prec(p, formulation, GML_PREC_I8X, nrows, nodes, theta, vec, ld, hv);
}

// the second pass of a Hessian-vector call: the directions are in dTheta (internal layout), the control block lists all rows;
// leaves H p (RISE, RPLE) / Hess Z p (logRISE) in dG
static int hv_second_pass(gml_problem *p, int64_t Rp, int ngroups, int npad, int formulation, int precision) {
    hipStream_t st = p->st;
    if (precision == GML_PREC_F64) {
        HIPCHK(hipMemsetAsync(p->dG, 0, sizeof(double) * Rp * p->d.Qp, st));
        launch_fwd_f64(p->d, p->dTheta, p->dRowcol, p->dGroups, npad, formulation + 4, p->dV, p->dF, st); // V <- h (x . p), in place
        launch_bwd_f64(p->d, p->dV, p->dGroups, ngroups, p->dG, st);
        HIPCHK(hipGetLastError());
        return GML_OK;
    }
    gml::I8Pass a{};
    a.theta = p->dTheta;
    a.srow = p->dSrow;
    a.rowcol = p->dRowcol;
    a.groups = p->dGroups;
    a.ngroups = ngroups;
    a.slot0 = 0;
    a.slot1 = (int)Rp;
    a.form = formulation;
    a.want_grad = true;
    a.F = nullptr;
    a.G = p->dG;
    a.hv = 1;
    a.vmap = p->dSrow;
    std::string err;
    const int rc = gml::i8_pass(&p->i8ws, p->d, p->ws_rows, a, st, nullptr, &err);
    if (rc) return fail(rc, "%s", err.c_str());
    return GML_OK;
}

static int hv_precision(int precision) {
    if (precision == GML_PREC_AUTO) return GML_PREC_I8X; // (an inexact Newton step needs no more; name f64 for the Float64-grade operator)
    if (precision == GML_PREC_I8X || precision == GML_PREC_F64) return precision;
    return -1;
}

extern "C" int gml_hessvec_batch_prec(gml_problem *p, int formulation, int precision, int64_t nrows, const int64_t *nodes, const double *theta,
                                      const double *vec, int64_t ld, double *hv) {
    if (!p || !nodes || !theta || !vec || !hv) return fail(GML_EINVAL, "NULL argument");
    {
        const int asked = precision;
        precision = hv_precision(asked);
        if (precision < 0)
            return fail(asked == GML_PREC_I8W ? GML_EUNSUPPORTED : GML_EINVAL,
                        "Hessian-vector products run at precision i8x (31-bit curvature weights, ~1e-8) or f64 (FP64 MFMA, 1e-12); got %d", asked);
    }
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (nrows <= 0) return fail(GML_EINVAL, "nrows must be positive");
    if (ld < p->P) return fail(GML_EINVAL, "ld %lld smaller than the %lld parameters per node", (long long)ld, (long long)p->P);
    for (int64_t r = 0; r < nrows; ++r)
        if (nodes[r] < 0 || nodes[r] >= p->n) return fail(GML_EINVAL, "node id %lld out of range", (long long)nodes[r]);
    HIPCHK(hipSetDevice(p->device));
    const int64_t Qp = p->d.Qp, P = p->P, Rp = round_up(nrows, 32);
    {
        const bool td = is_device_ptr(theta), vd = is_device_ptr(vec), hd = is_device_ptr(hv);
        if (td != vd || td != hd) return fail(GML_EINVAL, "theta, vec and hv must be all host or all device pointers");
        if (td) return hessvec_dev(p, formulation, precision, nrows, nodes, theta, vec, ld, hv);
    }
    // 1. objective + gradient pass at theta: leaves the curvature weights (limb planes of V) in the slots 0..nrows-1
    RowSet rs;
    rs.R = nrows;
    rs.node.assign(nodes, nodes + nrows);
    std::vector<NodeLayout> lay((size_t)nrows);
    std::vector<double> Th((size_t)nrows * Qp, 0.0), Vc((size_t)nrows * Qp, 0.0), Gi((size_t)nrows * Qp), Hv((size_t)nrows * Qp);
    std::vector<uint8_t> badrow((size_t)nrows, 0);
    parallel_for(nrows, [&](int64_t r) {
        build_layout(p, nodes[r], lay[r]);
        for (int64_t j = 0; j < P; ++j) {
            const double a = theta[r * ld + j], b = vec[r * ld + j];
            if (!std::isfinite(a) || !std::isfinite(b)) badrow[r] = 1;
            Th[(size_t)r * Qp + lay[r].cols[j]] = a;
            Vc[(size_t)r * Qp + lay[r].cols[j]] = b;
        }
    });
    for (int64_t r = 0; r < nrows; ++r)
        if (badrow[r]) return fail(GML_EINVAL, "row %lld contains a non-finite value", (long long)r);
    std::vector<uint8_t> act((size_t)nrows, 1);
    std::vector<double> fv((size_t)nrows);
    int rc = device_pass(p, rs, act, Th.data(), formulation, precision, true, fv.data(), Gi.data(), nullptr);
    if (rc) return rc;
    // 2. Hessian-vector pass: the rows of the direction through the same slots (vmap = identity)
    hipStream_t st = p->st;
    const int64_t W = p->ws_rows;
    std::memcpy(p->hTh, Vc.data(), sizeof(double) * nrows * Qp);
    HIPCHK(hipMemcpyAsync(p->dTheta, p->hTh, sizeof(double) * nrows * Qp, hipMemcpyHostToDevice, st));
    // control block of ALL rows: device_pass may have ended on a rescaled re-run of a subset (dense theta rows), which
    // leaves rowcol = -1 for the others and a shortened tile list
    int hv_npad = 0;
    {
        const int ng = (int)(Rp / 32);
        int npad = 0;
        for (int64_t r = 0; r < Rp; ++r) {
            p->hCtl[r] = (int)r; // slot = row
            p->hCtl[W + r] = r < nrows ? (int)nodes[r] : -1;
        }
        for (int g = 0; g < ng || (npad % 4); ++g, ++npad) p->hCtl[2 * W + g] = g < ng ? g : -1;
        HIPCHK(hipMemcpyAsync(p->dSrow, p->hCtl, sizeof(int) * (2 * W + npad), hipMemcpyHostToDevice, st));
        hv_npad = npad;
    }
    rc = hv_second_pass(p, Rp, (int)(Rp / 32), hv_npad, formulation, precision);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(p->hG, p->dG, sizeof(double) * nrows * Qp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    parallel_for(nrows, [&](int64_t r) {
        const double z = fv[r];
        double gv = 0.0;
        if (formulation == GML_LOGRISE) // Hess log Z = Hess Z / Z - g g^T with g = grad Z / Z (:279)
            for (int64_t j = 0; j < P; ++j) gv += Gi[(size_t)r * Qp + lay[r].cols[j]] / z * vec[r * ld + j];
        for (int64_t j = 0; j < P; ++j) {
            double v = p->hG[(size_t)r * Qp + lay[r].cols[j]];
            if (formulation == GML_LOGRISE) v = v / z - Gi[(size_t)r * Qp + lay[r].cols[j]] / z * gv;
            hv[r * ld + j] = v;
        }
    });
    return GML_OK;
}

// Test hook (not part of include/gml.h): the working-set Hessian blocks the solver builds its Newton steps from (gml_i8_hess.hip), for
// a few rows at theta -- an int8-limb objective pass leaves the curvature weights in the rows' slots, then i8_hessian runs over the
// configurations of every kstride-th block of 512 (Kh of them).  cols [nrows][m]: the working set of each row as parameter indices
// in the reference's order.  H [nrows][mp][mp] (mp = m rounded up to 32): the lower 32 x 32 tiles of sum_k h_k stat_i stat_j over
// the sub-sample, in the unit of f, not rescaled by the sub-sample's weight.
extern "C" int gml_test_hessian_blocks(gml_problem *p, int formulation, int precision, int64_t nrows, const int64_t *nodes, const double *theta,
                                       int64_t ld, const int32_t *cols, int m, int64_t Kh, int64_t kstride, double *H) {
    if (!p || !nodes || !theta || !cols || !H || nrows <= 0 || m <= 0 || m > 512) return fail(GML_EINVAL, "bad argument");
    if (!gml_is_i8(precision)) return fail(GML_EINVAL, "an int8-limb precision (the planes of V are what the kernel reads)");
    HIPCHK(hipSetDevice(p->device));
    const int64_t Qp = p->d.Qp, P = p->P, Rp = round_up(nrows, 32);
    const int mt = (m + 31) / 32, mp = mt * 32;
    RowSet rs;
    rs.R = nrows;
    rs.node.assign(nodes, nodes + nrows);
    std::vector<NodeLayout> lay((size_t)nrows);
    std::vector<double> Th((size_t)nrows * Qp, 0.0), Gi((size_t)nrows * Qp);
    std::vector<int> F((size_t)nrows * mp, 0);
    for (int64_t r = 0; r < nrows; ++r) {
        build_layout(p, nodes[r], lay[r]);
        for (int64_t j = 0; j < P; ++j) Th[(size_t)r * Qp + lay[r].cols[j]] = theta[r * ld + j];
        for (int a = 0; a < mp; ++a) {
            const int32_t j = cols[r * m + (a < m ? a : 0)]; // (the padding repeats the first entry, as the solver's lists do not care)
            if (j < 0 || j >= P) return fail(GML_EINVAL, "working-set entry out of range");
            F[(size_t)r * mp + a] = lay[r].cols[j];
        }
    }
    std::vector<uint8_t> act((size_t)nrows, 1);
    std::vector<double> fv((size_t)nrows);
    int rc = device_pass(p, rs, act, Th.data(), formulation, precision, true, fv.data(), Gi.data(), nullptr);
    if (rc) return rc;
    hipStream_t st = p->st;
    std::vector<int> ctl((size_t)(3 * nrows), 0); // rowcol | vslot | mt
    std::vector<long long> hoff((size_t)nrows + 1, 0);
    std::vector<int> hmt((size_t)nrows, mt);
    for (int64_t r = 0; r < nrows; ++r) {
        ctl[r] = (int)nodes[r];
        ctl[nrows + r] = (int)r;
        ctl[2 * nrows + r] = mt;
        hoff[r + 1] = hoff[r] + (long long)mp * mp;
    }
    int *dCtl = nullptr, *dF = nullptr;
    long long *dHoff = nullptr;
    double *dH = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        for (void *q : {(void *)dCtl, (void *)dF, (void *)dHoff, (void *)dH})
            if (q) (void)dev_free_synced(q);
    };
    const size_t htotal = (size_t)nrows * mp * mp;
    hipError_t e = dev_malloc(&dCtl, sizeof(int) * ctl.size());
    if (e == hipSuccess) e = dev_malloc(&dF, sizeof(int) * F.size());
    if (e == hipSuccess) e = dev_malloc(&dHoff, sizeof(long long) * hoff.size());
    if (e == hipSuccess) e = dev_malloc(&dH, sizeof(double) * htotal);
    if (e == hipSuccess) e = hipMemcpyAsync(dCtl, ctl.data(), sizeof(int) * ctl.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dF, F.data(), sizeof(int) * F.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dHoff, hoff.data(), sizeof(long long) * hoff.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(dH, 0, sizeof(double) * htotal, st);
    if (e != hipSuccess) {
        cleanup();
        return fail(GML_EHIP, "gml_test_hessian_blocks: %s", hipGetErrorString(e));
    }
    std::string err;
    rc = gml::i8_hessian(p->i8ws, p->d, dCtl, dCtl + nrows, dF, dCtl + 2 * nrows, hmt.data(), dHoff, (int64_t)htotal, (int)nrows, mp, formulation, Kh, kstride,
                         dH, st, &err, nullptr);
    if (rc == GML_OK) {
        e = hipMemcpyAsync(H, dH, sizeof(double) * htotal, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = GML_EHIP;
    }
    cleanup();
    (void)Rp;
    if (rc) return fail(rc, "%s", err.empty() ? "gml_test_hessian_blocks failed" : err.c_str());
    return GML_OK;
}

// Timing hook with the parameters RESIDENT in HBM: Theta is uploaded once, then `warmup + steps` passes run back
// to back on the handle's stream with no host round trip (a device-side optimiser would call the operator this
// way); f and the gradient of the last pass are downloaded once at the end.  kernel_ms[3] = device time per pass.
extern "C" int gml_bench_pass_resident(gml_problem *p, int formulation, int precision, const double *theta, int steps,
                                       int warmup, double kernel_ms[4], double *f_out, double *g_out, double *step_ms) {
    if (!p || !kernel_ms || !theta || steps < 1 || warmup < 0) return fail(GML_EINVAL, "bad argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    HIPCHK(hipSetDevice(p->device));
    const int64_t R = p->node1 - p->node0, Qp = p->d.Qp, P = p->P, Rp = round_up(R, 32);
    int rc = ensure_ws(p, R);
    if (rc) return rc;
    hipStream_t st = p->st;
    std::vector<NodeLayout> lay((size_t)R);
    std::memset(p->hTh, 0, sizeof(double) * Rp * Qp);
    parallel_for(R, [&](int64_t r) {
        build_layout(p, p->node0 + r, lay[r]);
        for (int64_t j = 0; j < P; ++j) p->hTh[(size_t)r * Qp + lay[r].cols[j]] = theta[r * P + j];
    });
    const int ngroups = (int)(Rp / 32);
    const int64_t W = p->ws_rows;
    for (int64_t r = 0; r < Rp; ++r) {
        p->hCtl[r] = (int)r; // slot = row
        p->hCtl[W + r] = r < R ? (int)(p->node0 + r) : -1;
    }
    int npad = 0;
    for (int g = 0; g < ngroups || (npad % 4); ++g, ++npad) p->hCtl[2 * W + g] = g < ngroups ? g : -1;
    HIPCHK(hipMemcpyAsync(p->dTheta, p->hTh, sizeof(double) * Rp * Qp, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->dSrow, p->hCtl, sizeof(int) * (2 * W + npad), hipMemcpyHostToDevice, st));
    {
        const int asked = precision;
        precision = gml_resolve_precision(p, asked);
        if (precision < 0) return fail(GML_EINVAL, "unknown precision %d", asked);
    }
    if (!gml_is_i8(precision)) {
        rc = gml_ensure_f64(p, p->ws_rows);
        if (rc) return rc;
    }
    std::vector<hipEvent_t> ev((size_t)3 * steps, nullptr);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int s = 0; s < warmup + steps; ++s) {
        hipEvent_t *e3 = s >= warmup ? ev.data() + (size_t)3 * (s - warmup) : nullptr;
        if (gml_is_i8(precision)) {
            std::string err;
            gml::I8Pass a{};
            a.theta = p->dTheta;
            a.srow = p->dSrow;
            a.rowcol = p->dRowcol;
            a.groups = p->dGroups;
            a.ngroups = ngroups;
            a.slot0 = 0;
            a.slot1 = (int)Rp;
            a.form = formulation;
            a.want_grad = true;
            a.F = p->dF;
            a.G = p->dG;
            a.wide = precision == GML_PREC_I8W;
            a.compact = bench_compact();
            rc = gml::i8_pass(&p->i8ws, p->d, W, a, st, e3, &err);
            if (rc) return fail(rc, "%s", err.c_str());
        } else {
            HIPCHK(hipMemsetAsync(p->dF, 0, sizeof(double) * Rp, st));
            HIPCHK(hipMemsetAsync(p->dG, 0, sizeof(double) * Rp * Qp, st));
            if (e3) HIPCHK(hipEventRecord(e3[0], st));
            launch_fwd_f64(p->d, p->dTheta, p->dRowcol, p->dGroups, npad, formulation, p->dV, p->dF, st);
            if (e3) HIPCHK(hipEventRecord(e3[1], st));
            launch_bwd_f64(p->d, p->dV, p->dGroups, ngroups, p->dG, st);
            if (e3) HIPCHK(hipEventRecord(e3[2], st));
        }
    }
    hipEvent_t e_end = nullptr;
    HIPCHK(hipEventCreate(&e_end));
    HIPCHK(hipEventRecord(e_end, st));
    HIPCHK(hipMemcpyAsync(p->hF, p->dF, sizeof(double) * Rp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(p->hG, p->dG, sizeof(double) * Rp * Qp, hipMemcpyDeviceToHost, st));
    std::vector<unsigned> mm;
    if (gml_is_i8(precision) && formulation != GML_RPLE) {
        mm.resize((size_t)Rp);
        const double *tau_ = nullptr;
        const unsigned *mm_ = nullptr;
        gml::i8_slot_results(p->i8ws, 0, &tau_, &mm_);
        HIPCHK(hipMemcpyAsync(mm.data(), mm_, sizeof(unsigned) * Rp, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    for (size_t r = 0; r < mm.size() && (int64_t)r < R; ++r)
        if (mm[r] < (1u << 23))
            return fail(GML_EUNSUPPORTED, "row %zu uses fewer than 23 bits of the fixed-point range at this theta: time it through "
                                         "gml_bench_pass (which rescales)", r);
    double sum[2] = {0, 0};
    float ms = 0;
    for (int s = 0; s < steps; ++s) {
        HIPCHK(hipEventElapsedTime(&ms, ev[(size_t)3 * s], ev[(size_t)3 * s + 1]));
        sum[0] += ms;
        HIPCHK(hipEventElapsedTime(&ms, ev[(size_t)3 * s + 1], ev[(size_t)3 * s + 2]));
        sum[1] += ms;
    }
    if (step_ms) // device time of every pass: from its forward launch to the next pass's (the last one: to the end)
        for (int s = 0; s < steps; ++s) {
            HIPCHK(hipEventElapsedTime(&ms, ev[(size_t)3 * s], s + 1 < steps ? ev[(size_t)3 * (s + 1)] : e_end));
            step_ms[s] = ms;
        }
    HIPCHK(hipEventElapsedTime(&ms, ev[0], e_end));
    kernel_ms[0] = sum[0] / steps;
    kernel_ms[1] = sum[1] / steps;
    kernel_ms[2] = kernel_ms[0] + kernel_ms[1];
    kernel_ms[3] = ms / steps; // from the first timed forward launch to the end of the last pass (quantisation of pass 1 excluded)
    for (auto &e : ev) (void)hipEventDestroy(e);
    (void)hipEventDestroy(e_end);
    if (f_out || g_out)
        parallel_for(R, [&](int64_t r) {
            const double z = p->hF[r];
            if (f_out) f_out[r] = formulation == GML_LOGRISE ? std::log(z) : z;
            if (g_out)
                for (int64_t j = 0; j < P; ++j) {
                    const double v = p->hG[(size_t)r * Qp + lay[r].cols[j]];
                    g_out[r * P + j] = formulation == GML_LOGRISE ? v / z : v;
                }
        });
    return GML_OK;
}

extern "C" int gml_bench_pass(gml_problem *p, int formulation, int precision, const double *theta, int steps,
                              int warmup, double kernel_ms[3]) {
    if (!p || !kernel_ms) return fail(GML_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    const int64_t R = p->node1 - p->node0, Qp = p->d.Qp, P = p->P;
    RowSet rs;
    rs.R = R;
    rs.node.resize((size_t)R);
    std::vector<double> Th((size_t)R * Qp, 0.0), Gi((size_t)R * Qp), fv((size_t)R);
    for (int64_t r = 0; r < R; ++r) rs.node[r] = p->node0 + r;
    if (theta)
        parallel_for(R, [&](int64_t r) {
            NodeLayout L;
            build_layout(p, rs.node[r], L);
            for (int64_t j = 0; j < P; ++j) Th[(size_t)r * Qp + L.cols[j]] = theta[r * P + j];
        });
    std::vector<uint8_t> act((size_t)R, 1);
    double sum[2] = {0, 0};
    for (int s = 0; s < warmup + steps; ++s) {
        float ms[2] = {0, 0};
        t_bench_call = true;
        int rc = device_pass(p, rs, act, Th.data(), formulation, precision, true, fv.data(), Gi.data(), nullptr, ms);
        t_bench_call = false;
        if (rc) return rc;
        if (s >= warmup) {
            sum[0] += ms[0];
            sum[1] += ms[1];
        }
    }
    kernel_ms[0] = sum[0] / steps;
    kernel_ms[1] = sum[1] / steps;
    kernel_ms[2] = kernel_ms[0] + kernel_ms[1];
    return GML_OK;
}

// Test hook (not part of include/gml.h): the block-diagonal preconditioner of the matrix-free rows on caller-given tiles -- tile t is
// a T x T row-major symmetric block of which the leading m_t x m_t part counts; z_t = (s1 H_t - s2 g_t g_t^T)^-1 r_t, by the same two
// kernels the CG uses (launch_tile_inverse, then launch_tile_apply with the tiles' entries laid out consecutively in one row).
// tests/test_gpu_newton_solve.py.
