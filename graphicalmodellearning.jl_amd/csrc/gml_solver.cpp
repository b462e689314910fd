// gml_learn: the batched l1 solver of libgml_hip, device-resident.
//
// Every local node u solves   min_x f_u(x) + lambda * sum_{j penalised} |x_j|   -- the problem the reference builds
// for Ipopt with the z >= |x| epigraph (GraphicalModelLearning.jl:166-177, one model per node in the loop :161) -- and
// all nodes advance in lock-step:
//   1. one device pass gives f and the full gradient of every active node (int8-limb or FP64 MFMA kernels);
//   2. pseudo-gradient / KKT residual and working set per node (k_select); converged nodes drop out;
//   3. Newton direction on the working set: Hessian by one device kernel over a sub-sample of the configurations +
//      batched Cholesky (working sets up to max_working entries), or matrix-free conjugate gradients with
//      Hessian-vector products from the same GEMM kernels (larger working sets: dense optima);
//   4. projected (orthant-wise) backtracking line search: the first trial is a full pass (it usually succeeds), further
//      trials are objective-only passes over the rows that need them.
// The iterates X, gradients G, trial points, directions and pseudo-gradients are [rows][Qp] arrays that stay in HBM;
// per iteration only per-row scalars (a few dozen bytes per node) and the small control blocks of the passes cross PCIe.
// A pass evaluates exactly the active rows, packed into consecutive slots of the int8-limb workspace (full MFMA tiles
// whatever subset is still active).
#include "gml_internal.h"
#include "gml_solver.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace gml;

#define RCCHK(expr)             \
    do {                        \
        const int rc__ = (expr); \
        if (rc__) return rc__;  \
    } while (0)

namespace {

// device allocations of one gml_learn call, released together
struct Arena {
    std::vector<void *> ptrs;
    ~Arena() {
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
    template <typename T> hipError_t get(T **out, size_t count) {
        *out = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(out), sizeof(T) * std::max<size_t>(count, 1));
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

// Small host <-> device transfers of the solver (control blocks up, per-row scalars down) go through one pinned arena:
// a copy from or to pageable memory is staged by the runtime and costs the host 20-30 us each, and an iteration makes
// about twenty-five of them -- at 128 rows per GPU that was a third of learn().  h2d copies the bytes into the arena and
// queues an asynchronous copy from there; d2h queues the copy into the arena and hands the bytes to the caller's buffer at
// the next sync(), which is the only place the stream is waited for.  Transfers too large for the arena go the plain way.
struct Stage {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    hipStream_t st = nullptr;
    struct Pend {
        void *host;
        const void *pin;
        size_t n;
    };
    std::vector<Pend> pend;
    void *take(size_t n) {
        const size_t a = (off + 63) & ~(size_t)63;
        if (a + n > cap) return nullptr;
        off = a + n;
        return base + a;
    }
    hipError_t h2d(void *dev, const void *host, size_t n) {
        if (n == 0) return hipSuccess;
        void *q = n <= cap / 4 ? take(n) : nullptr;
        if (!q) return hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st);
        std::memcpy(q, host, n);
        return hipMemcpyAsync(dev, q, n, hipMemcpyHostToDevice, st);
    }
    hipError_t d2h(void *host, const void *dev, size_t n) {
        if (n == 0) return hipSuccess;
        void *q = n <= cap / 4 ? take(n) : nullptr;
        if (!q) return hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st);
        pend.push_back({host, q, n});
        return hipMemcpyAsync(q, dev, n, hipMemcpyDeviceToHost, st);
    }
    hipError_t sync() {
        const hipError_t e = hipStreamSynchronize(st);
        for (const Pend &x : pend) std::memcpy(x.host, x.pin, x.n);
        pend.clear();
        off = 0;
        return e;
    }
};

} // namespace

extern "C" int gml_learn(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts_in, double *out,
                         double *kkt_out, gml_stats *stats_out) {
    if (!p || !out) return fail(GML_EINVAL, "NULL argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (formulation != GML_RISE && p->order != 2)
        return fail(GML_EUNSUPPORTED, "multi-body statistics are defined for RISE only (multiRISE, :83-152)");
    if (!(regularizer_c >= 0)) return fail(GML_EINVAL, "regularizer must be >= 0");
    gml_opts o;
    if (opts_in) o = *opts_in;
    else gml_default_opts(&o);
    if (!(o.tol > 0)) o.tol = 1e-9;
    if (o.max_iter <= 0) o.max_iter = 100;
    if (o.max_working < 32) o.max_working = 512;
    if (o.max_working > 512) o.max_working = 512;
    o.max_working = (int)gml_round_up(o.max_working, 32);
    if (o.max_add <= 0) o.max_add = 64;
    {
        // (auto: launch-bound sizes gain nothing from the int8 path and converge in fewer FP64 iterations)
        const int asked = o.precision;
        o.precision = gml_resolve_precision(p, asked);
        if (o.precision < 0) return fail(GML_EINVAL, "unknown precision %d", asked);
    }
    HIPCHK(hipSetDevice(p->device));
    const int64_t dbg_row = getenv("GML_DEBUG_ROW") ? atoll(getenv("GML_DEBUG_ROW")) : 0; // row traced at verbose >= 2
    gml_stats stl;
    std::memset(&stl, 0, sizeof stl);
    gml_stats *stats = &stl;
    const double t_start = gml_now_s();
    hipStream_t st = p->st;
    if (!p->stage) {
        p->stage_bytes = (size_t)8 << 20;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->stage), p->stage_bytes, hipHostMallocDefault));
    }
    Stage stg;
    stg.base = p->stage;
    stg.cap = p->stage_bytes;
    stg.st = st;
    const DevProblem &d = p->d;

    const int64_t R = p->node1 - p->node0, Rp = gml_round_up(R, 32), Qp = d.Qp, P = p->P;
    const int capW = o.max_working, capP = capW;
    const double lambda = gml_lambda(regularizer_c, p->n, p->M);
    stats->lambda = lambda;

    // ---- device state -------------------------------------------------------------------------------------------------
    Arena A;
    double *X, *G, *Xt, *Gt, *Xb, *D, *PG, *Gs = nullptr, *Rv = nullptr, *Pv = nullptr, *Hp = nullptr, *Zv = nullptr;
    uint8_t *kind;
    int *dNode, *dRows, *dRows2, *dRowsP, *dCtl, *dFidx, *dMt, *dMsCg, *dVslot, *dHv = nullptr;
    CgState *dCg;
    long long *dHoff;
    double *dBest, *dAlpha, *dScale, *dS1, *dS1cg, *dDinv, *dFs, *dOvr, *dgF, *dpgF, *dsol, *dSdiag, *dH = nullptr;
    SelectOut *dSel;
    TrialOut *dTrial;
    int64_t dH_elems = 0;
    const size_t nd = (size_t)Rp * Qp;
    // slots of the int8-limb workspace: every active row of a pass in its own slot, the passes of one iteration in
    // disjoint ranges (their V planes feed the next Hessians)
    // Objective-only passes (line-search trials whose V planes nobody reads) run in a scratch range above the main one.
    const int64_t Smain = o.precision == GML_PREC_I8X ? Rp + gml_round_up(std::max<int64_t>(R / 2, 96), 32) + 64 : Rp + 64;
    const int64_t Scap = Smain + Rp;
    {
        size_t freeb = 0, totalb = 0;
        HIPCHK(hipMemGetInfo(&freeb, &totalb));
        const double need = 7.0 * 8.0 * (double)nd + (double)nd;
        if (need > 0.9 * (double)freeb)
            return fail(GML_ENOMEM, "solver state of %.1f GB for %lld rows does not fit in %.1f GB free HBM", need / 1e9, (long long)R,
                        freeb / 1e9);
    }
    HIPCHK(A.get(&X, nd));
    HIPCHK(A.get(&G, nd));
    HIPCHK(A.get(&Xt, nd));
    HIPCHK(A.get(&Gt, nd));
    HIPCHK(A.get(&Xb, nd));
    HIPCHK(A.get(&D, nd));
    HIPCHK(A.get(&PG, nd));
    HIPCHK(A.get(&kind, nd));
    HIPCHK(A.get(&dNode, (size_t)Rp));
    HIPCHK(A.get(&dRows, (size_t)Rp));
    HIPCHK(A.get(&dRows2, (size_t)Rp));
    HIPCHK(A.get(&dRowsP, (size_t)Rp));
    HIPCHK(A.get(&dCtl, (size_t)(2 * Scap + Scap / 32 + 8)));
    HIPCHK(A.get(&dFidx, (size_t)Rp * capP));
    HIPCHK(A.get(&dMt, (size_t)3 * Rp));
    HIPCHK(A.get(&dMsCg, (size_t)Rp));
    HIPCHK(A.get(&dDinv, (size_t)Rp));
    HIPCHK(A.get(&dS1cg, (size_t)Rp));
    HIPCHK(A.get(&dCg, (size_t)Rp));
    HIPCHK(A.get(&dVslot, (size_t)Rp));
    HIPCHK(A.get(&dHoff, (size_t)Rp + 1));
    HIPCHK(A.get(&dBest, (size_t)Rp));
    HIPCHK(A.get(&dAlpha, (size_t)Rp));
    HIPCHK(A.get(&dScale, (size_t)Rp));
    HIPCHK(A.get(&dS1, (size_t)Rp));
    HIPCHK(A.get(&dFs, (size_t)Scap));
    HIPCHK(A.get(&dOvr, (size_t)Scap));
    HIPCHK(A.get(&dgF, (size_t)Rp * capP));
    HIPCHK(A.get(&dpgF, (size_t)Rp * capP));
    HIPCHK(A.get(&dsol, (size_t)Rp * capP));
    HIPCHK(A.get(&dSdiag, (size_t)Rp));
    HIPCHK(A.get(&dSel, (size_t)Rp));
    HIPCHK(A.get(&dTrial, (size_t)Rp));
    HIPCHK(hipMemsetAsync(X, 0, sizeof(double) * nd, st));
    HIPCHK(hipMemsetAsync(Xb, 0, sizeof(double) * nd, st));
    HIPCHK(hipMemsetAsync(G, 0, sizeof(double) * nd, st));
    HIPCHK(hipMemsetAsync(D, 0, sizeof(double) * nd, st));
    {
        std::vector<int> node((size_t)Rp, -1);
        for (int64_t r = 0; r < R; ++r) node[r] = (int)(p->node0 + r);
        std::vector<double> inf((size_t)Rp, INFINITY);
        HIPCHK(stg.h2d(dNode, node.data(), sizeof(int) * Rp));
        HIPCHK(stg.h2d(dBest, inf.data(), sizeof(double) * Rp));
        HIPCHK(stg.sync());
    }
    launch_kind(d, p->order, dNode, (int)Rp, kind, st);

    // ---- host state (scalars per row) -----------------------------------------------------------------------------------
    std::vector<double> f((size_t)R, 0.0), ft((size_t)R, 0.0), Fobj((size_t)R, 0.0), kkt((size_t)R, INFINITY), best((size_t)R, INFINITY),
        Z((size_t)R, 1.0), Zt((size_t)R, 1.0), alpha((size_t)R, 1.0), dd((size_t)R, 0.0), fn((size_t)R, 0.0), fnt((size_t)R, 0.0),
        l1t((size_t)R, 0.0), backv((size_t)R, 0.0);
    std::vector<uint8_t> done((size_t)R, 0), vstale((size_t)R, 0), atfloor((size_t)R, 0), nreg((size_t)R, 0), accepted_fwd((size_t)R, 0),
        need((size_t)R, 0), iscg((size_t)R, 0);
    std::vector<int> stall((size_t)R, 0), msz((size_t)R, 0), vslot((size_t)R, -1), vprev((size_t)R, -1), owner((size_t)Scap, -1);
    std::vector<double> Fbest((size_t)R, INFINITY);
    std::vector<SelectOut> sel((size_t)Rp);
    std::vector<TrialOut> trial((size_t)Rp);
    int64_t slot_next = 0;
    // Scale of the fixed-point V (int8 path): instead of the worst-case bound w_max exp(sum|theta|) every pass after a row's
    // first uses vref = max_k |V_rk| measured by its previous pass, times exp(||theta - theta_ref||_1), which bounds the new
    // weights rigorously (|E_k' - E_k| <= ||theta' - theta||_1).  Near the optimum the steps are tiny, so V keeps all 31 bits
    // relative to its actual maximum and the noise floor of f and grad drops by the bits the bound would have wasted.
    std::vector<double> vref((size_t)R, 0.0), dref((size_t)R, 0.0), stepn((size_t)R, 0.0);

    // Matrix-free rows admit, per iteration, only the violators within this fraction of the largest violation (measured on
    // the order-3 config at the reference's default regulariser: with every violator at once -- 0 -- or a quarter of the
    // largest -- 0.25 -- the projected Newton steps are damped to nothing by the line search; 0.5 takes full steps throughout;
    // 0.7 converges too, in 1.6x the iterations)
    const double viol_frac = getenv("GML_CG_VIOL_FRAC") ? atof(getenv("GML_CG_VIOL_FRAC")) : 0.5;
    const double cg_eta = getenv("GML_CG_ETA") ? atof(getenv("GML_CG_ETA")) : 0.05;
    int prec = o.precision; // switches to FP64 for the rows the int8-limb path cannot bring below tol ("polish")
    bool can_polish = false;
    if (o.precision == GML_PREC_I8X && o.polish >= 0) {
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
            const double need_b = (d.Xs ? 0.0 : 2.0 * (double)d.Kp * (double)Qp) + (p->dV && p->dVrows >= Rp ? 0.0 : 8.0 * (double)Rp * (double)d.Kp) +
                                  8.0 * (double)nd + 4.0 * (double)Scap * d.Kp /* the i8 workspace still to come */;
            can_polish = need_b < 0.8 * (double)freeb;
        }
    }
    int stall_cap = can_polish ? 4 : 10;

    auto stage = [&](const char *name) {
        if (o.verbose >= 3) {
            (void)stg.sync();
            fprintf(stderr, "[gml]     stage %s (last error: %s)\n", name, hipGetErrorString(hipGetLastError()));
            fflush(stderr);
        }
    };
    auto upload_rows = [&](const std::vector<int> &rows, int *dst) -> int {
        if (!rows.empty()) HIPCHK(stg.h2d(dst, rows.data(), sizeof(int) * rows.size()));
        return GML_OK;
    };

    // ---- one objective(/gradient) pass over the listed rows ---------------------------------------------------------------
    //   src: X or Xt; dst: G or Gt (want_grad); results per row: fo (f, or log Z), zo (Z, logRISE), no (noise of f)
    //   pp: the arithmetic of this pass (the solver's current precision; the FP64 phase borrows int8 passes for the V
    //   planes its matrix-free rows need)
    std::vector<int> pslot((size_t)R, -1); // slot each row occupied in the pass being re-run (scale re-runs happen in place)
    std::function<int(const std::vector<int> &, const double *, double *, bool, bool, std::vector<double> &, std::vector<double> &,
                      std::vector<double> &, const std::vector<double> *, int, int)>
        run_pass = [&](const std::vector<int> &rows, const double *src, double *dst, bool want_grad, bool at_trial,
                       std::vector<double> &fo, std::vector<double> &zo, std::vector<double> &no, const std::vector<double> *ovr_in,
                       int depth, int pp) -> int {
        const int64_t n = (int64_t)rows.size();
        if (n == 0) return GML_OK;
        const double t0 = gml_now_s();
        stage(want_grad ? "pass" : "fwd pass");
        const int64_t np = gml_round_up(n, 32);
        std::vector<double> fh, tauh;
        std::vector<unsigned> mmh;
        int64_t base = 0;
        const bool track = pp == GML_PREC_I8X && formulation != GML_RPLE;
        if (pp == GML_PREC_I8X) {
            // slots of this pass: a fresh consecutive range, or (re-run of some rows of a pass with a tighter scale: ovr_in)
            // the slots those rows already hold -- a re-run must not claim new slots, it could wrap around and overwrite
            // planes of its own pass
            std::vector<int64_t> slot((size_t)n);
            int64_t lo, hi;
            if (ovr_in) {
                lo = Scap;
                hi = 0;
                for (int64_t a = 0; a < n; ++a) {
                    slot[a] = pslot[rows[a]];
                    lo = std::min(lo, slot[a] / 32 * 32);
                    hi = std::max(hi, slot[a] / 32 * 32 + 32);
                }
            } else if (!want_grad && at_trial) {
                // objective-only trial: scratch slots, the rows keep the V planes of their iterates
                lo = Smain;
                hi = Smain + np;
                for (int64_t a = 0; a < n; ++a) slot[a] = Smain + a;
            } else {
                base = gml_round_up(slot_next, 32);
                if (base + np > Smain) base = 0; // wrap: the rows whose V planes are overwritten become stale below
                slot_next = base + np;
                lo = base;
                hi = base + np;
                for (int64_t a = 0; a < np; ++a) {
                    const int64_t s = base + a;
                    const int prev = owner[s];
                    if (prev >= 0 && vslot[prev] == s) {
                        vslot[prev] = -1;
                        vstale[prev] = 1;
                    }
                    owner[s] = a < n ? rows[a] : -1;
                }
                for (int64_t a = 0; a < n; ++a) {
                    const int r = rows[a];
                    slot[a] = base + a;
                    vprev[r] = at_trial ? vslot[r] : -1; // a rejected trial goes back to the planes of the iterate, if they survive
                    vslot[r] = (int)(base + a);
                    vstale[r] = 0;
                }
            }
            for (int64_t a = 0; a < n; ++a) pslot[rows[a]] = (int)slot[a];
            const int64_t ns = hi - lo;
            std::vector<int> srow((size_t)ns, 0), rowcol((size_t)ns, -1), groups;
            std::vector<double> ovr((size_t)ns, 0.0);
            std::vector<uint8_t> tile((size_t)(ns / 32), 0);
            for (int64_t a = 0; a < n; ++a) {
                const int64_t q = slot[a] - lo;
                const int r = rows[a];
                srow[q] = r;
                rowcol[q] = (int)(p->node0 + r);
                tile[q / 32] = 1;
                if (ovr_in) ovr[q] = (*ovr_in)[r];
                else if (track && vref[r] > 0.0)
                    ovr[q] = vref[r] * std::exp(dref[r] + (at_trial ? stepn[r] : 0.0)) * (1.0 + 1e-6) / 2130000000.0;
            }
            for (int64_t g = 0; g < ns / 32; ++g)
                if (tile[g]) groups.push_back((int)(lo / 32 + g));
            const int ng = (int)groups.size();
            while (groups.size() % 4) groups.push_back(-1);
            // device control block: srow | rowcol are indexed by slot, so they are placed at the slot range
            HIPCHK(stg.h2d(dCtl + lo, srow.data(), sizeof(int) * ns));
            HIPCHK(stg.h2d(dCtl + Scap + lo, rowcol.data(), sizeof(int) * ns));
            HIPCHK(stg.h2d(dCtl + 2 * Scap, groups.data(), sizeof(int) * groups.size()));
            HIPCHK(stg.h2d(dOvr + lo, ovr.data(), sizeof(double) * ns));
            I8Pass a{};
            a.theta = src;
            a.srow = dCtl;
            a.rowcol = dCtl + Scap;
            a.groups = dCtl + 2 * Scap;
            a.ngroups = ng;
            a.slot0 = (int)lo;
            a.slot1 = (int)hi;
            a.form = formulation;
            a.want_grad = want_grad;
            a.F = dFs;
            a.G = dst;
            a.tauovr = dOvr;
            std::string err;
            int rc = i8_pass(&p->i8ws, d, Scap, a, st, nullptr, &err);
            if (rc) return fail(rc, "%s", err.c_str());
            std::vector<double> fr((size_t)ns), taur((size_t)ns);
            std::vector<unsigned> mmr((size_t)ns);
            const double *dtau = nullptr;
            const unsigned *dmm = nullptr;
            i8_slot_results(p->i8ws, 0, &dtau, &dmm);
            HIPCHK(stg.d2h(fr.data(), dFs + lo, sizeof(double) * ns));
            HIPCHK(stg.d2h(taur.data(), dtau + lo, sizeof(double) * ns));
            HIPCHK(stg.d2h(mmr.data(), dmm + lo, sizeof(unsigned) * ns));
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync());
            fh.resize((size_t)n);
            tauh.resize((size_t)n);
            mmh.resize((size_t)n);
            for (int64_t a2 = 0; a2 < n; ++a2) {
                fh[a2] = fr[slot[a2] - lo];
                tauh[a2] = taur[slot[a2] - lo];
                mmh[a2] = mmr[slot[a2] - lo];
            }
        } else {
            // FP64 path: slot = row; a tile's backward GEMM writes every row of the tile, so the gradient goes to a
            // scratch array first and only the listed rows are copied out
            int rc = gml_ensure_f64(p, Rp);
            if (rc) return rc;
            if (!Gs) HIPCHK(A.get(&Gs, nd));
            std::vector<int> ctl((size_t)(Rp + Rp / 32 + 8), -1);
            std::vector<uint8_t> tile((size_t)(Rp / 32), 0);
            for (int64_t a = 0; a < n; ++a) {
                ctl[rows[a]] = (int)(p->node0 + rows[a]);
                tile[rows[a] >> 5] = 1;
            }
            int ng = 0;
            for (int64_t g = 0; g < Rp / 32; ++g)
                if (tile[g]) ctl[Rp + ng++] = (int)g;
            const int ng4 = (int)gml_round_up(ng, 4);
            HIPCHK(stg.h2d(dCtl, ctl.data(), sizeof(int) * (Rp + ng4)));
            HIPCHK(hipMemsetAsync(dFs, 0, sizeof(double) * Rp, st));
            launch_fwd_f64(d, src, dCtl, dCtl + Rp, ng4, formulation, p->dV, dFs, st);
            if (want_grad) {
                HIPCHK(hipMemsetAsync(Gs, 0, sizeof(double) * nd, st));
                launch_bwd_f64(d, p->dV, dCtl + Rp, ng, Gs, st);
                RCCHK(upload_rows(rows, dRowsP));
                launch_copy_rows(dRowsP, (int)n, Qp, Gs, dst, nullptr, nullptr, st);
            }
            fh.resize((size_t)Rp);
            HIPCHK(stg.d2h(fh.data(), dFs, sizeof(double) * Rp));
            for (int64_t a = 0; a < n; ++a) { // (V [row][Kp] of the FP64 path is indexed by row)
                vstale[rows[a]] = 0;
                if (vslot[rows[a]] < 0) vslot[rows[a]] = 0;
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(stg.sync());
        std::vector<int> again;
        std::vector<double> ovr2;
        std::vector<double> scale;
        for (int64_t a = 0; a < n; ++a) {
            const int r = rows[a];
            const double fv = pp == GML_PREC_I8X ? fh[a] : fh[r];
            // f64: summation rounding.  int8 limbs: every V_rk is rounded to a multiple of tau_r with a dither that is
            // equidistributed over the samples, so the errors (each within one unit, standard deviation 0.41 tau) add like a
            // random walk: 8 sigma of sqrt(K) terms (the worst case K * tau is never approached).
            double noise = 1e-13 * std::max(1.0, std::fabs(fv));
            if (track) {
                noise += 3.3 * std::sqrt((double)p->K) * tauh[a];
                const double vmax = ((double)mmh[a] + 1.0) * tauh[a]; // rigorous bound on max_k |V_rk|
                vref[r] = vmax;
                dref[r] = at_trial ? stepn[r] : 0.0; // distance from the current iterate to the point just evaluated
                // Dynamic range: tau_r was derived from a bound; when the largest |V_rk| actually seen is more than 8 bits
                // below it (dense theta), the row is re-run with tau_r taken from that maximum
                if (mmh[a] < (1u << 23)) {
                    if (ovr2.empty()) ovr2.assign((size_t)R, 0.0);
                    again.push_back(r);
                    ovr2[r] = vmax * (1.0 + 1e-12) / 2130000000.0;
                }
            }
            if (formulation == GML_LOGRISE) { // f = log Z, g = grad Z / Z   (:279)
                zo[r] = fv;
                fo[r] = std::log(fv);
                no[r] = noise / fv;
            } else {
                fo[r] = fv;
                no[r] = noise;
            }
        }
        stats->node_evals += n;
        if (want_grad) ++stats->passes;
        else ++stats->forward_passes;
        if (!again.empty()) {
            stats->t_pass += gml_now_s() - t0;
            if (depth >= 6) {
                // the weights exp(-E) of these rows underflow the fixed-point range even after six rescalings (|theta|_1 in the
                // hundreds): a trial point that far out is simply rejected; at the iterate itself it is an error
                if (!at_trial) return fail(GML_EUNSUPPORTED, "precision i8x: the weights exp(-E) of a row underflow its fixed-point range; use precision f64");
                for (int r : again) {
                    fo[r] = INFINITY;
                    no[r] = 0.0;
                }
                return GML_OK;
            }
            int rc = run_pass(again, src, dst, want_grad, at_trial, fo, zo, no, &ovr2, depth + 1, pp);
            if (rc) return rc;
            // rows of this call that were not re-run still need their logRISE scaling: fall through with them only
        }
        if (formulation == GML_LOGRISE && want_grad) {
            std::vector<int> keep;
            std::vector<double> sc((size_t)Rp, 1.0);
            for (int64_t a = 0; a < n; ++a) {
                const int r = rows[a];
                if (std::find(again.begin(), again.end(), r) != again.end()) continue; // scaled by the re-run
                keep.push_back(r);
                sc[r] = 1.0 / zo[r];
            }
            if (!keep.empty()) {
                HIPCHK(stg.h2d(dScale, sc.data(), sizeof(double) * Rp));
                RCCHK(upload_rows(keep, dRowsP));
                launch_scale_rows(dRowsP, (int)keep.size(), dScale, Qp, dst, st);
                HIPCHK(stg.sync()); // sc, keep are locals
            }
        }
        if (again.empty()) stats->t_pass += gml_now_s() - t0;
        return GML_OK;
    };

    // sub-sampled Newton: Hessians over Kh configurations -- every kstride-th block of 512, so that a sorted histogram is
    // sampled evenly -- rescaled by the weight of the sub-sample.  The budget (rows x configurations) is kept roughly
    // constant: as nodes converge, the remaining ones get more configurations, up to all of them -- an inexact Hessian only
    // costs iterations, and it costs the most on the few ill-conditioned nodes that are still active at the end.
    const int64_t Kh_base = o.hess_samples == 0 ? 32768 : (o.hess_samples < 0 ? d.Kp : (int64_t)o.hess_samples);
    const int64_t nblk512 = d.Kp / 512;
    int64_t Kh = d.Kp, kstride = 1;
    double hscale = 1.0;
    auto set_kh = [&](int64_t nactive) {
        int64_t want = Kh_base;
        if (o.hess_samples == 0 && nactive > 0) want = Kh_base * std::max<int64_t>(1, R / nactive);
        int64_t nb = std::min(nblk512, std::max<int64_t>(2, (want + 511) / 512));
        if (nb * 512 >= p->K) nb = nblk512; // (nearly) everything: take it all
        kstride = nblk512 / nb;
        Kh = nb * 512;
        double wsum = 0;
        for (int64_t cb = 0; cb < nb; ++cb) wsum += p->wblk[(size_t)(cb * kstride)];
        if (!(wsum > 0)) { // a sub-sample without weight (degenerate histogram): use every configuration
            nb = nblk512;
            kstride = 1;
            Kh = d.Kp;
            wsum = 1.0;
        }
        hscale = nb == nblk512 ? 1.0 : 1.0 / wsum;
    };
    set_kh(R);

    // ---- first pass at X = 0 ------------------------------------------------------------------------------------------------
    std::vector<int> rows_all((size_t)R);
    for (int64_t r = 0; r < R; ++r) rows_all[r] = (int)r;
    int rc = run_pass(rows_all, X, G, true, false, f, Z, fn, nullptr, 0, prec);
    if (rc) return rc;

    int it = 0;
    for (it = 0; it < o.max_iter; ++it) {
        const double th0 = gml_now_s();
        // ---- KKT residuals, working sets (device) -----------------------------------------------------------------------------
        std::vector<int> act;
        for (int64_t r = 0; r < R; ++r)
            if (!done[r]) act.push_back((int)r);
        RCCHK(upload_rows(act, dRows));
        stage("select");
        launch_select(dRows, (int)act.size(), X, G, kind, Qp, lambda, o.max_add, capW, capP, viol_frac, PG, dFidx, dgF, dpgF, dSel, dBest, Xb, st);
        HIPCHK(stg.d2h(sel.data(), dSel, sizeof(SelectOut) * Rp));
        HIPCHK(hipGetLastError());
        HIPCHK(stg.sync());
        int64_t nactive = 0, ncg = 0;
        double worst_all = 0;
        int maxm = 0;
        for (int r : act) {
            const SelectOut &s = sel[r];
            Fobj[r] = f[r] + s.l1;
            kkt[r] = std::isfinite(s.worst) && std::isfinite(f[r]) ? s.worst : INFINITY;
            // progress = a smaller KKT residual (the device made the same comparison and saved the iterate) or a smaller
            // objective beyond its noise: with thousands of coordinates entering at once (dense optima) the residual is
            // not monotone along a converging sequence, the objective is
            const bool fdown = Fobj[r] < Fbest[r] - std::max(10.0 * fn[r], 1e-13 * std::fabs(Fobj[r]));
            if (Fobj[r] < Fbest[r]) Fbest[r] = Fobj[r];
            if (kkt[r] < best[r]) {
                best[r] = kkt[r];
                stall[r] = 0;
            } else if (fdown) {
                stall[r] = 0;
            } else {
                ++stall[r];
            }
            if (kkt[r] <= o.tol) {
                done[r] = 1;
                continue;
            }
            if (stall[r] >= stall_cap) { // no progress: at the noise floor of the pass arithmetic (or a failed line search)
                done[r] = 1;
                atfloor[r] = 1;
                continue;
            }
            ++nactive;
            iscg[r] = s.m < 0;
            msz[r] = s.m < 0 ? s.pad : s.m; // working set of the Cholesky step, or the preconditioner block of a matrix-free row
            if (iscg[r]) ++ncg;
            maxm = std::max(maxm, msz[r]);
        }
        for (int64_t r = 0; r < R; ++r) {
            if (done[r]) msz[r] = 0;
            worst_all = std::max(worst_all, std::min(kkt[r], best[r]));
        }
        if (o.verbose)
            fprintf(stderr, "[gml] it %3d active %6lld (cg %lld)  max-kkt %.3e  max|W| %d  passes %d fwd %d%s\n", it, (long long)nactive,
                    (long long)ncg, worst_all, maxm, stats->passes, stats->forward_passes, prec == GML_PREC_F64 && o.precision != prec ? "  [fp64 polish]" : "");
        if (o.verbose >= 2 && dbg_row < R)
            fprintf(stderr, "[gml]   row %lld: kkt %.3e best %.3e F %.15e m %d nsupp %d nviol %d stall %d\n", (long long)dbg_row, kkt[dbg_row],
                    best[dbg_row], Fobj[dbg_row], sel[dbg_row].m, sel[dbg_row].nsupp, sel[dbg_row].nviol, stall[dbg_row]);
        stats->t_host += gml_now_s() - th0;
        if (nactive == 0) {
            // Polish: rows that the int8-limb arithmetic could not bring below tol (its gradient carries ~sqrt(K) 2^-31 of noise
            // relative to the largest weight, which an ill-conditioned, weakly regularised problem amplifies) continue on the
            // FP64 path from their best iterate, when that path fits in memory.
            std::vector<int> fl;
            for (int64_t r = 0; r < R; ++r)
                if (atfloor[r] && !(std::min(best[r], kkt[r]) <= o.tol)) fl.push_back((int)r);
            if (!(prec == GML_PREC_I8X && can_polish && !fl.empty())) break;
            if (gml_ensure_f64(p, Rp) != GML_OK) break; // does not fit after all: the rows stay as they are (reported not converged)
            prec = GML_PREC_F64;
            stall_cap = 10;
            RCCHK(upload_rows(fl, dRows));
            launch_copy_rows(dRows, (int)fl.size(), Qp, Xb, X, nullptr, nullptr, st); // back to the best iterate
            std::vector<double> inf((size_t)Rp, INFINITY);
            HIPCHK(stg.h2d(dBest, inf.data(), sizeof(double) * Rp));
            HIPCHK(stg.sync());
            for (int r : fl) {
                done[r] = 0;
                atfloor[r] = 0;
                stall[r] = 0;
                best[r] = INFINITY;
                Fbest[r] = INFINITY;
            }
            if (o.verbose) fprintf(stderr, "[gml] polish: %zu rows continue on the FP64 path\n", fl.size());
            rc = run_pass(fl, X, G, true, false, f, Z, fn, nullptr, 0, prec);
            if (rc) return rc;
            set_kh((int64_t)fl.size());
            stats->polished = 1;
            continue;
        }
        set_kh(nactive);

        // rows whose V planes were overwritten (a wrapped slot range) need a fresh pass before the curvature; such a pass can
        // itself overwrite planes that are still needed, hence the loop: its last round re-evaluates every active row from
        // slot 0 (they always fit)
        for (int round = 0; round < 3; ++round) {
            std::vector<int> stale;
            for (int64_t r = 0; r < R; ++r)
                if (!done[r] && (vstale[r] || vslot[r] < 0)) stale.push_back((int)r);
            if (stale.empty()) break;
            if (round == 2) {
                stale.clear();
                for (int64_t r = 0; r < R; ++r)
                    if (!done[r]) stale.push_back((int)r);
                slot_next = 0;
            }
            rc = run_pass(stale, X, G, true, false, f, Z, fn, nullptr, 0, prec);
            if (rc) return rc;
        }

        // ---- Newton directions -----------------------------------------------------------------------------------------------------
        const double th1 = gml_now_s();
        std::vector<int> chol_rows, cg_rows;
        for (int64_t r = 0; r < R; ++r)
            if (!done[r]) (iscg[r] ? cg_rows : chol_rows).push_back((int)r);
        if (!cg_rows.empty() && prec != GML_PREC_I8X) {
            // FP64 phase: the curvature weights of the matrix-free rows' Hessian-vector products come from an int8-limb objective
            // pass at the same iterate (those products run on the int8 cores either way; only the curvature is approximate)
            std::vector<double> tf((size_t)R), tz((size_t)R, 1.0), tn((size_t)R);
            rc = run_pass(cg_rows, X, nullptr, false, false, tf, tz, tn, nullptr, 0, GML_PREC_I8X);
            if (rc) return rc;
        }
        {
            // Hessian blocks (int8 kernel over the limb planes of the rows' last passes, or the FP64 MFMA kernel over V): the
            // working set of a Cholesky row, the preconditioner block of a matrix-free row
            std::vector<int> mt2((size_t)3 * R), mscg((size_t)R, 0);
            std::vector<long long> hoff((size_t)R + 1, 0);
            std::vector<double> s1((size_t)Rp, 1.0), s1cg((size_t)Rp, 1.0), dinv((size_t)Rp, 1.0);
            for (int64_t r = 0; r < R; ++r) {
                const int m = done[r] ? 0 : msz[r];
                mt2[r] = (m + 31) / 32;
                mt2[R + r] = (int)(p->node0 + r);
                mt2[2 * R + r] = iscg[r] ? 0 : m; // the Cholesky step solves these
                mscg[r] = iscg[r] ? m : 0;        // the preconditioner solves of the matrix-free rows
                hoff[r + 1] = hoff[r] + (long long)mt2[r] * 32 * mt2[r] * 32;
                const double zi = formulation == GML_LOGRISE ? 1.0 / Z[r] : 1.0; // Hess log Z = Hess Z / Z - g g^T
                s1[r] = hscale * zi; // sub-sampled blocks
                s1cg[r] = zi;        // the Hessian-vector products use every configuration
                // the common diagonal of the operator: sum_k h_k (RISE: f; logRISE: Z / Z = 1; RPLE: at most 1)
                dinv[r] = formulation == GML_RISE ? 1.0 / std::max(f[r], 1e-300) : 1.0;
            }
            const int64_t htotal = std::max<long long>(hoff[R], 1);
            if (htotal > dH_elems) {
                dH_elems = htotal + htotal / 4;
                HIPCHK(A.get(&dH, (size_t)dH_elems));
            }
            HIPCHK(stg.h2d(dMt, mt2.data(), sizeof(int) * 3 * R));
            HIPCHK(stg.h2d(dMsCg, mscg.data(), sizeof(int) * R));
            HIPCHK(stg.h2d(dHoff, hoff.data(), sizeof(long long) * (R + 1)));
            HIPCHK(stg.h2d(dS1, s1.data(), sizeof(double) * Rp));
            HIPCHK(stg.h2d(dDinv, dinv.data(), sizeof(double) * Rp));
            HIPCHK(stg.h2d(dS1cg, s1cg.data(), sizeof(double) * Rp));
            HIPCHK(stg.h2d(dVslot, vslot.data(), sizeof(int) * R));
            HIPCHK(hipMemsetAsync(dH, 0, sizeof(double) * htotal, st));
            stage("hessian");
            if (prec == GML_PREC_I8X)
                for (int64_t r = 0; r < R; ++r)
                    if (mt2[r] > 0 && (vslot[r] < 0 || vslot[r] >= Scap || owner[vslot[r]] != r || vstale[r]))
                        return fail(GML_EHIP, "internal: row %lld enters the Hessian without valid V planes (slot %d, owner %d, stale %d)",
                                    (long long)r, vslot[r], vslot[r] >= 0 && vslot[r] < Scap ? owner[vslot[r]] : -2, (int)vstale[r]);
            if (prec == GML_PREC_I8X) {
                std::string err;
                int hrc = i8_hessian(p->i8ws, d, dMt + R, dVslot, dFidx, dMt, mt2.data(), dHoff, htotal, (int)R, capP, formulation, Kh, kstride,
                                     dH, st, &err);
                if (hrc) return fail(hrc, "%s", err.empty() ? "int8 Hessian: working set above 512 entries" : err.c_str());
            } else {
                launch_hess_f64(d, p->dV, dMt + R, dFidx, dMt, dHoff, (int)R, capP, formulation, Kh, kstride, dH, st);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync()); // the vectors above are locals
            ++stats->hessian_passes;
        }
        const double s2 = formulation == GML_LOGRISE ? 1.0 : 0.0;
        if (!chol_rows.empty()) {
            // batched Cholesky on the device; the directions are scattered into D
            stage("cholesky");
            launch_newton_solve(dH, dHoff, dMt, dMt + 2 * R, dS1, s2, dgF, dpgF, (int)R, capP, dsol, dSdiag, st);
            RCCHK(upload_rows(chol_rows, dRows));
            launch_scatter_dir(dRows, (int)chol_rows.size(), dFidx, dsol, dMt + 2 * R, capP, Qp, D, st);
            HIPCHK(hipGetLastError());
        }
        if (!cg_rows.empty()) {
            // Matrix-free Newton-CG: H_WW d = -pg_W by preconditioned conjugate gradients, Hessian-vector products from the device
            // operator (forward GEMM of the direction, weights of the rows' last objective pass, backward GEMM), preconditioner =
            // the Cholesky-factored Hessian block of the row's strongest entries + the common diagonal elsewhere.  Inexact Newton:
            // the residual is reduced by eta = min(0.05, sqrt(kkt)), in at most max_cg steps.
            stage("pcg");
            if (!Rv) {
                HIPCHK(A.get(&Rv, nd));
                HIPCHK(A.get(&Pv, nd));
                HIPCHK(A.get(&Hp, nd));
                HIPCHK(A.get(&Zv, nd));
                HIPCHK(A.get(&dHv, (size_t)(3 * Rp + Rp / 32 + 8)));
            }
            RCCHK(upload_rows(cg_rows, dRows));
            launch_pcg_init(dRows, (int)cg_rows.size(), X, PG, kind, Qp, dFidx, dMsCg, capP, D, Rv, dpgF, dCg, st);
            launch_newton_solve(dH, dHoff, dMt, dMsCg, dS1, s2, dgF, dpgF, (int)R, capP, dsol, dSdiag, st);
            launch_pcg_dir(dRows, (int)cg_rows.size(), Qp, dFidx, dMsCg, capP, dsol, dDinv, Rv, Zv, Pv, 1, dCg, st);
            std::vector<CgState> cgs((size_t)Rp);
            std::vector<int> live = cg_rows;
            // at most 16 CG steps per Newton step: the number of Newton iterations is set by the admission of the violators, not by
            // the accuracy of the directions (order-3 probe: 32 iterations with a cap of 40, 16 or 15 -- 545 / 349 / 333
            // Hessian-vector passes; 40 iterations, 279 passes with a cap of 8)
            const int maxcg = o.max_cg > 0 ? o.max_cg : (getenv("GML_CG_MAX") ? atoi(getenv("GML_CG_MAX")) : 16);
            // limbs of the CG direction p in the Hessian-vector passes: 2 (14 bits of max|p|) are enough for an inexact Newton
            // step that stops at a residual of 5 % -- the iteration counts of the 64-node probe of config 5 are 59 / 59 / 56
            // with 4 / 3 / 2 limbs, and the forward GEMM of an H.v pass costs in proportion
            int hv_lf = getenv("GML_HV_LF") ? atoi(getenv("GML_HV_LF")) : 2;
            hv_lf = hv_lf < 2 ? 2 : (hv_lf > 5 ? 5 : hv_lf);
            // ... and the products u_k = h_k (x_k . p) go to the backward GEMM in 2 limbs (15 bits of the largest) instead of 4:
            // half the MFMAs and half the V reads of that GEMM; config 5 at the default regulariser 116 -> 93 s with
            // 10 % more iterations (GML_HV_LB=4 restores the 31 bits)
            const int hv_lb = getenv("GML_HV_LB") ? atoi(getenv("GML_HV_LB")) : 2;
            for (int ci = 0; ci < maxcg && !live.empty(); ++ci) {
                // Hp = H p for the live rows: an hv pass over slots [0, n) of the u-plane workspace
                const int64_t n = (int64_t)live.size(), np = gml_round_up(n, 32);
                std::vector<int> ctl((size_t)(3 * np + np / 32 + 4), -1);
                for (int64_t a = 0; a < np; ++a) {
                    ctl[a] = a < n ? live[a] : 0;
                    ctl[np + a] = a < n ? (int)(p->node0 + live[a]) : -1;
                    ctl[2 * np + a] = a < n ? vslot[live[a]] : 0;
                }
                for (int64_t g = 0; g < np / 32; ++g) ctl[3 * np + g] = (int)g;
                HIPCHK(stg.h2d(dHv, ctl.data(), sizeof(int) * ctl.size()));
                I8Pass a{};
                a.theta = Pv;
                a.srow = dHv;
                a.rowcol = dHv + np;
                a.vmap = dHv + 2 * np;
                a.groups = dHv + 3 * np;
                a.ngroups = (int)(np / 32);
                a.slot0 = 0;
                a.slot1 = (int)np;
                a.form = formulation;
                a.want_grad = true;
                a.F = nullptr;
                a.G = Hp;
                a.hv = hv_lb == 2 ? 2 : 1;
                a.lf = hv_lf;
                std::string err;
                rc = i8_pass(&p->i8ws, d, Scap, a, st, nullptr, &err);
                if (rc) return fail(rc, "%s", err.c_str());
                RCCHK(upload_rows(live, dRows));
                launch_pcg_step(dRows, (int)live.size(), X, PG, G, kind, Qp, dS1cg, s2, dFidx, dMsCg, capP, Hp, D, Rv, Pv, dpgF, dCg, st);
                HIPCHK(stg.d2h(cgs.data(), dCg, sizeof(CgState) * Rp));
                HIPCHK(hipGetLastError());
                HIPCHK(stg.sync());
                ++stats->hessian_passes;
                stats->hv_evals += n;
                std::vector<int> nxt;
                for (int r : live) {
                    const double eta = std::min(cg_eta, std::sqrt(std::max(kkt[r], 1e-300)));
                    if (cgs[r].pHp > 0 && cgs[r].rs > eta * eta * cgs[r].rs0) nxt.push_back(r);
                }
                if (o.verbose >= 2) fprintf(stderr, "[gml]   cg %2d: %zu rows live\n", ci, nxt.size());
                live.swap(nxt);
                if (live.empty()) break;
                RCCHK(upload_rows(live, dRows));
                launch_newton_solve(dH, dHoff, dMt, dMsCg, dS1, s2, dgF, dpgF, (int)R, capP, dsol, dSdiag, st);
                launch_pcg_dir(dRows, (int)live.size(), Qp, dFidx, dMsCg, capP, dsol, dDinv, Rv, Zv, Pv, 0, dCg, st);
            }
        }
        HIPCHK(stg.sync());
        stats->t_hess += gml_now_s() - th1;

        // ---- projected backtracking line search ----------------------------------------------------------------------------------------
        // Two acceptance regimes per row:
        //  * the predicted decrease is well above the uncertainty of f  -> Armijo on F;
        //  * otherwise ("noise regime": near the optimum, or a noisy int8-limb f) function values cannot certify the step; the trial
        //    is then a full pass and is accepted iff the directional derivative of F at the trial point back towards x is >= 0 (up
        //    to an overshoot allowance): F is convex, so such a trial cannot have increased F.
        // The passes of this iteration use fresh slot ranges: the V planes of the previous ones are no longer needed.
        slot_next = 0;
        for (int64_t r = 0; r < R; ++r) {
            need[r] = !done[r];
            alpha[r] = 1.0;
            accepted_fwd[r] = 0;
        }
        for (int ls = 0; ls < 30; ++ls) {
            const double th2 = gml_now_s();
            std::vector<int> rows;
            for (int64_t r = 0; r < R; ++r)
                if (need[r]) rows.push_back((int)r);
            if (rows.empty()) break;
            RCCHK(upload_rows(rows, dRows));
            HIPCHK(stg.h2d(dAlpha, alpha.data(), sizeof(double) * R));
            stage("trial");
            launch_trial(dRows, (int)rows.size(), X, D, PG, kind, Qp, lambda, dAlpha, Xt, dTrial, st);
            HIPCHK(stg.d2h(trial.data(), dTrial, sizeof(TrialOut) * Rp));
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync());
            bool anynoise = false;
            for (int r : rows) {
                dd[r] = trial[r].dd;
                stepn[r] = trial[r].stepn; // ||trial - x||_1: bounds the change of every energy
                l1t[r] = trial[r].l1t;
                nreg[r] = !(-0.1 * dd[r] > 8.0 * fn[r]); // the step's expected decrease (~|dd|/2) vs the uncertainty of f
                anynoise |= nreg[r] != 0;
            }
            stats->t_host += gml_now_s() - th2;
            const bool full = (ls == 0) || anynoise;
            rc = run_pass(rows, Xt, Gt, full, true, ft, Zt, fnt, nullptr, 0, prec);
            if (rc) return rc;
            const double th3 = gml_now_s();
            if (full) {
                launch_back(dRows, (int)rows.size(), X, Xt, Gt, kind, Qp, lambda, dTrial, st);
                HIPCHK(stg.d2h(trial.data(), dTrial, sizeof(TrialOut) * Rp));
                HIPCHK(stg.sync());
            }
            std::vector<int> acc;
            for (int r : rows) {
                bool ok;
                if (nreg[r]) {
                    const double back = trial[r].back;
                    ok = std::isfinite(ft[r]) && std::isfinite(back) && back >= -0.5 * std::fabs(dd[r]);
                } else {
                    const double Fn = ft[r] + l1t[r];
                    ok = std::isfinite(Fn) && Fn <= Fobj[r] + 1e-4 * dd[r] + fn[r] + fnt[r];
                }
                if (o.verbose >= 2 && r == dbg_row)
                    fprintf(stderr, "[gml]   row %d: ls %d alpha %.3g nreg %d ft %.12e Fobj %.12e dd %.3e fnt %.3e back %.3e ok %d\n", r, ls, alpha[r],
                            (int)nreg[r], ft[r], Fobj[r], dd[r], fnt[r], trial[r].back, (int)ok);
                if (ok) {
                    dref[r] = 0.0; // the iterate moves onto the point the scale was measured at
                    f[r] = ft[r];
                    fn[r] = fnt[r];
                    Z[r] = Zt[r];
                    acc.push_back(r);
                    if (!full) accepted_fwd[r] = 1; // (its V planes still belong to the old iterate)
                    need[r] = 0;
                } else {
                    if (prec != GML_PREC_I8X) {
                        vstale[r] = 1; // the FP64 path's V is indexed by row: every trial overwrites it
                    } else if (full) { // the planes just written belong to the rejected point: back to those of the iterate, if they survive
                        const int pv = vprev[r];
                        if (pv >= 0 && owner[pv] == r) vslot[r] = pv;
                        else vstale[r] = 1;
                    }
                    alpha[r] *= 0.5;
                    if (nreg[r] && alpha[r] < 1.0 / 64) {
                        need[r] = 0;   // cannot improve along this direction: the stall counter ends the row,
                        stall[r] += 3; // after at most three such line searches (each costs ~7 passes)
                    }
                }
            }
            RCCHK(upload_rows(acc, dRows2));
            launch_copy_rows(dRows2, (int)acc.size(), Qp, Xt, X, full ? Gt : nullptr, G, st);
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync());
            stats->t_host += gml_now_s() - th3;
        }
        // rows accepted on an objective-only trial still need their gradient (and V)
        {
            std::vector<int> rows;
            for (int64_t r = 0; r < R; ++r)
                if (accepted_fwd[r]) rows.push_back((int)r);
            if (!rows.empty()) {
                rc = run_pass(rows, X, G, true, false, f, Z, fn, nullptr, 0, prec);
                if (rc) return rc;
            }
        }
        // rows whose line search failed entirely stay where they are; the stall counter ends them
    }

    // ---- results in the reference layout -----------------------------------------------------------------------------------------------
    int notconv = 0;
    double maxk = 0;
    std::vector<int> frombest, fromx;
    for (int64_t r = 0; r < R; ++r) {
        const double k = std::min(best[r], kkt[r]);
        if (!(k <= o.tol)) ++notconv;
        maxk = std::max(maxk, k);
        if (kkt_out) kkt_out[r] = k;
        (best[r] <= kkt[r] ? frombest : fromx).push_back((int)r);
    }
    // best iterate per row -> Xt (free now), one download
    RCCHK(upload_rows(frombest, dRows));
    launch_copy_rows(dRows, (int)frombest.size(), Qp, Xb, Xt, nullptr, nullptr, st);
    RCCHK(upload_rows(fromx, dRows2));
    launch_copy_rows(dRows2, (int)fromx.size(), Qp, X, Xt, nullptr, nullptr, st);
    // reference layout on the device (the direction array D is free now), then ONE copy: to the caller's host matrix, or --
    // `out` a device pointer (gml_multi_learn's dev_out blocks, device-side callers) -- device to device
    double *dres = D; // [R][P], P <= Qp
    int32_t *dcols = nullptr;
    if (p->order != 2) { // multi-body key order (:94-104): column of every parameter slot, from the host
        std::vector<int32_t> cols((size_t)R * P);
        gml_parallel_for(R, [&](int64_t r) {
            NodeLayout L;
            gml_build_layout(p, p->node0 + r, L);
            std::memcpy(cols.data() + (size_t)r * P, L.cols.data(), sizeof(int32_t) * P);
        });
        HIPCHK(A.get(&dcols, (size_t)R * P));
        HIPCHK(hipMemcpyAsync(dcols, cols.data(), sizeof(int32_t) * R * P, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st)); // cols is a local
    }
    launch_rows_to_reference(Xt, R, Qp, P, p->node0, d.cconst, dcols, dres, st);
    HIPCHK(hipGetLastError());
    hipPointerAttribute_t attr;
    bool dev_out = false;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess) dev_out = (attr.type == hipMemoryTypeDevice);
    else (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(out, dres, sizeof(double) * R * P, dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIPCHK(stg.sync());
    stats->iterations = it;
    stats->max_kkt = maxk;
    stats->not_converged = notconv;
    stats->t_total = gml_now_s() - t_start;
    stats->t_pack = p->t_ingest[3];
    if (stats_out) *stats_out = *stats;
    if (notconv)
        return fail(GML_ENOTCONV, "%d of %lld nodes did not reach the KKT tolerance %.1e (worst %.3e)", notconv, (long long)R, o.tol, maxk);
    return GML_OK;
}
