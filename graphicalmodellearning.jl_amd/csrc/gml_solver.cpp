// gml_learn: the batched l1 solver of libgml_hip, device-resident.
//
// Every local node u solves   min_x f_u(x) + lambda * sum_{j penalised} |x_j|   -- the problem the reference builds
// for Ipopt with the z >= |x| epigraph (GraphicalModelLearning.jl:166-177, one model per node in the loop :161) -- and
// all nodes advance in lock-step (Solver::iterate):
//   1. select        pseudo-gradient / KKT residual and working set per node (k_select); converged nodes drop out;
//                    rows the int8-limb arithmetic cannot bring below tol continue on the FP64 path (polish);
//   2. directions    Hessian blocks by one device kernel over a sub-sample of the configurations + batched Cholesky
//                    (newton_blocks: working sets up to max_working entries), or matrix-free conjugate gradients with
//                    Hessian-vector products from the same GEMM kernels (newton_cg: larger working sets, dense optima);
//   3. line_search   projected (orthant-wise) backtracking: the first trial is a full pass (it usually succeeds), further
//                    trials are objective-only passes over the rows that need them.
// One device pass (run_pass) gives f and the full gradient of every listed row (int8-limb or FP64 MFMA kernels).
// The iterates X, gradients G, trial points, directions and pseudo-gradients are [rows][Qp] arrays that stay in HBM;
// per iteration only per-row scalars (a few dozen bytes per node) and the small control blocks of the passes cross PCIe.
// A pass evaluates exactly the active rows, packed into consecutive slots of the int8-limb workspace (full MFMA tiles
// whatever subset is still active).
//
// Host <-> device traffic of an iteration is batched: one upload per pass (its control block), one per direction phase, one
// download per pass (SlotResult), and the host waits for the stream only where it must decide something: after select,
// after the trial points, and after each pass (whose acceptance scalars ride along).  On small node shards these round
// trips, not the kernels, are what an iteration costs.
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
        // also reached by the early returns of a failed solve, with kernels possibly still queued on the blocks: wait once,
        // then the blocks may go back to the library's cache (which hands them out without waiting)
        if (!ptrs.empty()) (void)hipDeviceSynchronize();
        for (void *q : ptrs)
            if (q) (void)dev_free_synced(q);
    }
    // give one block back before the arena dies (regrown buffers)
    void release(void *q) {
        for (auto &x : ptrs)
            if (x == q && q) {
                (void)dev_free(q);
                x = nullptr;
            }
    }
    template <typename T> hipError_t get(T **out, size_t count) {
        *out = nullptr;
        hipError_t e = dev_malloc(reinterpret_cast<void **>(out), sizeof(T) * std::max<size_t>(count, 1));
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

// Small host <-> device transfers of the solver (control blocks up, per-row scalars down) go through one pinned arena:
// a copy from or to pageable memory is staged by the runtime and costs the host 20-30 us each.  h2d copies the bytes into
// the arena and queues an asynchronous copy from there -- the caller's buffer is free as soon as h2d returns; d2h queues
// the copy into the arena and hands the bytes to the caller's buffer at the next sync(), which is the only place the
// stream is waited for.  A transfer too large for the arena goes the plain way and is waited for at once.
struct Stage {
    char *base = nullptr;
    size_t cap = 0, off = 0, floor = 0; // [0, floor): arrays that live as long as the solve (persist); the rest is recycled at every sync()
    bool mapped = false;                // the device reads and writes the arena under its host address (checked by the solver's init)
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
    // Zero-copy: the arena is pinned, mapped host memory, so a kernel can read its control data from it and write its per-row results
    // into it directly.  On small node shards an iteration is a chain of short launches, and every hipMemcpyAsync in it was a 4-us blit
    // kernel plus 5 us of host time (profiles/r5_shard128_trace_before.txt: 15 per iteration).
    // persist: an array the kernels write and the host reads after sync(), for the whole solve (before the first take only)
    template <typename T> T *persist(size_t count) {
        const size_t a = (floor + 63) & ~(size_t)63;
        // only while nothing has been taken from the recycled part: [floor, off) may hold bytes of a copy still in flight
        if (!mapped || off != floor || !pend.empty() || a + sizeof(T) * count > cap / 2) return nullptr;
        floor = a + sizeof(T) * count;
        if (off < floor) off = floor;
        return reinterpret_cast<T *>(base + a);
    }
    // put: control data for kernels queued before the next sync(); NULL when it does not fit (the caller copies to a device buffer)
    template <typename T> const T *put(const T *host, size_t count) {
        void *q = mapped && sizeof(T) * count <= cap / 4 ? take(sizeof(T) * std::max<size_t>(count, 1)) : nullptr;
        if (q && count) std::memcpy(q, host, sizeof(T) * count);
        return reinterpret_cast<const T *>(q);
    }
    hipError_t h2d(void *dev, const void *host, size_t n) {
        if (n == 0) return hipSuccess;
        void *q = n <= cap / 4 ? take(n) : nullptr;
        if (!q) {
            hipError_t e = hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st);
            return e != hipSuccess ? e : hipStreamSynchronize(st); // the caller's buffer may die after return
        }
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
        off = floor;
        return e;
    }
};

// Device time of the direction phase (Hessians, Cholesky, CG): measured with events, so that the host need not wait for
// the stream there just to read a clock.
struct PhaseTimer {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipStream_t st = nullptr;
    ~PhaseTimer() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    void mark() {
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev.push_back(e);
        }
        (void)hipEventRecord(ev[used++], st);
    }
    double seconds() { // sum over the (begin, end) pairs; call after the stream has been waited for
        double s = 0;
        for (size_t i = 0; i + 1 < used; i += 2) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) s += ms * 1e-3;
        }
        return s;
    }
};

template <typename T> const T *rebase(const T *p, int64_t lo) { // p' with p'[lo + i] == p[i] (indexed by absolute slot on the device only)
    return reinterpret_cast<const T *>(reinterpret_cast<uintptr_t>(p) - sizeof(T) * (size_t)lo);
}

struct Solver {
    // ---- problem, options ----------------------------------------------------------------------------------------------
    gml_problem *p;
    const DevProblem &d;
    const int formulation;
    gml_opts o;
    gml_stats stats{};
    hipStream_t st;
    // (experiment, gml_test_tune GML_TUNE_DUAL_STREAMS: the passes on a stream of the lowest priority, everything else on one of the highest,
    // so that another handle's pass running on the same GPU lets this handle's short direction-phase kernels in as workgroups retire)
    hipStream_t st_pass = nullptr, st_own_hi = nullptr, st_own_lo = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bool dual = false;
    const double *x0 = nullptr; // warm start (gml_learn_warm): the rows to start from, reference layout [R][P], host or device; NULL = zeros
    int32_t *dColsRef = nullptr; // multi-body: column of every parameter slot of every local row (:94-104), built once (warm start, finish)
    int compact_skip = 0, compact_backoff = 0; // passes that do not try the column compaction after one that came out dense (run_pass)
    bool underflow = false; // the solve ended because a row's weights left the fixed-point range at an iterate (gml_learn: auto -> FP64)
    bool at_zero = false; // the pass being queued evaluates X = 0 (the first pass of a solve; also its rescaled re-runs)
    ~Solver() { // (each handle by itself: init may have returned between two of the four creations)
        if (st_own_hi) (void)hipStreamSynchronize(st_own_hi);
        if (st_own_lo) (void)hipStreamSynchronize(st_own_lo);
        if (ev_a) (void)hipEventDestroy(ev_a);
        if (ev_b) (void)hipEventDestroy(ev_b);
        if (st_own_hi) (void)hipStreamDestroy(st_own_hi);
        if (st_own_lo) (void)hipStreamDestroy(st_own_lo);
    }
    double tune[GML_NTUNE] = {}; // the experiment knobs as they stood when the solve began (gml_test_tune may be called meanwhile)
    Stage stg;
    Arena A;
    PhaseTimer dir_time;
    double t_dir_host = 0; // host time between the two marks of the direction phases (launches, and the waits of the CG steps)
    const int64_t R, Rp, Qp, P;
    const size_t nd;
    int capW = 0, capP = 0;
    double lambda = 0;
    int64_t Smain = 0, Scap = 0; // slots of the int8-limb workspace: main range (V planes feed the Hessians) + scratch range
    double viol_frac = 0.5, cg_eta = 0.05, eta_admit = 0.25;
    int maxcg = 16, hv_lf = 2, hv_lb = 2;
    int64_t dbg_row = 0, worst_row = 0;
    const double t_begin = gml_now_s();
    // precision i8w: the passes run in their coarse form (30-bit theta, 23-bit weights) until the first active row comes within
    // coarse_thr of its optimum, then at full width for the rest of the solve
    bool coarse_on = false;
    double coarse_thr = 1e-7;
    int prec = GML_PREC_I8X; // arithmetic of the passes: switches to FP64 for the rows the int8 path leaves above tol ("polish")
    bool can_polish = false;
    int stall_cap = 10;

    // ---- device state ----------------------------------------------------------------------------------------------------
    double *X = nullptr, *G = nullptr, *Xt = nullptr, *Gt = nullptr, *Xb = nullptr, *D = nullptr, *PG = nullptr, *Gs = nullptr;
    double *Rv = nullptr, *Pv = nullptr, *Hp = nullptr, *Zv = nullptr; // CG vectors, on first use
    uint8_t *Wm = nullptr;                                             // ... and the mask of the system's coordinates
    FaceOut *dFaces = nullptr;
    bool use_secant = true;
    int face_rounds = 2;      // re-solves of a CG system without the coordinates its step would push through zero
    double face_share = 0.05; // ... for the rows where those carry more than this share of the predicted decrease
    uint8_t *kind = nullptr;
    int *dNode = nullptr, *dRows = nullptr, *dRows2 = nullptr, *dRowsP = nullptr, *dFidx = nullptr, *dHv = nullptr;
    char *dPass = nullptr;  // control block of a pass: srow | rowcol | groups | tau overrides (one upload)
    char *dHctl = nullptr;  // control block of the direction phase (one upload)
    SlotResult *dRes = nullptr;
    CgState *dCg = nullptr;
    double *dBest = nullptr, *dAlpha = nullptr, *dFs = nullptr, *dgF = nullptr, *dpgF = nullptr, *dsol = nullptr,
           *dSdiag = nullptr, *dH = nullptr;
    SelectOut *dSel = nullptr;
    TrialOut *dTrial = nullptr;
    int64_t dH_elems = 0;
    // pointers into dHctl (set by direction_blocks)
    int *dMt = nullptr, *dVslot = nullptr;
    long long *dHoff = nullptr;
    double *dS1 = nullptr, *dS1cg = nullptr;
    // secant pairs of the Cholesky rows (k_secant): previous working set, x and g on it, the last two (s, y)
    int *dFprev = nullptr, *dMprev = nullptr, *dNpairs = nullptr;
    double *dXprev = nullptr, *dGprev = nullptr, *dSec = nullptr, *dYnoise = nullptr;
    // preconditioner tiles of the matrix-free rows (direction_blocks): control block, column lists, gradient entries
    static constexpr int kTile = 128;
    char *dTctl = nullptr;
    size_t dTctl_bytes = 0;
    int *dFV = nullptr, *dVm = nullptr, *dWrow = nullptr, *dLive = nullptr;
    double *dgV = nullptr;
    // the last rows' Hessian-vector products on the vector ALUs over their working sets only (gml_hv_sparse.hip)
    const long long *dT0m = nullptr; // first tile of each local row's working-set list (device; direction_blocks)
    int *dNw = nullptr;              // |W| by local row
    void *dHvsBuf = nullptr;
    size_t hvs_bytes = 0;
    bool hv_sparse = true;
    double hv_sparse_ratio = g_hv_sparse_ratio; // sparse when sum |W| of the live rows < ratio * columns * node tiles of the GEMM pass
    int64_t n_hv_sparse = 0;
    int64_t tile_cap = 0, ntiles = 0, tile_base = 0; // capacity of dFV / dgV in tiles; tiles of this iteration; offset of the first in dH

    // ---- host state (scalars per row) ------------------------------------------------------------------------------------
    std::vector<double> f, ft, Fobj, kkt, best, Z, Zt, alpha, dd, fn, fnt, l1t, Fbest;
    std::vector<uint8_t> done, vstale, atfloor, nreg, accepted_fwd, need, iscg;
    std::vector<int> stall, msz, nW, vslot, vprev, owner, pslot;
    // per-row results of k_select / k_trial + k_back / the passes' last kernels: written by the kernels straight into pinned host
    // memory (Stage::persist) and read here after the stream has been waited for; device arrays + one download each when the
    // pinned arena is too small for them (tens of thousands of local rows)
    SelectOut *sel = nullptr, *kSel = nullptr;
    TrialOut *trial = nullptr, *kTrial = nullptr;
    SlotResult *res = nullptr, *kRes = nullptr;
    std::vector<SelectOut> sel_v;
    std::vector<TrialOut> trial_v;
    std::vector<SlotResult> res_v;
    bool zc = false;
    double *dStepn = nullptr; // |trial - x|_1 by row, left on the device by k_trial: scales the weights' unit of the trial pass (k_quant_theta)
    int64_t slot_next = 0;
    // Scale of the fixed-point V (int8 path): instead of the worst-case bound w_max exp(sum|theta|) every pass after a row's
    // first uses vref = max_k |V_rk| measured by its previous pass, times exp(||theta - theta_ref||_1), which bounds the new
    // weights rigorously (|E_k' - E_k| <= ||theta' - theta||_1).  Near the optimum the steps are tiny, so V keeps all 31 bits
    // relative to its actual maximum and the noise floor of f and grad drops by the bits the bound would have wasted.
    std::vector<double> vref, dref, stepn;
    // sub-sampled Newton: Hessians over Kh configurations -- every kstride-th block of 512 (set_kh)
    int64_t Kh_base = 0, nblk512 = 0, Kh = 0, kstride = 1;
    double hscale = 1.0;

    Solver(gml_problem *p_, int form, const gml_opts &o_, double lam)
        : p(p_), d(p_->d), formulation(form), o(o_), st(p_->st), R(p_->node1 - p_->node0), Rp(gml_round_up(R, 32)), Qp(p_->d.Qp), P(p_->P),
          nd((size_t)Rp * p_->d.Qp), lambda(lam) {}

    void trace(const char *name) {
        if (o.verbose >= 3) {
            (void)stg.sync();
            fprintf(stderr, "[gml]     stage %s at %.4f s (last error: %s)\n", name, gml_now_s() - t_begin, hipGetErrorString(hipGetLastError()));
            fflush(stderr);
        }
    }
    // still building its support: violators at the scale of the support itself (k_select admits them by halves of the largest
    // violation; once they are few next to the support, all at once)
    bool admitting(int64_t r) const { return (int64_t)sel[r].nviol * 16 > sel[r].nsupp; }
    int upload_rows(const std::vector<int> &rows, int *dst) {
        if (!rows.empty()) HIPCHK(stg.h2d(dst, rows.data(), sizeof(int) * rows.size()));
        return GML_OK;
    }
    // a row list (or any small control array) for kernels queued before the next wait: read from the pinned arena where it fits,
    // else uploaded to `fallback`
    template <typename T> int ctl(const std::vector<T> &v, T *fallback, const T **out) {
        *out = fallback;
        if (v.empty()) return GML_OK;
        if (const T *q = stg.put(v.data(), v.size())) {
            *out = q;
            return GML_OK;
        }
        HIPCHK(stg.h2d(fallback, v.data(), sizeof(T) * v.size()));
        return GML_OK;
    }
    int fetch(void *host, const void *dev, size_t bytes) { // results the kernels left in device arrays (not zero-copy)
        if (!zc) HIPCHK(stg.d2h(host, dev, bytes));
        return GML_OK;
    }

    int init();
    void set_kh(int64_t nactive, int maxm = 512);
    // One objective(/gradient) pass over the listed rows.  src: X or Xt; dst: G or Gt (want_grad); results per row: fo (f,
    // or log Z), zo (Z, logRISE), no (noise of f).  pp: the arithmetic of this pass (the solver's current precision; the
    // FP64 phase borrows int8 passes for the V planes its matrix-free rows need).  after: queued on the stream right behind
    // the pass and before the host waits for it (the acceptance scalars of a trial ride along with the pass results).
    int run_pass(const std::vector<int> &rows, const double *src, double *dst, bool want_grad, bool at_trial, std::vector<double> &fo,
                 std::vector<double> &zo, std::vector<double> &no, const std::vector<double> *ovr_in, int depth, int pp,
                 const std::function<int()> *after = nullptr, bool step_on_device = false);
    int select(int it, int64_t *nactive);
    int start_polish(bool *started);
    int refresh_stale();
    int direction_blocks(const std::vector<int> &cg_rows);
    int newton_blocks(const std::vector<int> &chol_rows);
    int newton_cg(const std::vector<int> &cg_rows);
    int newton_cg_group(const std::vector<int> &cg_rows, bool subsample);
    int line_search();
    int finish(double *out, double *kkt_out, int iterations);
    int ref_cols();
    int load_x0();
    int iterate(double *out, double *kkt_out);
};

int Solver::init() {
    if (!(o.tol > 0)) o.tol = 1e-9;
    if (o.max_iter <= 0) o.max_iter = 100;
    if (o.max_working < 32) o.max_working = 512;
    if (o.max_working > 512) o.max_working = 512;
    o.max_working = (int)gml_round_up(o.max_working, 32);
    if (o.max_add <= 0) o.max_add = 64;
    capW = capP = o.max_working;
    // Matrix-free rows admit, per iteration, only the violators within this fraction of the largest violation (measured on
    // the order-3 config at the reference's default regulariser: with every violator at once -- 0 -- or a quarter of the
    // largest -- 0.25 -- the projected Newton steps are damped to nothing by the line search; 0.5 takes full steps throughout;
    // 0.7 converges too, in 1.6x the iterations)
    viol_frac = o.cg_viol_frac > 0 ? o.cg_viol_frac : 0.5;
    cg_eta = o.cg_eta > 0 ? o.cg_eta : 0.05;
    // at most 16 CG steps per Newton step: the number of Newton iterations is set by the admission of the violators, not by
    // the accuracy of the directions (order-3 probe: 32 iterations with a cap of 40, 16 or 15 -- 545 / 349 / 333
    // Hessian-vector passes; 40 iterations, 279 passes with a cap of 8)
    maxcg = o.max_cg > 0 ? o.max_cg : 16;
    // limbs of the CG direction p in the Hessian-vector passes: 2 (14 bits of max|p|) are enough for an inexact Newton
    // step that stops at a residual of 5 % -- the iteration counts of the 64-node probe of config 5 are 59 / 59 / 56
    // with 4 / 3 / 2 limbs, and the forward GEMM of an H.v pass costs in proportion
    hv_lf = o.hv_limbs_fwd ? std::min(5, std::max(2, (int)o.hv_limbs_fwd)) : 2;
    // ... and the products u_k = h_k (x_k . p) go to the backward GEMM in 2 limbs (15 bits of the largest) instead of 4:
    // half the MFMAs and half the V reads of that GEMM; config 5 at the default regulariser 116 -> 93 s with 10 % more
    // iterations
    hv_lb = o.hv_limbs_bwd == 4 ? 4 : 2;
    dbg_row = o.debug_row > 0 ? o.debug_row : 0;
    if (o.limbs_fwd != 0 && o.limbs_fwd != 3 && o.limbs_fwd != 4 && o.limbs_fwd != 5) return fail(GML_EINVAL, "limbs_fwd must be 3, 4 or 5");

    if (!p->stage) {
        p->stage_bytes = (size_t)8 << 20;
        // (portable + mapped: one process may drive several devices -- gml_multi -- and the kernels of this handle's device read and
        // write the arena directly)
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->stage), p->stage_bytes, hipHostMallocPortable | hipHostMallocMapped));
    }
    {
        void *dp = nullptr; // zero-copy only where the device sees the arena under the same address (unified addressing); else copies
        stg.mapped = hipHostGetDevicePointer(&dp, p->stage, 0) == hipSuccess && dp == static_cast<void *>(p->stage);
        if (!stg.mapped) (void)hipGetLastError();
        for (int i = 0; i < GML_NTUNE; ++i) tune[i] = g_tune[i];
        if (tune[GML_TUNE_NO_ZEROCOPY] > 0) stg.mapped = false; // (tests: the copy path that very large handles take)
    }
    st_pass = st;
    if (tune[GML_TUNE_DUAL_STREAMS] > 0 && gml_is_i8(o.precision)) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamSynchronize(st)); // (what the handle's own stream still holds: the samples' images)
        HIPCHK(hipStreamCreateWithPriority(&st_own_hi, hipStreamNonBlocking, greatest));
        HIPCHK(hipStreamCreateWithPriority(&st_own_lo, hipStreamNonBlocking, tune[GML_TUNE_DUAL_STREAMS] > 1 ? greatest : least));
        HIPCHK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
        st = st_own_hi;
        st_pass = st_own_lo;
        dual = true;
    }
    stg.base = p->stage;
    stg.cap = p->stage_bytes;
    stg.st = st;
    stg.off = stg.floor = 0;
    dir_time.st = st;
    stats.lambda = lambda;
    prec = o.precision;
    // (launch-bound problems gain nothing from cheaper passes and pay for the re-evaluation at the switch: config 2 takes 16 + 3
    // passes instead of 13 + 2)
    // (decided on the whole problem -- n, not the local rows -- so that a node shard runs the form its problem would run on one GPU;
    // WHEN the phase ends still depends on the rows of this handle: it ends for all of them with the first that gets close, a per-row
    // switch having been measured slower, DESIGN.md 4.2 -- trajectories differ between shardings, optima do not)
    coarse_on = gml_is_i8(o.precision) && o.coarse >= 0 && formulation != GML_RPLE && (double)p->K * (double)Qp * (double)p->n >= 17179869184.0;
    if (o.coarse > 0) coarse_thr = std::pow(10.0, -(double)o.coarse); // (tuning: the KKT residual at which the coarse phase ends)

    // slots of the int8-limb workspace: every active row of a pass in its own slot, the passes of one iteration in
    // disjoint ranges (their V planes feed the next Hessians).  Objective-only passes (line-search trials whose V planes
    // nobody reads) run in a scratch range above the main one.
    Smain = gml_is_i8(o.precision) ? Rp + gml_round_up(std::max<int64_t>(R / 2, 96), 32) + 64 : Rp + 64;
    Scap = Smain + Rp;
    {
        size_t freeb = 0, totalb = 0;
        HIPCHK(dev_mem_info(&freeb, &totalb));
        const double need_b = 7.0 * 8.0 * (double)nd + (double)nd;
        if (need_b > 0.9 * (double)freeb)
            return fail(GML_ENOMEM, "solver state of %.1f GB for %lld rows does not fit in %.1f GB free HBM", need_b / 1e9, (long long)R,
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
    HIPCHK(A.get(&dPass, (size_t)(sizeof(int) * (2 * Scap + Scap / 32 + 16) + sizeof(double) * Scap + 64)));
    HIPCHK(A.get(&dHctl, (size_t)(sizeof(int) * 5 * Rp + sizeof(long long) * (Rp + 1) + sizeof(double) * 3 * Rp + 256)));
    HIPCHK(A.get(&dLive, (size_t)Rp));
    HIPCHK(A.get(&dFprev, (size_t)Rp * capP));
    HIPCHK(A.get(&dMprev, (size_t)Rp));
    HIPCHK(A.get(&dNpairs, (size_t)Rp));
    HIPCHK(A.get(&dXprev, (size_t)Rp * capP));
    HIPCHK(A.get(&dGprev, (size_t)Rp * capP));
    HIPCHK(A.get(&dSec, (size_t)4 * Rp * capP)); // S[0], S[1], Y[0], Y[1]
    HIPCHK(hipMemsetAsync(dMprev, 0, sizeof(int) * Rp, st));
    HIPCHK(hipMemsetAsync(dNpairs, 0, sizeof(int) * Rp, st));
    HIPCHK(A.get(&dRes, (size_t)Scap));
    HIPCHK(A.get(&dFidx, (size_t)Rp * capP));
    HIPCHK(A.get(&dCg, (size_t)Rp));
    HIPCHK(A.get(&dBest, (size_t)Rp));
    HIPCHK(A.get(&dAlpha, (size_t)Rp));
    HIPCHK(A.get(&dFs, (size_t)Scap));
    HIPCHK(A.get(&dgF, (size_t)Rp * capP));
    HIPCHK(A.get(&dpgF, (size_t)Rp * capP));
    HIPCHK(A.get(&dsol, (size_t)Rp * capP));
    HIPCHK(A.get(&dSdiag, (size_t)Rp));
    HIPCHK(A.get(&dSel, (size_t)Rp));
    HIPCHK(A.get(&dTrial, (size_t)Rp));
    HIPCHK(A.get(&dStepn, (size_t)Rp));
    HIPCHK(hipMemsetAsync(dStepn, 0, sizeof(double) * Rp, st));
    sel = stg.persist<SelectOut>((size_t)Rp);
    trial = stg.persist<TrialOut>((size_t)Rp);
    res = stg.persist<SlotResult>((size_t)Scap);
    zc = sel && trial && res;
    if (zc) {
        kSel = sel;
        kTrial = trial;
        kRes = res;
        std::memset(sel, 0, sizeof(SelectOut) * Rp);
        std::memset(trial, 0, sizeof(TrialOut) * Rp);
        std::memset(res, 0, sizeof(SlotResult) * Scap);
    } else {
        stg.off = stg.floor = 0;
        sel_v.resize((size_t)Rp);
        trial_v.resize((size_t)Rp);
        res_v.resize((size_t)Scap);
        sel = sel_v.data();
        trial = trial_v.data();
        res = res_v.data();
        kSel = dSel;
        kTrial = dTrial;
        kRes = dRes;
    }
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
    }
    launch_kind(d, p->order, dNode, (int)Rp, kind, st);

    f.assign((size_t)R, 0.0);
    ft.assign((size_t)R, 0.0);
    Fobj.assign((size_t)R, 0.0);
    kkt.assign((size_t)R, INFINITY);
    best.assign((size_t)R, INFINITY);
    Z.assign((size_t)R, 1.0);
    Zt.assign((size_t)R, 1.0);
    alpha.assign((size_t)R, 1.0);
    dd.assign((size_t)R, 0.0);
    fn.assign((size_t)R, 0.0);
    fnt.assign((size_t)R, 0.0);
    l1t.assign((size_t)R, 0.0);
    Fbest.assign((size_t)R, INFINITY);
    for (auto *v : {&done, &vstale, &atfloor, &nreg, &accepted_fwd, &need, &iscg}) v->assign((size_t)R, 0);
    stall.assign((size_t)R, 0);
    msz.assign((size_t)R, 0);
    nW.assign((size_t)R, 0);
    vslot.assign((size_t)R, -1);
    vprev.assign((size_t)R, -1);
    pslot.assign((size_t)R, -1);
    owner.assign((size_t)Scap, -1);
    vref.assign((size_t)R, 0.0);
    dref.assign((size_t)R, 0.0);
    stepn.assign((size_t)R, 0.0);

    if (gml_is_i8(o.precision) && o.polish >= 0) {
        size_t freeb = 0, totalb = 0;
        if (dev_mem_info(&freeb, &totalb) == hipSuccess) {
            const double need_b = (d.Xt ? 0.0 : (double)d.Kp * (double)Qp) + (p->dV && p->dVrows >= Rp ? 0.0 : 8.0 * (double)Rp * (double)d.Kp) +
                                  8.0 * (double)nd + (o.precision == GML_PREC_I8W ? 6.0 : 4.0) * (double)Scap * d.Kp /* the i8 workspace still to come */;
            can_polish = need_b < 0.8 * (double)freeb;
        }
    }
    stall_cap = can_polish ? 4 : 10;

    // sub-sampled Newton: the budget (rows x configurations) is kept roughly constant: as nodes converge, the remaining
    // ones get more configurations, up to all of them -- an inexact Hessian only costs iterations, and it costs the most on
    // the few ill-conditioned nodes that are still active at the end.
    // The base: 32 768 configurations per row, more on small node shards -- below ~2^23 row-configurations a Hessian launch costs its
    // latency, not its work, so a shard of 128 rows takes 65 536 per row for the price of 32 768 and saves two of fourteen iterations
    // (profiles/r5_hess_budget_sweep.txt; 1 024 rows: 32 768 -> 109 ms, 49 152 -> 113 ms, 24 576 -> 112 ms).  Like the rule above this
    // makes a row's Newton trajectory -- not its optimum -- depend on how many rows share its GPU.
    // (int8 kernels only: the FP64 path's Hessian kernel is bound by its FP64 work at any size -- n = 200, K = 1e5 on that path took
    // 17 % longer with the larger budget, profiles/r5_robust_sweep.txt)
    Kh_base = 32768;
    if (gml_is_i8(o.precision)) Kh_base = std::min<int64_t>(131072, std::max<int64_t>(32768, ((int64_t)1 << 23) / std::max<int64_t>(R, 1)));
    if (tune[GML_TUNE_KH_BASE] > 0) Kh_base = (int64_t)tune[GML_TUNE_KH_BASE];
    if (o.hess_samples != 0) Kh_base = o.hess_samples < 0 ? d.Kp : (int64_t)o.hess_samples;
    nblk512 = d.Kp / 512;
    set_kh(R);
    return GML_OK;
}

void Solver::set_kh(int64_t nactive, int maxm) {
    // (a budget scaled to the working-set size -- 64 configurations per entry, 8192 at least -- was measured and dropped: the
    // headline problem then needs 20-22 iterations instead of 14)
    (void)maxm;
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
    hscale = nb == nblk512 ? 1.0 : 1.0 / wsum; // rescaled by the weight of the sub-sample
}

int Solver::run_pass(const std::vector<int> &rows, const double *src, double *dst, bool want_grad, bool at_trial, std::vector<double> &fo,
                     std::vector<double> &zo, std::vector<double> &no, const std::vector<double> *ovr_in, int depth, int pp,
                     const std::function<int()> *after, bool step_on_device) {
    const int64_t n = (int64_t)rows.size();
    if (n == 0) return GML_OK;
    const double t0 = gml_now_s();
    trace(want_grad ? "pass" : "fwd pass");
    const int64_t np = gml_round_up(n, 32);
    std::vector<double> fh, tauh;
    std::vector<unsigned> mmh;
    const bool track = gml_is_i8(pp) && formulation != GML_RPLE;
    const bool wide = pp == GML_PREC_I8W;
    if (gml_is_i8(pp)) {
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
            int64_t base = gml_round_up(slot_next, 32);
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
        // control block of the pass, one upload: srow [ns] | rowcol [ns] | tiles, padded with -1 to a multiple of 4 | tau [ns]
        std::vector<uint8_t> tile((size_t)(ns / 32), 0);
        for (int64_t a = 0; a < n; ++a) tile[(slot[a] - lo) / 32] = 1;
        int ng = 0;
        for (int64_t g = 0; g < ns / 32; ++g) ng += tile[g];
        const int ng4 = (int)gml_round_up(ng, 4);
        const size_t ints = (size_t)(2 * ns + ng4), ioff = (ints * sizeof(int) + 7) & ~(size_t)7;
        std::vector<char> blk(ioff + sizeof(double) * ns, 0);
        int *srow = reinterpret_cast<int *>(blk.data()), *rowcol = srow + ns, *groups = rowcol + ns;
        double *ovr = reinterpret_cast<double *>(blk.data() + ioff);
        for (int64_t q = 0; q < ns; ++q) rowcol[q] = -1;
        for (int64_t a = 0; a < n; ++a) {
            const int64_t q = slot[a] - lo;
            const int r = rows[a];
            srow[q] = r;
            rowcol[q] = (int)(p->node0 + r);
            if (ovr_in) ovr[q] = (*ovr_in)[r];
            else if (track && vref[r] > 0.0) // (step_on_device: times exp(|trial - x|_1), which k_trial left in dStepn)
                ovr[q] = vref[r] * std::exp(dref[r] + (at_trial && !step_on_device ? stepn[r] : 0.0)) * (1.0 + 1e-6) / i8_vdiv(wide);
        }
        {
            int k = 0;
            for (int64_t g = 0; g < ns / 32; ++g)
                if (tile[g]) groups[k++] = (int)(lo / 32 + g);
            for (; k < ng4; ++k) groups[k] = -1;
        }
        HIPCHK(stg.h2d(dPass, blk.data(), blk.size()));
        const int *dsrow = reinterpret_cast<const int *>(dPass);
        I8Pass a{};
        a.theta = src;
        a.srow = rebase(dsrow, lo); // indexed by slot on the device
        a.rowcol = rebase(dsrow + ns, lo);
        a.groups = dsrow + 2 * ns;
        a.ngroups = ng;
        a.slot0 = (int)lo;
        a.slot1 = (int)hi;
        a.form = formulation;
        a.want_grad = want_grad;
        a.F = dFs;
        a.G = dst;
        a.tauovr = rebase(reinterpret_cast<const double *>(dPass + ioff), lo);
        a.tauovr_lnrow = !ovr_in && track && step_on_device ? dStepn : nullptr;
        a.res = kRes;
        a.lf = o.limbs_fwd;
        a.wide = wide;
        a.coarse = coarse_on;
        a.zero_theta = at_zero && tune[GML_TUNE_NO_ZERO_SHORTCUT] == 0; // the first pass of a solve: X = 0 for every row
        // Column compaction of the forward GEMM.  It pays where the rows are sparse RELATIVE TO THE COLUMN COUNT: multi-body
        // statistics, thousands of spins (config 5 at c = 1.2: 25 non-zeros of 130 817 per node, learn() 3.5 -> 2.2 s).  With the
        // 1 024 columns of the headline problem a tile's 32 rows cover every column after the third pass and the two extra launches
        // cost a 128-node shard 5 %: problems below 4 096 columns do not try.  A pass whose tiles came out (mostly) dense makes the
        // next 2, 4, ... 16 passes skip the attempt.
        const bool may_compact = tune[GML_TUNE_NO_COMPACT] == 0 && (tune[GML_TUNE_SOLVER_COMPACT] > 0 || d.Qfp >= 4096);
        if (may_compact && compact_skip > 0) --compact_skip;
        else a.compact = may_compact;
        std::string err;
        if (dual) { // the pass behind what this handle has queued so far, on the low-priority stream
            HIPCHK(hipEventRecord(ev_a, st));
            HIPCHK(hipStreamWaitEvent(st_pass, ev_a, 0));
        }
        int rc = i8_pass(&p->i8ws, d, Scap, a, st_pass, nullptr, &err);
        if (rc) return fail(rc, "%s", err.c_str());
        if (dual) { // ... and everything that follows behind the pass
            HIPCHK(hipEventRecord(ev_b, st_pass));
            HIPCHK(hipStreamWaitEvent(st, ev_b, 0));
        }
        if (formulation == GML_LOGRISE && want_grad) // grad log Z = grad Z / Z (:279), Z from the pass results on the device
            launch_scale_slots_inv(a.srow, a.rowcol, (int)lo, (int)ns, kRes, Qp, dst, st);
        RCCHK(fetch(res + lo, dRes + lo, sizeof(SlotResult) * ns));
        if (after) RCCHK((*after)());
        HIPCHK(hipGetLastError());
        std::vector<int> ctab;
        if (a.compact) { // what the compaction made of this pass comes down with its results
            const int *dcnk = nullptr;
            int cs = 0;
            i8_compact_table(p->i8ws, &dcnk, &cs);
            if (dcnk) {
                ctab.assign((size_t)(hi - lo) / 32, 0);
                HIPCHK(stg.d2h(ctab.data(), dcnk + lo / 32, sizeof(int) * ctab.size()));
            }
        }
        HIPCHK(stg.sync());
        if (!ctab.empty()) {
            const int nk_all = (int)(d.Qfp / 64);
            long long swept = 0;
            for (int v : ctab) swept += v < 0 ? nk_all : v;
            if ((double)swept > 0.6 * (double)nk_all * (double)ctab.size()) {
                compact_backoff = std::min(16, std::max(2, 2 * compact_backoff));
                compact_skip = compact_backoff;
            } else {
                compact_backoff = 0;
            }
            if (o.verbose >= 2)
                fprintf(stderr, "[gml]   compaction: %lld of %lld column steps swept over %zu tiles%s\n", swept, (long long)nk_all * (long long)ctab.size(),
                        ctab.size(), compact_skip ? " (the next passes do not try)" : "");
        }
        if (step_on_device) // (the trial's scalars came down with the pass)
            for (int64_t a2 = 0; a2 < n; ++a2) stepn[rows[a2]] = trial[rows[a2]].stepn;
        fh.resize((size_t)n);
        tauh.resize((size_t)n);
        mmh.resize((size_t)n);
        for (int64_t a2 = 0; a2 < n; ++a2) {
            const SlotResult &q = res[(size_t)slot[a2]];
            fh[a2] = q.f;
            tauh[a2] = q.tau;
            mmh[a2] = q.mmax;
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
        int *dctl = reinterpret_cast<int *>(dPass);
        HIPCHK(stg.h2d(dctl, ctl.data(), sizeof(int) * (Rp + ng4)));
        HIPCHK(hipMemsetAsync(dFs, 0, sizeof(double) * Rp, st));
        launch_fwd_f64(d, src, dctl, dctl + Rp, ng4, formulation, p->dV, dFs, st);
        if (want_grad) {
            HIPCHK(hipMemsetAsync(Gs, 0, sizeof(double) * nd, st));
            launch_bwd_f64(d, p->dV, dctl + Rp, ng, Gs, st);
            RCCHK(upload_rows(rows, dRowsP));
            launch_copy_rows(dRowsP, (int)n, Qp, Gs, dst, nullptr, nullptr, st);
            if (formulation == GML_LOGRISE) launch_scale_rows_inv(dRowsP, (int)n, dFs, Qp, dst, st);
        }
        fh.resize((size_t)Rp);
        HIPCHK(stg.d2h(fh.data(), dFs, sizeof(double) * Rp));
        for (int64_t a = 0; a < n; ++a) { // (V [row][Kp] of the FP64 path is indexed by row)
            vstale[rows[a]] = 0;
            if (vslot[rows[a]] < 0) vslot[rows[a]] = 0;
        }
        if (after) RCCHK((*after)());
        HIPCHK(hipGetLastError());
        HIPCHK(stg.sync());
        if (step_on_device)
            for (int64_t a2 = 0; a2 < n; ++a2) stepn[rows[a2]] = trial[rows[a2]].stepn;
    }
    std::vector<int> again;
    std::vector<double> ovr2;
    for (int64_t a = 0; a < n; ++a) {
        const int r = rows[a];
        const double fv = gml_is_i8(pp) ? fh[a] : fh[r];
        // f64: summation rounding.  int8 limbs: every V_rk is rounded to a multiple of tau_r with a dither that is
        // equidistributed over the samples, so the errors (each within one unit, standard deviation 0.41 tau) add like a
        // random walk: 8 sigma of sqrt(K) terms (the worst case K * tau is never approached).
        double noise = 1e-13 * std::max(1.0, std::fabs(fv));
        if (track) {
            noise += 3.3 * std::sqrt((double)p->K) * tauh[a] * (coarse_on ? i8_coarse_unit(wide) : 1.0); // (coarse: multiples of 2^24 / 2^8 tau)
            const double vmax = ((double)mmh[a] + 1.0) * i8_mmax_unit(wide) * tauh[a]; // rigorous bound on max_k |V_rk|
            vref[r] = vmax;
            dref[r] = at_trial ? stepn[r] : 0.0; // distance from the current iterate to the point just evaluated
            // Dynamic range: tau_r was derived from a bound; when the largest |V_rk| actually seen is more than 8 bits
            // below it (dense theta), the row is re-run with tau_r taken from that maximum
            if (mmh[a] < (1u << 23)) {
                if (ovr2.empty()) ovr2.assign((size_t)R, 0.0);
                again.push_back(r);
                ovr2[r] = vmax * (1.0 + 1e-12) / i8_vdiv(wide);
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
    stats.node_evals += n;
    if (want_grad) ++stats.passes;
    else ++stats.forward_passes;
    if (!again.empty()) {
        stats.t_pass += gml_now_s() - t0;
        if (depth >= 6) {
            // the weights exp(-E) of these rows underflow the fixed-point range even after six rescalings (|theta|_1 in the
            // hundreds): a trial point that far out is simply rejected; at the iterate itself it is an error
            if (!at_trial) {
                underflow = true;
                return fail(GML_EUNSUPPORTED, "precision %s: the weights exp(-E) of a row underflow its fixed-point range at an iterate; use precision f64 "
                                              "(or auto, which does so by itself)", wide ? "i8w" : "i8x");
            }
            for (int r : again) {
                fo[r] = INFINITY;
                no[r] = 0.0;
            }
            return GML_OK;
        }
        // (the re-run repeats `after`: what it queued for these rows was computed from the first run's gradient)
        RCCHK(run_pass(again, src, dst, want_grad, at_trial, fo, zo, no, &ovr2, depth + 1, pp, after));
    }
    if (again.empty()) stats.t_pass += gml_now_s() - t0;
    return GML_OK;
}

// KKT residuals and working sets (device); decides which rows are done.
int Solver::select(int it, int64_t *nactive_out) {
    std::vector<int> act;
    for (int64_t r = 0; r < R; ++r)
        if (!done[r]) act.push_back((int)r);
    const int *rows_k = nullptr;
    RCCHK(ctl(act, dRows, &rows_k));
    trace("select");
    launch_select(rows_k, (int)act.size(), X, G, kind, Qp, lambda, o.max_add, capW, capP, viol_frac, PG, dFidx, dgF, dpgF, kSel, dBest, Xb, st);
    RCCHK(fetch(sel, dSel, sizeof(SelectOut) * Rp));
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
        msz[r] = s.m < 0 ? 0 : s.m; // working set of the Cholesky step
        nW[r] = s.m < 0 ? -s.m : 0; // ... of a matrix-free row (listed in tiles for its preconditioner)
        if (iscg[r]) ++ncg;
        maxm = std::max(maxm, msz[r]);
    }
    worst_row = 0;
    for (int64_t r = 0; r < R; ++r) {
        if (done[r]) msz[r] = 0;
        const double k = std::min(kkt[r], best[r]);
        if (k > worst_all) worst_row = r;
        worst_all = std::max(worst_all, k);
    }
    if (o.verbose)
        fprintf(stderr, "[gml] it %3d active %6lld (cg %lld)  max-kkt %.3e  max|W| %d  passes %d fwd %d  hv %lld  t %.2f s%s\n", it, (long long)nactive,
                (long long)ncg, worst_all, maxm, stats.passes, stats.forward_passes, (long long)stats.hv_evals, gml_now_s() - t_begin,
                prec == GML_PREC_F64 && o.precision != prec ? "  [fp64 polish]" : "");
    if (o.verbose >= 2)
        for (int64_t r : {dbg_row, worst_row})
            if (r < R && (r == dbg_row || worst_row != dbg_row))
                fprintf(stderr, "[gml]   row %lld: kkt %.3e best %.3e F %.15e m %d nsupp %d nviol %d stall %d\n", (long long)r, kkt[r], best[r], Fobj[r],
                        sel[r].m, sel[r].nsupp, sel[r].nviol, stall[r]);
    *nactive_out = nactive;
    return GML_OK;
}

// Polish: rows that the int8-limb arithmetic could not bring below tol (its gradient carries ~sqrt(K) 2^-31 of noise relative
// to the largest weight, which an ill-conditioned, weakly regularised problem amplifies) continue on the FP64 path from their
// best iterate, when that path fits in memory.
int Solver::start_polish(bool *started) {
    *started = false;
    std::vector<int> fl;
    for (int64_t r = 0; r < R; ++r)
        if (atfloor[r] && !(std::min(best[r], kkt[r]) <= o.tol)) fl.push_back((int)r);
    if (!(gml_is_i8(prec) && can_polish && !fl.empty())) return GML_OK;
    if (gml_ensure_f64(p, Rp) != GML_OK) return GML_OK; // does not fit after all: the rows stay as they are (reported not converged)
    prec = GML_PREC_F64;
    stall_cap = 10;
    RCCHK(upload_rows(fl, dRows));
    launch_copy_rows(dRows, (int)fl.size(), Qp, Xb, X, nullptr, nullptr, st); // back to the best iterate
    std::vector<double> inf((size_t)Rp, INFINITY);
    HIPCHK(stg.h2d(dBest, inf.data(), sizeof(double) * Rp));
    for (int r : fl) {
        done[r] = 0;
        atfloor[r] = 0;
        stall[r] = 0;
        best[r] = INFINITY;
        Fbest[r] = INFINITY;
    }
    if (o.verbose) fprintf(stderr, "[gml] polish: %zu rows continue on the FP64 path\n", fl.size());
    RCCHK(run_pass(fl, X, G, true, false, f, Z, fn, nullptr, 0, prec));
    set_kh((int64_t)fl.size());
    stats.polished = 1;
    *started = true;
    return GML_OK;
}

// rows whose V planes were overwritten (a wrapped slot range) need a fresh pass before the curvature; such a pass can
// itself overwrite planes that are still needed, hence the loop: its last round re-evaluates every active row from
// slot 0 (they always fit)
int Solver::refresh_stale() {
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
        RCCHK(run_pass(stale, X, G, true, false, f, Z, fn, nullptr, 0, prec));
    }
    return GML_OK;
}

// Hessian blocks (int8 kernel over the limb planes of the rows' last passes, or the FP64 MFMA kernel over V): the working
// set of a Cholesky row; for a matrix-free row the tiles of its block-diagonal preconditioner (gml_solver.hip), inverted here.
// One upload carries the control block of the rows, one that of the tiles.
int Solver::direction_blocks(const std::vector<int> &cg_rows) {
    if (!cg_rows.empty() && !gml_is_i8(prec)) {
        // FP64 phase: the curvature weights of the matrix-free rows' Hessian-vector products and preconditioner tiles come from
        // an int8-limb objective pass at the same iterate (those run on the int8 cores either way; only the curvature is
        // approximate)
        std::vector<double> tf((size_t)R), tz((size_t)R, 1.0), tn((size_t)R);
        RCCHK(run_pass(cg_rows, X, nullptr, false, false, tf, tz, tn, nullptr, 0, gml_is_i8(o.precision) ? o.precision : GML_PREC_I8X));
    }
    // layout of the block: mt [R] | node [R] | msz of the Cholesky rows [R] | (unused) [R] | vslot [R] | hoff [R+1] |
    // s1 [Rp] | ynoise [Rp] | s1cg [Rp]
    const size_t ibytes = (sizeof(int) * 5 * R + 7) & ~(size_t)7, lbytes = sizeof(long long) * (R + 1);
    std::vector<char> blk(ibytes + lbytes + sizeof(double) * 3 * Rp, 0);
    int *mt2 = reinterpret_cast<int *>(blk.data()), *vs = mt2 + 4 * R;
    long long *hoff = reinterpret_cast<long long *>(blk.data() + ibytes);
    double *s1 = reinterpret_cast<double *>(blk.data() + ibytes + lbytes), *yn = s1 + Rp, *s1cg = s1 + 2 * Rp;
    hoff[0] = 0;
    for (int64_t r = 0; r < Rp; ++r) s1[r] = s1cg[r] = 1.0;
    for (int64_t r = 0; r < R; ++r) yn[r] = 100.0 * fn[r]; // a gradient difference below this is noise of the pass arithmetic, not curvature
    for (int64_t r = 0; r < R; ++r) {
        const int m = done[r] || iscg[r] ? 0 : msz[r];
        mt2[r] = (m + 31) / 32;
        mt2[R + r] = (int)(p->node0 + r);
        mt2[2 * R + r] = m; // the Cholesky step solves these
        vs[r] = vslot[r];
        hoff[r + 1] = hoff[r] + (long long)mt2[r] * 32 * mt2[r] * 32;
        const double zi = formulation == GML_LOGRISE ? 1.0 / Z[r] : 1.0; // Hess log Z = Hess Z / Z - g g^T
        s1[r] = hscale * zi; // sub-sampled blocks
        s1cg[r] = zi;        // the Hessian-vector products use every configuration
    }
    // the tiles: kTile consecutive entries of W each, behind the rows' blocks
    constexpr int T = kTile;
    std::vector<long long> t0((size_t)R, 0);
    ntiles = 0;
    for (int r : cg_rows) {
        t0[r] = ntiles;
        ntiles += (nW[r] + T - 1) / T;
    }
    tile_base = hoff[R];
    const int64_t NV = R + ntiles, htotal = std::max<long long>(tile_base + ntiles * (long long)T * T, 1);
    std::vector<int> mtV, vmV, wrowV, hflag;
    std::vector<long long> hoffV;
    if (ntiles > 0) {
        mtV.assign((size_t)NV, T / 32);
        hoffV.resize((size_t)NV);
        vmV.resize((size_t)ntiles);
        wrowV.resize((size_t)ntiles);
        hflag.assign((size_t)R, 0);
        const bool rows_i8 = gml_is_i8(prec); // (FP64 phase: the rows' blocks come from launch_hess_f64, not from this call)
        for (int64_t r = 0; r < R; ++r) {
            mtV[r] = rows_i8 ? mt2[r] : 0;
            hoffV[r] = hoff[r];
            hflag[r] = mtV[r] > 0;
        }
        for (int r : cg_rows) {
            const int64_t nt = (nW[r] + T - 1) / T;
            hflag[r] = 1;
            for (int64_t t = 0; t < nt; ++t) {
                vmV[t0[r] + t] = (int)std::min<int64_t>(T, nW[r] - t * T);
                wrowV[t0[r] + t] = r;
                hoffV[R + t0[r] + t] = tile_base + (t0[r] + t) * (long long)T * T;
            }
        }
    }
    if (htotal > dH_elems) {
        // regrown: the superseded block goes back now (a long solve of rows that keep admitting violators would otherwise hold
        // every earlier size until gml_learn returns), and the new size is checked against what the device has left
        A.release(dH);
        dH = nullptr;
        dH_elems = htotal + htotal / 4;
        size_t freeb = 0, totalb = 0;
        HIPCHK(dev_mem_info(&freeb, &totalb));
        if (8.0 * (double)dH_elems > 0.95 * (double)freeb) {
            if (8.0 * (double)htotal > 0.95 * (double)freeb)
                return fail(GML_ENOMEM, "Hessian blocks of %.1f GB do not fit in %.1f GB free HBM (lower max_working)", 8.0 * htotal / 1e9, freeb / 1e9);
            dH_elems = htotal;
        }
        HIPCHK(A.get(&dH, (size_t)dH_elems));
    }
    for (int64_t r = 0; r < R; ++r) {
        const bool needs_planes = (gml_is_i8(prec) && mt2[r] > 0) || (ntiles > 0 && iscg[r] && !done[r]);
        if (needs_planes && (vslot[r] < 0 || vslot[r] >= Scap || owner[vslot[r]] != r || vstale[r]))
            return fail(GML_EHIP, "internal: row %lld enters the Hessian without valid V planes (slot %d, owner %d, stale %d)", (long long)r,
                        vslot[r], vslot[r] >= 0 && vslot[r] < Scap ? owner[vslot[r]] : -2, (int)vstale[r]);
    }
    HIPCHK(stg.h2d(dHctl, blk.data(), blk.size()));
    dMt = reinterpret_cast<int *>(dHctl);
    dVslot = dMt + 4 * R;
    dHoff = reinterpret_cast<long long *>(dHctl + ibytes);
    dS1 = reinterpret_cast<double *>(dHctl + ibytes + lbytes);
    dYnoise = dS1 + Rp;
    dS1cg = dS1 + 2 * Rp;
    HessTiles tl;
    int *dMtV = dMt;
    long long *dHoffV = dHoff;
    if (ntiles > 0) {
        // control block of the tiles: hoffV [NV] | t0 [R] | mtV [NV] | vm [ntiles] | wrow [ntiles] | hflag [R]
        const size_t tb = sizeof(long long) * (NV + R) + sizeof(int) * (NV + 2 * ntiles + R);
        if (tb > dTctl_bytes) {
            A.release(dTctl);
            dTctl_bytes = tb + tb / 4;
            HIPCHK(A.get(&dTctl, dTctl_bytes));
        }
        if (ntiles > tile_cap) {
            A.release(dFV);
            A.release(dgV);
            tile_cap = ntiles + ntiles / 4;
            HIPCHK(A.get(&dFV, (size_t)tile_cap * T));
            HIPCHK(A.get(&dgV, (size_t)tile_cap * T));
        }
        std::vector<char> tblk(tb);
        long long *h_hoff = reinterpret_cast<long long *>(tblk.data()), *h_t0 = h_hoff + NV;
        int *h_mt = reinterpret_cast<int *>(h_t0 + R), *h_vm = h_mt + NV, *h_wrow = h_vm + ntiles, *h_flag = h_wrow + ntiles;
        std::memcpy(h_hoff, hoffV.data(), sizeof(long long) * NV);
        std::memcpy(h_t0, t0.data(), sizeof(long long) * R);
        std::memcpy(h_mt, mtV.data(), sizeof(int) * NV);
        std::memcpy(h_vm, vmV.data(), sizeof(int) * ntiles);
        std::memcpy(h_wrow, wrowV.data(), sizeof(int) * ntiles);
        std::memcpy(h_flag, hflag.data(), sizeof(int) * R);
        HIPCHK(stg.h2d(dTctl, tblk.data(), tb));
        dHoffV = reinterpret_cast<long long *>(dTctl);
        const long long *dT0 = dHoffV + NV;
        dT0m = dT0;
        dMtV = reinterpret_cast<int *>(dTctl + sizeof(long long) * (NV + R));
        dVm = dMtV + NV;
        dWrow = dVm + ntiles;
        RCCHK(upload_rows(cg_rows, dRows2));
        launch_cg_tiles(dRows2, (int)cg_rows.size(), X, PG, G, kind, Qp, T, dT0, dFV, dgV, st);
        tl.n = ntiles;
        tl.T = T;
        tl.F = dFV;
        tl.wrow = dWrow;
        tl.hflag = dWrow + ntiles;
    }
    // (the int8 kernel's finalisation writes every lower tile of the rows' blocks and the solve reads nothing else of them; the FP64
    // kernel accumulates in place, and the tiles' inverses read whole tiles)
    if (!gml_is_i8(prec) || ntiles > 0) HIPCHK(hipMemsetAsync(dH, 0, sizeof(double) * htotal, st));
    trace("hessian");
    if (gml_is_i8(prec) || ntiles > 0) {
        std::string err;
        const int hrc = i8_hessian(p->i8ws, d, dMt + R, dVslot, dFidx, dMtV, ntiles > 0 ? mtV.data() : mt2, dHoffV, htotal, (int)R, capP, formulation, Kh,
                                   kstride, dH, st, &err, ntiles > 0 ? &tl : nullptr);
        if (hrc) return fail(hrc, "%s", err.empty() ? "int8 Hessian: working set above 512 entries" : err.c_str());
    }
    if (!gml_is_i8(prec)) launch_hess_f64(d, p->dV, dMt + R, dFidx, dMt, dHoff, (int)R, capP, formulation, Kh, kstride, dH, st);
    if (ntiles > 0) {
        trace("tile inverses");
        launch_tile_inverse(T, dH, dHoffV + R, dVm, dWrow, dS1, formulation == GML_LOGRISE ? 1.0 : 0.0, dgV, ntiles, st);
    }
    HIPCHK(hipGetLastError());
    ++stats.hessian_passes;
    return GML_OK;
}

// batched Cholesky on the device; the directions are scattered into D
int Solver::newton_blocks(const std::vector<int> &chol_rows) {
    if (chol_rows.empty()) return GML_OK;
    const double s2 = formulation == GML_LOGRISE ? 1.0 : 0.0;
    trace("cholesky");
    int maxm = 1;
    for (int r : chol_rows) maxm = std::max(maxm, msz[r]);
    const int *rows_k = nullptr;
    RCCHK(ctl(chol_rows, dRows, &rows_k));
    // (a block over every configuration is the Hessian itself: nothing to correct -- and a secant over a finite step would spoil it)
    const bool secant = use_secant && Kh < d.Kp;
    if (secant)
        launch_secant(rows_k, (int)chol_rows.size(), dFidx, dMt + 2 * R, capP, X, Qp, dgF, dH, dHoff, dMt, dS1, s2, dYnoise, dFprev, dMprev, dXprev, dGprev,
                      dSec, dSec + 2 * Rp * capP, dNpairs, Rp * (int64_t)capP, kCholLds, st);
    if (secant && o.verbose >= 2) {
        std::vector<int> npv((size_t)Rp);
        HIPCHK(hipMemcpyAsync(npv.data(), dNpairs, sizeof(int) * Rp, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int c[3] = {0, 0, 0};
        for (int r : chol_rows) ++c[std::min(2, npv[r])];
        fprintf(stderr, "[gml]   secant pairs in use: none %d rows, one %d, two %d\n", c[0], c[1], c[2]);
    }
    // (the kernel also re-solves on the orthant face where the projection of the line search would clip the step: as for the
    // matrix-free rows, newton_cg_group, but inside the one launch)
    NewtonFaces nf;
    nf.F = dFidx;
    nf.X = X;
    nf.kind = kind;
    nf.Qp = Qp;
    nf.share = face_share;
    // (a re-solve costs as much as the solve; what it saves is backtracking passes, whose cost grows with configurations x
    // statistics: config 2 -- 1e5 x 256 -- is 0.4 ms faster without, the headline problem 4 % faster with)
    nf.rounds = (double)p->K * (double)Qp >= 268435456.0 ? face_rounds : 0;
    if (tune[GML_TUNE_FACE_ROUNDS] > 0) nf.rounds = (int)tune[GML_TUNE_FACE_ROUNDS] - 1;
    // (blocks of up to kCholLds entries take the secant correction inside the solve, on the matrix in LDS; k_secant above kept the
    // pairs and corrected the larger blocks itself)
    SecantPairs sp;
    if (secant) sp = SecantPairs{dSec, dSec + 2 * Rp * capP, dNpairs, Rp * (int64_t)capP};
    launch_newton_solve(dH, dHoff, dMt, dMt + 2 * R, dS1, s2, dgF, dpgF, (int)R, capP, dsol, dSdiag, st, maxm, &nf, nullptr, nullptr, secant ? &sp : nullptr);
    launch_scatter_dir(rows_k, (int)chol_rows.size(), dFidx, dsol, dMt + 2 * R, capP, Qp, D, st);
    HIPCHK(hipGetLastError());
    return GML_OK;
}

// Matrix-free Newton-CG: H_WW d = -pg_W by preconditioned conjugate gradients, Hessian-vector products from the device
// operator (forward GEMM of the direction, weights of the rows' last objective pass, backward GEMM), preconditioner =
// the inverted Hessian blocks of tiles of kTile neighbouring entries of W (direction_blocks).  Inexact Newton:
// the residual is reduced by eta = min(cg_eta, sqrt(kkt)), in at most max_cg steps.
int Solver::newton_cg(const std::vector<int> &cg_rows) {
    if (cg_rows.empty()) return GML_OK;
    // Two groups.  Rows that are still ADMITTING violators (coordinates at zero whose pseudo-gradient is not: the support is
    // not final, the iteration count of a dense optimum is set by how many of them are let in per iteration, not by the
    // accuracy of the step) and are not yet within max(1e3 tol, 1e-5) of their optimum take Hessian-vector products over a
    // sub-sample of the configurations (below).  Rows whose support is final take every configuration: with the approximate
    // curvature a row converges only linearly (the residual halves per iteration at 60 configurations per working-set
    // entry -- the weights exp(-E) are far from uniform), and the last digits come from a few exact Newton steps (measured on
    // config 5 at the default regulariser: sub-sampled throughout, a handful of the 512 rows creep at 5e-6 for a hundred
    // iterations).
    std::vector<int> coarse, fine;
    const double thr = std::max(1e3 * o.tol, 1e-5);
    for (int r : cg_rows) (o.hv_subsample != 1 && admitting(r) && kkt[r] > thr ? coarse : fine).push_back(r);
    if (o.verbose >= 2) fprintf(stderr, "[gml]   cg groups: %zu rows admitting (sub-sampled curvature), %zu rows with every configuration\n", coarse.size(), fine.size());
    RCCHK(newton_cg_group(coarse, true));
    return newton_cg_group(fine, false);
}

int Solver::newton_cg_group(const std::vector<int> &cg_rows, bool subsample) {
    if (cg_rows.empty()) return GML_OK;
    const double s2 = formulation == GML_LOGRISE ? 1.0 : 0.0;
    trace("pcg");
    if (!Rv) {
        HIPCHK(A.get(&Rv, nd));
        HIPCHK(A.get(&Pv, nd));
        HIPCHK(A.get(&Hp, nd));
        HIPCHK(A.get(&Zv, nd));
        HIPCHK(A.get(&Wm, nd));
        HIPCHK(A.get(&dFaces, (size_t)Rp));
        HIPCHK(A.get(&dHv, (size_t)(3 * Rp + Rp / 32 + 8)));
    }
    RCCHK(upload_rows(cg_rows, dRows2));
    std::vector<int> liveflag((size_t)Rp, 0);
    for (int r : cg_rows) liveflag[r] = 1;
    HIPCHK(stg.h2d(dLive, liveflag.data(), sizeof(int) * Rp));
    // the rows' lists of W (k_cg_tiles, direction_blocks): the per-step vector kernels walk them instead of the 131 k columns
    if (!dNw) HIPCHK(A.get(&dNw, (size_t)Rp));
    HIPCHK(stg.h2d(dNw, nW.data(), sizeof(int) * R));
    const WList wl{dFV, dT0m, dNw, kTile};
    launch_pcg_init(dRows2, (int)cg_rows.size(), X, PG, kind, Qp, D, Rv, Zv, Pv, Wm, dCg, st);
    launch_tile_apply(kTile, dH + tile_base, dFV, dVm, dWrow, dLive, ntiles, Qp, Rv, Zv, st);
    launch_pcg_dir(dRows2, (int)cg_rows.size(), Qp, Wm, Rv, Zv, Pv, 1, dCg, wl, st);
    // Sub-sampled curvature: the Hessian-vector products run over ~1/ksub of the configurations -- as many as keep 32 of
    // them per working-set entry (the sample covariance of |W| statistics from 32 |W| configurations has its eigenvalues
    // within (1 +- 0.18)^2 of the true ones: a Newton step that is only solved to a 5 % residual loses nothing to that) --
    // spread over the whole histogram, rescaled by the weight of the part.  The gradient is always exact, so the optimum does
    // not move; one split plan serves every step of this CG solve (one operator).
    int64_t kchunk = 0, kpart = 0;
    // the split plan and the scales of the products over 1/ksub of the configurations
    auto set_plan = [&](int ksub, size_t nrows) -> int {
        int nsplit = 0;
        i8_split_plan(d, (int)(gml_round_up((int64_t)nrows, 32) / 32), ksub, &kchunk, &kpart, &nsplit);
        double wsub = 0;
        for (int c = 0; c < nsplit; ++c)
            for (int64_t b = c * kchunk / 512; b < std::min((c * kchunk + kpart) / 512, d.Kp / 512); ++b) wsub += p->wblk[(size_t)b];
        std::vector<double> sc((size_t)Rp);
        if (kpart < kchunk && wsub > 0) {
            for (int64_t r = 0; r < Rp; ++r) sc[r] = (r < R && formulation == GML_LOGRISE ? 1.0 / Z[r] : 1.0) / wsub;
            if (o.verbose >= 2) fprintf(stderr, "[gml]   cg: Hessian-vector products over 1/%d of the configurations (weight %.4f)\n", ksub, wsub);
        } else {
            kpart = kchunk;
            for (int64_t r = 0; r < Rp; ++r) sc[r] = r < R && formulation == GML_LOGRISE ? 1.0 / Z[r] : 1.0;
        }
        HIPCHK(stg.h2d(dS1cg, sc.data(), sizeof(double) * Rp)); // (the sub-sample's weight replaces the full sum)
        return GML_OK;
    };
    int64_t maxW = 1;
    for (int r : cg_rows) maxW = std::max<int64_t>(maxW, -(int64_t)sel[r].m);
    {
        int ksub = o.hv_subsample > 0 ? o.hv_subsample : (int)std::min<int64_t>(8, std::max<int64_t>(1, p->K / (32 * maxW)));
        if (!subsample) ksub = 1;
        RCCHK(set_plan(ksub, cg_rows.size()));
    }
    // Relaxed products (inexact Krylov: Simoncini & Szyld 2003, van den Eshof & Sleijpen 2004).  The error a product may carry
    // without moving the attainable residual grows like 1 / |r_j|: the first steps of an exact-curvature solve need every
    // configuration, the later ones -- whose search directions only correct a residual that is already a fraction of the
    // right-hand side -- do not.  From step 2 on the products run over 1/2, from step 4 on over 1/8 of the configurations
    // (as long as 16 configurations per working-set entry remain).  Config 5 at the default regulariser: the same 43-44 Newton
    // iterations and 59-67 k products, 36.4 -> 26.2 s; the step where the cut starts matters (over 1/4 from step 2: 28.1 s and
    // 70 k products; from step 1: 31.3 s, 56 iterations).  hv_subsample = 1 keeps every product exact.
    std::vector<std::pair<int, int>> relax;
    if (!subsample && o.hv_subsample == 0)
        for (const std::pair<int, int> sk : {std::pair<int, int>{2, 2}, std::pair<int, int>{4, 8}}) {
            const int ks = (int)std::min<int64_t>(sk.second, p->K / (16 * maxW));
            if (ks > 1) relax.push_back({sk.first, ks});
        }
    // The GEMM pass costs the same for a tile of 32 rows whatever the number of live ones in it; few rows go entry by entry
    // instead (same integers, gml_hv_sparse.hip: their iterates do not depend on the path)
    const bool sparse_ok = hv_sparse && p->i8ws != nullptr && formulation != GML_RPLE && hv_lf == 2 && hv_lb == 2 && d.ko <= 2 && maxW <= 65536 &&
                           dT0m != nullptr;
    // Hout = (sum_k h_k x_k x_k^T) theta for the listed rows: an hv pass over slots [0, n) of the u-plane workspace
    auto hv_pass = [&](const std::vector<int> &rows, const double *theta, double *Hout) -> int {
        const int64_t n = (int64_t)rows.size(), np = gml_round_up(n, 32);
        int64_t sumW = 0;
        for (int r : rows) sumW += nW[r];
        const bool sparse = sparse_ok && (double)sumW < hv_sparse_ratio * (double)d.Qfp * (double)(np / 32) &&
                            i8_hv_sparse_bytes(d, (int)n, maxW) <= ((size_t)2 << 30);
        std::vector<int> ctl((size_t)(3 * np + np / 32 + 4), -1);
        for (int64_t a = 0; a < np; ++a) {
            ctl[a] = a < n ? rows[a] : 0;
            ctl[np + a] = a < n ? (int)(p->node0 + rows[a]) : -1;
            ctl[2 * np + a] = a < n ? vslot[rows[a]] : 0;
        }
        for (int64_t g = 0; g < np / 32; ++g) ctl[3 * np + g] = (int)g;
        HIPCHK(stg.h2d(dHv, ctl.data(), sizeof(int) * ctl.size()));
        if (sparse) {
            const size_t need = i8_hv_sparse_bytes(d, (int)n, maxW);
            if (need > hvs_bytes) {
                const double tr = gml_now_s();
                A.release(dHvsBuf);
                dHvsBuf = nullptr;
                hvs_bytes = std::max(need + need / 2, (size_t)16 << 20);
                char *b = nullptr;
                HIPCHK(A.get(&b, hvs_bytes));
                dHvsBuf = b;
                if (o.verbose) fprintf(stderr, "[gml]   scratch of the entry-by-entry products: %.0f MB (%.1f ms)\n", hvs_bytes / 1048576.0, 1e3 * (gml_now_s() - tr));
            }
            std::string err;
            const int rc = i8_hv_sparse(p->i8ws, d, (int)n, dHv, dHv + np, dHv + 2 * np, dT0m, dNw, dFV, kTile, maxW, theta, Hout, kchunk, kpart, dHvsBuf,
                                        st, &err);
            if (rc) return fail(rc, "%s", err.c_str());
            ++stats.hessian_passes;
            ++n_hv_sparse;
            ++g_hv_sparse_calls;
            stats.hv_evals += n;
            return GML_OK;
        }
        I8Pass a{};
        a.theta = theta;
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
        a.G = Hout;
        a.hv = hv_lb == 2 ? 2 : 1;
        a.lf = hv_lf;
        a.kchunk = kchunk;
        a.kpart = kpart;
        std::string err;
        const int rc = i8_pass(&p->i8ws, d, Scap, a, st, nullptr, &err);
        if (rc) return fail(rc, "%s", err.c_str());
        ++stats.hessian_passes;
        stats.hv_evals += n;
        return GML_OK;
    };
    auto set_live = [&](const std::vector<int> &rows) -> int {
        RCCHK(upload_rows(rows, dRows2));
        std::fill(liveflag.begin(), liveflag.end(), 0);
        for (int r : rows) liveflag[r] = 1;
        HIPCHK(stg.h2d(dLive, liveflag.data(), sizeof(int) * Rp));
        return GML_OK;
    };
    std::vector<CgState> cgs((size_t)Rp);
    std::vector<FaceOut> faces((size_t)Rp);
    std::vector<int> cur = cg_rows; // rows of this round
    for (int round = 0;; ++round) {
        std::vector<int> live = cur;
        for (int ci = 0; ci < maxcg && !live.empty(); ++ci) {
            if (round == 0)
                for (const auto &sk : relax)
                    if (ci == sk.first) RCCHK(set_plan(sk.second, live.size()));
            RCCHK(hv_pass(live, Pv, Hp));
            RCCHK(upload_rows(live, dRows2));
            launch_pcg_step(dRows2, (int)live.size(), G, Wm, Qp, dS1cg, s2, Hp, D, Rv, Pv, dCg, wl, st);
            HIPCHK(stg.d2h(cgs.data(), dCg, sizeof(CgState) * Rp));
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync());
            std::vector<int> nxt;
            for (int r : live) {
                // a row that is still admitting violators needs a step that is good at the scale of its largest violation, not more
                const double eta = admitting(r) ? eta_admit : std::min(cg_eta, std::sqrt(std::max(kkt[r], 1e-300)));
                if (cgs[r].pHp > 0 && cgs[r].rs > eta * eta * cgs[r].rs0) nxt.push_back(r);
            }
            if (o.verbose >= 2) {
                fprintf(stderr, "[gml]   cg %2d: %zu rows live", ci, nxt.size());
                for (int64_t r : {dbg_row, worst_row})
                    if (r < R && std::find(cur.begin(), cur.end(), (int)r) != cur.end())
                        fprintf(stderr, "  row %lld |r|/|r0| %.3e", (long long)r, std::sqrt(cgs[r].rs / std::max(cgs[r].rs0, 1e-300)));
                fprintf(stderr, "\n");
            }
            live.swap(nxt);
            if (live.empty() || ci + 1 == maxcg) break; // (at the cap the next direction would not be used)
            RCCHK(set_live(live));
            launch_tile_apply(kTile, dH + tile_base, dFV, dVm, dWrow, dLive, ntiles, Qp, Rv, Zv, st);
            launch_pcg_dir(dRows2, (int)live.size(), Qp, Wm, Rv, Zv, Pv, 0, dCg, wl, st);
        }
        if (round >= face_rounds) break;
        // orthant faces (gml_solver.hip, k_pcg_faces): the rows whose projected step would lose a noticeable part of its
        // predicted decrease to the clipping solve again without the clipped coordinates
        RCCHK(upload_rows(cur, dRows2));
        launch_pcg_faces(dRows2, (int)cur.size(), X, PG, kind, Qp, D, Wm, dFaces, st);
        HIPCHK(stg.d2h(faces.data(), dFaces, sizeof(FaceOut) * Rp));
        HIPCHK(hipGetLastError());
        HIPCHK(stg.sync());
        std::vector<int> again;
        int64_t nfixed = 0;
        for (int r : cur) {
            nfixed += faces[r].nfixed;
            if (faces[r].nfixed > 0 && faces[r].mass > face_share * faces[r].total) again.push_back(r);
        }
        if (o.verbose >= 2)
            fprintf(stderr, "[gml]   cg faces (round %d): %lld coordinates fixed at zero in %zu rows; %zu rows solve again\n", round, (long long)nfixed,
                    cur.size(), again.size());
        if (again.empty()) break;
        cur.swap(again);
        if (!relax.empty()) RCCHK(set_plan(1, cur.size())); // (the re-solve starts from the exact residual of the clipped step)
        RCCHK(hv_pass(cur, D, Hp));
        RCCHK(set_live(cur));
        launch_pcg_resid(dRows2, (int)cur.size(), PG, G, Qp, dS1cg, s2, Hp, D, Wm, Rv, dCg, st);
        launch_tile_apply(kTile, dH + tile_base, dFV, dVm, dWrow, dLive, ntiles, Qp, Rv, Zv, st);
        launch_pcg_dir(dRows2, (int)cur.size(), Qp, Wm, Rv, Zv, Pv, 1, dCg, wl, st);
    }
    return GML_OK;
}

// Projected backtracking line search.  Two acceptance regimes per row:
//  * the predicted decrease is well above the uncertainty of f  -> Armijo on F;
//  * otherwise ("noise regime": near the optimum, or a noisy int8-limb f) function values cannot certify the step; the trial
//    is then a full pass and is accepted iff the directional derivative of F at the trial point back towards x is >= 0 (up
//    to an overshoot allowance): F is convex, so such a trial cannot have increased F.
// The passes of this iteration use fresh slot ranges: the V planes of the previous ones are no longer needed.
int Solver::line_search() {
    slot_next = 0;
    for (int64_t r = 0; r < R; ++r) {
        need[r] = !done[r];
        alpha[r] = 1.0;
        accepted_fwd[r] = 0;
    }
    for (int ls = 0; ls < 30; ++ls) {
        std::vector<int> rows;
        for (int64_t r = 0; r < R; ++r)
            if (need[r]) rows.push_back((int)r);
        if (rows.empty()) break;
        const int *rows_k = nullptr;
        const double *alpha_k = nullptr;
        RCCHK(ctl(rows, dRows, &rows_k));
        RCCHK(ctl(alpha, dAlpha, &alpha_k));
        trace("trial");
        launch_trial(rows_k, (int)rows.size(), X, D, PG, kind, Qp, lambda, alpha_k, Xt, kTrial, dStepn, st);
        // The first trial of an iteration is a full pass whatever the trial's scalars say, and the one thing the pass needs of them --
        // |trial - x|_1, which scales the unit of its weights -- is applied on the device (k_quant_theta): the host does not wait here,
        // the scalars come down with the results of the pass.  Later trials choose their kind of pass from the scalars.
        const bool early = ls > 0;
        auto read_trial = [&]() {
            for (int r : rows) {
                dd[r] = trial[r].dd;
                stepn[r] = trial[r].stepn; // ||trial - x||_1: bounds the change of every energy
                l1t[r] = trial[r].l1t;
                nreg[r] = !(-0.1 * dd[r] > 8.0 * fn[r]); // the step's expected decrease (~|dd|/2) vs the uncertainty of f
            }
        };
        bool anynoise = false;
        if (early) {
            RCCHK(fetch(trial, dTrial, sizeof(TrialOut) * Rp));
            HIPCHK(hipGetLastError());
            HIPCHK(stg.sync());
            read_trial();
            for (int r : rows) anynoise |= nreg[r] != 0;
        }
        const bool full = (ls == 0) || anynoise;
        // a full pass is followed on the stream by the directional derivatives back towards x: both come down in one wait
        const std::function<int()> back = [&]() -> int {
            const int *rk = nullptr; // (listed again: a re-run of some rows repeats this after a wait, which recycles the arena)
            RCCHK(ctl(rows, dRows, &rk));
            launch_back(rk, (int)rows.size(), X, Xt, Gt, kind, Qp, lambda, kTrial, st);
            RCCHK(fetch(trial, dTrial, sizeof(TrialOut) * Rp));
            return GML_OK;
        };
        RCCHK(run_pass(rows, Xt, Gt, full, true, ft, Zt, fnt, nullptr, 0, prec, full ? &back : nullptr, !early));
        if (!early) read_trial();
        std::vector<int> acc;
        for (int r : rows) {
            bool ok;
            if (nreg[r]) {
                const double bk = trial[r].back;
                ok = std::isfinite(ft[r]) && std::isfinite(bk) && bk >= -0.5 * std::fabs(dd[r]);
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
                if (!gml_is_i8(prec)) {
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
        if (o.verbose >= 2) {
            int nn = 0;
            for (int r : rows) nn += nreg[r] != 0;
            fprintf(stderr, "[gml]   ls %2d: %zu rows (%d in the noise regime), %zu accepted, %s pass\n", ls, rows.size(), nn, acc.size(), full ? "full" : "forward");
        }
        const int *acc_k = nullptr;
        RCCHK(ctl(acc, dRows2, &acc_k));
        launch_copy_rows(acc_k, (int)acc.size(), Qp, Xt, X, full ? Gt : nullptr, G, st);
        HIPCHK(hipGetLastError());
    }
    // rows accepted on an objective-only trial still need their gradient (and V)
    std::vector<int> rows;
    for (int64_t r = 0; r < R; ++r)
        if (accepted_fwd[r]) rows.push_back((int)r);
    if (!rows.empty()) RCCHK(run_pass(rows, X, G, true, false, f, Z, fn, nullptr, 0, prec));
    // rows whose line search failed entirely stay where they are; the stall counter ends them
    return GML_OK;
}

// results in the reference layout
int Solver::finish(double *out, double *kkt_out, int iterations) {
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
    // best iterate per row -> Xt (free now)
    RCCHK(upload_rows(frombest, dRows));
    launch_copy_rows(dRows, (int)frombest.size(), Qp, Xb, Xt, nullptr, nullptr, st);
    RCCHK(upload_rows(fromx, dRows2));
    launch_copy_rows(dRows2, (int)fromx.size(), Qp, X, Xt, nullptr, nullptr, st);
    // reference layout on the device (the direction array D is free now), then ONE copy: to the caller's host matrix, or --
    // `out` a device pointer (gml_multi_learn's dev_out blocks, device-side callers) -- device to device
    double *dres = D; // [R][P], P <= Qp
    RCCHK(ref_cols());
    const int32_t *dcols = dColsRef;
    launch_rows_to_reference(Xt, R, Qp, P, p->node0, d.cconst, dcols, dres, st);
    HIPCHK(hipGetLastError());
    hipPointerAttribute_t attr;
    bool dev_out = false;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess) dev_out = (attr.type == hipMemoryTypeDevice);
    else (void)hipGetLastError();
    HIPCHK(hipMemcpyAsync(out, dres, sizeof(double) * R * P, dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    HIPCHK(stg.sync());
    stats.iterations = iterations;
    stats.max_kkt = maxk;
    stats.not_converged = notconv;
    stats.t_hess = dir_time.seconds();
    // (the host does not wait at the end of the direction phase: what of it was still running when the first trial pass was queued
    // shows up in that pass's host-side wait)
    stats.t_pass = std::max(0.0, stats.t_pass - std::max(0.0, stats.t_hess - t_dir_host));
    if (notconv)
        return fail(GML_ENOTCONV, "%d of %lld nodes did not reach the KKT tolerance %.1e (worst %.3e)", notconv, (long long)R, o.tol, maxk);
    return GML_OK;
}

// multi-body key order (:94-104): the column of every parameter slot of every local row, from the host, once per solve
int Solver::ref_cols() {
    if (p->order == 2 || dColsRef) return GML_OK;
    std::vector<int32_t> cols((size_t)R * P);
    gml_parallel_for(R, [&](int64_t r) {
        NodeLayout L;
        gml_build_layout(p, p->node0 + r, L);
        std::memcpy(cols.data() + (size_t)r * P, L.cols.data(), sizeof(int32_t) * P);
    });
    HIPCHK(A.get(&dColsRef, (size_t)R * P));
    HIPCHK(hipMemcpyAsync(dColsRef, cols.data(), sizeof(int32_t) * R * P, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st)); // cols is a local
    return GML_OK;
}

// warm start: X <- the caller's rows (reference layout, host or device pointer)
int Solver::load_x0() {
    RCCHK(ref_cols());
    hipPointerAttribute_t attr;
    bool on_dev = false;
    if (hipPointerGetAttributes(&attr, x0) == hipSuccess) on_dev = (attr.type == hipMemoryTypeDevice);
    else (void)hipGetLastError();
    const double *src = x0;
    if (!on_dev) {
        double *tmp = nullptr;
        HIPCHK(A.get(&tmp, (size_t)R * P));
        HIPCHK(hipMemcpyAsync(tmp, x0, sizeof(double) * R * P, hipMemcpyHostToDevice, st));
        src = tmp;
    }
    int *dbad = nullptr;
    HIPCHK(A.get(&dbad, 4));
    HIPCHK(hipMemsetAsync(dbad, 0, sizeof(int) * 4, st));
    launch_ref_to_internal(src, P, R, P, Qp, dNode, d.cconst, dColsRef, X, dbad, st);
    HIPCHK(hipGetLastError());
    int bad[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(bad, dbad, sizeof bad, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (bad[0]) return fail(GML_EINVAL, "the starting point contains a non-finite value");
    return GML_OK;
}

int Solver::iterate(double *out, double *kkt_out) {
    // first pass at X = 0: every energy is 0, the forward kernel skips its sweeps over the columns (the same bits without the GEMM:
    // 3.4 -> 1 ms of the headline solve's first pass); a warm start (gml_learn_warm) begins at the caller's rows instead
    std::vector<int> rows_all((size_t)R);
    for (int64_t r = 0; r < R; ++r) rows_all[r] = (int)r;
    if (x0) RCCHK(load_x0());
    at_zero = x0 == nullptr;
    const int rc0 = run_pass(rows_all, X, G, true, false, f, Z, fn, nullptr, 0, prec);
    at_zero = false;
    RCCHK(rc0);
    int it = 0;
    for (it = 0; it < o.max_iter; ++it) {
        int64_t nactive = 0;
        const std::vector<uint8_t> done_before = coarse_on ? done : std::vector<uint8_t>();
        RCCHK(select(it, &nactive));
        if (coarse_on) {
            // A gradient of the coarse form (30-bit theta, 23-bit weights) must not certify anything: as soon as a row comes within
            // coarse_thr of its optimum -- or is declared converged -- by such a gradient, the coarse phase ends for good, those rows
            // are evaluated again at full width where they stand, their best-iterate records start over, and the selection is redone.
            const double thr = std::max(coarse_thr, 100.0 * o.tol);
            std::vector<int> low;
            bool dense = false; // a row has gone matrix-free: a dense optimum, whose many more iterations gain nothing from cheaper early
                                // passes (config 5 at the default regulariser: 52 iterations instead of 39 for the same wall-clock)
            for (int64_t r = 0; r < R; ++r) dense |= !done[r] && iscg[r];
            // (also the rows this selection ended at their stall count, above the threshold: declared done on coarse gradients)
            for (int64_t r = 0; r < R; ++r)
                if (!done_before[r] && (done[r] || !(kkt[r] > thr) || dense)) low.push_back((int)r);
            if (!low.empty()) {
                coarse_on = false;
                if (o.verbose)
                    fprintf(stderr, "[gml] it %3d: %s: the passes switch from the coarse form to full width (%zu rows re-evaluated)\n", it,
                            dense ? "rows have gone matrix-free (dense optimum)" : "rows have come within the coarse threshold of their optimum", low.size());
                const double inf = INFINITY;
                for (int r : low) {
                    done[r] = 0;
                    atfloor[r] = 0;
                    stall[r] = 0;
                    best[r] = INFINITY;
                    Fbest[r] = INFINITY;
                    HIPCHK(stg.h2d(dBest + r, &inf, sizeof(double)));
                }
                RCCHK(run_pass(low, X, G, true, false, f, Z, fn, nullptr, 0, prec));
                RCCHK(select(it, &nactive));
            }
        }
        if (nactive == 0) {
            bool polishing = false;
            RCCHK(start_polish(&polishing));
            if (!polishing) break;
            continue;
        }
        {
            int maxm = 32;
            for (int64_t r = 0; r < R; ++r)
                if (!done[r]) maxm = std::max(maxm, msz[r]);
            set_kh(nactive, maxm);
        }
        trace("refresh");
        RCCHK(refresh_stale());
        trace("refreshed");
        std::vector<int> chol_rows, cg_rows;
        for (int64_t r = 0; r < R; ++r)
            if (!done[r]) (iscg[r] ? cg_rows : chol_rows).push_back((int)r);
        dir_time.mark();
        const double t_dir0 = gml_now_s();
        RCCHK(direction_blocks(cg_rows));
        RCCHK(newton_blocks(chol_rows));
        RCCHK(newton_cg(cg_rows));
        trace("directions done");
        dir_time.mark();
        t_dir_host += gml_now_s() - t_dir0;
        RCCHK(line_search());
    }
    return finish(out, kkt_out, it);
}

} // namespace

extern "C" int gml_learn(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts_in, double *out,
                         double *kkt_out, gml_stats *stats_out) {
    return gml_learn_warm(p, formulation, regularizer_c, opts_in, nullptr, out, kkt_out, stats_out);
}

extern "C" int gml_learn_warm(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts_in, const double *x0, double *out,
                              double *kkt_out, gml_stats *stats_out) {
    if (!p || !out) return fail(GML_EINVAL, "NULL argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (formulation != GML_RISE && p->order != 2)
        return fail(GML_EUNSUPPORTED, "multi-body statistics are defined for RISE only (multiRISE, :83-152)");
    if (!(regularizer_c >= 0)) return fail(GML_EINVAL, "regularizer must be >= 0");
    gml_opts o;
    if (opts_in) o = *opts_in;
    else gml_default_opts(&o);
    bool asked_auto = false;
    {
        // (auto: gml_internal.h -- the 38/31-bit limbs, the FP64-grade ones for tight tolerances and for small problems)
        const int asked = o.precision;
        asked_auto = asked == GML_PREC_AUTO;
        o.precision = gml_resolve_precision(p, asked, o.tol > 0 ? o.tol : 1e-9);
        if (o.precision < 0) return fail(GML_EINVAL, "unknown precision %d", asked);
    }
    HIPCHK(hipSetDevice(p->device));
    const double t_start = gml_now_s();
    int rc;
    bool underflow = false;
    {
        Solver s(p, formulation, o, gml_lambda(regularizer_c, p->n, p->M));
        s.x0 = x0;
        rc = s.init();
        if (rc == GML_OK) rc = s.iterate(out, kkt_out);
        s.stats.t_total = gml_now_s() - t_start;
        s.stats.t_pack = p->t_ingest[3];
        // everything that is neither a pass nor the direction phase: selection, trial points, bookkeeping
        s.stats.t_host = std::max(0.0, s.stats.t_total - s.stats.t_pass - s.stats.t_hess);
        if (stats_out && (rc == GML_OK || rc == GML_ENOTCONV)) *stats_out = s.stats;
        underflow = s.underflow;
    }
    if (rc == GML_EUNSUPPORTED && underflow && asked_auto) {
        // `auto` is the reference's Float64 solve (:164-181) by other means: a histogram whose optimum lies where exp(-E) spreads over
        // hundreds of units (c = 0 on near-separable data, |theta|_1 in the hundreds) cannot be held by the int8 limbs at an iterate;
        // the reference returns a result there, so `auto` runs the solve again on the FP64-MFMA path.  A caller who named an
        // int8-limb precision keeps the error.
        const std::string first = gml_last_error();
        gml_opts o64 = o;
        o64.precision = GML_PREC_F64;
        Solver s(p, formulation, o64, gml_lambda(regularizer_c, p->n, p->M));
        s.x0 = x0;
        rc = s.init();
        if (rc == GML_OK) rc = s.iterate(out, kkt_out);
        if (rc != GML_OK && rc != GML_ENOTCONV)
            return fail(rc, "%s; the FP64 path, which precision auto falls back to, then failed: %s", first.c_str(), std::string(gml_last_error()).c_str());
        s.stats.t_total = gml_now_s() - t_start;
        s.stats.t_pack = p->t_ingest[3];
        s.stats.t_host = std::max(0.0, s.stats.t_total - s.stats.t_pass - s.stats.t_hess);
        s.stats.polished = 1; // (finished on the FP64 path)
        if (stats_out) *stats_out = s.stats;
    } else if (rc == GML_ENOTCONV && asked_auto && gml_is_i8(o.precision) && (double)p->K * (double)p->P * (double)p->n <= 268435456.0) {
        // The other way an int8-limb solve can differ from the Float64 solve `auto` stands for: no row leaves the fixed-point range,
        // but the iterates wander along a nearly flat direction on the noise of the limbs and never certify (RPLE at c = 0 on
        // near-separable data: profiles/r6_robust_sweep.txt; the FP64 path converges there in 22 iterations).  On SMALL problems --
        // every kernel launch-bound, the FP64 solve a few milliseconds -- `auto` therefore tries the FP64 path before it reports
        // "not converged", and keeps whichever solve ended with fewer unconverged rows (then the smaller residual).  Host `out` only.
        hipPointerAttribute_t attr;
        bool dev_out = false;
        if (hipPointerGetAttributes(&attr, out) == hipSuccess) dev_out = (attr.type == hipMemoryTypeDevice);
        else (void)hipGetLastError();
        if (!dev_out) {
            const std::string first = gml_last_error();
            const size_t R = (size_t)(p->node1 - p->node0);
            std::vector<double> out1(out, out + R * (size_t)p->P), kkt1;
            if (kkt_out) kkt1.assign(kkt_out, kkt_out + R);
            gml_stats st1{};
            if (stats_out) st1 = *stats_out;
            gml_opts o64 = o;
            o64.precision = GML_PREC_F64;
            Solver s(p, formulation, o64, gml_lambda(regularizer_c, p->n, p->M));
            s.x0 = x0;
            int rc2 = s.init();
            if (rc2 == GML_OK) rc2 = s.iterate(out, kkt_out);
            const bool usable = rc2 == GML_OK || rc2 == GML_ENOTCONV;
            const bool better = usable && (!stats_out || s.stats.not_converged < st1.not_converged ||
                                           (s.stats.not_converged == st1.not_converged && s.stats.max_kkt < st1.max_kkt));
            if (better) {
                rc = rc2;
                s.stats.t_total = gml_now_s() - t_start;
                s.stats.t_pack = p->t_ingest[3];
                s.stats.t_host = std::max(0.0, s.stats.t_total - s.stats.t_pass - s.stats.t_hess);
                s.stats.polished = 1;
                if (stats_out) *stats_out = s.stats;
            } else { // the first solve stands
                std::copy(out1.begin(), out1.end(), out);
                if (kkt_out) std::copy(kkt1.begin(), kkt1.end(), kkt_out);
                if (stats_out) {
                    *stats_out = st1;
                    stats_out->t_total = gml_now_s() - t_start;
                }
                (void)fail(GML_ENOTCONV, "%s", first.c_str());
            }
        }
    }
    return rc;
}
