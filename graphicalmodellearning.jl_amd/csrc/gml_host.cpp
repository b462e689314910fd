// libgml_hip: C ABI (include/gml.h) + host side of the MI355X learn() hot path.
//
// What lives here: histogram validation/packing, the per-node parameter layout of the
// reference (pairwise :162, multi-body :94-104), the batched working-set Newton solver that
// replaces the reference's per-node Ipopt solve (:164-181), and the orchestration of the
// device passes.  All arithmetic over the K configurations happens in HIP kernels
// (gml_kernels_f64.hip, gml_kernels_i8.hip); there is no CPU fallback for it.
#include "../../include/gml.h"
#include "gml_dev.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace gml;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

extern "C" const char *gml_last_error(void) { return g_err.c_str(); }

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// persistent worker pool for the host-side per-node loops (thread creation per call would cost
// more than most of these loops)
namespace {
class Pool {
  public:
    Pool() {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt == 0) nt = 1;
        if (nt > 16) nt = 16;
        nworkers_ = nt > 1 ? nt - 1 : 0;
        for (unsigned t = 0; t < nworkers_; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void run(int64_t n, const std::function<void(int64_t)> &fn) {
        if (n <= 0) return;
        if (nworkers_ == 0 || n == 1) {
            for (int64_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::lock_guard<std::mutex> serial(run_m_); // one job at a time
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            pending_ = nworkers_;
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            const int64_t i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    unsigned nworkers_ = 0;
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t n_ = 0;
    std::atomic<int64_t> next_{0};
    unsigned pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};
Pool &pool() {
    static Pool *p = new Pool(); // intentionally leaked: no destructor races at process exit
    return *p;
}
} // namespace

static void parallel_for(int64_t n, const std::function<void(int64_t)> &fn) { pool().run(n, fn); }

static int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------
// problem handle
// ------------------------------------------------------------------------------------------
struct gml_problem {
    int device = 0;
    hipStream_t st = nullptr;
    int64_t n = 0, K = 0, P = 0, node0 = 0, node1 = 0;
    int order = 2;
    double M = 0;
    DevProblem d{};
    std::vector<int32_t> gkeys; // [Q][ko] subsets of spins (feature keys), -1 padded
    int ko = 1;
    std::vector<int64_t> qoff; // qoff[q] = first column of the size-q subsets
    std::vector<double> wblk;    // wblk[j] = sum of w over the configurations [512 j, 512 j + 512)
    // workspace (sized for ws_rows rows)
    int64_t ws_rows = 0;
    double *dTheta = nullptr, *dV = nullptr, *dG = nullptr, *dF = nullptr;
    int *hCtl = nullptr; // pinned twin of dRowcol | dGroups
    int *dRowcol = nullptr, *dGroups = nullptr;
    double *hTh = nullptr, *hG = nullptr, *hF = nullptr; // pinned staging (ws_rows x Qp, ws_rows)
    // hessian workspace
    int64_t hs_rows = 0, hs_cap = 0, hs_elems = 0;
    int *dFidx = nullptr, *dMt = nullptr;
    long long *dHoff = nullptr;
    double *dH = nullptr, *dVec = nullptr;
    // i8 path workspace lives in gml_i8 (allocated lazily)
    void *i8ws = nullptr;
};

static int64_t binom(int64_t n, int64_t k) {
    if (k < 0 || k > n) return 0;
    int64_t r = 1;
    for (int64_t i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

extern "C" double gml_lambda(double c, int64_t n, double M) {
    // lambda = regularizer*sqrt(log((num_spins^2)/0.05)/num_samples)   (:157)
    return c * std::sqrt(std::log(((double)n * (double)n) / 0.05) / M);
}

extern "C" void gml_default_opts(gml_opts *o) {
    std::memset(o, 0, sizeof *o);
    o->tol = 1e-9;
    o->max_iter = 100;
    o->precision = GML_PREC_I8X; // the fast path; rows it leaves above tol are finished in FP64 (polish = 0)
    o->max_working = 512;
    o->max_add = 64;
    o->verbose = 0;
}

// next q-subset of {0..n-1} in lexicographic order; returns false after the last one
static bool next_comb(std::vector<int> &idx, int64_t n) {
    const int q = (int)idx.size();
    int t = q - 1;
    while (t >= 0 && idx[t] == (int)n - q + t) --t;
    if (t < 0) return false;
    ++idx[t];
    for (int s = t + 1; s < q; ++s) idx[s] = idx[s - 1] + 1;
    return true;
}

// Parameter j of node u (reference order, :94-104: (u), then (u,S) with S the ascending
// subsets of the other spins, by size then lexicographically) -> internal column.
static void node_cols(const gml_problem *p, int64_t u, std::vector<int32_t> &cols) {
    cols.clear();
    cols.reserve((size_t)p->P);
    cols.push_back((int32_t)p->d.cconst); // (u,) : the field, statistic s_u * 1
    const int fo = p->order - 1;
    if (fo >= 1) {
        for (int64_t i = 0; i < p->n; ++i)
            if (i != u) cols.push_back((int32_t)i);
    }
    for (int q = 2; q <= fo; ++q) {
        if (q > p->n) break;
        std::vector<int> idx(q);
        for (int t = 0; t < q; ++t) idx[t] = t;
        int64_t c = p->qoff[q];
        do {
            bool has = false;
            for (int t = 0; t < q; ++t) has |= (idx[t] == (int)u);
            if (!has) cols.push_back((int32_t)c);
            ++c;
        } while (next_comb(idx, p->n));
    }
}

// Host (pageable) -> device copy of a large buffer through two pinned staging buffers filled by the thread
// pool: a plain hipMemcpy from pageable memory runs at 4-5 GB/s, this at the speed of the parallel memcpy.
static int upload_pageable(void *dst, const void *src, size_t bytes, hipStream_t st) {
    constexpr size_t CH = (size_t)64 << 20;
    if (bytes < 2 * CH) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return GML_OK;
    }
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = GML_OK;
    for (int i = 0; i < 2 && rc == GML_OK; ++i) {
        if (hipHostMalloc(&stage[i], CH) != hipSuccess || hipEventCreate(&done[i]) != hipSuccess) rc = GML_ENOMEM;
    }
    if (rc == GML_OK) {
        int b = 0;
        for (size_t off = 0; off < bytes && rc == GML_OK; off += CH, b ^= 1) {
            const size_t len = std::min(CH, bytes - off);
            if (off >= 2 * CH && hipEventSynchronize(done[b]) != hipSuccess) rc = GML_EHIP; // its previous copy has left the buffer
            const char *sp = static_cast<const char *>(src) + off;
            char *dp = static_cast<char *>(stage[b]);
            const int64_t parts = (int64_t)((len + ((size_t)4 << 20) - 1) / ((size_t)4 << 20));
            parallel_for(parts, [&](int64_t q) {
                const size_t o = (size_t)q << 22, l = std::min((size_t)4 << 20, len - o);
                std::memcpy(dp + o, sp + o, l);
            });
            if (hipMemcpyAsync(static_cast<char *>(dst) + off, stage[b], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(done[b], st) != hipSuccess)
                rc = GML_EHIP;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = GML_EHIP;
    }
    for (int i = 0; i < 2; ++i) {
        if (stage[i]) (void)hipHostFree(stage[i]);
        if (done[i]) (void)hipEventDestroy(done[i]);
    }
    if (rc == GML_ENOMEM) { // no pinned memory: fall back to the plain copy
        (void)hipGetLastError();
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return GML_OK;
    }
    if (rc) return fail(rc, "staged upload failed");
    return GML_OK;
}

// Builds the device-resident problem from the +-1 configurations, given as host bytes `spins` (K x n row-major), or
// as device bytes `dbytes` -- sample-major [K][n] or, with dev_spin_major, spin-major [n][ld] -- which this function
// owns and frees on every path.  What stays resident is one bit per entry: the sign bits of the spins (Sb) and the two
// MFMA operand images derived from them (Xb, Xtb): 2/8 byte per (configuration, statistic) + 1/8 per (configuration, spin).
static int alloc_dev(gml_problem *p, const double *counts, const int8_t *spins, int8_t *dbytes = nullptr,
                     bool dev_spin_major = false, int64_t ld = 0) {
    struct Guard { // temporaries of this function: released on every return
        int8_t *bytes = nullptr;
        long long *bad = nullptr;
        ~Guard() {
            if (bytes) (void)hipFree(bytes);
            if (bad) (void)hipFree(bad);
        }
    } tmp;
    tmp.bytes = dbytes;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamCreate(&p->st));
    DevProblem &d = p->d;
    // Statistic columns: the non-empty subsets of spins up to size order-1 (singles, then pairs (i<j) lexicographic,
    // ...), zero padding up to Qfp, then the constant column `cconst` (the empty subset: the node's field) opening a
    // final 64-byte block.  Order 1 (fields only, :94-104 with interaction_order = 1) keeps the single-spin columns
    // too -- the FP64 path reads the nodes' signs from them -- but gives them no parameter (node_cols).
    const int fo = std::max(p->order - 1, 1);
    p->ko = fo;
    p->qoff.assign(fo + 2, 0);
    int64_t Qf = 0;
    for (int q = 1; q <= fo; ++q) {
        p->qoff[q] = Qf;
        Qf += binom(p->n, q);
    }
    p->qoff[fo + 1] = Qf;
    p->P = 0;
    for (int q = 0; q <= p->order - 1; ++q) p->P += binom(p->n - 1, q);
    d.K = p->K;
    d.n = p->n;
    d.Qf = Qf;
    d.Qfp = round_up(std::max<int64_t>(Qf, 1), 64);
    d.cconst = d.Qfp;
    d.Qp = d.Qfp + 64;
    d.Kp = round_up(p->K, 1024);
    d.ko = p->ko;
    const int64_t Q = Qf;
    if (d.Qfp / 64 > 32000) return fail(GML_EUNSUPPORTED, "more than 2^21 statistics per node");
    {
        size_t freeb = 0, totalb = 0;
        HIPCHK(hipMemGetInfo(&freeb, &totalb));
        const double need = 2.0 * (double)d.Kp * (double)round_up(d.Qfp, 256) / 8.0 + (double)d.Kp * (double)p->n / 8.0 + 8.0 * (double)d.Kp;
        if (need > 0.92 * (double)freeb)
            return fail(GML_ENOMEM, "the bit images of the %lld x %lld design matrix (%.1f GB) do not fit in %.1f GB of free HBM",
                        (long long)d.Kp, (long long)d.Qfp, need / 1e9, freeb / 1e9);
    }
    // feature keys
    p->gkeys.assign((size_t)std::max<int64_t>(Q, 1) * p->ko, -1);
    {
        int64_t c = 0;
        for (int q = 1; q <= fo && q <= p->n; ++q) {
            std::vector<int> idx(q);
            for (int t = 0; t < q; ++t) idx[t] = t;
            do {
                for (int t = 0; t < q; ++t) p->gkeys[(size_t)c * p->ko + t] = idx[t];
                ++c;
            } while (next_comb(idx, p->n));
        }
    }
    d.Xs = d.Xt = nullptr; // FP64 path only, built on first use (ensure_f64)
    HIPCHK(hipMalloc(&d.Sb, (size_t)p->n * (d.Kp / 8)));
    HIPCHK(hipMalloc(&d.keys, sizeof(int32_t) * p->gkeys.size()));
    HIPCHK(hipMalloc(&d.Xb, (size_t)d.Kp * (d.Qfp / 8)));
    HIPCHK(hipMalloc(&d.Xtb, (size_t)xtb_bytes(d)));
    HIPCHK(hipMalloc(&d.w, sizeof(double) * d.Kp));
    HIPCHK(hipMemsetAsync(d.Sb, 0, (size_t)p->n * (d.Kp / 8), p->st));
    HIPCHK(hipMemsetAsync(d.w, 0, sizeof(double) * d.Kp, p->st));
    // weights w_k = counts[k]/M  (:170)
    std::vector<double> w((size_t)p->K);
    d.wmax = 0;
    for (int64_t k = 0; k < p->K; ++k) {
        w[k] = (counts ? counts[k] : 1.0) / p->M;
        d.wmax = std::max(d.wmax, w[k]);
    }
    d.wuni = w[0];
    for (int64_t k = 1; k < p->K; ++k)
        if (w[k] != w[0]) {
            d.wuni = 0.0;
            break;
        }
    p->wblk.assign((size_t)(d.Kp / 512), 0.0); // weight of every block of 512 configurations (sub-sampled Hessians)
    for (int64_t k = 0; k < p->K; ++k) p->wblk[(size_t)(k >> 9)] += w[k];
    HIPCHK(hipMemcpyAsync(d.w, w.data(), sizeof(double) * p->K, hipMemcpyHostToDevice, p->st));
    HIPCHK(hipMemcpyAsync(d.keys, p->gkeys.data(), sizeof(int32_t) * p->gkeys.size(), hipMemcpyHostToDevice, p->st));
    if (!tmp.bytes) {
        HIPCHK(hipMalloc(&tmp.bytes, (size_t)p->K * p->n));
        int urc = upload_pageable(tmp.bytes, spins, (size_t)p->K * p->n, p->st);
        if (urc) return urc;
        // validation of the +-1 alphabet (the reference validates nothing): first offending configuration
        long long hbad = -1;
        HIPCHK(hipMalloc(&tmp.bad, sizeof(long long)));
        HIPCHK(hipMemcpyAsync(tmp.bad, &hbad, sizeof(long long), hipMemcpyHostToDevice, p->st));
        launch_check_pm1(tmp.bytes, p->K, p->n, tmp.bad, p->st);
        HIPCHK(hipMemcpyAsync(&hbad, tmp.bad, sizeof(long long), hipMemcpyDeviceToHost, p->st));
        HIPCHK(hipStreamSynchronize(p->st));
        if (hbad >= 0) return fail(GML_EINVAL, "configuration %lld holds a spin that is not +-1", hbad);
        dev_spin_major = false;
    }
    launch_spin_bits(tmp.bytes, dev_spin_major, p->K, p->n, ld, d.Kp, d.Sb, p->st);
    launch_pack_bits(d, p->st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(p->st));
    return GML_OK;
}

static int create_common_impl(const double *counts, const int8_t *spins, int64_t K, int64_t n, int order, int64_t node0,
                              int64_t node1, int device, gml_problem **out, int8_t *dspins, bool dev_spin_major, int64_t ld,
                              bool *taken);

// dspins: the configurations as validated +-1 bytes on the device (sample-major K x n, or spin-major n x ld with
// dev_spin_major); owned from here on, whatever the outcome
static int create_common(const double *counts, const int8_t *spins, int64_t K, int64_t n, int order,
                         int64_t node0, int64_t node1, int device, gml_problem **out,
                         int8_t *dspins = nullptr, bool dev_spin_major = false, int64_t ld = 0) {
    bool taken = false;
    const int rc = create_common_impl(counts, spins, K, n, order, node0, node1, device, out, dspins, dev_spin_major, ld, &taken);
    if (!taken && dspins) (void)hipFree(dspins);
    return rc;
}

static int create_common_impl(const double *counts, const int8_t *spins, int64_t K, int64_t n, int order, int64_t node0,
                              int64_t node1, int device, gml_problem **out, int8_t *dspins, bool dev_spin_major, int64_t ld,
                              bool *taken) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    if (K <= 0 || n <= 0) return fail(GML_EINVAL, "empty histogram (K=%lld, n=%lld)", (long long)K, (long long)n);
    if (order < 1 || order > 8) return fail(GML_EINVAL, "interaction order %d out of range [1,8]", order);
    if (node0 < 0 || node1 > n || node0 >= node1)
        return fail(GML_EINVAL, "bad node range [%lld,%lld) for n=%lld", (long long)node0, (long long)node1, (long long)n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GML_EHIP, "no HIP device available (libgml_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GML_EINVAL, "device %d out of range (%d devices)", device, ndev);
    double M = 0;
    for (int64_t k = 0; k < K; ++k) {
        double c = counts ? counts[k] : 1.0;
        if (!(c >= 0) || !std::isfinite(c)) return fail(GML_EINVAL, "count of configuration %lld is negative or not finite", (long long)k);
        M += c;
    }
    if (!(M > 0)) return fail(GML_EINVAL, "sum of counts is zero");
    gml_problem *p = new gml_problem();
    p->device = device;
    p->n = n;
    p->K = K;
    p->M = M;
    p->order = order;
    p->node0 = node0;
    p->node1 = node1;
    *taken = true;
    int rc = alloc_dev(p, counts, spins, dspins, dev_spin_major, ld);
    if (rc != GML_OK) {
        std::string keep = g_err;
        gml_problem_destroy(p);
        g_err = keep;
        return rc;
    }
    *out = p;
    return GML_OK;
}

extern "C" int gml_problem_create_spins(const double *counts, const int8_t *spins, int64_t K, int64_t n,
                                        int order, int64_t node0, int64_t node1, int device,
                                        gml_problem **out) {
    if (!spins) return fail(GML_EINVAL, "spins is NULL");
    // the +-1 check runs on the device, on the uploaded copy (alloc_dev)
    return create_common(counts, spins, K, n, order, node0, node1, device, out);
}

extern "C" int gml_problem_create(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld,
                                  int col_major, int order, int64_t node0, int64_t node1, int device,
                                  gml_problem **out) {
    if (!samples) return fail(GML_EINVAL, "samples is NULL");
    if (K <= 0 || n <= 0) return fail(GML_EINVAL, "empty histogram (K=%lld, n=%lld)", (long long)K, (long long)n);
    if (dtype != GML_I8 && dtype != GML_I32 && dtype != GML_I64 && dtype != GML_F64)
        return fail(GML_EINVAL, "unknown dtype %d", dtype);
    if (ld < (col_major ? K : n + 1)) return fail(GML_EINVAL, "leading dimension %lld too small", (long long)ld);
    if ((double)K * (double)n >= 16.0e6) {
        if (!out) return fail(GML_EINVAL, "out is NULL");
        *out = nullptr;
        if (order < 1 || order > 8) return fail(GML_EINVAL, "interaction order %d out of range [1,8]", order);
        if (node0 < 0 || node1 > n || node0 >= node1)
            return fail(GML_EINVAL, "bad node range [%lld,%lld) for n=%lld", (long long)node0, (long long)node1, (long long)n);
        // Large histogram (e.g. the 8 GB Matrix{Int64} of sample(), column-major): upload it as it is through
        // the staged copy and convert / validate on the device; the host only sees the K counts.
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            return fail(GML_EHIP, "no HIP device available (libgml_hip has no CPU fallback)");
        if (device < 0 || device >= ndev) return fail(GML_EINVAL, "device %d out of range (%d devices)", device, ndev);
        HIPCHK(hipSetDevice(device));
        const size_t esz = dtype == GML_I8 ? 1 : (dtype == GML_I32 ? 4 : 8);
        const size_t bytes = esz * (size_t)(col_major ? ld * (n + 1) - (ld - K) : (K - 1) * ld + (n + 1));
        void *dH = nullptr;
        int8_t *dS = nullptr;
        double *dC = nullptr;
        long long *dbad = nullptr, hbad = -1;
        hipStream_t st = nullptr;
        auto cleanup = [&](int rc) {
            void *ptrs[] = {dH, dC, dbad};
            for (void *q : ptrs)
                if (q) (void)hipFree(q);
            if (st) (void)hipStreamDestroy(st);
            if (rc != GML_OK && dS) (void)hipFree(dS);
            return rc;
        };
#define CCHK(expr)                                                                                               \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,       \
                                hipGetErrorString(e_)));                                                         \
    } while (0)
        CCHK(hipStreamCreate(&st));
        CCHK(hipMalloc(&dH, bytes));
        CCHK(hipMalloc(&dS, (size_t)K * n));
        CCHK(hipMalloc(&dC, sizeof(double) * K));
        CCHK(hipMalloc(&dbad, sizeof(long long)));
        CCHK(hipMemcpyAsync(dbad, &hbad, sizeof(long long), hipMemcpyHostToDevice, st));
        int urc = upload_pageable(dH, samples, bytes, st);
        if (urc) return cleanup(urc);
        // column-major input gives spin-major bytes [n][K], row-major input sample-major [K][n]: the bit packer takes both
        launch_convert_hist(dH, dtype, K, n, ld, col_major, dC, dS, dbad, st);
        std::vector<double> counts((size_t)K);
        CCHK(hipMemcpyAsync(counts.data(), dC, sizeof(double) * K, hipMemcpyDeviceToHost, st));
        CCHK(hipMemcpyAsync(&hbad, dbad, sizeof(long long), hipMemcpyDeviceToHost, st));
        CCHK(hipGetLastError());
        CCHK(hipStreamSynchronize(st));
#undef CCHK
        if (hbad >= 0) return cleanup(fail(GML_EINVAL, "configuration %lld holds a spin that is not +-1", hbad));
        double Msum = 0;
        for (int64_t k = 0; k < K; ++k) {
            if (!(counts[k] >= 0) || !std::isfinite(counts[k]))
                return cleanup(fail(GML_EINVAL, "count of configuration %lld is negative or not finite", (long long)k));
            Msum += counts[k];
        }
        if (!(Msum > 0)) return cleanup(fail(GML_EINVAL, "sum of counts is zero"));
        cleanup(GML_OK); // every check create_common repeats has passed: it takes ownership of dS
        return create_common(counts.data(), nullptr, K, n, order, node0, node1, device, out, dS, col_major != 0, K);
    }
    auto at = [&](int64_t k, int64_t j) -> double {
        const int64_t off = col_major ? k + j * ld : k * ld + j;
        switch (dtype) {
        case GML_I8: return (double)((const int8_t *)samples)[off];
        case GML_I32: return (double)((const int32_t *)samples)[off];
        case GML_I64: return (double)((const int64_t *)samples)[off];
        default: return ((const double *)samples)[off];
        }
    };
    // data_info (:76-81): column 1 = counts, the rest = spins
    std::vector<double> counts((size_t)K);
    std::vector<int8_t> spins((size_t)K * n);
    std::atomic<int64_t> bad(-1);
    parallel_for((K + 4095) / 4096, [&](int64_t b) {
        const int64_t k1 = std::min(K, (b + 1) * 4096);
        for (int64_t k = b * 4096; k < k1; ++k) {
            counts[k] = at(k, 0);
            for (int64_t i = 0; i < n; ++i) {
                double v = at(k, 1 + i);
                if (v == 1.0) spins[k * n + i] = 1;
                else if (v == -1.0) spins[k * n + i] = -1;
                else bad = k;
            }
        }
    });
    if (bad >= 0) return fail(GML_EINVAL, "configuration %lld holds a spin that is not +-1", (long long)bad.load());
    return create_common(counts.data(), spins.data(), K, n, order, node0, node1, device, out);
}

// ------------------------------------------------------------------------------------------
// gml_problem_create_sampled: sample on the device, then build the handle from the device-resident
// samples (the step before the path; src/sampling.jl:34-57, 94-106)
// ------------------------------------------------------------------------------------------
// Terms of one model: spins of term t = keys[t*stride .. +stride) (0-based, -1 = unused slot).
static int create_sampled_terms(const int32_t *keys, int stride, const double *weights, int64_t nterms, int64_t n,
                                int64_t N, uint64_t seed, int order, int64_t node0, int64_t node1, int device,
                                gml_problem **out) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    if ((nterms > 0 && (!keys || !weights)) || stride < 1) return fail(GML_EINVAL, "NULL or malformed term list");
    if (n <= 0 || N <= 0) return fail(GML_EINVAL, "n and N must be positive");
    if (order < 1 || order > 8) return fail(GML_EINVAL, "interaction order %d out of range [1,8]", order);
    if (node0 < 0 || node1 > n || node0 >= node1)
        return fail(GML_EINVAL, "bad node range [%lld,%lld) for n=%lld", (long long)node0, (long long)node1, (long long)n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GML_EHIP, "no HIP device available (libgml_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GML_EINVAL, "device %d out of range (%d devices)", device, ndev);
    for (int64_t t = 0; t < nterms; ++t) {
        if (!std::isfinite(weights[t])) return fail(GML_EINVAL, "weight of term %lld is not finite", (long long)t);
        for (int a = 0; a < stride; ++a) {
            const int32_t v = keys[t * stride + a];
            if (v < -1 || v >= n) return fail(GML_EINVAL, "term %lld names spin %d outside [0,%lld)", (long long)t, v, (long long)n);
        }
    }
    // connected components of the term hypergraph
    std::vector<int64_t> parent((size_t)n);
    for (int64_t i = 0; i < n; ++i) parent[i] = i;
    std::function<int64_t(int64_t)> find = [&](int64_t a) {
        while (parent[a] != a) a = parent[a] = parent[parent[a]];
        return a;
    };
    for (int64_t t = 0; t < nterms; ++t) {
        if (weights[t] == 0.0) continue;
        int64_t first = -1;
        for (int a = 0; a < stride; ++a) {
            const int32_t v = keys[t * stride + a];
            if (v < 0) continue;
            if (first < 0) first = v;
            else parent[find(v)] = find(first);
        }
    }
    std::vector<std::vector<int>> blocks;
    std::vector<int64_t> id((size_t)n, -1);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = find(i);
        if (id[r] < 0) {
            id[r] = (int64_t)blocks.size();
            blocks.emplace_back();
        }
        blocks[(size_t)id[r]].push_back((int)i);
    }
    size_t maxsb = 0;
    for (auto &b : blocks) maxsb = std::max(maxsb, b.size());
    if (maxsb > 22)
        return fail(GML_EUNSUPPORTED, "a connected component of the model has %zu spins: exact enumeration is limited to 22 "
                                     "(an MCMC sampler is not implemented)", maxsb);
    // per block: its terms as bit masks over the block's spins (a repeated spin cancels: s^2 = 1)
    std::vector<int> local((size_t)n, 0);
    for (auto &b : blocks)
        for (size_t i = 0; i < b.size(); ++i) local[(size_t)b[i]] = (int)i;
    std::vector<std::vector<unsigned>> bmask(blocks.size());
    std::vector<std::vector<double>> bwt(blocks.size());
    size_t maxnt = 1;
    for (int64_t t = 0; t < nterms; ++t) {
        if (weights[t] == 0.0) continue;
        unsigned mask = 0;
        int64_t any = -1;
        for (int a = 0; a < stride; ++a) {
            const int32_t v = keys[t * stride + a];
            if (v < 0) continue;
            mask ^= 1u << local[(size_t)v];
            any = v;
        }
        if (any < 0) continue; // the empty term: a constant energy
        const size_t b = (size_t)id[find(any)];
        bmask[b].push_back(mask);
        bwt[b].push_back(weights[t]);
        maxnt = std::max(maxnt, bmask[b].size());
    }
    HIPCHK(hipSetDevice(device));
    gml_problem *p = new gml_problem();
    p->device = device;
    p->n = n;
    p->K = N;
    p->M = (double)N;
    p->order = order;
    p->node0 = node0;
    p->node1 = node1;
    hipStream_t st = nullptr;
    int8_t *dS = nullptr;
    double *dwt = nullptr, *den = nullptr, *dcdf = nullptr;
    unsigned *dmask = nullptr;
    int *dmem = nullptr;
    auto cleanup = [&](int rc) {
        if (dwt) (void)hipFree(dwt);
        if (dmask) (void)hipFree(dmask);
        if (den) (void)hipFree(den);
        if (dcdf) (void)hipFree(dcdf);
        if (dmem) (void)hipFree(dmem);
        if (st) (void)hipStreamDestroy(st);
        return rc;
    };
#define SCHK(expr)                                                                                              \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if (dS) (void)hipFree(dS);                                                                          \
            delete p;                                                                                           \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,      \
                                hipGetErrorString(e_)));                                                        \
        }                                                                                                       \
    } while (0)
    SCHK(hipStreamCreate(&st));
    SCHK(hipMalloc(&dS, (size_t)N * n));
    SCHK(hipMalloc(&dwt, sizeof(double) * maxnt));
    SCHK(hipMalloc(&dmask, sizeof(unsigned) * maxnt));
    SCHK(hipMalloc(&den, sizeof(double) * ((size_t)1 << maxsb)));
    SCHK(hipMalloc(&dcdf, sizeof(double) * ((size_t)1 << maxsb)));
    SCHK(hipMalloc(&dmem, sizeof(int) * maxsb));
    for (size_t b = 0; b < blocks.size(); ++b) {
        const auto &mem = blocks[b];
        const int sb = (int)mem.size(), nt = (int)bmask[b].size();
        if (nt > 0) {
            SCHK(hipMemcpyAsync(dmask, bmask[b].data(), sizeof(unsigned) * nt, hipMemcpyHostToDevice, st));
            SCHK(hipMemcpyAsync(dwt, bwt[b].data(), sizeof(double) * nt, hipMemcpyHostToDevice, st));
        }
        SCHK(hipMemcpyAsync(dmem, mem.data(), sizeof(int) * sb, hipMemcpyHostToDevice, st));
        launch_block_sampler(dmask, dwt, nt, sb, dmem, N, n, (unsigned long long)seed, (int)b, den, dcdf, dS, st);
        SCHK(hipGetLastError());
        SCHK(hipStreamSynchronize(st)); // the staging buffers are reused by the next block
    }
#undef SCHK
    cleanup(0);
    int rc = alloc_dev(p, nullptr, nullptr, dS);
    if (rc != GML_OK) {
        std::string keep = g_err;
        gml_problem_destroy(p);
        g_err = keep;
        return rc;
    }
    *out = p;
    return GML_OK;
}

extern "C" int gml_problem_create_mcmc_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms,
                                             int64_t n, int64_t N, uint64_t seed, int sweeps, int order, int64_t node0,
                                             int64_t node1, int device, gml_problem **out) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    if ((nterms > 0 && (!keys || !weights)) || key_stride < 1) return fail(GML_EINVAL, "NULL or malformed term list");
    if (n <= 0 || N <= 0 || sweeps < 1) return fail(GML_EINVAL, "n, N and sweeps must be positive");
    if (order < 1 || order > 8) return fail(GML_EINVAL, "interaction order %d out of range [1,8]", order);
    if (node0 < 0 || node1 > n || node0 >= node1)
        return fail(GML_EINVAL, "bad node range [%lld,%lld) for n=%lld", (long long)node0, (long long)node1, (long long)n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GML_EHIP, "no HIP device available (libgml_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GML_EINVAL, "device %d out of range (%d devices)", device, ndev);
    // incidence lists: for every spin the terms it belongs to (weight + the other spins; a spin named twice cancels)
    std::vector<std::vector<std::pair<double, std::vector<int>>>> inc((size_t)n);
    for (int64_t t = 0; t < nterms; ++t) {
        if (!std::isfinite(weights[t])) return fail(GML_EINVAL, "weight of term %lld is not finite", (long long)t);
        std::vector<int> sp;
        for (int a = 0; a < key_stride; ++a) {
            const int32_t v = keys[t * key_stride + a];
            if (v < -1 || v >= n) return fail(GML_EINVAL, "term %lld names spin %d outside [0,%lld)", (long long)t, v, (long long)n);
            if (v < 0) continue;
            auto itv = std::find(sp.begin(), sp.end(), (int)v);
            if (itv != sp.end()) sp.erase(itv); // s^2 = 1
            else sp.push_back((int)v);
        }
        if (weights[t] == 0.0) continue;
        for (size_t a = 0; a < sp.size(); ++a) {
            std::vector<int> others;
            for (size_t b = 0; b < sp.size(); ++b)
                if (b != a) others.push_back(sp[b]);
            inc[(size_t)sp[a]].emplace_back(weights[t], std::move(others));
        }
    }
    std::vector<int> ioff((size_t)n + 1, 0), ooff(1, 0), oth;
    std::vector<double> iw;
    for (int64_t i = 0; i < n; ++i) {
        for (auto &e : inc[(size_t)i]) {
            iw.push_back(e.first);
            for (int j : e.second) oth.push_back(j);
            ooff.push_back((int)oth.size());
        }
        ioff[(size_t)i + 1] = (int)iw.size();
    }
    if (iw.empty()) iw.push_back(0.0);
    if (oth.empty()) oth.push_back(0);
    HIPCHK(hipSetDevice(device));
    gml_problem *p = new gml_problem();
    p->device = device;
    p->n = n;
    p->K = N;
    p->M = (double)N;
    p->order = order;
    p->node0 = node0;
    p->node1 = node1;
    hipStream_t st = nullptr;
    const int64_t Np = round_up(N, 256);
    int8_t *dSt = nullptr;
    int *dioff = nullptr, *dooff = nullptr, *doth = nullptr;
    double *diw = nullptr;
    auto cleanup = [&](int rc) {
        void *ptrs[] = {dioff, dooff, doth, diw};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
        if (st) (void)hipStreamDestroy(st);
        return rc;
    };
#define SCHK(expr)                                                                                              \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if (dSt) (void)hipFree(dSt);                                                                        \
            delete p;                                                                                           \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,      \
                                hipGetErrorString(e_)));                                                        \
        }                                                                                                       \
    } while (0)
    SCHK(hipStreamCreate(&st));
    SCHK(hipMalloc(&dSt, (size_t)n * Np));
    SCHK(hipMalloc(&dioff, sizeof(int) * ioff.size()));
    SCHK(hipMalloc(&dooff, sizeof(int) * ooff.size()));
    SCHK(hipMalloc(&doth, sizeof(int) * oth.size()));
    SCHK(hipMalloc(&diw, sizeof(double) * iw.size()));
    SCHK(hipMemcpyAsync(dioff, ioff.data(), sizeof(int) * ioff.size(), hipMemcpyHostToDevice, st));
    SCHK(hipMemcpyAsync(dooff, ooff.data(), sizeof(int) * ooff.size(), hipMemcpyHostToDevice, st));
    SCHK(hipMemcpyAsync(doth, oth.data(), sizeof(int) * oth.size(), hipMemcpyHostToDevice, st));
    SCHK(hipMemcpyAsync(diw, iw.data(), sizeof(double) * iw.size(), hipMemcpyHostToDevice, st));
    SCHK(hipMemsetAsync(dSt, 0, (size_t)n * Np, st));
    launch_glauber(dioff, diw, dooff, doth, n, N, Np, sweeps, (unsigned long long)seed, dSt, st);
    SCHK(hipGetLastError());
    SCHK(hipStreamSynchronize(st));
#undef SCHK
    cleanup(0);
    int rc = alloc_dev(p, nullptr, nullptr, dSt, true, Np); // the chains' final states, spin-major
    if (rc != GML_OK) {
        std::string keep = g_err;
        gml_problem_destroy(p);
        g_err = keep;
        return rc;
    }
    *out = p;
    return GML_OK;
}

extern "C" int gml_problem_create_sampled_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms,
                                                int64_t n, int64_t N, uint64_t seed, int order, int64_t node0,
                                                int64_t node1, int device, gml_problem **out) {
    return create_sampled_terms(keys, key_stride, weights, nterms, n, N, seed, order, node0, node1, device, out);
}

extern "C" int gml_problem_create_sampled(const double *model, int64_t n, int64_t N, uint64_t seed, int order,
                                          int64_t node0, int64_t node1, int device, gml_problem **out) {
    if (!model || !out) return fail(GML_EINVAL, "NULL argument");
    *out = nullptr;
    if (n <= 0) return fail(GML_EINVAL, "n and N must be positive");
    // the matrix as terms: 1/2 s^T A s = sum_{i<j} A_ij s_i s_j (sampling.jl:40), prior = diagonal (:41)
    std::vector<int32_t> keys;
    std::vector<double> wts;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            const double v = model[i * n + j];
            if (j < i && v != model[j * n + i])
                return fail(GML_EINVAL, "the model matrix is not symmetric at (%lld,%lld)", (long long)i, (long long)j);
            if (v == 0.0) continue;
            keys.push_back((int32_t)j);
            keys.push_back(j < i ? (int32_t)i : -1);
            wts.push_back(v);
        }
    return create_sampled_terms(keys.data(), 2, wts.data(), (int64_t)wts.size(), n, N, seed, order, node0, node1, device, out);
}

// the +-1 configurations held by the handle, K x n row-major (for tests and for callers that want the
// samples back, e.g. to build the reference's histogram)
extern "C" int gml_problem_get_spins(gml_problem *p, int8_t *spins) {
    if (!p || !spins) return fail(GML_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    // sign bits -> +-1 bytes, sample-major, on the device (in slabs of <= 2^22 samples), one copy per slab
    const int64_t slab = std::min<int64_t>(p->K, (int64_t)1 << 22);
    int8_t *dT = nullptr;
    HIPCHK(hipMalloc(&dT, (size_t)slab * p->n));
    int rc = GML_OK;
    for (int64_t k0 = 0; k0 < p->K && rc == GML_OK; k0 += slab) {
        const int64_t kk = std::min(slab, p->K - k0);
        launch_unpack_spins(p->d, k0, kk, dT, p->st);
        if (hipMemcpyAsync(spins + k0 * p->n, dT, (size_t)kk * p->n, hipMemcpyDeviceToHost, p->st) != hipSuccess ||
            hipStreamSynchronize(p->st) != hipSuccess)
            rc = fail(GML_EHIP, "download of the spins failed: %s", hipGetErrorString(hipGetLastError()));
    }
    (void)hipFree(dT);
    return rc;
}

namespace gml {
void i8_free(void *ws);
void i8_get_v(void *ws, const int8_t **Vq, const double **tau);
const unsigned *i8_get_mmax(void *ws);
int64_t i8_hess_kmax(const DevProblem &d);
int i8_hessian(void *ws, const DevProblem &d, const int *dRowcol, const int *dF, const int *dMt, const int *hMt,
               const long long *dHoff, int64_t htotal, int R, int cap, int form, int64_t Kh, int64_t kstride, double *dH,
               hipStream_t st, std::string *err);
}

extern "C" void gml_problem_destroy(gml_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->st) (void)hipStreamSynchronize(p->st);
    void *ptrs[] = {p->d.Xs, p->d.Xt, p->d.Sb, p->d.keys, p->d.Xb, p->d.Xtb, p->d.w, p->dTheta, p->dV, p->dG, p->dF, p->dRowcol, p->dFidx, p->dMt, p->dH};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    void *hptrs[] = {p->hTh, p->hG, p->hF, p->hCtl};
    for (void *q : hptrs)
        if (q) (void)hipHostFree(q);
    if (p->dHoff) (void)hipFree(p->dHoff);
    if (p->dVec) (void)hipFree(p->dVec);
    if (p->i8ws) gml::i8_free(p->i8ws);
    if (p->st) (void)hipStreamDestroy(p->st);
    delete p;
}

extern "C" int gml_problem_info(const gml_problem *p, int64_t *n, int64_t *K, double *M, int64_t *P,
                                int64_t *node0, int64_t *node1) {
    if (!p) return fail(GML_EINVAL, "problem is NULL");
    if (n) *n = p->n;
    if (K) *K = p->K;
    if (M) *M = p->M;
    if (P) *P = p->P;
    if (node0) *node0 = p->node0;
    if (node1) *node1 = p->node1;
    return GML_OK;
}

extern "C" int gml_multi_keys(const gml_problem *p, int64_t u, int32_t *keys) {
    if (!p || !keys) return fail(GML_EINVAL, "NULL argument");
    if (u < 0 || u >= p->n) return fail(GML_EINVAL, "node %lld out of range", (long long)u);
    std::vector<int32_t> cols;
    node_cols(p, u, cols);
    const int order = p->order;
    for (int64_t j = 0; j < p->P; ++j) {
        int32_t *k = keys + j * order;
        for (int t = 0; t < order; ++t) k[t] = -1;
        k[0] = (int32_t)u;
        const int32_t c = cols[j];
        if (c != (int32_t)p->d.cconst)
            for (int t = 0; t < p->ko && t + 1 < order; ++t) k[1 + t] = p->gkeys[(size_t)c * p->ko + t];
    }
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// device pass orchestration
// ------------------------------------------------------------------------------------------
namespace gml {
// implemented in gml_kernels_i8.hip: the exact int8-limb pass (same contract as the f64 one)
int i8_pass(void **ws, const DevProblem &d, const double *dTheta, const int *dRowcol, const int *dGroups, int ngroups,
            int Rp, int form, bool want_grad, double *dF, double *dG,
            hipStream_t st, hipEvent_t *ev /* [3] or NULL */, const double *hTauOvr, std::string *err);
}

static int ensure_ws(gml_problem *p, int64_t rows) {
    const int64_t Rp = round_up(rows, 32);
    if (Rp <= p->ws_rows) return GML_OK;
    void *ptrs[] = {p->dTheta, p->dV, p->dG, p->dF, p->dRowcol};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    void *hptrs[] = {p->hTh, p->hG, p->hF, p->hCtl};
    for (void *q : hptrs)
        if (q) (void)hipHostFree(q);
    p->hTh = p->hG = p->hF = nullptr;
    p->hCtl = nullptr;
    p->dTheta = p->dV = p->dG = p->dF = nullptr;
    p->dRowcol = p->dGroups = nullptr;
    p->ws_rows = 0;
    size_t freeb = 0, totalb = 0;
    HIPCHK(hipMemGetInfo(&freeb, &totalb));
    const double need = 2.0 * Rp * p->d.Qp * 8.0;
    if (need > 0.9 * (double)freeb)
        return fail(GML_ENOMEM, "workspace of %.1f GB for %lld rows does not fit in %.1f GB free HBM", need / 1e9,
                    (long long)Rp, freeb / 1e9);
    HIPCHK(hipMalloc(&p->dTheta, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipMalloc(&p->dG, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipMalloc(&p->dF, sizeof(double) * Rp));
    // control block: rowcol [Rp] | active tiles, padded with -1 [Rp/32 + 4]; one pinned twin, one upload per pass
    HIPCHK(hipMalloc(&p->dRowcol, sizeof(int) * (Rp + Rp / 32 + 4)));
    p->dGroups = p->dRowcol + Rp;
    HIPCHK(hipHostMalloc(&p->hCtl, sizeof(int) * (Rp + Rp / 32 + 4)));
    HIPCHK(hipHostMalloc(&p->hTh, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipHostMalloc(&p->hG, sizeof(double) * Rp * p->d.Qp));
    HIPCHK(hipHostMalloc(&p->hF, sizeof(double) * Rp));
    HIPCHK(hipMemsetAsync(p->dTheta, 0, sizeof(double) * Rp * p->d.Qp, p->st));
    p->ws_rows = Rp;
    return GML_OK;
}

// What only the FP64 path needs: the sample-major byte image Xs and V [ws_rows][Kp].
static int ensure_f64(gml_problem *p) {
    DevProblem &d = p->d;
    size_t freeb = 0, totalb = 0;
    if (!d.Xs) {
        HIPCHK(hipMemGetInfo(&freeb, &totalb));
        if (2.0 * (double)d.Kp * d.Qp > 0.9 * (double)freeb)
            return fail(GML_EUNSUPPORTED, "the FP64 path needs two %.1f GB byte images of the design matrix: use precision i8x",
                        (double)d.Kp * d.Qp / 1e9);
        HIPCHK(hipMalloc(&d.Xt, (size_t)d.Kp * d.Qp));
        HIPCHK(hipMalloc(&d.Xs, (size_t)d.Kp * d.Qp));
        HIPCHK(hipMemsetAsync(d.Xt, 0, (size_t)d.Kp * d.Qp, p->st));
        HIPCHK(hipMemsetAsync(d.Xs, 0, (size_t)d.Kp * d.Qp, p->st));
        launch_expand_xt(d, d.Xt, p->st);
        HIPCHK(hipMemsetAsync(d.Xt + d.cconst * d.Kp, 1, (size_t)p->K, p->st)); // the constant statistic
        launch_transpose_i8(d.Xt, d.cconst + 1, p->K, d.Kp, d.Xs, d.Qp, p->st);
    }
    if (!p->dV) {
        HIPCHK(hipMemGetInfo(&freeb, &totalb));
        if ((double)p->ws_rows * d.Kp * 8.0 > 0.9 * (double)freeb)
            return fail(GML_ENOMEM, "FP64 workspace of %.1f GB does not fit: use precision i8x", (double)p->ws_rows * d.Kp * 8.0 / 1e9);
        HIPCHK(hipMalloc(&p->dV, sizeof(double) * p->ws_rows * d.Kp));
        HIPCHK(hipMemsetAsync(p->dV, 0, sizeof(double) * p->ws_rows * d.Kp, p->st));
    }
    return GML_OK;
}

struct RowSet {
    int64_t R = 0;
    std::vector<int64_t> node; // node id per row
};

// One device pass over the rows flagged in `act` (size R).  theta: R x Qp host, internal
// layout.  Writes f[r], and g (R x Qp) when want_grad, for the active rows only.
static int device_pass(gml_problem *p, const RowSet &rs, const std::vector<uint8_t> &act, const double *theta,
                       int form, int precision, bool want_grad, double *f, double *g, gml_stats *stats,
                       float *ms /* [2]: fwd, bwd or NULL */ = nullptr,
                       double *fnoise /* R: absolute uncertainty of f[r] (before any log) or NULL */ = nullptr,
                       const std::vector<double> *tau_ovr = nullptr /* Rp per-row tau (0 = from the bound): the solver's
                       tracked scale, or the rescaled re-run below */,
                       int depth = 0, double *vmax_out = nullptr /* R: rigorous bound on max_k |V_rk| of the evaluated rows */) {
    const int64_t R = rs.R, Qp = p->d.Qp;
    const int64_t Rp = round_up(R, 32);
    int rc = ensure_ws(p, R);
    if (rc) return rc;
    std::vector<int> rowcol((size_t)Rp, -1), groups;
    int64_t nact = 0;
    for (int64_t r = 0; r < R; ++r)
        if (act[r]) {
            rowcol[r] = (int)rs.node[r]; // row of Xt holding s_u
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
    // control block (row -> node, active tiles) through its pinned twin: one asynchronous upload
    std::vector<int> gpad = groups;
    while (gpad.size() % 4) gpad.push_back(-1);
    std::memcpy(p->hCtl, rowcol.data(), sizeof(int) * Rp);
    std::memcpy(p->hCtl + p->ws_rows, gpad.data(), sizeof(int) * gpad.size());
    HIPCHK(hipMemcpyAsync(p->dRowcol, p->hCtl, sizeof(int) * (p->ws_rows + gpad.size()), hipMemcpyHostToDevice, st));
    if (precision != GML_PREC_I8X) { // the int8 pass zeroes its own accumulators (one kernel)
        HIPCHK(hipMemsetAsync(p->dF, 0, sizeof(double) * Rp, st));
        if (want_grad) HIPCHK(hipMemsetAsync(p->dG, 0, sizeof(double) * Rp * Qp, st));
    }
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    if (ms)
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    if (precision == GML_PREC_I8X) {
        std::string err;
        rc = gml::i8_pass(&p->i8ws, p->d, p->dTheta, p->dRowcol, p->dGroups, (int)groups.size(),
                          (int)Rp, form, want_grad, p->dF, p->dG, st, ms ? ev : nullptr, tau_ovr ? tau_ovr->data() : nullptr,
                          &err);
        if (rc) return fail(rc, "%s", err.c_str());
    } else {
        rc = ensure_f64(p);
        if (rc) return rc;
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
    const bool i8exp = precision == GML_PREC_I8X && form != GML_RPLE;
    if (precision == GML_PREC_I8X) {
        const int8_t *Vq = nullptr;
        const double *tau = nullptr;
        gml::i8_get_v(p->i8ws, &Vq, &tau);
        tauh.resize((size_t)Rp);
        HIPCHK(hipMemcpyAsync(tauh.data(), tau, sizeof(double) * Rp, hipMemcpyDeviceToHost, st));
        if (i8exp) {
            mmaxh.resize((size_t)Rp);
            HIPCHK(hipMemcpyAsync(mmaxh.data(), gml::i8_get_mmax(p->i8ws), sizeof(unsigned) * Rp, hipMemcpyDeviceToHost, st));
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    if (fnoise)
        for (int64_t r = 0; r < R; ++r) {
            if (!act[r]) continue;
            // f64: summation rounding.  int8 limbs: every V_rk is rounded to a multiple of tau_r with a dither
            // that is equidistributed over the samples, so the errors (each within one unit, standard deviation
            // 0.41 tau) add like a random walk: 8 sigma of sqrt(K) terms (the worst case K * tau is never approached).
            fnoise[r] = 1e-13 * std::max(1.0, std::fabs(fh[r]));
            if (precision == GML_PREC_I8X && form != GML_RPLE) fnoise[r] += 3.3 * std::sqrt((double)p->K) * tauh[r];
        }
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
        // the largest |V_rk| actually seen is more than 8 bits below that bound (dense theta), re-run the row with
        // tau_r taken from it: (mmax + 1) tau bounds every |V_rk| rigorously, so the re-run cannot overflow.
        std::vector<uint8_t> again((size_t)R, 0);
        std::vector<double> ovr((size_t)Rp, 0.0);
        int64_t nagain = 0;
        for (int64_t r = 0; r < R; ++r)
            if (act[r] && mmaxh[r] < (1u << 23)) {
                again[r] = 1;
                ovr[r] = ((double)mmaxh[r] + 1.0) * tauh[r] * (1.0 + 1e-12) / 2130000000.0;
                ++nagain;
            }
        if (vmax_out)
            for (int64_t r = 0; r < R; ++r)
                if (act[r]) vmax_out[r] = ((double)mmaxh[r] + 1.0) * tauh[r];
        if (nagain > 0) {
            if (depth >= 6) return fail(GML_EUNSUPPORTED, "precision i8x: the weights exp(-E) of a row underflow its fixed-point range; use precision f64");
            return device_pass(p, rs, again, theta, form, precision, want_grad, f, g, stats, nullptr, fnoise, &ovr, depth + 1, vmax_out);
        }
    }
    return GML_OK;
}

// Working-set Hessians of the active rows (int8 kernel over the limb planes of the last pass, or the
// FP64 MFMA kernel over V).  Fidx: R x cap column ids (padding = Qp-1), m[r] = working-set size
// (0 = skip).  The result is ragged: row r's block starts at hoff[r] in Hout and is mp x mp with
// mp = 32*ceil(m[r]/32) (lower 32x32 tiles filled).
static int device_newton(gml_problem *p, const RowSet &rs, const std::vector<int> &Fidx, const std::vector<int> &m,
                         int cap, int form, int precision, int64_t Kh, int64_t kstride, const std::vector<double> &s1, double s2,
                         const std::vector<double> &gF, const std::vector<double> &pgF, std::vector<double> &dout,
                         std::vector<double> &sdiag, gml_stats *stats) {
    std::vector<long long> hoff;
    const int64_t R = rs.R;
    const double t0 = now_s();
    std::vector<int> mt2((size_t)3 * R);
    hoff.assign((size_t)R + 1, 0);
    for (int64_t r = 0; r < R; ++r) {
        mt2[r] = (m[r] + 31) / 32;
        mt2[R + r] = (int)rs.node[r];
        mt2[2 * R + r] = m[r];
        hoff[r + 1] = hoff[r] + (long long)mt2[r] * 32 * mt2[r] * 32;
    }
    const int64_t htotal = std::max<long long>(hoff[R], 1);
    if (R > p->hs_rows || (int64_t)R * cap > p->hs_cap || htotal > p->hs_elems) {
        void *ptrs[] = {p->dFidx, p->dMt, p->dH, p->dHoff, p->dVec};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
        p->dFidx = p->dMt = nullptr;
        p->dH = p->dVec = nullptr;
        p->dHoff = nullptr;
        p->hs_rows = std::max(R, p->hs_rows);
        p->hs_cap = std::max<int64_t>((int64_t)R * cap, p->hs_cap);
        p->hs_elems = std::max<int64_t>(htotal + htotal / 4, p->hs_elems);
        HIPCHK(hipMalloc(&p->dFidx, sizeof(int) * p->hs_cap));
        HIPCHK(hipMalloc(&p->dMt, sizeof(int) * 3 * p->hs_rows));
        HIPCHK(hipMalloc(&p->dHoff, sizeof(long long) * (p->hs_rows + 1)));
        HIPCHK(hipMalloc(&p->dH, sizeof(double) * p->hs_elems));
        HIPCHK(hipMalloc(&p->dVec, sizeof(double) * (3 * p->hs_cap + 2 * p->hs_rows))); // gF | pgF | d | s1 | Sdiag
    }
    hipStream_t st = p->st;
    HIPCHK(hipMemcpyAsync(p->dFidx, Fidx.data(), sizeof(int) * R * cap, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->dMt, mt2.data(), sizeof(int) * 3 * R, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->dHoff, hoff.data(), sizeof(long long) * (R + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(p->dH, 0, sizeof(double) * htotal, st));
    bool done = false;
    if (precision == GML_PREC_I8X) {
        std::string err;
        int hrc = gml::i8_hessian(p->i8ws, p->d, p->dMt + R, p->dFidx, p->dMt, mt2.data(), p->dHoff, htotal, (int)R, cap, form,
                                  Kh, kstride, p->dH, st, &err);
        if (hrc == GML_OK) done = true;
        else if (hrc != GML_EUNSUPPORTED) return fail(hrc, "%s", err.c_str());
    }
    if (!done) {
        if (precision == GML_PREC_I8X)
            return fail(GML_EUNSUPPORTED, "a Newton block above 512 entries (the solver caps max_working at 512)");
        launch_hess_f64(p->d, p->dV, p->dMt + R, p->dFidx, p->dMt, p->dHoff, (int)R, cap, form, Kh, kstride, p->dH, st);
    }
    HIPCHK(hipGetLastError());
    // Newton systems solved in place on the device; only the directions come back
    double *dg = p->dVec, *dpg = dg + (int64_t)R * cap, *dd = dpg + (int64_t)R * cap, *ds1 = dd + (int64_t)R * cap,
           *dsd = ds1 + R;
    HIPCHK(hipMemcpyAsync(dg, gF.data(), sizeof(double) * R * cap, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dpg, pgF.data(), sizeof(double) * R * cap, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ds1, s1.data(), sizeof(double) * R, hipMemcpyHostToDevice, st));
    launch_newton_solve(p->dH, p->dHoff, p->dMt, p->dMt + 2 * R, ds1, s2, dg, dpg, (int)R, cap, dd, dsd, st);
    HIPCHK(hipGetLastError());
    dout.resize((size_t)R * cap);
    sdiag.resize((size_t)R);
    HIPCHK(hipMemcpyAsync(dout.data(), dd, sizeof(double) * R * cap, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(sdiag.data(), dsd, sizeof(double) * R, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (stats) {
        stats->t_hess += now_s() - t0;
        ++stats->hessian_passes;
    }
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// layouts: reference parameter vector <-> internal column layout
// ------------------------------------------------------------------------------------------
struct NodeLayout {
    std::vector<int32_t> cols; // reference slot j -> internal column
};

static void build_layout(const gml_problem *p, int64_t u, NodeLayout &L) {
    if (p->order == 2) { // slot i <-> spin i, slot u = field (:162)
        L.cols.resize((size_t)p->n);
        for (int64_t i = 0; i < p->n; ++i) L.cols[i] = (int32_t)(i == u ? p->d.cconst : i);
    } else {
        node_cols(p, u, L.cols);
    }
}

static inline double pseudo_grad(double x, double g, double lam) {
    if (lam == 0.0) return g;
    if (x > 0) return g + lam;
    if (x < 0) return g - lam;
    if (g + lam < 0) return g + lam;
    if (g - lam > 0) return g - lam;
    return 0.0;
}

// ------------------------------------------------------------------------------------------
// gml_learn: batched working-set orthant-wise Newton.
//
// Every local node u solves   min_x f_u(x) + lambda * sum_{j penalised} |x_j|   -- the problem
// the reference builds for Ipopt with the z >= |x| epigraph (:166-177) -- in lock-step:
//   1. one device pass gives f and the full gradient of every active node;
//   2. pseudo-gradient / KKT residual per node; converged nodes drop out;
//   3. working set = non-zeros + the largest violators; its Hessian comes from one device
//      kernel over the same K configurations; Newton step by Cholesky on the host;
//   4. projected (orthant) backtracking line search: first trial is a full pass (it usually
//      succeeds), further trials are objective-only passes over the rows that need them.
// ------------------------------------------------------------------------------------------
extern "C" int gml_learn(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts_in,
                         double *out, double *kkt_out, gml_stats *stats_out) {
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
    o.max_working = (int)round_up(o.max_working, 32);
    if (o.max_add <= 0) o.max_add = 64;
    HIPCHK(hipSetDevice(p->device));
    const int64_t dbg_row = getenv("GML_DEBUG_ROW") ? atoll(getenv("GML_DEBUG_ROW")) : 0; // row traced at verbose >= 2
    gml_stats stl;
    std::memset(&stl, 0, sizeof stl);
    gml_stats *stats = &stl;
    const double t_start = now_s();

    const int64_t R = p->node1 - p->node0, Qp = p->d.Qp, Q = p->d.Qp, P = p->P;
    const int32_t cconst = (int32_t)p->d.cconst;
    const double lambda = gml_lambda(regularizer_c, p->n, p->M);
    stats->lambda = lambda;
    // sub-sampled Newton: Hessians over the first Kh configurations, rescaled by M / M_h.  The budget
    // (rows x configurations) is kept roughly constant: as nodes converge, the remaining ones get more
    // configurations, up to all of them -- an inexact Hessian only costs iterations, and it costs the
    // most on the few ill-conditioned nodes that are still active at the end.
    // sub-sampled Newton: Hessians over Kh configurations -- every kstride-th block of 512, so that a sorted histogram
    // is sampled evenly -- rescaled by the weight of the sub-sample.  The budget (rows x configurations) is kept
    // roughly constant: as nodes converge, the remaining ones get more configurations, up to all of them -- an
    // inexact Hessian only costs iterations, and it costs the most on the few ill-conditioned nodes that are still
    // active at the end.
    const int64_t Kh_base = o.hess_samples == 0 ? 32768 : (o.hess_samples < 0 ? p->d.Kp : (int64_t)o.hess_samples);
    const int64_t nblk512 = p->d.Kp / 512;
    int64_t Kh = p->d.Kp, kstride = 1;
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
            Kh = p->d.Kp;
            wsum = 1.0;
        }
        hscale = nb == nblk512 ? 1.0 : 1.0 / wsum;
    };
    set_kh(R);

    RowSet rs;
    rs.R = R;
    rs.node.resize((size_t)R);
    for (int64_t r = 0; r < R; ++r) rs.node[r] = p->node0 + r;

    // kind[r][c]: 0 = structurally absent (key contains u), 1 = free, 2 = l1-penalised
    std::vector<uint8_t> kind((size_t)R * Qp, 0);
    std::vector<NodeLayout> lay((size_t)R);
    parallel_for(R, [&](int64_t r) {
        build_layout(p, rs.node[r], lay[r]);
        uint8_t *kr = kind.data() + r * Qp;
        for (int32_t c : lay[r].cols) kr[c] = (c == cconst) ? 1 : 2; // length(inter) > 1 is penalised (:118,:171)
    });

    std::vector<double> X((size_t)R * Qp, 0.0), G((size_t)R * Qp, 0.0), Xt((size_t)R * Qp, 0.0),
        Gt((size_t)R * Qp, 0.0), Xbest((size_t)R * Qp, 0.0);
    std::vector<double> f((size_t)R, 0.0), ft((size_t)R, 0.0), Fobj((size_t)R, 0.0), kkt((size_t)R, INFINITY),
        best((size_t)R, INFINITY), Z((size_t)R, 1.0), Zt((size_t)R, 1.0), alpha((size_t)R, 1.0), dd((size_t)R, 0.0),
        fn((size_t)R, 0.0), fnt((size_t)R, 0.0);
    std::vector<uint8_t> done((size_t)R, 0), act((size_t)R, 1), need((size_t)R, 0), vstale((size_t)R, 0), atfloor((size_t)R, 0);
    // FP64 polish of the rows the int8-limb path leaves above tol: possible when the FP64 workspaces fit
    bool can_polish = false;
    if (o.precision == GML_PREC_I8X && o.polish >= 0) {
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) == hipSuccess) {
            const double need_b = (p->d.Xs ? 0.0 : 2.0 * (double)p->d.Kp * (double)p->d.Qp) + (p->dV ? 0.0 : 8.0 * (double)round_up(R, 32) * (double)p->d.Kp);
            can_polish = need_b < 0.8 * (double)freeb;
        }
    }
    int stall_cap = can_polish ? 4 : 10;
    std::vector<int> stall((size_t)R, 0), msz((size_t)R, 0), mtot((size_t)R, 0), blk((size_t)R, 0);
    std::vector<std::vector<int>> Fset((size_t)R);
    std::vector<std::vector<double>> Dset((size_t)R), PGset((size_t)R);
    // Scale of the fixed-point V (int8 path): instead of the worst-case bound w_max exp(sum|theta|) every pass after
    // a row's first uses vref = max_k |V_rk| measured by its previous pass, times exp(||theta - theta_ref||_1),
    // which bounds the new weights rigorously (|E_k' - E_k| <= ||theta' - theta||_1).  Near the optimum the steps
    // are tiny, so V keeps all 31 bits relative to its actual maximum and the noise floor of f and grad drops by the
    // bits the bound would have wasted.
    int prec = o.precision; // switches to FP64 for the rows the int8-limb path cannot bring below tol (see "polish" below)
    bool track_scale = prec == GML_PREC_I8X && formulation != GML_RPLE;
    std::vector<double> vref((size_t)R, 0.0), dref((size_t)R, 0.0), stepn((size_t)R, 0.0), vnew((size_t)R, 0.0),
        ovr((size_t)round_up(R, 32), 0.0);
    auto scale_for = [&](const std::vector<uint8_t> &rows, bool at_trial) -> const std::vector<double> * {
        if (!track_scale) return nullptr;
        for (int64_t r = 0; r < R; ++r)
            ovr[r] = (rows[r] && vref[r] > 0.0) ? vref[r] * std::exp(dref[r] + (at_trial ? stepn[r] : 0.0)) * (1.0 + 1e-6) / 2130000000.0 : 0.0;
        return &ovr;
    };
    auto scale_seen = [&](const std::vector<uint8_t> &rows, bool at_trial) {
        if (!track_scale) return;
        for (int64_t r = 0; r < R; ++r)
            if (rows[r]) {
                vref[r] = vnew[r];
                dref[r] = at_trial ? stepn[r] : 0.0; // distance from the current iterate to the point just evaluated
            }
    };

    // logRISE post-processing of a pass: f = log Z, g = grad Z / Z   (:279)
    auto post = [&](const std::vector<uint8_t> &a, std::vector<double> &fv, std::vector<double> &gv,
                    std::vector<double> &zv, std::vector<double> &nv, bool grad) {
        if (formulation != GML_LOGRISE) return;
        parallel_for(R, [&](int64_t r) {
            if (!a[r]) return;
            const double z = fv[r];
            zv[r] = z;
            fv[r] = std::log(z);
            nv[r] = nv[r] / z; // uncertainty of log Z
            if (grad) {
                double *gr = gv.data() + r * Qp;
                for (int64_t c = 0; c < Q; ++c) gr[c] /= z;
            }
        });
    };

    int rc = device_pass(p, rs, act, X.data(), formulation, prec, true, f.data(), G.data(), stats, nullptr,
                         fn.data(), nullptr, 0, vnew.data());
    if (rc) return rc;
    scale_seen(act, false);
    post(act, f, G, Z, fn, true);

    int it = 0;
    for (it = 0; it < o.max_iter; ++it) {
        const double th0 = now_s();
        // ---- KKT residuals, working sets ----------------------------------------------
        parallel_for(R, [&](int64_t r) {
            if (done[r]) return;
            const double *x = X.data() + r * Qp, *g = G.data() + r * Qp;
            const uint8_t *kr = kind.data() + r * Qp;
            double F = f[r], worst = 0, worstW = 0;
            std::vector<std::pair<double, int>> viol;
            std::vector<int> &Fs = Fset[r];
            Fs.clear();
            for (int64_t c = 0; c < Q; ++c) {
                if (!kr[c]) continue;
                const double l = kr[c] == 2 ? lambda : 0.0;
                F += l * std::fabs(x[c]);
                const double pg = pseudo_grad(x[c], g[c], l);
                if (std::fabs(pg) > worst) worst = std::fabs(pg);
                if (x[c] != 0.0 || kr[c] == 1) {
                    Fs.push_back((int)c);
                    worstW = std::max(worstW, std::fabs(pg));
                } else if (pg != 0.0) {
                    viol.emplace_back(-std::fabs(pg), (int)c);
                }
            }
            if (!std::isfinite(worst)) worst = INFINITY;
            Fobj[r] = F;
            kkt[r] = worst;
            if (worst < best[r]) {
                best[r] = worst;
                std::memcpy(Xbest.data() + r * Qp, x, sizeof(double) * Qp);
                stall[r] = 0;
            } else {
                ++stall[r];
            }
            if (worst <= o.tol) {
                done[r] = 1;
                return;
            }
            if (stall[r] >= stall_cap) { // no progress: at the noise floor of the pass arithmetic (or a failed line search)
                done[r] = 1;
                atfloor[r] = 1;
                return;
            }
            // Only the max_add largest violators are admitted per iteration: at theta = 0 most
            // coordinates violate |g| <= lambda merely through <s_u><s_c> (non-zero magnetisations), and
            // stop doing so once the field and the strongest couplings have been fitted; and none at all
            // while the residual on the current support still dominates (the violations outside are then
            // largely an artefact of the unconverged support).
            if (worstW > worst * 0.999999 && worstW > 0 && !viol.empty() && (int)Fs.size() > 1) viol.clear();
            if ((int)viol.size() > o.max_add) {
                std::nth_element(viol.begin(), viol.begin() + o.max_add, viol.end());
                viol.resize(o.max_add);
            }
            // Newton block W: everything free if it fits the cap.  Otherwise (a denser optimum than the
            // cap: lambda at or below the sampling noise) block Gauss-Seidel: the free coordinates are
            // ranked by max(|pg|, |x| * f) and the iterations cycle through consecutive blocks of that
            // ranking (the unpenalised slot is in every block); all other coordinates stay fixed, so the
            // block Newton step cannot overshoot through couplings it ignores.
            const int capW = o.max_working;
            if ((int)(Fs.size() + viol.size()) <= capW) {
                for (auto &v : viol) Fs.push_back(v.second);
                blk[r] = 0;
            } else {
                std::vector<std::pair<double, int>> cand;
                cand.reserve(Fs.size() + viol.size());
                const double fs = std::max(std::fabs(formulation == GML_LOGRISE ? 1.0 : f[r]), 1e-300);
                int cfree = -1;
                for (int c : Fs) {
                    if (kr[c] == 1) {
                        cfree = c;
                        continue;
                    }
                    cand.emplace_back(-std::max(std::fabs(pseudo_grad(x[c], g[c], lambda)), std::fabs(x[c]) * fs), c);
                }
                for (auto &v : viol) cand.emplace_back(v.first, v.second);
                std::sort(cand.begin(), cand.end());
                const int per = capW - 1, nblk = ((int)cand.size() + per - 1) / per;
                const int b = blk[r] % nblk;
                blk[r] = (blk[r] + 1) % nblk;
                Fs.clear();
                if (cfree >= 0) Fs.push_back(cfree);
                for (int a = b * per; a < std::min<int>((b + 1) * per, (int)cand.size()); ++a) Fs.push_back(cand[a].second);
            }
            std::sort(Fs.begin(), Fs.end());
        });
        int64_t nactive = 0;
        double worst_all = 0;
        int maxm = 0;
        for (int64_t r = 0; r < R; ++r) {
            if (!done[r]) {
                ++nactive;
                maxm = std::max<int>(maxm, (int)Fset[r].size());
            }
            worst_all = std::max(worst_all, std::min(kkt[r], best[r]));
        }
        if (o.verbose)
            fprintf(stderr, "[gml] it %3d active %6lld  max-kkt %.3e  max|F| %d  passes %d fwd %d\n", it,
                    (long long)nactive, worst_all, maxm, stats->passes, stats->forward_passes);
        if (nactive == 0) {
            // Polish: rows that the int8-limb arithmetic could not bring below tol (its gradient carries ~sqrt(K) 2^-31
            // of noise relative to the largest weight, which an ill-conditioned, weakly regularised problem amplifies)
            // continue on the FP64 path from their best iterate, when that path fits in memory.
            int64_t nfloor = 0;
            for (int64_t r = 0; r < R; ++r) nfloor += (atfloor[r] && !(std::min(best[r], kkt[r]) <= o.tol));
            if (!(prec == GML_PREC_I8X && can_polish && nfloor > 0)) break;
            if (ensure_f64(p) != GML_OK) break; // does not fit after all: the rows stay as they are (reported not converged)
            prec = GML_PREC_F64;
            track_scale = false;
            stall_cap = 10;
            std::fill(need.begin(), need.end(), 0);
            for (int64_t r = 0; r < R; ++r) {
                if (!atfloor[r] || std::min(best[r], kkt[r]) <= o.tol) continue;
                if (best[r] <= kkt[r]) std::memcpy(X.data() + r * Qp, Xbest.data() + r * Qp, sizeof(double) * Qp);
                done[r] = 0;
                atfloor[r] = 0;
                stall[r] = 0;
                best[r] = INFINITY;
                blk[r] = 0;
                need[r] = 1;
            }
            if (o.verbose) fprintf(stderr, "[gml] polish: %lld rows continue on the FP64 path\n", (long long)nfloor);
            rc = device_pass(p, rs, need, X.data(), formulation, prec, true, f.data(), G.data(), stats, nullptr, fn.data(), nullptr, 0,
                             nullptr);
            if (rc) return rc;
            post(need, f, G, Z, fn, true);
            for (int64_t r = 0; r < R; ++r)
                if (need[r]) vstale[r] = 0;
            set_kh(nfloor);
            ++stats->polished;
            continue;
        }
        set_kh(nactive);

        // rows whose V was overwritten by a rejected trial need a fresh pass before the Hessian
        bool anystale = false;
        for (int64_t r = 0; r < R; ++r) {
            need[r] = (!done[r] && vstale[r]);
            anystale |= need[r] != 0;
        }
        stats->t_host += now_s() - th0;
        if (anystale) {
            rc = device_pass(p, rs, need, X.data(), formulation, prec, true, f.data(), G.data(), stats, nullptr,
                             fn.data(), scale_for(need, false), 0, vnew.data());
            if (rc) return rc;
            scale_seen(need, false);
            post(need, f, G, Z, fn, true);
            for (int64_t r = 0; r < R; ++r)
                if (need[r]) vstale[r] = 0;
        }

        // ---- Newton directions on the working sets (Hessian + Cholesky solve on the device) ----
        const double th1 = now_s();
        const int cap = (int)round_up(std::max(maxm, 1), 32);
        std::vector<int> Fidx((size_t)R * cap, (int)(Qp - 1));
        std::vector<double> gFm((size_t)R * cap, 0.0), pgFm((size_t)R * cap, 0.0), s1v((size_t)R, 1.0);
        parallel_for(R, [&](int64_t r) {
            msz[r] = done[r] ? 0 : (int)Fset[r].size();
            const double *x = X.data() + r * Qp, *g = G.data() + r * Qp;
            const uint8_t *kr = kind.data() + r * Qp;
            for (int a = 0; a < msz[r]; ++a) {
                const int c = Fset[r][a];
                Fidx[(size_t)r * cap + a] = c;
                gFm[(size_t)r * cap + a] = g[c];
                pgFm[(size_t)r * cap + a] = pseudo_grad(x[c], g[c], kr[c] == 2 ? lambda : 0.0);
            }
            s1v[r] = formulation == GML_LOGRISE ? hscale / Z[r] : hscale; // Hess log Z = Hess Z / Z - g g^T
        });
        stats->t_host += now_s() - th1;
        std::vector<double> Dn, Sd;
        rc = device_newton(p, rs, Fidx, msz, cap, formulation, prec, Kh, kstride, s1v, formulation == GML_LOGRISE ? 1.0 : 0.0,
                           gFm, pgFm, Dn, Sd, stats);
        if (rc) return rc;
        const double th1b = now_s();
        parallel_for(R, [&](int64_t r) {
            if (done[r]) return;
            const int m = msz[r];
            std::vector<double> bw(Dn.begin() + (size_t)r * cap, Dn.begin() + (size_t)r * cap + m);
            std::vector<double> pgv(pgFm.begin() + (size_t)r * cap, pgFm.begin() + (size_t)r * cap + m);
            mtot[r] = (int)Fset[r].size();
            Dset[r] = bw;
            PGset[r] = pgv;
        });
        stats->t_host += now_s() - th1b;

        // ---- projected backtracking line search ----------------------------------------
        // Two acceptance regimes per row:
        //  * the predicted decrease is well above the uncertainty of f  -> Armijo on F;
        //  * otherwise ("noise regime": near the optimum, or a noisy int8-limb f) function values
        //    cannot certify the step; the trial is then a full pass and is accepted iff it lowers
        //    the KKT residual (max |pseudo-gradient|), which is what convergence is measured by.
        std::vector<uint8_t> nreg((size_t)R, 0);
        for (int64_t r = 0; r < R; ++r) {
            need[r] = !done[r];
            alpha[r] = 1.0;
        }
        std::vector<uint8_t> accepted_fwd((size_t)R, 0);
        for (int ls = 0; ls < 30; ++ls) {
            const double th2 = now_s();
            bool any = false, anynoise = false;
            parallel_for(R, [&](int64_t r) {
                if (!need[r]) return;
                const double *x = X.data() + r * Qp;
                double *xt = Xt.data() + r * Qp;
                const uint8_t *kr = kind.data() + r * Qp;
                std::memcpy(xt, x, sizeof(double) * Qp);
                const std::vector<int> &Fs = Fset[r];
                double d_ = 0;
                for (int a = 0; a < mtot[r]; ++a) {
                    const int c = Fs[a];
                    double v = x[c] + alpha[r] * Dset[r][a];
                    if (kr[c] == 2 && lambda > 0) {
                        const double pg = PGset[r][a];
                        const double xi = x[c] != 0.0 ? (x[c] > 0 ? 1.0 : -1.0) : (pg < 0 ? 1.0 : -1.0);
                        if (v * xi < 0) v = 0.0; // crossed zero: clip to the orthant face
                    }
                    xt[c] = v;
                    d_ += PGset[r][a] * (v - x[c]);
                }
                if (!(d_ < 0)) {
                    // The projected step is not a descent direction: some coordinate's Newton step crossed zero and
                    // was clipped, while the other coordinates still carry the moves that were meant to accompany
                    // it.  Take only the clipping (each such coordinate moves towards its 1-D minimiser, so F
                    // decreases); the coordinate then leaves the working set and the next Newton system is right.
                    std::memcpy(xt, x, sizeof(double) * Qp);
                    d_ = 0;
                    for (int a = 0; a < mtot[r]; ++a) {
                        const int c = Fs[a];
                        if (kr[c] != 2 || !(lambda > 0) || x[c] == 0.0) continue;
                        const double v = x[c] + alpha[r] * Dset[r][a];
                        if (v * x[c] < 0) {
                            xt[c] = 0.0;
                            d_ += PGset[r][a] * (0.0 - x[c]);
                        }
                    }
                }
                dd[r] = d_;
                double sn = 0;
                for (int a = 0; a < mtot[r]; ++a) sn += std::fabs(xt[Fs[a]] - x[Fs[a]]);
                stepn[r] = sn; // ||trial - x||_1: bounds the change of every energy
                nreg[r] = !(-0.1 * d_ > 8.0 * fn[r]); // the step's expected decrease (~|dd|/2) vs the uncertainty of f
            });
            for (int64_t r = 0; r < R; ++r) {
                any |= need[r] != 0;
                anynoise |= (need[r] && nreg[r]);
            }
            stats->t_host += now_s() - th2;
            if (!any) break;
            const bool full = (ls == 0) || anynoise;
            rc = device_pass(p, rs, need, Xt.data(), formulation, prec, full, ft.data(), Gt.data(), stats, nullptr,
                             fnt.data(), scale_for(need, true), 0, vnew.data());
            if (rc) return rc;
            scale_seen(need, true);
            post(need, ft, Gt, Zt, fnt, full);
            const double th3 = now_s();
            parallel_for(R, [&](int64_t r) {
                if (!need[r]) return;
                vstale[r] = 1;
                const double *xt = Xt.data() + r * Qp;
                const uint8_t *kr = kind.data() + r * Qp;
                bool ok;
                if (nreg[r]) {
                    // F is convex: F(x) >= F(xt) + F'(xt; x - xt).  So a trial whose directional derivative
                    // BACK towards x is >= 0 cannot have increased F; a small negative value (overshoot of at
                    // most ~1.5x the minimiser along the step) is tolerated.  Coordinates clipped to zero
                    // contribute lambda|s_c| - g_c s_c >= 0 whenever they belong at zero.
                    const double *gt = Gt.data() + r * Qp, *x0 = X.data() + r * Qp;
                    double back = 0;
                    for (int a = 0; a < mtot[r]; ++a) {
                        const int c = Fset[r][a];
                        const double sc = x0[c] - xt[c]; // direction back to x
                        if (sc == 0.0) continue;
                        double gl = gt[c] * sc;
                        if (kr[c] == 2) gl += lambda * (xt[c] != 0.0 ? (xt[c] > 0 ? sc : -sc) : std::fabs(sc));
                        back += gl;
                    }
                    ok = std::isfinite(ft[r]) && std::isfinite(back) && back >= -0.5 * std::fabs(dd[r]);
                } else {
                    double Fn = ft[r];
                    for (int64_t c = 0; c < Q; ++c)
                        if (kr[c] == 2 && xt[c] != 0.0) Fn += lambda * std::fabs(xt[c]);
                    ok = std::isfinite(Fn) && Fn <= Fobj[r] + 1e-4 * dd[r] + fn[r] + fnt[r];
                }
                if (o.verbose >= 2 && r == dbg_row)
                    fprintf(stderr, "[gml]   row %lld: ls %d alpha %.3g nreg %d ft %.12e Fobj %.12e dd %.3e fnt %.3e ok %d\n", (long long)r, ls, alpha[r],
                            (int)nreg[r], ft[r], Fobj[r], dd[r], fnt[r], (int)ok);
                if (ok) {
                    dref[r] = 0.0; // the iterate moves onto the point the scale was measured at
                    std::memcpy(X.data() + r * Qp, xt, sizeof(double) * Qp);
                    f[r] = ft[r];
                    fn[r] = fnt[r];
                    Z[r] = Zt[r];
                    if (full) {
                        std::memcpy(G.data() + r * Qp, Gt.data() + r * Qp, sizeof(double) * Qp);
                        vstale[r] = 0;
                    } else {
                        accepted_fwd[r] = 1;
                    }
                    need[r] = 0;
                } else {
                    alpha[r] *= 0.5;
                    if (nreg[r] && alpha[r] < 1.0 / 64) {
                        need[r] = 0; // cannot improve along this direction: the stall counter ends the row,
                        stall[r] += 3; // after at most three such line searches (each costs ~7 passes)
                    }
                }
            });
            stats->t_host += now_s() - th3;
        }
        // rows accepted on an objective-only trial still need their gradient (and V)
        bool anyf = false;
        for (int64_t r = 0; r < R; ++r) anyf |= accepted_fwd[r] != 0;
        if (anyf) {
            rc = device_pass(p, rs, accepted_fwd, X.data(), formulation, prec, true, f.data(), G.data(), stats,
                             nullptr, fn.data(), scale_for(accepted_fwd, false), 0, vnew.data());
            if (rc) return rc;
            scale_seen(accepted_fwd, false);
            post(accepted_fwd, f, G, Z, fn, true);
            for (int64_t r = 0; r < R; ++r)
                if (accepted_fwd[r]) vstale[r] = 0;
        }
        // rows whose line search failed entirely: they stay where they are; the stall counter ends them
    }

    // ---- results in the reference layout --------------------------------------------------
    int notconv = 0;
    double maxk = 0;
    for (int64_t r = 0; r < R; ++r) {
        const double k = std::min(best[r], kkt[r]);
        if (!(k <= o.tol)) ++notconv;
        maxk = std::max(maxk, k);
        if (kkt_out) kkt_out[r] = k;
    }
    std::vector<double> res((size_t)R * P);
    parallel_for(R, [&](int64_t r) {
        const double *x = (best[r] <= kkt[r] ? Xbest.data() : X.data()) + r * Qp;
        for (int64_t j = 0; j < P; ++j) res[(size_t)r * P + j] = x[lay[r].cols[j]];
    });
    hipPointerAttribute_t attr;
    bool dev_out = false;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess) dev_out = (attr.type == hipMemoryTypeDevice);
    else (void)hipGetLastError();
    if (dev_out) {
        HIPCHK(hipMemcpy(out, res.data(), sizeof(double) * R * P, hipMemcpyHostToDevice));
    } else {
        std::memcpy(out, res.data(), sizeof(double) * R * P);
    }
    stats->iterations = it;
    stats->max_kkt = maxk;
    stats->not_converged = notconv;
    stats->t_total = now_s() - t_start;
    if (stats_out) *stats_out = *stats;
    if (notconv)
        return fail(GML_ENOTCONV, "%d of %lld nodes did not reach the KKT tolerance %.1e (worst %.3e)", notconv,
                    (long long)R, o.tol, maxk);
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
    for (int64_t r = 0; r < nrows; ++r)
        if (nodes[r] < 0 || nodes[r] >= p->n) return fail(GML_EINVAL, "node id %lld out of range", (long long)nodes[r]);
    HIPCHK(hipSetDevice(p->device));
    const int64_t Qp = p->d.Qp, P = p->P;
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
    int rc = device_pass(p, rs, act, Th.data(), formulation, precision, g != nullptr, fv.data(), Gi.data(), nullptr);
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

// Timing hook with the parameters RESIDENT in HBM: Theta is uploaded once, then `warmup + steps` passes run back
// to back on the handle's stream with no host round trip (a device-side optimiser would call the operator this
// way); f and the gradient of the last pass are downloaded once at the end.  kernel_ms[3] = device time per pass.
extern "C" int gml_bench_pass_resident(gml_problem *p, int formulation, int precision, const double *theta, int steps,
                                       int warmup, double kernel_ms[4], double *f_out, double *g_out) {
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
    for (int64_t r = 0; r < Rp; ++r) p->hCtl[r] = r < R ? (int)(p->node0 + r) : -1;
    int npad = 0;
    for (int g = 0; g < ngroups || (npad % 4); ++g, ++npad) p->hCtl[p->ws_rows + g] = g < ngroups ? g : -1;
    HIPCHK(hipMemcpyAsync(p->dTheta, p->hTh, sizeof(double) * Rp * Qp, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->dRowcol, p->hCtl, sizeof(int) * (p->ws_rows + npad), hipMemcpyHostToDevice, st));
    if (precision != GML_PREC_I8X) {
        rc = ensure_f64(p);
        if (rc) return rc;
    }
    std::vector<hipEvent_t> ev((size_t)3 * steps, nullptr);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int s = 0; s < warmup + steps; ++s) {
        hipEvent_t *e3 = s >= warmup ? ev.data() + (size_t)3 * (s - warmup) : nullptr;
        if (precision == GML_PREC_I8X) {
            std::string err;
            rc = gml::i8_pass(&p->i8ws, p->d, p->dTheta, p->dRowcol, p->dGroups, ngroups, (int)Rp, formulation, true, p->dF,
                              p->dG, st, e3, nullptr, &err);
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
    if (precision == GML_PREC_I8X && formulation != GML_RPLE) {
        mm.resize((size_t)Rp);
        HIPCHK(hipMemcpyAsync(mm.data(), gml::i8_get_mmax(p->i8ws), sizeof(unsigned) * Rp, hipMemcpyDeviceToHost, st));
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
        int rc = device_pass(p, rs, act, Th.data(), formulation, precision, true, fv.data(), Gi.data(), nullptr, ms);
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
