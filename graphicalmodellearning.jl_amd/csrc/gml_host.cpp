// libgml_hip: C ABI (include/gml.h) + host side of the MI355X learn() hot path.
//
// What lives here: histogram validation/packing, the per-node parameter layout of the
// reference (pairwise :162, multi-body :94-104), the batched working-set Newton solver that
// replaces the reference's per-node Ipopt solve (:164-181), and the orchestration of the
// device passes.  All arithmetic over the K configurations happens in HIP kernels
// (gml_kernels_f64*.hip, gml_i8_*.hip, gml_kernels_i8w.hip); there is no CPU fallback for it.
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

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int gml_fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *gml_last_error(void) { return g_err.c_str(); }

double gml_now_s() {
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

void gml_parallel_for(int64_t n, const std::function<void(int64_t)> &fn) { pool().run(n, fn); }


int64_t gml_binom(int64_t n, int64_t k) {
    if (k < 0 || k > n) return 0;
    int64_t r = 1;
    for (int64_t i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

extern "C" double gml_lambda(double c, int64_t n, double M) {
    // lambda = regularizer*sqrt(log((num_spins^2)/0.05)/num_samples)   (:157)
    return c * std::sqrt(std::log(((double)n * (double)n) / 0.05) / M);
}

// ABI identity (include/gml.h): the bindings compare these with their own mirrors when they load the library
extern "C" int gml_abi_version(void) { return GML_ABI_VERSION; }
extern "C" int64_t gml_sizeof_opts(void) { return (int64_t)sizeof(gml_opts); }
extern "C" int64_t gml_sizeof_stats(void) { return (int64_t)sizeof(gml_stats); }

extern "C" void gml_default_opts(gml_opts *o) {
    std::memset(o, 0, sizeof *o);
    o->tol = 1e-9;
    o->max_iter = 100;
    o->precision = GML_PREC_AUTO; // the int8-limb fast path (rows it leaves above tol are finished in FP64: polish = 0), FP64 for tiny problems
    o->max_working = 512;
    o->max_add = 64;
    o->verbose = 0;
}

// next q-subset of {0..n-1} in lexicographic order; returns false after the last one
bool gml_next_comb(std::vector<int> &idx, int64_t n) {
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
void gml_node_cols(const gml_problem *p, int64_t u, std::vector<int32_t> &cols) {
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
        } while (gml_next_comb(idx, p->n));
    }
}


extern "C" void gml_problem_destroy(gml_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->st) (void)hipStreamSynchronize(p->st);
    void *ptrs[] = {p->d.Xt, p->d.Sb, p->d.keys, p->d.Xb, p->d.Xtb, p->d.w, p->dTheta, p->dV, p->dG, p->dF, p->dSrow, p->opCols, p->opFlag, p->opSel, p->opG2};
    for (void *q : ptrs)
        if (q) (void)dev_free(q);
    void *hptrs[] = {p->hTh, p->hG, p->hF, p->hCtl, p->stage};
    for (void *q : hptrs)
        if (q) (void)hipHostFree(q);
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
    gml_node_cols(p, u, cols);
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
// layouts: reference parameter vector <-> internal column layout
// ------------------------------------------------------------------------------------------
void gml_build_layout(const gml_problem *p, int64_t u, NodeLayout &L) {
    if (p->order == 2) { // slot i <-> spin i, slot u = field (:162)
        L.cols.resize((size_t)p->n);
        for (int64_t i = 0; i < p->n; ++i) L.cols[i] = (int32_t)(i == u ? p->d.cconst : i);
    } else {
        gml_node_cols(p, u, L.cols);
    }
}

