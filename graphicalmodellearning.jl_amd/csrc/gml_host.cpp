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
static double now_s() { return gml_now_s(); }

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
static void parallel_for(int64_t n, const std::function<void(int64_t)> &fn) { gml_parallel_for(n, fn); }

static int64_t round_up(int64_t a, int64_t b) { return gml_round_up(a, b); }

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
    o->precision = GML_PREC_AUTO; // the int8-limb fast path (rows it leaves above tol are finished in FP64: polish = 0), FP64 for tiny problems
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

// ------------------------------------------------------------------------------------------
// Building the device-resident problem.  What stays resident of the samples is one bit per entry: the sign bits of
// the spins (Sb) and the two MFMA operand images derived from them (Xb, Xtb): 2/8 byte per (configuration, statistic)
// + 1/8 per (configuration, spin).
//   prob_layout   sizes, statistic keys, allocations (Sb zeroed)
//   prob_weights  w_k = counts_k / M (:170) and the host-side summaries of them
//   (Sb is filled by the host packer through the pinned stages below, or on the device from sampled bytes)
//   prob_images   Sb, keys -> Xb, Xtb
// ------------------------------------------------------------------------------------------
static int prob_layout(gml_problem *p) {
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
        HIPCHK(dev_mem_info(&freeb, &totalb));
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
    d.Xt = nullptr; // FP64 path only, built on first use (ensure_f64)
    HIPCHK(dev_malloc(&d.Sb, (size_t)p->n * (d.Kp / 8)));
    HIPCHK(dev_malloc(&d.keys, sizeof(int32_t) * p->gkeys.size()));
    HIPCHK(dev_malloc(&d.Xb, (size_t)d.Kp * (d.Qfp / 8)));
    HIPCHK(dev_malloc(&d.Xtb, (size_t)xtb_bytes(d)));
    HIPCHK(dev_malloc(&d.w, sizeof(double) * d.Kp));
    HIPCHK(hipMemcpyAsync(d.keys, p->gkeys.data(), sizeof(int32_t) * p->gkeys.size(), hipMemcpyHostToDevice, p->st));
    return GML_OK;
}

// host-side summaries of the weights, shared by the parts of a multi-GPU problem
struct WeightInfo {
    std::vector<double> w, wblk;
    double wmax = 0, wuni = 0;
};
static void weight_info(const double *counts /* NULL: all ones */, int64_t K, int64_t Kp, double M, WeightInfo &wi) {
    wi.w.assign((size_t)Kp, 0.0); // 0 on the padding configurations
    wi.wmax = 0;
    for (int64_t k = 0; k < K; ++k) {
        wi.w[k] = (counts ? counts[k] : 1.0) / M; // w_k = counts[k]/M  (:170)
        wi.wmax = std::max(wi.wmax, wi.w[k]);
    }
    wi.wuni = wi.w[0];
    for (int64_t k = 1; k < K; ++k)
        if (wi.w[k] != wi.w[0]) {
            wi.wuni = 0.0;
            break;
        }
    wi.wblk.assign((size_t)(Kp / 512), 0.0); // weight of every block of 512 configurations (sub-sampled Hessians)
    for (int64_t k = 0; k < K; ++k) wi.wblk[(size_t)(k >> 9)] += wi.w[k];
}
static int prob_weights(gml_problem *p, const WeightInfo &wi) {
    DevProblem &d = p->d;
    d.wmax = wi.wmax;
    d.wuni = wi.wuni;
    p->wblk = wi.wblk;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(d.w, wi.w.data(), sizeof(double) * d.Kp, hipMemcpyHostToDevice, p->st));
    HIPCHK(hipStreamSynchronize(p->st)); // wi may be a temporary of the caller
    return GML_OK;
}
static int prob_images(gml_problem *p) {
    HIPCHK(hipSetDevice(p->device));
    launch_pack_bits(p->d, p->st);
    HIPCHK(hipGetLastError());
    return GML_OK;
}

static int check_create_args(int64_t K, int64_t n, int order, int64_t node0, int64_t node1, int device) {
    if (K <= 0 || n <= 0) return fail(GML_EINVAL, "empty histogram (K=%lld, n=%lld)", (long long)K, (long long)n);
    if (order < 1 || order > 8) return fail(GML_EINVAL, "interaction order %d out of range [1,8]", order);
    if (node0 < 0 || node1 > n || node0 >= node1)
        return fail(GML_EINVAL, "bad node range [%lld,%lld) for n=%lld", (long long)node0, (long long)node1, (long long)n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GML_EHIP, "no HIP device available (libgml_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GML_EINVAL, "device %d out of range (%d devices)", device, ndev);
    return GML_OK;
}

static gml_problem *new_problem(int64_t K, int64_t n, double M, int order, int64_t node0, int64_t node1, int device) {
    gml_problem *p = new gml_problem();
    p->device = device;
    p->n = n;
    p->K = K;
    p->M = M;
    p->order = order;
    p->node0 = node0;
    p->node1 = node1;
    return p;
}

// Handle from +-1 bytes that are already on the device (the samplers): sample-major [K][n] or, with spin_major,
// [n][ld].  `dbytes` is owned from here on and freed on every path.  dedupe: the handle holds the DISTINCT configurations
// with their multiplicities (the reference's countmap, sampling.jl:52) instead of one row per draw.
static int create_from_device_bytes(gml_problem *p, int8_t *dbytes, bool spin_major, int64_t ld, const double *counts, gml_problem **out,
                                    bool dedupe = false) {
    struct Guard {
        void *b[3];
        ~Guard() {
            for (void *q : b)
                if (q) (void)dev_free(q);
        }
    } guard{{dbytes, nullptr, nullptr}};
    const double t0 = now_s();
    int rc = GML_OK;
    unsigned long long *dkeys = nullptr;
    std::vector<double> hcounts;
    if (dedupe) {
        if (p->n > 64 || p->K >= ((int64_t)1 << 31)) rc = fail(GML_EUNSUPPORTED, "histogramming on the device needs n <= 64 spins and fewer than 2^31 samples");
        hipStream_t st0 = nullptr;
        if (rc == GML_OK && (hipSetDevice(p->device) != hipSuccess || hipStreamCreate(&st0) != hipSuccess)) rc = fail(GML_EHIP, "hipStreamCreate failed");
        if (rc == GML_OK) {
            std::string err;
            int *dcnt = nullptr;
            int64_t Kd = 0;
            rc = dedupe_samples(dbytes, spin_major, ld, p->K, p->n, st0, &dkeys, &dcnt, &Kd, &err);
            guard.b[1] = dkeys;
            guard.b[2] = dcnt;
            if (rc) rc = fail(rc, "%s", err.c_str());
            else {
                std::vector<int> hc((size_t)Kd);
                if (hipMemcpy(hc.data(), dcnt, sizeof(int) * Kd, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GML_EHIP, "download of the counts failed");
                hcounts.assign(hc.begin(), hc.end());
                p->M = (double)p->K; // every draw counted once
                p->K = Kd;
                counts = hcounts.data();
            }
        }
        if (st0) (void)hipStreamDestroy(st0);
        (void)dev_free(dbytes); // the draws are no longer needed
        guard.b[0] = nullptr;
    }
    if (rc == GML_OK) rc = prob_layout(p);
    if (rc == GML_OK) {
        WeightInfo wi;
        weight_info(counts, p->K, p->d.Kp, p->M, wi);
        rc = prob_weights(p, wi);
    }
    if (rc == GML_OK) {
        if (hipMemsetAsync(p->d.Sb, 0, (size_t)p->n * (p->d.Kp / 8), p->st) != hipSuccess) rc = fail(GML_EHIP, "hipMemsetAsync failed");
    }
    if (rc == GML_OK) {
        if (dedupe) launch_bits_from_keys(dkeys, p->K, p->n, p->d.Kp, p->d.Sb, p->st);
        else launch_spin_bits(dbytes, spin_major, p->K, p->n, ld, p->d.Kp, p->d.Sb, p->st);
        const double t1 = now_s();
        rc = prob_images(p);
        if (rc == GML_OK && hipStreamSynchronize(p->st) != hipSuccess) rc = fail(GML_EHIP, "building the bit images failed: %s", hipGetErrorString(hipGetLastError()));
        p->t_ingest[2] = now_s() - t1;
    }
    p->t_ingest[3] = now_s() - t0;
    if (rc != GML_OK) {
        std::string keep = g_err;
        gml_problem_destroy(p);
        g_err = keep;
        return rc;
    }
    *out = p;
    return GML_OK;
}

// ---- pinned stages of the ingest pipeline: two buffers, allocated once per process (page-locking 32 MB costs more than
// packing a small histogram), handed to one ingest at a time (the packer uses the whole worker pool anyway)
namespace {
constexpr size_t kStageBytes = (size_t)16 << 20;
struct StagePair {
    std::mutex m;
    void *buf[2] = {nullptr, nullptr};
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (cap >= bytes) return GML_OK;
        for (auto &b : buf) {
            if (b) (void)hipHostFree(b);
            b = nullptr;
        }
        cap = 0;
        for (auto &b : buf)
            if (hipHostMalloc(&b, bytes, hipHostMallocPortable) != hipSuccess) {
                (void)hipGetLastError();
                return fail(GML_ENOMEM, "no pinned staging memory (%zu bytes)", bytes);
            }
        cap = bytes;
        return GML_OK;
    }
};
StagePair &stages() {
    static StagePair *s = new StagePair(); // leaked on purpose: no destructor order issues with the HIP runtime at exit
    return *s;
}
} // namespace

// Sb of every part from a producer of sign-word rows: `fill(i0, i1, dst)` writes the rows of the spins [i0, i1)
// ([i1 - i0][wpr] words) into pinned memory and returns the first bad configuration or -1.  Each chunk is produced
// ONCE and copied to every part (pack once, replicate); producing chunk c+1 overlaps the copies of chunk c.
static int fill_sign_bits(const std::vector<gml_problem *> &parts, const std::function<int64_t(int64_t, int64_t, uint32_t *)> &fill,
                          double *t_fill, double *t_wait) {
    gml_problem *p0 = parts[0];
    const int64_t n = p0->n, wpr = p0->d.Kp / 32;
    const size_t rowb = (size_t)wpr * 4;
    StagePair &sp = stages();
    std::lock_guard<std::mutex> lk(sp.m);
    int rc = sp.ensure(std::max(kStageBytes, rowb));
    if (rc) return rc;
    int64_t rows = std::max<int64_t>(1, (int64_t)(sp.cap / rowb));
    if (rows >= 32) rows = rows / 32 * 32; // whole 32-column groups for the row-major packer
    const size_t G = parts.size();
    std::vector<hipEvent_t> ev(2 * G, nullptr);
    auto cleanup = [&](int code) {
        for (size_t g = 0; g < G; ++g) {
            (void)hipSetDevice(parts[g]->device);
            (void)hipStreamSynchronize(parts[g]->st); // nothing may still read the stages
            for (int b = 0; b < 2; ++b)
                if (ev[b * G + g]) (void)hipEventDestroy(ev[b * G + g]);
        }
        return code;
    };
    for (size_t g = 0; g < G; ++g) {
        if (hipSetDevice(parts[g]->device) != hipSuccess) return cleanup(fail(GML_EHIP, "hipSetDevice failed"));
        for (int b = 0; b < 2; ++b)
            if (hipEventCreateWithFlags(&ev[b * G + g], hipEventDisableTiming) != hipSuccess) return cleanup(fail(GML_EHIP, "hipEventCreate failed"));
    }
    int b = 0;
    int64_t chunk = 0;
    for (int64_t i0 = 0; i0 < n; i0 += rows, b ^= 1, ++chunk) {
        const int64_t i1 = std::min(n, i0 + rows);
        double t0 = now_s();
        if (chunk >= 2)
            for (size_t g = 0; g < G; ++g)
                if (hipEventSynchronize(ev[b * G + g]) != hipSuccess) return cleanup(fail(GML_EHIP, "hipEventSynchronize failed"));
        double t1 = now_s();
        *t_wait += t1 - t0;
        const int64_t bad = fill(i0, i1, static_cast<uint32_t *>(sp.buf[b]));
        *t_fill += now_s() - t1;
        if (bad >= 0) return cleanup(fail(GML_EINVAL, "configuration %lld holds a spin that is not +-1", (long long)bad));
        for (size_t g = 0; g < G; ++g) {
            gml_problem *p = parts[g];
            if (hipSetDevice(p->device) != hipSuccess ||
                hipMemcpyAsync(reinterpret_cast<char *>(p->d.Sb) + (size_t)i0 * rowb, sp.buf[b], (size_t)(i1 - i0) * rowb, hipMemcpyHostToDevice, p->st) != hipSuccess ||
                hipEventRecord(ev[b * G + g], p->st) != hipSuccess)
                return cleanup(fail(GML_EHIP, "upload of the sign bits failed: %s", hipGetErrorString(hipGetLastError())));
        }
    }
    const double t0 = now_s();
    const int code = cleanup(GML_OK);
    *t_wait += now_s() - t0;
    return code;
}

// Handles for the node ranges `ranges` on `devices` from ONE host histogram: counts and sign bits are produced once.
static int create_parts(const std::function<int64_t(int64_t, int64_t, uint32_t *)> &fill, const double *counts, double M, int64_t K, int64_t n,
                        int order, const std::vector<std::pair<int64_t, int64_t>> &ranges, const std::vector<int> &devices, double t_counts,
                        std::vector<gml_problem *> &parts) {
    const double t_begin = now_s();
    const size_t G = ranges.size();
    parts.assign(G, nullptr);
    auto destroy_all = [&](int code) {
        std::string keep = g_err;
        for (auto &q : parts) {
            if (q) gml_problem_destroy(q);
            q = nullptr;
        }
        g_err = keep;
        return code;
    };
    WeightInfo wi;
    double t_alloc = 0, t_weights = 0;
    for (size_t g = 0; g < G; ++g) {
        parts[g] = new_problem(K, n, M, order, ranges[g].first, ranges[g].second, devices[g]);
        const double ta = now_s();
        int rc = prob_layout(parts[g]);
        const double tb = now_s();
        t_alloc += tb - ta;
        if (rc == GML_OK && g == 0) weight_info(counts, K, parts[0]->d.Kp, M, wi);
        if (rc == GML_OK) rc = prob_weights(parts[g], wi);
        t_weights += now_s() - tb;
        if (rc) return destroy_all(rc);
    }
    double t_fill = t_counts, t_wait = 0;
    int rc = fill_sign_bits(parts, fill, &t_fill, &t_wait);
    if (rc) return destroy_all(rc);
    const double t1 = now_s();
    for (size_t g = 0; g < G && rc == GML_OK; ++g) rc = prob_images(parts[g]);
    for (size_t g = 0; g < G && rc == GML_OK; ++g)
        if (hipSetDevice(parts[g]->device) != hipSuccess || hipStreamSynchronize(parts[g]->st) != hipSuccess)
            rc = fail(GML_EHIP, "building the bit images failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc) return destroy_all(rc);
    const double t2 = now_s();
    for (auto *q : parts) {
        q->t_ingest[0] = t_fill;                                       // host: counts + sign words (once for all parts)
        q->t_ingest[1] = (t1 - t_begin) - (t_fill - t_counts);        // allocations, weights, copies not hidden by the packing
        q->t_ingest[2] = t2 - t1;                                      // Xb, Xtb
        q->t_ingest[3] = t2 - t_begin + t_counts;
        q->t_ingest[4] = t_alloc;   // of t[1]: stream + device allocations (all parts)
        q->t_ingest[5] = t_weights; // of t[1]: weights w = counts / M, their summaries and upload
    }
    return GML_OK;
}

// histogram on the host (any layout) -> parts
static int create_parts_from_hist(const HistView &hv, int order, const std::vector<std::pair<int64_t, int64_t>> &ranges,
                                  const std::vector<int> &devices, std::vector<gml_problem *> &parts) {
    for (size_t g = 0; g < ranges.size(); ++g) {
        const int rc = check_create_args(hv.K, hv.n, order, ranges[g].first, ranges[g].second, devices[g]);
        if (rc) return rc;
    }
    const ParallelFor pf = [](int64_t cnt, const std::function<void(int64_t)> &fn) { gml_parallel_for(cnt, fn); };
    const double t0 = now_s();
    std::vector<double> counts((size_t)hv.K);
    double M = 0; // data_info (:76-81): column 1 = counts, M = their sum
    const int64_t badc = pack_counts(hv, counts.data(), &M, pf);
    if (badc >= 0) return fail(GML_EINVAL, "count of configuration %lld is negative or not finite", (long long)badc);
    if (!(M > 0)) return fail(GML_EINVAL, "sum of counts is zero");
    const double t_counts = now_s() - t0;
    auto fill = [&](int64_t i0, int64_t i1, uint32_t *dst) { return pack_spins(hv, i0, i1, gml_round_up(hv.K, 1024) / 32, dst, pf); };
    return create_parts(fill, hv.counts ? counts.data() : nullptr, M, hv.K, hv.n, order, ranges, devices, t_counts, parts);
}

static int create_one_from_hist(const HistView &hv, int order, int64_t node0, int64_t node1, int device, gml_problem **out) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    std::vector<gml_problem *> parts;
    const int rc = create_parts_from_hist(hv, order, {{node0, node1}}, {device}, parts);
    if (rc) return rc;
    *out = parts[0];
    return GML_OK;
}

extern "C" int gml_problem_create_spins(const double *counts, const int8_t *spins, int64_t K, int64_t n,
                                        int order, int64_t node0, int64_t node1, int device,
                                        gml_problem **out) {
    if (!spins) return fail(GML_EINVAL, "spins is NULL");
    HistView hv{};
    hv.base = spins;
    hv.dtype = GML_I8;
    hv.K = K;
    hv.n = n;
    hv.ld = n;
    hv.col_major = false;
    hv.spin_off = 0;
    hv.counts = counts;
    hv.counts_dtype = GML_F64;
    hv.counts_stride = 1;
    return create_one_from_hist(hv, order, node0, node1, device, out);
}

static int check_hist_args(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major) {
    if (!samples) return fail(GML_EINVAL, "samples is NULL");
    if (K <= 0 || n <= 0) return fail(GML_EINVAL, "empty histogram (K=%lld, n=%lld)", (long long)K, (long long)n);
    if (dtype != GML_I8 && dtype != GML_I32 && dtype != GML_I64 && dtype != GML_F64)
        return fail(GML_EINVAL, "unknown dtype %d", dtype);
    if (ld < (col_major ? K : n + 1)) return fail(GML_EINVAL, "leading dimension %lld too small", (long long)ld);
    return GML_OK;
}

extern "C" int gml_problem_create(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld,
                                  int col_major, int order, int64_t node0, int64_t node1, int device,
                                  gml_problem **out) {
    const int rc = check_hist_args(samples, dtype, K, n, ld, col_major);
    if (rc) return rc;
    return create_one_from_hist(hist_view(samples, dtype, K, n, ld, col_major != 0), order, node0, node1, device, out);
}

// The parts of a multi-GPU problem (gml_multi.cpp): the histogram is packed once, its bits are copied to every device.
int gml_create_parts(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, int order,
                     const std::vector<std::pair<int64_t, int64_t>> &ranges, const std::vector<int> &devices,
                     std::vector<gml_problem *> &parts) {
    const int rc = check_hist_args(samples, dtype, K, n, ld, col_major);
    if (rc) return rc;
    return create_parts_from_hist(hist_view(samples, dtype, K, n, ld, col_major != 0), order, ranges, devices, parts);
}

// ---- host-only packing entry points (no device needed) ------------------------------------------------------------------------
extern "C" int64_t gml_packed_words(int64_t K) { return K > 0 ? gml_round_up(K, 1024) / 32 : 0; }

extern "C" int gml_pack_histogram(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major,
                                  uint32_t *sign_bits, int64_t words_per_spin, double *counts, double *M) {
    const int rc = check_hist_args(samples, dtype, K, n, ld, col_major);
    if (rc) return rc;
    if (!sign_bits || !counts) return fail(GML_EINVAL, "NULL output");
    if (words_per_spin < (K + 31) / 32) return fail(GML_EINVAL, "words_per_spin %lld too small for K=%lld", (long long)words_per_spin, (long long)K);
    const HistView hv = hist_view(samples, dtype, K, n, ld, col_major != 0);
    const ParallelFor pf = [](int64_t cnt, const std::function<void(int64_t)> &fn) { gml_parallel_for(cnt, fn); };
    double Ms = 0;
    const int64_t badc = pack_counts(hv, counts, &Ms, pf);
    if (badc >= 0) return fail(GML_EINVAL, "count of configuration %lld is negative or not finite", (long long)badc);
    if (!(Ms > 0)) return fail(GML_EINVAL, "sum of counts is zero");
    if (M) *M = Ms;
    const int64_t bad = pack_spins(hv, 0, n, words_per_spin, sign_bits, pf);
    if (bad >= 0) return fail(GML_EINVAL, "configuration %lld holds a spin that is not +-1", (long long)bad);
    return GML_OK;
}

extern "C" int gml_problem_create_packed(const uint32_t *sign_bits, int64_t words_per_spin, const double *counts, int64_t K, int64_t n,
                                         int order, int64_t node0, int64_t node1, int device, gml_problem **out) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    if (!sign_bits) return fail(GML_EINVAL, "sign_bits is NULL");
    int rc = check_create_args(K, n, order, node0, node1, device);
    if (rc) return rc;
    if (words_per_spin < (K + 31) / 32) return fail(GML_EINVAL, "words_per_spin %lld too small for K=%lld", (long long)words_per_spin, (long long)K);
    double M = 0;
    for (int64_t k = 0; k < K; ++k) {
        const double c = counts ? counts[k] : 1.0;
        if (!(c >= 0) || !std::isfinite(c)) return fail(GML_EINVAL, "count of configuration %lld is negative or not finite", (long long)k);
        M += c;
    }
    if (!(M > 0)) return fail(GML_EINVAL, "sum of counts is zero");
    const int64_t wpr = gml_round_up(K, 1024) / 32, wreal = (K + 31) / 32;
    const uint32_t tailmask = (K & 31) ? ((1u << (K & 31)) - 1u) : 0xFFFFFFFFu;
    auto fill = [&](int64_t i0, int64_t i1, uint32_t *dst) -> int64_t {
        gml_parallel_for(i1 - i0, [&](int64_t a) {
            uint32_t *row = dst + a * wpr;
            std::memcpy(row, sign_bits + (i0 + a) * words_per_spin, sizeof(uint32_t) * wreal);
            row[wreal - 1] &= tailmask; // bits beyond K belong to padding configurations: they must be zero
            std::memset(row + wreal, 0, sizeof(uint32_t) * (wpr - wreal));
        });
        return -1;
    };
    std::vector<gml_problem *> parts;
    rc = create_parts(fill, counts, M, K, n, order, {{node0, node1}}, {device}, 0.0, parts);
    if (rc) return rc;
    *out = parts[0];
    return GML_OK;
}

extern "C" int gml_problem_get_sign_bits(gml_problem *p, uint32_t *sign_bits) {
    if (!p || !sign_bits) return fail(GML_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(sign_bits, p->d.Sb, (size_t)p->n * (p->d.Kp / 8), hipMemcpyDeviceToHost, p->st));
    HIPCHK(hipStreamSynchronize(p->st));
    return GML_OK;
}

extern "C" int gml_problem_ingest_times(const gml_problem *p, double t[6]) {
    if (!p || !t) return fail(GML_EINVAL, "NULL argument");
    for (int i = 0; i < 6; ++i) t[i] = p->t_ingest[i];
    return GML_OK;
}

// ------------------------------------------------------------------------------------------
// gml_problem_create_device_convert: the other ingest route -- the raw matrix goes over PCIe as it is (staged copy) and
// is validated and converted on the device; the host only sees the K counts.  64x the PCIe bytes of the packed route:
// for hosts whose cores are scarcer than their PCIe bandwidth.  Produces the same bits (tests/test_gpu_ingest.py).
// ------------------------------------------------------------------------------------------
extern "C" int gml_problem_create_device_convert(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld,
                                                 int col_major, int order, int64_t node0, int64_t node1, int device,
                                                 gml_problem **out) {
    int rc = check_hist_args(samples, dtype, K, n, ld, col_major);
    if (rc) return rc;
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    rc = check_create_args(K, n, order, node0, node1, device);
    if (rc) return rc;
    const double t_begin = now_s();
    HIPCHK(hipSetDevice(device));
    const size_t esz = dtype == GML_I8 ? 1 : (dtype == GML_I32 ? 4 : 8);
    const size_t bytes = esz * (size_t)(col_major ? ld * (n + 1) - (ld - K) : (K - 1) * ld + (n + 1));
    void *dH = nullptr;
    int8_t *dS = nullptr;
    double *dC = nullptr;
    long long *dbad = nullptr, hbad = -1;
    hipStream_t st = nullptr;
    auto cleanup = [&](int code) {
        void *ptrs[] = {dH, dC, dbad};
        for (void *q : ptrs)
            if (q) (void)dev_free(q);
        if (st) (void)hipStreamDestroy(st);
        if (code != GML_OK && dS) (void)dev_free(dS);
        return code;
    };
#define CCHK(expr)                                                                                               \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,       \
                                hipGetErrorString(e_)));                                                         \
    } while (0)
    CCHK(hipStreamCreate(&st));
    CCHK(dev_malloc(&dH, bytes));
    CCHK(dev_malloc(&dS, (size_t)K * n));
    CCHK(dev_malloc(&dC, sizeof(double) * K));
    CCHK(dev_malloc(&dbad, sizeof(long long)));
    CCHK(hipMemcpyAsync(dbad, &hbad, sizeof(long long), hipMemcpyHostToDevice, st));
    int urc = upload_pageable(dH, samples, bytes, st);
    if (urc) return cleanup(urc);
    const double t_up = now_s();
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
    cleanup(GML_OK); // dS passes to the handle
    gml_problem *p = new_problem(K, n, Msum, order, node0, node1, device);
    rc = create_from_device_bytes(p, dS, col_major != 0, K, counts.data(), out);
    if (rc == GML_OK) {
        (*out)->t_ingest[0] = 0.0;
        (*out)->t_ingest[1] = t_up - t_begin; // the raw upload
        (*out)->t_ingest[3] = now_s() - t_begin;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// gml_problem_create_sampled: sample on the device, then build the handle from the device-resident
// samples (the step before the path; src/sampling.jl:34-57, 94-106)
// ------------------------------------------------------------------------------------------
// Terms of one model: spins of term t = keys[t*stride .. +stride) (0-based, -1 = unused slot).
static int create_sampled_terms(const int32_t *keys, int stride, const double *weights, int64_t nterms, int64_t n,
                                int64_t N, uint64_t seed, int order, int64_t node0, int64_t node1, int device,
                                gml_problem **out, bool dedupe = false) {
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
        if (dwt) (void)dev_free(dwt);
        if (dmask) (void)dev_free(dmask);
        if (den) (void)dev_free(den);
        if (dcdf) (void)dev_free(dcdf);
        if (dmem) (void)dev_free(dmem);
        if (st) (void)hipStreamDestroy(st);
        return rc;
    };
#define SCHK(expr)                                                                                              \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if (dS) (void)dev_free(dS);                                                                          \
            delete p;                                                                                           \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,      \
                                hipGetErrorString(e_)));                                                        \
        }                                                                                                       \
    } while (0)
    SCHK(hipStreamCreate(&st));
    SCHK(dev_malloc(&dS, (size_t)N * n));
    SCHK(dev_malloc(&dwt, sizeof(double) * maxnt));
    SCHK(dev_malloc(&dmask, sizeof(unsigned) * maxnt));
    SCHK(dev_malloc(&den, sizeof(double) * ((size_t)1 << maxsb)));
    SCHK(dev_malloc(&dcdf, sizeof(double) * ((size_t)1 << maxsb)));
    SCHK(dev_malloc(&dmem, sizeof(int) * maxsb));
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
    return create_from_device_bytes(p, dS, false, 0, nullptr, out, dedupe);
}

static int create_mcmc_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms, int64_t n, int64_t N, uint64_t seed,
                             int sweeps, int order, int64_t node0, int64_t node1, int device, gml_problem **out, bool dedupe) {
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
            if (q) (void)dev_free(q);
        if (st) (void)hipStreamDestroy(st);
        return rc;
    };
#define SCHK(expr)                                                                                              \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) {                                                                                 \
            if (dSt) (void)dev_free(dSt);                                                                        \
            delete p;                                                                                           \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "%s failed: %s", #expr,      \
                                hipGetErrorString(e_)));                                                        \
        }                                                                                                       \
    } while (0)
    SCHK(hipStreamCreate(&st));
    SCHK(dev_malloc(&dSt, (size_t)n * Np));
    SCHK(dev_malloc(&dioff, sizeof(int) * ioff.size()));
    SCHK(dev_malloc(&dooff, sizeof(int) * ooff.size()));
    SCHK(dev_malloc(&doth, sizeof(int) * oth.size()));
    SCHK(dev_malloc(&diw, sizeof(double) * iw.size()));
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
    return create_from_device_bytes(p, dSt, true, Np, nullptr, out, dedupe); // the chains' final states, spin-major
}

extern "C" int gml_problem_create_mcmc_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms,
                                             int64_t n, int64_t N, uint64_t seed, int sweeps, int order, int64_t node0,
                                             int64_t node1, int device, gml_problem **out) {
    return create_mcmc_terms(keys, key_stride, weights, nterms, n, N, seed, sweeps, order, node0, node1, device, out, false);
}

extern "C" int gml_problem_create_sampled_hist(const int32_t *keys, int key_stride, const double *weights, int64_t nterms, int64_t n,
                                               int64_t N, uint64_t seed, int mcmc_sweeps, int order, int64_t node0, int64_t node1,
                                               int device, gml_problem **out) {
    if (n > 64) return fail(GML_EUNSUPPORTED, "histogramming on the device needs n <= 64 spins (n = %lld)", (long long)n);
    if (mcmc_sweeps > 0) return create_mcmc_terms(keys, key_stride, weights, nterms, n, N, seed, mcmc_sweeps, order, node0, node1, device, out, true);
    return create_sampled_terms(keys, key_stride, weights, nterms, n, N, seed, order, node0, node1, device, out, true);
}

extern "C" int gml_problem_get_counts(gml_problem *p, double *counts) {
    if (!p || !counts) return fail(GML_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(counts, p->d.w, sizeof(double) * p->K, hipMemcpyDeviceToHost, p->st));
    HIPCHK(hipStreamSynchronize(p->st));
    for (int64_t k = 0; k < p->K; ++k) counts[k] = std::nearbyint(counts[k] * p->M * 1e6) / 1e6; // w_k = counts_k / M (:170)
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
    HIPCHK(dev_malloc(&dT, (size_t)slab * p->n));
    int rc = GML_OK;
    for (int64_t k0 = 0; k0 < p->K && rc == GML_OK; k0 += slab) {
        const int64_t kk = std::min(slab, p->K - k0);
        launch_unpack_spins(p->d, k0, kk, dT, p->st);
        if (hipMemcpyAsync(spins + k0 * p->n, dT, (size_t)kk * p->n, hipMemcpyDeviceToHost, p->st) != hipSuccess ||
            hipStreamSynchronize(p->st) != hipSuccess)
            rc = fail(GML_EHIP, "download of the spins failed: %s", hipGetErrorString(hipGetLastError()));
    }
    (void)dev_free(dT);
    return rc;
}

extern "C" void gml_problem_destroy(gml_problem *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->st) (void)hipStreamSynchronize(p->st);
    void *ptrs[] = {p->d.Xt, p->d.Sb, p->d.keys, p->d.Xb, p->d.Xtb, p->d.w, p->dTheta, p->dV, p->dG, p->dF, p->dSrow};
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
                       int depth = 0) {
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
            if (depth >= 6) return fail(GML_EUNSUPPORTED, "precision i8x / i8w: the weights exp(-E) of a row underflow its fixed-point range; use precision f64");
            return device_pass(p, rs, again, theta, form, precision, want_grad, f, g, stats, nullptr, &ovr, depth + 1);
        }
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
        node_cols(p, u, L.cols);
    }
}
static void build_layout(const gml_problem *p, int64_t u, NodeLayout &L) { gml_build_layout(p, u, L); }

// ------------------------------------------------------------------------------------------
// gml_objgrad_batch: the operator (:191-208, :221-233)
// ------------------------------------------------------------------------------------------
extern "C" int gml_objgrad_batch(gml_problem *p, int formulation, int precision, int64_t nrows,
                                 const int64_t *nodes, const double *theta, int64_t ld, double *f, double *g) {
    if (!p || !nodes || !theta || !f) return fail(GML_EINVAL, "NULL argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (nrows <= 0) return fail(GML_EINVAL, "nrows must be positive");
    if (ld < p->P) return fail(GML_EINVAL, "ld %lld smaller than the %lld parameters per node", (long long)ld, (long long)p->P);
    {
        const int asked = precision;
        precision = gml_resolve_precision(p, asked);
        if (precision < 0) return fail(GML_EINVAL, "unknown precision %d", asked);
    }
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

// ------------------------------------------------------------------------------------------
// gml_hessvec_batch: curvature operator, H_u(theta) v for many nodes at once
// ------------------------------------------------------------------------------------------
extern "C" int gml_hessvec_batch(gml_problem *p, int formulation, int64_t nrows, const int64_t *nodes, const double *theta,
                                 const double *vec, int64_t ld, double *hv) {
    if (!p || !nodes || !theta || !vec || !hv) return fail(GML_EINVAL, "NULL argument");
    if (formulation < 0 || formulation > 2) return fail(GML_EINVAL, "unknown formulation %d", formulation);
    if (nrows <= 0) return fail(GML_EINVAL, "nrows must be positive");
    if (ld < p->P) return fail(GML_EINVAL, "ld %lld smaller than the %lld parameters per node", (long long)ld, (long long)p->P);
    for (int64_t r = 0; r < nrows; ++r)
        if (nodes[r] < 0 || nodes[r] >= p->n) return fail(GML_EINVAL, "node id %lld out of range", (long long)nodes[r]);
    HIPCHK(hipSetDevice(p->device));
    const int64_t Qp = p->d.Qp, P = p->P, Rp = round_up(nrows, 32);
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
    int rc = device_pass(p, rs, act, Th.data(), formulation, GML_PREC_I8X, true, fv.data(), Gi.data(), nullptr);
    if (rc) return rc;
    // 2. Hessian-vector pass: the rows of the direction through the same slots (vmap = identity)
    hipStream_t st = p->st;
    const int64_t W = p->ws_rows;
    std::memcpy(p->hTh, Vc.data(), sizeof(double) * nrows * Qp);
    HIPCHK(hipMemcpyAsync(p->dTheta, p->hTh, sizeof(double) * nrows * Qp, hipMemcpyHostToDevice, st));
    // control block of ALL rows: device_pass may have ended on a rescaled re-run of a subset (dense theta rows), which
    // leaves rowcol = -1 for the others and a shortened tile list
    {
        const int ng = (int)(Rp / 32);
        int npad = 0;
        for (int64_t r = 0; r < Rp; ++r) {
            p->hCtl[r] = (int)r; // slot = row
            p->hCtl[W + r] = r < nrows ? (int)nodes[r] : -1;
        }
        for (int g = 0; g < ng || (npad % 4); ++g, ++npad) p->hCtl[2 * W + g] = g < ng ? g : -1;
        HIPCHK(hipMemcpyAsync(p->dSrow, p->hCtl, sizeof(int) * (2 * W + npad), hipMemcpyHostToDevice, st));
    }
    gml::I8Pass a{};
    a.theta = p->dTheta;
    a.srow = p->dSrow;
    a.rowcol = p->dRowcol;
    a.groups = p->dGroups;
    a.ngroups = (int)(Rp / 32);
    a.slot0 = 0;
    a.slot1 = (int)Rp;
    a.form = formulation;
    a.want_grad = true;
    a.F = nullptr;
    a.G = p->dG;
    a.hv = 1;
    a.vmap = p->dSrow;
    std::string err;
    rc = gml::i8_pass(&p->i8ws, p->d, W, a, st, nullptr, &err);
    if (rc) return fail(rc, "%s", err.c_str());
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

// Test hook (not part of include/gml.h): the block-diagonal preconditioner of the matrix-free rows on caller-given tiles -- tile t is
// a T x T row-major symmetric block of which the leading m_t x m_t part counts; z_t = (s1 H_t - s2 g_t g_t^T)^-1 r_t, by the same two
// kernels the CG uses (launch_tile_inverse, then launch_tile_apply with the tiles' entries laid out consecutively in one row).
// tests/test_gpu_newton_solve.py.
extern "C" int gml_test_tile_precond(int T, int ntiles, const int *m, const double *tiles /* ntiles x T x T */, double s1, double s2,
                                     const double *g /* ntiles x T */, const double *r /* ntiles x T */, double *z_out /* ntiles x T */,
                                     int device) {
    if ((T != 64 && T != 128) || ntiles <= 0) return fail(GML_EINVAL, "bad tile size");
    HIPCHK(hipSetDevice(device));
    const size_t nt = (size_t)ntiles, ne = nt * T;
    std::vector<long long> hoff(nt);
    std::vector<int> wrow(nt, 0), fv(ne), live(1, 1);
    for (size_t t = 0; t < nt; ++t) hoff[t] = (long long)t * T * T;
    for (size_t e = 0; e < ne; ++e) fv[e] = (int)e; // tile t owns the columns [t T, (t + 1) T) of the one row
    double *dH = nullptr, *dS1 = nullptr, *dG = nullptr, *dR = nullptr, *dZ = nullptr;
    long long *dHoff = nullptr;
    int *dM = nullptr, *dWrow = nullptr, *dFv = nullptr, *dLive = nullptr;
    auto freeall = [&]() {
        void *ptrs[] = {dH, dS1, dG, dR, dZ, dHoff, dM, dWrow, dFv, dLive};
        for (void *q : ptrs)
            if (q) (void)dev_free(q);
    };
#define TCHK2(expr)                                                                              \
    do {                                                                                        \
        if ((expr) != hipSuccess) {                                                             \
            freeall();                                                                          \
            return fail(GML_EHIP, "%s failed: %s", #expr, hipGetErrorString(hipGetLastError())); \
        }                                                                                       \
    } while (0)
    TCHK2(dev_malloc(&dH, sizeof(double) * ne * T));
    TCHK2(dev_malloc(&dS1, sizeof(double)));
    TCHK2(dev_malloc(&dG, sizeof(double) * ne));
    TCHK2(dev_malloc(&dR, sizeof(double) * ne));
    TCHK2(dev_malloc(&dZ, sizeof(double) * ne));
    TCHK2(dev_malloc(&dHoff, sizeof(long long) * nt));
    TCHK2(dev_malloc(&dM, sizeof(int) * nt));
    TCHK2(dev_malloc(&dWrow, sizeof(int) * nt));
    TCHK2(dev_malloc(&dFv, sizeof(int) * ne));
    TCHK2(dev_malloc(&dLive, sizeof(int)));
    TCHK2(hipMemcpy(dH, tiles, sizeof(double) * ne * T, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dS1, &s1, sizeof(double), hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dG, g, sizeof(double) * ne, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dR, r, sizeof(double) * ne, hipMemcpyHostToDevice));
    TCHK2(hipMemset(dZ, 0, sizeof(double) * ne));
    TCHK2(hipMemcpy(dHoff, hoff.data(), sizeof(long long) * nt, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dM, m, sizeof(int) * nt, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dWrow, wrow.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dFv, fv.data(), sizeof(int) * ne, hipMemcpyHostToDevice));
    TCHK2(hipMemcpy(dLive, live.data(), sizeof(int), hipMemcpyHostToDevice));
    launch_tile_inverse(T, dH, dHoff, dM, dWrow, dS1, s2, dG, ntiles, nullptr);
    launch_tile_apply(T, dH, dFv, dM, dWrow, dLive, ntiles, (int64_t)ne, dR, dZ, nullptr);
    TCHK2(hipGetLastError());
    TCHK2(hipDeviceSynchronize());
    TCHK2(hipMemcpy(z_out, dZ, sizeof(double) * ne, hipMemcpyDeviceToHost));
#undef TCHK2
    freeall();
    return GML_OK;
}

// Test hook (not part of include/gml.h): the batched Newton solve on caller-given blocks -- A_r d_r = -pg_r for R symmetric positive
// definite m_r x m_r blocks (row-major, m_r <= cap <= 512), exactly as gml_learn's direction phase calls it.  tests/test_gpu_newton_solve.py.
static int test_newton_solve(int R, const int *m, int cap, const double *blocks, const double *pg, double s2, const double *g, double *d_out,
                             int device, const unsigned char *fix /* R x cap or NULL */, const double *dfix /* R x cap */) {
    HIPCHK(hipSetDevice(device));
    std::vector<long long> hoff((size_t)R + 1, 0);
    std::vector<int> mt((size_t)R);
    int maxm = 0;
    for (int r = 0; r < R; ++r) {
        if (m[r] < 0 || m[r] > cap || cap > 512) return fail(GML_EINVAL, "bad block size");
        mt[r] = (m[r] + 31) / 32;
        hoff[r + 1] = hoff[r] + (long long)mt[r] * 32 * mt[r] * 32;
        maxm = std::max(maxm, m[r]);
    }
    std::vector<double> H((size_t)std::max<long long>(hoff[R], 1), 0.0), s1((size_t)R, 1.0), gz((size_t)R * cap, 0.0);
    for (int r = 0; r < R; ++r) {
        const int hp = 32 * mt[r];
        for (int i = 0; i < m[r]; ++i)
            for (int j = 0; j < m[r]; ++j) H[(size_t)hoff[r] + (size_t)i * hp + j] = blocks[((size_t)r * cap + i) * cap + j];
        for (int i = m[r]; i < hp; ++i) H[(size_t)hoff[r] + (size_t)i * hp + i] = 1.0; // padding: identity
    }
    double *dH = nullptr, *dS1 = nullptr, *dG = nullptr, *dPg = nullptr, *dOut = nullptr, *dSd = nullptr, *dDfix = nullptr;
    long long *dHoff = nullptr;
    int *dMt = nullptr, *dM = nullptr, *dRedo = nullptr;
    uint8_t *dFix = nullptr;
    auto freeall = [&]() {
        void *ptrs[] = {dH, dS1, dG, dPg, dOut, dSd, dHoff, dMt, dM, dDfix, dRedo, dFix};
        for (void *q : ptrs)
            if (q) (void)dev_free(q);
    };
#define TCHK(expr)                                                           \
    do {                                                                     \
        if ((expr) != hipSuccess) {                                          \
            freeall();                                                       \
            return fail(GML_EHIP, "%s failed: %s", #expr, hipGetErrorString(hipGetLastError())); \
        }                                                                    \
    } while (0)
    TCHK(dev_malloc(&dH, sizeof(double) * H.size()));
    TCHK(dev_malloc(&dS1, sizeof(double) * R));
    TCHK(dev_malloc(&dG, sizeof(double) * R * cap));
    TCHK(dev_malloc(&dPg, sizeof(double) * R * cap));
    TCHK(dev_malloc(&dOut, sizeof(double) * R * cap));
    TCHK(dev_malloc(&dSd, sizeof(double) * R));
    TCHK(dev_malloc(&dHoff, sizeof(long long) * (R + 1)));
    TCHK(dev_malloc(&dMt, sizeof(int) * R));
    TCHK(dev_malloc(&dM, sizeof(int) * R));
    TCHK(hipMemcpy(dH, H.data(), sizeof(double) * H.size(), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dS1, s1.data(), sizeof(double) * R, hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dG, g ? g : gz.data(), sizeof(double) * R * cap, hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dPg, pg, sizeof(double) * R * cap, hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dHoff, hoff.data(), sizeof(long long) * (R + 1), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dMt, mt.data(), sizeof(int) * R, hipMemcpyHostToDevice));
    TCHK(hipMemcpy(dM, m, sizeof(int) * R, hipMemcpyHostToDevice));
    TCHK(hipMemset(dOut, 0, sizeof(double) * R * cap));
    if (fix) { // some entries fixed from the start
        TCHK(dev_malloc(&dFix, (size_t)R * cap));
        TCHK(dev_malloc(&dDfix, sizeof(double) * R * cap));
        TCHK(hipMemcpy(dFix, fix, (size_t)R * cap, hipMemcpyHostToDevice));
        TCHK(hipMemcpy(dDfix, dfix, sizeof(double) * R * cap, hipMemcpyHostToDevice));
    }
    launch_newton_solve(dH, dHoff, dMt, dM, dS1, s2, dG, dPg, R, cap, dOut, dSd, nullptr, maxm, nullptr, dFix, dDfix);
    TCHK(hipGetLastError());
    TCHK(hipDeviceSynchronize());
    TCHK(hipMemcpy(d_out, dOut, sizeof(double) * R * cap, hipMemcpyDeviceToHost));
#undef TCHK
    freeall();
    return GML_OK;
}

extern "C" int gml_test_newton_solve(int R, const int *m, int cap, const double *blocks /* R x cap x cap, block r uses its leading m_r x m_r */,
                                     const double *pg /* R x cap */, double s2, const double *g /* R x cap or NULL */, double *d_out /* R x cap */,
                                     int device) {
    return test_newton_solve(R, m, cap, blocks, pg, s2, g, d_out, device, nullptr, nullptr);
}
// ... and the re-solve with some entries fixed (fix != 0: d = dfix there; the others solve A_ff d_f = -pg_f - A_fx dfix_x)
extern "C" int gml_test_newton_solve_fixed(int R, const int *m, int cap, const double *blocks, const double *pg, double s2, const double *g,
                                           const unsigned char *fix, const double *dfix, double *d_out, int device) {
    return test_newton_solve(R, m, cap, blocks, pg, s2, g, d_out, device, fix, dfix);
}

// Experiment hook (not part of include/gml.h): bytes [off, off + bytes) of the V limb planes of the handle's int8 workspace.  The
// timing builds of the forward kernels (scripts/build_variant.sh ... -DABL_TIMING) leave per-workgroup timestamps there.
extern "C" int gml_debug_read_vq(gml_problem *p, int64_t off, int64_t bytes, void *out) {
    if (!p || !out) return fail(GML_EINVAL, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    const int8_t *vq = nullptr;
    int64_t total = 0;
    gml::i8_vq_buffer(p->i8ws, &vq, &total, p->d);
    if (!vq || off < 0 || off + bytes > total) return fail(GML_EINVAL, "range outside the %lld bytes of V planes", (long long)total);
    HIPCHK(hipStreamSynchronize(p->st));
    HIPCHK(hipMemcpy(out, vq + off, (size_t)bytes, hipMemcpyDeviceToHost));
    return GML_OK;
}
