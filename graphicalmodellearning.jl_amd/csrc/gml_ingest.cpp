// libgml_hip, host side: building a handle from host data -- the one-pass bit packer and its pinned upload pipeline, the
// packed form, the raw-upload + device-conversion route (GraphicalModelLearning.jl:73, :76-81: what every learn() does first).
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
static void parallel_for(int64_t n, const std::function<void(int64_t)> &fn) { gml_parallel_for(n, fn); }
static int64_t round_up(int64_t a, int64_t b) { return gml_round_up(a, b); }

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
        Qf += gml_binom(p->n, q);
    }
    p->qoff[fo + 1] = Qf;
    p->P = 0;
    for (int q = 0; q <= p->order - 1; ++q) p->P += gml_binom(p->n - 1, q);
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
            } while (gml_next_comb(idx, p->n));
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

int gml_check_create_args(int64_t K, int64_t n, int order, int64_t node0, int64_t node1, int device) {
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

gml_problem *gml_new_problem(int64_t K, int64_t n, double M, int order, int64_t node0, int64_t node1, int device) {
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
int gml_create_from_device_bytes(gml_problem *p, int8_t *dbytes, bool spin_major, int64_t ld, const double *counts, gml_problem **out,
                                 bool dedupe) {
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
        const std::string keep = gml_last_error();
        gml_problem_destroy(p);
        return fail(rc, "%s", keep.c_str());
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
        const std::string keep = gml_last_error();
        for (auto &q : parts) {
            if (q) gml_problem_destroy(q);
            q = nullptr;
        }
        return fail(code, "%s", keep.c_str());
    };
    WeightInfo wi;
    double t_alloc = 0, t_weights = 0;
    for (size_t g = 0; g < G; ++g) {
        parts[g] = gml_new_problem(K, n, M, order, ranges[g].first, ranges[g].second, devices[g]);
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
        const int rc = gml_check_create_args(hv.K, hv.n, order, ranges[g].first, ranges[g].second, devices[g]);
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
    int rc = gml_check_create_args(K, n, order, node0, node1, device);
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
    rc = gml_check_create_args(K, n, order, node0, node1, device);
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
    gml_problem *p = gml_new_problem(K, n, Msum, order, node0, node1, device);
    rc = gml_create_from_device_bytes(p, dS, col_major != 0, K, counts.data(), out);
    if (rc == GML_OK) {
        (*out)->t_ingest[0] = 0.0;
        (*out)->t_ingest[1] = t_up - t_begin; // the raw upload
        (*out)->t_ingest[3] = now_s() - t_begin;
    }
    return rc;
}

