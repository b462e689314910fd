// gml_multi_*: one problem on several GPUs of a node from ONE caller (the reference's host language has no
// torch.distributed: a Julia `learn(samples, RISE(), HIP(devices = 0:7))` lands here).
//
// The node-wise problems are independent (the loop GraphicalModelLearning.jl:161 touches only row u, :181), so GPU g
// owns the contiguous node range [g n / G, (g+1) n / G): its own gml_problem handle (the sample bits are replicated,
// 1/8 byte per entry), its own host thread, no communication while solving.  The learned row blocks are gathered
//   * into the caller's host matrix directly (every thread writes its rows), and/or
//   * into a device-resident copy of the full matrix on EVERY GPU by one RCCL all-gather over xGMI (gml_multi_learn's
//     dev_out), for callers that keep working on the devices.  librccl is loaded at run time (dlopen): the library has no
//     link-time dependency on it, and falls back to peer copies when it is absent or the device list repeats a GPU.
#include "gml_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <thread>

namespace {

// the few RCCL entry points used, resolved at run time
struct Rccl {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl() {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD); // the copy a host framework has already loaded, if any
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(h, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather;
    }
};
Rccl &rccl() {
    static Rccl r;
    return r;
}
constexpr int kNcclFloat64 = 8; // ncclDouble / ncclFloat64 (nccl.h)

} // namespace

struct gml_multi {
    std::vector<gml_problem *> part;
    std::vector<int> device;
    int64_t n = 0, K = 0, P = 0;
    double M = 0;
    std::vector<Rccl::comm_t> comm; // one per part when RCCL is usable for this device list
    char gather_kind[32] = "host";
    std::string rccl_note;          // how the communicators came about (gml_multi_diag): a first 8-GPU run must be diagnosable
    std::vector<gml_stats> last;    // per-part statistics of the last gml_multi_learn
};

extern "C" void gml_multi_destroy(gml_multi *m) {
    if (!m) return;
    for (size_t g = 0; g < m->comm.size(); ++g)
        if (m->comm[g]) {
            (void)hipSetDevice(m->device[g]);
            (void)rccl().CommDestroy(m->comm[g]);
        }
    for (gml_problem *p : m->part) gml_problem_destroy(p);
    delete m;
}

extern "C" int gml_multi_create(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, int order,
                                const int *devices, int ndev, gml_multi **out) {
    if (!out) return fail(GML_EINVAL, "out is NULL");
    *out = nullptr;
    if (!devices || ndev < 1) return fail(GML_EINVAL, "empty device list");
    if (n < ndev) return fail(GML_EINVAL, "more devices (%d) than nodes (%lld)", ndev, (long long)n);
    gml_multi *m = new gml_multi();
    m->device.assign(devices, devices + ndev);
    // contiguous node ranges; the histogram is packed once and its bits are copied to every device
    std::vector<std::pair<int64_t, int64_t>> ranges;
    for (int g = 0; g < ndev; ++g) ranges.emplace_back((int64_t)g * n / ndev, (int64_t)(g + 1) * n / ndev);
    const int rc = gml_create_parts(samples, dtype, K, n, ld, col_major, order, ranges, m->device, m->part);
    if (rc) {
        const std::string keep = gml_last_error();
        m->part.clear();
        gml_multi_destroy(m);
        return fail(rc, "%s", keep.c_str());
    }
    (void)gml_problem_info(m->part[0], &m->n, &m->K, &m->M, &m->P, nullptr, nullptr);
    // RCCL communicators, when every part sits on its own GPU
    std::vector<int> uniq(m->device);
    std::sort(uniq.begin(), uniq.end());
    const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
    if (!rccl().ok) {
        m->rccl_note = "librccl not loadable (dlopen): gathers use peer copies";
    } else if (!distinct) {
        m->rccl_note = "the device list repeats a GPU: gathers use peer copies (RCCL refuses two ranks on one device)";
    } else {
        m->comm.assign((size_t)ndev, nullptr);
        const int e = rccl().CommInitAll(m->comm.data(), ndev, devices);
        if (e != 0) {
            m->comm.clear(); // fall back to peer copies
            m->rccl_note = std::string("ncclCommInitAll failed (") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?") + "): gathers use peer copies";
        } else {
            m->rccl_note = "RCCL communicators ready: " + std::to_string(ndev) + " ranks, one per GPU";
        }
    }
    *out = m;
    return GML_OK;
}

extern "C" int gml_multi_info(const gml_multi *m, int64_t *n, int64_t *K, double *M, int64_t *P, int *ndev, char *gather_kind /* >= 32 bytes or NULL */) {
    if (!m) return fail(GML_EINVAL, "handle is NULL");
    if (n) *n = m->n;
    if (K) *K = m->K;
    if (M) *M = m->M;
    if (P) *P = m->P;
    if (ndev) *ndev = (int)m->part.size();
    if (gather_kind) std::strcpy(gather_kind, m->gather_kind);
    return GML_OK;
}

// Diagnostics of the collective path (not needed for results): how the RCCL communicators of this handle came about, and which
// path the last dev_out gather took.
extern "C" int gml_multi_diag(const gml_multi *m, char *buf, int nbuf) {
    if (!m || !buf || nbuf < 1) return fail(GML_EINVAL, "NULL argument");
    const std::string s = m->rccl_note + "; last gather: " + m->gather_kind;
    std::strncpy(buf, s.c_str(), (size_t)nbuf - 1);
    buf[nbuf - 1] = 0;
    return GML_OK;
}

// Test hook (not part of include/gml.h): build RCCL communicators for this handle's device list even when it repeats a GPU, so
// that the error path of the collective set-up can be exercised on a 1-GPU box: RCCL refuses, and the refusal must come back as a
// clean GML_EHIP carrying ncclGetErrorString's text.  (Should an RCCL build accept the list, the communicators are kept and the
// next dev_out gather takes the all-gather path.)
extern "C" int gml_test_multi_force_rccl(gml_multi *m) {
    if (!m) return fail(GML_EINVAL, "handle is NULL");
    if (!rccl().ok) return fail(GML_EUNSUPPORTED, "librccl not loadable");
    if (!m->comm.empty()) return GML_OK;
    const int ndev = (int)m->device.size();
    m->comm.assign((size_t)ndev, nullptr);
    const int e = rccl().CommInitAll(m->comm.data(), ndev, m->device.data());
    if (e != 0) {
        m->comm.clear();
        m->rccl_note = std::string("ncclCommInitAll failed (") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?") + ")";
        return fail(GML_EHIP, "ncclCommInitAll over %d ranks failed: %s", ndev, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
    }
    m->rccl_note = "RCCL communicators forced on a device list that repeats a GPU";
    return GML_OK;
}

extern "C" int gml_multi_part_stats(const gml_multi *m, gml_stats *parts) {
    if (!m || !parts) return fail(GML_EINVAL, "NULL argument");
    for (size_t g = 0; g < m->part.size(); ++g) parts[g] = g < m->last.size() ? m->last[g] : gml_stats{};
    return GML_OK;
}

namespace {
// streams of the gather, destroyed on every path; the caller's current device is restored
struct GatherStreams {
    std::vector<hipStream_t> st;
    const std::vector<int> &dev;
    int caller_dev = 0;
    explicit GatherStreams(const std::vector<int> &d) : st(d.size(), nullptr), dev(d) { (void)hipGetDevice(&caller_dev); }
    ~GatherStreams() {
        for (size_t g = 0; g < st.size(); ++g)
            if (st[g]) {
                (void)hipSetDevice(dev[g]);
                (void)hipStreamSynchronize(st[g]);
                (void)hipStreamDestroy(st[g]);
            }
        (void)hipSetDevice(caller_dev);
    }
};
} // namespace

extern "C" int gml_multi_learn(gml_multi *m, int formulation, double regularizer_c, const gml_opts *opts, double *out, double *kkt,
                               gml_stats *stats, double **dev_out) {
    if (!m || (!out && !dev_out)) return fail(GML_EINVAL, "NULL argument");
    const int G = (int)m->part.size();
    const int64_t n = m->n, P = m->P;
    if (dev_out)
        for (int g = 0; g < G; ++g)
            if (!dev_out[g]) return fail(GML_EINVAL, "dev_out[%d] is NULL", g);
    std::vector<int> rc((size_t)G, GML_OK);
    std::vector<std::string> msg((size_t)G);
    m->last.assign((size_t)G, gml_stats{});
    std::vector<gml_stats> &st = m->last;
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g] {
            const int64_t n0 = (int64_t)g * n / G, n1 = (int64_t)(g + 1) * n / G;
            // Each part writes its rows where they belong: with dev_out, straight into its block of the device-resident matrix
            // on its own GPU (device to device inside gml_learn: the block sits where the all-gather expects it), and from
            // there once to the caller's host matrix; without, into the disjoint row block of the host matrix.
            double *dst = dev_out ? dev_out[g] + n0 * P : out + n0 * P;
            rc[g] = gml_learn(m->part[g], formulation, regularizer_c, opts, dst, kkt ? kkt + n0 : nullptr, &st[g]);
            if (rc[g]) msg[g] = gml_last_error();
            if ((rc[g] == GML_OK || rc[g] == GML_ENOTCONV) && dev_out && out) {
                if (hipSetDevice(m->device[g]) != hipSuccess ||
                    hipMemcpy(out + n0 * P, dst, sizeof(double) * (n1 - n0) * P, hipMemcpyDeviceToHost) != hipSuccess) {
                    rc[g] = GML_EHIP;
                    msg[g] = std::string("download of the learned rows failed: ") + hipGetErrorString(hipGetLastError());
                }
            }
        });
    for (auto &t : th) t.join();
    if (stats) { // totals over the parts, the wall-clock entries of the slowest part (per part: gml_multi_part_stats)
        std::memset(stats, 0, sizeof *stats);
        for (int g = 0; g < G; ++g) {
            stats->iterations = std::max(stats->iterations, st[g].iterations);
            stats->passes += st[g].passes;
            stats->forward_passes += st[g].forward_passes;
            stats->hessian_passes += st[g].hessian_passes;
            stats->node_evals += st[g].node_evals;
            stats->hv_evals += st[g].hv_evals;
            stats->max_kkt = std::max(stats->max_kkt, st[g].max_kkt);
            stats->lambda = st[g].lambda;
            stats->t_pack = std::max(stats->t_pack, st[g].t_pack);
            stats->t_pass = std::max(stats->t_pass, st[g].t_pass);
            stats->t_hess = std::max(stats->t_hess, st[g].t_hess);
            stats->t_host = std::max(stats->t_host, st[g].t_host);
            stats->t_total = std::max(stats->t_total, st[g].t_total);
            stats->not_converged += st[g].not_converged;
            stats->polished |= st[g].polished;
        }
    }
    int worst = GML_OK;
    std::string wmsg;
    for (int g = 0; g < G; ++g)
        if (rc[g] && (worst == GML_OK || rc[g] != GML_ENOTCONV)) {
            worst = rc[g];
            wmsg = msg[g];
        }
    if (worst != GML_OK && worst != GML_ENOTCONV) return fail(worst, "%s", wmsg.c_str());
    if (dev_out) {
        // the full matrix on every GPU: one in-place all-gather of the row blocks over xGMI (RCCL), or peer copies
        const bool even = n % G == 0;
        GatherStreams gs(m->device);
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipSetDevice(m->device[g]));
            HIPCHK(hipStreamCreate(&gs.st[g]));
        }
        if (!m->comm.empty() && even) {
            std::strcpy(m->gather_kind, "rccl-allgather");
            int e = rccl().GroupStart();
            if (e == 0) {
                for (int g = 0; g < G && e == 0; ++g)
                    e = rccl().AllGather(dev_out[g] + (int64_t)g * (n / G) * P, dev_out[g], (size_t)(n / G * P), kNcclFloat64, m->comm[g], gs.st[g]);
                const int e2 = rccl().GroupEnd(); // always closed: an open group would hang the process's later RCCL calls
                if (e == 0) e = e2;
            }
            if (e != 0) return fail(GML_EHIP, "RCCL all-gather failed: %s", rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
        } else {
            std::strcpy(m->gather_kind, "peer-copy");
            for (int g = 0; g < G; ++g)
                for (int h = 0; h < G; ++h) {
                    if (h == g || dev_out[h] == dev_out[g]) continue;
                    const int64_t n0 = (int64_t)h * n / G, n1 = (int64_t)(h + 1) * n / G;
                    HIPCHK(hipSetDevice(m->device[g]));
                    HIPCHK(hipMemcpyPeerAsync(dev_out[g] + n0 * P, m->device[g], dev_out[h] + n0 * P, m->device[h], sizeof(double) * (n1 - n0) * P, gs.st[g]));
                }
        }
        for (int g = 0; g < G; ++g) {
            HIPCHK(hipSetDevice(m->device[g]));
            HIPCHK(hipStreamSynchronize(gs.st[g]));
        }
    }
    if (worst == GML_ENOTCONV) return fail(GML_ENOTCONV, "%s", wmsg.c_str());
    return GML_OK;
}
