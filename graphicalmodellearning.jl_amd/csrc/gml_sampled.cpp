// libgml_hip, host side: handles whose samples are drawn on the device -- the step before the path (src/sampling.jl:34-106):
// exact block sampling, Glauber chains, histogramming of the draws.
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

static int64_t round_up(int64_t a, int64_t b) { return gml_round_up(a, b); }

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
    return gml_create_from_device_bytes(p, dS, false, 0, nullptr, out, dedupe);
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
    return gml_create_from_device_bytes(p, dSt, true, Np, nullptr, out, dedupe); // the chains' final states, spin-major
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

