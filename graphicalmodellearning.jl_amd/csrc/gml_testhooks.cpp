// libgml_hip: test and experiment hooks (not part of include/gml.h).
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

namespace gml {
double g_hv_sparse_ratio = 0.3; // (config 5 at the default regulariser: 25.6 s never, 23.6 s at 0.3 and at 0.6)
std::atomic<long long> g_hv_sparse_calls{0}; // (every part of a gml_multi_learn counts from its own host thread)
double g_tune[GML_NTUNE] = {};               // experiment knobs of the solver, 0 = the built-in rule (gml_solver.h)
}
// the solver's switch between the GEMM form and the entry-by-entry form of a Hessian-vector pass (gml_solver.h); returns the old value
extern "C" long long gml_test_hv_sparse_calls(void) { return g_hv_sparse_calls.load(); }
// experiment knob `id` of the solver (gml_solver.h: GML_TUNE_*); returns the old value.  Not part of include/gml.h; set between solves only
extern "C" double gml_test_tune(int id, double value) {
    if (id < 0 || id >= GML_NTUNE) return NAN;
    const double old = g_tune[id];
    g_tune[id] = value;
    return old;
}
extern "C" double gml_test_hv_sparse_ratio(double ratio) {
    const double old = g_hv_sparse_ratio;
    g_hv_sparse_ratio = ratio;
    return old;
}


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

