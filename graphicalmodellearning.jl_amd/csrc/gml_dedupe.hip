// Histogramming on the device: the `countmap` of the reference's sampler (sampling.jl:52: one row per DISTINCT
// configuration, column 1 = how often it was drawn) for samples that never leave HBM.
//
// For n <= 64 spins a configuration is one 64-bit key (bit i set <=> spin i is -1): keys of the N samples -> radix sort
// over the n significant bits -> run-length encoding = the distinct configurations, ascending, with their counts
// (hipCUB / rocPRIM device primitives; the sort dominates: a few passes over 8 N bytes).  The handle is then built from the
// K' distinct rows instead of the N draws: K' <= min(N, 2^n), so a 9-spin model sampled 1e8 times becomes 512 rows.
#include "../../include/gml.h"
#include "gml_dev.h"

#include <hipcub/hipcub.hpp>

namespace gml {

// key of sample k from +-1 bytes: sample-major S [N][n] or spin-major S [n][ld]
__global__ __launch_bounds__(256) void k_make_keys(const int8_t *__restrict__ S, int spin_major, int64_t ld, int64_t N, int n,
                                                   unsigned long long *__restrict__ keys) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    unsigned long long v = 0;
    for (int i = 0; i < n; ++i) {
        const int8_t s = spin_major ? S[(int64_t)i * ld + k] : S[k * n + i];
        v |= (unsigned long long)(s < 0) << i;
    }
    keys[k] = v;
}

// sign words of the distinct configurations: Sb [n][wpr], bit j of word w <-> row 32 w + j (gml_bits.h)
__global__ __launch_bounds__(256) void k_bits_from_keys(const unsigned long long *__restrict__ keys, int64_t K, int n, int64_t wpr,
                                                        unsigned *__restrict__ Sb) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (w >= wpr || i >= n) return;
    unsigned v = 0;
    for (int j = 0; j < 32; ++j) {
        const int64_t k = w * 32 + j;
        if (k < K) v |= (unsigned)((keys[k] >> i) & 1ull) << j;
    }
    Sb[(int64_t)i * wpr + w] = v;
}

#define DCHK(expr)                                                               \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            if (err) *err = std::string(#expr) + " failed: " + hipGetErrorString(e_); \
            cleanup();                                                           \
            return e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP;            \
        }                                                                        \
    } while (0)

// Distinct configurations of the N samples held as +-1 bytes on the device: *dkeys_out [K'] ascending keys, *dcounts_out [K']
// multiplicities (both device, owned by the caller), *K_out = K'.
int dedupe_samples(const int8_t *dS, bool spin_major, int64_t ld, int64_t N, int64_t n, hipStream_t st, unsigned long long **dkeys_out,
                   int **dcounts_out, int64_t *K_out, std::string *err) {
    unsigned long long *k0 = nullptr, *k1 = nullptr, *uniq = nullptr;
    int *cnt = nullptr, *nruns = nullptr;
    void *tmp = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = {k0, k1, uniq, cnt, nruns, tmp};
        for (void *q : ptrs)
            if (q) (void)dev_free(q);
    };
    DCHK(dev_malloc(&k0, sizeof(unsigned long long) * N));
    DCHK(dev_malloc(&k1, sizeof(unsigned long long) * N));
    hipLaunchKernelGGL(k_make_keys, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dS, spin_major ? 1 : 0, ld, N, (int)n, k0);
    size_t tb = 0;
    DCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, k0, k1, (int)N, 0, (int)n, st));
    DCHK(dev_malloc(&tmp, tb ? tb : 1));
    DCHK(hipcub::DeviceRadixSort::SortKeys(tmp, tb, k0, k1, (int)N, 0, (int)n, st));
    (void)dev_free(tmp);
    tmp = nullptr;
    // k0 is free again: it receives the distinct keys (at most N of them)
    uniq = k0;
    k0 = nullptr;
    DCHK(dev_malloc(&cnt, sizeof(int) * N));
    DCHK(dev_malloc(&nruns, sizeof(int)));
    tb = 0;
    DCHK(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb, k1, uniq, cnt, nruns, (int)N, st));
    DCHK(dev_malloc(&tmp, tb ? tb : 1));
    DCHK(hipcub::DeviceRunLengthEncode::Encode(tmp, tb, k1, uniq, cnt, nruns, (int)N, st));
    int hruns = 0;
    DCHK(hipMemcpyAsync(&hruns, nruns, sizeof(int), hipMemcpyDeviceToHost, st));
    DCHK(hipStreamSynchronize(st));
    *dkeys_out = uniq;
    *dcounts_out = cnt;
    *K_out = hruns;
    uniq = nullptr;
    cnt = nullptr;
    cleanup();
    return GML_OK;
}

void launch_bits_from_keys(const unsigned long long *dkeys, int64_t K, int64_t n, int64_t Kp, unsigned *Sb, hipStream_t st) {
    const int64_t wpr = Kp / 32;
    hipLaunchKernelGGL(k_bits_from_keys, dim3((unsigned)((wpr + 255) / 256), (unsigned)n), dim3(256), 0, st, dkeys, K, (int)n, wpr, Sb);
}

} // namespace gml
