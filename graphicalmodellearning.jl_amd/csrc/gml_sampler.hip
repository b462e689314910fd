// On-device exact sampler: the step BEFORE the learn() path.
//
// The reference's `sample(gm, N)` enumerates all 2^n states, weighs them with exp(sum_t w_t prod_{i in t} s_i)
// (src/sampling.jl:26-30, 60-64) and draws N of them (:34-57, :67-88).  Here the same exact scheme is
// applied per connected component ("block") of the term hypergraph, for any interaction order, so that models far
// beyond n ~ 25 can be sampled as long as every block has at most 22 spins, and the +-1 samples are
// written straight into HBM in the layout gml_problem_create_spins expects (no host histogram, no
// PCIe upload).  Random numbers: a counter-based splitmix64 hash of (seed, block, sample).
#include "../../include/gml.h"
#include "gml_dev.h"

namespace gml {

// energies of all 2^sb states of one block: e(state) = sum_t w_t prod_{i in t} s_i (weigh_proba, sampling.jl:60-64;
// the pairwise form 1/2 s^T A s + h^T s of :26-30 is the same sum over the pair and field terms).  Bit i of
// `state` = spin i (int_to_spin, sampling.jl:11-14: little-endian bits -> +-1).  A term is a bit mask over the
// block's spins: prod s_i = (-1)^(number of -1 spins in the term) = 1 - 2 (popcount(mask & ~state) & 1).
__global__ __launch_bounds__(256) void k_block_energies(const unsigned *__restrict__ masks, const double *__restrict__ wts,
                                                        int nt, int sb, double *__restrict__ en) {
    __shared__ unsigned sm[1024];
    __shared__ double sw[1024];
    const int64_t st = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned nst = ~(unsigned)st;
    double e = 0.0;
    for (int t0 = 0; t0 < nt; t0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024 && t0 + t < nt; t += 256) {
            sm[t] = masks[t0 + t];
            sw[t] = wts[t0 + t];
        }
        __syncthreads();
        const int m = nt - t0 < 1024 ? nt - t0 : 1024;
        for (int t = 0; t < m; ++t) e += (__popc(sm[t] & nst) & 1) ? -sw[t] : sw[t];
    }
    if (st < ((int64_t)1 << sb)) en[st] = e;
}

// single-workgroup max + exclusive->inclusive scan of exp(en - max) into cdf (normalised to cdf[last] = 1)
__global__ __launch_bounds__(1024) void k_block_cdf(const double *__restrict__ en, int64_t ns, double *__restrict__ cdf) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    double mx = -INFINITY;
    for (int64_t i = tid; i < ns; i += 1024) mx = fmax(mx, en[i]);
    red[tid] = mx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    // each thread owns a contiguous chunk
    const int64_t per = (ns + 1023) / 1024, b0 = tid * per, b1 = b0 + per < ns ? b0 + per : ns;
    double sum = 0.0;
    for (int64_t i = b0; i < b1; ++i) sum += exp(en[i] - mx);
    red[tid] = sum;
    __syncthreads();
    __shared__ double total;
    if (tid == 0) { // 1024 partial sums: serial exclusive scan
        double acc = 0.0;
        for (int t = 0; t < 1024; ++t) {
            const double v = red[t];
            red[t] = acc;
            acc += v;
        }
        total = acc;
    }
    __syncthreads();
    double acc = red[tid];
    for (int64_t i = b0; i < b1; ++i) {
        acc += exp(en[i] - mx);
        cdf[i] = acc;
    }
    __syncthreads();
    const double inv = 1.0 / total;
    for (int64_t i = b0; i < b1; ++i) cdf[i] *= inv;
}

__device__ __forceinline__ double u01(unsigned long long seed, unsigned long long block, unsigned long long k) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (k + 1) + 0xD1B54A32D192ED03ull * (block + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// one thread per sample: CDF inversion by binary search, spins written sample-major
__global__ __launch_bounds__(256) void k_block_draw(const double *__restrict__ cdf, int64_t ns, int sb,
                                                    const int *__restrict__ members /* sb global spin ids */,
                                                    int64_t N, int64_t n, unsigned long long seed, int block,
                                                    int8_t *__restrict__ S /* N x n */) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const double u = u01(seed, (unsigned long long)block, (unsigned long long)k);
    int64_t lo = 0, hi = ns - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid;
        else lo = mid + 1;
    }
    for (int t = 0; t < sb; ++t) S[k * n + members[t]] = (lo >> t) & 1 ? (int8_t)1 : (int8_t)-1;
}

// ------------------------------------------------------------------------------------------
// Glauber (single-spin heat-bath) dynamics for models whose components are too large to enumerate: N
// independent chains, one thread each, `sweeps` sequential-scan sweeps from a uniformly random start; the
// final states are the samples.  (Beyond the reference, whose "Gibbs" sampler is exact enumeration; SURVEY.md
// 8(f) #2 names this as the follow-up.)  State spin-major St [n][Np] so that a wave's reads of one spin are
// contiguous; the model as incidence lists: spin i is in incidences [ioff[i], ioff[i+1]), incidence e has
// weight iw[e] and the other spins of its term oth[ooff[e] .. ooff[e+1]).
//   field_i = sum_e iw[e] prod_{j in others(e)} s_j,   P(s_i = +1 | rest) = 1 / (1 + exp(-2 field_i))
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_glauber(const int *__restrict__ ioff, const double *__restrict__ iw,
                                                 const int *__restrict__ ooff, const int *__restrict__ oth, int64_t n,
                                                 int64_t N, int64_t Np, int sweeps, unsigned long long seed,
                                                 int8_t *__restrict__ St) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    for (int64_t i = 0; i < n; ++i) St[i * Np + k] = u01(seed, 0xFFFFFFFFull, (unsigned long long)(k * n + i)) < 0.5 ? (int8_t)1 : (int8_t)-1;
    for (int sw = 0; sw < sweeps; ++sw) {
        for (int64_t i = 0; i < n; ++i) {
            double field = 0.0;
            for (int e = ioff[i]; e < ioff[i + 1]; ++e) {
                int pr = 1;
                for (int a = ooff[e]; a < ooff[e + 1]; ++a) pr *= (int)St[(int64_t)oth[a] * Np + k];
                field += iw[e] * (double)pr;
            }
            const double pup = 1.0 / (1.0 + exp(-2.0 * field));
            const double u = u01(seed, (unsigned long long)sw, (unsigned long long)(k * n + i));
            St[i * Np + k] = u < pup ? (int8_t)1 : (int8_t)-1;
        }
    }
}

void launch_glauber(const int *dioff, const double *diw, const int *dooff, const int *doth, int64_t n, int64_t N, int64_t Np,
                    int sweeps, unsigned long long seed, int8_t *dSt, hipStream_t st) {
    hipLaunchKernelGGL(k_glauber, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dioff, diw, dooff, doth, n, N, Np, sweeps, seed,
                       dSt);
}

void launch_block_sampler(const unsigned *dmasks, const double *dwts, int nt, int sb, const int *dmembers, int64_t N, int64_t n,
                          unsigned long long seed, int block, double *den, double *dcdf, int8_t *dS, hipStream_t st) {
    const int64_t ns = (int64_t)1 << sb;
    hipLaunchKernelGGL(k_block_energies, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, dmasks, dwts, nt, sb, den);
    hipLaunchKernelGGL(k_block_cdf, dim3(1), dim3(1024), 0, st, den, ns, dcdf);
    hipLaunchKernelGGL(k_block_draw, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dcdf, ns, sb, dmembers, N, n, seed, block,
                       dS);
}

} // namespace gml
