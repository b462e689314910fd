// Result assembly of the multi-body learn() on the device: what the reference does after its node loop
// (GraphicalModelLearning.jl:129-132 `reconstruction[inter] = ...` per node, :135-149 group by sorted key + `mean`, :151
// FactorGraph(order, n, :spin, reconstruction)) without a key table and without a host loop.
//
// The n solved rows (n x P doubles, row u in the key order of :94-104 = gml_multi_keys(u)) become ONE array of weights in the
// order the reference itself lists a model's terms in (models.jl:61,72: sort by (length(key), key)):
//   symmetrised     every ascending key S, |S| <= order: sizes 1..order, lexicographic within a size; C(n,1) + ... + C(n,order)
//                   values, value = mean over u in S of row u's entry for the key (u, S \ {u})      (:135-149)
//   unsymmetrised   every key (u, S'), S' an ascending subset of the other spins: by size, then u, then S'; n P values   (:129-132)
// A key <-> its position is closed-form (combinatorial number system, lexicographic): the kernel unranks its output index,
// ranks S \ {u} among the (|S|-1)-subsets of the n-1 other spins to find row u's slot, and adds the |S| entries in ascending
// u.  C5 (n = 512, order 3): 67.0 M row entries (536 MB) -> 22.5 M terms (179 MB), one launch.
#include "gml_internal.h"

#include <algorithm>
#include <cstring>

namespace {

constexpr int MAXORD = 8; // (an order-9 model with n >= 32 has > 2^24 parameters per node; the statistics side stops far earlier)

struct TermPlan {
    int64_t n;
    int order;
    int pairwise; // order 2: the rows keep the pairwise slot layout of the C ABI (slot i <-> spin i, slot u = field, :162)
    int64_t off[MAXORD + 2];  // symmetrised: first position of the size-s keys, off[order + 1] = total
    int64_t uoff[MAXORD + 2]; // unsymmetrised: the same
    int64_t roff[MAXORD + 2]; // first slot of the size-s keys within a node's row (multi-body layout): roff[1] = 0
    int64_t rcnt[MAXORD + 2]; // C(n-1, s-1): size-s keys of one node
};

__host__ __device__ inline int64_t binom(int64_t m, int q) {
    if (q < 0 || q > m) return 0;
    int64_t r = 1;
    for (int i = 1; i <= q; ++i) r = r * (m - q + i) / i; // exact at every step: r = C(m - q + i, i)
    return r;
}

// the q-subset of {0..m-1} of lexicographic rank r -> c[0..q)
__host__ __device__ inline void unrank_lex(int64_t r, int64_t m, int q, int32_t *c) {
    int64_t base = 0; // the items below base are spent
    for (int i = 0; i < q; ++i) {
        const int rem = q - i;
        const int64_t mm = m - base;
        const int64_t tot = binom(mm, rem);
        // subsets of the remaining items whose first element lies below base + a: tot - C(mm - a, rem); the largest a with that <= r
        int64_t lo = 0, hi = mm - rem;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (tot - binom(mm - mid, rem) <= r) lo = mid;
            else hi = mid - 1;
        }
        r -= tot - binom(mm - lo, rem);
        c[i] = (int32_t)(base + lo);
        base += lo + 1;
    }
}

// lexicographic rank of the ascending q-subset c of {0..m-1}
__host__ __device__ inline int64_t rank_lex(const int32_t *c, int q, int64_t m) {
    int64_t r = 0, base = 0;
    for (int i = 0; i < q; ++i) {
        r += binom(m - base, q - i) - binom(m - c[i], q - i);
        base = (int64_t)c[i] + 1;
    }
    return r;
}

// slot of the key (u, S \ {u}) in row u; S = c[0..s) ascending, u = c[a]
__host__ __device__ inline int64_t row_slot(const TermPlan &pl, const int32_t *c, int s, int a) {
    const int32_t u = c[a];
    if (pl.pairwise) return s == 1 ? u : c[1 - a];
    if (s == 1) return 0;
    int32_t o[MAXORD];
    for (int t = 0, j = 0; t < s; ++t)
        if (t != a) o[j++] = c[t] - (c[t] > u); // the other spins renumbered 0..n-2 (`neighbours`, :96)
    return pl.roff[s] + rank_lex(o, s - 1, pl.n - 1);
}

__global__ __launch_bounds__(256) void k_terms_sym(TermPlan pl, const double *__restrict__ rows, int64_t ld, double *__restrict__ out, int64_t tmax) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tmax) return;
    int s = 1;
    while (t >= pl.off[s + 1]) ++s;
    int32_t c[MAXORD];
    unrank_lex(t - pl.off[s], pl.n, s, c);
    double sum = 0.0;
    for (int a = 0; a < s; ++a) sum += rows[(int64_t)c[a] * ld + row_slot(pl, c, s, a)]; // ascending u
    out[t] = sum / (double)s; // `mean` (:147)
}

// The triples of an order-3 model (99 % of its terms; C5: 22.2 M of 22.4 M) without unranking and with every access coalesced.
// A workgroup owns one smallest spin a and a 32 x 32 tile of (b, c), a < b < c.  Term (a, b, c) is the mean of
//   row a, slot of (b, c)   consecutive in c   -- read along the tile's c
//   row b, slot of (a, c)   consecutive in c   -- read along the tile's c
//   row c, slot of (a, b)   consecutive in b, one ROW per c: a column walk of the rows matrix -- read along b (32 consecutive slots
//                           of one row per wave half), transposed through LDS
// and lands at off[3] + rank(a, b, c), consecutive in c.  HBM-bound: 3 x 8 B read + 8 B written per term (C5: 536 MB + 179 MB).
constexpr int TT = 32;
__device__ inline int64_t c2(int64_t x) { return x * (x - 1) / 2; }
__device__ inline int64_t c3(int64_t x) { return x * (x - 1) * (x - 2) / 6; }
__global__ __launch_bounds__(256) void k_terms_sym3(int64_t n, int64_t roff3, int64_t off3, const double *__restrict__ rows, int64_t ld,
                                                    double *__restrict__ out) {
    const int64_t a = blockIdx.z, b0 = (int64_t)blockIdx.y * TT, c0 = (int64_t)blockIdx.x * TT;
    if (c0 + TT - 1 <= b0 || b0 + TT - 1 <= a) return; // no (b, c) of the tile has a < b < c
    __shared__ double tile[TT][TT + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t m = n - 1;
    // pairs (x < y) of the m other spins of a row: slot roff3 + C(m,2) - C(m-x,2) + (y - x - 1)
#pragma unroll
    for (int i = 0; i < TT / 8; ++i) { // row c's entry for (a, b): a and b lie below c, so they keep their numbers among c's others
        const int64_t c = c0 + ty + 8 * i, b = b0 + tx;
        double v = 0.0;
        if (c < n && b > a && b < c) v = rows[c * ld + roff3 + c2(m) - c2(m - a) + (b - a - 1)];
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TT / 8; ++i) {
        const int64_t b = b0 + ty + 8 * i, c = c0 + tx;
        if (b <= a || c <= b || c >= n) continue;
        const double va = rows[a * ld + roff3 + c2(m) - c2(m - (b - 1)) + (c - b - 1)]; // (b, c) above a: renumbered b - 1, c - 1
        const double vb = rows[b * ld + roff3 + c2(m) - c2(m - a) + (c - 1 - a - 1)];   // a below b, c above: a, c - 1
        const double vc = tile[tx][ty + 8 * i];
        const int64_t t = off3 + c3(n) - c3(n - a) + c2(n - a - 1) - c2(n - b) + (c - b - 1);
        out[t] = ((va + vb) + vc) / 3.0; // ascending u, then `mean` (:147): the same operations as k_terms_sym
    }
}

// unsymmetrised: a strided copy -- row u's slots of size s go to uoff[s] + u rcnt[s] + (slot - roff[s]).  One grid row per node, the
// slots along x: no division per element, both sides coalesced
__global__ __launch_bounds__(256) void k_terms_unsym(TermPlan pl, const double *__restrict__ rows, int64_t ld, double *__restrict__ out, int64_t u0) {
    const int64_t u = u0 + blockIdx.y, slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t P = pl.pairwise ? pl.n : pl.roff[pl.order + 1];
    if (slot >= P) return;
    int64_t t;
    if (pl.pairwise) {
        t = slot == u ? u : pl.uoff[2] + u * pl.rcnt[2] + (slot - (slot > u)); // slot i <-> spin i, slot u = the field (:162)
    } else {
        int s = 1;
        while (slot >= pl.roff[s + 1]) ++s;
        t = pl.uoff[s] + u * pl.rcnt[s] + (slot - pl.roff[s]);
    }
    out[t] = rows[u * ld + slot];
}

int make_plan(int64_t n, int order, TermPlan &pl) {
    if (n < 1 || order < 1) return fail(GML_EINVAL, "terms: n = %lld, order = %d", (long long)n, order);
    if (order > MAXORD) return fail(GML_EUNSUPPORTED, "terms: interaction orders above %d are not assembled on the device", MAXORD);
    std::memset(&pl, 0, sizeof pl);
    pl.n = n;
    pl.order = order;
    pl.pairwise = order == 2;
    const double lim = 1.0e12; // (8 TB of doubles; keeps the exact binomials and their intermediate products inside int64)
    double chk = 0;
    for (int s = 1; s <= order; ++s) {
        double b = 1; // C(n, s) in floating point, against overflow of the exact one
        for (int i = 1; i <= s; ++i) b = b * (double)(n - s + i) / i;
        chk += b * s;
        if (chk > lim) return fail(GML_EUNSUPPORTED, "terms: the model has more than 1e12 terms");
        pl.rcnt[s] = binom(n - 1, s - 1);
        pl.off[s + 1] = pl.off[s] + binom(n, s);
        pl.uoff[s + 1] = pl.uoff[s] + n * pl.rcnt[s];
        pl.roff[s + 1] = pl.roff[s] + pl.rcnt[s];
    }
    return GML_OK;
}

bool is_device_ptr(const void *q, int *dev) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, q) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) {
            if (dev) *dev = attr.device;
            return true;
        }
        return false;
    }
    (void)hipGetLastError();
    return false;
}

} // namespace

extern "C" int64_t gml_terms_count(int64_t n, int order, int symmetrize) {
    TermPlan pl;
    if (make_plan(n, order, pl) != GML_OK) return -1;
    return symmetrize ? pl.off[order + 1] : pl.uoff[order + 1];
}

// device-side core: rows and out on `device`, launched on st
int gml_terms_assemble_dev(const double *drows, int64_t ld, int64_t n, int order, int symmetrize, double *dout, hipStream_t st) {
    TermPlan pl;
    const int rc = make_plan(n, order, pl);
    if (rc != GML_OK) return rc;
    const int64_t T = symmetrize ? pl.off[order + 1] : pl.uoff[order + 1];
    const unsigned nb = (unsigned)((T + 255) / 256);
    if (symmetrize && order == 3 && n >= 3 && n <= 65535) {
        // fields and pairs by the generic kernel, the triples by the tiled one
        const int64_t t2 = pl.off[3];
        hipLaunchKernelGGL(k_terms_sym, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, pl, drows, ld, dout, t2);
        const unsigned nt = (unsigned)((n + TT - 1) / TT);
        hipLaunchKernelGGL(k_terms_sym3, dim3(nt, nt, (unsigned)n), dim3(256), 0, st, n, pl.roff[3], pl.off[3], drows, ld, dout);
    } else if (symmetrize) {
        hipLaunchKernelGGL(k_terms_sym, dim3(nb), dim3(256), 0, st, pl, drows, ld, dout, T);
    } else {
        const int64_t P = pl.pairwise ? n : pl.roff[order + 1];
        for (int64_t u0 = 0; u0 < n; u0 += 65535) // (grid rows: at most 65 535 per launch)
            hipLaunchKernelGGL(k_terms_unsym, dim3((unsigned)((P + 255) / 256), (unsigned)std::min<int64_t>(65535, n - u0)), dim3(256), 0, st, pl,
                               drows, ld, dout, u0);
    }
    HIPCHK(hipGetLastError());
    return GML_OK;
}

extern "C" int gml_terms_assemble(const double *rows, int64_t ld, int64_t n, int order, int symmetrize, int device, double *out) {
    if (!rows || !out) return fail(GML_EINVAL, "NULL argument");
    TermPlan pl;
    int rc = make_plan(n, order, pl);
    if (rc != GML_OK) return rc;
    const int64_t P = pl.pairwise ? n : pl.roff[order + 1];
    if (ld < P) return fail(GML_EINVAL, "terms: leading dimension %lld < %lld parameters per node", (long long)ld, (long long)P);
    const int64_t T = symmetrize ? pl.off[order + 1] : pl.uoff[order + 1];
    int rdev = device, odev = device;
    const bool rows_dev = is_device_ptr(rows, &rdev), out_dev = is_device_ptr(out, &odev);
    if (rows_dev) device = rdev;
    else if (out_dev) device = odev;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = nullptr; // the device's null stream: ordered after whatever the caller queued there
    double *drows = nullptr, *dout = nullptr;
    auto cleanup = [&]() {
        if (drows) (void)gml::dev_free(drows);
        if (dout) (void)gml::dev_free(dout);
    };
    const double *src = rows;
    if (!rows_dev) {
        const size_t bytes = sizeof(double) * (size_t)((n - 1) * ld + P);
        hipError_t e = gml::dev_malloc(&drows, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(drows, rows, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            cleanup();
            return fail(e == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "terms: staging the rows failed: %s", hipGetErrorString(e));
        }
        src = drows;
    }
    double *dst = out;
    if (!out_dev) {
        const hipError_t e = gml::dev_malloc(&dout, sizeof(double) * (size_t)T);
        if (e != hipSuccess) {
            cleanup();
            return fail(e == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "terms: allocating the result failed: %s", hipGetErrorString(e));
        }
        dst = dout;
    }
    rc = gml_terms_assemble_dev(src, ld, n, order, symmetrize, dst, st);
    hipError_t e = hipSuccess;
    if (rc == GML_OK && !out_dev) e = hipMemcpyAsync(out, dout, sizeof(double) * (size_t)T, hipMemcpyDeviceToHost, st);
    if (rc == GML_OK && e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (rc != GML_OK) return rc;
    if (e != hipSuccess) return fail(GML_EHIP, "terms: %s", hipGetErrorString(e));
    return GML_OK;
}

// keys of the terms [first, first + count): `order` int32 each, 0-based spins, -1 in the unused slots.  Host only.
extern "C" int gml_terms_keys(int64_t n, int order, int symmetrize, int64_t first, int64_t count, int32_t *keys) {
    TermPlan pl;
    const int rc = make_plan(n, order, pl);
    if (rc != GML_OK) return rc;
    const int64_t T = symmetrize ? pl.off[order + 1] : pl.uoff[order + 1];
    if (first < 0 || count < 0 || first + count > T) return fail(GML_EINVAL, "terms: [%lld, %lld) is outside the %lld terms", (long long)first, (long long)(first + count), (long long)T);
    if (count == 0) return GML_OK;
    if (!keys) return fail(GML_EINVAL, "NULL argument");
    const int64_t chunk = 1 << 16, nchunk = (count + chunk - 1) / chunk;
    gml_parallel_for(nchunk, [&](int64_t ci) {
        const int64_t t0 = first + ci * chunk, t1 = std::min(first + count, t0 + chunk);
        int s = 0;
        int64_t u = -1, left = 0; // positions left in the current (size[, node]) run
        int32_t c[MAXORD];
        const int64_t m = symmetrize ? n : n - 1;
        for (int64_t t = t0; t < t1; ++t) {
            int32_t *k = keys + (t - first) * order;
            if (left == 0) { // (re)start: unrank
                const int64_t *off = symmetrize ? pl.off : pl.uoff;
                s = 1;
                while (t >= off[s + 1]) ++s;
                int64_t r = t - off[s];
                int q = s;
                if (!symmetrize) {
                    u = r / pl.rcnt[s];
                    r = r % pl.rcnt[s];
                    q = s - 1;
                    left = pl.rcnt[s] - r;
                } else {
                    left = off[s + 1] - t;
                }
                unrank_lex(r, m, q, c);
            } else { // next subset in lexicographic order
                const int q = symmetrize ? s : s - 1;
                int i = q - 1;
                while (i >= 0 && c[i] == (int32_t)(m - q + i)) --i;
                ++c[i];
                for (int j = i + 1; j < q; ++j) c[j] = c[j - 1] + 1;
            }
            --left;
            for (int j = 0; j < order; ++j) k[j] = -1;
            if (symmetrize) {
                for (int j = 0; j < s; ++j) k[j] = c[j];
            } else {
                k[0] = (int32_t)u;
                for (int j = 0; j + 1 < s; ++j) k[1 + j] = c[j] + (c[j] >= u); // `neighbours` back to spin ids (:96)
            }
        }
    });
    return GML_OK;
}

// position of a key (len spins, 0-based) among the terms; -1 if it is not a key of the model
extern "C" int64_t gml_terms_rank(int64_t n, int order, int symmetrize, const int32_t *key, int len) {
    TermPlan pl;
    if (make_plan(n, order, pl) != GML_OK) return -1;
    if (!key || len < 1 || len > order) return -1;
    for (int j = 0; j < len; ++j)
        if (key[j] < 0 || key[j] >= n) return -1;
    if (symmetrize) {
        for (int j = 1; j < len; ++j)
            if (key[j] <= key[j - 1]) return -1;
        return pl.off[len] + rank_lex(key, len, n);
    }
    int32_t o[MAXORD];
    const int32_t u = key[0];
    for (int j = 1; j < len; ++j) {
        if (key[j] == u || (j > 1 && key[j] <= key[j - 1])) return -1;
        o[j - 1] = key[j] - (key[j] > u);
    }
    return pl.uoff[len] + (int64_t)u * pl.rcnt[len] + rank_lex(o, len - 1, n - 1);
}

// ------------------------------------------------------------------------------------------
// Pairwise result assembly: reconstruction <- 0.5 (reconstruction + reconstruction') (GraphicalModelLearning.jl:184-186) on the
// device.  32 x 32 tiles, the transposed operand through LDS: both reads and the write coalesced.  (a + b) * 0.5 in FP64 -- the
// same bits as the host expression.  On the host this line costs 16 ms at n = 1024 and 0.22 s at n = 4096 (a strided walk of the
// transposed operand): a sixth of the headline solve, as much as a whole config-4 solve on eight GPUs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pair_sym(const double *__restrict__ R, int64_t ld, int64_t n, double *__restrict__ out, int64_t ldo) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) { // R[j0 + r][i0 + c], read along c
        const int64_t r = j0 + ty + 8 * q, c = i0 + tx;
        tile[ty + 8 * q][tx] = (r < n && c < n) ? R[r * ld + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + ty + 8 * q, j = j0 + tx;
        if (i < n && j < n) out[i * ldo + j] = (R[i * ld + j] + tile[tx][ty + 8 * q]) * 0.5;
    }
}

int gml_pair_symmetrize_dev(const double *drows, int64_t ld, int64_t n, double *dout, int64_t ldo, hipStream_t st) {
    const unsigned nt = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(k_pair_sym, dim3(nt, nt), dim3(256), 0, st, drows, ld, n, dout, ldo);
    HIPCHK(hipGetLastError());
    return GML_OK;
}

extern "C" int gml_matrix_symmetrize(const double *rows, int64_t ld, int64_t n, int device, double *out) {
    if (!rows || !out || n < 1 || ld < n) return fail(GML_EINVAL, "gml_matrix_symmetrize: bad argument");
    int rdev = device, odev = device;
    const bool rows_dev = is_device_ptr(rows, &rdev), out_dev = is_device_ptr(out, &odev);
    if (rows_dev) device = rdev;
    else if (out_dev) device = odev;
    HIPCHK(hipSetDevice(device));
    hipStream_t st = nullptr;
    double *drows = nullptr, *dout = nullptr;
    auto cleanup = [&]() {
        if (drows) (void)gml::dev_free(drows);
        if (dout) (void)gml::dev_free(dout);
    };
    hipError_t e = hipSuccess;
    const double *src = rows;
    int64_t lds = ld;
    if (!rows_dev) {
        e = gml::dev_malloc(&drows, sizeof(double) * (size_t)n * (size_t)n);
        if (e == hipSuccess) e = hipMemcpy2DAsync(drows, sizeof(double) * n, rows, sizeof(double) * ld, sizeof(double) * n, (size_t)n, hipMemcpyHostToDevice, st);
        src = drows;
        lds = n;
    }
    double *dst = out;
    if (e == hipSuccess && (!out_dev || out == rows)) { // (in place is allowed: through a second block)
        e = gml::dev_malloc(&dout, sizeof(double) * (size_t)n * (size_t)n);
        dst = dout;
    }
    int rc = GML_OK;
    if (e == hipSuccess) rc = gml_pair_symmetrize_dev(src, lds, n, dst, n, st);
    if (e == hipSuccess && rc == GML_OK && dst != out)
        e = hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n * (size_t)n, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == GML_OK) e = hipStreamSynchronize(st);
    cleanup();
    if (rc != GML_OK) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "gml_matrix_symmetrize: %s", hipGetErrorString(e));
    return GML_OK;
}

// gml_learn over all nodes of a pairwise handle + the symmetrisation, on the device (include/gml.h)
extern "C" int gml_learn_matrix(gml_problem *p, int formulation, double regularizer_c, int symmetrize, const gml_opts *opts, double *out,
                                double *kkt, gml_stats *stats) {
    if (!p || !out) return fail(GML_EINVAL, "NULL argument");
    if (p->order != 2) return fail(GML_EINVAL, "gml_learn_matrix is for pairwise handles (order 2); multi-body results: gml_learn_terms");
    if (!symmetrize) return gml_learn(p, formulation, regularizer_c, opts, out, kkt, stats);
    if (p->node0 != 0 || p->node1 != p->n)
        return fail(GML_EINVAL, "gml_learn_matrix symmetrises the rows of ALL nodes (this handle holds [%lld, %lld) of %lld): gather the rows of "
                                "gml_learn and call gml_matrix_symmetrize", (long long)p->node0, (long long)p->node1, (long long)p->n);
    HIPCHK(hipSetDevice(p->device));
    const size_t nn = (size_t)p->n * (size_t)p->n;
    double *drows = nullptr, *dsym = nullptr;
    HIPCHK(gml::dev_malloc(&drows, sizeof(double) * nn));
    gml_stats st_local;
    std::memset(&st_local, 0, sizeof st_local);
    int rc = gml_learn(p, formulation, regularizer_c, opts, drows, kkt, &st_local);
    std::string learn_msg = gml_last_error();
    int rc2 = GML_OK;
    if (rc == GML_OK || rc == GML_ENOTCONV) {
        const double t0 = gml_now_s();
        int odev = 0;
        const bool out_dev = is_device_ptr(out, &odev);
        hipError_t e = hipSuccess;
        double *dst = out;
        if (!out_dev) {
            e = gml::dev_malloc(&dsym, sizeof(double) * nn);
            dst = dsym;
        }
        if (e == hipSuccess) {
            rc2 = gml_pair_symmetrize_dev(drows, p->n, p->n, dst, p->n, p->st);
            if (rc2 == GML_OK && !out_dev) e = hipMemcpyAsync(out, dsym, sizeof(double) * nn, hipMemcpyDeviceToHost, p->st);
            if (rc2 == GML_OK && e == hipSuccess) e = hipStreamSynchronize(p->st);
        }
        if (e != hipSuccess) rc2 = fail(e == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "gml_learn_matrix: %s", hipGetErrorString(e));
        st_local.t_assemble = gml_now_s() - t0;
        st_local.t_total += st_local.t_assemble;
    }
    if (dsym) (void)gml::dev_free(dsym);
    (void)gml::dev_free(drows);
    if (stats && (rc == GML_OK || rc == GML_ENOTCONV)) *stats = st_local;
    if (rc2 != GML_OK) return rc2;
    if (rc == GML_ENOTCONV) return fail(GML_ENOTCONV, "%s", learn_msg.c_str());
    return rc;
}

// gml_learn for every node of the handle + the assembly above, the rows never leaving the device (include/gml.h)
extern "C" int gml_learn_terms(gml_problem *p, int formulation, double regularizer_c, int symmetrize, const gml_opts *opts, double *terms,
                               double *kkt, gml_stats *stats) {
    if (!p || !terms) return fail(GML_EINVAL, "NULL argument");
    if (p->node0 != 0 || p->node1 != p->n)
        return fail(GML_EINVAL, "gml_learn_terms needs a handle over all nodes (this one holds [%lld, %lld) of %lld): gather the rows of "
                                "gml_learn and call gml_terms_assemble", (long long)p->node0, (long long)p->node1, (long long)p->n);
    TermPlan pl;
    int rc = make_plan(p->n, p->order, pl);
    if (rc != GML_OK) return rc;
    HIPCHK(hipSetDevice(p->device));
    const int64_t T = symmetrize ? pl.off[p->order + 1] : pl.uoff[p->order + 1];
    double *drows = nullptr, *dout = nullptr;
    HIPCHK(gml::dev_malloc(&drows, sizeof(double) * (size_t)p->n * (size_t)p->P));
    gml_stats st_local;
    std::memset(&st_local, 0, sizeof st_local);
    rc = gml_learn(p, formulation, regularizer_c, opts, drows, kkt, &st_local);
    std::string learn_msg = gml_last_error();
    int rc2 = GML_OK;
    if (rc == GML_OK || rc == GML_ENOTCONV) {
        const double t0 = gml_now_s();
        int odev = 0;
        const bool out_dev = is_device_ptr(terms, &odev);
        hipError_t e = hipSuccess;
        double *dst = terms;
        if (!out_dev) {
            e = gml::dev_malloc(&dout, sizeof(double) * (size_t)T);
            dst = dout;
        }
        if (e == hipSuccess) {
            rc2 = gml_terms_assemble_dev(drows, p->P, p->n, p->order, symmetrize, dst, p->st);
            if (rc2 == GML_OK && !out_dev) e = hipMemcpyAsync(terms, dout, sizeof(double) * (size_t)T, hipMemcpyDeviceToHost, p->st);
            if (rc2 == GML_OK && e == hipSuccess) e = hipStreamSynchronize(p->st);
        }
        if (e != hipSuccess) rc2 = fail(e == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP, "terms: %s", hipGetErrorString(e));
        st_local.t_assemble = gml_now_s() - t0;
        st_local.t_total += st_local.t_assemble;
    }
    if (dout) (void)gml::dev_free(dout);
    (void)gml::dev_free(drows);
    if (stats && (rc == GML_OK || rc == GML_ENOTCONV)) *stats = st_local;
    if (rc2 != GML_OK) return rc2;
    if (rc == GML_ENOTCONV) return fail(GML_ENOTCONV, "%s", learn_msg.c_str());
    return rc;
}
