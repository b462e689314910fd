// Host-side single-pass packer of the sample histogram (the ingest step of gml_problem_create / gml_multi_create).
//
// The reference hands `learn` a K x (1+n) matrix of 8-byte elements (sampling.jl:52-54: Matrix{Int64}, column 1 = counts;
// readdlm gives Float64, test/runtests.jl:71), copied once at GraphicalModelLearning.jl:73 and re-read n times by the
// nodal_stat comprehension (:162).  Here it is read exactly once, on the host, and leaves as ONE BIT per spin: the
// spin-major sign words of the device image Sb (gml_bits.h) plus the K counts -- 1/64 of the bytes cross PCIe, and for
// several GPUs the packing is done once and the bits are replicated.
//
// A Julia matrix is column-major: column 1+i holds spin i of all K configurations contiguously, so 32 consecutive
// elements give one sign word (AVX2: compare + movemask, 8 loads per word for Int64/Float64).  A row-major (numpy)
// matrix is packed through 32 x 32 bit transposes of row words.  The +-1 alphabet is validated in the same sweep (the
// reference validates nothing): a residue that is zero iff every element is +-1 is OR-ed per block and looked at once.
//
// Plain C++ (no HIP): unit-tested on the CPU box through gml_pack_histogram (tests/test_host_pack.py).
#include "gml_pack.h"

#include "gml_bits.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <immintrin.h>

namespace gml {
namespace {

// ---- 32 consecutive elements -> sign word (bit j set <=> element j is -1); *bad |= non-zero iff some element is not +-1
template <typename T> struct Elem;
template <> struct Elem<int8_t> {
    static inline bool neg(int8_t v) { return v == -1; }
    static inline bool ok(int8_t v) { return v == 1 || v == -1; }
    static inline double num(int8_t v) { return (double)v; }
};
template <> struct Elem<int32_t> {
    static inline bool neg(int32_t v) { return v == -1; }
    static inline bool ok(int32_t v) { return v == 1 || v == -1; }
    static inline double num(int32_t v) { return (double)v; }
};
template <> struct Elem<int64_t> {
    static inline bool neg(int64_t v) { return v == -1; }
    static inline bool ok(int64_t v) { return v == 1 || v == -1; }
    static inline double num(int64_t v) { return (double)v; }
};
template <> struct Elem<double> {
    static inline bool neg(double v) { return v == -1.0; }
    static inline bool ok(double v) { return v == 1.0 || v == -1.0; }
    static inline double num(double v) { return v; }
};

template <typename T> inline uint32_t word_scalar(const T *p, int cnt, uint64_t *bad) {
    uint32_t m = 0;
    for (int j = 0; j < cnt; ++j) {
        const T v = p[j];
        if (Elem<T>::neg(v)) m |= 1u << j;
        else if (!Elem<T>::ok(v)) *bad |= 1;
    }
    return m;
}

// AVX2 forms for a full word.  Residue: integers (v + 1) & ~2 (zero iff v in {-1, +1}); doubles (|bits|) ^ bits(1.0).
__attribute__((target("avx2"))) inline uint32_t word_avx2(const int64_t *p, uint64_t *bad) {
    const __m256i one = _mm256_set1_epi64x(1), keep = _mm256_set1_epi64x(~2ll);
    __m256i acc = _mm256_setzero_si256();
    uint32_t m = 0;
    for (int q = 0; q < 8; ++q) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 4 * q));
        m |= (uint32_t)_mm256_movemask_pd(_mm256_castsi256_pd(v)) << (4 * q);
        acc = _mm256_or_si256(acc, _mm256_and_si256(_mm256_add_epi64(v, one), keep));
    }
    *bad |= (uint64_t)!_mm256_testz_si256(acc, acc);
    return m;
}
__attribute__((target("avx2"))) inline uint32_t word_avx2(const double *p, uint64_t *bad) {
    const __m256i absm = _mm256_set1_epi64x(0x7FFFFFFFFFFFFFFFll), onebits = _mm256_set1_epi64x(0x3FF0000000000000ll);
    __m256i acc = _mm256_setzero_si256();
    uint32_t m = 0;
    for (int q = 0; q < 8; ++q) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 4 * q));
        m |= (uint32_t)_mm256_movemask_pd(_mm256_castsi256_pd(v)) << (4 * q);
        acc = _mm256_or_si256(acc, _mm256_xor_si256(_mm256_and_si256(v, absm), onebits));
    }
    *bad |= (uint64_t)!_mm256_testz_si256(acc, acc);
    return m;
}
__attribute__((target("avx2"))) inline uint32_t word_avx2(const int32_t *p, uint64_t *bad) {
    const __m256i one = _mm256_set1_epi32(1), keep = _mm256_set1_epi32(~2);
    __m256i acc = _mm256_setzero_si256();
    uint32_t m = 0;
    for (int q = 0; q < 4; ++q) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 8 * q));
        m |= (uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(v)) << (8 * q);
        acc = _mm256_or_si256(acc, _mm256_and_si256(_mm256_add_epi32(v, one), keep));
    }
    *bad |= (uint64_t)!_mm256_testz_si256(acc, acc);
    return m;
}
__attribute__((target("avx2"))) inline uint32_t word_avx2(const int8_t *p, uint64_t *bad) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
    const __m256i res = _mm256_and_si256(_mm256_add_epi8(v, _mm256_set1_epi8(1)), _mm256_set1_epi8((char)~2));
    *bad |= (uint64_t)!_mm256_testz_si256(res, res);
    return (uint32_t)_mm256_movemask_epi8(v);
}

bool have_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}

// ---- column-major: the words [w0, w1) of spin column `col` (K contiguous elements) ---------------------------------------
template <typename T> void pack_col_tail(const T *col, int64_t K, int64_t w, int64_t w1, uint32_t *out, uint64_t *bad) {
    for (; w < w1; ++w) {
        const int64_t left = K - 32 * w;
        out[w] = left > 0 ? word_scalar(col + 32 * w, (int)std::min<int64_t>(left, 32), bad) : 0u;
    }
}
template <typename T> __attribute__((target("avx2"))) void pack_col_range_avx2(const T *col, int64_t K, int64_t w0, int64_t w1, uint32_t *out, uint64_t *bad) {
    const int64_t wf = std::min(w1, K / 32); // words made of 32 real samples
    for (int64_t w = w0; w < wf; ++w) out[w] = word_avx2(col + 32 * w, bad);
    pack_col_tail(col, K, std::max(w0, wf), w1, out, bad);
}
template <typename T> void pack_col_range_scalar(const T *col, int64_t K, int64_t w0, int64_t w1, uint32_t *out, uint64_t *bad) {
    const int64_t wf = std::min(w1, K / 32);
    for (int64_t w = w0; w < wf; ++w) out[w] = word_scalar(col + 32 * w, 32, bad);
    pack_col_tail(col, K, std::max(w0, wf), w1, out, bad);
}

// ---- row-major: a block of 16 words (512 samples) x 32 columns through 32 x 32 bit transposes ------------------------------
// base points at element (0, spin 0); out at the word row of column c0
template <typename T> __attribute__((target("avx2"))) void pack_row_block_avx2(const T *base, int64_t ld, int64_t K, int64_t c0, int cnt, int64_t wb0, int64_t wb1, int64_t wpr, uint32_t *out, uint64_t *bad) {
    for (int64_t w = wb0; w < wb1; ++w) {
        uint32_t a[32];
        for (int r = 0; r < 32; ++r) {
            const int64_t k = 32 * w + r;
            if (k >= K) a[r] = 0u;
            else if (cnt == 32) a[r] = word_avx2(base + k * ld + c0, bad);
            else a[r] = word_scalar(base + k * ld + c0, cnt, bad);
        }
        transpose32(a); // a[c] bit r <- old a[r] bit c
        for (int c = 0; c < cnt; ++c) out[(int64_t)c * wpr + w] = a[c];
    }
}
template <typename T> void pack_row_block_scalar(const T *base, int64_t ld, int64_t K, int64_t c0, int cnt, int64_t wb0, int64_t wb1, int64_t wpr, uint32_t *out, uint64_t *bad) {
    for (int64_t w = wb0; w < wb1; ++w) {
        uint32_t a[32];
        for (int r = 0; r < 32; ++r) {
            const int64_t k = 32 * w + r;
            a[r] = k < K ? word_scalar(base + k * ld + c0, cnt, bad) : 0u;
        }
        transpose32(a);
        for (int c = 0; c < cnt; ++c) out[(int64_t)c * wpr + w] = a[c];
    }
}

template <typename T> int64_t first_bad_config(const HistView &h, int64_t i0, int64_t i1) {
    const T *base = static_cast<const T *>(h.base);
    int64_t best = -1;
    for (int64_t i = i0; i < i1; ++i)
        for (int64_t k = 0; k < h.K && (best < 0 || k < best); ++k) {
            const T v = h.col_major ? base[(h.spin_off + i) * h.ld + k] : base[k * h.ld + h.spin_off + i];
            if (!Elem<T>::ok(v)) {
                best = k;
                break;
            }
        }
    return best;
}

template <typename T> int64_t pack_spins_t(const HistView &h, int64_t i0, int64_t i1, int64_t wpr, uint32_t *out, const ParallelFor &pf) {
    const T *base = static_cast<const T *>(h.base);
    const bool avx = have_avx2();
    std::atomic<uint64_t> anybad(0);
    const int64_t wreal = (h.K + 31) / 32;
    if (h.col_major) {
        // task = (spin, slab of 8192 words = 256 Ki samples): 2 MB of Int64 input per task
        const int64_t slab = 8192, nslab = (wpr + slab - 1) / slab;
        pf((i1 - i0) * nslab, [&](int64_t t) {
            const int64_t i = i0 + t / nslab, s = t % nslab;
            const int64_t w0 = s * slab, w1 = std::min(wpr, w0 + slab);
            uint64_t bad = 0;
            const T *col = base + (h.spin_off + i) * h.ld;
            uint32_t *row = out + (i - i0) * wpr;
            if (avx) pack_col_range_avx2<T>(col, h.K, w0, w1, row, &bad);
            else pack_col_range_scalar<T>(col, h.K, w0, w1, row, &bad);
            if (bad) anybad.store(1, std::memory_order_relaxed);
        });
    } else {
        // task = (group of 32 columns, block of 16 words = 512 samples): every output line (64 B) is written by one task
        const int64_t ncg = (i1 - i0 + 31) / 32, nwb = (wpr + 15) / 16;
        pf(ncg * nwb, [&](int64_t t) {
            const int64_t cg = t / nwb, wb = t % nwb;
            const int64_t c0 = i0 + 32 * cg;
            const int cnt = (int)std::min<int64_t>(32, i1 - c0);
            const int64_t w0 = wb * 16, w1 = std::min(wpr, w0 + 16);
            uint64_t bad = 0;
            uint32_t *row = out + (c0 - i0) * wpr;
            if (w0 >= wreal) {
                for (int c = 0; c < cnt; ++c)
                    for (int64_t w = w0; w < w1; ++w) row[(int64_t)c * wpr + w] = 0u;
                return;
            }
            if (avx) pack_row_block_avx2<T>(base + h.spin_off, h.ld, h.K, c0, cnt, w0, w1, wpr, row, &bad);
            else pack_row_block_scalar<T>(base + h.spin_off, h.ld, h.K, c0, cnt, w0, w1, wpr, row, &bad);
            if (bad) anybad.store(1, std::memory_order_relaxed);
        });
    }
    if (!anybad.load()) return -1;
    return first_bad_config<T>(h, i0, i1);
}

template <typename T> int64_t pack_counts_t(const HistView &h, double *counts, double *Msum, const ParallelFor &pf) {
    const T *cb = static_cast<const T *>(h.counts);
    const int64_t blk = 65536, nb = (h.K + blk - 1) / blk;
    std::vector<double> part((size_t)nb, 0.0);
    std::atomic<int64_t> bad(-1);
    pf(nb, [&](int64_t b) {
        const int64_t k1 = std::min(h.K, (b + 1) * blk);
        double s = 0;
        for (int64_t k = b * blk; k < k1; ++k) {
            const double c = cb ? Elem<T>::num(cb[k * h.counts_stride]) : 1.0;
            counts[k] = c;
            if (!(c >= 0) || !std::isfinite(c)) {
                int64_t cur = bad.load();
                while ((cur < 0 || k < cur) && !bad.compare_exchange_weak(cur, k)) {
                }
            }
            s += c;
        }
        part[(size_t)b] = s;
    });
    double M = 0;
    for (double s : part) M += s; // fixed order: M does not depend on the thread count
    *Msum = M;
    return bad.load();
}

} // namespace

int64_t pack_spins(const HistView &h, int64_t i0, int64_t i1, int64_t wpr, uint32_t *out, const ParallelFor &pf) {
    switch (h.dtype) {
    case GML_I8: return pack_spins_t<int8_t>(h, i0, i1, wpr, out, pf);
    case GML_I32: return pack_spins_t<int32_t>(h, i0, i1, wpr, out, pf);
    case GML_I64: return pack_spins_t<int64_t>(h, i0, i1, wpr, out, pf);
    default: return pack_spins_t<double>(h, i0, i1, wpr, out, pf);
    }
}

int64_t pack_counts(const HistView &h, double *counts, double *Msum, const ParallelFor &pf) {
    switch (h.counts_dtype) {
    case GML_I8: return pack_counts_t<int8_t>(h, counts, Msum, pf);
    case GML_I32: return pack_counts_t<int32_t>(h, counts, Msum, pf);
    case GML_I64: return pack_counts_t<int64_t>(h, counts, Msum, pf);
    default: return pack_counts_t<double>(h, counts, Msum, pf);
    }
}

HistView hist_view(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, bool col_major) {
    HistView h{};
    h.base = samples;
    h.dtype = dtype;
    h.K = K;
    h.n = n;
    h.ld = ld;
    h.col_major = col_major;
    h.spin_off = 1; // column 0 = counts (sampling.jl:52-54)
    h.counts = samples;
    h.counts_dtype = dtype;
    h.counts_stride = col_major ? 1 : ld;
    return h;
}

} // namespace gml
