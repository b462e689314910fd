// Host-side packer of the sample histogram: interface (gml_pack.cpp).  Plain C++, no HIP types.
#pragma once
#include "../../include/gml.h"

#include <cstdint>
#include <functional>
#include <vector>

namespace gml {

// A histogram matrix as the caller holds it (K x (1+n), column 0 = counts: sampling.jl:52-54), or split inputs
// (spins K x n + counts, gml_problem_create_spins): `spin_off` = first spin column, `counts` = first count (NULL: all ones).
struct HistView {
    const void *base;
    int dtype;          // GML_I8 / I32 / I64 / F64 of the spins
    int64_t K, n, ld;   // element (k, j) at base[k + j ld] (col_major) or base[k ld + j]
    bool col_major;
    int64_t spin_off;
    const void *counts; // element k at counts[k * counts_stride], type counts_dtype
    int counts_dtype;
    int64_t counts_stride;
};
HistView hist_view(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, bool col_major);

typedef std::function<void(int64_t, const std::function<void(int64_t)> &)> ParallelFor;

// counts -> double [K] (finite, >= 0), *Msum = their sum.  Returns the first offending configuration, or -1.
int64_t pack_counts(const HistView &h, double *counts, double *Msum, const ParallelFor &pf);
// Sign words of the spins [i0, i1): out[(i - i0) * wpr + w], bit j of word w <-> configuration 32 w + j, set <=> -1;
// all wpr words of every row are written (zero beyond K).  Returns the smallest configuration index that holds an
// element other than +-1 in those spins, or -1.
int64_t pack_spins(const HistView &h, int64_t i0, int64_t i1, int64_t wpr, uint32_t *out, const ParallelFor &pf);

} // namespace gml
