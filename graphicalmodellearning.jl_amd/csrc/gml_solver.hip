// Device side of the batched l1 solver (gml_solver.cpp): everything the iteration does with the dense per-row arrays
// -- pseudo-gradient / KKT residual, working-set selection, trial points of the projected line search, acceptance
// tests, the vector operations of the matrix-free Newton-CG -- runs here, on [rows][Qp] FP64 arrays that never leave
// HBM.  Only a few scalars per row cross PCIe per iteration.
//
// The problem of row r (node u = node[r]) is   min_x f_u(x) + lambda * sum_{c penalised} |x_c|   -- what the
// reference hands to Ipopt through the z >= |x| epigraph (GraphicalModelLearning.jl:166-177); kind[r][c] says which
// columns are parameters of the row: 0 = none (the key contains u, or padding), 1 = free (the field), 2 = penalised.
#include "../../include/gml.h"
#include "gml_dev.h"
#include "gml_solver.h"

namespace gml {

__device__ __forceinline__ double pseudo_grad(double x, double g, double lam) {
    if (lam == 0.0) return g;
    if (x > 0) return g + lam;
    if (x < 0) return g - lam;
    if (g + lam < 0) return g + lam;
    if (g - lam > 0) return g - lam;
    return 0.0;
}

__device__ __forceinline__ double block_sum(double v, double *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double block_max(double v, double *red) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__device__ __forceinline__ int block_sum_i(int v, int *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// kind[r][c] from the statistic keys: column c is a parameter of node u iff its key does not contain u and its size
// is at most order - 1 (:94-104); the constant column is the field (free), every other parameter is penalised (:118, :171)
__global__ __launch_bounds__(256) void k_kind(const int32_t *__restrict__ keys, int ko, int64_t Qf, int64_t Qp, int64_t cconst,
                                              int order, const int *__restrict__ node, uint8_t *__restrict__ kind) {
    const int r = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Qp) return;
    const int u = node[r];
    uint8_t k = 0;
    if (u >= 0) {
        if (c == cconst) k = 1;
        else if (c < Qf) {
            int sz = 0;
            bool has = false;
            for (int t = 0; t < ko; ++t) {
                const int i = keys[c * ko + t];
                if (i >= 0) {
                    ++sz;
                    has |= (i == u);
                }
            }
            if (!has && sz <= order - 1) k = 2;
        }
    }
    kind[(int64_t)r * Qp + c] = k;
}

void launch_kind(const DevProblem &d, int order, const int *dnode, int R, uint8_t *kind, hipStream_t st) {
    hipLaunchKernelGGL(k_kind, dim3((unsigned)((d.Qp + 255) / 256), (unsigned)R), dim3(256), 0, st, d.keys, d.ko, d.Qf, d.Qp, d.cconst, order,
                       dnode, kind);
}

// G[r][:] *= scale[r] for the listed rows (logRISE: grad log Z = grad Z / Z, :279)
__global__ __launch_bounds__(256) void k_scale_rows(const int *__restrict__ rows, const double *__restrict__ scale, int64_t Qp,
                                                    double *__restrict__ G) {
    const int r = rows[blockIdx.y];
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < Qp) G[(int64_t)r * Qp + c] *= scale[r];
}
void launch_scale_rows(const int *drows, int nrows, const double *dscale, int64_t Qp, double *G, hipStream_t st) {
    if (nrows > 0)
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((Qp + 255) / 256), (unsigned)nrows), dim3(256), 0, st, drows, dscale, Qp, G);
}

// dst[r][:] = src[r][:] for the listed rows (up to two pairs of arrays)
__global__ __launch_bounds__(256) void k_copy_rows(const int *__restrict__ rows, int64_t Qp, const double *__restrict__ s0,
                                                   double *__restrict__ d0, const double *__restrict__ s1, double *__restrict__ d1) {
    const int r = rows[blockIdx.y];
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Qp) return;
    d0[(int64_t)r * Qp + c] = s0[(int64_t)r * Qp + c];
    if (s1) d1[(int64_t)r * Qp + c] = s1[(int64_t)r * Qp + c];
}
// Rows of the internal column layout -> the reference's parameter order: out[r][j] = X[r][col(r, j)].  Pairwise (cols ==
// NULL): slot j <-> spin j, slot u = the field (the u-th column of nodal_stat is s_u, GraphicalModelLearning.jl:162);
// multi-body: cols [R][P] from the host (node_cols: the key order of :94-104).
__global__ __launch_bounds__(256) void k_rows_to_reference(const double *__restrict__ X, int64_t Qp, int64_t P, int64_t node0,
                                                           int64_t cconst, const int32_t *__restrict__ cols, double *__restrict__ out) {
    const int64_t r = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const int64_t c = cols ? cols[r * P + j] : (j == node0 + r ? cconst : j);
    out[r * P + j] = X[r * Qp + c];
}

void launch_rows_to_reference(const double *X, int64_t R, int64_t Qp, int64_t P, int64_t node0, int64_t cconst, const int32_t *cols, double *out,
                              hipStream_t st) {
    if (R > 0)
        hipLaunchKernelGGL(k_rows_to_reference, dim3((unsigned)((P + 255) / 256), (unsigned)R), dim3(256), 0, st, X, Qp, P, node0, cconst, cols, out);
}

// logRISE: grad log Z = grad Z / Z (:279), with Z taken from the pass results on the device (no host round trip between the
// pass and whatever consumes its gradient).  By slot (int8 path: row srow[slot], Z = res[slot].f) or by row (FP64 path).
__global__ __launch_bounds__(256) void k_scale_slots_inv(const int *__restrict__ srow, const int *__restrict__ rowcol, int slot0,
                                                         const SlotResult *__restrict__ res, int64_t Qp, double *__restrict__ G) {
    const int s = slot0 + blockIdx.y;
    if (rowcol[s] < 0) return;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < Qp) G[(int64_t)srow[s] * Qp + c] *= 1.0 / res[s].f;
}
__global__ __launch_bounds__(256) void k_scale_rows_inv(const int *__restrict__ rows, const double *__restrict__ fs, int64_t Qp,
                                                        double *__restrict__ G) {
    const int r = rows[blockIdx.y];
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < Qp) G[(int64_t)r * Qp + c] *= 1.0 / fs[r];
}
void launch_scale_slots_inv(const int *srow, const int *rowcol, int slot0, int ns, const SlotResult *res, int64_t Qp, double *G, hipStream_t st) {
    if (ns > 0)
        hipLaunchKernelGGL(k_scale_slots_inv, dim3((unsigned)((Qp + 255) / 256), (unsigned)ns), dim3(256), 0, st, srow, rowcol, slot0, res, Qp, G);
}
void launch_scale_rows_inv(const int *drows, int nrows, const double *fs, int64_t Qp, double *G, hipStream_t st) {
    if (nrows > 0)
        hipLaunchKernelGGL(k_scale_rows_inv, dim3((unsigned)((Qp + 255) / 256), (unsigned)nrows), dim3(256), 0, st, drows, fs, Qp, G);
}

void launch_copy_rows(const int *drows, int nrows, int64_t Qp, const double *s0, double *d0, const double *s1, double *d1, hipStream_t st) {
    if (nrows > 0)
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((Qp + 255) / 256), (unsigned)nrows), dim3(256), 0, st, drows, Qp, s0, d0, s1, d1);
}

// ------------------------------------------------------------------------------------------
// KKT residual and working set of one row (one workgroup per listed row).
//   PG[r][c]  pseudo-gradient (minimum-norm subgradient of F = f + lambda |x|_1 on the penalised columns)
//   out[r]    {l1 = lambda sum |x_c|, worst = max |pg|, worstW = max |pg| on the current support, m, nsupp, nviol}
//   working set W = support (x != 0, or the free column) + the at most max_add largest violators (x = 0, pg != 0) --
//   none while the residual on the support still dominates (worstW ~ worst: the violations outside are then largely
//   an artefact of the unconverged support) -- written in ascending column order to F[r][0..m), with g and pg
//   gathered next to it; when W exceeds capW the row is flagged for the matrix-free Newton-CG instead (m = -|W|,
//   W = every column with x != 0 or pg != 0 after the admission cut; nothing is gathered).
// Also keeps the best iterate: if worst < best[r], best[r] = worst and Xbest[r] = X[r].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ G,
                                                const uint8_t *__restrict__ kind, int64_t Qp, double lambda, int max_add, int capW,
                                                int capP, double viol_frac, double *__restrict__ PG, int *__restrict__ F, double *__restrict__ gF,
                                                double *__restrict__ pgF, SelectOut *__restrict__ out, double *__restrict__ best,
                                                double *__restrict__ Xbest) {
    const int r = rows[blockIdx.x];
    const int tid = threadIdx.x;
    const double *x = X + (int64_t)r * Qp, *g = G + (int64_t)r * Qp;
    const uint8_t *kr = kind + (int64_t)r * Qp;
    double *pgr = PG + (int64_t)r * Qp;
    const double bprev = best[r]; // read by every thread before thread 0 updates it (barriers in between)
    __shared__ double red[4];
    __shared__ int redi[4];
    __shared__ int scan[257];
    double l1 = 0, worst = 0, worstW = 0;
    int nsupp = 0, nviol = 0;
    for (int64_t c = tid; c < Qp; c += 256) {
        const uint8_t k = kr[c];
        double pg = 0.0;
        if (k) {
            const double l = k == 2 ? lambda : 0.0, xc = x[c];
            l1 += l * fabs(xc);
            pg = pseudo_grad(xc, g[c], l);
            const double a = fabs(pg);
            worst = fmax(worst, a);
            if (xc != 0.0 || k == 1) {
                ++nsupp;
                worstW = fmax(worstW, a);
            } else if (pg != 0.0) {
                ++nviol;
            }
        }
        pgr[c] = pg;
    }
    l1 = block_sum(l1, red);
    worst = block_max(worst, red);
    worstW = block_max(worstW, red);
    nsupp = block_sum_i(nsupp, redi);
    nviol = block_sum_i(nviol, redi);
    if (!(worst == worst)) worst = INFINITY; // NaN
    const bool addv = !(worstW > worst * 0.999999 && worstW > 0 && nsupp > 1);
    // The Hessian blocks and the Cholesky work on whole 32-entry tiles: admit as many violators as fill the working set up
    // to a multiple of 32 rather than one entry into the next tile (support 1 + 64 violators = 65 entries would be three
    // tiles, 2.2x the Hessian work of the 64 that two tiles hold), as long as at least half of max_add still get in.
    // A row with violators by the ten thousand has a dense optimum ahead of it (config 5 at the reference's default
    // regulariser: 93 000 at the start, 7 000 in the solution): twice the entries per iteration until it goes matrix-free
    // (13 -> 9 iterations of that phase, 36.6 -> 34.9 s on one box, the sparse-optimum run at c = 1.2 4.7 -> 3.7 s; four times as many
    // lengthen the tail instead).
    if (nviol > 16 * capW) max_add *= 2;
    {
        const int full = (nsupp + max_add) / 32 * 32;
        if (full - nsupp >= max_add / 2) max_add = full - nsupp;
    }
    // threshold of the max_add largest violators: bisection on the float pattern of |pg| (monotone as unsigned)
    // (a support that alone exceeds capW makes the row matrix-free whatever the threshold: no search -- 31 sweeps of the row)
    unsigned thr = 0;
    if (addv && nviol > max_add && nsupp <= capW) {
        unsigned lo = 0, hi = 0x7f800000u; // invariant: count(v >= lo) > max_add >= count(v >= hi)
        constexpr int NREG = 8;
        if (Qp <= 256 * NREG) {
            // rows of up to 2048 columns (the pairwise configurations): the candidates' float patterns in registers, counted by
            // ballots -- the 31 steps touch no memory but the four per-wave counts (31 strided sweeps of the row with a block
            // reduction each were 45 of this kernel's 60 us on a 128-node shard)
            unsigned v[NREG];
#pragma unroll
            for (int q = 0; q < NREG; ++q) {
                const int64_t c = tid + 256 * q;
                v[q] = (c < Qp && kr[c] == 2 && x[c] == 0.0 && pgr[c] != 0.0) ? __float_as_uint((float)fabs(pgr[c])) : 0u;
            }
            auto count_ge = [&](unsigned t) {
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < NREG; ++q) cnt += __popcll(__ballot(v[q] >= t && v[q] != 0u));
                __syncthreads();
                if ((tid & 63) == 0) redi[tid >> 6] = cnt;
                __syncthreads();
                return redi[0] + redi[1] + redi[2] + redi[3];
            };
            while (hi - lo > 1) {
                const unsigned mid = lo + (hi - lo) / 2;
                if (count_ge(mid) > max_add) lo = mid;
                else hi = mid;
            }
            thr = count_ge(hi) > 0 ? hi : lo;
        } else {
            while (hi - lo > 1) {
                const unsigned mid = lo + (hi - lo) / 2;
                int cnt = 0;
                for (int64_t c = tid; c < Qp; c += 256)
                    if (kr[c] == 2 && x[c] == 0.0 && pgr[c] != 0.0 && __float_as_uint((float)fabs(pgr[c])) >= mid) ++cnt;
                cnt = block_sum_i(cnt, redi);
                if (cnt > max_add) lo = mid;
                else hi = mid;
            }
            int cnt = 0;
            for (int64_t c = tid; c < Qp; c += 256)
                if (kr[c] == 2 && x[c] == 0.0 && pgr[c] != 0.0 && __float_as_uint((float)fabs(pgr[c])) >= hi) ++cnt;
            cnt = block_sum_i(cnt, redi);
            thr = cnt > 0 ? hi : lo; // nothing above the tie class at `lo`: take that class (a large one sends the row to CG)
        }
    }
    SelectOut o;
    o.l1 = l1;
    o.worst = worst;
    o.worstW = worstW;
    o.nsupp = nsupp;
    o.nviol = nviol;
    // ordered compaction: thread t owns the columns [t*chunk, (t+1)*chunk)
    const int64_t chunk = (Qp + 255) / 256, c0 = tid * chunk, c1 = c0 + chunk < Qp ? c0 + chunk : Qp;
    int m = nsupp; // (a support above capW: matrix-free whatever the violators -- the ordered count below, a strided sweep, is not needed)
    if (nsupp <= capW) {
        int cnt = 0;
        for (int64_t c = c0; c < c1; ++c) {
            const uint8_t k = kr[c];
            if (!k) continue;
            cnt += (x[c] != 0.0 || k == 1) || (addv && pgr[c] != 0.0 && __float_as_uint((float)fabs(pgr[c])) >= thr);
        }
        scan[tid + 1] = cnt;
        if (tid == 0) scan[0] = 0;
        __syncthreads();
        if (tid == 0)
            for (int t = 1; t <= 256; ++t) scan[t] += scan[t - 1];
        __syncthreads();
        m = scan[256];
    }
    if (m > capW) {
        // Matrix-free row: W = the support + the admitted violators (dense vectors, nothing gathered here: k_cg_tiles lists W
        // for the preconditioner once the host knows its size, m = -|W|).
        // Only the violators within viol_frac of the largest one enter W in this iteration (the others keep pg = 0 in the
        // dense array, which defines W for the CG and the line search): thousands of coordinates leaving zero at once, most
        // of them to come back, make the projected Newton step a poor one.
        if (viol_frac > 0.0 && nviol * 16 > nsupp) { // (a few violators next to a large support: all of them)
            double mv = 0.0;
            for (int64_t c = tid; c < Qp; c += 256)
                if (kr[c] == 2 && x[c] == 0.0) mv = fmax(mv, fabs(pgr[c]));
            mv = block_max(mv, red);
            const double cut = viol_frac * mv;
            for (int64_t c = tid; c < Qp; c += 256)
                if (kr[c] == 2 && x[c] == 0.0 && fabs(pgr[c]) < cut) pgr[c] = 0.0;
            __syncthreads();
        }
        int cw = 0;
        for (int64_t c = tid; c < Qp; c += 256) cw += kr[c] && (x[c] != 0.0 || pgr[c] != 0.0);
        cw = block_sum_i(cw, redi);
        o.m = -cw;
        o.pad = 0;
    } else {
        int pos = scan[tid];
        for (int64_t c = c0; c < c1; ++c) {
            const uint8_t k = kr[c];
            if (!k) continue;
            if ((x[c] != 0.0 || k == 1) || (addv && pgr[c] != 0.0 && __float_as_uint((float)fabs(pgr[c])) >= thr)) {
                F[(int64_t)r * capP + pos] = (int)c;
                gF[(int64_t)r * capP + pos] = g[c];
                pgF[(int64_t)r * capP + pos] = pgr[c];
                ++pos;
            }
        }
        for (int a = m + tid; a < capP; a += 256) { // padding
            F[(int64_t)r * capP + a] = (int)(Qp - 1);
            gF[(int64_t)r * capP + a] = 0.0;
            pgF[(int64_t)r * capP + a] = 0.0;
        }
        o.m = m;
        o.pad = 0;
    }
    const bool better = worst < bprev;
    if (tid == 0) {
        out[r] = o;
        if (better) best[r] = worst;
    }
    if (better)
        for (int64_t c = tid; c < Qp; c += 256) Xbest[(int64_t)r * Qp + c] = x[c];
}

void launch_select(const int *drows, int nrows, const double *X, const double *G, const uint8_t *kind, int64_t Qp, double lambda,
                   int max_add, int capW, int capP, double viol_frac, double *PG, int *F, double *gF, double *pgF, SelectOut *out, double *best,
                   double *Xbest, hipStream_t st) {
    if (nrows > 0)
        hipLaunchKernelGGL(k_select, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, G, kind, Qp, lambda, max_add, capW, capP, viol_frac, PG, F, gF,
                           pgF, out, best, Xbest);
}

// D[r][:] = 0, D[r][F[r][a]] = dsol[r][a] (a < m[r]) for the listed rows: the Newton direction of the Cholesky rows
__global__ __launch_bounds__(256) void k_scatter_dir(const int *__restrict__ rows, const int *__restrict__ F, const double *__restrict__ dsol,
                                                     const int *__restrict__ msz, int capP, int64_t Qp, double *__restrict__ D) {
    const int r = rows[blockIdx.x];
    double *d = D + (int64_t)r * Qp;
    for (int64_t c = threadIdx.x; c < Qp; c += 256) d[c] = 0.0;
    __syncthreads();
    const int m = msz[r];
    for (int a = threadIdx.x; a < m; a += 256) d[F[(int64_t)r * capP + a]] = dsol[(int64_t)r * capP + a];
}
void launch_scatter_dir(const int *drows, int nrows, const int *F, const double *dsol, const int *msz, int capP, int64_t Qp, double *D,
                        hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_scatter_dir, dim3((unsigned)nrows), dim3(256), 0, st, drows, F, dsol, msz, capP, Qp, D);
}

// ------------------------------------------------------------------------------------------
// Secant correction of the sub-sampled Hessian blocks (Cholesky rows).  The blocks come from a few per cent of the
// configurations, so a row whose working set is final converges linearly (the headline problem: 8 of its 14 iterations
// go that way, residual x 0.3-0.5 each).  The gradients, however, are exact: while the working set of a row stays the same
// from one iteration to the next, s = x - x_prev and y = g - g_prev on it are a secant pair of the true Hessian, and the
// fresh block A is corrected by the BFGS formula  A <- A - (A s)(A s)^T / (s^T A s) + y y^T / (y^T s)  for the last
// (at most two) pairs, oldest first -- positive definite as long as y^T s > 0, exact along the last steps.
//   F / gF: this iteration's working set and gradient on it (k_select);  H: the blocks (lower tiles, pitch 32 mt), the update
//   is applied to H in the units newton_solve reads (A = s1 H - s2 g g^T);  state: previous working set, x and g on it, pairs.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_secant(const int *__restrict__ rows, const int *__restrict__ F, const int *__restrict__ msz, int cap,
                                                const double *__restrict__ X, int64_t Qp, const double *__restrict__ gF, double *__restrict__ H,
                                                const long long *__restrict__ hoff, const int *__restrict__ mt, const double *__restrict__ s1,
                                                double s2, const double *__restrict__ ynoise, int *__restrict__ Fprev, int *__restrict__ mprev,
                                                double *__restrict__ xprev, double *__restrict__ gprev, double *__restrict__ S,
                                                double *__restrict__ Y, int *__restrict__ npairs, int64_t pair_stride,
                                                int apply_above /* blocks of up to this many entries are corrected by their solve kernel, in LDS */) {
    constexpr int L = 2;
    const int r = rows[blockIdx.x], tid = threadIdx.x;
    const int m = msz[r];
    if (m == 0) return;
    const int64_t base = (int64_t)r * cap;
    __shared__ double sS[512], sY[512], sV[512], sG[512], red[4];
    __shared__ int redi[4];
    // same working set as last time?
    int diff = m != mprev[r];
    for (int a = tid; a < m; a += 256) diff |= Fprev[base + a] != F[base + a];
    diff = block_sum_i(diff, redi);
    int np = diff ? 0 : npairs[r];
    double ss = 0, yy = 0, ys = 0, ymax = 0;
    for (int a = tid; a < m; a += 256) {
        const double xa = X[(int64_t)r * Qp + F[base + a]], ga = gF[base + a];
        const double sa = xa - xprev[base + a], ya = ga - gprev[base + a];
        sS[a] = sa;
        sY[a] = ya;
        sG[a] = s2 != 0.0 ? ga : 0.0;
        ss += sa * sa;
        yy += ya * ya;
        ys += sa * ya;
        ymax = fmax(ymax, fabs(ya));
        xprev[base + a] = xa;
        gprev[base + a] = ga;
        Fprev[base + a] = F[base + a];
    }
    ss = block_sum(ss, red);
    yy = block_sum(yy, red);
    ys = block_sum(ys, red);
    ymax = block_max(ymax, red);
    if (!diff && ss > 0.0 && ys > 1e-4 * sqrt(ss * yy) && ymax > ynoise[r]) { // a usable pair: keep the last L
        if (np == L) {
            for (int l = 0; l + 1 < L; ++l)
                for (int a = tid; a < m; a += 256) {
                    S[l * pair_stride + base + a] = S[(l + 1) * pair_stride + base + a];
                    Y[l * pair_stride + base + a] = Y[(l + 1) * pair_stride + base + a];
                }
            np = L - 1;
        }
        for (int a = tid; a < m; a += 256) {
            S[np * pair_stride + base + a] = sS[a];
            Y[np * pair_stride + base + a] = sY[a];
        }
        ++np;
    }
    __syncthreads();
    if (tid == 0) {
        npairs[r] = np;
        mprev[r] = m;
    }
    if (np == 0 || m <= apply_above) return;
    const int hp = 32 * mt[r];
    double *A = H + hoff[r];
    const double sc = s1[r];
    for (int l = 0; l < np; ++l) {
        __syncthreads();
        for (int a = tid; a < m; a += 256) {
            sS[a] = S[l * pair_stride + base + a];
            sY[a] = Y[l * pair_stride + base + a];
        }
        __syncthreads();
        double gs = 0;
        if (s2 != 0.0) {
            for (int a = tid; a < m; a += 256) gs += sG[a] * sS[a];
            gs = block_sum(gs, red);
        }
        double sAs = 0, ysl = 0;
        for (int i = tid; i < m; i += 256) {
            double v = 0;
            for (int j = 0; j <= i; ++j) v = fma(A[(int64_t)i * hp + j], sS[j], v);
            for (int j = i + 1; j < m; ++j) v = fma(A[(int64_t)j * hp + i], sS[j], v);
            v = sc * v - s2 * sG[i] * gs;
            sV[i] = v;
            sAs += sS[i] * v;
            ysl += sS[i] * sY[i];
        }
        sAs = block_sum(sAs, red);
        ysl = block_sum(ysl, red);
        if (!(sAs > 0.0 && ysl > 0.0)) continue; // (uniform)
        const double ia = 1.0 / (sAs * sc), iy = 1.0 / (ysl * sc);
        for (int idx = tid; idx < m * m; idx += 256) {
            const int i = idx / m, j = idx - i * m;
            if (j <= i) A[(int64_t)i * hp + j] += sY[i] * sY[j] * iy - sV[i] * sV[j] * ia;
        }
    }
}
void launch_secant(const int *drows, int nrows, const int *F, const int *msz, int cap, const double *X, int64_t Qp, const double *gF, double *H,
                   const long long *hoff, const int *mt, const double *s1, double s2, const double *ynoise, int *Fprev, int *mprev, double *xprev,
                   double *gprev, double *S, double *Y, int *npairs, int64_t pair_stride, int apply_above, hipStream_t st) {
    if (nrows > 0)
        hipLaunchKernelGGL(k_secant, dim3((unsigned)nrows), dim3(256), 0, st, drows, F, msz, cap, X, Qp, gF, H, hoff, mt, s1, s2, ynoise, Fprev, mprev,
                           xprev, gprev, S, Y, npairs, pair_stride, apply_above);
}

// ------------------------------------------------------------------------------------------
// Trial point of the projected (orthant-wise) line search: xt = P(x + alpha d), where a penalised coordinate that
// would cross zero is clipped to it (orthant face: the sign of x, or of -pg at zero).  out: dd = pg . (xt - x) (the
// directional derivative of F along the projected step), stepn = |xt - x|_1, l1t = lambda sum |xt_c|.
// When the projected step is not a descent direction (dd >= 0: a coordinate's step crossed zero and was clipped while
// the others still carry the moves that were meant to accompany it) only the clipping is taken.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trial(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ D,
                                               const double *__restrict__ PG, const uint8_t *__restrict__ kind, int64_t Qp, double lambda,
                                               const double *__restrict__ alpha, double *__restrict__ Xt, TrialOut *__restrict__ out,
                                               double *__restrict__ stepn /* [rows] or NULL: |xt - x|_1 again, for the trial pass on the device */) {
    const int r = rows[blockIdx.x];
    const int tid = threadIdx.x;
    const double *x = X + (int64_t)r * Qp, *d = D + (int64_t)r * Qp, *pg = PG + (int64_t)r * Qp;
    const uint8_t *kr = kind + (int64_t)r * Qp;
    double *xt = Xt + (int64_t)r * Qp;
    const double al = alpha[r];
    __shared__ double red[4];
    double dd = 0, sn = 0, l1 = 0;
    for (int64_t c = tid; c < Qp; c += 256) {
        const double xc = x[c];
        double v = xc;
        const uint8_t k = kr[c];
        if (k && d[c] != 0.0) {
            v = xc + al * d[c];
            if (k == 2 && lambda > 0) {
                const double xi = xc != 0.0 ? (xc > 0 ? 1.0 : -1.0) : (pg[c] < 0 ? 1.0 : -1.0);
                if (v * xi < 0) v = 0.0;
            }
            dd += pg[c] * (v - xc);
            sn += fabs(v - xc);
        }
        if (k == 2) l1 += lambda * fabs(v);
        xt[c] = v;
    }
    dd = block_sum(dd, red);
    if (!(dd < 0)) { // uniform over the workgroup
        dd = 0;
        sn = 0;
        l1 = 0;
        for (int64_t c = tid; c < Qp; c += 256) {
            const double xc = x[c];
            double v = xc;
            const uint8_t k = kr[c];
            if (k == 2 && lambda > 0 && xc != 0.0 && d[c] != 0.0 && (xc + al * d[c]) * xc < 0) {
                v = 0.0;
                dd += pg[c] * (0.0 - xc);
                sn += fabs(xc);
            }
            if (k == 2) l1 += lambda * fabs(v);
            xt[c] = v;
        }
        dd = block_sum(dd, red);
    }
    sn = block_sum(sn, red);
    l1 = block_sum(l1, red);
    if (tid == 0) {
        TrialOut o;
        o.dd = dd;
        o.stepn = sn;
        o.l1t = l1;
        o.back = 0.0;
        out[r] = o;
        if (stepn) stepn[r] = sn;
    }
}
void launch_trial(const int *drows, int nrows, const double *X, const double *D, const double *PG, const uint8_t *kind, int64_t Qp,
                  double lambda, const double *alpha, double *Xt, TrialOut *out, double *stepn, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_trial, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, D, PG, kind, Qp, lambda, alpha, Xt, out, stepn);
}

// back[r] = F'(xt; x - xt): directional derivative of F at the trial point back towards x (gradient Gt at xt).  F is
// convex, so back >= 0 implies F(xt) <= F(x): the acceptance test when function values are below their noise.
__global__ __launch_bounds__(256) void k_back(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ Xt,
                                              const double *__restrict__ Gt, const uint8_t *__restrict__ kind, int64_t Qp, double lambda,
                                              TrialOut *__restrict__ out) {
    const int r = rows[blockIdx.x];
    const double *x = X + (int64_t)r * Qp, *xt = Xt + (int64_t)r * Qp, *gt = Gt + (int64_t)r * Qp;
    const uint8_t *kr = kind + (int64_t)r * Qp;
    __shared__ double red[4];
    double back = 0;
    for (int64_t c = threadIdx.x; c < Qp; c += 256) {
        const double sc = x[c] - xt[c]; // direction back to x
        if (sc == 0.0 || !kr[c]) continue;
        double gl = gt[c] * sc;
        if (kr[c] == 2) gl += lambda * (xt[c] != 0.0 ? (xt[c] > 0 ? sc : -sc) : fabs(sc));
        back += gl;
    }
    back = block_sum(back, red);
    if (threadIdx.x == 0) out[r].back = back;
}
void launch_back(const int *drows, int nrows, const double *X, const double *Xt, const double *Gt, const uint8_t *kind, int64_t Qp,
                 double lambda, TrialOut *out, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_back, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, Xt, Gt, kind, Qp, lambda, out);
}

// ------------------------------------------------------------------------------------------
// Matrix-free Newton-CG on the rows whose working set is too large for a Cholesky block (dense optima: lambda at the
// level of the sampling noise): solve H_WW d = -pg_W by preconditioned conjugate gradients, W = {c : x_c != 0 or
// pg_c != 0}, with H p from the device operator (two GEMM passes, i8_pass with hv = 1).
//
// Preconditioner: block-diagonal.  W is listed in column order and cut into tiles of T consecutive entries; each tile's
// (sub-sampled) Hessian block is inverted once per Newton iteration (k_tile_inverse) and applied as a dense T x T product
// per CG step (k_tile_apply).  Why tiles of neighbours: the statistics are products of spins, so
// H[(j,k),(l,m)] = E_h[s_j s_k s_l s_m] is largest between statistics that share a spin or pair up correlated ones -- and
// the column order (sorted keys) puts those next to each other.  The common diagonal alone (every H_cc = sum_k h_k) or the
// block of the |W|/14 strongest entries leaves CG at the condition number of the spin correlations squared (measured on a
// 256-spin analogue of config 5: 11 steps to a 5 % residual, 29 to 1e-3; tiles of 64: 6 / 16; of 128: 5 / 13).
// Vectors live in [rows][Qp] arrays; the per-row scalars in CgState.
// ------------------------------------------------------------------------------------------
// W of the listed rows in column order, in tiles: FV[(t0[r] T + a)] = a-th column of W, gV = its gradient entry (logRISE:
// the rank-one term of the block); the last tile of a row is padded with the all-plus column Qp - 1 (never used: the
// inverse and the product stop at the tile's entry count).
__global__ __launch_bounds__(256) void k_cg_tiles(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ PG,
                                                  const double *__restrict__ G, const uint8_t *__restrict__ kind, int64_t Qp, int T,
                                                  const long long *__restrict__ t0, int *__restrict__ FV, double *__restrict__ gV) {
    const int r = rows[blockIdx.x], tid = threadIdx.x;
    const int64_t base = (int64_t)r * Qp;
    __shared__ int scan[257];
    const int64_t chunk = (Qp + 255) / 256, c0 = tid * chunk, c1 = c0 + chunk < Qp ? c0 + chunk : Qp;
    int cnt = 0;
    for (int64_t c = c0; c < c1; ++c) cnt += kind[base + c] && (X[base + c] != 0.0 || PG[base + c] != 0.0);
    scan[tid + 1] = cnt;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= 256; ++t) scan[t] += scan[t - 1];
    __syncthreads();
    int *fv = FV + t0[r] * T;
    double *gv = gV + t0[r] * T;
    int pos = scan[tid];
    for (int64_t c = c0; c < c1; ++c)
        if (kind[base + c] && (X[base + c] != 0.0 || PG[base + c] != 0.0)) {
            fv[pos] = (int)c;
            gv[pos] = G[base + c];
            ++pos;
        }
    const int m = scan[256], mp = (m + T - 1) / T * T;
    for (int a = m + tid; a < mp; a += 256) {
        fv[a] = (int)(Qp - 1);
        gv[a] = 0.0;
    }
}
void launch_cg_tiles(const int *drows, int nrows, const double *X, const double *PG, const double *G, const uint8_t *kind, int64_t Qp, int T,
                     const long long *t0, int *FV, double *gV, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_cg_tiles, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, PG, G, kind, Qp, T, t0, FV, gV);
}

// z_tile = Minv_tile r_tile for every tile of a live row (one workgroup per tile; Minv symmetric, pitch T, read by columns:
// coalesced)
template <int T>
__global__ __launch_bounds__(256) void k_tile_apply(const double *__restrict__ Minv, const int *__restrict__ FV, const int *__restrict__ vm,
                                                    const int *__restrict__ wrow, const int *__restrict__ live, int64_t Qp,
                                                    const double *__restrict__ Rv, double *__restrict__ Zv) {
    const int64_t v = blockIdx.x;
    const int wr = wrow[v];
    if (!live[wr]) return;
    const int m = vm[v], tid = threadIdx.x;
    constexpr int NH = 256 / T; // threads per output entry
    __shared__ double rt[T], part[256];
    const int *fv = FV + v * T;
    if (tid < T) rt[tid] = tid < m ? Rv[(int64_t)wr * Qp + fv[tid]] : 0.0;
    __syncthreads();
    const int i = tid % T, hf = tid / T;
    const double *Mi = Minv + v * T * T;
    double s0 = 0.0, s1 = 0.0;
    const int j0 = hf * (T / NH), j1 = min(m, j0 + T / NH);
    int j = j0;
    for (; j + 1 < j1; j += 2) {
        s0 = fma(Mi[(int64_t)j * T + i], rt[j], s0);
        s1 = fma(Mi[(int64_t)(j + 1) * T + i], rt[j + 1], s1);
    }
    if (j < j1) s0 = fma(Mi[(int64_t)j * T + i], rt[j], s0);
    part[tid] = s0 + s1;
    __syncthreads();
    if (tid < T && tid < m) {
        double z = part[tid];
#pragma unroll
        for (int q = 1; q < NH; ++q) z += part[tid + q * T];
        Zv[(int64_t)wr * Qp + fv[tid]] = z;
    }
}
void launch_tile_apply(int T, const double *Minv, const int *FV, const int *vm, const int *wrow, const int *live, int64_t ntiles, int64_t Qp,
                       const double *Rv, double *Zv, hipStream_t st) {
    if (ntiles <= 0) return;
    if (T == 64) hipLaunchKernelGGL(k_tile_apply<64>, dim3((unsigned)ntiles), dim3(256), 0, st, Minv, FV, vm, wrow, live, Qp, Rv, Zv);
    else hipLaunchKernelGGL(k_tile_apply<128>, dim3((unsigned)ntiles), dim3(256), 0, st, Minv, FV, vm, wrow, live, Qp, Rv, Zv);
}

// r = -pg on W, d = 0, z = 0, p = 0;  Wm = the mask of W (k_pcg_faces shrinks it).  The vectors are dense [Qp] arrays that are zero
// outside W: the per-step kernels below walk the row's LIST of W (k_cg_tiles) and never touch the rest.
__global__ __launch_bounds__(256) void k_pcg_init(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ PG,
                                                  const uint8_t *__restrict__ kind, int64_t Qp, double *__restrict__ D, double *__restrict__ Rv,
                                                  double *__restrict__ Zv, double *__restrict__ Pv, uint8_t *__restrict__ Wm,
                                                  CgState *__restrict__ cg) {
    const int r = rows[blockIdx.x];
    __shared__ double red[4];
    double rs = 0;
    for (int64_t c = threadIdx.x; c < Qp; c += 256) {
        const int64_t i = (int64_t)r * Qp + c;
        const bool inW = kind[i] && (X[i] != 0.0 || PG[i] != 0.0);
        const double v = inW ? -PG[i] : 0.0;
        Rv[i] = v;
        D[i] = 0.0;
        Zv[i] = 0.0;
        Pv[i] = 0.0;
        Wm[i] = inW;
        rs += v * v;
    }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) {
        cg[r].rs = rs;
        cg[r].rs0 = rs;
        cg[r].pHp = 0.0;
        cg[r].rz = 0.0;
    }
}
void launch_pcg_init(const int *drows, int nrows, const double *X, const double *PG, const uint8_t *kind, int64_t Qp, double *D, double *Rv,
                     double *Zv, double *Pv, uint8_t *Wm, CgState *cg, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_pcg_init, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, PG, kind, Qp, D, Rv, Zv, Pv, Wm, cg);
}

// With z = M^-1 r in Zv (k_tile_apply over the tiles of the original W; the preconditioner of the current, possibly smaller,
// W is its restriction: z is masked):  beta = r.z / (r.z)_old (0 on the first call);  p = z + beta p.
__global__ __launch_bounds__(256) void k_pcg_dir(const int *__restrict__ rows, int64_t Qp, const uint8_t *__restrict__ Wm,
                                                 const double *__restrict__ Rv, const double *__restrict__ Zv, double *__restrict__ Pv, int first,
                                                 CgState *__restrict__ cg, const WList wl) {
    const int r = rows[blockIdx.x];
    const int64_t base = (int64_t)r * Qp;
    const int *fv = wl.FV + wl.t0[r] * wl.T;
    const int m = wl.nw[r];
    __shared__ double red[4];
    const double rzo = cg[r].rz;
    double rz = 0;
    for (int a = threadIdx.x; a < m; a += 256) {
        const int64_t i = base + fv[a];
        if (Wm[i]) rz += Rv[i] * Zv[i];
    }
    rz = block_sum(rz, red);
    const double be = (!first && rzo > 0) ? rz / rzo : 0.0;
    for (int a = threadIdx.x; a < m; a += 256) {
        const int64_t i = base + fv[a];
        Pv[i] = Wm[i] ? Zv[i] + (first ? 0.0 : be * Pv[i]) : 0.0;
    }
    if (threadIdx.x == 0) cg[r].rz = rz;
}
void launch_pcg_dir(const int *drows, int nrows, int64_t Qp, const uint8_t *Wm, const double *Rv, const double *Zv, double *Pv, int first,
                    CgState *cg, const WList &wl, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_pcg_dir, dim3((unsigned)nrows), dim3(256), 0, st, drows, Qp, Wm, Rv, Zv, Pv, first, cg, wl);
}

// Orthant faces.  The Newton system is solved on W without its sign constraints; the line search then projects the step onto
// the orthant of the iterate (k_trial): a coordinate at zero may only move against its pseudo-gradient, a non-zero one not
// past zero.  With correlated statistics the unconstrained solution is full of large moves that cancel each other; clipping
// one of a pair leaves the other uncompensated and the projected step climbs (config 5 at the default regulariser: a
// quarter of the rows accepted only alpha = 1/16 .. 1/64).  So after the CG solve the coordinates whose step leaves the face
// are fixed where the projection would put them -- at zero -- and removed from W, and the system is solved again for the
// others (bound-constrained QP by an active-set CG): out = {number fixed, their share of the predicted decrease}.
__global__ __launch_bounds__(256) void k_pcg_faces(const int *__restrict__ rows, const double *__restrict__ X, const double *__restrict__ PG,
                                                   const uint8_t *__restrict__ kind, int64_t Qp, double *__restrict__ D,
                                                   uint8_t *__restrict__ Wm, FaceOut *__restrict__ out) {
    const int r = rows[blockIdx.x];
    const int64_t base = (int64_t)r * Qp;
    __shared__ double red[4];
    __shared__ int redi[4];
    int nf = 0;
    double mass = 0, total = 0;
    for (int64_t c = threadIdx.x; c < Qp; c += 256) {
        const int64_t i = base + c;
        if (!Wm[i]) continue;
        const double x = X[i], dc = D[i], pg = PG[i];
        total += fabs(pg * dc);
        if (kind[i] != 2) continue;
        if (x == 0.0 ? dc * pg > 0.0 : (x + dc) * x < 0.0) {
            const double fixed = x == 0.0 ? 0.0 : -x;
            mass += fabs(pg * (dc - fixed));
            D[i] = fixed;
            Wm[i] = 0;
            ++nf;
        }
    }
    nf = block_sum_i(nf, redi);
    mass = block_sum(mass, red);
    total = block_sum(total, red);
    if (threadIdx.x == 0) {
        FaceOut o;
        o.nfixed = nf;
        o.pad = 0;
        o.mass = mass;
        o.total = total;
        out[r] = o;
    }
}
void launch_pcg_faces(const int *drows, int nrows, const double *X, const double *PG, const uint8_t *kind, int64_t Qp, double *D, uint8_t *Wm,
                      FaceOut *out, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_pcg_faces, dim3((unsigned)nrows), dim3(256), 0, st, drows, X, PG, kind, Qp, D, Wm, out);
}

// Residual of the shrunk system: given Hd = (sum_k h_k x_k x_k^T) d,  r = -pg - (s1 Hd - s2 g (g . d)) on W, 0 elsewhere
__global__ __launch_bounds__(256) void k_pcg_resid(const int *__restrict__ rows, const double *__restrict__ PG, const double *__restrict__ G,
                                                   int64_t Qp, const double *__restrict__ s1, double s2, const double *__restrict__ Hd,
                                                   const double *__restrict__ D, const uint8_t *__restrict__ Wm, double *__restrict__ Rv,
                                                   CgState *__restrict__ cg) {
    const int r = rows[blockIdx.x];
    const int64_t base = (int64_t)r * Qp;
    __shared__ double red[4];
    double gd = 0;
    if (s2 != 0.0) {
        for (int64_t c = threadIdx.x; c < Qp; c += 256) gd += G[base + c] * D[base + c];
        gd = block_sum(gd, red);
    }
    const double sc = s1[r];
    double rs = 0;
    for (int64_t c = threadIdx.x; c < Qp; c += 256) {
        const int64_t i = base + c;
        const double v = Wm[i] ? -PG[i] - (sc * Hd[i] - s2 * G[i] * gd) : 0.0;
        Rv[i] = v;
        rs += v * v;
    }
    rs = block_sum(rs, red);
    if (threadIdx.x == 0) {
        cg[r].rs = rs;
        cg[r].pHp = 0.0;
        cg[r].rz = 0.0;
    }
}
void launch_pcg_resid(const int *drows, int nrows, const double *PG, const double *G, int64_t Qp, const double *s1, double s2, const double *Hd,
                      const double *D, const uint8_t *Wm, double *Rv, CgState *cg, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_pcg_resid, dim3((unsigned)nrows), dim3(256), 0, st, drows, PG, G, Qp, s1, s2, Hd, D, Wm, Rv, cg);
}

// Given Hp = (sum_k h_k x_k x_k^T) p from the device:
//   Hp <- s1[r] * Hp - s2 * g (g . p), restricted to W (mask Wm)   (logRISE: Hess log Z = Hess Z / Z - g g^T, s1 = 1/Z, s2 = 1)
//   alpha = r.z / p.Hp;  d += alpha p;  r -= alpha Hp;  rs = r.r
__global__ __launch_bounds__(256) void k_pcg_step(const int *__restrict__ rows, const double *__restrict__ G, const uint8_t *__restrict__ Wm,
                                                  int64_t Qp, const double *__restrict__ s1, double s2, double *__restrict__ Hp,
                                                  double *__restrict__ D, double *__restrict__ Rv, const double *__restrict__ Pv,
                                                  CgState *__restrict__ cg, const WList wl) {
    const int r = rows[blockIdx.x];
    const int64_t base = (int64_t)r * Qp;
    const int *fv = wl.FV + wl.t0[r] * wl.T;
    const int m = wl.nw[r];
    __shared__ double red[4];
    const double rzo = cg[r].rz;
    double gp = 0;
    if (s2 != 0.0) {
        for (int a = threadIdx.x; a < m; a += 256) {
            const int64_t i = base + fv[a];
            gp += G[i] * Pv[i];
        }
        gp = block_sum(gp, red);
    }
    const double sc = s1[r];
    double pHp = 0;
    for (int a = threadIdx.x; a < m; a += 256) {
        const int64_t i = base + fv[a];
        const double h = Wm[i] ? sc * Hp[i] - s2 * G[i] * gp : 0.0;
        Hp[i] = h;
        pHp += Pv[i] * h;
    }
    pHp = block_sum(pHp, red);
    const double al = pHp > 0 ? rzo / pHp : 0.0;
    double rsn = 0;
    for (int a = threadIdx.x; a < m; a += 256) {
        const int64_t i = base + fv[a];
        D[i] += al * Pv[i];
        const double rv = Rv[i] - al * Hp[i];
        Rv[i] = rv;
        rsn += rv * rv;
    }
    rsn = block_sum(rsn, red);
    if (threadIdx.x == 0) {
        cg[r].rs = rsn;
        cg[r].pHp = pHp;
    }
}
void launch_pcg_step(const int *drows, int nrows, const double *G, const uint8_t *Wm, int64_t Qp, const double *s1, double s2, double *Hp,
                     double *D, double *Rv, const double *Pv, CgState *cg, const WList &wl, hipStream_t st) {
    if (nrows > 0) hipLaunchKernelGGL(k_pcg_step, dim3((unsigned)nrows), dim3(256), 0, st, drows, G, Wm, Qp, s1, s2, Hp, D, Rv, Pv, cg, wl);
}

} // namespace gml
