/*
 * gml_oracle.c -- CPU restatement of GraphicalModelLearning.jl's learn() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (graphicalmodellearning.jl_amd/,
 * the C-ABI library) may include, link or call this file.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * Parity pin: the reference itself (Julia + JuMP + Ipopt) cannot run in this image, so the
 * oracle is pinned against the reference's own golden vectors
 * (test/data/{a,b,c,mvt}_{RISE,logRISE,RPLE}_learned.csv, copied to tests/golden/) by
 * tests/test_oracle_golden.py.  multiRISE at order >= 3 has no golden in the reference
 * ("parity unpinned" there; only the reference's cross-formulation check
 * multiRISE(c,false,2) == RISE(c,false), test/runtests.jl:132-158, pins it).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference/src/GraphicalModelLearning.jl unless another file is named.
 *
 * Plain C99, FP64, no dependencies.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GML_RISE 0
#define GML_LOGRISE 1
#define GML_RPLE 2

/* ------------------------------------------------------------------------------------------
 * lambda = c * sqrt(log(n^2 / 0.05) / M)            (:157, same at :86, :213, :266, :304)
 * ---------------------------------------------------------------------------------------- */
double gml_oracle_lambda(double c, int64_t n, double M) {
    return c * sqrt(log(((double)n * (double)n) / 0.05) / M);
}

/* ------------------------------------------------------------------------------------------
 * Multi-body key enumeration.
 * Keys of node u (0-based here, 1-based in the reference), in the order the reference
 * builds them (:94-104 with models.jl:228-246 `permutations`, which yields strictly
 * ascending tuples sorted lexicographically):
 *   p = 1 : (u)
 *   p = 2 : (u, i)        i ascending over {0..n-1} \ {u}
 *   p = 3 : (u, i, j)     i < j ascending over the same set, lexicographic
 *   ...
 * keys is written as P rows of `order` int32, unused trailing slots = -1.
 * ---------------------------------------------------------------------------------------- */
static int64_t binom(int64_t n, int64_t k) {
    if (k < 0 || k > n) return 0;
    int64_t r = 1;
    for (int64_t i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
}

int64_t gml_oracle_multi_nparams(int64_t n, int order) {
    int64_t P = 0;
    for (int p = 1; p <= order; ++p) P += binom(n - 1, p - 1);
    return P;
}

void gml_oracle_multi_keys(int64_t n, int order, int64_t u, int32_t *keys) {
    int64_t row = 0;
    int32_t nb[4096];
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i)
        if (i != u) nb[m++] = (int32_t)i;
    for (int p = 1; p <= order; ++p) {
        int q = p - 1; /* size of the subset of neighbours */
        if (q > m) break;
        int idx[16];
        for (int t = 0; t < q; ++t) idx[t] = t;
        for (;;) {
            int32_t *k = keys + row * order;
            for (int t = 0; t < order; ++t) k[t] = -1;
            k[0] = (int32_t)u;
            for (int t = 0; t < q; ++t) k[1 + t] = nb[idx[t]];
            ++row;
            /* next combination in lexicographic order */
            int t = q - 1;
            while (t >= 0 && idx[t] == (int)m - q + t) --t;
            if (t < 0) break;
            ++idx[t];
            for (int s = t + 1; s < q; ++s) idx[s] = idx[s - 1] + 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Node statistics.
 * Pairwise  (:162): stat[k,i] = s_u^k * (i == u ? 1 : s_i^k)
 * Multibody (:107): stat[k,key] = prod_{i in key} s_i^k     (key[0] == u)
 * A row is materialised into `row` (length P) -- the oracle never holds the K x P matrix
 * except inside the small-n Newton solver's Hessian accumulation, which is row-at-a-time too.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t K, n, P;
    const double *counts; /* K   (samples[:,1])          */
    const int8_t *spins;  /* K*n row-major (samples[:,2:end]) */
    int64_t u;
    int order;           /* 0 => pairwise layout (P == n, slot i <-> spin i, slot u = field) */
    const int32_t *keys; /* P*order when order > 0 */
    double M;            /* sum(counts)  (:79) */
} node_t;

static inline void stat_row(const node_t *nd, int64_t k, double *row) {
    const int8_t *s = nd->spins + k * nd->n;
    double su = (double)s[nd->u];
    if (nd->order == 0) {
        for (int64_t i = 0; i < nd->n; ++i) row[i] = su * (double)s[i];
        row[nd->u] = su;
    } else {
        for (int64_t j = 0; j < nd->P; ++j) {
            const int32_t *key = nd->keys + j * nd->order;
            double p = 1.0;
            for (int t = 0; t < nd->order && key[t] >= 0; ++t) p *= (double)s[key[t]];
            row[j] = p;
        }
    }
}

/* pointwise pieces of the three objectives.
 * RISE    (:169-172, explicit form :191-208):  sum_k w_k exp(-E_k)
 * logRISE (:278-281):                           log sum_k w_k exp(-E_k)
 * RPLE    (:316-319):                           sum_k w_k log(1 + exp(-2 E_k))
 */
static inline double softplus_m2(double E) { /* log(1+exp(-2E)), stable */
    double t = -2.0 * E;
    return t > 0 ? t + log1p(exp(-t)) : log1p(exp(t));
}

/* f, g (length P) and optionally H (P x P, row-major, full) at theta.  Smooth part only. */
static void node_eval(const node_t *nd, int form, const double *theta, double *f, double *g,
                      double *H, double *row /* scratch P */) {
    const int64_t K = nd->K, P = nd->P;
    memset(g, 0, sizeof(double) * P);
    if (H) memset(H, 0, sizeof(double) * P * P);
    double shift = 0.0;
    if (form == GML_LOGRISE) { /* max-shift for the log-sum-exp */
        double mx = -INFINITY;
        for (int64_t k = 0; k < K; ++k) {
            if (nd->counts[k] <= 0) continue;
            stat_row(nd, k, row);
            double E = 0;
            for (int64_t j = 0; j < P; ++j) E += theta[j] * row[j];
            if (-E > mx) mx = -E;
        }
        shift = mx;
    }
    double acc = 0.0;
    for (int64_t k = 0; k < K; ++k) {
        double w = nd->counts[k] / nd->M; /* samples[k,1]/num_samples (:170) */
        if (w == 0) continue;
        stat_row(nd, k, row);
        double E = 0;
        for (int64_t j = 0; j < P; ++j) E += theta[j] * row[j];
        double a, h; /* a: d f_k / dE (negated weight on row), h: second derivative weight */
        if (form == GML_RPLE) {
            double s = 1.0 / (1.0 + exp(2.0 * E)); /* sigma(-2E) */
            acc += w * softplus_m2(E);
            a = 2.0 * w * s;
            h = 4.0 * w * s * (1.0 - s);
        } else {
            double e = w * exp(-E - shift);
            acc += e;
            a = e;
            h = e;
        }
        for (int64_t j = 0; j < P; ++j) g[j] -= a * row[j]; /* g[i] = sum_k stat[k,i]*partial[k] (:204-207) */
        if (H) {
            for (int64_t i = 0; i < P; ++i) {
                double hi = h * row[i];
                double *Hi = H + i * P;
                for (int64_t j = 0; j <= i; ++j) Hi[j] += hi * row[j];
            }
        }
    }
    if (form == GML_LOGRISE) {
        /* f = log Z ; g = grad Z / Z ; H = Hess Z / Z - g g^T */
        double Z = acc;
        *f = log(Z) + shift;
        for (int64_t j = 0; j < P; ++j) g[j] /= Z;
        if (H)
            for (int64_t i = 0; i < P; ++i)
                for (int64_t j = 0; j <= i; ++j) H[i * P + j] = H[i * P + j] / Z - g[i] * g[j];
    } else {
        *f = acc;
    }
    if (H)
        for (int64_t i = 0; i < P; ++i)
            for (int64_t j = 0; j < i; ++j) H[j * P + i] = H[i * P + j];
}

/* The Hessian of the same smooth part on the free coordinates fr[0..m) only (m x m, row-major, full): what node_eval's H holds
 * at (fr[a], fr[b]).  The Newton step of node_solve uses nothing else of H, and K m^2 instead of K P^2 is what lets the
 * restatement solve order-3 problems of a few hundred parameters per node in seconds (tests: n = 36, P = 631).  g: the gradient
 * node_eval returned at theta (logRISE: Hess log Z = Hess Z / Z - g g^T). */
static void node_hess_free(const node_t *nd, int form, const double *theta, const double *g, const int64_t *fr, int64_t m,
                           double *HF, double *row /* scratch P */) {
    const int64_t K = nd->K, P = nd->P;
    memset(HF, 0, sizeof(double) * m * m);
    double shift = 0.0;
    if (form == GML_LOGRISE) {
        double mx = -INFINITY;
        for (int64_t k = 0; k < K; ++k) {
            if (nd->counts[k] <= 0) continue;
            stat_row(nd, k, row);
            double E = 0;
            for (int64_t j = 0; j < P; ++j) E += theta[j] * row[j];
            if (-E > mx) mx = -E;
        }
        shift = mx;
    }
    double acc = 0.0;
    double *rf = malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int64_t k = 0; k < K; ++k) {
        double w = nd->counts[k] / nd->M;
        if (w == 0) continue;
        stat_row(nd, k, row);
        double E = 0;
        for (int64_t j = 0; j < P; ++j) E += theta[j] * row[j];
        double h;
        if (form == GML_RPLE) {
            double sg = 1.0 / (1.0 + exp(2.0 * E));
            h = 4.0 * w * sg * (1.0 - sg);
        } else {
            h = w * exp(-E - shift);
            acc += h;
        }
        for (int64_t a = 0; a < m; ++a) rf[a] = row[fr[a]];
        for (int64_t a = 0; a < m; ++a) {
            double ha = h * rf[a];
            double *Ha = HF + a * m;
            for (int64_t b = 0; b <= a; ++b) Ha[b] += ha * rf[b];
        }
    }
    free(rf);
    if (form == GML_LOGRISE)
        for (int64_t a = 0; a < m; ++a)
            for (int64_t b = 0; b <= a; ++b) HF[a * m + b] = HF[a * m + b] / acc - g[fr[a]] * g[fr[b]];
    for (int64_t a = 0; a < m; ++a)
        for (int64_t b = 0; b < a; ++b) HF[b * m + a] = HF[a * m + b];
}

/* pseudo-gradient of f + lambda * sum_{j penalised} |x_j|  (minimum-norm subgradient) */
static inline double pseudo_grad(double x, double g, double lam) {
    if (lam == 0.0) return g;
    if (x > 0) return g + lam;
    if (x < 0) return g - lam;
    if (g + lam < 0) return g + lam;
    if (g - lam > 0) return g - lam;
    return 0.0;
}

/* dense Cholesky solve of A d = b on an m x m system (A destroyed); returns 0 on success */
static int chol_solve(double *A, double *b, int64_t m) {
    for (int64_t j = 0; j < m; ++j) {
        double d = A[j * m + j];
        for (int64_t k = 0; k < j; ++k) d -= A[j * m + k] * A[j * m + k];
        if (!(d > 0)) return 1;
        d = sqrt(d);
        A[j * m + j] = d;
        for (int64_t i = j + 1; i < m; ++i) {
            double s = A[i * m + j];
            for (int64_t k = 0; k < j; ++k) s -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = s / d;
        }
    }
    for (int64_t i = 0; i < m; ++i) {
        double s = b[i];
        for (int64_t k = 0; k < i; ++k) s -= A[i * m + k] * b[k];
        b[i] = s / A[i * m + i];
    }
    for (int64_t i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int64_t k = i + 1; k < m; ++k) s -= A[k * m + i] * b[k];
        b[i] = s / A[i * m + i];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * One node's l1-regularised solve.  Restates the optimisation problem the reference hands to
 * Ipopt (:164-181): min_x f(x) + lambda * sum_{j penalised} z_j, z_j >= |x_j|, i.e. the
 * l1-penalised convex problem -- solved here to its exact optimum by an orthant-wise
 * active-set Newton method (the reference's interior-point iterate differs from the optimum
 * by Ipopt's barrier residual; see tests/test_oracle_golden.py for the measured gap).
 * pen[j] = 1 if slot j is penalised (j != u for pairwise :171; len(key) > 1 for multi :118).
 * Returns the final max |pseudo-gradient| (KKT residual); *iters_out = Newton iterations.
 * ---------------------------------------------------------------------------------------- */
static double node_solve(const node_t *nd, int form, double lam, const uint8_t *pen, double *x,
                         double tol, int maxit, int *iters_out) {
    const int64_t P = nd->P;
    double *g = malloc(sizeof(double) * P), *H = malloc(sizeof(double) * P * P);
    double *row = malloc(sizeof(double) * P), *pg = malloc(sizeof(double) * P);
    double *HF = malloc(sizeof(double) * P * P), *d = malloc(sizeof(double) * P);
    double *xn = malloc(sizeof(double) * P), *gn = malloc(sizeof(double) * P);
    int64_t *fr = malloc(sizeof(int64_t) * P);
    double *xbest = malloc(sizeof(double) * P);
    double f, kkt = INFINITY, best = INFINITY, Fbest = INFINITY;
    int it, stall = 0;
    memcpy(xbest, x, sizeof(double) * P);
    for (it = 0; it < maxit; ++it) {
        node_eval(nd, form, x, &f, g, NULL, row);
        double F = f;
        kkt = 0;
        int64_t m = 0;
        for (int64_t j = 0; j < P; ++j) {
            double l = pen[j] ? lam : 0.0;
            F += l * fabs(x[j]);
            pg[j] = pseudo_grad(x[j], g[j], l);
            if (fabs(pg[j]) > kkt) kkt = fabs(pg[j]);
            if (x[j] != 0.0 || pg[j] != 0.0) fr[m++] = j;
        }
        /* progress = a smaller KKT residual (beyond its last digits) or a smaller objective (beyond its summation noise): while
         * many coordinates still enter and leave the support -- order-3 problems with hundreds of violators -- the residual is
         * not monotone along a converging sequence, the objective is */
        const int fdown = F < Fbest - 1e-12 * fmax(1.0, fabs(F));
        if (F < Fbest) Fbest = F;
        if (kkt < best) {
            if (kkt < 0.99 * best || fdown) stall = 0;
            else ++stall;
            best = kkt;
            memcpy(xbest, x, sizeof(double) * P);
        } else if (fdown) {
            stall = 0;
        } else {
            ++stall;
        }
        if (stall >= 3) break; /* at the FP64 noise floor: keep the best iterate */
        if (kkt <= tol) break;
        node_hess_free(nd, form, x, g, fr, m, H, row); /* H: the m x m block on the free coordinates */
        double ridge = 0.0;
        for (;;) {
            for (int64_t a = 0; a < m; ++a) {
                for (int64_t b = 0; b < m; ++b) HF[a * m + b] = H[a * m + b];
                HF[a * m + a] += ridge;
                d[a] = -pg[fr[a]];
            }
            if (chol_solve(HF, d, m) == 0) break;
            ridge = ridge == 0.0 ? 1e-12 : ridge * 100.0;
        }
        /* projected backtracking line search on the full objective */
        double t = 1.0;
        int ok = 0;
        for (int ls = 0; ls < 60; ++ls, t *= 0.5) {
            memcpy(xn, x, sizeof(double) * P);
            double dd = 0.0;
            for (int64_t a = 0; a < m; ++a) {
                int64_t j = fr[a];
                double v = x[j] + t * d[a];
                if (pen[j] && lam > 0) {
                    double xi = x[j] != 0.0 ? (x[j] > 0 ? 1.0 : -1.0) : (pg[j] < 0 ? 1.0 : -1.0);
                    if (v * xi < 0) v = 0.0; /* crossed zero: clip to the orthant face */
                }
                xn[j] = v;
                dd += pg[j] * (xn[j] - x[j]);
            }
            double fn;
            node_eval(nd, form, xn, &fn, gn, NULL, row);
            double Fn = fn;
            for (int64_t j = 0; j < P; ++j)
                if (pen[j]) Fn += lam * fabs(xn[j]);
            /* Armijo with an FP64-noise allowance so full Newton steps survive near the optimum */
            if (Fn <= F + 1e-4 * dd + 4e-16 * fmax(1.0, fabs(F))) {
                ok = 1;
                break;
            }
            /* Below the summation noise of f (K terms: ~1e-13 relative at K = 4e4, where the expected decrease of a step at KKT
             * 1e-7 is 1e-14) function values cannot rank the trial: it is accepted iff its KKT residual, from the gradient
             * just evaluated, is smaller than the iterate's.  The reference's own fixtures (K <= 512) never get here. */
            if (-dd < 1e-11 * fmax(1.0, fabs(F)) && Fn <= F + 1e-11 * fmax(1.0, fabs(F))) {
                double kn = 0.0;
                for (int64_t j = 0; j < P; ++j) {
                    double pj = fabs(pseudo_grad(xn[j], gn[j], pen[j] ? lam : 0.0));
                    if (pj > kn) kn = pj;
                }
                if (kn < kkt) {
                    ok = 1;
                    break;
                }
            }
        }
        if (!ok) break; /* cannot improve within FP64 resolution */
        memcpy(x, xn, sizeof(double) * P);
    }
    if (iters_out) *iters_out = it;
    memcpy(x, xbest, sizeof(double) * P);
    kkt = best;
    free(xbest);
    free(g); free(H); free(row); free(pg); free(HF); free(d); free(xn); free(gn); free(fr);
    return kkt;
}

static double sum_counts(const double *c, int64_t K) { /* data_info (:76-81) */
    double M = 0;
    for (int64_t k = 0; k < K; ++k) M += c[k];
    return M;
}

/* ------------------------------------------------------------------------------------------
 * Public: objective + gradient for one pairwise node at theta (length n; slot u = field).
 * This is the reference's hand-written operator pair risea_obj / grad_risea_obj (:191-208),
 * extended to the logRISE (:278-281) and RPLE (:316-319) pointwise functions.
 * Smooth part only (no l1 term).
 * ---------------------------------------------------------------------------------------- */
void gml_oracle_objgrad_pair(int form, int64_t K, int64_t n, const double *counts,
                             const int8_t *spins, int64_t u, const double *theta, double *f,
                             double *g) {
    node_t nd = {K, n, n, counts, spins, u, 0, NULL, sum_counts(counts, K)};
    double *row = malloc(sizeof(double) * n);
    node_eval(&nd, form, theta, f, g, NULL, row);
    free(row);
}

/* Fast variant of the same math for the CPU baseline timing (RISE only): no row
 * materialisation, energies straight from the int8 spins.  `nodes` lists which nodes to
 * evaluate; theta/g are nn x n row-major.  OpenMP over nodes when built with -fopenmp. */
void gml_oracle_objgrad_rise_nodes(int64_t K, int64_t n, const double *counts,
                                   const int8_t *spins, const int64_t *nodes, int64_t nn,
                                   const double *theta, double *f, double *g) {
    const double M = sum_counts(counts, K);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t a = 0; a < nn; ++a) {
        const int64_t u = nodes[a];
        const double *th = theta + a * n;
        double *ga = g + a * n;
        double *acc = calloc((size_t)n, sizeof(double));
        double facc = 0.0, gu = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            const int8_t *s = spins + k * n;
            double dot = 0.0;
            for (int64_t i = 0; i < n; ++i) dot += th[i] * (double)s[i];
            double su = (double)s[u];
            /* E = su * (dot - th[u]*su) + th[u]*su  :  slot u holds the field (:162) */
            double E = su * (dot - th[u] * su + th[u]);
            double e = (counts[k] / M) * exp(-E);
            facc += e;
            double v = e * su;
            for (int64_t i = 0; i < n; ++i) acc[i] += v * (double)s[i];
            gu += v;
        }
        for (int64_t i = 0; i < n; ++i) ga[i] = -acc[i];
        ga[u] = -gu;
        f[a] = facc;
        free(acc);
    }
}

/* ------------------------------------------------------------------------------------------
 * Public: learn() for the pairwise formulations  (:154-189 RISE, :263-298 logRISE,
 * :301-336 RPLE).  out is n x n ROW-major: out[u*n + i] = reconstruction[u, i] (:181),
 * diagonal = fields.  Symmetrisation 0.5*(R + R^T) when requested (:184-186).
 * kkt[u] = final max|pseudo-gradient| of node u.  Returns max over nodes of kkt.
 * ---------------------------------------------------------------------------------------- */
double gml_oracle_learn_pair(int form, int64_t K, int64_t n, const double *counts,
                             const int8_t *spins, double c, int symmetrize, double tol,
                             double *out, double *kkt, int32_t *iters) {
    const double M = sum_counts(counts, K);
    const double lam = gml_oracle_lambda(c, n, M);
    double worst = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t u = 0; u < n; ++u) {
        node_t nd = {K, n, n, counts, spins, u, 0, NULL, M};
        uint8_t *pen = malloc((size_t)n);
        for (int64_t j = 0; j < n; ++j) pen[j] = (j != u); /* current_spin != j (:171) */
        double *x = calloc((size_t)n, sizeof(double));
        int it = 0;
        double r = node_solve(&nd, form, lam, pen, x, tol, 500, &it);
        memcpy(out + u * n, x, sizeof(double) * n);
        if (kkt) kkt[u] = r;
        if (iters) iters[u] = it;
#pragma omp critical
        if (r > worst) worst = r;
        free(pen);
        free(x);
    }
    if (symmetrize) {
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = i + 1; j < n; ++j) {
                double m = 0.5 * (out[i * n + j] + out[j * n + i]);
                out[i * n + j] = out[j * n + i] = m;
            }
    }
    return worst;
}

/* ------------------------------------------------------------------------------------------
 * Public: multi-body objective/gradient and learn  (multiRISE, :83-152).
 * theta/out are per node P-vectors in the key order of gml_oracle_multi_keys.
 * Symmetrisation (:135-149: group by sorted key, mean) is host glue and lives in the caller.
 * ---------------------------------------------------------------------------------------- */
void gml_oracle_objgrad_multi(int64_t K, int64_t n, int order, const double *counts,
                              const int8_t *spins, int64_t u, const double *theta, double *f,
                              double *g) {
    int64_t P = gml_oracle_multi_nparams(n, order);
    int32_t *keys = malloc(sizeof(int32_t) * P * order);
    gml_oracle_multi_keys(n, order, u, keys);
    node_t nd = {K, n, P, counts, spins, u, order, keys, sum_counts(counts, K)};
    double *row = malloc(sizeof(double) * P);
    node_eval(&nd, GML_RISE, theta, f, g, NULL, row);
    free(row);
    free(keys);
}

double gml_oracle_learn_multi(int64_t K, int64_t n, int order, const double *counts,
                              const int8_t *spins, double c, double tol, double *out,
                              double *kkt) {
    const double M = sum_counts(counts, K);
    const double lam = gml_oracle_lambda(c, n, M); /* still n^2, independent of order (:86) */
    const int64_t P = gml_oracle_multi_nparams(n, order);
    double worst = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t u = 0; u < n; ++u) {
        int32_t *keys = malloc(sizeof(int32_t) * P * order);
        gml_oracle_multi_keys(n, order, u, keys);
        node_t nd = {K, n, P, counts, spins, u, order, keys, M};
        uint8_t *pen = malloc((size_t)P);
        for (int64_t j = 0; j < P; ++j) pen[j] = (order > 1 && keys[j * order + 1] >= 0); /* length(inter)>1 (:118) */
        double *x = calloc((size_t)P, sizeof(double));
        int it = 0;
        double r = node_solve(&nd, GML_RISE, lam, pen, x, tol, 500, &it);
        memcpy(out + u * P, x, sizeof(double) * P);
        if (kkt) kkt[u] = r;
#pragma omp critical
        if (r > worst) worst = r;
        free(pen);
        free(x);
        free(keys);
    }
    return worst;
}
