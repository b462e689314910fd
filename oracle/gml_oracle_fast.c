/*
 * gml_oracle_fast.c -- cache-blocked, OpenMP CPU restatement of the learn() hot path, for sizes the
 * plain restatement (gml_oracle.c) cannot reach in seconds.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as gml_oracle.c): used by tests/ as the full-size checker and
 * by bench.py's `cpu_baseline` leg as the timed CPU path.  Nothing in the product may link or call it.
 *
 * Same math as gml_oracle.c, FP64 throughout, formulated the way a CPU wants it:
 *   - objective + gradient for MANY nodes at once (blocks of 32 nodes share one sweep over the +-1
 *     configurations; the energies and the gradient are two small GEMMs per block of samples), any
 *     of RISE (:169-172, :191-208), logRISE (:278-281), RPLE (:316-319);
 *   - the order-3 multi-body objective/gradient (:94-119) without materialising the statistics;
 *   - learn() for the pairwise formulations as a batched working-set orthant-wise Newton method:
 *     the algorithm of the device solver (graphicalmodellearning.jl_amd/csrc/gml_host.cpp) re-stated
 *     for the host, so that "CPU learn() wall-clock" is timed at the same tolerance with the same
 *     method.  It is pinned like everything else here: tests/test_oracle_golden.py checks it against
 *     the reference's golden vectors and against gml_oracle.c's dense Newton.
 * Reference line numbers: /root/reference/src/GraphicalModelLearning.jl.
 *
 * Plain C99 + OpenMP, no dependencies.  Built into libgml_oracle.so (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GML_RISE 0
#define GML_LOGRISE 1
#define GML_RPLE 2

#define NB 32 /* nodes per block: 8 AVX2 vectors of energies per sample */
#define SB 2  /* samples per inner block: 16 accumulator vectors */

double gml_oracle_lambda(double c, int64_t n, double M);

/* number of OpenMP threads of the calls below (0 = the runtime's default); the caller sizes it to the CPU time the
 * process may actually use (a container's cgroup quota can be far below the visible core count) */
void gml_oracle_set_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

static int nthreads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static double sum_counts(const double *c, int64_t K) { /* data_info (:76-81) */
    double M = 0;
    for (int64_t k = 0; k < K; ++k) M += c ? c[k] : 1.0;
    return M;
}

static inline double softplus_m2(double E) { /* log(1+exp(-2E)), stable */
    double t = -2.0 * E;
    return t > 0 ? t + log1p(exp(-t)) : log1p(exp(t));
}

/* One block of <= NB nodes over the configurations [k0, k1).
 *   ThT [n][NB]  theta transposed (column r = node r of the block, zero padded)
 *   u   [NB]     node ids (the slot that holds the field, :162), thu[r] = theta_r[u_r]
 * Accumulates  fsum[r] += sum_k phi,  Gt[i][r] += sum_k a_rk s_u s_i,  gu[r] += sum_k a_rk s_u
 * (a = -d phi / dE >= 0), and, if hw != NULL, stores the Hessian weight of (k, r) in hw[(k-k0)*NB + r]. */
static void block_eval(int form, int64_t k0, int64_t k1, int64_t n, const double *counts, double M,
                       const int8_t *spins, const int64_t *u, const double *ThT, const double *thu,
                       double *fsum, double *Gt, double *gu, double *sd /* scratch SB*n */) {
    for (int64_t k = k0; k < k1; k += SB) {
        const int sb = (int)(k1 - k < SB ? k1 - k : SB);
        double acc[SB][NB];
        for (int q = 0; q < SB; ++q) {
            for (int r = 0; r < NB; ++r) acc[q][r] = 0.0;
            const int8_t *s = spins + (k + (q < sb ? q : 0)) * n;
            double *d = sd + (int64_t)q * n;
            for (int64_t i = 0; i < n; ++i) d[i] = (double)s[i];
        }
        /* energies: acc[q][r] = sum_i s_i^k theta_r[i]   (inner sum of :170 / :196) */
        for (int64_t i = 0; i < n; ++i) {
            const double *t = ThT + i * NB;
            const double s0 = sd[i], s1 = sd[n + i];
            for (int r = 0; r < NB; ++r) {
                acc[0][r] += s0 * t[r];
                acc[1][r] += s1 * t[r];
            }
        }
        double v[SB][NB];
        for (int q = 0; q < SB; ++q) {
            const double w = q < sb ? (counts ? counts[k + q] : 1.0) / M : 0.0; /* samples[k,1]/num_samples (:170) */
            const int8_t *s = spins + (k + (q < sb ? q : 0)) * n;
            for (int r = 0; r < NB; ++r) {
                const double su = (double)s[u[r]];
                /* stat[k,i] = s_u s_i (i != u), stat[k,u] = s_u (:162)  =>  E = s_u (dot - theta_u s_u) + theta_u s_u */
                const double E = su * acc[q][r] - thu[r] + thu[r] * su;
                double a;
                if (form == GML_RPLE) {
                    const double sg = 1.0 / (1.0 + exp(2.0 * E));
                    fsum[r] += w * softplus_m2(E);
                    a = 2.0 * w * sg;
                } else {
                    a = w * exp(-E);
                    fsum[r] += a;
                }
                v[q][r] = a * su;
                gu[r] += a * su;
            }
        }
        /* gradient: Gt[i][r] += s_i^k * a_rk s_u^k   (:204-207) */
        for (int64_t i = 0; i < n; ++i) {
            double *G = Gt + i * NB;
            const double s0 = sd[i], s1 = sd[n + i];
            for (int r = 0; r < NB; ++r) G[r] += s0 * v[0][r] + s1 * v[1][r];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Public: smooth objective and gradient of `nn` pairwise nodes (any ids, repeats allowed) at
 * theta (nn x n row-major, slot u = field): the batched form of risea_obj / grad_risea_obj
 * (:191-208) and of the logRISE / RPLE smooth parts.  counts may be NULL (all ones).
 * logRISE is evaluated without a max-shift (|E| of a few hundred is fine in FP64).
 * ---------------------------------------------------------------------------------------- */
void gml_oracle_objgrad_nodes(int form, int64_t K, int64_t n, const double *counts, const int8_t *spins,
                              const int64_t *nodes, int64_t nn, const double *theta, double *f, double *g) {
    const double M = sum_counts(counts, K);
    const int64_t nblk = (nn + NB - 1) / NB;
    int64_t nch = (2 * (int64_t)nthreads() + nblk - 1) / nblk;
    if (nch > (K + 255) / 256) nch = (K + 255) / 256;
    if (nch < 1) nch = 1;
    const int64_t kch = ((K + nch - 1) / nch + SB - 1) / SB * SB;
    const int64_t ntask = nblk * nch;
    double *ThT = calloc((size_t)(nblk * n * NB), sizeof(double));
    double *Gt = calloc((size_t)(ntask * n * NB), sizeof(double));
    double *fs = calloc((size_t)(ntask * NB), sizeof(double)), *gus = calloc((size_t)(ntask * NB), sizeof(double));
    double *thu = calloc((size_t)(nblk * NB), sizeof(double));
    int64_t *uu = calloc((size_t)(nblk * NB), sizeof(int64_t));
    for (int64_t a = 0; a < nn; ++a) {
        const int64_t b = a / NB, r = a % NB;
        uu[a] = nodes[a];
        thu[a] = theta[a * n + nodes[a]];
        for (int64_t i = 0; i < n; ++i) ThT[(b * n + i) * NB + r] = theta[a * n + i];
    }
#pragma omp parallel
    {
        double *sd = malloc(sizeof(double) * SB * (size_t)n);
#pragma omp for schedule(dynamic, 1)
        for (int64_t t = 0; t < ntask; ++t) {
            const int64_t b = t / nch, c = t % nch;
            const int64_t k0 = c * kch, k1 = k0 + kch < K ? k0 + kch : K;
            if (k0 < k1)
                block_eval(form, k0, k1, n, counts, M, spins, uu + b * NB, ThT + b * n * NB, thu + b * NB, fs + t * NB,
                           Gt + t * n * NB, gus + t * NB, sd);
        }
        free(sd);
    }
    /* reduce the per-chunk partial sums block by block (contiguous sweeps), then scatter to the caller's layout */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < nblk; ++b) {
        double *G0 = Gt + (b * nch) * n * NB, *f0 = fs + (b * nch) * NB, *g0 = gus + (b * nch) * NB;
        for (int64_t c = 1; c < nch; ++c) {
            const double *Gc = Gt + (b * nch + c) * n * NB;
            for (int64_t e = 0; e < n * NB; ++e) G0[e] += Gc[e];
            for (int r = 0; r < NB; ++r) {
                f0[r] += fs[(b * nch + c) * NB + r];
                g0[r] += gus[(b * nch + c) * NB + r];
            }
        }
        for (int r = 0; r < NB && b * NB + r < nn; ++r) {
            const int64_t a = b * NB + r;
            double fa = f0[r];
            if (g) {
                double *ga = g + a * n;
                for (int64_t i = 0; i < n; ++i) ga[i] = -G0[i * NB + r];
                ga[nodes[a]] = -g0[r];
                if (form == GML_LOGRISE) /* g = grad Z / Z (:279) */
                    for (int64_t i = 0; i < n; ++i) ga[i] /= fa;
            }
            f[a] = form == GML_LOGRISE ? log(fa) : fa;
        }
    }
    free(ThT); free(Gt); free(fs); free(gus); free(thu); free(uu);
}

/* ------------------------------------------------------------------------------------------
 * Public: order-3 multi-body RISE objective and gradient (multiRISE(., ., 3), :94-119) of a few
 * nodes.  theta / g: nn x P with P = 1 + (n-1) + C(n-1,2) in the reference's key order
 * ((u), (u,i) ascending, (u,i,j) lexicographic; models.jl:228-246).  With t = the other spins in
 * ascending order, E = s_u * (th0 + sum_a t_a (th1[a] + sum_{b>a} th2[a,b] t_b)).
 * OpenMP over the configurations, one private gradient per thread.
 * ---------------------------------------------------------------------------------------- */
void gml_oracle_objgrad_multi3_nodes(int64_t K, int64_t n, const double *counts, const int8_t *spins,
                                     const int64_t *nodes, int64_t nn, const double *theta, double *f, double *g) {
    const double M = sum_counts(counts, K);
    const int64_t m = n - 1, P = 1 + m + m * (m - 1) / 2;
    const int T = nthreads();
    for (int64_t a = 0; a < nn; ++a) {
        const int64_t u = nodes[a];
        const double *th = theta + a * P, *th1 = th + 1, *th2 = th + 1 + m;
        double *gacc = calloc((size_t)T * (size_t)P, sizeof(double)), *facc = calloc((size_t)T, sizeof(double));
#pragma omp parallel
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num();
#else
            const int tid = 0;
#endif
            double *gt = gacc + (size_t)tid * (size_t)P, *g1 = gt + 1, *g2 = gt + 1 + m;
            double *t = malloc(sizeof(double) * 4 * (size_t)m);
            double fl = 0.0;
#pragma omp for schedule(static)
            for (int64_t k4 = 0; k4 < (K + 3) / 4; ++k4) {
                const int64_t k = 4 * k4;
                const int sb = (int)(K - k < 4 ? K - k : 4);
                double su[4] = {0, 0, 0, 0}, D[4], cq[4];
                for (int q = 0; q < 4; ++q) {
                    const int8_t *s = spins + (k + (q < sb ? q : 0)) * n;
                    double *tq = t + (int64_t)q * m;
                    int64_t j = 0;
                    for (int64_t i = 0; i < n; ++i)
                        if (i != u) tq[j++] = (double)s[i];
                    su[q] = (double)s[u];
                    D[q] = th[0];
                }
                const double *row = th2;
                for (int64_t aa = 0; aa < m; ++aa) {
                    const int64_t len = m - aa - 1;
                    double in0 = th1[aa], in1 = th1[aa], in2 = th1[aa], in3 = th1[aa];
                    const double *t0 = t + aa + 1, *t1 = t0 + m, *t2 = t1 + m, *t3 = t2 + m;
                    for (int64_t b = 0; b < len; ++b) {
                        const double c = row[b];
                        in0 += c * t0[b];
                        in1 += c * t1[b];
                        in2 += c * t2[b];
                        in3 += c * t3[b];
                    }
                    D[0] += t[aa] * in0;
                    D[1] += t[m + aa] * in1;
                    D[2] += t[2 * m + aa] * in2;
                    D[3] += t[3 * m + aa] * in3;
                    row += len;
                }
                for (int q = 0; q < 4; ++q) {
                    const double w = q < sb ? (counts ? counts[k + q] : 1.0) / M : 0.0;
                    const double e = w * exp(-su[q] * D[q]);
                    fl += e;
                    cq[q] = -e * su[q]; /* d/d theta_key = -w exp(-E) * stat_key, stat_key = s_u * prod t */
                    gt[0] += cq[q];
                }
                double *grow = g2;
                for (int64_t aa = 0; aa < m; ++aa) {
                    const int64_t len = m - aa - 1;
                    const double c0 = cq[0] * t[aa], c1 = cq[1] * t[m + aa], c2 = cq[2] * t[2 * m + aa], c3 = cq[3] * t[3 * m + aa];
                    g1[aa] += c0 + c1 + c2 + c3;
                    const double *t0 = t + aa + 1, *t1 = t0 + m, *t2 = t1 + m, *t3 = t2 + m;
                    for (int64_t b = 0; b < len; ++b) grow[b] += c0 * t0[b] + c1 * t1[b] + c2 * t2[b] + c3 * t3[b];
                    grow += len;
                }
            }
            facc[tid] = fl;
            free(t);
        }
        double fa = 0.0;
        for (int tt = 0; tt < T; ++tt) fa += facc[tt];
        f[a] = fa;
        if (g) {
#pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < P; ++j) {
                double s = 0.0;
                for (int tt = 0; tt < T; ++tt) s += gacc[(size_t)tt * (size_t)P + j];
                g[a * P + j] = s;
            }
        }
        free(gacc);
        free(facc);
    }
}

/* ------------------------------------------------------------------------------------------
 * learn() on the CPU: batched working-set orthant-wise Newton (the device solver's method).
 * ---------------------------------------------------------------------------------------- */
static inline double pseudo_grad(double x, double g, double lam) {
    if (lam == 0.0) return g;
    if (x > 0) return g + lam;
    if (x < 0) return g - lam;
    if (g + lam < 0) return g + lam;
    if (g - lam > 0) return g - lam;
    return 0.0;
}

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

/* Hessian of one node on its working set W (m entries) at theta x:  H = sum_k h_k stat_W stat_W^T
 * (RISE h = w exp(-E); logRISE Hess Z / Z - g g^T; RPLE h = 4 w s (1-s)) over every `stride`-th
 * configuration, rescaled by the weight of the sub-sample (sub-sampled Newton, like the device solver:
 * the gradient stays exact, so only the convergence rate depends on it).  Energies are recomputed from
 * the (sparse) x, whose non-zeros all lie in W.  Serial: the caller parallelises over nodes. */
static void node_hessian(int form, int64_t K, int64_t n, const double *counts, double M, const int8_t *spins, int64_t u,
                         const double *x, const int *W, int m, const double *gW, int64_t stride, double *H, double *st) {
    double Z = 0.0, wsum = 0.0;
    memset(H, 0, sizeof(double) * (size_t)m * m);
    for (int64_t k = 0; k < K; k += stride) {
        const int8_t *s = spins + k * n;
        const double su = (double)s[u], w = (counts ? counts[k] : 1.0) / M;
        if (w == 0.0) continue;
        wsum += w;
        double E = 0.0;
        for (int a = 0; a < m; ++a) {
            st[a] = W[a] == u ? su : su * (double)s[W[a]];
            E += x[W[a]] * st[a];
        }
        double h;
        if (form == GML_RPLE) {
            const double sg = 1.0 / (1.0 + exp(2.0 * E));
            h = 4.0 * w * sg * (1.0 - sg);
        } else {
            h = w * exp(-E);
            Z += h;
        }
        for (int a = 0; a < m; ++a) {
            const double ha = h * st[a];
            double *Hr = H + (size_t)a * m;
            for (int b = 0; b <= a; ++b) Hr[b] += ha * st[b];
        }
    }
    for (int a = 0; a < m; ++a)
        for (int b = 0; b <= a; ++b) {
            double v = H[(size_t)a * m + b];
            if (form == GML_LOGRISE) v = v / Z - gW[a] * gW[b];
            else v /= wsum;
            H[(size_t)a * m + b] = H[(size_t)b * m + a] = v;
        }
}

/* ------------------------------------------------------------------------------------------
 * Public: learn(samples, RISE/logRISE/RPLE(c, .)) for the nodes [node0, node1), un-symmetrised rows
 * (out: (node1-node0) x n, row u = reconstruction[u, :] of :181).  All nodes advance in lock-step so that
 * the objective/gradient passes are the blocked batched ones above; per iteration and node: KKT
 * residual, working set = non-zeros + at most `max_add` largest violators, exact Hessian on the working
 * set, Cholesky, projected backtracking (Armijo, then monotone-KKT acceptance at the FP64 noise floor).
 * stats[0] = Newton iterations, [1] = batched passes, [2] = node evaluations.  Returns the worst KKT.
 * ---------------------------------------------------------------------------------------- */
static double learn_ids(int form, int64_t K, int64_t n, const double *counts, const int8_t *spins, const int64_t *ids_in,
                        int64_t R, double c, double tol, int max_iter, double *out, double *kkt_out, double *stats) {
    const double M = sum_counts(counts, K);
    const double lam = gml_oracle_lambda(c, n, M);
    const int max_add = 64;
    double *X = calloc((size_t)(R * n), sizeof(double)), *G = calloc((size_t)(R * n), sizeof(double));
    double *Xt = calloc((size_t)(R * n), sizeof(double)), *Gtr = calloc((size_t)(R * n), sizeof(double));
    double *f = calloc((size_t)R, sizeof(double)), *ft = calloc((size_t)R, sizeof(double)), *kkt = calloc((size_t)R, sizeof(double));
    double *best = calloc((size_t)R, sizeof(double)), *Xbest = calloc((size_t)(R * n), sizeof(double));
    int64_t *ids = malloc(sizeof(int64_t) * (size_t)R), *act = malloc(sizeof(int64_t) * (size_t)R);
    uint8_t *done = calloc((size_t)R, 1);
    int *stall = calloc((size_t)R, sizeof(int));
    double *pack = malloc(sizeof(double) * (size_t)(R * n)), *fpack = malloc(sizeof(double) * (size_t)R);
    double *gpack = malloc(sizeof(double) * (size_t)(R * n));
    double npass = 0, nevals = 0;
    for (int64_t r = 0; r < R; ++r) {
        ids[r] = ids_in[r];
        best[r] = INFINITY;
        kkt[r] = INFINITY;
    }
    /* one batched pass over the rows listed in act[0..na): theta from `src`, results into fdst / gdst */
#define PASS(src, fdst, gdst, na)                                                                              \
    do {                                                                                                       \
        for (int64_t a_ = 0; a_ < (na); ++a_) memcpy(pack + a_ * n, (src) + act[a_] * n, sizeof(double) * n);  \
        int64_t *nid_ = malloc(sizeof(int64_t) * (size_t)(na));                                                \
        for (int64_t a_ = 0; a_ < (na); ++a_) nid_[a_] = ids[act[a_]];                                         \
        gml_oracle_objgrad_nodes(form, K, n, counts, spins, nid_, (na), pack, fpack, gpack);                   \
        for (int64_t a_ = 0; a_ < (na); ++a_) {                                                                \
            (fdst)[act[a_]] = fpack[a_];                                                                       \
            memcpy((gdst) + act[a_] * n, gpack + a_ * n, sizeof(double) * n);                                  \
        }                                                                                                      \
        free(nid_);                                                                                            \
        npass += 1;                                                                                            \
        nevals += (double)(na);                                                                                \
    } while (0)
    int64_t na = 0;
    for (int64_t r = 0; r < R; ++r) act[na++] = r;
    PASS(X, f, G, na);
    int it;
    /* per-row step data kept for the line search */
    int *Wm = calloc((size_t)R, sizeof(int)), *Wall = malloc(sizeof(int) * (size_t)(R * n));
    double *Dall = malloc(sizeof(double) * (size_t)(R * n)), *PGall = malloc(sizeof(double) * (size_t)(R * n));
    double *Fobj = calloc((size_t)R, sizeof(double)), *dd = calloc((size_t)R, sizeof(double)), *alpha = calloc((size_t)R, sizeof(double));
    uint8_t *need = calloc((size_t)R, 1);
    int64_t nactive = R;
    for (it = 0; it < max_iter; ++it) {
        /* Hessians over a strided sub-sample of 32768 x (rows / active rows) configurations (all of them for the
         * last few rows): the device solver's budget rule */
        int64_t Kh = 32768 * (R / (nactive > 0 ? nactive : 1));
        const int64_t stride = Kh >= K ? 1 : K / Kh;
        nactive = 0;
#pragma omp parallel reduction(+ : nactive)
        {
            int *W = malloc(sizeof(int) * (size_t)n), *vidx = malloc(sizeof(int) * (size_t)n);
            double *pgW = malloc(sizeof(double) * n), *gW = malloc(sizeof(double) * n), *d = malloc(sizeof(double) * n);
            double *viol = malloc(sizeof(double) * n), *st = malloc(sizeof(double) * n);
            double *H = NULL, *A = NULL;
            size_t hcap = 0;
#pragma omp for schedule(dynamic, 1)
            for (int64_t r = 0; r < R; ++r) {
                if (done[r]) continue;
                const int64_t u = ids[r];
                const double *x = X + r * n, *g = G + r * n;
                double F = f[r], worst = 0.0, worstW = 0.0;
                int m = 0, nv = 0;
                for (int64_t j = 0; j < n; ++j) {
                    const double l = j == u ? 0.0 : lam; /* current_spin != j is penalised (:171) */
                    F += l * fabs(x[j]);
                    const double pg = pseudo_grad(x[j], g[j], l);
                    if (fabs(pg) > worst) worst = fabs(pg);
                    if (x[j] != 0.0 || j == u) {
                        W[m++] = (int)j;
                        if (fabs(pg) > worstW) worstW = fabs(pg);
                    } else if (pg != 0.0) {
                        viol[nv] = fabs(pg);
                        vidx[nv++] = (int)j;
                    }
                }
                Fobj[r] = F;
                kkt[r] = worst;
                if (worst < best[r]) {
                    best[r] = worst;
                    memcpy(Xbest + r * n, x, sizeof(double) * n);
                    stall[r] = 0;
                } else {
                    ++stall[r];
                }
                if (worst <= tol || stall[r] >= 4) {
                    done[r] = 1;
                    continue;
                }
                ++nactive;
                if (worstW > worst * 0.999999 && worstW > 0 && m > 1) nv = 0; /* support not yet converged: no new entries */
                for (int a = 0; a < nv && a < max_add; ++a) { /* the max_add largest violators (partial selection sort) */
                    int bi = a;
                    for (int b = a + 1; b < nv; ++b)
                        if (viol[b] > viol[bi]) bi = b;
                    double tv = viol[a];
                    viol[a] = viol[bi];
                    viol[bi] = tv;
                    int ti = vidx[a];
                    vidx[a] = vidx[bi];
                    vidx[bi] = ti;
                    W[m++] = vidx[a];
                }
                for (int a = 0; a < m; ++a) {
                    gW[a] = g[W[a]];
                    pgW[a] = pseudo_grad(x[W[a]], g[W[a]], W[a] == u ? 0.0 : lam);
                }
                if ((size_t)m * m > hcap) {
                    hcap = (size_t)m * m;
                    free(H);
                    free(A);
                    H = malloc(sizeof(double) * hcap);
                    A = malloc(sizeof(double) * hcap);
                }
                node_hessian(form, K, n, counts, M, spins, u, x, W, m, gW, stride, H, st);
                double ridge = 0.0;
                for (;;) {
                    memcpy(A, H, sizeof(double) * (size_t)m * m);
                    for (int a = 0; a < m; ++a) {
                        A[(size_t)a * m + a] += ridge;
                        d[a] = -pgW[a];
                    }
                    if (chol_solve(A, d, m) == 0) break;
                    ridge = ridge == 0.0 ? 1e-12 : ridge * 100.0;
                }
                Wm[r] = m;
                memcpy(Wall + r * n, W, sizeof(int) * (size_t)m);
                memcpy(Dall + r * n, d, sizeof(double) * (size_t)m);
                memcpy(PGall + r * n, pgW, sizeof(double) * (size_t)m);
            }
            free(W); free(vidx); free(pgW); free(gW); free(d); free(viol); free(st); free(H); free(A);
        }
        if (nactive == 0) break;
        for (int64_t r = 0; r < R; ++r) {
            need[r] = !done[r];
            alpha[r] = 1.0;
        }
        for (int ls = 0; ls < 40; ++ls) {
            na = 0;
            for (int64_t r = 0; r < R; ++r) {
                if (!need[r]) continue;
                const int64_t u = ids[r];
                const double *x = X + r * n;
                double *xt = Xt + r * n;
                memcpy(xt, x, sizeof(double) * n);
                double dsum = 0.0;
                for (int a = 0; a < Wm[r]; ++a) {
                    const int j = Wall[r * n + a];
                    double v = x[j] + alpha[r] * Dall[r * n + a];
                    if (j != u && lam > 0) {
                        const double pg = PGall[r * n + a];
                        const double xi = x[j] != 0.0 ? (x[j] > 0 ? 1.0 : -1.0) : (pg < 0 ? 1.0 : -1.0);
                        if (v * xi < 0) v = 0.0; /* crossed zero: clip to the orthant face */
                    }
                    xt[j] = v;
                    dsum += PGall[r * n + a] * (v - x[j]);
                }
                dd[r] = dsum;
                act[na++] = r;
            }
            if (na == 0) break;
            PASS(Xt, ft, Gtr, na);
            for (int64_t a = 0; a < na; ++a) {
                const int64_t r = act[a], u = ids[r];
                const double *xt = Xt + r * n;
                double Fn = ft[r];
                for (int64_t j = 0; j < n; ++j)
                    if (j != u) Fn += lam * fabs(xt[j]);
                int ok = isfinite(Fn) && Fn <= Fobj[r] + 1e-4 * dd[r] + 4e-16 * fmax(1.0, fabs(Fobj[r]));
                if (!ok && isfinite(Fn) && -dd[r] < 1e-13 * fmax(1.0, fabs(Fobj[r]))) {
                    /* at the FP64 noise floor of f: accept iff the KKT residual does not grow */
                    double worst = 0.0;
                    for (int64_t j = 0; j < n; ++j) {
                        const double pg = pseudo_grad(xt[j], Gtr[r * n + j], j == u ? 0.0 : lam);
                        if (fabs(pg) > worst) worst = fabs(pg);
                    }
                    ok = worst < kkt[r];
                }
                if (ok) {
                    memcpy(X + r * n, xt, sizeof(double) * n);
                    memcpy(G + r * n, Gtr + r * n, sizeof(double) * n);
                    f[r] = ft[r];
                    need[r] = 0;
                } else {
                    alpha[r] *= 0.5;
                    if (alpha[r] < 1e-9) {
                        need[r] = 0;
                        stall[r] += 2;
                    }
                }
            }
        }
    }
    double worst = 0.0;
    for (int64_t r = 0; r < R; ++r) {
        const double kk = best[r] < kkt[r] ? best[r] : kkt[r];
        memcpy(out + r * n, (best[r] <= kkt[r] ? Xbest : X) + r * n, sizeof(double) * n);
        if (kkt_out) kkt_out[r] = kk;
        if (kk > worst) worst = kk;
    }
    if (stats) {
        stats[0] = it;
        stats[1] = npass;
        stats[2] = nevals;
    }
#undef PASS
    free(X); free(G); free(Xt); free(Gtr); free(f); free(ft); free(kkt); free(best); free(Xbest); free(ids); free(act);
    free(done); free(stall); free(pack); free(fpack); free(gpack);
    free(Wm); free(Wall); free(Dall); free(PGall); free(Fobj); free(dd); free(alpha); free(need);
    return worst;
}

/* the node loop of learn (:161) over the range [node0, node1) ... */
double gml_oracle_learn_pair_fast(int form, int64_t K, int64_t n, const double *counts, const int8_t *spins, int64_t node0,
                                  int64_t node1, double c, double tol, int max_iter, double *out, double *kkt_out,
                                  double *stats) {
    const int64_t R = node1 - node0;
    int64_t *ids = malloc(sizeof(int64_t) * (size_t)(R > 0 ? R : 1));
    for (int64_t r = 0; r < R; ++r) ids[r] = node0 + r;
    const double w = learn_ids(form, K, n, counts, spins, ids, R, c, tol, max_iter, out, kkt_out, stats);
    free(ids);
    return w;
}

/* ... and over a list of nodes (row r of `out` = node nodes[r]): the full-size parity tests solve a sample of the nodes of a
 * problem whose every node the oracle cannot solve in seconds */
double gml_oracle_learn_nodes_fast(int form, int64_t K, int64_t n, const double *counts, const int8_t *spins, const int64_t *nodes,
                                   int64_t nnodes, double c, double tol, int max_iter, double *out, double *kkt_out,
                                   double *stats) {
    return learn_ids(form, K, n, counts, spins, nodes, nnodes, c, tol, max_iter, out, kkt_out, stats);
}
