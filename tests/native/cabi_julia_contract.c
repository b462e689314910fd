/*
 * C-level contract test of the Julia binding (graphicalmodellearning.jl_amd/julia/GraphicalModelLearningHIP.jl).
 * There is no Julia in the image, so this program calls the C ABI (include/gml.h) EXACTLY as the `ccall`s of that file do:
 *   - the samples as a column-major Int64 matrix K x (1+n) (what `sample()` returns, sampling.jl:52-54; `ld = K`),
 *   - gml_opts filled field by field in the order of `GmlOpts`, gml_stats read back as `GmlStats`,
 *   - `out` read as the Julia `Array{Float64}(undef, P, R)` (C row-major R x P == Julia column-major P x R),
 *   - the same for gml_multi_* with devices {0, 0} (HIP(devices = [0, 0])).
 * Modes:  layout                       print sizeof / offsetof of the two structs (no device needed)
 *         run <samples.csv> <learned.csv> <c> <symmetrize>   learn RISE(c, symmetrize) and compare (needs a GPU)
 *         terms <samples.csv> <c> <symmetrize> <order>       learn multiRISE(c, symmetrize, order) and print its terms (needs a GPU)
 */
#include "gml.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int read_csv(const char *path, double **out, int *rows, int *cols) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t cap = 1024, cnt = 0;
    double *v = malloc(cap * sizeof *v);
    char line[1 << 16];
    int r = 0, c0 = -1;
    while (fgets(line, sizeof line, f)) {
        int c = 0;
        for (char *tok = strtok(line, ",\n\r"); tok; tok = strtok(NULL, ",\n\r")) {
            if (cnt == cap) v = realloc(v, (cap *= 2) * sizeof *v);
            v[cnt++] = strtod(tok, NULL);
            ++c;
        }
        if (c == 0) continue;
        if (c0 < 0) c0 = c;
        if (c != c0) return -2;
        ++r;
    }
    fclose(f);
    *out = v;
    *rows = r;
    *cols = c0;
    return 0;
}

#define OFF(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))

static int layout(void) {
    printf("sizeof gml_opts %zu\n", sizeof(gml_opts));
    OFF(gml_opts, tol); OFF(gml_opts, max_iter); OFF(gml_opts, precision); OFF(gml_opts, max_working); OFF(gml_opts, max_add);
    OFF(gml_opts, verbose); OFF(gml_opts, hess_samples); OFF(gml_opts, polish); OFF(gml_opts, max_cg);
    OFF(gml_opts, limbs_fwd); OFF(gml_opts, hv_limbs_fwd); OFF(gml_opts, hv_limbs_bwd); OFF(gml_opts, debug_row);
    OFF(gml_opts, hv_subsample); OFF(gml_opts, coarse);
    OFF(gml_opts, cg_viol_frac); OFF(gml_opts, cg_eta);
    printf("sizeof gml_stats %zu\n", sizeof(gml_stats));
    OFF(gml_stats, iterations); OFF(gml_stats, passes); OFF(gml_stats, forward_passes); OFF(gml_stats, hessian_passes);
    OFF(gml_stats, node_evals); OFF(gml_stats, max_kkt); OFF(gml_stats, lambda); OFF(gml_stats, t_pack); OFF(gml_stats, t_pass);
    OFF(gml_stats, t_hess); OFF(gml_stats, t_host); OFF(gml_stats, t_total); OFF(gml_stats, not_converged); OFF(gml_stats, polished);
    OFF(gml_stats, hv_evals); OFF(gml_stats, t_assemble);
    /* what the .jl file checks in __init__ (no device needed) */
    printf("const GML_ABI_VERSION %d\n", GML_ABI_VERSION);
    printf("call gml_abi_version %d\ncall gml_sizeof_opts %lld\ncall gml_sizeof_stats %lld\n", gml_abi_version(), (long long)gml_sizeof_opts(),
           (long long)gml_sizeof_stats());
    /* the constants of the .jl file */
    printf("const GML_OK %d\nconst GML_ENOTCONV %d\nconst GML_RISE %d\nconst GML_LOGRISE %d\nconst GML_RPLE %d\n", GML_OK, GML_ENOTCONV, GML_RISE,
           GML_LOGRISE, GML_RPLE);
    printf("const GML_I64 %d\nconst GML_F64 %d\nconst GML_PREC_F64 %d\nconst GML_PREC_I8X %d\nconst GML_PREC_AUTO %d\nconst GML_PREC_I8W %d\n", GML_I64,
           GML_F64, GML_PREC_F64, GML_PREC_I8X, GML_PREC_AUTO, GML_PREC_I8W);
    return 0;
}

/* reconstruction (R x P, from the Julia-shaped P x R buffer) -> optional 0.5 (R + R'), max |. - want| */
static double compare(const double *out, int n, int symmetrize, const double *want) {
    double worst = 0;
    for (int u = 0; u < n; ++u)
        for (int j = 0; j < n; ++j) {
            /* Julia: out[j+1, u+1] of the P x R array == C out[u*P + j]; permutedims(out)[u+1, j+1] */
            double v = out[(size_t)u * n + j];
            if (symmetrize) v = 0.5 * (v + out[(size_t)j * n + u]); /* GraphicalModelLearning.jl:184-186 */
            const double d = fabs(v - want[(size_t)u * n + j]);
            if (d > worst) worst = d;
        }
    return worst;
}

static int run(const char *samples_csv, const char *learned_csv, double c, int symmetrize) {
    double *S, *W;
    int K, cols, wr, wc;
    if (read_csv(samples_csv, &S, &K, &cols) || read_csv(learned_csv, &W, &wr, &wc)) {
        fprintf(stderr, "cannot read the fixtures\n");
        return 2;
    }
    const int n = cols - 1;
    if (wr != n || wc != n) return 2;
    /* Matrix{Int64}, column-major: element (k, j) at s[k + j*K] */
    int64_t *s = malloc(sizeof(int64_t) * (size_t)K * cols);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < cols; ++j) s[k + (size_t)j * K] = (int64_t)llround(S[(size_t)k * cols + j]);

    /* ---- solve_rows(): gml_problem_create / gml_problem_info / gml_learn / gml_problem_destroy ------------------ */
    gml_problem *h = NULL;
    int rc = gml_problem_create(s, GML_I64, K, n, K, 1 /* column-major */, 2, 0, n, 0 /* method.device */, &h);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_problem_create: %s\n", gml_last_error());
        return 3;
    }
    int64_t P = 0;
    gml_problem_info(h, NULL, NULL, NULL, &P, NULL, NULL);
    if (P != n) return 4;
    const int R = n;
    double *out = malloc(sizeof(double) * (size_t)P * R); /* Array{Float64}(undef, P, R) */
    /* GmlOpts(m.tol, m.max_iter, precision_id, m.max_working, m.max_add, m.verbose, m.hess_samples, m.polish ? 0 : -1, 0) */
    gml_opts o;
    o.tol = 1e-10;
    o.max_iter = 100;
    o.precision = GML_PREC_AUTO;
    o.max_working = 512;
    o.max_add = 64;
    o.verbose = 0;
    o.hess_samples = 0;
    o.polish = 0;
    o.max_cg = 0;
    o.limbs_fwd = o.hv_limbs_fwd = o.hv_limbs_bwd = o.debug_row = o.hv_subsample = o.coarse = 0;
    o.cg_viol_frac = o.cg_eta = 0.0;
    gml_stats st;
    memset(&st, 0xEE, sizeof st); /* Ref{GmlStats}() is uninitialised memory */
    /* learn(samples, ::RISE, ::HIP): symmetrization -> gml_learn_matrix (solve + 0.5 (R + R') on the device, :184-186), else gml_learn */
    if (symmetrize) rc = gml_learn_matrix(h, GML_RISE, c, 1, &o, out, NULL /* C_NULL */, &st);
    else rc = gml_learn(h, GML_RISE, c, &o, out, NULL /* C_NULL */, &st);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_learn: %s\n", gml_last_error());
        return 5;
    }
    gml_problem_destroy(h);
    const double e1 = compare(out, n, 0 /* already symmetrised by the library */, W);
    printf("single max_abs_diff %.3e iterations %d passes %d max_kkt %.3e lambda %.6e not_converged %d\n", e1, st.iterations, st.passes,
           st.max_kkt, st.lambda, st.not_converged);
    const double lam = c * sqrt(log((double)n * n / 0.05) / 1e6); /* the fixtures hold 1e6 samples (:157) */
    (void)lam;
    if (st.not_converged != 0 || !(st.max_kkt <= 1e-10) || st.iterations <= 0 || st.iterations > 100) return 6;

    /* ---- solve_rows_multi(): gml_multi_create / gml_multi_info / gml_multi_learn / gml_multi_destroy ------------- */
    int devs[2] = {0, 0};
    gml_multi *m = NULL;
    rc = gml_multi_create(s, GML_I64, K, n, K, 1, 2, devs, 2, &m);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_multi_create: %s\n", gml_last_error());
        return 7;
    }
    int64_t P2 = 0;
    gml_multi_info(m, NULL, NULL, NULL, &P2, NULL, NULL);
    if (P2 != n) return 8;
    double *out2 = malloc(sizeof(double) * (size_t)P2 * n); /* Array{Float64}(undef, P, n) */
    memset(&st, 0xEE, sizeof st);
    rc = gml_multi_learn(m, GML_RISE, c, &o, out2, NULL, &st, NULL);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_multi_learn: %s\n", gml_last_error());
        return 9;
    }
    gml_multi_destroy(m);
    if (symmetrize) { /* the gathered rows through gml_matrix_symmetrize, in place (as the .jl does on `rt`) */
        rc = gml_matrix_symmetrize(out2, n, n, devs[0], out2);
        if (rc != GML_OK) {
            fprintf(stderr, "gml_matrix_symmetrize: %s\n", gml_last_error());
            return 12;
        }
    }
    const double e2 = compare(out2, n, 0, W);
    printf("multi max_abs_diff %.3e iterations %d not_converged %d\n", e2, st.iterations, st.not_converged);
    if (st.not_converged != 0) return 10;
    return (e1 <= 5e-8 && e2 <= 5e-8) ? 0 : 11;
}

/* learn(samples, ::multiRISE, ::HIP) as the .jl file calls it: gml_terms_count -> gml_problem_create(order) -> gml_learn_terms ->
 * gml_terms_keys; prints every term as "term k1 k2 ... weight" (1-based keys, like the Dict the .jl builds) */
static int terms(const char *samples_csv, double c, int symmetrize, int order) {
    double *S;
    int K, cols;
    if (read_csv(samples_csv, &S, &K, &cols)) return 2;
    const int n = cols - 1;
    int64_t *s = malloc(sizeof(int64_t) * (size_t)K * cols);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < cols; ++j) s[k + (size_t)j * K] = (int64_t)llround(S[(size_t)k * cols + j]);
    const int64_t nterms = gml_terms_count(n, order, symmetrize);
    if (nterms < 0) return 3;
    double *weights = malloc(sizeof(double) * (size_t)nterms); /* Vector{Float64}(undef, nterms) */
    gml_problem *h = NULL;
    int rc = gml_problem_create(s, GML_I64, K, n, K, 1, order, 0, n, 0, &h);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_problem_create: %s\n", gml_last_error());
        return 4;
    }
    gml_opts o;
    memset(&o, 0, sizeof o);
    o.tol = 1e-10;
    o.max_iter = 100;
    o.precision = GML_PREC_AUTO;
    o.max_working = 512;
    o.max_add = 64;
    gml_stats st;
    memset(&st, 0xEE, sizeof st);
    rc = gml_learn_terms(h, GML_RISE, c, symmetrize, &o, weights, NULL, &st);
    gml_problem_destroy(h);
    if (rc != GML_OK) {
        fprintf(stderr, "gml_learn_terms: %s\n", gml_last_error());
        return 5;
    }
    int32_t *keys = malloc(sizeof(int32_t) * (size_t)order * (size_t)nterms); /* Array{Int32}(undef, order, nterms) */
    rc = gml_terms_keys(n, order, symmetrize, 0, nterms, keys);
    if (rc != GML_OK) return 6;
    printf("nterms %lld not_converged %d t_assemble_positive %d\n", (long long)nterms, st.not_converged, st.t_assemble > 0.0);
    for (int64_t t = 0; t < nterms; ++t) {
        printf("term");
        for (int j = 0; j < order; ++j)
            if (keys[t * order + j] >= 0) printf(" %d", keys[t * order + j] + 1);
        printf(" %.17g\n", weights[t]);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "layout")) return layout();
    if (argc >= 6 && !strcmp(argv[1], "terms")) return terms(argv[2], atof(argv[3]), atoi(argv[4]), atoi(argv[5]));
    if (argc >= 6 && !strcmp(argv[1], "run")) return run(argv[2], argv[3], atof(argv[4]), atoi(argv[5]));
    fprintf(stderr, "usage: %s layout | run samples.csv learned.csv c symmetrize\n", argv[0]);
    return 64;
}
