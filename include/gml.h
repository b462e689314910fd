/*
 * gml.h -- C ABI of the MI355X-native learn() hot path of GraphicalModelLearning.jl.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / C++ types.  Every
 * entry point names the reference interface it replaces (paths relative to
 * /root/reference/src/GraphicalModelLearning.jl).  The reference-side binding (Julia
 * `ccall`) is shown in INTEGRATION.md and graphicalmodellearning.jl_amd/julia/.
 *
 * All functions return 0 on success or a GML_E* code; gml_last_error() returns a
 * thread-local message for the last failure.  The library never falls back to a CPU
 * implementation: without a usable HIP device every compute entry point fails with
 * GML_EHIP.
 */
#ifndef GML_H
#define GML_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------- */
#define GML_OK 0
#define GML_EINVAL 1     /* bad argument: shape, dtype, non +-1 spin, negative count, ...   */
#define GML_ENOTCONV 2   /* a node did not reach the KKT tolerance (reference: the
                            `@assert termination_status == LOCALLY_SOLVED` at :127,180,251,289,327) */
#define GML_EHIP 3       /* HIP runtime error / no device                                   */
#define GML_ENOMEM 4     /* the problem does not fit the device                             */
#define GML_EUNSUPPORTED 5

/* ---- formulations: the GMLFormulation subtypes (:20-56) ----------------------------- */
#define GML_RISE 0       /* RISE    (:30-35), objective :169-172; RISEA (:37-42, :191-260) is the same math */
#define GML_LOGRISE 1    /* logRISE (:44-49), objective :278-281                            */
#define GML_RPLE 2       /* RPLE    (:51-56), objective :316-319                            */

/* ---- element types of the sample histogram ------------------------------------------ */
#define GML_I8 0
#define GML_I32 1
#define GML_I64 2        /* what `sample()` returns (sampling.jl:52-54)                     */
#define GML_F64 3        /* what `readdlm` returns (test/runtests.jl:71)                    */

/* ---- arithmetic of the device objective/gradient pass ------------------------------- */
#define GML_PREC_F64 0   /* FP64 MFMA (v_mfma_f64_16x16x4_f64)                             */
#define GML_PREC_I8X 1   /* fixed point on v_mfma_i32_*_i8: Theta in 38-bit and V in 31-bit int8 limbs (V rounded
                            with a dither), integer GEMMs without further rounding: f, grad to ~1e-9 relative      */
#define GML_PREC_I8W 3   /* FP64-grade fixed point on the same int8 matrix cores: Theta in 54-bit (7 limb planes; entries within a
                            factor two of a row's largest are exact), V in dithered 47-bit int8 limbs (6 planes), exp in FP64,
                            integer GEMMs without further rounding: f, grad agree with a Float64 evaluation (:191-208) to the
                            1e-12 the GML_PREC_F64 path is held to, at ~6x its speed.  gml_learn builds its Hessians and
                            Hessian-vector products from the top 31 bits of the same V planes                              */
#define GML_PREC_AUTO 2  /* gml_objgrad_batch (the pair an external solver registers in place of the reference's Float64 obj / grad):
                            GML_PREC_I8W.  gml_learn: GML_PREC_I8X, except for solves with tol < 2e-10 and for small problems
                            (samples x parameters x spins <= 2^28: a property of the problem, not of the call or of the node
                            shard -- the reference's own fixtures, the README example; every kernel is launch-bound there), which
                            take GML_PREC_I8W (as few iterations as Float64; the 31-bit weights would stall at their noise floor).  Never GML_EUNSUPPORTED for a
                            valid histogram: beyond 2^24 configurations the int8 path keeps one set of i32
                            gradient accumulators per 2^23 configurations and adds them in int64             */

typedef struct gml_problem gml_problem; /* opaque: packed spins + weights resident in HBM  */

typedef struct gml_opts {
    double tol;          /* KKT tolerance: max |pseudo-gradient| per node (default 1e-9)    */
    int32_t max_iter;    /* outer (Newton) iterations (default 100)                         */
    int32_t precision;   /* GML_PREC_* (default GML_PREC_AUTO)                               */
    int32_t max_working; /* cap on a node's Newton block (default = max = 512, multiple of 32); a
                            denser optimum is solved by cycling blocks (block Gauss-Seidel)    */
    int32_t max_add;     /* new (violating) coordinates admitted per node per iteration (default 64; twice that while a node has
                            more than 16 max_working violators: a dense optimum) */
    int32_t verbose;     /* 0 silent, 1 per-iteration line on stderr                        */
    int32_t hess_samples; /* Newton Hessians use the first hess_samples configurations (< 0 = all;
                            0 = adaptive: 32768 x (local nodes / nodes still active), so the
                            last few nodes get all of them); the gradient always uses all     */
    int32_t polish;      /* precision i8x only: 0 = rows that stall above tol at the noise floor of the int8-limb
                            arithmetic continue on the FP64 path when its workspaces fit (default); -1 = never */
    int32_t max_cg;      /* conjugate-gradient iterations per Newton step of the matrix-free rows (working sets above
                            max_working); each costs one Hessian-vector pass (default 16) */
    /* tuning of the fixed-point arithmetic and of the matrix-free Newton-CG; 0 = the default named */
    int32_t limbs_fwd;   /* int8 limb planes of Theta in the objective passes: 3, 4 or 5 (default 5: 38 significant bits)  */
    int32_t hv_limbs_fwd; /* ... of the CG direction in the Hessian-vector passes: 2..5 (default 2: 14 bits, ample for an
                            inexact Newton step that stops at a 5 % residual)                                              */
    int32_t hv_limbs_bwd; /* limb planes of the products h_k (x_k . p) in those passes: 2 or 4 (default 2)                  */
    int32_t debug_row;   /* local row traced on stderr when verbose >= 2 (default 0)                                       */
    int32_t hv_subsample; /* the Hessian-vector products of the matrix-free rows run over 1/hv_subsample of the configurations,
                            spread over the whole histogram (the gradient always uses all of them, so only the convergence
                            rate is affected).  0 = automatic: rows still admitting their support: as many as keep >= 32
                            configurations per working-set entry, at most 8; rows with their support final: every configuration
                            for the first two steps of a solve, 1/2 from step 2, 1/8 from step 4 on (inexact Krylov: the later a
                            step, the less accurate its product needs to be).  1 = every configuration, always.  n > 1: the
                            admitting rows over 1/n, the others over every configuration                                     */
    int32_t coarse;      /* int8-limb precisions, exp forms: 0 = while every active node is farther than max(1e-7, 100 tol) (KKT) from its
                            optimum the passes run in a cheap form -- theta in 30 bits (i8w: its top four limb planes, one forward sweep
                            instead of two; i8x: four limb planes instead of five), the weights in three limb planes (dithered 23 bits),
                            one 3-plane backward launch -- and at full width for the rest of the solve, starting with a re-evaluation of
                            the rows that ended the phase: a coarse gradient certifies nothing (default); -1 = every pass at full width;
                            e > 0: the coarse phase ends at 10^-e                                                                  */
    double cg_viol_frac; /* matrix-free rows admit, per iteration, the violators within this fraction of the largest
                            violation (default 0.5: full Newton steps throughout on dense optima)           */
    double cg_eta;       /* CG stops at a residual reduced by min(cg_eta, sqrt(kkt)) (default 0.05); rows that are still building
                            their support (violators > 1/16 of the support) stop at 0.25                                   */
} gml_opts;

typedef struct gml_stats {
    int32_t iterations;      /* outer iterations                                            */
    int32_t passes;          /* full objective+gradient passes (all local nodes)            */
    int32_t forward_passes;  /* objective-only passes (line-search trials)                  */
    int32_t hessian_passes;
    int64_t node_evals;      /* sum over the objective(/gradient) passes of the number of nodes evaluated (their time: t_pass) */
    double max_kkt;          /* worst final KKT residual over local nodes                   */
    double lambda;           /* the regulariser actually used (:157)                        */
    double t_pack, t_pass, t_hess, t_host, t_total; /* seconds; t_pack = building the handle (gml_problem_ingest_times
                                t[3]: host packing + uploads + operand images), not part of t_total */
    int32_t not_converged;   /* number of local nodes above tol                             */
    int32_t polished;        /* 1 if rows were finished on the FP64 path (precision i8x, see gml_opts.polish) */
    int64_t hv_evals;        /* node evaluations of the Hessian-vector passes of the matrix-free rows (their time: t_hess) */
    double t_assemble;       /* gml_learn_terms: the device-side assembly of the term array + its copy to the caller (part of t_total) */
} gml_stats;

/*
 * ABI identity.  gml_opts and gml_stats are plain structs mirrored field by field in the bindings (the Julia file, the ctypes
 * twin): a binding built against another revision of this header would read and write the wrong bytes without any error.  Every
 * binding therefore checks, when it loads the library, that gml_abi_version() is the GML_ABI_VERSION it was written against and that
 * gml_sizeof_opts() / gml_sizeof_stats() equal the sizes of its own mirrors -- and refuses to run otherwise.  GML_ABI_VERSION is
 * bumped by every change of a struct layout, of an argument list or of the meaning of a constant.
 */
#define GML_ABI_VERSION 6
int gml_abi_version(void);
int64_t gml_sizeof_opts(void);
int64_t gml_sizeof_stats(void);

const char *gml_last_error(void);
void gml_default_opts(gml_opts *o);

/*
 * gml_problem_create -- replaces what every `learn` method does first: `data_info(samples)`
 * (:76-81) and the per-node `nodal_stat` comprehension (:162, :218, :271, :309; multi-body
 * :94-108), which it never materialises.
 *
 *   samples   K x (1+n) histogram, column 0 = counts, columns 1..n = spins in {-1,+1}
 *             (the `Array{T,2}` produced by sampling.jl:52-54).  ld = leading dimension;
 *             col_major != 0 for a Julia matrix (element (k,j) at samples[k + j*ld]),
 *             0 for a C/numpy matrix (samples[k*ld + j]).
 *   order     interaction order of the statistics: 2 = pairwise (RISE/logRISE/RPLE),
 *             p >= 2 = multiRISE(..., p) (:22-28).  order 1 = fields only.
 *   node0,node1  this handle solves nodes [node0, node1) (0-based); node-wise sharding for
 *             one-process-per-GPU runs (the `for current_spin = 1:num_spins` loop, :161).
 *             What a node range changes and what it does not: objective, gradient and Hessian-vector values of a row are the
 *             same bits whatever range the handle covers (int8-limb precisions).  The SOLVE's trajectory is not: the Newton
 *             Hessians are built from a sub-sample whose size depends on how many rows share the GPU (2^23 / rows, 32 768 ..
 *             131 072 configurations; gml_opts.hess_samples pins it), and the coarse phase ends for all rows of a handle at
 *             once -- so a row's iteration count and the last bits of its solution depend on the sharding (and on the GPU count),
 *             while every sharding ends within tol of the same optimum (tests: 2e-9 between shardings at tol 1e-9).
 *   device    HIP device ordinal.
 * The caller keeps ownership of `samples`; it is not referenced after return.
 */
int gml_problem_create(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld,
                       int col_major, int order, int64_t node0, int64_t node1, int device,
                       gml_problem **out);

/*
 * Ingest.  gml_problem_create reads the caller's matrix exactly ONCE, on the host: a streaming packer (worker threads,
 * AVX2) turns 32 consecutive 8-byte elements of a spin column into one sign word, validates the +-1 alphabet and the
 * counts in the same sweep, and K n / 8 bytes of sign bits + 8 K bytes of weights cross PCIe instead of the
 * 8 K (n+1) of the Matrix{Int64} (`copy` of the Adjoint at :73; C3: 0.13 GB instead of 8.2 GB).  Packing overlaps the
 * copies (two pinned stages).  gml_problem_ingest_times reports the split:
 *   t[0] host packing (counts + sign words), t[1] allocations / uploads not hidden behind it, t[2] MFMA operand images
 *   (device), t[3] whole create call, t[4] of t[1]: stream + device allocations, t[5] of t[1]: weights; seconds.
 */
int gml_problem_ingest_times(const gml_problem *p, double t[6]);

/* The other route: the raw matrix is uploaded as it is and converted / validated on the device (64x the PCIe bytes;
 * for hosts with few cores).  Same arguments and the same resulting handle, bit for bit. */
int gml_problem_create_device_convert(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld,
                                      int col_major, int order, int64_t node0, int64_t node1, int device,
                                      gml_problem **out);

/*
 * Packed form of a histogram, for callers that build several handles from one matrix (other processes, other GPUs) or
 * keep the samples packed: sign_bits [n][words_per_spin] dwords, spin-major, bit j of word w <-> configuration 32 w + j,
 * set <=> the spin is -1 (bits beyond K ignored); counts [K] doubles (NULL = all ones).
 *   gml_packed_words(K)         words_per_spin of the device image (K rounded up to 1024, / 32)
 *   gml_pack_histogram          host only (no device needed): matrix -> sign_bits, counts, *M = sum of counts (:76-81)
 *   gml_problem_create_packed   handle from the packed form
 *   gml_problem_get_sign_bits   the packed form of a handle's samples ([n][gml_packed_words(K)] dwords, host pointer),
 *                               e.g. of a handle sampled on the device
 */
int64_t gml_packed_words(int64_t K);
int gml_pack_histogram(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major,
                       uint32_t *sign_bits, int64_t words_per_spin, double *counts, double *M);
int gml_problem_create_packed(const uint32_t *sign_bits, int64_t words_per_spin, const double *counts, int64_t K, int64_t n,
                              int order, int64_t node0, int64_t node1, int device, gml_problem **out);
int gml_problem_get_sign_bits(gml_problem *p, uint32_t *sign_bits);

/* Same, from split inputs: counts (K doubles, NULL = all ones) and spins (K x n int8,
 * row-major).  Used by the synthetic benchmark so that no 8-byte histogram is built. */
int gml_problem_create_spins(const double *counts, const int8_t *spins, int64_t K, int64_t n,
                             int order, int64_t node0, int64_t node1, int device,
                             gml_problem **out);

/*
 * gml_problem_create_sampled -- the step BEFORE the path: replaces `sample(gm, N)` for pairwise models
 * (src/sampling.jl:34-57, 94-106) and feeds the result straight into a handle, all on the device.
 * `model` is the n x n symmetric matrix of FactorGraph(matrix) (models.jl:105-134): off-diagonal =
 * couplings, diagonal = fields.  Sampling is exact (enumeration + CDF inversion, the reference's own
 * method) per connected component of the coupling graph, so every component must have <= 22 spins
 * (GML_EUNSUPPORTED otherwise).  The handle holds N rows with count 1 each (M = N).
 */
int gml_problem_create_sampled(const double *model, int64_t n, int64_t N, uint64_t seed, int order,
                               int64_t node0, int64_t node1, int device, gml_problem **out);

/* Same for a model of any interaction order given as a term list (the reference's general sampler,
 * src/sampling.jl:60-88, dispatch :94-106): term t couples the spins keys[t*key_stride .. +key_stride)
 * (0-based, -1 = unused slot) with weight weights[t]; P(s) ~ exp(sum_t w_t prod_{i in t} s_i).  Exact per
 * connected component of the term hypergraph (<= 22 spins each). */
int gml_problem_create_sampled_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms,
                                     int64_t n, int64_t N, uint64_t seed, int order, int64_t node0,
                                     int64_t node1, int device, gml_problem **out);

/* Beyond the reference (whose only sampler is exact enumeration): N independent Glauber (heat-bath) chains of
 * `sweeps` sequential sweeps from a random start, for models whose components exceed 22 spins (lattices, ...).
 * Same term-list arguments as above; the final states of the chains become the handle's samples. */
int gml_problem_create_mcmc_terms(const int32_t *keys, int key_stride, const double *weights, int64_t nterms,
                                  int64_t n, int64_t N, uint64_t seed, int sweeps, int order, int64_t node0,
                                  int64_t node1, int device, gml_problem **out);

/*
 * gml_problem_create_sampled_hist -- sample AND histogram on the device: what `sample(gm, N)` returns is the countmap of the
 * draws (sampling.jl:52-54: one row per distinct configuration, column 1 = its count).  Same term-list arguments as above
 * (mcmc_sweeps = 0: exact sampling, > 0: Glauber chains); n <= 64, N < 2^31.  The N draws become 64-bit keys, are radix-sorted
 * and run-length encoded on the device, and the handle holds the K' <= min(N, 2^n) DISTINCT configurations (ascending key
 * order: bit i of the key <=> spin i is -1) with their multiplicities as counts, M = N.  A 9-spin model sampled 1e8 times
 * gives a 512-row handle, and learn() on it costs 512 rows, not 1e8.
 */
int gml_problem_create_sampled_hist(const int32_t *keys, int key_stride, const double *weights, int64_t nterms, int64_t n,
                                    int64_t N, uint64_t seed, int mcmc_sweeps, int order, int64_t node0, int64_t node1,
                                    int device, gml_problem **out);

/* The counts of a handle's K rows (host pointer, K doubles): column 1 of the histogram (sampling.jl:54). */
int gml_problem_get_counts(gml_problem *p, double *counts);

/* The +-1 configurations held by a handle, K x n row-major (host pointer). */
int gml_problem_get_spins(gml_problem *p, int8_t *spins);

void gml_problem_destroy(gml_problem *p);

/* Device blocks of >= 1 MB that handles and solves release are kept by the library, per device and size, for the next handle of the
 * same shape (up to a quarter of the device's memory; an allocation that fails empties the cache and tries again): on MI355X /
 * ROCm 7 a hipMalloc of a multi-GB block after a hipFree sporadically takes 0.4-1.3 s, several times the headline solve.
 * gml_trim_cache() returns all of it to the driver; the return value is the number of bytes released. */
int64_t gml_trim_cache(void);
/* Cap of that cache in bytes per device: 0 = keep nothing (every release goes back to the driver: for processes that share a GPU
 * with other allocators -- a framework's caching allocator, other ranks), < 0 = the default, a quarter of the device's memory.
 * When a release does not fit, the blocks released longest ago are returned to the driver first. */
void gml_set_cache_limit(int64_t bytes_per_device);

/* sizes: n, K (rows given), M = sum(counts) (:79), P = parameters per node
 * (n for order 2; sum_{p<=order} C(n-1,p-1) in general), local node range */
int gml_problem_info(const gml_problem *p, int64_t *n, int64_t *K, double *M, int64_t *P,
                     int64_t *node0, int64_t *node1);

/* lambda = c*sqrt(log(n^2/0.05)/M)  (:157; identical at :86, :213, :266, :304) */
double gml_lambda(double c, int64_t n, double M);

/* Multi-body keys of node u in the reference's construction order (:94-104 with
 * models.jl:228-246): P rows of `order` int32 (0-based spin ids, -1 = unused slot). */
int gml_multi_keys(const gml_problem *p, int64_t u, int32_t *keys);

/*
 * gml_objgrad_batch -- the operator boundary.  Replaces the pair the reference registers
 * with JuMP, `obj(x...)` / `grad(g, x...)` = risea_obj / grad_risea_obj (:191-208, :221-233),
 * for many nodes at once, and the @NLobjective smooth parts of logRISE (:278-281) and
 * RPLE (:316-319).
 *
 *   nodes[r]        node id (0-based, any node, repeats allowed), r < nrows
 *   theta[r*ld + j] parameter j of row r in the REFERENCE's layout: pairwise -> j = spin
 *                   index, slot j == nodes[r] is the field (the u-th column of nodal_stat
 *                   is s_u, :162); multi-body -> j indexes gml_multi_keys(nodes[r]).
 *   f[r]            smooth objective (no l1 term)
 *   g[r*ld + j]     its gradient, same layout as theta
 * One "node evaluation" = one row.
 *
 * Sparse rows are cheaper: the forward GEMM of a 32-row node tile sweeps only the statistics columns on which one of its rows is
 * non-zero (the compact column list and the bit image of those columns are built on the device in front of the pass; lists longer than
 * a quarter of the columns sweep everything).  The sums are the same integers, so f and g are the same bits as a sweep over all columns; a
 * pass at rows with ~15 non-zeros of 1024 costs 7.7 ms instead of 10.7 (headline size, precision i8w).
 *
 * theta, f and g are host pointers, or -- all three -- DEVICE pointers on the handle's GPU (detected): rows that live in HBM (a
 * device-side optimiser, torch / CuPy / Julia GPU arrays) are scattered into the internal layout, evaluated and gathered back by
 * kernels, with no staging copy and nothing crossing PCIe but the control block and two small per-row arrays (headline pass
 * n = 1024, K = 1e6: 10.4 ms instead of 13.4 through host pointers; the same bits either way).  `nodes` is a host array in both
 * forms.  Ordering: the kernels run on the handle's own stream, a blocking stream -- ordered after everything the caller enqueued on
 * the device's null stream (torch's default stream) before the call; work on other streams must have completed.  The call returns
 * when f and g are written.
 */
int gml_objgrad_batch(gml_problem *p, int formulation, int precision, int64_t nrows,
                      const int64_t *nodes, const double *theta, int64_t ld, double *f,
                      double *g);

/*
 * gml_hessvec_batch -- beyond the reference (whose registered operator is first-order, :221-233, so Ipopt falls back
 * to a limited-memory Hessian there): the curvature operator  hv[r] = Hess f_{nodes[r]}(theta[r]) * vec[r]  for
 * second-order / Newton-CG external solvers, same layouts as gml_objgrad_batch.  Two GEMM passes on the int8 matrix
 * cores with the curvature weights of an objective pass at theta (precision i8x; ~1e-8 relative).  This is the
 * operator gml_learn's matrix-free Newton-CG uses for working sets above max_working.
 * theta, vec and hv: host pointers, or all three device pointers (as gml_objgrad_batch).
 */
int gml_hessvec_batch(gml_problem *p, int formulation, int64_t nrows, const int64_t *nodes, const double *theta,
                      const double *vec, int64_t ld, double *hv);
/* The same with the arithmetic named: GML_PREC_I8X (what gml_hessvec_batch runs; GML_PREC_AUTO means this), or GML_PREC_F64 --
 * both passes on the FP64 matrix cores (an objective pass at theta, then U_k = h_k (x_k . vec) formed in place of its weights and
 * contracted by the same backward GEMM): the curvature operator at the accuracy of the reference's Float64 arithmetic (1e-12
 * against a dense numpy Hessian, tests/test_gpu_operator_export.py), for external second-order solvers that want their Hessian
 * as good as their gradient; ~10x the time of the int8 form.  GML_PREC_I8W: GML_EUNSUPPORTED (the Hessian-vector forms read
 * 31-bit curvature weights whatever the objective pass wrote). */
int gml_hessvec_batch_prec(gml_problem *p, int formulation, int precision, int64_t nrows, const int64_t *nodes, const double *theta,
                           const double *vec, int64_t ld, double *hv);

/*
 * gml_learn -- replaces learn(samples, formulation, method) for the handle's node range
 * (:154-189 RISE, :263-298 logRISE, :301-336 RPLE, :210-260 RISEA, :83-133 multiRISE up to
 * the per-node solve).  For every local node it minimises
 *     f_u(x) + lambda * sum_{j penalised} |x_j|
 * (the problem the reference hands to Ipopt through the z >= |x| epigraph, :166-181).
 *
 *   out   (node1-node0) x P, row-major: row r = solution of node node0+r in the reference's
 *         layout (pairwise: reconstruction[u, 1:n] of :181, diagonal slot = field).
 *         May be a host pointer or a device pointer (detected).
 *   kkt   optional (node1-node0) host array: final max|pseudo-gradient| per node.
 * Symmetrisation (:184-186, :135-149) needs all rows: handles over all nodes take gml_learn_matrix / gml_learn_terms (solve and
 * result assembly in one call, on the device), node shards gather their rows and call gml_matrix_symmetrize / gml_terms_assemble.
 * Returns GML_ENOTCONV if some node stayed above opts->tol (out is still filled).
 */
int gml_learn(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts,
              double *out, double *kkt, gml_stats *stats);

/*
 * gml_learn_warm -- gml_learn from a starting point instead of x = 0: x0 = (node1-node0) x P rows in the layout of `out` (host or
 * device pointer; NULL = zeros = gml_learn).  The optimum does not depend on it (the problems are convex); the number of iterations
 * does.  For regularisation paths -- the same samples solved at a sequence of c, each from the previous solution (the reference
 * re-solves from scratch: JuMP start values are never set, :166-167) -- and for re-solves after more samples arrived.
 * out may be x0 itself.
 */
int gml_learn_warm(gml_problem *p, int formulation, double regularizer_c, const gml_opts *opts, const double *x0, double *out,
                   double *kkt, gml_stats *stats);

/*
 * Result assembly of multiRISE on the device -- replaces the tail of learn(samples, ::multiRISE, ...): the per-node
 * `reconstruction[inter] = ...` (:129-132), the symmetrisation (group by sorted key, `mean`: :135-149) and the Dict that
 * FactorGraph(order, n, :spin, reconstruction) (:151, models.jl:8-17) is built from.  The terms of the learned model are returned
 * as ONE array of weights in the order the reference itself lists a model's terms in (models.jl:61,72: `sort(..., by = x ->
 * (length(x), x))`), so no key table exists anywhere; a key <-> its position is closed-form (combinatorial number system):
 *   symmetrize != 0   every ascending key S with |S| <= order: by size, lexicographic within a size; C(n,1) + ... + C(n,order)
 *                     weights, each the mean over u in S of row u's entry for (u, S \ {u}), added in ascending u
 *   symmetrize == 0   every key (u, S'), S' an ascending subset of the other spins: by size, then u, then S'; n P weights
 *
 *   gml_terms_count     number of terms (-1: unsupported order / overflow)
 *   gml_terms_assemble  rows: the n x P solved rows (row u in the layout of gml_learn's `out`, leading dimension ld), host OR device
 *                       pointer; out: gml_terms_count doubles, host OR device pointer.  One kernel launch on `device` (on the
 *                       device that holds rows / out when one of them is a device pointer); host rows are staged through it
 *   gml_terms_keys      host only: the keys of the terms [first, first + count), `order` int32 per term, 0-based spins, -1 = unused
 *   gml_terms_rank      host only: position of a key of `len` spins (0-based), -1 if the model has no such key
 *   gml_learn_terms     gml_learn over ALL nodes of the handle + the assembly, the rows never leaving the device: `terms` (host or
 *                       device pointer) receives the term array; kkt, stats as gml_learn (stats->t_assemble = the assembly).
 *                       GML_EINVAL for a handle over a node sub-range (gather the rows, then gml_terms_assemble).
 *                       C5 (n = 512, order 3): 67.0 M row entries -> 22.5 M terms (179 MB) in one launch.
 */
int64_t gml_terms_count(int64_t n, int order, int symmetrize);
int gml_terms_assemble(const double *rows, int64_t ld, int64_t n, int order, int symmetrize, int device, double *out);
int gml_terms_keys(int64_t n, int order, int symmetrize, int64_t first, int64_t count, int32_t *keys);
int64_t gml_terms_rank(int64_t n, int order, int symmetrize, const int32_t *key, int len);
int gml_learn_terms(gml_problem *p, int formulation, double regularizer_c, int symmetrize, const gml_opts *opts, double *terms,
                    double *kkt, gml_stats *stats);

/*
 * The pairwise counterpart: `reconstruction = 0.5 * (reconstruction + transpose(reconstruction))` (:184-186) on the device.
 * On the host that one line walks the transposed operand with a stride of n doubles: 16 ms at n = 1024 (a sixth of the headline
 * solve), 0.22 s at n = 4096 (as much as a whole config-4 solve on eight GPUs).  (a + b) * 0.5 in FP64: the same bits.
 *   gml_learn_matrix        gml_learn over ALL nodes of a pairwise handle + the symmetrisation (symmetrize != 0), the rows never
 *                           leaving the device: out = the n x n matrix the reference's learn returns (host or device pointer);
 *                           stats->t_assemble = the symmetrisation + the copy.  symmetrize == 0: plain gml_learn.
 *   gml_matrix_symmetrize   the same for rows gathered from node shards: rows n x n (leading dimension ld), out n x n, host or
 *                           device pointers, in place allowed
 */
int gml_learn_matrix(gml_problem *p, int formulation, double regularizer_c, int symmetrize, const gml_opts *opts, double *out,
                     double *kkt, gml_stats *stats);
int gml_matrix_symmetrize(const double *rows, int64_t ld, int64_t n, int device, double *out);

/*
 * gml_multi_* -- the node loop of `learn` (:161: `for current_spin = 1:num_spins`, rows stored at :181) over several
 * GPUs of one node from a single caller (the reference's host language has no process group: Julia's
 * `learn(samples, RISE(), HIP(devices = 0:7))` binds these).  GPU g owns the nodes [g n / G, (g+1) n / G): one handle and
 * one host thread per device, the sample bits replicated, no communication while solving.
 *
 *   gml_multi_create   as gml_problem_create, for the devices listed (a device may be listed more than once): the matrix
 *                      is packed ONCE on the host and each chunk of sign bits is copied to every device while the next
 *                      one is being packed (C4: 0.5 GB per GPU instead of 32.8 GB per GPU)
 *   gml_multi_learn    out: n x P row-major host matrix (may be NULL), rows written by the part that owns them; kkt: n.
 *                      dev_out: NULL, or one device pointer per part (n x P doubles on that part's GPU): the full
 *                      matrix is left on EVERY GPU by one RCCL all-gather over xGMI (librccl is loaded at run time;
 *                      peer copies when it is absent, the device list repeats a GPU, or n is not a multiple of G).
 *                      stats: totals over the parts, times of the slowest part.
 *                      With dev_out every part writes its rows device to device into its own block of dev_out[g]
 *                      (nothing returns to the host on the way) and the all-gather runs in place.
 *   gml_multi_info     gather_kind (>= 32 bytes or NULL): how the last dev_out gather was done
 *                      ("rccl-allgather", "peer-copy", "host" when none was asked for)
 *   gml_multi_part_stats  parts: ndev gml_stats, the statistics of every part of the last gml_multi_learn (iterations,
 *                      node evaluations, t_total ...): exposes stragglers among the node shards
 */
typedef struct gml_multi gml_multi;
int gml_multi_create(const void *samples, int dtype, int64_t K, int64_t n, int64_t ld, int col_major, int order,
                     const int *devices, int ndev, gml_multi **out);
int gml_multi_info(const gml_multi *m, int64_t *n, int64_t *K, double *M, int64_t *P, int *ndev, char *gather_kind);
int gml_multi_learn(gml_multi *m, int formulation, double regularizer_c, const gml_opts *opts, double *out, double *kkt,
                    gml_stats *stats, double **dev_out);
int gml_multi_part_stats(const gml_multi *m, gml_stats *parts);
/* diagnostics of the collective path, as text (nbuf >= 160 holds it): whether librccl was loaded, whether ncclCommInitAll succeeded
 * (with ncclGetErrorString's text when not) and which path the last dev_out gather took */
int gml_multi_diag(const gml_multi *m, char *buf, int nbuf);
void gml_multi_destroy(gml_multi *m);

/* Timing hook for the benchmark: runs `steps` full objective+gradient passes over the local
 * nodes at the given theta ((node1-node0) x P, reference layout, host) on the handle's
 * stream and returns the average device time of the dominant kernels measured with HIP
 * events.  kernel_ms[0] = forward (energies + pointwise), [1] = backward (gradient),
 * [2] = whole pass. */
int gml_bench_pass(gml_problem *p, int formulation, int precision, const double *theta,
                   int steps, int warmup, double kernel_ms[3]);

/* The same with Theta resident in HBM: uploaded once, then warmup + steps passes back to back without host round
 * trips (how a device-side optimiser would drive the operator); f ((node1-node0)) and g ((node1-node0) x P) of the
 * last pass are returned if not NULL.  kernel_ms[3] = device time per pass; step_ms (steps doubles, or NULL) = the
 * device time of every timed pass. */
int gml_bench_pass_resident(gml_problem *p, int formulation, int precision, const double *theta, int steps,
                            int warmup, double kernel_ms[4], double *f_out, double *g_out, double *step_ms);

#ifdef __cplusplus
}
#endif
#endif /* GML_H */
