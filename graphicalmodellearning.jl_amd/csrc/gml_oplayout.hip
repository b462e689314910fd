// Layout kernels of the device-pointer operator calls (gml_objgrad_batch / gml_hessvec_batch with theta, f, g, vec, hv in HBM:
// include/gml.h): the caller's rows in the REFERENCE's parameter order (pairwise: slot j <-> spin j, slot u = the field --
// the u-th column of nodal_stat is s_u, GraphicalModelLearning.jl:162; multi-body: the key order of :94-104) <-> the internal
// column layout the passes run on, without a host round trip.
#include "gml_dev.h"

namespace gml {

// X[r][col(r, j)] = theta[r ld + j] for the rows with rowcol[r] >= 0 (X zeroed by the caller); *bad |= 1 on a non-finite entry
__global__ __launch_bounds__(256) void k_ref_to_internal(const double *__restrict__ theta, int64_t ld, int64_t P, int64_t Qp,
                                                         const int *__restrict__ rowcol, int64_t cconst, const int32_t *__restrict__ cols,
                                                         double *__restrict__ X, int *__restrict__ bad) {
    const int64_t r = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int u = rowcol[r];
    if (j >= P || u < 0) return;
    const double v = theta[r * ld + j];
    if (!isfinite(v)) atomicOr(bad, 1);
    const int64_t c = cols ? cols[r * P + j] : (j == u ? cconst : j);
    X[r * Qp + c] = v;
}

// the selected rows back: f[r] = F[r] (logRISE: log Z), g[r ld + j] = G[r][col(r, j)] (logRISE: / Z, :279)
__global__ __launch_bounds__(256) void k_internal_to_ref(const double *__restrict__ G, const double *__restrict__ F, int64_t Qp, int64_t P,
                                                         int64_t ld, const int *__restrict__ rowcol, const uint8_t *__restrict__ sel,
                                                         int64_t cconst, const int32_t *__restrict__ cols, int logz, double *__restrict__ f,
                                                         double *__restrict__ g) {
    const int64_t r = blockIdx.y, j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (!sel[r]) return;
    const double z = F ? F[r] : 1.0;
    if (j == 0 && f) f[r] = logz ? log(z) : z;
    if (j >= P || !g) return;
    const int u = rowcol[r];
    const int64_t c = cols ? cols[r * P + j] : (j == u ? cconst : j);
    const double v = G[r * Qp + c];
    g[r * ld + j] = logz ? v / z : v;
}

// Hessian-vector rows back (Gz: grad Z in the reference order, [r][P]).  logRISE: Hess log Z = Hess Z / Z - g g^T with g = grad Z / Z (:279): hv = Hv / Z - g (g . vec);
// one workgroup per row (the dot product first)
__global__ __launch_bounds__(256) void k_hv_to_ref(const double *__restrict__ Hv, const double *__restrict__ Gz, const double *__restrict__ F,
                                                   int64_t Qp, int64_t P, int64_t ld, const int *__restrict__ rowcol, int64_t cconst,
                                                   const int32_t *__restrict__ cols, int logz, const double *__restrict__ vec,
                                                   double *__restrict__ hv) {
    const int64_t r = blockIdx.x;
    const int u = rowcol[r];
    if (u < 0) return;
    __shared__ double red[256];
    const double z = logz ? F[r] : 1.0;
    double gv = 0.0;
    if (logz) {
        for (int64_t j = threadIdx.x; j < P; j += 256) {
            gv += Gz[r * P + j] / z * vec[r * ld + j];
        }
        red[threadIdx.x] = gv;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        gv = red[0];
    }
    for (int64_t j = threadIdx.x; j < P; j += 256) {
        const int64_t c = cols ? cols[r * P + j] : (j == u ? cconst : j);
        double v = Hv[r * Qp + c];
        if (logz) v = v / z - Gz[r * P + j] / z * gv;
        hv[r * ld + j] = v;
    }
}

void launch_ref_to_internal(const double *theta, int64_t ld, int64_t R, int64_t P, int64_t Qp, const int *rowcol, int64_t cconst,
                            const int32_t *cols, double *X, int *bad, hipStream_t st) {
    if (R > 0)
        hipLaunchKernelGGL(k_ref_to_internal, dim3((unsigned)((P + 255) / 256), (unsigned)R), dim3(256), 0, st, theta, ld, P, Qp, rowcol, cconst,
                           cols, X, bad);
}
void launch_internal_to_ref(const double *G, const double *F, int64_t R, int64_t Qp, int64_t P, int64_t ld, const int *rowcol, const uint8_t *sel,
                            int64_t cconst, const int32_t *cols, int logz, double *f, double *g, hipStream_t st) {
    if (R > 0)
        hipLaunchKernelGGL(k_internal_to_ref, dim3((unsigned)((P + 255) / 256), (unsigned)R), dim3(256), 0, st, G, F, Qp, P, ld, rowcol, sel, cconst,
                           cols, logz, f, g);
}
void launch_hv_to_ref(const double *Hv, const double *Gz, const double *F, int64_t R, int64_t Qp, int64_t P, int64_t ld, const int *rowcol,
                      int64_t cconst, const int32_t *cols, int logz, const double *vec, double *hv, hipStream_t st) {
    if (R > 0)
        hipLaunchKernelGGL(k_hv_to_ref, dim3((unsigned)R), dim3(256), 0, st, Hv, Gz, F, Qp, P, ld, rowcol, cconst, cols, logz, vec, hv);
}

} // namespace gml
