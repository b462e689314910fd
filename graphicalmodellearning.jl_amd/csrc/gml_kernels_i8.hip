// Exact fixed-point device pass of the learn() hot path on the int8 matrix cores (gfx950,
// v_mfma_i32_32x32x32_i8).
//
// Idea: the statistics are +-1 (GraphicalModelLearning.jl:162, :107), so both contractions of
// the objective/gradient pass,
//     A[r][k] = sum_c Theta[r][c] X[k][c]        (energies, inner sum of :170 / :196)
//     G[r][c] = sum_k V[r][k]     X[k][c]        (gradient, :205-207)
// become EXACT integer GEMMs once the real operand is written in balanced base-256 digits
// ("limbs"):  Theta[r][c] = sigma_r * sum_l 256^l t_l[r][c],  V[r][k] = tau_r * sum_l 256^l v_l[r][k],
// t_l, v_l in [-128,127].  Each limb plane is one int8 operand of the i8 MFMA (2x the bf16
// rate), the i32 accumulators cannot overflow (|sum| <= 128 * 2^24), and the limb planes are
// recombined in int64 / FP64 exactly.  The only roundings are the two quantisations (sigma_r,
// sigma_r is a power of two, tau_r = bound/2.13e9, chosen per node: 8*LF-2 resp. 31 significant bits), so the result is
// deterministic and independent of tiling, split-K order and GPU count.
//
// Layout of the limb planes ("planar tiles"): rows are grouped by 32-node tile `t` and limb `l`:
//   Tq row (t*LF + l)*32 + rl  holds limb l of node row t*32+rl, pitch Qfp      (forward B operand)
//   Vq row (t*LB + l)*32 + rl  holds limb l of V   row t*32+rl, pitch Kp        (backward A operand)
// so that a wave's 32x32 MFMA tiles of the different limbs share lane <-> node and
// register <-> sample, and the limbs combine lane-locally.
#include "../../include/gml.h"
#include "gml_dev.h"
#include <string>

namespace gml {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int LB = 4; // limbs of V (30 significant bits relative to the per-node bound)

struct I8Ws {
    int64_t rows = 0; // capacity in node rows (multiple of 32)
    int LF = 4;
    int8_t *Tq = nullptr, *Vq = nullptr;
    int32_t *Gacc = nullptr;
    double *sigma = nullptr, *tau = nullptr, *invtau = nullptr;
    long long *qconst = nullptr, *csum = nullptr, *asum = nullptr;
    int *pairs = nullptr;
};

// ------------------------------------------------------------------------------------------
// quantise Theta rows into limb planes.  One workgroup per node row.
// ------------------------------------------------------------------------------------------
template <int LF>
__global__ __launch_bounds__(256) void k_quant_theta(const double *__restrict__ Theta,
                                                     const int *__restrict__ rowcol, int64_t Qp,
                                                     int64_t Qfp, int64_t cconst, double wmax, int form,
                                                     int8_t *__restrict__ Tq, double *__restrict__ sigma,
                                                     double *__restrict__ tau, double *__restrict__ invtau,
                                                     long long *__restrict__ qconst) {
    const int r = blockIdx.x;
    if (rowcol[r] < 0) return;
    const double *th = Theta + (int64_t)r * Qp;
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double mx = 0.0;
    for (int64_t c = tid; c < Qfp; c += 256) mx = fmax(mx, fabs(th[c]));
    if (tid == 0) mx = fmax(mx, fabs(th[cconst]));
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    // sigma = 2^(ex - (8LF-2)) with mx < 2^ex  =>  |q| <= 2^(8LF-2)
    int ex = 0;
    if (mx > 0) (void)frexp(mx, &ex);
    const double sg = ldexp(1.0, ex - (8 * LF - 2));
    const double isg = ldexp(1.0, (8 * LF - 2) - ex);
    const int tile = r >> 5, rl = r & 31;
    double sabs = 0.0;
    for (int64_t c = tid; c < Qfp; c += 256) {
        long long q = (long long)rint(th[c] * isg);
        sabs += fabs((double)q);
#pragma unroll
        for (int l = 0; l < LF; ++l) {
            const long long dgt = ((q + 128) & 255) - 128;
            q = (q - dgt) >> 8;
            Tq[((int64_t)(tile * LF + l) * 32 + rl) * Qfp + c] = (int8_t)dgt;
        }
    }
    long long q0 = 0;
    if (tid == 0) {
        q0 = (long long)rint(th[cconst] * isg);
        sabs += fabs((double)q0);
    }
    red[tid] = sabs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        // |E| <= sigma * sum|q|  (|X| <= 1)  =>  bound on |V|
        const double emax = red[0] * sg;
        const double B = (form == 2) ? 2.0 * wmax : wmax * exp(emax);
        // |V|/tau <= 2.13e9: the largest magnitude whose 4 balanced base-256 digits fit the packed
        // (q + 0x80808080) ^ 0x80808080 form used by the forward epilogue
        const double t = B * (1.0 + 1e-12) / 2130000000.0;
        sigma[r] = sg;
        qconst[r] = q0;
        tau[r] = t;
        invtau[r] = 1.0 / t;
    }
}

// ------------------------------------------------------------------------------------------
// shared GEMM pieces: LDS tiles of [rows][64 bytes] with the 16-byte slots XOR-swizzled by
// (row>>2)&3, which makes the ds_read_b128 fragment reads (lane = row) conflict-free.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

#define MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)

// exp(x) for |x| < 700 to ~1e-15 relative: 2^(n/64) table (in LDS) times a degree-6 polynomial
__device__ __forceinline__ double exp_tab(double x, const double *__restrict__ tab) {
    const double t = rint(x * 92.33248261689366);      // 64/ln2
    double r = fma(t, -0.01083042469326756, x);          // ln2/64, high part (low 21 bits zero: t*hi exact)
    r = fma(t, -2.9815858269852933e-12, r);                 // low part
    double p = 1.3888888888888889e-03;  // 1/720
    p = fma(p, r, 8.3333333333333332e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int n = (int)t;
    return ldexp(tab[n & 63] * p, n >> 6);
}

// ------------------------------------------------------------------------------------------
// forward: C[k][m] = sum_c Xs[k][c] * Tq[m][c] on i8 MFMA, then the pointwise epilogue
//   E = s * sigma_r * (sum_l 256^l C_l + q0),  V = -w_k exp(-E) s  (RISE / logRISE),
//   V -> LB balanced limbs -> Vq planes (via an LDS transpose so that global stores are 16 B).
// Workgroup = 4 waves (2 along samples x 2 node tiles): 64*WM samples x 2 node tiles x LF limbs.
// ------------------------------------------------------------------------------------------
template <int WM, int LF>
__global__ __launch_bounds__(256, 2) void k_fwd_i8(
    const int8_t *__restrict__ Xs, const int8_t *__restrict__ Xt, const int8_t *__restrict__ Tq,
    const int *__restrict__ rowcol, const int *__restrict__ pairs, int npairs, const double *__restrict__ w,
    const double *__restrict__ sigma, const long long *__restrict__ qconst, const double *__restrict__ invtau,
    int64_t Qp, int64_t Qfp, int64_t Kp, int ntiles_k, int form, int8_t *__restrict__ Vq,
    long long *__restrict__ csum, long long *__restrict__ asum, double *__restrict__ fsum) {
    constexpr int BM = 64 * WM, BN = 64 * LF;
    constexpr int TILE = (BM + BN) * 64;
    constexpr int PITCH = 32 * WM + 16; // staging row pitch (bytes)
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    __shared__ double etab[64];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    if (tid < 64) etab[tid] = exp2((double)tid / 64.0);

    // XCD-aware mapping: the blocks of one sample tile (all node-tile pairs) share an XCD / L2
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int st = (bi / npairs) * 8 + xcd, pr = bi % npairs;
    if (st >= ntiles_k) return;
    const int64_t k0 = (int64_t)st * BM;
    const int t0 = pairs[2 * pr], t1 = pairs[2 * pr + 1];
    const int mytile = wn ? t1 : t0;

    // ---- global -> register staging descriptors -----------------------------------------
    const int8_t *asrc[WM];
    int adst[WM];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
        const int q = tid + 256 * j, row = q >> 2, slot = q & 3;
        asrc[j] = Xs + (k0 + row) * Qp + slot * 16;
        adst[j] = lds_off(row, slot);
    }
    const int8_t *bsrc[LF];
    int bdst[LF];
#pragma unroll
    for (int j = 0; j < LF; ++j) {
        const int q = tid + 256 * j, row = q >> 2, slot = q & 3;
        const int half = row / (32 * LF), l = (row >> 5) % LF, rl = row & 31;
        int tl = half ? t1 : t0;
        if (tl < 0) tl = t0;
        bsrc[j] = Tq + ((int64_t)(tl * LF + l) * 32 + rl) * Qfp + slot * 16;
        bdst[j] = BM * 64 + lds_off(row, slot);
    }

    v16i acc[WM][LF];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int l = 0; l < LF; ++l)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;

    const int nk = (int)(Qfp / 64);
    v4i ra[WM], rb[LF];
#pragma unroll
    for (int j = 0; j < WM; ++j) ra[j] = *reinterpret_cast<const v4i *>(asrc[j]);
#pragma unroll
    for (int j = 0; j < LF; ++j) rb[j] = *reinterpret_cast<const v4i *>(bsrc[j]);
#pragma unroll
    for (int j = 0; j < WM; ++j) *reinterpret_cast<v4i *>(lds + adst[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < LF; ++j) *reinterpret_cast<v4i *>(lds + bdst[j]) = rb[j];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (kt & 1) * TILE, nxt = TILE - cur;
        if (kt + 1 < nk) {
            const int64_t c0 = (int64_t)(kt + 1) * 64;
#pragma unroll
            for (int j = 0; j < WM; ++j) ra[j] = *reinterpret_cast<const v4i *>(asrc[j] + c0);
#pragma unroll
            for (int j = 0; j < LF; ++j) rb[j] = *reinterpret_cast<const v4i *>(bsrc[j] + c0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = 2 * t + h;
            v4i fa[WM], fb[LF];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                fa[i] = *reinterpret_cast<const v4i *>(lds + cur + lds_off(wm * 32 * WM + i * 32 + lr, slot));
#pragma unroll
            for (int l = 0; l < LF; ++l)
                fb[l] = *reinterpret_cast<const v4i *>(lds + cur + BM * 64 + lds_off((wn * LF + l) * 32 + lr, slot));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int l = 0; l < LF; ++l) acc[i][l] = MFMA_I8(fa[i], fb[l], acc[i][l]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int j = 0; j < WM; ++j) *reinterpret_cast<v4i *>(lds + nxt + adst[j]) = ra[j];
#pragma unroll
            for (int j = 0; j < LF; ++j) *reinterpret_cast<v4i *>(lds + nxt + bdst[j]) = rb[j];
        }
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // lane <-> node row (lr), register e <-> sample (e&3) + 8*(e>>2) + 4*h within the 32-sample tile
    const bool valid = mytile >= 0;
    const int r = (valid ? mytile : 0) * 32 + lr;
    const int rc = valid ? rowcol[r] : -1;
    const bool active = rc >= 0;
    const double sg = active ? sigma[r] : 0.0;
    const double q0 = active ? (double)qconst[r] : 0.0;
    const double it = active ? invtau[r] : 0.0;
    int8_t *stage = lds + wave * (LB * 32 * PITCH);
    long long cs = 0, as = 0;
    double fp = 0.0;
    const int64_t kw = k0 + wm * 32 * WM; // first sample of this wave
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t kk = kw + i * 32 + 8 * g + 4 * h;
            unsigned sw = 0;
            if (active) sw = *reinterpret_cast<const unsigned *>(Xt + (int64_t)rc * Kp + kk);
            unsigned dj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * g + j;
                double a = (double)acc[i][LF - 1][e];
#pragma unroll
                for (int l = LF - 2; l >= 0; --l) a = fma(a, 256.0, (double)acc[i][l][e]);
                a += q0;
                const int sb = (int)(int8_t)((sw >> (8 * j)) & 0xff); // s_u^k (0 on padded samples)
                const double s = (double)sb;
                const double E = s * sg * a;
                const double wk = w[kk + j] * it;
                double val;
                if (form == 2) { // RPLE (:317)
                    const double ex = exp_tab(2.0 * E, etab);
                    const double sgm = 1.0 / (1.0 + ex);
                    val = -2.0 * wk * sgm * s;
                    const double tt = -2.0 * E;
                    fp += w[kk + j] * (tt > 0 ? tt + log1p(exp(-tt)) : log1p(exp(tt)));
                } else { // RISE (:196,:204) / logRISE Z (:279)
                    val = -wk * exp_tab(-E, etab) * s;
                }
                const int vq = (int)rint(val);
                cs += vq;
                as -= (long long)vq * sb;
                dj[j] = ((unsigned)vq + 0x80808080u) ^ 0x80808080u; // 4 balanced base-256 digits
            }
            // 4 samples x 4 limbs byte transpose -> one dword per limb plane
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                const unsigned t01 = __builtin_amdgcn_perm(dj[1], dj[0], sel);
                const unsigned t23 = __builtin_amdgcn_perm(dj[3], dj[2], sel);
                const unsigned pl = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                *reinterpret_cast<unsigned *>(stage + (lb * 32 + lr) * PITCH + i * 32 + 8 * g + 4 * h) = pl;
            }
        }
    }
    cs += __shfl_xor(cs, 32);
    as += __shfl_xor(as, 32);
    if (active && h == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs);
        atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), (unsigned long long)as);
    }
    if (form == 2) {
        fp += __shfl_xor(fp, 32);
        if (active && h == 0) unsafeAtomicAdd(&fsum[r], fp);
    }
    __syncthreads();
    // coalesced store of the wave's LB*32 rows x 32*WM bytes
    constexpr int CH = 2 * WM; // 16-byte chunks per row
#pragma unroll
    for (int ps = 0; ps < (LB * 32 * CH) / 64; ++ps) {
        const int q = ps * 64 + lane, row = q / CH, slot = q % CH;
        const int lb = row >> 5, rl = row & 31;
        const v4i dat = *reinterpret_cast<const v4i *>(stage + row * PITCH + slot * 16);
        if (valid && rowcol[mytile * 32 + rl] >= 0)
            *reinterpret_cast<v4i *>(Vq + ((int64_t)(mytile * LB + lb) * 32 + rl) * Kp + kw + slot * 16) = dat;
    }
}

// ------------------------------------------------------------------------------------------
// backward: Gacc[m][c] += sum_k Vq[m][k] * Xt[c][k]  (i32, split-K with integer atomics).
// Workgroup tile: 128 rows (one node tile x 4 limbs) x 256 columns; waves 2 x 2, each 64 x 128.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_bwd_i8(const int8_t *__restrict__ Vq, const int8_t *__restrict__ Xt,
                                                const int *__restrict__ groups, int ngroups, int nNt,
                                                int64_t Qfp, int64_t Kp, int64_t kchunk, int nsplit,
                                                int32_t *__restrict__ Gacc) {
    constexpr int BM = 128, BN = 256, TILE = (BM + BN) * 64;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int T = ngroups * nNt;
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int chunk = (bi / T) * 8 + xcd, ti = bi % T;
    if (chunk >= nsplit) return;
    const int tile = groups[ti / nNt], nt = ti % nNt;
    const int64_t kb = (int64_t)chunk * kchunk;
    const int64_t ke = (kb + kchunk < Kp) ? kb + kchunk : Kp;
    const int64_t n0 = (int64_t)nt * BN;

    const int8_t *asrc[2], *bsrc[4];
    int adst[2], bdst[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + 256 * j, row = q >> 2, slot = q & 3;
        asrc[j] = Vq + ((int64_t)tile * 128 + row) * Kp + slot * 16;
        adst[j] = lds_off(row, slot);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = tid + 256 * j, row = q >> 2, slot = q & 3;
        int64_t c = n0 + row;
        if (c >= Qfp) c = Qfp - 1; // columns beyond the matrix: computed, never stored
        bsrc[j] = Xt + c * Kp + slot * 16;
        bdst[j] = BM * 64 + lds_off(row, slot);
    }
    v16i acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0;

    v4i ra[2], rb[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const v4i *>(asrc[j] + kb);
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const v4i *>(bsrc[j] + kb);
#pragma unroll
    for (int j = 0; j < 2; ++j) *reinterpret_cast<v4i *>(lds + adst[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<v4i *>(lds + bdst[j]) = rb[j];
    __syncthreads();

    int it = 0;
    for (int64_t kk = kb; kk < ke; kk += 64, ++it) {
        const int cur = (it & 1) * TILE, nxt = TILE - cur;
        const bool more = kk + 64 < ke;
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ra[j] = *reinterpret_cast<const v4i *>(asrc[j] + kk + 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const v4i *>(bsrc[j] + kk + 64);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = 2 * t + h;
            v4i fa[2], fb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[i] = *reinterpret_cast<const v4i *>(lds + cur + lds_off(wm * 64 + i * 32 + lr, slot));
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
                fb[jn] = *reinterpret_cast<const v4i *>(lds + cur + BM * 64 + lds_off(wn * 128 + jn * 32 + lr, slot));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) acc[i][jn] = MFMA_I8(fa[i], fb[jn], acc[i][jn]);
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<v4i *>(lds + nxt + adst[j]) = ra[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<v4i *>(lds + nxt + bdst[j]) = rb[j];
        }
        __syncthreads();
    }
    // C layout: column (lane&31) <-> Xt row (c), register e <-> Vq row (e&3)+8*(e>>2)+4*h
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int64_t c = n0 + wn * 128 + jn * 32 + lr;
            if (c < Qfp) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mrow = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    atomicAdd(&Gacc[((int64_t)tile * 128 + mrow) * Qfp + c], acc[i][jn][e]);
                }
            }
        }
}

// G[r][c] = tau_r * sum_l 256^l Gacc[(t*4+l)*32+rl][c];  G[r][cconst] = tau_r * csum[r];
// f[r] = tau_r * asum[r]  (= sum_k w_k exp(-E) for RISE / logRISE; RPLE keeps its FP64 sum)
__global__ __launch_bounds__(256) void k_finalize_i8(const int32_t *__restrict__ Gacc, const double *__restrict__ tau,
                                                     const long long *__restrict__ csum,
                                                     const long long *__restrict__ asum,
                                                     const int *__restrict__ rowcol, int64_t Qp, int64_t Qfp,
                                                     int64_t cconst, int form, int want_grad,
                                                     double *__restrict__ G, double *__restrict__ f) {
    const int r = blockIdx.y;
    if (rowcol[r] < 0) return;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double t = tau[r];
    if (c == 0 && form != 2) f[r] = t * (double)asum[r];
    if (!want_grad || c >= Qp) return;
    double v = 0.0;
    if (c < Qfp) {
        const int tile = r >> 5, rl = r & 31;
        long long s = 0;
#pragma unroll
        for (int l = LB - 1; l >= 0; --l) s = s * 256 + (long long)Gacc[((int64_t)(tile * LB + l) * 32 + rl) * Qfp + c];
        v = t * (double)s;
    } else if (c == cconst) {
        v = t * (double)csum[r];
    }
    G[(int64_t)r * Qp + c] = v;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#define I8CHK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            if (err) *err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                 \
            return e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP;                                 \
        }                                                                                             \
    } while (0)

void i8_get_v(void *p, const int8_t **Vq, const double **tau) {
    I8Ws *w = static_cast<I8Ws *>(p);
    *Vq = w ? w->Vq : nullptr;
    *tau = w ? w->tau : nullptr;
}

void i8_free(void *p) {
    I8Ws *w = static_cast<I8Ws *>(p);
    if (!w) return;
    void *ptrs[] = {w->Tq, w->Vq, w->Gacc, w->sigma, w->tau, w->invtau, w->qconst, w->csum, w->asum, w->pairs};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    delete w;
}

static int i8_ensure(void **wsp, const DevProblem &d, int Rp, int LF, std::string *err) {
    I8Ws *w = static_cast<I8Ws *>(*wsp);
    if (w && w->rows >= Rp && w->LF == LF) return GML_OK;
    if (w) i8_free(w);
    *wsp = nullptr;
    w = new I8Ws();
    w->LF = LF;
    I8CHK(hipMalloc(&w->Tq, (size_t)Rp * LF * d.Qfp));
    I8CHK(hipMalloc(&w->Vq, (size_t)Rp * LB * d.Kp));
    I8CHK(hipMalloc(&w->Gacc, sizeof(int32_t) * (size_t)Rp * LB * d.Qfp));
    I8CHK(hipMalloc(&w->sigma, sizeof(double) * Rp));
    I8CHK(hipMalloc(&w->tau, sizeof(double) * Rp));
    I8CHK(hipMalloc(&w->invtau, sizeof(double) * Rp));
    I8CHK(hipMalloc(&w->qconst, sizeof(long long) * Rp));
    I8CHK(hipMalloc(&w->csum, sizeof(long long) * Rp));
    I8CHK(hipMalloc(&w->asum, sizeof(long long) * Rp));
    I8CHK(hipMalloc(&w->pairs, sizeof(int) * (Rp / 32 + 2)));
    I8CHK(hipMemset(w->Tq, 0, (size_t)Rp * LF * d.Qfp));
    I8CHK(hipMemset(w->Vq, 0, (size_t)Rp * LB * d.Kp));
    w->rows = Rp;
    *wsp = w;
    return GML_OK;
}

int i8_limbs_forward() {
    static int lf = [] {
        const char *e = getenv("GML_I8_LF");
        int v = e ? atoi(e) : 5;
        return (v == 3 || v == 4 || v == 5) ? v : 5;
    }();
    return lf;
}

template <int WM, int LF>
static void launch_fwd(const I8Ws *w, const DevProblem &d, const int *dRowcol, int npairs, int form, double *dF,
                       hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * LF;
    constexpr int gemm_bytes = 2 * (BM + BN) * 64;
    constexpr int stage_bytes = 4 * LB * 32 * (32 * WM + 16);
    constexpr int shmem = gemm_bytes > stage_bytes ? gemm_bytes : stage_bytes;
    const int ntk = (int)(d.Kp / BM);
    const int grid = ((ntk + 7) / 8) * 8 * npairs;
    hipLaunchKernelGGL((k_fwd_i8<WM, LF>), dim3(grid), dim3(256), shmem, st, d.Xs, d.Xt, w->Tq, dRowcol, w->pairs, npairs,
                       d.w, w->sigma, w->qconst, w->invtau, d.Qp, d.Qfp, d.Kp, ntk, form, w->Vq, w->csum, w->asum, dF);
}

int i8_pass(void **wsp, const DevProblem &d, const double *dTheta, const int *dRowcol, const int *hRowcol,
            const int *hGroups, int ngroups, int Rp, int form, bool want_grad, double *dF, double *dG, hipStream_t st,
            hipEvent_t *ev, std::string *err) {
    (void)hRowcol;
    if (d.Kp > (int64_t)1 << 24) {
        if (err) *err = "GML_PREC_I8X supports up to 2^24 configurations per handle (i32 accumulators)";
        return GML_EUNSUPPORTED;
    }
    const int LF = i8_limbs_forward();
    int rc = i8_ensure(wsp, d, Rp, LF, err);
    if (rc) return rc;
    I8Ws *w = static_cast<I8Ws *>(*wsp);
    // pairs of active node tiles (a workgroup of the forward kernel serves two)
    int npairs = (ngroups + 1) / 2;
    {
        int tmp[4096];
        if (ngroups + 1 > 4096) {
            if (err) *err = "too many node tiles";
            return GML_EUNSUPPORTED;
        }
        for (int i = 0; i < ngroups; ++i) tmp[i] = hGroups[i];
        if (ngroups & 1) tmp[ngroups] = -1;
        I8CHK(hipMemcpyAsync(w->pairs, tmp, sizeof(int) * 2 * npairs, hipMemcpyHostToDevice, st));
        I8CHK(hipStreamSynchronize(st)); // tmp is a stack buffer
    }
    I8CHK(hipMemsetAsync(w->csum, 0, sizeof(long long) * Rp, st));
    I8CHK(hipMemsetAsync(w->asum, 0, sizeof(long long) * Rp, st));
    if (want_grad) I8CHK(hipMemsetAsync(w->Gacc, 0, sizeof(int32_t) * (size_t)Rp * LB * d.Qfp, st));
    switch (LF) {
    case 3:
        hipLaunchKernelGGL((k_quant_theta<3>), dim3(Rp), dim3(256), 0, st, dTheta, dRowcol, d.Qp, d.Qfp, d.cconst, d.wmax,
                           form, w->Tq, w->sigma, w->tau, w->invtau, w->qconst);
        break;
    case 5:
        hipLaunchKernelGGL((k_quant_theta<5>), dim3(Rp), dim3(256), 0, st, dTheta, dRowcol, d.Qp, d.Qfp, d.cconst, d.wmax,
                           form, w->Tq, w->sigma, w->tau, w->invtau, w->qconst);
        break;
    default:
        hipLaunchKernelGGL((k_quant_theta<4>), dim3(Rp), dim3(256), 0, st, dTheta, dRowcol, d.Qp, d.Qfp, d.cconst, d.wmax,
                           form, w->Tq, w->sigma, w->tau, w->invtau, w->qconst);
    }
    if (ev) I8CHK(hipEventRecord(ev[0], st));
    switch (LF) {
    case 3: launch_fwd<2, 3>(w, d, dRowcol, npairs, form, dF, st); break;
    case 5: launch_fwd<2, 5>(w, d, dRowcol, npairs, form, dF, st); break;
    default: launch_fwd<2, 4>(w, d, dRowcol, npairs, form, dF, st);
    }
    if (ev) I8CHK(hipEventRecord(ev[1], st));
    if (want_grad) {
        const int nNt = (int)((d.Qfp + 255) / 256);
        int nsplit = 16;
        int64_t kchunk = (d.Kp + nsplit - 1) / nsplit;
        kchunk = (kchunk + 63) / 64 * 64;
        if (kchunk < 1024) kchunk = 1024;
        nsplit = (int)((d.Kp + kchunk - 1) / kchunk);
        const int T = ngroups * nNt;
        const int grid = ((nsplit + 7) / 8) * 8 * T;
        hipLaunchKernelGGL(k_bwd_i8, dim3(grid), dim3(256), 2 * (128 + 256) * 64, st, w->Vq, d.Xt, w->pairs, ngroups, nNt,
                           d.Qfp, d.Kp, kchunk, nsplit, w->Gacc);
    }
    if (ev) I8CHK(hipEventRecord(ev[2], st));
    hipLaunchKernelGGL(k_finalize_i8, dim3((unsigned)((d.Qp + 255) / 256), (unsigned)Rp), dim3(256), 0, st, w->Gacc, w->tau,
                       w->csum, w->asum, dRowcol, d.Qp, d.Qfp, d.cconst, form, want_grad ? 1 : 0, dG, dF);
    I8CHK(hipGetLastError());
    return GML_OK;
}

} // namespace gml
