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
//   Tq image (t, kt) = [LF*32 rows][64 B]: row l*32 + rl holds limb l of node row t*32+rl, columns
//                      [64kt, 64kt+64); images are contiguous (t major)            (forward B operand)
//   Vq image (t, k/64) = [LB*32 rows][64 B]: row l*32 + rl holds limb l of V row t*32+rl, samples
//                      [64(k/64), +64); images are contiguous (t major), see vq_off()  (backward A operand)
// so that a wave's 32x32 MFMA tiles of the different limbs share lane <-> node and
// register <-> sample, and the limbs combine lane-locally.
//
// The +-1 operand of the forward GEMM is kept as ONE BIT per entry (Xb, bit set <=> -1) and expanded
// to 0/1 bytes in registers: sum_c q_c x_c = sum_c q_c - 2 sum_c q_c b_c.  An int8 image of it would
// make the kernel L2->LDS bandwidth bound (measured: 98 MAC per loaded byte against the ~140 the CU
// needs); with bits the loop loads 12 KB instead of 26.6 KB per 64-column step.
#include "gml_i8.h"
#include <algorithm>
#include <string>
#include <type_traits>

namespace gml {

// ------------------------------------------------------------------------------------------
// quantise Theta rows into limb planes.  One workgroup per node row.
// ------------------------------------------------------------------------------------------
template <int LF>
__global__ __launch_bounds__(256) void k_quant_theta(const double *__restrict__ Theta, const int *__restrict__ srow,
                                                     const int *__restrict__ rowcol, int slot0, int64_t Qp,
                                                     int64_t Qfp, int64_t cconst, double wmax, int form, int hv,
                                                     const int *__restrict__ vmap, const double *__restrict__ tauV,
                                                     int8_t *__restrict__ Tq, double *__restrict__ sigma,
                                                     double *__restrict__ tau, double *__restrict__ invtau,
                                                     long long *__restrict__ qconst, const double *__restrict__ tauovr,
                                                     double vdiv /* largest |V| / tau the planes of this pass hold */,
                                                     double vsrc_scale /* hv: unit of the V planes read, in multiples of tauV */) {
    const int r = slot0 + blockIdx.x; // slot
    if (rowcol[r] < 0) return;
    const double *th = Theta + (int64_t)srow[r] * Qp;
    __shared__ double red[256];
    __shared__ double red1[256];
    const int tid = threadIdx.x;
    double mx = 0.0, s1 = 0.0;
    for (int64_t c = tid; c < Qfp; c += 256) {
        mx = fmax(mx, fabs(th[c]));
        if (LF > 5) s1 += fabs(th[c]);
    }
    if (tid == 0) {
        mx = fmax(mx, fabs(th[cconst]));
        if (LF > 5) s1 += fabs(th[cconst]);
    }
    red[tid] = mx;
    red1[tid] = s1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] = fmax(red[tid], red[tid + s]);
            if (LF > 5) red1[tid] += red1[tid + s];
        }
        __syncthreads();
    }
    mx = red[0];
    s1 = red1[0];
    __syncthreads();
    // sigma = 2^(ex - (8LF-2)) with mx < 2^ex  =>  |q| <= 2^(8LF-2)
    int ex = 0;
    if (mx > 0) (void)frexp(mx, &ex);
    int sx = ex - (8 * LF - 2);
    if (LF > 5) { // 54-bit digits: keep sum_c |q_c| below 2^61, so that every integer sum of the pass fits 64 bits
        int e1 = 0;
        if (s1 > 0) (void)frexp(s1 * 1.0000001, &e1); // sum |theta| < 2^e1
        if (e1 - 61 > sx) sx = e1 - 61;
    }
    const double sg = ldexp(1.0, sx);
    const double isg = ldexp(1.0, -sx);
    const int tile = r >> 5, rl = r & 31;
    const int64_t nk = Qfp >> 6;
    double sabs = 0.0;
    long long ssum = 0; // sum_c q_c: the energy of the all-(+1) configuration (the forward GEMM runs on b = [x = -1])
    for (int64_t c = tid; c < Qfp; c += 256) {
        long long q = (long long)rint(th[c] * isg);
        sabs += fabs((double)q);
        ssum += q;
        int8_t *img = Tq + ((((int64_t)tile * nk + (c >> 6)) * LF) * 32 + rl) * 64 + (c & 63);
#pragma unroll
        for (int l = 0; l < LF; ++l) {
            const long long dgt = ((q + 128) & 255) - 128;
            q = (q - dgt) >> 8;
            img[l * 32 * 64] = (int8_t)dgt;
        }
    }
    long long q0 = 0;
    if (tid == 0) {
        q0 = (long long)rint(th[cconst] * isg);
        sabs += fabs((double)q0);
    }
    __shared__ long long redl[256];
    red[tid] = sabs;
    redl[tid] = ssum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            redl[tid] += redl[tid + s];
        }
        __syncthreads();
    }
    q0 += redl[0];
    if (tid == 0) {
        // |E| <= sigma * sum|q|  (|X| <= 1)
        const double emax = red[0] * sg;
        double t, it;
        if (hv) {
            // Hessian-vector pass: the row is a direction p, the forward epilogue forms u_k = h_k (x_k . p) with
            // h_k = tau_V |q_V| the weights of the row's last objective pass (slot vmap[r]); |x_k . p| <= emax, so
            // u_k / (tau_V emax) = |q_V| * (x_k . p) / emax is an integer of at most 31 bits.
            // (hv == 2: the products go to the backward GEMM in 2 limbs instead of 4 -- 15 bits of the largest one, which is
            // ample for the matrix-free Newton-CG that stops at a 5 % residual; the factor 1.01 keeps |u| inside the
            // +-32639 of two balanced digits)
            const double pn = (emax > 0.0 ? emax : 1.0) * (hv == 2 ? 65536.0 * 1.01 : 1.0);
            t = tauV[vmap[r]] * vsrc_scale * pn;
            it = 1.0 / pn;
        } else {
            const double B = (form == 2) ? 2.0 * wmax : wmax * exp(emax); // bound on |V|
            // |V|/tau <= 2.13e9: the largest magnitude whose 4 balanced base-256 digits fit the packed
            // (q + 0x80808080) ^ 0x80808080 form used by the forward epilogue.
            // The bound exp(sum|theta|) can exceed the largest actual |V| by many orders of magnitude (dense
            // theta); the caller then re-runs the row with tau taken from the largest |V| the first pass saw, and
            // the solver passes max|V| of its previous pass times exp(||step||_1), which bounds the new weights.
            t = B * (1.0 + 1e-12) / vdiv;
            if (tauovr && tauovr[r] > 0.0 && tauovr[r] < t) t = tauovr[r]; // a tighter rigorous scale from the caller
            it = 1.0 / t;
        }
        sigma[r] = sg;
        qconst[r] = q0;
        tau[r] = t;
        invtau[r] = it;
    }
}

// V / tau of one element of the exp forms: -s rint(w/tau exp(-s E) + dither), E = s Ea.  Ea to 3e-10 relative before
// the rounding.  FP64 range reduction with one FMA (the product t * ln2/64 is not rounded inside an FMA), FP32
// polynomial for expm1 of the reduced argument, table of 2^(j/64), exponent added as an integer, and the final rounding
// to an integer through the 1.5 * 2^52 trick, after adding a dither in [-1/2, 1/2) that is a fixed function of
// (node, sample): the rounding is then "stochastic" -- still deterministic and within one unit, but uncorrelated across
// samples.  Round-to-nearest is coherent whenever a sparse theta row leaves only a few distinct energies (thousands of
// samples share each rounding error), which made the realised error of f and grad approach the K * tau / 2 worst case
// instead of ~ sqrt(K) * tau.
// The epilogue is bound by the number of vector instructions, so this is written for few of them:
//   * sb (bit 0: s = +1) flips the sign of Ea going in (x = -s E) and of the result coming out by adding sb << 31 to
//     the high word -- round-half-even is symmetric, so rounding -y gives minus the rounding of y;
//   * everything from the weight on is scaled by 2^32 (wk32 = 2^32 w / tau; exact): the dither is then the hash itself,
//     converted int -> double, and the integer is read off below 1.5 * 2^84;
//   * the table holds 2^(j/64) with j << 14 taken off the high word: the exponent of 2^(n >> 6), n = 64 q + j, goes on
//     as n << 14 (= (q << 20) + (j << 14)) in one shift-add.
__device__ __forceinline__ int vq_exp(double Ea, unsigned sb, double wk32, unsigned dh, const double *__restrict__ tabb) {
    const double MAGIC = 6755399441055744.0;                  // 1.5 * 2^52
    const double MAGIC32 = 6755399441055744.0 * 4294967296.0; // 1.5 * 2^84: rounds to multiples of 2^32
    const int flip = (int)(sb << 31);
    const double x = __hiloint2double(__double2hiint(Ea) + flip, __double2loint(Ea)); // -s E
    const double tm = fma(x, 92.33248261689366, MAGIC);                               // 64/ln2
    const int n = __double2loint(tm);
    const double t = tm - MAGIC;
    const double r = fma(t, -0.010830424696249145, x); // ln2/64
    const float rf = (float)r;
    float d = fmaf(rf, 4.1666668e-02f, 1.6666667e-01f);
    d = fmaf(d, rf, 0.5f);
    d = fmaf(d, rf, 1.0f);
    d = d * rf; // expm1(r)
    const double tj0 = tabb[n & 63];
    const double tj = __hiloint2double((int)((unsigned)__double2hiint(tj0) + ((unsigned)n << 14)), __double2loint(tj0)); // 2^(n/64)
    const double res = fma(tj, (double)d, tj);
    const double y = fma(wk32, res, (double)(int)dh); // 2^32 (|V| / tau + dither)
    const double ys = __hiloint2double(__double2hiint(y) + flip, __double2loint(y));
    return __double2loint(ys + MAGIC32);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA ring shared by both GEMM kernels.  A stage image is [AROWS + BROWS][64 bytes]; it is
// filled by global_load_lds_dwordx4 in 1-KB pieces (16 rows): the LDS destination of a piece is
// linear (base + lane*16), so the XOR swizzle of lds_off() is applied to the per-lane SOURCE
// address instead.  Three stages: the loads of tile kt+2 are issued right after the barrier that
// retires tile kt-1, and stay in flight across two compute phases (counted vmcnt, raw s_barrier).
// ------------------------------------------------------------------------------------------

template <int NP>
__device__ __forceinline__ void ring_issue(const int8_t *const (&src)[NP], int64_t off, int8_t *stage_base, int wave,
                                           int npiece) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        int pc = wave + 4 * j;
        if (pc >= npiece) pc = npiece - 1; // duplicate piece: keeps the per-wave vmcnt count uniform
        __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + off), (lptr_t)(stage_base + pc * 1024), 16, 0, 0);
    }
}

template <int NP>
__device__ __forceinline__ void ring_issue8(const int8_t *const (&src)[NP], int64_t off, int8_t *stage_base, int wave,
                                            int npiece) {
    (void)npiece;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + off), (lptr_t)(stage_base + (wave + 8 * j) * 1024), 16, 0, 0);
}

template <int NP>
__device__ __forceinline__ void ring_wait(bool more) {
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------
// Bit images.  The only stored form of the samples is Sb, the spin-major sign bits (gml_bits.h); the two MFMA
// operand images are derived from it: a statistic of key S is the XOR of the rows of its spins.
// ------------------------------------------------------------------------------------------
// Sb from sample-major bytes S [K][n]: thread <-> (spin i fastest, word w)
__global__ __launch_bounds__(256) void k_bits_from_rows(const int8_t *__restrict__ S, int64_t K, int64_t n, int64_t wpr,
                                                        unsigned *__restrict__ Sb) {
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x, w = blockIdx.x;
    if (i >= n) return;
    unsigned v = 0;
    for (int j = 0; j < 32; ++j) {
        const int64_t k = w * 32 + j;
        if (k < K && S[k * n + i] < 0) v |= 1u << j;
    }
    Sb[i * wpr + w] = v;
}

// Sb from spin-major bytes St [n][ld]: thread <-> (word w fastest, spin i)
__global__ __launch_bounds__(256) void k_bits_from_cols(const int8_t *__restrict__ St, int64_t K, int64_t ld, int64_t wpr,
                                                        unsigned *__restrict__ Sb) {
    const int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x, i = blockIdx.x;
    if (w * 32 >= K) return;
    const int8_t *row = St + i * ld + w * 32;
    unsigned v = 0;
    for (int j = 0; j < 32; ++j)
        if (w * 32 + j < K && row[j] < 0) v |= 1u << j;
    Sb[i * wpr + w] = v;
}

void launch_spin_bits(const int8_t *S, bool spin_major, int64_t K, int64_t n, int64_t ld, int64_t Kp, unsigned *Sb, hipStream_t st) {
    const int64_t wpr = Kp / 32, nw = (K + 31) / 32;
    if (spin_major)
        hipLaunchKernelGGL(k_bits_from_cols, dim3((unsigned)n, (unsigned)((nw + 255) / 256)), dim3(256), 0, st, S, K, ld, wpr, Sb);
    else
        hipLaunchKernelGGL(k_bits_from_rows, dim3((unsigned)nw, (unsigned)((n + 255) / 256)), dim3(256), 0, st, S, K, n, wpr, Sb);
}

// natural-order word w of statistic column c: XOR of the rows of its spins (keys [Qf][ko], -1 = unused slot)
__device__ __forceinline__ unsigned stat_word(const unsigned *__restrict__ Sb, int64_t wpr, const int32_t *__restrict__ keys, int ko,
                                              int64_t Qf, int64_t c, int64_t w) {
    if (c >= Qf) return 0u; // zero padding columns
    unsigned v = 0;
    for (int t = 0; t < ko; ++t) {
        const int i = keys[c * ko + t];
        if (i >= 0) v ^= Sb[(int64_t)i * wpr + w];
    }
    return v;
}

// Forward operand Xb (sample-major): one thread builds the 32 dwords (k = 32w .. 32w+31, kt, h) from the 32 words
// of the columns that dword covers (bit j <-> column 64kt + xb_col(j, h)) by a 32 x 32 bit transpose.
// Pieces [K/128][nk][128 samples][2 h] dwords, one LDS-DMA instruction moves one 1-KB piece.
__global__ __launch_bounds__(256) void k_build_xb(const unsigned *__restrict__ Sb, int64_t wpr, const int32_t *__restrict__ keys,
                                                  int ko, int64_t Qf, int nk, unsigned *__restrict__ Xb) {
    const int64_t w = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
    const int h = threadIdx.x & 1, kt = blockIdx.y;
    if (w >= wpr) return;
    unsigned a[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) a[j] = stat_word(Sb, wpr, keys, ko, Qf, 64 * (int64_t)kt + xb_col(j, h), w);
    transpose32(a);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int64_t k = w * 32 + s;
        Xb[((((k >> 7) * nk + kt) * 128) + (k & 127)) * 2 + h] = a[s];
    }
}

// Backward operand Xtb (feature-major): dword (c, kt, h) = xtb_from_natural(word 2kt + h of column c);
// pieces [Qc/128][Kp/64][128 columns][2 h], Qc = Qfp rounded up to 256 (columns beyond Qf: zero bits).
__global__ __launch_bounds__(256) void k_build_xtb(const unsigned *__restrict__ Sb, int64_t wpr, const int32_t *__restrict__ keys,
                                                   int ko, int64_t Qf, int64_t nkk, unsigned *__restrict__ Xtb) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x; // (kt, h) = natural word index
    const int64_t c = blockIdx.x;                              // columns on x: there can be more than 65535 of them
    if (j >= 2 * nkk) return;
    const unsigned v = xtb_from_natural(stat_word(Sb, wpr, keys, ko, Qf, c, j));
    Xtb[((((c >> 7) * nkk + (j >> 1)) * 128) + (c & 127)) * 2 + (j & 1)] = v;
}

int64_t xtb_bytes(const DevProblem &d) { return (d.Qfp + 255) / 256 * 256 * (d.Kp / 8); }

void launch_pack_bits(const DevProblem &d, hipStream_t st) {
    const int nk = (int)(d.Qfp / 64);
    const int64_t wpr = d.Kp / 32;
    hipLaunchKernelGGL(k_build_xb, dim3((unsigned)((wpr + 127) / 128), (unsigned)nk), dim3(256), 0, st, d.Sb, wpr, d.keys, d.ko, d.Qf, nk,
                       d.Xb);
    const int64_t nkk = d.Kp / 64, Qc = (d.Qfp + 255) / 256 * 256;
    hipLaunchKernelGGL(k_build_xtb, dim3((unsigned)Qc, (unsigned)((2 * nkk + 255) / 256)), dim3(256), 0, st, d.Sb, wpr, d.keys, d.ko,
                       d.Qf, nkk, d.Xtb);
}

// byte forms, on demand: the configurations back as +-1 bytes [kk][n] (gml_problem_get_spins), and the feature-major
// byte image Xt [Qp][Kp] of the FP64 path (padding samples and columns zero; the constant column is set by the caller)
__global__ __launch_bounds__(256) void k_unpack_spins(const unsigned *__restrict__ Sb, int64_t wpr, int64_t n, int64_t k0, int64_t kk,
                                                      int8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x, kl = blockIdx.x, k = k0 + kl;
    if (i >= n || kl >= kk) return;
    out[kl * n + i] = ((Sb[i * wpr + (k >> 5)] >> (k & 31)) & 1u) ? (int8_t)-1 : (int8_t)1;
}

void launch_unpack_spins(const DevProblem &d, int64_t k0, int64_t kk, int8_t *out, hipStream_t st) {
    hipLaunchKernelGGL(k_unpack_spins, dim3((unsigned)kk, (unsigned)((d.n + 255) / 256)), dim3(256), 0, st, d.Sb, d.Kp / 32, d.n, k0, kk, out);
}

__global__ __launch_bounds__(256) void k_expand_xt(const unsigned *__restrict__ Sb, int64_t wpr, const int32_t *__restrict__ keys, int ko,
                                                   int64_t Qf, int64_t K, int64_t Kp, int8_t *__restrict__ Xt) {
    const int64_t w = (int64_t)blockIdx.y * 256 + threadIdx.x, c = blockIdx.x;
    if (w >= wpr) return;
    const unsigned v = stat_word(Sb, wpr, keys, ko, Qf, c, w);
    unsigned out[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        unsigned o = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t k = w * 32 + 4 * q + b;
            const unsigned byte = k < K ? (((v >> (4 * q + b)) & 1u) ? 0xFFu : 0x01u) : 0u;
            o |= byte << (8 * b);
        }
        out[q] = o;
    }
    v4i *dst = reinterpret_cast<v4i *>(Xt + c * Kp + w * 32);
    dst[0] = (v4i){(int)out[0], (int)out[1], (int)out[2], (int)out[3]};
    dst[1] = (v4i){(int)out[4], (int)out[5], (int)out[6], (int)out[7]};
}

void launch_expand_xt(const DevProblem &d, int8_t *Xt, hipStream_t st) {
    if (d.Qf > 0)
        hipLaunchKernelGGL(k_expand_xt, dim3((unsigned)d.Qf, (unsigned)((d.Kp / 32 + 255) / 256)), dim3(256), 0, st, d.Sb, d.Kp / 32, d.keys,
                           d.ko, d.Qf, d.K, d.Kp, Xt);
}

// ------------------------------------------------------------------------------------------
// forward: C[k][m] = sum_c b[k][c] * Tq[m][c] on i8 MFMA (b = [x = -1] from the bit image), then the
// pointwise epilogue
//   E = s * sigma_r * (q0 + S - 2 sum_l 256^l C_l),  V = -w_k exp(-E) s  (RISE / logRISE),
//   V -> LB balanced limbs -> Vq planes (via an LDS transpose so that global stores are 16 B).
// Workgroup = 4 waves along the samples: 256 samples x one 32-node tile x LF limb planes.
// Stage image of the 4-deep LDS-DMA ring: 2 KB of bits (two 128-sample pieces) + the (tile, kt) image
// of Tq.  The A fragments never touch LDS as bytes: each lane expands its dword of bits in registers.
// ------------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void ring_wait_ahead(int ahead) {
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int LF, int FORM /* 0: exp forms (RISE, logRISE), 2: RPLE; Hessian-vector products: 3 (exp forms), 4 (RPLE) */,
          bool WANTF,
          bool WIDE /* more than 32768 statistics columns: |acc_l| <= 128 Qfp no longer leaves room for the int32 pairing */,
          bool UNIW /* every real sample has the weight wuni (all counts equal): no weight loads.  A template parameter, not
                       a run-time test: a branch per element would put each of the epilogue's 32 dependent chains (range
                       reduction -> table read -> polynomial -> rounding) into its own basic block and serialise them */>
__global__ __launch_bounds__(256, 2) void k_fwd_i8(
    const unsigned *__restrict__ Xb, const unsigned *__restrict__ Sb, const int8_t *__restrict__ Tq,
    const int *__restrict__ rowcol, const int *__restrict__ groups, int ngroups, const double *__restrict__ w,
    const double *__restrict__ sigma, const long long *__restrict__ qconst, const double *__restrict__ invtau,
    int64_t Kp, int ntiles_k, int nk /* 64-column steps */, double wuni /* > 0: every real sample has this weight */,
    int64_t Kreal, int8_t *__restrict__ Vq, long long *__restrict__ csum, long long *__restrict__ asum,
    double *__restrict__ fsum, unsigned *__restrict__ mmax,
    // Hessian-vector forms only: the limb planes of V written by the rows' last objective pass, the slot that holds
    // them for each slot of this pass, and their scales
    const int8_t *__restrict__ Vsrc, const int *__restrict__ vmap, const double *__restrict__ tauV,
    int vsrc_lbt /* planes of a source image */, int vsrc_pl0 /* first of the 4 planes read */, double vsrc_scale /* their unit / tauV */,
    // sub-sampled passes (Hessian-vector products over a part of the configurations): compact sample tile t stands for the
    // tile (t / part_tiles) * chunk_tiles + t % part_tiles -- the first part_tiles tiles of every split-K chunk of the
    // backward kernel.  chunk_tiles == part_tiles: every configuration.
    int chunk_tiles, int part_tiles) {
    constexpr int WM = 2;                 // 32-sample MFMA tiles per wave
    constexpr bool HV = FORM >= 3;
    constexpr int BR = 32 * LF;           // rows of the Tq image
    constexpr int NPIECE = 2 + BR / 16, NP = (NPIECE + 3) / 4;
    // Ring stages hold DS consecutive 64-column steps: one barrier per DS steps (the waves of a workgroup then re-align
    // half as often, and the LDS reads of a stage's second step issue under the MFMAs of its first).
    constexpr int DS = 2;
    constexpr int STEP = NPIECE * 1024, STAGE = DS * STEP, NS = 3;
    constexpr int RING = NS * STAGE;
    extern __shared__ __attribute__((aligned(16))) int8_t lds[]; // ring, then the exp (and log) tables
    double *etab = reinterpret_cast<double *>(lds + RING);

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    if (tid < 64) {
        const double v = exp2((double)tid / 64.0);
        // exp forms: the table of vq_exp(), j << 14 taken off the high word
        etab[tid] = FORM == 0 ? __hiloint2double(__double2hiint(v) - (tid << 14), __double2loint(v)) : v;
    }
    if (FORM == 2 && tid < 64) { // log table for RPLE: c_j = 1 + (j + 1/2)/64 -> 1/c_j, log c_j
        const double cj = 1.0 + ((double)tid + 0.5) / 64.0;
        etab[64 + tid] = 1.0 / cj;
        etab[128 + tid] = log(cj);
    }
    __syncthreads(); // tables visible to every wave (the ring uses raw s_barrier without an LDS wait)

    // XCD-aware L2 blocking.  Blocks b and b+8 share an XCD (round-robin dispatch); XCD x owns the
    // sample tiles st = 8*i + x.  Within an XCD: groups of TG node tiles (outer), sample tiles
    // (middle), the TG node tiles (inner): Tq of the group stays resident in the XCD's L2 over the
    // sweep and each bit piece is fetched once per node-tile group.  The last group holds ngroups % TG tiles; the grid
    // has no idle workgroups beyond the sample tiles that pad ntiles_k to a multiple of 8 (a node-sharded rank runs few
    // node tiles: half of its launch would otherwise be workgroups that start only to exit).
    constexpr int TG = 8;
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int ntk8 = (ntiles_k + 7) >> 3;
    const int nfull = ngroups / TG, per_full = ntk8 * TG;
    int st, gi;
    if (bi < nfull * per_full) {
        const int rem = bi % per_full;
        st = (rem / TG) * 8 + xcd;
        gi = (bi / per_full) * TG + rem % TG;
    } else {
        const int lastn = ngroups - nfull * TG, rem = bi - nfull * per_full;
        st = (rem / lastn) * 8 + xcd;
        gi = nfull * TG + rem % lastn;
    }
    if (st >= ntiles_k) return;
    if (chunk_tiles != part_tiles) st = (st / part_tiles) * chunk_tiles + st % part_tiles;
    const int64_t k0 = (int64_t)st * 256;
    if (k0 >= Kp) return;
    const int mytile = groups[gi];

    // per-lane source of each 1-KB piece this wave loads, and its advance per 64-column step
    const int8_t *src[NP];
    int adv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        int pc = wave + 4 * j;
        if (pc >= NPIECE) pc = NPIECE - 1; // duplicate piece: keeps the per-wave vmcnt count uniform
        if (pc < 2) {
            src[j] = reinterpret_cast<const int8_t *>(Xb) + ((int64_t)(2 * st + pc) * nk) * 1024 + lane * 16;
            adv[j] = 1024;
        } else {
            const int row = (pc - 2) * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3); // XOR swizzle applied to the source (LDS side is linear)
            src[j] = Tq + ((int64_t)mytile * nk * BR + row) * 64 + slot * 16;
            adv[j] = BR * 64;
        }
    }
    const int nst = (nk + DS - 1) / DS; // ring stages of this tile
    auto issue = [&](int ks) { // stage ks = steps DS ks .. DS ks + DS - 1 (a step beyond the last one: the last one again, so
                               // that every stage counts the same number of loads for the vmcnt waits)
        int8_t *stage_base = lds + (ks % NS) * STAGE;
#pragma unroll
        for (int sub = 0; sub < DS; ++sub) {
            int kt = DS * ks + sub;
            kt = kt < nk ? kt : nk - 1;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int pc = wave + 4 * j;
                if (pc >= NPIECE) pc = NPIECE - 1;
                __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)kt * adv[j]), (lptr_t)(stage_base + sub * STEP + pc * 1024), 16, 0, 0);
            }
        }
    };

    v16i acc[WM][LF]; // first written by the peeled step 0 below (C operand = the constant 0: no clearing moves)

    // the epilogue's per-lane inputs are fetched now, so that their latency hides under the GEMM
    const int r = mytile * 32 + lr;
    const int rc = rowcol[r];
    const bool active = rc >= 0;
    // the node's sign bits for this wave's 64 samples (word i <-> MFMA tile i), shifted so that bit 8g + j is this
    // lane's sample 8g + 4h + j of the tile
    unsigned sgn[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) sgn[i] = active ? (Sb[(int64_t)rc * (Kp >> 5) + ((k0 + wave * 64) >> 5) + i] >> (4 * h)) : 0u;
    // samples at or beyond Kreal are padding (they carry no weight): this lane's element (i, g, j) sits 32 i + 8 g + j
    // samples after its first one, k0 + 64 wave + 4 h
    const int64_t left = Kreal - (k0 + wave * 64 + 4 * h);
    const int nreal = left > 64 ? 64 : (left < 0 ? 0 : (int)left);
    const double sg = active ? sigma[r] : 0.0;
    const double q0 = active ? (double)qconst[r] : 0.0;
    const double it = active ? invtau[r] : 0.0;

    // Two workgroups share a CU, one wave of each per SIMD.  The wave that is in its GEMM gets the issue priority over
    // the one that is in its epilogue: the matrix pipe is the scarcer resource (-3 % forward time, interleaved A/B).
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) issue(s);
    auto gemm_stage = [&](int ks, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        ring_wait_ahead<DS * NP>(nst - 1 - ks > NS - 2 ? NS - 2 : nst - 1 - ks); // NS - 2 later stages may still be in flight
        if (ks + NS - 1 < nst) issue(ks + NS - 1);
#pragma unroll
        for (int sub = 0; sub < DS; ++sub) {
            if (sub > 0 && DS * ks + sub >= nk) break; // (an odd number of steps: the last stage is half full)
            const int8_t *cur = lds + (ks % NS) * STAGE + sub * STEP;
            unsigned vb[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = wave * 64 + i * 32 + lr;
                vb[i] = *reinterpret_cast<const unsigned *>(cur + (row >> 7) * 1024 + (((row & 127) * 2 + h) << 2));
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4i fa[WM], fb[LF];
#pragma unroll
                for (int l = 0; l < LF; ++l)
                    fb[l] = *reinterpret_cast<const v4i *>(cur + 2048 + lds_off(l * 32 + lr, 2 * t + h));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][e] = (int)((vb[i] >> (4 * t + e)) & 0x01010101u);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int l = 0; l < LF; ++l) {
                        if (FIRST && sub == 0 && t == 0) acc[i][l] = MFMA_I8(fa[i], fb[l], ((v16i){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}));
                        else acc[i][l] = MFMA_I8(fa[i], fb[l], acc[i][l]);
                    }
            }
        }
    };
    gemm_stage(0, std::true_type{}); // nk >= 1: Qfp >= 64
    for (int ks = 1; ks < nst; ++ks) gemm_stage(ks, std::false_type{});
    __builtin_amdgcn_s_setprio(0);
    // ---- epilogue ----------------------------------------------------------------------------
    // lane <-> node row (lr), register e <-> sample (e&3) + 8*(e>>2) + 4*h within the 32-sample tile.  The
    // Vq image stores a step's samples in the order vq_pos() (gml_dev.h), in which this lane's 16 samples of
    // tile i are 16 contiguous bytes per limb: no LDS transpose, two 16-byte stores per limb.
    const int form = FORM;
    int8_t *vimg = Vq + vq_off(mytile * 32 + lr, 0, k0 + wave * 64, Kp) + h * 32; // row (limb 0, lr) of the wave's image
    const int8_t *vsrc = nullptr; // Hessian-vector forms: the same bytes of the row's V image
    double tvh = 0.0;
    if (HV && active) {
        const int vs = vmap[r];
        vsrc = Vsrc + vq_off(vs, vsrc_pl0, k0 + wave * 64, Kp, vsrc_lbt) + h * 32;
        tvh = tauV[vs] * vsrc_scale;
    }
    long long cs = 0, as = 0;
    double fp = 0.0;
    int mx = 0;
    const int64_t kw = k0 + wave * 64; // first sample of this wave
    const double sgq0 = sg * q0, wk32 = 4294967296.0 * (wuni * it); // 2^32 w / tau (vq_exp)
    double sg2 = -2.0 * sg;
    // dither of the V rounding: golden-ratio (Weyl) sequence in the global sample index, offset per node --
    // independent of tiling, node sharding and compaction, so results stay bit-identical across GPU counts
    const unsigned dh0 = (unsigned)rc * 0x85EBCA6Bu + (unsigned)(kw + 4 * h) * 0x9E3779B9u;
    if constexpr (FORM == 0) {
        // Exp forms (RISE, logRISE): the arithmetic of vq_exp(), laid out in STAGES over 8 elements at a time (two 4-sample
        // groups).  Every stage is 8 independent copies of a short chain, fenced by sched_barriers: a wave in its epilogue
        // then issues back to back instead of waiting out the 16-20 cycle latency of each dependent FP64 instruction (the
        // element-at-a-time form left the scheduler, at 200+ live registers, emitting each element's chain serially).
        // Fewer instructions per element as well: the sign is applied to the rounded magnitude in integers (one bit-field
        // extract serves both sign flips), sum_k V comes from dot4 over the packed digit planes, max|V| from the unsigned
        // magnitudes, and padding samples are masked in a branch only the last sample tile takes.
        constexpr double MAGIC = 6755399441055744.0, MAGIC32 = 6755399441055744.0 * 4294967296.0;
        constexpr unsigned GOLD = 0x9E3779B9u, CB = 0x80808080u;
        const int wleft = (int)((Kreal - kw) < 64 ? (Kreal - kw) : 64); // wave-uniform: real samples among this wave's 64
        int csl[LB] = {0, 0, 0, 0};
        unsigned mxu = 0;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const unsigned nsg = ~sgn[i]; // bit 8g + j set <=> s = +1
            v4i pl[LB];
#pragma unroll
            for (int hg = 0; hg < 2; ++hg) {
                // Layers of 8 independent instructions each, fenced (SB): whatever order the scheduler picks inside a layer, a
                // result is not needed before 8 issue slots later.
#define SB __builtin_amdgcn_sched_barrier(0)
                double a[8], Ea[8], wk[8], tm[8], x[8], tj0[8], yy[8];
                int mneg[8], nn[8];
                float rf[8], dd[8];
                unsigned mag[8];
                // A: exact recombination of the limb planes (pairs in int32, then FP64; all planes through FP64 when WIDE)
                if (WIDE || LF != 5) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = 8 * hg + q;
                        if (WIDE) {
                            a[q] = (double)acc[i][LF - 1][e];
#pragma unroll
                            for (int l = LF - 2; l >= 0; --l) a[q] = fma(a[q], 256.0, (double)acc[i][l][e]);
                        } else if (LF == 4) {
                            const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                            const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                            a[q] = fma((double)mid, 65536.0, (double)lo);
                        } else {
                            const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                            a[q] = fma((double)acc[i][2][e], 65536.0, (double)lo);
                        }
                    }
                    SB;
                } else { // LF == 5, the production form, layer by layer
                    int lo[8], mid[8];
                    double c4[8], cm[8], cl[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = 8 * hg + q;
                        lo[q] = acc[i][0][e] + (acc[i][1][e] << 8);
                        mid[q] = acc[i][2][e] + (acc[i][3][e] << 8);
                        c4[q] = (double)acc[i][4][e];
                    }
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        cm[q] = (double)mid[q];
                        cl[q] = (double)lo[q];
                    }
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = fma(c4[q], 65536.0, cm[q]);
                    SB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = fma(a[q], 65536.0, cl[q]);
                    SB;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pos = 8 * (2 * hg + (q >> 2)) + (q & 3);
                    Ea[q] = fma(a[q], sg2, sgq0);
                    // -1 iff s = +1 (v_bfe_i32 spelled out: the generic lowering is a shift pair, and the compiler then
                    // re-derives the two sign flips below from the shifted word with an and + an arithmetic shift each)
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(mneg[q]) : "v"(nsg), "n"(pos));
                    if (!UNIW) wk[q] = w[kw + i * 32 + 8 * (2 * hg + (q >> 2)) + 4 * h + (q & 3)];
                }
                SB;
                // B: x = -s E, range reduction n = rint(64 x / ln2), r = x - n ln2 / 64
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x[q] = __hiloint2double(__double2hiint(Ea[q]) + (mneg[q] << 31), __double2loint(Ea[q]));
                    tm[q] = fma(x[q], 92.33248261689366, MAGIC);
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    nn[q] = __double2loint(tm[q]);
                    tm[q] = tm[q] - MAGIC;
                    tj0[q] = etab[nn[q] & 63]; // 2^(j/64) (exponent bits of j << 14 taken off: see the table's construction)
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tm[q], -0.010830424696249145, x[q]); // r
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) rf[q] = (float)x[q];
                SB;
                // C: expm1(r) in FP32
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(rf[q], 4.1666668e-02f, 1.6666667e-01f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(dd[q], rf[q], 0.5f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = fmaf(dd[q], rf[q], 1.0f);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) dd[q] = dd[q] * rf[q];
                SB;
                // D: 2^32 (w / tau exp(-E) + dither), rounded to an integer magnitude
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3);
                    tj0[q] = __hiloint2double((int)((unsigned)__double2hiint(tj0[q]) + ((unsigned)nn[q] << 14)), __double2loint(tj0[q]));
                    x[q] = (double)dd[q];
                    yy[q] = (double)(int)(dh0 + (unsigned)idx * GOLD); // the dither
                }
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = fma(tj0[q], x[q], tj0[q]); // exp(-E)
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) yy[q] = fma(UNIW ? wk32 : 4294967296.0 * (wk[q] * it), x[q], yy[q]);
                SB;
#pragma unroll
                for (int q = 0; q < 8; ++q) mag[q] = (unsigned)__double2loint(yy[q] + MAGIC32); // >= 0: y > -2^31
                if (UNIW && wleft < 64) { // the last sample tile: padding samples carry no weight
                    asm volatile("; padding samples" ::: "memory"); // (keeps this a branch: as selects it costs every tile 2 instructions per element)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (i * 32 + 8 * (2 * hg + (q >> 2)) + (q & 3) >= nreal) mag[q] = 0u;
                }
                SB;
#undef SB
                // E: sign, 4 balanced base-256 digits per sample, 4 samples x 4 limbs byte transpose
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    unsigned dj[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = 4 * gg + j;
                        mxu = mag[q] > mxu ? mag[q] : mxu;
                        if (WANTF) as += (long long)mag[q];
                        // V / tau = -s |V| / tau = (mag ^ m) - m, then the 4 balanced digits (v + CB) ^ CB: one v_xad
                        unsigned tq;
                        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(tq) : "v"(mag[q]), "v"(mneg[q]), "v"(CB - (unsigned)mneg[q]));
                        dj[j] = tq ^ CB;
                    }
#pragma unroll
                    for (int lb = 0; lb < LB; ++lb) {
                        const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                        const unsigned t01 = __builtin_amdgcn_perm(dj[1], dj[0], sel);
                        const unsigned t23 = __builtin_amdgcn_perm(dj[3], dj[2], sel);
                        const unsigned pk = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                        pl[lb][2 * hg + gg] = (int)pk;
                        csl[lb] = __builtin_amdgcn_sdot4((int)pk, 0x01010101, csl[lb], false); // sum of the 4 digits
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (active) {
#pragma unroll
                for (int lb = 0; lb < LB; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
            }
        }
        cs = (long long)csl[0] + 256ll * csl[1] + 65536ll * csl[2] + 16777216ll * csl[3];
        cs += __shfl_xor(cs, 32);
        as += __shfl_xor(as, 32);
        const unsigned mo = (unsigned)__shfl_xor((int)mxu, 32);
        mxu = mo > mxu ? mo : mxu;
        if (active && h == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs);
            if (WANTF) atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), (unsigned long long)as);
            atomicMax(&mmax[r], mxu);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        v4i pl[LB], pv[LB];
        if (HV) {
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) pv[lb] = active ? *reinterpret_cast<const v4i *>(vsrc + lb * 32 * 64 + i * 16) : (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t kk = kw + i * 32 + 8 * g + 4 * h;
            unsigned dj[4], dv[4];
            if (HV) { // the 4 balanced digits of V of each of the group's 4 samples (inverse of the transpose below)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned sel = ((4u + j) << 8) | (unsigned)j;
                    const unsigned t01 = __builtin_amdgcn_perm((unsigned)pv[1][g], (unsigned)pv[0][g], sel);
                    const unsigned t23 = __builtin_amdgcn_perm((unsigned)pv[3][g], (unsigned)pv[2][g], sel);
                    dv[j] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
                }
            }
            if (FORM == 2) { // RPLE: gate each 4-sample group on the previous one (its longer arithmetic otherwise
                             // interleaves across groups and spills); pure arithmetic floats across sched_barriers
                asm volatile("" : "+v"(sg2), "+v"(fp));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * g + j;
                // exact recombination of the limb planes: pairs in int32 (|acc_l| <= 128 Qfp <= 2^22, so
                // |acc_l + 256 acc_{l+1}| < 2^31), then FP64; beyond 32768 columns every plane goes through FP64
                // (|a| < 2^53 always: a is the integer sum_c q_c b_c with |q_c| <= 2^38)
                double a;
                if (WIDE) {
                    a = (double)acc[i][LF - 1][e];
#pragma unroll
                    for (int l = LF - 2; l >= 0; --l) a = fma(a, 256.0, (double)acc[i][l][e]);
                } else if (LF == 5) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                    a = fma((double)acc[i][4][e], 65536.0, (double)mid);
                    a = fma(a, 65536.0, (double)lo);
                } else if (LF == 4) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    const int mid = acc[i][2][e] + (acc[i][3][e] << 8);
                    a = fma((double)mid, 65536.0, (double)lo);
                } else if (LF == 3) {
                    const int lo = acc[i][0][e] + (acc[i][1][e] << 8);
                    a = fma((double)acc[i][2][e], 65536.0, (double)lo);
                } else { // 2 limbs: directions of the Hessian-vector passes
                    a = (double)(acc[i][0][e] + (acc[i][1][e] << 8));
                }
                const double Ea = fma(a, sg2, sgq0);                // |E| pre-sign: sigma * (q0 + S - 2 A)
                const double dith = (double)(int)(dh0 + (unsigned)(i * 32 + 8 * g + j) * 0x9E3779B9u) * 2.3283064365386963e-10; // [-1/2, 1/2)
                const bool neg = ((sgn[i] >> (8 * g + j)) & 1u) != 0; // s_u^k = -1
                int vq;
                if (HV) {
                    // Hessian-vector product: u_k = h_k (x_k . p), h_k the curvature weight of the row's iterate:
                    // |V_k| for the exp forms, 2a(1 - a/(2w)) with a = |V_k| for RPLE.  In units of tau_V * emax:
                    // u = (h_k / tau_V) * (x_k . p) / emax, |.| <= 2^31 (Ea = x_k . p, it = 1 / emax)
                    const int qv = (int)((dv[j] ^ 0x80808080u) - 0x80808080u);
                    double hh = (double)(qv < 0 ? -qv : qv);
                    if (FORM == 4) {
                        const double wk0 = UNIW ? (i * 32 + 8 * g + j < nreal ? wuni : 0.0) : w[kk + j];
                        hh = wk0 > 0.0 ? 2.0 * hh * (1.0 - hh * tvh / (2.0 * wk0)) : 0.0;
                    }
                    vq = __double2loint(fma(hh, Ea * it, dith) + 6755399441055744.0);
                } else if (FORM == 2) { // RPLE (:317): f = w log(1 + exp(-2E)), V = -2 w s / (1 + exp(2E)), E = s * Ea
                    const double wk0 = UNIW ? (i * 32 + 8 * g + j < nreal ? wuni : 0.0) : w[kk + j];
                    const double E2 = neg ? -2.0 * Ea : 2.0 * Ea;
                    const double u = exp_tab(-fabs(E2), etab); // in (0, 1]
                    const double opu = 1.0 + u;
                    double rc = __builtin_amdgcn_rcp(opu); // 1 / (1 + u), two Newton steps
                    rc = fma(fma(-opu, rc, 1.0), rc, rc);
                    rc = fma(fma(-opu, rc, 1.0), rc, rc);
                    const double sig = E2 >= 0.0 ? u * rc : rc; // 1 / (1 + exp(2E))
                    const int mag = __double2loint(fma(2.0 * wk0 * it, sig, dith) + 6755399441055744.0);
                    vq = neg ? mag : -mag;
                    // log(1 + u), 1 + u in (1, 2]: table of log c_j on 64 intervals + log1p of the residual
                    int jt = (int)(u * 64.0);
                    jt = jt > 63 ? 63 : jt;
                    const double r1 = fma(opu, etab[64 + jt], -1.0); // |r1| <= 1/128
                    double lp = fma(r1, 1.0 / 7.0, -1.0 / 6.0);
                    lp = fma(lp, r1, 0.2);
                    lp = fma(lp, r1, -0.25);
                    lp = fma(lp, r1, 1.0 / 3.0);
                    lp = fma(lp, r1, -0.5);
                    lp = fma(lp, r1, 1.0);
                    const double l1p = fma(lp, r1, etab[128 + jt]);
                    fp += wk0 * ((E2 < 0.0 ? -E2 : 0.0) + l1p);
                } else { // RISE (:196,:204) / logRISE Z (:279): V = -w exp(-E) s
                    const unsigned sb = ~sgn[i] >> (8 * g + j); // bit 0: s = +1
                    const unsigned dh = dh0 + (unsigned)(i * 32 + 8 * g + j) * 0x9E3779B9u;
                    if (UNIW) {
                        vq = vq_exp(Ea, sb, wk32, dh, etab);
                        if (i * 32 + 8 * g + j >= nreal) vq = 0; // padding samples carry no weight
                    } else {
                        vq = vq_exp(Ea, sb, 4294967296.0 * (w[kk + j] * it), dh, etab);
                    }
                    const int nvq = -vq, mag = vq > nvq ? vq : nvq;
                    mx = mag > mx ? mag : mx;
                    if (WANTF) as += mag;
                }
                cs += vq;
                dj[j] = ((unsigned)vq + 0x80808080u) ^ 0x80808080u; // 4 balanced base-256 digits
            }
            // 4 samples x 4 limbs byte transpose -> one dword per limb plane
#pragma unroll
            for (int lb = 0; lb < LB; ++lb) {
                const unsigned sel = ((4u + lb) << 8) | (unsigned)lb;
                const unsigned t01 = __builtin_amdgcn_perm(dj[1], dj[0], sel);
                const unsigned t23 = __builtin_amdgcn_perm(dj[3], dj[2], sel);
                pl[lb][g] = (int)__builtin_amdgcn_perm(t23, t01, 0x05040100u);
            }
        }
        if (active) {
#pragma unroll
            // (plain stores: the two 16-byte halves of a 64-byte row come from two instructions and merge in L2; as
            // non-temporal stores they reach HBM separately -- 8.2 GB written instead of 5.1 -- for 2 % less time)
            for (int lb = 0; lb < LB; ++lb) *reinterpret_cast<v4i *>(vimg + lb * 32 * 64 + i * 16) = pl[lb];
        }
    }
    cs += __shfl_xor(cs, 32);
    as += __shfl_xor(as, 32);
    if (active && h == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&csum[r]), (unsigned long long)cs);
        if (WANTF) atomicAdd(reinterpret_cast<unsigned long long *>(&asum[r]), (unsigned long long)as);
    }
    if (form == 0) {
        const int mo = __shfl_xor(mx, 32);
        mx = mo > mx ? mo : mx;
        if (active && h == 0) atomicMax(&mmax[r], (unsigned)mx);
    }
    if (form == 2) {
        fp += __shfl_xor(fp, 32);
        if (active && h == 0) unsafeAtomicAdd(&fsum[r], fp);
    }
}

// ------------------------------------------------------------------------------------------
// backward: Gacc[m][c] += sum_k Vq[m][k] * b[k][c]  (i32, split-K with integer atomics), b = [x = -1]
// from the feature-major bit image; the gradient is tau * (sum_k V - 2 sum_l 256^l Gacc_l).
// Workgroup tile: 2 node tiles (256 rows of Vq: 4 limbs x 32 nodes each) x 256 columns,
// 8 waves as 2 (M) x 4 (N), each 128 x 64; 4-deep LDS-DMA ring of 18-KB stages (two 8-KB Vq images +
// 2 KB of bits), 2 waves/SIMD.  All (tile, column-tile) blocks of one k-chunk run on one XCD so that
// the chunk's slabs of Vq and of the bit image are fetched from HBM once and shared through that XCD's L2.
// ------------------------------------------------------------------------------------------
template <int TM /* node tiles per workgroup: 1 (4 waves, two workgroups per CU) or 2 (8 waves) */,
          int NL /* limb planes of Vq multiplied: 4; 2 for the products of a 2-limb Hessian-vector pass; 3 for one half of the
                    6 planes of the i8w pass (TM = 1) */>
__global__ __launch_bounds__(256 * TM, 2) void k_bwd_i8(
    const int8_t *__restrict__ Vq, const unsigned *__restrict__ Xtb, const int *__restrict__ groups, int ngroups_t,
    int nNt, int64_t Qfp, int64_t Kp, int64_t kchunk, int nsplit, int32_t *__restrict__ Gacc,
    int chunks_per_plane /* split-K chunks that share one set of i32 accumulators (<= 2^24 configurations: |sum| < 2^31) */,
    int64_t plane_stride /* elements between the accumulator sets */,
    int64_t kpart /* configurations of every chunk that take part (== kchunk: all; less: sub-sampled Hessian-vector products) */,
    int lbt /* limb planes of a Vq image (and of the accumulator rows of a node tile) */, int pl0 /* first plane multiplied */) {
    constexpr int NW = 4 * TM;
    constexpr int AR = 128 * TM, NPIECE = 8 * TM + 2, STAGE = NPIECE * 1024, NS = 4;
    constexpr int WMT = NL, WNT = 2; // wave tile 128 (96, 64) x 64: MFMA tile i <-> limb plane pl0 + i of the node tile
    static_assert(NL == 4 || ((NL == 2 || NL == 3) && TM == 1), "the 2- and 3-limb forms exist for 4-wave workgroups");
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    const int wm = wave % TM, wn = wave / TM; // wn in 0..3
    const int T = ngroups_t * nNt; // ngroups_t = number of TM-groups of node tiles
    const int b = blockIdx.x, xcd = b & 7, bi = b >> 3;
    const int chunk = (bi / T) * 8 + xcd, ti = bi % T; // all tiles of one k-chunk on one XCD
    if (chunk >= nsplit) return;
    Gacc += (int64_t)(chunk / chunks_per_plane) * plane_stride;
    const int gi = ti / nNt, nt = ti % nNt;
    int tiles[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) tiles[t] = groups[gi * TM + t]; // -1: padding (computed on tile 0, not stored)
    const int64_t kb = (int64_t)chunk * kchunk;
    const int64_t ke = (kb + kpart < Kp) ? kb + kpart : Kp;
    const int64_t n0 = (int64_t)nt * 256, nkk = Kp >> 6, kt0 = kb >> 6;

    // 8*TM + 2 pieces over 4*TM waves: waves 0 and 1 load three (the third is a piece of bits), the others two.
    // NL = 2: only the four pieces of limb planes 0 and 1 (rows 0..63 of the image) + the bits: two per wave.
    // NL = 3: the six pieces of three planes + the bits: two per wave.
    const bool three = NL == 4 && wave < 2;
    const int8_t *src[3];
    int adv[3], dst[3];
    const int img = lbt * 2048; // bytes of a Vq image
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int pc = wave + NW * j; // piece of the full stage image: 0 .. 8 TM - 1 rows of Vq, then the bits
        if (NL == 2) pc = j == 0 ? wave : 8 * TM + (wave & 1);
        if (NL == 3) pc = wave + 4 * j < 6 ? wave + 4 * j : 8 * TM + ((wave + 4 * j - 6) & 1);
        dst[j] = pc * 1024;
        if (pc < 8 * TM) {
            int tl = tiles[pc >> 3];
            if (tl < 0) tl = tiles[0];
            const int row = (pc & 7) * 16 + (lane >> 2);
            const int slot = (lane & 3) ^ ((row >> 2) & 3);
            src[j] = Vq + ((int64_t)tl * nkk + kt0) * img + (pl0 * 32 + row) * 64 + slot * 16;
            adv[j] = img;
        } else {
            const int pb = pc < NPIECE ? pc - 8 * TM : 0;
            src[j] = reinterpret_cast<const int8_t *>(Xtb) + ((int64_t)(2 * nt + pb) * nkk + kt0) * 1024 + lane * 16;
            adv[j] = 1024;
        }
    }
    auto issue = [&](int kt) {
        int8_t *stage_base = lds + (kt & (NS - 1)) * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)kt * adv[j]), (lptr_t)(stage_base + dst[j]), 16, 0, 0);
        if (three) __builtin_amdgcn_global_load_lds((gptr_t)(src[2] + (int64_t)kt * adv[2]), (lptr_t)(stage_base + dst[2]), 16, 0, 0);
    };
    v16i acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0;

    const int nk = (int)((ke - kb) / 64);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (three) ring_wait_ahead<3>(nk - 1 - kt);
        else ring_wait_ahead<2>(nk - 1 - kt);
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
        const int8_t *cur = lds + (kt & (NS - 1)) * STAGE;
        unsigned vb[WNT];
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn) {
            const int cw = wn * 64 + jn * 32 + lr;
            vb[jn] = *reinterpret_cast<const unsigned *>(cur + AR * 64 + (cw >> 7) * 1024 + (((cw & 127) * 2 + h) << 2));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            v4i fa[WMT], fb[WNT];
#pragma unroll
            for (int i = 0; i < WMT; ++i)
                fa[i] = *reinterpret_cast<const v4i *>(cur + lds_off(wm * 128 + i * 32 + lr, 2 * t + h));
#pragma unroll
            for (int jn = 0; jn < WNT; ++jn)
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[jn][e] = (int)((vb[jn] >> (4 * t + e)) & 0x01010101u);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int jn = 0; jn < WNT; ++jn) acc[i][jn] = MFMA_I8(fa[i], fb[jn], acc[i][jn]);
        }
    }
    // C layout: column (lane&31) <-> column c, register e <-> Vq row (e&3)+8*(e>>2)+4*h
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int jn = 0; jn < WNT; ++jn) {
            const int64_t c = n0 + wn * 64 + jn * 32 + lr;
            const int grow = wm * 128 + i * 32; // first row of this MFMA tile within the workgroup tile
            const int tl = tiles[grow >> 7];
            if (c < Qfp && tl >= 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mrow = (grow & 127) + (e & 3) + 8 * (e >> 2) + 4 * h;
                    atomicAdd(&Gacc[((int64_t)tl * lbt * 32 + pl0 * 32 + mrow) * Qfp + c], acc[i][jn][e]);
                }
            }
        }
}

// G[row][c] = tau_r * (csum[r] - 2 sum_l 256^l Gacc[(t*4+l)*32+rl][c])  (x = 1 - 2b);  G[row][cconst] = tau_r * csum[r];
// f[r] = tau_r * asum[r]  (= sum_k w_k exp(-E) for RISE / logRISE; RPLE keeps its FP64 sum).  r = slot, row = srow[r].
__global__ __launch_bounds__(256) void k_finalize_i8(const int32_t *__restrict__ Gacc, const double *__restrict__ tau,
                                                     const long long *__restrict__ csum,
                                                     const long long *__restrict__ asum, const int *__restrict__ srow,
                                                     const int *__restrict__ rowcol, int slot0, int64_t Qp, int64_t Qfp, int64_t Qf,
                                                     int64_t cconst, int form, int want_grad, int hv,
                                                     double *__restrict__ G, double *__restrict__ f, int nplanes,
                                                     int64_t plane_stride, const unsigned *__restrict__ mmax,
                                                     SlotResult *__restrict__ res) {
    const int r = slot0 + blockIdx.y;
    if (rowcol[r] < 0) return;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double t = tau[r];
    if (c == 0) {
        double fv = f ? f[r] : 0.0; // RPLE: the forward kernel's FP64 sum
        if (form != 2 && !hv) {
            if (want_grad) { // f = sum_k w exp(-E) = -sum_k V_k s_k = -G[r][u] (u = the node's own, masked, column)
                const int tile = r >> 5, rl = r & 31, u = rowcol[r];
                long long s = 0;
#pragma unroll
                for (int l = LB - 1; l >= 0; --l) {
                    long long a = 0;
                    for (int pl = 0; pl < nplanes; ++pl) a += (long long)Gacc[pl * plane_stride + ((int64_t)(tile * LB + l) * 32 + rl) * Qfp + u];
                    s = s * 256 + a;
                }
                fv = -t * (double)(csum[r] - 2 * s);
            } else {
                fv = t * (double)asum[r];
            }
            f[r] = fv;
        }
        if (res) res[r] = SlotResult{fv, t, mmax[r], 0u};
    }
    if (!want_grad || c >= Qp) return;
    double v = 0.0;
    if (c < Qf) {
        const int tile = r >> 5, rl = r & 31;
        long long s = 0;
#pragma unroll
        for (int l = LB - 1; l >= 0; --l) {
            long long a = 0;
            for (int pl = 0; pl < nplanes; ++pl) a += (long long)Gacc[pl * plane_stride + ((int64_t)(tile * LB + l) * 32 + rl) * Qfp + c];
            s = s * 256 + a;
        }
        v = t * (double)(csum[r] - 2 * s);
    } else if (c == cconst) {
        v = t * (double)csum[r];
    }
    G[(int64_t)srow[r] * Qp + c] = v;
}


// ------------------------------------------------------------------------------------------
// Working-set Hessian on the int8 matrix cores.
//   H_r[i][j] = sum_k h_rk x_ki x_kj,  x = +-1 = 1 - 2b  (b = 1 where x = -1)
//             = S - 2 T_ii - 2 T_jj + 4 T_ij,   T_ij = sum_k h_rk b_ki b_kj,  S = sum_k h_rk,
// computed per base-256 digit plane h_l of the (non-negative, 31-bit) weight as (mask_i & h_l) * b_j with the byte
// masks 0x00 / 0xFF expanded from bits in LDS: exact integer GEMMs.
//
// Sub-sampled Newton: the sum runs over `Kh` configurations taken as every `kstride`-th block of 512 (block cb of
// the compact index <-> samples [512 cb kstride, +512)): spread over the whole histogram, whose rows are usually
// sorted, instead of its first rows.
// ------------------------------------------------------------------------------------------
// Hessian weights of the active rows as limb planes over the compact index, in the sample order of the bit images
// (vq_pos within each 64): RISE / logRISE h = |V|; RPLE h = 2a(1 - a/(2w)), a = |V|
__global__ __launch_bounds__(256) void k_make_hw(const int8_t *__restrict__ Vq, const unsigned *__restrict__ Sb,
                                                 const double *__restrict__ w, const double *__restrict__ tau,
                                                 const int *__restrict__ rowcol /* row -> node */,
                                                 const int *__restrict__ vslot /* row -> slot of its V planes */,
                                                 const int *__restrict__ mt /* rows with mt[r] = 0 are skipped */, int64_t Kp,
                                                 int64_t Hpitch, int64_t kstride, int64_t Kh, int form, int8_t *__restrict__ Hq,
                                                 long long *__restrict__ hS, int vlbt, int vpl0, double vscale) {
    // A thread owns 4 consecutive bytes of a 64-sample row piece: the V image and the weight planes share the byte order
    // vq_pos() within a piece, and 4 consecutive positions are 4 consecutive samples, so the four limbs come in as four
    // dwords and leave as four dwords (one byte per element and limb before: 4x the memory instructions).
    const int r = blockIdx.y;
    if (mt[r] == 0) return;
    const int u = rowcol[r], vs = vslot[r];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; // (piece, dword)
    const int64_t jc = (t >> 4) * 64;                            // compact index of the piece's first sample
    const int p4 = (int)(t & 15) * 4;                            // byte position within the piece
    const int tile = r >> 5, rl = r & 31;
    long long sm = 0;
    if (jc < Kh) {
        const int64_t kc = (jc >> 9) * kstride * 512 + (jc & 511); // the configuration the piece starts at
        // samples of the positions p4 .. p4 + 3: s = s0 .. s0 + 3 (inverse of vq_pos)
        const int s0 = (((p4 >> 5) & 1) << 2) | (((p4 >> 2) & 3) << 3) | (((p4 >> 4) & 1) << 5);
        unsigned q[4] = {0u, 0u, 0u, 0u};
        unsigned sg = 0;
        if (kc < Kp) {
            const int8_t *vq = Vq + vq_off(vs, vpl0, kc, Kp, vlbt) + p4; // (vq_pos(0) = 0: the piece's first byte)
#pragma unroll
            for (int l = 0; l < 4; ++l) q[l] = *reinterpret_cast<const unsigned *>(vq + l * 32 * 64);
            const int64_t k0 = kc + s0;
            sg = (Sb[(int64_t)u * (Kp >> 5) + (k0 >> 5)] >> (k0 & 31)) & 15u; // s_u^k = 1 - 2 bit
        }
        unsigned dgw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qv = (int)(int8_t)(q[0] >> (8 * e)) + 256 * ((int)(int8_t)(q[1] >> (8 * e)) + 256 * ((int)(int8_t)(q[2] >> (8 * e)) + 256 * (int)(int8_t)(q[3] >> (8 * e))));
            int mag = ((sg >> e) & 1u) ? qv : -qv; // V = -w exp(-E) s: |V| = -q s >= 0
            if (mag < 0) mag = 0; // (cannot happen: the top four planes of a 6-plane image are V / 65536 tau rounded to nearest, same sign or 0)
            if (form == 2) {
                const double tt = tau[vs] * vscale, a = (double)mag * tt, wk = kc < Kp ? w[kc + s0 + e] : 0.0;
                mag = wk > 0 ? (int)rint(2.0 * a * (1.0 - a / (2.0 * wk)) / tt) : 0;
            }
            sm += mag;
            const unsigned dg = ((unsigned)mag + 0x80808080u) ^ 0x80808080u;
#pragma unroll
            for (int l = 0; l < 4; ++l) dgw[l] |= ((dg >> (8 * l)) & 0xffu) << (8 * e);
        }
        int8_t *hq = Hq + ((int64_t)tile * 128 + rl) * Hpitch + jc + p4;
#pragma unroll
        for (int l = 0; l < 4; ++l) *reinterpret_cast<unsigned *>(hq + (int64_t)l * 32 * Hpitch) = dgw[l];
    }
    // S = sum of the weights
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    __shared__ long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&hS[r]), (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}

// Row-major twin of Xtb for the gathered-row DMA of the Hessian kernel: Mb [Qp][Kp/64][2 h] dwords, same dword format
// and sample order as Xtb (rows at and beyond Qfp -- the constant column and the padding -- hold zero bits: x = +1).
__global__ __launch_bounds__(256) void k_build_mb(const unsigned *__restrict__ Xtb, int64_t nkk, unsigned *__restrict__ Mb) {
    const int64_t kt = (int64_t)blockIdx.y * 256 + threadIdx.x, c = blockIdx.x;
    if (kt >= nkk) return;
    const uint2 v = *reinterpret_cast<const uint2 *>(Xtb + ((((c >> 7) * nkk + kt) * 128) + (c & 127)) * 2);
    *reinterpret_cast<uint2 *>(Mb + (c * nkk + kt) * 2) = v;
}

constexpr int kHessSmall = 4; // working sets of up to this many 32-entry tiles take the 2 x 2 kernel, larger ones the 2 x 4
// Blocked kernel: a workgroup computes the tile block (rows 2a, 2a+1) x (columns BT b .. BT b + BT - 1) of one row's
// working-set matrix (needed iff BT b <= 2a + 1: lower triangle) over one chunk of the compact index.  Per group of 8
// steps (512 samples) it DMAs the 64 + 32 BT gathered rows x 64 B of bits and 4 x 512 B of weight limbs into a
// 3-stage ring.  Per step the four waves first expand the operands cooperatively into LDS, in MFMA fragment layout --
// wave w expands B tile w (0/1 bytes, both K-halves; w < BT) and A-mask fragment (i = w >> 1, t = w & 1) (0x00/0xFF
// bytes) -- then every wave runs the 2 x BT tile block for ITS weight limb l = wave:
//   acc[i][j] += (mask_i & h_l) * b_j  =  sum_k h_lk b_ik b_jk        (one barrier per step)
// BT = 2 serves working sets of up to 4 tiles (128 entries), BT = 4 the larger ones (fewer, fatter blocks).
template <int BT>
__global__ __launch_bounds__(256, 2) void k_hess_bits_blk(const unsigned *__restrict__ Mb, const int8_t *__restrict__ Hq,
                                                          const int *__restrict__ F, const int *__restrict__ mt,
                                                          const long long *__restrict__ hoff, int cap, int64_t Kh,
                                                          int64_t Kp, int64_t Hpitch, int64_t kchunk /* multiple of 512 */,
                                                          int64_t kstride, long long *__restrict__ H64, int z0, int R0,
                                                          const int *__restrict__ tF, const int *__restrict__ trow, int tT) {
    constexpr int AR = 64, BR = 32 * BT, RP = (AR + BR) / 16; // row pieces
    constexpr int STAGE = (AR + BR) * 64 + 4 * 512, NPIECE = STAGE / 1024, NSG = 3;
    constexpr int NPJ = (NPIECE + 3) / 4;                     // pieces of the waves that carry one more
    constexpr int EBUF = (4 + 2 * BT) * 1024;                 // expanded operands of one step
    extern __shared__ __attribute__((aligned(16))) int8_t lds[];
    int8_t *eb = lds + NSG * STAGE;
    const int r = blockIdx.z + z0; // block: a row's working set (r < R0), or tile r - R0 of the matrix-free rows' preconditioner
    const int m = mt[r];
    if (m == 0 || (m <= kHessSmall) != (BT == 2)) return; // one launch per size class
    // decode the block index: a = tile-row pair, b = group of BT tile columns, needed iff BT b <= 2a+1
    int a = 0, b = blockIdx.y;
    for (;;) {
        const int nb = (2 * a + 1) / BT + 1;
        if (b < nb) break;
        b -= nb;
        ++a;
        if (2 * a >= m) return;
    }
    if (2 * a >= m) return;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, h = lane >> 5;
    const int64_t kb = (int64_t)blockIdx.x * kchunk; // compact index
    if (kb >= Kh) return;
    const int64_t ke = (kb + kchunk < Kh) ? kb + kchunk : Kh;
    const int ngrp = (int)((ke - kb + 511) / 512);
    const int wr = r < R0 ? r : trow[r - R0]; // the row whose weights this block uses
    const int tile = wr >> 5, rl = wr & 31;
    const int *Fr = r < R0 ? F + (int64_t)r * cap : tF + (int64_t)(r - R0) * tT;
    const int mrows = m * 32;
    const int64_t nkk = Kp >> 6;

    // DMA sources of this wave's pieces (wave + 4 j); per group: + 64 kstride B (bits: 8 steps x 8 B of every
    // kstride-th block) resp. + 512 B (limb bytes, compact)
    const bool extra = wave < NPIECE - 4 * (NPJ - 1);
    const int8_t *src[NPJ];
    int64_t adv[NPJ];
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
        const int pc = wave + 4 * j;
        if (pc < RP) {
            const int row = pc * 16 + (lane >> 2); // A rows then B rows
            int fr = row < AR ? 2 * a * 32 + row : BT * b * 32 + (row - AR);
            if (fr >= mrows) fr = 0;
            const int slot = (lane & 3) ^ ((row >> 2) & 3); // swizzle on the source (LDS side is linear)
            src[j] = reinterpret_cast<const int8_t *>(Mb) + ((int64_t)Fr[fr] * nkk + (kb >> 9) * kstride * 8) * 8 + slot * 16;
            adv[j] = 64 * kstride;
        } else {
            const int l = 2 * (pc < NPIECE ? pc - RP : 0) + (lane >> 5);
            src[j] = Hq + ((int64_t)tile * 128 + l * 32 + rl) * Hpitch + kb + (lane & 31) * 16;
            adv[j] = 512;
        }
    }
    auto issue = [&](int g) {
        int8_t *sb = lds + (g % NSG) * STAGE;
#pragma unroll
        for (int j = 0; j < NPJ - 1; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + (int64_t)g * adv[j]), (lptr_t)(sb + (wave + 4 * j) * 1024), 16, 0, 0);
        if (extra)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[NPJ - 1] + (int64_t)g * adv[NPJ - 1]), (lptr_t)(sb + (wave + 4 * (NPJ - 1)) * 1024), 16, 0, 0);
    };
    v16i acc[2][BT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < BT; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0;

    // LDS offsets (stage-relative) of the dwords this lane expands: its row of B tile `wave` and of A tile wave >> 1
    const int rowB = AR + (wave < BT ? wave : 0) * 32 + lr, rowA = (wave >> 1) * 32 + lr;
    const int swB = (rowB >> 2) & 3, swA = (rowA >> 2) & 3;
    issue(0);
    if (ngrp > 1) issue(1);
    for (int g = 0; g < ngrp; ++g) {
        // this wave's pieces of stage g have landed (stage g + 1 may still be in flight), then every wave's
        if (g + 1 < ngrp) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPJ) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPJ - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < ngrp) issue(g + 2);
        const int8_t *st = lds + (g % NSG) * STAGE;
        const int left = (int)((ke - kb - (int64_t)g * 512 + 63) / 64);
        const int nsteps = left < 8 ? left : 8;
        for (int ks = 0; ks < nsteps; ++ks) {
            int8_t *e = eb + ((g * 8 + ks) & 1) * EBUF;
            // cooperative expansion of step ks: logical 16-byte slot ks >> 1 of the row, dword (ks & 1) * 2 + h
            {
                const unsigned vA = *reinterpret_cast<const unsigned *>(st + rowA * 64 + ((((ks >> 1) ^ swA)) << 4) + (((ks & 1) * 2 + h) << 2));
                v4i fm;
#pragma unroll
                for (int d = 0; d < 4; ++d) fm[d] = (int)(((vA >> (4 * (wave & 1) + d)) & 0x01010101u) * 0xFFu);
                *reinterpret_cast<v4i *>(e + wave * 1024 + lane * 16) = fm; // fragment (i = wave >> 1, t = wave & 1)
                if (wave < BT) {
                    const unsigned vB = *reinterpret_cast<const unsigned *>(st + rowB * 64 + ((((ks >> 1) ^ swB)) << 4) + (((ks & 1) * 2 + h) << 2));
                    v4i f0, f1;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        f0[d] = (int)((vB >> d) & 0x01010101u);
                        f1[d] = (int)((vB >> (4 + d)) & 0x01010101u);
                    }
                    *reinterpret_cast<v4i *>(e + 4096 + (wave * 2 + 0) * 1024 + lane * 16) = f0;
                    *reinterpret_cast<v4i *>(e + 4096 + (wave * 2 + 1) * 1024 + lane * 16) = f1;
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const v4i mg = *reinterpret_cast<const v4i *>(st + (AR + BR) * 64 + wave * 512 + ks * 64 + (2 * t + h) * 16);
                v4i fa[2], fb[BT];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const v4i *>(e + (i * 2 + t) * 1024 + lane * 16) & mg;
#pragma unroll
                for (int jn = 0; jn < BT; ++jn) fb[jn] = *reinterpret_cast<const v4i *>(e + 4096 + (jn * 2 + t) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < BT; ++jn) acc[i][jn] = MFMA_I8(fa[i], fb[jn], acc[i][jn]);
            }
        }
    }
    long long *Hr = H64 + hoff[r];
    const int hp = 32 * m;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < BT; ++jn) {
            const int ti = 2 * a + i, tj = BT * b + jn;
            if (ti < m && tj <= ti) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ii = ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, jj = tj * 32 + lr;
                    const long long v = ((long long)acc[i][jn][e]) * (1ll << (8 * wave));
                    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&Hr[(int64_t)ii * hp + jj]), (unsigned long long)v);
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_hess_i8_fin(const long long *__restrict__ H64, const long long *__restrict__ hS,
                                                     const double *__restrict__ tau, const int *__restrict__ vslot,
                                                     const int *__restrict__ mt,
                                                     const long long *__restrict__ hoff, double *__restrict__ H, int y0, int R0,
                                                     const int *__restrict__ trow, double vscale) {
    const int r = blockIdx.y + y0;
    const int wr = r < R0 ? r : trow[r - R0];
    const int m = mt[r] * 32;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * m) return;
    const int i = idx / m, j = idx % m;
    if ((j >> 5) > (i >> 5)) return;
    const long long *Hr = H64 + hoff[r];
    const long long T = Hr[(int64_t)i * m + j], Ti = Hr[(int64_t)i * m + i], Tj = Hr[(int64_t)j * m + j];
    H[hoff[r] + (int64_t)i * m + j] = tau[vslot[wr]] * vscale * (double)(hS[wr] - 2 * Ti - 2 * Tj + 4 * T);
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

// per-slot results of the last pass of the given kind (device pointers): tau (scale of the V / u planes), mmax
void i8_slot_results(void *p, int hv, const double **tau, const unsigned **mmax) {
    I8Ws *w = static_cast<I8Ws *>(p);
    *tau = w ? w->sc[hv ? 1 : 0].tau : nullptr;
    *mmax = w ? w->sc[hv ? 1 : 0].mmax : nullptr;
}

void i8_vq_buffer(void *p, const int8_t **vq, int64_t *bytes, const DevProblem &d) {
    I8Ws *w = static_cast<I8Ws *>(p);
    *vq = w ? w->Vq : nullptr;
    *bytes = w ? (int64_t)w->slots * w->LBT * d.Kp : 0;
}

void i8_free(void *p) {
    I8Ws *w = static_cast<I8Ws *>(p);
    if (!w) return;
    (void)hipDeviceSynchronize(); // once for all the blocks below (dev_free_synced)
    void *ptrs[] = {w->Tq, w->Vq, w->Uq, w->Gacc, w->tauovr, w->Hq, w->hS, w->H64, w->Mb};
    for (void *q : ptrs)
        if (q) (void)dev_free_synced(q);
    for (auto &sc : w->sc) {
        void *qs[] = {sc.sigma, sc.tau, sc.invtau, sc.qconst, sc.csum, sc.asum, sc.csum2, sc.asum2, sc.mmax};
        for (void *q : qs)
            if (q) (void)dev_free_synced(q);
    }
    delete w;
}

// `wide`: 1 = the workspace must hold 6-plane V images and 7 planes of Theta (objective passes of precision i8w), 0 = 4 / 5
// (i8x), -1 = whatever it holds (Hessian-vector passes: they read the V planes that are there and write Uq)
static int i8_ensure(void **wsp, const DevProblem &d, int64_t slots, int wide, hipStream_t st, std::string *err) {
    I8Ws *w = static_cast<I8Ws *>(*wsp);
    if (w && w->slots >= slots && (wide < 0 || (w->LBT == LBW) == (wide == 1))) return GML_OK;
    if (w) {
        // blocks go back to the library's cache, which hands them to the next caller without waiting: nothing of this
        // stream may still be using them
        (void)hipStreamSynchronize(st);
        i8_free(w);
    }
    *wsp = nullptr;
    w = new I8Ws();
    *wsp = w; // owned by the handle from here on: a failed allocation below is released by i8_free
    if (wide == 1) {
        w->LF = LFW;
        w->LBT = LBW;
    }
    I8CHK(dev_malloc(&w->Tq, (size_t)slots * w->LF * d.Qfp));
    I8CHK(dev_malloc(&w->Vq, (size_t)slots * w->LBT * d.Kp));
    // i32 accumulators of the backward GEMM hold |sum_k v_k b_k| <= 128 K: exact up to 2^24 configurations per set
    // (beyond 2^24: sets of <= 2^23 configurations + the slack of whole split-K chunks, see i8_pass)
    w->gplanes = d.Kp <= ((int64_t)1 << 24) ? 1 : (int)((d.Kp + ((int64_t)1 << 23) - 1) >> 23);
    I8CHK(dev_malloc(&w->Gacc, sizeof(int32_t) * (size_t)w->gplanes * slots * w->LBT * d.Qfp));
    for (auto &sc : w->sc) {
        I8CHK(dev_malloc(&sc.sigma, sizeof(double) * slots));
        I8CHK(dev_malloc(&sc.tau, sizeof(double) * slots));
        I8CHK(dev_malloc(&sc.invtau, sizeof(double) * slots));
        I8CHK(dev_malloc(&sc.qconst, sizeof(long long) * slots));
        I8CHK(dev_malloc(&sc.csum, sizeof(long long) * slots));
        I8CHK(dev_malloc(&sc.asum, sizeof(long long) * slots));
        I8CHK(dev_malloc(&sc.csum2, sizeof(long long) * slots));
        I8CHK(dev_malloc(&sc.asum2, sizeof(long long) * slots));
        I8CHK(dev_malloc(&sc.mmax, sizeof(unsigned) * slots));
    }
    I8CHK(dev_malloc(&w->tauovr, sizeof(double) * slots));
    I8CHK(hipMemsetAsync(w->Tq, 0, (size_t)slots * w->LF * d.Qfp, st));
    I8CHK(hipMemsetAsync(w->Vq, 0, (size_t)slots * w->LBT * d.Kp, st));
    w->slots = slots;
    return GML_OK;
}

// Largest number of configurations one int8 Hessian call can use (pitch of its weight planes): all of them.
int64_t i8_hess_kmax(const DevProblem &d) { return d.Kp; }

template <int BT>
static void launch_hess_blk(const I8Ws *w, const DevProblem &d, const int *dF, const int *dMt, const long long *dHoff, int R, int cap,
                            int maxm, int64_t Kh, int64_t kstride, int64_t nrows_active /* blocks of this size class */, hipStream_t st,
                            const HessTiles &tl) {
    // blocks (a, b) with BT b <= 2a + 1 for a < ceil(maxm / 2)
    int nblk = 0;
    for (int a = 0; 2 * a < maxm; ++a) nblk += (2 * a + 1) / BT + 1;
    // k-split so that the grid fills the chip (~4096 workgroups), in chunks of whole 512-sample groups
    const int maxsplit = (int)(Kh / 1024) > 0 ? (int)(Kh / 1024) : 1;
    // (counted on the rows that have a working set: late in a solve a handful of rows remain, each with all K configurations,
    // and sized on R they would get a few long workgroups each)
    const int64_t wg = (int64_t)(nrows_active > 0 ? nrows_active : 1) * nblk;
    int ns = (int)((4096 + wg - 1) / wg);
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    int64_t kc = (Kh + ns - 1) / ns;
    kc = (kc + 511) / 512 * 512;
    if (kc > ((int64_t)1 << 24)) kc = (int64_t)1 << 24; // i32 sums per workgroup: |sum| <= 128 kc
    ns = (int)((Kh + kc - 1) / kc);
    constexpr int shmem = 3 * ((64 + 32 * BT) * 64 + 4 * 512) + 2 * (4 + 2 * BT) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_hess_bits_blk<BT>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    const int64_t nv = R + tl.n;
    // (grid z is limited to 65 535; slices of 8 192 blocks so that the slicing is exercised by config 5's 28 k tiles, not only
    // by problems ten times its size)
    for (int64_t z0 = 0; z0 < nv; z0 += 8192)
        hipLaunchKernelGGL(k_hess_bits_blk<BT>, dim3((unsigned)ns, (unsigned)nblk, (unsigned)std::min<int64_t>(8192, nv - z0)), dim3(256), shmem,
                           st, w->Mb, w->Hq, dF, dMt, dHoff, cap, Kh, d.Kp, w->hKh, kc, kstride, w->H64, (int)z0, R, tl.F, tl.wrow, tl.T);
}

// Working-set Hessians of the rows 0..R-1 of the caller's arrays (mt[r] = 0: skip) from the int8 limb planes their
// last objective passes left in the slots vslot[r], over Kh configurations (a multiple of 512; block cb of the
// compact index = samples [512 cb kstride, +512)).  Returns GML_EUNSUPPORTED when a working set exceeds 512 entries
// (the solver handles larger ones matrix-free).
int i8_hessian(void *wsp, const DevProblem &d, const int *dRowcol /* row -> node */, const int *dVslot /* row -> slot */, const int *dF,
               const int *dMt, const int *hMt, const long long *dHoff, int64_t htotal, int R, int cap, int form, int64_t Kh,
               int64_t kstride, double *dH, hipStream_t st, std::string *err, const HessTiles *tiles) {
    const HessTiles tl = tiles ? *tiles : HessTiles{};
    const int *dFlag = tl.n > 0 ? tl.hflag : dMt; // rows whose weights are needed
    I8Ws *w = static_cast<I8Ws *>(wsp);
    if (!w) {
        if (err) *err = "no int8 pass has run on this handle";
        return GML_EINVAL;
    }
    int maxm = 0, maxsmall = 0;
    int64_t nsmall = 0, nlarge = 0;
    for (int r = 0; r < R; ++r) {
        maxm = hMt[r] > maxm ? hMt[r] : maxm;
        if (hMt[r] <= kHessSmall) maxsmall = hMt[r] > maxsmall ? hMt[r] : maxsmall;
        if (hMt[r] > kHessSmall) ++nlarge;
        else if (hMt[r] > 0) ++nsmall;
    }
    if (maxm > 16) return GML_EUNSUPPORTED;
    const bool tiles_small = tl.T / 32 <= kHessSmall;
    if (tl.n > 0) { // the tiles: tl.n blocks of T / 32 <= 4 tiles each
        maxm = std::max(maxm, tl.T / 32);
        if (tiles_small) {
            maxsmall = std::max(maxsmall, tl.T / 32);
            nsmall += tl.n;
        } else {
            nlarge += tl.n;
        }
    }
    const int64_t pitch = d.Kp, Rp = (R + 31) / 32 * 32;
    if (Kh > pitch) Kh = pitch;
    if (w->hKh != pitch || w->hrows < Rp) {
        if (w->Hq) (void)dev_free(w->Hq);
        if (w->hS) (void)dev_free(w->hS);
        w->Hq = nullptr;
        w->hS = nullptr;
        I8CHK(dev_malloc(&w->Hq, (size_t)Rp * LB * pitch));
        I8CHK(dev_malloc(&w->hS, sizeof(long long) * Rp));
        w->hKh = pitch;
        w->hrows = Rp;
    }
    if (!w->Mb) {
        I8CHK(dev_malloc(&w->Mb, (size_t)d.Qp * (d.Kp / 8)));
        I8CHK(hipMemsetAsync(w->Mb, 0, (size_t)d.Qp * (d.Kp / 8), st));
        hipLaunchKernelGGL(k_build_mb, dim3((unsigned)d.Qfp, (unsigned)((d.Kp / 64 + 255) / 256)), dim3(256), 0, st, d.Xtb, d.Kp / 64, w->Mb);
    }
    const int64_t need = htotal;
    if (need > w->hcap_elems) {
        if (w->H64) (void)dev_free(w->H64);
        w->H64 = nullptr;
        I8CHK(dev_malloc(&w->H64, sizeof(long long) * need));
        w->hcap_elems = need;
    }
    I8CHK(hipMemsetAsync(w->H64, 0, sizeof(long long) * need, st));
    I8CHK(hipMemsetAsync(w->hS, 0, sizeof(long long) * Rp, st));
    hipLaunchKernelGGL(k_make_hw, dim3((unsigned)((Kh / 4 + 255) / 256), (unsigned)R), dim3(256), 0, st, w->Vq, d.Sb, d.w, w->sc[0].tau, dRowcol,
                       dVslot, dFlag, d.Kp, pitch, kstride, Kh, form, w->Hq, w->hS, w->LBT, w->vpl0(), w->vscale());
    HessTiles rows_only = tl; // (the tiles are all of one size class: the other launch covers the rows' own blocks only)
    rows_only.n = 0;
    if (maxsmall > 0) launch_hess_blk<2>(w, d, dF, dMt, dHoff, R, cap, maxsmall, Kh, kstride, nsmall, st, tiles_small ? tl : rows_only);
    if (maxm > kHessSmall) launch_hess_blk<4>(w, d, dF, dMt, dHoff, R, cap, maxm, Kh, kstride, nlarge, st, tiles_small ? rows_only : tl);
    hipLaunchKernelGGL(k_hess_i8_fin, dim3((unsigned)((maxm * 32 * maxm * 32 + 255) / 256), (unsigned)R), dim3(256), 0, st, w->H64, w->hS,
                       w->sc[0].tau, dVslot, dMt, dHoff, dH, 0, R, tl.wrow, w->vscale());
    const int tm = tl.T / 32;
    for (int64_t y0 = 0; y0 < tl.n; y0 += 8192)
        hipLaunchKernelGGL(k_hess_i8_fin, dim3((unsigned)((tm * 32 * tm * 32 + 255) / 256), (unsigned)std::min<int64_t>(8192, tl.n - y0)), dim3(256),
                           0, st, w->H64, w->hS, w->sc[0].tau, dVslot, dMt, dHoff, dH, (int)(R + y0), R, tl.wrow, w->vscale());
    I8CHK(hipGetLastError());
    return GML_OK;
}

// one launch zeroes every accumulator of a pass over the slots [slot0, slot0 + ns): slot sums, maxima, f, and the i32
// gradient planes of those slots' tiles
__global__ __launch_bounds__(256) void k_zero_pass(long long *__restrict__ csum, long long *__restrict__ asum,
                                                   long long *__restrict__ csum2, long long *__restrict__ asum2,
                                                   unsigned *__restrict__ mmax, double *__restrict__ f, int slot0, int ns,
                                                   v4i *__restrict__ gacc, int64_t ngacc, int nplanes, int64_t plane_stride4) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (i0 < ns) {
        csum[slot0 + i0] = 0;
        asum[slot0 + i0] = 0;
        csum2[slot0 + i0] = 0;
        asum2[slot0 + i0] = 0;
        mmax[slot0 + i0] = 0;
        if (f) f[slot0 + i0] = 0.0;
    }
    const v4i z = {0, 0, 0, 0};
    for (int pl = 0; pl < nplanes; ++pl)
        for (int64_t i = i0; i < ngacc; i += stride) gacc[pl * plane_stride4 + i] = z;
}

struct FwdLaunch {
    int chunk_tiles, part_tiles, ntk; // sample tiles: per backward chunk, of them taking part, compact count
    const I8Ws *w;
    const DevProblem *d;
    const SlotScalars *sc;
    const int *rowcol, *groups, *vmap;
    int ngroups;
    double *F;
    int8_t *Vout;
    hipStream_t st;
};

template <int LF, int FORM, bool WANTF, bool WIDE, bool UNIW>
static void launch_fwd4(const FwdLaunch &a) {
    constexpr int STAGE = 2 * (2 + 2 * LF) * 1024; // two 64-column steps per ring stage, three stages
    constexpr int shmem = 3 * STAGE + 512 + 1024;   // ring + exp, log tables
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd_i8<LF, FORM, WANTF, WIDE, UNIW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, shmem); // per device: set on every launch
    const DevProblem &d = *a.d;
    const int ntk = a.ntk;
    const int grid = ((ntk + 7) / 8) * 8 * a.ngroups; // one workgroup per (sample tile, node tile); see the kernel's block mapping
    hipLaunchKernelGGL((k_fwd_i8<LF, FORM, WANTF, WIDE, UNIW>), dim3(grid), dim3(256), shmem, a.st, d.Xb, d.Sb, a.w->Tq, a.rowcol, a.groups,
                       a.ngroups, d.w, a.sc->sigma, a.sc->qconst, a.sc->invtau, d.Kp, ntk, (int)(d.Qfp / 64), d.wuni, d.K, a.Vout,
                       a.sc->csum, a.sc->asum, a.F, a.sc->mmax, a.w->Vq, a.vmap, a.w->sc[0].tau, a.w->LBT, a.w->vpl0(), a.w->vscale(),
                       a.chunk_tiles, a.part_tiles);
}

template <int LF, int FORM, bool WANTF, bool WIDE>
static void launch_fwd3(const FwdLaunch &a) {
    if (a.d->wuni > 0.0) launch_fwd4<LF, FORM, WANTF, WIDE, true>(a);
    else launch_fwd4<LF, FORM, WANTF, WIDE, false>(a);
}

template <int LF, int FORM, bool WANTF>
static void launch_fwd2(const FwdLaunch &a) {
    if (a.d->Qfp > 32768) launch_fwd3<LF, FORM, WANTF, true>(a);
    else launch_fwd3<LF, FORM, WANTF, false>(a);
}

template <int LF>
static void launch_fwd(const FwdLaunch &a, int form, bool wantf, int hv) {
    if (hv) {
        if (form == 2) launch_fwd2<LF, 4, false>(a);
        else launch_fwd2<LF, 3, false>(a);
    } else if (form == 2) launch_fwd2<LF, 2, true>(a);
    else if (wantf) launch_fwd2<LF, 0, true>(a);
    else launch_fwd2<LF, 0, false>(a);
}

// Split-K plan of the backward GEMM for `ngroups` node tiles: nsplit chunks of kchunk configurations each, of which the first
// kpart take part (ksub > 1: a sub-sampled Hessian-vector pass over ~1/ksub of the configurations, spread over the whole
// histogram chunk by chunk; kpart is then a multiple of 512, the granularity of gml_problem's block weights).
void i8_split_plan(const DevProblem &d, int ngroups, int ksub, int64_t *kchunk_out, int64_t *kpart_out, int *nsplit_out) {
    const int nNt = (int)((d.Qfp + 255) / 256);
    const int T = ngroups * nNt;
    const int gplanes = d.Kp <= ((int64_t)1 << 24) ? 1 : (int)((d.Kp + ((int64_t)1 << 23) - 1) >> 23);
    // a multiple of 8 chunks (one XCD each).  24 chunks, or -- with few node tiles (node-sharded ranks, late solver
    // iterations) -- as many as it takes to give each of the 512 resident workgroup slots one workgroup.
    // Measured at the headline problem (backward ms at 16 / 24 / 32 / 64 chunks): 128 nodes 0.52 / 0.47 / 0.39 / 0.42,
    // 256 nodes 0.75 / 0.73 / 0.77 / 0.75, 512 nodes 1.57 / 1.48 / 1.49 / 1.50, 1024 nodes 2.98 whatever the count.
    int nsplit = (int)(((512 + T - 1) / T + 7) / 8 * 8);
    if (nsplit < 24) nsplit = 24;
    if (nsplit > 256) nsplit = 256;
    int64_t kchunk = (d.Kp + nsplit - 1) / nsplit;
    if (ksub < 1) ksub = 1;
    const int64_t gran = ksub > 1 ? 512 * (int64_t)ksub : 256; // (whole 256-sample forward tiles either way)
    kchunk = (kchunk + gran - 1) / gran * gran;
    if (kchunk < 2048) kchunk = (2048 + gran - 1) / gran * gran;
    if (gplanes > 1 && kchunk > ((int64_t)1 << 22)) kchunk = ((int64_t)1 << 22) / gran * gran;
    *nsplit_out = (int)((d.Kp + kchunk - 1) / kchunk);
    *kchunk_out = kchunk;
    *kpart_out = kchunk / ksub;
}

// One pass of the int8-limb operator over the slots the caller lists (I8Pass, gml_dev.h).
int i8_pass(void **wsp, const DevProblem &d, int64_t slot_capacity, const I8Pass &a, hipStream_t st, hipEvent_t *ev, std::string *err) {
    const int hv = a.hv ? (a.hv == 2 ? 2 : 1) : 0; // 2: Hessian-vector products in 2 backward limbs
    const bool wide = a.wide && !hv;                // (Hessian-vector passes are 31-bit passes whatever the workspace holds)
    int rc = i8_ensure(wsp, d, slot_capacity, hv ? -1 : (wide ? 1 : 0), st, err);
    if (rc) return rc;
    I8Ws *w = static_cast<I8Ws *>(*wsp);
    int LF = wide ? LFW : (a.lf ? a.lf : 5);
    if (LF > w->LF) LF = w->LF;
    if (a.ngroups + 1 > 65536 || a.slot1 > w->slots || a.slot0 % 32 || a.slot1 % 32) {
        if (err) *err = "bad slot range";
        return GML_EINVAL;
    }
    if (hv && !w->Uq) {
        I8CHK(dev_malloc(&w->Uq, (size_t)w->slots * LB * d.Kp));
        I8CHK(hipMemsetAsync(w->Uq, 0, (size_t)w->slots * LB * d.Kp, st));
    }
    const SlotScalars &sc = w->sc[hv ? 1 : 0];
    const int ns = a.slot1 - a.slot0;
    const bool grad = a.want_grad || hv;
    const int lbg = wide ? LBW : LB; // limb planes of this pass's V and of its gradient accumulators
    int32_t *gacc0 = w->Gacc + (int64_t)a.slot0 * lbg * d.Qfp;
    const int64_t gplane_stride = (int64_t)w->slots * lbg * d.Qfp;
    hipLaunchKernelGGL(k_zero_pass, dim3(1024), dim3(256), 0, st, sc.csum, sc.asum, sc.csum2, sc.asum2, sc.mmax, a.F, a.slot0, ns,
                       reinterpret_cast<v4i *>(gacc0), grad ? (int64_t)ns * lbg * d.Qfp / 4 : 0, w->gplanes, gplane_stride / 4);
#define QUANT(LFV)                                                                                                                    \
    hipLaunchKernelGGL((k_quant_theta<LFV>), dim3(ns), dim3(256), 0, st, a.theta, a.srow, a.rowcol, a.slot0, d.Qp, d.Qfp, d.cconst,   \
                       d.wmax, a.form, hv, a.vmap, w->sc[0].tau, w->Tq, sc.sigma, sc.tau, sc.invtau, sc.qconst, a.tauovr,             \
                       wide ? kVdiv6 : kVdiv4, w->vscale())
    if (LF < 3 && !hv) LF = 3; // 2 limbs exist for the directions of Hessian-vector passes only
    switch (LF) {
    case 2: QUANT(2); break;
    case 3: QUANT(3); break;
    case 4: QUANT(4); break;
    case 7: QUANT(7); break;
    default: QUANT(5);
    }
#undef QUANT
    // split-K plan of the backward GEMM (made here: a sub-sampled pass runs its forward kernel over the same parts)
    const int nNt = (int)((d.Qfp + 255) / 256);
    constexpr int TM = 1; // node tiles per backward workgroup (the 8-wave form with two, TM = 2, measured slower)
    const int ngt = (a.ngroups + TM - 1) / TM;
    const int T = ngt * nNt;
    int64_t kchunk = 0, kpart = 0;
    int nsplit = 0;
    if (hv && a.kchunk > 0) {
        kchunk = a.kchunk;
        kpart = a.kpart > 0 ? a.kpart : a.kchunk;
        nsplit = (int)((d.Kp + kchunk - 1) / kchunk);
    } else {
        i8_split_plan(d, a.ngroups, hv ? a.ksub : 1, &kchunk, &kpart, &nsplit);
    }
    const int ksub = kpart < kchunk ? (int)(kchunk / kpart) : 1;
    if (ev) I8CHK(hipEventRecord(ev[0], st));
    if (wide) {
        FwdWArgs fw{&d, w->Tq, &sc, a.rowcol, a.groups, a.ngroups, a.form, !a.want_grad, a.F, w->Vq, st};
        launch_fwd_i8w(fw);
    } else {
        FwdLaunch fl{(int)(kchunk / 256), (int)(kpart / 256), 0, w, &d, &sc, a.rowcol, a.groups, a.vmap, a.ngroups, a.F, hv ? w->Uq : w->Vq, st};
        fl.ntk = ksub > 1 ? nsplit * fl.part_tiles : (int)(d.Kp / 256);
        if (ksub == 1) fl.chunk_tiles = fl.part_tiles = 1; // (every tile: no remapping)
        if (!hv && w->LBT != LB) {
            if (err) *err = "a 31-bit objective pass on a workspace of 6-plane V images";
            return GML_EINVAL;
        }
        switch (LF) {
        // with the gradient requested, f comes out of the backward GEMM for free (column u of row u)
        case 2: // (Hessian-vector forms only)
            if (a.form == 2) launch_fwd2<2, 4, false>(fl);
            else launch_fwd2<2, 3, false>(fl);
            break;
        case 3: launch_fwd<3>(fl, a.form, !a.want_grad, hv); break;
        case 4: launch_fwd<4>(fl, a.form, !a.want_grad, hv); break;
        default: launch_fwd<5>(fl, a.form, !a.want_grad, hv);
        }
    }
    if (ev) I8CHK(hipEventRecord(ev[1], st));
    if (grad) {
        // chunks per set of i32 accumulators: one set up to 2^24 configurations; beyond, gplanes = ceil(Kp / 2^23) sets of
        // cpp chunks each: cpp * kchunk < (Kp + kchunk) / gplanes + kchunk <= 2^23 + 1.5 * 2^22 < 2^24, so |sum| < 2^31
        const int cpp = (nsplit + w->gplanes - 1) / w->gplanes;
        const int grid = ((nsplit + 7) / 8) * 8 * T;
        const int shmem = 4 * (8 * TM + 2) * 1024;
        const int8_t *Vin = hv ? w->Uq : w->Vq;
        if (wide) { // the two halves of the 6 planes: two launches of the 3-plane form
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd_i8<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
            for (int half = 0; half < 2; ++half)
                hipLaunchKernelGGL((k_bwd_i8<1, 3>), dim3(grid), dim3(256), shmem, st, Vin, d.Xtb, a.groups, ngt, nNt, d.Qfp, d.Kp, kchunk,
                                   nsplit, w->Gacc, cpp, gplane_stride, kpart, LBW, 3 * half);
        } else if (hv == 2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd_i8<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
            hipLaunchKernelGGL((k_bwd_i8<1, 2>), dim3(grid), dim3(256), shmem, st, Vin, d.Xtb, a.groups, ngt, nNt, d.Qfp, d.Kp, kchunk,
                               nsplit, w->Gacc, cpp, gplane_stride, kpart, LB, 0);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd_i8<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
            hipLaunchKernelGGL((k_bwd_i8<1, 4>), dim3(grid), dim3(256), shmem, st, Vin, d.Xtb, a.groups, ngt, nNt, d.Qfp, d.Kp, kchunk,
                               nsplit, w->Gacc, cpp, gplane_stride, kpart, LB, 0);
        }
    }
    if (ev) I8CHK(hipEventRecord(ev[2], st));
    if (wide)
        launch_finalize_i8w(w->Gacc, sc, a.srow, a.rowcol, a.slot0, ns, d.Qp, d.Qfp, d.Qf, d.cconst, a.form, grad ? 1 : 0, a.G, a.F, w->gplanes,
                            gplane_stride, a.res, st);
    else
        hipLaunchKernelGGL(k_finalize_i8, dim3((unsigned)((d.Qp + 255) / 256), (unsigned)ns), dim3(256), 0, st, w->Gacc, sc.tau, sc.csum, sc.asum,
                           a.srow, a.rowcol, a.slot0, d.Qp, d.Qfp, d.Qf, d.cconst, a.form, grad ? 1 : 0, hv, a.G, a.F, w->gplanes, gplane_stride,
                           sc.mmax, a.res);
    I8CHK(hipGetLastError());
    return GML_OK;
}

} // namespace gml
