// Int8-limb path, part 1: the bit images of the samples and the quantisation of the parameter rows into limb planes
// (overview: gml_i8.h).
#include "gml_i8.h"
#include <algorithm>
#include <string>
#include <type_traits>

namespace gml {

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
// Column compaction of the forward GEMM.  The iterates of an l1-regularised solve are sparse -- a node's row holds a few dozen
// non-zeros among thousands of columns -- and a column on which all 32 rows of a node tile are zero contributes exactly nothing to
// the tile's integer sums.  k_col_union lists, per listed tile, the columns with a non-zero in one of the tile's active rows
// (ascending; one workgroup per tile: flags by column, ordered append through wave ballots); k_build_xc builds the forward bit
// image of those columns alone (k_build_xb through the list).  The quantisation kernel then writes the digits of those columns
// into the first cnk[tile] steps of the tile's Tq image and the forward kernel sweeps cnk[tile] steps instead of Qfp / 64.
// Tiles whose list exceeds the capacity (a quarter of the columns, at most 32 steps) keep the sweep over all columns: cnk = -1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_col_union(const double *__restrict__ Theta, const int *__restrict__ srow, const int *__restrict__ rowcol,
                                                   const int *__restrict__ groups, int64_t Qp, int64_t Qfp, int csteps, int *__restrict__ cnk,
                                                   int *__restrict__ cmap) {
    const int tile = groups[blockIdx.x];
    if (tile < 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ const double *rows[32];
    __shared__ int wcnt[4];
    __shared__ int base;
    if (tid < 32) {
        const int slot = tile * 32 + tid;
        rows[tid] = rowcol[slot] >= 0 ? Theta + (int64_t)srow[slot] * Qp : nullptr;
    }
    if (tid == 0) base = 0;
    __syncthreads();
    const int cap = csteps * 64;
    int *cm = cmap + (int64_t)tile * cap;
    for (int64_t c0 = 0; c0 < Qfp; c0 += 256) {
        const int64_t c = c0 + tid;
        bool nz = false;
        if (c < Qfp) {
#pragma unroll 8
            for (int rl = 0; rl < 32; ++rl) {
                const double *th = rows[rl];
                if (th) nz |= th[c] != 0.0;
            }
        }
        const unsigned long long m = __ballot(nz);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int v = 0; v < wave; ++v) off += wcnt[v];
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (nz && pos < cap) cm[pos] = (int)c;
        __syncthreads();
        if (tid == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    const int cnt = base;
    if (cnt > cap) {
        if (tid == 0) cnk[tile] = -1;
        return;
    }
    const int nkt = (cnt + 63) >> 6;
    for (int j = cnt + tid; j < nkt * 64; j += 256) cm[j] = -1;
    if (tid == 0) cnk[tile] = nkt;
}

// forward bit image of one tile's compact columns: k_build_xb with the column taken from the list (-1: a zero column)
__global__ __launch_bounds__(256) void k_build_xc(const unsigned *__restrict__ Sb, int64_t wpr, const int32_t *__restrict__ keys, int ko,
                                                  int64_t Qf, const int *__restrict__ groups, const int *__restrict__ cnk,
                                                  const int *__restrict__ cmap, int csteps, int64_t xc_tile, unsigned *__restrict__ Xc) {
    const int tile = groups[blockIdx.z];
    if (tile < 0) return;
    const int nk = cnk[tile], kt = blockIdx.y;
    if (kt >= nk) return; // (also the tiles that run on all columns: nk = -1)
    const int64_t w = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
    const int h = threadIdx.x & 1;
    if (w >= wpr) return;
    const int *cm = cmap + (int64_t)tile * csteps * 64 + 64 * kt;
    unsigned a[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int c = cm[xb_col(j, h)];
        a[j] = c >= 0 ? stat_word(Sb, wpr, keys, ko, Qf, c, w) : 0u;
    }
    transpose32(a);
    unsigned *img = Xc + (int64_t)tile * (xc_tile / 4);
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int64_t k = w * 32 + s;
        img[((((k >> 7) * nk + kt) * 128) + (k & 127)) * 2 + h] = a[s];
    }
}

void launch_col_compact(const I8Pass &a, const DevProblem &d, I8Ws *w, hipStream_t st) {
    hipLaunchKernelGGL(k_col_union, dim3((unsigned)a.ngroups), dim3(256), 0, st, a.theta, a.srow, a.rowcol, a.groups, d.Qp, d.Qfp, w->csteps, w->cnk,
                       w->cmap);
    const int64_t wpr = d.Kp / 32;
    hipLaunchKernelGGL(k_build_xc, dim3((unsigned)((wpr + 127) / 128), (unsigned)w->csteps, (unsigned)a.ngroups), dim3(256), 0, st, d.Sb, wpr, d.keys,
                       d.ko, d.Qf, a.groups, w->cnk, w->cmap, w->csteps, w->xc_tile, reinterpret_cast<unsigned *>(w->Xc));
}

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
                                                     long long *__restrict__ qconst, long long *__restrict__ qconst2, const double *__restrict__ tauovr,
                                                     const double *__restrict__ tauovr_lnrow,
                                                     double vdiv /* largest |V| / tau the planes of this pass hold */,
                                                     double vsrc_scale /* hv: unit of the V planes read, in multiples of tauV */,
                                                     const int *__restrict__ cnk, const int *__restrict__ cmap, int cstride) {
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
    // sum_c |q_c| as an INTEGER (< 2^61 by the choice of sigma): the bound it feeds must not depend on the order of the additions --
    // a compacted pass (below) visits other columns per thread than a pass over all of them, and both must give the same tau
    long long sabs = 0;
    long long ssum = 0; // sum_c q_c: the energy of the all-(+1) configuration (the forward GEMM runs on b = [x = -1])
    long long shi = 0;  // (7 planes) the same sum for the number the top four planes alone spell: q_hi = q / 2^24 rounded to nearest
    // compacted tile: the digits of the tile's non-zero columns (cmap: ascending, -1 padded) go to the first cnk[tile] steps of the
    // tile's image; every other column of this row is zero and contributes nothing to any sum
    const int ck = cnk ? cnk[tile] : -1;
    const int64_t ncol = ck >= 0 ? (int64_t)ck * 64 : Qfp;
    const int *cm = ck >= 0 ? cmap + (int64_t)tile * cstride : nullptr;
    for (int64_t j = tid; j < ncol; j += 256) {
        const int64_t c = cm ? cm[j] : j;
        long long q = c >= 0 ? (long long)rint(th[c] * isg) : 0;
        sabs += q < 0 ? -q : q;
        ssum += q;
        int8_t *img = Tq + ((((int64_t)tile * nk + (j >> 6)) * LF) * 32 + rl) * 64 + (j & 63);
#pragma unroll
        for (int l = 0; l < LF; ++l) {
            if (LF > 5 && l == 3) shi += q; // what is left after three balanced digits
            const long long dgt = ((q + 128) & 255) - 128;
            q = (q - dgt) >> 8;
            img[l * 32 * 64] = (int8_t)dgt;
        }
    }
    long long q0 = 0, q0hi = 0;
    if (tid == 0) {
        q0 = (long long)rint(th[cconst] * isg);
        sabs += q0 < 0 ? -q0 : q0;
        if (LF > 5) {
            long long q = q0;
            for (int l = 0; l < 3; ++l) q = (q - (((q + 128) & 255) - 128)) >> 8;
            q0hi = q;
        }
    }
    __shared__ long long redl[256];
    __shared__ long long redh[256];
    __shared__ long long reda[256];
    reda[tid] = sabs;
    redl[tid] = ssum;
    redh[tid] = shi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            reda[tid] += reda[tid + s];
            redl[tid] += redl[tid + s];
            if (LF > 5) redh[tid] += redh[tid + s];
        }
        __syncthreads();
    }
    q0 += redl[0];
    q0hi += redh[0];
    if (tid == 0) {
        // |E| <= sigma * sum|q|  (|X| <= 1)
        const double emax = (double)reda[0] * sg;
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
            if (tauovr && tauovr[r] > 0.0) { // a tighter rigorous scale from the caller
                const double to = tauovr_lnrow ? tauovr[r] * exp(tauovr_lnrow[srow[r]]) : tauovr[r];
                if (to < t) t = to;
            }
            it = 1.0 / t;
        }
        sigma[r] = sg;
        qconst[r] = q0;
        if (LF > 5) qconst2[r] = q0hi;
        tau[r] = t;
        invtau[r] = it;
    }
}


void launch_quant_theta(int LF, int ns, const I8Pass &a, const DevProblem &d, int hv, const double *tauV, int8_t *Tq, const SlotScalars &sc,
                        double vdiv, double vsrc_scale, hipStream_t st, const ColCompact *cc) {
    const int *cnk = cc ? cc->cnk : nullptr, *cmap = cc ? cc->cmap : nullptr;
    const int cstride = cc ? cc->cstride : 0;
#define QUANT(LFV)                                                                                                                    \
    hipLaunchKernelGGL((k_quant_theta<LFV>), dim3(ns), dim3(256), 0, st, a.theta, a.srow, a.rowcol, a.slot0, d.Qp, d.Qfp, d.cconst,   \
                       d.wmax, a.form, hv, a.vmap, tauV, Tq, sc.sigma, sc.tau, sc.invtau, sc.qconst, sc.qconst2, a.tauovr, a.tauovr_lnrow, vdiv, vsrc_scale, \
                       cnk, cmap, cstride)
    switch (LF) {
    case 2: QUANT(2); break;
    case 3: QUANT(3); break;
    case 4: QUANT(4); break;
    case 7: QUANT(7); break;
    default: QUANT(5);
    }
#undef QUANT
}

} // namespace gml
