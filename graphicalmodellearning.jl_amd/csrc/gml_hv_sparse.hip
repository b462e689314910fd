// Hessian-vector products of a FEW matrix-free rows, over their working sets only (overview of the int8-limb path: gml_i8.h).
//
// The GEMM form of a Hessian-vector pass (i8_pass, hv) multiplies a whole 32-row node tile against every statistics column:
// right for hundreds of live rows, wasteful for the last ones of a solve (config 5 at the reference's default regulariser: 555 of
// 809 passes carry one tile with 1 - 8 live rows in it, a working set of 5 % of the 131 k columns each).  For those this file
// computes the same integers on the vector ALUs, entry by entry:
//
//   quantise   p -> q_c = rint(p_c / sigma), sigma = 2^(ex - 14)                    (k_quant_theta<2>, hv = 2)
//   forward    A_k = sum_{c in W} q_c b_kc,  E_k = sigma (q0 - 2 A_k),  u_k = round(|V_k| E_k / pn + dither_k)   (k_fwd_i8 HV epilogue)
//   backward   S_c = sum_k u_k b_kc,  (H p)_c = t (sum_k u_k - 2 S_c)              (k_bwd_i8<1,2>, k_finalize_i8)
//
// with b_kc the sign bit of statistic c in configuration k, taken straight from the spin-major sign rows Sb (a statistic's bits are
// the XOR of the rows of its spins: no operand image is gathered), the same scales, the same dither sequence and the same
// sub-sample of the configurations.  Every sum is an exact integer, so the result is BIT-IDENTICAL to the GEMM pass whatever the
// order of the additions -- which path a row takes (it depends on how many rows are live on this GPU) cannot change its iterates.
// Cost: ~3 VALU operations per (configuration, entry) and direction; measured 2.4 ms per row of 4 600 entries over 1e6
// configurations against 14 ms for a tile of the GEMM pass, i.e. an entry costs about five columns of the tile: the solver takes
// this path when the live rows' entries sum to less than 0.3 x (columns x node tiles)  (gml_solver.cpp, newton_cg_group).
#include "gml_i8.h"
#include <algorithm>
#include <string>

namespace gml {

namespace {

struct HvsRow {
    double sg2, sgq0, it, t;
    long long csum;
    long long pad;
};

__device__ __forceinline__ int wave_sum_i(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per listed row: the scales of k_quant_theta<2> (hv = 2), the digits q of the row's working-set entries, the spins
// of their statistics; zeroes the row's accumulators.
__global__ __launch_bounds__(256) void k_hvs_quant(const int *__restrict__ rows, const int *__restrict__ vslot, const long long *__restrict__ t0,
                                                   const int *__restrict__ nw, const int *__restrict__ FV, int T, const double *__restrict__ P,
                                                   int64_t Qp, int64_t Qf, int64_t cconst, const int32_t *__restrict__ keys, int ko,
                                                   const double *__restrict__ tauV, double vscale, int64_t wcap, int *__restrict__ qW,
                                                   int2 *__restrict__ spins, long long *__restrict__ S, HvsRow *__restrict__ hs) {
    const int i = blockIdx.x, r = rows[i], m = nw[r], tid = threadIdx.x;
    const int *list = FV + t0[r] * T;
    const double *p = P + (int64_t)r * Qp;
    __shared__ double red[256];
    __shared__ long long redl[256];
    double mx = 0.0;
    for (int a = tid; a < m; a += 256) mx = fmax(mx, fabs(p[list[a]]));
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    int ex = 0;
    if (mx > 0) (void)frexp(mx, &ex);
    const int sx = ex - 14; // 2 limbs: |q| <= 2^14
    const double sg = ldexp(1.0, sx), isg = ldexp(1.0, -sx);
    double sabs = 0.0;
    long long ssum = 0, qc = 0;
    for (int a = tid; a < m; a += 256) {
        const int c = list[a];
        const long long q = (long long)rint(p[c] * isg);
        sabs += fabs((double)q);
        int2 sp = {-1, -1};
        if (c == cconst) {
            qc = q; // the constant statistic: no bits, it only enters q0
        } else if (c < Qf) {
            ssum += q;
            sp.x = keys[(int64_t)c * ko];
            sp.y = ko > 1 ? keys[(int64_t)c * ko + 1] : -1;
            if (sp.x < 0) { sp.x = sp.y; sp.y = -1; }
        }
        qW[(int64_t)i * wcap + a] = c < Qf ? (int)q : 0;
        spins[(int64_t)i * wcap + a] = sp;
        S[(int64_t)i * wcap + a] = 0;
    }
    red[tid] = sabs;
    redl[tid] = ssum + qc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[tid] += red[tid + s];
            redl[tid] += redl[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double emax = red[0] * sg;
        const double pn = (emax > 0.0 ? emax : 1.0) * (65536.0 * 1.01);
        HvsRow h;
        h.sg2 = -2.0 * sg;
        h.sgq0 = sg * (double)redl[0];
        h.it = 1.0 / pn;
        h.t = tauV[vslot[i]] * vscale * pn;
        h.csum = 0;
        h.pad = 0;
        hs[i] = h;
    }
}

// The three kernels below share one decomposition: thread <-> one natural-order word of 32 configurations (compact index cd over the
// sub-sample), workgroup <-> 8 192 configurations x one chunk of EC working-set entries of one row.  Splitting the entries over
// workgroups is what fills the GPU: a sub-sampled pass of one row has only 15 words-blocks, and a thread that walked all 4 600
// entries of its row alone would take longer than the GEMM pass it replaces.
constexpr int EC = 256; // entries per workgroup

struct WordIdx {
    bool valid;
    int64_t w, cd;
};
__device__ __forceinline__ WordIdx word_index(int ntk, int chunk_tiles, int part_tiles, int64_t Kp) {
    WordIdx o;
    o.cd = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ct = (int)(o.cd >> 3);
    int st = ct;
    if (chunk_tiles != part_tiles) st = (ct / part_tiles) * chunk_tiles + ct % part_tiles;
    o.valid = ct < ntk && (int64_t)st * 256 < Kp;
    o.w = o.valid ? (int64_t)st * 8 + (o.cd & 7) : 0;
    return o;
}

// forward: A[j][cd] += sum over the chunk's entries of q_c b_kc
__global__ __launch_bounds__(256) void k_hvs_fwd(const int *__restrict__ rows, const int *__restrict__ nw, int64_t wcap, const int *__restrict__ qW,
                                                 const int2 *__restrict__ spins, const unsigned *__restrict__ Sb, int64_t Kp, int *__restrict__ Abuf,
                                                 int64_t ncdp, int ntk, int chunk_tiles, int part_tiles) {
    const int i = blockIdx.z, r = rows[i], m = nw[r], tid = threadIdx.x, base = blockIdx.y * EC;
    if (base >= m) return;
    const int cnt = min(EC, m - base);
    const int64_t wpr = Kp >> 5;
    const WordIdx wi_ = word_index(ntk, chunk_tiles, part_tiles, Kp);
    const int64_t w = wi_.w;
    __shared__ int sq[EC];
    __shared__ int2 ss[EC];
    if (tid < cnt) {
        sq[tid] = qW[(int64_t)i * wcap + base + tid];
        ss[tid] = spins[(int64_t)i * wcap + base + tid];
    }
    __syncthreads();
    int acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0;
    // four entries at a time: their (L2-resident) sign words are requested together, ahead of the arithmetic
    for (int a0 = 0; a0 < cnt; a0 += 4) {
        unsigned x[4];
        int q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + u < cnt ? a0 + u : cnt - 1;
            const int2 s = ss[a];
            q[u] = a0 + u < cnt && s.x >= 0 ? sq[a] : 0; // (the constant statistic has no bits)
            x[u] = s.x >= 0 ? Sb[(int64_t)s.x * wpr + w] : 0u;
            if (s.y >= 0) x[u] ^= Sb[(int64_t)s.y * wpr + w];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] = __mul24((int)__builtin_amdgcn_ubfe(x[u], j, 1), q[u]) + acc[j];
    }
    if (!wi_.valid) return;
    int *A = Abuf + (int64_t)i * 32 * ncdp + wi_.cd;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (acc[j] != 0) atomicAdd(A + (int64_t)j * ncdp, acc[j]);
}

// u_k of every word's 32 configurations: the arithmetic of the GEMM pass's epilogue (k_fwd_i8, HV), configuration 32 w + j
__global__ __launch_bounds__(256) void k_hvs_mid(const int *__restrict__ node, const int *__restrict__ vslot, int64_t Kp, const int8_t *__restrict__ Vq,
                                                 int lbt, int pl0, HvsRow *__restrict__ hs, int *__restrict__ Abuf /* in: A, out: u */, int64_t ncdp,
                                                 int ntk, int chunk_tiles, int part_tiles) {
    const int i = blockIdx.z, tid = threadIdx.x;
    const WordIdx wi_ = word_index(ntk, chunk_tiles, part_tiles, Kp);
    const bool valid = wi_.valid;
    const int64_t w = wi_.w;
    __shared__ int redi[4];
    const HvsRow h = hs[i];
    const int vs = vslot[i];
    const int hh = (int)(w & 1);
    const int8_t *vrow = Vq + (((((int64_t)(vs >> 5) * (Kp >> 6) + (w >> 1)) * lbt + pl0) * 32 + (vs & 31)) * 64);
    uint4 run[4][2]; // [plane][run]: the 16 bytes at positions 16 hh .. and 32 + 16 hh .. of the row of the step
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            run[l][u] = valid ? *reinterpret_cast<const uint4 *>(vrow + (int64_t)l * 32 * 64 + 32 * u + 16 * hh) : make_uint4(0, 0, 0, 0);
    const unsigned dh0 = (unsigned)node[i] * 0x85EBCA6Bu + (unsigned)(w * 32) * 0x9E3779B9u;
    int *A = Abuf + (int64_t)i * 32 * ncdp + wi_.cd;
    int cs = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        // configuration j of the word sits at byte vq_pos(32 hh + j) of the row: run (j >> 2) & 1, byte ((j >> 3) << 2) | (j & 3)
        const int u = (j >> 2) & 1, off = ((j >> 3) << 2) | (j & 3);
        unsigned dv = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const uint4 v4 = run[l][u];
            const unsigned word = (off >> 2) == 0 ? v4.x : (off >> 2) == 1 ? v4.y : (off >> 2) == 2 ? v4.z : v4.w;
            dv |= ((word >> (8 * (off & 3))) & 0xffu) << (8 * l);
        }
        const int qv = (int)((dv ^ 0x80808080u) - 0x80808080u);
        const double hv = (double)(qv < 0 ? -qv : qv);
        const int a = valid ? A[(int64_t)j * ncdp] : 0;
        const double Ea = fma((double)a, h.sg2, h.sgq0);
        const double dith = (double)(int)(dh0 + (unsigned)j * 0x9E3779B9u) * 2.3283064365386963e-10;
        const int v = valid ? __double2loint(fma(hv, Ea * h.it, dith) + 6755399441055744.0) : 0;
        if (valid) A[(int64_t)j * ncdp] = v;
        cs += v;
    }
    cs = wave_sum_i(cs);
    if ((tid & 63) == 0) redi[tid >> 6] = cs;
    __syncthreads();
    if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&hs[i].csum), (unsigned long long)(long long)(redi[0] + redi[1] + redi[2] + redi[3]));
}

// backward: S_c += sum over the workgroup's configurations of u_k b_kc, for the chunk's entries
__global__ __launch_bounds__(256) void k_hvs_bwd(const int *__restrict__ rows, const int *__restrict__ nw, int64_t wcap, const int2 *__restrict__ spins,
                                                 const unsigned *__restrict__ Sb, int64_t Kp, const int *__restrict__ Ubuf, int64_t ncdp,
                                                 long long *__restrict__ S, int ntk, int chunk_tiles, int part_tiles) {
    const int i = blockIdx.z, r = rows[i], m = nw[r], tid = threadIdx.x, base = blockIdx.y * EC;
    if (base >= m) return;
    const int cnt = min(EC, m - base);
    const int64_t wpr = Kp >> 5;
    const WordIdx wi_ = word_index(ntk, chunk_tiles, part_tiles, Kp);
    const int64_t w = wi_.w;
    __shared__ int2 ss[EC];
    if (tid < cnt) ss[tid] = spins[(int64_t)i * wcap + base + tid];
    __syncthreads();
    int vq[32];
    const int *U = Ubuf + (int64_t)i * 32 * ncdp + wi_.cd;
#pragma unroll
    for (int j = 0; j < 32; ++j) vq[j] = wi_.valid ? U[(int64_t)j * ncdp] : 0;
    long long *Sr = S + (int64_t)i * wcap + base;
    for (int a0 = 0; a0 < cnt; a0 += 4) {
        unsigned x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + u < cnt ? a0 + u : cnt - 1;
            const int2 s = ss[a];
            x[u] = a0 + u < cnt && s.x >= 0 ? Sb[(int64_t)s.x * wpr + w] : 0u;
            if (a0 + u < cnt && s.y >= 0) x[u] ^= Sb[(int64_t)s.y * wpr + w];
        }
        int part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            part[u] = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) part[u] = __mul24((int)__builtin_amdgcn_ubfe(x[u], j, 1), vq[j]) + part[u];
        }
        // four wave sums at once: the cross-lane steps of the four chains overlap
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += __shfl_xor(part[u], o);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (a0 + u < cnt && part[u] != 0)
                    atomicAdd(reinterpret_cast<unsigned long long *>(&Sr[a0 + u]), (unsigned long long)(long long)part[u]);
        }
    }
}

// (H p)_c = t (sum_k u_k - 2 S_c) on the working-set entries (k_finalize_i8, hv); the constant statistic: t sum_k u_k
__global__ __launch_bounds__(256) void k_hvs_finalize(const int *__restrict__ rows, const long long *__restrict__ t0, const int *__restrict__ nw,
                                                      const int *__restrict__ FV, int T, int64_t wcap, const long long *__restrict__ S,
                                                      const HvsRow *__restrict__ hs, int64_t Qp, int64_t Qf, int64_t cconst,
                                                      double *__restrict__ Hout) {
    const int i = blockIdx.y, r = rows[i], m = nw[r];
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= m) return;
    const int c = FV[t0[r] * T + a];
    const HvsRow h = hs[i];
    double v = 0.0;
    if (c < Qf) v = h.t * (double)(h.csum - 2 * S[(int64_t)i * wcap + a]);
    else if (c == cconst) v = h.t * (double)h.csum;
    Hout[(int64_t)r * Qp + c] = v;
}

} // namespace

static int64_t hvs_ncdp(const DevProblem &d) { return (d.Kp / 32 + 255) / 256 * 256; } // words of a full pass, padded to whole workgroups

size_t i8_hv_sparse_bytes(const DevProblem &d, int nrows, int64_t wcap) {
    return (size_t)nrows * ((size_t)wcap * (sizeof(int) + sizeof(int2) + sizeof(long long)) + sizeof(HvsRow) + (size_t)32 * hvs_ncdp(d) * sizeof(int)) + 512;
}

// The products of the `nrows` rows listed on the device (rows: local row, node: its spin, vslot: the slot of its V planes, all three by
// position in the list; nw, t0 indexed by local row: size and first tile of its working-set list FV) for the direction P, into Hout at
// the entries of the list.  buf: i8_hv_sparse_bytes(d, nrows, wcap) bytes, wcap >= every listed nw.
int i8_hv_sparse(void *ws, const DevProblem &d, int nrows, const int *rows, const int *node, const int *vslot, const long long *t0, const int *nw,
                 const int *FV, int T, int64_t wcap, const double *P, double *Hout, int64_t kchunk, int64_t kpart, void *buf, hipStream_t st,
                 std::string *err) {
    I8Ws *w = static_cast<I8Ws *>(ws);
    if (!w || !w->Vq || nrows <= 0 || nrows > 65535 || d.ko > 2 || wcap > 65536) {
        if (err) *err = "i8_hv_sparse: no V planes, more than two spins per statistic or a working set above 65536 entries";
        return GML_EINVAL;
    }
    const int nsplit = (int)((d.Kp + kchunk - 1) / kchunk);
    int chunk_tiles = (int)(kchunk / 256), part_tiles = (int)(kpart / 256), ntk = nsplit * part_tiles;
    if (kpart >= kchunk) {
        chunk_tiles = part_tiles = 1;
        ntk = (int)(d.Kp / 256);
    }
    const unsigned nblk = (unsigned)(((int64_t)ntk * 8 + 255) / 256);
    const int64_t ncdp = (int64_t)nblk * 256; // (<= hvs_ncdp(d))
    char *b = static_cast<char *>(buf);
    HvsRow *hs = reinterpret_cast<HvsRow *>(b);
    b += ((size_t)nrows * sizeof(HvsRow) + 255) / 256 * 256;
    long long *S = reinterpret_cast<long long *>(b);
    b += (size_t)nrows * wcap * sizeof(long long);
    int2 *spins = reinterpret_cast<int2 *>(b);
    b += (size_t)nrows * wcap * sizeof(int2);
    int *qW = reinterpret_cast<int *>(b);
    b += ((size_t)nrows * wcap * sizeof(int) + 255) / 256 * 256;
    int *Abuf = reinterpret_cast<int *>(b);
    I8CHK(hipMemsetAsync(Abuf, 0, (size_t)nrows * 32 * ncdp * sizeof(int), st));
    hipLaunchKernelGGL(k_hvs_quant, dim3((unsigned)nrows), dim3(256), 0, st, rows, vslot, t0, nw, FV, T, P, d.Qp, d.Qf, d.cconst, d.keys, d.ko,
                       w->sc[0].tau, w->vscale(), wcap, qW, spins, S, hs);
    const unsigned nch = (unsigned)((wcap + EC - 1) / EC);
    hipLaunchKernelGGL(k_hvs_fwd, dim3(nblk, nch, (unsigned)nrows), dim3(256), 0, st, rows, nw, wcap, qW, spins, d.Sb, d.Kp, Abuf, ncdp, ntk, chunk_tiles,
                       part_tiles);
    hipLaunchKernelGGL(k_hvs_mid, dim3(nblk, 1, (unsigned)nrows), dim3(256), 0, st, node, vslot, d.Kp, w->Vq, w->LBT, w->vpl0(), hs, Abuf, ncdp, ntk,
                       chunk_tiles, part_tiles);
    hipLaunchKernelGGL(k_hvs_bwd, dim3(nblk, nch, (unsigned)nrows), dim3(256), 0, st, rows, nw, wcap, spins, d.Sb, d.Kp, Abuf, ncdp, S, ntk, chunk_tiles,
                       part_tiles);
    hipLaunchKernelGGL(k_hvs_finalize, dim3((unsigned)((wcap + 255) / 256), (unsigned)nrows), dim3(256), 0, st, rows, t0, nw, FV, T, wcap, S, hs,
                       d.Qp, d.Qf, d.cconst, Hout);
    I8CHK(hipGetLastError());
    return GML_OK;
}

} // namespace gml
