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
//
// Files: gml_i8_pack.hip (bit images, quantisation of Theta), gml_i8_fwd.hip (forward kernel of the 38/31-bit pass "i8x" and of the
// Hessian-vector forms), gml_i8_bwd.hip (backward kernel, finalisation), gml_i8_hess.hip (working-set Hessians),
// gml_kernels_i8w.hip (forward kernel and finalisation of the FP64-grade 54/47-bit pass "i8w"), gml_i8_pass.hip (workspace and
// the orchestration of a pass).  This header: what they share.  Internal, device side.
#pragma once
#include "../../include/gml.h"
#include "gml_dev.h"
#include "gml_bits.h"
#include <string>

namespace gml {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int LB = 4;  // limb planes of V of the i8x pass (31 significant bits relative to the per-node bound)
constexpr int LBW = 6; // ... of the i8w pass (47 bits): two halves of 3 planes, V / tau = lo + 2^24 hi
constexpr int LFW = 7; // limb planes of Theta of the i8w pass (54 bits): swept as 4 + 3 planes

// largest |V| / tau whose balanced base-256 digits fit the planes: 0x7F7F7F7F resp. 0x7F7F7F7F7F7F, with a margin
constexpr double kVdiv4 = 2130000000.0, kVdiv6 = 1.400e14;

// per-slot scalars of one kind of pass (objective/gradient passes, Hessian-vector passes)
struct SlotScalars {
    double *sigma = nullptr, *tau = nullptr, *invtau = nullptr;
    long long *qconst = nullptr, *csum = nullptr, *asum = nullptr;
    long long *qconst2 = nullptr; // i8w: the constant of the coarse form (the number the top four planes of Theta spell)
    long long *csum2 = nullptr, *asum2 = nullptr; // i8w: the high halves (sum of the planes 3..5; sum of |V| >> 32)
    unsigned *mmax = nullptr; // largest |V| / tau seen per slot in the last pass (i8w: >> 16) (dynamic-range check)
};

// Workspace of the int8-limb passes.  Everything is indexed by SLOT: a pass evaluates the node rows its caller lists
// in consecutive slots (32 slots = one MFMA node tile), so that the tiles it runs are full whatever subset of the
// rows is still active; `srow` maps a slot to the row of the caller's Theta / G arrays.
struct I8Ws {
    int64_t slots = 0; // capacity (multiple of 32)
    int LF = 5;        // limb planes the Tq buffer is sized for (7 when wide)
    int LBT = 4;       // limb planes of a Vq image: 4 (i8x) or 6 (i8w); Gacc is sized for as many
    int8_t *Tq = nullptr, *Vq = nullptr, *Uq = nullptr; // Uq: the V-like limb planes of Hessian-vector passes (on first use; always 4 planes)
    int32_t *Gacc = nullptr; // [gplanes][slots * LBT][Qfp]: one set of i32 gradient accumulators per 2^24 configurations
    int gplanes = 1;
    SlotScalars sc[2]; // [0] objective/gradient passes, [1] Hessian-vector passes
    double *tauovr = nullptr; // per-slot tau imposed by the caller (tracked scale, rescaled re-run), 0 = derive from the bound
    // working-set Hessian on the int8 cores (indexed by ROW of the caller's arrays)
    int64_t hKh = 0, hrows = 0, hcap_elems = 0;
    int8_t *Hq = nullptr;   // limb planes of the Hessian weights over the compact (sub-sampled) index
    unsigned *Mb = nullptr; // row-major twin of Xtb (gathered-row DMA of the Hessian kernel), built on first use
    long long *hS = nullptr, *H64 = nullptr;
    // The consumers of V as a 31-bit number (Hessian weights, Hessian-vector forms) read 4 planes of the image from plane vpl0()
    // on: all of an i8x image, the top four of an i8w image (balanced digits: the top planes are V / (65536 tau) rounded to
    // nearest), whose unit is vscale() * tau.
    // Column compaction of the forward GEMM (objective passes; gml_i8_pack.hip: k_col_union): per slot tile, the list of the
    // statistics columns on which a row of the tile is non-zero, and the forward bit image of those columns alone
    int *cnk = nullptr;    // [slots / 32] 64-column steps of the tile's compact image; -1: the tile runs on all columns
    int *cmap = nullptr;   // [slots / 32][csteps * 64] the columns, ascending, -1 padded
    int8_t *Xc = nullptr;  // [slots / 32][xc_tile bytes] compact forward images, Xb piece layout with the tile's own step count
    int csteps = 0;        // capacity in steps per tile
    int64_t xc_tile = 0;   // = Kp * csteps * 8
    int vpl0() const { return LBT - 4; }
    double vscale() const { return LBT == 6 ? 65536.0 : 1.0; }
};

// LDS tiles of [rows][64 bytes] with the 16-byte slots XOR-swizzled by (row>>2)&3, which makes the ds_read_b128 fragment
// reads (lane = row) conflict-free.
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

#define MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// LDS-DMA ring shared by the GEMM kernels: wait until at most `ahead` later stages of NP loads each are in flight, then the
// workgroup barrier (counted vmcnt, raw s_barrier)
template <int NP>
__device__ __forceinline__ void ring_wait_ahead(int ahead) {
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}


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

#define I8CHK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            if (err) *err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                 \
            return e_ == hipErrorOutOfMemory ? GML_ENOMEM : GML_EHIP;                                 \
        }                                                                                             \
    } while (0)

// what a compacted forward pass hands its kernels (NULL: every tile runs on all columns)
struct ColCompact {
    const int *cnk, *cmap;
    int cstride; // entries of cmap per tile
    const int8_t *Xc;
    int64_t xc_tile;
};

// ---- launchers across the kernel files ----------------------------------------------------------------------------------
// gml_i8_pack.hip
void launch_quant_theta(int LF, int ns, const I8Pass &a, const DevProblem &d, int hv, const double *tauV, int8_t *Tq, const SlotScalars &sc,
                        double vdiv, double vsrc_scale, hipStream_t st, const ColCompact *cc = nullptr);
// the non-zero columns of every listed tile's rows -> cnk / cmap, then the compact forward images of the tiles that got one
void launch_col_compact(const I8Pass &a, const DevProblem &d, I8Ws *w, hipStream_t st);
// gml_i8_fwd.hip
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
    bool coarse = false; // exp forms, LF = 4: V to multiples of 2^8 tau (planes 1..3)
    bool zero_theta = false; // every row of Theta is zero (the first pass of a solve): all energies are 0, the column sweep is skipped
    const ColCompact *cc = nullptr; // objective passes: tiles with a compact column list sweep those columns only
};
void launch_fwd_i8(const FwdLaunch &a, int LF, int form, bool wantf, int hv);
// gml_i8_bwd.hip
void launch_zero_pass(const SlotScalars &sc, double *F, const int *rowcol, int slot0, int ns, int32_t *gacc0, int64_t ngacc4, int nplanes, int64_t plane_stride4,
                      hipStream_t st);
void launch_bwd_i8(int NL, const int8_t *Vin, const DevProblem &d, const int *groups, int ngt, int nNt, int64_t kchunk, int nsplit, int32_t *Gacc,
                   int cpp, int64_t plane_stride, int64_t kpart, int lbt, int pl0, hipStream_t st);
void launch_finalize_i8(const int32_t *Gacc, const SlotScalars &sc, const int *srow, const int *rowcol, int slot0, int ns, const DevProblem &d,
                        int form, int want_grad, int hv, double *G, double *F, int nplanes, int64_t plane_stride, SlotResult *res, hipStream_t st);

// ---- the FP64-grade pass (gml_kernels_i8w.hip) ---------------------------------------------------------------------------
struct FwdWArgs {
    const DevProblem *d;
    const int8_t *Tq;
    const SlotScalars *sc;
    const int *rowcol, *groups;
    int ngroups, form;
    bool want_f; // objective only: sum |V| per slot (with the gradient, f comes out of the backward GEMM)
    bool coarse; // the 30 / 23-bit form of the pass: the top four planes of Theta, V to three planes (planes 3..5; plane 2 zero)
    double *F;   // RPLE: the FP64 sum of the objective terms per slot
    int8_t *Vq;
    hipStream_t st;
    bool zero_theta = false; // every row of Theta is zero: no column sweeps (the energies are 0)
    const ColCompact *cc = nullptr; // tiles with a compact column list sweep those columns only
};
void launch_fwd_i8w(const FwdWArgs &a);
void launch_finalize_i8w(const int32_t *Gacc, const SlotScalars &sc, const int *srow, const int *rowcol, int slot0, int ns, int64_t Qp,
                         int64_t Qfp, int64_t Qf, int64_t cconst, int form, int want_grad, double *G, double *f, int nplanes,
                         int64_t plane_stride, SlotResult *res, bool coarse, hipStream_t st);

} // namespace gml
