// Shared pieces of the int8-limb kernels (gml_kernels_i8.hip: the 38/31-bit pass "i8x", Hessians, Hessian-vector forms;
// gml_kernels_i8w.hip: the FP64-grade 54/47-bit pass "i8w").  Internal, device side.
#pragma once
#include "../../include/gml.h"
#include "gml_dev.h"
#include "gml_bits.h"

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
    int vpl0() const { return LBT - 4; }
    double vscale() const { return LBT == 6 ? 65536.0 : 1.0; }
};

// LDS tiles of [rows][64 bytes] with the 16-byte slots XOR-swizzled by (row>>2)&3, which makes the ds_read_b128 fragment
// reads (lane = row) conflict-free.
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

#define MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_32x32x32_i8((a), (b), (c), 0, 0, 0)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

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

// ---- the FP64-grade pass (gml_kernels_i8w.hip) ---------------------------------------------------------------------------
struct FwdWArgs {
    const DevProblem *d;
    const int8_t *Tq;
    const SlotScalars *sc;
    const int *rowcol, *groups;
    int ngroups, form;
    bool want_f; // objective only: sum |V| per slot (with the gradient, f comes out of the backward GEMM)
    double *F;   // RPLE: the FP64 sum of the objective terms per slot
    int8_t *Vq;
    hipStream_t st;
};
void launch_fwd_i8w(const FwdWArgs &a);
void launch_finalize_i8w(const int32_t *Gacc, const SlotScalars &sc, const int *srow, const int *rowcol, int slot0, int ns, int64_t Qp,
                         int64_t Qfp, int64_t Qf, int64_t cconst, int form, int want_grad, double *G, double *f, int nplanes,
                         int64_t plane_stride, SlotResult *res, hipStream_t st);

} // namespace gml
