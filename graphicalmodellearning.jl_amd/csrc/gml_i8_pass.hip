// Int8-limb path, part 5: the workspace of the passes and the orchestration of one pass (quantise -> forward -> backward ->
// finalise) for both widths, "i8x" (38/31 bits) and "i8w" (54/47 bits, FP64-grade).  Overview: gml_i8.h.
#include "gml_i8.h"
#include <algorithm>
#include <string>

namespace gml {

// per-slot results of the last pass of the given kind (device pointers): tau (scale of the V / u planes), mmax
void i8_slot_results(void *p, int hv, const double **tau, const unsigned **mmax) {
    I8Ws *w = static_cast<I8Ws *>(p);
    *tau = w ? w->sc[hv ? 1 : 0].tau : nullptr;
    *mmax = w ? w->sc[hv ? 1 : 0].mmax : nullptr;
}

// the compaction table of the last objective pass (device pointer; NULL: never compacted): steps per slot tile, -1 = all columns
void i8_compact_table(void *p, const int **cnk, int *csteps) {
    I8Ws *w = static_cast<I8Ws *>(p);
    *cnk = w && w->csteps > 0 ? w->cnk : nullptr;
    *csteps = w ? w->csteps : 0;
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
    void *ptrs[] = {w->Tq, w->Vq, w->Uq, w->Gacc, w->tauovr, w->Hq, w->hS, w->H64, w->Mb, w->cnk, w->cmap, w->Xc};
    for (void *q : ptrs)
        if (q) (void)dev_free_synced(q);
    for (auto &sc : w->sc) {
        void *qs[] = {sc.sigma, sc.tau, sc.invtau, sc.qconst, sc.qconst2, sc.csum, sc.asum, sc.csum2, sc.asum2, sc.mmax};
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
        I8CHK(dev_malloc(&sc.qconst2, sizeof(long long) * slots));
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
    const bool coarse = !hv && a.coarse && a.form != GML_RPLE; // the cheap form of an objective pass, either width
    if (wide && d.Qfp > ((int64_t)1 << 21)) {
        // the partial sums sum_c q_c b_c over 4 digit planes (|q| < 2^31) are folded in FP64: exact below 2^53, i.e. up to 2^21 columns
        if (err) *err = "precision i8w holds at most 2^21 statistics columns: use precision i8x";
        return GML_EUNSUPPORTED;
    }
    int rc = i8_ensure(wsp, d, slot_capacity, hv ? -1 : (wide ? 1 : 0), st, err);
    if (rc) return rc;
    I8Ws *w = static_cast<I8Ws *>(*wsp);
    int LF = wide ? LFW : (coarse ? 4 : (a.lf ? a.lf : 5));
    if (LF > w->LF) LF = w->LF;
    if (a.ngroups + 1 > 65536 || a.slot1 > w->slots || a.slot0 % 32 || a.slot1 % 32) {
        if (err) *err = "bad slot range";
        return GML_EINVAL;
    }
    if (hv && !w->Uq) {
        I8CHK(dev_malloc(&w->Uq, (size_t)w->slots * LB * d.Kp));
        I8CHK(hipMemsetAsync(w->Uq, 0, (size_t)w->slots * LB * d.Kp, st));
    }
    // column compaction of the forward GEMM (objective passes over sparse rows): buffers on first use; a workspace that cannot have
    // them (no memory left) simply keeps sweeping all columns
    const int nk_all = (int)(d.Qfp / 64);
    const bool compact = a.compact && !hv && !a.zero_theta && nk_all >= 2;
    ColCompact ccv{};
    const ColCompact *cc = nullptr;
    if (compact) {
        if (!w->cnk && w->csteps == 0) {
            // capacity: a quarter of the columns, 32 steps at most.  Building a tile's compact image costs K * 8 B of HBM writes per
            // step; above a quarter of the columns that eats what the shorter sweep saves (config 4's first passes: 32 of 64 steps,
            // 4 GB of images per pass, 2 % slower than sweeping everything)
            const int csteps = std::min(32, std::max(1, nk_all / 4));
            const int64_t ntile = w->slots / 32, xc_tile = d.Kp * (int64_t)csteps * 8;
            size_t freeb = 0, totalb = 0;
            if (dev_mem_info(&freeb, &totalb) == hipSuccess && (double)ntile * (double)xc_tile < 0.25 * (double)freeb &&
                dev_malloc(&w->cnk, sizeof(int) * ntile) == hipSuccess && dev_malloc(&w->cmap, sizeof(int) * ntile * csteps * 64) == hipSuccess &&
                dev_malloc(&w->Xc, (size_t)ntile * xc_tile) == hipSuccess) {
                w->csteps = csteps;
                w->xc_tile = xc_tile;
            } else {
                (void)hipGetLastError();
                w->csteps = -1; // (tried: not again)
            }
        }
        if (w->csteps > 0) {
            ccv = ColCompact{w->cnk, w->cmap, w->csteps * 64, w->Xc, w->xc_tile};
            cc = &ccv;
        }
    }
    const SlotScalars &sc = w->sc[hv ? 1 : 0];
    const int ns = a.slot1 - a.slot0;
    const bool grad = a.want_grad || hv;
    const int lbg = wide ? LBW : LB; // limb planes of this pass's V and of its gradient accumulators
    int32_t *gacc0 = w->Gacc + (int64_t)a.slot0 * lbg * d.Qfp;
    const int64_t gplane_stride = (int64_t)w->slots * lbg * d.Qfp;
    launch_zero_pass(sc, a.F, a.rowcol, a.slot0, ns, gacc0, grad ? (int64_t)ns * lbg * d.Qfp / 4 : 0, w->gplanes, gplane_stride / 4, st);
    if (LF < 3 && !hv) LF = 3; // 2 limbs exist for the directions of Hessian-vector passes only
    if (cc) launch_col_compact(a, d, w, st); // cnk / cmap / Xc of the listed tiles, in front of the quantisation that follows them
    launch_quant_theta(LF, ns, a, d, hv, w->sc[0].tau, w->Tq, sc, wide ? kVdiv6 : kVdiv4, w->vscale(), st, cc);
    // split-K plan of the backward GEMM (made here: a sub-sampled pass runs its forward kernel over the same parts)
    const int nNt = (int)((d.Qfp + 255) / 256);
    constexpr int TM = 1; // node tiles per backward workgroup (the 8-wave form with two, TM = 2, measured slower)
    const int ngt = (a.ngroups + TM - 1) / TM;
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
        FwdWArgs fw{&d, w->Tq, &sc, a.rowcol, a.groups, a.ngroups, a.form, !a.want_grad, coarse, a.F, w->Vq, st};
        fw.zero_theta = a.zero_theta && !hv;
        fw.cc = cc;
        launch_fwd_i8w(fw);
    } else {
        FwdLaunch fl{(int)(kchunk / 256), (int)(kpart / 256), 0, w, &d, &sc, a.rowcol, a.groups, a.vmap, a.ngroups, a.F, hv ? w->Uq : w->Vq, st};
        fl.coarse = coarse;
        fl.cc = cc;
        fl.zero_theta = a.zero_theta && !hv && kpart == kchunk;
        fl.ntk = ksub > 1 ? nsplit * fl.part_tiles : (int)(d.Kp / 256);
        if (ksub == 1) fl.chunk_tiles = fl.part_tiles = 1; // (every tile: no remapping)
        if (!hv && w->LBT != LB) {
            if (err) *err = "a 31-bit objective pass on a workspace of 6-plane V images";
            return GML_EINVAL;
        }
        // with the gradient requested, f comes out of the backward GEMM for free (column u of row u)
        launch_fwd_i8(fl, LF, a.form, !a.want_grad, hv);
    }
    if (ev) I8CHK(hipEventRecord(ev[1], st));
    if (grad) {
        // chunks per set of i32 accumulators: one set up to 2^24 configurations; beyond, gplanes = ceil(Kp / 2^23) sets of
        // cpp chunks each: cpp * kchunk < (Kp + kchunk) / gplanes + kchunk <= 2^23 + 1.5 * 2^22 < 2^24, so |sum| < 2^31
        const int cpp = (nsplit + w->gplanes - 1) / w->gplanes;
        const int8_t *Vin = hv ? w->Uq : w->Vq;
        if (wide) {
#ifndef I8W_BWD33
            // all six planes in ONE launch (wave tile 192 x 64, 230 registers): the bit operand is loaded and expanded once for six
            // planes.  Rounds 4 ran it as two launches of the 3-plane form (-DI8W_BWD33): 4.52 against 4.43 ms per headline pass, the
            // pass 10.89 against 10.74 ms, interleaved on one box (profiles/r5_ab_i8w_bwd6.txt); the sums are integers either way --
            // the same bits.  Coarse passes: the high half only, the 3-plane form.
            if (!coarse) launch_bwd_i8(6, Vin, d, a.groups, ngt, nNt, kchunk, nsplit, w->Gacc, cpp, gplane_stride, kpart, LBW, 0, st);
            else launch_bwd_i8(3, Vin, d, a.groups, ngt, nNt, kchunk, nsplit, w->Gacc, cpp, gplane_stride, kpart, LBW, 3, st);
#else
            for (int half = coarse ? 1 : 0; half < 2; ++half)
                launch_bwd_i8(3, Vin, d, a.groups, ngt, nNt, kchunk, nsplit, w->Gacc, cpp, gplane_stride, kpart, LBW, 3 * half, st);
#endif
        } else if (coarse) { // the planes 1..3 carry the 23-bit value (plane 0 is zero, its accumulators stay zero)
            launch_bwd_i8(3, Vin, d, a.groups, ngt, nNt, kchunk, nsplit, w->Gacc, cpp, gplane_stride, kpart, LB, 1, st);
        } else {
            launch_bwd_i8(hv == 2 ? 2 : 4, Vin, d, a.groups, ngt, nNt, kchunk, nsplit, w->Gacc, cpp, gplane_stride, kpart, LB, 0, st);
        }
    }
    if (ev) I8CHK(hipEventRecord(ev[2], st));
    if (wide)
        launch_finalize_i8w(w->Gacc, sc, a.srow, a.rowcol, a.slot0, ns, d.Qp, d.Qfp, d.Qf, d.cconst, a.form, grad ? 1 : 0, a.G, a.F, w->gplanes,
                            gplane_stride, a.res, coarse, st);
    else
        launch_finalize_i8(w->Gacc, sc, a.srow, a.rowcol, a.slot0, ns, d, a.form, grad ? 1 : 0, hv, a.G, a.F, w->gplanes, gplane_stride, a.res, st);
    I8CHK(hipGetLastError());
    return GML_OK;
}


} // namespace gml
