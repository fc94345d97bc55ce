// snmf_tu_hstep.hip -- dispatch of the H-step kernel family (k_hstep, k_hstep_rp, k_hstep_rh): a translation unit of its own so that the
// template instantiations compile in parallel with the other families (snmf_internal.h).
#include "snmf_internal.h"

// k_hstep dispatch over (NW, NT, NL, BM, OBJ, UPD)
template <int NW, int NT, int NL, int BM, int TT>
static int launch_hstep_nb(snmf_plan* pl, const StepArgs& a, bool obj, bool upd) {
    dim3 g(pl->grid_h), b((NW + NL) * 64);
    hipStream_t st = pl->ctx->stream;
    if (obj && upd) return launch_big(k_hstep<NW, NT, NL, BM, true, true, false, TT>, g, b, pl->lds_h, st, a);
    if (!obj && upd) return launch_big(k_hstep<NW, NT, NL, BM, false, true, false, TT>, g, b, pl->lds_h, st, a);
    if (obj && !upd) return launch_big(k_hstep<NW, NT, NL, BM, true, false, false, TT>, g, b, pl->lds_h, st, a);
    return SNMF_OK;
}
template <int NW, int NT, int NL, int TT = 32>
static int launch_hstep_g(snmf_plan* pl, const StepArgs& a, bool obj, bool upd) {
    if (pl->bm == BM_KL) return launch_hstep_nb<NW, NT, NL, BM_KL, TT>(pl, a, obj, upd);
    if (pl->bm == BM_EUC) return launch_hstep_nb<NW, NT, NL, BM_EUC, TT>(pl, a, obj, upd);
    return launch_hstep_nb<NW, NT, NL, BM_GEN, TT>(pl, a, obj, upd);
}
// MDI pass (src/snmf_mdi.m:251-257 fused into the Lam pass): synchronous-staging geometry, V rewritten in place
template <int BM>
static int launch_hstep_mdi_b(snmf_plan* pl, const StepArgs& a, bool obj, bool upd) {
    dim3 g(pl->grid_mdi), b(8 * 64);
    hipStream_t st = pl->ctx->stream;
    if (upd) return obj ? launch_big(k_hstep<8, 1, 0, BM, true, true, true>, g, b, pl->lds_mdi, st, a)
                        : launch_big(k_hstep<8, 1, 0, BM, false, true, true>, g, b, pl->lds_mdi, st, a);
    return launch_big(k_hstep<8, 1, 0, BM, true, false, true>, g, b, pl->lds_mdi, st, a);  // imputation (+ objective)
}

int launch_hstep(snmf_plan* pl, bool obj, bool upd) {
    if (pl->generic) {
        ScopedTimer tm(pl->ctx, FAM_HSTEP);
        return generic_hstep(pl, obj, upd);
    }
    StepArgs a = make_args(pl);
    a.n_tiles = pl->Tp / (pl->TTH * pl->NT);
    a.stagger = pl->stagger_h;
    ScopedTimer tm(pl->ctx, FAM_HSTEP);
    if (pl->M) {
        a.n_tiles = pl->Tp / 32;
        a.stagger = 0;
        a.M = pl->M;
        a.Vw = pl->V;
        a.impute = pl->it_done >= 1 ? 1 : 0;  // the first Lam pass precedes any imputation (:175 only)
        if (pl->bm == BM_KL) return launch_hstep_mdi_b<BM_KL>(pl, a, obj, upd);
        if (pl->bm == BM_EUC) return launch_hstep_mdi_b<BM_EUC>(pl, a, obj, upd);
        return launch_hstep_mdi_b<BM_GEN>(pl, a, obj, upd);
    }
    if (pl->sr && upd) return launch_hstep_sr(pl, a, obj);  // KL update launches, r <= 64 on 3..16 row tiles: a tile per workgroup cut by row tiles (snmf_tu_smallr.hip)
    if (pl->sf && upd) return launch_hstep_sf(pl, a, obj);  // KL update launches, F <= 64: a tile per wave (snmf_tu_smallf.hip)
    if (pl->rh && upd) return launch_hstep_rh(pl, a, obj);  // KL update launches of the 9..16-row-tile geometry (snmf_tu_hstep_rh.hip)
    if (pl->NWH == 8 && pl->NLH == 4) {
#ifdef SNMF_EXPERIMENTS
        if (pl->hm && pl->bm == BM_KL && upd && !a.S) return launch_hstep_m(pl, a, obj);  // merged roles, one wave per SIMD (experiments/snmf_tu_hstep_m.hip)
#endif
        if (pl->hstep_rp && pl->bm == BM_KL && upd) return launch_hstep_rp(pl, a, obj);  // KL update launches: the role pipeline (snmf_tu_hstep_rp.hip)
        return launch_hstep_g<8, 1, 4>(pl, a, obj, upd);
    }
    if (pl->NWH == 4) return pl->NT == 2 ? launch_hstep_g<4, 2, 0>(pl, a, obj, upd) : launch_hstep_g<4, 1, 0>(pl, a, obj, upd);
    if (pl->TTH == 16) return launch_hstep_g<8, 1, 0, 16>(pl, a, obj, upd);
    return pl->NT == 2 ? launch_hstep_g<8, 2, 0>(pl, a, obj, upd) : launch_hstep_g<8, 1, 0>(pl, a, obj, upd);
}
