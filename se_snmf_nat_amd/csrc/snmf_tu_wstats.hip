// snmf_tu_wstats.hip -- entry of the W-statistics dispatch (k_wstats) + the NK = 16 geometries; the NK = 4 / NK = 8 geometries
// are instantiated in snmf_tu_wstats4.hip / snmf_tu_wstats8.hip so that they compile in parallel (snmf_internal.h).
#include "snmf_wstats_dispatch.h"

int launch_wstats(snmf_plan* pl, bool obj) {
    if (pl->generic) {
        ScopedTimer tm(pl->ctx, FAM_WSTATS);
        return generic_wstats(pl, obj);
    }
    StepArgs a = make_args(pl);
    a.n_tiles = (pl->p.T + pl->TTW - 1) / pl->TTW;
    a.ldh = pl->ldhw;
    a.stagger = pl->stagger_w;
    a.nbuf = pl->nbw;
    a.til = pl->til;

    ScopedTimer tm(pl->ctx, FAM_WSTATS);
    if (pl->wsr) return launch_wstats_sr(pl, a, obj);  // r <= 64 on 3..16 row tiles, KL: statistics rows per wave, operands straight into the MFMA layouts (snmf_tu_smallr.hip)
    if (pl->wsf) return launch_wstats_sf(pl, a, obj);  // F <= 64, r <= 128, KL: a tile per wave (snmf_tu_smallf.hip)
    if (pl->NKT == 4) return launch_wstats_nk4(pl, a, obj);
    if (pl->NKT == 8) return launch_wstats_nk8(pl, a, obj);
    if (pl->TTW == 16) return launch_wstats_geo<16, 4, 0, 1, 16>(pl, a, obj);
    return launch_wstats_geo<16, 4, 0, 1>(pl, a, obj);
}
