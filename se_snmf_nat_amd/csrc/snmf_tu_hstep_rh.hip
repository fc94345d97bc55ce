// snmf_tu_hstep_rh.hip -- launch of the half-tile KL role pipeline k_hstep_rh (9..16 row tiles: F = 513, the geometry the
// reference ships, settings/initial_setting_SNMF_NAT.m:21-29).  A translation unit of its own (snmf_internal.h).
#include "snmf_internal.h"

int launch_hstep_rh(snmf_plan* pl, StepArgs a, bool obj) {
    dim3 g(pl->rp_grid), b(768);
    a.n_tiles = pl->rp_tiles;
    a.n_full = pl->rp_full;
    a.part_S = pl->rp_S;
    a.part_buf = pl->part_buf;
    a.part_cnt = pl->part_cnt;
    if (pl->rh_lxh) {  // P2 cut over the contraction: 1 = nk 4 (r = 97..100) four ways, 2 = nk 7 (r = 193..200) in pairs
        a.lxh = pl->rh_lxh;
        if (pl->rh_lxh == 2)
            return obj ? launch_big(k_hstep_rh<true, 2>, g, b, pl->lds_rh, pl->ctx->stream, a)
                       : launch_big(k_hstep_rh<false, 2>, g, b, pl->lds_rh, pl->ctx->stream, a);
        return obj ? launch_big(k_hstep_rh<true, 1>, g, b, pl->lds_rh, pl->ctx->stream, a)
                   : launch_big(k_hstep_rh<false, 1>, g, b, pl->lds_rh, pl->ctx->stream, a);
    }
    return obj ? launch_big(k_hstep_rh<true>, g, b, pl->lds_rh, pl->ctx->stream, a)
               : launch_big(k_hstep_rh<false>, g, b, pl->lds_rh, pl->ctx->stream, a);
}
