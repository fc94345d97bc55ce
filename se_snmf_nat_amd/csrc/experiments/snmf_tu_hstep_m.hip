// snmf_tu_hstep_m.hip -- launch of k_hstep_m, the KL H half-step with merged roles (one wave per SIMD does P1, both epilogues and
// P2 of its own rows / columns: snmf_hstep_m.h).  A translation unit of its own (snmf_internal.h).
// EXPERIMENT: compiled and linked only into -DSNMF_EXPERIMENTS builds (se_snmf_nat_amd/_lib.py: SNMF_EXPERIMENTS=1).
#include "../snmf_internal.h"
#include "snmf_hstep_m.h"

int launch_hstep_m(snmf_plan* pl, StepArgs a, bool obj) {
    dim3 g(pl->hm_grid), b(256);
    a.n_tiles = pl->rp_tiles;
    a.n_full = pl->rp_tiles;
    a.part_S = 0;
    return obj ? launch_big(k_hstep_m<true>, g, b, pl->lds_h, pl->ctx->stream, a)
               : launch_big(k_hstep_m<false>, g, b, pl->lds_h, pl->ctx->stream, a);
}
