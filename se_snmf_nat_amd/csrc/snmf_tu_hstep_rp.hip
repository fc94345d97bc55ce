// snmf_tu_hstep_rp.hip -- launch of the KL role pipeline k_hstep_rp (double-buffered 32-frame tiles: F = 257, the headline geometry).
// A translation unit of its own: the kernel is the one iterated on most, and its two instantiations compile in parallel with
// the rest of the library (snmf_internal.h).
#include "snmf_internal.h"

int launch_hstep_rp(snmf_plan* pl, StepArgs a, bool obj) {
    dim3 g(pl->rp_grid), b(768);
    a.n_tiles = pl->rp_tiles;
    a.n_full = pl->rp_full;
    a.part_S = pl->rp_S;
    a.part_buf = pl->part_buf;
    a.part_cnt = pl->part_cnt;
    a.lxh = pl->rp_cut;  // (k_hstep_rh's field, free here: 2 = the pair form of the cut)
    if (pl->rp_cut)  // r <= 64: P2 cut over the contraction (snmf_kernels.h: k_hstep_rp<OBJ, CUT>)
        return obj ? launch_big(k_hstep_rp<true, true>, g, b, pl->lds_h, pl->ctx->stream, a)
                   : launch_big(k_hstep_rp<false, true>, g, b, pl->lds_h, pl->ctx->stream, a);
    return obj ? launch_big(k_hstep_rp<true>, g, b, pl->lds_h, pl->ctx->stream, a)
               : launch_big(k_hstep_rp<false>, g, b, pl->lds_h, pl->ctx->stream, a);
}
