// snmf_tu_smallr.hip -- launches of the small-rank kernels (snmf_smallr.h: r <= 64 on 3..16 row tiles, the reference's R = 20 / 10 /
// 30 / 50 settings at F = 513 and r = 32 at F = 257).  A translation unit of its own (snmf_internal.h).
#include "snmf_internal.h"
#include "snmf_smallf.h"
#include "snmf_smallr.h"

int launch_hstep_sr(snmf_plan* pl, StepArgs a, bool obj) {
    a.n_tiles = pl->rp_tiles;  // only tiles that hold a frame (the pad tiles of both H buffers are zero and stay zero)
    a.stagger = pl->sr_stagger;
    dim3 g(pl->sr_grid), b(snmf::kSrWaves * 64);
    hipStream_t st = pl->ctx->stream;
    // (one column tile only: with two the operand prefetch, the resident W fragments and the accumulators do not fit 256 registers --
    //  136-210 spilled in a first build; r = 33..64 keep k_hstep_rp<., CUT>)
    if (pl->nk != 1) return fail(SNMF_ERR_INTERNAL, "k_hstep_sr: nk = %d", pl->nk);
    return obj ? launch_big(snmf::k_hstep_sr<1, true>, g, b, pl->lds_sr, st, a) : launch_big(snmf::k_hstep_sr<1, false>, g, b, pl->lds_sr, st, a);
}

template <int NK, bool OBJ>
static int launch_wstats_sr_n(snmf_plan* pl, const StepArgs& a) {
    auto kern = snmf::k_wstats_sr<NK, OBJ>;
    SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_wsr));
    hipLaunchKernelGGL(kern, dim3(pl->n_chunks), dim3(snmf::kSrWaves * 64), pl->lds_wsr, pl->ctx->stream, a, pl->n_chunks, 0, pl->n_mat);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
int launch_wstats_sr(snmf_plan* pl, const StepArgs& a, bool obj) {
    if (pl->nk != 1) return fail(SNMF_ERR_INTERNAL, "k_wstats_sr: nk = %d", pl->nk);
    StepArgs as = a;
    as.stagger = pl->sr_stagger;
    return obj ? launch_wstats_sr_n<1, true>(pl, as) : launch_wstats_sr_n<1, false>(pl, as);
}
