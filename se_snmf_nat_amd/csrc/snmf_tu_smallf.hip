// snmf_tu_smallf.hip -- launches of the small-F kernels (snmf_smallf.h: spectrograms of at most two 32-row tiles, the Mel solves).
#include "snmf_internal.h"
#include "snmf_smallf.h"

template <int NF, int NK>
static int launch_hstep_sf_n(snmf_plan* pl, const StepArgs& a, bool obj) {
    dim3 g(pl->sf_grid), b(snmf::kSfWaves * 64);
    return obj ? launch_big(snmf::k_hstep_sf<NF, NK, true>, g, b, pl->lds_sf, pl->ctx->stream, a)
               : launch_big(snmf::k_hstep_sf<NF, NK, false>, g, b, pl->lds_sf, pl->ctx->stream, a);
}
template <int NF>
static int launch_hstep_sf_f(snmf_plan* pl, const StepArgs& a, bool obj) {
    switch (pl->nk) {
        case 1: return launch_hstep_sf_n<NF, 1>(pl, a, obj);
        case 2: return launch_hstep_sf_n<NF, 2>(pl, a, obj);
        case 3: return launch_hstep_sf_n<NF, 3>(pl, a, obj);
        case 4: return launch_hstep_sf_n<NF, 4>(pl, a, obj);
        case 5: return launch_hstep_sf_n<NF, 5>(pl, a, obj);
        case 6: return launch_hstep_sf_n<NF, 6>(pl, a, obj);
        case 7: return launch_hstep_sf_n<NF, 7>(pl, a, obj);
        default: return launch_hstep_sf_n<NF, 8>(pl, a, obj);
    }
}
int launch_hstep_sf(snmf_plan* pl, StepArgs a, bool obj) {
    a.stagger = pl->sf_stagger;
    a.n_tiles = pl->rp_tiles;  // only tiles that hold a frame (the pad tiles of both H buffers are zero and stay zero)
    a.n_full = pl->sf_nfull;   // ... of which [sf_nfull, rp_tiles) are shared by four waves each
    a.part_S = pl->sf_share ? 4 : 0;
    return pl->nf == 1 ? launch_hstep_sf_f<1>(pl, a, obj) : launch_hstep_sf_f<2>(pl, a, obj);
}

template <int NF, int NK>
static int launch_wstats_sf_n(snmf_plan* pl, const StepArgs& a, bool obj) {
    dim3 g(pl->n_chunks), b(snmf::kSfWaves * 64);
    hipStream_t st = pl->ctx->stream;
    if (obj) {
        auto kern = snmf::k_wstats_sf<NF, NK, true>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_wsf));
        hipLaunchKernelGGL(kern, g, b, pl->lds_wsf, st, a, pl->n_chunks, 0, pl->n_mat);
    } else {
        auto kern = snmf::k_wstats_sf<NF, NK, false>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_wsf));
        hipLaunchKernelGGL(kern, g, b, pl->lds_wsf, st, a, pl->n_chunks, 0, pl->n_mat);
    }
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
template <int NF>
static int launch_wstats_sf_f(snmf_plan* pl, const StepArgs& a, bool obj) {
    switch (pl->nk) {
        case 1: return launch_wstats_sf_n<NF, 1>(pl, a, obj);
        case 2: return launch_wstats_sf_n<NF, 2>(pl, a, obj);
        case 3: return launch_wstats_sf_n<NF, 3>(pl, a, obj);
        default: return launch_wstats_sf_n<NF, 4>(pl, a, obj);
    }
}
int launch_wstats_sf(snmf_plan* pl, const StepArgs& a0, bool obj) {
    StepArgs a = a0;
    a.part_S = pl->wsf_share ? 4 : 0;  // a workgroup's single remainder tile is shared by its eight waves (snmf_smallf.h)
    return pl->nf == 1 ? launch_wstats_sf_f<1>(pl, a, obj) : launch_wstats_sf_f<2>(pl, a, obj);
}

