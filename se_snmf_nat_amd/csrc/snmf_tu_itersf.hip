// snmf_tu_itersf.hip -- launch of k_iter_sf (snmf_smallf.h): the H half-step and the W statistics of a full KL update in one
// launch, for spectrograms of two 32-row tiles (F = 33..64: 64 Mel bands) and r = 65..128 (three or four column tiles: the
// reference's R = 100).  A translation unit of its own: twelve instantiations that compile beside the rest (snmf_internal.h).
#include "snmf_internal.h"
#include "snmf_smallf.h"

StepArgs make_args(snmf_plan* pl);  // snmf_api.hip
template <int NK, bool OBJ, int SK>
static int launch_iter_sf_k(snmf_plan* pl, const StepArgs& a) {
    dim3 g(pl->n_chunks), b(snmf::kSfWaves * 64);
    auto kern = snmf::k_iter_sf<NK, OBJ, SK>;
    SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_isf));
    hipLaunchKernelGGL(kern, g, b, pl->lds_isf, pl->ctx->stream, a, pl->n_chunks, pl->n_mat);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
template <int NK, bool OBJ>
static int launch_iter_sf_s(snmf_plan* pl, const StepArgs& a) {
    if (a.S) return launch_iter_sf_k<NK, OBJ, 2>(pl, a);
    return a.lam_is_u ? launch_iter_sf_k<NK, OBJ, 0>(pl, a) : launch_iter_sf_k<NK, OBJ, 1>(pl, a);
}
int launch_iter_sf(snmf_plan* pl, bool obj) {
    StepArgs a = make_args(pl);
    a.n_tiles = pl->rp_tiles;
    a.part_S = pl->isf_share ? 4 : 0;  // a chunk's single remainder tile is shared by the four pairs (snmf_smallf.h)
    ScopedTimer tm(pl->ctx, FAM_HSTEP);
    if (pl->nk == 3) return obj ? launch_iter_sf_s<3, true>(pl, a) : launch_iter_sf_s<3, false>(pl, a);
    return obj ? launch_iter_sf_s<4, true>(pl, a) : launch_iter_sf_s<4, false>(pl, a);
}
