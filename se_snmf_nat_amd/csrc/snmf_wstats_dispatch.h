// snmf_wstats_dispatch.h -- the template dispatch of the W-statistics kernel family (k_wstats) over divergence, objective and
// leftover-column variants; included by the translation units that instantiate one group of geometries each
// (snmf_tu_wstats.hip: NK = 16 + the entry, snmf_tu_wstats4.hip: NK = 4, snmf_tu_wstats8.hip: NK = 8).
#pragma once
#include "snmf_internal.h"
#include "snmf_generic.h"

// k_wstats dispatch
template <int NK, int NWB, int NL, int WPS, int WM, int BM, bool OBJ, int TT = 32, int LX = 0, bool TIL = false>
static int launch_wstats_one(snmf_plan* pl, const StepArgs& a, int mat_index) {
    // consumer teams (plan->til > 1: fewer row tiles than consumer waves) are instantiated for the KL statistics of the NK = 4
    // loader geometries only -- the Mel solves; other divergences at such shapes keep every consumer wave on every tile
    if constexpr (!TIL && NK == 4 && NL > 0 && WM == 0 && BM == BM_KL && TT == 32) {
        if (pl->til > 1) return launch_wstats_one<NK, NWB, NL, WPS, WM, BM, OBJ, TT, LX, true>(pl, a, mat_index);
    }
    const bool split = pl->n_ch1 > 0;  // uneven row-group split: 1-D grid, group 0's chunks first
    dim3 g(split ? pl->n_chunks + (pl->n_fg - 1) * pl->n_ch1 : pl->n_chunks, split ? 1 : pl->n_fg, pl->n_kg), b((NWB + NL) * 64);
    StepArgs as = a;
    as.n_ch1 = split ? pl->n_ch1 : 0;
    if (!TIL) as.til = 1;
    if constexpr (TIL) {
        auto kern = k_wstats_teams<NK, NWB, NL, WPS, WM, BM, OBJ, TT, LX>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_w));
        hipLaunchKernelGGL(kern, g, b, pl->lds_w, pl->ctx->stream, as, pl->n_chunks, mat_index, pl->n_mat);
    } else {
        auto kern = k_wstats<NK, NWB, NL, WPS, WM, BM, OBJ, TT, LX>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_w));
        hipLaunchKernelGGL(kern, g, b, pl->lds_w, pl->ctx->stream, as, pl->n_chunks, mat_index, pl->n_mat);
    }
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
// The "P" statistics of a Euclidean W step without the Lam' pass: P = max(W*H, flr) * H' = W * (H*H') wherever W*H is above
// the 1e-9 floor (everywhere that matters: an entry at the floor contributes < 1e-9 * sum(h) either way, far below the
// engine's fp32 rounding of P).  H*H' is the V*H' launch with the H image as "V" (r rows, no extra row), by kappa-groups
// like the Q launch; its chunk slabs are added in fp64, and W * Gram is one small GEMM into the P slab of chunk 0 (the P
// slabs of the other chunks stay zero from plan creation), so k_reduce / k_wfin / k_wapply see an ordinary P.
template <int NK, int NWB, int NL, int WPS>
static int launch_gram_p(snmf_plan* pl, const StepArgs& a) {
    StepArgs ag = a;
    ag.V = a.Hin;
    ag.F = pl->p.r;
    ag.Fp = ag.Fm = ag.Fq = pl->rp;
    ag.nf = pl->rp / 32;
    ag.xr = 0;
    ag.n_ch1 = 0;
    ag.til = 1;  // (the consumer teams are a property of the plan's F: here the rows are H's)
    ag.slabs = pl->gram_slabs;
    if (pl->kq_kg) {  // r > 256: by kappa-groups on the loader-wave geometry, like the Q launch
        ag.ldh = 260;
        ag.kc = 1;
        ag.nbuf = 2;  // (kq_lds holds two tile buffers)
        auto kern = k_wstats<8, 4, 4, 2, 3, BM_EUC, false, 32>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->kq_lds));
        hipLaunchKernelGGL(kern, dim3(pl->gram_chunks, (ag.nf + 3) / 4, pl->kq_kg), dim3(512), pl->kq_lds, pl->ctx->stream, ag,
                           pl->gram_chunks, 0, 1);
    } else {          // r <= 256: the plan's own V * H' kernel
        auto kern = k_wstats<NK, NWB, NL, WPS, 3, BM_EUC, false, 32>;
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_w));
        hipLaunchKernelGGL(kern, dim3(pl->gram_chunks, (ag.nf + NWB - 1) / NWB, 1), dim3((NWB + NL) * 64), pl->lds_w, pl->ctx->stream, ag,
                           pl->gram_chunks, 0, 1);
    }
    HIP_TRY(hipGetLastError());
    const size_t n = (size_t)pl->rp * pl->rp;
    hipLaunchKernelGGL(k_gram_sum, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, pl->ctx->stream,
                       (const float*)pl->gram_slabs, pl->gram_chunks, n, pl->gram32, (const int*)&pl->st->stop);
    HIP_TRY(hipGetLastError());
    // P(f, k) = sum_j W(f, j) Gram(j, k), element (f, k) at k * Fp + f of a P slab.  The product is tiny and, as one
    // launch over K = r, a chain of r / 16 dependent tile steps on a few workgroups (33 us at r = 256): the contraction is
    // cut into up to eight ranges whose partial products go to the P slabs of chunks 0 .. 7 -- k_reduce adds the chunks in
    // fixed order anyway; the P slabs of the remaining chunks stay zero from plan creation.
    const int nsplit = std::max(1, std::min(std::min(8, pl->n_chunks), (pl->p.r + 31) / 32));
    const int kchunk = (((pl->p.r + nsplit - 1) / nsplit) + 15) / 16 * 16;
    const long long nW = (long long)pl->Fp * pl->rp;
    return g_gemm(pl, pl->Wcf, 1, pl->Fp, pl->gram32, 1, pl->rp, pl->slabs + nW, 1, pl->Fp, pl->p.F, pl->p.r, pl->p.r, kchunk,
                  nW * pl->n_mat);
}
template <int NK, int NWB, int NL, int WPS, int TT = 32>
static int launch_wstats_geo(snmf_plan* pl, const StepArgs& a, bool obj) {
    if (pl->bm == BM_KL) {
        // statistics columns past the last full 32-column tile: up to 8 go through the VALU (k_wstats<..., LX>; loader
        // geometries only: r = 100 at the reference's settings)
        const int left = pl->p.r - 32 * (pl->nk - 1);
        const int lx = (pl->nk >= 2 && pl->n_kg == 1 && left <= 8) ? (left + 3) / 4 : 0;
        // (NK = 4 geometries only: at NK = 8 -- 128 accumulator registers -- the extra code spills: 69 VGPRs at LX = 2)
        if constexpr (NL > 0 && TT == 32 && NK == 4) {
            if (lx == 1) return obj ? launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, true, TT, 1>(pl, a, 0)
                                    : launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, false, TT, 1>(pl, a, 0);
            // (two leftover groups only on the four-consumer geometry: at its 168-register limit the eight-consumer one answered the
            //  second group with 55-79 spilled VGPRs -- r = 101..104 at F = 513 take the padded fourth MFMA tile there instead)
            if constexpr (NWB == 4) {
                if (lx == 2) return obj ? launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, true, TT, 2>(pl, a, 0)
                                        : launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, false, TT, 2>(pl, a, 0);
            }
        }
        return obj ? launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, true, TT>(pl, a, 0)
                   : launch_wstats_one<NK, NWB, NL, WPS, 0, BM_KL, false, TT>(pl, a, 0);
    }
    if (pl->bm == BM_EUC) {
        if (pl->gram_p && !pl->M && !obj && TT == 32) SN_TRY((launch_gram_p<NK, NWB, NL, WPS>(pl, a)));
        else
            SN_TRY(obj ? (launch_wstats_one<NK, NWB, NL, WPS, 1, BM_EUC, true, TT>(pl, a, 1))
                       : (launch_wstats_one<NK, NWB, NL, WPS, 1, BM_EUC, false, TT>(pl, a, 1)));
        if (pl->kq_kg) {  // V * H^T by 256-column kappa-groups on the loader-wave geometry (see snmf_plan_create)
            StepArgs aq = a;
            aq.ldh = 260;
            aq.kc = 1;
            aq.nbuf = 2;
            aq.n_ch1 = 0;
            aq.til = 1;
            auto kern = k_wstats<8, 4, 4, 2, 3, BM_EUC, false, 32>;
            SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->kq_lds));
            hipLaunchKernelGGL(kern, dim3(pl->kq_chunks, pl->n_fg, pl->kq_kg), dim3(512), pl->kq_lds, pl->ctx->stream, aq, pl->kq_chunks, 0,
                               pl->n_mat);
            HIP_TRY(hipGetLastError());
            return SNMF_OK;
        }
        return launch_wstats_one<NK, NWB, NL, WPS, 3, BM_EUC, false, TT>(pl, a, 0);
    }
    SN_TRY(obj ? (launch_wstats_one<NK, NWB, NL, WPS, 1, BM_GEN, true, TT>(pl, a, 1))
               : (launch_wstats_one<NK, NWB, NL, WPS, 1, BM_GEN, false, TT>(pl, a, 1)));
    return launch_wstats_one<NK, NWB, NL, WPS, 2, BM_GEN, false, TT>(pl, a, 0);
}
