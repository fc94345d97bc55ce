// snmf_tu_small.hip -- dispatch of the persistent single-launch H-only solvers (k_hsolve_small, k_hsolve_frame) (snmf_internal.h).
#include "snmf_internal.h"

// persistent single-launch H-only solves: n_solves independent workgroups of tps <= 32 frames each
int launch_small(snmf_plan* pl, int n_solves, int tps, double* divh, double* costh, DevState* st,
                        float* recon, int recon_rx) {
    StepArgs a = make_args(pl);
    a.Hout = pl->H[pl->cur];  // in place
    a.n_tiles = 1;
    SmallArgs sa{};
    sa.max_iter = pl->p.max_iter;
    sa.cost_check = pl->p.cost_check;
    sa.conv_eps = pl->p.conv_eps;
    sa.divh = divh;
    sa.costh = costh;
    sa.st = st;
    sa.tps = tps;
    sa.recon = (tps == 1 && pl->frame_fb) ? recon : nullptr;  // only k_hsolve_frame produces the reconstructions
    sa.wn = pl->wn;
    sa.Rx = recon_rx;
    auto launch = [&](auto kern) -> int {
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_small));
        hipLaunchKernelGGL(kern, dim3(n_solves), dim3(512), pl->lds_small, pl->ctx->stream, a, sa);
        HIP_TRY(hipGetLastError());
        return SNMF_OK;
    };
    ScopedTimer tm(pl->ctx, FAM_HSTEP);
    const bool obj = pl->p.cost_check != 0;
    if (tps == 1 && pl->frame_fb) {
        auto launch_f = [&](auto kern) -> int {
            SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_frame));
            hipLaunchKernelGGL(kern, dim3(n_solves), dim3(512), pl->lds_frame, pl->ctx->stream, a, sa, (const float*)pl->Wcf);
            HIP_TRY(hipGetLastError());
            return SNMF_OK;
        };
        auto by_bm = [&](auto fbc, auto kbc) -> int {
            constexpr int FB = decltype(fbc)::value, KB = decltype(kbc)::value;
            auto by_obj = [&](auto bmc, auto rc) -> int {
                constexpr int BM = decltype(bmc)::value;
                constexpr bool RC = decltype(rc)::value;
                return obj ? launch_f(k_hsolve_frame<FB, KB, BM, true, RC>) : launch_f(k_hsolve_frame<FB, KB, BM, false, RC>);
            };
            auto by_rc = [&](auto bmc) -> int {
                return sa.recon ? by_obj(bmc, std::true_type{}) : by_obj(bmc, std::false_type{});
            };
            if (pl->bm == BM_KL) return by_rc(std::integral_constant<int, BM_KL>{});
            if (pl->bm == BM_EUC) return by_rc(std::integral_constant<int, BM_EUC>{});
            return by_rc(std::integral_constant<int, BM_GEN>{});
        };
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        using I25 = std::integral_constant<int, 25>;
        if (pl->frame_fb == 4) return pl->frame_kb == 16 ? by_bm(I4{}, I16{}) : by_bm(I4{}, I25{});
        return pl->frame_kb == 16 ? by_bm(I8{}, I16{}) : by_bm(I8{}, I25{});
    }
    if (pl->bm == BM_KL) return obj ? launch(k_hsolve_small<BM_KL, true>) : launch(k_hsolve_small<BM_KL, false>);
    if (pl->bm == BM_EUC) return obj ? launch(k_hsolve_small<BM_EUC, true>) : launch(k_hsolve_small<BM_EUC, false>);
    return obj ? launch(k_hsolve_small<BM_GEN, true>) : launch(k_hsolve_small<BM_GEN, false>);
}
