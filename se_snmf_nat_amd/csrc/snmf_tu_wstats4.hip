// snmf_tu_wstats4.hip -- the NK = 4 geometries of k_wstats (r <= 128: the reference's R = 100), see snmf_wstats_dispatch.h.
#include "snmf_wstats_dispatch.h"

int launch_wstats_nk4(snmf_plan* pl, const StepArgs& a, bool obj) {
    if (pl->NWB == 8) return launch_wstats_geo<4, 8, 4, 3>(pl, a, obj);
    return pl->NLW ? launch_wstats_geo<4, 4, 4, 2>(pl, a, obj) : launch_wstats_geo<4, 4, 0, 2>(pl, a, obj);
}
