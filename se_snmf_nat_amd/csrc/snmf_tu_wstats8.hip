// snmf_tu_wstats8.hip -- the NK = 8 geometries of k_wstats (128 < r <= 256: the headline), see snmf_wstats_dispatch.h.
#include "snmf_wstats_dispatch.h"

int launch_wstats_nk8(snmf_plan* pl, const StepArgs& a, bool obj) {
    return pl->NLW ? launch_wstats_geo<8, 4, 4, 2>(pl, a, obj) : launch_wstats_geo<8, 4, 0, 2>(pl, a, obj);
}
