// snmf_prof.h -- DIAGNOSTIC builds only (-DSNMF_PROF, scripts/phase_prof.sh): prints the phase stamps the kernels of such a
// build collect (shares per phase, wave start / end times, in-kernel clock) when a plan is destroyed.  The product
// library is built without SNMF_PROF and contains none of this.
#pragma once
static void snmf_prof_report(snmf_plan* pl) {
    if (pl->prof) {  // diagnostic build: phase shares of the LAST big-kernel launch
        const bool wlast = getenv("SNMF_PROF_W") != nullptr;
        const int nw = wlast ? pl->n_chunks * pl->n_fg * pl->NWB : pl->grid_h * pl->NWH;
        std::vector<unsigned long long> hp((size_t)nw * 12);
        hipMemcpy(hp.data(), pl->prof + (wlast ? (size_t)4096 * 12 : 0), hp.size() * 8, hipMemcpyDeviceToHost);
        if (const char* dump = getenv("SNMF_PROF_DUMP")) {  // raw per-wave phase cycles [nw][12] for offline analysis
            if (FILE* f = fopen(dump, "wb")) {
                fwrite(hp.data(), 8, hp.size(), f);
                fclose(f);
            }
        }
        double tot[12] = {0};
        for (int i = 0; i < nw; ++i)
            for (int j = 0; j < 12; ++j) tot[j] += (double)hp[(size_t)i * 12 + j];
        double all = 0;
        for (int j = 0; j < 12; ++j) all += tot[j];
        static const char* nmh[12] = {"bar_top", "stage", "bar_stage", "p1_pre", "p1_mfma", "p1_epi", "xrow", "bar_p2",
                                      "p2_pre", "p2_mfma", "p2_epi", "stage_out"};
        static const char* nmw[12] = {"loop", "barrier", "ssum+xrow", "p3_mfma", "p3_epi", "p4_mfma", "-", "-", "-", "-", "-", "-"};
        const char* const* nm = wlast ? nmw : nmh;
        fprintf(stderr, "[SNMF_PROF] %s phase shares (avg cycles/wave = %.0f):", wlast ? "k_wstats" : "k_hstep", all / nw);
        for (int j = 0; j < 12; ++j) fprintf(stderr, " %s=%.1f%%", nm[j], 100.0 * tot[j] / all);
        if (!wlast && pl->hstep_rp && pl->NWH == 8) {
            // role pipeline: waves 0-3 of a workgroup are the A team (slots: 4 = wait for ready + contraction, 5 = wait
            // for V + epilogues, 6 = bookkeeping, 11 = post p1b + extra row), waves 4-7 the B team (9 = gates +
            // contraction, 10 = epilogues, 11 = post p2done + loop top)
            double ta[12] = {0}, tb[12] = {0}, sa = 0, sb = 0;
            for (int i = 0; i < nw; ++i)
                for (int j = 0; j < 12; ++j) ((i & 7) < 4 ? ta : tb)[j] += (double)hp[(size_t)i * 12 + j];
            for (int j = 0; j < 12; ++j) { sa += ta[j]; sb += tb[j]; }
            fprintf(stderr, " | A team (cycles/wave %.0f): wait for ready %.1f%% loop %.1f%% epilogues %.1f%% other %.1f%% p1b+xrow %.1f%% | B team (%.0f): gates+loop %.1f%% epilogues %.1f%% post+top %.1f%%",
                    sa / (nw / 2), 100 * ta[3] / sa, 100 * ta[4] / sa, 100 * ta[5] / sa, 100 * ta[6] / sa, 100 * ta[11] / sa, sb / (nw / 2), 100 * tb[9] / sb,
                    100 * tb[10] / sb, 100 * tb[11] / sb);
        }
        std::vector<unsigned long long> hc((size_t)2 * nw);
        hipMemcpy(hc.data(), pl->prof + 98304 + (wlast ? (size_t)2 * 4096 : 0), hc.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> ghz, span;
        for (int i = 0; i < nw; ++i)
            if (hc[2 * i + 1]) {
                ghz.push_back((double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1);
                span.push_back((double)hc[2 * i + 1] * 0.01);
            }
        {   // when did the waves start and end, relative to the first start (100 MHz ticks -> us)
            std::vector<unsigned long long> hs0((size_t)nw);
            hipMemcpy(hs0.data(), pl->prof + 98304 + 16384 + (wlast ? (size_t)4096 : 0), hs0.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> st, en;
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < nw; ++i)
                if (hs0[i]) t0 = std::min(t0, hs0[i]);
            for (int i = 0; i < nw; ++i)
                if (hs0[i]) {
                    st.push_back((double)(hs0[i] - t0) * 0.01);
                    en.push_back((double)(hs0[i] - t0 + hc[2 * i + 1]) * 0.01);
                }
            if (!wlast && !st.empty()) {
                // where do the stragglers sit?  mean end time of the workgroups by blockIdx % 8 (the XCD a workgroup lands
                // on with round-robin dispatch) and by blockIdx / 32 (position in the grid)
                const int wpg = nw / std::max(1, pl->hstep_rp ? pl->rp_grid : pl->grid_h);  // stamped waves per workgroup
                double sx[8] = {0}, nx[8] = {0}, sg[8] = {0}, ng[8] = {0};
                for (int i = 0; i < nw; ++i)
                    if (hs0[i] && wpg > 0) {
                        const int wg = i / wpg;
                        const double e = (double)(hs0[i] - t0 + hc[2 * i + 1]) * 0.01;
                        sx[wg % 8] += e; nx[wg % 8] += 1;
                        sg[(wg / 32) % 8] += e; ng[(wg / 32) % 8] += 1;
                    }
                fprintf(stderr, " | mean wave end us by blockIdx%%8:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", nx[x] ? sx[x] / nx[x] : 0.0);
                fprintf(stderr, " ; by blockIdx/32:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", ng[x] ? sg[x] / ng[x] : 0.0);
            }
            if (wlast && !st.empty()) {
                // k_wstats: wave end times by (row group, tiles in the workgroup's chunk) and by blockIdx % 8 (XCD)
                const int nt = (pl->p.T + pl->TTW - 1) / pl->TTW;
                std::map<std::pair<int, int>, std::pair<double, double>> agg;  // -> (sum, max)
                std::map<std::pair<int, int>, int> cnt;
                double sx[8] = {0}, nx[8] = {0}, mx[8] = {0};
                for (int i = 0; i < nw; ++i) {
                    if (!hs0[i]) continue;
                    const int wg = i / pl->NWB;
                    int grp = 0, chunk = wg, nch = pl->n_chunks;
                    if (pl->n_ch1 > 0) {
                        if (wg >= pl->n_chunks) { grp = 1 + (wg - pl->n_chunks) / pl->n_ch1; chunk = (wg - pl->n_chunks) % pl->n_ch1; nch = pl->n_ch1; }
                    } else { grp = wg / pl->n_chunks; chunk = wg % pl->n_chunks; }
                    const int tiles = (int)(((long long)nt * (chunk + 1)) / nch) - (int)(((long long)nt * chunk) / nch);
                    const double e = (double)(hs0[i] - t0 + hc[2 * i + 1]) * 0.01;
                    auto& a2 = agg[{grp, tiles}];
                    a2.first += e; a2.second = std::max(a2.second, e); cnt[{grp, tiles}] += 1;
                    sx[wg % 8] += e; nx[wg % 8] += 1; mx[wg % 8] = std::max(mx[wg % 8], e);
                }
                fprintf(stderr, " | end us by (group, tiles): ");
                for (auto& kv : agg) fprintf(stderr, "(g%d, %d tiles, %d waves): mean %.1f max %.1f; ", kv.first.first, kv.first.second, cnt[kv.first], kv.second.first / cnt[kv.first], kv.second.second);
                fprintf(stderr, "| by blockIdx%%8 mean/max:");
                for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f/%.1f", nx[x] ? sx[x] / nx[x] : 0.0, mx[x]);
                // phase cycles per wave of the workgroups that end last / first (what are the stragglers doing?)
                std::vector<std::pair<double, int>> byend;
                for (int i = 0; i < nw; ++i)
                    if (hs0[i]) byend.push_back({(double)(hs0[i] - t0 + hc[2 * i + 1]) * 0.01, i});
                std::sort(byend.begin(), byend.end());
                const size_t nq = byend.size() / 10;
                for (int side = 0; side < 2 && nq > 0; ++side) {
                    double ph[6] = {0};
                    for (size_t q2 = 0; q2 < nq; ++q2) {
                        const int i = byend[side ? byend.size() - 1 - q2 : q2].second;
                        for (int j = 0; j < 6; ++j) ph[j] += (double)hp[(size_t)i * 12 + j];
                    }
                    fprintf(stderr, " | %s 10 %% of the waves, k cycles/wave:", side ? "LAST" : "first");
                    for (int j = 0; j < 6; ++j) fprintf(stderr, " %s=%.1f", nmw[j], ph[j] / nq / 1e3);
                }
            }
            if (!st.empty()) {
                std::sort(st.begin(), st.end());
                std::sort(en.begin(), en.end());
                auto q = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                fprintf(stderr, " | wave start us (10/50/90/100 %%): %.1f %.1f %.1f %.1f; wave end us (0/10/50/90/100 %%): %.1f %.1f %.1f %.1f %.1f",
                        q(st, .1), q(st, .5), q(st, .9), q(st, 1.), q(en, 0.), q(en, .1), q(en, .5), q(en, .9), q(en, 1.));
            }
        }
        if (!wlast) {  // per-tile periods of consumer wave 0: the first workgroup, one from the middle, the last
            std::vector<unsigned long long> tt((size_t)16384);
            hipMemcpy(tt.data(), pl->prof + 98304 + 24576, tt.size() * 8, hipMemcpyDeviceToHost);
            const int wgs[3] = {0, pl->grid_h / 2, pl->grid_h - 1};
            for (int wi = 0; wi < 3; ++wi) {
                const int b = wgs[wi];
                if (b < 0 || b >= 1024) continue;
                fprintf(stderr, " | wg %d tile periods us:", b);
                for (int i = 1; i < 16 && tt[(size_t)b * 16 + i]; ++i)
                    fprintf(stderr, " %.1f", (double)(tt[(size_t)b * 16 + i] - tt[(size_t)b * 16 + i - 1]) * 0.01);
            }
        }
        if (!ghz.empty()) {
            std::sort(ghz.begin(), ghz.end());
            std::sort(span.begin(), span.end());
            fprintf(stderr, " | in-kernel clock %.3f GHz (median of %zu waves; min %.3f max %.3f), stamped span %.1f us median, %.1f max",
                    ghz[ghz.size() / 2], ghz.size(), ghz.front(), ghz.back(), span[span.size() / 2], span.back());
        }
        fprintf(stderr, "\n");
        hipFree(pl->prof);
    }
}
