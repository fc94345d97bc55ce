// snmf_api.hip -- host side of libsnmf_hip.so: the C ABI declared in include/snmf.h.
//
// Orchestrates the kernels of snmf_kernels.h into the loop of the reference solver
// (lordet01/SE_SNMF_NAT src/sparse_nmf.m:157-292).  No CPU compute fallback exists here: every
// numeric step is a HIP kernel launch; without a device the entry points fail.
#define SNMF_AUX_KERNELS 1  // this translation unit launches k_reduce, k_wapply, k_check, ... (snmf_kernels.h)
#include "snmf_internal.h"
#include "snmf_frontend.h"
#include "snmf_generic.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
static const char* kFamNames[FAM_N] = {"hstep", "wstats", "wapply", "reduce", "wfin"};

static void drain_timers(snmf_ctx* c) {
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto& tp : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tp.a, tp.b) == hipSuccess) {
            c->fam_ms[tp.fam] += ms;
            c->fam_n[tp.fam] += 1;
        }
        hipEventDestroy(tp.a);
        hipEventDestroy(tp.b);
    }
    c->pending.clear();
}

extern "C" int snmf_abi_version(void) { return SNMF_ABI_VERSION; }
extern "C" const char* snmf_last_error(void) { return g_err.c_str(); }
extern "C" int snmf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int snmf_ctx_create(snmf_ctx** out, int device) {
    if (!out) return fail(SNMF_ERR_INVALID, "snmf_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(SNMF_ERR_NO_DEVICE, "no HIP device available (the engine has no CPU fallback)");
    if (device < 0 || device >= n) return fail(SNMF_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snmf_ctx* c = new snmf_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("SNMF_DEVCACHE_MB")) c->cache_cap = (size_t)std::max(0, atoi(e)) << 20;
    c->lds_max = prop.sharedMemPerBlock > 0 ? (size_t)prop.sharedMemPerBlock : 64 * 1024;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
            c->lds_max = std::max(c->lds_max, (size_t)v);
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(SNMF_ERR_NO_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
    *out = c;
    return SNMF_OK;
}

extern "C" int snmf_ctx_set_stream(snmf_ctx* c, void* s) {
    if (!c) return fail(SNMF_ERR_INVALID, "ctx is NULL");
    drain_timers(c);
    if (c->own_stream && c->stream) {
        hipStreamSynchronize(c->stream);
        hipStreamDestroy(c->stream);
    }
    if (s) {
        c->stream = (hipStream_t)s;
        c->own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return SNMF_OK;
}

extern "C" int snmf_ctx_sync(snmf_ctx* c) {
    if (!c) return fail(SNMF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SNMF_OK;
}

void* ctx_take(snmf_ctx* c, size_t bytes) {
    for (size_t i = 0; i < c->cache.size(); ++i)
        if (c->cache[i].second == bytes) {
            void* p = c->cache[i].first;
            c->cache.erase(c->cache.begin() + i);
            c->cache_bytes -= bytes;
            return p;
        }
    return nullptr;
}
void ctx_give(snmf_ctx* c, void* p, size_t bytes) {
    if (bytes > c->cache_cap) {
        hipFree(p);
        return;
    }
    while (c->cache_bytes + bytes > c->cache_cap || c->cache.size() >= 256) {  // oldest first
        hipFree(c->cache.front().first);
        c->cache_bytes -= c->cache.front().second;
        c->cache.erase(c->cache.begin());
    }
    c->cache.emplace_back(p, bytes);
    c->cache_bytes += bytes;
}

extern "C" void snmf_ctx_destroy(snmf_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    drain_timers(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& b : c->cache) hipFree(b.first);
    c->cache.clear();
    if (c->aux) snmf_ctx_destroy(c->aux);
    xfer_destroy(c);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int snmf_ctx_timing(snmf_ctx* c, int enable) {
    if (!c) return fail(SNMF_ERR_INVALID, "ctx is NULL");
    drain_timers(c);
    c->timing = enable != 0;
    if (enable) {
        for (int i = 0; i < FAM_N; ++i) {
            c->fam_ms[i] = 0;
            c->fam_n[i] = 0;
        }
    }
    return SNMF_OK;
}

extern "C" int snmf_ctx_timing_get(snmf_ctx* c, const char* family, double* avg_ms, int64_t* launches) {
    if (!c || !family) return fail(SNMF_ERR_INVALID, "NULL argument");
    drain_timers(c);
    for (int i = 0; i < FAM_N; ++i) {
        if (!strcmp(family, kFamNames[i])) {
            if (avg_ms) *avg_ms = c->fam_n[i] ? c->fam_ms[i] / (double)c->fam_n[i] : 0.0;
            if (launches) *launches = c->fam_n[i];
            return SNMF_OK;
        }
    }
    return fail(SNMF_ERR_INVALID, "unknown kernel family '%s'", family);
}

// ------------------------------------------------------------------------------------------
// plan

int validate_params(const snmf_params* p) {
    if (!p) return fail(SNMF_ERR_INVALID, "params is NULL");
    if (p->F <= 0 || p->T <= 0) return fail(SNMF_ERR_INVALID, "F and T must be positive (F=%d, T=%d)", p->F, p->T);
    if (p->r <= 0) return fail(SNMF_ERR_NO_INIT, "Number of components or initialization must be given");
    if (p->max_iter < 0) return fail(SNMF_ERR_INVALID, "max_iter must be >= 0");
    if (!(p->beta == p->beta)) return fail(SNMF_ERR_INVALID, "beta is NaN");
    if (p->sparsity_kind < 0 || p->sparsity_kind > 2) return fail(SNMF_ERR_INVALID, "bad sparsity_kind");
    if (p->cost_check != 0 && p->cost_check != 1)
        return fail(SNMF_ERR_NO_FIELD, "cost_check must be given as 0 or 1 (src/sparse_nmf.m:260 has no default)");
    return SNMF_OK;
}

#ifdef SNMF_PROF
#include "snmf_prof.h"
#endif

extern "C" void snmf_plan_destroy(snmf_plan* pl) {
    if (!pl) return;
    hipSetDevice(pl->ctx->device);
    hipStreamSynchronize(pl->ctx->stream);
#ifdef SNMF_PROF
    snmf_prof_report(pl);
#endif
    void* ptrs[] = {pl->V,     pl->H[0],  pl->H[1], pl->Wc,   pl->Wt4,   pl->Wk4,  pl->dphv, pl->colsum, pl->lamk,
                    pl->S,     pl->slabs, pl->spart, pl->part, pl->stats, pl->divh, pl->costh, pl->wn,    pl->st,
                    pl->w_ind, pl->staging, pl->wx, pl->Wcf, pl->M, pl->part_buf, pl->part_cnt,
                    pl->gLam,  pl->gR,    pl->gD,   pl->gNum, pl->gDen, pl->gram_slabs, pl->gram32};
    for (void* q : ptrs)
        if (q) {
            auto it = pl->blocks.find(q);
            if (it != pl->blocks.end()) ctx_give(pl->ctx, q, it->second);
            else hipFree(q);
        }
    delete pl;
}

static int hupd_parts(const snmf_plan* pl);  // (objective partials of an H-update launch; defined with the step API below)

extern "C" int snmf_plan_create(snmf_ctx* ctx, const snmf_params* p, snmf_plan** out) {
    if (!ctx || !out) return fail(SNMF_ERR_INVALID, "NULL argument");
    *out = nullptr;
    SN_TRY(validate_params(p));
    (void)hipGetLastError();  // start from a clean (sticky, per-thread) HIP error state, see PLAN_CHECK
    HIP_TRY(hipSetDevice(ctx->device));
    snmf_plan* pl = new snmf_plan();
    pl->ctx = ctx;
    pl->p = *p;
    pl->p.w_update_ind = nullptr;
    pl->p.h_update_ind = nullptr;
    const int F = p->F, T = p->T, r = p->r;
    // masks (src/sparse_nmf.m:142-148, :176-179)
    int n_h = 0, n_w = 0;
    pl->h_w_ind.assign(r, 1);
    for (int k = 0; k < r; ++k) {
        const bool hk = p->h_update_ind ? p->h_update_ind[k] != 0 : true;
        const bool wk = p->w_update_ind ? p->w_update_ind[k] != 0 : true;
        n_h += hk;
        n_w += wk;
        pl->h_w_ind[k] = wk;
    }
    if (n_h != 0 && n_h != r) {
        delete pl;
        // bsxfun(@plus, sum(w(:,h_ind))', p.sparsity) with sum(h_ind) ~= r rows: MATLAB size error
        return fail(SNMF_ERR_DIM,
                    "partial h_update_ind (%d of %d rows): dimension mismatch in src/sparse_nmf.m:192/197/202", n_h, r);
    }
    pl->upd_h = n_h > 0;
    pl->upd_w = n_w > 0;
    pl->bm = (p->beta == 1.0) ? BM_KL : (p->beta == 2.0 ? BM_EUC : BM_GEN);
    pl->n_mat = pl->bm == BM_KL ? 1 : 2;

    // row geometry (see snmf_kernels.h): F = 32*nf + 1 (257, 513, ...) -> extra-row mode
    pl->xr = (F % 32 == 1 && F > 32) ? 1 : 0;
    pl->nf = pl->xr ? F / 32 : (int)roundup(F, 32) / 32;
    pl->Fm = 32 * pl->nf;
    pl->Fp = pl->Fm + 4 * pl->xr;
    pl->Fq = pl->Fm + 8 * pl->xr;
    pl->rp = (int)roundup(r, 32);
    pl->Tp = (int)roundup(T + 32, 64);  // >= 32 zero columns of slack: a 32-frame tile may start at any frame
    pl->nk = pl->rp / 32;
    pl->ldh = pl->rp + 4;
    pl->ldr = pl->Fq + 4;
    // k_hstep geometry: prefer two 4-wave workgroups per CU on 32-frame tiles (their phases
    // de-synchronise and keep the matrix pipe fed); otherwise one 8-wave workgroup per CU on the
    // widest tile whose H image + ratio image fit the 160 KiB LDS.
    const size_t per_col = (size_t)(pl->ldh + pl->ldr) * 4;
    const size_t lds_cap = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may take all of it
    const size_t lds_extra = (size_t)pl->rp * 4 + 128;  // extra row of W + the roles' progress slots (6 signals x 4 waves) + the split tile's flag
    const size_t lds1 = 32 * per_col + lds_extra, lds2 = 64 * per_col + lds_extra;
    if (2 * lds1 - lds_extra <= lds_cap) { pl->NWH = 8; pl->NT = 1; pl->NLH = 4; }  // double-buffered
    else if (lds2 <= lds_cap && pl->Tp / 64 >= ctx->n_cu) { pl->NWH = 8; pl->NT = 2; }
    else if (lds1 <= lds_cap) { pl->NWH = 8; pl->NT = 1; }
    else if (16 * per_col + lds_extra <= lds_cap) {
        // the H image + ratio image of 32 frames do not fit (F + r > 1272, e.g. the reference's exemplar setting
        // R_x = R_d = 500 at F = 513, settings/bak_IS16_results/initial_setting_Exemplar.m:47-48): 16-frame tiles
        pl->NWH = 8; pl->NT = 1; pl->TTH = 16;
    } else {
        // F + r beyond what a 16-frame tile's H block + ratio image take of the LDS (~2540): the out-of-envelope path
        pl->generic = true;
        pl->NWH = 8; pl->NT = 1; pl->TTH = 16;
    }
    if (const char* e = getenv("SNMF_HSTEP_RP")) pl->hstep_rp = atoi(e) != 0;
    // A workgroup with a single tile has nothing to pipeline: the role pipelines' hand-offs then only add latency (C1,
    // 257 x 2000 r = 40, 63 tiles: k_hstep 16.9 us against k_hstep_rp 18.5), so such problems take the barrier-phased kernel
    if ((T + 31) / 32 <= ctx->n_cu) pl->hstep_rp = false;
    // F = 513 (9..16 row tiles): two whole tile buffers do not fit, but two H blocks + ONE ratio image do -- k_hstep_rh
    // pipelines on half tiles.  One pair of column tiles per wave of its P2 team: rp <= 256.
    pl->lds_rh = std::max<size_t>(((size_t)2 * 32 * pl->ldh + (size_t)32 * pl->ldr + pl->rp) * 4 + 160, 2 * kMaxNW * 64 * sizeof(double));
    pl->rh = pl->hstep_rp && pl->NLH != 4 && pl->bm == BM_KL && pl->nf >= 9 && pl->nf <= 16 && pl->rp <= 256 && pl->lds_rh <= lds_cap;
    // r = 97..100 on 16 row tiles (the reference's R = 100 at F = 513): P2 cut over the contraction, the 1..4 real columns of
    // the fourth column tile as 4x4x1 MFMAs on the same ratio fragments (k_hstep_rh<OBJ, LXH>); needs 52 KB more LDS for the waves' partial tiles
    {
        const size_t lx = pl->lds_rh + 4 * 3 * 1024 * 4 + 4 * 64 * 16;  // the B waves' partial tiles + partial leftover columns
        pl->rh_lxh = (pl->rh && pl->nf == 16 && pl->nk == 4 && r > 96 && r <= 100 && lx <= lds_cap) ? 1 : 0;
        if (pl->rh_lxh) pl->lds_rh = lx;
        // r = 193..200 on 16 row tiles (R_x + R_d = 200 at F = 513, run_basis_DNMF.m:40): seven column tiles -> the B waves work
        // in pairs over three full tiles each, cut in two over the contraction (k_hstep_rh<OBJ, 2>); 29 KB more LDS
        const size_t lx2 = pl->lds_rh + (size_t)(4 * 6 * 256 + 4 * 64 * 4 + pl->rp) * 4;
        const char* ec = getenv("SNMF_RH_CUT2");
        if (pl->rh && !pl->rh_lxh && pl->nf == 16 && pl->nk == 7 && r > 192 && r <= 200 && lx2 <= lds_cap && !(ec && atoi(ec) == 0)) {
            pl->rh_lxh = 2;
            pl->lds_rh = lx2;
        }
    }
    pl->lds_h = std::max<size_t>(pl->NLH ? 2 * lds1 - lds_extra : (pl->NT == 1 ? (size_t)pl->TTH * per_col + lds_extra : lds2),
                                 2 * kMaxNW * 64 * sizeof(double));
    // one or two column tiles (r <= 64) on the double-buffered role pipeline: P2 cut over the contraction (k_hstep_rp<., CUT>):
    // 32 KB of partial tiles + 1 ./ dph + two more signals behind the buffers
    {
        const char* e = getenv("SNMF_RP_CUT");
        const size_t more = 32 + (size_t)4 * pl->nk * 1024 * 4 + (size_t)pl->rp * 4;
        const bool shape_ok = pl->NLH == 4 && pl->NWH == 8 && pl->bm == BM_KL && pl->nk <= 2 && pl->nf >= 4 && !(e && atoi(e) == 0);
        // (rp_cut = 2: the PAIR form -- two column tiles, 8 KB of partials -- where the four-way form's 32 KB do not fit: 513 rows, r = 33..64)
        const size_t more2 = 32 + (size_t)2048 * 4 + (size_t)pl->rp * 4;
        pl->rp_cut = !shape_ok ? 0 : pl->lds_h + more <= lds_cap ? 1 : (pl->nk == 2 && pl->lds_h + more2 <= lds_cap) ? 2 : 0;
        if (pl->rp_cut) pl->lds_h += pl->rp_cut == 2 ? more2 : more;
    }
    pl->lds_mdi = std::max<size_t>(lds1, 2 * kMaxNW * 64 * sizeof(double));  // MDI pass: (NW=8, NT=1, NL=0)
    pl->grid_mdi = std::max(1, std::min(pl->Tp / 32, ctx->n_cu));
    const int n_tiles_h = pl->Tp / (pl->TTH * pl->NT);
    // without loaders the NT == 1 kernels are register-bounded for two workgroups per CU
    int wg_per_cu = (pl->lds_h * 2 <= lds_cap && pl->NT == 1 && !pl->NLH) ? 2 : 1;
    pl->grid_h = std::max(1, std::min(n_tiles_h, ctx->n_cu * wg_per_cu));
    // k_hstep_rp: only tiles that hold a frame (the pad tiles of both H buffers are zero and stay zero), and the last
    // PARTIAL round split by rows over the workgroups that would idle through it (snmf_kernels.h, "the split last round"):
    // 4 parts per tile when 4 * (tiles of that round) workgroups exist, else 2, else the round stays whole.
    // SNMF_HSTEP_SPLIT=0 keeps every tile in the pipeline (tests compare the two).
    {
        const int G = ctx->n_cu;
        pl->rp_tiles = (T + 31) / 32;
        pl->rp_full = pl->rp_tiles;
        pl->rp_grid = std::max(1, std::min(pl->rp_tiles, G));  // (k_hstep_rh launches on the same grid)
        pl->rp_S = 0;
        const char* e = getenv("SNMF_HSTEP_SPLIT");
        // (k_hstep_rh splits by CONTIGUOUS row tiles within a half: 16 row tiles only, F = 505..513)
        if ((pl->NLH == 4 || (pl->rh && pl->nf == 16)) && !pl->rp_cut && !(e && atoi(e) == 0)) {
            // (only a partial round BEHIND whole ones: a problem of fewer tiles than workgroups is latency-bound, and there
            //  the split's extra steps -- partial stores, the arrival counter, the finishing pass -- cost more than the
            //  shorter MFMA loops save: C1, 257 x 2000 r = 40, ran 23.3 k iterations/s split against 26.7 k whole)
            const int full = (pl->rp_tiles / G) * G, R = pl->rp_tiles - full;
            int S = (full > 0 && R > 0) ? (4 * R <= G ? 4 : (2 * R <= G ? 2 : 0)) : 0;
            while (S > pl->nf) S >>= 1;
            if (S >= 2) {
                pl->rp_S = S;
                pl->rp_full = full;
                pl->rp_grid = G;
            }
        }
    }
    // k_hstep_m (merged roles: one wave per SIMD runs P1, both epilogues and P2 of its own rows / columns, snmf_hstep_m.h): the
    // double-buffered geometry with exactly 8 row tiles and 8 column tiles (C2).  SNMF_HSTEP_M=1 selects it (A/B against k_hstep_rp).
    // An EXPERIMENT (15 % slower than k_hstep_rp, kept as the vehicle of the counters in profiles/r05_experiments.md section 1): only
    // in builds with -DSNMF_EXPERIMENTS (SNMF_EXPERIMENTS=1 python scripts/build_variant.py exp), whose sources then include
    // csrc/experiments/; the product library has neither the kernel nor the switch.
#ifdef SNMF_EXPERIMENTS
    {
        const char* e = getenv("SNMF_HSTEP_M");
        pl->hm = e && atoi(e) != 0 && pl->hstep_rp && pl->NWH == 8 && pl->NLH == 4 && pl->bm == BM_KL && pl->nf == 8 && pl->nk == 8 && !pl->generic;
        pl->hm_grid = std::max(1, std::min(pl->rp_tiles, ctx->n_cu));
    }
#endif
    // At most two row tiles and eight column tiles (the Mel solves, r <= 256): a tile per WAVE, nothing handed between waves
    // (snmf_smallf.h).  Follows SNMF_HSTEP_RP (tests compare against the barrier-phased kernels); SNMF_HSTEP_SF=0 keeps the
    // role pipeline.
    {
        const char* e = getenv("SNMF_HSTEP_RP");
        const char* e2 = getenv("SNMF_HSTEP_SF");
        pl->lds_sf = ((size_t)pl->nf * pl->rp * 32 + (size_t)pl->nk * pl->Fq * 32 + 2 * (size_t)pl->rp) * 4 + 2 * 8 * sizeof(double);
        pl->sf = pl->bm == BM_KL && pl->upd_h && !pl->xr && pl->nf <= 2 && pl->nk <= 8 && pl->lds_sf <= lds_cap && !pl->generic &&
                 !(e && atoi(e) == 0) && !(e2 && atoi(e2) == 0);
        pl->sf_grid = std::max(1, std::min((T + 31) / 32, ctx->n_cu));
        // the shared last tile: when the partial wave level behind the whole ones is the FIRST on its SIMDs (level 0 or 4 of 8: any other
        // level runs beside whole tiles of the same round on other SIMDs and sharing it would not end the launch earlier).  SNMF_HSTEP_SPLIT=0
        // keeps every tile whole, like the split last round of k_hstep_rp.
        {
            const int nw = 8 * pl->sf_grid, R = pl->rp_tiles % nw, wfull = R / pl->sf_grid, xb = R % pl->sf_grid;
            const char* e4 = getenv("SNMF_HSTEP_SPLIT");
            const size_t more = (size_t)pl->nf * 16384 + 16;
            pl->sf_share = 0;
            pl->sf_nfull = pl->rp_tiles;
            // (not for the full updates that snmf_plan_run fuses into k_iter_sf: the step API's two launches -- the sharded loop -- stay
            //  bit for bit what the fused launch computes with every tile whole, tests/test_gpu_parity.py::test_fused_small_f_iteration_equals_the_two_launches;
            //  k_iter_sf shares a chunk's remainder tile in its own way, isf_share)
            const bool isf_shape = pl->upd_h && pl->upd_w && pl->nf == 2 && pl->nk >= 3 && pl->nk <= 4;
            if (pl->sf && !isf_shape && xb > 0 && (wfull == 0 || wfull == 4) && pl->nk >= 2 && pl->lds_sf + more <= lds_cap && !(e4 && atoi(e4) == 0)) {
                pl->sf_share = xb;
                pl->sf_nfull = pl->rp_tiles - xb;
                pl->lds_sf += more;
            }
        }
        pl->sf_stagger = 8000;
        if (const char* e3 = getenv("SNMF_SF_STAG")) pl->sf_stagger = atoi(e3);
    }
    // k_wstats geometry: 4-wave workgroups, each wave owns one 32-row tile x NKT 32-column tiles of
    // the statistics in registers.  NKT <= 8 (128 accumulator VGPRs): two workgroups per CU.
    if (pl->nk <= 4) { pl->NKT = 4; pl->WPS = 2; }
    else if (pl->nk <= 8) { pl->NKT = 8; pl->WPS = 2; }
    else { pl->NKT = 16; pl->WPS = 1; }
    // (eight consumer waves, two per SIMD, for narrow statistics over at least eight row tiles: see the kernel)
    pl->NWB = (pl->NKT == 4 && pl->nf >= 8) ? 8 : 4;
    pl->n_kg = (pl->nk + pl->NKT - 1) / pl->NKT;
    pl->n_fg = (pl->nf + pl->NWB - 1) / pl->NWB;
    // H image of k_wstats: rows padded to whole NKT-tile groups (branch-free P4, see the kernel)
    pl->ldhw = std::max(pl->rp, 32 * pl->NKT * pl->n_kg) + 4;  // (NKT = 4: always 132 -- k_wstats<4, ...> has it as a compile-time constant)
    if (((size_t)32 * pl->ldhw + (size_t)32 * 32 * pl->NWB) * 4 + (size_t)pl->rp * 4 + 320 > lds_cap && pl->NKT == 16)
        pl->TTW = 16;  // large r: 16-frame tiles (the 32-frame H + V images do not fit the LDS)
    const int n_tiles_w = (T + pl->TTW - 1) / pl->TTW;  // tiles that hold a frame (an all-padding tile adds exact zeros: skipped)
    {
        // loaders + double buffering when the accumulators allow 2 waves per SIMD and LDS has room; the loader waves
        // stage only the row group's 32 * NWB columns of V (the kernel's ldv), so F = 513 fits as well
        const size_t buf_ld = ((size_t)pl->TTW * pl->ldhw + (size_t)pl->TTW * 32 * pl->NWB) * 4;
        pl->NLW = (pl->WPS == 2 && 2 * buf_ld + (size_t)pl->rp * 4 + 512 + (size_t)pl->NWB * std::min(pl->rp, 256) * 4 <= lds_cap) ? 4 : 0;
        // (loader waves pay from the second tile of a workgroup on; with one tile each -- C1: 63 tiles -- the synchronous
        //  4-wave geometry is faster: 12.7 us against 16.4)
        if (n_tiles_w <= ctx->n_cu / std::max(1, ((pl->nf + 3) / 4) * pl->n_kg)) pl->NLW = 0;
        if (const char* e = getenv("SNMF_WSTATS_NL")) pl->NLW = (atoi(e) == 4 && pl->NLW == 4) ? 4 : 0;
        size_t buf = buf_ld;
        if (!pl->NLW && pl->NWB == 8) {  // the eight-consumer geometry exists with loader waves only
            pl->NWB = 4;
            pl->n_fg = (pl->nf + pl->NWB - 1) / pl->NWB;
            buf = ((size_t)pl->TTW * pl->ldhw + (size_t)pl->TTW * 32 * pl->NWB) * 4;
        }
        // (the fixed-order sums at the end of the kernel use [4][rp] floats / one double per thread of the same memory)
        // (+ ready/done slots + the extra row's V values [2][32] + the consumers' partial extra rows of the slab, NK <= 8)
        const size_t gxs = pl->NKT <= 8 ? (size_t)pl->NWB * std::min(pl->rp, 256) * 4 : 0;
        // a THIRD tile buffer where it fits (r <= 128 at F = 513, C2's geometry): the loader waves then stage two tiles ahead
        // and the consumers stop waiting for `ready` (k_wstats; SNMF_WSTATS_NBUF=2 keeps two)
        const size_t tail = (size_t)pl->rp * 4 + 512 + gxs;  // extra row of W, progress slots + the extra row's V values, gxs
        pl->nbw = pl->NLW ? 2 : 1;
        if (pl->NLW && 3 * buf_ld + tail <= lds_cap) pl->nbw = 3;
        if (const char* e = getenv("SNMF_WSTATS_NBUF")) if (pl->NLW && atoi(e) == 2) pl->nbw = 2;
        pl->lds_w = std::max<size_t>(std::max<size_t>((pl->NLW ? pl->nbw * buf_ld : buf) + tail,
                                                      (size_t)std::max(4, pl->NWB) * pl->rp * 4),
                                     (size_t)(pl->NWB + pl->NLW) * 64 * sizeof(double));
    }
    // Fewer row tiles than consumer waves (F = 64, a Mel spectrogram: two): the consumer waves form teams that take the chunk's
    // tiles in turn (StepArgs::til) instead of leaving half the SIMDs without an MFMA wave.  Needs the loader geometry (the
    // teams' partial statistics meet in the tile buffers at the end), one row group, one kappa-group, no extra row.
    pl->til = 1;
    if (pl->NLW && pl->TTW == 32 && !pl->xr && pl->n_fg == 1 && pl->n_kg == 1 && pl->upd_w && pl->NKT == 4 && pl->bm == BM_KL) {
        int til = 1;
        while (til * 2 * pl->nf <= pl->NWB) til *= 2;
        const size_t per_wave = (size_t)(pl->NKT * 16 + 8) * 64 * 4;
        while (til > 1 && (size_t)(til - 1) * (pl->NWB / til) * per_wave > pl->lds_w) til /= 2;
        if (const char* e = getenv("SNMF_WSTATS_TIL")) if (atoi(e) == 1) til = 1;
        pl->til = til;
    }
    // ... and the KL statistics of the same shapes (r <= 128) through k_wstats_sf; follows SNMF_WSTATS_NL (tests compare against
    // the synchronously staging kernels); SNMF_WSTATS_SF=0 keeps k_wstats_teams
    {
        const char* e = getenv("SNMF_WSTATS_NL");
        const char* e2 = getenv("SNMF_WSTATS_SF");
        const int ncl = 8 / std::max(1, pl->nf);
        pl->lds_wsf = ((size_t)(ncl - 1) * pl->nf * pl->nk * 1024 + (size_t)ncl * pl->rp) * 4 + 8 * sizeof(double) + 64;
        pl->wsf = pl->bm == BM_KL && pl->upd_w && !pl->xr && pl->nf <= 2 && pl->nk <= 4 && pl->TTW == 32 && pl->n_kg == 1 && pl->n_fg == 1 &&
                  pl->NLW && !pl->generic && pl->lds_wsf <= lds_cap && !(e && atoi(e) == 0) && !(e2 && atoi(e2) == 0);
        if (pl->wsf) pl->til = 1;
        const char* e4 = getenv("SNMF_HSTEP_SPLIT");
        pl->wsf_share = pl->wsf && pl->nf == 2 && pl->nk >= 2 && !(pl->upd_h && pl->nk >= 3) && !(e4 && atoi(e4) == 0);  // (not the shapes of k_iter_sf: see sf_share)
    }
    // ... and, for FULL updates of those shapes, both half-steps in one launch (k_iter_sf; the run loop only: the step API keeps
    // the two launches, between which a multi-rank caller sums nothing but could).  SNMF_ITER_SF=0 keeps two launches.
    {
        const char* e = getenv("SNMF_ITER_SF");
        const int ncl = 4;  // SIMD pairs (H wave + W wave) per workgroup = chunk lanes of k_wstats_sf at two row tiles
        // (images, 1 ./ dph and lambda, ncl + 1 hand-off buffers, 16 progress words, the H waves' fp64 objective sums and the W waves' row sums per lane)
        const size_t body = ((size_t)pl->nf * pl->rp * 32 + (size_t)pl->nk * pl->Fq * 32 + 2 * (size_t)pl->rp + (size_t)(ncl + 1) * 32 * (32 * pl->nk + 4) + 16) * 4 + (size_t)ncl * 64 * 2 * sizeof(double) + (size_t)ncl * 64 * 4 * 4;
        const size_t tail = ((size_t)ncl * pl->nf * pl->nk * 1024 + (size_t)ncl * pl->rp) * 4 + 2 * ncl * sizeof(double);
        pl->lds_isf = std::max(body, tail) + 64;
        pl->isf = pl->sf && pl->wsf && pl->nf == 2 && pl->nk >= 3 && pl->upd_h && pl->upd_w && pl->lds_isf <= lds_cap && !(e && atoi(e) == 0);
        const char* e4 = getenv("SNMF_HSTEP_SPLIT");
        pl->isf_share = pl->isf && !(e4 && atoi(e4) == 0);
    }
    // Small rank on tall spectrograms (r <= 32 on 3..16 row tiles; the reference's R = 20 / 10 / 30 at F = 513): a tile per workgroup
    // cut by ROW TILES over its eight waves, every operand straight into the MFMA layouts (snmf_smallr.h).  Follow SNMF_HSTEP_RP /
    // SNMF_WSTATS_NL like the other fast paths (tests compare against the plain kernels); SNMF_HSTEP_SR=0 / SNMF_WSTATS_SR=0 keep the role pipelines.
    {
        const char* e = getenv("SNMF_HSTEP_SR");
        const char* e2 = getenv("SNMF_WSTATS_SR");
        const char* e3 = getenv("SNMF_WSTATS_NL");
        const bool shape = pl->bm == BM_KL && pl->nf >= 3 && pl->nf <= 16 && pl->nk == 1 && !pl->generic && pl->TTW == 32 && pl->TTH == 32;
        pl->lds_sr = sr_hstep_lds_bytes(pl->nk, pl->Fq, pl->rp);
        pl->sr = shape && pl->upd_h && pl->hstep_rp && pl->lds_sr <= lds_cap && !(e && atoi(e) == 0);
        pl->sr_grid = std::max(1, std::min((T + 31) / 32, ctx->n_cu));
        if (const char* e4 = getenv("SNMF_SR_STAG")) pl->sr_stagger = atoi(e4);
        pl->lds_wsr = sr_wstats_lds_bytes(pl->rp);
        pl->wsr = shape && pl->upd_w && pl->NLW && !(e2 && atoi(e2) == 0) && !(e3 && atoi(e3) == 0);
        if (pl->wsr) {  // one workgroup per frame chunk carries every row tile: no row groups
            pl->n_fg = 1;
            pl->n_kg = 1;
            pl->til = 1;
        }
    }
    const int wg_w = pl->NLW ? 1 : pl->WPS;  // workgroups per CU
    pl->n_chunks = std::max(1, std::min(n_tiles_w, ctx->n_cu * wg_w / std::max(1, pl->n_fg * pl->n_kg)));
    // Two row groups, only group 0 carries the extra row: deal the workgroups out so that both finish together.
    // Relative cost x of the extra row per tile: ~2.1 k cycles at rp = 256 against 21 k for the two MFMA loops (phase
    // stamps; a sweep of the split point on C2 had its optimum where this x puts it: 131..135 chunks for group 0,
    // k_wstats 0.2573 -> 0.2481 ms, profiles/r02_experiments.md).
    pl->n_ch1 = 0;
    if (pl->xr && pl->n_fg >= 2 && pl->n_kg == 1 && pl->NLW && pl->upd_w && n_tiles_w >= 4 * pl->n_chunks) {
        const int tot = pl->n_fg * pl->n_chunks, ng1 = pl->n_fg - 1;  // group 0: n0 workgroups, every other group n1
        // (the row is shared by the group's NWB waves.  Round 4 re-measured it with phase stamps at 513 x 72000, r = 100 -- a group-0
        //  tile takes 12 % longer per wave -- and swept x over 0.065 .. 0.22: the split this model picks (x = 0.065 there) is within
        //  0.5 % of the best one, larger x loses 4 % to the tile-count quantisation; SNMF_WSTATS_X overrides it for such sweeps)
        double x = (600.0 + 6.0 * pl->rp) / (82.0 * (pl->rp / 2 + 16 * pl->nk)) * 4.0 / pl->NWB;
        if (const char* e = getenv("SNMF_WSTATS_X")) x = atof(e);  // (experiment: relative cost of the extra row per tile)
        auto n1_of = [&](int n0) { return (tot - n0) / ng1; };
        auto cost = [&](int n0) {
            return std::max(std::ceil((double)n_tiles_w / n0) * (1.0 + x), std::ceil((double)n_tiles_w / n1_of(n0)));
        };
        int best = pl->n_chunks;
        for (int n0 = pl->n_chunks + 1; n0 <= pl->n_chunks + pl->n_chunks / 4 && n1_of(n0) >= 1; ++n0)
            if (cost(n0) < cost(best) - 1e-9) best = n0;
        if (best != pl->n_chunks) {
            pl->n_ch1 = n1_of(best);
            pl->n_chunks = best;
        }
    }
    // start-up stagger (cycles) of the second half of each grid: about half a tile period when two
    // workgroups share a CU.
    {
        const int mf_h = (pl->nf + pl->NWH - 1) / pl->NWH * pl->NT * (pl->rp / 2) +
                         (pl->nk + pl->NWH - 1) / pl->NWH * pl->NT * (pl->Fq / 2);
        pl->stagger_h = (pl->grid_h > ctx->n_cu) ? mf_h * 64 : 0;
        const int mf_w = pl->rp / 2 + 16 * pl->NKT;
        pl->stagger_w = 0;
        if (const char* e = getenv("SNMF_WSTAG")) pl->stagger_w = atoi(e);  // (experiment: intra-workgroup stagger of k_wstats' second consumer wave per SIMD, cycles)
        (void)mf_w;
    }
    if (pl->lds_w > lds_cap && pl->upd_w) pl->generic = true;  // r too large for k_wstats' H image
    if (pl->bm == BM_EUC && pl->NKT == 16 && pl->TTW == 32 && pl->upd_w) {
        // Q = V * H^T of the Euclidean W step needs no Lam', so nothing is recomputed when the statistics' columns are cut
        // into 256-wide kappa-groups: the NK = 16 geometry (256 accumulator registers, no room for loader waves, 116
        // spilled VGPRs) is replaced for this launch by <8,4,4,2> with double-buffered LDS-DMA staging
        pl->kq_kg = (pl->nk + 7) / 8;
        pl->kq_chunks = std::max(1, std::min(std::min(n_tiles_w, pl->n_chunks), ctx->n_cu / std::max(1, pl->n_fg * pl->kq_kg)));
        pl->kq_lds = (size_t)2 * 32 * (260 + 32 * 4) * 4 + (size_t)pl->rp * 4 + 512 + (size_t)4 * 256 * 4;
    }
    // Euclidean full updates: P through the Gram matrix (launch_gram_p) wherever it is the cheaper form (2 r^2 T against
    // 4 F T r; W-only solves take their objective from the P launch's Lam' and keep it)
    // SNMF_GRAM_P=0 opts out: P = max(W*H, flr)*H' is then formed from the Lam' pass exactly as src/sparse_nmf.m:228-233 writes
    // it (the two forms differ only where W*H sits below the 1e-9 floor, by at most flr * sum(h) per entry: include/snmf.h)
    const char* gp_env = getenv("SNMF_GRAM_P");
    if (pl->bm == BM_EUC && pl->upd_w && pl->upd_h && pl->TTW == 32 && r < 2 * F && !(gp_env && atoi(gp_env) == 0)) {
        pl->gram_p = true;
        const int nfg_g = (pl->rp / 32 + pl->NWB - 1) / pl->NWB;
        pl->gram_chunks = pl->kq_kg ? pl->kq_chunks
                                    : std::max(1, std::min(n_tiles_w, ctx->n_cu * (pl->NLW ? 1 : pl->WPS) / std::max(1, nfg_g)));
    }
    // k_wstats keeps the row sums of H (KL) and the extra row of the slab (F = 32n+1) in per-thread registers: 1024 columns
    if (pl->rp > 4 * pl->NWB * 64 && pl->upd_w && (pl->bm == BM_KL || pl->xr)) pl->generic = true;

    pl->lds_wfin = (size_t)9 * pl->n_mat * pl->Fp * sizeof(double);
    pl->wfin = pl->upd_w && pl->lds_wfin + 12 * 1024 <= lds_cap;
    // very few columns (r <= 32: the reference's R = 20 / 10 / 30): cut every column's rows into slices, a workgroup each (k_wfin,
    // gridDim.y): r * S workgroups of at least eight 16-byte cells each.  Measured (513 x 72000): r = 10 W-only 13 824 -> 14 474 it/s,
    // r = 20 7 615 -> 7 744; from r = 100 up the gather costs what the wider read saves (a11 17.3 -> 17.4 us, Mel 11.3 -> 12.5), so
    // those keep one workgroup per column.  SNMF_WFIN_SPLIT=0: never split.
    {
        const char* e = getenv("SNMF_WFIN_SPLIT");
        int S = r <= 32 ? std::max(1, std::min(8, ctx->n_cu / std::max(1, r))) : 1;
        while (S > 1 && (pl->Fp / 4 + S - 1) / S < 8) --S;
        pl->wfin_S = (pl->wfin && !(e && atoi(e) == 0)) ? S : 1;
    }

    // persistent single-launch path for the online shape (H-only, at most one 32-frame tile)
    {
        const size_t need = ((size_t)32 * (pl->ldh + pl->ldr) + ((pl->rp + 3) & ~3)) * 4 + 2 * 512 * sizeof(double);
        pl->small_ok = pl->upd_h && !pl->upd_w && need <= lds_cap;   // shape admits the persistent kernel
        // SNMF_NO_SMALL (tests): 1 = no persistent kernel at all (the plan loop), 2 = no register-resident frame kernel
        const char* ns = getenv("SNMF_NO_SMALL");
        const int no_small = ns ? atoi(ns) : 0;
        pl->small = pl->small_ok && T <= 32 && no_small != 1;
        pl->lds_small = need;
        // one frame per solve: register-resident dictionary (k_hsolve_frame), F <= 64*FB + 1, r <= 8*KB
        if (pl->small_ok && no_small == 0) {
            static const int fbs[2] = {4, 8}, kbs[2] = {16, 25};
            for (int fi = 0; fi < 2 && !pl->frame_fb; ++fi)
                for (int ki = 0; ki < 2 && !pl->frame_fb; ++ki)
                    if (F <= 64 * fbs[fi] + 1 && r <= 8 * kbs[ki]) {
                        pl->frame_fb = fbs[fi];
                        pl->frame_kb = kbs[ki];
                    }
            if (pl->frame_fb) {
                const int Fm2 = 64 * pl->frame_fb, RB = 8 * pl->frame_kb, nv = pl->bm == BM_KL ? 1 : 2;
                pl->lds_frame = (size_t)(40 + 4 * RB + 3 * (Fm2 + 4) + 8 * Fm2 + nv * 16 * (RB + 1)) * 4;
                if (pl->lds_frame > lds_cap) pl->frame_fb = pl->frame_kb = 0;
            }
        }
    }
    if (pl->generic) {
        // none of the fused geometries applies; contractions over the frames are split into chunks of kGChunkT frames,
        // whose slabs k_reduce adds like the fast path's
        pl->hstep_rp = pl->rh = false;
        pl->rh_lxh = 0;
        pl->rp_S = 0;
        pl->kq_kg = 0;
        pl->gram_p = false;
        pl->n_ch1 = 0;
        pl->small_ok = pl->small = false;
        pl->frame_fb = pl->frame_kb = 0;
        pl->wfin = false;
        pl->n_fg = pl->n_kg = 1;
        pl->n_chunks = (T + kGChunkT - 1) / kGChunkT;
        pl->grid_h = pl->grid_mdi = kGBlocks;
    }
    // allocations
    const size_t nV = (size_t)pl->Fp * pl->Tp, nH = (size_t)pl->rp * pl->Tp, nW = (size_t)pl->Fp * pl->rp;
    const size_t nWt = (size_t)pl->Fm * pl->rp, nWk = (size_t)pl->Fq * pl->rp;
    int s = SNMF_OK;
    auto A = [&](int st) { if (s == SNMF_OK) s = st; };
    // (through the context's block cache; the plan remembers the sizes so that snmf_plan_destroy can hand the blocks back)
    auto palloc = [&](auto** p, size_t n) {
        size_t bytes = 0;
        const int st = dalloc(p, n, ctx, &bytes);
        if (st == SNMF_OK) pl->blocks[(void*)*p] = bytes;
        return st;
    };
    A(palloc(&pl->V, nV));
    A(palloc(&pl->H[0], nH));
    A(palloc(&pl->H[1], nH));
    A(palloc(&pl->Wc, nW));
    A(palloc(&pl->Wcf, nW));
    A(palloc(&pl->Wt4, nWt));
    A(palloc(&pl->Wk4, nWk));
    A(palloc(&pl->wx, (size_t)pl->rp));
    A(palloc(&pl->dphv, (size_t)pl->rp));
    A(palloc(&pl->colsum, (size_t)pl->rp));
    A(palloc(&pl->lamk, (size_t)pl->rp));
    if (p->sparsity_kind == SNMF_SPARSITY_FULL) A(palloc(&pl->S, nH));
    {
        const char* e = getenv("SNMF_HFOLD");
        pl->fold_obj = pl->upd_h && !pl->upd_w && !pl->generic && !(e && atoi(e) == 0);
    }
    if (pl->upd_w) {
        A(palloc(&pl->slabs, (size_t)pl->n_chunks * pl->n_mat * nW));
        A(palloc(&pl->spart, (size_t)pl->n_chunks * pl->rp));
        if (pl->wfin_S > 1) {
            A(palloc(&pl->qp_buf, (size_t)r * pl->n_mat * pl->Fp));
            A(palloc(&pl->fin_cnt, (size_t)r));
        }
    }
    if (pl->gram_p) {
        A(palloc(&pl->gram_slabs, (size_t)pl->gram_chunks * pl->rp * pl->rp));
        A(palloc(&pl->gram32, (size_t)pl->rp * pl->rp));
    }
    if (pl->generic) {
        A(palloc(&pl->gLam, nV));
        A(palloc(&pl->gR, nV));
        if (pl->bm != BM_KL) A(palloc(&pl->gD, nV));
        if (pl->upd_h) {
            A(palloc(&pl->gNum, nH));
            if (pl->bm != BM_KL) A(palloc(&pl->gDen, nH));
        }
    }
    if (pl->rp_S) {
        A(palloc(&pl->part_buf, (size_t)pl->rp_grid * 32 * pl->rp));
        A(palloc(&pl->part_cnt, (size_t)(pl->rp_tiles - pl->rp_full)));
    }
    pl->n_part = std::max(std::max(pl->grid_h, pl->grid_mdi), pl->n_chunks * pl->n_fg);
    pl->n_part = std::max(pl->n_part, pl->rp_grid);
    pl->n_part = std::max(pl->n_part, 1024);
    if (pl->generic) pl->n_part = kGBlocks + 256;  // (+ the slots of k_sum_sh behind the Lam pass's)
    A(palloc(&pl->part, (size_t)2 * pl->n_part));
    A(palloc(&pl->stats, (size_t)pl->n_mat * nW + pl->rp + 2));
    A(palloc(&pl->divh, (size_t)std::max(1, p->max_iter)));
    A(palloc(&pl->costh, (size_t)std::max(1, p->max_iter)));
    A(palloc(&pl->wn, (size_t)pl->rp));
    A(palloc(&pl->st, (size_t)1 + (sizeof(FoldBlock) + sizeof(DevState) - 1) / sizeof(DevState)));  // DevState, then the FoldBlock
    A(palloc(&pl->w_ind, (size_t)pl->rp));
#ifdef SNMF_PROF
    A(palloc(&pl->prof, (size_t)1024 * 8 * 12 + 3 * 8192 + 16384));  // phase slots, then (cycles, 100 MHz ticks) and start tick per wave
    hipMemset(pl->prof, 0, ((size_t)1024 * 8 * 12 + 3 * 8192 + 16384) * 8);
#endif
    if (s != SNMF_OK) {
        snmf_plan_destroy(pl);
        return s;
    }
    hipStream_t st = ctx->stream;
    hipMemsetAsync(pl->Wc, 0, nW * 8, st);
    hipMemsetAsync(pl->Wcf, 0, nW * 4, st);
    hipMemsetAsync(pl->Wt4, 0, nWt * 4, st);
    hipMemsetAsync(pl->Wk4, 0, nWk * 4, st);
    hipMemsetAsync(pl->wx, 0, (size_t)pl->rp * 4, st);
    if (pl->part_cnt) hipMemsetAsync(pl->part_cnt, 0, (size_t)(pl->rp_tiles - pl->rp_full) * 4, st);
    if (pl->fin_cnt) hipMemsetAsync(pl->fin_cnt, 0, (size_t)r * 4, st);
    if (pl->slabs) hipMemsetAsync(pl->slabs, 0, (size_t)pl->n_chunks * pl->n_mat * nW * 4, st);
    hipMemsetAsync(pl->H[0], 0, nH * 4, st);
    hipMemsetAsync(pl->H[1], 0, nH * 4, st);
    hipMemsetAsync(pl->colsum, 0, pl->rp * 4, st);
    hipMemsetAsync(pl->stats, 0, ((size_t)pl->n_mat * nW + pl->rp + 2) * 8, st);
    hipMemsetAsync(pl->w_ind, 0, pl->rp, st);
    hipMemcpyAsync(pl->w_ind, pl->h_w_ind.data(), r, hipMemcpyHostToDevice, st);
    hipMemsetAsync(pl->st, 0, sizeof(DevState), st);  // solve_frames never goes through snmf_plan_init
    pl->fold_host = FoldBlock{0u, hupd_parts(pl), p->conv_eps, pl->stats + (size_t)pl->n_mat * pl->rp * pl->Fp + pl->rp, pl->divh, pl->costh};
    hipMemcpyAsync(pl->st + 1, &pl->fold_host, sizeof(FoldBlock), hipMemcpyHostToDevice, st);
    hipMemsetAsync(pl->divh, 0, sizeof(double) * std::max(1, p->max_iter), st);
    hipMemsetAsync(pl->costh, 0, sizeof(double) * std::max(1, p->max_iter), st);
    {
        std::vector<float> lk(pl->rp, 0.f);
        if (p->sparsity_kind == SNMF_SPARSITY_SCALAR)
            for (int k = 0; k < r; ++k) lk[k] = (float)p->sparsity_scalar;
        hipMemcpyAsync(pl->lamk, lk.data(), pl->rp * 4, hipMemcpyHostToDevice, st);
        // pad rows of H divide by dphv: keep it positive there (0 * acc / 1 = 0, never 0/0)
        std::vector<float> ones(pl->rp, 1.f);
        hipMemcpyAsync(pl->dphv, ones.data(), pl->rp * 4, hipMemcpyHostToDevice, st);
        hipStreamSynchronize(st);
    }
    pl->have_s = p->sparsity_kind == SNMF_SPARSITY_SCALAR;
    *out = pl;
    return SNMF_OK;
}

extern "C" int64_t snmf_plan_stats_len(const snmf_plan* pl) {
    if (!pl) return 0;
    return (int64_t)pl->n_mat * pl->Fp * pl->rp + pl->rp + 2;
}

extern "C" int snmf_plan_describe(const snmf_plan* pl, char* buf, size_t n) {
    if (!pl || !buf) return fail(SNMF_ERR_INVALID, "NULL argument");
    const bool kl_pipe = pl->NWH == 8 && pl->NLH == 4 && pl->bm == BM_KL && pl->upd_h && !pl->M && pl->hstep_rp;
    char hs[256];
    const bool rh_pipe = pl->rh && pl->upd_h && !pl->M;
    const bool sf_pipe = pl->sf && !pl->M;
    const bool sr_pipe = pl->sr && !pl->M;
    if (sr_pipe)
        snprintf(hs, sizeof hs, "k_hstep_sr (a tile per workgroup cut by row tiles over 8 waves, operands straight into the MFMA layouts, partial numerators meet in LDS; %d tiles, grid %d)", pl->rp_tiles, pl->sr_grid);
    else if (sf_pipe && pl->isf && pl->wfin)
        snprintf(hs, sizeof hs, "k_iter_sf (H step + W statistics of a full update in ONE launch, 4 SIMD pairs of an H wave and a W wave per workgroup%s; %d tiles, grid %d; step API: k_hstep_sf)", pl->isf_share ? ", a chunk's single remainder tile shared by the four pairs" : "", pl->rp_tiles, pl->n_chunks);
    else if (sf_pipe)
        snprintf(hs, sizeof hs, "k_hstep_sf (a tile per wave from first load to last store, 8 waves per workgroup; %d tiles, the last %d shared by four waves each, grid %d)", pl->rp_tiles, pl->sf_share, pl->sf_grid);
    else if (rh_pipe)
        snprintf(hs, sizeof hs, "k_hstep_rh (4 P1 + 4 P2 + 4 loader waves on half tiles%s; %d of %d tiles pipelined, last round split %d ways, grid %d)",
                 pl->rh_lxh == 1 ? ", P2 cut four ways over the contraction + leftover columns as 4x4x1 MFMAs" : (pl->rh_lxh == 2 ? ", P2 in wave pairs cut over the contraction + leftover columns as 4x4x1 MFMAs" : ""), pl->rp_full, pl->rp_tiles, pl->rp_S, pl->rp_grid);
    else if (kl_pipe && pl->hm && !pl->S)
        snprintf(hs, sizeof hs, "k_hstep_m (merged roles: 4 waves, one per SIMD, each P1 + P2 of its own row / column tiles; %d tiles whole, grid %d)",
                 pl->rp_tiles, pl->hm_grid);
    else if (kl_pipe)
        snprintf(hs, sizeof hs, "k_hstep_rp (4 P1 + 4 P2 + 4 loader waves%s; %d of %d tiles pipelined, last round split %d ways, grid %d)",
                 pl->rp_cut == 2 ? ", P2 in wave pairs cut over the contraction" : pl->rp_cut ? ", P2 cut four ways over the contraction" : "", pl->rp_full, pl->rp_tiles, pl->rp_S, pl->rp_grid);
    else
        snprintf(hs, sizeof hs, "k_hstep");
    if (pl->generic) {
        snprintf(buf, n, "F=%d T=%d r=%d beta=%g | out-of-envelope path (intermediates in HBM: k_g_gemm / k_g_ratio / k_g_hupd, %d frame splits) | n_cu=%d",
                 pl->p.F, pl->p.T, pl->p.r, pl->p.beta, pl->n_chunks, pl->ctx->n_cu);
        return SNMF_OK;
    }
    snprintf(buf, n,
             "F=%d T=%d r=%d beta=%g | Fm=%d(+%d VALU row) rp=%d Tp=%d | hstep: %s, tile=%d frames, grid=%d x %d thr, lds=%zu B | "
             "wstats: NK=%d waves=%d+%d grid=(%d chunks,%d fgroups,%d kgroups; group-1 chunks %d) lds=%zu B%s | W finish (run loop): %s | n_cu=%d",
             pl->p.F, pl->p.T, pl->p.r, pl->p.beta, pl->Fm, pl->xr, pl->rp, pl->Tp, hs, pl->TTH * pl->NT,
             sr_pipe ? pl->sr_grid : sf_pipe ? pl->sf_grid : ((kl_pipe || rh_pipe) ? pl->rp_grid : pl->grid_h), (sr_pipe || sf_pipe) ? 512 : (rh_pipe ? 768 : (pl->NWH + pl->NLH) * 64),
             sr_pipe ? pl->lds_sr : sf_pipe ? pl->lds_sf : (rh_pipe ? pl->lds_rh : pl->lds_h), pl->NKT, pl->NWB, pl->NLW, pl->n_chunks, pl->n_fg, pl->n_kg, pl->n_ch1 ? pl->n_ch1 : pl->n_chunks, pl->lds_w,
             pl->gram_p ? ", P = W*(H*H') through the Gram matrix" : (pl->wsr ? ", k_wstats_sr: statistics rows per wave, operands straight into the MFMA layouts" : pl->wsf ? (pl->wsf_share ? ", k_wstats_sf: a tile per wave, a single remainder tile shared by the eight waves" : ", k_wstats_sf: a tile per wave") : (pl->til > 1 ? (pl->til == 2 ? ", 2 consumer teams take the tiles in turn" : ", 4+ consumer teams take the tiles in turn") : "")), pl->wfin ? "k_wfin" : (pl->upd_w ? "k_reduce + k_wapply" : (pl->fold_obj && !pl->M ? "none (objective fold + convergence test on the H step's last workgroup)" : "none (objective fold + convergence test: k_reduce)")), pl->ctx->n_cu);
    return SNMF_OK;
}

// ---- data movement ---------------------------------------------------------------------------
static int ensure_staging(snmf_plan* pl, size_t bytes) {
    if (pl->staging_bytes >= bytes) return SNMF_OK;
    if (pl->staging) {
        hipStreamSynchronize(pl->ctx->stream);
        hipFree(pl->staging);
        pl->staging = nullptr;
        pl->staging_bytes = 0;
    }
    hipError_t e = hipMalloc(&pl->staging, bytes);
    if (e != hipSuccess) return fail(SNMF_ERR_NOMEM, "hipMalloc(staging %zu): %s", bytes, hipGetErrorString(e));
    pl->staging_bytes = bytes;
    return SNMF_OK;
}


// Host arrays go through the chunked, pinned, overlapped pipeline of snmf_tu_xfer.hip (pack_in returns as soon as the caller's
// array has been read; the tail of the pipeline is ordered on the engine's stream); device arrays are converted in place.
template <typename TIn, typename TDst = float>
static int pack_in(snmf_plan* pl, const TIn* src, int64_t ld, int rows, int cols, TDst* dst, int rowsP, int colsP,
                   bool do_floor, int is_device) {
    if (!src) return fail(SNMF_ERR_INVALID, "source pointer is NULL");
    if (ld < rows) return fail(SNMF_ERR_INVALID, "leading dimension %lld < rows %d", (long long)ld, rows);
    HIP_TRY(hipSetDevice(pl->ctx->device));
    if (!is_device) return xfer_pack_in<TIn, TDst>(pl->ctx, src, ld, rows, cols, dst, rowsP, colsP, do_floor);
    const size_t n = (size_t)rowsP * colsP;
    hipLaunchKernelGGL((k_pack<TIn, TDst>), dim3(grid_for(n)), dim3(256), 0, pl->ctx->stream, src, ld, rows, cols, dst, rowsP, colsP, kFlr,
                       do_floor ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

template <typename TOut, typename TSrc = float>
static int unpack_out(snmf_plan* pl, const TSrc* src, int rowsP, int rows, int cols, TOut* dst, int64_t ld,
                      int is_device) {
    if (!dst) return fail(SNMF_ERR_INVALID, "destination pointer is NULL");
    if (ld < rows) return fail(SNMF_ERR_INVALID, "leading dimension %lld < rows %d", (long long)ld, rows);
    HIP_TRY(hipSetDevice(pl->ctx->device));
    if (!is_device) return xfer_unpack_out<TOut, TSrc>(pl->ctx, src, rowsP, rows, cols, dst, ld);
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL((k_unpack<TOut, TSrc>), dim3(grid_for(n)), dim3(256), 0, pl->ctx->stream, src, rowsP, rows, cols, dst, ld);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

template <typename T>
int set_v(snmf_plan* pl, const T* V, int64_t ld, int dev) {
    PLAN_CHECK(pl);
    SN_TRY(pack_in<T>(pl, V, ld, pl->p.F, pl->p.T, pl->V, pl->Fp, pl->Tp, pl->p.floor_v != 0, dev));
    pl->have_v = true;
    pl->mdi_v_fresh = true;
    return SNMF_OK;
}
// MDI: the mask turns the plan into a solve of src/snmf_mdi.m / src/snmf_mdi_Sm.m
template <typename T>
static int set_mask(snmf_plan* pl, const T* M, int64_t ld, int dev) {
    PLAN_CHECK(pl);
    if (!pl->upd_h && !pl->upd_w) return fail(SNMF_ERR_UNSUPPORTED, "MDI with neither factor updated is not implemented");
    if (pl->TTH != 32 || pl->TTW != 32 || pl->generic)
        return fail(SNMF_ERR_UNSUPPORTED, "MDI needs the 32-frame tile images in LDS: F + r = %d is too large", pl->p.F + pl->p.r);
    if (!pl->M) {
        SN_TRY(dalloc(&pl->M, (size_t)pl->Fp * pl->Tp));
    }
    SN_TRY(pack_in<T>(pl, M, ld, pl->p.F, pl->p.T, pl->M, pl->Fp, pl->Tp, false, dev));
    pl->small = false;  // the persistent kernels carry no imputation step
    pl->inited = false;
    return SNMF_OK;
}
extern "C" int snmf_plan_set_mask_f64(snmf_plan* pl, const double* M, int64_t ld, int dev) { return set_mask<double>(pl, M, ld, dev); }
extern "C" int snmf_plan_set_mask_f32(snmf_plan* pl, const float* M, int64_t ld, int dev) { return set_mask<float>(pl, M, ld, dev); }
template <typename T>
int set_w(snmf_plan* pl, const T* W, int64_t ld, int dev) {
    PLAN_CHECK(pl);
    SN_TRY((pack_in<T, double>(pl, W, ld, pl->p.F, pl->p.r, pl->Wc, pl->Fp, pl->rp, false, dev)));
    pl->have_w = true;
    pl->w_dirty = true;
    pl->inited = false;
    return SNMF_OK;
}
template <typename T>
int set_h(snmf_plan* pl, const T* H, int64_t ld, int dev) {
    PLAN_CHECK(pl);
    SN_TRY(pack_in<T>(pl, H, ld, pl->p.r, pl->p.T, pl->H[0], pl->rp, pl->Tp, false, dev));
    pl->have_h = true;
    pl->inited = false;
    pl->cur = 0;
    return SNMF_OK;
}
template <typename T>
int set_s(snmf_plan* pl, const T* S, int dev) {
    PLAN_CHECK(pl);
    if (pl->p.sparsity_kind == SNMF_SPARSITY_SCALAR) return fail(SNMF_ERR_STATE, "plan has scalar sparsity");
    if (pl->p.sparsity_kind == SNMF_SPARSITY_RVEC)
        SN_TRY(pack_in<T>(pl, S, pl->p.r, pl->p.r, 1, pl->lamk, pl->rp, 1, false, dev));
    else
        SN_TRY(pack_in<T>(pl, S, pl->p.r, pl->p.r, pl->p.T, pl->S, pl->rp, pl->Tp, false, dev));
    pl->have_s = true;
    pl->inited = false;
    return SNMF_OK;
}

extern "C" int snmf_plan_set_v_f64(snmf_plan* pl, const double* V, int64_t ld, int d) { return set_v(pl, V, ld, d); }
extern "C" int snmf_plan_set_v_f32(snmf_plan* pl, const float* V, int64_t ld, int d) { return set_v(pl, V, ld, d); }
extern "C" int snmf_plan_set_w_f64(snmf_plan* pl, const double* W, int64_t ld, int d) { return set_w(pl, W, ld, d); }
extern "C" int snmf_plan_set_w_f32(snmf_plan* pl, const float* W, int64_t ld, int d) { return set_w(pl, W, ld, d); }
extern "C" int snmf_plan_set_h_f64(snmf_plan* pl, const double* H, int64_t ld, int d) { return set_h(pl, H, ld, d); }
extern "C" int snmf_plan_set_h_f32(snmf_plan* pl, const float* H, int64_t ld, int d) { return set_h(pl, H, ld, d); }
extern "C" int snmf_plan_set_sparsity_f64(snmf_plan* pl, const double* S, int d) { return set_s(pl, S, d); }
extern "C" int snmf_plan_set_sparsity_f32(snmf_plan* pl, const float* S, int d) { return set_s(pl, S, d); }
// (used by the other translation units: snmf_tu_online.hip, snmf_tu_multi.hip, snmf_tu_dnmf.hip)
template int set_v<float>(snmf_plan*, const float*, int64_t, int);
template int set_v<double>(snmf_plan*, const double*, int64_t, int);
template int set_w<float>(snmf_plan*, const float*, int64_t, int);
template int set_w<double>(snmf_plan*, const double*, int64_t, int);
template int set_h<float>(snmf_plan*, const float*, int64_t, int);
template int set_h<double>(snmf_plan*, const double*, int64_t, int);
template int set_s<float>(snmf_plan*, const float*, int);
template int set_s<double>(snmf_plan*, const double*, int);

extern "C" int snmf_plan_get_w_f64(snmf_plan* pl, double* W, int64_t ld, int d) {
    PLAN_CHECK(pl);
    return unpack_out<double, double>(pl, pl->Wc, pl->Fp, pl->p.F, pl->p.r, W, ld, d);
}
extern "C" int snmf_plan_get_w_f32(snmf_plan* pl, float* W, int64_t ld, int d) {
    PLAN_CHECK(pl);
    return unpack_out<float, double>(pl, pl->Wc, pl->Fp, pl->p.F, pl->p.r, W, ld, d);
}

// ---- launch helpers --------------------------------------------------------------------------
StepArgs make_args(snmf_plan* pl) {
    StepArgs a{};
    a.V = pl->V;
    a.Hin = pl->H[pl->cur];
    a.Hout = pl->H[pl->cur ^ 1];
    a.Wt4 = pl->Wt4;
    a.Wk4 = pl->Wk4;
    a.dphv = pl->dphv;
    a.colsum = pl->colsum;
    a.lamk = pl->lamk;
    a.S = pl->S;
    a.slabs = pl->slabs;
    a.spart = pl->spart;
    a.part = pl->part;
    a.stop = &pl->st->stop;
    a.stagger_shift = -1;
    a.nbuf = 2;
    a.til = 1;
    a.prof = pl->prof;
    a.wx = pl->wx;
    a.Fm = pl->Fm;
    a.Fq = pl->Fq;
    a.xr = pl->xr;
    a.F = pl->p.F;
    a.T = pl->p.T;
    a.Fp = pl->Fp;
    a.rp = pl->rp;
    a.Tp = pl->Tp;
    a.nf = pl->nf;
    a.nk = pl->nk;
    a.nqk = (pl->p.r + 7) / 8;
    a.ldh = pl->ldh;
    a.ldr = pl->ldr;
    a.lam_is_u = pl->p.sparsity_kind == SNMF_SPARSITY_SCALAR ? 1 : 0;
    a.lam_u = (float)pl->p.sparsity_scalar;
    a.beta = (float)pl->p.beta;
    const double bb1 = pl->p.beta * (pl->p.beta - 1.0);
    a.inv_bb1 = bb1 != 0.0 ? (float)(1.0 / bb1) : 0.f;
    a.fold_it = pl->fold_now;  // (> 0 only while snmf_plan_run issues the H step of an H-only iteration)
    return a;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: the cache of "already raised to"
// is keyed by (device, kernel) and guarded, so that a second context on another device, or two host threads, cannot
// skip a call they need (launches with more than 64 KiB of LDS would fail) or race on the map.
int ensure_dyn_lds(int device, const void* kern, size_t lds) {
    if (lds <= 64 * 1024) return SNMF_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> raised;
    std::lock_guard<std::mutex> lk(mu);
    size_t& cur = raised[{device, kern}];
    if (cur < lds) {
        HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cur = lds;
    }
    return SNMF_OK;
}

// ---- the out-of-envelope path (csrc/snmf_generic.h) ---------------------------------------------
int g_gemm(snmf_plan* pl, const float* A, long long rsA, long long csA, const float* B, long long rsB, long long csB, float* C,
                  long long rsC, long long csC, int M, int N, int K, int kchunk, long long zC) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.kchunk = kchunk > 0 ? kchunk : K;
    g.rsA = rsA; g.csA = csA; g.rsB = rsB; g.csB = csB; g.rsC = rsC; g.csC = csC; g.zC = zC;
    g.stop = &pl->st->stop;
    const int nz = (K + g.kchunk - 1) / g.kchunk;
    hipLaunchKernelGGL(k_g_gemm, dim3((N + 63) / 64, (M + 63) / 64, std::max(1, nz)), dim3(256), 0, pl->ctx->stream, g);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
// Lam = W * H (H = the current iterate), then the ratio images (+ the divergence terms of that iterate)
template <int BM>
static int g_lam_ratio(snmf_plan* pl, const float* H, bool obj) {
    const int F = pl->p.F, T = pl->p.T, r = pl->p.r;
    SN_TRY(g_gemm(pl, pl->Wcf, 1, pl->Fp, H, 1, pl->rp, pl->gLam, 1, pl->Fp, F, T, r, 0, 0));
    const double bb1 = pl->p.beta * (pl->p.beta - 1.0);
    const float inv_bb1 = bb1 != 0.0 ? (float)(1.0 / bb1) : 0.f;
    if (obj)
        hipLaunchKernelGGL((k_g_ratio<BM, true>), dim3(kGBlocks), dim3(256), 0, pl->ctx->stream, (const float*)pl->V, (const float*)pl->gLam,
                           pl->gR, pl->gD, F, pl->Fp, T, (float)pl->p.beta, inv_bb1, pl->part, (const int*)&pl->st->stop);
    else
        hipLaunchKernelGGL((k_g_ratio<BM, false>), dim3(kGBlocks), dim3(256), 0, pl->ctx->stream, (const float*)pl->V, (const float*)pl->gLam,
                           pl->gR, pl->gD, F, pl->Fp, T, (float)pl->p.beta, inv_bb1, pl->part, (const int*)&pl->st->stop);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
static int g_lam_ratio(snmf_plan* pl, const float* H, bool obj) {
    if (pl->bm == BM_KL) return g_lam_ratio<BM_KL>(pl, H, obj);
    if (pl->bm == BM_EUC) return g_lam_ratio<BM_EUC>(pl, H, obj);
    return g_lam_ratio<BM_GEN>(pl, H, obj);
}
int generic_hstep(snmf_plan* pl, bool obj, bool upd) {
    const int F = pl->p.F, T = pl->p.T, r = pl->p.r;
    const float* Hin = pl->H[pl->cur];
    SN_TRY(g_lam_ratio(pl, Hin, obj));
    if (!upd) return SNMF_OK;
    // num = W' * R, den = W' * D (beta != 1)
    SN_TRY(g_gemm(pl, pl->Wcf, pl->Fp, 1, pl->gR, 1, pl->Fp, pl->gNum, 1, pl->rp, r, T, F, 0, 0));
    if (pl->bm != BM_KL) SN_TRY(g_gemm(pl, pl->Wcf, pl->Fp, 1, pl->gD, 1, pl->Fp, pl->gDen, 1, pl->rp, r, T, F, 0, 0));
    float* Hout = pl->H[pl->cur ^ 1];
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(kGBlocks), dim3(256), 0, pl->ctx->stream, Hin, Hout, (const float*)pl->gNum, (const float*)pl->gDen,
                           (const float*)pl->S, (const float*)pl->lamk, (const float*)pl->colsum, (const float*)pl->dphv, r, pl->rp, T, pl->part,
                           (const int*)&pl->st->stop);
    };
    if (pl->bm == BM_KL) { if (obj) go(k_g_hupd<true, true>); else go(k_g_hupd<true, false>); }
    else { if (obj) go(k_g_hupd<false, true>); else go(k_g_hupd<false, false>); }
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
// W statistics of the current H as split-T slabs (+ the divergence of the previous iterate in W-only solves)
int generic_wstats(snmf_plan* pl, bool obj) {
    const int F = pl->p.F, T = pl->p.T, r = pl->p.r;
    const float* H = pl->H[pl->cur];
    SN_TRY(g_lam_ratio(pl, H, obj));
    const long long nW = (long long)pl->Fp * pl->rp, zC = nW * pl->n_mat;
    // M0 = Q = R * H' (KL: G), M1 = P = D * H'; element (f, k) of a slab at k * Fp + f
    SN_TRY(g_gemm(pl, pl->gR, 1, pl->Fp, H, pl->rp, 1, pl->slabs, 1, pl->Fp, F, r, T, kGChunkT, zC));
    if (pl->bm != BM_KL) {
        SN_TRY(g_gemm(pl, pl->gD, 1, pl->Fp, H, pl->rp, 1, pl->slabs + nW, 1, pl->Fp, F, r, T, kGChunkT, zC));
    } else {
        hipLaunchKernelGGL(k_g_rowsum, dim3((r + 255) / 256, pl->n_chunks), dim3(256), 0, pl->ctx->stream, H, pl->rp, r, T, kGChunkT,
                           pl->spart, (const int*)&pl->st->stop);
        HIP_TRY(hipGetLastError());
    }
    return SNMF_OK;
}

static ReduceArgs make_reduce_args(snmf_plan* pl, double* stats, bool do_mats, bool do_obj, int n_part, bool sh_const) {
    ReduceArgs ra{};
    ra.slabs = pl->slabs;
    ra.spart = pl->spart;
    ra.part = pl->part;
    ra.stats = stats;
    ra.stop = &pl->st->stop;
    ra.n_chunks = pl->n_chunks;
    ra.qp_buf = pl->qp_buf;
    ra.fin_cnt = pl->fin_cnt;
    ra.n_mat = pl->n_mat;
    ra.n_part = n_part;
    ra.rp = pl->rp;
    ra.r = pl->p.r;
    ra.Fp = pl->Fp;
    ra.do_mats = do_mats;
    ra.do_obj = do_obj;
    ra.sh_const = pl->sh_const;
    ra.use_sh_const = sh_const;
    return ra;
}
static int launch_reduce(snmf_plan* pl, double* stats, bool do_mats, bool do_obj, int n_part, bool sh_const, int check_it = 0) {
    ReduceArgs ra = make_reduce_args(pl, stats, do_mats, do_obj, n_part, sh_const);
    ra.check_it = check_it;
    ra.conv_eps = pl->p.conv_eps;
    ra.divh = pl->divh;
    ra.costh = pl->costh;
    ra.st = pl->st;
    if (const snmf_plan::Exchange* x = pl->xpush) {  // the multi-device entry: reduce and push in one launch
        pl->xpush = nullptr;
        ra.npush = x->n;
        for (int q = 0; q < x->n; ++q) {
            ra.push_dst[q] = x->push_dst[q];
            ra.push_flag[q] = x->push_flag[q];
        }
        ra.push_done = x->push_done;
        ra.push_seq = x->seq;
    }
    const size_t tot = do_mats ? ((size_t)pl->n_mat * pl->p.r * pl->Fp) / 4 : 1;
    ScopedTimer tm(pl->ctx, FAM_REDUCE);
    hipLaunchKernelGGL(k_reduce, dim3((int)std::max<size_t>(1, std::min<size_t>((tot + 31) / 32, 4096))), dim3(256),
                       0, pl->ctx->stream, ra);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

static ApplyArgs make_apply_args(snmf_plan* pl, const double* stats, int check_it, bool do_update, bool init_mode) {
    ApplyArgs aa{};
    aa.stats = stats;
    aa.Wc = pl->Wc;
    aa.Wcf = pl->Wcf;
    aa.Wt4 = pl->Wt4;
    aa.Wk4 = pl->Wk4;
    aa.dphv = pl->dphv;
    aa.colsum = pl->colsum;
    aa.lamk = pl->lamk;
    aa.w_ind = pl->w_ind;
    aa.divh = pl->divh;
    aa.costh = pl->costh;
    aa.st = pl->st;
    aa.F = pl->p.F;
    aa.r = pl->p.r;
    aa.Fp = pl->Fp;
    aa.rp = pl->rp;
    aa.n_mat = pl->n_mat;
    aa.wx = pl->wx;
    aa.Fm = pl->Fm;
    aa.Fq = pl->Fq;
    aa.xr = pl->xr;
    aa.check_it = check_it;
    aa.do_update = do_update;
    aa.init_mode = init_mode;
    aa.conv_eps = pl->p.conv_eps;
    aa.wn = pl->wn;
    return aa;
}
int launch_wapply(snmf_plan* pl, const double* stats, int check_it, bool do_update, bool init_mode) {
    ApplyArgs aa = make_apply_args(pl, stats, check_it, do_update, init_mode);
    size_t lds = 0;
    if (const snmf_plan::Exchange* x = pl->xgather) {  // the multi-device entry: sum the ranks' slots and apply in one launch
        pl->xgather = nullptr;
        aa.gather = x->gather;
        aa.ngather = x->n;
        aa.gather_len = x->len;
        aa.gflags = x->gflags;
        aa.gseq = x->seq;
        aa.fault = &pl->st->fault;
        lds = (size_t)2 * pl->Fp * sizeof(double);
    }
    ScopedTimer tm(pl->ctx, FAM_WAPPLY);
    hipLaunchKernelGGL(k_wapply, dim3(pl->p.r), dim3(256), lds, pl->ctx->stream, aa);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

// k_reduce + k_wapply in one launch (the loop of snmf_plan_run; see k_wfin)
static int launch_wfin(snmf_plan* pl, double* stats, bool do_obj, int n_part, bool sh_const, int check_it) {
    const ReduceArgs ra = make_reduce_args(pl, stats, true, do_obj, n_part, sh_const);
    const ApplyArgs aa = make_apply_args(pl, stats, check_it, true, false);
    ScopedTimer tm(pl->ctx, FAM_WFIN);
    auto launch = [&](auto kern) -> int {
        SN_TRY(ensure_dyn_lds(pl->ctx->device, (const void*)kern, pl->lds_wfin));
        hipLaunchKernelGGL(kern, dim3(pl->p.r, pl->wfin_S), dim3(768), pl->lds_wfin, pl->ctx->stream, ra, aa);
        HIP_TRY(hipGetLastError());
        return SNMF_OK;
    };
    return pl->n_mat == 2 ? launch(k_wfin<2>) : launch(k_wfin<1>);
}

static int launch_check(snmf_plan* pl, const double* stats, int it) {
    const size_t off = (size_t)pl->n_mat * pl->rp * pl->Fp + pl->rp;
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, pl->ctx->stream, stats, off, pl->divh, pl->costh, pl->st, it,
                       pl->p.conv_eps);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

// ---- init: src/sparse_nmf.m:157-173 ----------------------------------------------------------
extern "C" int snmf_plan_init(snmf_plan* pl) {
    PLAN_CHECK(pl);
    if (!pl->have_v || !pl->have_w || !pl->have_h) return fail(SNMF_ERR_STATE, "set_v, set_w and set_h must precede init");
    if (!pl->have_s) return fail(SNMF_ERR_STATE, "set_sparsity must precede init for this sparsity kind");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    hipStream_t st = pl->ctx->stream;
    HIP_TRY(hipMemsetAsync(pl->st, 0, sizeof(DevState), st));
    // arrival counters of the split tiles: "last to arrive" is old % S == S - 1, so a launch that ended in the fault path
    // (or was otherwise left partial) must not leave them misaligned for the next solve
    if (pl->part_cnt) HIP_TRY(hipMemsetAsync(pl->part_cnt, 0, (size_t)(pl->rp_tiles - pl->rp_full) * 4, st));
    if (pl->fin_cnt) HIP_TRY(hipMemsetAsync(pl->fin_cnt, 0, (size_t)pl->p.r * 4, st));  // (k_wfin's split form: the same alignment argument)
    HIP_TRY(hipMemsetAsync(&reinterpret_cast<FoldBlock*>(pl->st + 1)->cnt, 0, 4, st));  // (the H-only loop's folding H step: likewise)
    HIP_TRY(hipMemsetAsync(pl->divh, 0, sizeof(double) * std::max(1, pl->p.max_iter), st));
    HIP_TRY(hipMemsetAsync(pl->costh, 0, sizeof(double) * std::max(1, pl->p.max_iter), st));
    // wn = sqrt(sum(w.^2)); w = w./wn  (+ operand images, colsum, dphv).  When only V / H changed since
    // the last init (online separation: the dictionary stays) the normalised W and wn are reused:
    // normalising an already-identical input would reproduce them bit for bit.
    if (pl->w_dirty) SN_TRY(launch_wapply(pl, pl->stats, 0, false, true));
    pl->w_dirty = false;
    pl->small_done = false;
    if (pl->M) {
        // v = max(v .* M, flr) (src/snmf_mdi.m:175).  V is state in an MDI solve (re-imputed every iteration),
        // so each solve needs its own snmf_plan_set_v.
        if (!pl->mdi_v_fresh) return fail(SNMF_ERR_STATE, "MDI: snmf_plan_set_v must precede every snmf_plan_init (V is rewritten by the solve)");
        hipLaunchKernelGGL(k_mdi_start, dim3(grid_for((size_t)pl->Fp * pl->p.T)), dim3(256), 0, st, pl->V, (const float*)pl->M, pl->Fp,
                           pl->p.F, pl->p.T, kFlr);
        HIP_TRY(hipGetLastError());
        pl->mdi_v_fresh = false;
        pl->mdi_final = false;
    }
    // h = h .* wn'
    const size_t nH = (size_t)pl->rp * pl->Tp;
    hipLaunchKernelGGL(k_scale_h, dim3(grid_for(nH)), dim3(256), 0, st, pl->H[pl->cur], pl->wn, pl->rp, pl->p.r, nH);
    HIP_TRY(hipGetLastError());
    pl->it_done = 0;
    pl->final_done = false;
    pl->inited = true;
    pl->sh_const = 0.0;
    if (!pl->upd_h && pl->p.cost_check) {
        // H never changes: sum(sum(sparsity .* h)) (:261) is a constant of the solve
        const int g = 256;
        hipLaunchKernelGGL(k_sum_sh, dim3(g), dim3(256), 0, st, pl->H[pl->cur], pl->S, pl->lamk, pl->rp, pl->p.r,
                           pl->p.T, pl->part);
        HIP_TRY(hipGetLastError());
        std::vector<double> hp(g);
        HIP_TRY(hipMemcpyAsync(hp.data(), pl->part, g * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        double s = 0.0;
        for (double x : hp) s += x;
        pl->sh_const = s;
    }
    return SNMF_OK;
}

// ---- the iteration, split so that an all-reduce can sit between wstats and wapply ----------
// Iteration j (1-based) = hstep(j) -> wstats(j) -> [reduce] -> wapply(j).  The objective of
// iterate j-1 is produced by the first pass of iteration j that forms Lam = W_{j-1} * H_{j-1}.
static bool want_obj(const snmf_plan* pl, int j) { return pl->p.cost_check && j > 1; }
// objective partials written by an H-UPDATE launch of k_hstep* (= its grid; the objective-only launches use grid_h)
// objective partials written by a k_wstats launch with the objective (W-only solves) = its workgroups
static int wstats_parts(const snmf_plan* pl) {
    if (pl->generic) return kGBlocks;
    return pl->n_ch1 ? pl->n_chunks + (pl->n_fg - 1) * pl->n_ch1 : pl->n_chunks * pl->n_fg;
}
static bool hupd_is_rp(const snmf_plan* pl) {
    return !pl->M && (pl->rh || (pl->NWH == 8 && pl->NLH == 4 && pl->hstep_rp && pl->bm == BM_KL));
}
static int hupd_parts(const snmf_plan* pl) {
    if (pl->generic) return kGBlocks;
    if (pl->M) return pl->grid_mdi;
    if (pl->sr) return pl->sr_grid;
    if (pl->sf) return pl->sf_grid;
    if (hupd_is_rp(pl)) return (pl->hm && !pl->rh && !pl->S) ? pl->hm_grid : pl->rp_grid;
    return pl->grid_h;
}

extern "C" int snmf_plan_hstep(snmf_plan* pl) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    const int j = pl->it_done + 1;
    if (!pl->upd_h) {
        // W-only MDI: the Lam pass still runs, for the re-imputation of V (src/snmf_mdi.m:251-254) and the objective
        // of the previous iterate; the W statistics then read the re-imputed V
        if (pl->M && (pl->it_done >= 1 || want_obj(pl, j))) SN_TRY(launch_hstep(pl, true, false));
        return SNMF_OK;
    }
    SN_TRY(launch_hstep(pl, want_obj(pl, j), true));
    pl->cur ^= 1;
    return SNMF_OK;
}

// after hstep(j): H[cur] = H_j.  NOTE: when the device-side stop flag is set the kernels are
// no-ops and the H buffers keep H_{n_iter}; get_h accounts for that.
extern "C" int snmf_plan_wstats(snmf_plan* pl, double* stats) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    if (!stats) return fail(SNMF_ERR_INVALID, "stats is NULL");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    const int j = pl->it_done + 1;
    const bool obj = want_obj(pl, j);
    const bool mdi_wonly = pl->M && !pl->upd_h;  // objective partials come from the MDI Lam pass of snmf_plan_hstep
    if (pl->upd_w) {
        // W-only mode: the divergence of iterate j-1 comes from this pass (Lam' = W_{j-1} * H)
        SN_TRY(launch_wstats(pl, obj && !pl->upd_h && !mdi_wonly));
    }
    const int n_part = pl->upd_h ? hupd_parts(pl) : pl->M ? pl->grid_mdi : wstats_parts(pl);
    if (!pl->upd_h && !pl->upd_w && obj) {
        // neither factor is updated: the loop only re-evaluates the objective
        SN_TRY(launch_hstep(pl, true, false));
        SN_TRY(launch_reduce(pl, stats, false, true, pl->grid_h, true));
        return SNMF_OK;
    }
    SN_TRY(launch_reduce(pl, stats, pl->upd_w, obj, n_part, !pl->upd_h));
    return SNMF_OK;
}

extern "C" int snmf_plan_wapply(snmf_plan* pl, const double* stats) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    if (!stats) return fail(SNMF_ERR_INVALID, "stats is NULL");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    const int j = pl->it_done + 1;
    const bool obj = want_obj(pl, j);
    if (pl->upd_w) SN_TRY(launch_wapply(pl, stats, obj ? j - 1 : 0, true, false));
    else if (obj) SN_TRY(launch_check(pl, stats, j - 1));
    pl->it_done = j;
    return SNMF_OK;
}

extern "C" int snmf_plan_objstats(snmf_plan* pl, double* stats) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    if (!stats) return fail(SNMF_ERR_INVALID, "stats is NULL");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    SN_TRY(launch_hstep(pl, true, false));
    // sum(S.*H) of the final iterate: the objective-only pass does not visit H's rows, so take it
    // from a dedicated reduction over the current H.
    const int g = 256;
    const int np = pl->M ? pl->grid_mdi : pl->grid_h;  // workgroups of the Lam pass = objective partials
    hipLaunchKernelGGL(k_sum_sh, dim3(g), dim3(256), 0, pl->ctx->stream, pl->H[pl->cur], pl->S, pl->lamk, pl->rp,
                       pl->p.r, pl->p.T, pl->part + 2 * (size_t)np);
    HIP_TRY(hipGetLastError());
    SN_TRY(launch_reduce(pl, stats, false, true, np, false));
    const size_t off = (size_t)pl->n_mat * pl->rp * pl->Fp + pl->rp;
    hipLaunchKernelGGL(k_fold_sh, dim3(1), dim3(64), 0, pl->ctx->stream, pl->part + 2 * (size_t)np, 256,
                       stats + off, &pl->st->stop);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}

extern "C" int snmf_plan_objapply(snmf_plan* pl, const double* stats) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    if (pl->it_done < 1) return SNMF_OK;
    SN_TRY(launch_check(pl, stats, pl->it_done));
    pl->final_done = true;
    return SNMF_OK;
}

int read_state(snmf_plan* pl, DevState* hs) {
    HIP_TRY(hipMemcpyAsync(hs, pl->st, sizeof(DevState), hipMemcpyDeviceToHost, pl->ctx->stream));
    HIP_TRY(hipStreamSynchronize(pl->ctx->stream));
    if (hs->fault) return fail(SNMF_ERR_INTERNAL, "a device-side producer/consumer wait timed out: results are invalid");
    return SNMF_OK;
}

extern "C" int snmf_plan_stopped(snmf_plan* pl, int32_t* stopped) {
    PLAN_CHECK(pl);
    DevState hs{};
    SN_TRY(read_state(pl, &hs));
    if (stopped) *stopped = hs.stop;
    return SNMF_OK;
}

extern "C" int snmf_plan_run_sharded(snmf_plan* pl, int32_t n_iters, double* stats, snmf_allreduce_fn all_reduce, void* user,
                                     int32_t poll_every, int32_t finalize, int32_t* iters_done) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    if (!stats) return fail(SNMF_ERR_INVALID, "stats is NULL");
    if (iters_done) *iters_done = 0;
    const int64_t full = snmf_plan_stats_len(pl);
    const bool w_any = pl->upd_w;
    double* ar_ptr = w_any ? stats : stats + full - 2;  // H-only solves exchange nothing but the two cost scalars
    const int64_t ar_len = w_any ? full : 2;
    const bool can_stop = pl->p.cost_check && pl->p.conv_eps > 0 && poll_every > 0;
    int since = 0, ran = 0;
    bool stopped = false;
    while (ran < n_iters && pl->it_done < pl->p.max_iter) {
        SN_TRY(snmf_plan_hstep(pl));
        SN_TRY(snmf_plan_wstats(pl, stats));
        if (all_reduce && all_reduce(ar_ptr, ar_len, user) != 0) return fail(SNMF_ERR_INVALID, "all_reduce callback failed");
        SN_TRY(snmf_plan_wapply(pl, stats));
        ++ran;
        if (can_stop && ++since >= poll_every) {
            since = 0;
            int32_t st = 0;
            SN_TRY(snmf_plan_stopped(pl, &st));
            if (st) {
                stopped = true;
                break;
            }
        }
    }
    if (iters_done) *iters_done = ran;
    if (finalize && !stopped && pl->it_done >= pl->p.max_iter && pl->p.cost_check && !pl->final_done && pl->it_done > 0) {
        SN_TRY(snmf_plan_objstats(pl, stats));
        if (all_reduce && all_reduce(stats + full - 2, 2, user) != 0) return fail(SNMF_ERR_INVALID, "all_reduce callback failed");
        SN_TRY(snmf_plan_objapply(pl, stats));
    }
    return SNMF_OK;
}

// final objective (iterate max_iter): objective-only pass + sum(S.*H) + check
static int finalize_objective(snmf_plan* pl) {
    if (pl->final_done || !pl->p.cost_check || pl->it_done < 1) return SNMF_OK;
    SN_TRY(snmf_plan_objstats(pl, pl->stats));
    return snmf_plan_objapply(pl, pl->stats);
}

extern "C" int snmf_plan_run(snmf_plan* pl, int32_t n_iters, int32_t* iters_done) {
    PLAN_CHECK(pl);
    if (!pl->inited) return fail(SNMF_ERR_STATE, "plan not initialised");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    if (pl->small && pl->it_done == 0 && n_iters >= pl->p.max_iter && !pl->small_done) {
        SN_TRY(launch_small(pl, 1, pl->p.T, pl->divh, pl->costh, pl->st));
        pl->it_done = pl->p.max_iter;
        pl->final_done = true;
        pl->small_done = true;
        if (iters_done) {
            DevState hs{};
            SN_TRY(read_state(pl, &hs));
            *iters_done = hs.stop ? hs.n_iter : pl->it_done;
        }
        return SNMF_OK;
    }
    const int target = std::min(pl->p.max_iter, pl->it_done + std::max(0, n_iters));
    const bool can_stop = pl->p.cost_check && pl->p.conv_eps > 0.0;
    int since_poll = 0;
    bool stopped = false;
    while (pl->it_done < target) {
        if (pl->isf && pl->wfin && !pl->M && pl->n_chunks == std::max(1, std::min((pl->p.T + 31) / 32, pl->ctx->n_cu))) {
            // F <= 64, r <= 128, full KL update: H step + W statistics in ONE launch (k_iter_sf), then the reduction + W update
            const int j = pl->it_done + 1;
            const bool obj = want_obj(pl, j);
            SN_TRY(launch_iter_sf(pl, obj));
            pl->cur ^= 1;
            SN_TRY(launch_wfin(pl, pl->stats, obj, pl->n_chunks, false, obj ? j - 1 : 0));
            pl->it_done = j;
        } else {
        // H-only: the objective fold and the convergence test of iteration j - 1 ride on the H step of iteration j (its last
        // workgroup to arrive folds, obj_partial_out); without that, k_reduce with the test behind it (the step API's k_reduce + k_check)
        const bool h_only = pl->upd_h && !pl->upd_w && !pl->M;
        pl->fold_now = (h_only && pl->fold_obj && want_obj(pl, pl->it_done + 1)) ? pl->it_done : 0;
        const int rc_h = snmf_plan_hstep(pl);
        const bool folded = pl->fold_now > 0;
        pl->fold_now = 0;
        SN_TRY(rc_h);
        if (pl->wfin) {
            // (what snmf_plan_wstats + snmf_plan_wapply do for a W update, the reduction and the update in one launch)
            const int j = pl->it_done + 1;
            const bool obj = want_obj(pl, j), mdi_wonly = pl->M && !pl->upd_h;
            SN_TRY(launch_wstats(pl, obj && !pl->upd_h && !mdi_wonly));
            const int n_part = pl->upd_h ? hupd_parts(pl) : pl->M ? pl->grid_mdi : wstats_parts(pl);
            SN_TRY(launch_wfin(pl, pl->stats, obj, n_part, !pl->upd_h, obj ? j - 1 : 0));
            pl->it_done = j;
        } else if (pl->upd_h && !pl->upd_w && !pl->M) {
            const int j = pl->it_done + 1;
            if (want_obj(pl, j) && !folded) SN_TRY(launch_reduce(pl, pl->stats, false, true, hupd_parts(pl), false, j - 1));
            pl->it_done = j;
        } else {
            SN_TRY(snmf_plan_wstats(pl, pl->stats));
            SN_TRY(snmf_plan_wapply(pl, pl->stats));
        }
        }
        if (can_stop && ++since_poll >= 4) {
            since_poll = 0;
            DevState hs{};
            SN_TRY(read_state(pl, &hs));
            if (hs.stop) {
                stopped = true;
                break;
            }
        }
    }
    if (!stopped && pl->it_done >= pl->p.max_iter) SN_TRY(finalize_objective(pl));
    if (pl->M && !stopped && pl->it_done >= pl->p.max_iter && !pl->mdi_final) {
        // the re-imputation of the last iteration (src/snmf_mdi.m:251-254) rides on the final objective pass;
        // without cost_check that pass does not exist, so run the Lam pass for its imputation alone
        if (!pl->p.cost_check && pl->it_done >= 1) SN_TRY(launch_hstep(pl, true, false));
        pl->mdi_final = true;
    }
    if (iters_done) {
        DevState hs{};
        SN_TRY(read_state(pl, &hs));
        *iters_done = hs.stop ? hs.n_iter : pl->it_done;
    }
    return SNMF_OK;
}

// Which H buffer holds the result?  Without a stop: H[cur].  With a stop recorded at iteration
// n (detected while iteration n+1 was in flight): hstep(n+1) already wrote H_{n+1} into the
// other buffer before the flag was raised, later launches were no-ops although the host kept
// flipping `cur`; H_n is the buffer with index parity n (H_0 lives in buffer 0 after set_h).
int result_h_index(snmf_plan* pl, int* idx) {
    DevState hs{};
    SN_TRY(read_state(pl, &hs));
    if (hs.stop && pl->upd_h && !pl->small_done) *idx = hs.n_iter & 1;
    else *idx = pl->cur;
    return SNMF_OK;
}

extern "C" int snmf_plan_get_h_f64(snmf_plan* pl, double* H, int64_t ld, int d) {
    PLAN_CHECK(pl);
    int idx = 0;
    SN_TRY(result_h_index(pl, &idx));
    return unpack_out<double>(pl, pl->H[idx], pl->rp, pl->p.r, pl->p.T, H, ld, d);
}
extern "C" int snmf_plan_get_h_f32(snmf_plan* pl, float* H, int64_t ld, int d) {
    PLAN_CHECK(pl);
    int idx = 0;
    SN_TRY(result_h_index(pl, &idx));
    return unpack_out<float>(pl, pl->H[idx], pl->rp, pl->p.r, pl->p.T, H, ld, d);
}

// v_MDI of src/snmf_mdi.m:296-306 from the solve's final (W, H, V)
template <typename T>
static int get_v_mdi(snmf_plan* pl, T* V, int64_t ld, int dev) {
    PLAN_CHECK(pl);
    if (!pl->M) return fail(SNMF_ERR_STATE, "not an MDI plan (snmf_plan_set_mask was not called)");
    if (!pl->inited || pl->it_done < 1) return fail(SNMF_ERR_STATE, "snmf_plan_run must precede get_v_mdi");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    hipStream_t st = pl->ctx->stream;
    int idx = 0;
    SN_TRY(result_h_index(pl, &idx));
    const int F = pl->p.F, Tn = pl->p.T;
    float* tmp = nullptr;
    SN_TRY(dalloc(&tmp, (size_t)F * Tn));
    constexpr int NC = 8;
    const size_t lds = (size_t)(NC * pl->rp + NC * F + 1) * 4 + 2 * NC * 4 * sizeof(double);
    hipLaunchKernelGGL(k_mdi_final<NC>, dim3((Tn + NC - 1) / NC), dim3(256), lds, st, (const float*)pl->V, (const float*)pl->M,
                       (const float*)pl->Wcf, (const float*)pl->H[idx], F, pl->Fp, pl->p.r, pl->rp, Tn, kFlr, tmp);
    int rc = hipGetLastError() == hipSuccess ? SNMF_OK : fail(SNMF_ERR_NO_DEVICE, "k_mdi_final launch failed");
    if (rc == SNMF_OK) rc = unpack_out<T, float>(pl, tmp, F, F, Tn, V, ld, dev);
    hipStreamSynchronize(st);
    hipFree(tmp);
    return rc;
}
extern "C" int snmf_plan_get_v_mdi_f64(snmf_plan* pl, double* V, int64_t ld, int dev) { return get_v_mdi<double>(pl, V, ld, dev); }
extern "C" int snmf_plan_get_v_mdi_f32(snmf_plan* pl, float* V, int64_t ld, int dev) { return get_v_mdi<float>(pl, V, ld, dev); }

extern "C" int snmf_plan_get_objective(snmf_plan* pl, double* div_out, double* cost_out, int32_t* n_iter_out) {
    PLAN_CHECK(pl);
    HIP_TRY(hipSetDevice(pl->ctx->device));
    DevState hs{};
    SN_TRY(read_state(pl, &hs));
    const int mi = pl->p.max_iter;
    // iterations executed: the stop index (:279-281) or every launched iteration
    const int n_exec = hs.stop ? hs.n_iter : pl->it_done;
    if (n_iter_out) *n_iter_out = n_exec;
    if (div_out) {
        std::fill(div_out, div_out + mi, 0.0);
        if (pl->p.cost_check && hs.n_iter > 0)
            HIP_TRY(hipMemcpy(div_out, pl->divh, sizeof(double) * std::min(mi, hs.n_iter), hipMemcpyDeviceToHost));
    }
    if (cost_out) {
        std::fill(cost_out, cost_out + mi, 0.0);
        if (pl->p.cost_check && hs.n_iter > 0)
            HIP_TRY(hipMemcpy(cost_out, pl->costh, sizeof(double) * std::min(mi, hs.n_iter), hipMemcpyDeviceToHost));
    }
    return SNMF_OK;
}

// ---- online stream ---------------------------------------------------------------------------
template <typename T>
static int solve_frames_impl(snmf_plan* pl, int32_t tps, const T* V, int64_t ldV, int32_t n_solves, const T* H0,
                             T* H_out, int32_t* n_iter_out, double* cost_out) {
    PLAN_CHECK(pl);
    if (!V || !H0 || !H_out || n_solves <= 0) return fail(SNMF_ERR_INVALID, "V, H0, H_out and n_solves are required");
    if (pl->upd_w || !pl->upd_h) return fail(SNMF_ERR_STATE, "solve_frames needs an H-only plan (w_update_ind all false)");
    if (!pl->have_w) return fail(SNMF_ERR_STATE, "snmf_plan_set_w must precede solve_frames");
    if (!pl->have_s) return fail(SNMF_ERR_STATE, "set_sparsity must precede solve_frames for this sparsity kind");
    if (pl->p.sparsity_kind == SNMF_SPARSITY_FULL) return fail(SNMF_ERR_UNSUPPORTED, "solve_frames: full sparsity matrix");
    if (tps < 1 || tps > 32) return fail(SNMF_ERR_UNSUPPORTED, "frames per solve must be in [1,32] (got %d)", tps);
    if (!pl->small_ok) return fail(SNMF_ERR_UNSUPPORTED, "F + r too large for the persistent online kernel");
    const int F = pl->p.F, r = pl->p.r;
    const size_t ncols = (size_t)n_solves * tps;
    if (ncols > (size_t)pl->p.T) return fail(SNMF_ERR_INVALID, "n_solves*frames_per_solve = %zu exceeds the plan's T = %d", ncols, pl->p.T);
    if (ldV < F) return fail(SNMF_ERR_INVALID, "ldV < F");
    HIP_TRY(hipSetDevice(pl->ctx->device));
    hipStream_t st = pl->ctx->stream;
    const int mi = std::max(1, pl->p.max_iter);
    // device staging: [V | H0 | H_all] as T, then per-solve state, objective histories, outputs
    const size_t bV = ((ncols - 1) * (size_t)ldV + F) * sizeof(T), bH0 = (size_t)r * tps * sizeof(T),
                 bHo = (size_t)r * ncols * sizeof(T);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t oH0 = al(bV), oHo = oH0 + al(bH0), oSt = oHo + al(bHo), oDv = oSt + al((size_t)n_solves * sizeof(DevState)),
                 oCs = oDv + al((size_t)n_solves * mi * 8), oNi = oCs + al((size_t)n_solves * mi * 8),
                 oCo = oNi + al((size_t)n_solves * 4), tot = oCo + al((size_t)n_solves * 8);
    SN_TRY(ensure_staging(pl, tot));
    char* sb = (char*)pl->staging;
    HIP_TRY(hipMemcpyAsync(sb, V, bV, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sb + oH0, H0, bH0, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(sb + oSt, 0, oNi - oSt, st));
    // V: all frames at once; floor as src/sparse_nmf.m:169
    const size_t nVp = (size_t)pl->Fp * pl->Tp;
    hipLaunchKernelGGL(k_pack<T>, dim3(grid_for(nVp)), dim3(256), 0, st, (const T*)sb, ldV, F, (int)ncols, pl->V, pl->Fp,
                       pl->Tp, kFlr, pl->p.floor_v ? 1 : 0);
    HIP_TRY(hipGetLastError());
    pl->have_v = true;
    // W: normalised once per set_w (:157-159)
    if (pl->w_dirty) SN_TRY(launch_wapply(pl, pl->stats, 0, false, true));
    pl->w_dirty = false;
    // H0 .* wn' replicated for every solve (:160)
    pl->cur = 0;
    hipLaunchKernelGGL(k_tile_h0<T>, dim3(grid_for(ncols * pl->rp)), dim3(256), 0, st, (const T*)(sb + oH0), pl->wn, r,
                       pl->rp, tps, n_solves, pl->H[0]);
    HIP_TRY(hipGetLastError());
    pl->have_h = true;
    pl->inited = false;  // the plan's single-solve state is not valid after a stream call
    SN_TRY(launch_small(pl, n_solves, tps, (double*)(sb + oDv), (double*)(sb + oCs), (DevState*)(sb + oSt)));
    hipLaunchKernelGGL(k_unpack<T>, dim3(grid_for((size_t)r * ncols)), dim3(256), 0, st, pl->H[0], pl->rp, r, (int)ncols,
                       (T*)(sb + oHo), (int64_t)r);
    hipLaunchKernelGGL(k_collect, dim3((n_solves + 255) / 256), dim3(256), 0, st, (const DevState*)(sb + oSt),
                       (const double*)(sb + oCs), n_solves, pl->p.max_iter, pl->p.cost_check, (int*)(sb + oNi),
                       (double*)(sb + oCo));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(H_out, sb + oHo, bHo, hipMemcpyDeviceToHost, st));
    if (n_iter_out) HIP_TRY(hipMemcpyAsync(n_iter_out, sb + oNi, (size_t)n_solves * 4, hipMemcpyDeviceToHost, st));
    if (cost_out) HIP_TRY(hipMemcpyAsync(cost_out, sb + oCo, (size_t)n_solves * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return SNMF_OK;
}
extern "C" int snmf_plan_solve_frames_f64(snmf_plan* pl, int32_t tps, const double* V, int64_t ldV, int32_t n,
                                          const double* H0, double* H_out, int32_t* n_iter_out, double* cost_out) {
    return solve_frames_impl<double>(pl, tps, V, ldV, n, H0, H_out, n_iter_out, cost_out);
}
extern "C" int snmf_plan_solve_frames_f32(snmf_plan* pl, int32_t tps, const float* V, int64_t ldV, int32_t n,
                                          const float* H0, float* H_out, int32_t* n_iter_out, double* cost_out) {
    return solve_frames_impl<float>(pl, tps, V, ldV, n, H0, H_out, n_iter_out, cost_out);
}

// ---- one-shot drop-in ------------------------------------------------------------------------
// The pages of a result array that was allocated a moment ago (mxCreateDoubleMatrix, np.empty: MATLAB's value semantics) do not
// exist yet: the first write to each is a page fault, and at 205 MB (C2's activations) those faults -- 50 000 of them when the
// allocation did not get huge pages -- were up to 7 ms INSIDE the download of a one-shot call (round 5: Plan.get_h 3.4 ms into
// an array with huge pages, 10-11 ms behind a solve in the same process).  The out-of-place entries know the destination from the
// start, so a few host threads make its pages exist WHILE the device solves: madvise(MADV_POPULATE_WRITE) (Linux 5.14: faults the
// range in without writing to it), else one write per page -- joined before the download, so nothing can be overwritten.
#include <cerrno>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
namespace {
struct Prefault {
    std::vector<std::thread> th;
    void start(void* ptr, size_t bytes, int n_thr = 4) {
        if (!ptr || bytes < ((size_t)8 << 20)) return;
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
        const uintptr_t lo = ((uintptr_t)ptr + pg - 1) & ~(uintptr_t)(pg - 1), hi = ((uintptr_t)ptr + bytes) & ~(uintptr_t)(pg - 1);
        if (hi <= lo) return;
        const size_t n_pg = (hi - lo) / pg;
        for (int i = 0; i < n_thr; ++i) {
            const uintptr_t a = lo + (n_pg * i / n_thr) * pg, b = lo + (n_pg * (i + 1) / n_thr) * pg;
            if (b <= a) continue;
            try {
                th.emplace_back([a, b, pg] {
                    (void)madvise((void*)a, b - a, MADV_HUGEPAGE);  // (a hint where transparent huge pages are opt-in: 100 faults instead of 50 000)
                    if (madvise((void*)a, b - a, MADV_POPULATE_WRITE) == 0) return;
                    // kernels without MADV_POPULATE_WRITE answer EINVAL: one write per page instead (the range is an OUTPUT
                    // array nothing has been stored into yet: the caller checked that it overlaps no input).  Any other errno
                    // (ENOMEM, EFAULT, EPERM on a sealed mapping): leave the pages alone, the copy-out takes the faults
                    if (errno != EINVAL) return;
                    for (uintptr_t q = a; q < b; q += pg) *(volatile char*)q = 0;
                });
            } catch (...) {  // no thread to be had: the copy-out takes the faults itself (nothing may throw across the C ABI)
                break;
            }
        }
    }
    void join() {
        for (auto& t : th) t.join();
        th.clear();
    }
    ~Prefault() { join(); }
};
}  // namespace

// [a, a + na) and [b, b + nb) share a byte
static bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
    if (!a || !b || !na || !nb) return false;
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

// oop: called through an out-of-place entry (W / H are result arrays of their own: only then may their pages be touched early)
template <typename T>
static int sparse_nmf_impl(snmf_ctx* ctx, const snmf_params* p, const T* V, int64_t ldV, const T* W0, const T* H0, T* W, T* H,
                           const T* sparsity, double* div_out, double* cost_out, int32_t* n_iter_out, bool oop) {
    if (!ctx) return fail(SNMF_ERR_INVALID, "ctx is NULL");
    if (!V || !W || !H || !W0 || !H0) return fail(SNMF_ERR_INVALID, "V, W and H must be non-NULL");
    snmf_plan* pl = nullptr;
    SN_TRY(snmf_plan_create(ctx, p, &pl));
    int s = SNMF_OK;
    SN_STEP(s, set_v<T>(pl, V, ldV, 0));
    SN_STEP(s, set_w<T>(pl, W0, p->F, 0));
    SN_STEP(s, set_h<T>(pl, H0, p->r, 0));
    if (p->sparsity_kind != SNMF_SPARSITY_SCALAR) {
        if (!sparsity) SN_STEP(s, fail(SNMF_ERR_INVALID, "sparsity array required for this sparsity_kind"));
        else SN_STEP(s, set_s<T>(pl, sparsity, 0));
    }
    Prefault pf;
    if (s == SNMF_OK && oop) {
        // out-of-place: H holds nothing yet -- unless the caller handed in a result array that overlaps one of the inputs
        // (partially aliased buffers are legal for the copy-out, which runs after every input has been consumed, but the
        // pre-fault's fallback WRITES into the pages while the inputs may still be being read)
        const size_t hb = (size_t)p->r * p->T * sizeof(T);
        const size_t sb = p->sparsity_kind == SNMF_SPARSITY_SCALAR ? 0 : (p->sparsity_kind == SNMF_SPARSITY_RVEC ? (size_t)p->r : (size_t)p->r * p->T) * sizeof(T);
        const bool clash = ranges_overlap(H, hb, H0, hb) || ranges_overlap(H, hb, W0, (size_t)p->F * p->r * sizeof(T)) ||
                           ranges_overlap(H, hb, V, (size_t)ldV * p->T * sizeof(T)) || ranges_overlap(H, hb, sparsity, sb) ||
                           ranges_overlap(H, hb, W, (size_t)p->F * p->r * sizeof(T));
        if (!clash) pf.start(H, hb);
    }
    SN_STEP(s, snmf_plan_init(pl));
    if (s == SNMF_OK) SN_STEP(s, snmf_plan_run(pl, p->max_iter, nullptr));
    pf.join();
    if (s == SNMF_OK) {
        if (sizeof(T) == 8) {
            SN_STEP(s, snmf_plan_get_w_f64(pl, (double*)W, p->F, 0));
            SN_STEP(s, snmf_plan_get_h_f64(pl, (double*)H, p->r, 0));
        } else {
            SN_STEP(s, snmf_plan_get_w_f32(pl, (float*)W, p->F, 0));
            SN_STEP(s, snmf_plan_get_h_f32(pl, (float*)H, p->r, 0));
        }
    }
    if (s == SNMF_OK) SN_STEP(s, snmf_plan_get_objective(pl, div_out, cost_out, n_iter_out));
    snmf_plan_destroy(pl);
    return s;
}

extern "C" int snmf_sparse_nmf_f64(snmf_ctx* ctx, const snmf_params* p, const double* V, int64_t ldV, double* W,
                                   double* H, const double* sparsity, double* div_out, double* cost_out,
                                   int32_t* n_iter_out) {
    return sparse_nmf_impl<double>(ctx, p, V, ldV, W, H, W, H, sparsity, div_out, cost_out, n_iter_out, false);
}
extern "C" int snmf_sparse_nmf_f32(snmf_ctx* ctx, const snmf_params* p, const float* V, int64_t ldV, float* W,
                                   float* H, const float* sparsity, double* div_out, double* cost_out,
                                   int32_t* n_iter_out) {
    return sparse_nmf_impl<float>(ctx, p, V, ldV, W, H, W, H, sparsity, div_out, cost_out, n_iter_out, false);
}
// The same with the initial factors read-only and the results in arrays of their own: MATLAB's value semantics (inputs are
// never modified, outputs freshly allocated: SURVEY.md section 8b) without duplicating init_h first -- 14 ms per 205 MB on the host.
extern "C" int snmf_sparse_nmf_oop_f64(snmf_ctx* ctx, const snmf_params* p, const double* V, int64_t ldV, const double* W0,
                                       const double* H0, const double* sparsity, double* W, double* H, double* div_out,
                                       double* cost_out, int32_t* n_iter_out) {
    return sparse_nmf_impl<double>(ctx, p, V, ldV, W0, H0, W, H, sparsity, div_out, cost_out, n_iter_out, true);
}
extern "C" int snmf_sparse_nmf_oop_f32(snmf_ctx* ctx, const snmf_params* p, const float* V, int64_t ldV, const float* W0,
                                       const float* H0, const float* sparsity, float* W, float* H, double* div_out,
                                       double* cost_out, int32_t* n_iter_out) {
    return sparse_nmf_impl<float>(ctx, p, V, ldV, W0, H0, W, H, sparsity, div_out, cost_out, n_iter_out, true);
}

// ---- spectrogram front-end ---------------------------------------------------------------------
extern "C" int64_t snmf_stft_num_frames(const snmf_stft_params* sp, int64_t L) {
    if (!sp || sp->frameshift <= 0) return 0;
    // while size_crnt < length(s) - fftlen, size_crnt = 1 + i*shift   (src/stft_fft.m:15,:21,:35)
    const int64_t lim = L - sp->fftlength - 1;
    if (lim <= 0) return 0;
    return (lim + sp->frameshift - 1) / sp->frameshift;
}

int validate_stft(const snmf_stft_params* sp) {
    if (!sp || !sp->window) return fail(SNMF_ERR_INVALID, "stft params / window is NULL");
    const int N = sp->fftlength;
    if (N < 64 || N > 4096 || (N & (N - 1))) return fail(SNMF_ERR_UNSUPPORTED, "fftlength must be a power of two in [64,4096]");
    if (sp->framelength < 1 || sp->framelength > N) return fail(SNMF_ERR_INVALID, "framelength must be in [1, fftlength]");
    if (sp->frameshift < 1) return fail(SNMF_ERR_INVALID, "frameshift must be positive");
    if (sp->dcbin < 1)
        return fail(SNMF_ERR_UNSUPPORTED, "DCbin must be >= 1 (with DCbin = 0 a silent frame becomes an all-zero column, "
                                          "which run_basis_train.m:61 removes: data-dependent compaction is not implemented)");
    if (sp->dcbin > N / 2 + 1 || sp->splice < 0) return fail(SNMF_ERR_INVALID, "bad DCbin / Splice");
    return SNMF_OK;
}

// features into a device buffer `dst` (column t at dst + t*ld); scratch allocations are freed on return
int stft_to_device(snmf_ctx* ctx, const snmf_stft_params* sp, const float* samples, int64_t n_samples,
                          int samples_on_device, float* dst, int64_t ld, int64_t n_frames) {
    hipStream_t st = ctx->stream;
    const int N = sp->fftlength, K = N / 2 + 1, S = sp->splice;
    float *d_s = nullptr, *d_win = nullptr, *d_tmp = nullptr;
    float2* d_tw = nullptr;
    int rc = SNMF_OK;
    auto cleanup = [&]() {
        hipStreamSynchronize(st);
        if (d_s && !samples_on_device) hipFree(d_s);
        if (d_win) hipFree(d_win);
        if (d_tw) hipFree(d_tw);
        if (d_tmp) hipFree(d_tmp);
    };
#define FE_TRY(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            rc = fail(SNMF_ERR_NO_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
            cleanup();                                                            \
            return rc;                                                            \
        }                                                                         \
    } while (0)
    if (samples_on_device) d_s = const_cast<float*>(samples);
    else {
        FE_TRY(hipMalloc((void**)&d_s, (size_t)n_samples * 4));
        FE_TRY(hipMemcpyAsync(d_s, samples, (size_t)n_samples * 4, hipMemcpyHostToDevice, st));
    }
    std::vector<float> hw(sp->framelength);
    for (int i = 0; i < sp->framelength; ++i) hw[i] = (float)sp->window[i];
    std::vector<float2> htw(N / 2);
    for (int q = 0; q < N / 2; ++q) {
        const double ang = -2.0 * M_PI * (double)q / (double)N;
        htw[q] = make_float2((float)cos(ang), (float)sin(ang));
    }
    FE_TRY(hipMalloc((void**)&d_win, hw.size() * 4));
    FE_TRY(hipMalloc((void**)&d_tw, htw.size() * 8));
    FE_TRY(hipMemcpyAsync(d_win, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, st));
    FE_TRY(hipMemcpyAsync(d_tw, htw.data(), htw.size() * 8, hipMemcpyHostToDevice, st));
    StftArgs a{};
    a.s = d_s;
    a.n_samples = n_samples;
    a.sz = sp->framelength;
    a.shift = sp->frameshift;
    a.dcbin = sp->dcbin;
    a.preemph = (float)sp->preemph;
    a.win = d_win;
    a.tw = d_tw;
    a.powv = (float)sp->pow;
    a.n_frames = (int)n_frames;
    if (S == 0) {
        a.floorv = (float)sp->nonzerofloor;
        a.out = dst;
        a.ld = ld;
    } else {
        FE_TRY(hipMalloc((void**)&d_tmp, (size_t)K * n_frames * 4));
        a.floorv = 0.f;
        a.out = d_tmp;
        a.ld = K;
    }
    dim3 g((unsigned)n_frames), b(256);
    switch (N) {
        case 64: hipLaunchKernelGGL(k_stft<6>, g, b, 0, st, a); break;
        case 128: hipLaunchKernelGGL(k_stft<7>, g, b, 0, st, a); break;
        case 256: hipLaunchKernelGGL(k_stft<8>, g, b, 0, st, a); break;
        case 512: hipLaunchKernelGGL(k_stft<9>, g, b, 0, st, a); break;
        case 1024: hipLaunchKernelGGL(k_stft<10>, g, b, 0, st, a); break;
        case 2048: hipLaunchKernelGGL(k_stft<11>, g, b, 0, st, a); break;
        default: hipLaunchKernelGGL(k_stft<12>, g, b, 0, st, a); break;
    }
    FE_TRY(hipGetLastError());
    if (S > 0) {
        const size_t n = (size_t)(2 * S + 1) * K * n_frames;
        hipLaunchKernelGGL(k_splice, dim3(grid_for(n)), dim3(256), 0, st, d_tmp, (int64_t)K, K, (int)n_frames, S,
                           (float)sp->nonzerofloor, dst, ld);
        FE_TRY(hipGetLastError());
    }
#undef FE_TRY
    cleanup();
    return SNMF_OK;
}

extern "C" int snmf_stft_features_f32(snmf_ctx* ctx, const snmf_stft_params* sp, const float* samples,
                                      int64_t n_samples, int samples_on_device, float* V_out, int64_t ld,
                                      int out_on_device, int32_t* n_frames_out) {
    if (!ctx || !samples || !V_out) return fail(SNMF_ERR_INVALID, "NULL argument");
    SN_TRY(validate_stft(sp));
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t nfr = snmf_stft_num_frames(sp, n_samples);
    const int64_t F = (int64_t)(2 * sp->splice + 1) * (sp->fftlength / 2 + 1);
    if (n_frames_out) *n_frames_out = (int32_t)nfr;
    if (nfr <= 0) return SNMF_OK;
    if (ld < F) return fail(SNMF_ERR_INVALID, "ld < feature rows %lld", (long long)F);
    if (out_on_device) return stft_to_device(ctx, sp, samples, n_samples, samples_on_device, V_out, ld, nfr);
    float* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, (size_t)F * nfr * 4));
    int rc = stft_to_device(ctx, sp, samples, n_samples, samples_on_device, d_out, F, nfr);
    if (rc == SNMF_OK) {
        hipError_t e = hipMemcpy2D(V_out, (size_t)ld * 4, d_out, (size_t)F * 4, (size_t)F * 4, (size_t)nfr,
                                   hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(SNMF_ERR_NO_DEVICE, "hipMemcpy2D: %s", hipGetErrorString(e));
    }
    hipFree(d_out);
    return rc;
}

extern "C" int snmf_plan_set_v_from_audio_f32(snmf_plan* pl, const snmf_stft_params* sp, const float* samples,
                                              int64_t n_samples, int samples_on_device) {
    PLAN_CHECK(pl);
    if (!samples) return fail(SNMF_ERR_INVALID, "samples is NULL");
    SN_TRY(validate_stft(sp));
    HIP_TRY(hipSetDevice(pl->ctx->device));
    const int64_t nfr = snmf_stft_num_frames(sp, n_samples);
    const int64_t F = (int64_t)(2 * sp->splice + 1) * (sp->fftlength / 2 + 1);
    if (F != pl->p.F || nfr != pl->p.T)
        return fail(SNMF_ERR_DIM, "audio gives %lld x %lld features, the plan is %d x %d", (long long)F, (long long)nfr,
                    pl->p.F, pl->p.T);
    // pad rows/columns of the resident V stay zero; features are >= nonzerofloor^... > 0, the
    // solver's own floor (src/sparse_nmf.m:169) is applied by clamping the floor value from below
    HIP_TRY(hipMemsetAsync(pl->V, 0, (size_t)pl->Fp * pl->Tp * 4, pl->ctx->stream));
    SN_TRY(stft_to_device(pl->ctx, sp, samples, n_samples, samples_on_device, pl->V, pl->Fp, nfr));
    if (pl->p.floor_v) {
        // v = max(v, 1e-9) on the real entries (a no-op whenever nonzerofloor >= 1e-9)
        hipLaunchKernelGGL(k_floor_real, dim3(grid_for((size_t)pl->Fp * pl->p.T)), dim3(256), 0, pl->ctx->stream, pl->V,
                           pl->Fp, pl->p.F, pl->p.T, kFlr);
        HIP_TRY(hipGetLastError());
    }
    pl->have_v = true;
    return SNMF_OK;
}

extern "C" int snmf_mel_features_f32(snmf_ctx* ctx, const float* mel, int32_t M, int32_t n, int32_t K, const float* V,
                                     int64_t ldv, int32_t T, float* out, int64_t ldo, int on_device) {
    if (!ctx || !mel || !V || !out) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (M < 1 || n < 1 || K < 1 || T < 1 || ldv < (int64_t)K * n || ldo < (int64_t)K * M)
        return fail(SNMF_ERR_INVALID, "bad Mel projection sizes");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    float *d_mel = nullptr, *d_v = nullptr, *d_o = nullptr;
    HIP_TRY(hipMalloc((void**)&d_mel, (size_t)M * n * 4));
    HIP_TRY(hipMemcpyAsync(d_mel, mel, (size_t)M * n * 4, hipMemcpyHostToDevice, st));
    int rc = SNMF_OK;
    if (on_device) {
        d_v = const_cast<float*>(V);
        d_o = out;
    } else {
        if (hipMalloc((void**)&d_v, (size_t)ldv * T * 4) != hipSuccess || hipMalloc((void**)&d_o, (size_t)ldo * T * 4) != hipSuccess)
            rc = fail(SNMF_ERR_NOMEM, "hipMalloc failed");
        else if (hipMemcpy2DAsync(d_v, (size_t)ldv * 4, V, (size_t)ldv * 4, (size_t)K * n * 4, (size_t)T, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(SNMF_ERR_NO_DEVICE, "hipMemcpy2DAsync (Mel input) failed");
    }
    if (rc == SNMF_OK) {
        hipLaunchKernelGGL(k_mel, dim3(grid_for((size_t)K * M * T)), dim3(256), 0, st, d_mel, M, n, K, d_v, ldv, T, d_o, ldo);
        if (hipGetLastError() != hipSuccess) rc = fail(SNMF_ERR_NO_DEVICE, "k_mel launch failed");
        if (!on_device && rc == SNMF_OK &&
            hipMemcpy2DAsync(out, (size_t)ldo * 4, d_o, (size_t)ldo * 4, (size_t)K * M * 4, (size_t)T, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(SNMF_ERR_NO_DEVICE, "hipMemcpy2DAsync (Mel output) failed");
    }
    hipStreamSynchronize(st);
    hipFree(d_mel);
    if (!on_device) {
        if (d_v) hipFree(d_v);
        if (d_o) hipFree(d_o);
    }
    return rc;
}

// TF_DD feature transform (src/TF_DD.m, run_basis_train.m:64-67): see snmf_frontend.h.  X / out: F x T column-major,
// host or device (both the same side); out may alias X.
extern "C" int snmf_tf_dd_f32(snmf_ctx* ctx, double alpha_eta, int32_t F, int32_t T, const float* X, int64_t ldx, float* out,
                              int64_t ldo, int on_device) {
    if (!ctx || !X || !out) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (F < 1 || T < 1 || ldx < F || ldo < F) return fail(SNMF_ERR_INVALID, "bad TF_DD sizes");
    if (!(alpha_eta == alpha_eta)) return fail(SNMF_ERR_INVALID, "alpha_eta is NaN");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nch = (T + kDdChunk - 1) / kDdChunk;
    float *d_x = nullptr, *d_o = nullptr;
    double* d_c = nullptr;
    int rc = SNMF_OK;
    if (hipMalloc((void**)&d_c, (size_t)nch * F * sizeof(double)) != hipSuccess) return fail(SNMF_ERR_NOMEM, "hipMalloc failed");
    if (on_device) {
        d_x = const_cast<float*>(X);
        d_o = out;
    } else if (hipMalloc((void**)&d_x, (size_t)ldx * T * 4) != hipSuccess || hipMalloc((void**)&d_o, (size_t)ldo * T * 4) != hipSuccess) {
        rc = fail(SNMF_ERR_NOMEM, "hipMalloc failed");
    } else {
        // only the F x T entries belong to the caller (ldx may exceed F, and the last column then holds just F elements)
        if (hipMemcpy2DAsync(d_x, (size_t)ldx * 4, X, (size_t)ldx * 4, (size_t)F * 4, (size_t)T, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(SNMF_ERR_NO_DEVICE, "hipMemcpy2DAsync (TF_DD input) failed");
    }
    if (rc == SNMF_OK) {
        const dim3 g(nch, (F + 255) / 256);
        hipLaunchKernelGGL(k_tfdd_carry, g, dim3(256), 0, st, (const float*)d_x, ldx, (int)F, (int)T, alpha_eta, d_c);
        hipLaunchKernelGGL(k_tfdd_state, dim3((F + 255) / 256), dim3(256), 0, st, (const float*)d_x, (int)F, (int)T, alpha_eta, d_c);
        hipLaunchKernelGGL(k_tfdd_apply, g, dim3(256), 0, st, (const float*)d_x, ldx, (int)F, (int)T, alpha_eta, (const double*)d_c, d_o, ldo);
        if (hipGetLastError() != hipSuccess) rc = fail(SNMF_ERR_NO_DEVICE, "TF_DD launch failed");
        // (rows F .. ldo-1 of the caller's out are padding the header does not promise: left untouched)
        if (!on_device && rc == SNMF_OK &&
            hipMemcpy2DAsync(out, (size_t)ldo * 4, d_o, (size_t)ldo * 4, (size_t)F * 4, (size_t)T, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(SNMF_ERR_NO_DEVICE, "hipMemcpy2DAsync (TF_DD output) failed");
    }
    hipStreamSynchronize(st);
    hipFree(d_c);
    if (!on_device) {
        if (d_x) hipFree(d_x);
        if (d_o) hipFree(d_o);
    }
    return rc;
}

