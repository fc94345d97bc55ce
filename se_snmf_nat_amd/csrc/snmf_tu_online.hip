// snmf_tu_online.hip -- the online separation loop behind the C ABI (snmf_online_*), kernels in snmf_online.h (snmf_internal.h).
#include "snmf_internal.h"

// ---- online separation loop (include/snmf.h: snmf_online_*) -------------------------------------
// Host side of src/bnmf_sep_event_RT_IS16.m + the frame loop of src/NTF_sep_event_RT.m:54-135.  The
// host only sequences launches: per frame it reads one 32-byte status (did the adaptation condition
// fire?) and, when it did, runs the W-only adaptation solve through the engine's ordinary plan.
#include "snmf_online.h"

constexpr size_t kTraceCap = 1u << 16;  // diagnostics ring: the newest 65536 frames (~11 min at 100 frames/s)
struct snmf_online {
    snmf_ctx* ctx = nullptr;
    snmf_online_params p{};
    int F = 0, r = 0, N = 0, nov = 0;
    int Fs = 0;               // rows of the solves: F, or F_order in Mel mode
    int mel = 0, mel_conv = 0, n1 = 0;  // B_sep_mode = 'Mel' (snmf_online_set_mel)
    float *melmat = nullptr, *Bmf = nullptr, *Ymel = nullptr;
    double *Bm = nullptr, *Bmtmp = nullptr;  // [n1 x r] Mel dictionaries [B_Mel_x | B_Mel_d], fp64 like B
    snmf_plan* hp = nullptr;  // frame solve: Fs x 1, rank r, H-only
    snmf_plan* ap = nullptr;  // adaptation solve: F x m_a, rank R_a, W-only
    snmf_plan* hsemi = nullptr;  // semi-supervised frame solve (basis_update_N / _E): generic path, W reset every frame
    snmf_plan* hb = nullptr;  // fixed dictionary (no adaptation): the frame solves of a whole batch in one launch
    DevState* bst = nullptr;
    double *bdiv = nullptr, *bcost = nullptr;
    OnlineStatus* bstatus = nullptr;
    float *recon1 = nullptr, *breco = nullptr;
    // cooperative single-launch adaptation solve (k_wadapt)
    bool wadapt = false;
    int wa_nwg = 0;
    size_t wa_lds = 0;
    double *wa_W = nullptr, *wa_p1 = nullptr, *wa_p2 = nullptr, *wa_cost = nullptr;
    int* wa_nit = nullptr;
    unsigned* wa_bar = nullptr;  // B_x*A_x | B_d*A_d from the frame solve: one frame / a batch
    double *B = nullptr, *Bfix = nullptr, *Btmp = nullptr;  // fp64 like the engine's W master copy (k_wapply)
    float *Bf = nullptr;                                    // fp32 mirror of B for the reconstructions
    float *H0 = nullptr, *lambda_dav = nullptr, *Xm_tilde = nullptr,
          *r_blk = nullptr, *ldblk = nullptr, *adblk = nullptr, *Vad = nullptr, *Had = nullptr, *win_s = nullptr,
          *win_i = nullptr, *syn_tail = nullptr, *syn_tail_x = nullptr, *syn_tail_d = nullptr;
    float2* tw = nullptr;
    uint8_t* rup = nullptr;
    OnlineDev* dev = nullptr;
    OnlineStatus* status = nullptr;
    DevState* hst = nullptr;
    double *hdiv = nullptr, *hcost = nullptr;
    OnlineStatus* h_status = nullptr;  // pinned
    // per-call buffers (grown on demand)
    int cap_frames = 0;
    float *sig = nullptr, *Ym = nullptr, *Xt = nullptr, *Xh = nullptr, *Dh = nullptr, *syn = nullptr, *outf = nullptr;
    float2* Yph = nullptr;
    int16_t* out16 = nullptr;
    // host state of the driver loop
    std::vector<float> pending, hist;
    int64_t l = 0;  // frames processed
    bool finished = false;
    bool failed = false;  // a device batch failed midway: frame counter, history and rings are no longer consistent
    std::deque<snmf_online_frame> trace;  // bounded: the newest kTraceCap frames (a real-time stream runs for days)
};

static void online_free_call_buffers(snmf_online* o) {
    void* ptrs[] = {o->sig, o->Ym, o->Xt, o->Xh, o->Dh, o->syn, o->outf, o->Yph, o->out16, o->bst, o->bdiv, o->bcost, o->bstatus, o->breco, o->Ymel};
    o->Ymel = nullptr;
    o->breco = nullptr;
    if (o->hb) {
        snmf_plan_destroy(o->hb);
        o->hb = nullptr;
    }
    o->bst = nullptr;
    o->bdiv = o->bcost = nullptr;
    o->bstatus = nullptr;
    for (void* q : ptrs)
        if (q) hipFree(q);
    o->sig = o->Ym = o->Xt = o->Xh = o->Dh = o->syn = o->outf = nullptr;
    o->Yph = nullptr;
    o->out16 = nullptr;
    o->cap_frames = 0;
}

extern "C" void snmf_online_destroy(snmf_online* o) {
    if (!o) return;
    hipSetDevice(o->ctx->device);
    hipStreamSynchronize(o->ctx->stream);
#ifdef SNMF_PROF_WA
    {
        unsigned long long pf[10] = {0};
        hipMemcpyFromSymbol(pf, HIP_SYMBOL(g_wa_prof), sizeof pf);
        if (pf[9]) {
            static const char* nm[8] = {"setup", "product Lam'", "product G + partial sums", "exchange A", "W update + partial sums", "exchange B", "normalise", "store"};
            fprintf(stderr, "[SNMF_PROF_WA] %llu solves, %.2f loops per solve; k cycles per solve:", pf[9], (double)pf[8] / pf[9]);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%.1f", nm[i], (double)pf[i] / pf[9] * 1e-3);
            fprintf(stderr, "\n");
            unsigned long long z[10] = {0};
            hipMemcpyToSymbol(HIP_SYMBOL(g_wa_prof), z, sizeof z);
        }
    }
#endif
    if (o->hp) snmf_plan_destroy(o->hp);
    if (o->ap) snmf_plan_destroy(o->ap);
    if (o->hsemi) snmf_plan_destroy(o->hsemi);
    online_free_call_buffers(o);
    void* ptrs[] = {o->B,   o->Bfix, o->Btmp,  o->H0,    o->lambda_dav, o->Xm_tilde, o->r_blk, o->ldblk, o->adblk,  o->Vad,
                    o->Had, o->win_s, o->win_i, o->syn_tail, o->tw,       o->rup,      o->dev,   o->status, o->hst,   o->hdiv,
                    o->hcost, o->syn_tail_x, o->syn_tail_d, o->Bf, o->recon1, o->wa_W, o->wa_p1, o->wa_p2, o->wa_cost, o->wa_nit, o->wa_bar, o->melmat, o->Bmf, o->Bm, o->Bmtmp};
    for (void* q : ptrs)
        if (q) hipFree(q);
    if (o->h_status) hipHostFree(o->h_status);
    delete o;
}

static int online_validate(const snmf_online_params* p) {
    if (!p) return fail(SNMF_ERR_INVALID, "online params is NULL");
    const int N = p->fftlength;
    if (N < 64 || N > 4096 || (N & (N - 1))) return fail(SNMF_ERR_UNSUPPORTED, "fftlength must be a power of two in [64,4096]");
    if (p->framelength < 1 || p->framelength > N || p->frameshift < 1 || p->frameshift > p->framelength)
        return fail(SNMF_ERR_INVALID, "need 1 <= frameshift <= framelength <= fftlength");
    const int F = N / 2 + 1;
    if (p->dcbin < 0 || p->dcbin > F || p->dcbin_back < 0 || p->dcbin_back > F || p->delay < 0)
        return fail(SNMF_ERR_INVALID, "bad DCbin / DCbin_back / delay");
    if (p->R_x < 1 || p->R_d < 1) return fail(SNMF_ERR_INVALID, "R_x and R_d must be positive");
    if (p->max_iter < 1) return fail(SNMF_ERR_INVALID, "max_iter must be positive");
    if (p->enhance_method != 0 && p->enhance_method != 1) return fail(SNMF_ERR_INVALID, "enhance_method: 0 Wiener, 1 MMSE");
    if (p->blk_sparse) {
        if (p->blk_gap < 1 || p->blk_gap % 2 == 0) return fail(SNMF_ERR_INVALID, "blk_gap must be odd (src/blk_sparse.m:4)");
        if (p->P_len_k < 2 || p->P_len_k % 2 || p->P_len_l < 1) return fail(SNMF_ERR_INVALID, "P_len_k must be even and >= 2, P_len_l >= 1");
        if (p->P_len_k + p->dcbin > F) return fail(SNMF_ERR_INVALID, "P_len_k + DCbin exceeds the number of bins");
    }
    if (p->adapt_train_N) {
        if (p->R_a < 1 || p->R_a > p->R_d || p->m_a < 1) return fail(SNMF_ERR_INVALID, "need 1 <= R_a <= R_d and m_a >= 1");
    }
    return SNMF_OK;
}

// (re)create the resident solves for o->Fs rows: the frame solve, the optional semi-supervised variant, the
// adaptation plan and the cooperative adaptation kernel's buffers
static int online_make_solvers(snmf_online* o) {
    const snmf_online_params* p = &o->p;
    snmf_ctx* ctx = o->ctx;
    const int F = o->Fs, r = o->r;
    const int Ra = p->adapt_train_N ? p->R_a : 1, ma = p->adapt_train_N ? p->m_a : 1;
    hipStreamSynchronize(ctx->stream);
    for (snmf_plan** q : {&o->hp, &o->hsemi, &o->ap}) {
        if (*q) snmf_plan_destroy(*q);
        *q = nullptr;
    }
    for (void** q : {(void**)&o->wa_W, (void**)&o->wa_p1, (void**)&o->wa_p2, (void**)&o->wa_cost, (void**)&o->wa_nit, (void**)&o->wa_bar}) {
        if (*q) hipFree(*q);
        *q = nullptr;
    }
    o->wadapt = false;
    int s = SNMF_OK;
    auto A = [&](int v) { if (s == SNMF_OK) s = v; };
    // the two resident solves
    snmf_params hp{};
    hp.F = F; hp.T = 1; hp.r = r; hp.beta = p->beta_div; hp.max_iter = p->max_iter; hp.conv_eps = p->conv_eps;
    hp.cost_check = p->cost_check; hp.floor_v = 1; hp.sparsity_kind = SNMF_SPARSITY_SCALAR; hp.sparsity_scalar = p->sparsity;
    std::vector<uint8_t> zeros(std::max(r, Ra), 0), ones(std::max(r, Ra), 1);
    hp.w_update_ind = zeros.data();  // supervised (:139)
    hp.h_update_ind = ones.data();   // :148
    A(snmf_plan_create(ctx, &hp, &o->hp));
    // !small_ok (F + r too large for the persistent single-launch kernels, e.g. the exemplar setting R_x = R_d = 500 of
    // settings/bak_IS16_results/initial_setting_Exemplar.m:47-48): the frame solve runs through the ordinary plan loop
    if (p->basis_update_N || p->basis_update_E) {
        snmf_params sp = hp;
        std::vector<uint8_t> wm(r, 0);
        for (int k = 0; k < r; ++k) wm[k] = p->basis_update_N ? (k >= p->R_x) : (k < p->R_x);  // :125-131
        sp.w_update_ind = wm.data();
        A(snmf_plan_create(ctx, &sp, &o->hsemi));
    }
    if (p->adapt_train_N) {
        snmf_params ap = hp;
        ap.T = ma; ap.r = Ra;
        ap.w_update_ind = ones.data();   // :330 (the per-solve subset r_up is written on the device)
        ap.h_update_ind = zeros.data();  // :331
        A(snmf_plan_create(ctx, &ap, &o->ap));
    }
    auto D = [&](auto** ptr, size_t n) { if (s == SNMF_OK) s = dalloc(ptr, n); };
    if (p->adapt_train_N && p->beta_div == 1.0 && p->R_a <= kWaRP && !getenv("SNMF_NO_WADAPT")) {
        int coop = 0;
        hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device);
        o->wa_nwg = (F + kWaRB - 1) / kWaRB;
        o->wa_lds = (size_t)(kWaRB * kWaRP + 4 * kWaRP + 32 + 256) * 8 +
                    (size_t)(2 * kWaRB * kWaRP + kWaRP + 2 * kWaRB * p->m_a + p->R_a * p->m_a + p->m_a * (kWaRP + 1) +
                             ((p->m_a <= 128 && p->m_a % 4 == 0) ? p->R_a * 128 + p->m_a * kWaRP : 0)) * 4;  // (+ the products' operand images)
        o->wadapt = coop != 0 && o->wa_nwg <= ctx->n_cu && o->wa_nwg <= 2 * kWaQ && o->wa_lds <= 160 * 1024;
        if (o->wadapt) {
            D(&o->wa_W, (size_t)p->R_a * F);
            D(&o->wa_p1, (size_t)o->wa_nwg * (kWaRP + 1));
            D(&o->wa_p2, (size_t)o->wa_nwg * 2 * kWaRP);
            D(&o->wa_cost, (size_t)p->max_iter);
            D(&o->wa_nit, (size_t)1);
            D(&o->wa_bar, (size_t)1);
        }
    }
    return s;
}

extern "C" int snmf_online_create(snmf_ctx* ctx, const snmf_online_params* p, const float* Bx, const float* Bd, const float* H0,
                                  const float* Ad0, const float* win_stft, const float* win_istft, snmf_online** out) {
    if (!ctx || !out || !Bx || !Bd || !H0 || !win_stft || !win_istft) return fail(SNMF_ERR_INVALID, "NULL argument");
    *out = nullptr;
    SN_TRY(online_validate(p));
    if (p->adapt_train_N && !Ad0) return fail(SNMF_ERR_INVALID, "Ad_blk0 is required when adapt_train_N is set");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    snmf_online* o = new snmf_online();
    o->ctx = ctx;
    o->p = *p;
    const int N = p->fftlength, F = N / 2 + 1, r = p->R_x + p->R_d, sz = p->framelength, hop = p->frameshift;
    const int Ra = p->adapt_train_N ? p->R_a : 1, ma = p->adapt_train_N ? p->m_a : 1, Pl = p->blk_sparse ? p->P_len_l : 1;
    o->F = F;
    o->r = r;
    o->N = N;
    o->nov = (sz + hop - 1) / hop;
    int s = SNMF_OK;
    auto A = [&](int v) { if (s == SNMF_OK) s = v; };
    o->Fs = F;
    A(online_make_solvers(o));
    auto D = [&](auto** ptr, size_t n) { if (s == SNMF_OK) s = dalloc(ptr, n); };
    D(&o->B, (size_t)F * r); D(&o->Bfix, (size_t)F * p->R_d); D(&o->Btmp, (size_t)F * p->R_d); D(&o->H0, (size_t)r);
    D(&o->Bf, (size_t)F * r);
    D(&o->recon1, (size_t)2 * F);
    D(&o->lambda_dav, (size_t)F); D(&o->Xm_tilde, (size_t)F); D(&o->r_blk, (size_t)F * Pl); D(&o->ldblk, (size_t)F * ma);
    D(&o->adblk, (size_t)Ra * ma); D(&o->Vad, (size_t)F * ma); D(&o->Had, (size_t)Ra * ma); D(&o->win_s, (size_t)sz);
    D(&o->win_i, (size_t)sz); D(&o->syn_tail, (size_t)std::max(1, o->nov - 1) * sz); D(&o->tw, (size_t)N / 2);
    if (p->class_outputs) {
        D(&o->syn_tail_x, (size_t)std::max(1, o->nov - 1) * sz);
        D(&o->syn_tail_d, (size_t)std::max(1, o->nov - 1) * sz);
    }
    D(&o->rup, (size_t)Ra); D(&o->dev, (size_t)1); D(&o->status, (size_t)1); D(&o->hst, (size_t)1);
    D(&o->hdiv, (size_t)p->max_iter); D(&o->hcost, (size_t)p->max_iter);
    if (s == SNMF_OK && hipHostMalloc((void**)&o->h_status, sizeof(OnlineStatus)) != hipSuccess) A(fail(SNMF_ERR_NOMEM, "hipHostMalloc"));
    if (s != SNMF_OK) {
        snmf_online_destroy(o);
        return s;
    }
    std::vector<float2> htw(N / 2);
    for (int q = 0; q < N / 2; ++q) {
        const double ang = -2.0 * M_PI * (double)q / (double)N;
        htw[q] = make_float2((float)cos(ang), (float)sin(ang));
    }
    OnlineDev d0{0, 1, 0, 0};  // update_switch = 1 (src/init_buff.m:42)
    std::vector<double> hB((size_t)F * r);
    for (size_t i = 0; i < (size_t)F * p->R_x; ++i) hB[i] = (double)Bx[i];
    for (size_t i = 0; i < (size_t)F * p->R_d; ++i) hB[(size_t)F * p->R_x + i] = (double)Bd[i];
    hipMemcpyAsync(o->B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->Bfix, hB.data() + (size_t)F * p->R_x, (size_t)F * p->R_d * 8, hipMemcpyHostToDevice, st);  // B_Mel_d in DFT mode (:328)
    hipMemcpyAsync(o->Bf, Bx, (size_t)F * p->R_x * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->Bf + (size_t)F * p->R_x, Bd, (size_t)F * p->R_d * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->H0, H0, (size_t)r * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->win_s, win_stft, (size_t)sz * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->win_i, win_istft, (size_t)sz * 4, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->tw, htw.data(), htw.size() * 8, hipMemcpyHostToDevice, st);
    hipMemcpyAsync(o->dev, &d0, sizeof d0, hipMemcpyHostToDevice, st);
    hipMemsetAsync(o->lambda_dav, 0, (size_t)F * 4, st);
    hipMemsetAsync(o->Xm_tilde, 0, (size_t)F * 4, st);
    hipMemsetAsync(o->r_blk, 0, (size_t)F * Pl * 4, st);
    hipMemsetAsync(o->ldblk, 0, (size_t)F * ma * 4, st);
    hipMemsetAsync(o->adblk, 0, (size_t)Ra * ma * 4, st);
    hipMemsetAsync(o->syn_tail, 0, (size_t)std::max(1, o->nov - 1) * sz * 4, st);
    if (p->class_outputs) {
        hipMemsetAsync(o->syn_tail_x, 0, (size_t)std::max(1, o->nov - 1) * sz * 4, st);
        hipMemsetAsync(o->syn_tail_d, 0, (size_t)std::max(1, o->nov - 1) * sz * 4, st);
    }
    hipMemsetAsync(o->rup, 0, (size_t)Ra, st);
    if (p->adapt_train_N) hipMemcpyAsync(o->adblk, Ad0, (size_t)Ra * ma * 4, hipMemcpyHostToDevice, st);  // column-major R_a x m_a
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        snmf_online_destroy(o);
        return fail(SNMF_ERR_NO_DEVICE, "online create: %s", hipGetErrorString(e));
    }
    A(set_w<double>(o->hp, o->B, F, 1));
    if (s != SNMF_OK) {
        snmf_online_destroy(o);
        return s;
    }
    o->hist.assign((size_t)(sz - hop), 0.f);
    *out = o;
    return SNMF_OK;
}

// B_sep_mode = 'Mel' (src/bnmf_sep_event_RT_IS16.m:106-120, src/init_buff.m:45-47): the solves run on Mel features
extern "C" int snmf_online_set_mel(snmf_online* o, int32_t F_order, int32_t mel_conv, const float* melmat, const float* BMx,
                                   const float* BMd) {
    if (!o || !melmat || !BMx || !BMd) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (o->l != 0 || !o->pending.empty()) return fail(SNMF_ERR_STATE, "snmf_online_set_mel must precede the first process call");
    if (F_order < 2 || F_order > o->F) return fail(SNMF_ERR_INVALID, "F_order must be in [2, fftlength/2+1]");
    HIP_TRY(hipSetDevice(o->ctx->device));
    hipStream_t st = o->ctx->stream;
    const int n1 = F_order, r = o->r, F = o->F, Rx = o->p.R_x, Rd = o->p.R_d;
    o->mel = 1;
    o->mel_conv = mel_conv != 0;
    o->n1 = n1;
    o->Fs = n1;
    SN_TRY(online_make_solvers(o));
    for (void** q : {(void**)&o->melmat, (void**)&o->Bmf, (void**)&o->Bm, (void**)&o->Bmtmp}) {
        if (*q) hipFree(*q);
        *q = nullptr;
    }
    SN_TRY(dalloc(&o->melmat, (size_t)n1 * F));
    SN_TRY(dalloc(&o->Bmf, (size_t)n1 * r));
    SN_TRY(dalloc(&o->Bm, (size_t)n1 * r));
    SN_TRY(dalloc(&o->Bmtmp, (size_t)n1 * Rd));
    std::vector<double> hB((size_t)n1 * r);
    for (size_t i = 0; i < (size_t)n1 * Rx; ++i) hB[i] = (double)BMx[i];
    for (size_t i = 0; i < (size_t)n1 * Rd; ++i) hB[(size_t)n1 * Rx + i] = (double)BMd[i];
    HIP_TRY(hipMemcpyAsync(o->Bm, hB.data(), hB.size() * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->Bmf, BMx, (size_t)n1 * Rx * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->Bmf + (size_t)n1 * Rx, BMd, (size_t)n1 * Rd * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(o->melmat, melmat, (size_t)n1 * F * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    SN_TRY(set_w<double>(o->hp, o->Bm, n1, 1));
    online_free_call_buffers(o);  // batch buffers depend on the solve geometry
    return SNMF_OK;
}

/* Current B_Mel_d (n1 x R_d): what the Mel-mode adaptation updates (src/bnmf_sep_event_RT_IS16.m:318). */
extern "C" int snmf_online_get_mel_basis_f32(snmf_online* o, float* BMd, int64_t ld) {
    if (!o || !BMd) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (!o->mel) return fail(SNMF_ERR_STATE, "not in Mel mode");
    if (ld < o->n1) return fail(SNMF_ERR_INVALID, "ld < F_order");
    HIP_TRY(hipSetDevice(o->ctx->device));
    HIP_TRY(hipStreamSynchronize(o->ctx->stream));
    HIP_TRY(hipMemcpy2D(BMd, (size_t)ld * 4, o->Bmf + (size_t)o->n1 * o->p.R_x, (size_t)o->n1 * 4, (size_t)o->n1 * 4, (size_t)o->p.R_d,
                        hipMemcpyDeviceToHost));
    return SNMF_OK;
}

static int online_reserve(snmf_online* o, int n) {
    if (n <= o->cap_frames) return SNMF_OK;
    hipStreamSynchronize(o->ctx->stream);
    online_free_call_buffers(o);
    const int cap = std::max(n, 64);
    const size_t F = o->F, sz = o->p.framelength, hop = o->p.frameshift;
    SN_TRY(dalloc(&o->sig, (sz - hop) + (size_t)cap * hop));
    SN_TRY(dalloc(&o->Ym, F * cap));
    SN_TRY(dalloc(&o->Yph, F * cap));
    if (o->mel) SN_TRY(dalloc(&o->Ymel, (size_t)o->n1 * cap));
    SN_TRY(dalloc(&o->Xt, F * cap));
    if (o->p.class_outputs) {
        SN_TRY(dalloc(&o->Xh, F * cap));
        SN_TRY(dalloc(&o->Dh, F * cap));
    }
    SN_TRY(dalloc(&o->syn, (size_t)(cap + o->nov - 1) * sz));
    SN_TRY(dalloc(&o->outf, (size_t)cap * hop));
    SN_TRY(dalloc(&o->out16, (size_t)cap * hop));
    if (!o->p.adapt_train_N && !o->hsemi && o->hp->small_ok) {
        snmf_params bp = o->hp->p;
        bp.T = cap;
        std::vector<uint8_t> zeros(o->r, 0), ones(o->r, 1);
        bp.w_update_ind = zeros.data();
        bp.h_update_ind = ones.data();
        SN_TRY(snmf_plan_create(o->ctx, &bp, &o->hb));
        SN_TRY(set_w<double>(o->hb, o->mel ? o->Bm : o->B, o->Fs, 1));
        SN_TRY(dalloc(&o->bst, (size_t)cap));
        SN_TRY(dalloc(&o->bdiv, (size_t)cap * o->p.max_iter));
        SN_TRY(dalloc(&o->bcost, (size_t)cap * o->p.max_iter));
        SN_TRY(dalloc(&o->bstatus, (size_t)cap));
        SN_TRY(dalloc(&o->breco, (size_t)cap * 2 * o->Fs));
    }
    o->cap_frames = cap;
    return SNMF_OK;
}

template <typename K>
static void launch_by_logn(K&& f, int N) {
    switch (N) {
        case 64: f(std::integral_constant<int, 6>{}); break;
        case 128: f(std::integral_constant<int, 7>{}); break;
        case 256: f(std::integral_constant<int, 8>{}); break;
        case 512: f(std::integral_constant<int, 9>{}); break;
        case 1024: f(std::integral_constant<int, 10>{}); break;
        case 2048: f(std::integral_constant<int, 11>{}); break;
        default: f(std::integral_constant<int, 12>{}); break;
    }
}

// the frame solve (:148-154): V = Ym (device), W resident, H0 the fixed start; leaves A in hp->H[0]
static int online_solve_frame(snmf_online* o, const float* dV, const float** A_out, const DevState** st_out, const float** recon_out) {
    hipStream_t st = o->ctx->stream;
    if (o->hsemi) {
        // semi-supervised: an ordinary solve with part of W free; init_w = [B_DFT_x, B_DFT_d] again every frame (:140-146)
        snmf_plan* ps = o->hsemi;
        SN_TRY(set_v<float>(ps, dV, o->Fs, 1));
        SN_TRY(set_w<double>(ps, o->mel ? o->Bm : o->B, o->Fs, 1));
        SN_TRY(set_h<float>(ps, o->H0, o->r, 1));
        SN_TRY(snmf_plan_init(ps));
        SN_TRY(snmf_plan_run(ps, o->p.max_iter, nullptr));
        int idx = 0;
        SN_TRY(result_h_index(ps, &idx));
        *A_out = ps->H[idx];
        *st_out = ps->st;
        *recon_out = nullptr;
        return SNMF_OK;
    }
    snmf_plan* pl = o->hp;
    if (!pl->small_ok) {
        // large rank: the ordinary plan loop (16-frame tiles), one solve per frame; reconstructions are formed in k_opost
        SN_TRY(set_v<float>(pl, dV, o->Fs, 1));
        SN_TRY(set_h<float>(pl, o->H0, o->r, 1));
        SN_TRY(snmf_plan_init(pl));  // W and its norms are reused unless the adaptation has replaced the dictionary
        SN_TRY(snmf_plan_run(pl, o->p.max_iter, nullptr));
        int idx = 0;
        SN_TRY(result_h_index(pl, &idx));
        *A_out = pl->H[idx];
        *st_out = pl->st;
        *recon_out = nullptr;
        return SNMF_OK;
    }
    *A_out = pl->H[0];
    *st_out = o->hst;
    *recon_out = (pl->frame_fb && (!o->mel || o->mel_conv)) ? o->recon1 : nullptr;  // Mel without MelConv: B_DFT*A, formed in k_opost
    const size_t nVp = (size_t)pl->Fp * pl->Tp;
    hipLaunchKernelGGL(k_pack<float>, dim3(grid_for(nVp)), dim3(256), 0, st, dV, (int64_t)o->Fs, o->Fs, 1, pl->V, pl->Fp, pl->Tp, kFlr,
                       pl->p.floor_v ? 1 : 0);
    HIP_TRY(hipGetLastError());
    pl->have_v = true;
    if (pl->w_dirty) SN_TRY(launch_wapply(pl, pl->stats, 0, false, true));  // wn, w./wn (:157-159)
    pl->w_dirty = false;
    pl->cur = 0;
    hipLaunchKernelGGL(k_tile_h0<float>, dim3(grid_for((size_t)pl->rp)), dim3(256), 0, st, (const float*)o->H0, pl->wn, o->r, pl->rp, 1,
                       1, pl->H[0]);  // h .* wn' (:160)
    HIP_TRY(hipGetLastError());
    pl->have_h = true;
    pl->inited = false;
    HIP_TRY(hipMemsetAsync(o->hst, 0, sizeof(DevState), st));
    return launch_small(pl, 1, 1, o->hdiv, o->hcost, o->hst, (o->mel && !o->mel_conv) ? nullptr : o->recon1, o->p.R_x);
}

// :296-336 once the status says the solve is due
static int online_adapt(snmf_online* o, int32_t* iters) {
    const snmf_online_params& p = o->p;
    hipStream_t st = o->ctx->stream;
    snmf_plan* ap = o->ap;
    // DFT mode adapts B_DFT_d on lambda_d_blk (:320-338); Mel mode adapts B_Mel_d on melmat*lambda_d_blk (:298-318)
    const int Fs = o->Fs;
    double* Ball = o->mel ? o->Bm : o->B;
    double* Bd = Ball + (size_t)Fs * p.R_x;
    double* Btmp = o->mel ? o->Bmtmp : o->Btmp;
    float* mirror = (o->mel ? o->Bmf : o->Bf) + (size_t)Fs * p.R_x;
    const double* Bfix = o->mel ? Bd : o->Bfix;  // columns beyond R_a never change; :328 takes them from B_Mel_d
    if (o->mel) {
        hipLaunchKernelGGL(k_oprep_mel, dim3(p.m_a), dim3(256), 0, st, (const float*)o->ldblk, (const float*)o->adblk, (const uint8_t*)o->rup,
                           (const OnlineDev*)o->dev, (const float*)o->melmat, o->F, o->n1, p.R_a, p.m_a, o->Vad, o->Had, ap->w_ind);
    } else {
        const size_t n = (size_t)o->F * p.m_a + (size_t)p.R_a * p.m_a + p.R_a;
        hipLaunchKernelGGL(k_oprep, dim3(grid_for(n)), dim3(256), 0, st, (const float*)o->ldblk, (const float*)o->adblk,
                           (const uint8_t*)o->rup, (const OnlineDev*)o->dev, o->F, p.R_a, p.m_a, o->Vad, o->Had, ap->w_ind);
    }
    HIP_TRY(hipGetLastError());
    const double* Wres = nullptr;
    int ldw = 0;
    if (o->wadapt) {
        // the whole solve in one cooperative launch (k_wadapt)
        WAdaptArgs wa{};
        wa.V = o->Vad; wa.H = o->Had; wa.W0 = Bd; wa.w_ind = ap->w_ind; wa.Wout = o->wa_W; wa.part1 = o->wa_p1; wa.part2 = o->wa_p2;
        wa.costh = o->wa_cost; wa.n_iter_out = o->wa_nit; wa.F = Fs; wa.Ra = p.R_a; wa.ma = p.m_a; wa.max_iter = p.max_iter;
        wa.cost_check = p.cost_check; wa.sparsity = (float)p.sparsity; wa.flr = kFlr; wa.conv_eps = p.conv_eps;
        SN_TRY(ensure_dyn_lds(o->ctx->device, (const void*)k_wadapt, o->wa_lds));
        wa.bar = o->wa_bar;
        HIP_TRY(hipMemsetAsync(o->wa_bar, 0, 4, st));
        void* kargs[] = {&wa};
        if (hipLaunchCooperativeKernel((const void*)k_wadapt, dim3(o->wa_nwg), dim3(kWaNT), kargs, (unsigned)o->wa_lds, st) == hipSuccess) {
            HIP_TRY(hipMemcpyAsync(iters, o->wa_nit, 4, hipMemcpyDeviceToHost, st));
            // the solve's verdict is read BEFORE its W is merged into the dictionary: a timed-out grid barrier leaves
            // wa_W invalid, and B_d, its fp32 mirror and the frame-solve plan must not see it
            HIP_TRY(hipStreamSynchronize(st));
            if (*iters < 0) return fail(SNMF_ERR_INTERNAL, "adaptation kernel: grid barrier timed out (dictionary left untouched)");
            Wres = o->wa_W;
            ldw = Fs;
        } else {
            (void)hipGetLastError();  // cooperative launch refused (e.g. CUs not all available): generic path from now on
            o->wadapt = false;
        }
    }
    if (!Wres) {
        SN_TRY(set_v<float>(ap, o->Vad, Fs, 1));       // lambda_d_blk[_Mel] (floored at 1e-9 inside, sparse_nmf.m:169)
        SN_TRY(set_w<double>(ap, Bd, Fs, 1));          // init_w: first R_a noise columns (:332)
        SN_TRY(set_h<float>(ap, o->Had, p.R_a, 1));    // init_h (:333)
        SN_TRY(snmf_plan_init(ap));
        SN_TRY(snmf_plan_run(ap, p.max_iter, iters));
        Wres = ap->Wc;
        ldw = ap->Fp;
    }
    hipLaunchKernelGGL(k_oassemble, dim3(p.R_d), dim3(256), 0, st, (const double*)Bd, Wres, ldw, Bfix, (const uint8_t*)o->rup, Fs, p.R_a,
                       p.R_d, Btmp, mirror);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(Bd, Btmp, (size_t)Fs * p.R_d * 8, hipMemcpyDeviceToDevice, st));
    SN_TRY(set_w<double>(o->hp, Ball, Fs, 1));         // next frame's init_w (:140-146)
    return SNMF_OK;  // `iters` is already on the host (both paths synchronised when they read it)
}

// n frames whose samples are sig = [history | n hops] (host); appends the hops the driver would write
static int online_run_frames(snmf_online* o, const std::vector<float>& sig, int n, std::vector<float>* outf,
                             std::vector<int16_t>* out16, std::vector<float>* xh, std::vector<float>* dh) {
    const snmf_online_params& p = o->p;
    const int F = o->F, sz = p.framelength, hop = p.frameshift, nov = o->nov;
    hipStream_t st = o->ctx->stream;
    SN_TRY(online_reserve(o, n));
    HIP_TRY(hipMemcpyAsync(o->sig, sig.data(), sig.size() * 4, hipMemcpyHostToDevice, st));
    OStftArgs sa{};
    sa.sig = o->sig; sa.sz = sz; sa.hop = hop; sa.dcbin = p.dcbin; sa.preemph = (float)p.preemph; sa.win = o->win_s; sa.tw = o->tw;
    sa.powv = (float)p.pow; sa.floorv = (float)p.nonzerofloor; sa.Ym = o->Ym; sa.Yph = o->Yph; sa.ld = F; sa.n_frames = n;
    launch_by_logn([&](auto L) { hipLaunchKernelGGL(k_ostft<decltype(L)::value>, dim3(n), dim3(256), 0, st, sa); }, o->N);
    HIP_TRY(hipGetLastError());
    if (o->mel) {
        hipLaunchKernelGGL(k_omel_frame, dim3(n), dim3(256), (size_t)(o->n1 + 2) * 4, st, (const float*)o->Ym, (const float*)o->melmat, F, o->n1, n,
                           o->Ymel);
        HIP_TRY(hipGetLastError());
    }
    const size_t lds_post = (size_t)(o->r + 7 * F + 3 * o->n1) * 4;
    auto post_args = [&](int i, int64_t l) {
        OPostArgs a{};
        a.B = o->Bf; a.Ym = o->Ym + (size_t)i * F; a.lambda_dav = o->lambda_dav; a.Xm_tilde = o->Xm_tilde;
        a.r_blk = o->r_blk; a.ldblk = o->ldblk; a.adblk = o->adblk; a.rup = o->rup; a.dev = o->dev;
        a.Xt_out = o->Xt + (size_t)i * F;
        a.Xh_out = o->Xh ? o->Xh + (size_t)i * F : nullptr;
        a.Dh_out = o->Dh ? o->Dh + (size_t)i * F : nullptr;
        a.F = F; a.Rx = p.R_x; a.Rd = p.R_d; a.Ra = p.adapt_train_N ? p.R_a : 1; a.ma = p.adapt_train_N ? p.m_a : 1;
        a.Pl = p.blk_sparse ? p.P_len_l : 1; a.Pk = p.P_len_k; a.dcbin = p.dcbin; a.gap = p.blk_gap;
        a.l = (int)std::min<int64_t>(l, 1 << 30);
        a.blk_sparse = p.blk_sparse; a.adapt = p.adapt_train_N; a.wiener = p.enhance_method == 0; a.init_N_len = p.init_N_len;
        a.switch_at = (int)std::floor(p.overlap_m_a * p.m_a);
        a.alpha_p = (float)p.alpha_p; a.alpha_eta = (float)p.alpha_eta; a.alpha_d = (float)p.alpha_d; a.beta0 = (float)p.beta;
        a.beta_max = (float)p.beta_max; a.Ar_up = (float)p.Ar_up; a.flr = (float)p.nonzerofloor;
        a.n = 1;
        a.a_stride = 0;
        a.mel = o->mel; a.mel_conv = o->mel_conv; a.n1 = o->n1; a.melmat = o->melmat; a.Bmf = o->Bmf;
        a.Ymel = o->mel ? o->Ymel + (size_t)i * o->n1 : nullptr;
        a.recon_len = o->Fs;
        return a;
    };
    if (!p.adapt_train_N && !o->hsemi && o->hb) {
        // Fixed dictionary: nothing the host decides sits between frames.  All frame solves of the batch run
        // in ONE launch (one workgroup per frame, W normalised once), then ONE k_opost launch walks the
        // sequential post-filter recurrences.
        snmf_plan* pl = o->hb;
        const size_t nVp = (size_t)pl->Fp * pl->Tp;
        hipLaunchKernelGGL(k_pack<float>, dim3(grid_for(nVp)), dim3(256), 0, st, (const float*)(o->mel ? o->Ymel : o->Ym), (int64_t)o->Fs, o->Fs, n,
                           pl->V, pl->Fp, pl->Tp, kFlr, pl->p.floor_v ? 1 : 0);
        HIP_TRY(hipGetLastError());
        pl->have_v = true;
        if (pl->w_dirty) SN_TRY(launch_wapply(pl, pl->stats, 0, false, true));
        pl->w_dirty = false;
        pl->cur = 0;
        hipLaunchKernelGGL(k_tile_h0<float>, dim3(grid_for((size_t)n * pl->rp)), dim3(256), 0, st, (const float*)o->H0, pl->wn, o->r, pl->rp,
                           1, n, pl->H[0]);
        HIP_TRY(hipGetLastError());
        pl->have_h = true;
        pl->inited = false;
        HIP_TRY(hipMemsetAsync(o->bst, 0, (size_t)n * sizeof(DevState), st));
        SN_TRY(launch_small(pl, n, 1, o->bdiv, o->bcost, o->bst, (o->mel && !o->mel_conv) ? nullptr : o->breco, p.R_x));
        OPostArgs a = post_args(0, o->l + 1);
        a.A = pl->H[0]; a.hst = o->bst; a.status = o->bstatus; a.n = n; a.a_stride = pl->rp;
        a.recon = (pl->frame_fb && (!o->mel || o->mel_conv)) ? o->breco : nullptr;
        hipLaunchKernelGGL(k_opost, dim3(1), dim3(1024), lds_post, st, a);
        HIP_TRY(hipGetLastError());
        std::vector<OnlineStatus> hst((size_t)n);
        HIP_TRY(hipMemcpyAsync(hst.data(), o->bstatus, (size_t)n * sizeof(OnlineStatus), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (const OnlineStatus& hs : hst) {
            snmf_online_frame tr{};
            tr.n_iter = hs.n_iter; tr.trig = hs.trig; tr.n_up = hs.n_up; tr.beta = hs.beta; tr.A_x_mag = hs.A_x_mag; tr.A_d_mag = hs.A_d_mag;
            tr.Q_control = hs.Q_control;
            o->trace.push_back(tr);
            if (o->trace.size() > kTraceCap) o->trace.pop_front();
        }
    } else {
        for (int i = 0; i < n; ++i) {
            OPostArgs a = post_args(i, o->l + 1 + i);
            SN_TRY(online_solve_frame(o, o->mel ? o->Ymel + (size_t)i * o->n1 : o->Ym + (size_t)i * F, &a.A, &a.hst, &a.recon));
            a.status = o->status;
            hipLaunchKernelGGL(k_opost, dim3(1), dim3(1024), lds_post, st, a);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(o->h_status, o->status, sizeof(OnlineStatus), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const OnlineStatus hs = *o->h_status;
            snmf_online_frame tr{};
            tr.n_iter = hs.n_iter; tr.trig = hs.trig; tr.n_up = hs.n_up; tr.beta = hs.beta; tr.A_x_mag = hs.A_x_mag; tr.A_d_mag = hs.A_d_mag;
            tr.Q_control = hs.Q_control;
            if (hs.do_solve && hs.n_up > 0) {
                int32_t it = 0;
                SN_TRY(online_adapt(o, &it));
                tr.solved = 1;
                tr.adapt_iters = it;
            }
            o->trace.push_back(tr);
            if (o->trace.size() > kTraceCap) o->trace.pop_front();
        }
    }
    // inverse STFT of the n frames behind the nov-1 frames kept from the previous call, overlap-add
    const int l0 = (int)std::min<int64_t>(o->l + 1, 1 << 30);
    const int i_first = (int)std::max<int64_t>(0, (int64_t)p.delay + 1 - l0);
    const int n_out = std::max(0, n - i_first);
    auto synth = [&](const float* mag, std::vector<float>* of, std::vector<int16_t>* o16, bool keep_tail, float* tail) -> int {
        if (nov > 1) HIP_TRY(hipMemcpyAsync(o->syn, tail, (size_t)(nov - 1) * sz * 4, hipMemcpyDeviceToDevice, st));
        OIstftArgs ia{};
        ia.mag = mag; ia.ph = o->Yph; ia.ld = F; ia.n_frames = n; ia.sz = sz; ia.dcb = p.dcbin_back; ia.powv = (float)p.pow;
        ia.scale = (float)(p.overlapscale / (double)o->N); ia.preemph = (float)p.preemph; ia.win = o->win_i; ia.tw = o->tw;
        ia.syn = o->syn + (size_t)(nov - 1) * sz;
        launch_by_logn([&](auto L) { hipLaunchKernelGGL(k_oistft<decltype(L)::value>, dim3(n), dim3(256), 0, st, ia); }, o->N);
        HIP_TRY(hipGetLastError());
        if (n_out > 0) {
            hipLaunchKernelGGL(k_oola, dim3(grid_for((size_t)n_out * hop)), dim3(256), 0, st, (const float*)o->syn, n, l0, p.delay, sz, hop, nov,
                               i_first, n_out, o->outf, o16 ? o->out16 : nullptr);
            HIP_TRY(hipGetLastError());
            if (of) {
                const size_t at = of->size();
                of->resize(at + (size_t)n_out * hop);
                HIP_TRY(hipMemcpyAsync(of->data() + at, o->outf, (size_t)n_out * hop * 4, hipMemcpyDeviceToHost, st));
            }
            if (o16) {
                const size_t at = o16->size();
                o16->resize(at + (size_t)n_out * hop);
                HIP_TRY(hipMemcpyAsync(o16->data() + at, o->out16, (size_t)n_out * hop * 2, hipMemcpyDeviceToHost, st));
            }
        }
        if (keep_tail && nov > 1)
            HIP_TRY(hipMemcpyAsync(tail, o->syn + (size_t)n * sz, (size_t)(nov - 1) * sz * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return SNMF_OK;
    };
    SN_TRY(synth(o->Xt, outf, out16, true, o->syn_tail));
    if (o->p.class_outputs) {  // x_hat / d_hat of :350-361 (summed over the classes), same synthesis
        SN_TRY(synth(o->Xh, xh, nullptr, true, o->syn_tail_x));
        SN_TRY(synth(o->Dh, dh, nullptr, true, o->syn_tail_d));
    }
    o->l += n;
    return SNMF_OK;
}

extern "C" int snmf_online_process_f32(snmf_online* o, const float* pcm, int64_t n, int flush, float* xt_f32, int16_t* xt_i16,
                                       float* xh_f32, float* dh_f32, int64_t cap, int64_t* n_out) {
    if (!o) return fail(SNMF_ERR_INVALID, "online handle is NULL");
    if (n_out) *n_out = 0;
    if (n < 0 || (n > 0 && !pcm)) return fail(SNMF_ERR_INVALID, "pcm is NULL");
    if (o->finished) return fail(SNMF_ERR_STATE, "the stream was flushed; create a new separator");
    if (o->failed) return fail(SNMF_ERR_STATE, "an earlier call failed midway through a batch; the separator state is not reusable, create a new one");
    if ((xh_f32 || dh_f32) && !o->p.class_outputs) return fail(SNMF_ERR_STATE, "class outputs were not requested at creation");
    (void)hipGetLastError();  // clean sticky error state, see PLAN_CHECK
    HIP_TRY(hipSetDevice(o->ctx->device));
    const int sz = o->p.framelength, hop = o->p.frameshift;
    o->pending.insert(o->pending.end(), pcm, pcm + n);
    const int64_t nfr = (int64_t)(o->pending.size() / (size_t)hop);
    const int64_t tail_frames = flush ? o->p.delay + 1 : 0;
    const int64_t max_out = (nfr + tail_frames) * hop;
    if ((xt_f32 || xt_i16 || xh_f32 || dh_f32) && cap < max_out) {
        o->pending.resize(o->pending.size() - (size_t)n);
        return fail(SNMF_ERR_INVALID, "output capacity %lld < %lld samples", (long long)cap, (long long)max_out);
    }
    std::vector<float> of, ox, od;
    std::vector<int16_t> o16;
    const int64_t chunk = 4096;  // frames per device batch
    int64_t done = 0;
    while (done < nfr) {
        const int nb = (int)std::min(chunk, nfr - done);
        std::vector<float> sig(o->hist);
        sig.insert(sig.end(), o->pending.begin() + done * hop, o->pending.begin() + (done + nb) * hop);
        if (int rc = online_run_frames(o, sig, nb, xt_f32 ? &of : nullptr, xt_i16 ? &o16 : nullptr, xh_f32 ? &ox : nullptr, dh_f32 ? &od : nullptr)) {
            o->failed = true;  // frames of this call were consumed and the device state advanced: never retry on it
            return rc;
        }
        o->hist.assign(sig.end() - (sz - hop), sig.end());
        done += nb;
    }
    o->pending.erase(o->pending.begin(), o->pending.begin() + nfr * hop);
    if (flush) {
        // a partial hop is dropped and delay+1 all-zero frames follow (src/NTF_sep_event_RT.m:69-76)
        std::vector<float> sig((size_t)(sz - hop) + (size_t)tail_frames * hop, 0.f);
        if (int rc = online_run_frames(o, sig, (int)tail_frames, xt_f32 ? &of : nullptr, xt_i16 ? &o16 : nullptr, xh_f32 ? &ox : nullptr, dh_f32 ? &od : nullptr)) {
            o->failed = true;
            return rc;
        }
        o->pending.clear();
        o->finished = true;
    }
    if (xt_f32) std::memcpy(xt_f32, of.data(), of.size() * 4);
    if (xt_i16) std::memcpy(xt_i16, o16.data(), o16.size() * 2);
    if (xh_f32) std::memcpy(xh_f32, ox.data(), ox.size() * 4);
    if (dh_f32) std::memcpy(dh_f32, od.data(), od.size() * 4);
    if (n_out) *n_out = (int64_t)std::max(std::max(of.size(), o16.size()), std::max(ox.size(), od.size()));
    return SNMF_OK;
}

extern "C" int snmf_online_get_basis_f32(snmf_online* o, float* Bd, int64_t ld) {
    if (!o || !Bd) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (ld < o->F) return fail(SNMF_ERR_INVALID, "ld < F");
    HIP_TRY(hipSetDevice(o->ctx->device));
    HIP_TRY(hipStreamSynchronize(o->ctx->stream));
    HIP_TRY(hipMemcpy2D(Bd, (size_t)ld * 4, o->Bf + (size_t)o->F * o->p.R_x, (size_t)o->F * 4, (size_t)o->F * 4, (size_t)o->p.R_d,
                        hipMemcpyDeviceToHost));
    return SNMF_OK;
}

extern "C" int snmf_online_trace(snmf_online* o, snmf_online_frame* out, int64_t cap, int64_t* n) {
    if (!o) return fail(SNMF_ERR_INVALID, "online handle is NULL");
    if (n) *n = (int64_t)o->trace.size();
    if (out && cap > 0) std::copy_n(o->trace.begin(), (size_t)std::min<int64_t>(cap, (int64_t)o->trace.size()), out);
    return SNMF_OK;
}
