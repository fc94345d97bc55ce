// snmf_tu_dnmf.hip -- the reference's two training CALLERS of sparse_nmf behind the C ABI, device-resident (snmf_internal.h):
//
//   B_hat = run_basis_DNMF(x, d, B, p)                    run_basis_DNMF.m:1-55 (and its Mel twin run_basis_DNMF_Mel.m:1-95)
//   [B_DFT, B_Mel, A_DFT, A_Mel] = run_basis_train(...)   run_basis_train.m:58-91 for one event class
//
// Through three separate sparse_nmf calls the 3-solve loop moved A_hat (r x T) device -> host -> device and paid a full
// host round trip per solve (round 3: 0.326 s per call at BASELINE config 4 for 0.039 s of kernels).  Here the mixture /
// clean / noise features are uploaded ONCE each (or formed on the device from the two waveforms), A_hat stays in HBM between
// solve 1 and solves 2 / 3, and the uploads of X and D run on a second stream under solve 1.  Only B_hat (and A_hat when the
// caller asks for it) comes back.  The training entry forms TF_mag / TF_Mel in HBM, gathers the exemplar columns there and
// runs both solves without V ever crossing PCIe.
#include "snmf_internal.h"
#include "snmf_frontend.h"

#include <thread>

namespace snmf {

// ---- uniform draws for an initial H the caller does not supply ----------------------------------------------------------
// src/sparse_nmf.m:133-134: h = rand(r, n) after rand('seed', s) (:112-114).  MATLAB's legacy generator cannot be reproduced
// (SURVEY.md section 8c), so a caller that wants ITS OWN draws passes them (the MATLAB wrapper does); a caller that passes none
// gets Philox-4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = the column-major element index / 4: r x n numbers
// in (0, 1), reproducible on any host (se_snmf_nat_amd/api.py: philox_uniform restates it in NumPy for the tests), and 160 MB
// that do not cross PCIe at BASELINE config 4.
__device__ __host__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// H[t * rp + k] = u(k + r * (t_off + t)) for k < r, t < T; pads zero.  t_off: this plan's first column in the r x n matrix the draw
// is defined on (a shard of the frames starts from ITS columns of the unsharded draw: dist.h0_columns, snmf_run_basis_dnmf_multi_*).
// value = ((x >> 9) + 0.5) * 2^-23: every one of the 2^23 results is exactly representable and lies strictly inside (0, 1)
// (with 24 bits and + 0.5 the sum needed 25 bits: ties rounded to even, the largest draw came out as 1.0 -- round 4's advisor).
static __global__ __launch_bounds__(256) void k_rand_h(float* __restrict__ H, int rp, int r, int T, uint64_t seed, uint64_t t_off) {
    const uint64_t e_lo = (uint64_t)r * t_off, e_hi = e_lo + (uint64_t)r * (uint64_t)T;
    const uint64_t q_lo = e_lo / 4, n4 = (e_hi + 3) / 4 - q_lo;
    for (uint64_t qi = (uint64_t)blockIdx.x * 256 + threadIdx.x; qi < n4; qi += (uint64_t)gridDim.x * 256) {
        const uint64_t q = q_lo + qi;
        uint32_t o[4];
        philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        for (int j = 0; j < 4; ++j) {
            const uint64_t e = 4 * q + j;
            if (e >= e_lo && e < e_hi) {
                const uint64_t le = e - e_lo;
                H[(le / r) * rp + (le % r)] = ((float)(o[j] >> 9) + 0.5f) * (1.0f / 8388608.0f);
            }
        }
    }
}
static __global__ void k_add_sig(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] + d[i];
}
// out[j * rows + f] = V[idx[j] * ld + f]: the exemplar columns TF_mag(:, sample_idx) of run_basis_train.m:82-83
static __global__ void k_gather_cols(const float* __restrict__ V, int64_t ld, int rows, const int64_t* __restrict__ idx, int n,
                                     float* __restrict__ out) {
    const size_t tot = (size_t)rows * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % rows);
        const size_t j = i / rows;
        out[i] = V[(size_t)idx[j] * ld + f];
    }
}

}  // namespace snmf

// the second context of a context: same device, its own stream and transfer pipeline (uploads under a running solve)
static int aux_ctx(snmf_ctx* c, snmf_ctx** out) {
    if (!c->aux) SN_TRY(snmf_ctx_create(&c->aux, c->device));
    *out = c->aux;
    return SNMF_OK;
}

int rand_h(snmf_plan* pl, uint64_t seed, int64_t col0) {
    HIP_TRY(hipSetDevice(pl->ctx->device));
    hipStream_t st = pl->ctx->stream;
    const size_t nH = (size_t)pl->rp * pl->Tp;
    HIP_TRY(hipMemsetAsync(pl->H[0], 0, nH * 4, st));
    hipLaunchKernelGGL(k_rand_h, dim3(grid_for(((size_t)pl->p.r * pl->p.T + 3) / 4 + 1)), dim3(256), 0, st, pl->H[0], pl->rp, pl->p.r, pl->p.T, seed,
                       (uint64_t)std::max<int64_t>(0, col0));
    HIP_TRY(hipGetLastError());
    pl->have_h = true;
    pl->inited = false;
    pl->cur = 0;
    return SNMF_OK;
}

extern "C" int snmf_plan_set_h_random(snmf_plan* pl, uint64_t seed) {
    PLAN_CHECK(pl);
    return rand_h(pl, seed, 0);
}

namespace {

struct Loop3 {  // the three plans of run_basis_DNMF.m:36-55
    snmf_ctx *c1 = nullptr, *c2 = nullptr;
    snmf_plan *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
    std::vector<uint8_t> on, off;
    ~Loop3() {
        if (p1) snmf_plan_destroy(p1);
        if (p2) snmf_plan_destroy(p2);
        if (p3) snmf_plan_destroy(p3);
    }
};

int loop3_create(snmf_ctx* ctx, const snmf_params* p, int R_x, int R_d, Loop3* L) {
    if (!ctx || !p) return fail(SNMF_ERR_INVALID, "NULL argument");
    if (R_x < 1 || R_d < 1 || p->r != R_x + R_d) return fail(SNMF_ERR_DIM, "params->r = %d must equal R_x + R_d = %d + %d", p->r, R_x, R_d);
    if (p->sparsity_kind != SNMF_SPARSITY_SCALAR)
        // an r x 1 or r x n p.sparsity has R_x + R_d rows: solves 2 / 3 (R_x, R_d rows) are a MATLAB dimension error (src/sparse_nmf.m:192)
        return fail(SNMF_ERR_DIM, "run_basis_DNMF needs a scalar p.sparsity (its W-only solves have R_x / R_d rows)");
    L->c1 = ctx;
    SN_TRY(aux_ctx(ctx, &L->c2));
    L->on.assign(p->r, 1);
    L->off.assign(p->r, 0);
    snmf_params q = *p;
    q.w_update_ind = L->off.data();  // run_basis_DNMF.m:37
    q.h_update_ind = L->on.data();   // :38
    SN_TRY(snmf_plan_create(L->c1, &q, &L->p1));
    q.r = R_x;
    q.w_update_ind = L->on.data();   // :43
    q.h_update_ind = L->off.data();  // :44
    SN_TRY(snmf_plan_create(L->c2, &q, &L->p2));
    q.r = R_d;                       // :49-50
    SN_TRY(snmf_plan_create(L->c2, &q, &L->p3));
    return SNMF_OK;
}

// solve 1, then (A_hat staying in HBM) solves 2 and 3; `upload23` brings V / W of plans 2 and 3 in and runs on a second host
// thread under solve 1 (it may be empty when they are already resident)
template <typename T>
int loop3_run(Loop3* L, const std::function<int()>& upload23, int R_x, int R_d, T* B_hat, int64_t ldBh, T* A_hat, int64_t ldA,
              int32_t* n_iter3) {
    snmf_plan *p1 = L->p1, *p2 = L->p2, *p3 = L->p3;
    const int F = p1->p.F;
    if (ldBh < F) return fail(SNMF_ERR_INVALID, "leading dimension of B_hat < F");
    int rc_up = SNMF_OK;
    std::string err_up;
    std::thread up;
    if (upload23) up = std::thread([&] {
        rc_up = upload23();
        if (rc_up != SNMF_OK) err_up = g_err;
    });
    int32_t n1 = 0, n2 = 0, n3 = 0;
    int s = snmf_plan_init(p1);
    SN_STEP(s, snmf_plan_run(p1, p1->p.max_iter, &n1));  // [~, A_hat] = sparse_nmf(Y, p)   (:40)
    if (up.joinable()) up.join();
    if (s != SNMF_OK) return s;
    if (rc_up != SNMF_OK) return fail(rc_up, "%s", err_up.c_str());
    int idx = 0;
    SN_TRY(result_h_index(p1, &idx));  // (synchronises solve 1's stream: A_hat is complete)
    const float* A = p1->H[idx];
    // p.init_h = A_hat(1:R_x,:) / A_hat(R_x+1:end,:)   (:46, :52): rows of the resident fp32 H, no host round trip
    SN_TRY(set_h<float>(p2, A, p1->rp, 1));
    SN_TRY(set_h<float>(p3, A + R_x, p1->rp, 1));
    int rc_a = SNMF_OK;
    std::string err_a;
    std::thread dl;
    if (A_hat) dl = std::thread([&] {  // A_hat -> host under solves 2 / 3 (plan 1's stream and transfer pipeline are idle now)
        rc_a = sizeof(T) == 8 ? snmf_plan_get_h_f64(p1, (double*)A_hat, ldA, 0) : snmf_plan_get_h_f32(p1, (float*)A_hat, ldA, 0);
        if (rc_a != SNMF_OK) err_a = g_err;
    });
    s = snmf_plan_init(p2);
    SN_STEP(s, snmf_plan_run(p2, p2->p.max_iter, &n2));  // [B_hat_x, ~] = sparse_nmf(X, p)  (:47)
    SN_STEP(s, snmf_plan_init(p3));
    SN_STEP(s, snmf_plan_run(p3, p3->p.max_iter, &n3));  // [B_hat_d, ~] = sparse_nmf(D, p)  (:53)
    if (sizeof(T) == 8) {
        SN_STEP(s, snmf_plan_get_w_f64(p2, (double*)B_hat, ldBh, 0));  // B_hat = [B_hat_x, B_hat_d]   (:55)
        SN_STEP(s, snmf_plan_get_w_f64(p3, (double*)B_hat + (size_t)R_x * ldBh, ldBh, 0));
    } else {
        SN_STEP(s, snmf_plan_get_w_f32(p2, (float*)B_hat, ldBh, 0));
        SN_STEP(s, snmf_plan_get_w_f32(p3, (float*)B_hat + (size_t)R_x * ldBh, ldBh, 0));
    }
    if (dl.joinable()) dl.join();
    if (s != SNMF_OK) return s;
    if (rc_a != SNMF_OK) return fail(rc_a, "%s", err_a.c_str());
    if (n_iter3) {
        n_iter3[0] = n1;
        n_iter3[1] = n2;
        n_iter3[2] = n3;
    }
    return SNMF_OK;
}

template <typename T>
int dnmf_impl(snmf_ctx* ctx, const snmf_params* p, int R_x, int R_d, const T* Y, int64_t ldY, const T* X, int64_t ldX, const T* D,
              int64_t ldD, const T* B, int64_t ldB, const T* H0, uint64_t seed, T* B_hat, int64_t ldBh, T* A_hat, int64_t ldA,
              int32_t* n_iter3) {
    if (!Y || !X || !D || !B || !B_hat) return fail(SNMF_ERR_INVALID, "Y, X, D, B and B_hat must be non-NULL");
    (void)hipGetLastError();
    Loop3 L;
    SN_TRY(loop3_create(ctx, p, R_x, R_d, &L));
    if (ldB < p->F) return fail(SNMF_ERR_INVALID, "leading dimension of B < F");
    if (A_hat && ldA < p->r) return fail(SNMF_ERR_INVALID, "leading dimension of A_hat < R_x + R_d");
    SN_TRY(set_v<T>(L.p1, Y, ldY, 0));
    SN_TRY(set_w<T>(L.p1, B, ldB, 0));  // p.init_w = B   (:39)
    if (H0) SN_TRY(set_h<T>(L.p1, H0, p->r, 0));
    else SN_TRY(rand_h(L.p1, seed, 0));
    auto upload23 = [&]() -> int {
        SN_TRY(set_v<T>(L.p2, X, ldX, 0));
        SN_TRY(set_v<T>(L.p3, D, ldD, 0));
        SN_TRY(set_w<T>(L.p2, B, ldB, 0));                          // p.init_w = B(:,1:R_x)            (:45)
        SN_TRY(set_w<T>(L.p3, B + (size_t)R_x * ldB, ldB, 0));      // p.init_w = B(:,R_x+1:R_x+R_d)    (:51)
        return SNMF_OK;
    };
    return loop3_run<T>(&L, upload23, R_x, R_d, B_hat, ldBh, A_hat, ldA, n_iter3);
}

// device features of one signal into a plan's resident V: |STFT|.^pow + floor (run_basis_DNMF.m:13-34), optionally the Mel
// projection of run_basis_DNMF_Mel.m:21-69 (mel: M x n row-major on the HOST, scratch: device K*n x T)
int features_to_plan(snmf_plan* pl, const snmf_stft_params* sp, const float* d_sig, int64_t n, const float* mel, int M, float* d_scratch) {
    if (!mel) return snmf_plan_set_v_from_audio_f32(pl, sp, d_sig, n, 1);
    const int nb = sp->fftlength / 2 + 1, K = 2 * sp->splice + 1;
    const int64_t T = snmf_stft_num_frames(sp, n);
    SN_TRY(stft_to_device(pl->ctx, sp, d_sig, n, 1, d_scratch, (int64_t)K * nb, T));
    hipStream_t st = pl->ctx->stream;
    HIP_TRY(hipMemsetAsync(pl->V, 0, (size_t)pl->Fp * pl->Tp * 4, st));
    SN_TRY(snmf_mel_features_f32(pl->ctx, mel, M, nb, K, d_scratch, (int64_t)K * nb, (int32_t)T, pl->V, pl->Fp, 1));
    if (pl->p.floor_v) {
        hipLaunchKernelGGL(k_floor_real, dim3(grid_for((size_t)pl->Fp * pl->p.T)), dim3(256), 0, st, pl->V, pl->Fp, pl->p.F, pl->p.T, kFlr);
        HIP_TRY(hipGetLastError());
    }
    pl->have_v = true;
    pl->mdi_v_fresh = true;
    return SNMF_OK;
}

struct DevBuf {  // scratch that dies with the call
    std::vector<void*> ptrs;
    template <typename T>
    int alloc(T** p, size_t n) {
        SN_TRY(dalloc(p, n));
        ptrs.push_back(*p);
        return SNMF_OK;
    }
    ~DevBuf() {
        for (void* q : ptrs) hipFree(q);
    }
};

}  // namespace

extern "C" int snmf_run_basis_dnmf_f64(snmf_ctx* ctx, const snmf_params* p, int32_t R_x, int32_t R_d, const double* Y, int64_t ldY,
                                       const double* X, int64_t ldX, const double* D, int64_t ldD, const double* B, int64_t ldB,
                                       const double* H0, uint64_t seed, double* B_hat, int64_t ldBh, double* A_hat, int64_t ldA,
                                       int32_t* n_iter_out) {
    return dnmf_impl<double>(ctx, p, R_x, R_d, Y, ldY, X, ldX, D, ldD, B, ldB, H0, seed, B_hat, ldBh, A_hat, ldA, n_iter_out);
}
extern "C" int snmf_run_basis_dnmf_f32(snmf_ctx* ctx, const snmf_params* p, int32_t R_x, int32_t R_d, const float* Y, int64_t ldY,
                                       const float* X, int64_t ldX, const float* D, int64_t ldD, const float* B, int64_t ldB,
                                       const float* H0, uint64_t seed, float* B_hat, int64_t ldBh, float* A_hat, int64_t ldA,
                                       int32_t* n_iter_out) {
    return dnmf_impl<float>(ctx, p, R_x, R_d, Y, ldY, X, ldX, D, ldD, B, ldB, H0, seed, B_hat, ldBh, A_hat, ldA, n_iter_out);
}

// B_hat = run_basis_DNMF(x, d, B, p) from the two WAVEFORMS: the truncation to equal length (:5-9), y = x + d (:10), the
// three spectrogram feature sets (:13-34) and the loop (:36-55) on the device; with `mel` the Mel twin run_basis_DNMF_Mel.m.
extern "C" int snmf_run_basis_dnmf_audio_f64(snmf_ctx* ctx, const snmf_params* p, const snmf_stft_params* sp, int32_t R_x, int32_t R_d,
                                             const float* x, int64_t n_x, const float* d, int64_t n_d, const float* mel, int32_t mel_M,
                                             const double* B, int64_t ldB, const double* H0, uint64_t seed, double* B_hat,
                                             int64_t ldBh, double* A_hat, int64_t ldA, int32_t* n_iter_out) {
    if (!ctx || !p || !x || !d || !B || !B_hat) return fail(SNMF_ERR_INVALID, "NULL argument");
    SN_TRY(validate_stft(sp));
    (void)hipGetLastError();
    const int64_t n = std::min(n_x, n_d);  // :5-9
    const int64_t T = snmf_stft_num_frames(sp, n);
    const int nb = sp->fftlength / 2 + 1, K = 2 * sp->splice + 1;
    const int64_t F = mel ? (int64_t)K * mel_M : (int64_t)K * nb;
    if (T < 1) return fail(SNMF_ERR_INVALID, "the signals are shorter than one analysis frame");
    if (F != p->F || T != p->T)
        return fail(SNMF_ERR_DIM, "the signals give %lld x %lld features, params say %d x %d", (long long)F, (long long)T, p->F, p->T);
    if (mel && mel_M < 1) return fail(SNMF_ERR_INVALID, "mel_M must be positive");
    Loop3 L;
    SN_TRY(loop3_create(ctx, p, R_x, R_d, &L));
    if (ldB < p->F) return fail(SNMF_ERR_INVALID, "leading dimension of B < F");
    if (A_hat && ldA < p->r) return fail(SNMF_ERR_INVALID, "leading dimension of A_hat < R_x + R_d");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf buf;
    float *dx = nullptr, *dd = nullptr, *dy = nullptr, *scr1 = nullptr, *scr2 = nullptr;
    SN_TRY(buf.alloc(&dx, (size_t)n));
    SN_TRY(buf.alloc(&dd, (size_t)n));
    SN_TRY(buf.alloc(&dy, (size_t)n));
    if (mel) {
        SN_TRY(buf.alloc(&scr1, (size_t)K * nb * T));
        SN_TRY(buf.alloc(&scr2, (size_t)K * nb * T));
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dd, d, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_add_sig, dim3(grid_for((size_t)n)), dim3(256), 0, st, (const float*)dx, (const float*)dd, dy, (size_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // the signals are read on the second stream as well
    SN_TRY(features_to_plan(L.p1, sp, dy, n, mel, mel_M, scr1));
    SN_TRY(set_w<double>(L.p1, B, ldB, 0));
    if (H0) SN_TRY(set_h<double>(L.p1, H0, p->r, 0));
    else SN_TRY(rand_h(L.p1, seed, 0));
    auto upload23 = [&]() -> int {
        HIP_TRY(hipSetDevice(ctx->device));
        SN_TRY(features_to_plan(L.p2, sp, dx, n, mel, mel_M, scr2));
        SN_TRY(features_to_plan(L.p3, sp, dd, n, mel, mel_M, scr2));
        SN_TRY(set_w<double>(L.p2, B, ldB, 0));
        SN_TRY(set_w<double>(L.p3, B + (size_t)R_x * ldB, ldB, 0));
        return SNMF_OK;
    };
    const int rc = loop3_run<double>(&L, upload23, R_x, R_d, B_hat, ldBh, A_hat, ldA, n_iter_out);
    hipStreamSynchronize(L.c1->stream);
    hipStreamSynchronize(L.c2->stream);
    return rc;
}

// [B_DFT_init, A_DFT_init] and [B_Mel_init, A_Mel_init] of run_basis_train.m:58-91 for one event class from its training
// signal: TF_mag (:60-63) with the optional TF_DD (:64-67), TF_Mel (:70-78), the exemplar columns (:82-83) and the two
// full-update solves (:84-91) -- features, exemplars and both V matrices never leave HBM.
extern "C" int snmf_run_basis_train_audio_f64(snmf_ctx* ctx, const snmf_params* p, const snmf_stft_params* sp, double alpha_eta_dd,
                                              const float* mel, int32_t mel_M, const float* s_full, int64_t n_samples,
                                              const int64_t* sample_idx, int32_t train_exemplar, const double* H0, uint64_t seed,
                                              double* B_DFT, double* A_DFT, double* B_Mel, double* A_Mel, int32_t* n_iter_out) {
    if (!ctx || !p || !s_full || !sample_idx || !B_DFT) return fail(SNMF_ERR_INVALID, "NULL argument");
    SN_TRY(validate_stft(sp));
    (void)hipGetLastError();
    const int nb = sp->fftlength / 2 + 1, K = 2 * sp->splice + 1;
    const int64_t T = snmf_stft_num_frames(sp, n_samples), F = (int64_t)K * nb, Fm = (int64_t)K * mel_M;
    const int r = p->r;
    if (T < 1) return fail(SNMF_ERR_INVALID, "the signal is shorter than one analysis frame");
    if (F != p->F || T != p->T)
        return fail(SNMF_ERR_DIM, "the signal gives %lld x %lld features, params say %d x %d", (long long)F, (long long)T, p->F, p->T);
    if ((mel != nullptr) != (B_Mel != nullptr)) return fail(SNMF_ERR_INVALID, "mel and B_Mel must be given together");
    if (mel && mel_M < 1) return fail(SNMF_ERR_INVALID, "mel_M must be positive");
    if (p->sparsity_kind == SNMF_SPARSITY_FULL) return fail(SNMF_ERR_UNSUPPORTED, "a full sparsity matrix is not supported by the training entry");
    for (int j = 0; j < r; ++j)
        if (sample_idx[j] < 0 || sample_idx[j] >= T) return fail(SNMF_ERR_INVALID, "sample_idx[%d] = %lld outside [0, %lld)", j, (long long)sample_idx[j], (long long)T);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    snmf_params q = *p;
    q.w_update_ind = q.h_update_ind = nullptr;  // :85-86 all true
    struct Plans {
        snmf_plan *a = nullptr, *b = nullptr;
        ~Plans() {
            if (a) snmf_plan_destroy(a);
            if (b) snmf_plan_destroy(b);
        }
    } P;
    SN_TRY(snmf_plan_create(ctx, &q, &P.a));
    snmf_plan* pa = P.a;
    DevBuf buf;
    float *ds = nullptr, *ex = nullptr, *exm = nullptr;
    int64_t* didx = nullptr;
    SN_TRY(buf.alloc(&ds, (size_t)n_samples));
    SN_TRY(buf.alloc(&didx, (size_t)r));
    SN_TRY(buf.alloc(&ex, (size_t)F * r));
    HIP_TRY(hipMemcpyAsync(ds, s_full, (size_t)n_samples * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(didx, sample_idx, (size_t)r * 8, hipMemcpyHostToDevice, st));
    // TF_mag straight into the DFT plan's resident V (leading dimension Fp; pad rows / columns stay zero)
    HIP_TRY(hipMemsetAsync(pa->V, 0, (size_t)pa->Fp * pa->Tp * 4, st));
    SN_TRY(stft_to_device(ctx, sp, ds, n_samples, 1, pa->V, pa->Fp, T));
    if (alpha_eta_dd >= 0.0) SN_TRY(snmf_tf_dd_f32(ctx, alpha_eta_dd, (int32_t)F, (int32_t)T, pa->V, pa->Fp, pa->V, pa->Fp, 1));  // :64-67
    hipLaunchKernelGGL(k_gather_cols, dim3(grid_for((size_t)F * r)), dim3(256), 0, st, (const float*)pa->V, (int64_t)pa->Fp, (int)F,
                       (const int64_t*)didx, r, ex);  // B_DFT_init = TF_mag(:, sample_idx)   (:82)
    HIP_TRY(hipGetLastError());
    if (mel) {
        q.F = (int32_t)Fm;
        SN_TRY(snmf_plan_create(ctx, &q, &P.b));
        SN_TRY(buf.alloc(&exm, (size_t)Fm * r));
        HIP_TRY(hipMemsetAsync(P.b->V, 0, (size_t)P.b->Fp * P.b->Tp * 4, st));
        SN_TRY(snmf_mel_features_f32(ctx, mel, mel_M, nb, K, pa->V, pa->Fp, (int32_t)T, P.b->V, P.b->Fp, 1));  // :70-78 (before the V floor, like the reference)
        hipLaunchKernelGGL(k_gather_cols, dim3(grid_for((size_t)Fm * r)), dim3(256), 0, st, (const float*)P.b->V, (int64_t)P.b->Fp, (int)Fm,
                           (const int64_t*)didx, r, exm);  // B_Mel_init = TF_Mel(:, sample_idx)   (:83)
        HIP_TRY(hipGetLastError());
    }
    int32_t nit[2] = {0, 0};
    auto solve = [&](snmf_plan* pl, const float* w0, int64_t rows, double* Bo, double* Ao, int32_t* ni) -> int {
        if (train_exemplar) {  // :84, :95-96: the exemplars ARE the dictionary
            HIP_TRY(hipStreamSynchronize(st));
            std::vector<float> h((size_t)rows * r);
            HIP_TRY(hipMemcpy(h.data(), w0, h.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < h.size(); ++i) Bo[i] = (double)h[i];
            return SNMF_OK;
        }
        if (pl->p.floor_v) {  // v = max(v, flr) (src/sparse_nmf.m:169) on the real entries
            hipLaunchKernelGGL(k_floor_real, dim3(grid_for((size_t)pl->Fp * pl->p.T)), dim3(256), 0, st, pl->V, pl->Fp, pl->p.F, pl->p.T, kFlr);
            HIP_TRY(hipGetLastError());
        }
        pl->have_v = true;
        pl->mdi_v_fresh = true;
        SN_TRY(set_w<float>(pl, w0, rows, 1));  // p.init_w = B_*_init   (:87, :90)
        if (H0) SN_TRY(set_h<double>(pl, H0, r, 0));
        else SN_TRY(rand_h(pl, seed, 0));       // the reference re-seeds per call (:112-114): both solves start from the SAME h
        SN_TRY(snmf_plan_init(pl));
        SN_TRY(snmf_plan_run(pl, pl->p.max_iter, ni));
        SN_TRY(snmf_plan_get_w_f64(pl, Bo, rows, 0));
        if (Ao) SN_TRY(snmf_plan_get_h_f64(pl, Ao, r, 0));
        return SNMF_OK;
    };
    int s = SNMF_OK;
    // (the Mel exemplars and TF_Mel were formed above from the UNFLOORED TF_mag; the solver's own floor comes last)
    SN_STEP(s, solve(pa, ex, F, B_DFT, A_DFT, &nit[0]));                       // :88
    if (mel) SN_STEP(s, solve(P.b, exm, Fm, B_Mel, A_Mel, &nit[1]));           // :91
    hipStreamSynchronize(st);
    if (s == SNMF_OK && n_iter_out) {
        n_iter_out[0] = nit[0];
        n_iter_out[1] = nit[1];
    }
    return s;
}
