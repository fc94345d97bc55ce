/*
 * snmf.h -- C ABI of libsnmf_hip.so, the MI355X (gfx950) sparse-NMF multiplicative-update engine.
 *
 * Drop-in boundary for the hot path of lordet01/SE_SNMF_NAT:
 *     [w, h, objective] = sparse_nmf(v, p)            (reference: src/sparse_nmf.m:1)
 *     [w, h, objective] = sparse_nmf_GPU(v, p)        (reference: src/sparse_nmf_GPU.m:1)
 * and the 3-solve loop of run_basis_DNMF.m:36-55 built on it.  The reference is MATLAB; the
 * binding a maintainer adds is a MEX shim (integration/sparse_nmf_mex.cpp) or, from Python,
 * ctypes (se_snmf_nat_amd/_lib.py).  Every entry point below names the reference lines it replaces.
 *
 * Conventions: all matrices are column-major (MATLAB layout) with explicit leading dimensions
 * where given; plain pointers and sizes only; every function returns an snmf_status and never
 * throws; snmf_last_error() returns a thread-local message for the last failure.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * SNMF_ERR_NO_DEVICE.
 */
#ifndef SNMF_H_
#define SNMF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNMF_ABI_VERSION 5  /* 5: snmf_multi_release_cache, snmf_multi_cached_teams, snmf_rccl_*, snmf_plan_run_sharded_rccl (the device RNG of snmf_plan_set_h_random changed with 4: draws seeded under ABI 3 are not reproducible); 4: snmf_run_basis_dnmf_multi_*; 3: snmf_run_basis_dnmf_*, snmf_run_basis_train_audio_f64, snmf_sparse_nmf_oop_*, snmf_plan_set_h_random, snmf_ctx_xfer_stats; 2: snmf_multi_* */

typedef enum snmf_status {
    SNMF_OK = 0,
    SNMF_ERR_INVALID = 1,   /* bad argument (NULL, non-positive size, ...) */
    SNMF_ERR_NO_INIT = 2,   /* neither init_w nor r given: src/sparse_nmf.m:117-119 */
    SNMF_ERR_DIM = 3,       /* MATLAB dimension error on the path (e.g. partial h_update_ind, :192) */
    SNMF_ERR_NO_FIELD = 4,  /* p.cost_check missing: src/sparse_nmf.m:260 */
    SNMF_ERR_NO_DEVICE = 5, /* no HIP device / HIP runtime failure */
    SNMF_ERR_NOMEM = 6,
    SNMF_ERR_STATE = 7,     /* call order violated (e.g. run before set_v) */
    SNMF_ERR_UNSUPPORTED = 8,
    SNMF_ERR_INTERNAL = 9   /* a bounded device-side wait gave up (never expected): the results are invalid */
} snmf_status;

/* sparsity argument forms of src/sparse_nmf.m:150-155 */
typedef enum snmf_sparsity_kind {
    SNMF_SPARSITY_SCALAR = 0, /* length(p.sparsity)==1 -> ones(r,n)*s   (:151-152) */
    SNMF_SPARSITY_RVEC = 1,   /* r x 1 column -> repmat(.,1,n)         (:153-154) */
    SNMF_SPARSITY_FULL = 2    /* r x n given in full                    (:155)     */
} snmf_sparsity_kind;

/* Solver parameters = the fields of `p` the reference reads (src/sparse_nmf.m:79-164). */
typedef struct snmf_params {
    int32_t F;          /* rows of v       (m, :71) */
    int32_t T;          /* columns of v    (n, :72) */
    int32_t r;          /* rank            (:116-131) */
    double beta;        /* divergence order after the cf switch (:99-110): is=0, kl=1, ed=2 */
    int32_t max_iter;   /* :79-81, default 100 */
    double conv_eps;    /* :91-93, default 0 */
    int32_t cost_check; /* :260 (no default in the reference; the wrappers enforce presence) */
    int32_t floor_v;    /* 1: v = max(v,1e-9) as sparse_nmf.m:169; 0: sparse_nmf_GPU.m (absent) */
    int32_t sparsity_kind;     /* snmf_sparsity_kind */
    double sparsity_scalar;    /* used when kind == SCALAR */
    const uint8_t* w_update_ind; /* r flags or NULL = all true (:142-144) */
    const uint8_t* h_update_ind; /* r flags or NULL = all true (:146-148) */
} snmf_params;

typedef struct snmf_ctx snmf_ctx;   /* device + stream + kernel configuration */
typedef struct snmf_plan snmf_plan; /* one problem resident in HBM: V, W, H, workspaces */

/* ---- library ---------------------------------------------------------------------------- */
int snmf_abi_version(void);
const char* snmf_last_error(void);
int snmf_device_count(void); /* number of HIP devices visible, 0 if none (never fails) */

/* ---- context ---------------------------------------------------------------------------- */
/* Replaces the implicit gpuArray device state of src/sparse_nmf_GPU.m:161-166.
 * A context keeps the device blocks of destroyed plans for the next plan of the same sizes (a caller that solves problem after
 * problem of one shape -- every call site of sparse_nmf in the reference -- does not pay hipMalloc / hipFree per call): at most
 * SNMF_DEVCACHE_MB megabytes (environment, read at creation; default 4096, 0 = off), released by snmf_ctx_destroy. */
int snmf_ctx_create(snmf_ctx** out, int device);
/* Use a caller-owned hipStream_t; NULL = the context's own (non-blocking) stream.  The legacy
 * default stream has handle 0 == NULL and therefore cannot be selected: a caller that must order
 * other work (e.g. an RCCL all-reduce) against the solver passes an explicitly created stream and
 * issues that work on it (se_snmf_nat_amd/dist.py). */
int snmf_ctx_set_stream(snmf_ctx* ctx, void* hip_stream);
int snmf_ctx_sync(snmf_ctx* ctx);
void snmf_ctx_destroy(snmf_ctx* ctx);

/* ---- one-shot drop-in ------------------------------------------------------------------- */
/* [w,h,objective] = sparse_nmf(v,p) with explicit initial factors (the MATLAB wrapper draws
 * them with MATLAB's own RNG, src/sparse_nmf.m:112-140, so RNG parity is by construction).
 *   V      F x T  (ldV >= F), never modified                       (:71-72, :169)
 *   W      F x r  in: init_w (:116-131)   out: w, unit-L2 columns  (:242)
 *   H      r x T  in: init_h (:133-140)   out: h
 *   sparsity: NULL for SCALAR, r doubles for RVEC, r x T doubles for FULL (:150-155)
 *   div_out, cost_out: max_iter doubles each, zero-filled then objective.div/.cost (:171-173,
 *                      :263-264); may be NULL
 *   n_iter_out: number of iterations executed = length of the truncated vectors (:279-280)
 */
int snmf_sparse_nmf_f64(snmf_ctx* ctx, const snmf_params* p, const double* V, int64_t ldV,
                        double* W, double* H, const double* sparsity, double* div_out,
                        double* cost_out, int32_t* n_iter_out);
int snmf_sparse_nmf_f32(snmf_ctx* ctx, const snmf_params* p, const float* V, int64_t ldV,
                        float* W, float* H, const float* sparsity, double* div_out,
                        double* cost_out, int32_t* n_iter_out);
/* Out-of-place form: W0 (F x r) and H0 (r x T) are read-only, the results go to W and H (tight: leading dimensions F and r).
 * This is MATLAB's value semantics without a copy of init_h on the host first -- what the MEX shim calls. */
int snmf_sparse_nmf_oop_f64(snmf_ctx* ctx, const snmf_params* p, const double* V, int64_t ldV, const double* W0,
                            const double* H0, const double* sparsity, double* W, double* H, double* div_out,
                            double* cost_out, int32_t* n_iter_out);
int snmf_sparse_nmf_oop_f32(snmf_ctx* ctx, const snmf_params* p, const float* V, int64_t ldV, const float* W0,
                            const float* H0, const float* sparsity, float* W, float* H, double* div_out,
                            double* cost_out, int32_t* n_iter_out);

/* ---- resident-plan API (benchmarks, on-device pipelines, multi-GPU sharding) -------------- */
/* A plan owns fp32 device copies of V (F x T), W (F x r), H (r x T) in the engine's padded
 * layouts plus all workspaces.  Call order: create -> set_v/set_w/set_h[/set_sparsity] ->
 * init -> run (or the step functions) -> get_*.  `p->T` is the LOCAL column count of this
 * shard when the frame axis is sharded across ranks.
 * One documented deviation from the reference's expression: a Euclidean (beta = 2) FULL update with r < 2F forms the W step's
 * P = max(W*H, 1e-9) * H' (src/sparse_nmf.m:166, :228-233) as W * (H*H'), i.e. without the floor on W*H.  The two differ only
 * where W*H < 1e-9 (silent rows / frames), by at most 1e-9 * sum(h) per entry of P.  Environment SNMF_GRAM_P=0 at plan creation
 * selects the reference's expression (one more Lam' pass per iteration); tests/test_gpu_parity.py pins both. */
int snmf_plan_create(snmf_ctx* ctx, const snmf_params* p, snmf_plan** out);
void snmf_plan_destroy(snmf_plan* plan);

/* is_device != 0: pointer is device memory on the context's device (fp32/fp64 as named).  The library reads it on
 * the CONTEXT's stream (snmf_ctx_set_stream): a device buffer produced on another stream must be complete -- or that
 * stream ordered ahead of the context's with an event -- before the call; the host mirror synchronises the producer. */
int snmf_plan_set_v_f64(snmf_plan* plan, const double* V, int64_t ld, int is_device);
int snmf_plan_set_v_f32(snmf_plan* plan, const float* V, int64_t ld, int is_device);
int snmf_plan_set_w_f64(snmf_plan* plan, const double* W, int64_t ld, int is_device);
int snmf_plan_set_w_f32(snmf_plan* plan, const float* W, int64_t ld, int is_device);
int snmf_plan_set_h_f64(snmf_plan* plan, const double* H, int64_t ld, int is_device);
int snmf_plan_set_h_f32(snmf_plan* plan, const float* H, int64_t ld, int is_device);
/* RVEC: r values; FULL: r x T (ld = r).  SCALAR needs no call. */
int snmf_plan_set_sparsity_f64(snmf_plan* plan, const double* S, int is_device);
int snmf_plan_set_sparsity_f32(snmf_plan* plan, const float* S, int is_device);

/* src/sparse_nmf.m:157-169: normalise W columns, rescale H rows (V is floored at set_v); resets
 * the iteration counter and the objective history.  If W was not set again since the last init its
 * normalised form and norms are reused (same bits as normalising the same input again). */
int snmf_plan_init(snmf_plan* plan);

/* The hot loop src/sparse_nmf.m:186-286: up to `n_iters` more iterations (bounded by
 * max_iter), early stop per :273-282 evaluated on the device.  Asynchronous on the context's
 * stream unless conv_eps > 0 (then it synchronises every few iterations to read the stop flag).
 * iters_done (optional) receives the total iterations executed so far (after a sync). */
int snmf_plan_run(snmf_plan* plan, int32_t n_iters, int32_t* iters_done);

/* Step functions for frame-sharded multi-GPU training; one iteration is
 *     hstep -> wstats(stats) -> [all-reduce(sum) of stats across ranks] -> wapply(stats)
 * stats is a DEVICE buffer of snmf_plan_stats_len() doubles:
 *     [ G or Q (F*r) | P (F*r, beta != 1 only) | s (r) | div | sum(S.*H) ]
 * hstep  : src/sparse_nmf.m:189-208 on the local columns (no-op when no row of h is updated)
 * wstats : the T-reductions of :215-239 (+ the objective sums of :248-261 for the PREVIOUS
 *          iterate) over the local columns
 * wapply : the F x r epilogue of :215-244 on the reduced statistics, the convergence test of
 *          :272-284 on the reduced cost, identical on every rank.
 * finalize: objective of the last iterate (needs one more local pass + all-reduce of the two
 *          scalars): objstats -> [all-reduce] -> objapply. */
int64_t snmf_plan_stats_len(const snmf_plan* plan);
int snmf_plan_hstep(snmf_plan* plan);
int snmf_plan_wstats(snmf_plan* plan, double* stats_dev);
int snmf_plan_wapply(snmf_plan* plan, const double* stats_dev);
int snmf_plan_objstats(snmf_plan* plan, double* stats_dev);
int snmf_plan_objapply(snmf_plan* plan, const double* stats_dev);
/* 1 when the device-side convergence test has fired (synchronises the stream). */
int snmf_plan_stopped(snmf_plan* plan, int32_t* stopped);
/* The loop above in ONE call for a process-per-GPU host: up to n_iters iterations of
 *     hstep -> wstats(stats) -> all_reduce(stats, len, user) -> wapply(stats)
 * with the caller's collective as a callback (it must enqueue an in-place SUM of the `len` doubles at `stats` over the ranks on
 * the context's stream, or order its own stream against it; it returns 0, anything else aborts the loop with
 * SNMF_ERR_INVALID).  `len` is the whole buffer, or 2 (the two cost scalars at the tail: `stats + stats_len - 2`) when no
 * column of W is updated.  all_reduce = NULL: a single rank.  poll_every > 0: the convergence flag is read every poll_every
 * iterations when conv_eps > 0 (every rank reads the same value).  finalize != 0: after the last iteration of the SOLVE
 * (max_iter reached, no early stop) the objective of the last iterate is formed (objstats -> all_reduce -> objapply).
 * iters_done: iterations this call ran.  src/sparse_nmf.m:186-286 per rank. */
typedef int (*snmf_allreduce_fn)(double* stats_dev, int64_t len, void* user);
int snmf_plan_run_sharded(snmf_plan* plan, int32_t n_iters, double* stats_dev, snmf_allreduce_fn all_reduce, void* user,
                          int32_t poll_every, int32_t finalize, int32_t* iters_done);
/* The same loop with the collective issued by the library itself: ncclAllReduce (RCCL over xGMI) of the statistics on the
 * context's stream, once per iteration -- no callback into the host language (a ctypes / MEX trampoline per iteration was 29 us of
 * host work, a quarter of an iteration on a short shard).  librccl.so is resolved with dlopen when first needed (SNMF_RCCL_LIB, else
 * the copy the process already holds -- PyTorch's --, else the system's), so this library loads without it; snmf_rccl_available()
 * says whether it was found.  The communicator is set up by the caller's own out-of-band channel: rank 0 calls
 * snmf_rccl_get_unique_id (128 bytes), every rank receives them (torch.distributed broadcast, MPI, a file) and calls
 * snmf_rccl_comm_create(device, id, n_ranks, rank) -- ncclCommInitRank, collective over the ranks -- and later
 * snmf_rccl_comm_destroy.  One communicator per rank, on the plan's device.  Reference: the one sum per iteration that sharding the
 * frames of run_basis_train.m:88 / run_basis_DNMF.m:47,53 over ranks needs (the statistics of src/sparse_nmf.m:215-239, :248-261). */
typedef struct snmf_rccl_comm snmf_rccl_comm;
int snmf_rccl_available(void);
int snmf_rccl_get_unique_id(void* id_out, int64_t cap);
int snmf_rccl_comm_create(int32_t device, const void* id, int32_t n_ranks, int32_t rank, snmf_rccl_comm** out);
void snmf_rccl_comm_destroy(snmf_rccl_comm* comm);
int snmf_plan_run_sharded_rccl(snmf_plan* plan, int32_t n_iters, double* stats_dev, snmf_rccl_comm* comm, int32_t poll_every,
                               int32_t finalize, int32_t* iters_done);

/* Online separation stream (src/NTF_sep_event_RT.m:67-107 -> src/bnmf_sep_event_RT_IS16.m:138-154):
 * the SAME dictionary W and the SAME initial activations H0 (the reference re-seeds its generator
 * in every call, src/sparse_nmf.m:112-114, so H0 is identical for every frame) are used for
 * n_solves INDEPENDENT H-only solves, one per group of `frames_per_solve` (<= 32) consecutive
 * columns of V; each has its own objective and its own convergence test (:272-284).
 * Result-identical to n_solves calls of snmf_sparse_nmf_* with w_update_ind all false, but W stays
 * resident and normalised, V is uploaded once, and every solve is ONE persistent workgroup (all of
 * its iterations inside one launch), the solves running concurrently across the CUs.
 *   plan    : H-only plan whose T >= n_solves*frames_per_solve (capacity); snmf_plan_set_w called
 *   V       : F x (n_solves*frames_per_solve) host matrix, leading dimension ldV
 *   H0      : r x frames_per_solve host matrix
 *   H_out   : r x (n_solves*frames_per_solve) host matrix (leading dimension r)
 *   n_iter_out[n_solves], cost_out[n_solves] (last recorded cost, 0 without cost_check): optional
 * Scalar / r-vector sparsity only. */
int snmf_plan_solve_frames_f64(snmf_plan* plan, int32_t frames_per_solve, const double* V, int64_t ldV,
                               int32_t n_solves, const double* H0, double* H_out, int32_t* n_iter_out,
                               double* cost_out);
int snmf_plan_solve_frames_f32(snmf_plan* plan, int32_t frames_per_solve, const float* V, int64_t ldV,
                               int32_t n_solves, const float* H0, float* H_out, int32_t* n_iter_out,
                               double* cost_out);

/* Results.  W: F x r, H: r x T(local).  Host or device destinations. */
int snmf_plan_get_w_f64(snmf_plan* plan, double* W, int64_t ld, int is_device);
int snmf_plan_get_w_f32(snmf_plan* plan, float* W, int64_t ld, int is_device);
int snmf_plan_get_h_f64(snmf_plan* plan, double* H, int64_t ld, int is_device);
int snmf_plan_get_h_f32(snmf_plan* plan, float* H, int64_t ld, int is_device);
/* objective.div / objective.cost (src/sparse_nmf.m:263-264,:279-280): writes n_iter entries
 * of each (arrays must hold max_iter doubles; the tail is zero-filled). */
int snmf_plan_get_objective(snmf_plan* plan, double* div_out, double* cost_out,
                            int32_t* n_iter_out);

/* ---- spectrogram front-end on the device (the step before the path in every call stack) ---- */
/* Parameters of src/stft_fft.m + run_basis_train.m:60-63 (shipped values:
 * settings/initial_setting_SNMF_NAT.m:21-37,53,88-90). */
typedef struct snmf_stft_params {
    int32_t framelength;   /* sz,    p.framelength (640) */
    int32_t frameshift;    /* shift, p.frameshift  (160) */
    int32_t fftlength;     /* power of two in [64, 4096], p.fftlength (1024) */
    int32_t dcbin;         /* first DCbin magnitudes are set to 1e-6 (src/stft_fft.m:31); must be >= 1 */
    int32_t splice;        /* p.Splice (src/frame_splice.m), 0 = none */
    double preemph;        /* p.preemph */
    double pow;            /* p.pow: features are |STFT|.^pow + nonzerofloor */
    double nonzerofloor;   /* p.nonzerofloor */
    const double* window;  /* framelength values, p.win_STFT */
} snmf_stft_params;

/* Number of frames src/stft_fft.m:21 processes for an n_samples signal (its trailing all-zero
 * columns are the ones run_basis_train.m:61 removes).  Pure host arithmetic, never fails. */
int64_t snmf_stft_num_frames(const snmf_stft_params* sp, int64_t n_samples);

/* TF_mag of run_basis_train.m:60-63 / run_basis_DNMF.m:13-16: F x n_frames, F = (2*splice+1)*(fftlength/2+1).
 * samples: float audio (host or device); V_out: host or device, leading dimension ld >= F. */
int snmf_stft_features_f32(snmf_ctx* ctx, const snmf_stft_params* sp, const float* samples, int64_t n_samples,
                           int samples_on_device, float* V_out, int64_t ld, int out_on_device,
                           int32_t* n_frames_out);
/* Same, written straight into a plan's resident V (plan F must equal the feature rows and plan T the
 * frame count); the V floor of src/sparse_nmf.m:169 is applied as in snmf_plan_set_v. */
int snmf_plan_set_v_from_audio_f32(snmf_plan* plan, const snmf_stft_params* sp, const float* samples,
                                   int64_t n_samples, int samples_on_device);
/* Mel projection of run_basis_train.m:70-78: out (K*M x T) = blockdiag(mel) * V (K*n x T);
 * mel: M x n ROW-major host matrix (mel_matrix(...)'), V / out host or device (both the same side). */
int snmf_mel_features_f32(snmf_ctx* ctx, const float* mel, int32_t M, int32_t n, int32_t K, const float* V,
                          int64_t ldv, int32_t T, float* out, int64_t ldo, int on_device);
/* TF_DD of src/TF_DD.m:1-9 (run_basis_train.m:64-67, p.domain_DD): recursive average along the frames, row by row,
 * X_DD(:,1) = X(:,1), X_DD(:,l) = alpha_eta X_DD(:,l-1) + (1 - alpha_eta) X(:,l).  X / out: F x T column-major, host or
 * device (both the same side); out may be X. */
int snmf_tf_dd_f32(snmf_ctx* ctx, double alpha_eta, int32_t F, int32_t T, const float* X, int64_t ldx, float* out,
                   int64_t ldo, int on_device);

/* ---- the training callers, device-resident ------------------------------------------------------- */
/* h = rand(r, n) of src/sparse_nmf.m:133-134 for a caller that supplies no init_h: Philox-4x32-10 keyed by `seed`, counter =
 * column-major element index / 4, value = ((x >> 9) + 0.5) * 2^-23 (exactly representable, strictly inside (0, 1)), written straight into the plan's resident H
 * (MATLAB's legacy rand('seed', s) stream is not reproducible; a wrapper that wants ITS draws passes them to snmf_plan_set_h).
 * Every entry below that takes `H0` uses this generator when H0 is NULL. */
int snmf_plan_set_h_random(snmf_plan* plan, uint64_t seed);

/* B_hat = run_basis_DNMF(x, d, B, p), the 3-solve loop run_basis_DNMF.m:36-55 on formed features:
 *     [~, A_hat]   = sparse_nmf(Y, p)   H-only,  init_w = B                                         (:37-40)
 *     [B_hat_x, ~] = sparse_nmf(X, p)   W-only,  init_w = B(:,1:R_x),        init_h = A_hat(1:R_x,:)      (:43-47)
 *     [B_hat_d, ~] = sparse_nmf(D, p)   W-only,  init_w = B(:,R_x+1:R_x+R_d), init_h = A_hat(R_x+1:end,:)  (:49-53)
 *     B_hat = [B_hat_x, B_hat_d]                                                                   (:55)
 * Result-identical to three snmf_sparse_nmf_* calls, but Y, X, D cross PCIe once each (X and D on a second stream under solve
 * 1), A_hat never leaves HBM, and only B_hat (and A_hat if asked for) comes back.
 *   p      : F, T, r = R_x + R_d, beta, max_iter, conv_eps, cost_check, floor_v, scalar sparsity; the update masks are set by
 *            the loop (p->w_update_ind / h_update_ind are ignored)
 *   Y, X, D: F x T host matrices (mixture / clean / noise features, :13-34), never modified
 *   B      : F x (R_x + R_d) exemplar basis;  H0: (R_x + R_d) x T initial activations of solve 1 (leading dimension r) or NULL
 *   B_hat  : F x (R_x + R_d) out;  A_hat: (R_x + R_d) x T out or NULL;  n_iter_out: 3 iteration counts or NULL */
int snmf_run_basis_dnmf_f64(snmf_ctx* ctx, const snmf_params* p, int32_t R_x, int32_t R_d, const double* Y, int64_t ldY,
                            const double* X, int64_t ldX, const double* D, int64_t ldD, const double* B, int64_t ldB,
                            const double* H0, uint64_t seed, double* B_hat, int64_t ldBh, double* A_hat, int64_t ldA,
                            int32_t* n_iter_out);
int snmf_run_basis_dnmf_f32(snmf_ctx* ctx, const snmf_params* p, int32_t R_x, int32_t R_d, const float* Y, int64_t ldY,
                            const float* X, int64_t ldX, const float* D, int64_t ldD, const float* B, int64_t ldB,
                            const float* H0, uint64_t seed, float* B_hat, int64_t ldBh, float* A_hat, int64_t ldA,
                            int32_t* n_iter_out);
/* The same from the two WAVEFORMS, as the reference's function takes them (run_basis_DNMF.m:1): truncation to equal length
 * (:5-9), y = x + d (:10), the three feature sets (:13-34) and the loop on the device -- only audio in, B_hat out.
 * mel != NULL: run_basis_DNMF_Mel.m (mel: mel_M x (fftlength/2+1) ROW-major = mel_matrix(...)', features projected as :21-69,
 * B the Mel exemplar basis).  p->F / p->T must equal the feature rows and snmf_stft_num_frames(sp, min(n_x, n_d)). */
int snmf_run_basis_dnmf_audio_f64(snmf_ctx* ctx, const snmf_params* p, const snmf_stft_params* sp, int32_t R_x, int32_t R_d,
                                  const float* x, int64_t n_x, const float* d, int64_t n_d, const float* mel, int32_t mel_M,
                                  const double* B, int64_t ldB, const double* H0, uint64_t seed, double* B_hat, int64_t ldBh,
                                  double* A_hat, int64_t ldA, int32_t* n_iter_out);
/* run_basis_train.m:58-91 for one event class from its assembled training signal: TF_mag (:60-63; alpha_eta_dd >= 0: TF_DD,
 * :64-67), TF_Mel (:70-78; mel / B_Mel NULL: DFT only), exemplar columns sample_idx (r ZERO-based frame indices -- the wrapper
 * draws them, rng(1); randsample(...), :80-81), and the two full-update solves (:84-91; train_exemplar != 0: none, the exemplars
 * are returned).  V is formed in HBM and never crosses PCIe.  p: F = DFT feature rows, T = frames, r = number of exemplars,
 * solver fields.  B_DFT: F x r, A_DFT: r x T or NULL, B_Mel: (2*splice+1)*mel_M x r, A_Mel: r x T or NULL (all fp64, tight);
 * the post-normalisation (+1e-9, :113-116) and the k-means rank reduction (:118-134) stay with the caller. */
int snmf_run_basis_train_audio_f64(snmf_ctx* ctx, const snmf_params* p, const snmf_stft_params* sp, double alpha_eta_dd,
                                   const float* mel, int32_t mel_M, const float* s_full, int64_t n_samples,
                                   const int64_t* sample_idx, int32_t train_exemplar, const double* H0, uint64_t seed,
                                   double* B_DFT, double* A_DFT, double* B_Mel, double* A_Mel, int32_t* n_iter_out);

/* ---- missing-data imputation variants --------------------------------
 * [v_MDI, h, objective] = snmf_mdi(v, Dm, p)     src/snmf_mdi.m:1      (binary observed mask)
 * [v_MDI, h, objective] = snmf_mdi_Sm(v, Sm, p)  src/snmf_mdi_Sm.m:1   (soft mask in [0,1])
 * The same solver with a masked start (:175), a re-imputation of v after every iteration (:251-254)
 * and a gain-matched final imputation (:296-306).  A plan becomes an MDI solve by giving it a mask
 * (F x T, 1 = observed) before snmf_plan_set_v / snmf_plan_init; at least one factor must be updated.
 * The wrappers map p.sparsity_mdi and p.conv_eps_mdi onto the plan's sparsity / conv_eps. */
int snmf_plan_set_mask_f64(snmf_plan* plan, const double* M, int64_t ld, int on_device);
int snmf_plan_set_mask_f32(snmf_plan* plan, const float* M, int64_t ld, int on_device);
/* v_MDI (:296-306) after snmf_plan_run: F x T, observed entries kept, the rest Nt .* max(w*h, flr). */
int snmf_plan_get_v_mdi_f64(snmf_plan* plan, double* V, int64_t ld, int on_device);
int snmf_plan_get_v_mdi_f32(snmf_plan* plan, float* V, int64_t ld, int on_device);

/* ---- online separation loop (BASELINE config 3) -------------------
 * Device-resident replacement of the per-frame function
 *   [x_hat_i, d_hat_i, x_tilde, g] = bnmf_sep_event_RT_IS16(y, l, g, p)   src/bnmf_sep_event_RT_IS16.m:1
 * together with its state g (src/init_buff.m:17-42) and the hop queueing / overlap-add / int16 output of
 * the driver loop src/NTF_sep_event_RT.m:54-135, for the configuration the reference ships
 * (blk_len_sep = 1, Splice = 0, one channel; B_sep_mode 'DFT', or 'Mel' through snmf_online_set_mel).
 * Per frame: STFT -> H-only solve against [B_DFT_x, B_DFT_d] -> reconstructions, block sparsity,
 * adaptive beta, Wiener / MMSE gain -> noise-reference rings and (when triggered) the W-only
 * adaptation solve + dictionary re-assembly -> inverse STFT, overlap-add.  Only PCM in, PCM out and a
 * 32-byte status per frame cross PCIe. */
typedef struct snmf_online snmf_online;

typedef struct snmf_online_params {
    /* signal (settings/initial_setting_SNMF_NAT.m:21-37,53,88-92) */
    int32_t fftlength, framelength, frameshift;
    int32_t dcbin, dcbin_back;
    int32_t delay;            /* p.delay: frames before the first hop is written */
    double preemph, pow, nonzerofloor, overlapscale;
    /* dictionaries */
    int32_t R_x, R_d;
    /* per-frame and adaptation solves (src/sparse_nmf.m parameters) */
    double beta_div;          /* 1 = 'kl', 2 = 'ed', 0 = 'is' */
    double sparsity;
    int32_t max_iter, cost_check;
    double conv_eps;
    /* enhancement filter (:221-261) */
    int32_t enhance_method;   /* 0 = 'Wiener', 1 = 'MMSE' */
    int32_t init_N_len;
    double alpha_eta, alpha_d, beta, beta_max;
    /* block sparsity (src/blk_sparse.m) */
    int32_t blk_sparse, P_len_k, P_len_l, blk_gap;
    double alpha_p;
    /* noise dictionary adaptation (:263-347) */
    int32_t adapt_train_N, R_a, m_a;
    double overlap_m_a, Ar_up;
    int32_t class_outputs;    /* also synthesise the event / noise estimates (x_hat, d_hat) */
    /* semi-supervised frame solve (:125-139): the frame solve also updates the noise (N) or the event (E)
     * columns of its private copy of W; only the activations are used afterwards, as in the reference */
    int32_t basis_update_N, basis_update_E;
} snmf_online_params;

typedef struct snmf_online_frame {   /* per-frame diagnostics, in frame order */
    int32_t n_iter;           /* iterations of the frame solve */
    int32_t trig;             /* adaptation condition :266 */
    int32_t solved;           /* an adaptation solve ran */
    int32_t n_up;             /* sum(r_up) */
    int32_t adapt_iters;      /* iterations of the adaptation solve */
    float beta, A_x_mag, A_d_mag, Q_control;
} snmf_online_frame;

/* B_DFT_x: F x R_x, B_DFT_d: F x R_d (column-major, F = fftlength/2+1); H0: r values standing in for
 * rand(r,1) of src/sparse_nmf.m:133-134 (the same vector every frame, as the reference re-seeds per call);
 * Ad_blk0: R_a x m_a standing in for rand(R_a, m_a) of src/init_buff.m:39; windows: framelength values. */
int snmf_online_create(snmf_ctx* ctx, const snmf_online_params* p, const float* B_DFT_x, const float* B_DFT_d,
                       const float* H0, const float* Ad_blk0, const float* win_stft, const float* win_istft,
                       snmf_online** out);
/* B_sep_mode = 'Mel' (src/bnmf_sep_event_RT_IS16.m:106-120; src/init_buff.m:45-47): call once right after create.
 * melmat: F_order x F ROW-major (g.melmat = mel_matrix(fs, F_order, fftlength, 1, fs/2)'); B_Mel_x / B_Mel_d:
 * F_order x R_x / R_d column-major.  mel_conv = p.MelConv (0: Mel activations on the DFT bases). */
int snmf_online_set_mel(snmf_online* o, int32_t F_order, int32_t mel_conv, const float* melmat, const float* B_Mel_x,
                        const float* B_Mel_d);
/* Current B_Mel_d (Mel mode adapts this one; B_DFT_d stays). */
int snmf_online_get_mel_basis_f32(snmf_online* o, float* B_Mel_d, int64_t ld);
/* Feed n PCM samples (int16-valued floats); every complete hop becomes a frame.  flush != 0 ends the
 * stream the way the driver does at end of file (delay+1 all-zero frames, src/NTF_sep_event_RT.m:69-76).
 * Outputs (host, each may be NULL): the denoised signal before rounding, the int16 the driver writes, and
 * with class_outputs the event / noise estimates; capacity `cap` samples each, *n_out samples written
 * (at most (n/frameshift + delay + 2) * frameshift). */
int snmf_online_process_f32(snmf_online* o, const float* pcm, int64_t n, int flush, float* x_tilde_f32,
                            int16_t* x_tilde_i16, float* x_hat_f32, float* d_hat_f32, int64_t cap, int64_t* n_out);
/* Current B_DFT_d (what src/NTF_sep_event_RT.m:138-140 saves to B_D_u.mat). */
int snmf_online_get_basis_f32(snmf_online* o, float* B_DFT_d, int64_t ld);
/* Diagnostics of the most recent frames: the separator keeps a ring of the newest 65536 frames (a real-time stream
 * runs unbounded); copies the oldest min(cap, *n) of them in order, *n = frames held (= all frames for shorter runs). */
int snmf_online_trace(snmf_online* o, snmf_online_frame* out, int64_t cap, int64_t* n);
void snmf_online_destroy(snmf_online* o);

/* ---- multi-GPU solves: one process, several devices ------------------------------------- */
/* The reference's host is ONE MATLAB interpreter (run_basis_train.m:88, run_basis_DNMF.m:40,47,53 call sparse_nmf from
 * a single thread), so the MEX shim cannot bring a process-per-GPU launcher with it: this is the entry that puts the
 * multi-GPU path behind the same call.  The frame axis is sharded over `n_dev` ranks (columns are independent given
 * W, src/sparse_nmf.m:189-208), rank g on devices[g] (a device may appear more than once: the ranks then share it --
 * single-GPU testing); W is replicated.  Per iteration ONE exchange of the fp64 statistics
 *     [ G or Q | P (beta != 1) | rowsum(H) | div | sum(S.*H) ]        (src/sparse_nmf.m:215-239, :248-261)
 * as a one-shot all-reduce over peer-mapped (fine-grained) memory: every rank stores its buffer into its slot on every
 * peer, every rank sums the slots in rank order, so the W replicas and the stop decision (:272-284) are bit-identical everywhere.
 * H-only solves exchange only the two cost scalars.  Call order as for a plan: create -> set_v / set_w / set_h
 * [/ set_sparsity] -> init -> run -> get_*.  Matrices are HOST buffers of the WHOLE problem (V: F x T, H: r x T,
 * column-major; every rank's shard moves through its own pinned pipeline, all ranks at once); params->T is the total frame
 * count.  snmf_multi_run creates its rank threads per call and restores
 * the calling thread's current HIP device before it returns; the one-shot entries use min(n_dev, T) ranks.
 *   col_begin: n_dev + 1 ascending column offsets (col_begin[0] = 0, col_begin[n_dev] = T), or NULL = balanced. */
typedef struct snmf_multi snmf_multi;
int snmf_multi_create(const int32_t* devices, int32_t n_dev, const snmf_params* params, const int64_t* col_begin,
                      snmf_multi** out);
void snmf_multi_destroy(snmf_multi* m);
/* What a device list keeps for the life of the process (csrc/snmf_multi.h, MultiTeam): per rank two contexts -- each with
 * up to 3 x 16 MiB of pinned host bounce buffers, as much device staging and the cached device blocks of destroyed plans --
 * plus the fine-grained gather buffers, arrival words and events; an 8-device list retains about 1.5 GB of pinned host
 * memory after one run_basis_dnmf(devices = ...) call.  At most SNMF_TEAM_CACHE (default 2) idle device lists are kept; a
 * team on which a run FAILED is never kept (its exchange numbers and arrival words may disagree between the ranks).
 * snmf_multi_release_cache destroys every idle team now and returns how many it destroyed (teams in use are not touched):
 * call it when the host is done with its device lists -- a MEX file from its mexAtExit hook (integration/sparse_nmf_mex.cpp
 * does).  snmf_multi_cached_teams: idle teams currently held. */
int32_t snmf_multi_release_cache(void);
int32_t snmf_multi_cached_teams(void);
/* How push and sum of the per-iteration exchange are ordered (csrc/snmf_multi.h).  FLAGS: on the devices (arrival words
 * polled by the summing kernel; a rank's host thread only enqueues) -- AUTO picks it when every rank has a device of its
 * own.  EVENTS: hipEvents + a host barrier per iteration -- AUTO picks it when ranks share a device (single-GPU testing),
 * where FLAGS can stall on a shared hardware queue until its 5 s device-side time-out.  FLAGS across two PHYSICAL devices
 * has only been exercised where tests/test_gpu_multi_abi.py finds two devices.  Between init and the first run only. */
#define SNMF_EXCHANGE_AUTO 0
#define SNMF_EXCHANGE_FLAGS 1
#define SNMF_EXCHANGE_EVENTS 2
int snmf_multi_set_exchange(snmf_multi* m, int32_t mode);
int snmf_multi_set_v_f64(snmf_multi* m, const double* V, int64_t ld);
int snmf_multi_set_v_f32(snmf_multi* m, const float* V, int64_t ld);
int snmf_multi_set_w_f64(snmf_multi* m, const double* W, int64_t ld);
int snmf_multi_set_w_f32(snmf_multi* m, const float* W, int64_t ld);
int snmf_multi_set_h_f64(snmf_multi* m, const double* H, int64_t ld);
int snmf_multi_set_h_f32(snmf_multi* m, const float* H, int64_t ld);
/* r doubles (RVEC) or r x T, leading dimension r (FULL) */
int snmf_multi_set_sparsity_f64(snmf_multi* m, const double* S);
int snmf_multi_set_sparsity_f32(snmf_multi* m, const float* S);
int snmf_multi_init(snmf_multi* m);
/* as snmf_plan_run; one host thread per rank issues that rank's launches, the ranks meet only in the exchange */
int snmf_multi_run(snmf_multi* m, int32_t n_iters, int32_t* iters_done);
int snmf_multi_get_w_f64(snmf_multi* m, double* W, int64_t ld);
int snmf_multi_get_w_f32(snmf_multi* m, float* W, int64_t ld);
/* the replica of W held by one rank (the replicas are bit-identical by construction; tests check it) */
int snmf_multi_get_w_rank_f64(snmf_multi* m, int32_t rank, double* W, int64_t ld);
int snmf_multi_get_h_f64(snmf_multi* m, double* H, int64_t ld);
int snmf_multi_get_h_f32(snmf_multi* m, float* H, int64_t ld);
int snmf_multi_get_objective(snmf_multi* m, double* div_out, double* cost_out, int32_t* n_iter_out);
/* One-shot drop-in with a device list: snmf_sparse_nmf_f64 / _f32 sharded over `n_dev` ranks. */
int snmf_sparse_nmf_multi_f64(const int32_t* devices, int32_t n_dev, const snmf_params* p, const double* V, int64_t ldV,
                              double* W, double* H, const double* sparsity, double* div_out, double* cost_out,
                              int32_t* n_iter_out);
int snmf_sparse_nmf_multi_f32(const int32_t* devices, int32_t n_dev, const snmf_params* p, const float* V, int64_t ldV,
                              float* W, float* H, const float* sparsity, double* div_out, double* cost_out,
                              int32_t* n_iter_out);
/* B_hat = run_basis_DNMF(x, d, B, p) with a device list (run_basis_DNMF.m:36-55; BASELINE config 4): snmf_run_basis_dnmf_f64 / _f32
 * with the frames of all three solves sharded over `n_dev` ranks of this process -- the same arguments, `devices` in place of the
 * context.  Y, X, D cross PCIe once each (every rank its shard, all ranks at once, X and D under solve 1), each rank keeps ITS
 * columns of A_hat in HBM for solves 2 / 3, one exchange of the W statistics per iteration, only B_hat (and A_hat when asked
 * for) comes back.  H0 == NULL: rank g draws columns [col_g, col_g+1) of the (R_x + R_d) x T Philox draw, so the result does not
 * depend on the number of ranks beyond the order in which the ranks' statistics are summed (a few fp32 ulp on B_hat; A_hat
 * bit for bit).  The ranks' contexts, gather buffers and peer-access grants are built once per device list and kept for later calls
 * on the same list (csrc/snmf_multi.h: teams; the two most recently used idle lists, SNMF_TEAM_CACHE=n overrides, 0 = none). */
int snmf_run_basis_dnmf_multi_f64(const int32_t* devices, int32_t n_dev, const snmf_params* p, int32_t R_x, int32_t R_d,
                                  const double* Y, int64_t ldY, const double* X, int64_t ldX, const double* D, int64_t ldD,
                                  const double* B, int64_t ldB, const double* H0, uint64_t seed, double* B_hat, int64_t ldBh,
                                  double* A_hat, int64_t ldA, int32_t* n_iter_out);
int snmf_run_basis_dnmf_multi_f32(const int32_t* devices, int32_t n_dev, const snmf_params* p, int32_t R_x, int32_t R_d,
                                  const float* Y, int64_t ldY, const float* X, int64_t ldX, const float* D, int64_t ldD,
                                  const float* B, int64_t ldB, const float* H0, uint64_t seed, float* B_hat, int64_t ldBh,
                                  float* A_hat, int64_t ldA, int32_t* n_iter_out);

/* ---- instrumentation (bench.py: HIP-event timing on the engine's own stream) ------------ */
/* Average device time in milliseconds per launch of the named kernel family over the launches
 * recorded since snmf_ctx_timing(ctx, 1) was switched on.  Families: "hstep", "wstats",
 * "wapply", "reduce".  Timing inserts hipEvents around each launch (off by default). */
int snmf_ctx_timing(snmf_ctx* ctx, int enable);
/* Host <-> device transfer counters of a context since the last reset (host arrays move as a pipeline of column chunks through
 * pinned bounce buffers, csrc/snmf_tu_xfer.hip).  out8 = { host->device: bytes of the callers' arrays, wall seconds inside the
 * calls, seconds of host-side copying, calls;  device->host: the same four }. */
int snmf_ctx_xfer_stats(snmf_ctx* ctx, double* out8, int reset);
int snmf_ctx_timing_get(snmf_ctx* ctx, const char* family, double* avg_ms, int64_t* launches);
/* Kernel geometry chosen for a plan, for DESIGN.md / profiles bookkeeping. */
int snmf_plan_describe(const snmf_plan* plan, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* SNMF_H_ */
