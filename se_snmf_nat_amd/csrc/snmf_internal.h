// snmf_internal.h -- what the translation units of libsnmf_hip.so share on the HOST side: error plumbing, the context
// and plan structures, and the prototypes of the launch dispatchers.  The library is built from several .hip files
// compiled in parallel (se_snmf_nat_amd/_lib.py): snmf_api.hip (context, plans, data movement, the iteration loop, the
// front-end), snmf_tu_hstep*.hip / snmf_tu_wstats*.hip / snmf_tu_small.hip (the template instantiations of the three big
// kernel families and their dispatch), snmf_tu_online.hip, snmf_tu_multi.hip, snmf_tu_dnmf.hip.  Nothing here is part of
// the C ABI (include/snmf.h).
#pragma once
#include "snmf_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "snmf.h"

using namespace snmf;

// ---- errors (defined in snmf_api.hip) ----------------------------------------------------------
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? SNMF_ERR_NOMEM : SNMF_ERR_NO_DEVICE, "%s: %s", \
                        #expr, hipGetErrorString(e_));                                             \
    } while (0)
// lazy chain: the call is only MADE while no earlier one has failed, so the first failure's status AND message survive
#define SN_STEP(s, expr)                      \
    do {                                      \
        if ((s) == SNMF_OK) (s) = (expr);     \
    } while (0)
#define SN_TRY(expr)              \
    do {                          \
        int s_ = (expr);          \
        if (s_ != SNMF_OK) return s_; \
    } while (0)

// ---- context -----------------------------------------------------------------------------------
struct TimerPair {
    hipEvent_t a, b;
    int fam;
};
enum { FAM_HSTEP = 0, FAM_WSTATS, FAM_WAPPLY, FAM_REDUCE, FAM_WFIN, FAM_N };

struct HostXfer;  // pinned bounce buffers + copy stream of the chunked host <-> device pipeline (snmf_tu_xfer.hip)
struct XferStats {
    double h2d_bytes = 0, h2d_wall = 0, h2d_host = 0, h2d_calls = 0, d2h_bytes = 0, d2h_wall = 0, d2h_host = 0, d2h_calls = 0;
};

struct snmf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    size_t lds_max = 160 * 1024;
    bool timing = false;
    std::vector<TimerPair> pending;
    double fam_ms[FAM_N] = {0, 0, 0, 0, 0};
    int64_t fam_n[FAM_N] = {0, 0, 0, 0, 0};
    HostXfer* xf = nullptr;  // created by the first host-array transfer
    snmf_ctx* aux = nullptr; // a second stream + transfer pipeline on the same device (uploads under a running solve: snmf_tu_dnmf.hip)
    XferStats xs;
    // Device blocks of destroyed plans, kept for the next plan of this context (a caller that solves problem after problem of one
    // shape -- every MATLAB call site of sparse_nmf -- otherwise pays hipMalloc + hipFree of ~30 buffers per call: 3-5 ms).  Exact
    // size matches only; at most cache_cap bytes (SNMF_DEVCACHE_MB, default 4096; 0 = off); released with the context.
    std::vector<std::pair<void*, size_t>> cache;
    size_t cache_bytes = 0, cache_cap = (size_t)4096 << 20;
};
void* ctx_take(snmf_ctx* c, size_t bytes);          // a cached block of exactly `bytes`, or nullptr
void ctx_give(snmf_ctx* c, void* p, size_t bytes);   // hand a block back (cached, or freed when the cache is full)

struct ScopedTimer {
    snmf_ctx* c;
    TimerPair tp;
    bool on;
    ScopedTimer(snmf_ctx* c_, int fam) : c(c_), on(c_->timing) {
        if (on) {
            tp.fam = fam;
            hipEventCreate(&tp.a);
            hipEventCreate(&tp.b);
            hipEventRecord(tp.a, c->stream);
        }
    }
    ~ScopedTimer() {
        if (on) {
            hipEventRecord(tp.b, c->stream);
            c->pending.push_back(tp);
        }
    }
};

// ---- plan --------------------------------------------------------------------------------------
struct snmf_plan {
    snmf_ctx* ctx = nullptr;
    snmf_params p{};
    // geometry
    int Fp = 0, rp = 0, Tp = 0, nf = 0, nk = 0;
    int Fm = 0, Fq = 0, xr = 0;
    int NT = 1, NWH = 8, NLH = 0;  // k_hstep: frame tile = 32*NT, NWH consumer + NLH loader waves
    int TTH = 32, TTW = 32;        // frames per tile of k_hstep (NT == 1) / k_wstats; 16 = narrow tiles (images too big for 32 frames)
    bool hstep_rp = true;          // KL update launches of the (8, 1, 4) geometry use the role pipeline k_hstep_rp (SNMF_HSTEP_RP=0: k_hstep)
    // k_hstep_rp launch geometry: tiles [0, rp_full) through the pipeline on rp_grid workgroups, the tiles of the last
    // partial round [rp_full, rp_tiles) cut into rp_S row parts, one workgroup each (rp_S = 0: no split)
    int rp_tiles = 0, rp_full = 0, rp_S = 0, rp_grid = 1;
    int rp_cut = 0;                // k_hstep_rp<., CUT>: r <= 64, P2 cut over the contraction -- 1: four ways, every column tile; 2: wave pairs, a tile each (SNMF_RP_CUT=0: the column-tile deal)
    bool hm = false;               // KL update launches run k_hstep_m (merged roles, one wave per SIMD: snmf_hstep_m.h); SNMF_HSTEP_M=0/1
    int hm_grid = 1;
    bool rh = false;               // KL update launches run k_hstep_rh (9..16 row tiles, e.g. F = 513: one ratio image, pipelined by half tiles)
    size_t lds_rh = 0;
    float* part_buf = nullptr;     // partial numerators of the split tiles [rp_grid][32][rp]
    unsigned* part_cnt = nullptr;  // arrivals per split tile (monotonic)
    int NKT = 8, NWB = 4, WPS = 2, NLW = 0;  // k_wstats template geometry (NLW loader waves)
    int nbw = 2;                             // k_wstats tile buffers in LDS with loader waves (3 where they fit)
    int n_fg = 1, n_kg = 1, n_chunks = 1;
    bool sf = false;      // KL H-update launches through k_hstep_sf (F <= 64, r <= 128: snmf_smallf.h)
    int sf_grid = 1;
    bool wsf = false;     // KL statistics through k_wstats_sf (F <= 64, r <= 128)
    size_t lds_wsf = 0;
    bool isf_share = false;  // k_iter_sf: a chunk's single remainder tile is shared by the four pairs (SNMF_HSTEP_SPLIT=0: whole)
    bool wsf_share = false;  // k_wstats_sf / k_iter_sf: a workgroup's single remainder tile is shared by its waves (SNMF_HSTEP_SPLIT=0: whole)
    bool isf = false;     // full KL updates of those shapes: H step + W statistics in ONE launch (k_iter_sf); SNMF_ITER_SF=0 keeps two
    size_t lds_isf = 0;
    int sf_stagger = 0;   // cycles by which the second wave of each SIMD starts late (k_hstep_sf)
    size_t lds_sf = 0;
    // k_hstep_sf: the tiles [sf_nfull, rp_tiles) -- one per workgroup, the partial wave level behind the whole ones -- are shared by the
    // four waves of that level (snmf_smallf.h, "the shared last tile"); 0 = every tile whole
    int sf_share = 0, sf_nfull = 0;
    // small rank on tall spectrograms (r <= 64, 3..16 row tiles: snmf_smallr.h): a tile per WORKGROUP cut by row tiles, operands straight
    // into the MFMA layouts; KL H-update launches through k_hstep_sr, KL statistics through k_wstats_sr (SNMF_HSTEP_SR / SNMF_WSTATS_SR = 0: the role pipelines)
    bool sr = false, wsr = false;
    int sr_grid = 1;
    int sr_stagger = 1300;  // cycles by which the second wave of each SIMD starts late (k_hstep_sr / k_wstats_sr; SNMF_SR_STAG)
    size_t lds_sr = 0, lds_wsr = 0;
    int til = 1;  // k_wstats: consumer teams that share a chunk's tiles (StepArgs::til)
    int n_ch1 = 0;  // k_wstats: chunks of row group 1 when the two row groups are split unevenly (else 0)
    // beta = 2, r > 256: the V*H^T launch (needs no Lam') runs the loader-wave geometry <8,4,4,2> once per 256-column
    // kappa-group, each staging only its own columns of H (kq_chunks frame chunks, kq_kg kappa-groups; 0 = off)
    int kq_chunks = 0, kq_kg = 0;
    // snmf_plan_run: k_reduce + k_wapply as one launch (k_wfin) when a column's chunk-group sums fit the LDS
    int rh_lxh = 0;  // k_hstep_rh: P2 cut over the contraction (1: r = 97..100 four ways; 2: r = 193..200 in pairs), leftover columns as 4x4x1 MFMAs
    bool wfin = false;
    int wfin_S = 1;             // k_wfin: row slices per column (few columns: r <= 128), gathered by the column's last arriver
    double* qp_buf = nullptr;   // [r][n_mat * Fp]
    unsigned* fin_cnt = nullptr;  // [r] arrivals per column (monotonic)
    size_t lds_wfin = 0;
    // H-only loop of snmf_plan_run: the objective fold + convergence test ride on the H step (obj_partial_out in snmf_kernels.h:
    // the last workgroup to arrive folds) instead of a k_reduce launch per iteration.  SNMF_HFOLD=0 keeps the launch.
    bool fold_obj = false;
    FoldBlock fold_host{};         // what the FoldBlock behind *st holds (the source of the copy at plan creation)
    int fold_now = 0;              // > 0 while snmf_plan_run issues the H step that carries the test of that iteration
    // shapes beyond the fused kernels' LDS / register envelope: the same iteration with its intermediates in HBM
    // (csrc/snmf_generic.h); Lam / ratio / denominator images [Tp][Fp], numerator / denominator of the H update [Tp][rp]
    // Euclidean W step, r > 256, full updates: P = max(W*H, flr) * H' is formed as W * (H*H') -- the r x r Gram matrix
    // costs 2 r^2 T flop instead of the P launch's 4 F T r (C5: 4.75 -> ~2.5 ms); see launch_gram_p
    bool gram_p = false;
    int gram_chunks = 0;
    float *gram_slabs = nullptr, *gram32 = nullptr;
    bool generic = false;
    float *gLam = nullptr, *gR = nullptr, *gD = nullptr, *gNum = nullptr, *gDen = nullptr;
    size_t kq_lds = 0;
    int grid_h = 1;
    int ldh = 0, ldr = 0, ldhw = 0;
    int stagger_h = 0, stagger_w = 0;
    size_t lds_h = 0, lds_w = 0;
    int bm = BM_KL;
    int n_mat = 1;
    bool upd_h = true, upd_w = true;
    // device buffers
    float *V = nullptr, *H[2] = {nullptr, nullptr}, *Wt4 = nullptr, *Wk4 = nullptr;
    double* Wc = nullptr;  // fp64 master copy of W (see k_wapply)
    float* Wcf = nullptr;  // fp32 rounding of Wc, column-major [rp][Fp] (k_hsolve_frame)
    int frame_fb = 0, frame_kb = 0;  // register-block geometry of k_hsolve_frame (0: shape not admitted)
    float* M = nullptr;    // MDI: observed/missing mask in V's layout (src/snmf_mdi.m); non-null = MDI solve
    bool mdi_v_fresh = false, mdi_final = false;
    size_t lds_mdi = 0;
    int grid_mdi = 1;
    size_t lds_frame = 0;
    float *dphv = nullptr, *colsum = nullptr, *lamk = nullptr, *S = nullptr, *wx = nullptr;
    float *slabs = nullptr, *spart = nullptr;
    double *part = nullptr, *stats = nullptr, *divh = nullptr, *costh = nullptr, *wn = nullptr;
    DevState* st = nullptr;
    unsigned long long* prof = nullptr;
    uint8_t* w_ind = nullptr;
    void* staging = nullptr;
    size_t staging_bytes = 0;
    int n_part = 0;
    // the multi-device entry's fused exchange (csrc/snmf_multi.h): set around snmf_plan_wstats / snmf_plan_wapply of one
    // iteration, consumed (and cleared) by launch_reduce / launch_wapply
    struct Exchange {
        int n = 0;                       // ranks
        double* push_dst[16] = {};       // this rank's slot on every rank
        unsigned* push_flag[16] = {};    // FLAGS: this rank's arrival word on every rank (else nullptr)
        unsigned* push_done = nullptr;   // FLAGS: this rank's workgroup counter
        const double* gather = nullptr;  // this rank's gather buffer of the exchange's parity
        size_t len = 0;                  // doubles per slot
        const unsigned* gflags = nullptr;
        unsigned seq = 0;
    };
    const Exchange* xpush = nullptr;     // next launch_reduce pushes while it reduces
    const Exchange* xgather = nullptr;   // next launch_wapply sums the slots while it applies
    // state
    bool have_v = false, have_w = false, have_h = false, have_s = false, inited = false;
    bool w_dirty = true;      // W changed since its last normalisation (online: W stays, only V/H change)
    bool small = false;       // T <= 32 H-only solve: one persistent single-workgroup launch
    bool small_ok = false;
    size_t lds_small = 0;
    bool small_done = false;
    int cur = 0;          // H[cur] holds the current iterate
    int it_done = 0;      // update iterations launched
    bool final_done = false;
    double sh_const = 0.0;
    std::vector<uint8_t> h_w_ind;
    std::map<void*, size_t> blocks;  // device blocks of this plan by size (handed back to the context's cache on destruction)
};

static inline size_t roundup(size_t x, size_t m) { return (x + m - 1) / m * m; }

template <typename T>
static int dalloc(T** p, size_t n, snmf_ctx* cache_of = nullptr, size_t* bytes_out = nullptr) {
    *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (bytes_out) *bytes_out = bytes;
    if (cache_of)
        if (void* q = ctx_take(cache_of, bytes)) {
            *p = (T*)q;
            return SNMF_OK;
        }
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e != hipSuccess) return fail(SNMF_ERR_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    return SNMF_OK;
}

static inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 4096); }

// Every plan entry point starts from a clean HIP error state: hipGetLastError() is sticky per thread, so an error some
// EARLIER, unrelated call left behind (a refused device ordinal, the caller's own HIP code, torch) would otherwise be
// reported by the first kernel-launch check of this library as if the launch had failed.
#define PLAN_CHECK(pl)                                           \
    if (!(pl)) return fail(SNMF_ERR_INVALID, "plan is NULL");    \
    (void)hipGetLastError()

// ---- shared host functions ------------------------------------------------------------------------
// snmf_api.hip
int validate_params(const snmf_params* p);
StepArgs make_args(snmf_plan* pl);
int ensure_dyn_lds(int device, const void* kern, size_t lds);
int g_gemm(snmf_plan* pl, const float* A, long long rsA, long long csA, const float* B, long long rsB, long long csB, float* C,
           long long rsC, long long csC, int M, int N, int K, int kchunk, long long zC);
int generic_hstep(snmf_plan* pl, bool obj, bool upd);
int generic_wstats(snmf_plan* pl, bool obj);
int launch_wapply(snmf_plan* pl, const double* stats, int check_it, bool do_update, bool init_mode);
int read_state(snmf_plan* pl, DevState* hs);
int validate_stft(const snmf_stft_params* sp);
int stft_to_device(snmf_ctx* ctx, const snmf_stft_params* sp, const float* samples, int64_t n_samples, int samples_on_device,
                   float* dst, int64_t ld, int64_t n_frames);
int result_h_index(snmf_plan* pl, int* idx);
int rand_h(snmf_plan* pl, uint64_t seed, int64_t col0);  // snmf_tu_dnmf.hip: Philox draw of an initial H, columns [col0, col0 + T) of the r x n draw
template <typename T> int set_v(snmf_plan* pl, const T* V, int64_t ld, int dev);
template <typename T> int set_w(snmf_plan* pl, const T* W, int64_t ld, int dev);
template <typename T> int set_h(snmf_plan* pl, const T* H, int64_t ld, int dev);
template <typename T> int set_s(snmf_plan* pl, const T* S, int dev);
// snmf_tu_xfer.hip: a host matrix (column-major, leading dimension ld) <-> the padded device layout, as a chunked pipeline
template <typename TIn, typename TDst>
int xfer_pack_in(snmf_ctx* ctx, const TIn* src, int64_t ld, int rows, int cols, TDst* dst, int rowsP, int colsP, bool do_floor);
template <typename TOut, typename TSrc>
int xfer_unpack_out(snmf_ctx* ctx, const TSrc* src, int rowsP, int rows, int cols, TOut* dst, int64_t ld);
void xfer_destroy(snmf_ctx* c);
int xfer_sync(snmf_ctx* c);
// snmf_tu_hstep.hip / snmf_tu_wstats.hip / snmf_tu_small.hip
int launch_hstep(snmf_plan* pl, bool obj, bool upd);
int launch_hstep_rp(snmf_plan* pl, StepArgs a, bool obj);  // snmf_tu_hstep_rp.hip
int launch_hstep_rh(snmf_plan* pl, StepArgs a, bool obj);  // snmf_tu_hstep_rh.hip
int launch_hstep_m(snmf_plan* pl, StepArgs a, bool obj);   // snmf_tu_hstep_m.hip
int launch_hstep_sf(snmf_plan* pl, StepArgs a, bool obj);  // snmf_tu_smallf.hip
int launch_wstats_sf(snmf_plan* pl, const StepArgs& a, bool obj);  // snmf_tu_smallf.hip
int launch_hstep_sr(snmf_plan* pl, StepArgs a, bool obj);            // snmf_tu_smallr.hip
int launch_wstats_sr(snmf_plan* pl, const StepArgs& a, bool obj);   // snmf_tu_smallr.hip
int launch_iter_sf(snmf_plan* pl, bool obj);  // snmf_tu_smallf.hip: H step + W statistics of one full KL iteration, H[cur] -> H[cur ^ 1]
int launch_wstats(snmf_plan* pl, bool obj);
int launch_wstats_nk4(snmf_plan* pl, const StepArgs& a, bool obj);  // snmf_tu_wstats4.hip
int launch_wstats_nk8(snmf_plan* pl, const StepArgs& a, bool obj);  // snmf_tu_wstats8.hip
int launch_small(snmf_plan* pl, int n_solves, int tps, double* divh, double* costh, DevState* st, float* recon = nullptr,
                 int recon_rx = 0);

template <typename K>
static int launch_big(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, StepArgs a) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));  // every caller has set the plan's device
    SN_TRY(ensure_dyn_lds(dev, (const void*)kern, lds));
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    HIP_TRY(hipGetLastError());
    return SNMF_OK;
}
