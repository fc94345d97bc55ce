// snmf_kernels.h -- gfx950 (MI355X / CDNA4) device code of the sparse-NMF engine.
//
// Reference algorithm: lordet01/SE_SNMF_NAT  src/sparse_nmf.m:186-286 (multiplicative updates
// of H and W for the beta-divergence with an L1 penalty on H, W kept column-normalised).
//
// Design: everything is built from fp32 MFMA 32x32x2 tiles
// (v_mfma_f32_32x32x2_f32, exact f32, 64 cycles / SIMD).  A workgroup of NW waves owns a tile of
// Tt = 32*NT spectrogram frames.  The chain of contractions of one half-iteration runs inside
// ONE kernel, the F x Tt ratio tile lives only in LDS / registers and never goes to HBM:
//
//   k_hstep : Lam = W*H (P1) -> ratio tile in LDS -> W^T*ratio (P2) -> H <- H .* dmh ./ dph
//             (+ the divergence / cost sums of the PREVIOUS iterate, which P1's Lam is)
//   k_wstats: Lam'^T = H^T*W^T (P3) -> ratio in registers (already the A operand of) ->
//             G += ratio * H^T (P4), accumulated in registers over the workgroup's whole frame
//             chunk and written once as a split-T partial slab
//   k_reduce: partial slabs + partial objective sums -> fp64 statistics (fixed order)
//   k_wapply: F x r epilogue (dpw, dmw, update, column normalise) + convergence test
//   k_wfin  : k_reduce + k_wapply in one launch (the loop of one device: nothing is exchanged between them)
// KL update launches of the headline geometries are role pipelines of the same arithmetic: k_hstep_rp (two tile buffers in
// LDS), k_hstep_rh (F = 513: one ratio image, pipelined by half tiles); k_wstats has LDS-DMA loader waves.  Shapes that do
// not fit a tile's images into the LDS at all take snmf_generic.h (intermediates in HBM).
//
// MFMA 32x32x2 f32 operand map (lane l, fl = l&31, h = l>>5):
//   A[i=fl][k=h], B[k=h][j=fl], D[row=(reg&3)+8*(reg>>2)+4*h][col=fl]  (reg in [0,16)).
// The contraction index is visited in the permuted order k(q,h,e) = 8q+4h+e so that one 16-byte
// read per lane feeds four MFMAs, and so that a D tile can be handed to the next product as an
// operand without any lane movement (its row index IS that order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snmf {

typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kFlr = 1e-9f;  // src/sparse_nmf.m:166
constexpr int kMaxNW = 12;     // most waves per workgroup (reduction scratch sizing)

// beta modes (template parameter BM)
constexpr int BM_GEN = 0;  // generic beta (incl. IS, beta = 0)
constexpr int BM_KL = 1;   // beta == 1
constexpr int BM_EUC = 2;  // beta == 2

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_pow(float x, float y) {
    // Generic-beta path only (x > 0 always: clamped at 1e-9).  The bare transcendental-unit form
    // exp2(y*log2 x) carries a systematic ~1e-5 relative bias at |y*log2 x| ~ 30, which showed up
    // in the objective; the OCML powf is accurate to ~1 ulp.
    return powf(x, y);
}
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

// D-tile register -> row inside the 32-row tile
__device__ __forceinline__ constexpr int drow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// den = lam^(beta-1), and the map den -> lam^(beta-2) used to form num = v * lam^(beta-2)
template <int BM>
__device__ __forceinline__ float den_of_lam(float lam, float beta) {
    if (BM == BM_EUC) return lam;
    return fast_pow(lam, beta - 1.f);
}
template <int BM>
__device__ __forceinline__ float numfac_of_lam(float lam, float beta) {
    if (BM == BM_EUC) return 1.f;
    return fast_pow(lam, beta - 2.f);
}

// one element of the divergence sum, src/sparse_nmf.m:248-258
template <int BM>
__device__ __forceinline__ float div_term(float v, float lam, float beta, float inv_bb1) {
    if (BM == BM_KL) {
        // v*log(v/lam) - v + lam
        return v * (fast_ln(v * fast_rcp(lam)) - 1.f) + lam;
    } else if (BM == BM_EUC) {
        float d = v - lam;
        return d * d;
    } else {
        if (beta == 0.f) {
            float q = v * fast_rcp(lam);
            return q - fast_ln(q) - 1.f;
        }
        float lb1 = fast_pow(lam, beta - 1.f);
        return (fast_pow(v, beta) + (beta - 1.f) * lb1 * lam - beta * v * lb1) * inv_bb1;
    }
}

// Device-side solver state (fp64 objective history and the convergence flag).
struct DevState {
    int stop;      // set by the convergence test, src/sparse_nmf.m:275-281
    int n_iter;    // iterations whose objective has been recorded
    int fault;     // a bounded device-side wait (LDS producer/consumer counters) gave up: results are invalid
    int pad1;
};
// Index of this wave in its workgroup, as a SCALAR: `threadIdx.x >> 6` alone is a per-lane value to the compiler, and
// every role switch, row / column-tile loop and operand base pointer derived from it then becomes vector code under exec
// masks (with the register live ranges of all roles overlapping).  readfirstlane makes the uniformity visible.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// `stop` points at DevState::stop; the fault word sits two ints behind it
__device__ __forceinline__ void raise_fault(const int* stop) {
    if (stop) atomicExch(const_cast<int*>(stop) + 2, 1);
}
constexpr int kSpinLimit = 1 << 22;  // x >= 64 cycles of s_sleep: ~0.1 s, four orders of magnitude over a tile period

// ------------------------------------------------------------------------------------------
// Arguments shared by the two big kernels.  Layouts (all fp32, zero padded):
//   V   [Tp][Fp]            column-major F x T, leading dimension Fp
//   H   [Tp][rp]            column-major r x T, rp = 32*nk
//   Wt4 [nf][rp/8][2][32][4] = W[32*phi+f][8q+4h+e]   (A operand of W*H, B operand of H^T*W^T)
//   Wk4 [nk][Fq/8][2][32][4] = W[8q+4h+e][32*kap+k]   (A operand of W^T*ratio)
// Row geometry.  Spectrograms have F = 2^n + 1 rows (257, 513): padding that to a multiple of 32
// would waste a whole 32-row MFMA tile (+12.5 % flops at 257) and, worse, leave 9 tiles for 8
// waves.  In "extra-row" mode (xr = 1, F = 32*nf + 1) the MFMA tiles cover Fm = 32*nf rows and the
// last row is a 1 x r x Tt dot product on the VALU (1/257 of the flops, hidden under the MFMAs);
// it re-enters the W^T*ratio contraction as one more 8-deep k-block (Fq = Fm + 8).
//   xr = 0: Fm = Fp = Fq = 32*ceil(F/32).     xr = 1: Fp = Fm + 4, Fq = Fm + 8.
// ------------------------------------------------------------------------------------------
struct StepArgs {
    const float* V;
    const float* Hin;
    float* Hout;
    const float* Wt4;
    const float* Wk4;
    const float* dphv;    // KL, scalar/rvec sparsity: max(colsum(W)+lambda_k, flr)  [rp]
    const float* colsum;  // KL, full sparsity: colsum(W) [rp]
    const float* lamk;    // lambda_k [rp] (scalar/rvec kinds)
    const float* S;       // full sparsity r x T in H layout, or nullptr
    float* slabs;         // k_wstats: [n_chunks][n_mat][rp][Fp]
    float* spart;         // k_wstats: [n_chunks][rp]  partial row sums of H
    double* part;         // [grid][2] partial (div, sum S.*H)
    const int* stop;      // device flag: convergence reached -> kernels become no-ops
    unsigned long long* prof;  // -DSNMF_PROF diagnostic builds only: per-wave phase cycle sums
    const float* wx;      // extra-row mode: W[Fm, k]  [rp]
    const float* M;       // MDI: observed/missing mask in V's layout (1 = observed), src/snmf_mdi.m
    float* Vw;            // MDI: V, writable (re-imputed in place by the Lam pass)
    int impute;           // MDI: this pass carries the re-imputation of the previous iteration (:251-254)
    int n_ch1;            // k_wstats: > 0 = row group 1 has its own, smaller, number of frame chunks (1-D grid)
    int lxh;              // k_hstep_rh: P2 cut over the contraction, leftover columns on the VALU (nk = 4, r = 97..100)
    int kc;               // k_wstats, WM = 3 (V * H^T needs no Lam'): each kappa-group stages only ITS 32*NK columns of H
    int nbuf;             // k_wstats with loader waves: tile buffers in LDS (2, or 3 where they fit: the loaders then run two tiles ahead)
    int F, T, Fp, rp, Tp, nf, nk;
    int nqk;              // 8-deep k-blocks of the contractions over the components = ceil(r / 8): W's columns / H's rows
                          // r .. rp-1 are zero padding, so the blocks past it (r = 100: 3 of 16) only add zeros and are skipped
    int Fm;               // rows covered by MFMA tiles = 32*nf
    int Fq;               // contraction length of W^T*ratio = Fm + 8*xr
    int xr;               // 1: F = Fm + 1, the last row ("Nyquist bin") is handled on the VALU
    int n_tiles;          // tiles of this kernel's tile width
    int ldh, ldr;         // LDS leading dimensions (floats)
    int stagger;          // cycles by which half of the workgroups start late (0 = off)
    int stagger_shift;    // which half: bit `shift` of the linear block id (-1: upper half of the grid)
    // k_hstep_rp, split last round: tiles [n_full, n_tiles) are cut into part_S row parts (part p = the 32-row tiles
    // phi = p, p + part_S, ...; the last part also owns the extra row), one workgroup each; their partial W^T*ratio
    // numerators go to part_buf [(tile - n_full) * part_S + p][rp / 32][4][64] f32x4 (MFMA fragment order) and the last
    // of a tile's workgroups to arrive adds them in part order and applies the H update.  part_S = 0: no split.
    int n_full, part_S;
    float* part_buf;
    unsigned* part_cnt;   // [n_tiles - n_full] arrivals per split tile (monotonic; a launch adds part_S to each)
    int lam_is_u;         // scalar sparsity: every real row has the same lambda (pad rows of H are zero anyway)
    float lam_u;          // ... that lambda
    float beta, inv_bb1;
    int til;              // (LAST: a field in the middle moves the kernel-argument offsets of everything behind it, and the 168-VGPR
                          // geometries of k_wstats answered the different scalar loads with 176 spilled VGPRs)
                          // k_wstats with loader waves, FEWER row tiles than consumer waves (nf <= NWB / 2, no extra row): the consumers
                          // are til teams of NWB / til waves; team p takes the tiles it % til == p of the chunk and the teams' partial
                          // statistics are added through LDS at the end (fixed order).  0 / 1: every consumer wave on every tile
    // (APPENDED behind til, for the reason given there; ONE scalar: k_hstep_rp<true, false> holds 63 spilled SGPRs in one VGPR's
    // 64 lanes, and the seven words of a first version -- pointers, eps, counts as kernel arguments -- gave it a scratch slot.)
    // The H-only loop of snmf_plan_run, fold_it > 0: the objective fold and the convergence test of iteration fold_it ride on
    // this H step -- the workgroup that arrives LAST adds the partials in k_reduce's order and records (obj_partial_out below;
    // everything else it needs sits in the FoldBlock behind the plan's DevState); no k_reduce launch follows.
    int fold_it;
};
// behind the plan's DevState in the same allocation (stop -> DevState -> + 1), written once at plan creation
struct FoldBlock {
    unsigned cnt;     // arrivals (monotonic; a folding launch adds gridDim.x)
    int n;            // objective partials of an H-update launch (= its grid)
    double eps;       // conv_eps
    double* sc;       // (div, sum S.*H) of the statistics buffer
    double* divh;
    double* costh;
};


// Compiler fence for software pipelining.  hipcc otherwise sinks a prefetching load into the
// iteration that consumes it (observed: global_load -> s_waitcnt vmcnt(0) -> MFMA, i.e. the full
// L2 latency exposed every k-step).  Loads cannot move across this statement, so everything issued
// before it stays in flight while the MFMAs after it run.
// The asm statement pins loads at the IR level; sched_barrier(0) additionally stops the machine
// scheduler from hoisting the (memory-free) MFMAs above the loads, which has the same effect.
// Phase stamps (diagnostic build -DSNMF_PROF only; the shipped kernels contain none).
#ifdef SNMF_PROF
#define SNMF_STAMP_DECL unsigned long long pf_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long pl_ = __builtin_amdgcn_s_memtime(); \
    const unsigned long long pc0_ = pl_, pr0_ = __builtin_amdgcn_s_memrealtime();
#define SNMF_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); pf_[i] += t_ - pl_; pl_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SNMF_STAMP_OUT(base, nph) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < nph; ++i_) (base)[i_] = pf_[i_]; } while (0)
// start tick (100 MHz) of tile `it` of workgroup `wg`, consumer wave 0, first 16 tiles of the first 1024 workgroups
#define SNMF_STAMP_TILE(prof, wg, it) do { if (threadIdx.x == 0 && (it) < 16 && (wg) < 1024 && (prof)) \
    (prof)[98304 + 24576 + (size_t)(wg) * 16 + (it)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// in-kernel clock of this wave's stamped span: shader cycles over the 100 MHz real-time counter (slot pair idx behind the phase slots)
#define SNMF_STAMP_CLK(prof, idx) do { if ((threadIdx.x & 63) == 0) { (prof)[98304 + 2 * (idx)] = __builtin_amdgcn_s_memtime() - pc0_; \
    (prof)[98304 + 2 * (idx) + 1] = __builtin_amdgcn_s_memrealtime() - pr0_; (prof)[98304 + 16384 + (idx)] = pr0_; } } while (0)
#else
#define SNMF_STAMP_DECL
#define SNMF_STAMP(i)
#define SNMF_STAMP_OUT(base, nph)
#define SNMF_STAMP_CLK(prof, idx)
#define SNMF_STAMP_TILE(prof, wg, it)
#endif

// (-DSNMF_STRESS: the note in front of rp_post below)
#ifdef SNMF_STRESS
__device__ __forceinline__ void stress_jitter() {
    unsigned t = (unsigned)__builtin_amdgcn_s_memtime() ^ ((unsigned)threadIdx.x >> 6) * 0x85EBCA6Bu ^ (unsigned)blockIdx.x * 0xC2B2AE35u;
    t ^= t >> 7;
    t *= 0x9E3779B1u;
    t ^= t >> 15;
    const int n = __builtin_amdgcn_readfirstlane((t & 3u) == 0u ? (int)((t >> 2) & 63u) : 0);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#else
__device__ __forceinline__ void stress_jitter() {}
#endif

// Convergence test of src/sparse_nmf.m:260-285 for iteration `it` (1-based) whose (div, sh) sit
// in the reduced statistics.  Every workgroup evaluates it identically (no races: it reads only
// the statistics and the cost of iteration it-1, written by an earlier launch); one thread
// records.  Returns true when the loop must stop at `it`.
// last_pre: costh[it - 2] loaded by the caller ahead of time (k_wfin issues every global load of the launch up front), or nullptr
__device__ __forceinline__ bool conv_test(const double* sc, double* divh, double* costh, DevState* st, int it,
                                          double conv_eps, bool recorder, const double* last_pre = nullptr) {
    const double div = sc[0], cost = sc[0] + sc[1];
    bool stopnow = false;
    if (it > 1 && conv_eps > 0.0) {
        const double last = last_pre ? *last_pre : costh[it - 2];
        const double e = fabs(cost - last) / last;
        stopnow = e < conv_eps;
    }
    if (recorder) {
        divh[it - 1] = div;
        costh[it - 1] = cost;
        st->n_iter = it;
        if (stopnow) st->stop = 1;
    }
    return stopnow;
}

// The end of every H-step kernel with the objective: wave 0 hands in the workgroup's partial sums (d, s valid in lane 0; all 64
// lanes call).  Ordinarily that is two stores and k_reduce / k_wfin fold the partials of the launch.  In the H-only loop of
// snmf_plan_run (a.fold_it > 0) the fold rides on this launch: partials go out as agent-scope (sc1, write-through) stores, are
// acknowledged, then the arrival is counted; the wave that arrives last reads all of them with sc1 loads and adds them in
// k_reduce's order exactly -- thread t of its 256 adds partials t, t + 256, ... and a tree 128, 64, ..., 1 follows: lane l of
// this wave plays threads l, l + 64, l + 128, l + 192 -- so cost histories and stopping iterations are those of the separate launch
// to the bit.  (No fences: a __threadfence() would write back the H tiles this XCD's L2 still holds, see k_wfin.)
__device__ __forceinline__ void obj_partial_out(const StepArgs& a, int slot, double d, double s) {
    const int lane = threadIdx.x & 63;
    if (a.fold_it <= 0) {
        if (lane == 0) {
            a.part[2 * slot] = d;
            a.part[2 * slot + 1] = s;
        }
        return;
    }
    DevState* st = reinterpret_cast<DevState*>(const_cast<int*>(a.stop));
    FoldBlock* fb = reinterpret_cast<FoldBlock*>(st + 1);
    int last = 0;
    if (lane == 0) {
        __hip_atomic_store(a.part + 2 * slot, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part + 2 * slot + 1, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    stress_jitter();  // (-DSNMF_STRESS builds only)
    if (lane == 0) {
        const unsigned old = __hip_atomic_fetch_add(&fb->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old % gridDim.x) == gridDim.x - 1u;
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    const int n = fb->n;
    double xd[4], xs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        xd[q] = 0.0;
        xs[q] = 0.0;
        for (int c = lane + 64 * q; c < n; c += 256) {
            xd[q] += __hip_atomic_load(a.part + 2 * c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xs[q] += __hip_atomic_load(a.part + 2 * c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    xd[0] += xd[2];
    xs[0] += xs[2];
    xd[1] += xd[3];
    xs[1] += xs[3];
    xd[0] += xd[1];
    xs[0] += xs[1];
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        xd[0] += __shfl_down(xd[0], sh, 64);
        xs[0] += __shfl_down(xs[0], sh, 64);
    }
    if (lane == 0) {
        double sc[2] = {xd[0], xs[0]};
        fb->sc[0] = sc[0];
        fb->sc[1] = sc[1];
        conv_test(sc, fb->divh, fb->costh, st, a.fold_it, fb->eps, true);
    }
}

#define SNMF_PIN()                          \
    do {                                    \
        asm volatile("" ::: "memory");      \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)

// One k-block of a contraction: four MFMAs per frame sub-tile from one f32x4 of W and one f32x4 per sub-tile of the LDS image.
//   SWAP=false: W is the MFMA A operand (P1, P2);  SWAP=true: the LDS tile is A (P3).
template <int NT, bool SWAP>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[NT], const f32x4& w, const f32x4 (&sf)[NT]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            if (SWAP) acc[tau] = mfma32(sf[tau][e], w[e], acc[tau]);
            else acc[tau] = mfma32(w[e], sf[tau][e], acc[tau]);
        }
    }
}

// gate(): called once by the contraction loops that take one, after the W loads of the first stage have been issued and
// before the first LDS read: the caller's wait for the LDS image goes there, so the L2 round trip of the first fragments is
// spent while the wave waits anyway.
struct NoGate {
    __device__ __forceinline__ void operator()() const {}
};

// ---- operand fragments WITHOUT vector address arithmetic ------------------------------------------------------------
// For v_mfma_f32_32x32x2_f32 every other instruction a SIMD issues is time its matrix pipe does not get
// (scripts/mfma_samewave.hip), and an operand load is not just its own issue slot: with a 64-bit per-lane address
// (global_load) each W fragment cost ~30 cycles and each LDS fragment ~14 in the loops above -- 78 cycles per MFMA for
// k_wstats' P3 loop alone on a SIMD where the pipe needs 64 (scripts/p3_loop_probe.hip).  Here the W image is read
// through a buffer descriptor: lane offset fixed (16 B per lane), k-block offset SCALAR (q * 1024 B on the SALU), LDS
// fragments at immediate offsets from one moving base, and no clamps on the prefetch past the last block (the W image is
// bounds-checked by the descriptor, the LDS reads stay inside the workgroup's allocation: callers see to that).  The same
// probe: 69 cycles per MFMA, 66 with the 32 k-blocks of rp = 256 fully unrolled.
// The wave index that selects the image block must be scalar (wave_index()).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wimage_rsrc(const float* img, size_t n_floats) {
    const size_t bytes = n_floats * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, bytes > 0xfffffff0ull ? (int)0xfffffff0u : (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ldw_buf(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
// acc[tau] += sum over k-blocks q of Wfrag(q) (x) Sfrag_tau(q): W fragments by
// descriptor + scalar offset, LDS fragments at immediate offsets from one moving base per frame sub-tile, no clamps.
template <int NT, bool SWAP>
__device__ __forceinline__ void contract_buf(f32x16 (&acc)[NT], __amdgpu_buffer_rsrc_t rs, int voff, int soff0, const float* sp,
                                             int sstride, int nq) {
    f32x4 wA[2], wB[2], sA[2][NT], sB[2][NT];
    auto ldw = [&](int q) { return ldw_buf(rs, voff, soff0 + q * 1024); };
    const float* bp[NT];  // moving bases: block q + j of sub-tile tau at bp[tau] + 8 * j
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) bp[tau] = sp + tau * sstride;
    auto lds2 = [&](f32x4 (&sf)[2][NT], int j0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int tau = 0; tau < NT; ++tau) sf[j][tau] = *reinterpret_cast<const f32x4*>(bp[tau] + 8 * (j0 + j));
    };
    wA[0] = ldw(0);
    wA[1] = ldw(1);
    lds2(sA, 0);
    int q = 0;
    for (; q + 3 < nq; q += 4) {
        wB[0] = ldw(q + 2);
        wB[1] = ldw(q + 3);
        lds2(sB, 2);
        SNMF_PIN();
        mfma_block<NT, SWAP>(acc, wA[0], sA[0]);
        mfma_block<NT, SWAP>(acc, wA[1], sA[1]);
        wA[0] = ldw(q + 4);
        wA[1] = ldw(q + 5);
        lds2(sA, 4);
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) bp[tau] += 32;
        SNMF_PIN();
        mfma_block<NT, SWAP>(acc, wB[0], sB[0]);
        mfma_block<NT, SWAP>(acc, wB[1], sB[1]);
    }
    // remainder (nq % 4 blocks): wA / sA hold blocks q, q + 1
    if (q < nq) mfma_block<NT, SWAP>(acc, wA[0], sA[0]);
    if (q + 1 < nq) mfma_block<NT, SWAP>(acc, wA[1], sA[1]);
    if (q + 2 < nq) {
        const f32x4 w2 = ldw(q + 2);
        f32x4 s2[NT];
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) s2[tau] = *reinterpret_cast<const f32x4*>(bp[tau] + 16);
        mfma_block<NT, SWAP>(acc, w2, s2);
    }
}
// acc += sum_q Sfrag(q) (x) Wfrag(q)  (LDS tile = A operand, W = B: k_wstats' P3), one accumulator chain.
//   rs / voff / soff0: W image descriptor, this lane's byte offset (16 * lane), byte offset of the image block (scalar)
//   sp: this lane's LDS row;  gate(): see contract_sb.  UNROLL32: nq == 32 known at compile time (rp = 256).
template <bool UNROLL32, typename Gate>
__device__ __forceinline__ void contract_p3_buf(f32x16& acc, __amdgpu_buffer_rsrc_t rs, int voff, int soff0, const float* sp,
                                                int nq, Gate gate) {
    f32x4 wA[2], wB[2], sA[2], sB[2];
    // k-block q: its 4 KiB group (q >> 2) on the scalar offset, its place in the group as the instruction's 12-bit
    // immediate; the scalar base is made opaque so that the unrolled loop's offsets are re-derived per call on the SALU
    // instead of being hoisted out of the caller's tile loop into 32 live scalars (they were spilled to VGPR lanes and
    // read back with a v_readlane per load: k_wstats 0.2360 -> 0.2336 ms without that)
    int sbase = soff0;
    asm volatile("" : "+s"(sbase));
    auto ldw = [&](int q) { return ldw_buf(rs, voff + (q & 3) * 1024, sbase + (q >> 2) * 4096); };
    auto mmstage = [&](const f32x4 (&w)[2], const f32x4 (&sf)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(sf[j][e], w[j][e], acc);
    };
    wA[0] = ldw(0);
    wA[1] = ldw(1);
    SNMF_PIN();
    gate();
    sA[0] = *reinterpret_cast<const f32x4*>(sp);
    sA[1] = *reinterpret_cast<const f32x4*>(sp + 8);
    if (UNROLL32) {
#pragma unroll
        for (int q = 0; q < 32; q += 4) {
            wB[0] = ldw(q + 2);
            wB[1] = ldw(q + 3);
            sB[0] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 2));
            sB[1] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 3));
            SNMF_PIN();
            mmstage(wA, sA);
            wA[0] = ldw(q + 4);
            wA[1] = ldw(q + 5);
            sA[0] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 4));
            sA[1] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 5));
            SNMF_PIN();
            mmstage(wB, sB);
        }
        return;
    }
    const float* bp = sp;  // moving base: block q + j at bp + 8 * j
    int q = 0;
    for (; q + 3 < nq; q += 4) {
        wB[0] = ldw(q + 2);
        wB[1] = ldw(q + 3);
        sB[0] = *reinterpret_cast<const f32x4*>(bp + 16);
        sB[1] = *reinterpret_cast<const f32x4*>(bp + 24);
        SNMF_PIN();
        mmstage(wA, sA);
        wA[0] = ldw(q + 4);
        wA[1] = ldw(q + 5);
        sA[0] = *reinterpret_cast<const f32x4*>(bp + 32);
        sA[1] = *reinterpret_cast<const f32x4*>(bp + 40);
        bp += 32;
        SNMF_PIN();
        mmstage(wB, sB);
    }
    // remainder (nq % 4 blocks): wA / sA hold blocks q, q + 1
    if (q < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(sA[0][e], wA[0][e], acc);
    }
    if (q + 1 < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(sA[1][e], wA[1][e], acc);
    }
    if (q + 2 < nq) {
        const f32x4 w2 = ldw(q + 2);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(bp + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(s2[e], w2[e], acc);
    }
}

// Two workgroups share a CU.  Launched together with identical work they would run in lockstep:
// both staging (HBM burst, matrix pipe idle), then both issuing MFMAs (pipe contended).  Starting
// the second half of the grid half a tile period late makes one workgroup's staging / epilogues
// coincide with the other's MFMA phase for the rest of the kernel.  Purely a speed matter.
__device__ __forceinline__ void stagger_start(int cycles, unsigned linear_block, unsigned n_blocks, int shift) {
    const bool late = shift >= 0 ? ((linear_block >> shift) & 1u) != 0 : linear_block >= n_blocks / 2;
    if (cycles > 0 && late) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(32);
    }
}

// Cooperative copy of a contiguous [cols][rowlen] global tile into an LDS image with leading
// dimension ld (rowlen, ld multiples of 4 floats), and back.  Staging BOTH the H tile and the V
// tile through LDS keeps every HBM-latency access out of the MFMA loops: vmcnt retires in issue
// order, so a single outstanding HBM load (or store) in front of the W-fragment loads would stall
// the first wait of the loop for the whole HBM latency.
template <int NTHREADS>
__device__ __forceinline__ void stage_in(const float* __restrict__ src, float* dst, int cols, int rowlen, int ld,
                                         int tid) {
    const int r4 = rowlen / 4;
    const int n4 = cols * r4;
    constexpr int B = 8;  // loads in flight per thread and batch
    for (int i0 = tid; i0 < n4; i0 += B * NTHREADS) {
        f32x4 x[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * NTHREADS;
            if (i < n4) x[b] = *reinterpret_cast<const f32x4*>(src + 4 * (size_t)i);
        }
        SNMF_PIN();
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * NTHREADS;
            if (i < n4) {
                const int t = i / r4, k4 = i - t * r4;
                *reinterpret_cast<f32x4*>(dst + t * ld + 4 * k4) = x[b];
            }
        }
    }
}
// Two tiles (H and V) staged with ALL their loads in flight at once: one HBM round trip for a loader.
// BA / BB = f32x4 per thread for each array (compile-time bounds; guarded by the real counts).
template <int NTHREADS, int BA, int BB>
__device__ __forceinline__ void stage_in2(const float* __restrict__ srcA, float* dstA, int colsA, int rowlenA, int ldA,
                                          const float* __restrict__ srcB, float* dstB, int colsB, int rowlenB, int ldB,
                                          int tid) {
    const int rA = rowlenA / 4, nA = colsA * rA, rB = rowlenB / 4, nB = colsB * rB;
    if (nA > BA * NTHREADS || nB > BB * NTHREADS) {  // shape too big for one batch: generic path
        stage_in<NTHREADS>(srcA, dstA, colsA, rowlenA, ldA, tid);
        stage_in<NTHREADS>(srcB, dstB, colsB, rowlenB, ldB, tid);
        return;
    }
    f32x4 xa[BA], xb[BB];
#pragma unroll
    for (int b = 0; b < BA; ++b) {
        const int i = tid + b * NTHREADS;
        if (i < nA) xa[b] = *reinterpret_cast<const f32x4*>(srcA + 4 * (size_t)i);
    }
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const int i = tid + b * NTHREADS;
        if (i < nB) xb[b] = *reinterpret_cast<const f32x4*>(srcB + 4 * (size_t)i);
    }
    SNMF_PIN();
#pragma unroll
    for (int b = 0; b < BA; ++b) {
        const int i = tid + b * NTHREADS;
        if (i < nA) {
            const int t = i / rA, k4 = i - t * rA;
            *reinterpret_cast<f32x4*>(dstA + t * ldA + 4 * k4) = xa[b];
        }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b) {
        const int i = tid + b * NTHREADS;
        if (i < nB) {
            const int t = i / rB, k4 = i - t * rB;
            *reinterpret_cast<f32x4*>(dstB + t * ldB + 4 * k4) = xb[b];
        }
    }
}

template <int NTHREADS>
__device__ __forceinline__ void stage_out(float* __restrict__ dst, const float* src, int cols, int rowlen, int ld,
                                          int tid) {
    const int r4 = rowlen / 4;
    const int n4 = cols * r4;
    constexpr int B = 8;
    for (int i0 = tid; i0 < n4; i0 += B * NTHREADS) {
        f32x4 x[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * NTHREADS;
            if (i < n4) {
                const int t = i / r4, k4 = i - t * r4;
                x[b] = *reinterpret_cast<const f32x4*>(src + t * ld + 4 * k4);
            }
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int i = i0 + b * NTHREADS;
            if (i < n4) *reinterpret_cast<f32x4*>(dst + 4 * (size_t)i) = x[b];
        }
    }
}

// Geometry.  A workgroup has NW CONSUMER waves (they issue every MFMA) and NL LOADER waves.
// Loaders keep HBM latency away from the matrix pipe: while the consumers work on tile i they bring
// tile i+1 (its H and V blocks) into the other LDS buffer and copy the updated H tile i-1 out;
// consumer waves never touch HBM (their only global loads are L2-resident W fragments).
// vmcnt retires in issue order per wave, so this split -- not an async copy issued by the consumer
// itself -- is what keeps the W-fragment waits short.  NL = 0: consumers stage synchronously (used
// when the LDS budget of the shape leaves no room for a second buffer).
//   (NW=8, NT=1, NL=4): one workgroup per CU, 3 waves per SIMD, two LDS buffers   <- default
//   (NW=4, NT=1, NL=0): two workgroups per CU, synchronous staging
//   (NW=8, NT=2, NL=0): one workgroup per CU, W fragments shared by two frame sub-tiles
constexpr int kPF = 20;  // f32x4 a loader thread keeps in flight (covers 32*(rp+Fp) <= 20480 floats at NL=4)

// VG: V is read from global memory (persistent small-problem kernel: the LDS V image would be
// destroyed by the in-place ratio, and V is L2-resident there) instead of the staged LDS image.
// MDI (src/snmf_mdi.m:251-257): the Lam this pass forms is also the estimate that re-imputes V,
//   v <- max(v.*M + Lam.*(1-M), flr), written back for the W step, BEFORE the objective and the ratio use it.
// TT < 32 (NT == 1 only): NARROW tiles of TT frames for shapes whose 32-frame images do not fit the LDS (F + r > 1272).
// The MFMAs still span 32 columns; lanes fl >= TT feed them a duplicate of frame fl & (TT-1) and take no part in any
// epilogue, so (32-TT)/32 of the matrix work of this fallback path is wasted and nothing else changes.
template <int NW, int NT, int BM, bool OBJ, bool VG = false, bool MDI = false, int TT = 32>
__device__ __forceinline__ void hstep_p1_tiles(const StepArgs& a, float* Hs, float* Rs, int t0, int w, int lane,
                                               bool upd, double& acc_div) {
    constexpr int Tt = NT == 1 ? TT : 32 * NT;
    static_assert(TT == 32 || (NT == 1 && (TT == 16 || TT == 8)), "narrow tiles: one sub-tile of 16 or 8 frames");
    const int fl = lane & 31, h = lane >> 5;
    const int flt = fl & (Tt < 32 ? Tt - 1 : 31);  // LDS / global row this lane addresses
    const bool fvalid = Tt >= 32 || fl < Tt;
    const int rp = a.rp, ldh = a.ldh, ldr = a.ldr;
    // ---- P1: Lam[phi] = W[phi,:] * H[:, tile]  -> ratio / den image (in place over the staged V)
    for (int phi = w; phi < a.nf; phi += NW) {
        f32x16 acc[NT];
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) acc[tau] = zero16();
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wt4 + (size_t)phi * rp * 32) + lane;
        f32x4 vfr[NT][4], mfr[NT][4];
        if (VG || MDI) {
#pragma unroll
            for (int tau = 0; tau < NT; ++tau)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const size_t off = (size_t)(t0 + tau * 32 + flt) * a.Fp + phi * 32 + 4 * h + 8 * g;
                    vfr[tau][g] = *reinterpret_cast<const f32x4*>(a.V + off);
                    if (MDI) mfr[tau][g] = *reinterpret_cast<const f32x4*>(a.M + off);
                }
        }
        contract_buf<NT, false>(acc, wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32), lane * 16, phi * rp * 128, Hs + flt * ldh + 4 * h,
                                32 * ldh, a.nqk);
        // epilogue: lane (t = fl, h), reg -> f = 32*phi + drow(reg,h)
        float dsum = 0.f;
        const bool interior = OBJ && Tt >= 32 && phi * 32 + 32 <= a.F && t0 + Tt <= a.T;
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            if (!fvalid) break;
            const int t = t0 + tau * 32 + fl;
            float* rsp = Rs + (tau * 32 + fl) * ldr + phi * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = (VG || MDI) ? vfr[tau][g] : *reinterpret_cast<const f32x4*>(rsp + 8 * g);  // staged V
                if (MDI && a.impute) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = phi * 32 + 8 * g + 4 * h + j;
                        const float lam = fmaxf(acc[tau][4 * g + j], kFlr), mk = mfr[tau][g][j];
                        v[j] = (f < a.F && t < a.T) ? fmaxf(v[j] * mk + lam * (1.f - mk), kFlr) : 0.f;  // pads stay zero
                    }
                    *reinterpret_cast<f32x4*>(a.Vw + (size_t)t * a.Fp + phi * 32 + 4 * h + 8 * g) = v;
                }
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lam = fmaxf(acc[tau][4 * g + j], kFlr);
                    if (OBJ) {
                        const int f = phi * 32 + 8 * g + 4 * h + j;
                        float d = div_term<BM>(v[j], lam, a.beta, a.inv_bb1);
                        // (bounds masks only where the tile has padding: a wave-uniform test; the unmasked add stays
                        //  un-contracted so that both forms round alike, see k_wstats)
                        if (interior) {
#pragma clang fp contract(off)
                            dsum = dsum + d;
                        } else {
                            dsum += (f < a.F && t < a.T) ? d : 0.f;
                        }
                    }
                    if (BM == BM_KL) o[j] = v[j] * fast_rcp(lam);
                    else o[j] = den_of_lam<BM>(lam, a.beta);
                }
                if (upd) *reinterpret_cast<f32x4*>(rsp + 8 * g) = o;
            }
        }
        if (OBJ) acc_div += (double)dsum;
    }
}

// ---- cross-lane sums on the DPP path (1 VALU op per step; __shfl_* goes through ds_bpermute) ------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over each aligned group of 4 lanes, result in all 4
__device__ __forceinline__ float quad_sum_f(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
// sum over each row of 16 lanes, result in all 16
__device__ __forceinline__ float row_sum_f(float v) {
    v = quad_sum_f(v);
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}
// sum over the wave, result wave-uniform
__device__ __forceinline__ float wave_sum_f(float v) {
    v = row_sum_f(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)) +
           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
}

// The sum over each row of 16 lanes as the steps xor 8, 4, 2, 1 of a butterfly on the DPP path (two v_mov_dpp + the add per step;
// __shfl_xor is two ds_bpermute round trips per step, and these sums sit on the latency chain of k_wfin / k_wapply: phase stamps in
// profiles/r06_experiments.md section 10).  row_ror:8 pairs lane i with i xor 8; row_ror:4 reads lane (i - 4) mod 16 of the row,
// whose value -- every lane pair (i, i xor 8) holds the same sum by then -- is lane (i xor 4)'s; the quad permutations are xor 2 and
// xor 1 themselves.  A fixed order on every rank and in both callers (k_wfin = k_reduce + k_wapply bit for bit).
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double butterfly_8421(double v) {
    v += dpp_d<0x128>(v);  // row_ror:8
    v += dpp_d<0x124>(v);  // row_ror:4
    v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
    return v;
}

// fixed-order sum over the 256 threads of a workgroup: each wave's sum (the same tree on every rank), then the four wave sums in wave
// order.  One workgroup barrier per call instead of the eight of an LDS tree.
// a wave's sum, wave-uniform: the four row sums (DPP), read off lanes 0 / 16 / 32 / 48 and added in row order
__device__ __forceinline__ double rdlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_total_d(double v) {
    v = butterfly_8421(v);
    return (rdlane_d(v, 0) + rdlane_d(v, 16)) + (rdlane_d(v, 32) + rdlane_d(v, 48));
}
__device__ __forceinline__ double wg_sum_256(double v, double* scratch /*[4]*/, int tid) {
    v = wave_total_d(v);
    __syncthreads();  // scratch of the previous call has been read
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// two sums at once (each with wg_sum_256's tree, so each comes out bit for bit as a call of its own would): one pair of barriers
// and one exposed shuffle chain instead of two -- the F x r epilogue is a chain of latencies on a mostly idle chip
__device__ __forceinline__ void wg_sum2_256(double& v0, double& v1, double* scratch0 /*[4]*/, double* scratch1 /*[4]*/, int tid) {
    v0 = wave_total_d(v0);
    v1 = wave_total_d(v1);
    __syncthreads();
    if ((tid & 63) == 0) {
        scratch0[tid >> 6] = v0;
        scratch1[tid >> 6] = v1;
    }
    __syncthreads();
    v0 = (scratch0[0] + scratch0[1]) + (scratch0[2] + scratch0[3]);
    v1 = (scratch1[0] + scratch1[1]) + (scratch1[2] + scratch1[3]);
}

// extra row (F = 32*nf + 1): lam_x[t] = sum_k W[Fm,k] H[k,t] on the VALU, 4 columns x 16 lanes at a time
template <int NW, int NT, int BM, bool OBJ, bool VG = false, bool MDI = false, int TT = 32>
__device__ __forceinline__ void hstep_p1_xrow(const StepArgs& a, float* Hs, float* Rs, const float* wxs, int t0, int w,
                                              int lane, bool upd, double& acc_div) {
    constexpr int Tt = NT == 1 ? TT : 32 * NT;
    const int rp = a.rp, ldh = a.ldh, ldr = a.ldr;
    // narrow tiles: fewer than 4 frames per wave -> the first Tt/4 waves take 4 frames each
    constexpr int CPW = Tt >= 4 * NW ? Tt / NW : 4;
    if (a.xr && w * CPW < Tt) {
        // extra row: lam_x[t] = sum_k W[Fm,k] H[k,t]; 4 columns x 16 lanes at a time
        float dsum = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < CPW; c0 += 4) {
            const int tl = w * CPW + c0 + (lane >> 4);
            const int kl = lane & 15;
            const float* hrow = Hs + tl * ldh;
            float s0 = 0.f, s1 = 0.f;
            for (int k = 4 * kl; k < rp; k += 64) {  // 16 lanes x 4 consecutive k per step
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wxs + k);
                const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + k);
                s0 += wv[0] * hv[0] + wv[1] * hv[1];
                s1 += wv[2] * hv[2] + wv[3] * hv[3];
            }
            const float s = row_sum_f(s0 + s1);  // DPP: the same balanced tree as an xor butterfly, no ds_bpermute
            if (kl == 0) {
                const int t = t0 + tl;
                float v = (VG || MDI) ? a.V[(size_t)t * a.Fp + a.Fm] : Rs[tl * ldr + a.Fm];  // staged V
                const float lam = fmaxf(s, kFlr);
                if (MDI && a.impute) {
                    const float mk = a.M[(size_t)t * a.Fp + a.Fm];
                    v = t < a.T ? fmaxf(v * mk + lam * (1.f - mk), kFlr) : 0.f;
                    a.Vw[(size_t)t * a.Fp + a.Fm] = v;
                }
                if (OBJ) dsum += (t < a.T) ? div_term<BM>(v, lam, a.beta, a.inv_bb1) : 0.f;
                if (upd) Rs[tl * ldr + a.Fm] = (BM == BM_KL) ? v * fast_rcp(lam) : den_of_lam<BM>(lam, a.beta);
            }
        }
        if (OBJ) acc_div += (double)dsum;
    }
}

// P1 of one wave.  The second half of the waves (the SIMD partners of the first half) run their
// VALU-only extra-row work FIRST: the two waves of a SIMD then reach their MFMA loops, and later
// their VALU epilogues, at different times instead of colliding on both.
template <int NW, int NT, int BM, bool OBJ, bool VG = false, bool MDI = false, int TT = 32>
__device__ __forceinline__ void hstep_p1(const StepArgs& a, float* Hs, float* Rs, const float* wxs, int t0, int w,
                                         int lane, bool upd, double& acc_div) {
    const bool xfirst = a.xr && (w >= NW / 2);
    if (xfirst) hstep_p1_xrow<NW, NT, BM, OBJ, VG, MDI, TT>(a, Hs, Rs, wxs, t0, w, lane, upd, acc_div);
    hstep_p1_tiles<NW, NT, BM, OBJ, VG, MDI, TT>(a, Hs, Rs, t0, w, lane, upd, acc_div);
    if (a.xr && !xfirst) hstep_p1_xrow<NW, NT, BM, OBJ, VG, MDI, TT>(a, Hs, Rs, wxs, t0, w, lane, upd, acc_div);
}

// beta != 1: in-place transform of this wave's part of the image, den = lam^(b-1) -> num = V .* lam^(b-2)
template <int NW, int NT, int BM, int TT = 32>
__device__ __forceinline__ void hstep_den_to_num(const StepArgs& a, float* Rs, int t0, int w, int lane) {
    constexpr int Tt = NT == 1 ? TT : 32 * NT;
    const int fl = lane & 31, h = lane >> 5;
    const int Fp = a.Fp, ldr = a.ldr;
    for (int phi = w; phi < a.nf; phi += NW) {
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            if (Tt < 32 && fl >= Tt) break;  // narrow tiles: lanes past the tile own no frame
            const int t = t0 + tau * 32 + fl;
            const float* vp = a.V + (size_t)t * Fp + phi * 32 + 4 * h;
            float* rsp = Rs + (tau * 32 + fl) * ldr + phi * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = *reinterpret_cast<const f32x4*>(vp + 8 * g);
                f32x4 d = *reinterpret_cast<f32x4*>(rsp + 8 * g);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BM == BM_EUC) o[j] = v[j];
                    else {
                        // den = lam^(b-1)  ->  lam^(b-2) = den^((b-2)/(b-1))
                        float lf = (a.beta == 0.f) ? d[j] * d[j] : fast_pow(d[j], (a.beta - 2.f) / (a.beta - 1.f));
                        o[j] = v[j] * lf;
                    }
                }
                *reinterpret_cast<f32x4*>(rsp + 8 * g) = o;
            }
        }
    }
    constexpr int CPW = Tt >= 4 * NW ? Tt / NW : 4;
    if (a.xr && w * CPW < Tt) {
        if ((lane & 15) == 0) {
#pragma unroll
            for (int c0 = 0; c0 < CPW; c0 += 4) {
                const int tl = w * CPW + c0 + (lane >> 4);
                const float v = a.V[(size_t)(t0 + tl) * Fp + a.Fm];
                const float d = Rs[tl * ldr + a.Fm];
                float o;
                if (BM == BM_EUC) o = v;
                else o = v * ((a.beta == 0.f) ? d * d : fast_pow(d, (a.beta - 2.f) / (a.beta - 1.f)));
                Rs[tl * ldr + a.Fm] = o;
            }
        }
    }
}

// ---- P2: contraction over f with W^T.  KL: dmh = W^T*ratio; H <- H .* dmh ./ dphv.
// beta != 1, pass 0: dph = W^T*den + S; Hs <- H ./ max(dph, flr);   pass 1: dmh = W^T*num; H <- Hs .* dmh
template <int NW, int NT, int BM, bool OBJ, int TT = 32>
__device__ __forceinline__ void hstep_p2(const StepArgs& a, float* Hs, const float* Rs, int t0, int w, int lane,
                                         int pass, double& acc_sh) {
    constexpr int Tt = NT == 1 ? TT : 32 * NT;
    const int fl = lane & 31, h = lane >> 5;
    const int flt = fl & (Tt < 32 ? Tt - 1 : 31);
    const bool fvalid = Tt >= 32 || fl < Tt;
    const int rp = a.rp, ldh = a.ldh, ldr = a.ldr;
    for (int kap = w; kap < a.nk; kap += NW) {
        f32x16 acc[NT];
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) acc[tau] = zero16();
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wk4 + (size_t)kap * a.Fq * 32) + lane;
        // per-k constants of the epilogue, issued before the MFMA loop
        f32x4 dpf[4], spf[4];
        if (!a.S) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k0 = kap * 32 + 8 * g + 4 * h;
                if (BM == BM_KL) dpf[g] = *reinterpret_cast<const f32x4*>(a.dphv + k0);
                if (OBJ || BM != BM_KL) spf[g] = *reinterpret_cast<const f32x4*>(a.lamk + k0);
            }
        }
        contract_buf<NT, false>(acc, wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32), lane * 16, kap * a.Fq * 128,
                                Rs + flt * ldr + 4 * h, 32 * ldr, a.Fq / 8);
        // epilogue: lane (t = fl, h), reg -> k = 32*kap + drow(reg,h)
        float shsum = 0.f;
#pragma unroll
        for (int tau = 0; tau < NT; ++tau) {
            if (!fvalid) break;
            const int t = t0 + tau * 32 + fl;
            float* hsp = Hs + (tau * 32 + fl) * ldh + kap * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k0 = kap * 32 + 8 * g + 4 * h;
                f32x4 ho = *reinterpret_cast<f32x4*>(hsp + 8 * g);
                f32x4 sp;
                if (a.S) sp = *reinterpret_cast<const f32x4*>(a.S + (size_t)t * rp + k0);
                else sp = spf[g];
                f32x4 o;
                if (BM == BM_KL) {
                    f32x4 dp;
                    if (a.S) {
                        f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) dp[j] = fmaxf(cs[j] + sp[j], kFlr);
                    } else {
                        dp = dpf[g];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = ho[j] * acc[tau][4 * g + j] * fast_rcp(dp[j]);
                        if (OBJ) shsum += sp[j] * ho[j];
                    }
                } else if (pass == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float dp = fmaxf(acc[tau][4 * g + j] + sp[j], kFlr);
                        o[j] = ho[j] * fast_rcp(dp);
                        if (OBJ) shsum += sp[j] * ho[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = ho[j] * acc[tau][4 * g + j];
                }
                *reinterpret_cast<f32x4*>(hsp + 8 * g) = o;  // in place; copied out after P2
            }
        }
        if (OBJ) acc_sh += (double)shsum;
    }
}

// ============================================================================================
// k_hstep: H half-step, src/sparse_nmf.m:189-208, fused with the objective of the previous
// iterate (:248-261).  UPD=false gives the objective-only pass.
//
// NL = 0: NW consumer waves, one LDS buffer, the consumers stage each tile synchronously.
// NL > 0: NW consumers + NL loaders, TWO LDS buffers.  Barrier schedule of tile i (every wave,
//         whatever its role, executes the same sequence; buffers cur = i&1, nxt = cur^1):
//   B1 | consumers: P1(cur)            loaders: copy H tile i-1 out of nxt, issue tile i+1 -> regs
//   B2 | consumers: P2(cur) [+2 barriers for beta != 1]     loaders: regs -> nxt
// so the only thing the consumers ever wait for is each other.
// KL update launches (one pass per tile) drop the consumers' half of B1: P1 of tile i touches only buffer cur, so a
// consumer that has finished P2 of tile i-1 does not have to wait for the others -- it bumps an LDS counter
// (arrive only) and goes on; the LOADERS wait for the counter before they touch nxt.  One workgroup barrier per
// tile instead of two, and a wave's epilogue overlaps its neighbours' MFMA loop across the tile boundary.  The
// loaders' wait is a bounded spin: a lost signal raises DevState::fault (the host then fails the call), never a hang.
// ============================================================================================
template <int NW, int NT, int NL, int BM, bool OBJ, bool UPD, bool MDI = false, int TT = 32>
__global__ __launch_bounds__((NW + NL) * 64, (NL > 0 ? 3 : 2)) void k_hstep(StepArgs a) {
    static_assert(!MDI || NL == 0, "the MDI pass reads and rewrites V in global memory: synchronous staging only");
    static_assert(TT == 32 || (NL == 0 && NT == 1 && !MDI), "narrow tiles: synchronous staging, one sub-tile");
    constexpr int NTHR = (NW + NL) * 64;
    constexpr int NBUF = NL > 0 ? 2 : 1;
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Tt = NT == 1 ? TT : 32 * NT;
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp, ldh = a.ldh, ldr = a.ldr;
    const int bufsz = Tt * (ldh + ldr);  // floats per buffer: Hs [Tt][ldh] then Rs [Tt][ldr]
    float* wxs = lds + NBUF * bufsz;     // [rp]  extra row of W
    double acc_div = 0.0, acc_sh = 0.0;
    if (a.xr) {
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
        // the 7 unused cells of the extra 8-deep k-block stay zero for the whole kernel
        for (int i = threadIdx.x; i < NBUF * Tt * 8; i += NTHR) {
            const int bsel = i / (Tt * 8), ii = i - bsel * Tt * 8;
            lds[bsel * bufsz + Tt * ldh + (ii >> 3) * ldr + a.Fm + (ii & 7)] = 0.f;
        }
    }
    constexpr int NPASS = (BM == BM_KL) ? 1 : 2;
    constexpr bool SIG = NL > 0 && UPD && NPASS == 1;
    unsigned* sig = reinterpret_cast<unsigned*>(wxs + rp);  // consumers' "P2 of the previous tile done" count
    if (SIG && threadIdx.x == 0) *sig = 0u;

    if (NL > 0 && w >= NW) {
        // ================================ loader role =========================================
        // Same index -> (row, column) map for the copy-out and the load of an H image, so a loader
        // thread only ever re-writes cells it has itself read: no loader-loader barrier is needed.
        constexpr int NLT = NL * 64;
        const int lt = threadIdx.x - NW * 64;
        int tile = blockIdx.x, it = 0, prev = -1;
        if (tile < a.n_tiles) {
            stage_in<NLT>(a.Hin + (size_t)tile * Tt * rp, lds, Tt, rp, ldh, lt);
            stage_in<NLT>(a.V + (size_t)tile * Tt * Fp, lds + Tt * ldh, Tt, Fp, ldr, lt);
        }
        for (; tile < a.n_tiles; tile += gridDim.x, ++it) {
            float* nH = lds + ((it & 1) ^ 1) * bufsz;
            const int nt = tile + (int)gridDim.x;
            if (!SIG || it == 0) {
                __syncthreads();  // B1
            } else {
                const unsigned target = (unsigned)(NW * it);
                int spin = 0;
                while (__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                    if (++spin > kSpinLimit) {
                        raise_fault(a.stop);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (UPD && prev >= 0) stage_out<NLT>(a.Hout + (size_t)prev * Tt * rp, nH, Tt, rp, ldh, lt);
            if (nt < a.n_tiles)  // both blocks of tile i+1, one HBM round trip (the ratio image of tile i-1 is dead)
                stage_in2<NLT, 10, 10>(a.Hin + (size_t)nt * Tt * rp, nH, Tt, rp, ldh, a.V + (size_t)nt * Tt * Fp,
                                       nH + Tt * ldh, Tt, Fp, ldr, lt);
            __syncthreads();  // B2
            if (UPD && NPASS == 2) {
                __syncthreads();
                __syncthreads();
            }
            prev = tile;
        }
        __syncthreads();  // the last tile's P2 is complete
        if (UPD && prev >= 0)
            stage_out<NLT>(a.Hout + (size_t)prev * Tt * rp, lds + ((it - 1) & 1) * bufsz, Tt, rp, ldh, lt);
    } else {
        // ================================ consumer role =======================================
        if (NL == 0) stagger_start(a.stagger, blockIdx.x, gridDim.x, a.stagger_shift);
        SNMF_STAMP_DECL
        int it = 0;
        for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x, ++it) {
            const int t0 = tile * Tt;
            float* Hs = lds + (NL > 0 ? (it & 1) * bufsz : 0);
            float* Rs = Hs + Tt * ldh;  // V tile, overwritten in place by the ratio tile
            if (NL == 0) {
                __syncthreads();  // previous tile fully consumed / copied out
                SNMF_STAMP(0);
                stage_in<NTHR>(a.Hin + (size_t)t0 * rp, Hs, Tt, rp, ldh, threadIdx.x);
                stage_in<NTHR>(a.V + (size_t)t0 * Fp, Rs, Tt, Fp, ldr, threadIdx.x);
                SNMF_STAMP(1);
            }
            if (!SIG || it == 0) {
                __syncthreads();  // B1
            } else {  // arrive only: this wave's P2 of the previous tile (its LDS writes included) is complete
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            SNMF_STAMP(2);
            SNMF_STAMP_TILE(a.prof, blockIdx.x, it);
            hstep_p1<NW, NT, BM, OBJ, false, MDI, TT>(a, Hs, Rs, wxs, t0, w, lane, UPD, acc_div);
            SNMF_STAMP(4);
            if (UPD || NL > 0) __syncthreads();  // B2
            SNMF_STAMP(7);
            if (UPD) {
                hstep_p2<NW, NT, BM, OBJ, TT>(a, Hs, Rs, t0, w, lane, 0, acc_sh);
                SNMF_STAMP(9);
                if (NPASS == 2) {
                    __syncthreads();  // every wave finished reading the den image
                    hstep_den_to_num<NW, NT, BM, TT>(a, Rs, t0, w, lane);
                    __syncthreads();
                    double dummy = 0.0;
                    hstep_p2<NW, NT, BM, false, TT>(a, Hs, Rs, t0, w, lane, 1, dummy);
                }
                if (NL == 0) {
                    __syncthreads();  // the updated H tile leaves through one coalesced copy
                    SNMF_STAMP(10);
                    stage_out<NTHR>(a.Hout + (size_t)t0 * rp, Hs, Tt, rp, ldh, threadIdx.x);
                    SNMF_STAMP(11);
                }
            }
        }
        if (NL > 0) __syncthreads();  // matches the loaders' final barrier
        SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * NW + w) * 12, 12);
        SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * NW + w);
    }

    if (OBJ) {
        // deterministic workgroup reduction of the two fp64 partial sums
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);  // [2][NTHR]
        red[threadIdx.x] = acc_div;
        red[NTHR + threadIdx.x] = acc_sh;
        __syncthreads();
        // NTHR is 256, 512 or 768: fold the tail above the largest power of two first
        constexpr int P2 = NTHR >= 512 ? 512 : 256;
        if ((int)threadIdx.x + P2 < NTHR) {
            red[threadIdx.x] += red[threadIdx.x + P2];
            red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + P2];
        }
        __syncthreads();
        for (int s = P2 / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                red[threadIdx.x] += red[threadIdx.x + s];
                red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) obj_partial_out(a, blockIdx.x, red[0], red[NTHR]);
    }
}

// ============================================================================================
// k_hstep_rp: the KL H half-step as a ROLE PIPELINE (round 2).  Same arithmetic as k_hstep<8,1,4,BM_KL,OBJ,true>,
// different schedule.  In k_hstep all eight consumer waves run P1 of a tile, meet at a barrier, then all run P2.
// Here the consumer waves are split by ROLE, one wave of each role per SIMD:
//     A team (waves 0-3) : P1 of tile i+1   Lam = W*H -> ratio image (in place over the staged V)
//     B team (waves 4-7) : P2 of tile i     W^T*ratio -> H update in LDS
//     loaders (waves 8-11): copy the updated H tile i out and bring tile i+2 into the buffer that tile i leaves
//     (the extra row, F = 32n+1, is the A team's work after its last epilogue of a tile)
// What this buys is NOT "one wave's epilogue under another wave's MFMAs": while a wave issues back-to-back MFMAs of
// this shape no other wave of its SIMD issues anything (scripts/mfma_valu_overlap.hip), so a SIMD's time is its MFMA
// cycles plus everything else its waves issue.  What a schedule can avoid is a SIMD on which every wave WAITS.  The two
// MFMA waves of a SIMD share the pipe and therefore leave their loops together; if P2 of the next tile could only start
// after the A team's whole epilogue (and the extra row), nobody would issue an MFMA through all of it.  Hence P2 in
// phases: the first 4*NA k-blocks need only the ratio rows of the row tiles 0..NA-1, the rest every row tile, and only
// the very last k-block the extra row, which the A team computes after everything else.
// No workgroup barrier inside the tile loop: six signals, each four per-wave progress words in LDS, order the roles
//     ready  (loaders)   "the H block of tile j is staged"           A waits (loop), and the loaders' extra-row pass
//     vready (loaders)   "the V block of tile j is staged"           A waits (epilogues), and the extra-row pass
//     p1a    (A team)    "ratio rows of the row tiles 0..NA-1 whole"  B waits (phase 1)
//     p1b    (A team)    "every ratio row tile is whole"              B waits (phase 2)
//     xdone  (A team)    "the extra row of the ratio image is done"   B waits (last k-block)
//     p2done (B team)    "H_j is updated, its ratio dead"             loaders wait
// (dependencies run strictly forward in the tile index, so the waits cannot form a cycle; every wait is a bounded spin
// that raises DevState::fault instead of hanging).  Each wave owns TWO 32-row (A) / 32-column (B) output tiles whose
// MFMA chains share every LDS fragment: one ds_read_b128 feeds 8 MFMAs instead of 4, and two independent accumulator
// chains alternate in the pipe.  The loaders fetch tile i+2 into REGISTERS before they wait for p2done(i), so the
// reload that sits between P2(i) and P1(i+2) is an LDS write, not an HBM round trip.
// The loaders are the part to keep SHORT: beside two MFMA waves a loader wave gets an instruction in only where they
// stall, so the time from p2done(i) to ready(i+2) -- a dependency cycle of two tile periods runs through it -- is the
// number of instructions in between times tens of cycles.  With ~390 of them (index arithmetic per cell, 64-bit
// addresses) the kernel took 0.2595 ms on C2; with ~60 (buffer instructions with scalar offsets, a row mapping for the H
// block, one precomputed LDS offset per V cell) 0.246 ms.
// Fences are LDS-only ("local"): LDS operations of a wave complete in order, and a workgroup-wide release that also
// drained vmcnt would make the loaders wait for their H stores to be acknowledged by HBM.
// ============================================================================================
// Progress slots: every producer wave of a role owns ONE word per signal and writes the number of tiles it has finished;
// a consumer waits until all four words of the role have reached its tile.  (A single counter incremented by all four
// waves -- what k_hstep's `sig` and k_wstats' ready / done do -- is a TOTAL: two arrivals of a fast wave can stand in for
// the missing arrival of a slow one whenever a wave may run a tile ahead of its team.  The loaders stage tiles 0 and 1
// back to back, so with a total `ready` the A team, or the loaders' own extra-row pass, could start on a tile that one
// loader wave had not finished staging; a probe over awkward shapes found it.)
// -DSNMF_STRESS (scripts/build_variant.py stress -DSNMF_STRESS; tests/test_gpu_fuzz.py and the pipelined-vs-plain lists run on it:
// profiles/r06_stress.log): a pseudo-random delay at EVERY hand-off of the role pipelines -- in front of each progress post (the
// signal arrives late) and behind each satisfied wait (the consumer starts late), at the split tiles' arrival counter, the fused
// small-F iteration's pair hand-off and the adaptation kernel's grid exchange.  The unit tests run with deterministic timing; a
// protocol that is only right for the timing the kernels happen to have (round 5's DMA refill under another wave's copy-out) shows
// when one hand-off in four is late by up to 64 x 64 cycles -- a fifth of a tile period.  The shipped kernels contain none of this.
// (stress_jitter itself: near the top of this file, with the objective fold that also uses it)

__device__ __forceinline__ void rp_post(unsigned* slots, int wave_in_role, unsigned tiles_done, int lane) {
    stress_jitter();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(slots + wave_in_role, tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void rp_await(const unsigned* slots, unsigned target, const int* stop) {
    int spin = 0;
    for (;;) {
        const unsigned a0 = __hip_atomic_load(slots + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned a1 = __hip_atomic_load(slots + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned a2 = __hip_atomic_load(slots + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned a3 = __hip_atomic_load(slots + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned lo01 = a0 < a1 ? a0 : a1, lo23 = a2 < a3 ? a2 : a3;
        if ((lo01 < lo23 ? lo01 : lo23) >= target) break;
        if (++spin > kSpinLimit) {
            raise_fault(stop);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    stress_jitter();
}

// acc[i] += sum_q Wfrag_i(q) (x) Sfrag(q), i < NA: NA output tiles that share the LDS operand stream (one ds_read_b128
// feeds 4 * NA MFMAs).  Three named stages, each one k-block, fragments two blocks ahead of their use (a wave that has
// the pipe to itself still hides the L2 latency).
// gate(): called between the W loads of the first two blocks and their LDS loads.  W does not depend on the tile, so
// the caller puts its WAIT for the LDS image there: the L2 round trip of the loop's first fragments (~0.8 k cycles per
// call, measured with one team's loops removed: a lone A loop ran at 77 cycles per MFMA, a lone two-phase B loop at 86)
// is then spent while the wave waits anyway.
// W fragments through the buffer path (see contract_p3_buf): soff[i] = byte offset of image block i (scalar).  LDS fragments at immediate offsets from a base that moves once per three k-blocks; no clamps: the
// prefetch reads up to two k-blocks (64 B of the LDS row, 2 KB of the image) past the last one.
template <int NA, typename Gate>
__device__ __forceinline__ void contract_shared_buf(f32x16 (&acc)[NA], __amdgpu_buffer_rsrc_t rs, int voff, const int (&soff)[NA],
                                                    const float* sp, int nq, Gate gate) {
    f32x4 wA[NA], wB[NA], wC[NA], sA, sB, sC;
    auto ldw = [&](f32x4 (&w)[NA], int q) {
#pragma unroll
        for (int i = 0; i < NA; ++i) w[i] = ldw_buf(rs, voff, soff[i] + q * 1024);
    };
    auto mm = [&](const f32x4 (&w)[NA], const f32x4& sf) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = mfma32(w[i][e], sf[e], acc[i]);
    };
    ldw(wA, 0);
    ldw(wB, 1);
    SNMF_PIN();
    gate();
    const float* bp = sp;  // moving base: block q + j at bp + 8 * j
    sA = *reinterpret_cast<const f32x4*>(bp);
    sB = *reinterpret_cast<const f32x4*>(bp + 8);
    int q = 0;
    for (; q + 2 < nq; q += 3) {
        ldw(wC, q + 2);
        sC = *reinterpret_cast<const f32x4*>(bp + 16);
        SNMF_PIN();
        mm(wA, sA);
        ldw(wA, q + 3);
        sA = *reinterpret_cast<const f32x4*>(bp + 24);
        SNMF_PIN();
        mm(wB, sB);
        ldw(wB, q + 4);
        sB = *reinterpret_cast<const f32x4*>(bp + 32);
        bp += 24;
        SNMF_PIN();
        mm(wC, sC);
    }
    if (q < nq) mm(wA, sA);
    if (q + 1 < nq) mm(wB, sB);
}

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction.  Block b = lane / 4: A[i] is the a operand
// of lane 4b + i, B[j] the b operand of lane 4b + j, D[i][j] lands in register i of lane 4b + j (scripts/mfma_4x4x1_probe.hip
// checks this on the device; 15 cycles an instruction there, i.e. half the 32x32x2 shape's MAC rate -- but none of it padding).
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// contract_shared_buf + the LEFTOVER COLUMNS of the numerator: r = 100 is three full 32-column tiles + 4 columns, r = 200 six + 8
// (the reference's own ranks, settings/initial_setting_SNMF_NAT.m:48-49); a fourth / seventh 32-column MFMA tile would be 87 % /
// 75 % padding.  One group of four leftover columns rides along with the full tiles' contraction instead: the SAME ratio fragment
// sf[e] (lane (t = fl, h): ratio[8q + 4h + e][t]) is the B operand of a 4x4x1 MFMA whose A operand is W[8q + 4h + e][c0 + (lane & 3)]
// -- a 16-byte piece of the ordinary Wk4 image (column tile c0 / 32, column c0 % 32 + (lane & 3)), so no table of its own --
// and gl[i] accumulates column c0 + i for frame fl over this lane half's rows: per k-block one more 16-byte load and four
// short MFMAs.  (Round 3 did these columns on the VALU from a small LDS copy of W: 1.6 k cycles per tile and SIMD at r = 100.)
//   voff_l / soff_l: this lane's byte offset into the leftover columns' image block ((h * 128 + (c0 % 32 + (lane & 3)) * 4) * 4)
//   and the block's scalar byte offset ((c0 / 32) * Fq * 128 + first k-block * 1024).
template <int NA, typename Gate>
__device__ __forceinline__ void contract_shared_buf_lx(f32x16 (&acc)[NA], f32x4& gl, __amdgpu_buffer_rsrc_t rs, int voff, const int (&soff)[NA],
                                                       int voff_l, int soff_l, const float* sp, int nq, Gate gate) {
    f32x4 wA[NA], wB[NA], wC[NA], lA, lB, lC, sA, sB, sC;
    auto ldw = [&](f32x4 (&w)[NA], f32x4& l, int q) {
#pragma unroll
        for (int i = 0; i < NA; ++i) w[i] = ldw_buf(rs, voff, soff[i] + q * 1024);
        l = ldw_buf(rs, voff_l, soff_l + q * 1024);
    };
    auto mm = [&](const f32x4 (&w)[NA], const f32x4& l, const f32x4& sf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = mfma32(w[i][e], sf[e], acc[i]);
            gl = mfma4(l[e], sf[e], gl);
        }
    };
    ldw(wA, lA, 0);
    ldw(wB, lB, 1);
    SNMF_PIN();
    gate();
    const float* bp = sp;  // moving base: block q + j at bp + 8 * j
    sA = *reinterpret_cast<const f32x4*>(bp);
    sB = *reinterpret_cast<const f32x4*>(bp + 8);
    int q = 0;
    for (; q + 2 < nq; q += 3) {
        ldw(wC, lC, q + 2);
        sC = *reinterpret_cast<const f32x4*>(bp + 16);
        SNMF_PIN();
        mm(wA, lA, sA);
        ldw(wA, lA, q + 3);
        sA = *reinterpret_cast<const f32x4*>(bp + 24);
        SNMF_PIN();
        mm(wB, lB, sB);
        ldw(wB, lB, q + 4);
        sB = *reinterpret_cast<const f32x4*>(bp + 32);
        bp += 24;
        SNMF_PIN();
        mm(wC, lC, sC);
    }
    if (q < nq) mm(wA, lA, sA);
    if (q + 1 < nq) mm(wB, lB, sB);
}

// P1 epilogue of one 32-row tile: Lam -> ratio in place over the staged V (+ the divergence terms of the previous iterate).
// The per-element bounds masks of the objective are only evaluated for the tiles that need them: a wave-uniform test
// picks the unmasked loop for interior tiles (a quarter of the epilogue's VALU instructions; no measurable effect on
// the kernel -- a neighbour wave's VALU, LDS or load traffic costs an MFMA-issuing wave nothing, scripts/mfma_valu_overlap.hip).
template <bool OBJ, bool MASKED>
__device__ __forceinline__ void rp_p1_epilogue_t(const StepArgs& a, const f32x16& acc, float* Rs, int phi, int t0, int lane,
                                                 float& dsum) {
    const int fl = lane & 31, h = lane >> 5;
    const int t = t0 + fl;
    float* rsp = Rs + fl * a.ldr + phi * 32 + 4 * h;
    f32x4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(rsp + 8 * g);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lam = fmaxf(acc[4 * g + j], kFlr);
            if (OBJ) {
                const float d = div_term<BM_KL>(v[g][j], lam, a.beta, a.inv_bb1);
                if (MASKED) {
                    const int f = phi * 32 + 8 * g + 4 * h + j;
                    dsum += (f < a.F && t < a.T) ? d : 0.f;
                } else {
                    dsum += d;
                }
            }
            o[j] = v[g][j] * fast_rcp(lam);
        }
        *reinterpret_cast<f32x4*>(rsp + 8 * g) = o;
    }
}
template <bool OBJ>
__device__ __forceinline__ void rp_p1_epilogue(const StepArgs& a, const f32x16& acc, float* Rs, int phi, int t0, int lane,
                                               float& dsum) {
    if (!OBJ || (phi * 32 + 32 <= a.F && t0 + 32 <= a.T)) rp_p1_epilogue_t<OBJ, false>(a, acc, Rs, phi, t0, lane, dsum);
    else rp_p1_epilogue_t<OBJ, true>(a, acc, Rs, phi, t0, lane, dsum);
}

// P2 epilogue of one 32-column tile of W^T*ratio: H <- H .* dmh ./ dph in LDS (src/sparse_nmf.m:192-195).
// dpf: 1 ./ max(colsum(W)+lambda, flr) of this lane's 16 columns (scalar / r-vector sparsity) -- a constant of the
// launch, formed once per wave by rp_p2_consts; with a full r x T sparsity matrix dph is formed here from S.
template <bool OBJ>
__device__ __forceinline__ void rp_p2_epilogue(const StepArgs& a, const f32x16& acc, float* Hs, int kap, int t0, int lane,
                                               const f32x4 (&dpf)[4], float& shsum) {
    const int fl = lane & 31, h = lane >> 5;
    const int t = t0 + fl;
    float* hsp = Hs + fl * a.ldh + kap * 32 + 4 * h;
    f32x4 spv[4];  // lambda of this lane's 16 columns: only the objective needs it, so it is fetched here (L2) and used last
    if ((OBJ && !a.lam_is_u) || a.S) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = kap * 32 + 8 * g + 4 * h;
            spv[g] = a.S ? *reinterpret_cast<const f32x4*>(a.S + (size_t)t * a.rp + k0)
                         : *reinterpret_cast<const f32x4*>(a.lamk + k0);
        }
    }
    f32x4 hov[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int k0 = kap * 32 + 8 * g + 4 * h;
        hov[g] = *reinterpret_cast<f32x4*>(hsp + 8 * g);
        f32x4 dp;
        if (a.S) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dp[j] = fmaxf(cs[j] + spv[g][j], kFlr);
        } else {
            dp = dpf[g];
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = hov[g][j] * acc[4 * g + j] * (a.S ? fast_rcp(dp[j]) : dp[j]);  // dpf holds 1 ./ dph
        *reinterpret_cast<f32x4*>(hsp + 8 * g) = o;
    }
    if (OBJ) {
        if (a.lam_is_u && !a.S) {  // scalar sparsity: sum(S .* H) = lambda * sum(H), one multiply per tile
            float hs = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) hs += hov[g][j];
            shsum += a.lam_u * hs;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) shsum += spv[g][j] * hov[g][j];
        }
    }
}
// The same for register groups [GB, GE) of a tile only (columns kap*32 + 8g + 4h + j, g in [GB, GE)): the CUT2 mode of k_hstep_rh
// hands a wave one whole tile and HALF of another.  rdph: 1 ./ max(colsum(W) + lambda, flr) for all rp columns in LDS (scalar /
// r-vector sparsity; the waves of that mode finish parts of three different tiles, too many constants to keep in registers).
template <bool OBJ, int GB, int GE>
__device__ __forceinline__ void rh_cut_epilogue(const StepArgs& a, const f32x16& acc, float* Hs, int kap, int t0, int lane,
                                                const float* rdph, float& shsum) {
    const int fl = lane & 31, h = lane >> 5;
    const int t = t0 + fl;
    float* hsp = Hs + fl * a.ldh + kap * 32 + 4 * h;
    float hs = 0.f;
#pragma unroll
    for (int g = GB; g < GE; ++g) {
        const int k0 = kap * 32 + 8 * g + 4 * h;
        const f32x4 hov = *reinterpret_cast<f32x4*>(hsp + 8 * g);
        f32x4 dp, spv = {0.f, 0.f, 0.f, 0.f};
        if (a.S) {
            const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
            spv = *reinterpret_cast<const f32x4*>(a.S + (size_t)t * a.rp + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dp[j] = fast_rcp(fmaxf(cs[j] + spv[j], kFlr));
        } else {
            dp = *reinterpret_cast<const f32x4*>(rdph + k0);
            if (OBJ && !a.lam_is_u) spv = *reinterpret_cast<const f32x4*>(a.lamk + k0);
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = hov[j] * acc[4 * g + j] * dp[j];
        *reinterpret_cast<f32x4*>(hsp + 8 * g) = o;
        if (OBJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (a.lam_is_u && !a.S) hs += hov[j];
                else shsum += spv[j] * hov[j];
            }
        }
    }
    if (OBJ && a.lam_is_u && !a.S) shsum += a.lam_u * hs;
}
// ... and for ONE register group g chosen at run time (wave-uniform), the group's four numerators in `num`: k_hstep_rp<., CUT>, where
// wave w finishes group w of every column tile.
template <bool OBJ>
__device__ __forceinline__ void rp_cut_group_epilogue(const StepArgs& a, const f32x4& num, float* Hs, int kap, int g, int t0, int lane,
                                                      const float* rdph, float& shsum) {
    const int fl = lane & 31, h = lane >> 5;
    const int t = t0 + fl, k0 = kap * 32 + 8 * g + 4 * h;
    float* hsp = Hs + fl * a.ldh + k0;
    const f32x4 hov = *reinterpret_cast<f32x4*>(hsp);
    f32x4 dp, spv = {0.f, 0.f, 0.f, 0.f};
    if (a.S) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
        spv = *reinterpret_cast<const f32x4*>(a.S + (size_t)t * a.rp + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[j] = fast_rcp(fmaxf(cs[j] + spv[j], kFlr));
    } else {
        dp = *reinterpret_cast<const f32x4*>(rdph + k0);
        if (OBJ && !a.lam_is_u) spv = *reinterpret_cast<const f32x4*>(a.lamk + k0);
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = hov[j] * num[j] * dp[j];
    *reinterpret_cast<f32x4*>(hsp) = o;
    if (OBJ) {
        if (a.lam_is_u && !a.S) shsum += a.lam_u * (((hov[0] + hov[1]) + hov[2]) + hov[3]);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j) shsum += spv[j] * hov[j];
    }
}
__device__ __forceinline__ void rp_p2_consts(const StepArgs& a, int kap, int lane, f32x4 (&dpf)[4]) {
    const int h = lane >> 5;
    if (!a.S) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.dphv + kap * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) dpf[g][j] = fast_rcp(d[j]);
        }
    }
}

// ---- the split last round of k_hstep_rp ------------------------------------------------------------------------------
// n_tiles tiles on G workgroups are ceil(n_tiles / G) rounds of the pipeline, and in the last one only n_tiles mod G
// workgroups have a tile (C2: 3125 tiles on 256 -> 53 workgroups work through a 13th tile period while 203 CUs idle,
// 6-7 % of the launch).  A tile can be cut by ROWS without duplicating any work: P1 needs only a workgroup's own rows of
// W and V (and the whole H tile), and P2 is linear in the ratio rows, so part p of S forms the ratio rows of its own
// 32-row tiles phi = p, p + S, ... and contracts W^T*ratio over exactly those rows -- a PARTIAL numerator [rp x 32].
// The tiles of the last partial round are dealt out S parts each over the workgroups that would otherwise idle, and a
// workgroup's part is one more item of its list, staged by the loaders like any tile, in the SECOND TO LAST place:
//   A team: P1 of the part's row tiles.  A part with at most two row tiles (F = 257, S = 4) is ALSO cut in two over k so
//       that all four A waves have an item; the two partial Lam tiles of a row tile meet in the ratio image's cells of
//       row tiles the part does not own;
//   B team: W^T*ratio over the part's k-blocks (4 per own row tile, + the extra row's for the last part) -> partial
//       numerator to HBM in MFMA fragment order, through agent-scope (sc1, write-through) stores;
//   loaders: when every B wave has seen its stores acknowledged, bump the tile's agent-scope arrival counter (all of this
//       runs under the workgroup's last whole tile);
//   at the end of the kernel the LAST of a tile's S workgroups to have arrived (nobody waits for anybody) adds the S
//       partials in part order -- sc1 loads, so no stale L2 line of its XCD can answer -- and applies the H update from
//       the H block it still has in LDS, exactly as rp_p2_epilogue forms it (+ the tile's share of sum(S .* H)).
// Summation order of a split tile: Lam possibly in two k ranges, the numerator in S row parts -- fp32, a few ulp from
// the one-workgroup order of the pipelined tiles (tests/test_gpu_pipelined_vs_plain.py states the tolerance).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kAuxSC1 = 16;  // cache-policy bit of the raw buffer builtins: sc1 = agent scope (coherent across the XCDs' L2s)
constexpr int kAuxNT = 2;    // ... nt = non-temporal (streaming): the line is not kept in the L2 for re-use
// 16-byte buffer store.  A buffer_store_dwordx4 with a REGISTER scalar offset still reads its data registers after it
// has issued (found by scripts/rp_shape_probe.py in round 2: the compiler reused one for an LDS address in the very next
// instruction and the address reached memory; LLVM's hazard recogniser pads only immediate offsets): the s_nop carries
// the data registers as operands, so nothing can be scheduled between the store and the padding that rewrites them.
template <int AUX = 0>
__device__ __forceinline__ void buf_store_b128(__amdgpu_buffer_rsrc_t rs, int voff, int soff, f32x4 x) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), rs, voff, soff, AUX);
    asm volatile("s_nop 2" : "+v"(x) : : "memory");
}
// scratch cells of item `it` of a k-split part (S >= 4, nf = nrt * S): the it-th row tile the part does not own
__device__ __forceinline__ int part_scratch_tile(int it, int pp, int S) {
    const int q = it / (S - 1), rr = it - q * (S - 1);
    return q * S + (pp + 1 + rr) % S;
}

struct PartGeo {
    int S, pp, nrt;  // parts per tile, this workgroup's part, its row tiles phi = pp + i * S
    bool ks2;        // P1 also cut in two over k (parts of at most two row tiles)
    bool own_x;      // the last part owns the extra row
};
__device__ __forceinline__ PartGeo part_geo(const StepArgs& a) {
    PartGeo g;
    g.S = a.part_S;
    g.pp = (int)blockIdx.x % g.S;
    g.nrt = g.pp < a.nf ? (a.nf - 1 - g.pp) / g.S + 1 : 0;
    g.ks2 = g.nrt <= 2 && g.S >= 4 && a.nf == g.nrt * g.S;
    g.own_x = a.xr && g.pp == g.S - 1;
    return g;
}
// The steps of a split tile are NOT a branch inside the tile loops of their role: the loops sit at the 168-VGPR limit of
// three waves per SIMD, and with the part's code in the loop body the pipelined tiles got 5 % slower (22 spilled VGPRs);
// as real (not inlined) functions they need scratch for their arguments and are slower still.  The loops are peeled
// instead: the part and the workgroup's last whole tile are straight-line code behind them.
// A team, wave w: P1 of the part's row tiles; returns the divergence terms (OBJ).  cnt: the kernel's progress slots.
template <bool OBJ>
__device__ __forceinline__ double rp_part_p1(const StepArgs& a, float* Hs, const float* wxs, unsigned* cnt, int j, int ptile, int w, int lane) {
    constexpr int NA = 4;
    const PartGeo pg = part_geo(a);
    const int rp = a.rp, ldh = a.ldh, ldr = a.ldr, pS = pg.S, pp = pg.pp, pnrt = pg.nrt;
    unsigned *ready = cnt, *p1a = cnt + 4, *p1b = cnt + 8, *xdone = cnt + 16, *vready = cnt + 20;
    float* Rs = Hs + 32 * ldh;
    const int t0 = ptile * 32, fl = lane & 31, h = lane >> 5;
    const float* sp = Hs + fl * ldh + 4 * h;
    bool waited = false, vwaited = false;
    auto gate_ready = [&]() {
        if (!waited) rp_await(ready, (unsigned)(j + 1), a.stop);
        waited = true;
    };
    auto gate_v = [&]() {
        if (!vwaited) rp_await(vready, (unsigned)(j + 1), a.stop);
        vwaited = true;
    };
    float dsum = 0.f;
    double acc_div = 0.0;
    const __amdgpu_buffer_rsrc_t rsw = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);
    if (!pg.ks2) {
        for (int i = w; i < pnrt; i += NA) {
            const int phi = pp + i * pS;
            f32x16 acc[1] = {zero16()};
            const int so[1] = {phi * rp * 128};
            contract_shared_buf<1>(acc, rsw, lane * 16, so, sp, a.nqk, gate_ready);
            gate_v();
            rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
        }
        gate_ready();
        gate_v();
        rp_post(p1a, w, (unsigned)(j + 1), lane);
    } else {
        // item w = (row tile w >> 1, k half w & 1): partial Lam -> the cells of a row tile the part does not own
        const int it_i = w >> 1, it_k = w & 1;
        const int nq0 = (a.nqk + 1) / 2, qb = it_k ? nq0 : 0, nqs = it_k ? a.nqk - nq0 : nq0;  // this item's k-blocks [qb, qb + nqs)
        if (w < 2 * pnrt) {
            f32x16 acc[1] = {zero16()};
            const int so[1] = {(pp + it_i * pS) * rp * 128 + qb * 1024};
            contract_shared_buf<1>(acc, rsw, lane * 16, so, sp + 8 * qb, nqs, gate_ready);
            gate_v();  // (the V commit of the loaders must not land on top of the scratch cells)
            float* d = Rs + fl * ldr + part_scratch_tile(w, pp, pS) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]};
                *reinterpret_cast<f32x4*>(d + 8 * g) = o;
            }
        }
        gate_ready();
        gate_v();
        rp_post(p1a, w, (unsigned)(j + 1), lane);  // "my partial Lam tile is in its cells"
        rp_await(p1a, (unsigned)(j + 1), a.stop);
        if (w < 2 * pnrt) {  // wave w: rows 16 (w & 1) .. + 15 of row tile w >> 1: Lam = the two k halves, in order
            const int phi = pp + it_i * pS;
            const float* s0 = Rs + fl * ldr + part_scratch_tile(2 * it_i, pp, pS) * 32 + 4 * h;
            const float* s1 = Rs + fl * ldr + part_scratch_tile(2 * it_i + 1, pp, pS) * 32 + 4 * h;
            float* rsp = Rs + fl * ldr + phi * 32 + 4 * h;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * it_k + gg;
                const f32x4 l0 = *reinterpret_cast<const f32x4*>(s0 + 8 * g), l1 = *reinterpret_cast<const f32x4*>(s1 + 8 * g);
                const f32x4 v = *reinterpret_cast<const f32x4*>(rsp + 8 * g);
                f32x4 o;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float lam = fmaxf(l0[jj] + l1[jj], kFlr);
                    if (OBJ) {
                        const int f = phi * 32 + 8 * g + 4 * h + jj;
                        const float dv = div_term<BM_KL>(v[jj], lam, a.beta, a.inv_bb1);
                        dsum += (f < a.F && t0 + fl < a.T) ? dv : 0.f;
                    }
                    o[jj] = v[jj] * fast_rcp(lam);
                }
                *reinterpret_cast<f32x4*>(rsp + 8 * g) = o;
            }
        }
    }
    if (OBJ) acc_div += (double)dsum;
    rp_post(p1b, w, (unsigned)(j + 1), lane);
    if (pg.own_x) {
        hstep_p1_xrow<NA, 1, BM_KL, OBJ>(a, Hs, Rs, wxs, t0, w, lane, true, acc_div);
        rp_post(xdone, w, (unsigned)(j + 1), lane);
    }
    return acc_div;
}
// B team, wave wb: W^T*ratio over the part's own ratio rows -> partial numerator (fragment order [kap][g][lane]).  The
// part's k-blocks are scattered over the contraction axis (4 per own row tile, + the extra row's): block i of the list
// sits at blk(i); a ring of four fragment sets keeps three blocks' loads in flight.
__device__ __forceinline__ void rp_part_p2(const StepArgs& a, const float* Rs, unsigned* cnt, int j, int wb, int lane) {
    constexpr int NB = 4;
    const PartGeo pg = part_geo(a);
    const int rp = a.rp, ldr = a.ldr, pS = pg.S, pp = pg.pp, pnrt = pg.nrt;
    const int fl = lane & 31, h = lane >> 5;
    const float* sp = Rs + fl * ldr + 4 * h;
    rp_await(cnt + 8, (unsigned)(j + 1), a.stop);                  // p1b
    if (pg.own_x) rp_await(cnt + 16, (unsigned)(j + 1), a.stop);  // xdone
    const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
    const __amdgpu_buffer_rsrc_t rsp_ = __builtin_amdgcn_make_buffer_rsrc(a.part_buf + (size_t)blockIdx.x * 32 * rp, 0, 32 * rp * 4, 0x00020000);
    const int nb = 4 * pnrt + (pg.own_x ? 1 : 0);
    auto blk = [&](int i) { return i < 4 * pnrt ? 4 * (pp + (i >> 2) * pS) + (i & 3) : 4 * a.nf; };
    for (int kap = wb; kap < a.nk; kap += 2 * NB) {
        const bool two = kap + NB < a.nk;
        const int kap1 = two ? kap + NB : kap;
        f32x16 acc[2] = {zero16(), zero16()};
        f32x4 wq0[4], wq1[4], sq[4];
        auto ldq = [&](int u, int i) {
            const int b = blk(i);
            wq0[u] = ldw_buf(rsk, lane * 16, kap * a.Fq * 128 + b * 1024);
            wq1[u] = ldw_buf(rsk, lane * 16, kap1 * a.Fq * 128 + b * 1024);
            sq[u] = *reinterpret_cast<const f32x4*>(sp + 8 * b);
        };
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < nb) ldq(u, u);
        SNMF_PIN();
        for (int i0 = 0; i0 < nb; i0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u < nb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = mfma32(wq0[u][e], sq[u][e], acc[0]);
                        acc[1] = mfma32(wq1[u][e], sq[u][e], acc[1]);
                    }
                }
                if (i0 + u + 4 < nb) ldq(u, i0 + u + 4);
                SNMF_PIN();
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == 1 && !two) break;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
                buf_store_b128<kAuxSC1>(rsp_, lane * 16, ((c ? kap1 : kap) * 1024 + g * 256) * 4, o);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial is at the coherence point before this wave reports
    rp_post(cnt + 12, wb, (unsigned)(j + 1), lane);     // p2done of the part's place in the list: the loaders bump the arrival counter
}
// All waves, at the very end of the kernel: the workgroup that bumped the tile's counter to a multiple of S (loader wave 0,
// right after the part's P2, a whole tile period ago) was the last of the tile's S workgroups and finishes the tile:
// numerator = the S partials in part order, then H <- H .* dmh ./ dph as rp_p2_epilogue forms it; returns the tile's
// share of sum(S .* H) of the previous iterate (OBJ).  Hs: the split tile's H block, staged for P1.
template <bool OBJ>
__device__ __forceinline__ double rp_part_finish(const StepArgs& a, const float* Hs, const unsigned* plast, int ptile) {
    constexpr int NTHR = 768, Tt = 32;
    const PartGeo pg = part_geo(a);
    const int rp = a.rp, ldh = a.ldh, pS = pg.S;
    __syncthreads();  // every role is through its list; loader wave 0 wrote *plast when it bumped the tile's arrival counter
    if (!*plast) return 0.0;
    const int u0 = (int)blockIdx.x - pg.pp;  // first of the tile's S units
    const __amdgpu_buffer_rsrc_t rsp_ = __builtin_amdgcn_make_buffer_rsrc(a.part_buf + (size_t)u0 * Tt * rp, 0, pS * Tt * rp * 4, 0x00020000);
    float shsum = 0.f;
    // The partials come from memory (sc1: ~2 us a round trip): all loads of a batch of three elements per thread are issued
    // before the first is used (one element at a time it was three dependent round trips for rp = 256: the most
    // expensive part of the whole split)
    constexpr int NB3 = 3;
    for (int idx0 = threadIdx.x; idx0 < a.nk * 256; idx0 += NB3 * NTHR) {
        f32x4 x[NB3][4];  // S <= 4 here (8-way splits do not exist yet)
#pragma unroll
        for (int b = 0; b < NB3; ++b) {
            const int idx = idx0 + b * NTHR;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < pS && idx < a.nk * 256)
                    x[b][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsp_, idx * 16, p * Tt * rp * 4, kAuxSC1));
        }
#pragma unroll
        for (int b = 0; b < NB3; ++b) {
            const int idx = idx0 + b * NTHR;
            if (idx >= a.nk * 256) break;
            const int kap = idx >> 8, g = (idx >> 6) & 3, ln = idx & 63, fl = ln & 31, h = ln >> 5;
            const int t = ptile * Tt + fl, k0 = kap * 32 + 8 * g + 4 * h;
            const f32x4 ho = *reinterpret_cast<const f32x4*>(Hs + fl * ldh + k0);
            f32x4 spv = {0.f, 0.f, 0.f, 0.f}, o, den;
            if (a.S) {
                spv = *reinterpret_cast<const f32x4*>(a.S + (size_t)t * rp + k0);
                const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) den[jj] = fmaxf(cs[jj] + spv[jj], kFlr);
            } else {
                if (OBJ && !a.lam_is_u) spv = *reinterpret_cast<const f32x4*>(a.lamk + k0);
                den = *reinterpret_cast<const f32x4*>(a.dphv + k0);
            }
            f32x4 num = x[b][0];
#pragma unroll
            for (int p = 1; p < 4; ++p)
                if (p < pS) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) num[jj] += x[b][p][jj];
                }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) o[jj] = ho[jj] * num[jj] * fast_rcp(den[jj]);
            *reinterpret_cast<f32x4*>(a.Hout + (size_t)t * rp + k0) = o;
            if (OBJ) {
                if (a.lam_is_u && !a.S) shsum += a.lam_u * ((ho[0] + ho[1]) + (ho[2] + ho[3]));
                else shsum += (spv[0] * ho[0] + spv[1] * ho[1]) + (spv[2] * ho[2] + spv[3] * ho[3]);
            }
        }
    }
    return (double)shsum;
}

// CUT (round 5; r <= 64: one or two column tiles -- settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48 R_x = 20,
// R_d = 10, initial_setting_IMCRA.m:47-48 R = 50): P2's column tiles were ONE B wave's work (nk = 1: 4 (nf + ..) k-blocks x 4 MFMAs
// on one SIMD -- at F = 513 260 MFMAs beside 64 of P1 -- while three B waves idled).  Here every B wave contracts a quarter of the
// k-blocks for ALL column tiles, the partial tiles meet in LDS (k_hstep_rh<., 1>'s exchange: wdone / rdone) and wave w finishes
// register group w (rows 8 w .. 8 w + 7 of each 32-column tile) summed in wave order, so both orders of arrival give the same
// bits.  A template parameter: the instantiation the headline runs (CUT = false) is the code it was.
template <bool OBJ, bool CUT = false>
__global__ __launch_bounds__(768, 3) void k_hstep_rp(StepArgs a) {
    constexpr int NA = 4, NB = 4, NL = 4, NTHR = (NA + NB + NL) * 64, NLT = NL * 64, Tt = 32;
    constexpr int PA = 10, PB = 10;  // f32x4 per loader thread of the H / V block held in registers (as stage_in2)
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp, ldh = a.ldh, ldr = a.ldr;
    const int bufsz = Tt * (ldh + ldr);  // floats per buffer: Hs [Tt][ldh] then Rs [Tt][ldr]
    float* wxs = lds + 2 * bufsz;        // [rp] extra row of W
    unsigned* cnt = reinterpret_cast<unsigned*>(wxs + rp);
    // p1a: "the ratio rows of the row tiles 0..NA-1 are whole" (each A wave after the epilogue of its FIRST row tile);
    // p1b: "every row tile is" (each A wave after its last epilogue); xdone: "the extra row is" (each A wave after its
    // share of it).  The B team starts P2 on the first 4*NA k-blocks (rows
    // 0..32*NA-1) at p1a, needs p1b for the rest and xdone for the extra row's k-block only: in steady state both teams
    // leave their MFMA loops together (they share the pipe), and the A team's epilogue is then the one stretch with nobody
    // in a loop -- B waits for half of it only.  Every signal is four per-wave progress words (rp_post / rp_await), so
    // each has a single kind of producer and a wave that runs ahead cannot stand in for one that lags.
    static_assert(NA == 4 && NB == 4 && NL == 4, "four progress slots per role");
    // vready: "the tile's V block is staged" -- a signal of its own because the A team needs V only in its epilogues: the
    // loaders commit the H block first and post `ready`, so the V commit is off the path the A team's next loop waits for.
    unsigned *ready = cnt, *p1a = cnt + 4, *p1b = cnt + 8, *p2done = cnt + 12, *xdone = cnt + 16, *vready = cnt + 20;
    unsigned *wdone = cnt + 28, *rdone = cnt + 32;               // CUT: partial tiles written / read (B team internal)
    float* const Ps = reinterpret_cast<float*>(cnt + 40);        // CUT: [4 waves][nk <= 2 tiles][4 g][64 lanes][4] partial numerators
    const bool cut_pair = CUT && a.lxh == 2;                     // CUT, two column tiles, 16 row tiles (no room for 32 KB of partials): the pair form below
    float* const rdph = Ps + (cut_pair ? 2048 : 4 * a.nk * 1024);  // CUT: [rp] 1 ./ dph (rp_cut_group_epilogue)
    double acc_div = 0.0, acc_sh = 0.0;
    if (CUT && !a.S)
        for (int k = threadIdx.x; k < rp; k += NTHR) rdph[k] = fast_rcp(a.dphv[k]);
    if (a.xr) {
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
        // the 7 unused cells of the extra 8-deep k-block stay zero for the whole kernel
        for (int i = threadIdx.x; i < 2 * Tt * 8; i += NTHR) {
            const int bsel = i / (Tt * 8), ii = i - bsel * Tt * 8;
            lds[bsel * bufsz + Tt * ldh + (ii >> 3) * ldr + a.Fm + (ii & 7)] = 0.f;
        }
    }
    if (threadIdx.x < (CUT ? 36 : 25)) cnt[threadIdx.x] = 0u;
    __syncthreads();
    // A wave WITHOUT work -- an A wave past the last row tile, a B wave without a column tile -- used to walk the tile list
    // anyway, waiting for and posting every signal: a wave that polls LDS takes issue slots from the MFMA wave of its SIMD
    // (F = 64: two of four A waves, r <= 32: three of four B waves).  Where nothing else needs it (no split last round, whose
    // parts are dealt to all four waves of a team; no extra row, which the A waves share) its progress words are set to
    // "done with every tile" once and it waits at the final barrier instead.
    bool idle = false;
    if (a.part_S == 0) {
        const bool bs = a.nf <= 2 && a.nk >= 2;  // (b_shift: see the B team)
        const bool idle_a = w < NA && w >= a.nf && !a.xr;
        const bool idle_b = !CUT && w >= NA && w < NA + NB && (bs ? w - NA < 2 : w - NA >= a.nk);
        if (lane == 0) {
            if (idle_a) p1a[w] = p1b[w] = xdone[w] = 0xffffffffu;
            if (idle_b) p2done[w - NA] = 0xffffffffu;
        }
        idle = idle_a || idle_b;
    }
    __syncthreads();
    // Tiles [0, n_full) go through the pipeline, dealt out round-robin.  The tiles of the last PARTIAL round,
    // [n_full, n_tiles), are split over part_S workgroups each: workgroup u < (n_tiles - n_full) * part_S
    // takes part u % part_S of tile n_full + u / part_S as one more staged tile AFTER its pipelined ones.
    const int n_full = (!CUT && a.part_S > 0) ? a.n_full : a.n_tiles;
    const int nmy = (int)blockIdx.x < n_full ? (n_full - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    // (CUT launches keep every tile whole: host rp_S = 0 -- the split's code is not in that instantiation)
    const bool has_part = !CUT && a.part_S > 0 && (int)blockIdx.x < (a.n_tiles - n_full) * a.part_S;
    const int ptile = has_part ? n_full + (int)blockIdx.x / a.part_S : 0;
    // A workgroup's list of nst items: its nmy pipelined tiles with, when it has one, its part of a split tile in the
    // SECOND TO LAST place -- so that the part's tail (partial stores acknowledged, the arrival counter) runs under the
    // last whole tile instead of at the end of the kernel, where only the last arriver's finishing pass is left.
    const int nst = nmy + (has_part ? 1 : 0);
    const int ppos = has_part ? nmy - 1 : 0x7fffffff;  // (has_part: n_full >= gridDim.x, so nmy >= 1)
    auto tile_of = [&](int j) { return j < ppos ? (int)blockIdx.x + j * (int)gridDim.x : (j == ppos ? ptile : (int)blockIdx.x + (j - 1) * (int)gridDim.x); };
    unsigned* const plast = cnt + 24;  // 1: this workgroup was the last of its split tile's to arrive (set by loader wave 0)

    if (w >= NA + NB) {
        // ================================ loaders ===================================================
        const int lt = threadIdx.x - (NA + NB) * 64;
        const int rA = rp / 4, nA = Tt * rA, rB = Fp / 4, nB = Tt * rB;
        const bool fits = rp <= 256 && nB < PB * NLT;  // (H: one piece per row; V: the last slot is the straddling cell's)
        // floor(i / d) = umulhi(i, ceil(2^32 / d)) for 0 <= i < 2^16 (i < 20 * 256 here), d >= 1
        const unsigned invB = (unsigned)((0x100000000ull + (unsigned)rB - 1) / (unsigned)rB);
        const int lw = w - (NA + NB);
        // Beside two waves that issue MFMAs back to back a loader wave gets an instruction in only where they stall
        // (scripts/mfma_valu_overlap.hip), so the time from p2done(j) to ready(j+2) -- which the A team waits for -- is
        // set by the NUMBER of instructions between them.  That was ~390 (index arithmetic per cell and tile, 64-bit
        // addresses); this path has none of it:
        //  * global side through buffer instructions: descriptor (SGPRs, rebuilt per tile) + SCALAR offset of the cell +
        //    one fixed lane offset;
        //  * the H block by rows (wave lw takes rows lw, lw+4, ..., one 1 KiB piece each: rp <= 256), so an LDS address is
        //    (scalar row offset) + the same lane offset: one add per cell and tile;
        //  * the V block by linear cells (its rows are not a whole number of pieces) with one precomputed LDS offset per
        //    cell; slots past the end fall back onto the thread's own first cell, the LAST slot is the straddling cell's;
        //  * lanes past the end of a short H row (rp < 256) duplicate lane 0.
        // Every access stays unconditional (see below).
        constexpr int PR = Tt / NL;  // H rows per loader wave
        const int hv = lane * 4 < rp ? lane * 16 : 0;
        const unsigned gv = 16u * (unsigned)lt;
        const int nfB = nB / NLT;  // V cells wholly inside (>= 1)
        unsigned gbB = (lt + nfB * NLT < nB) ? gv + 16u * (unsigned)(nfB * NLT) : gv;
        asm volatile("" : "+v"(gbB));
        int loB[PB];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int ic = b == PB - 1 ? (int)(gbB >> 4) : (b < nfB ? lt + b * NLT : lt);
            const int t = (int)__umulhi((unsigned)ic, invB), k4 = ic - t * rB;
            loB[b] = Tt * ldh + t * ldr + 4 * k4;
            asm volatile("" : "+v"(loB[b]));
        }
        auto ldA = [&](__amdgpu_buffer_rsrc_t rs, int i) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hv, (lw + NL * i) * rp * 4, 0));
        };
        auto ldB = [&](__amdgpu_buffer_rsrc_t rs, int b) {
            if (b == PB - 1) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gbB, 0, 0));
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)gv, b < nfB ? b * NLT * 16 : 0, 0));
        };
        // (H_new leaves as NON-TEMPORAL stores: nobody reads it before the next launch, and as ordinary stores its 100 MB a launch pushed the
        //  W images -- which every workgroup re-reads from the L2 for every tile where they do not fit the LDS -- out of the L2s: C2 2 159-2 165 ->
        //  2 179-2 191 it/s, k_hstep_rp 228.6 -> 225.0 us, the k_wstats behind it 223.5 -> 222.2.  nt on the loaders' V / H LOADS, on
        //  k_wstats' loads or slab stores, sc1 instead of nt: all slower, profiles/r06_experiments.md section 11; k_hstep_rh, whose
        //  images fit the LDS: no difference.)
        auto stA = [&](__amdgpu_buffer_rsrc_t rs, int i, const f32x4& x) { buf_store_b128<kAuxNT>(rs, hv, (lw + NL * i) * rp * 4, x); };
        auto rsrc_of = [&](const float* base, int n_cells) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n_cells * 16, 0x00020000);
        };
        // CUT launches also run on 16 row tiles (513 rows at r <= 64), where a V tile is 17 cells a loader thread and does not fit the
        // register path (`fits`; with PB = 17 the loaders spill 66 registers and every CUT shape got slower).  The refill after
        // p2done was stage_in's batches of 8 cells -- two or three exposed HBM round trips per tile: the A team waited 10 k of
        // its 19 k cycles per tile for `ready` (phase stamps, 513 x 72000 r = 20).  Under CUT that path is LDS-DMA: every piece of
        // the tile (a padded row of H, three pieces of a row of V) in flight at once, no registers, one round trip.
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        auto dma_v = [&](int tile, float* bH) {
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)tile * Tt * Fp), 0, Tt * Fp * 4, 0x00020000);
            for (int t = lw; t < Tt; t += NL)
                for (int pc = 0; pc * 256 < Fp; ++pc)
                    if (pc * 256 + lane * 4 < Fp)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(bH + Tt * ldh + t * ldr + pc * 256), 16, lane * 16,
                                                                 (t * Fp + pc * 256) * 4, 0, 0);
        };
        auto dma_h = [&](int tile, float* bH) {
            const __amdgpu_buffer_rsrc_t rh =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)tile * Tt * rp), 0, Tt * rp * 4, 0x00020000);
            for (int t = lw; t < Tt; t += NL)
                if (lane * 4 < rp)  // (rp <= 64 under CUT: one piece per row)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(bH + t * ldh), 16, lane * 16, t * rp * 4, 0, 0);
        };
        if (CUT && !fits) {
            for (int j = 0; j < 2 && j < nst; ++j) {
                dma_h(tile_of(j), lds + j * bufsz);
                dma_v(tile_of(j), lds + j * bufsz);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing else orders a ds_read behind an LDS-DMA
            for (int j = 0; j < 2 && j < nst; ++j) {
                rp_post(ready, lw, (unsigned)(j + 1), lane);
                rp_post(vready, lw, (unsigned)(j + 1), lane);
            }
        } else {
            for (int j = 0; j < 2 && j < nst; ++j) {
                float* bH = lds + j * bufsz;
                stage_in2<NLT, PA, PB>(a.Hin + (size_t)tile_of(j) * Tt * rp, bH, Tt, rp, ldh, a.V + (size_t)tile_of(j) * Tt * Fp,
                                       bH + Tt * ldh, Tt, Fp, ldr, lt);
                rp_post(ready, lw, (unsigned)(j + 1), lane);
                rp_post(vready, lw, (unsigned)(j + 1), lane);
            }
        }
        for (int j = 0; j < nst; ++j) {
            float* bH = lds + (j & 1) * bufsz;
            const bool more = j + 2 < nst;  // (the workgroup's share of a split tile is staged like one more tile)
            // tile j+2 -> registers while tile j is still being worked on.  Every access below is UNCONDITIONAL: a slot past
            // the end of a block falls back onto the thread's OWN first cell (the same 16 bytes read and written again, by
            // the same thread, so program order keeps "copy out, then overwrite" intact -- a cell owned by another thread
            // could already hold the next tile), so the loader is straight-line code and the compiler's s_waitcnt
            // bookkeeping stays exact.  With predicated
            // accesses it fell back to vmcnt(0) in front of the first LDS write of the prefetched registers, i.e. it also
            // waited for the H stores issued just before to be acknowledged by HBM: 7 us of every 21 us tile period in
            // which neither team had a tile to work on (phase stamps of the diagnostic build).
            f32x4 xa[PR], xb[PB];
            if (more && fits) {
                const float* srcA = a.Hin + (size_t)tile_of(j + 2) * Tt * rp;
                const float* srcB = a.V + (size_t)tile_of(j + 2) * Tt * Fp;
#pragma unroll
                for (int b = 0; b < PR; ++b) xa[b] = ldA(rsrc_of(srcA, nA), b);
#pragma unroll
                for (int b = 0; b < PB; ++b) xb[b] = ldB(rsrc_of(srcB, nB), b);
            }
            SNMF_PIN();
            rp_await(p2done, (unsigned)(j + 1), a.stop);
            if (j == ppos) {
                // the part: nothing to copy out (its H block stays for the finishing pass: no later item takes this buffer).
                // Every B wave has waited for its partial stores: bump the tile's arrival counter; whoever brings it to a
                // multiple of S is the last of the tile's S workgroups and will finish it at the end of the kernel
                if (lw == 0) stress_jitter();
                if (lw == 0 && lane == 0) {
                    const unsigned old = __hip_atomic_fetch_add(a.part_cnt + (ptile - n_full), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *plast = (old % (unsigned)a.part_S == (unsigned)a.part_S - 1u) ? 1u : 0u;
                }
                continue;
            }
            float* const dstH = a.Hout + (size_t)tile_of(j) * Tt * rp;
            if (!fits) {  // shape too big for the register path: plain copy-out, then a refill that starts only now
                if constexpr (CUT) {
                    if (more) dma_v(tile_of(j + 2), bH);  // (the ratio image is free: P2 of tile j is done)
                    // H_new leaves BY ROWS, wave lw the rows lw, lw + 4, ... it is about to overwrite: stage_out's linear cell deal
                    // left another wave's cells of such a row unread when this wave's DMA landed in it (found by the shape fuzzer:
                    // 6 of 39 solves with three or more tiles per workgroup off by 1e-3 .. 1e-1; two tiles per workgroup -- all
                    // the unit tests had -- never refill)
#ifdef SNMF_REVERT_R5_RACE  // (diagnostic builds only: round 5's race put back, to show that the -m gpu fuzz test fails on it: profiles/r06_experiments.md)
                    stage_out<NLT>(dstH, bH, Tt, rp, ldh, lt);
#else
                    {
                        const char* const bHr = reinterpret_cast<const char*>(bH) + hv;
#pragma unroll
                        for (int b0 = 0; b0 < PR; b0 += 4) {
                            f32x4 ho[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) ho[u] = *reinterpret_cast<const f32x4*>(bHr + (lw + NL * (b0 + u)) * ldh * 4);
#pragma unroll
                            for (int u = 0; u < 4; ++u) stA(rsrc_of(dstH, nA), b0 + u, ho[u]);
                        }
                    }
#endif
                    if (more) {
                        dma_h(tile_of(j + 2), bH);         // (behind THIS wave's reads of the same rows in program order)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        rp_post(ready, lw, (unsigned)(j + 3), lane);
                        rp_post(vready, lw, (unsigned)(j + 3), lane);
                    }
                } else {
                    stage_out<NLT>(dstH, bH, Tt, rp, ldh, lt);
                    if (more) {
                        stage_in<NLT>(a.Hin + (size_t)tile_of(j + 2) * Tt * rp, bH, Tt, rp, ldh, lt);
                        stage_in<NLT>(a.V + (size_t)tile_of(j + 2) * Tt * Fp, bH + Tt * ldh, Tt, Fp, ldr, lt);
                        rp_post(ready, lw, (unsigned)(j + 3), lane);
                        rp_post(vready, lw, (unsigned)(j + 3), lane);
                    }
                }
                continue;
            }
            // the updated H tile leaves (LDS -> registers -> HBM; the stores are only ISSUED here) ...
            // (four LDS reads in flight at a time: one read, one wait, one store at a time put eight LDS latencies between
            //  p2done and ready)
            const char* const bHl = reinterpret_cast<const char*>(bH) + hv;  // this lane's column of the H image
#pragma unroll
            for (int b0 = 0; b0 < PR; b0 += 4) {
                f32x4 ho[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ho[u] = *reinterpret_cast<const f32x4*>(bHl + (lw + NL * (b0 + u)) * ldh * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) stA(rsrc_of(dstH, nA), b0 + u, ho[u]);
            }
            // ... and the prefetched tile takes its place (its loads completed long ago): the H block first, so that `ready`
            // does not wait for the V commit
            if (more) {
#pragma unroll
                for (int b = 0; b < PR; ++b)
                    *reinterpret_cast<f32x4*>(const_cast<char*>(bHl) + (lw + NL * b) * ldh * 4) = xa[b];
                rp_post(ready, lw, (unsigned)(j + 3), lane);
#pragma unroll
                for (int b = 0; b < PB; ++b) *reinterpret_cast<f32x4*>(bH + loB[b]) = xb[b];
                rp_post(vready, lw, (unsigned)(j + 3), lane);
            }
        }
    } else if (idle) {
        // (nothing: straight to the final reduction)
    } else if (w < NA) {
        // ================================ A team: P1 ================================================
        SNMF_STAMP_DECL
        auto a_item = [&](const int j) {  // P1 of the whole tile in place j of the list
            const int t0 = tile_of(j) * Tt;
            float* Hs = lds + (j & 1) * bufsz;
            float* Rs = Hs + Tt * ldh;
            SNMF_STAMP(11);
            bool waited = false;
            auto gate_ready = [&]() {  // the tile's H image: waited for once, behind the first W fragments of the tile
                if (!waited) {
                    SNMF_STAMP(4);
                    rp_await(ready, (unsigned)(j + 1), a.stop);
                    SNMF_STAMP(3);  // (diagnostic builds: the wait for the staged tile by itself)
                }
                waited = true;
            };
            bool vwaited = false;
            auto gate_v = [&]() {  // the tile's V block: needed by the epilogues only
                if (!vwaited) rp_await(vready, (unsigned)(j + 1), a.stop);
                vwaited = true;
            };
            const int fl = lane & 31, h = lane >> 5;
            const float* sp = Hs + fl * ldh + 4 * h;
            float dsum = 0.f;
            const bool early_pair = CUT && a.nf > 2 * NA;  // (see the B team's k-ranges under CUT)
            for (int phi = w; phi < a.nf; phi += 2 * NA) {
                if (phi + NA < a.nf) {
                    f32x16 acc[2] = {zero16(), zero16()};
                    const int so[2] = {phi * rp * 128, (phi + NA) * rp * 128};
                    contract_shared_buf<2>(acc, wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32), lane * 16, so, sp, a.nqk, gate_ready);
                    SNMF_STAMP(4);
                    gate_v();
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                    if (phi == w && !early_pair) rp_post(p1a, w, (unsigned)(j + 1), lane);  // row tile w < NA: this wave's share of the first 32*NA ratio rows
                    rp_p1_epilogue<OBJ>(a, acc[1], Rs, phi + NA, t0, lane, dsum);
                    if (phi == w && early_pair) rp_post(p1a, w, (unsigned)(j + 1), lane);   // CUT, 9..16 row tiles: the first 64*NA rows
                    SNMF_STAMP(5);
                } else {
                    f32x16 acc[1] = {zero16()};
                    const int so[1] = {phi * rp * 128};
                    contract_shared_buf<1>(acc, wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32), lane * 16, so, sp, a.nqk, gate_ready);
                    SNMF_STAMP(4);
                    gate_v();
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                    if (phi == w) rp_post(p1a, w, (unsigned)(j + 1), lane);
                    SNMF_STAMP(5);
                }
            }
            gate_ready();  // (a wave without a row tile has not waited yet)
            gate_v();
            if (w >= a.nf) rp_post(p1a, w, (unsigned)(j + 1), lane);  // a wave without a row tile still reports
            if (OBJ) acc_div += (double)dsum;
            SNMF_STAMP(6);
            rp_post(p1b, w, (unsigned)(j + 1), lane);
            // The extra row (F = 32n+1): 8 frames per A wave, AFTER p1b -- the B team's P2 needs it for its very last
            // k-block only (xdone), so it is off every critical path, and on an MFMA wave its ~80 instructions cost their
            // issue cycles; on the loader waves, which only get an instruction in where the MFMA waves stall, they cost
            // 4.3 % of the kernel (0.2485 -> 0.2379 ms with the arithmetic removed).
            if (a.xr) {
                hstep_p1_xrow<NA, 1, BM_KL, OBJ>(a, Hs, Rs, wxs, t0, w, lane, true, acc_div);
                rp_post(xdone, w, (unsigned)(j + 1), lane);
            }
        };
        // (the part and the last tile come after the loop, not as a branch inside it: the loop sits at the register limit)
        for (int j = 0; j < (has_part ? nmy - 1 : nmy); ++j) a_item(j);
        if (has_part) {
            acc_div += rp_part_p1<OBJ>(a, lds + (ppos & 1) * bufsz, wxs, cnt, ppos, ptile, w, lane);
            a_item(nmy);
        }
        SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * (NA + NB) + w) * 12, 12);
        SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * (NA + NB) + w);
    } else {
        // ================================ B team: P2 ================================================
        const int wb = w - NA;
        SNMF_STAMP_DECL
        // 1 ./ dph of this wave's columns: constants of the launch when the wave owns one pair of column tiles (nk <= 8)
        // Which column tiles this wave owns: tile wb and wb + NB, ... -- unless the A team has work for two waves only
        // (nf <= 2: a Mel spectrogram of 64 bands): A waves 0 and 1 share their SIMDs with B waves 0 and 1, and with P1 AND P2
        // of a tile on SIMDs 0 / 1 while SIMDs 2 / 3 carry a quarter of P2 each, the tile period was the loaded SIMDs' (Mel
        // 64 x 72000 r = 100: 84 of a tile's 232 MFMAs on each of them).  Then B waves 2 and 3 take ALL column tiles in pairs
        // (kb, kb + 2): P1 on SIMDs 0 / 1, P2 on SIMDs 2 / 3.
        const bool b_shift = a.nf <= 2 && a.nk >= 2;
        const int kb = b_shift ? wb - 2 : wb, kpair = b_shift ? 2 : NB;
        const bool one_group = a.nk <= 2 * kpair;
        f32x4 dp0[4], dp1[4];
        if (one_group && kb >= 0 && kb < a.nk) {
            rp_p2_consts(a, kb, lane, dp0);
            if (kb + kpair < a.nk) rp_p2_consts(a, kb + kpair, lane, dp1);
        }
        unsigned lx_seq = 0;  // CUT: tiles this wave has put through the partial buffers
        auto b_item = [&](const int j) {  // P2 of the whole tile in place j of the list
            const int t0 = tile_of(j) * Tt;
            float* Hs = lds + (j & 1) * bufsz;
            const float* Rs = Hs + Tt * ldh;
            SNMF_STAMP(11);
            if constexpr (CUT) {
                const int fl = lane & 31, h = lane >> 5;
                const float* sp = Rs + fl * ldr + 4 * h;
                const int nq = a.Fq / 8, nqm = a.xr ? nq - 1 : nq;
                const int ne0 = 4 * NA * (a.nf > 2 * NA ? 2 : 1), ne = ne0 < nqm ? ne0 : nqm;  // k-blocks over the ratio rows p1a covers
                if (cut_pair) {
                    // PAIR form (two column tiles where the four-way form's 32 KB of partial tiles do not fit: 513 rows at r = 33..64,
                    // settings/bak_IS16_results/initial_setting_IMCRA.m:47-48 R = 50): B waves (2 i, 2 i + 1) share column tile i, each
                    // half of the contraction (the odd wave also the extra row's block); a wave stores the two register groups its
                    // partner finishes (2 KB a wave, 8 KB in all), reads the partner's two for its own, adds them in wave order and runs
                    // the epilogue on groups 2 hf, 2 hf + 1.  Same MFMA count per wave as the four-way form.
                    const int ti = wb >> 1, hf = wb & 1;
                    auto gate_e = [&]() { rp_await(p1a, (unsigned)(j + 1), a.stop); };
                    auto gate_l = [&]() { rp_await(p1b, (unsigned)(j + 1), a.stop); };
                    auto gate_x = [&]() { rp_await(xdone, (unsigned)(j + 1), a.stop); };
                    const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                    f32x16 a1[1] = {zero16()};
                    {   // this wave's half of the EARLY rows (those p1a covers), then its half of the late ones
                        const int qlo = ne * hf / 2, qhi = ne * (hf + 1) / 2;
                        const int so[1] = {ti * a.Fq * 128 + qlo * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so, sp + 8 * qlo, qhi - qlo, gate_e);
                    }
                    if (nqm > ne) {
                        const int qlo = ne + (nqm - ne) * hf / 2, qhi = ne + (nqm - ne) * (hf + 1) / 2;
                        const int so[1] = {ti * a.Fq * 128 + qlo * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so, sp + 8 * qlo, qhi - qlo, gate_l);
                    } else {
                        gate_l();
                    }
                    if (hf == 1 && a.xr) {
                        const int so3[1] = {ti * a.Fq * 128 + nqm * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so3, sp + 8 * nqm, 1, gate_x);
                    }
                    SNMF_STAMP(9);
                    rp_await(rdone, lx_seq, a.stop);
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {  // the partner's groups 2 (1 - hf) + gg, wave-uniform register choice by two selects
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = hf ? a1[0][4 * gg + e] : a1[0][8 + 4 * gg + e];
                        *reinterpret_cast<f32x4*>(Ps + (wb * 2 + gg) * 256 + lane * 4) = o;
                    }
                    ++lx_seq;
                    rp_post(wdone, wb, lx_seq, lane);
                    rp_await(wdone, lx_seq, a.stop);
                    f32x4 tot[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(Ps + ((wb ^ 1) * 2 + gg) * 256 + lane * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mine = hf ? a1[0][8 + 4 * gg + e] : a1[0][4 * gg + e];
                            tot[gg][e] = hf ? x[e] + mine : mine + x[e];  // (the even wave's partial first: one order on both sides)
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    rp_post(rdone, wb, lx_seq, lane);
                    float shsum = 0.f;
                    rp_cut_group_epilogue<OBJ>(a, tot[0], Hs, ti, 2 * hf, t0, lane, rdph, shsum);
                    rp_cut_group_epilogue<OBJ>(a, tot[1], Hs, ti, 2 * hf + 1, t0, lane, rdph, shsum);
                    if (OBJ) acc_sh += (double)shsum;
                    rp_post(p2done, wb, (unsigned)(j + 1), lane);
                    SNMF_STAMP(10);
                    return;
                }
                // this wave's k-blocks of every column tile: a quarter of the EARLY ratio rows (those p1a covers: the first 32 NA, or
                // 64 NA on more than 2 NA row tiles) and a quarter of the late ones (p1b) -- every B wave starts while the A team is
                // still at work and has only half of its blocks left when the last ratio row lands (as contiguous quarters two
                // waves sat idle until p1b); wave 3 also takes the extra row's block
                auto gate_e = [&]() { rp_await(p1a, (unsigned)(j + 1), a.stop); };
                auto gate_l = [&]() { rp_await(p1b, (unsigned)(j + 1), a.stop); };
                auto gate_x = [&]() { rp_await(xdone, (unsigned)(j + 1), a.stop); };
                const int qe0 = ne * wb / NB, qe1 = ne * (wb + 1) / NB;
                const int ql0 = ne + (nqm - ne) * wb / NB, ql1 = ne + (nqm - ne) * (wb + 1) / NB;
                const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                const bool two = a.nk > 1;
                f32x16 acc[2] = {zero16(), zero16()};
                if (two) {
                    {
                        const int so[2] = {qe0 * 1024, a.Fq * 128 + qe0 * 1024};
                        contract_shared_buf<2>(acc, rsk, lane * 16, so, sp + 8 * qe0, qe1 - qe0, gate_e);
                    }
                    if (ql1 > ql0) {
                        const int so[2] = {ql0 * 1024, a.Fq * 128 + ql0 * 1024};
                        contract_shared_buf<2>(acc, rsk, lane * 16, so, sp + 8 * ql0, ql1 - ql0, gate_l);
                    } else {
                        gate_l();
                    }
                    if (wb == NB - 1 && a.xr) {
                        const int so3[2] = {nqm * 1024, a.Fq * 128 + nqm * 1024};
                        contract_shared_buf<2>(acc, rsk, lane * 16, so3, sp + 8 * nqm, 1, gate_x);
                    }
                } else {
                    f32x16 a1[1] = {zero16()};
                    {
                        const int so[1] = {qe0 * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so, sp + 8 * qe0, qe1 - qe0, gate_e);
                    }
                    if (ql1 > ql0) {
                        const int so[1] = {ql0 * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so, sp + 8 * ql0, ql1 - ql0, gate_l);
                    } else {
                        gate_l();
                    }
                    if (wb == NB - 1 && a.xr) {
                        const int so3[1] = {nqm * 1024};
                        contract_shared_buf<1>(a1, rsk, lane * 16, so3, sp + 8 * nqm, 1, gate_x);
                    }
                    acc[0] = a1[0];
                }
                // partial tiles -> LDS, once every wave has read the previous tile's
                SNMF_STAMP(9);
                rp_await(rdone, lx_seq, a.stop);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !two) break;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(Ps + ((wb * a.nk + i) * 4 + g) * 256 + lane * 4) = o;
                    }
                }
                ++lx_seq;
                rp_post(wdone, wb, lx_seq, lane);
                rp_await(wdone, lx_seq, a.stop);
                // register group wb of every column tile, the four waves' partials in wave order
                f32x4 tot[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !two) break;
#pragma unroll
                    for (int ww = 0; ww < NB; ++ww) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(Ps + ((ww * a.nk + i) * 4 + wb) * 256 + lane * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[i][e] += x[e];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rp_post(rdone, wb, lx_seq, lane);
                float shsum = 0.f;
                rp_cut_group_epilogue<OBJ>(a, tot[0], Hs, 0, wb, t0, lane, rdph, shsum);
                if (two) rp_cut_group_epilogue<OBJ>(a, tot[1], Hs, 1, wb, t0, lane, rdph, shsum);
                if (OBJ) acc_sh += (double)shsum;
                rp_post(p2done, wb, (unsigned)(j + 1), lane);
                SNMF_STAMP(10);
                return;
            }
            auto gate_p1a = [&]() { rp_await(p1a, (unsigned)(j + 1), a.stop); };
            const int fl = lane & 31, h = lane >> 5;
            const float* sp = Rs + fl * ldr + 4 * h;
            float shsum = 0.f;
            // k-blocks over the ratio rows of the row tiles 0..NA-1 (never the extra row's block) / the rest
            const int nq = a.Fq / 8, nq1 = 4 * (a.nf < NA ? a.nf : NA);
            const int nqm = a.xr ? nq - 1 : nq;  // k-blocks over the ratio rows proper; the extra row's is a phase of its own, gated by xdone
            auto gate_p1b = [&]() { rp_await(p1b, (unsigned)(j + 1), a.stop); };
            auto gate_x = [&]() { rp_await(xdone, (unsigned)(j + 1), a.stop); };
            for (int kap = kb < 0 ? a.nk : kb; kap < a.nk; kap += 2 * kpair) {
                if (kap + kpair < a.nk) {
                    f32x16 acc[2] = {zero16(), zero16()};
                    if (!one_group) {
                        rp_p2_consts(a, kap, lane, dp0);
                        rp_p2_consts(a, kap + kpair, lane, dp1);
                    }
                    {
                        const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                        const int so[2] = {kap * a.Fq * 128, (kap + kpair) * a.Fq * 128};
                        contract_shared_buf<2>(acc, rsk, lane * 16, so, sp, nq1, gate_p1a);
                        if (nqm > nq1) {
                            const int so2[2] = {so[0] + nq1 * 1024, so[1] + nq1 * 1024};
                            contract_shared_buf<2>(acc, rsk, lane * 16, so2, sp + 8 * nq1, nqm - nq1, gate_p1b);
                        } else {
                            gate_p1b();
                        }
                        if (nq > nqm) {
                            const int so3[2] = {so[0] + nqm * 1024, so[1] + nqm * 1024};
                            contract_shared_buf<2>(acc, rsk, lane * 16, so3, sp + 8 * nqm, nq - nqm, gate_x);
                        }
                    }
                    SNMF_STAMP(9);
                    rp_p2_epilogue<OBJ>(a, acc[0], Hs, kap, t0, lane, dp0, shsum);
                    rp_p2_epilogue<OBJ>(a, acc[1], Hs, kap + kpair, t0, lane, dp1, shsum);
                    SNMF_STAMP(10);
                } else {
                    f32x16 acc[1] = {zero16()};
                    if (!one_group) rp_p2_consts(a, kap, lane, dp0);
                    {
                        const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                        const int so[1] = {kap * a.Fq * 128};
                        contract_shared_buf<1>(acc, rsk, lane * 16, so, sp, nq1, gate_p1a);
                        if (nqm > nq1) {
                            const int so2[1] = {so[0] + nq1 * 1024};
                            contract_shared_buf<1>(acc, rsk, lane * 16, so2, sp + 8 * nq1, nqm - nq1, gate_p1b);
                        } else {
                            gate_p1b();
                        }
                        if (nq > nqm) {
                            const int so3[1] = {so[0] + nqm * 1024};
                            contract_shared_buf<1>(acc, rsk, lane * 16, so3, sp + 8 * nqm, nq - nqm, gate_x);
                        }
                    }
                    rp_p2_epilogue<OBJ>(a, acc[0], Hs, kap, t0, lane, dp0, shsum);
                }
            }
            if (OBJ) acc_sh += (double)shsum;
            // a wave WITHOUT a column tile (nk < NB) has waited for nothing yet: it must not run ahead and count towards
            // p2done of a later tile before the working waves are there (the counters are totals, not per-tile flags)
            gate_p1a();
            gate_p1b();
            if (a.xr) gate_x();
            rp_post(p2done, wb, (unsigned)(j + 1), lane);
        };
        for (int j = 0; j < (has_part ? nmy - 1 : nmy); ++j) b_item(j);
        if (has_part) {
            rp_part_p2(a, lds + (ppos & 1) * bufsz + Tt * ldh, cnt, ppos, wb, lane);
            b_item(nmy);
        }
        SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * (NA + NB) + w) * 12, 12);
        SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * (NA + NB) + w);
    }

    if (has_part) acc_sh += rp_part_finish<OBJ>(a, lds + (ppos & 1) * bufsz, plast, ptile);

    if (OBJ) {
        // deterministic workgroup reduction of the two fp64 partial sums (as k_hstep)
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);  // [2][NTHR]
        red[threadIdx.x] = acc_div;
        red[NTHR + threadIdx.x] = acc_sh;
        __syncthreads();
        if ((int)threadIdx.x + 512 < NTHR) {
            red[threadIdx.x] += red[threadIdx.x + 512];
            red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + 512];
        }
        __syncthreads();
        for (int s = 256; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                red[threadIdx.x] += red[threadIdx.x + s];
                red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) obj_partial_out(a, blockIdx.x, red[0], red[NTHR]);
    }
}

// ============================================================================================
// k_hstep_rh: the role pipeline for spectrograms of 9..16 row tiles -- F = 513, the geometry the reference ships
// (settings/initial_setting_SNMF_NAT.m:21-29: 40 ms window -> 1024-point FFT -> 513 rows; run_basis_train.m:88 at
// r = 100, run_basis_DNMF.m:40 at r = 200) -- whose two tile buffers do not fit the 160 KiB LDS (a 32-frame ratio image
// of 513 rows alone is 67 KB).  Same arithmetic and MFMA order per tile as k_hstep_rp / k_hstep<8,1,4,KL>, but the
// pipeline runs on HALF tiles: the H block is double-buffered as before, the ratio image exists ONCE and its two row
// halves (row tiles 0-7 | 8-15 + the extra row) are the pipeline's buffers:
//     unit u = (tile j, half hf).   A team: P1 of the half's row tiles (two per wave) -> ratio rows in place over the
//     staged V half;   B team: the half's k-blocks of W^T*ratio, accumulating over both halves of a tile in registers,
//     then the H update;   loaders: V half (j+1, hf) -> registers while the B team still reads ratio half (j, hf),
//     committed when the B team is through with it (bdone), H block j+2 as in k_hstep_rp.
// While the B team contracts half (j, hf) the A team runs the loop of the unit after the next one (its MFMA loop needs
// only the H block; the V half is needed by its epilogue), so a V commit has one A loop of slack.  Signals, each four
// per-wave progress words as in k_hstep_rp: ready (H block of tile j), vready (V half of unit u), p1 (ratio half of unit
// u whole), xdone (extra row of tile j), bdone (B team through with ratio half u), p2done (H_j updated).
// Needs rp <= 256 (one pair of column tiles per B wave: the ratio halves are read once per tile).
// ============================================================================================
template <int NACC, bool OBJ, typename G0, typename G1, typename GX>
__device__ __forceinline__ void rh_p2_tile(const StepArgs& a, float* Hs, const float* Rs, int kap, int t0, int lane, int wb, int j,
                                           unsigned* bdone, const f32x4 (&dp0)[4], const f32x4 (&dp1)[4], float& shsum, G0 g0, G1 g1,
                                           GX gx) {
    constexpr int NB = 4;
    const int fl = lane & 31, h = lane >> 5;
    const float* sp = Rs + fl * a.ldr + 4 * h;
    const int nq1 = 4 * (a.nf - 8);  // k-blocks of the second half's row tiles
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = zero16();
    const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
    int so[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) so[i] = (kap + i * NB) * a.Fq * 128;
    contract_shared_buf<NACC>(acc, rsk, lane * 16, so, sp, 32, g0);
    rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
#pragma unroll
    for (int i = 0; i < NACC; ++i) so[i] += 32 * 1024;
    contract_shared_buf<NACC>(acc, rsk, lane * 16, so, sp + 256, nq1, g1);
    if (a.xr) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) so[i] += nq1 * 1024;
        contract_shared_buf<NACC>(acc, rsk, lane * 16, so, sp + 256 + 8 * nq1, 1, gx);
    }
    rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
    rp_p2_epilogue<OBJ>(a, acc[0], Hs, kap, t0, lane, dp0, shsum);
    if (NACC == 2) rp_p2_epilogue<OBJ>(a, acc[NACC - 1], Hs, kap + NB, t0, lane, dp1, shsum);
}

template <bool OBJ, int LXH = 0>
__global__ __launch_bounds__(768, 3) void k_hstep_rh(StepArgs a) {
    constexpr int NA = 4, NB = 4, NL = 4, NTHR = (NA + NB + NL) * 64, Tt = 32, PR = Tt / NL;
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp, ldh = a.ldh, ldr = a.ldr;
    const int hsz = Tt * ldh;          // floats per H buffer [Tt][ldh]
    float* Rs = lds + 2 * hsz;         // [Tt][ldr] the one ratio image (V staged into it by halves)
    float* wxs = Rs + Tt * ldr;        // [rp] extra row of W
    unsigned* cnt = reinterpret_cast<unsigned*>(wxs + rp);
    unsigned *ready = cnt, *vready = cnt + 4, *p1 = cnt + 8, *xdone = cnt + 12, *bdone = cnt + 16, *p2done = cnt + 20;
    double acc_div = 0.0, acc_sh = 0.0;
    if (a.xr) {
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
        // the unused cells of the extra 8-deep k-block stay zero for the whole kernel (the V commits write Fm .. Fm+3)
        for (int i = threadIdx.x; i < Tt * 8; i += NTHR) Rs[(i >> 3) * ldr + a.Fm + (i & 7)] = 0.f;
    }
    if (threadIdx.x < 40) cnt[threadIdx.x] = 0u;
    // lxh (nk = 4 with 1..4 real columns in the fourth column tile: r = 97..100, the reference's R = 100): P2 is cut over
    // the CONTRACTION instead of over the column tiles -- B wave wb takes k-blocks [16 wb, 16 wb + 16) (+ the extra row's
    // for the last wave) of the three full column tiles (three accumulators share each LDS fragment: 192 MFMAs a wave
    // instead of 260 on a tile that is 87 % padding), the leftover columns of its rows as VALU work from a small LDS copy
    // of those columns of W (wl), and the waves' partial tiles are added through LDS (Ps, Pl) before the epilogue: every
    // SIMD then carries the same work (a re-deal of whole tiles cannot balance them).
    // LXH == 2 (CUT2; nk = 7 with 1..8 real columns in the seventh tile: r = 193..200, run_basis_DNMF.m:40 at the shipped
    // R_x + R_d): seven column tiles on four B waves are 2, 2, 2, 1 -- three SIMDs carry 920 MFMAs a tile, the fourth 660.  Here
    // the B waves work in PAIRS: pair p owns the full tiles 3p .. 3p+2 and the leftover column group p (columns 192 + 4p .. + 3,
    // 4x4x1 MFMAs on the same fragments); inside a pair wave s takes k-blocks [16 s, 16 s + 16) of EACH ratio half (+ the extra
    // row's, s = 1): 396 + 132 short MFMAs a wave, every SIMD the same.  The partners swap what the other finishes through LDS
    // (one whole tile + half of the third, 6 KB a wave) and each runs 1.5 tile epilogues; wave 0 also the leftover columns'.
    unsigned *wdone = cnt + 28, *rdone = cnt + 32;            // B waves: partials of place j written / read
    float* Ps = reinterpret_cast<float*>(cnt + 40);          // LXH 1: [4 waves][3 tiles][4 g][64 lanes][4]; LXH 2: [4 waves][6][64 lanes][4]
    float* Pl = Ps + (LXH == 2 ? 4 * 6 * 256 : 4 * 3 * 1024);  // [4 waves][64 lanes][4] partial leftover columns
    float* rdph = Pl + 4 * 64 * 4;                            // LXH 2: [rp] 1 ./ dph (rh_cut_epilogue)
    if (LXH == 2 && !a.S)
        for (int k = threadIdx.x; k < rp; k += NTHR) rdph[k] = fast_rcp(a.dphv[k]);
    __syncthreads();
    // The split last round as in k_hstep_rp ("the split last round" above), with 16 row tiles: part p of S owns the
    // CONTIGUOUS row tiles [p nfp, (p+1) nfp), nfp = 16 / S -- half a half (S = 4: one row tile per A wave) or a whole
    // half (S = 2: the ordinary pairs), so a part has work in ONE unit of its place and its k-blocks are one range.
    const int n_full = a.part_S > 0 ? a.n_full : a.n_tiles;
    const int nmy = (int)blockIdx.x < n_full ? (n_full - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const bool has_part = a.part_S > 0 && (int)blockIdx.x < (a.n_tiles - n_full) * a.part_S;
    const int pS = a.part_S > 0 ? a.part_S : 1, unit = (int)blockIdx.x, pp = unit % pS;
    const int ptile = has_part ? n_full + unit / pS : 0;
    const int nfp = a.nf / pS, php = (pp * nfp) / 8;  // row tiles per part; the half its row tiles are in
    const bool own_x = a.xr && pp == pS - 1;
    const int nst = nmy + (has_part ? 1 : 0);
    const int ppos = has_part ? nmy - 1 : 0x7fffffff;  // the part: second to last in the list (has_part: nmy >= 1)
    auto tile_of = [&](int j) { return j < ppos ? (int)blockIdx.x + j * (int)gridDim.x : (j == ppos ? ptile : (int)blockIdx.x + (j - 1) * (int)gridDim.x); };
    unsigned* const plast = cnt + 24;  // 1: this workgroup was the last of its split tile's to arrive (set by loader wave 0)

    if (w >= NA + NB) {
        // ================================ loaders ===================================================
        // As k_hstep_rp's: buffer instructions with scalar offsets, one lane offset per block, every access unconditional.
        //  * H block by rows: wave lw takes rows lw, lw+4, ..., one 1 KiB piece each (lanes past a short row duplicate lane 0);
        //  * V half hf by rows as well: columns [256 hf, 256 hf + 256) of a frame row are 64 16-byte cells, lane <-> cell; the
        //    second half ends at Fp (its last cell carries the extra row's value): lanes past it duplicate lane 0.
        const int lw = w - (NA + NB);
        const int hv = lane * 4 < rp ? lane * 16 : 0;
        const int nc1 = (Fp - 256) / 4;  // cells per frame row of the second half
        const int vv1 = lane < nc1 ? lane * 16 : 0;
        // F = 513: the second half has 65 cells a row -- the 65th (columns 512..515: the extra row's value) of this wave's
        // eight rows is one more load, eight lanes wide (the others duplicate them)
        const bool xcell = nc1 > 64;
        const int xoff_g = ((lw + NL * (lane & 7)) * Fp + 512) * 4, xoff_l = ((lw + NL * (lane & 7)) * ldr + 512) * 4;
        auto rsrc_of = [&](const float* base, int bytes) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
        };
        // (H_in is dead behind this launch: as NON-TEMPORAL loads its 100 MB at r = 200 do not displace V -- which k_wstats / the next launch read again --
        //  in the cache behind the L2s: c4h 2 990-2 998 -> 3 021-3 025 it/s; no difference at r = 100 or for k_hstep_rp, profiles/r06_experiments.md section 11)
        auto ldA = [&](__amdgpu_buffer_rsrc_t rs, int i) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hv, (lw + NL * i) * rp * 4, kAuxNT));
        };
        auto ldV = [&](__amdgpu_buffer_rsrc_t rs, int hf, int i) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hf ? vv1 : lane * 16, ((lw + NL * i) * Fp + 256 * hf) * 4, 0));
        };
        auto stageH = [&](int tile, float* dst) {  // start-up only: straight through registers
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.Hin + (size_t)tile * Tt * rp, Tt * rp * 4);
            f32x4 x[PR];
#pragma unroll
            for (int i = 0; i < PR; ++i) x[i] = ldA(rs, i);
#pragma unroll
            for (int i = 0; i < PR; ++i) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dst) + hv + (lw + NL * i) * ldh * 4) = x[i];
        };
        auto commitV = [&](int hf, const f32x4 (&x)[PR]) {
            char* base = reinterpret_cast<char*>(Rs) + (hf ? vv1 + 1024 : lane * 16);
#pragma unroll
            for (int i = 0; i < PR; ++i) *reinterpret_cast<f32x4*>(base + (lw + NL * i) * ldr * 4) = x[i];
        };
        auto ldX = [&](__amdgpu_buffer_rsrc_t rs) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff_g, 0, 0));
        };
        auto commitX = [&](const f32x4& x) { *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(Rs) + xoff_l) = x; };
        for (int j = 0; j < 2 && j < nst; ++j) {
            stageH(tile_of(j), lds + j * hsz);
            rp_post(ready, lw, (unsigned)(j + 1), lane);
        }
        if (nst > 0) {
            const __amdgpu_buffer_rsrc_t rv = rsrc_of(a.V + (size_t)tile_of(0) * Tt * Fp, Tt * Fp * 4);
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 x[PR];
#pragma unroll
                for (int i = 0; i < PR; ++i) x[i] = ldV(rv, hf, i);
                commitV(hf, x);
                if (hf && xcell) commitX(ldX(rv));
                rp_post(vready, lw, (unsigned)(hf + 1), lane);
            }
        }
        for (int j = 0; j < nst; ++j) {
            float* bH = lds + (j & 1) * hsz;
            const bool more_v = j + 1 < nst, more_h = j + 2 < nst;
            const __amdgpu_buffer_rsrc_t rv = rsrc_of(a.V + (size_t)tile_of(more_v ? j + 1 : j) * Tt * Fp, Tt * Fp * 4);
            f32x4 xv[PR], xa[PR], xx;
            // unit (j, 0): V half (j+1, 0) and the H block of tile j+2 -> registers, then wait for the B team to leave half 0
            if (more_v) {
#pragma unroll
                for (int i = 0; i < PR; ++i) xv[i] = ldV(rv, 0, i);
            }
            if (more_h) {
                const __amdgpu_buffer_rsrc_t rh = rsrc_of(a.Hin + (size_t)tile_of(j + 2) * Tt * rp, Tt * rp * 4);
#pragma unroll
                for (int i = 0; i < PR; ++i) xa[i] = ldA(rh, i);
            }
            SNMF_PIN();
            rp_await(bdone, (unsigned)(2 * j + 1), a.stop);
            if (more_v) {
                commitV(0, xv);
                rp_post(vready, lw, (unsigned)(2 * j + 3), lane);
#pragma unroll
                for (int i = 0; i < PR; ++i) xv[i] = ldV(rv, 1, i);
                if (xcell) xx = ldX(rv);
            }
            SNMF_PIN();
            rp_await(bdone, (unsigned)(2 * j + 2), a.stop);
            if (more_v) {
                commitV(1, xv);
                if (xcell) commitX(xx);
                rp_post(vready, lw, (unsigned)(2 * j + 4), lane);
            }
            rp_await(p2done, (unsigned)(j + 1), a.stop);
            if (j == ppos) {  // the part: no H to copy out (its H block stays for the finishing pass); bump the tile's arrival counter
                if (lw == 0) stress_jitter();
                if (lw == 0 && lane == 0) {
                    const unsigned old = __hip_atomic_fetch_add(a.part_cnt + (ptile - n_full), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *plast = (old % (unsigned)pS == (unsigned)pS - 1u) ? 1u : 0u;
                }
                continue;
            }
            // the updated H tile leaves (LDS -> registers -> HBM; four LDS reads in flight at a time) ...
            const __amdgpu_buffer_rsrc_t ro = rsrc_of(a.Hout + (size_t)tile_of(j) * Tt * rp, Tt * rp * 4);
            const char* const bHl = reinterpret_cast<const char*>(bH) + hv;
#pragma unroll
            for (int b0 = 0; b0 < PR; b0 += 4) {
                f32x4 ho[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ho[u] = *reinterpret_cast<const f32x4*>(bHl + (lw + NL * (b0 + u)) * ldh * 4);
#pragma unroll
                // (non-temporal, like k_hstep_rp's: with H_in's loads and H_new's stores both streaming, the 205 MB of V at 513 x 100000 stay in the
                //  cache behind the L2s from one H-only iteration to the next: c4h 3 034 -> 3 051 it/s on top of the loads' 2 994 -> 3 034)
                for (int u = 0; u < 4; ++u) buf_store_b128<kAuxNT>(ro, hv, (lw + NL * (b0 + u)) * rp * 4, ho[u]);
            }
            // ... and the H block of tile j+2 takes its place
            if (more_h) {
#pragma unroll
                for (int i = 0; i < PR; ++i) *reinterpret_cast<f32x4*>(const_cast<char*>(bHl) + (lw + NL * i) * ldh * 4) = xa[i];
                rp_post(ready, lw, (unsigned)(j + 3), lane);
            }
        }
    } else if (w < NA) {
        // ================================ A team: P1, two units per tile ==============================
        const __amdgpu_buffer_rsrc_t rsw = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);
        auto a_item = [&](const int j) {  // P1 of the whole tile in place j of the list
            const int t0 = tile_of(j) * Tt;
            float* Hs = lds + (j & 1) * hsz;
            const int fl = lane & 31, h = lane >> 5;
            const float* sp = Hs + fl * ldh + 4 * h;
            float dsum = 0.f;
            bool waited = false;
            auto gate_ready = [&]() {  // the tile's H block: waited for once, behind the first W fragments of the tile
                if (!waited) rp_await(ready, (unsigned)(j + 1), a.stop);
                waited = true;
            };
            for (int hf = 0; hf < 2; ++hf) {
                const int u = 2 * j + hf, phi = 8 * hf + w;
                if (phi + NA < a.nf) {
                    f32x16 acc[2] = {zero16(), zero16()};
                    const int so[2] = {phi * rp * 128, (phi + NA) * rp * 128};
                    contract_shared_buf<2>(acc, rsw, lane * 16, so, sp, a.nqk, gate_ready);
                    rp_await(vready, (unsigned)(u + 1), a.stop);  // the V half, and with it: the B team is through with ratio half u-2
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                    rp_p1_epilogue<OBJ>(a, acc[1], Rs, phi + NA, t0, lane, dsum);
                } else if (phi < a.nf) {
                    f32x16 acc[1] = {zero16()};
                    const int so[1] = {phi * rp * 128};
                    contract_shared_buf<1>(acc, rsw, lane * 16, so, sp, a.nqk, gate_ready);
                    rp_await(vready, (unsigned)(u + 1), a.stop);
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                } else {
                    gate_ready();
                    rp_await(vready, (unsigned)(u + 1), a.stop);
                }
                rp_post(p1, w, (unsigned)(u + 1), lane);
            }
            if (OBJ) acc_div += (double)dsum;
            if (a.xr) {  // the extra row: after the last epilogue, the B team needs it for its very last k-block only
                hstep_p1_xrow<NA, 1, BM_KL, OBJ>(a, Hs, Rs, wxs, t0, w, lane, true, acc_div);
                rp_post(xdone, w, (unsigned)(j + 1), lane);
            }
        };
        for (int j = 0; j < (has_part ? nmy - 1 : nmy); ++j) a_item(j);
        if (has_part) {
            // the part (place ppos): its nfp row tiles sit in half php -- S = 4: row tile 4 pp + w, one per wave; S = 2: the
            // half's ordinary pairs (w, w + 4)
            const int j = ppos, t0 = ptile * Tt;
            float* Hs = lds + (j & 1) * hsz;
            const int fl = lane & 31, h = lane >> 5;
            const float* sp = Hs + fl * ldh + 4 * h;
            float dsum = 0.f;
            bool waited = false;
            auto gate_ready = [&]() {
                if (!waited) rp_await(ready, (unsigned)(j + 1), a.stop);
                waited = true;
            };
            for (int hf = 0; hf < 2; ++hf) {
                const int u = 2 * j + hf;
                if (hf == php && nfp == 8) {
                    const int phi = 8 * hf + w;
                    f32x16 acc[2] = {zero16(), zero16()};
                    const int so[2] = {phi * rp * 128, (phi + NA) * rp * 128};
                    contract_shared_buf<2>(acc, rsw, lane * 16, so, sp, a.nqk, gate_ready);
                    rp_await(vready, (unsigned)(u + 1), a.stop);
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                    rp_p1_epilogue<OBJ>(a, acc[1], Rs, phi + NA, t0, lane, dsum);
                } else if (hf == php) {
                    const int phi = pp * nfp + w;
                    f32x16 acc[1] = {zero16()};
                    const int so[1] = {phi * rp * 128};
                    contract_shared_buf<1>(acc, rsw, lane * 16, so, sp, a.nqk, gate_ready);
                    rp_await(vready, (unsigned)(u + 1), a.stop);
                    rp_p1_epilogue<OBJ>(a, acc[0], Rs, phi, t0, lane, dsum);
                } else {
                    gate_ready();
                    rp_await(vready, (unsigned)(u + 1), a.stop);
                }
                rp_post(p1, w, (unsigned)(u + 1), lane);
            }
            if (OBJ) acc_div += (double)dsum;
            if (own_x) hstep_p1_xrow<NA, 1, BM_KL, OBJ>(a, Hs, Rs, wxs, t0, w, lane, true, acc_div);
            if (a.xr) rp_post(xdone, w, (unsigned)(j + 1), lane);  // (every place posts it: the slots are progress numbers)
            a_item(nmy);
        }
    } else {
        // ================================ B team: P2 =================================================
        const int wb = w - NA;
        f32x4 dp0[4], dp1[4];
        if (wb < a.nk) {
            rp_p2_consts(a, wb, lane, dp0);
            if (wb + NB < a.nk) rp_p2_consts(a, wb + NB, lane, dp1);
        }
        unsigned lx_seq = 0;  // LXH: tiles this wave has put through the partial buffers
        auto b_item = [&](const int j) {  // P2 of the whole tile in place j of the list
            const int t0 = tile_of(j) * Tt;
            float* Hs = lds + (j & 1) * hsz;
            float shsum = 0.f;
            auto g0 = [&]() { rp_await(p1, (unsigned)(2 * j + 1), a.stop); };
            auto g1 = [&]() { rp_await(p1, (unsigned)(2 * j + 2), a.stop); };
            auto gx = [&]() { rp_await(xdone, (unsigned)(j + 1), a.stop); };
            if constexpr (LXH == 2) {
                const int fl = lane & 31, h = lane >> 5;
                const float* sp = Rs + fl * ldr + 4 * h;
                const int pr = wb >> 1, sh = wb & 1, qb = 16 * sh;
                const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                f32x16 acc[3] = {zero16(), zero16(), zero16()};
                int so[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) so[i] = (3 * pr + i) * a.Fq * 128 + qb * 1024;
                f32x4 gl = {0.f, 0.f, 0.f, 0.f};  // leftover column 192 + 4 pr + c for frame fl, partial over this wave's k-blocks / lane half
                const int voff_l = (h * 128 + (4 * pr + (lane & 3)) * 4) * 4;
                int so_l = 6 * a.Fq * 128 + qb * 1024;
                contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, so_l, sp + 8 * qb, 16, g0);
                rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
#pragma unroll
                for (int i = 0; i < 3; ++i) so[i] += 32 * 1024;
                so_l += 32 * 1024;
                contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, so_l, sp + 8 * (32 + qb), 16, g1);
                if (sh == 1 && a.xr) {  // the extra row's k-block: the pair's second wave
#pragma unroll
                    for (int i = 0; i < 3; ++i) so[i] = (3 * pr + i) * a.Fq * 128 + 64 * 1024;
                    contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, 6 * a.Fq * 128 + 64 * 1024, sp + 8 * 64, 1, gx);
                }
                rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
                // what the partner finishes -> LDS (once every wave has read the previous tile's): its whole tile (slots 0..3) and its
                // half of the pair's third tile (slots 4, 5: register groups 2 sh' .. 2 sh' + 1 of the PARTNER, sh' = 1 - sh)
                rp_await(rdone, lx_seq, a.stop);
                float* pw = Ps + wb * 6 * 256 + lane * 4;
                auto put = [&](int slot, const f32x16& x, int g) {
                    const f32x4 o = {x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(pw + slot * 256) = o;
                };
                if (sh == 0) {
                    put(0, acc[1], 0); put(1, acc[1], 1); put(2, acc[1], 2); put(3, acc[1], 3);
                    put(4, acc[2], 2); put(5, acc[2], 3);
                } else {
                    put(0, acc[0], 0); put(1, acc[0], 1); put(2, acc[0], 2); put(3, acc[0], 3);
                    put(4, acc[2], 0); put(5, acc[2], 1);
                }
                *reinterpret_cast<f32x4*>(Pl + (wb * 64 + lane) * 4) = gl;
                ++lx_seq;
                rp_post(wdone, wb, lx_seq, lane);
                rp_await(wdone, lx_seq, a.stop);
                const float* pq = Ps + (wb ^ 1) * 6 * 256 + lane * 4;  // the partner's slots
                f32x4 in[6];
#pragma unroll
                for (int q6 = 0; q6 < 6; ++q6) in[q6] = *reinterpret_cast<const f32x4*>(pq + q6 * 256);
                f32x16 t6 = zero16();
                if (wb == 0) {  // leftover columns: group h of this lane half = the sum over that pair's two waves and both row halves
                    const float* pl0 = Pl + (2 * h) * 256 + fl * 4;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(pl0), x1 = *reinterpret_cast<const f32x4*>(pl0 + 128);
                    const f32x4 x2 = *reinterpret_cast<const f32x4*>(pl0 + 256), x3 = *reinterpret_cast<const f32x4*>(pl0 + 384);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t6[e] = (x0[e] + x1[e]) + (x2[e] + x3[e]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rp_post(rdone, wb, lx_seq, lane);
                // own partial + the partner's, in wave order (s = 0 first), so that both orders of arrival give the same bits
                if (sh == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[0][4 * g + e] += in[g][e];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[2][4 * g + e] += in[4 + g][e];
                    rh_cut_epilogue<OBJ, 0, 4>(a, acc[0], Hs, 3 * pr, t0, lane, rdph, shsum);
                    rh_cut_epilogue<OBJ, 0, 2>(a, acc[2], Hs, 3 * pr + 2, t0, lane, rdph, shsum);
                    if (wb == 0) rh_cut_epilogue<OBJ, 0, 1>(a, t6, Hs, 6, t0, lane, rdph, shsum);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[1][4 * g + e] = in[g][e] + acc[1][4 * g + e];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[2][8 + 4 * g + e] = in[4 + g][e] + acc[2][8 + 4 * g + e];
                    rh_cut_epilogue<OBJ, 0, 4>(a, acc[1], Hs, 3 * pr + 1, t0, lane, rdph, shsum);
                    rh_cut_epilogue<OBJ, 2, 4>(a, acc[2], Hs, 3 * pr + 2, t0, lane, rdph, shsum);
                }
                if (OBJ) acc_sh += (double)shsum;
                rp_post(p2done, wb, (unsigned)(j + 1), lane);
            } else if constexpr (LXH == 1) {
                const int fl = lane & 31, h = lane >> 5;
                const float* sp = Rs + fl * ldr + 4 * h;
                // k-blocks [8 wb, 8 wb + 8) of EACH ratio half (a wave with all its blocks in one half would leave its SIMD idle
                // through the other half's phase and crowd it in its own: 0.150 -> 0.167 ms), + the extra row's for the last wave
                const int qb = 8 * wb;
                const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
                f32x16 acc[3] = {zero16(), zero16(), zero16()};
                int so[3] = {qb * 1024, a.Fq * 128 + qb * 1024, 2 * a.Fq * 128 + qb * 1024};
                // the leftover columns 96 .. 99 ride along as 4x4x1 MFMAs on the same ratio fragments (contract_shared_buf_lx):
                // gl[c] = column 96 + c for frame fl, partial over this wave's k-blocks and this lane half's rows
                f32x4 gl = {0.f, 0.f, 0.f, 0.f};
                const int voff_l = (h * 128 + (lane & 3) * 4) * 4;
                int so_l = 3 * a.Fq * 128 + qb * 1024;
                contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, so_l, sp + 8 * qb, 8, g0);
                rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
#pragma unroll
                for (int i = 0; i < 3; ++i) so[i] += 32 * 1024;
                so_l += 32 * 1024;
                contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, so_l, sp + 8 * (32 + qb), 8, g1);
                if (wb == 3 && a.xr) {  // the extra row's k-block (rows Fm+1 .. Fm+7 of the ratio image and of W are zero)
#pragma unroll
                    for (int i = 0; i < 3; ++i) so[i] = i * a.Fq * 128 + 64 * 1024;
                    contract_shared_buf_lx<3>(acc, gl, rsk, lane * 16, so, voff_l, 3 * a.Fq * 128 + 64 * 1024, sp + 8 * 64, 1, gx);
                }
                rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
                // partial tiles -> LDS, once every wave has read the previous tile's
                rp_await(rdone, lx_seq, a.stop);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(Ps + ((wb * 3 + i) * 4 + g) * 256 + lane * 4) = o;
                    }
                *reinterpret_cast<f32x4*>(Pl + (wb * 64 + lane) * 4) = gl;
                ++lx_seq;
                rp_post(wdone, wb, lx_seq, lane);
                rp_await(wdone, lx_seq, a.stop);
                f32x16 tot = zero16();
                if (wb < 3) {
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(Ps + ((ww * 3 + wb) * 4 + g) * 256 + lane * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) tot[4 * g + e] += x[e];
                        }
                } else if (h == 0) {  // columns 96 .. 99 = this column tile's registers 0 .. 3 of the lanes h = 0
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(Pl + (ww * 64 + fl) * 4);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(Pl + (ww * 64 + 32 + fl) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[e] += x0[e] + x1[e];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                rp_post(rdone, wb, lx_seq, lane);
                rp_p2_epilogue<OBJ>(a, tot, Hs, wb, t0, lane, dp0, shsum);
                if (OBJ) acc_sh += (double)shsum;
                rp_post(p2done, wb, (unsigned)(j + 1), lane);
            } else {
                if (wb + NB < a.nk) rh_p2_tile<2, OBJ>(a, Hs, Rs, wb, t0, lane, wb, j, bdone, dp0, dp1, shsum, g0, g1, gx);
                else if (wb < a.nk) rh_p2_tile<1, OBJ>(a, Hs, Rs, wb, t0, lane, wb, j, bdone, dp0, dp1, shsum, g0, g1, gx);
                else {  // a wave without a column tile keeps step (the slots are progress numbers)
                    g0();
                    rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
                    g1();
                    if (a.xr) gx();
                    rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
                }
                if (OBJ) acc_sh += (double)shsum;
                rp_post(p2done, wb, (unsigned)(j + 1), lane);
            }
        };
        for (int j = 0; j < (has_part ? nmy - 1 : nmy); ++j) b_item(j);
        if (has_part) {
            // the part: W^T*ratio over its own k-blocks [4 pp nfp, 4 (pp+1) nfp) (+ the extra row's for the last part) ->
            // partial numerator, fragment order, agent-scope stores (as rp_part_p2)
            const int j = ppos, fl = lane & 31, h = lane >> 5;
            const float* sp = Rs + fl * ldr + 4 * h;
            const int qb = 4 * pp * nfp, nqp = 4 * nfp;
            auto gp = [&]() { rp_await(p1, (unsigned)(2 * j + php + 1), a.stop); };
            auto gx = [&]() { rp_await(xdone, (unsigned)(j + 1), a.stop); };
            const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
            const __amdgpu_buffer_rsrc_t rsp_ = __builtin_amdgcn_make_buffer_rsrc(a.part_buf + (size_t)unit * Tt * rp, 0, Tt * rp * 4, 0x00020000);
            if (php == 1) rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);  // (nobody reads the part's other ratio half)
            if (wb < a.nk) {
                const bool two = wb + NB < a.nk;
                const int kap1 = two ? wb + NB : wb;
                f32x16 acc[2] = {zero16(), zero16()};
                int so[2] = {wb * a.Fq * 128 + qb * 1024, kap1 * a.Fq * 128 + qb * 1024};
                contract_shared_buf<2>(acc, rsk, lane * 16, so, sp + 8 * qb, nqp, gp);
                if (own_x) {
                    so[0] = wb * a.Fq * 128 + (a.Fm / 8) * 1024;
                    so[1] = kap1 * a.Fq * 128 + (a.Fm / 8) * 1024;
                    contract_shared_buf<2>(acc, rsk, lane * 16, so, sp + a.Fm, 1, gx);
                }
                rp_await(p1, (unsigned)(2 * j + 2), a.stop);  // (the A team is through with both units of the place)
                if (a.xr) gx();
                if (php == 0) rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
                rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c == 1 && !two) break;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = {acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]};
                        buf_store_b128<kAuxSC1>(rsp_, lane * 16, ((c ? kap1 : wb) * 1024 + g * 256) * 4, o);
                    }
                }
            } else {
                rp_await(p1, (unsigned)(2 * j + 2), a.stop);
                if (a.xr) gx();
                if (php == 0) rp_post(bdone, wb, (unsigned)(2 * j + 1), lane);
                rp_post(bdone, wb, (unsigned)(2 * j + 2), lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial is at the coherence point before this wave reports
            rp_post(p2done, wb, (unsigned)(j + 1), lane);
            b_item(nmy);
        }
    }
    if (has_part) acc_sh += rp_part_finish<OBJ>(a, lds + (ppos & 1) * hsz, plast, ptile);

    if (OBJ) {
        // deterministic workgroup reduction of the two fp64 partial sums (as k_hstep)
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);  // [2][NTHR]
        red[threadIdx.x] = acc_div;
        red[NTHR + threadIdx.x] = acc_sh;
        __syncthreads();
        if ((int)threadIdx.x + 512 < NTHR) {
            red[threadIdx.x] += red[threadIdx.x + 512];
            red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + 512];
        }
        __syncthreads();
        for (int s = 256; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                red[threadIdx.x] += red[threadIdx.x + s];
                red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) obj_partial_out(a, blockIdx.x, red[0], red[NTHR]);
    }
}

// ============================================================================================
// k_hsolve_small: the WHOLE H-only solve of src/sparse_nmf.m:186-286 for T <= 32 frames in ONE
// launch by ONE workgroup: the online separation call (src/bnmf_sep_event_RT_IS16.m:138-154,
// T = 1, W = [B_x, B_d] fixed) is latency-bound -- ~27 iterations of two GEMV-sized products --
// so per-iteration kernel launches and host-side convergence polling would dominate.  H lives in
// LDS across the iterations; the objective sums, the convergence test (:272-284) and the
// objective vectors are produced on the device; W fragments stream from L2.
// Iteration j: P1 on H_{j-1} gives cost_{j-1} (tested BEFORE H moves, so a stop leaves H_{j-1}).
// ============================================================================================
struct SmallArgs {
    int max_iter;
    int cost_check;
    double conv_eps;
    double* divh;   // [gridDim.x][max_iter]
    double* costh;  // [gridDim.x][max_iter]
    DevState* st;   // [gridDim.x]
    int tps;        // frames per solve (<= 32); workgroup b solves columns [b*tps, (b+1)*tps)
    // k_hsolve_frame only: also emit the two reconstructions the online post-filter needs
    // (src/bnmf_sep_event_RT_IS16.m:158-202), [gridDim.x][2][F]: B(:,1:Rx)*A(1:Rx) and B(:,Rx+1:r)*A(Rx+1:r),
    // with B = W .* wn' (the un-normalised dictionary) -- the block is already in registers here.
    float* recon;
    const double* wn;
    int Rx;
};

// One workgroup = one independent solve (gridDim.x solves run concurrently: the batched online
// stream, one frame per CU).
template <int BM, bool OBJ>
__global__ __launch_bounds__(512, 2) void k_hsolve_small(StepArgs a, SmallArgs sa) {
    constexpr int NW = 8, NT = 1, NTHR = 512, Tt = 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {   // re-base everything on this workgroup's columns
        const size_t c0 = (size_t)blockIdx.x * sa.tps;
        a.V += c0 * a.Fp;
        a.Hin += c0 * a.rp;
        a.Hout += c0 * a.rp;
        if (a.S) a.S += c0 * a.rp;
        a.T = sa.tps;
        sa.divh += (size_t)blockIdx.x * sa.max_iter;
        sa.costh += (size_t)blockIdx.x * sa.max_iter;
        sa.st += blockIdx.x;
    }
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, ldh = a.ldh, ldr = a.ldr;
    float* Hs = lds;                 // [32][ldh]
    float* Rs = Hs + Tt * ldh;       // [32][ldr] ratio / den / num image
    float* wxs = Rs + Tt * ldr;      // [rp]
    double* red = reinterpret_cast<double*>(wxs + ((rp + 3) & ~3));  // [2][512]
    if (a.xr)
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
    for (int i = threadIdx.x; i < Tt * ldr; i += NTHR) Rs[i] = 0.f;
    stage_in<NTHR>(a.Hin, Hs, Tt, rp, ldh, threadIdx.x);
    __syncthreads();
    constexpr int NPASS = (BM == BM_KL) ? 1 : 2;
    double last_cost = 0.0;
    int n_rec = 0;
    bool stopped = false;
    for (int j = 1; j <= sa.max_iter + 1; ++j) {
        if (j > sa.max_iter && !(OBJ && sa.max_iter >= 1)) break;
        const bool upd = j <= sa.max_iter;
        double acc_div = 0.0, acc_sh = 0.0;
        hstep_p1<NW, NT, BM, OBJ, true>(a, Hs, Rs, wxs, 0, w, lane, upd, acc_div);
        if (OBJ && j > 1) {
            // sum(S .* H_{j-1}) over the real entries (pad rows of H are zero)
            float sh = 0.f;
            for (int i = threadIdx.x; i < a.T * rp; i += NTHR) {
                const int t = i / rp, k = i - t * rp;
                const float sv = a.S ? a.S[(size_t)t * rp + k] : a.lamk[k];
                sh += sv * Hs[t * ldh + k];
            }
            acc_sh = (double)sh;
        }
        __syncthreads();
        if (OBJ && j > 1) {
            red[threadIdx.x] = acc_div;
            red[NTHR + threadIdx.x] = acc_sh;
            __syncthreads();
            for (int s2 = NTHR / 2; s2 > 0; s2 >>= 1) {
                if ((int)threadIdx.x < s2) {
                    red[threadIdx.x] += red[threadIdx.x + s2];
                    red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + s2];
                }
                __syncthreads();
            }
            const double div = red[0], cost = red[0] + red[NTHR];
            __syncthreads();  // everyone has read red[] before it is reused
            const int it = j - 1;
            bool stopnow = false;
            if (it > 1 && sa.conv_eps > 0.0) stopnow = fabs(cost - last_cost) / last_cost < sa.conv_eps;
            if (threadIdx.x == 0) {
                sa.divh[it - 1] = div;
                sa.costh[it - 1] = cost;
            }
            n_rec = it;
            last_cost = cost;
            if (stopnow) {
                stopped = true;
                break;
            }
        }
        if (!upd) break;
        double dummy = 0.0;
        hstep_p2<NW, NT, BM, false>(a, Hs, Rs, 0, w, lane, 0, dummy);
        if (NPASS == 2) {
            __syncthreads();
            hstep_den_to_num<NW, NT, BM>(a, Rs, 0, w, lane);
            __syncthreads();
            hstep_p2<NW, NT, BM, false>(a, Hs, Rs, 0, w, lane, 1, dummy);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sa.st->n_iter = n_rec;
        sa.st->stop = stopped ? 1 : 0;
    }
    __syncthreads();
    stage_out<NTHR>(a.Hout, Hs, sa.tps, rp, ldh, threadIdx.x);  // only this solve's columns
}

// ============================================================================================
// k_hsolve_frame: the same whole-solve-in-one-launch as k_hsolve_small for ONE frame per solve
// (tps = 1: the real online call, src/bnmf_sep_event_RT_IS16.m:148-154).  A single column wastes
// 31/32 of every MFMA tile, which made k_hsolve_small matrix-pipe bound on one CU (2 x 1792 MFMAs
// = 37 us per iteration at 513 x 200).  Here the dictionary lives in REGISTERS for the whole solve
// and both products are plain fp32 FMAs with no padding:
//   thread (kb = wave 0..7, fb = lane 0..63) holds the FB x KB block W[FB*fb + i][KB*kb + kk]
//   P1  lam_part[i] = sum_kk W[i][kk] h[kk]      -> LDS [8][Fm] -> 8-way sum, floor, ratio, div term
//   P2  d_part[kk]  = sum_i  W[i][kk] ratio[i]   -> LDS [64][8*KB] -> 64-way sum, H update
// i.e. 2*FB*KB FMAs per thread per iteration and four barriers; W is read from L2 once per
// solve instead of twice per iteration.  The extra row (F = 64*FB + 1) is a wave-level dot product.
// Loop semantics, objective recording and the stop test are those of k_hsolve_small.
// ============================================================================================
// P1 of k_hsolve_frame: this thread's FB x KB register block times the wave's KB entries of hv -> the wave's partial Lam
// (lamp[kb][Fm]) and, with the extra row, its partial of W[Fm,:] * hv.  Shared by the solve's loop and the reconstruction passes.
template <int FB, int KB>
__device__ __forceinline__ void hsolve_frame_p1(const f32x2 (&wr)[FB / 2][KB], const float* hv, const float* wxs, float* lamp, float* xlam,
                                                int Fm, bool xr, int kb, int fb, int lane) {
    f32x2 lp[FB / 2];
#pragma unroll
    for (int i2 = 0; i2 < FB / 2; ++i2) lp[i2] = f32x2{0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
        const float hk = hv[kb * KB + kk];  // wave-uniform address: LDS broadcast
        const f32x2 hk2 = f32x2{hk, hk};
#pragma unroll
        for (int i2 = 0; i2 < FB / 2; ++i2) lp[i2] = __builtin_elementwise_fma(wr[i2][kk], hk2, lp[i2]);
    }
#pragma unroll
    for (int i2 = 0; i2 < FB / 2; ++i2) *reinterpret_cast<f32x2*>(lamp + kb * Fm + fb * FB + 2 * i2) = lp[i2];
    if (xr) {  // extra row: this wave's KB columns of W[Fm,:] * hv, lanes 0..KB-1
        static_assert(KB <= 64, "one lane per column");
        const float px = wave_sum_f(lane < KB ? wxs[kb * KB + lane] * hv[kb * KB + lane] : 0.f);
        if (lane == 0) xlam[kb] = px;
    }
}

template <int FB, int KB, int BM, bool OBJ, bool RECON = false>
__global__ __launch_bounds__(512, 2) void k_hsolve_frame(StepArgs a, SmallArgs sa, const float* __restrict__ Wcf) {
    constexpr int NTHR = 512, RB = 8 * KB, LDP = RB + 1;  // LDP odd: the partial rows hit distinct banks
    constexpr int NPR = 16;  // partial rows of the W^T product after the quad pre-reduction
    constexpr int NV = (BM == BM_KL) ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {   // this workgroup's frame
        const size_t c0 = (size_t)blockIdx.x;
        a.V += c0 * a.Fp;
        a.Hin += c0 * a.rp;
        a.Hout += c0 * a.rp;
        if (a.S) a.S += c0 * a.rp;
        sa.divh += (size_t)blockIdx.x * sa.max_iter;
        sa.costh += (size_t)blockIdx.x * sa.max_iter;
        sa.st += blockIdx.x;
    }
    const int tid = threadIdx.x, lane = tid & 63, kb = tid >> 6, fb = lane;
    const int Fm = 64 * FB;              // rows held in registers (>= F - xr; rows >= F are zero)
    const int rp = a.rp, F = a.F;
    const bool xr = F > Fm;              // one extra row, index Fm
    double* red = reinterpret_cast<double*>(lds);  // [16] objective partial sums
    float* xlam = lds + 32;              // [8]      per-wave partials of the extra row's Lam
    float* hs = lds + 40;                // [RB]     activations (k >= r stay zero)
    float* vs = hs + RB;                 // [Fm + 4] the frame (floored V)
    float* va = vs + Fm + 4;             // [Fm + 4] ratio (KL) / num vector
    float* vb = va + Fm + 4;             // [Fm + 4] den vector (beta != 1)
    float* wxs = vb + Fm + 4;            // [RB]     extra row of W
    float* sps = wxs + RB;               // [RB]     sparsity weights of this frame
    float* dps = sps + RB;               // [RB]     KL: max(colsum + S, flr)
    float* lamp = dps + RB;              // [8][Fm]  P1 partials
    float* dmp = lamp + 8 * Fm;          // [NV][NPR][LDP] P2 partials

    // ---- one-time loads -----------------------------------------------------------------------
    // rows are held in PAIRS (2*i2, 2*i2+1) so that both products map onto v_pk_fma_f32 with the SAME
    // register pairing (a second pairing would make the compiler keep two copies of the block)
    static_assert(FB % 2 == 0, "FB must be even");
    f32x2 wr[FB / 2][KB];
    if (Fm <= a.Fp && (a.Fp & 3) == 0 && (FB & 3) == 0) {
        // whole register rows lie inside the padded column (rows F..Fp-1 of Wcf are zero): 16-byte loads
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int k = kb * KB + kk;
            const f32x4* src = reinterpret_cast<const f32x4*>(Wcf + (size_t)(k < rp ? k : 0) * a.Fp + fb * FB);
#pragma unroll
            for (int i4 = 0; i4 < FB / 4; ++i4) {
                f32x4 v = src[i4];
                if (k >= rp) v = f32x4{0.f, 0.f, 0.f, 0.f};
                wr[2 * i4][kk] = f32x2{v[0], v[1]};
                wr[2 * i4 + 1][kk] = f32x2{v[2], v[3]};
            }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int k = kb * KB + kk;
#pragma unroll
            for (int i2 = 0; i2 < FB / 2; ++i2) {
                const int f = fb * FB + 2 * i2;
                const bool ok = k < rp && f < Fm;
                wr[i2][kk].x = (ok && f < F) ? Wcf[(size_t)k * a.Fp + f] : 0.f;
                wr[i2][kk].y = (ok && f + 1 < F) ? Wcf[(size_t)k * a.Fp + f + 1] : 0.f;
            }
        }
    }
    for (int k = tid; k < RB; k += NTHR) {
        const bool in = k < rp;
        hs[k] = in ? a.Hin[k] : 0.f;
        wxs[k] = (in && xr) ? a.wx[k] : 0.f;
        const float sp = in ? (a.S ? a.S[k] : a.lamk[k]) : 0.f;
        sps[k] = sp;
        dps[k] = in ? (a.S ? fmaxf(a.colsum[k] + sp, kFlr) : a.dphv[k]) : 1.f;
    }
    for (int f = tid; f < Fm + 4; f += NTHR) {
        vs[f] = f < F ? a.V[f] : 0.f;
        va[f] = 0.f;
        vb[f] = 0.f;
    }
    __syncthreads();

    double last_cost = 0.0;
    int n_rec = 0;
    bool stopped = false;
    for (int j = 1; j <= sa.max_iter + 1; ++j) {
        if (j > sa.max_iter && !(OBJ && sa.max_iter >= 1)) break;
        const bool upd = j <= sa.max_iter;
        // ---- P1: Lam = W * h ---------------------------------------------------------------------
        hsolve_frame_p1<FB, KB>(wr, hs, wxs, lamp, xlam, Fm, xr, kb, fb, lane);
        __syncthreads();
        float dterm = 0.f;
        for (int f = tid; f < Fm; f += NTHR) {
            float s = lamp[f];
#pragma unroll
            for (int q = 1; q < 8; ++q) s += lamp[q * Fm + f];
            const float lam = fmaxf(s, kFlr);
            const float v = vs[f];
            const bool real = f < F;
            if (OBJ) dterm += real ? div_term<BM>(v, lam, a.beta, a.inv_bb1) : 0.f;
            if (BM == BM_KL) {
                va[f] = real ? v * fast_rcp(lam) : 0.f;
            } else {
                const float den = den_of_lam<BM>(lam, a.beta);
                float lf = 1.f;  // lam^(beta-2) from den, as hstep_den_to_num
                if (BM != BM_EUC) lf = (a.beta == 0.f) ? den * den : fast_pow(den, (a.beta - 2.f) / (a.beta - 1.f));
                vb[f] = real ? den : 0.f;
                va[f] = real ? v * lf : 0.f;
            }
        }
        if (xr && tid == 0) {  // extra row: the 8 per-wave partials were formed in P1
            float s = xlam[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) s += xlam[q];
            const float lam = fmaxf(s, kFlr);
            const float v = vs[Fm];
            if (OBJ) dterm += div_term<BM>(v, lam, a.beta, a.inv_bb1);
            if (BM == BM_KL) {
                va[Fm] = v * fast_rcp(lam);
            } else {
                const float den = den_of_lam<BM>(lam, a.beta);
                float lf = 1.f;
                if (BM != BM_EUC) lf = (a.beta == 0.f) ? den * den : fast_pow(den, (a.beta - 2.f) / (a.beta - 1.f));
                vb[Fm] = den;
                va[Fm] = v * lf;
            }
        }
        if (OBJ && j > 1) {
            // cost_{j-1} = div(V, W*H_{j-1}) + sum(S .* H_{j-1}); fixed-order fp64 sums
            const float shf = lane < KB ? sps[kb * KB + lane] * hs[kb * KB + lane] : 0.f;  // this wave's columns
            const double dv = (double)wave_sum_f(dterm), sh = (double)wave_sum_f(shf);  // 64 terms in fp32, then fp64
            if (lane == 0) {
                red[kb] = dv;
                red[8 + kb] = sh;
            }
            __syncthreads();
            double div = 0.0, shs = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                div += red[q];
                shs += red[8 + q];
            }
            const double cost = div + shs;
            const int it = j - 1;
            bool stopnow = false;
            if (it > 1 && sa.conv_eps > 0.0) stopnow = fabs(cost - last_cost) / last_cost < sa.conv_eps;
            if (tid == 0) {
                sa.divh[it - 1] = div;
                sa.costh[it - 1] = cost;
            }
            n_rec = it;
            last_cost = cost;
            if (stopnow) {
                stopped = true;
                break;
            }
        } else {
            __syncthreads();
        }
        if (!upd) break;
        // ---- P2: W^T * ratio (and W^T * den for beta != 1) -----------------------------------------
        {
            f32x2 ra[FB / 2], rb[FB / 2];
#pragma unroll
            for (int i2 = 0; i2 < FB / 2; ++i2) {
                ra[i2] = *reinterpret_cast<const f32x2*>(va + fb * FB + 2 * i2);
                if (NV == 2) rb[i2] = *reinterpret_cast<const f32x2*>(vb + fb * FB + 2 * i2);
            }
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
                f32x2 pa = f32x2{0.f, 0.f}, pb = f32x2{0.f, 0.f};
#pragma unroll
                for (int i2 = 0; i2 < FB / 2; ++i2) {
                    pa = __builtin_elementwise_fma(wr[i2][kk], ra[i2], pa);
                    if (NV == 2) pb = __builtin_elementwise_fma(wr[i2][kk], rb[i2], pb);
                }
                // pre-reduce over the 4 row-blocks of a lane quad on the DPP path: 16 partial rows instead of 64
                const float qa = quad_sum_f(pa.x + pa.y);
                if ((lane & 3) == 0) dmp[(fb >> 2) * LDP + kb * KB + kk] = qa;
                if (NV == 2) {
                    const float qb = quad_sum_f(pb.x + pb.y);
                    if ((lane & 3) == 0) dmp[NPR * LDP + (fb >> 2) * LDP + kb * KB + kk] = qb;
                }
            }
        }
        __syncthreads();
        // Each wave finishes ITS OWN KB columns (lane <-> column): P1, the extra-row partial and the sparsity
        // sum of the next iteration read exactly those entries of h back, from the same wave, so no workgroup
        // barrier is needed here (LDS operations of one wave complete in order).
        if (lane < KB) {
            const int k = kb * KB + lane;
            float sa_ = 0.f, sb_ = 0.f;
#pragma unroll
            for (int q = 0; q < NPR; ++q) {
                sa_ += dmp[q * LDP + k];
                if (NV == 2) sb_ += dmp[NPR * LDP + q * LDP + k];
            }
            if (xr) {
                sa_ = fmaf(wxs[k], va[Fm], sa_);
                if (NV == 2) sb_ = fmaf(wxs[k], vb[Fm], sb_);
            }
            const float ho = hs[k];
            float hn;
            if (BM == BM_KL) hn = ho * sa_ * fast_rcp(dps[k]);                 // H .* dmh ./ dph  (:194-195)
            else hn = ho * fast_rcp(fmaxf(sb_ + sps[k], kFlr)) * sa_;          // :196-205
            hs[k] = k < a.rp ? hn : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid == 0) {
        sa.st->n_iter = n_rec;
        sa.st->stop = stopped ? 1 : 0;
    }
    for (int k = tid; k < rp; k += NTHR) a.Hout[k] = k < RB ? hs[k] : 0.f;
    if (RECON) {
        // Xm_hat_sum = B_x*A_x and Dm_hat_sum = B_d*A_d: two more passes of P1 ITSELF (the same inlined code, so the register
        // block keeps the allocation of the solve's loop: with the class select inside the unrolled products this variant --
        // the one the online loop launches -- had 32 spilled VGPRs and 36 B of scratch) over two masked copies of h .* wn,
        // the first with the columns k < Rx, the second with the rest; they live in sps / dps, which are dead by now.
        float* out = sa.recon + (size_t)blockIdx.x * 2 * F;
        __syncthreads();  // H has been copied out and nobody reads sps / dps any more
        if (lane < KB) {
            const int k = kb * KB + lane;
            const float hw = k < a.rp ? (float)((double)hs[k] * sa.wn[k]) : 0.f;
            sps[k] = k < sa.Rx ? hw : 0.f;
            dps[k] = k < sa.Rx ? 0.f : hw;
        }
        __builtin_amdgcn_wave_barrier();
        for (int part = 0; part < 2; ++part) {
            hsolve_frame_p1<FB, KB>(wr, part ? dps : sps, wxs, lamp, xlam, Fm, xr, kb, fb, lane);
            __syncthreads();
            for (int f = tid; f < Fm && f < F; f += NTHR) {
                float s = lamp[f];
#pragma unroll
                for (int q = 1; q < 8; ++q) s += lamp[q * Fm + f];
                out[part * F + f] = s;
            }
            if (xr && tid == 0) {
                float s = xlam[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) s += xlam[q];
                out[part * F + Fm] = s;
            }
            __syncthreads();
        }
    }
}

// ============================================================================================
// k_wstats: the T-reductions of the W half-step, src/sparse_nmf.m:215-239.
//   WM 0 (KL)      : slab = (V ./ Lam') * H^T        and s = rowsum(H)
//   WM 1 (P)       : slab = Lam'^(beta-1) * H^T
//   WM 2 (Q)       : slab = (V .* Lam'^(beta-2)) * H^T
//   WM 3 (Q, EUC)  : slab = V * H^T                  (no Lam' needed)
// grid = (n_chunks, n_fgroups); wave w of a workgroup owns f-tile phi = fgroup*NWB + w and keeps
// its NK x (32x32) accumulators in registers over the whole chunk of frames.
// OBJ: additionally sums the divergence of (W, H) (W-only mode: Lam' IS the objective's Lam).
// ============================================================================================
// NL > 0: NL loader waves stage tile i+1 into the second LDS buffer (and keep the row sums of H)
// while the NWB consumer waves -- ONE per SIMD, so nothing contends for the matrix pipe -- run
// P3/P4 on tile i; one barrier per tile.  NL = 0: consumers stage synchronously (two workgroups per
// CU when the accumulators leave room).
// TT = frames per tile (32; 16 = narrow tiles for shapes whose 32-frame H image does not fit the LDS: lanes / rows past
// the tile carry duplicates that are masked out of the ratio, P4 contracts over TT frames only, P3 wastes half its MFMAs).
// LX: 16-byte groups (1 or 2) of statistics columns past the last FULL 32-column tile that are accumulated on the VALU
// instead of a nearly empty MFMA tile (r = 100: 4 columns, r = 200: 8); 0 = every column tile through the MFMAs.  A template
// parameter: as a run-time branch the extra code cost the 8+4-wave geometry 113 spilled VGPRs.
// TIL: the consumer teams of StepArgs::til.  A compile-time switch, and a kernel of its own (k_wstats_teams): the eight-consumer
// geometries of the F = 513 shapes sit at their 168-VGPR limit, and the team indices as run-time values in their tile loop cost them
// 19 spilled VGPRs (the reference's 513 x 72000 r = 100 W step 0.1350 -> 0.1407 ms); profiles/r04_resources.csv is the check.
template <int NK, int NWB, int NL, int WPS, int WM, int BM, bool OBJ, int TT, int LX, bool TIL>
__device__ __forceinline__ void wstats_body(StepArgs a, int n_chunks, int mat_index, int n_mat) {
    static_assert(TT == 32 || (TT == 16 && NL == 0), "narrow tiles: 16 frames, synchronous staging");
    constexpr int NTHR = (NWB + NL) * 64;
    // Tile buffers: 1 without loader waves; with them 2, or 3 where the LDS has room (host: a.nbuf).  With two buffers the DMA
    // of tile i+1 can only be issued once the SLOWEST consumer has finished tile i-1 and must have landed before the fastest
    // finishes tile i: the consumers of a11 (513 x 72000, r = 100) spent 11.6 % of their time waiting for `ready` (phase stamps
    // of the diagnostic build).  With three the loaders run up to two tiles ahead and the wait disappears.
    const int NBUF = NL > 0 ? (a.nbuf == 3 ? 3 : 2) : 1;
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // V image: only the columns of this workgroup's row group are staged, so it is [TT][32 * NWB] (F = 513, four row
    // groups: 16 KiB instead of 64 -- what lets two tile buffers fit -- and the V block crosses L2 once per iteration, not
    // once per row group); the extra row's V values go to a small array of their own (vx)
    const int ldv = 32 * NWB;
    // NK = 4 (r <= 128) with full tiles: the H image's leading dimension is always 128 + 4 (host: ldhw), a COMPILE-TIME constant
    // here -- P4's sixteen row bases become one base + immediate offsets (15 VGPRs less: the LX variants of the 8+4-wave
    // geometry spilled 1..4 registers into scratch, which costs a kernel ~4 us per launch, and carried the adds per tile)
    // (NK = 8, r <= 256, one kappa-group: always 256 + 4 -- the host's ldhw and the kappa-group launches' 260)
    constexpr bool LDHC = (NK == 4 || NK == 8) && TT == 32;
    const int ldh = LDHC ? (NK == 4 ? 132 : 260) : a.ldh;
    const int bufsz = TT * (ldh + ldv);     // floats per buffer: Hs [TT][ldh] then Vs [TT][ldv]
    float* wxs = lds + NBUF * bufsz;        // [rp] extra row of W
    const int lane = threadIdx.x & 63, w = wave_index();
    const bool is_loader = NL > 0 && w >= NWB;
    const int fl = lane & 31, h = lane >> 5;
    const int rp = a.rp, Fp = a.Fp;
    // Row group and frame chunk of this workgroup.  Row groups of which only group 0 carries the extra row are not
    // equally expensive per tile (the VALU row costs ~14 % of a tile on the MFMA-issuing waves), so the host may give
    // them DIFFERENT numbers of frame chunks: a 1-D grid of n_chunks workgroups of group 0 followed by a.n_ch1 for each
    // further group.  The slabs of the chunks those groups do not have stay zero (set once at plan creation).
    int chunk = blockIdx.x, by = blockIdx.y, nch = n_chunks;
    if (a.n_ch1 > 0) {
        by = 0;
        if (chunk >= n_chunks) {
            const int c1 = chunk - n_chunks;
            by = 1 + c1 / a.n_ch1;
            chunk = c1 - (by - 1) * a.n_ch1;
            nch = a.n_ch1;
        }
    }
    const bool kc = WM == 3 && NK == 8 && a.kc != 0;  // kappa-compact H image (columns [256 z, 256 z + 256) at offset 0)
    const bool do_x = a.xr && by == 0 && (blockIdx.z == 0 || kc);  // extra row: one f-group only (its k range: see kc)
    constexpr int CPW = TT / NWB;  // columns of the extra-row dot product per wave
    static_assert(CPW % 4 == 0, "the extra row takes 4 frames x 16 lanes at a time");
    // 256-column pieces of an H row this geometry can have: NK <= 8 is rp <= 256 (one kappa-group, n_kg = 1 on the host)
    constexpr int NPC = NK <= 8 ? 1 : 4;
    // extra row of the slab: lane <-> k = 256*(i/4) + 4*lane + i%4  (rp <= 256 * NPC).  One 256-column piece (NK <= 8):
    // the four partial sums live in LDS, one row per consumer wave (gxs; a read-modify-write per tile) -- as registers
    // they were what the 168-VGPR geometries spilled, and a scratch round trip per tile on the waves with the extra row
    // paces the whole workgroup.  NK = 16: registers.
    constexpr bool GXL = NPC == 1;
    const int gxw = a.rp < 256 ? a.rp : 256;
    float gx[4 * NPC];
#pragma unroll
    for (int i = 0; i < 4 * NPC; ++i) gx[i] = 0.f;
    if (do_x && WM != 3)
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
    if (GXL && do_x)  // (the consumers' partial extra rows start at zero; a barrier separates this from their first use)
        for (int k = threadIdx.x; k < NWB * gxw; k += NTHR) (lds + NBUF * TT * (ldh + 32 * NWB) + a.rp + 128)[k] = 0.f;
    // F = 64 (a Mel spectrogram) has two row tiles: two of four consumer waves -- two of four SIMDs -- would carry the whole
    // chunk.  With a.til > 1 the consumers form til TEAMS of nfl = NWB / til waves; wave w is row tile w % nfl of team w / nfl.
    int til = 1, nfl = NWB, wl = w, tph = 0;
    if constexpr (TIL) {
        til = (NL > 0 && a.til > 1) ? a.til : 1;
        nfl = NWB / til;
        wl = til > 1 ? w % nfl : w;
        tph = til > 1 ? w / nfl : 0;
    }
    const int phi = by * NWB + (TIL ? wl : w);
    const int fc = (TIL ? wl : w) * 32 + fl;  // this lane's column of the staged V image
    const bool active = !is_loader && phi < a.nf;
    const int kap_base = blockIdx.z * NK;  // kappa-group (r > 32*NK: P3 is recomputed per group)
    const bool do_obj = OBJ && blockIdx.z == 0;
    // contiguous, balanced range of 32-frame tiles for this chunk
    const int tb = (int)(((long long)a.n_tiles * chunk) / nch);
    const int te = (int)(((long long)a.n_tiles * (chunk + 1)) / nch);

    f32x16 G[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) G[k] = zero16();
    // Leftover columns (LX): r = 100 is three full 32-column tiles + 4 columns, r = 200 six + 8 -- the reference's own
    // ranks (settings/initial_setting_SNMF_NAT.m:48-49).  A fourth / seventh MFMA tile would be 87 % / 75 % padding; the
    // few columns are accumulated here instead: this lane holds ratio[f, t] for its 16 frames, so
    // gl[j] += sum_i R[i] * H[k0 + j, t_i]  (one broadcast ds_read_b128 per frame), the two lane halves are added at the end.
    const int nkm = LX ? a.nk - 1 : a.nk;  // column tiles that go through the MFMAs
    float gl[LX ? 4 * LX : 1];
#pragma unroll
    for (int j = 0; j < (LX ? 4 * LX : 1); ++j) gl[j] = 0.f;
    // row sums of H: kept by the staging threads (loaders, or everybody when NL = 0);
    // thread <-> k = sid + j*NST, j < 4 (rp <= 4*NST checked on the host)
    constexpr int NST = NL > 0 ? NL * 64 : NWB * 64;
    const int sid = NL > 0 ? (int)threadIdx.x - NWB * 64 : (int)threadIdx.x;
    const bool do_s = WM == 0 && by == 0 && blockIdx.z == 0;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f};
    double acc_div = 0.0;

    // NL > 0: no workgroup barrier inside the tile loop.  The consumer waves never exchange data (each owns its rows
    // of G), so all they need is "tile i is staged" -- an LDS counter the loader waves bump (`ready`, NL per tile) --
    // and all the loaders need before they refill a buffer is "every consumer is done with tile i-1" (`done`, NWB
    // per tile).  Waits are bounded spins: a lost signal raises DevState::fault (the host then fails the call), never a hang.
    // Per-wave progress slots (rp_post / rp_await): a single counter bumped by all four waves is a TOTAL, and a consumer
    // that drifts a tile ahead of a slow team-mate (nothing synchronises the consumers with each other) could have stood
    // in for it and let the loaders refill a buffer that wave was still reading.
    // NWB = 8 (narrow statistics, NK = 4: 64 accumulator VGPRs): TWO consumer waves per SIMD, each with its own row tile --
    // one wave's epilogue, extra row and wait for the tile then run beside the other's MFMA loops, which at rp <= 128 are
    // too short (128 MFMAs per tile) to amortise them, and the H tile is staged once per eight row tiles instead of four.
    static_assert(NL == 0 || (NL == 4 && (NWB == 4 || NWB == 8)), "progress slots: four per loader role, NWB consumers");
    unsigned* ready = reinterpret_cast<unsigned*>(wxs + rp);  // [4] loader waves: tiles staged
    unsigned* done = ready + 4;                               // [NWB] consumer waves: tiles finished
    float* vx = reinterpret_cast<float*>(done + NWB);         // [NBUF][32] V of the extra row, one value per frame (DMA loaders)
    float* gxs = wxs + rp + 128;                              // [NWB][gxw] partial extra row of the slab per consumer wave (GXL)
    // (the read-modify-write is inline assembly: as C++ stores into the LDS array inside the tile loop they alias every
    //  tile read as far as the compiler knows, and the 168-VGPR geometries went from 16 to 211 spilled registers)
    const unsigned gxa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(gxs + w * gxw + 4 * lane);
    // A consumer wave without a row tile (and without a share of the extra row) has nothing to do in the tile loop but wait
    // and report -- polling LDS beside the MFMA wave of its SIMD: its progress word is set to "done with every tile" instead.
    // (NWB = 4 geometries only: in the eight-consumer kernels of the F = 513 shapes the extra loop condition cost 0.7 %)
    const bool idle_c = NL > 0 && NWB == 4 && !is_loader && !active && !do_x;
    if (NL > 0) {
        if (threadIdx.x < 4 + NWB) ready[threadIdx.x] = 0u;
        __syncthreads();  // slots and wxs are set
        if constexpr (NWB == 4) {  // (the eight-consumer kernels sit at their register limit: this code as dead code cost them 176 spilled VGPRs)
            if (idle_c && lane == 0) done[w] = 0xffffffffu;
            __syncthreads();
        }
    }

    // The extra row of one tile (row group 0 only): ratio_x[t] for this wave's CPW columns, then gx[k] += ratio_x[t] * H[k,t].
    // It stays on the CONSUMER waves: moved to the loader waves -- which only get an instruction in where their SIMD's
    // MFMA wave stalls -- it delayed the staging of the next tile (k_wstats 0.249 -> 0.277 ms on C2).
    auto xrow_tile = [&](const float* xH, int xt0, int xw, const float* vxc) {
            // extra row: ratio_x[t] for this wave's CPW columns, then gx[k] += ratio_x[t] * H[k,t]
            float rxv[CPW];
            float dsum = 0.f;
#pragma unroll
            for (int c0 = 0; c0 < CPW; c0 += 4) {
                const int tl = xw * CPW + c0 + (lane >> 4);
                const int kl = lane & 15;
                const float* hrow = xH + tl * ldh;
                const int t = xt0 + tl;
                const float v = vxc[tl];  // (the extra row's V values are staged into a compact array)
                float rv;
                if (WM != 3) {
                    float s0 = 0.f, s1 = 0.f;
                    for (int k = 4 * kl; k < rp; k += 64) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wxs + k);
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + k);
                        s0 += wv[0] * hv[0] + wv[1] * hv[1];
                        s1 += wv[2] * hv[2] + wv[3] * hv[3];
                    }
                    const float s = row_sum_f(s0 + s1);
                    const float lam = fmaxf(s, kFlr);
                    if (OBJ && kl == 0) dsum += (t < a.T) ? div_term<BM>(v, lam, a.beta, a.inv_bb1) : 0.f;
                    if (WM == 0) rv = v * fast_rcp(lam);
                    else if (WM == 1) rv = den_of_lam<BM>(lam, a.beta);
                    else rv = v * numfac_of_lam<BM>(lam, a.beta);
                } else {
                    rv = v;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    rxv[c0 + c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv), 16 * c));
            }
            if (OBJ) acc_div += (double)dsum;
            // (one ds_read_b128 per column and 256 rows of H: beside its own MFMAs every instruction of this wave counts)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) {
                const int k0 = 256 * pc + 4 * lane;
                if (k0 + (kc ? kap_base * 32 : 0) < rp) {
                    f32x4 g4;
                    if (GXL) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(g4) : "v"(gxa));
#pragma unroll
                    for (int c = 0; c < CPW; ++c) {
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(xH + (xw * CPW + c) * ldh + k0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (GXL) g4[e] += rxv[c] * hv[e];
                            else gx[4 * pc + e] += rxv[c] * hv[e];
                        }
                    }
                    if (GXL) asm volatile("ds_write_b128 %0, %1" ::"v"(gxa), "v"(g4));
                }
            }
    };
    // ================================ loader role, LDS-DMA =====================================
    // Beside a wave that issues v_mfma_f32_32x32x2_f32 back to back a second wave of the SIMD gets an instruction in only
    // where the MFMA wave stalls (scripts/mfma_valu_overlap.hip), so how long a loader takes to stage a tile is set by
    // how many instructions that takes, not by the bytes: ~300 per tile with loads into registers + ds_write + index
    // arithmetic + the row sums, which made `ready` arrive late (phase stamps: the consumers waited 1.0 k cycles per tile
    // for it, and any cycle their loops saved went into that wait).  Here a tile is ~17 buffer_load ... lds per wave
    // (scalar addressing: one 1 KiB piece per padded H row, one piece per frame row for this row group's columns of V) and the row sums are taken
    // over the wave's OWN rows (complete as soon as its own DMA has landed) with one ds_read_b128 per row and 256 columns.
    float rs4[NPC][4];  // row sums of H over this wave's rows: k = 256 p + 4 lane + e   (rp <= 256 * NPC)
#pragma unroll
    for (int pp = 0; pp < NPC; ++pp)
#pragma unroll
        for (int e = 0; e < 4; ++e) rs4[pp][e] = 0.f;
    if (is_loader) {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        const int lw = w - NWB;
        const int npc = (rp + 255) >> 8;    // 1 KiB pieces per H row
        const int nVb = TT * Fp * 4;        // bytes of a V tile
        // V: only the columns of THIS workgroup's row group (a.Fm rows in groups of 32*NWB) -- every frame row in one piece of
        // up to 512 B -- and, for the group that owns it, the extra row's 32 values into a compact array: with two row groups
        // the V block crosses L2 once per iteration instead of twice (447 -> 344 MB per launch on C2)
        const int c0 = by * NWB * 32;                                  // first column (float) of the group
        const int cw = (a.Fm - c0 < NWB * 32 ? a.Fm - c0 : NWB * 32);  // its width in floats (multiple of 32)
        auto dma_tile = [&](int tile, float* dst, int bi) {
            const __amdgpu_buffer_rsrc_t rh =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)tile * TT * rp), 0, TT * rp * 4, 0x00020000);
            const int pc0 = kc ? (int)blockIdx.z : 0, pcn = kc ? 1 : npc;  // kc: only this kappa-group's 1 KiB piece of every row
            for (int t = lw; t < TT; t += NL)
                for (int pc = 0; pc < pcn; ++pc)
                    if ((pc0 + pc) * 256 + lane * 4 < rp)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + t * ldh + pc * 256), 16, lane * 16,
                                                                 (t * rp + (pc0 + pc) * 256) * 4, 0, 0);
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)tile * TT * Fp), 0, nVb, 0x00020000);
            for (int t = lw; t < TT; t += NL)
                for (int pc = 0; pc * 256 < cw; ++pc)
                    if (pc * 256 + lane * 4 < cw)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + TT * ldh + t * ldv + pc * 256), 16, lane * 16,
                                                                 (t * Fp + c0 + pc * 256) * 4, 0, 0);
            if (do_x && lw == 0 && lane < TT)  // one dword per frame: V[Fm, t] -> vx[buffer][t]
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(vx + bi * 32), 4, lane * Fp * 4, a.Fm * 4, 0, 0);
        };
        auto sums_of = [&](const float* H) {
            if (!do_s) return;
            for (int t = lw; t < TT; t += NL)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
                    if (pc < npc && pc * 256 + lane * 4 < rp) {
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(H + t * ldh + pc * 256 + lane * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) rs4[pc][e] += hv[e];
                    }
        };
        if (tb < te) {
            dma_tile(tb, lds, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing else orders a ds_read behind an LDS-DMA
            sums_of(lds);
            rp_post(ready, lw, 1u, lane);
        }
        int nb = 0;  // buffer of the tile being staged: tile `it + 1` of the chunk lives in buffer (it + 1) % NBUF
        for (int tile = tb, it = 0; tile + 1 < te; ++tile, ++it) {
            nb = nb + 1 == NBUF ? 0 : nb + 1;
            float* nH = lds + nb * bufsz;
            // the buffer's last tenant was tile it + 1 - NBUF: every consumer must have finished it + 2 - NBUF tiles
            if (it + 2 - NBUF > 0) {
                rp_await(done, (unsigned)(it + 2 - NBUF), a.stop);
                if (NWB == 8) rp_await(done + 4, (unsigned)(it + 2 - NBUF), a.stop);
            }
            dma_tile(tile + 1, nH, nb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sums_of(nH);
            rp_post(ready, lw, (unsigned)(it + 2), lane);
        }
    }

    // Two consumer waves per SIMD (NWB = 8) start together and do identical work, so left alone they run in LOCKSTEP: both wait
    // for the first W fragments of P3 at the same time, both sit in their epilogues at the same time -- every latency is
    // exposed on both at once and the SIMD idles (a11: 20 % of its cycles with neither an MFMA nor a VALU instruction in
    // flight, rocprofv3 PMC).  Nothing synchronises the consumers with each other, so a one-off delay of the second wave of
    // each SIMD at kernel start persists for the whole chunk: one wave's stalls then fall into the other's MFMA loops.
    if (NL > 0 && NWB == 8 && !is_loader && w >= 4 && a.stagger > 0 && tb < te) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - ts < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(16);
    }
    SNMF_STAMP_DECL
    int cb = NL > 0 ? NBUF - 1 : 0;  // buffer of the consumers' current tile (advanced at the loop top)
    // (an idle consumer's loop is empty.  Through the loop's END, not one more term in its condition: as `&& !(NWB == 4 && idle_c)`
    //  -- constant true in the eight-consumer kernels -- it still cost THEM 176 spilled VGPRs at their 168-register limit)
    int te_c = te;
    if constexpr (NWB == 4) {
        if (idle_c) te_c = tb;
    }
    for (int tile = tb, it = 0; tile < te_c && !is_loader; ++tile, ++it) {
        const int t0 = tile * TT;
        SNMF_STAMP(0);
        if (NL > 0) cb = cb + 1 == NBUF ? 0 : cb + 1;
        float* Hs = lds + cb * bufsz;  // [32][ldh]
        float* Vs = Hs + TT * ldh;                            // [TT][ldv]  (no HBM access in the MFMA loops)
        if (NL == 0) {
            __syncthreads();
            stage_in<NST>(a.Hin + (size_t)t0 * rp, Hs, TT, rp, ldh, sid);
            {   // this row group's columns of the V tile: cw4 16-byte cells per frame row, all loads of a batch in flight
                const int c0 = by * NWB * 32, cw4 = (a.Fm - c0 < NWB * 32 ? a.Fm - c0 : NWB * 32) / 4, n4 = TT * cw4;
                const float* vsrc = a.V + (size_t)t0 * Fp + c0;
                constexpr int B = 8;
                for (int i0 = sid; i0 < n4; i0 += B * NST) {
                    f32x4 x[B];
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const int i = i0 + b * NST;
                        if (i < n4) x[b] = *reinterpret_cast<const f32x4*>(vsrc + (size_t)(i / cw4) * Fp + 4 * (i % cw4));
                    }
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const int i = i0 + b * NST;
                        if (i < n4) *reinterpret_cast<f32x4*>(Vs + (i / cw4) * ldv + 4 * (i % cw4)) = x[b];
                    }
                }
                if (do_x && sid < TT) vx[sid] = a.V[(size_t)(t0 + sid) * Fp + a.Fm];
            }
            __syncthreads();
        }
        // NL > 0: the wait for the staged tile.  (Taken behind P3's first W loads instead, as k_hstep_rp does with its
        // waits, it cost 5 %: 0.236 -> 0.247 ms.)
        if (NL > 0) rp_await(ready, (unsigned)(it + 1), a.stop);
        SNMF_STAMP(1);
        if (NL == 0 && do_s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = sid + j * NST;
                if (k < rp) {
                    float sacc = 0.f;
                    for (int t = 0; t < TT; ++t) sacc += Hs[t * ldh + k];
                    ssum[j] += sacc;
                }
            }
        }
        if (do_x) xrow_tile(Hs, t0, w, vx + cb * 32);
        SNMF_STAMP(2);
        if (!active || (TIL && til > 1 && it % til != tph)) {  // (another team's tile: only the progress report)
            if (NL > 0) rp_post(done, w, (unsigned)(it + 1), lane);
            continue;
        }

        float R[16];
        if (WM != 3) {
            // ---- P3: Lam'^T[t, f] = sum_k H[k,t] W[f,k]  (A = H from LDS, B = W from L2) -----
            f32x16 acc1[1] = {zero16()};
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wt4 + (size_t)phi * rp * 32) + lane;
            {
                const __amdgpu_buffer_rsrc_t rsw = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);
                const float* spl = Hs + (fl & (TT - 1)) * ldh + 4 * h;
                if (a.nqk == 32) contract_p3_buf<true>(acc1[0], rsw, lane * 16, phi * rp * 128, spl, 32, NoGate());
                else contract_p3_buf<false>(acc1[0], rsw, lane * 16, phi * rp * 128, spl, a.nqk, NoGate());
            }
            SNMF_STAMP(3);
            const f32x16 acc = acc1[0];
            // lane (f = fl, h), reg -> t = t0 + drow(reg,h)
            const int f = phi * 32 + fl;
            float dsum = 0.f;
            // (the bounds masks of the objective only where a tile has padding: a wave-uniform test, as rp_p1_epilogue)
            auto ratio_rows = [&](auto masked) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (TT < 32 && i >= TT / 2) {  // rows drow(i,h) >= TT: duplicates of the narrow tile
                        R[i] = 0.f;
                        continue;
                    }
                    const int t = t0 + drow(i, h);
                    const float v = Vs[drow(i, h) * ldv + fc];
                    float lam = fmaxf(acc[i], kFlr);
                    if (OBJ) {
                        if (do_obj) {
                            float d = div_term<BM>(v, lam, a.beta, a.inv_bb1);
                            // (the unmasked add must not be contracted with the term's last product into an fma: the masked
                            //  form -- a select between product and add -- cannot be, the two would differ in the last bit,
                            //  and the early-stop decisions in the online loop's Euclidean test hang on less.  __fadd_rn is
                            //  a plain "+" to the compiler and does not stop it)
                            if (decltype(masked)::value) {
                                dsum += (f < a.F && t < a.T) ? d : 0.f;
                            } else {
#pragma clang fp contract(off)
                                dsum = dsum + d;
                            }
                        }
                    }
                    if (WM == 0) R[i] = v * fast_rcp(lam);
                    else if (WM == 1) R[i] = den_of_lam<BM>(lam, a.beta);
                    else R[i] = v * numfac_of_lam<BM>(lam, a.beta);
                }
            };
            if (OBJ && TT == 32 && phi * 32 + 32 <= a.F && t0 + 32 <= a.T) ratio_rows(std::false_type{});
            else ratio_rows(std::true_type{});
            if (OBJ) acc_div += (double)dsum;
        } else {
            const int f = phi * 32 + fl;
#pragma unroll
            for (int i = 0; i < 16; ++i) R[i] = (TT < 32 && i >= TT / 2) ? 0.f : Vs[drow(i, h) * ldv + fc];
        }
        // ---- P4: G[phi, kap] += ratio[f, t] * H[k, t]  (A = ratio registers, B = H from LDS)
        // B fragments (one ds_read_b32 per MFMA) are fetched a whole kappa-tile (16 reads) ahead.
        // kappa-tiles beyond nk (NK is a template bound) are skipped by a scalar test.
        SNMF_STAMP(4);
        // The H image of this kernel is padded to 32*NK columns per kappa-group (host: ldh), so the prefetch of a tile
        // beyond nk reads finite padding -- no clamps, and every read is (one of 16 row bases) + an immediate offset.
        const float* hb = Hs + (4 * h) * ldh + fl + (kc ? 0 : kap_base * 32);
        const float* hrow[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) hrow[i] = hb + ((i & 3) + 8 * (i >> 2)) * ldh;
        float b0[16], b1[16];
        auto ldb = [&](float (&b)[16], int kap) {
#pragma unroll
            for (int i = 0; i < TT / 2; ++i) b[i] = hrow[i][kap * 32];
        };
        // two named buffers alternate (a register copy would have to wait for the load it follows);
        // the tile index is a compile-time constant so that G[] is never indexed dynamically
#define SNMF_KTILE(KP, CUR, NXT)                                                  \
    if constexpr ((KP) < NK) {                                                    \
        if (kap_base + (KP) < nkm) { /* (scalar test: tiles past the last one that goes through the MFMAs are skipped) */ \
            if constexpr ((KP) + 1 < NK) ldb(NXT, (KP) + 1);                      \
            SNMF_PIN();                                                           \
            _Pragma("unroll") for (int i = 0; i < TT / 2; ++i) G[KP] = mfma32(R[i], CUR[i], G[KP]); \
        }                                                                         \
    }
        ldb(b0, 0);
        SNMF_KTILE(0, b0, b1)
        SNMF_KTILE(1, b1, b0)
        SNMF_KTILE(2, b0, b1)
        SNMF_KTILE(3, b1, b0)
        SNMF_KTILE(4, b0, b1)
        SNMF_KTILE(5, b1, b0)
        SNMF_KTILE(6, b0, b1)
        SNMF_KTILE(7, b1, b0)
        SNMF_KTILE(8, b0, b1)
        SNMF_KTILE(9, b1, b0)
        SNMF_KTILE(10, b0, b1)
        SNMF_KTILE(11, b1, b0)
        SNMF_KTILE(12, b0, b1)
        SNMF_KTILE(13, b1, b0)
        SNMF_KTILE(14, b0, b1)
        SNMF_KTILE(15, b1, b0)
#undef SNMF_KTILE
        if constexpr (LX > 0) {  // the leftover columns, on the VALU
            const int koff = (a.nk - 1) * 32 - fl;  // hrow[i] points at column fl of frame t_i
            // (four frames at a time, fenced: left alone the compiler hoists all 16 / 32 reads -- 64+ live registers, spills)
#pragma unroll
            for (int i0 = 0; i0 < TT / 2; i0 += 4) {
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) {
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(hrow[i] + koff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gl[e] += R[i] * h0[e];
                }
                if constexpr (LX > 1) {
#pragma unroll
                    for (int i = i0; i < i0 + 4; ++i) {
                        const f32x4 h1 = *reinterpret_cast<const f32x4*>(hrow[i] + koff + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) gl[4 + e] += R[i] * h1[e];
                    }
                }
                SNMF_PIN();
            }
        }
        if (NL > 0) rp_post(done, w, (unsigned)(it + 1), lane);  // this wave's last LDS read of the tile fed the MFMAs above
        SNMF_STAMP(5);
    }
#ifdef SNMF_PROF
    if (!is_loader && a.prof) {
        const size_t wv = ((size_t)(blockIdx.x + gridDim.x * blockIdx.y)) * NWB + w;  // 1-D grid in split mode
        if (wv < 4096) {
            SNMF_STAMP_OUT(a.prof + (4096 + wv) * 12, 12);
            SNMF_STAMP_CLK(a.prof, 4096 + wv);
        }
    }
#endif

    if constexpr (TIL && NL > 0) if (til > 1) {
        // the teams' partial statistics -> team 0, through the tile buffers (every consumer is past its last tile and the
        // loaders past their last DMA at the first barrier), added in team order
        float* xs = lds;  // [(til - 1) * nfl][NK * 16 + 4 * LX][64]
        constexpr int NV = NK * 16 + (LX ? 4 * LX : 0);
        __syncthreads();
        if (active && tph > 0) {
            float* dst = xs + (size_t)((tph - 1) * nfl + wl) * NV * 64 + lane;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[(k * 16 + i) * 64] = G[k][i];
            if constexpr (LX > 0) {
#pragma unroll
                for (int j = 0; j < 4 * LX; ++j) dst[(NK * 16 + j) * 64] = gl[j];
            }
        }
        __syncthreads();
        if (active && tph == 0) {
            for (int p = 1; p < til; ++p) {
                const float* src = xs + (size_t)((p - 1) * nfl + wl) * NV * 64 + lane;
#pragma unroll
                for (int k = 0; k < NK; ++k)
#pragma unroll
                    for (int i = 0; i < 16; ++i) G[k][i] += src[(k * 16 + i) * 64];
                if constexpr (LX > 0) {
#pragma unroll
                    for (int j = 0; j < 4 * LX; ++j) gl[j] += src[(NK * 16 + j) * 64];
                }
            }
        }
        __syncthreads();  // (the row-sum reduction below reuses the same memory)
    }
    const bool writes = TIL ? (active && tph == 0) : active;
    // ---- write the partial slab: D tile lane (k = fl, h), reg -> f = 32*phi + drow(reg,h)
    if (writes) {
        float* slab = a.slabs + ((size_t)chunk * n_mat + mat_index) * rp * Fp;
#pragma unroll
        for (int kap = 0; kap < NK; ++kap) {
            if (kap_base + kap < nkm) {
                float* dst = slab + (size_t)((kap_base + kap) * 32 + fl) * Fp + phi * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o = {G[kap][4 * g], G[kap][4 * g + 1], G[kap][4 * g + 2], G[kap][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = o;
                }
            }
        }
    }
    if (writes && LX > 0) {  // leftover columns: slab[k0 + j][f] = the two lane halves' partial sums (fixed order: h = 0, then h = 1)
        float* slab = a.slabs + ((size_t)chunk * n_mat + mat_index) * rp * Fp;
        const int k0 = (a.nk - 1) * 32;
#pragma unroll
        for (int j = 0; j < 4 * LX; ++j) {
            const float other = __shfl_xor(gl[j], 32, 64);
            if (h == 0 && k0 + j < rp) slab[(size_t)(k0 + j) * Fp + phi * 32 + fl] = gl[j] + other;
        }
    }
    if (NL > 0) {
        if (do_s) {  // fixed-order sum of the loader waves' partial row sums, through LDS
            __syncthreads();
            float* red = lds;  // [NL][rp]
            if (is_loader) {
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = pc * 256 + lane * 4 + e;
                        if (k < rp) red[(w - NWB) * rp + k] = rs4[pc][e];
                    }
            }
            __syncthreads();
            for (int k = threadIdx.x; k < rp; k += NTHR) {
                float sk = 0.f;
                for (int ww = 0; ww < NL; ++ww) sk += red[ww * rp + k];
                a.spart[(size_t)chunk * rp + k] = sk;
            }
            __syncthreads();
        }
    } else
    if (do_s && sid >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = sid + j * NST;
            if (k < rp) a.spart[(size_t)chunk * rp + k] = ssum[j];
        }
    }
    if (do_x) {
        // fixed-order sum of the consumers' partial extra rows, through LDS
        if (GXL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the assembly stores above are not the compiler's to wait for)
        __syncthreads();
        float* red = GXL ? gxs : lds;  // [NWB][rp]  (NWB*rp <= 32*ldh)
        const int rw = GXL ? gxw : rp;
        if (!GXL && !is_loader) {
#pragma unroll
            for (int i = 0; i < 4 * NPC; ++i) {
                const int k = 256 * (i >> 2) + 4 * lane + (i & 3);
                if (k < rp) red[w * rw + k] = gx[i];
            }
        }
        __syncthreads();
        float* slab = a.slabs + ((size_t)chunk * n_mat + mat_index) * rp * Fp;
        const int koff = kc ? kap_base * 32 : 0, kn = kc ? 32 * NK : rp;  // kc: this kappa-group's columns only
        for (int k = threadIdx.x; k < kn && koff + k < rp; k += NTHR) {
            float s = 0.f;
            for (int ww = 0; ww < NWB; ++ww) s += red[ww * rw + k];
            slab[(size_t)(koff + k) * Fp + a.Fm] = s;
        }
        __syncthreads();
    }
    if (OBJ) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);
        red[threadIdx.x] = acc_div;
        __syncthreads();
        // NTHR is 256, 512 or 768: fold the tail above the largest power of two first
        constexpr int P2 = NTHR >= 512 ? 512 : 256;
        if ((int)threadIdx.x + P2 < NTHR) red[threadIdx.x] += red[threadIdx.x + P2];
        __syncthreads();
        for (int s = P2 / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0 && blockIdx.z == 0) {
            // (uneven row-group split: a 1-D grid, one slot per workgroup; n_chunks * n_fg slots are reduced, the unused ones stay 0)
            const int slot = a.n_ch1 > 0 ? (int)blockIdx.x : by * n_chunks + chunk;
            a.part[2 * slot] = red[0];
            a.part[2 * slot + 1] = 0.0;
        }
    }
}

template <int NK, int NWB, int NL, int WPS, int WM, int BM, bool OBJ, int TT = 32, int LX = 0>
__global__ __launch_bounds__((NWB + NL) * 64, WPS) void k_wstats(StepArgs a, int n_chunks, int mat_index,
                                                                 int n_mat) {
    wstats_body<NK, NWB, NL, WPS, WM, BM, OBJ, TT, LX, false>(a, n_chunks, mat_index, n_mat);
}
// the same with consumer TEAMS (StepArgs::til): a kernel of its own, see the note on TIL above
template <int NK, int NWB, int NL, int WPS, int WM, int BM, bool OBJ, int TT = 32, int LX = 0>
__global__ __launch_bounds__((NWB + NL) * 64, WPS) void k_wstats_teams(StepArgs a, int n_chunks, int mat_index,
                                                                       int n_mat) {
    wstats_body<NK, NWB, NL, WPS, WM, BM, OBJ, TT, LX, true>(a, n_chunks, mat_index, n_mat);
}

// (conv_test: near the top of this file, in front of obj_partial_out)

// ============================================================================================
// Statistics buffer (fp64, device), the unit that is all-reduced across ranks:
//   [ M0 (rp*Fp) | M1 (rp*Fp, beta != 1) | s (rp) | div | sh ]
// M0 = G (KL) or Q;  M1 = P.  Element (f,k) at k*Fp + f.
// ============================================================================================
struct ReduceArgs {
    const float* slabs;   // [n_chunks][n_mat][rp*Fp]
    const float* spart;   // [n_chunks][rp]
    const double* part;   // [n_part][2]
    double* stats;
    const int* stop;
    int n_chunks, n_mat, n_part, rp, Fp;
    double* qp_buf;       // k_wfin with gridDim.y = S > 1 (few columns: r <= 128): [r][n_mat * Fp] summed statistics of a column, a row slice per workgroup
    unsigned* fin_cnt;    // ... [r] arrivals per column (monotonic; a launch adds S to each)
    int r;                // real columns of W: rows k >= r of every slab are zero and stay zero in the statistics
    int do_mats;          // reduce slabs + s
    int do_obj;           // reduce objective partials
    double sh_const;      // W-only mode: sum(S.*H) is constant, added here
    int use_sh_const;
    // > 0: the convergence test of that iteration runs behind the objective fold (H-only loop of snmf_plan_run: one
    // launch instead of k_reduce + k_check)
    int check_it;
    double conv_eps;
    double* divh;
    double* costh;
    DevState* st;
    // Fused push of the multi-device entry (csrc/snmf_multi.h): npush > 0 = every reduced value ALSO goes straight into this
    // rank's slot of the gather buffer of npush ranks (peer stores) instead of a second launch copying the statistics there;
    // push_done / push_flag / push_seq: the FLAGS ordering's completion protocol (the last workgroup posts the sequence
    // number on every peer), nullptr in EVENTS mode.
    int npush;
    double* push_dst[16];
    unsigned* push_flag[16];
    unsigned* push_done;
    unsigned push_seq;
};

#ifdef SNMF_AUX_KERNELS  // k_reduce .. k_mdi_start: launched by snmf_api.hip only (the other translation units skip their code generation)
static __global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
    if (a.stop && *a.stop) return;
    const size_t nel = (size_t)a.rp * a.Fp;   // multiple of 4
    const size_t nmat = nel * a.n_mat;
    double* sc = a.stats + nmat + a.rp;
    if (a.do_mats) {
        // slabs: a block owns 32 x 4 consecutive elements; its 8 thread groups each add a contiguous
        // eighth of the chunks (8 loads in flight), then the 8 partial sums are added in group
        // order.  Every order is fixed => bitwise reproducible and identical on every rank.
        __shared__ double part[8][32][4];
        // only the r real rows of each matrix (r = 100: 22 % fewer bytes than rp = 128): virtual index v4 -> f32x4 position
        const size_t n4r = (size_t)a.r * a.Fp / 4, n4 = n4r * a.n_mat;
        auto pos_of = [&](size_t v4) {
            const size_t m = v4 / n4r;
            return m * (nel / 4) + (v4 - m * n4r);
        };
        const size_t cstride = nmat;  // floats between consecutive chunks
        const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
        const int cb = (int)(((long long)a.n_chunks * g) / 8), ce = (int)(((long long)a.n_chunks * (g + 1)) / 8);
        for (size_t b4 = (size_t)blockIdx.x * 32; b4 < n4; b4 += (size_t)gridDim.x * 32) {
            const size_t i4 = b4 + e;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            if (i4 < n4) {
                const float* p = a.slabs + 4 * pos_of(i4);
                int c = cb;
                for (; c + 16 <= ce; c += 16) {  // (sixteen loads in flight: a 128-chunk launch is one round trip per group)
                    f32x4 x[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) x[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(c + j) * cstride);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        s0 += (double)x[j][0];
                        s1 += (double)x[j][1];
                        s2 += (double)x[j][2];
                        s3 += (double)x[j][3];
                    }
                }
                for (; c + 8 <= ce; c += 8) {
                    f32x4 x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(c + j) * cstride);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s0 += (double)x[j][0];
                        s1 += (double)x[j][1];
                        s2 += (double)x[j][2];
                        s3 += (double)x[j][3];
                    }
                }
                for (; c < ce; ++c) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(p + (size_t)c * cstride);
                    s0 += (double)x[0];
                    s1 += (double)x[1];
                    s2 += (double)x[2];
                    s3 += (double)x[3];
                }
            }
            __syncthreads();
            part[g][e][0] = s0;
            part[g][e][1] = s1;
            part[g][e][2] = s2;
            part[g][e][3] = s3;
            __syncthreads();
            if (threadIdx.x < 128) {
                const int ee = threadIdx.x >> 2, j = threadIdx.x & 3;
                if (b4 + ee < n4) {
                    double t = 0.0;
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg) t += part[gg][ee][j];
                    const size_t idx = 4 * pos_of(b4 + ee) + j;
                    a.stats[idx] = t;
                    for (int q = 0; q < a.npush; ++q) a.push_dst[q][idx] = t;
                }
            }
        }
        // row sums of H (KL only; zeros otherwise): same two-level fixed-order scheme, 32 rows a block
        for (int k0 = blockIdx.x * 32; k0 < a.rp; k0 += gridDim.x * 32) {
            double sk = 0.0;
            int c = cb;
            for (; c + 8 <= ce; c += 8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = a.spart[(size_t)(c + j) * a.rp + k0 + e];
#pragma unroll
                for (int j = 0; j < 8; ++j) sk += (double)x[j];
            }
            for (; c < ce; ++c) sk += (double)a.spart[(size_t)c * a.rp + k0 + e];
            __syncthreads();
            part[g][e][0] = sk;
            __syncthreads();
            if (threadIdx.x < 32) {
                double t = 0.0;
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) t += part[gg][threadIdx.x][0];
                a.stats[nmat + k0 + threadIdx.x] = t;
                for (int q = 0; q < a.npush; ++q) a.push_dst[q][nmat + k0 + threadIdx.x] = t;
            }
        }
    }
    // objective partial sums: the LAST block folds them (fixed strided + tree order)
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ double red2[2][256];
        double d = 0.0, h = 0.0;
        if (a.do_obj) {
            for (int c = threadIdx.x; c < a.n_part; c += 256) {
                d += a.part[2 * c];
                h += a.part[2 * c + 1];
            }
        }
        red2[0][threadIdx.x] = d;
        red2[1][threadIdx.x] = h;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                red2[0][threadIdx.x] += red2[0][threadIdx.x + st];
                red2[1][threadIdx.x] += red2[1][threadIdx.x + st];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            sc[0] = a.do_obj ? red2[0][0] : 0.0;
            sc[1] = a.do_obj ? (a.use_sh_const ? a.sh_const : red2[1][0]) : 0.0;
            for (int q = 0; q < a.npush; ++q) {
                a.push_dst[q][nmat + a.rp] = sc[0];
                a.push_dst[q][nmat + a.rp + 1] = sc[1];
            }
            if (a.check_it > 0) conv_test(sc, a.divh, a.costh, a.st, a.check_it, a.conv_eps, true);
        }
    }
    if (a.npush > 0 && a.push_done) {  // FLAGS ordering (as k_push_stats): the last workgroup announces the push on every peer
        __threadfence_system();        // this thread's peer stores are performed before it reports
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.push_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1) {
                __hip_atomic_store(a.push_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                for (int q = 0; q < a.npush; ++q) __hip_atomic_store(a.push_flag[q], a.push_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
#endif  // SNMF_AUX_KERNELS (k_reduce)

struct ApplyArgs {
    const double* stats;
    double* Wc;       // [rp][Fp] master copy, column-major, fp64 (see k_wapply)
    float* Wcf;       // [rp][Fp] the same rounded to fp32 (k_hsolve_frame loads its register blocks from it)
    float* Wt4;
    float* Wk4;
    float* dphv;
    float* colsum;
    const float* lamk;
    const uint8_t* w_ind;  // [rp] (pads 0)
    double* divh;
    double* costh;
    DevState* st;
    float* wx;        // extra-row mode: W[Fm, :]
    int F, r, Fp, rp, n_mat;
    int Fm, Fq, xr;
    int check_it;     // >0: run the convergence test for that iteration first
    int do_update;    // apply the W update (0: check only)
    int init_mode;    // 1: normalise the given W only (src/sparse_nmf.m:157-159), write wn
    double conv_eps;
    double* wn;       // init_mode: column norms out [rp]
    // Fused sum of the multi-device entry (csrc/snmf_multi.h): gather != nullptr = the statistics are the sum, in RANK ORDER,
    // of ngather slots of gather_len doubles each (the one-shot exchange's gather buffer of this rank) -- every column's
    // workgroup adds its own column, instead of a launch of its own summing the whole buffer first.  gflags / gseq / fault: the
    // FLAGS ordering's arrival words (nullptr: EVENTS -- the stream already waited for every push).
    const double* gather;
    int ngather;
    size_t gather_len;
    const unsigned* gflags;
    unsigned gseq;
    int* fault;
};

// One workgroup (256 threads) per column k of W.  src/sparse_nmf.m:215-244.
// The master copy of W is fp64 although every contraction runs on fp32 images of it: the update is
// multiplicative, so an entry that underflows fp32 (rows the data does not excite shrink by orders of
// magnitude per iteration) would be pinned at exactly 0 for the rest of the solve, while in the
// reference's doubles it stays positive and comes back as soon as V./Lam is large there -- which the
// noise-dictionary adaptation (src/bnmf_sep_event_RT_IS16.m:296-336) does all the time.  W is
// F x r: keeping it in fp64 costs nothing measurable.
// Q / P: this column of the reduced statistics; P == nullptr: KL, the
// "P" of every row is the row sum sk of H.
// wpre: this thread's rows of the column (wc[tid + 256 i], i < 5; F <= 1280) loaded by the caller ahead of time, or nullptr
// (by reference + a flag, every index a compile-time constant: through a pointer the array went to scratch)
// w_ind_k / lamk_k: w_ind[k] and lambda_k[k], LOADED BY THE CALLER at the top of its kernel: read here they were two more dependent
// global round trips on the launch's critical path (the first decides everything that follows, the second sits between the last
// column sum and the last store) -- 1.5-2 us each on a cold, mostly idle chip (profiles/r06_experiments.md section 10: phase stamps)
__device__ __forceinline__ void wapply_column(const ApplyArgs& a, int k, int tid, const double* Q, const double* P,
                                              double sk, double (&red)[3][256], const double (&wpre)[5], bool use_pre,
                                              int w_ind_k, float lamk_k) {
    double* wc = a.Wc + (size_t)k * a.Fp;
    const bool upd = a.init_mode ? false : (w_ind_k != 0);
    // This thread's rows f = tid, tid + 256, ... of the column live in registers for the whole epilogue (F <= 1024 + 1
    // rows: up to NR = 5 per thread; larger F loops again through memory like before).
    constexpr int NR = 5;
    const bool in_regs = a.F <= NR * 256;
    double wv[NR], qv[NR], pv[NR];
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int f = tid + 256 * i;
            const bool ok = f < a.F;
            wv[i] = use_pre ? wpre[i] : (ok ? wc[f] : 0.0);
            qv[i] = (ok && upd) ? Q[f] : 0.0;
            pv[i] = (ok && upd) ? (P ? P[f] : sk) : 0.0;
        }
    }
    // pass 1: column sums needed by the update
    double cQW = 0.0, cPW = 0.0, cW = 0.0;
    if (upd) {
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                cQW += qv[i] * wv[i];
                cPW += pv[i] * wv[i];
            }
        } else {
            for (int f = tid; f < a.F; f += 256) {
                const double w0 = wc[f];
                cQW += Q[f] * w0;
                cPW += (P ? P[f] : sk) * w0;
            }
        }
        wg_sum2_256(cQW, cPW, &red[0][0], &red[1][0], tid);
    }
    // updated (un-normalised) entry: dpw = max(P + W.*colsum(Q.*W), flr); dmw = Q + W.*colsum(P.*W)
    auto upd_val = [&](double w0, double Qv, double Pv) -> double {
        if (upd) {
            double dpw = Pv + w0 * cQW;
            dpw = dpw > 1e-9 ? dpw : 1e-9;
            w0 = w0 * (Qv + w0 * cPW) / dpw;
        }
        return w0;
    };
    auto updated = [&](int f) -> double { return upd_val(wc[f], upd ? Q[f] : 0.0, upd ? (P ? P[f] : sk) : 0.0); };
    // pass 2: squared norm of the updated column
    double ssq = 0.0;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            wv[i] = (tid + 256 * i < a.F) ? upd_val(wv[i], qv[i], pv[i]) : 0.0;
            ssq += wv[i] * wv[i];
        }
    } else {
        for (int f = tid; f < a.F; f += 256) {
            const double w1 = updated(f);
            ssq += w1 * w1;
        }
    }
    const double nrm = sqrt(wg_sum_256(ssq, &red[2][0], tid));
    // pass 3: normalise (ALL columns, :242), write the three images, column sum
    auto emit = [&](int f, double w1) {
        const double wd = w1 / nrm;
        const float wf = (float)wd;  // operand images: an entry below fp32 range adds < 1e-38*max(h) to Lam, far under the 1e-9 floor
        cW += (double)wf;
        wc[f] = wd;
        a.Wcf[(size_t)k * a.Fp + f] = wf;
        if (f < a.Fm) {  // Wt4[phi][q][h][f32][e] = W[32phi+f32][8q+4h+e]
            const int phi = f >> 5, f32 = f & 31, q = k >> 3, hh = (k >> 2) & 1, e = k & 3;
            a.Wt4[(((size_t)phi * (a.rp / 8) + q) * 2 + hh) * 128 + f32 * 4 + e] = wf;
        } else {
            a.wx[k] = wf;  // extra row
        }
        {   // Wk4[kap][q][h][k32][e] = W[8q+4h+e][32kap+k32]
            const int kap = k >> 5, k32 = k & 31, q = f >> 3, hh = (f >> 2) & 1, e = f & 3;
            a.Wk4[(((size_t)kap * (a.Fq / 8) + q) * 2 + hh) * 128 + k32 * 4 + e] = wf;
        }
    };
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (tid + 256 * i < a.F) emit(tid + 256 * i, wv[i]);
    } else {
        // the updated value is recomputed from the ORIGINAL column: write only after every read of this thread's rows
        for (int f = tid; f < a.F; f += 256) emit(f, updated(f));
    }
    cW = wg_sum_256(cW, &red[0][0], tid);
    if (tid == 0) {
        const float cs = (float)cW;
        a.colsum[k] = cs;
        a.dphv[k] = fmaxf(cs + lamk_k, kFlr);
        if (a.init_mode) a.wn[k] = nrm;
    }
}

#ifdef SNMF_AUX_KERNELS
static __global__ __launch_bounds__(256) void k_wapply(ApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) double wap_cols[];  // gather mode: [n_mat][Fp] this column's summed statistics
    __shared__ double red[3][256];
    __shared__ double scs[2];
    if (!a.init_mode && a.st->stop) return;  // normalising a fresh W never depends on an earlier solve's flag
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    const size_t nel = (size_t)a.rp * a.Fp;
    const int w_ind_k = a.w_ind[k];   // (issued here, used by wapply_column: see there)
    const float lamk_k = a.lamk[k];
    if (a.gather) {
        if (a.gflags) {  // FLAGS ordering: every rank's push of this exchange must have arrived (k_sum_ranks' bounded wait)
            if (tid < a.ngather && !__hip_atomic_load(a.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while ((int)(__hip_atomic_load(a.gflags + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.gseq) < 0) {
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {  // 5 s
                        atomicExch(a.fault, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(32);
                }
            }
            __syncthreads();
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (tid < 2) {  // the two cost scalars, summed like everything else: slot 0 + slot 1 + ...
            const size_t i = nel * a.n_mat + a.rp + tid;
            double v = a.gather[i];
            for (int q = 1; q < a.ngather; ++q) v += a.gather[(size_t)q * a.gather_len + i];
            scs[tid] = v;
        }
        __syncthreads();
    }
    if (a.check_it > 0) {
        const double* sc = a.gather ? scs : a.stats + nel * a.n_mat + a.rp;
        bool stopnow = conv_test(sc, a.divh, a.costh, a.st, a.check_it, a.conv_eps, k == 0 && tid == 0);
        if (stopnow) return;
    }
    if (!a.do_update && !a.init_mode) return;
    if (k >= a.r) return;
    const double no_pre[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (a.gather) {
        double* Qs = wap_cols;
        double* Ps = wap_cols + a.Fp;
        for (int f = tid; f < a.Fp; f += 256) {
            const size_t i = (size_t)k * a.Fp + f;
            double v = a.gather[i];
            for (int q = 1; q < a.ngather; ++q) v += a.gather[(size_t)q * a.gather_len + i];
            Qs[f] = v;
            if (a.n_mat == 2) {
                double u = a.gather[nel + i];
                for (int q = 1; q < a.ngather; ++q) u += a.gather[(size_t)q * a.gather_len + nel + i];
                Ps[f] = u;
            }
        }
        double sk = 0.0;
        if (a.n_mat != 2) {
            const size_t i = nel * a.n_mat + k;
            sk = a.gather[i];
            for (int q = 1; q < a.ngather; ++q) sk += a.gather[(size_t)q * a.gather_len + i];
        }
        __syncthreads();
        wapply_column(a, k, tid, Qs, a.n_mat == 2 ? Ps : nullptr, sk, red, no_pre, false, w_ind_k, lamk_k);
        return;
    }
    const double* Q = a.stats + (size_t)k * a.Fp;
    const double* P = (a.n_mat == 2) ? a.stats + nel + (size_t)k * a.Fp : nullptr;
    const double sk = (a.n_mat == 2 || a.init_mode) ? 0.0 : a.stats[nel * a.n_mat + k];
    wapply_column(a, k, tid, Q, P, sk, red, no_pre, false, w_ind_k, lamk_k);
}
#endif  // SNMF_AUX_KERNELS (k_wapply)

// ============================================================================================
// k_wfin: k_reduce and k_wapply in ONE launch, for the loop of a single device (snmf_plan_run): nothing is exchanged
// between the two there, and as two launches they were 21 us of every iteration that do not shrink with T (the chunk
// slabs cross HBM once, 34 MB at C2; the rest was a second launch and the fp64 statistics' round trip through memory).
// One workgroup per column k of W: it adds the column's n_chunks slab pieces in k_reduce's order (eight contiguous chunk
// groups, then the eight partial sums in group order: the same doubles, bit for bit), keeps the sums in LDS and runs
// wapply_column on them.  Every workgroup folds the objective partials itself (k_reduce's order again) and evaluates the
// convergence test on its own copy.  The step API (snmf_plan_wstats / snmf_plan_wapply: the statistics leave the device
// or are summed over ranks in between) keeps the two kernels; tests compare the two paths bit for bit.
// Dynamic LDS: (8 + 1) * n_mat * Fp doubles.
// ============================================================================================
template <int NMAT>
__global__ __launch_bounds__(768) void k_wfin(ReduceArgs ra, ApplyArgs a) {
    extern __shared__ __attribute__((aligned(16))) double wfin_lds[];
    __shared__ double red[3][256];
    __shared__ double red2[2][256];
    __shared__ double skp[10];  // [8] row-sum groups, [8]: the split form's last-arriver flag (a static of its own would shift the dynamic LDS off 16 bytes)
    const int k = blockIdx.x, tid = threadIdx.x;
    const size_t nel = (size_t)a.rp * a.Fp;
    // Few columns (r <= 128: the reference's R = 100, 20, 10): r workgroups are a fraction of the chip, and each read its column of
    // EVERY chunk's slab -- 6.8 of the launch's 19 us at 513 x 72000, r = 20.  With gridDim.y = S > 1 the column's rows are cut into S
    // slices, a workgroup each (the sums per element are the same sums: bit-identical statistics); the slices meet in qp_buf and the
    // LAST of a column's S workgroups to arrive (nobody waits for anybody) runs the epilogue on the whole column.
    const int S = (int)gridDim.y, nEa = a.Fp / 4, nEs = (nEa + S - 1) / S;
    const int e_lo = (int)blockIdx.y * nEs, e_hi = e_lo + nEs < nEa ? e_lo + nEs : nEa;
    const int nE = e_hi > e_lo ? e_hi - e_lo : 0, nI = NMAT * nE;  // f32x4 of this workgroup's slice of the column (of both matrices)
    double* part = wfin_lds;                  // [8][nI][4]
    double* QP = wfin_lds + (size_t)8 * NMAT * nEs * 4;  // [NMAT][Fp]
    const size_t cstride = nel * NMAT;
    // Round 6: the launch was a CHAIN of dependent global round trips on a mostly idle chip (the stop flag, then the slab pieces,
    // then the objective partials, then the previous cost for the convergence test, then the column of W: ~1.5 us each of the
    // kernel's 14-16 us, which do not shrink with T).  Every global load whose address does not depend on another load is now
    // ISSUED before the first one is waited for: vmcnt retires in issue order, so the slab pieces (issued first) are consumed
    // first while the rest is already on its way.
    const int stop_now = __hip_atomic_load(&a.st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The column's slab pieces: item (g, e) = chunk group g, f32x4 e.  768 threads for this phase only (sixteen loads of
    // a thread in flight: one memory round trip for a column of C2, where 256 threads needed four); the W update behind
    // it is wapply_column's 256 threads, the other waves end at the barrier.
    auto item_geo = [&](int i, int& cb, int& ce, const float*& p) {
        const int g = i / nI, e = i - g * nI, m = e / nE, e4 = e - m * nE;
        cb = (int)(((long long)ra.n_chunks * g) / 8);
        ce = (int)(((long long)ra.n_chunks * (g + 1)) / 8);
        p = ra.slabs + (size_t)m * nel + (size_t)k * a.Fp + 4 * (e_lo + e4);
    };
    f32x4 x[16];  // sixteen chunks of the current item in flight
    int it_i = tid, cb = 0, ce = 0;
    const float* p = nullptr;
    auto load_batch = [&](int c) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (c + j < ce) x[j] = *reinterpret_cast<const f32x4*>(p + (size_t)(c + j) * cstride);
    };
    if (it_i < 8 * nI) {
        item_geo(it_i, cb, ce, p);
        load_batch(cb);
    }
    // this thread's rows of the column of W (wapply_column's register layout), the first 256 objective partials, the previous cost
    constexpr int NR = 5;
    double wpre[NR] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double pd0 = 0.0, ph0 = 0.0, last_cost = 0.0;
    const bool pre_w = a.F <= NR * 256 && k < a.r;
    if (tid < 256) {
        if (pre_w) {
            const double* wc = a.Wc + (size_t)k * a.Fp;
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (tid + 256 * i < a.F) wpre[i] = wc[tid + 256 * i];
        }
        if (ra.do_obj && tid < ra.n_part) {
            pd0 = ra.part[2 * tid];
            ph0 = ra.part[2 * tid + 1];
        }
        if (a.check_it > 1) last_cost = a.costh[a.check_it - 2];
    }
    const int w_ind_k = a.w_ind[k];  // (wapply_column's two scalars: see there)
    const float lamk_k = a.lamk[k];
    SNMF_PIN();
    if (stop_now) return;
    while (it_i < 8 * nI) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int c = cb;;) {  // (chunks in ascending order, as k_reduce adds them)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (c + j < ce) {
                    s0 += (double)x[j][0];
                    s1 += (double)x[j][1];
                    s2 += (double)x[j][2];
                    s3 += (double)x[j][3];
                }
            c += 16;
            if (c >= ce) break;
            load_batch(c);
        }
        part[(size_t)it_i * 4 + 0] = s0;
        part[(size_t)it_i * 4 + 1] = s1;
        part[(size_t)it_i * 4 + 2] = s2;
        part[(size_t)it_i * 4 + 3] = s3;
        it_i += 768;
        if (it_i < 8 * nI) {
            item_geo(it_i, cb, ce, p);
            load_batch(cb);
        }
    }
    // row sum of H (KL): eight chunk groups as well
    if (NMAT == 1 && tid >= 760) {  // (threads of the last wave: it has the fewest items above)
        const int g = tid - 760;
        const int cb = (int)(((long long)ra.n_chunks * g) / 8), ce = (int)(((long long)ra.n_chunks * (g + 1)) / 8);
        double sk = 0.0;
        int c = cb;
        for (; c + 8 <= ce; c += 8) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = ra.spart[(size_t)(c + j) * a.rp + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) sk += (double)x[j];
        }
        for (; c < ce; ++c) sk += (double)ra.spart[(size_t)c * a.rp + k];
        skp[g] = sk;
    }
    // objective partials (k_reduce's last block: strided, then a tree)
    if (tid < 256) {
        double d = pd0, h = ph0;
        if (ra.do_obj) {
            for (int c = tid + 256; c < ra.n_part; c += 256) {
                d += ra.part[2 * c];
                h += ra.part[2 * c + 1];
            }
        }
        red2[0][tid] = d;
        red2[1][tid] = h;
    }
    __syncthreads();
    if (tid >= 256) return;  // (ended waves do not count at later barriers)
    for (int idx = tid; idx < nI * 4; idx += 256) {  // element idx of this slice: (matrix m, row 4 e_lo + ...)
        double t = 0.0;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) t += part[((size_t)gg * nI) * 4 + idx];
        const int m = idx / (4 * nE), fo = idx - m * 4 * nE;
        QP[m * a.Fp + 4 * e_lo + fo] = t;
    }
    if (S > 1) {
        // this slice -> qp_buf; the last arriver of the column gathers the S slices and goes on alone.  The slices come from other
        // XCDs' L2s: agent-scope (sc1, write-through) stores acknowledged before the arrival is counted, sc1 loads on the other side --
        // NOT fences: a __threadfence() here writes back everything dirty in the XCD's L2 (the H step's output) and made the launch
        // 34-46 us where it had been 11-17
        double* qg = ra.qp_buf + (size_t)k * NMAT * a.Fp;
        for (int idx = tid; idx < nI * 4; idx += 256) {
            const int m = idx / (4 * nE), fo = idx - m * 4 * nE;
            __hip_atomic_store(qg + m * a.Fp + 4 * e_lo + fo, QP[m * a.Fp + 4 * e_lo + fo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) stress_jitter();  // (-DSNMF_STRESS builds only)
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ra.fin_cnt + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            skp[8] = (old % (unsigned)S == (unsigned)S - 1u) ? 1.0 : 0.0;
        }
        __syncthreads();
        if (skp[8] == 0.0) return;
        for (int idx = tid; idx < NMAT * a.Fp; idx += 256) QP[idx] = __hip_atomic_load(qg + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    // (k_reduce's tree 128, 64, ..., 1 -- red[t] += red[t + st] for t < st -- with the same pairs: wave 0 holds elements t, t + 64, t + 128,
    //  t + 192 of the array, adds 128 and 64 apart in registers and the rest by shuffles: two barriers instead of eight)
    if (tid < 64) {
        double d0 = red2[0][tid] + red2[0][tid + 128], d1 = red2[0][tid + 64] + red2[0][tid + 192];
        double h0 = red2[1][tid] + red2[1][tid + 128], h1 = red2[1][tid + 64] + red2[1][tid + 192];
        d0 += d1;
        h0 += h1;
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) {
            d0 += __shfl_down(d0, st, 64);
            h0 += __shfl_down(h0, st, 64);
        }
        if (tid == 0) {
            red2[0][0] = d0;
            red2[1][0] = h0;
        }
    }
    __syncthreads();
    double sc[2];
    sc[0] = ra.do_obj ? red2[0][0] : 0.0;
    sc[1] = ra.do_obj ? (ra.use_sh_const ? ra.sh_const : red2[1][0]) : 0.0;
    if (k == 0 && tid == 0) {
        double* scg = ra.stats + cstride + a.rp;
        scg[0] = sc[0];
        scg[1] = sc[1];
    }
    if (a.check_it > 0) {
        const bool stopnow = conv_test(sc, a.divh, a.costh, a.st, a.check_it, a.conv_eps, k == 0 && tid == 0, &last_cost);
        if (stopnow) return;
    }
    double sk = 0.0;
    if (NMAT == 1) {
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) sk += skp[gg];
    }
    wapply_column(a, k, tid, QP, NMAT == 2 ? QP + a.Fp : nullptr, sk, red, wpre, pre_w, w_ind_k, lamk_k);
}


#ifdef SNMF_AUX_KERNELS
// Convergence check alone (H-only mode and the final objective pass): one thread.
static __global__ void k_check(const double* stats, size_t sc_off, double* divh, double* costh, DevState* st, int it,
                        double conv_eps) {
    if (st->stop) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) conv_test(stats + sc_off, divh, costh, st, it, conv_eps, true);
}

// ---- small utility kernels -------------------------------------------------------------------
// h = h .* wn'  (src/sparse_nmf.m:160), H in [Tp][rp] layout
static __global__ void k_scale_h(float* H, const double* wn, int rp, int r, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(i % rp);
        if (k < r) H[i] = (float)((double)H[i] * wn[k]);
    }
}

// partial sums of S.*H over the real entries (W-only mode: constant over the iterations)
static __global__ __launch_bounds__(256) void k_sum_sh(const float* H, const float* S, const float* lamk, int rp, int r, int T,
                                                double* out /*[grid]*/) {
    __shared__ double red[256];
    double s = 0.0;
    const size_t n = (size_t)rp * T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int k = (int)(i % rp);
        if (k < r) s += (double)(S ? S[i] : lamk[k]) * (double)H[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// k_sum_sh partials -> stats.sh (tiny kernel: 256 doubles)
static __global__ void k_fold_sh(const double* part, int n, double* sc, const int* stop) {
    if (*stop) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        sc[1] = s;
    }
}

// per-solve bookkeeping of the online stream: iteration count and last recorded cost
static __global__ void k_collect(const DevState* st, const double* costh, int n_solves, int max_iter, int cost_check,
                          int* n_iter_out, double* cost_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_solves) {
        n_iter_out[b] = st[b].stop ? st[b].n_iter : max_iter;
        cost_out[b] = (cost_check && st[b].n_iter > 0) ? costh[(size_t)b * max_iter + st[b].n_iter - 1] : 0.0;
    }
}
#endif  // SNMF_AUX_KERNELS (k_check, k_scale_h, k_sum_sh, k_fold_sh, k_collect)
// H[0][(s*tps + t)*rp + k] = H0[t*r + k] * wn[k]: the same initial activations for every solve,
// already rescaled by the column norms of init_w (src/sparse_nmf.m:160)
template <typename TIn>
__global__ void k_tile_h0(const TIn* __restrict__ H0, const double* __restrict__ wn, int r, int rp, int tps,
                          int n_solves, float* __restrict__ H) {
    const size_t n = (size_t)n_solves * tps * rp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % rp);
        const size_t col = i / rp;
        const int t = (int)(col % tps);
        // same two roundings as the single-solve path (pack to fp32, then scale in fp64)
        H[i] = k < r ? (float)((double)(float)H0[(size_t)t * r + k] * wn[k]) : 0.f;
    }
}

// Layout conversion: user column-major (rows x cols, ld) -> padded [colsP][rowsP] fp32, optional floor.
// (TDst = double only for the W master copy, which keeps the reference's fp64 range: see k_wapply.)
template <typename TIn, typename TDst = float>
__global__ void k_pack(const TIn* __restrict__ src, int64_t ld, int rows, int cols, TDst* __restrict__ dst, int rowsP,
                       int colsP, float floor_val, int do_floor) {
    const size_t n = (size_t)rowsP * colsP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int rr = (int)(i % rowsP);
        const size_t c = i / rowsP;
        TDst v = (TDst)0;
        if (rr < rows && c < (size_t)cols) {
            v = (TDst)src[(size_t)c * ld + rr];
            if (do_floor) v = v > (TDst)floor_val ? v : (TDst)floor_val;
        }
        dst[i] = v;
    }
}
template <typename TOut, typename TSrc = float>
__global__ void k_unpack(const TSrc* __restrict__ src, int rowsP, int rows, int cols, TOut* __restrict__ dst,
                         int64_t ld) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int rr = (int)(i % rows);
        const size_t c = i / rows;
        dst[c * ld + rr] = (TOut)src[c * rowsP + rr];
    }
}

// ---- MDI helpers (src/snmf_mdi.m) ---------------------------------------------------------------
#ifdef SNMF_AUX_KERNELS
// v = max(v .* M, flr) on the real entries (:175); pads stay zero
static __global__ void k_mdi_start(float* V, const float* __restrict__ M, int Fp, int F, int T, float flr) {
    const size_t n = (size_t)Fp * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if ((int)(i % Fp) < F) V[i] = fmaxf(V[i] * M[i], flr);
}
#endif  // SNMF_AUX_KERNELS (k_mdi_start)

// v_MDI = max(v.*M + Nt .* max(w*h,flr) .* (1-M), flr),  Nt = sum(v.*M) ./ max(sum(max(w*h,flr).*M), flr)  (:298-306)
// One 256-thread workgroup per group of NC columns; W from its fp32 column-major copy, h columns in LDS.
template <int NC>
__global__ __launch_bounds__(256) void k_mdi_final(const float* __restrict__ V, const float* __restrict__ M,
                                                   const float* __restrict__ Wcf, const float* __restrict__ H, int F,
                                                   int Fp, int r, int rp, int T, float flr, float* __restrict__ out) {
    extern __shared__ float sm[];
    float* hs = sm;                       // [NC][rp]
    float* lam = hs + NC * rp;            // [NC][F]
    double* red = reinterpret_cast<double*>(lam + NC * F + ((NC * F + NC * rp) & 1));  // [2][NC][4 waves]
    const int t0 = blockIdx.x * NC, tid = threadIdx.x;
    for (int i = tid; i < NC * rp; i += 256) {
        const int c = i / rp, k = i - c * rp;
        hs[i] = (t0 + c < T) ? H[(size_t)(t0 + c) * rp + k] : 0.f;
    }
    __syncthreads();
    double sa[NC], sb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) sa[c] = sb[c] = 0.0;
    for (int f = tid; f < F; f += 256) {
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
        for (int k = 0; k < r; ++k) {
            const float wv = Wcf[(size_t)k * Fp + f];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(wv, hs[c * rp + k], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float l = fmaxf(acc[c], flr);
            lam[c * F + f] = l;
            if (t0 + c < T) {
                const float mk = M[(size_t)(t0 + c) * Fp + f];
                sa[c] += (double)(V[(size_t)(t0 + c) * Fp + f] * mk);
                sb[c] += (double)(l * mk);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double a = sa[c], b = sb[c];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o, 64);
            b += __shfl_down(b, o, 64);
        }
        if ((tid & 63) == 0) {
            red[(c * 4 + (tid >> 6)) * 2] = a;
            red[(c * 4 + (tid >> 6)) * 2 + 1] = b;
        }
    }
    __syncthreads();
    for (int c = 0; c < NC; ++c) {
        if (t0 + c >= T) break;
        double a = 0.0, b = 0.0;
        for (int q = 0; q < 4; ++q) {
            a += red[(c * 4 + q) * 2];
            b += red[(c * 4 + q) * 2 + 1];
        }
        const float Nt = (float)(a / fmax(b, (double)flr));
        for (int f = tid; f < F; f += 256) {
            const size_t i = (size_t)(t0 + c) * Fp + f;
            const float mk = M[i];
            out[(size_t)(t0 + c) * F + f] = fmaxf(V[i] * mk + Nt * lam[c * F + f] * (1.f - mk), flr);
        }
    }
}

// ---- small-rank family (snmf_smallr.h): what the host's geometry selection needs of it -----------------------------------------
constexpr int kSrWaves = 8;  // waves per workgroup (two per SIMD), one workgroup per CU
// LDS of k_hstep_sr: Wk4 image [NK][Fq/8][2][32][4] | wx [rp] | 1 ./ dph [rp] | lambda_k [rp] | partial slots [2 buffers][8 waves][NK][16][64]
// | ratio of the extra row [2][32] | progress words [32] | doubles [2][8]
inline size_t sr_hstep_lds_bytes(int nk, int Fq, int rp) {
    return ((size_t)nk * Fq * 32 + 3 * (size_t)rp + (size_t)2 * kSrWaves * nk * 1024 + 64 + 32) * 4 + 2 * kSrWaves * sizeof(double);
}
// LDS of k_wstats_sr: wx [rp] | row sums [rp] | extra row of the slab [rp] | pad [16] | doubles [8]
inline size_t sr_wstats_lds_bytes(int rp) { return ((size_t)3 * rp + 16) * 4 + kSrWaves * sizeof(double); }

}  // namespace snmf
