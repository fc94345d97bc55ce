// snmf_frontend.h -- on-device spectrogram front-end: the step right before
// the solver in every call stack of the reference, so that V is produced in HBM and only the raw
// audio crosses PCIe.
//   src/stft_fft.m:15-37        framing, pre-emphasis, window, zero-padded FFT, |.|, DC-bin value
//   run_basis_train.m:60-63     (drop all-zero columns,) splice, .^pow + nonzerofloor
//   src/frame_splice.m:1-24     context splicing
//   run_basis_train.m:70-78     Mel projection (the Mel matrix itself is a host-side parameter table)
// HBM-bound byte work (2.5 KB of samples in, 2 KB of features out per frame); the FFT is a
// radix-2 Stockham autosort in LDS, one 256-thread workgroup per frame.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snmf {

struct StftArgs {
    const float* s;      // samples (device)
    int64_t n_samples;
    int sz, shift, dcbin;
    float preemph;
    const float* win;    // [sz]
    const float2* tw;    // [N/2] exp(-2*pi*i*q/N)
    float powv, floorv;  // floorv is added here only when there is no splicing pass
    float* out;          // column t at out + t*ld
    int64_t ld;
    int n_frames;
};

template <int LOGN>
__global__ __launch_bounds__(256) void k_stft(StftArgs a) {
    constexpr int N = 1 << LOGN;
    __shared__ float2 bufA[N];
    __shared__ float2 bufB[N];
    const int t = blockIdx.x;
    if (t >= a.n_frames) return;
    const int64_t start = (int64_t)t * a.shift;  // 0-based first sample of frame t (size_crnt - 1)
    for (int n = threadIdx.x; n < N; n += 256) {
        float x = 0.f;
        if (n < a.sz) {
            const float cur = a.s[start + n];
            const float prev = n > 0 ? a.s[start + n - 1] : 0.f;  // filter([1 -preemph],1,.) with zero state
            x = (cur - a.preemph * prev) * a.win[n];
        }
        bufA[n] = make_float2(x, 0.f);
    }
    __syncthreads();
    float2* x = bufA;
    float2* y = bufB;
    // Stockham autosort, decimation in frequency: result in natural order after LOGN stages
    for (int l = N / 2, m = 1; l >= 1; l >>= 1, m <<= 1) {
        const int tstep = N / (2 * l);
        for (int idx = threadIdx.x; idx < N / 2; idx += 256) {
            const int j = idx / m, k = idx - j * m;
            const float2 c0 = x[k + j * m];
            const float2 c1 = x[k + j * m + l * m];
            const float2 w = a.tw[j * tstep];
            const float2 d = make_float2(c0.x - c1.x, c0.y - c1.y);
            y[k + 2 * j * m] = make_float2(c0.x + c1.x, c0.y + c1.y);
            y[k + 2 * j * m + m] = make_float2(w.x * d.x - w.y * d.y, w.x * d.y + w.y * d.x);
        }
        __syncthreads();
        float2* tmp = x;
        x = y;
        y = tmp;
    }
    float* o = a.out + (int64_t)t * a.ld;
    for (int f = threadIdx.x; f <= N / 2; f += 256) {
        const float2 c = x[f];
        float mag = sqrtf(c.x * c.x + c.y * c.y);  // abs(S_frame), src/stft_fft.m:27
        if (f < a.dcbin) mag = 0.000001f;          // :31
        float v;
        if (a.powv == 2.f) v = mag * mag;
        else if (a.powv == 1.f) v = mag;
        else v = powf(mag, a.powv);
        o[f] = v + a.floorv;  // run_basis_train.m:63
    }
}

// src/frame_splice.m:8-23 on the powered magnitudes, then + nonzerofloor (run_basis_train.m:62-63):
// out[(S+s)*K + f, t] = src[f, t+s],  out[(S-s)*K + f, t] = src[f, t-s]  (zero outside 1..T)
static __global__ void k_splice(const float* __restrict__ src, int64_t ld_src, int K, int T, int S, float floorv,
                         float* __restrict__ out, int64_t ld_out) {
    const int64_t n = (int64_t)(2 * S + 1) * K * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i % ((2 * S + 1) * K));
        const int t = (int)(i / ((2 * S + 1) * K));
        const int blk = row / K, f = row - blk * K;
        const int ts = t + (blk - S);
        float v = 0.f;
        if (ts >= 0 && ts < T) {
            // the reference's if/elseif order: when t-s < 1 the upper block takes t+s (valid or not is
            // impossible here since t+s < T is checked by ts < T) and the lower block is zero
            v = src[(int64_t)ts * ld_src + f];
        }
        out[(int64_t)t * ld_out + row] = v + floorv;
    }
}

// run_basis_train.m:70-78: out[k*M + m, t] = sum_f mel[m, f] * V[k*n + f, t]; one thread per output
static __global__ void k_mel(const float* __restrict__ mel /*[M][n] row-major*/, int M, int n, int K,
                      const float* __restrict__ V, int64_t ldv, int T, float* __restrict__ out, int64_t ldo) {
    const int64_t tot = (int64_t)K * M * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i % (K * M));
        const int t = (int)(i / (K * M));
        const int k = row / M, m = row - k * M;
        const float* mr = mel + (int64_t)m * n;
        const float* vc = V + (int64_t)t * ldv + (int64_t)k * n;
        float s = 0.f;
        for (int f = 0; f < n; ++f) s += mr[f] * vc[f];
        out[(int64_t)t * ldo + row] = s;
    }
}

// v = max(v, flr) on the real F x T entries of a padded [T][Fp] matrix (src/sparse_nmf.m:169)
static __global__ void k_floor_real(float* V, int Fp, int F, int T, float flr) {
    const int64_t n = (int64_t)Fp * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if ((int)(i % Fp) < F) V[i] = fmaxf(V[i], flr);
}

// ---- TF_DD (src/TF_DD.m:1-9, called by run_basis_train.m:64-67 when p.domain_DD is set): a one-tap recursive average
// along the frame axis, row by row:  X_DD(:,1) = X(:,1);  X_DD(:,l) = a X_DD(:,l-1) + (1-a) X(:,l).
// Rows are independent and the matrix is column-major, so thread <-> row gives coalesced frame-by-frame accesses; the
// frame axis is cut into chunks of kDdChunk frames scanned in parallel:
//   k_tfdd_carry : every (chunk, row) runs the recursion over its chunk from state 0 -> its carry c
//   k_tfdd_state : per row, the states at the chunk starts: S_{j+1} = a^len_j S_j + c_j  (sequential over the few chunks);
//                  S_0 = X(:,1), which makes the first column come out as itself
//   k_tfdd_apply : every (chunk, row) re-runs its chunk from the true start state and writes the result
// State and coefficients in fp64 (the features are fp32; the recursion then adds no error of its own).
constexpr int kDdChunk = 256;
static __global__ __launch_bounds__(256) void k_tfdd_carry(const float* __restrict__ X, int64_t ld, int F, int T, double a, double* __restrict__ carry) {
    const int f = blockIdx.y * 256 + threadIdx.x, j = blockIdx.x;
    if (f >= F) return;
    const int t0 = j * kDdChunk, t1 = min(T, t0 + kDdChunk);
    double s = 0.0;
    for (int t = t0; t < t1; ++t) s = a * s + (1.0 - a) * (double)X[(int64_t)t * ld + f];
    carry[(int64_t)j * F + f] = s;
}
static __global__ __launch_bounds__(256) void k_tfdd_state(const float* __restrict__ X, int F, int T, double a, double* __restrict__ carry /* in: carries, out: start states */) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int nch = (T + kDdChunk - 1) / kDdChunk;
    double s = (double)X[f];  // state "before" column 1: a x + (1-a) x = x
    for (int j = 0; j < nch; ++j) {
        const int len = min(T, (j + 1) * kDdChunk) - j * kDdChunk;
        const double c = carry[(int64_t)j * F + f];
        carry[(int64_t)j * F + f] = s;
        s = pow(a, (double)len) * s + c;
    }
}
static __global__ __launch_bounds__(256) void k_tfdd_apply(const float* __restrict__ X, int64_t ld, int F, int T, double a, const double* __restrict__ state,
                                                    float* __restrict__ out, int64_t ldo) {
    const int f = blockIdx.y * 256 + threadIdx.x, j = blockIdx.x;
    if (f >= F) return;
    const int t0 = j * kDdChunk, t1 = min(T, t0 + kDdChunk);
    double s = state[(int64_t)j * F + f];
    for (int t = t0; t < t1; ++t) {
        s = a * s + (1.0 - a) * (double)X[(int64_t)t * ld + f];
        out[(int64_t)t * ldo + f] = (float)s;
    }
}

}  // namespace snmf
