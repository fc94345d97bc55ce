// snmf_online.h -- the online separation loop around the per-frame solve, kept on the device
// (BASELINE config 3).  Reference (shipped configuration: blk_len_sep = 1,
// Splice = 0, B_sep_mode = 'DFT'):
//   src/bnmf_sep_event_RT_IS16.m:65-81     frame STFT: |Y|^pow with DC bins zeroed + floor, phase
//   src/bnmf_sep_event_RT_IS16.m:158-202   reconstructions  Xm_hat = B_x*A_x,  Dm_hat = B_d*A_d
//   src/blk_sparse.m:1-37                  Hoyer block sparsity Q
//   src/bnmf_sep_event_RT_IS16.m:220-261   adaptive beta, smoothed noise PSD, Wiener / MMSE gain
//   src/bnmf_sep_event_RT_IS16.m:263-347   noise-reference rings, r_up, dictionary re-assembly
//   src/synth_ifft_buff.m:1-32             inverse STFT of a frame
//   src/NTF_sep_event_RT.m:104-124         overlap-add, int16 output
// Everything here is vector work on F ~ 513 bins per frame: latency-bound, one workgroup per
// frame for the transforms and ONE workgroup for the sequential post-solve step.  The solves
// themselves are the engine's kernels (k_hsolve_small for the frame, k_wstats/k_wapply for the
// adaptation).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdint.h>

namespace snmf {

struct OnlineStatus {  // written once per frame, read back by the host
    int trig, do_solve, n_up, n_iter;
    float beta, A_x_mag, A_d_mag, Q_control;
};

struct OnlineDev {     // device-resident scalar state of the loop
    int n_push;        // pushes into the noise-reference rings (lambda_d_blk / Ad_blk)
    int update_switch; // src/init_buff.m:42
    int pad0, pad1;
};

// radix-2 Stockham autosort FFT of N = 2^LOGN points held in LDS; returns the buffer with the result
template <int LOGN>
__device__ __forceinline__ float2* fft_lds(float2* x, float2* y, const float2* __restrict__ tw) {
    constexpr int N = 1 << LOGN;
    for (int l = N / 2, m = 1; l >= 1; l >>= 1, m <<= 1) {
        const int tstep = N / (2 * l);
        for (int idx = threadIdx.x; idx < N / 2; idx += blockDim.x) {
            const int j = idx / m, k = idx - j * m;
            const float2 c0 = x[k + j * m];
            const float2 c1 = x[k + j * m + l * m];
            const float2 w = tw[j * tstep];
            const float2 d = make_float2(c0.x - c1.x, c0.y - c1.y);
            y[k + 2 * j * m] = make_float2(c0.x + c1.x, c0.y + c1.y);
            y[k + 2 * j * m + m] = make_float2(w.x * d.x - w.y * d.y, w.x * d.y + w.y * d.x);
        }
        __syncthreads();
        float2* t = x;
        x = y;
        y = t;
    }
    return x;
}

struct OStftArgs {
    const float* sig;  // [(sz - hop) history | n_frames * hop new samples]; frame i starts at i*hop
    int sz, hop, dcbin;
    float preemph;
    const float* win;
    const float2* tw;
    float powv, floorv;
    float* Ym;         // column i at Ym + i*ld
    float2* Yph;       // exp(i*angle(Y)) per bin, same layout
    int64_t ld;
    int n_frames;
};

// src/bnmf_sep_event_RT_IS16.m:65-81
template <int LOGN>
__global__ __launch_bounds__(256) void k_ostft(OStftArgs a) {
    constexpr int N = 1 << LOGN;
    __shared__ float2 bufA[N];
    __shared__ float2 bufB[N];
    const int t = blockIdx.x;
    if (t >= a.n_frames) return;
    const float* s = a.sig + (int64_t)t * a.hop;
    for (int n = threadIdx.x; n < N; n += 256) {
        float x = 0.f;
        if (n < a.sz) {
            const float cur = s[n];
            const float prev = n > 0 ? s[n - 1] : 0.f;  // filter([1 -preemph],1,y), zero state (:66)
            x = (cur - a.preemph * prev) * a.win[n];     // :67
        }
        bufA[n] = make_float2(x, 0.f);
    }
    __syncthreads();
    const float2* X = fft_lds<LOGN>(bufA, bufB, a.tw);
    float* om = a.Ym + (int64_t)t * a.ld;
    float2* op = a.Yph + (int64_t)t * a.ld;
    for (int f = threadIdx.x; f <= N / 2; f += 256) {
        const float2 c = X[f];
        const float mag = sqrtf(c.x * c.x + c.y * c.y);
        float v;
        if (a.powv == 2.f) v = mag * mag;
        else if (a.powv == 1.f) v = mag;
        else v = powf(mag, a.powv);
        if (f < a.dcbin) v = 0.f;                        // :74
        om[f] = v + a.floorv;                            // :77
        op[f] = mag > 0.f ? make_float2(c.x / mag, c.y / mag) : make_float2(1.f, 0.f);  // angle(0) = 0
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// fixed-order block sum, result in every thread; red holds one double per wave
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max_f(float v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)v;
    __syncthreads();
    float s = (float)red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, (float)red[i]);
    return s;
}

struct OPostArgs {
    const float* A;       // [r] activations of this frame (solver's H buffer, first r of rp)
    const DevState* hst;  // the frame solve's state (n_iter)
    const float* B;       // [F x r] column-major, current [B_DFT_x | B_DFT_d]
    const float* recon;   // [2][F] per frame: B_x*A_x and B_d*A_d from the frame solve (NULL: computed here from B)
    const float* Ym;      // [F]
    float* lambda_dav;    // [F] state
    float* Xm_tilde;      // [F] state
    float* r_blk;         // [Pl][F] ring of SNR_local columns
    float* ldblk;         // [ma][F] ring  lambda_d_blk
    float* adblk;         // [ma][Ra] ring Ad_blk
    uint8_t* rup;         // [Ra]
    OnlineDev* dev;
    OnlineStatus* status;
    float* Xt_out;        // [F] G .* Ym of this frame
    float* Xh_out;        // [F] Xm_hat_sum (may be NULL)
    float* Dh_out;        // [F] Dm_hat_sum (may be NULL)
    int F, Rx, Rd, Ra, ma, Pl, Pk, dcbin, gap;
    int l;                // 1-based frame index
    int blk_sparse, adapt, wiener, init_N_len, switch_at;
    float alpha_p, alpha_eta, alpha_d, beta0, beta_max, Ar_up, flr;
    // B_sep_mode = 'Mel' (:106-120): the solve ran on Mel features
    const float* melmat;  // [n1][F] row-major (g.melmat)
    const float* Ymel;    // [n1] this frame's normalised Mel features
    const float* Bmf;     // [n1 x r] fp32 mirror of [B_Mel_x | B_Mel_d] (reconstruction fallback)
    int mel, mel_conv, n1;
    int recon_len;        // rows of one reconstruction in `recon` (F, or n1 with MelConv)
    int n;                // frames handled by this launch, one after the other (> 1 only without adaptation)
    int a_stride;         // distance between the activation vectors of consecutive frames
};

// Everything between the frame solve and the inverse STFT, src/bnmf_sep_event_RT_IS16.m:158-292, for one frame.
__device__ __forceinline__ void opost_frame(const OPostArgs& a, float* sm, double* red) {
    const int F = a.F, r = a.Rx + a.Rd;
    float* sA = sm;
    float* Xs = sA + r;
    float* Ds = Xs + F;
    float* Q = Ds + F;
    float* rs1 = Q + F;
    float* rs2 = rs1 + F;
    float* Gs = rs2 + F;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n_push0 = a.dev->n_push, sw0 = a.dev->update_switch;
    for (int k = tid; k < r; k += nt) sA[k] = a.A[k];
    __syncthreads();
    // A_x_mag, A_d_mag (:228-229)
    double sx = 0.0, sd = 0.0;
    for (int k = tid; k < r; k += nt) {
        if (k < a.Rx) sx += (double)sA[k];
        else sd += (double)sA[k];
    }
    sx = block_sum_d(sx, red);
    sd = block_sum_d(sd, red);
    const float A_x_mag = (float)(sx / a.Rx), A_d_mag = (float)(sd / a.Rd);
    // Xm_hat_sum = B_x*A_x, Dm_hat_sum = B_d*A_d (:158-202; any class partition sums to these)
    float* Ymd = Gs + F;  // [F] Ym_Mel_DFT (first frame only), then [3*n1] Mel-domain vectors
    if (a.mel && a.mel_conv) {
        // :165-171,:185-192: reconstructions in the Mel domain, mapped back with melmat'; :205-211 Ym_Mel_DFT
        float* Xm = Ymd + F;
        float* Dm = Xm + a.n1;
        float* Ym1 = Dm + a.n1;
        for (int m = tid; m < a.n1; m += nt) {
            float x, d;
            if (a.recon) {
                x = a.recon[m];
                d = a.recon[a.recon_len + m];
            } else {
                x = 0.f;
                d = 0.f;
                const float* b = a.Bmf + m;
                for (int k = 0; k < a.Rx; ++k) x = fmaf(b[(size_t)k * a.n1], sA[k], x);
                b += (size_t)a.Rx * a.n1;
                for (int k = 0; k < a.Rd; ++k) d = fmaf(b[(size_t)k * a.n1], sA[a.Rx + k], d);
            }
            Xm[m] = x;
            Dm[m] = d;
            Ym1[m] = a.Ymel[m];
        }
        __syncthreads();
        for (int f = tid; f < F; f += nt) {
            float x = 0.f, d = 0.f, y = 0.f;
            for (int m = 0; m < a.n1; ++m) {
                const float mm = a.melmat[(size_t)m * F + f];
                x = fmaf(mm, Xm[m], x);
                d = fmaf(mm, Dm[m], d);
                y = fmaf(mm, Ym1[m], y);
            }
            Xs[f] = x;
            Ds[f] = d;
            Ymd[f] = y;
        }
    } else if (a.recon) {
        for (int f = tid; f < F; f += nt) {
            Xs[f] = a.recon[f];
            Ds[f] = a.recon[a.recon_len + f];
            Ymd[f] = a.Ym[f];
        }
    } else {
        for (int f = tid; f < F; f += nt) {
            const float* b = a.B + f;
            float x = 0.f, d = 0.f;
            for (int k = 0; k < a.Rx; ++k) x = fmaf(b[(size_t)k * F], sA[k], x);
            b += (size_t)a.Rx * F;
            for (int k = 0; k < a.Rd; ++k) d = fmaf(b[(size_t)k * F], sA[a.Rx + k], d);
            Xs[f] = x;
            Ds[f] = d;
            Ymd[f] = a.Ym[f];
        }
    }
    __syncthreads();
    // ---- src/blk_sparse.m ----
    if (a.blk_sparse) {
        float mx = 0.f;
        for (int f = tid; f < F; f += nt) {
            const float s = Xs[f] / fmaxf(Ds[f], a.flr);  // :10
            rs1[f] = s;
            mx = fmaxf(mx, s);
        }
        mx = block_max_f(mx, red);
        float* col = a.r_blk + (size_t)((a.l - 1) % a.Pl) * F;  // newest column of the ring (:14)
        for (int f = tid; f < F; f += nt) {
            col[f] = rs1[f] / mx;                                // :12
            Q[f] = f < a.dcbin ? 0.f : 0.1f;                     // :16
        }
        __syncthreads();
        if (a.l > a.Pl) {
            for (int f = tid; f < F; f += nt) {
                float s1 = 0.f, s2 = 0.f;
                for (int c = 0; c < a.Pl; ++c) {
                    const float v = a.r_blk[(size_t)c * F + f];
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
                rs1[f] = s1;
                rs2[f] = s2;
            }
            __syncthreads();
            const int k2 = a.Pk / 2, gN2 = (a.gap - 1) / 2;
            const int kfirst = k2 + a.dcbin, klast = F - k2;  // 1-based, :20
            const int nwin = klast >= kfirst ? (klast - kfirst) / a.gap + 1 : 0;
            const double sqn = sqrt((double)a.Pl * (double)a.Pk);
            for (int j = tid; j < nwin; j += nt) {
                const int k = kfirst + j * a.gap;
                double l1 = 0.0, l2 = 0.0;
                for (int row = k - k2; row < k + k2; ++row) {  // 1-based rows k-k2+1 .. k+k2
                    l1 += (double)rs1[row];
                    l2 += (double)rs2[row];
                }
                Gs[j] = (float)((sqn - l1 / sqrt(l2)) / (sqn - 1.0));  // :26
            }
            __syncthreads();
            if (gN2 >= 1) {
                // blk_gap >= 3: window k reads Q(k-1), which no other window writes (window k-gap ends at
                // k-gap+gN2 < k-1), so the recursion of :28 sees the initial value and windows are independent
                for (int j = tid; j < nwin; j += nt) {
                    const int k = kfirst + j * a.gap;
                    const float qprev = (k - 2) < a.dcbin ? 0.f : 0.1f;
                    const float pv = a.alpha_p * qprev + (1.f - a.alpha_p) * Gs[j];
                    for (int i = k - gN2 - 1; i <= k + gN2 - 1; ++i) Q[i] = pv;  // :29-30
                }
            } else if (tid == 0) {
                // blk_gap = 1: a genuine first-order recursion along frequency
                for (int j = 0; j < nwin; ++j) {
                    const int k = kfirst + j;
                    Q[k - 1] = a.alpha_p * Q[k - 2] + (1.f - a.alpha_p) * Gs[j];
                }
            }
            __syncthreads();
            const float qv = Q[a.Pk + a.dcbin - 1];
            __syncthreads();
            for (int f = tid; f < a.Pk - 1; f += nt) Q[f] = qv;  // :32
            __syncthreads();
        }
        for (int f = tid; f < a.dcbin; f += nt) Q[f] = 0.f;      // :36
    } else {
        for (int f = tid; f < F; f += nt) Q[f] = 1.f;            // :217
    }
    __syncthreads();
    double qs = 0.0;
    for (int f = tid; f < F; f += nt) qs += (double)Q[f];
    qs = block_sum_d(qs, red);
    const float meanQ = (float)(qs / F);
    // ---- gain (:221-261) ----
    float beta = (float)(20.0 * log10((double)A_d_mag / (double)A_x_mag)) * a.beta0;  // :230-231
    if (beta < a.beta0) beta = a.beta0;
    else if (beta >= a.beta_max) beta = a.beta_max;
    const bool init = a.l <= a.init_N_len;
    for (int f = tid; f < F; f += nt) {
        const float ym = a.Ym[f];
        float ld = a.l == 1 ? Ymd[f] : a.lambda_dav[f];                   // :223-225 (Ym_Mel_DFT)
        ld = a.alpha_d * ld + (1.f - a.alpha_d) * Ds[f] * beta;           // :241
        a.lambda_dav[f] = ld;
        float G;
        if (a.wiener) {
            G = Xs[f] / (Xs[f] + Ds[f]);                                  // :245
        } else {
            float eta = (a.alpha_eta * a.Xm_tilde[f] + (1.f - a.alpha_eta) * Xs[f] * Q[f]) / fmaxf(ld, a.flr);  // :247
            eta = fmaxf(0.0031f, eta);                                    // :251
            G = eta / (eta + 1.f);
        }
        G = fminf(G, 1.f);                                                // :254 (min ignores NaN, as MATLAB's)
        if (init) G = a.flr;                                              // :256-258
        Gs[f] = G;
        const float xt = G * ym;                                          // :260
        a.Xm_tilde[f] = xt;
        a.Xt_out[f] = xt;
        if (a.Xh_out) a.Xh_out[f] = Xs[f];
        if (a.Dh_out) a.Dh_out[f] = Ds[f];
    }
    const float A_x_eff = init ? a.flr : A_x_mag;                         // :258
    const float Q_control = (1.f - meanQ) * a.Ar_up;                      // :264
    const bool trig = a.adapt && (Q_control * A_d_mag > A_x_eff);         // :266
    int do_solve = 0, n_up = 0;
    __syncthreads();
    if (trig) {
        const int head = n_push0 % a.ma;  // overwrites the oldest column == shift + append (:282,:285)
        for (int f = tid; f < F; f += nt) {
            const float ym = a.Ym[f];
            const float mref = f < a.dcbin ? a.flr : 1.f - Gs[f];         // :271-272
            a.ldblk[(size_t)head * F + f] = init ? ym : ym * mref;        // :268-274
        }
        for (int k = tid; k < a.Ra; k += nt) a.adblk[(size_t)head * a.Ra + k] = sA[a.Rx + k];
        __syncthreads();
        int cnt = 0;
        for (int k = tid; k < a.Ra; k += nt) {
            double s = 0.0;
            for (int c = 0; c < a.ma; ++c) s += (double)a.adblk[(size_t)c * a.Ra + k];
            const bool up = (double)Q_control * (s / a.ma) > (double)A_x_eff;  // :288
            a.rup[k] = up ? 1 : 0;
            cnt += up;
        }
        n_up = (int)(block_sum_d((double)cnt, red) + 0.5);
        do_solve = sw0 == a.switch_at;                                    // :294
        if (tid == 0) {
            a.dev->n_push = n_push0 + 1;
            a.dev->update_switch = do_solve ? 1 : sw0 + 1;                // :343-345
        }
    }
    if (tid == 0) {
        OnlineStatus s;
        s.trig = trig;
        s.do_solve = do_solve;
        s.n_up = n_up;
        s.n_iter = a.hst->n_iter;
        s.beta = beta;
        s.A_x_mag = A_x_eff;
        s.A_d_mag = A_d_mag;
        s.Q_control = Q_control;
        *a.status = s;
    }
}

// One workgroup; dynamic LDS = (r + 7*F + 3*n1) floats.  The post-filter recurrences (smoothed noise PSD, the
// previous frame's G.*Y, the SNR ring) make the frames sequential, but when the dictionary is fixed
// (no adaptation) nothing the host must decide sits between them: the frame solves of a whole batch run
// in parallel first and this kernel then walks the batch in ONE launch.
__global__ __launch_bounds__(1024) void k_opost(OPostArgs a0) {
    extern __shared__ float sm[];
    __shared__ double red[16];
    for (int i = 0; i < a0.n; ++i) {
        OPostArgs a = a0;
        a.A += (size_t)i * a0.a_stride;
        if (a.recon) a.recon += (size_t)i * 2 * a0.recon_len;
        if (a.Ymel) a.Ymel += (size_t)i * a0.n1;
        a.hst += i;
        a.Ym += (size_t)i * a0.F;
        a.Xt_out += (size_t)i * a0.F;
        if (a.Xh_out) a.Xh_out += (size_t)i * a0.F;
        if (a.Dh_out) a.Dh_out += (size_t)i * a0.F;
        a.status += i;
        a.l += i;
        opost_frame(a, sm, red);
        __syncthreads();  // state written by this frame (global + LDS scratch) is visible to the next
    }
}

// Inputs of the adaptation solve (:296-335) in time order: V = lambda_d_blk, H = Ad_blk with the
// rows not flagged by r_up zeroed (the reference drops those rows/columns; a zero activation row
// contributes nothing to Lam, G or the cost, so the flagged columns see the same problem), and
// the engine's W-update mask = r_up.
__global__ void k_oprep(const float* __restrict__ ldblk, const float* __restrict__ adblk, const uint8_t* __restrict__ rup,
                        const OnlineDev* dev, int F, int Ra, int ma, float* __restrict__ Vad, float* __restrict__ Had,
                        uint8_t* __restrict__ w_ind) {
    const int oldest = dev->n_push % ma;
    const size_t nv = (size_t)F * ma, nh = (size_t)Ra * ma;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + nh + Ra; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nv) {
            const int c = (int)(i / F), f = (int)(i - (size_t)c * F);
            Vad[i] = ldblk[(size_t)((oldest + c) % ma) * F + f];
        } else if (i < nv + nh) {
            const size_t j = i - nv;
            const int c = (int)(j / Ra), k = (int)(j - (size_t)c * Ra);
            Had[j] = rup[k] ? adblk[(size_t)((oldest + c) % ma) * Ra + k] : 0.f;
        } else {
            const int k = (int)(i - nv - nh);
            w_ind[k] = rup[k];
        }
    }
}

// :106-120 for a batch of frames: Ym_Mel = melmat*Ym, normalised to unit norm (+1e-9) and scaled to ||Ym||.
// One workgroup per frame, one wave per group of outputs.
__global__ __launch_bounds__(256) void k_omel_frame(const float* __restrict__ Ym, const float* __restrict__ melmat, int F, int n1,
                                                    int n_frames, float* __restrict__ Ymel) {
    extern __shared__ float sm[];  // [n1] + 2
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (t >= n_frames) return;
    const float* y = Ym + (size_t)t * F;
    float tn2 = 0.f;
    for (int f = tid; f < F; f += 256) tn2 = fmaf(y[f], y[f], tn2);
    tn2 = wave_sum_f(tn2);
    __shared__ float part[4];
    if (lane == 0) part[w] = tn2;
    for (int m = w; m < n1; m += 4) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s = fmaf(melmat[(size_t)m * F + f], y[f], s);
        s = wave_sum_f(s);
        if (lane == 0) sm[m] = s;
    }
    __syncthreads();
    const float tn = sqrtf(part[0] + part[1] + part[2] + part[3]);
    float vn2 = 0.f;
    for (int m = tid; m < n1; m += 256) vn2 = fmaf(sm[m], sm[m], vn2);
    vn2 = wave_sum_f(vn2);
    __syncthreads();
    if (lane == 0) part[w] = vn2;
    __syncthreads();
    const float vn = sqrtf(part[0] + part[1] + part[2] + part[3]);
    for (int m = tid; m < n1; m += 256) Ymel[(size_t)t * n1 + m] = (sm[m] / vn + 1e-9f) * tn;
}

// Mel-mode inputs of the adaptation solve (:298-313): lambda_d_blk_Mel = melmat * lambda_d_blk in time order,
// Ad_blk rows masked by r_up, the update mask.  One workgroup per ring column.
__global__ __launch_bounds__(256) void k_oprep_mel(const float* __restrict__ ldblk, const float* __restrict__ adblk,
                                                   const uint8_t* __restrict__ rup, const OnlineDev* dev,
                                                   const float* __restrict__ melmat, int F, int n1, int Ra, int ma,
                                                   float* __restrict__ Vad, float* __restrict__ Had, uint8_t* __restrict__ w_ind) {
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int oldest = dev->n_push % ma;
    const float* col = ldblk + (size_t)((oldest + c) % ma) * F;
    for (int m = w; m < n1; m += 4) {
        float s = 0.f;
        for (int f = lane; f < F; f += 64) s = fmaf(melmat[(size_t)m * F + f], col[f], s);
        s = wave_sum_f(s);
        if (lane == 0) Vad[(size_t)c * n1 + m] = s;
    }
    for (int k = tid; k < Ra; k += 256) {
        Had[(size_t)c * Ra + k] = rup[k] ? adblk[(size_t)((oldest + c) % ma) * Ra + k] : 0.f;
        if (c == 0) w_ind[k] = rup[k];
    }
}

// B_DFT_d = [B_d_rem, B_d_tmp, B_d_fix] (:336): kept columns first, then the re-trained ones, then
// the columns beyond R_a taken from the original dictionary (:328).  One workgroup per column.
__global__ void k_oassemble(const double* __restrict__ Bd_old, const double* __restrict__ Wc, int Fp,
                            const double* __restrict__ Bfix, const uint8_t* __restrict__ rup, int F, int Ra, int Rd,
                            double* __restrict__ Bd_new, float* __restrict__ Bd_f32) {
    const int j = blockIdx.x;
    if (j >= Rd) return;
    const double* src;
    if (j >= Ra) {
        src = Bfix + (size_t)j * F;
    } else {
        int n_rem = 0;
        for (int k = 0; k < Ra; ++k) n_rem += rup[k] ? 0 : 1;
        const bool want_up = j >= n_rem;
        int need = want_up ? j - n_rem : j, k = 0;
        for (; k < Ra; ++k) {
            if ((rup[k] != 0) == want_up) {
                if (need == 0) break;
                --need;
            }
        }
        src = want_up ? Wc + (size_t)k * Fp : Bd_old + (size_t)k * F;
    }
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const double v = src[f];
        Bd_new[(size_t)j * F + f] = v;
        Bd_f32[(size_t)j * F + f] = (float)v;
    }
}

struct OIstftArgs {
    const float* mag;   // column i at mag + i*ld  (magnitude^pow domain)
    const float2* ph;
    int64_t ld;
    int n_frames, sz, dcb;
    float powv, scale, preemph;  // scale = overlapscale / N
    const float* win;
    const float2* tw;
    float* syn;         // frame i at syn + i*sz
};

// src/synth_ifft_buff.m:10-28 (+ the overlapscale of src/bnmf_sep_event_RT_IS16.m:363)
template <int LOGN>
__global__ __launch_bounds__(256) void k_oistft(OIstftArgs a) {
    constexpr int N = 1 << LOGN;
    __shared__ float2 bufA[N];
    __shared__ float2 bufB[N];
    const int t = blockIdx.x;
    if (t >= a.n_frames) return;
    const float* mg = a.mag + (int64_t)t * a.ld;
    const float2* ph = a.ph + (int64_t)t * a.ld;
    // real(ifft(X)) = real(fft(conj(X)))/N with X(N-k) = conj(X(k)) for k = 1..N/2-1 (:16-18)
    for (int k = threadIdx.x; k < N; k += 256) {
        const int kk = k <= N / 2 ? k : N - k;
        float m = kk < a.dcb ? 0.f : mg[kk];                    // :10
        if (a.powv == 2.f) m = sqrtf(m);                        // :11
        else if (a.powv != 1.f) m = powf(m, 1.f / a.powv);
        const float2 p = ph[kk];
        bufA[k] = make_float2(m * p.x, k <= N / 2 ? -m * p.y : m * p.y);
    }
    __syncthreads();
    float2* X = fft_lds<LOGN>(bufA, bufB, a.tw);
    float* o = a.syn + (int64_t)t * a.sz;
    if (a.preemph == 0.f) {
        for (int n = threadIdx.x; n < a.sz; n += 256) o[n] = X[n].x * a.scale * a.win[n];  // :19-24
    } else {
        for (int n = threadIdx.x; n < a.sz; n += 256) X[n].y = X[n].x * a.scale * a.win[n];
        __syncthreads();
        if (threadIdx.x == 0) {  // filter(1, [1 -preemph], .) (:26)
            float acc = 0.f;
            for (int n = 0; n < a.sz; ++n) {
                acc = X[n].y + a.preemph * acc;
                o[n] = acc;
            }
        }
    }
}

// Overlap-add of src/NTF_sep_event_RT.m:104-124 in closed form: the hop written at frame l is the sum
// over the frames l-q (q = nov-1 .. 0, oldest first, only frames > delay were ever accumulated) of
// their samples [q*hop, q*hop + hop).  syn holds nov-1 frames of the previous call, then the new ones.
__global__ void k_oola(const float* __restrict__ syn, int n_new, int l0, int delay, int sz, int hop, int nov, int i_first,
                       int n_out, float* __restrict__ outf, int16_t* __restrict__ out16) {
    const size_t n = (size_t)n_out * hop;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(e / hop), s = (int)(e - (size_t)j * hop);
        const int i = i_first + j;  // index among the new frames; global frame l = l0 + i
        float acc = 0.f;
        for (int q = nov - 1; q >= 0; --q) {
            const int lq = l0 + i - q, off = q * hop + s;
            if (lq > delay && lq >= 1 && off < sz) acc += syn[(size_t)(i - q + nov - 1) * sz + off];
        }
        if (outf) outf[e] = acc;
        if (out16) {
            float rr = copysignf(floorf(fabsf(acc) + 0.5f), acc);  // fwrite(..,'int16'): round half away, saturate
            rr = fminf(fmaxf(rr, -32768.f), 32767.f);
            if (!(acc == acc)) rr = 0.f;  // NaN -> 0 as MATLAB's integer conversion
            out16[e] = (int16_t)rr;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_wadapt: the whole W-only adaptation solve (src/bnmf_sep_event_RT_IS16.m:330-335 ->
// src/sparse_nmf.m:157-286 with h_update_ind all false, KL) in ONE cooperative launch.
// The problem is tiny (513 x 100, rank <= 64) and the generic path costs three launches per
// iteration (~30 us); here V and H never change, so every workgroup keeps H (both orientations) and its
// RB = 8 rows of V and W in LDS for the whole solve and an iteration is two small products per row block
// plus two grid barriers for the column sums (colsum(G.*W), then the column norms).  The convergence
// test runs identically in every workgroup on the same reduced numbers.  Master copy of W in fp64
// (see k_wapply).  Grid = ceil(F / 8) workgroups of 256 threads, launched cooperatively.
// ---------------------------------------------------------------------------------------------

// exchange accesses: agent-scope relaxed atomics = sc1 write-through stores / coherent loads (see grid_bar)
__device__ __forceinline__ void xstore(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xload(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Cross-workgroup sum of `ncol` (<= 2*RP+1) column quantities, part[q*stride + c], q < nwg, into out[c]
// (LDS).  After the barrier's acquire these loads come from memory, ~2 us each: every thread issues all of
// its loads before the first add (4 thread groups x 64 columns, <= kWaQ workgroups per group), then the
// sums are combined in a fixed order (bit-reproducible).
constexpr int kWaQ = 40;  // workgroups per thread group: covers nwg <= 80
__device__ __forceinline__ void cross_sum(const double* __restrict__ part, int stride, int ncol, int nwg, double* scratch /*[2][128]*/,
                                          double* out /*[128]*/) {
    const int g = threadIdx.x >> 7, c = threadIdx.x & 127;
    if (threadIdx.x < 256) {  // (a workgroup may have more threads than the exchange needs)
        const int per = (nwg + 1) / 2, q0 = g * per;
        double v[kWaQ];
#pragma unroll
        for (int i = 0; i < kWaQ; ++i) {
            const int q = q0 + i;
            v[i] = (i < per && q < nwg && c < ncol) ? xload(part + (size_t)q * stride + c) : 0.0;
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < kWaQ; ++i) s += v[i];
        scratch[g * 128 + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 128 && threadIdx.x < ncol) out[threadIdx.x] = scratch[c] + scratch[128 + c];
    __syncthreads();
}

struct WAdaptArgs {
    const float* V;        // [ma][F]  lambda_d_blk in time order
    const float* H;        // [ma][Ra] Ad_blk in time order, rows not in r_up zeroed
    const double* W0;      // [Ra][F]  init_w (first R_a columns of B_DFT_d)
    const uint8_t* w_ind;  // [Ra]     r_up
    double* Wout;          // [Ra][F]  result
    double* part1;         // [nwg][RP + 1]  colsum(G.*W) partials + divergence partial
    double* part2;         // [nwg][2][RP]   squared-norm and column-sum partials
    double* costh;         // [max_iter]
    int* n_iter_out;
    unsigned* bar;         // grid-barrier counter, zero at launch
    int F, Ra, ma, max_iter, cost_check;
    float sparsity, flr;
    double conv_eps;
};

constexpr int kWaRB = 8, kWaRP = 64;    // rows of W per workgroup (32 threads each), padded rank (32 rows x 1024 threads: 2261 instead of 2743 frames/s)
constexpr int kWaLPR = 32;                // threads per row of W (64 -- a wave per row, eight waves per workgroup -- made the products no
                                          // faster and both exchanges slower: 3434 against 3620 frames/s)
constexpr int kWaNT = kWaRB * kWaLPR;     // threads per workgroup

// Grid barrier on a monotonic device counter (zeroed before the launch).  cooperative_groups' grid.sync()
// measured ~20 us per call here, and an agent-scope release/acquire pair costs a write-back plus an
// invalidate of the XCD's whole L2 on every workgroup.  Instead, everything the workgroups exchange goes
// through agent-scope (sc1, write-through / coherent) relaxed atomic stores and loads (xstore / xload):
// __syncthreads() waits for those stores to be acknowledged by the coherence point, one relaxed agent-scope
// add publishes the arrival, and the readers' sc1 loads cannot hit a stale line.  The kernel is launched
// cooperatively, so every workgroup is resident; the spin is bounded all the same so that a lost workgroup
// ends in wrong numbers (flagged through n_iter_out = -1), never in a hung GPU.
// (Tried: no barrier at all, every thread re-loading the partial rows until a sentinel value is gone -- "the data
// is the flag".  Correct, but 256 pollers per workgroup flood the coherent path: 2090 instead of 2790 frames/s.
// Tried: 16 / 32 rows per workgroup: the exchange does not get cheaper with fewer workgroups, the products do
// get slower: 2650 / 2400 frames/s.)
// `ok_sp`: one int of the caller's DYNAMIC LDS.  (A static __shared__ here preceded the dynamic region and shifted its base
// by 4 bytes: every 8- / 16-byte LDS access of the kernel was then off its natural alignment and replayed at 64 cycles per
// wave-instruction -- /opt/skills/guides/cdna_hip_programming.md, "statics totalling != 0 (mod 16) shift the base".)
__device__ __forceinline__ bool grid_bar(unsigned* ctr, unsigned nwg, unsigned& gen, int* ok_sp) {
    int& ok_s = *ok_sp;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's exchange stores are acknowledged ...
    __syncthreads();                                   // ... and so are everybody's in the workgroup
    ++gen;
    if (threadIdx.x < 64) stress_jitter();  // (-DSNMF_STRESS builds only: snmf_kernels.h)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = gen * nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24))
            __builtin_amdgcn_s_sleep(1);
        ok_s = spins < (1u << 24);
    }
    __syncthreads();
    return ok_s != 0;
}

#ifdef SNMF_PROF_WA  // diagnostic builds only: cycles of workgroup 0 by phase, summed over the solves of a run
__device__ unsigned long long g_wa_prof[10];
#define WA_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    atomicAdd(&g_wa_prof[i], t_ - wa_t_); wa_t_ = t_; } } while (0)
#else
#define WA_STAMP(i)
#endif
__global__ __launch_bounds__(kWaNT) void k_wadapt(WAdaptArgs a) {
#ifdef SNMF_PROF_WA
    unsigned long long wa_t_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr int RB = kWaRB, RP = kWaRP, NT = kWaNT, NWV = kWaNT / 64;
    unsigned gen = 0;
    bool bar_ok = true;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int F = a.F, Ra = a.Ra, ma = a.ma, tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int f0 = wg * RB;
    double* Wd = reinterpret_cast<double*>(sm);          // [RB][RP]
    double* cq = Wd + RB * RP;                            // [RP] reduced column quantities
    double* cs = cq + RP;                                 // [RP] colsum(W)
    double* red = cs + RP;                                // [NWV + 1] (padded to 32)
    int* oks = reinterpret_cast<int*>(red + 31);          // the grid barrier's verdict (red is padded to 32 doubles)
    double* scr = red + 32;                               // [2][128] cross_sum scratch
    double* tmp = scr + 256;                              // [2*RP] reduced quantities of one exchange
    float* Wf = reinterpret_cast<float*>(tmp + 2 * RP);   // [RB][RP]
    float* Gs = Wf + RB * RP;                             // [RB][RP]
    float* sk = Gs + RB * RP;                             // [RP] rowsum(H)
    float* Vs = sk + RP;                                  // [RB][ma]
    float* Rs = Vs + RB * ma;                             // [RB][ma]
    float* Hs = Rs + RB * ma;                             // [Ra][ma]
    float* HT = Hs + Ra * ma;                             // [ma][RP + 1]
    // the operand images of the two products: what ONE lane needs for one k (its four frames l, l + 32, l + 64, l + 96) resp.
    // one frame (its two columns l, l + 32) side by side, so that it is ONE wide read instead of four / two (ma <= 128)
    float* Hp = HT + ma * (RP + 1);                       // [Ra][128]: Hp[k][4 l + j] = h[k][l + 32 j]  (0 past ma)
    float* HTp = Hp + Ra * 128;                           // [ma][RP]:  HTp[t][2 l + c] = h[l + 32 c][t] (0 past Ra)
    static_assert(kWaLPR == 32 && RP == 64, "operand images of k_wadapt's products");
    const bool wide = a.ma <= 128 && (a.ma & 3) == 0;    // (16-byte reads of rows of ma floats)
    const int f = tid / kWaLPR, l32 = tid % kWaLPR;       // row of the block, lane within the row's threads
    static_assert(RP % kWaLPR == 0 && 128 % kWaLPR == 0, "whole columns of G / frames per lane");
    const bool row_ok = f0 + f < F;

    // ---- load + src/sparse_nmf.m:157-169 ------------------------------------------------------
    // (consecutive threads take consecutive ROWS of one column / frame: the block's 8 rows are one 64- / 32-byte piece of memory)
    for (int i = tid; i < RB * RP; i += NT) {
        const int k = i / RB, ff = i - k * RB;
        Wd[ff * RP + k] = (k < Ra && f0 + ff < F) ? a.W0[(size_t)k * F + f0 + ff] : 0.0;
    }
    for (int i = tid; i < RB * ma; i += NT) {
        const int t = i / RB, ff = i - t * RB;
        Vs[ff * ma + t] = (f0 + ff < F) ? fmaxf(a.V[(size_t)t * F + f0 + ff], a.flr) : 0.f;   // :169
    }
    for (int i = tid; i < Ra * ma; i += NT) {
        const int t = i / Ra, k = i - t * Ra;
        Hs[k * ma + t] = a.H[i];
    }
    __syncthreads();
    // wn = sqrt(sum(w.^2)) and colsum(w): partials over this block's rows, ONE exchange (the column sums of the normalised
    // W are colsum(w) ./ wn, as after every update below)
    if (tid < RP) {
        double s2 = 0.0, s1 = 0.0;
        for (int ff = 0; ff < RB; ++ff) {
            const double w = Wd[ff * RP + tid];
            s2 += w * w;
            s1 += w;
        }
        xstore(a.part2 + (size_t)wg * 2 * RP + tid, s2);
        xstore(a.part2 + (size_t)wg * 2 * RP + RP + tid, s1);
    }
    bar_ok &= grid_bar(a.bar, (unsigned)nwg, gen, oks);
    cross_sum(a.part2, 2 * RP, 2 * RP, nwg, scr, tmp);
    if (tid < RP) {
        cq[tid] = tid < Ra ? sqrt(tmp[tid]) : 1.0;  // wn
        cs[tid] = tmp[RP + tid] / cq[tid];
    }
    __syncthreads();
    for (int i = tid; i < RB * RP; i += NT) {
        const int k = i % RP;
        const double w = k < Ra ? Wd[i] / cq[k] : 0.0;   // w = w ./ wn
        Wd[i] = w;
        Wf[i] = (float)w;
    }
    for (int i = tid; i < Ra * ma; i += NT) {
        const int k = i / ma;
        Hs[i] = (float)((double)Hs[i] * cq[k]);          // h = h .* wn'  (:160)
    }
    __syncthreads();
    for (int i = tid; i < ma * (RP + 1); i += NT) {
        const int t = i / (RP + 1), k = i - t * (RP + 1);
        HT[i] = k < Ra ? Hs[k * ma + t] : 0.f;
    }
    if (wide) {
        for (int i = tid; i < Ra * 128; i += NT) {
            const int k = i >> 7, l = (i & 127) >> 2, j = i & 3, t = l + 32 * j;
            Hp[i] = t < ma ? Hs[k * ma + t] : 0.f;
        }
        for (int i = tid; i < ma * RP; i += NT) {
            const int t = i / RP, l = (i % RP) >> 1, c = i & 1, k = l + 32 * c;
            HTp[i] = k < Ra ? Hs[k * ma + t] : 0.f;
        }
    }
    if (tid < RP) {
        float s = 0.f;
        if (tid < Ra)
            for (int t = 0; t < ma; ++t) s += Hs[tid * ma + t];
        sk[tid] = s;                                     // sum(h,2)
    }
    __syncthreads();
    double sh_const = 0.0;                               // sum(sum(sparsity .* h)) (:261), constant: H is fixed
    for (int k = 0; k < Ra; ++k) sh_const += (double)a.sparsity * (double)sk[k];
    WA_STAMP(0);

    double last_cost = 0.0;
    int n_rec = 0;
    bool stopped = false;
    for (int j = 1; j <= a.max_iter + 1; ++j) {
        if (j > a.max_iter && !a.cost_check) break;
        // ---- Lam' = max(W*H, flr), ratio, divergence of iterate j-1 ------------------------------
        // Both products keep the summation order of the plain loops (one fma chain per output element, k resp. t ascending), but
        // as written before -- one output at a time, its LDS operands read inside the chain, the result stored into the same
        // LDS array the next chain reads from -- every fma waited for an LDS round trip (18 k cycles per product on a wave that
        // has its SIMD to itself: 115 cycles per fma; phase stamps of the diagnostic build).  Here a thread's outputs (four
        // frames, two columns) advance together and nothing is stored inside the loops, so the reads pipeline.
        float dterm = 0.f;
        if (wide) {
            constexpr int NJ = 4;
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
            const float* wr = Wf + f * RP;        // (columns k >= Ra of Wf are zero)
            const float* hp = Hp + 4 * l32;
            // four k per step: one 16-byte read of the W row + four 16-byte reads of the lane's frames; the chains are the plain
            // loop's (per frame, k ascending; a padded k adds fma(0, h, acc) = acc)
            const int n4 = (Ra + 3) >> 2;
            for (int g = 0; g < n4; ++g) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * g);
                f32x4 h4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 4 * g + u < Ra ? 4 * g + u : Ra - 1;  // (w = 0 there)
                    h4[u] = *reinterpret_cast<const f32x4*>(hp + k * 128);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) a4[jj] = fmaf(w4[u], h4[u][jj], a4[jj]);
            }
            float ac[NJ] = {a4[0], a4[1], a4[2], a4[3]};
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int t = l32 + kWaLPR * j;
                if (t < ma) {
                    const float lam = fmaxf(ac[j], a.flr), v = Vs[f * ma + t];
                    Rs[f * ma + t] = row_ok ? v * fast_rcp(lam) : 0.f;
                    if (row_ok) dterm += div_term<BM_KL>(v, lam, 1.f, 0.f);
                }
            }
        } else {
            for (int t = l32; t < ma; t += kWaLPR) {
                float acc = 0.f;
                for (int k = 0; k < Ra; ++k) acc = fmaf(Wf[f * RP + k], Hs[k * ma + t], acc);
                const float lam = fmaxf(acc, a.flr), v = Vs[f * ma + t];
                Rs[f * ma + t] = row_ok ? v * fast_rcp(lam) : 0.f;
                if (row_ok) dterm += div_term<BM_KL>(v, lam, 1.f, 0.f);
            }
        }
        __syncthreads();
        WA_STAMP(1);
        // ---- G = (V./Lam') * H' -----------------------------------------------------------------
        {
            constexpr int NC = RP / kWaLPR;  // columns l32 + kWaLPR * c of this lane (HT's columns k >= Ra are zero)
            float g[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) g[c] = 0.f;
            const float* rr = Rs + f * ma;
            if (wide) {
                // four frames per step: one 16-byte read of the ratio row + four 8-byte reads of the lane's two columns
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const float* xp = HTp + 2 * l32;
                const int m4 = ma >> 2;
                for (int q = 0; q < m4; ++q) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(rr + 4 * q);
                    f32x2 x2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x2[u] = *reinterpret_cast<const f32x2*>(xp + (4 * q + u) * RP);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        g[0] = fmaf(r4[u], x2[u][0], g[0]);
                        g[1] = fmaf(r4[u], x2[u][1], g[1]);
                    }
                }
                for (int t = 4 * m4; t < ma; ++t) {
                    const float r = rr[t];
                    g[0] = fmaf(r, xp[t * RP], g[0]);
                    g[1] = fmaf(r, xp[t * RP + 1], g[1]);
                }
            } else {
                const float* hc = HT + l32;
                for (int t = 0; t < ma; ++t) {
                    const float r = rr[t];
#pragma unroll
                    for (int c = 0; c < NC; ++c) g[c] = fmaf(r, hc[t * (RP + 1) + kWaLPR * c], g[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) Gs[f * RP + l32 + kWaLPR * c] = l32 + kWaLPR * c < Ra ? g[c] : 0.f;
        }
        const float dw = wave_sum_f(dterm);
        if ((tid & 63) == 0) red[tid >> 6] = (double)dw;
        __syncthreads();
        if (tid < RP) {
            double s = 0.0;
            for (int ff = 0; ff < RB; ++ff) s += (double)Gs[ff * RP + tid] * Wd[ff * RP + tid];
            xstore(a.part1 + (size_t)wg * (RP + 1) + tid, s);  // colsum(G .* W) partial (:217)
        }
        if (tid == RP) {
            double dsum = 0.0;
#pragma unroll
            for (int q = 0; q < NWV; ++q) dsum += red[q];
            xstore(a.part1 + (size_t)wg * (RP + 1) + RP, dsum);
        }
        WA_STAMP(2);
        bar_ok &= grid_bar(a.bar, (unsigned)nwg, gen, oks);
        cross_sum(a.part1, RP + 1, RP + 1, nwg, scr, tmp);     // colsum(G .* W) | div
        if (tid < RP) cq[tid] = tmp[tid];
        if (tid == RP) red[NWV] = tmp[RP];
        __syncthreads();
        WA_STAMP(3);
        if (a.cost_check && j > 1) {                      // cost of iterate j-1 (:260-284)
            const double cost = red[NWV] + sh_const;
            const int it = j - 1;
            bool stopnow = false;
            if (it > 1 && a.conv_eps > 0.0) stopnow = fabs(cost - last_cost) / last_cost < a.conv_eps;
            if (wg == 0 && tid == 0) a.costh[it - 1] = cost;
            n_rec = it;
            last_cost = cost;
            if (stopnow) {
                stopped = true;
                break;
            }
        }
        if (j > a.max_iter) break;
        // ---- W update (:215-222) on this block's rows, then the norms ------------------------------
        for (int i = tid; i < RB * RP; i += NT) {
            const int k = i % RP;
            double wv = Wd[i];
            if (k < Ra && a.w_ind[k]) {
                const double s = (double)sk[k];
                double dpw = s + wv * cq[k];
                dpw = dpw > (double)a.flr ? dpw : (double)a.flr;
                wv = wv * ((double)Gs[i] + wv * (s * cs[k])) / dpw;
            }
            Wd[i] = wv;
        }
        __syncthreads();
        if (tid < RP) {
            double s2 = 0.0, s1 = 0.0;
            for (int ff = 0; ff < RB; ++ff) {
                const double w = Wd[ff * RP + tid];
                s2 += w * w;
                s1 += w;
            }
            xstore(a.part2 + (size_t)wg * 2 * RP + tid, s2);
            xstore(a.part2 + (size_t)wg * 2 * RP + RP + tid, s1);
        }
        WA_STAMP(4);
        bar_ok &= grid_bar(a.bar, (unsigned)nwg, gen, oks);
        cross_sum(a.part2, 2 * RP, 2 * RP, nwg, scr, tmp);
        if (tid < RP) {
            const double nrm = tid < Ra ? sqrt(tmp[tid]) : 1.0;
            cq[tid] = nrm;
            cs[tid] = tmp[RP + tid] / nrm;               // colsum of the normalised W
        }
        __syncthreads();
        WA_STAMP(5);
        for (int i = tid; i < RB * RP; i += NT) {
            const int k = i % RP;
            const double w = k < Ra ? Wd[i] / cq[k] : 0.0;   // :242, ALL columns
            Wd[i] = w;
            Wf[i] = (float)w;
        }
        __syncthreads();
        WA_STAMP(6);
#ifdef SNMF_PROF_WA
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_wa_prof[8], 1ull);
#endif
    }
    __syncthreads();
    for (int i = tid; i < RB * RP; i += NT) {
        const int k = i / RB, ff = i - k * RB;
        if (k < Ra && f0 + ff < F) a.Wout[(size_t)k * F + f0 + ff] = Wd[ff * RP + k];
    }
    if (wg == 0 && tid == 0) *a.n_iter_out = !bar_ok ? -1 : (stopped ? n_rec : a.max_iter);
    WA_STAMP(7);
#ifdef SNMF_PROF_WA
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_wa_prof[9], 1ull);
#endif
}

}  // namespace snmf
