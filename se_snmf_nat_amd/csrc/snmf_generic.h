// ============================================================================================
// The out-of-envelope path: shapes the fused kernels cannot hold.
//
// k_hstep / k_wstats keep a whole frame tile's images (H block + ratio image, rows F + r) in the 160 KiB LDS of a CU,
// and k_wstats keeps per-column sums in registers: that covers every setting the reference ships (F <= 513, r <= 1000)
// and everything up to F + r ~ 2540 / r <= 1024 for W updates -- but src/sparse_nmf.m itself accepts any size.  Beyond the
// envelope the plan runs the SAME iteration (src/sparse_nmf.m:186-258) with its intermediates in HBM:
//     Lam = W*H                               k_g_gemm
//     ratio images R = V.*Lam.^(beta-2), D = Lam.^(beta-1), divergence terms      k_g_ratio
//     H step: num = W'*R (den = W'*D, beta != 1)   k_g_gemm;   H <- H .* num ./ max(den + sparsity, flr)   k_g_hupd
//     W step: Lam' = W*H;  Q = R*H', P = D*H' (KL: P = row sums of H) as split-T slabs   k_g_gemm, k_g_rowsum
//             then the ordinary k_reduce (fixed-order fp64 sum of the slabs) and k_wapply.
// So the statistics buffer, the W update, the objective bookkeeping, the early stop and the multi-rank exchange are the
// ones of the fast path; only the two big products and the element-wise passes differ.  Correctness first: the GEMM is
// a plain LDS-tiled kernel (64 x 64 x 16 tiles, one f32-MFMA quadrant per wave, no operand pipelining), contractions over the frames are cut into chunks of kGChunkT frames whose
// fp32 partial sums are added in fp64 by k_reduce, like the fast path's chunk slabs.
// ============================================================================================
#pragma once

#include "snmf_kernels.h"

namespace snmf {

constexpr int kGBlocks = 1024;   // workgroups of the element-wise passes = objective partial slots
constexpr int kGChunkT = 2048;   // frames per split of a contraction over T

// C[z](m, n) = sum_{k in split z} A(m, k) * B(k, n); element (i, j) of X at X[i * rsX + j * csX]; split z covers
// k in [z * kchunk, min(K, (z + 1) * kchunk)) and writes C + z * zC.
struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, kchunk;
    long long rsA, csA, rsB, csB, rsC, csC, zC;
    const int* stop;
};

static __global__ __launch_bounds__(256) void k_g_gemm(GemmArgs g) {
    if (g.stop && *g.stop) return;
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;  // (x runs over N: the frames, the one dimension that can pass 65535 tiles)
    const int k_lo = blockIdx.z * g.kchunk, k_hi = min(g.K, k_lo + g.kchunk);
    // four waves, one 32 x 32 quadrant of the block tile each, on the f32 MFMA (operand map: snmf_kernels.h).  The product
    // is formed as C^T = B^T * A^T when C's unit stride runs along m, so that the lanes of a D register (its columns)
    // always run along C's unit stride and the stores coalesce.
    const int w = tid >> 6, lane = tid & 63, fl = lane & 31, h = lane >> 5;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const bool c_m_fast = g.rsC == 1;
    f32x16 acc = zero16();
    // the faster-running index of each operand picks how a tile is read (coalesced along the unit stride)
    const bool a_m_fast = g.rsA == 1, b_n_fast = g.csB == 1;
    for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + 256 * e;  // 1024 elements of each tile
            {
                const int mm = a_m_fast ? (i & 63) : (i >> 4), kk = a_m_fast ? (i >> 6) : (i & 15);
                const int m = m0 + mm, k = k0 + kk;
                As[kk][mm] = (m < g.M && k < k_hi) ? g.A[(long long)m * g.rsA + (long long)k * g.csA] : 0.f;
            }
            {
                const int nn = b_n_fast ? (i & 63) : (i >> 4), kk = b_n_fast ? (i >> 6) : (i & 15);
                const int n = n0 + nn, k = k0 + kk;
                Bs[kk][nn] = (n < g.N && k < k_hi) ? g.B[(long long)k * g.rsB + (long long)n * g.csB] : 0.f;
            }
        }
        __syncthreads();
        if (c_m_fast) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) acc = mfma32(Bs[kk + h][wn + fl], As[kk + h][wm + fl], acc);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) acc = mfma32(As[kk + h][wm + fl], Bs[kk + h][wn + fl], acc);
        }
        __syncthreads();
    }
    float* C = g.C + (long long)blockIdx.z * g.zC;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // D[row = drow(i, h)][col = fl]: rows come from the first MFMA operand, columns from the second
        const int m = m0 + wm + (c_m_fast ? fl : drow(i, h)), n = n0 + wn + (c_m_fast ? drow(i, h) : fl);
        if (m < g.M && n < g.N) C[(long long)m * g.rsC + (long long)n * g.csC] = acc[i];
    }
}

// workgroup sum of one double per thread (256 threads), result in thread 0
__device__ __forceinline__ double g_block_sum(double v, double* red /*[256]*/) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    return red[0];
}

// Lam -> R = V .* Lam^(beta-2) (KL: V ./ Lam) and, beta != 1, D = Lam^(beta-1); OBJ: the block's divergence terms
// (src/sparse_nmf.m:248-258) into part[2 b], part[2 b + 1] = 0.  V, Lam, R, D are [T][Fp] images.
template <int BM, bool OBJ>
__global__ __launch_bounds__(256) void k_g_ratio(const float* __restrict__ V, const float* __restrict__ Lam, float* __restrict__ R,
                                                 float* __restrict__ D, int F, int Fp, int T, float beta, float inv_bb1,
                                                 double* __restrict__ part, const int* stop) {
    if (stop && *stop) return;
    __shared__ double red[256];
    double dsum = 0.0;
    const long long n = (long long)F * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = i / F;
        const int f = (int)(i - t * F);
        const long long idx = t * Fp + f;
        const float v = V[idx], lam = fmaxf(Lam[idx], kFlr);
        if (OBJ) dsum += (double)div_term<BM>(v, lam, beta, inv_bb1);
        if (BM == BM_KL) {
            R[idx] = v / lam;
        } else {
            R[idx] = v * numfac_of_lam<BM>(lam, beta);
            D[idx] = den_of_lam<BM>(lam, beta);
        }
    }
    if (OBJ) {
        const double s = g_block_sum(dsum, red);
        if (threadIdx.x == 0) {
            part[2 * blockIdx.x] = s;
            part[2 * blockIdx.x + 1] = 0.0;
        }
    }
}

// H <- H .* num ./ max(den + sparsity, flr) (src/sparse_nmf.m:189-208); KL: den = colsum(W) (dphv = max(colsum + lambda,
// flr) when the sparsity is per row).  OBJ: the block's share of sum(sparsity .* H) of the previous iterate -> part[2 b + 1].
template <bool KL, bool OBJ>
__global__ __launch_bounds__(256) void k_g_hupd(const float* __restrict__ Hin, float* __restrict__ Hout, const float* __restrict__ Num,
                                                const float* __restrict__ Den, const float* __restrict__ S,
                                                const float* __restrict__ lamk, const float* __restrict__ colsum,
                                                const float* __restrict__ dphv, int r, int rp, int T, double* __restrict__ part,
                                                const int* stop) {
    if (stop && *stop) return;
    __shared__ double red[256];
    double sh = 0.0;
    const long long n = (long long)r * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = i / r;
        const int k = (int)(i - t * r);
        const long long idx = t * rp + k;
        const float h = Hin[idx];
        const float sp = S ? S[idx] : lamk[k];
        float den;
        if (KL) den = S ? fmaxf(colsum[k] + sp, kFlr) : dphv[k];
        else den = fmaxf(Den[idx] + sp, kFlr);
        Hout[idx] = h * Num[idx] / den;
        if (OBJ) sh += (double)sp * (double)h;
    }
    if (OBJ) {
        const double s = g_block_sum(sh, red);
        if (threadIdx.x == 0) part[2 * blockIdx.x + 1] = s;
    }
}

// spart[z][k] = sum over the frames of split z of H[k, t]  (KL W step: the "P" of every row)
static __global__ __launch_bounds__(256) void k_g_rowsum(const float* __restrict__ H, int rp, int r, int T, int kchunk, float* __restrict__ spart,
                                                  const int* stop) {
    if (stop && *stop) return;
    const int z = blockIdx.y, t_lo = z * kchunk, t_hi = min(T, t_lo + kchunk);
    for (int k = blockIdx.x * 256 + threadIdx.x; k < r; k += gridDim.x * 256) {
        float s = 0.f;
        for (int t = t_lo; t < t_hi; ++t) s += H[(long long)t * rp + k];
        spart[(long long)z * rp + k] = s;
    }
}

// out[i] = sum over the chunks of slabs[c][i] (fp64 accumulation, chunk order), i < n: the Gram matrix H*H' of the
// Euclidean W step (see launch_gram_p in snmf_api.hip)
static __global__ __launch_bounds__(256) void k_gram_sum(const float* __restrict__ slabs, int n_chunks, size_t n, float* __restrict__ out,
                                                  const int* stop) {
    if (stop && *stop) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double s = 0.0;
        int c = 0;
        for (; c + 8 <= n_chunks; c += 8) {  // eight loads in flight, added in chunk order
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = slabs[(size_t)(c + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)x[j];
        }
        for (; c < n_chunks; ++c) s += (double)slabs[(size_t)c * n + i];
        out[i] = (float)s;
    }
}

}  // namespace snmf
