// snmf_smallf.h -- the KL half-steps for spectrograms of at most two 32-row tiles (F <= 64: the Mel solves,
// run_basis_train.m:90-91 and run_basis_DNMF_Mel.m:75-88 on p.Mel_bands = 64 rows) with r <= 256 components.
//
// Why a family of its own.  With F = 64 a 32-frame tile is 232 MFMAs for a whole compute unit; the role pipelines
// (k_hstep_rp, k_wstats with loader waves) pay a fixed hand-off chain per tile -- loader -> A team -> B team -> loader, one
// tile in flight per workgroup -- and were latency-bound at a quarter of the MFMA peak and a third of the HBM rate
// (profiles/r04_experiments.md section 6: 2.2 us per staged tile whatever its size).  scripts/tile_stream_probe.hip shows
// what the memory system does when nothing waits for anything else: ONE wave per CU streaming private 24 KB tiles into
// registers moves 5 TB/s, four waves per CU 6.5 TB/s.  So here a tile belongs to ONE wave from its first load to its last
// store, nothing is handed between waves, and eight waves per CU hide each other's latencies:
//
//   k_hstep_sf   lane (t, h) of the wave loads its frame's H column and V column straight into the MFMA operand layout
//                (16-byte pieces at k = 8q + 4h: the B operand of Lam = W*H), Lam's D tile IS the operand layout of the
//                ratio for W^T*ratio (the permuted contraction order of snmf_kernels.h), whose D tile IS the layout the H
//                column was loaded in -- so the update is register-to-register and the store mirrors the load.
//                No LDS image of H, V or the ratio; the only shared data are W's two operand images (LDS, read-only).
//
// Same arithmetic per element as k_hstep_rp (same MFMA order over the contraction, same epilogue expressions), so the two agree
// bit for bit on H and to the summation order of the objective partials.
#pragma once
#include "snmf_kernels.h"
#include <type_traits>

namespace snmf {

constexpr int kSfWaves = 8;  // waves per workgroup (two per SIMD), one workgroup per CU

// NF row tiles (1..2), NK column tiles of H (1..8; 254 VGPRs at 8, no scratch).  Dynamic LDS: Wt4 image [NF][rp/8][2][32][4], Wk4 image [NK][Fq/8][2][32][4],
// 1 ./ dph [rp], lambda_k [rp], then [2][kSfWaves] doubles for the objective partials.
template <int NF, int NK, bool OBJ>
__global__ __launch_bounds__(kSfWaves * 64, 2) void k_hstep_sf(StepArgs a) {
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp;
    const int nq8 = rp / 8, nqf = a.Fq / 8;  // k-blocks of an image row tile / column tile
    float* const wt = lds;                                  // Wt4: NF * rp * 32 floats
    float* const wk = wt + (size_t)NF * rp * 32;            // Wk4: NK * Fq * 32 floats
    float* const rdp = wk + (size_t)NK * a.Fq * 32;         // 1 ./ dph   [rp]   (scalar / per-row sparsity)
    float* const lmk = rdp + rp;                            // lambda_k   [rp]
    double* const red = reinterpret_cast<double*>(lmk + rp);  // [2][kSfWaves]

    // this wave's tiles: wave-major over the grid, so that the waves with one tile more are spread over all CUs first
    const int gw = w * (int)gridDim.x + (int)blockIdx.x, nw = kSfWaves * (int)gridDim.x;
    const int t = lane & 31, h = lane >> 5;

    f32x4 hq[NK * 4], vq[NF * 4];
    auto load_tile = [&](int tile) {
        const float* hp = a.Hin + ((size_t)tile * 32 + t) * rp + 4 * h;
        const float* vp = a.V + ((size_t)tile * 32 + t) * Fp + 4 * h;
#pragma unroll
        for (int q = 0; q < NK * 4; ++q) hq[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
#pragma unroll
        for (int q = 0; q < NF * 4; ++q) vq[q] = *reinterpret_cast<const f32x4*>(vp + 8 * q);
    };
    {
        const int n4 = (NF * rp * 32 + NK * a.Fq * 32) / 4, nt4 = NF * rp * 32 / 4;
        for (int i = threadIdx.x; i < n4; i += kSfWaves * 64) {
            const f32x4 x = i < nt4 ? reinterpret_cast<const f32x4*>(a.Wt4)[i] : reinterpret_cast<const f32x4*>(a.Wk4)[i - nt4];
            reinterpret_cast<f32x4*>(lds)[i] = x;
        }
        for (int k = threadIdx.x; k < rp; k += kSfWaves * 64) {
            rdp[k] = a.S ? 0.f : fast_rcp(a.dphv[k]);
            lmk[k] = a.S ? 0.f : a.lamk[k];
        }
    }
    __syncthreads();

    // The two waves of a SIMD start together and do identical work: left alone they run in lockstep -- both wait for their
    // tile's loads at the same time, both sit in their epilogues at the same time.  A one-off delay of the second wave of each
    // SIMD puts one wave's memory waits under the other's MFMA loops, and nothing ever re-synchronises them.
    if (a.stagger > 0 && w >= 4) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - ts < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(16);
    }
    const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + lane;  // fragment (phi, q): wtl[(phi * nq8 + q) * 64]
    const f32x4* const wkl = reinterpret_cast<const f32x4*>(wk) + lane;  // fragment (kap, q): wkl[(kap * nqf + q) * 64]
    double acc_div = 0.0, acc_sh = 0.0;

    // The sparsity kind is a compile-time constant of the tile loop (SK 0: one lambda for every row, 1: a lambda per row, 2: an
    // r x T matrix in H's layout): as run-time branches inside the update they cut it into dozens of basic blocks, each ending
    // in register copies.
    auto tiles = [&](auto sk_tag) {
    constexpr int SK = decltype(sk_tag)::value;
    for (int tile = gw; tile < a.n_tiles; tile += nw) {
        const int t0 = tile * 32;
        // (the loads sit at the TOP of the loop: issued behind the previous tile's stores they would take 96 registers of their own)
        load_tile(tile);
        // ---- P1: Lam[f, t] = sum_k W[f, k] H[k, t]; A = W fragment (LDS), B = this lane's H pieces.  Both row tiles at once: two
        // accumulator chains that share the B operand alternate in the pipe; the fragments of block q + 1 are read before the
        // MFMAs of block q (the guards are scalar: blocks past ceil(r / 8) are zero padding) ----
        float dsum = 0.f;
        {
            f32x16 acc[NF];
            f32x4 wa[NF], wb[NF];
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) {
                acc[phi] = zero16();
                wa[phi] = wtl[(phi * nq8) * 64];
            }
            // (nk = ceil(r / 32), so the first 4 (NK - 1) blocks always hold components; only the last column tile's blocks can be
            //  zero padding: 4 NK - 3 <= nqk = ceil(r / 8) <= 4 NK.  The test LEAVES the straight-line code -- as a guard around
            //  each block it made every block end in a copy of both accumulators)
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                if (q > 4 * (NK - 1) && q >= a.nqk) break;
                if (q + 1 < NK * 4) {
#pragma unroll
                    for (int phi = 0; phi < NF; ++phi) wb[phi] = wtl[(phi * nq8 + q + 1) * 64];
                }
                SNMF_PIN();
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int phi = 0; phi < NF; ++phi) acc[phi] = mfma32(wa[phi][e], hq[q][e], acc[phi]);
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) wa[phi] = wb[phi];
            }
            // ratio in place over V: lane (t, h), element (g, j) <-> f = 32 phi + 8 g + 4 h + j  (rp_p1_epilogue)
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) {
                const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = vq[phi * 4 + g][j];
                        const float lam = fmaxf(acc[phi][4 * g + j], kFlr);
                        if (OBJ) {
                            const float d = div_term<BM_KL>(v, lam, a.beta, a.inv_bb1);
                            if (edge) dsum += (phi * 32 + 8 * g + 4 * h + j < a.F && t0 + t < a.T) ? d : 0.f;
                            else dsum += d;
                        }
                        vq[phi * 4 + g][j] = v * fast_rcp(lam);
                    }
                    if (OBJ) __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler runs all 32 logarithms side by side: 75 spilled VGPRs)
                }
            }
        }
        if (OBJ) acc_div += (double)dsum;
        // ---- P2: num[k, t] = sum_f W[f, k] ratio[f, t]; A = W^T fragment (LDS), B = this lane's ratio pieces; column tiles in
        // pairs (two chains on one B operand); then the update ----
        float shsum = 0.f;
        auto p2_update = [&](const f32x16& acc, const int kap) {
            // H update: lane (t, h), element (g, j) <-> k = 32 kap + 8 g + 4 h + j  (rp_p2_epilogue)
            float hs = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k0 = kap * 32 + 8 * g + 4 * h;
                const f32x4 ho = hq[kap * 4 + g];
                f32x4 sp, dp;
                if constexpr (SK == 2) {
                    sp = *reinterpret_cast<const f32x4*>(a.S + ((size_t)t0 + t) * rp + k0);
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dp[j] = fast_rcp(fmaxf(cs[j] + sp[j], kFlr));
                } else {
                    dp = *reinterpret_cast<const f32x4*>(rdp + k0);
                    if constexpr (OBJ && SK == 1) sp = *reinterpret_cast<const f32x4*>(lmk + k0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) hq[kap * 4 + g][j] = ho[j] * acc[4 * g + j] * dp[j];
                if constexpr (OBJ) {
                    if constexpr (SK == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) hs += ho[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) shsum += sp[j] * ho[j];
                    }
                }
            }
            if constexpr (OBJ && SK == 0) shsum += a.lam_u * hs;
        };
#pragma unroll
        for (int kap = 0; kap < NK; kap += 2) {
            constexpr int NQ = NF * 4;  // = Fq / 8 (no extra row in this family)
            if (kap + 1 < NK) {
                f32x16 acc0 = zero16(), acc1 = zero16();
                f32x4 wa0 = wkl[(kap * nqf) * 64], wa1 = wkl[((kap + 1) * nqf) * 64], wb0, wb1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + 1 < NQ) {
                        wb0 = wkl[(kap * nqf + q + 1) * 64];
                        wb1 = wkl[((kap + 1) * nqf + q + 1) * 64];
                    }
                    SNMF_PIN();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = mfma32(wa0[e], vq[q][e], acc0);
                        acc1 = mfma32(wa1[e], vq[q][e], acc1);
                    }
                    wa0 = wb0;
                    wa1 = wb1;
                }
                p2_update(acc0, kap);
                p2_update(acc1, kap + 1);
            } else {
                f32x16 acc0 = zero16();
                f32x4 wa0 = wkl[(kap * nqf) * 64], wb0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + 1 < NQ) wb0 = wkl[(kap * nqf + q + 1) * 64];
                    SNMF_PIN();
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc0 = mfma32(wa0[e], vq[q][e], acc0);
                    wa0 = wb0;
                }
                p2_update(acc0, kap);
            }
        }
        if (OBJ) acc_sh += (double)shsum;
        // ---- the updated column leaves the way it came ----
        {
            float* op = a.Hout + ((size_t)t0 + t) * rp + 4 * h;
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) *reinterpret_cast<f32x4*>(op + 8 * q) = hq[q];
        }
    }
    };
    if (a.S) tiles(std::integral_constant<int, 2>{});
    else if (a.lam_is_u) tiles(std::integral_constant<int, 0>{});
    else tiles(std::integral_constant<int, 1>{});

    if (OBJ) {
        // fixed-order reduction: lanes (shuffles), then the eight waves through LDS
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            acc_div += __shfl_xor(acc_div, s, 64);
            acc_sh += __shfl_xor(acc_sh, s, 64);
        }
        if (lane == 0) {
            red[w] = acc_div;
            red[kSfWaves + w] = acc_sh;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double d = 0.0, s = 0.0;
            for (int i = 0; i < kSfWaves; ++i) {
                d += red[i];
                s += red[kSfWaves + i];
            }
            a.part[2 * blockIdx.x] = d;
            a.part[2 * blockIdx.x + 1] = s;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// k_wstats_sf: the KL statistics of the W half-step (src/sparse_nmf.m:215-222: G = (V ./ Lam') * H', s = sum(H, 2)) in the same
// style: nothing is staged for anybody else.  Wave w of a workgroup is row tile phi = w / NCL of "chunk lane" c = w % NCL
// (NCL = 8 / NF): it walks the tiles tb + c, tb + c + NCL, ... of the workgroup's frame range with its NK accumulator tiles of G
// in registers.  Per tile:
//   P3  Lam'^T[t, f] = sum_k H[k, t] W[f, k]: A = the lane's H pieces (lane (t, h), loaded as in k_hstep_sf), B = W fragment
//       (LDS); the D tile has frames in registers and rows in lanes -- the A-operand layout of
//   P4  G[f, k] += sum_t ratio[t, f] H[k, t]: A = ratio registers, B = H[32 kap + lane][t] as 4-byte loads (the tile's rows were
//       fetched a moment ago by this wave: L2 hits), one column tile ahead of its MFMAs; the same reads give the row sums.
// At the end the chunk lanes' accumulators meet in LDS (fixed order) and the workgroup writes ONE slab, so k_wfin / k_reduce see
// what k_wstats would have written for n_chunks = gridDim.x.
// Dynamic LDS: max(Wt4 image NF * rp * 32 floats, (NCL - 1) * NF partial tiles of NK * 16 * 64 floats) + [rp] row sums + doubles.
template <int NF, int NK, bool OBJ>
__global__ __launch_bounds__(kSfWaves * 64, 2) void k_wstats_sf(StepArgs a, int n_chunks, int mat_index, int n_mat) {
    if (a.stop && *a.stop) return;
    constexpr int NCL = kSfWaves / NF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp, nq8 = rp / 8;
    const int fl = lane & 31, h = lane >> 5;
    const int phi = w / NCL, c = w % NCL;
    const int chunk = blockIdx.x;
    const int tb = (int)(((long long)a.n_tiles * chunk) / n_chunks), te = (int)(((long long)a.n_tiles * (chunk + 1)) / n_chunks);
    {
        const int n4 = NF * rp * 32 / 4;
        for (int i = threadIdx.x; i < n4; i += kSfWaves * 64) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(a.Wt4)[i];
    }
    __syncthreads();
    const f32x4* const wtl = reinterpret_cast<const f32x4*>(lds) + lane + (size_t)phi * nq8 * 64;

    f32x16 G[NK];
    float ssum[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        G[k] = zero16();
        ssum[k] = 0.f;
    }
    double acc_div = 0.0;
    const bool do_s = phi == 0;

    for (int tile = tb + c; tile < te; tile += NCL) {
        const int t0 = tile * 32;
        // this lane's H pieces (operand layout, as k_hstep_sf), its V values in the D layout of P3, the first column tile of P4
        f32x4 hq[NK * 4];
        float v[16], b0[16], b1[16];
        {
            const float* hp = a.Hin + ((size_t)t0 + fl) * rp + 4 * h;
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) hq[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
            // (buffer loads: ONE lane offset per array, the register's row as a scalar offset -- sixteen 64-bit lane addresses per
            //  array were what spilled)
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
            const int vo = ((4 * h) * Fp + phi * 32 + fl) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vo, drow(i, 0) * Fp * 4, 0));
        }
        // H[32 kap + fl][t0 + drow(i, h)]: lane offset (4 h) * rp + fl, scalar offset drow(i, 0) * rp + 32 kap
        const __amdgpu_buffer_rsrc_t rh =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)t0 * rp), 0, 32 * rp * 4, 0x00020000);
        const int ho = ((4 * h) * rp + fl) * 4;
        auto ldb = [&](float (&b)[16], int kap) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, ho, (drow(i, 0) * rp + kap * 32) * 4, 0));
        };
        ldb(b0, 0);
        // ---- P3 ----
        f32x16 acc = zero16();
        {
            f32x4 wa = wtl[0], wb;
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                if (q > 4 * (NK - 1) && q >= a.nqk) break;  // (zero padding past ceil(r / 8): see k_hstep_sf)
                if (q + 1 < NK * 4) wb = wtl[(q + 1) * 64];
                SNMF_PIN();
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(hq[q][e], wa[e], acc);
                wa = wb;
            }
        }
        // ---- ratio (and the objective of a W-only solve): lane (f = fl, h), register i <-> frame t0 + drow(i, h) ----
        float R[16];
        {
            const int f = phi * 32 + fl;
            const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
            float dsum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float lam = fmaxf(acc[i], kFlr);
                if (OBJ) {
                    const float d = div_term<BM_KL>(v[i], lam, a.beta, a.inv_bb1);
                    if (edge) dsum += (f < a.F && t0 + drow(i, h) < a.T) ? d : 0.f;
                    else dsum += d;
                }
                R[i] = v[i] * fast_rcp(lam);
            }
            if (OBJ) acc_div += (double)dsum;
        }
        // ---- P4, the row sums riding on its operand reads ----
        auto ktile = [&](f32x16& g, float& ss, const float (&b)[16]) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g = mfma32(R[i], b[i], g);
            if (do_s) {
                float s4 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s4 += b[i];
                ss += s4;
            }
        };
#pragma unroll
        for (int kap = 0; kap < NK; kap += 2) {
            if (kap + 1 < NK) ldb(b1, kap + 1);
            SNMF_PIN();
            ktile(G[kap], ssum[kap], b0);
            if (kap + 1 < NK) {
                if (kap + 2 < NK) ldb(b0, kap + 2);
                SNMF_PIN();
                ktile(G[kap + 1], ssum[kap + 1], b1);
            }
        }
    }

    // ---- the chunk lanes' partial statistics -> lane 0 of each row tile, through LDS, in lane order ----
    __syncthreads();  // (every wave is through with the W image)
    float* xs = lds;                                             // [(NCL - 1) * NF][NK * 16][64]
    float* sred = lds + (size_t)(NCL - 1) * NF * NK * 16 * 64;   // [NCL][rp] row sums
    double* dred = reinterpret_cast<double*>(sred + NCL * rp);   // [kSfWaves]
    if (c > 0) {
        float* dst = xs + (size_t)((c - 1) * NF + phi) * NK * 16 * 64 + lane;
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(k * 16 + i) * 64] = G[k][i];
    }
    if (do_s) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float other = __shfl_xor(ssum[k], 32, 64);  // the two lane halves hold the two halves of a tile's frames
            if (h == 0) sred[c * rp + k * 32 + fl] = ssum[k] + other;
        }
    }
    if (OBJ) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc_div += __shfl_xor(acc_div, s, 64);
        if (lane == 0) dred[w] = acc_div;
    }
    __syncthreads();
    if (c == 0) {
        for (int p = 1; p < NCL; ++p) {
            const float* src = xs + (size_t)((p - 1) * NF + phi) * NK * 16 * 64 + lane;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) G[k][i] += src[(k * 16 + i) * 64];
        }
        // slab: D tile lane (k = fl, h), register -> f = 32 phi + drow(reg, h)  (k_wstats' layout)
        float* slab = a.slabs + ((size_t)chunk * n_mat + mat_index) * rp * Fp;
#pragma unroll
        for (int kap = 0; kap < NK; ++kap) {
            float* dst = slab + (size_t)(kap * 32 + fl) * Fp + phi * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {G[kap][4 * g], G[kap][4 * g + 1], G[kap][4 * g + 2], G[kap][4 * g + 3]};
                *reinterpret_cast<f32x4*>(dst + 8 * g) = o;
            }
        }
    }
    for (int k = threadIdx.x; k < rp; k += kSfWaves * 64) {
        float sk = 0.f;
        for (int p = 0; p < NCL; ++p) sk += sred[p * rp + k];
        a.spart[(size_t)chunk * rp + k] = sk;
    }
    if (OBJ && threadIdx.x == 0) {
        double d = 0.0;
        for (int i = 0; i < kSfWaves; ++i) d += dred[i];
        a.part[2 * chunk] = d;
        a.part[2 * chunk + 1] = 0.0;
    }
}

}  // namespace snmf
