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

// The W images into LDS by LDS-DMA: they are contiguous on both sides -- 1 KiB per instruction, a few per wave, all in flight at
// once, no registers (as a load / store loop the fill was two dependent round trips of eight loads a thread: 1.5 us of every launch
// of this family).  The caller waits (s_waitcnt vmcnt(0)) in front of its barrier: nothing else orders a ds_read behind an LDS-DMA.
__device__ __forceinline__ void sf_fill_image(const float* src, float* dst, int nbytes, int w, int lane) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    for (int o = w * 1024; o < nbytes; o += kSfWaves * 1024)
        if (o + lane * 16 < nbytes) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + o / 4), 16, lane * 16, o, 0, 0);
}


// This lane's index, re-derived where it is needed (two VALU instructions, no input register) and OPAQUE to the optimiser: everything
// computed from it -- frame / half, operand offsets, image pointers -- is then a value of the tile it is used in, not a loop invariant
// that lives through the whole kernel.  k_iter_sf kept the thread index and a dozen values derived from it alive under its 128
// accumulators and SPILLED them (the cheapest values there are); a kernel that touches scratch at all pays 6-7 us per launch.
__device__ __forceinline__ int fresh_lane() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// LDS progress words between waves of a workgroup (the shared last tile of k_hstep_sf, the pair hand-off of k_iter_sf)
__device__ __forceinline__ void sf_post(unsigned* word, unsigned val, int lane) {
    stress_jitter();  // (-DSNMF_STRESS builds only: snmf_kernels.h)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sf_await(const unsigned* word, unsigned target, const int* stop) {
    int spin = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if (++spin > kSpinLimit) {
            raise_fault(stop);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    stress_jitter();
}


// THE SHARED LAST TILE (a.part_S == 4; plan: sf_share).  Tiles go to waves wave-major over the grid, so 3125 tiles on 2048 waves are
// one tile for everybody, a second one for waves 0..3 of every workgroup -- one per SIMD -- and 53 more for wave 4 of 53 workgroups:
// those SIMDs run two tiles side by side while every other SIMD of the chip runs one, and the launch ends a tile period later than
// its work asks for (64 x 100000, r = 200: 77.5 us against 68.3 us at 98304 frames, scripts/gpu_r6_k.sh).  Those tiles [n_full,
// n_tiles) are therefore SHARED by the four waves of their level (w0 .. w0 + 3, one per SIMD): wave v takes the column tiles
// kap = CPW v .. of H -- its k range of Lam = W*H, whose four partial sums meet in LDS and are added in wave order by every wave
// (so all four hold the same Lam and ratio), then W^T*ratio, the update and the store of its own rows of H.  No other data is exchanged.
// The four waves take the shared tile FIRST, then their own tiles (the SIMDs' loads are the same either way; see the call).
// Another order of additions for Lam than the whole tile's one chain: SNMF_HSTEP_SPLIT=0 keeps every tile whole (tests compare).

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
    f32x4* const xch = reinterpret_cast<f32x4*>(red + 2 * kSfWaves);  // shared last tile: partial Lam [4 waves][NF][4][64] f32x4 ...
    unsigned* const xfl = reinterpret_cast<unsigned*>(xch + 4 * NF * 4 * 64);  // ... and their four "written" words
    const int n_main = a.part_S ? a.n_full : a.n_tiles;  // tiles of the wave-major loop

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
        sf_fill_image(a.Wt4, wt, NF * rp * 32 * 4, w, lane);
        sf_fill_image(a.Wk4, wk, NK * a.Fq * 32 * 4, w, lane);
        for (int k = threadIdx.x; k < rp; k += kSfWaves * 64) {
            rdp[k] = a.S ? 0.f : fast_rcp(a.dphv[k]);
            lmk[k] = a.S ? 0.f : a.lamk[k];
        }
        if (a.part_S && threadIdx.x < 4) xfl[threadIdx.x] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

    // ---- the shared last tile (see the note above the kernel): waves w0 .. w0 + NP - 1 of the workgroups that have one ----
    auto shared_tile = [&](auto sk_tag) {
        constexpr int SK = decltype(sk_tag)::value;
        constexpr int CPW = (NK + 3) / 4, NP = (NK + CPW - 1) / CPW;  // column tiles per wave, participating waves
        const int wv = w - ((a.n_full / (int)gridDim.x) & 7);         // (the level behind the whole ones: 0 or 4, the plan's condition)
        const int tile = a.n_full + (int)blockIdx.x;
        if (NP < 2 || wv < 0 || wv >= NP || tile >= a.n_tiles) return;
        const int kap0 = wv * CPW;
        const bool two = CPW == 2 && kap0 + 1 < NK;  // (an odd NK leaves the last wave one column tile)
        // (lane-derived values of THIS block of code: taken from the kernel's own they stay alive across the tile loop above, which
        //  has no register to spare -- 29 / 63 spilled VGPRs at NK = 7 / 8)
        const int lane = fresh_lane(), t = lane & 31, h = lane >> 5;
        int tile_o = tile, rp_o = rp, Fp_o = Fp;
        asm volatile("" : "+s"(tile_o), "+s"(rp_o), "+s"(Fp_o));  // (nothing of this block is computed ahead of the tile loop)
        const int t0 = tile_o * 32;
        const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + lane;
        const f32x4* const wkl = reinterpret_cast<const f32x4*>(wk) + lane;
        f32x4 hs[CPW * 4], vs[NF * 4];
        {
            const float* hp = a.Hin + ((size_t)t0 + t) * rp_o + 32 * kap0 + 4 * h;
            const float* vp = a.V + ((size_t)t0 + t) * Fp_o + 4 * h;
#pragma unroll
            for (int q = 0; q < CPW * 4; ++q) hs[q] = (q < 4 || two) ? *reinterpret_cast<const f32x4*>(hp + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NF * 4; ++q) vs[q] = *reinterpret_cast<const f32x4*>(vp + 8 * q);
        }
        // P1 over this wave's k blocks, partial Lam -> LDS, the four partials added in wave order
        float dsum = 0.f;
        {
            f32x16 acc[NF];
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) acc[phi] = zero16();
#pragma unroll
            for (int ql = 0; ql < CPW * 4; ++ql) {
                const int q = kap0 * 4 + ql;
                if (q >= a.nqk || (ql >= 4 && !two)) break;
                f32x4 wa[NF];
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) wa[phi] = wtl[(phi * nq8 + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int phi = 0; phi < NF; ++phi) acc[phi] = mfma32(wa[phi][e], hs[ql][e], acc[phi]);
            }
#pragma unroll
            for (int phi = 0; phi < NF; ++phi)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xch[((wv * NF + phi) * 4 + g) * 64 + lane] = f32x4{acc[phi][4 * g], acc[phi][4 * g + 1], acc[phi][4 * g + 2], acc[phi][4 * g + 3]};
            sf_post(xfl + wv, 1u, lane);
#pragma unroll
            for (int v = 0; v < NP; ++v) sf_await(xfl + v, 1u, a.stop);
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) {
                const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 lam4 = xch[((0 * NF + phi) * 4 + g) * 64 + lane];
#pragma unroll
                    for (int v = 1; v < NP; ++v) {
                        const f32x4 x = xch[((v * NF + phi) * 4 + g) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j) lam4[j] += x[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = vs[phi * 4 + g][j];
                        const float lam = fmaxf(lam4[j], kFlr);
                        if (OBJ && wv == 0) {  // (the tile's divergence terms count once)
                            const float d = div_term<BM_KL>(v, lam, a.beta, a.inv_bb1);
                            if (edge) dsum += (phi * 32 + 8 * g + 4 * h + j < a.F && t0 + t < a.T) ? d : 0.f;
                            else dsum += d;
                        }
                        vs[phi * 4 + g][j] = v * fast_rcp(lam);
                    }
                    if (OBJ) __builtin_amdgcn_sched_barrier(0);  // (as in the tile loop: the logarithms one group at a time)
                }
            }
        }
        if (OBJ) acc_div += (double)dsum;
        // P2 + update of this wave's column tiles
        float shsum = 0.f;
        auto upd = [&](const f32x16& acc, const int c) {  // (p2_update of the tile loop on the local pieces hs[4 c ..])
            const int kap = kap0 + c;
            float hsm = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k0 = kap * 32 + 8 * g + 4 * h;
                const f32x4 ho = hs[c * 4 + g];
                f32x4 sp, dp;
                if constexpr (SK == 2) {
                    sp = *reinterpret_cast<const f32x4*>(a.S + ((size_t)t0 + t) * rp_o + k0);
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(a.colsum + k0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dp[j] = fast_rcp(fmaxf(cs[j] + sp[j], kFlr));
                } else {
                    dp = *reinterpret_cast<const f32x4*>(rdp + k0);
                    if constexpr (OBJ && SK == 1) sp = *reinterpret_cast<const f32x4*>(lmk + k0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) hs[c * 4 + g][j] = ho[j] * acc[4 * g + j] * dp[j];
                if constexpr (OBJ) {
                    if constexpr (SK == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) hsm += ho[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) shsum += sp[j] * ho[j];
                    }
                }
            }
            if constexpr (OBJ && SK == 0) shsum += a.lam_u * hsm;
        };
        constexpr int NQ = NF * 4;
        if (two) {
            if constexpr (CPW == 2) {
                f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 wa0 = wkl[(kap0 * nqf + q) * 64], wa1 = wkl[((kap0 + 1) * nqf + q) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = mfma32(wa0[e], vs[q][e], acc0);
                        acc1 = mfma32(wa1[e], vs[q][e], acc1);
                    }
                }
                upd(acc0, 0);
                upd(acc1, 1);
            }
        } else {
            f32x16 acc0 = zero16();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 wa0 = wkl[(kap0 * nqf + q) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0 = mfma32(wa0[e], vs[q][e], acc0);
            }
            upd(acc0, 0);
        }
        if (OBJ) acc_sh += (double)shsum;
        {
            float* op = a.Hout + ((size_t)t0 + t) * rp_o + 32 * kap0 + 4 * h;
#pragma unroll
            for (int q = 0; q < CPW * 4; ++q)
                if (q < 4 || two) *reinterpret_cast<f32x4*>(op + 8 * q) = hs[q];
        }
    };
    // The sparsity kind is a compile-time constant of the tile loop (SK 0: one lambda for every row, 1: a lambda per row, 2: an
    // r x T matrix in H's layout): as run-time branches inside the update they cut it into dozens of basic blocks, each ending
    // in register copies.
    auto tiles = [&](auto sk_tag) {
    constexpr int SK = decltype(sk_tag)::value;
    if (a.part_S) shared_tile(sk_tag);  // (FIRST: behind the tile loop its values stayed alive across the loop -- 29 / 55 spilled VGPRs at NK = 7 / 8)
    for (int tile = gw; tile < n_main; tile += nw) {
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
        if (threadIdx.x < 64) {
            double d = 0.0, s = 0.0;
            for (int i = 0; i < kSfWaves; ++i) {
                d += red[i];
                s += red[kSfWaves + i];
            }
            obj_partial_out(a, blockIdx.x, d, s);
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
    sf_fill_image(a.Wt4, lds, NF * rp * 32 * 4, w, lane);
    // (the shared remainder tile, below: partial Lam' [8 waves][4][64] f32x4 and the waves' "written" words, behind the W image in what
    //  becomes the partial-statistics area at the end of the kernel -- NF = 2, NK >= 2: three times the image)
    f32x4* const xch = reinterpret_cast<f32x4*>(lds + (size_t)NF * rp * 32);
    unsigned* const xfl = reinterpret_cast<unsigned*>(xch + kSfWaves * 4 * 64);
    if (a.part_S && threadIdx.x < kSfWaves) xfl[threadIdx.x] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

    // Whole rounds of NCL tiles, then the remainder of m = n % NCL tiles: remainder tile j goes to chunk lane (j + m) % NCL -- the
    // deal of k_iter_sf's W tasks (whose H task of that tile sits on pair j, so that no SIMD gets an extra tile of both kinds);
    // the two kernels accumulate the same tiles in the same order, which keeps their slabs bit for bit the same.
    const int n_my = te - tb, n_rnd = n_my / NCL, n_rem0 = n_my - n_rnd * NCL;
    // THE SHARED REMAINDER TILE (a.part_S == 4; the plan's wsf_share).  12 or 13 tiles for the four chunk lanes of a workgroup are three
    // rounds -- or three rounds and ONE more tile on one SIMD while the other three idle (64 x 100000, r = 100: 48.2 us against 41.1 us
    // at 98304 frames, scripts/gpu_r6_k.sh).  A single remainder tile is therefore shared by all eight waves, FIRST (as k_hstep_sf's
    // shared tile; the accumulators are still zero): wave (phi, c) takes column tile kap = c -- its k range of Lam', whose NK partial
    // sums meet in LDS and are added in lane order by every wave of the row tile, then its own column tile of G and of the row sums.
    // Another order of additions than the whole tile's; SNMF_HSTEP_SPLIT=0 keeps the tile whole (tests compare).
    const bool share = NF == 2 && NK >= 2 && a.part_S == 4 && n_rem0 == 1;
    const int n_rem = share ? 0 : n_rem0;
    if (share && c < NK) {
        const int t0 = (tb + n_rnd * NCL) * 32;
        f32x4 hq[4];
        float v[16], b0[16];
        {
            const float* hp = a.Hin + ((size_t)t0 + fl) * rp + 32 * c + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) hq[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
            const int vo = ((4 * h) * Fp + phi * 32 + fl) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vo, drow(i, 0) * Fp * 4, 0));
            const __amdgpu_buffer_rsrc_t rh =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)t0 * rp), 0, 32 * rp * 4, 0x00020000);
            const int ho = ((4 * h) * rp + fl) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                b0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, ho, (drow(i, 0) * rp + c * 32) * 4, 0));
        }
        f32x16 acc = zero16();
#pragma unroll
        for (int ql = 0; ql < 4; ++ql) {
            const int q = 4 * c + ql;
            if (q >= a.nqk) break;
            const f32x4 wa = wtl[q * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(hq[ql][e], wa[e], acc);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) xch[(w * 4 + g) * 64 + lane] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        sf_post(xfl + w, 1u, lane);
#pragma unroll
        for (int p = 0; p < NK; ++p) sf_await(xfl + phi * NCL + p, 1u, a.stop);
        float R[16];
        {
            const int f = phi * 32 + fl;
            const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
            float dsum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 lam4 = xch[((phi * NCL + 0) * 4 + g) * 64 + lane];
#pragma unroll
                for (int p = 1; p < NK; ++p) {
                    const f32x4 x = xch[((phi * NCL + p) * 4 + g) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) lam4[j] += x[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * g + j;
                    const float lam = fmaxf(lam4[j], kFlr);
                    if (OBJ && c == 0) {  // (the tile's divergence terms count once per row tile)
                        const float d = div_term<BM_KL>(v[i], lam, a.beta, a.inv_bb1);
                        if (edge) dsum += (f < a.F && t0 + drow(i, h) < a.T) ? d : 0.f;
                        else dsum += d;
                    }
                    R[i] = v[i] * fast_rcp(lam);
                }
            }
            if (OBJ) acc_div += (double)dsum;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (c == k) {
#pragma unroll
                for (int i = 0; i < 16; ++i) G[k] = mfma32(R[i], b0[i], G[k]);
                if (do_s) {
                    float s4 = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s4 += b0[i];
                    ssum[k] += s4;
                }
            }
        }
    }
    const int jx = (c - n_rem + NCL) % NCL;  // the remainder tile of this lane, if jx < n_rem
    const int n_it = n_rnd + (jx < n_rem ? 1 : 0);
    const bool x_first = n_rem == 1 && jx == 0;  // (a single remainder tile is its lane's FIRST tile: see k_iter_sf)
    for (int itx = 0; itx < n_it; ++itx) {
        const int itr = x_first ? itx - 1 : itx;  // index among the whole rounds (-1 / n_rnd: the remainder tile)
        const int tile = (itr >= 0 && itr < n_rnd) ? tb + c + itr * NCL : tb + n_rnd * NCL + jx;
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

// -------------------------------------------------------------------------------------------------------------------
// k_iter_sf (round 5): the H half-step AND the W statistics of a full KL update in ONE launch, for spectrograms of two 32-row
// tiles (F = 33..64: run_basis_train.m:90-91 on 64 Mel bands) and three or four column tiles (r = 65..128: R = 100).
// V and H cross HBM once per iteration instead of twice, the W images are filled once, and one launch with its ramp, its
// final reduction and its drain disappears.
//
// Eight waves, two per SIMD, in four PAIRS (waves c and 4 + c: SIMD partners under the cyclic wave placement):
//   H wave c      the tile body of k_hstep_sf: operand loads straight into the MFMA layout, P1, ratio, P2, the H update in
//                 registers, H_new stored the way it came -- and written once more into the pair's LDS hand-off buffer
//                 ([32 frames][32 NK + 4], 16-byte writes in the layout it was computed in);
//   W wave 4 + c  the tile body of k_wstats_sf on the tile its partner has just finished: P3's A operand (H_new pieces, lane
//                 (t, h)) comes out of the hand-off buffer with the 16-byte reads that mirror the writes, P4's B operand (H_new
//                 with the components in lanes) with 4-byte reads of the same buffer -- no second trip to L2 for H, nothing of
//                 it in the wave's vmcnt queue; V once more with the rows in lanes (4-byte loads of lines its partner fetched
//                 a tile earlier).  It owns the statistics of BOTH row tiles: NF * NK accumulator tiles (pinning them to a[0:127] with an "a" asm constraint
//                 fixed the kernel-wide register split at 128 / 128 and made the H waves spill: they are ordinary values).
// Why pairs and not one wave that does everything: the statistics of both row tiles are 128 registers and a tile's other state
// ~240, so a do-everything wave is alone on its SIMD -- built first, 68.8 us against 73.0 for the two launches (Mel 64 x 72000,
// r = 100): with one wave per SIMD every latency of the tile (the operand loads' HBM round trip first of all) is exposed.
// Here each wave stays under 256 registers, both waves of a SIMD carry 232 MFMAs per tile, and one wave's loads, LDS round
// trips and epilogues run beside the other's MFMA loops.  The hand-off is ONE-WAY and per SIMD pair -- `full` (H wave: tile i
// is in the buffer) / `empty` (W wave: tile i is read) -- not the workgroup-wide chain the role pipelines paid per tile.
// Work split: pair c walks the tiles tb + c, tb + c + 4, ... of the workgroup's contiguous frame range: exactly the tiles, in
// exactly the order, in which waves (phi, c) of k_wstats_sf accumulate them, and the pairs' statistics meet in LDS in the
// same order -- with n_chunks workgroups the slabs, the row sums and therefore W equal the two-launch path BIT FOR BIT, and H_new
// is per tile the same arithmetic (tests/test_gpu_parity.py::test_fused_small_f_iteration_equals_the_two_launches).
// Dynamic LDS: Wt4 image (P1 and P3), Wk4 image (P2), 1 ./ dph and lambda_k [rp] each, 4 hand-off buffers, 8 progress words; the
// end of the kernel reuses it for the pairs' partial statistics as k_wstats_sf does.

// The H wave's side of the hand-off WITHOUT compiler-visible memory effects: a fence (or a "memory" clobber) inside the tile loop
// orders every LDS access of the tile around it as far as the compiler knows, and the H waves' code went from 206 registers
// to 256 + 11 spilled (either of the two fences alone did it).  The hardware needs none of that: LDS operations of a wave
// complete in order, the data written behind the wait and ahead of the post are registers of this wave.
__device__ __forceinline__ void sf_post_raw(unsigned word_addr, unsigned val) {
    stress_jitter();
    asm volatile("ds_write_b32 %0, %1" ::"v"(word_addr), "v"(val));
}
__device__ __forceinline__ void sf_await_raw(unsigned word_addr, unsigned target, const int* stop) {
    int spin = 0;
    for (;;) {
        unsigned x;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(word_addr));
        if (__builtin_amdgcn_readfirstlane((int)x) >= (int)target) break;
        if (++spin > kSpinLimit) {
            raise_fault(stop);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    stress_jitter();
}

// SK: the sparsity kind (0: one lambda for every row, 1: a lambda per row, 2: an r x T matrix in H's layout) -- a parameter of the
// KERNEL here, not of a tile-loop lambda as in k_hstep_sf (three loops in one function: the register allocation of the worst).
template <int NK, bool OBJ, int SK>
__global__ __launch_bounds__(kSfWaves * 64, 2) void k_iter_sf(StepArgs a, int n_chunks, int n_mat) {
    if (a.stop && *a.stop) return;
    constexpr int NF = 2, NP = kSfWaves / 2, NTHR = kSfWaves * 64, LDT = 32 * NK + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = wave_index();
    const bool is_w = w >= NP;          // W wave of pair c
    const int c = is_w ? w - NP : w;
    const int rp = a.rp, Fp = a.Fp;
    const int nq8 = rp / 8, nqf = a.Fq / 8;
    float* const wt = lds;                                    // Wt4: NF * rp * 32 floats
    float* const wk = wt + (size_t)NF * rp * 32;              // Wk4: NK * Fq * 32 floats
    float* const rdp = wk + (size_t)NK * a.Fq * 32;           // 1 ./ dph [rp]
    float* const lmk = rdp + rp;                              // lambda_k [rp]
    float* const hb = lmk + rp + (size_t)c * 32 * LDT;        // this pair's hand-off buffer [32][LDT]
    float* const xb = lmk + rp + (size_t)NP * 32 * LDT;       // one more buffer: a chunk's single remainder tile (below)
    unsigned* const sig = reinterpret_cast<unsigned*>(xb + 32 * LDT);  // full[NP], empty[NP], xfull; the shared remainder tile: hxc, xfc, wxc, wdc (arrival counts)
    double* const accd = reinterpret_cast<double*>(sig + 16);           // [NP H waves][64 lanes][2] fp64 partial sums of the objective
    float* const ssl = reinterpret_cast<float*>(accd + (size_t)NP * 64 * 2);  // [NP W waves][64 lanes][4] row sums of H_new per lane (NK <= 4)
    unsigned* const full = sig + c;
    unsigned* const empty = sig + NP + c;
    const unsigned full_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)full;
    const unsigned empty_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)empty;
    const unsigned xfull_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(sig + 2 * NP);
    const unsigned wdc_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(sig + 2 * NP + 4);
    const int chunk = blockIdx.x;
    const int tb = (int)(((long long)a.n_tiles * chunk) / n_chunks), te = (int)(((long long)a.n_tiles * (chunk + 1)) / n_chunks);
    // THE SHARED REMAINDER TILE (a.part_S == 4; the plan's isf_share).  A chunk of 4 R + 1 tiles is R rounds and one more tile: as an H
    // task on pair 0 and a W task on pair 1 it made those two SIMDs carry 2 R + 1 tasks against 2 R on the others (64 x 72000: 5 against
    // 4, 55.7 us where 65536 frames take 46.5).  Now the four H waves share its H task FIRST (each its column tile: a k range of Lam, the
    // partial sums through LDS in wave order, then its rows of W^T*ratio, update, store and its columns of the extra hand-off buffer), and
    // the four W waves share its W task first (each: its k range of Lam' for both row tiles, partial sums through LDS, then column tile c
    // of G and of the row sums).  The exchanges live in the pairs' own hand-off buffers, which are idle until the H waves' first
    // whole tile is done; four arrival counts order it all (hxc: H partials written; xfc: H_new pieces in the extra buffer; wxc: W
    // partials written; wdc: W partials read -- only then may an H wave write its pair's buffer).
    const bool xshare = a.part_S == 4 && (te - tb) % NP == 1;
    unsigned* const hxc = sig + 2 * NP + 1;
    unsigned* const xfc = sig + 2 * NP + 2;
    unsigned* const wxc = sig + 2 * NP + 3;
    unsigned* const wdc = sig + 2 * NP + 4;
    auto count_up = [&](unsigned* word, int lane_) {  // (release: this wave's LDS writes are complete before the arrival is counted)
        stress_jitter();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane_ == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    {
        sf_fill_image(a.Wt4, wt, NF * rp * 32 * 4, w, (int)(threadIdx.x & 63));
        sf_fill_image(a.Wk4, wk, NK * a.Fq * 32 * 4, w, (int)(threadIdx.x & 63));
        for (int k = threadIdx.x; k < rp; k += NTHR) {
            rdp[k] = a.S ? 0.f : fast_rcp(a.dphv[k]);
            lmk[k] = a.S ? 0.f : a.lamk[k];
        }
        if (threadIdx.x < 16) sig[threadIdx.x] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing else orders a ds_read behind an LDS-DMA
    }
    __syncthreads();
    // (fragment (phi, q) of an image: wtl[(phi * nq8 + q) * 64], wkl[(kap * nqf + q) * 64] with wtl / wkl = image + this lane: formed per tile, see fresh_lane)

    // (the end of the kernel -- barrier, partial statistics into LDS, barrier -- is written out in BOTH role branches: the statistics
    //  tiles must not be live in the H waves' code, whose registers they would take: 128 of the 256 a wave has at two per SIMD)
    float* const xs = lds;                                            // [NP][NF * NK][16][64]   (after the first barrier)
    float* const sred = lds + (size_t)NP * NF * NK * 16 * 64;         // [NP][rp] row sums
    double* const dred = reinterpret_cast<double*>(sred + NP * rp);   // [2][NP]

    if (!is_w) {
        // The two fp64 partial sums of the objective live in LDS, a slot pair per lane, updated once per tile with inline ds_read_b64 /
        // ds_write_b64 (no compiler-visible memory effects: see sf_post_raw): as registers they were four of the VGPRs this kernel does
        // not have -- they were spilled across the tile body, and a kernel that touches scratch at all pays 6-7 us per launch
        // (scripts/scratch_probe.hip), a tenth of this one's run time.
        if (OBJ) {
            const unsigned acc_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(accd + ((size_t)c * 64 + fresh_lane()) * 2);
            const double z = 0.0;
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8" ::"v"(acc_a), "v"(z));
        }
        // ================================ H wave: k_hstep_sf's tile body + the hand-off ================================
        // this pair's H tasks: tile c of every whole round of NP tiles and remainder tile c (if there is one).  A SINGLE remainder tile
        // (n = 4 R + 1: Mel 64 x 72000 is 9 tiles a chunk) is pair 0's FIRST task and goes into the extra buffer, which nobody has to
        // release: pair 1's W wave takes it while it would otherwise wait for its own partner's second tile, and no SIMD carries an
        // extra task of both kinds behind a chain of hand-offs (5 tasks on the two busiest SIMDs instead of 6 on one).
        const int n_my = te - tb, n_rnd = n_my / NP, n_rem = xshare ? 0 : n_my - n_rnd * NP;
        const bool x_first = n_rem == 1 && c == 0;
        const unsigned n_h = (unsigned)(n_rnd + (c < n_rem ? 1 : 0));
        SNMF_STAMP_DECL
        if (xshare && c < NK) {
            // ---- this wave's quarter of the shared remainder tile's H task ----
            const int ln = fresh_lane(), t = ln & 31, h = ln >> 5;
            const int hlo = (t * rp + 4 * h) * 4, vlo = (t * Fp + 4 * h) * 4;
            const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + ln;
            const f32x4* const wkl = reinterpret_cast<const f32x4*>(wk) + ln;
            const int t0 = (tb + n_rnd * NP) * 32;
            f32x4 hs[4], vs[NF * 4];
            {
                const __amdgpu_buffer_rsrc_t rhi =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)t0 * rp), 0, 32 * rp * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rvi =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
#pragma unroll
                for (int ql = 0; ql < 4; ++ql) hs[ql] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rhi, hlo, 32 * (4 * c + ql), 0));
#pragma unroll
                for (int q = 0; q < NF * 4; ++q) vs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rvi, vlo, 32 * q, 0));
            }
            float dsum = 0.f;
            {
                f32x16 acc[NF];
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) acc[phi] = zero16();
#pragma unroll
                for (int ql = 0; ql < 4; ++ql) {
                    const int q = 4 * c + ql;
                    if (q >= a.nqk) break;
                    f32x4 wa[NF];
#pragma unroll
                    for (int phi = 0; phi < NF; ++phi) wa[phi] = wtl[(phi * nq8 + q) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int phi = 0; phi < NF; ++phi) acc[phi] = mfma32(wa[phi][e], hs[ql][e], acc[phi]);
                }
                f32x4* const xw = reinterpret_cast<f32x4*>(hb) + ln;  // this pair's hand-off buffer: [phi][g][lane] f32x4 (8 of its 16.5 KB)
#pragma unroll
                for (int phi = 0; phi < NF; ++phi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) xw[(phi * 4 + g) * 64] = f32x4{acc[phi][4 * g], acc[phi][4 * g + 1], acc[phi][4 * g + 2], acc[phi][4 * g + 3]};
                count_up(hxc, ln);
                sf_await(hxc, (unsigned)NK, a.stop);
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) {
                    const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 lam4 = reinterpret_cast<const f32x4*>(lmk + rp)[(phi * 4 + g) * 64 + ln];
#pragma unroll
                        for (int p = 1; p < NK; ++p) {
                            const f32x4 x = reinterpret_cast<const f32x4*>(lmk + rp + (size_t)p * 32 * LDT)[(phi * 4 + g) * 64 + ln];
#pragma unroll
                            for (int j = 0; j < 4; ++j) lam4[j] += x[j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = vs[phi * 4 + g][j];
                            const float lam = fmaxf(lam4[j], kFlr);
                            if (OBJ && c == 0) {  // (the tile's divergence terms count once)
                                const float d = div_term<BM_KL>(v, lam, a.beta, a.inv_bb1);
                                if (edge) dsum += (phi * 32 + 8 * g + 4 * h + j < a.F && t0 + t < a.T) ? d : 0.f;
                                else dsum += d;
                            }
                            vs[phi * 4 + g][j] = v * fast_rcp(lam);
                        }
                        if (OBJ) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            float shsum = 0.f;
            {
                constexpr int NQ = NF * 4;
                f32x16 acc0 = zero16();
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 wa0 = wkl[(c * nqf + q) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc0 = mfma32(wa0[e], vs[q][e], acc0);
                }
                float hsm = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // (p2_update of the tile loop on the local pieces)
                    const int k0 = c * 32 + 8 * g + 4 * h;
                    const f32x4 ho = hs[g];
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
                    for (int j = 0; j < 4; ++j) hs[g][j] = ho[j] * acc0[4 * g + j] * dp[j];
                    if constexpr (OBJ) {
                        if constexpr (SK == 0) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) hsm += ho[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) shsum += sp[j] * ho[j];
                        }
                    }
                }
                if constexpr (OBJ && SK == 0) shsum += a.lam_u * hsm;
            }
            if (OBJ) {
                const unsigned acc_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(accd + ((size_t)c * 64 + ln) * 2);
                double ad, as;
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(ad), "=v"(as) : "v"(acc_a));
                ad += (double)dsum;
                as += (double)shsum;
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(acc_a), "v"(ad), "v"(as));
            }
            {
                const __amdgpu_buffer_rsrc_t rho = __builtin_amdgcn_make_buffer_rsrc(a.Hout + (size_t)t0 * rp, 0, 32 * rp * 4, 0x00020000);
                f32x4* const xp = reinterpret_cast<f32x4*>(xb + t * LDT + 4 * h + 32 * c);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    buf_store_b128(rho, hlo, 32 * (4 * c + g), hs[g]);
                    xp[2 * g] = hs[g];
                }
            }
            count_up(xfc, ln);
        }
        for (unsigned ith = 0; ith < n_h; ++ith) {
            const int ln = fresh_lane(), t = ln & 31, h = ln >> 5;  // (lane (t, h): frame t, component / row half h)
            const int hlo = (t * rp + 4 * h) * 4, vlo = (t * Fp + 4 * h) * 4;  // this lane's byte offsets inside a tile of H / V
            const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + ln;
            const f32x4* const wkl = reinterpret_cast<const f32x4*>(wk) + ln;
            const unsigned acc_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(accd + ((size_t)c * 64 + ln) * 2);
            const bool is_x = x_first && ith == 0;                  // the remainder tile, into the extra buffer
            const unsigned it = x_first ? ith - 1u : ith;           // index among the tiles that go through this pair's own buffer
            const int tile = is_x ? tb + n_rnd * NP : tb + c + (int)it * NP;
            const int t0 = tile * 32;
            f32x4 hq[NK * 4], vq[NF * 4];
            {   // (buffer loads: descriptor + ONE lane offset per array + immediates -- as 64-bit lane pointers the three tile bases were
                //  six long-lived VGPRs that this kernel does not have: they were what the H waves spilled)
                const __amdgpu_buffer_rsrc_t rhi =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)t0 * rp), 0, 32 * rp * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rvi =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < NK * 4; ++q) hq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rhi, hlo, 32 * q, 0));
#pragma unroll
                for (int q = 0; q < NF * 4; ++q) vq[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rvi, vlo, 32 * q, 0));
            }
            // ---- P1 + ratio ----
            float dsum = 0.f;
            {
                f32x16 acc[NF];
                f32x4 wa[NF], wb[NF];
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) {
                    acc[phi] = zero16();
                    wa[phi] = wtl[(phi * nq8) * 64];
                }
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
                SNMF_STAMP(0);
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
                        if (OBJ) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (OBJ) {
                double ad;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ad) : "v"(acc_a));
                ad += (double)dsum;
                asm volatile("ds_write_b64 %0, %1" ::"v"(acc_a), "v"(ad));
            }
            SNMF_STAMP(1);
            // ---- P2 + the H update ----
            float shsum = 0.f;
            auto p2_update = [&](const f32x16& acc, const int kap) {
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
                constexpr int NQ = NF * 4;
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
            if (OBJ) {
                double as;
                asm volatile("ds_read_b64 %0, %1 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(as) : "v"(acc_a));
                as += (double)shsum;
                asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(acc_a), "v"(as));
            }
            SNMF_STAMP(2);
            // ---- H_new leaves the way it came ... ----
            {
                const __amdgpu_buffer_rsrc_t rho = __builtin_amdgcn_make_buffer_rsrc(a.Hout + (size_t)t0 * rp, 0, 32 * rp * 4, 0x00020000);
#pragma unroll
                for (int q = 0; q < NK * 4; ++q) buf_store_b128(rho, hlo, 32 * q, hq[q]);
            }
            // ---- ... and goes to the partner: [frame][component], once the partner has read the previous tile ----
            // (the writes are inline assembly: as C++ stores into the LDS array inside the tile loop they alias every W-fragment read of
            //  the tile as far as the compiler knows, and the H waves' code went from 206 registers to 256 + 11 spilled.  LDS operations
            //  of a wave complete in order, so the progress word posted behind them cannot overtake them)
            SNMF_STAMP(3);
            if (!is_x && it > 0) sf_await_raw(empty_a, it, a.stop);
            if (xshare && it == 0) sf_await_raw(wdc_a, (unsigned)NK, a.stop);  // (the W waves are through with the exchange that lived in this buffer)
            SNMF_STAMP(4);
            {
                const unsigned tpa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)((is_x ? xb : hb) + t * LDT + 4 * h);
#pragma unroll
                for (int q = 0; q < NK * 4; ++q) asm volatile("ds_write_b128 %0, %1" ::"v"(tpa + 32u * q), "v"(hq[q]));
            }
            sf_post_raw(is_x ? xfull_a : full_a, is_x ? 1u : it + 1);
            SNMF_STAMP(3);
        }
        __syncthreads();  // (every wave is through with the W images and the hand-off buffers)
        SNMF_STAMP(10);
        if (OBJ) {
            double acc_div, acc_sh;  // (the slots are this lane's own, outside xs / sred / dred)
            const unsigned acc_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(accd + ((size_t)c * 64 + fresh_lane()) * 2);
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(acc_div), "=v"(acc_sh) : "v"(acc_a));
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                acc_div += __shfl_xor(acc_div, s, 64);
                acc_sh += __shfl_xor(acc_sh, s, 64);
            }
            if (fresh_lane() == 0) {
                dred[c] = acc_div;
                dred[NP + c] = acc_sh;
            }
        }
        __syncthreads();
        SNMF_STAMP(11);
        SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * 8 + w) * 12, 12);
        SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * 8 + w);
    } else {
        // ================================ W wave: k_wstats_sf's tile body on the partner's tile ================================
        f32x16 G[NF][NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) G[phi][k] = zero16();
        }
        // (the row sums of H_new: per-lane LDS slots updated with ds_add_f32 -- one lane, one slot, program order: the same sums in the
        //  same order as a register would hold -- instead of NK more registers under the 128 accumulators)
        {
            const unsigned sa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ssl + ((size_t)c * 64 + fresh_lane()) * 4);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            asm volatile("ds_write_b128 %0, %1" ::"v"(sa), "v"(z));
        }
        SNMF_STAMP_DECL
        // One W task: the statistics of `tile` out of hand-off buffer `hbx`, once its H wave has posted `target` on `fullx`.
        auto w_tile = [&](const int tile, const float* const hbx, const unsigned* const fullx, const unsigned target) {
            const int t0 = tile * 32;
            const int ln = fresh_lane(), t = ln & 31, h = ln >> 5;
            const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + ln;
            const unsigned sa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ssl + ((size_t)c * 64 + ln) * 4);
            // V with the rows in lanes (lane (f, h): frames drow(i, h)): lines the partner fetched a tile ago.  ONE row tile's values
            // at a time: those of the second are loaded into the same registers in front of its P3.
            float vt[16];
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
            auto ld_vt = [&](int phi) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    vt[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, ((4 * h) * Fp + phi * 32 + t) * 4, drow(i, 0) * Fp * 4, 0));
            };
            ld_vt(0);
            SNMF_STAMP(9);
            sf_await(fullx, target, a.stop);
            SNMF_STAMP(5);
            // One row tile after the other (the statistics take 128 of the wave's 256 registers: the ratio of ONE row tile at a time,
            // and the 4-byte operand reads of P4 once per row tile -- LDS reads are what this wave has to spare):
#pragma unroll
            for (int phi = 0; phi < NF; ++phi) {
                // ---- P3: Lam'^T[t, f]; A = H_new pieces of lane (t, h) (hand-off buffer), B = W fragment ----
                SNMF_PIN();               // (not earlier: hoisted into the previous row tile's P4 the sixteen values were spilled at once)
                if (phi > 0) ld_vt(phi);  // (P3's 52 MFMAs are between these loads and their use)
                SNMF_PIN();
                float R[16];
                {
                    const float* ap = hbx + t * LDT + 4 * h;
                    f32x16 acc = zero16();
                    f32x4 wa = wtl[(phi * nq8) * 64], wb, ha = *reinterpret_cast<const f32x4*>(ap), hn;
#pragma unroll
                    for (int q = 0; q < NK * 4; ++q) {
                        if (q > 4 * (NK - 1) && q >= a.nqk) break;
                        if (q + 1 < NK * 4) {
                            wb = wtl[(phi * nq8 + q + 1) * 64];
                            hn = *reinterpret_cast<const f32x4*>(ap + 8 * (q + 1));
                        }
                        SNMF_PIN();
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma32(ha[e], wa[e], acc);
                        wa = wb;
                        ha = hn;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) R[i] = vt[i] * fast_rcp(fmaxf(acc[i], kFlr));
                }
                SNMF_STAMP(6);

                // ---- P4: G[f, k] += sum_t ratio'[t, f] H_new[k, t]; B = H_new with the components in lanes (frame drow(i, h),
                // component 32 kap + fl), one column tile ahead; the row sums ride on the reads of the first row tile ----
                {
                    const float* bp = hbx + (4 * h) * LDT + t;
                    // (one operand set, read right in front of its MFMAs: the partner wave's MFMAs cover the LDS round trip of the first
                    //  read, and a second set in flight cost this wave 16 of the registers it does not have)
#pragma unroll
                    for (int kap = 0; kap < NK; ++kap) {
                        float b0[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) b0[i] = bp[drow(i, 0) * LDT + kap * 32];
#pragma unroll
                        for (int i = 0; i < 16; ++i) G[phi][kap] = mfma32(R[i], b0[i], G[phi][kap]);
                        if (phi == 0) {
                            float s4 = 0.f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) s4 += b0[i];
                            asm volatile("ds_add_f32 %0, %1" ::"v"(sa + 4u * (unsigned)kap), "v"(s4));
                        }
                    }
                }
                SNMF_STAMP(7);
            }
        };
        // this pair's W tasks: the tiles its own H wave hands over (whole rounds) and -- so that no SIMD carries an extra tile of BOTH kinds --
        // the remainder tile whose H task sits on pair jx = (c - m) mod NP.  A single remainder tile comes FIRST, out of the extra
        // buffer (see the H waves); two or three come last, out of their pairs' own buffers.  k_wstats_sf deals its tiles the same way.
        const int n_my = te - tb, n_rnd = n_my / NP, n_rem = xshare ? 0 : n_my - n_rnd * NP;
        const int jx = (c - n_rem + NP) % NP;
        if (xshare && c < NK) {
            // ---- this wave's quarter of the shared remainder tile's W task (H_new of the tile: the extra buffer, once all NK pieces are in) ----
            const int t0 = (tb + n_rnd * NP) * 32;
            const int ln = fresh_lane(), t = ln & 31, h = ln >> 5;
            const f32x4* const wtl = reinterpret_cast<const f32x4*>(wt) + ln;
            const unsigned sa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ssl + ((size_t)c * 64 + ln) * 4);
            float vt[NF][16];
            {
                const __amdgpu_buffer_rsrc_t rv =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
#pragma unroll
                for (int phi = 0; phi < NF; ++phi)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        vt[phi][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, ((4 * h) * Fp + phi * 32 + t) * 4, drow(i, 0) * Fp * 4, 0));
            }
            sf_await(xfc, (unsigned)NK, a.stop);
            f32x4* const xw = reinterpret_cast<f32x4*>(hb) + ln;  // [phi][g][lane] f32x4 in this pair's hand-off buffer
            {
                const float* ap = xb + t * LDT + 4 * h + 32 * c;
#pragma unroll
                for (int phi = 0; phi < NF; ++phi) {
                    f32x16 acc = zero16();
#pragma unroll
                    for (int ql = 0; ql < 4; ++ql) {
                        const int q = 4 * c + ql;
                        if (q >= a.nqk) break;
                        const f32x4 ha = *reinterpret_cast<const f32x4*>(ap + 8 * ql), wa = wtl[(phi * nq8 + q) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc = mfma32(ha[e], wa[e], acc);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) xw[(phi * 4 + g) * 64] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                }
            }
            count_up(wxc, ln);
            sf_await(wxc, (unsigned)NK, a.stop);
            float R[NF][16];
#pragma unroll
            for (int phi = 0; phi < NF; ++phi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 lam4 = reinterpret_cast<const f32x4*>(lmk + rp)[(phi * 4 + g) * 64 + ln];
#pragma unroll
                    for (int p = 1; p < NK; ++p) {
                        const f32x4 x = reinterpret_cast<const f32x4*>(lmk + rp + (size_t)p * 32 * LDT)[(phi * 4 + g) * 64 + ln];
#pragma unroll
                        for (int j = 0; j < 4; ++j) lam4[j] += x[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) R[phi][4 * g + j] = vt[phi][4 * g + j] * fast_rcp(fmaxf(lam4[j], kFlr));
                }
            count_up(wdc, ln);  // (the partials are read: the H waves may use the buffers)
            {
                const float* bp = xb + (4 * h) * LDT + t + c * 32;
                float b0[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) b0[i] = bp[drow(i, 0) * LDT];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (c == k) {
#pragma unroll
                        for (int phi = 0; phi < NF; ++phi)
#pragma unroll
                            for (int i = 0; i < 16; ++i) G[phi][k] = mfma32(R[phi][i], b0[i], G[phi][k]);
                    }
                }
                float s4 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s4 += b0[i];
                asm volatile("ds_add_f32 %0, %1" ::"v"(sa + 4u * (unsigned)c), "v"(s4));
            }
        }
        if (n_rem == 1 && jx == 0) w_tile(tb + n_rnd * NP, xb, sig + 2 * NP, 1u);
        for (int it = 0; it < n_rnd; ++it) {
            w_tile(tb + c + it * NP, hb, full, (unsigned)(it + 1));
            sf_post(empty, (unsigned)(it + 1), fresh_lane());  // (the last reads of the buffer have returned: the MFMAs consumed them)
        }
        if (n_rem > 1 && jx < n_rem)  // (the last tile of pair jx's buffer: nobody waits for `empty` behind it)
            w_tile(tb + n_rnd * NP + jx, lmk + rp + (size_t)jx * 32 * LDT, sig + jx, (unsigned)(n_rnd + 1));
        SNMF_STAMP(8);
        __syncthreads();  // (every wave is through with the W images and the hand-off buffers)
        SNMF_STAMP(10);
        // Every W wave writes its NF * NK tiles; tile j = phi * NK + kap is then summed over the pairs IN PAIR ORDER (the order in which
        // k_wstats_sf's lane 0 adds them) by wave j % NP, which writes that tile's rows of the slab: the four waves share the
        // reduction instead of waiting for pair 0's W wave (its 384 dependent LDS reads were ~4 us at the end of the kernel).
        const int lane = fresh_lane(), t = lane & 31, h = lane >> 5;
#pragma unroll
        for (int phi = 0; phi < NF; ++phi)
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float* dst = xs + (size_t)(c * NF * NK + phi * NK + k) * 16 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[i * 64] = G[phi][k][i];
            }
        {
            f32x4 ssum;
            const unsigned sa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ssl + ((size_t)c * 64 + lane) * 4);
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ssum) : "v"(sa));
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const float other = __shfl_xor(ssum[k], 32, 64);  // the two lane halves hold the two halves of a tile's frames
                if (h == 0) sred[c * rp + k * 32 + t] = ssum[k] + other;
            }
        }
        __syncthreads();
        for (int j = c; j < NF * NK; j += NP) {
            const int phi = j / NK, kap = j - phi * NK;
            float g[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = xs[(size_t)j * 16 * 64 + i * 64 + lane];
            for (int p = 1; p < NP; ++p) {
                const float* src = xs + (size_t)(p * NF * NK + j) * 16 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] += src[i * 64];
            }
            // slab: D tile lane (k = fl, h), register -> f = 32 phi + drow(reg, h)  (k_wstats' layout).  Buffer stores: the 64-bit
            // lane pointers of these few stores were computed at the top of the kernel and carried (spilled) through the tile loop
            const __amdgpu_buffer_rsrc_t rsl =
                __builtin_amdgcn_make_buffer_rsrc(a.slabs + ((size_t)chunk * n_mat) * rp * Fp, 0, rp * Fp * 4, 0x00020000);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 o = {g[4 * gq], g[4 * gq + 1], g[4 * gq + 2], g[4 * gq + 3]};
                buf_store_b128(rsl, (t * Fp + 4 * h) * 4, ((kap * 32) * Fp + phi * 32 + 8 * gq) * 4, o);
            }
        }
        SNMF_STAMP(11);
        SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * 8 + w) * 12, 12);
        SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * 8 + w);
    }
    const int tid_e = w * 64 + fresh_lane();  // (not threadIdx.x: that register would live -- spilled -- through the whole kernel)
    for (int k = tid_e; k < rp; k += NTHR) {
        float sk = 0.f;
        for (int p = 0; p < NP; ++p) sk += sred[p * rp + k];
        a.spart[(size_t)chunk * rp + k] = sk;
    }
    if (OBJ && tid_e == 0) {
        double d = 0.0, s2 = 0.0;
        for (int i = 0; i < NP; ++i) {
            d += dred[i];
            s2 += dred[NP + i];
        }
        a.part[2 * chunk] = d;
        a.part[2 * chunk + 1] = s2;
    }
}

}  // namespace snmf
