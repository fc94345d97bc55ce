// snmf_smallr.h -- the KL half-steps for SMALL RANK on tall spectrograms: r <= 32 components (ONE 32-column tile; the kernels are
// written for NK column tiles but two do not fit the registers) on 3..16 row tiles (F = 65..544: the reference's R = 20 / 10, 30 settings at F = 513,
// settings/bak_IS16_results/initial_setting_SNMF_Techwin_201603_RT.m:47-48, initial_setting_IMCRA.m:47-48, and r = 32 at F = 257).
//
// Why a family of its own (round 6).  At r <= 64 a 32-frame tile is a few hundred MFMAs for a whole compute unit and 70 KB of
// spectrogram: the role pipeline (k_hstep_rp<., CUT> + k_wstats with loader waves) runs it as the chain loader -> A team -> B team
// -> loader on two LDS tile buffers and was bound by that chain, not by a roof (profiles/r05_experiments.md section 4: MFMA pipe
// 32-35 % busy at 2.2-2.5 TB/s; 2.5x from either roof at r = 20).  The small-F family (snmf_smallf.h) showed what works when a
// tile is small: no staging, no roles -- every wave loads its operands straight into the MFMA layouts and nothing is handed between
// waves per k-block.  A whole tile per wave does not carry over, though: 72 000 frames are 2250 tiles, 2.2 per SIMD -- the busiest
// SIMD would run 3 tiles where 2.2 are needed.  So here a tile belongs to a WORKGROUP of eight waves, cut by ROW TILES:
//
//   k_hstep_sr   wave w takes the row tiles phi = w, w + 8 of every tile of its workgroup: it loads the H block (4 KB, the same for
//                all eight waves: L2) and ITS rows of V straight into the operand layouts of k_hstep_sf (one tile ahead, in
//                registers), forms Lam = W*H for its rows (W fragments of P1 from L2 through a buffer descriptor, a tile-independent
//                address stream), the ratio in registers, and contracts W^T*ratio over ITS rows only -- a PARTIAL numerator
//                [32 NK x 32].  The eight partials meet in LDS; the tile's REDUCER (wave tile % 8: the role rotates, so no SIMD
//                carries it alone) adds them in wave order, handles the extra row (F = 32 n + 1) on the VALU, applies the update
//                and stores the H block the way it came.  Two progress signals per tile (wrote / freed), no workgroup barrier in
//                the tile loop, so the waves drift apart and hide each other's latencies.  W^T's image (P2's A operand) is the
//                only big LDS resident.
//   k_wstats_sr  wave w owns the statistics rows phi = w, w + 8 for the workgroup's whole frame chunk (G tiles in registers: no
//                cross-wave reduction at all, the slab is written once as k_wstats writes it); per tile Lam'^T from the lane's H
//                pieces and W fragments (L2), the ratio in registers is the A operand of G += ratio * H^T; row sums of H ride on
//                the B-operand reads (wave 0), the extra row's slab row is accumulated per lane and reduced once per chunk.
//
// Same arithmetic per element as the other KL kernels (same MFMA order over the contraction within a row tile, same epilogue
// expressions); the numerator is summed over row tiles in another order than k_hstep_rp's (wave order instead of one chain), i.e.
// to fp32 rounding of the sum -- tests/test_gpu_parity.py states the tolerance against the plain kernels and the oracle.
#pragma once
#include "snmf_kernels.h"
#include "snmf_smallf.h"  // sf_fill_image, sf_post / sf_await
#include <type_traits>

namespace snmf {

// (kSrWaves, sr_hstep_lds_bytes, sr_wstats_lds_bytes: snmf_kernels.h -- the host's geometry selection needs them without this header)

// max(x, 1e-9) as ONE instruction: fmaxf(x, c) compiles to a canonicalising v_max_f32 x, x in front of the real one (IEEE mode: a
// signalling NaN must be quieted); v_max_f32 itself already returns the other operand for a NaN, which is MATLAB's max(x, flr).
__device__ __forceinline__ float sr_floor(float x) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(kFlr), "v"(x));
    return r;
}

// wait until all EIGHT progress words of an array have reached `target`: one 32-byte read per poll (polling the words one after the
// other cost seven LDS round trips per wait, two waits per tile: more than the reducer chain it replaced)
__device__ __forceinline__ void sr_await8(const unsigned* words, unsigned target, const int* stop) {
    int spin = 0;
    for (;;) {
        const u32x4_t x = *reinterpret_cast<const volatile u32x4_t*>(words), y = *reinterpret_cast<const volatile u32x4_t*>(words + 4);
        const unsigned m0 = x[0] < x[1] ? x[0] : x[1], m1 = x[2] < x[3] ? x[2] : x[3], m2 = y[0] < y[1] ? y[0] : y[1], m3 = y[2] < y[3] ? y[2] : y[3];
        const unsigned m01 = m0 < m1 ? m0 : m1, m23 = m2 < m3 ? m2 : m3;
        if ((m01 < m23 ? m01 : m23) >= target) break;
        if (++spin > kSpinLimit) {
            raise_fault(stop);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    stress_jitter();
}

// The exchange of the partial numerators.  First form (one REDUCER wave per tile, the role rotating): the reducer's extra work --
// wait for seven partials, add them, the extra row, the update, the store: ~2 k cycles -- made it late for the NEXT tile, whose
// reducer then waited for its partial: a chain of late reducers, 12 us of a 70 us launch (measured with the exchange compiled out),
// and the per-tile rendezvous kept the two waves of every SIMD in lockstep (both in their MFMA loops, then both in their epilogues:
// no MFMA / VALU overlap between SIMD partners).  Now EVERY wave finishes a SLICE of the tile (two of the sixteen registers of the
// numerator tile: wave w the columns k = 8 (w >> 1) + 4 h + 2 (w & 1) + {0, 1}), one tile LATE: the partials of tile j are written
// into buffer j & 1 behind the MFMAs of tile j and summed behind the MFMAs of tile j + 1, when every wave has long posted them --
// a full tile of slack, so the waves of a workgroup may drift up to a tile apart and the start-up stagger of each SIMD's second
// wave persists.  Two progress words per wave: wrote (partials written) and rdone (slices read: the buffer may be rewritten).
template <int NK, bool OBJ>
__global__ __launch_bounds__(kSrWaves * 64, 2) void k_hstep_sr(StepArgs a) {
    static_assert(NK == 1, "the slice deal below is for one column tile (16 numerator registers, 2 per wave)");
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp;
    const int nq8 = rp / 8, nqf = a.Fq / 8;
    float* const wk = lds;                                          // Wk4 image
    float* const wxs = wk + (size_t)NK * a.Fq * 32;                 // [rp] extra row of W
    float* const rdp = wxs + rp;                                    // [rp] 1 ./ dph (scalar / per-row sparsity)
    float* const lmk = rdp + rp;                                    // [rp] lambda_k
    float* const ps = lmk + rp;                                     // [2 buffers][8 waves][4 g][64 lanes][4] partial numerators
    float* const rxs = ps + (size_t)2 * kSrWaves * 1024;            // [2 buffers][32] ratio of the extra row per frame (formed by one wave per tile)
    unsigned* const wrote = reinterpret_cast<unsigned*>(rxs + 64);  // [8] tiles whose partial wave w has written
    unsigned* const rdone = wrote + 8;                              // [8] tiles whose slice wave w has read
    double* const red = reinterpret_cast<double*>(wrote + 32);      // [2][8]
    const int t = lane & 31, h = lane >> 5;

    sf_fill_image(a.Wk4, wk, NK * a.Fq * 32 * 4, w, lane);
    for (int k = threadIdx.x; k < rp; k += kSrWaves * 64) {
        wxs[k] = a.xr ? a.wx[k] : 0.f;
        rdp[k] = a.S ? 0.f : fast_rcp(a.dphv[k]);
        lmk[k] = a.S ? 0.f : a.lamk[k];
    }
    if (threadIdx.x < 32) wrote[threadIdx.x] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing else orders a ds_read behind an LDS-DMA)
    __syncthreads();

    // this workgroup's tiles: b, b + G, ...; this wave's row tiles: w and w + 8 (nf <= 16)
    const int G = (int)gridDim.x;
    const int nmy = (int)blockIdx.x < a.n_tiles ? (a.n_tiles - 1 - (int)blockIdx.x) / G + 1 : 0;
    const bool has0 = w < a.nf, has1 = w + kSrWaves < a.nf;
    const __amdgpu_buffer_rsrc_t rsw = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);
    const f32x4* const wkl = reinterpret_cast<const f32x4*>(wk) + lane;  // fragment (kap, q): wkl[(kap * nqf + q) * 64]
    // this wave's slice of every tile: numerator registers 2 w, 2 w + 1 = group sg, elements 2 sh, 2 sh + 1 <-> columns k0, k0 + 1
    const int sg = w >> 1, sh = w & 1, k0 = 8 * sg + 4 * h + 2 * sh;

    f32x4 hq[NK * 4], vq[2][4], hn[NK * 4], vn[2][4];
    float vx = 0.f, vxn = 0.f;
    auto load_tile = [&](int tile, f32x4 (&H)[NK * 4], f32x4 (&V)[2][4], float& x) {
        const float* hp = a.Hin + ((size_t)tile * 32 + t) * rp + 4 * h;
        const float* vp = a.V + ((size_t)tile * 32 + t) * Fp + 4 * h;
#pragma unroll
        for (int q = 0; q < NK * 4; ++q) H[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
        if (has0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) V[0][g] = *reinterpret_cast<const f32x4*>(vp + w * 32 + 8 * g);
        }
        if (has1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) V[1][g] = *reinterpret_cast<const f32x4*>(vp + (w + kSrWaves) * 32 + 8 * g);
        }
        if (a.xr) x = a.V[((size_t)tile * 32 + t) * Fp + a.Fm];  // the extra row's value of this lane's frame
    };
    double acc_div = 0.0, acc_sh = 0.0;
    // P1's A operand: this wave's row tiles are the SAME for every tile of the kernel, so their W fragments (2 row tiles x ceil(r / 8)
    // k-blocks x 16 bytes per lane) stay in REGISTERS.  (First build: fetched from L2 per tile through the buffer path -- vmcnt retires
    // in issue order, so the first fragment of every tile waited behind the NEXT tile's operand prefetch from HBM: 86 us against the
    // role pipeline's 72 at 513 x 72000, r = 20.)  No vector-memory load is left inside a tile but the prefetch itself.
    f32x4 wr[2][NK * 4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < NK * 4; ++q)
            wr[s][q] = (s == 0 ? has0 : has1) ? ldw_buf(rsw, lane * 16, ((w + s * kSrWaves) * nq8 + q) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (nmy > 0) load_tile((int)blockIdx.x, hq, vq, vx);
    // the second wave of each SIMD starts late by about half a row tile's work (k_hstep_sf's stagger): its MFMA loops then run under
    // its partner's epilogues for the rest of the kernel -- nothing below re-synchronises them to less than a tile
    if (a.stagger > 0 && w >= 4) {
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - ts < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(16);
    }

    auto tiles = [&](auto sk_tag) {
    constexpr int SK = decltype(sk_tag)::value;
    SNMF_STAMP_DECL
    // state of the tile whose slice is still to be finished (tile j - 1 while tile j is computed)
    float ho0 = 0.f, ho1 = 0.f;
    // finish the slice of tile jp (its partials sit in buffer jp & 1): sum in wave order, extra row, update, store
    auto finish = [&](int jp, float my0, float my1) {
        const int t0p = ((int)blockIdx.x + jp * G) * 32;
        sr_await8(wrote, (unsigned)(jp + 1), a.stop);  // (this wave's own word passes by program order)
        SNMF_STAMP(7);
        const float* src = ps + (size_t)(jp & 1) * kSrWaves * 1024 + (sg * 64 + lane) * 4 + 2 * sh;
        float n0 = 0.f, n1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < kSrWaves; ++ww) {  // wave order, this wave's own partial (still in registers) in its place
            f32x2 x;
            if (ww == w) x = f32x2{my0, my1};
            else x = *reinterpret_cast<const f32x2*>(src + ww * 1024);
            n0 += x[0];
            n1 += x[1];
        }
        // (ratio_x: posted with the partials by the tile's x-wave -- read BEFORE this wave reports the buffer as read: a first build
        //  read it behind the post, and the x-wave of tile jp + 2 could overwrite it in between; 1 % errors in H on a few frames,
        //  found by tests/test_gpu_fuzz.py[pipe] and test_pipelined_kernels_equal_plain_kernels[F257_r32_T26000])
        const float rx_p = a.xr ? rxs[(jp & 1) * 32 + t] : 0.f;
        sf_post(rdone + w, (unsigned)(jp + 1), lane);
        if (a.xr) {  // the extra row's k-block of W^T * ratio: W[Fm, k] * ratio_x[t]
            n0 += wxs[k0] * rx_p;
            n1 += wxs[k0 + 1] * rx_p;
        }
        float dp0, dp1, sp0 = 0.f, sp1 = 0.f;
        if constexpr (SK == 2) {
            const f32x2 sp = *reinterpret_cast<const f32x2*>(a.S + ((size_t)t0p + t) * rp + k0);
            sp0 = sp[0];
            sp1 = sp[1];
            dp0 = fast_rcp(fmaxf(a.colsum[k0] + sp0, kFlr));
            dp1 = fast_rcp(fmaxf(a.colsum[k0 + 1] + sp1, kFlr));
        } else {
            dp0 = rdp[k0];
            dp1 = rdp[k0 + 1];
            if constexpr (OBJ && SK == 1) {
                sp0 = lmk[k0];
                sp1 = lmk[k0 + 1];
            }
        }
        const f32x2 o = {ho0 * n0 * dp0, ho1 * n1 * dp1};
        *reinterpret_cast<f32x2*>(a.Hout + ((size_t)t0p + t) * rp + k0) = o;
        if constexpr (OBJ) {
            if constexpr (SK == 0) acc_sh += (double)(a.lam_u * (ho0 + ho1));
            else acc_sh += (double)(sp0 * ho0 + sp1 * ho1);
        }
    };
    float my0 = 0.f, my1 = 0.f;  // this wave's own partial of its slice, tile j - 1
    for (int j = 0; j < nmy; ++j) {
        const int tile = (int)blockIdx.x + j * G, t0 = tile * 32;
        // the next tile's operands are requested before this tile's MFMAs (a whole tile period ahead of their use)
        if (j + 1 < nmy) load_tile(tile + G, hn, vn, vxn);
        f32x16 num = zero16();
        float dsum = 0.f;
        SNMF_STAMP(0);  // (diagnostic builds: issue of the next tile's operand loads)
        // ---- this wave's row tiles: P1 (A = resident W fragment, B = the lane's H pieces), ratio in place over V, P2 partial ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0 ? !has0 : !has1) continue;
            const int phi = w + s * kSrWaves;
            f32x16 acc = zero16();
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                if (q > 4 * (NK - 1) && q >= a.nqk) break;  // (zero padding past ceil(r / 8): see k_hstep_sf)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(wr[s][q][e], hq[q][e], acc);
            }
            SNMF_STAMP(1);  // P1
            {   // ratio in place over V: lane (t, h), element (g, jj) <-> f = 32 phi + 8 g + 4 h + jj  (rp_p1_epilogue)
                const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float v = vq[s][g][jj];
                        const float lam = sr_floor(acc[4 * g + jj]);
                        if (OBJ) {
                            const float d = div_term<BM_KL>(v, lam, a.beta, a.inv_bb1);
                            if (edge) dsum += (phi * 32 + 8 * g + 4 * h + jj < a.F && t0 + t < a.T) ? d : 0.f;
                            else dsum += d;
                        }
                        vq[s][g][jj] = v * fast_rcp(lam);
                    }
                    if (OBJ) __builtin_amdgcn_sched_barrier(0);
                }
            }
            SNMF_STAMP(2);  // ratio
            // P2 over this row tile's four k-blocks: num += W[32 phi .., :]^T * ratio
            {
                f32x4 wa = wkl[(4 * phi) * 64], wb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q + 1 < 4) wb = wkl[(4 * phi + q + 1) * 64];
                    SNMF_PIN();
#pragma unroll
                    for (int e = 0; e < 4; ++e) num = mfma32(wa[e], vq[s][q][e], num);
                    wa = wb;
                }
            }
            SNMF_STAMP(3);  // P2
        }
        if (OBJ) acc_div += (double)dsum;
        // ---- the extra row (F = 32 n + 1): lam_x[t] = sum_k W[Fm, k] H[k, t] from the lane's H pieces (the other half of the components
        // sits in lane t + 32), by ONE wave per tile (the role rotates; all eight forming it cost 0.9 k cycles per tile and wave) ----
        float rx = 0.f;
        const bool x_wave = a.xr && w == (j & (kSrWaves - 1));
        if (x_wave) {
            float s0 = 0.f;
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wxs + 8 * q + 4 * h);
                s0 += (wv[0] * hq[q][0] + wv[1] * hq[q][1]) + (wv[2] * hq[q][2] + wv[3] * hq[q][3]);
            }
            const float so = __shfl_xor(s0, 32, 64);
            const float lamx = sr_floor(h == 0 ? s0 + so : so + s0);  // (the half with h = 0 first on both lanes)
            if (OBJ && h == 0 && t0 + t < a.T) acc_div += (double)div_term<BM_KL>(vx, lamx, a.beta, a.inv_bb1);
            rx = vx * fast_rcp(lamx);
        }
        // ---- this tile's partial into buffer j & 1 (once every wave has read its slice of tile j - 2 out of it) ----
        SNMF_STAMP(4);  // extra row
        if (j >= 2) sr_await8(rdone, (unsigned)(j - 1), a.stop);
        SNMF_STAMP(5);  // wait: buffer free
        {
            float* dst = ps + ((size_t)(j & 1) * kSrWaves + w) * 1024 + lane * 4;  // [4 g][64 lanes][4]
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(dst + g * 256) = f32x4{num[4 * g], num[4 * g + 1], num[4 * g + 2], num[4 * g + 3]};
            if (x_wave && h == 0) rxs[(j & 1) * 32 + t] = rx;
            sf_post(wrote + w, (unsigned)(j + 1), lane);
        }
        SNMF_STAMP(6);  // write partial
        // this tile's slice state for the next round: own partial and old H values (wave-uniform register choice by selects)
        float my0n, my1n, ho0n, ho1n;
        {
            const f32x4 ng = sg == 0 ? f32x4{num[0], num[1], num[2], num[3]} : sg == 1 ? f32x4{num[4], num[5], num[6], num[7]}
                           : sg == 2 ? f32x4{num[8], num[9], num[10], num[11]} : f32x4{num[12], num[13], num[14], num[15]};
            const f32x4 hg = sg == 0 ? hq[0] : sg == 1 ? hq[1] : sg == 2 ? hq[2] : hq[3];
            my0n = sh ? ng[2] : ng[0];
            my1n = sh ? ng[3] : ng[1];
            ho0n = sh ? hg[2] : hg[0];
            ho1n = sh ? hg[3] : hg[1];
        }
        // the prefetched tile becomes the current one -- BEFORE the previous tile's slice is finished: that ends in a store, and
        // vmcnt retires in issue order (as the last thing of the iteration the store's acknowledgement from HBM sat in front of the
        // wait for the prefetched registers: 1.2 k cycles per tile in the phase stamps)
#pragma unroll
        for (int q = 0; q < NK * 4; ++q) hq[q] = hn[q];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) vq[s][g] = vn[s][g];
        vx = vxn;
        SNMF_STAMP(9);  // the wait for the prefetched operands + register moves
        // ---- the slice of the PREVIOUS tile (its partials were posted a tile ago) ----
        if (j >= 1) finish(j - 1, my0, my1);
        SNMF_STAMP(8);  // finish the previous tile's slice (its wait: slot 7)
        my0 = my0n;
        my1 = my1n;
        ho0 = ho0n;
        ho1 = ho1n;
    }
    if (nmy > 0) finish(nmy - 1, my0, my1);
    SNMF_STAMP(8);
    SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * kSrWaves + w) * 12, 12);
    SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * kSrWaves + w);
    };
    if (a.S) tiles(std::integral_constant<int, 2>{});
    else if (a.lam_is_u) tiles(std::integral_constant<int, 0>{});
    else tiles(std::integral_constant<int, 1>{});

    if (OBJ) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            acc_div += __shfl_xor(acc_div, s, 64);
            acc_sh += __shfl_xor(acc_sh, s, 64);
        }
        if (lane == 0) {
            red[w] = acc_div;
            red[kSrWaves + w] = acc_sh;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            double d = 0.0, s = 0.0;
            for (int i = 0; i < kSrWaves; ++i) {
                d += red[i];
                s += red[kSrWaves + i];
            }
            obj_partial_out(a, blockIdx.x, d, s);
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------
// k_wstats_sr: the KL statistics G = (V ./ Lam') * H', s = sum(H, 2) (src/sparse_nmf.m:215-222) for the same shapes.  Workgroup =
// frame chunk (contiguous, balanced range of 32-frame tiles); wave w owns the statistics of row tiles phi = w, w + 8 for the whole
// chunk: 2 NK accumulator tiles in registers, written once into the chunk's slab in k_wstats' layout (k_wfin / k_reduce unchanged).
// Per tile and wave: the lane's H pieces (A operand of P3, lane (t, h)), H with rows in lanes (B operand of P4: 4-byte loads of the
// rows the wave fetched a moment ago), its 2 x 16 values of V in P3's D layout.  The extra row (row group of wave 0 ... handled by
// the wave `xw` below): ratio_x[t] from the lane's H pieces, gx[k] += ratio_x[t] H[k, t] per lane, summed over the lanes once per chunk.
// Dynamic LDS: [rp] row sums, [rp] extra row, wx [rp], [8] doubles.
// -------------------------------------------------------------------------------------------------------------------
template <int NK, bool OBJ>
__global__ __launch_bounds__(kSrWaves * 64, 2) void k_wstats_sr(StepArgs a, int n_chunks, int mat_index, int n_mat) {
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = wave_index();
    const int rp = a.rp, Fp = a.Fp, nq8 = rp / 8;
    const int fl = lane & 31, h = lane >> 5;
    float* const wxs = lds;            // [rp] extra row of W
    float* const sred = wxs + rp;      // [rp] row sums of H (wave 0)
    float* const gxr = sred + rp;      // [rp] extra row of the slab (wave xw)
    double* const dred = reinterpret_cast<double*>(gxr + rp + 16);  // [8]
    const int chunk = blockIdx.x;
    const int tb = (int)(((long long)a.n_tiles * chunk) / n_chunks), te = (int)(((long long)a.n_tiles * (chunk + 1)) / n_chunks);
    for (int k = threadIdx.x; k < rp; k += kSrWaves * 64) wxs[k] = a.xr ? a.wx[k] : 0.f;
    __syncthreads();
    const bool has0 = w < a.nf, has1 = w + kSrWaves < a.nf;
    // the extra row rides on the wave with the fewest row tiles (the last one), the row sums on wave 0
    const int xw = kSrWaves - 1;
    const bool do_x = a.xr && w == xw, do_s = w == 0;
    const __amdgpu_buffer_rsrc_t rsw = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);

    f32x16 G[2][NK];
    float ssum[NK];
    f32x4 gx[NK * 4];  // extra row of the slab, per-lane partial sums in the layout of the H pieces: k = 8 q + 4 h + e
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        G[0][k] = zero16();
        G[1][k] = zero16();
        ssum[k] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < NK * 4; ++q) gx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    double acc_div = 0.0;
    // P3's B operand: this wave's row tiles are the same for the whole chunk -- their W fragments stay in registers (see k_hstep_sr)
    f32x4 wr[2][NK * 4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < NK * 4; ++q)
            wr[s][q] = (s == 0 ? has0 : has1) ? ldw_buf(rsw, lane * 16, ((w + s * kSrWaves) * nq8 + q) * 1024) : f32x4{0.f, 0.f, 0.f, 0.f};

    // Operands of a tile: the lane's H pieces (A operand of P3) and its 2 x 16 values of V in P3's D layout are requested ONE TILE
    // AHEAD (HBM: the V rows are this wave's own; the H block is the same for all eight waves, so whoever asks first brings it to L2);
    // H with rows in lanes (B operand of P4, the block the prefetch fetched a tile ago: L2) is requested at the top of the tile, IN
    // FRONT of the next tile's prefetch in the wave's in-order vmcnt queue, and lands under P3.
    f32x4 hq[NK * 4], hn[NK * 4];
    float v[2][16], vn[2][16], vx = 0.f, vxn = 0.f;
    auto load_ahead = [&](int tile, f32x4 (&H)[NK * 4], float (&V)[2][16], float& x) {
        const int t0 = tile * 32;
        const float* hp = a.Hin + ((size_t)t0 + fl) * rp + 4 * h;
#pragma unroll
        for (int q = 0; q < NK * 4; ++q) H[q] = *reinterpret_cast<const f32x4*>(hp + 8 * q);
        const __amdgpu_buffer_rsrc_t rv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)t0 * Fp), 0, 32 * Fp * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0 ? !has0 : !has1) continue;
            const int vo = ((4 * h) * Fp + (w + s * kSrWaves) * 32 + fl) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                V[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vo, drow(i, 0) * Fp * 4, 0));
        }
        if (do_x) x = a.V[((size_t)t0 + fl) * Fp + a.Fm];
    };
    if (tb < te) load_ahead(tb, hq, v, vx);
    if (a.stagger > 0 && w >= 4) {  // (the second wave of each SIMD starts late: see k_hstep_sr; nothing here ever re-synchronises the waves)
        const unsigned long long ts = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - ts < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(16);
    }

    for (int tile = tb; tile < te; ++tile) {
        const int t0 = tile * 32;
        float b0[16], b1[16];
        const __amdgpu_buffer_rsrc_t rh =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)t0 * rp), 0, 32 * rp * 4, 0x00020000);
        const int ho = ((4 * h) * rp + fl) * 4;
        auto ldb = [&](float (&b)[16], int kap) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                b[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, ho, (drow(i, 0) * rp + kap * 32) * 4, 0));
        };
        ldb(b0, 0);
        if (NK > 1) ldb(b1, 1);
        if (tile + 1 < te) load_ahead(tile + 1, hn, vn, vxn);
        SNMF_PIN();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0 ? !has0 : !has1) continue;
            const int phi = w + s * kSrWaves;
            // ---- P3: Lam'^T[t, f] = sum_k H[k, t] W[f, k]; A = the lane's H pieces, B = W fragment (registers) ----
            f32x16 acc = zero16();
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                if (q > 4 * (NK - 1) && q >= a.nqk) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(hq[q][e], wr[s][q][e], acc);
            }
            // ---- ratio: lane (f = fl, h), register i <-> frame t0 + drow(i, h) ----
            float R[16];
            {
                const int f = phi * 32 + fl;
                const bool edge = OBJ && !(phi * 32 + 32 <= a.F && t0 + 32 <= a.T);
                float dsum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float lam = sr_floor(acc[i]);
                    if (OBJ) {
                        const float d = div_term<BM_KL>(v[s][i], lam, a.beta, a.inv_bb1);
                        if (edge) dsum += (f < a.F && t0 + drow(i, h) < a.T) ? d : 0.f;
                        else dsum += d;
                    }
                    R[i] = v[s][i] * fast_rcp(lam);
                }
                if (OBJ) acc_div += (double)dsum;
            }
            // ---- P4: G[phi, kap] += ratio * H^T ----
#pragma unroll
            for (int i = 0; i < 16; ++i) G[s][0] = mfma32(R[i], b0[i], G[s][0]);
            if (NK > 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) G[s][NK - 1] = mfma32(R[i], b1[i], G[s][NK - 1]);
            }
        }
        // row sums of H (wave 0): from the B operand (rows in lanes, this lane half's 16 frames)
        if (do_s) {
            float s4 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s4 += b0[i];
            ssum[0] += s4;
            if (NK > 1) {
                float s5 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) s5 += b1[i];
                ssum[NK - 1] += s5;
            }
        }
        if (do_x) {  // the extra row: lane (t = fl, h) holds H[8 q + 4 h + e][t]
            float s0 = 0.f;
#pragma unroll
            for (int q = 0; q < NK * 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wxs + 8 * q + 4 * h);
                s0 += (wv[0] * hq[q][0] + wv[1] * hq[q][1]) + (wv[2] * hq[q][2] + wv[3] * hq[q][3]);
            }
            const float so = __shfl_xor(s0, 32, 64);
            const float lamx = fmaxf(h == 0 ? s0 + so : so + s0, kFlr);
            if (OBJ && h == 0 && t0 + fl < a.T) acc_div += (double)div_term<BM_KL>(vx, lamx, a.beta, a.inv_bb1);
            const float rx = vx * fast_rcp(lamx);
#pragma unroll
            for (int q = 0; q < NK * 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) gx[q][e] += rx * hq[q][e];
        }
        // the prefetched tile becomes the current one
#pragma unroll
        for (int q = 0; q < NK * 4; ++q) hq[q] = hn[q];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[s][i] = vn[s][i];
        vx = vxn;
    }

    // ---- slab: D tile lane (k = fl, h), register -> f = 32 phi + drow(reg, h)  (k_wstats' layout) ----
    float* slab = a.slabs + ((size_t)chunk * n_mat + mat_index) * rp * Fp;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 0 ? !has0 : !has1) continue;
        const int phi = w + s * kSrWaves;
#pragma unroll
        for (int kap = 0; kap < NK; ++kap) {
            float* dst = slab + (size_t)(kap * 32 + fl) * Fp + phi * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {G[s][kap][4 * g], G[s][kap][4 * g + 1], G[s][kap][4 * g + 2], G[s][kap][4 * g + 3]};
                *reinterpret_cast<f32x4*>(dst + 8 * g) = o;
            }
        }
    }
    if (do_s) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float other = __shfl_xor(ssum[k], 32, 64);  // the two lane halves hold the two halves of a tile's frames
            if (h == 0) sred[k * 32 + fl] = ssum[k] + other;
        }
    }
    if (do_x) {
        // sum over the 32 frames (lanes t of one half): xor butterfly within the half, a fixed tree; lane t = 0 of each half writes
#pragma unroll
        for (int q = 0; q < NK * 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = gx[q][e];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                if (fl == 0) gxr[8 * q + 4 * h + e] = x;
            }
    }
    if (OBJ) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc_div += __shfl_xor(acc_div, s, 64);
        if (lane == 0) dred[w] = acc_div;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < rp; k += kSrWaves * 64) {
        a.spart[(size_t)chunk * rp + k] = sred[k];
        if (a.xr) slab[(size_t)k * Fp + a.Fm] = gxr[k];
    }
    if (OBJ && threadIdx.x == 0) {
        double d = 0.0;
        for (int i = 0; i < kSrWaves; ++i) d += dred[i];
        a.part[2 * chunk] = d;
        a.part[2 * chunk + 1] = 0.0;
    }
}

}  // namespace snmf
