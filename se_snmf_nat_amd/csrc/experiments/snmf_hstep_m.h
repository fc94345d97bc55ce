// snmf_hstep_m.h -- k_hstep_m: the KL H half-step with the roles MERGED (round 5).
//
// Same arithmetic per tile as k_hstep<8,1,4,BM_KL> / k_hstep_rp (src/sparse_nmf.m:189-208, + :248-261 of the previous iterate):
// P1 Lam = W*H -> ratio = V ./ max(Lam, flr) (in place over nothing: V comes from registers, the ratio goes to the LDS image)
// -> P2 W^T*ratio -> H <- H .* num ./ dph.  Different SCHEDULE: k_hstep_rp runs three waves per SIMD -- an A wave (P1 of tile
// j+1), a B wave (P2 of tile j), a loader wave -- that hand the tile to each other through six LDS signals.  Under
// v_mfma_f32_32x32x2_f32 a SIMD is a sequential machine (profiles/r02_experiments.md: while a wave issues these MFMAs no other
// wave of its SIMD issues anything), so three waves buy no overlap; what they cost is the hand-offs: every wait of one role
// for another is a stretch in which the SIMD has no MFMA to issue unless its other MFMA wave happens to be inside a loop,
// and a wave that has VALU work (an epilogue) beside a partner in an MFMA loop is starved until that loop ends (stamps:
// the B team's 140-instruction epilogue took 10.7 k cycles).  Here ONE wave per SIMD does everything for its own rows and
// columns of the tile, in program order:
//     wave w:  P1 of row tiles w, w+4  ->  epilogue 1 (ratio rows -> LDS image)  ->  P2 of column tiles w, w+4 over ALL ratio
//              rows  ->  epilogue 2 (H update of its columns, stored straight from the registers)
// Per tile a wave waits for its three peers exactly twice: `rdone` (every wave's ratio rows are in the image: the step
// between epilogue 1 and P2 -- the four waves do identical work, so they arrive together) and `hready` (every wave's share
// of the next H block has landed: issued a whole P2 earlier, never late).  No loader waves: the H block of tile j+1 comes
// by LDS-DMA (8 one-KiB pieces per wave) and the V pieces of tile j+1 go straight into registers in the D-tile layout
// (16 bytes per lane: the rows 8g+4h..+3 of frame fl), both issued at ONE point per tile -- right behind the P1 loop, in
// front of epilogue 1 -- because vmcnt retires in order: an HBM access in front of a waited-for W fragment exposes the whole
// HBM latency, and behind this point comes a thousand-odd cycles of VALU work before the next fragment is waited for.  The
// first two W fragment stages of each contraction are issued BEFORE the epilogue that precedes it (W does not depend on
// the tile), so the L2 round trip of a loop's first fragments is spent in the epilogue.
//
// LDS: two buffers of [Tt][ldh] H block + [Tt][ldr] ratio image (k_hstep_rp's layout and size), the extra row of W, 8 progress
// words.  Buffer hazards, all resolved by program order + the two waits (no further signal):
//   * ratio image (j & 1) is written by epilogue 1 of tile j; its last readers were the P2 loops of tile j-2, and a wave can
//     only be past `rdone(j-1)` when every wave has posted it, i.e. has finished its P2 loop of tile j-2;
//   * H buffer ((j+1) & 1) is refilled (DMA issued behind the P1 loop of tile j) while its last readers were P1 / the extra
//     row / the H values of epilogue 2 (fetched into registers BEFORE rdone is posted) of tile j-1: every wave posted
//     rdone(j-1) after them, and this wave has waited for that;
//   * the DMA pieces a wave issued behind P1(j) have landed when its P2(j) loop has consumed a W fragment issued after them
//     (in-order return), so `hready(j+1)` is posted behind that loop.
// Envelope (host: snmf_api.hip): KL, H update, 8 row tiles (+ the extra row), 8 column tiles, no r x T sparsity matrix; every
// tile whole (no split last round yet).  Bit-identical H to k_hstep / k_hstep_rp: same MFMA order per output tile.
#pragma once
#include "../snmf_kernels.h"

namespace snmf {

// one k-block stage of NA output tiles' W fragments
template <int NA>
__device__ __forceinline__ void hm_ldw(f32x4 (&w)[NA], __amdgpu_buffer_rsrc_t rs, int voff, const int (&soff)[NA], int q) {
#pragma unroll
    for (int i = 0; i < NA; ++i) w[i] = ldw_buf(rs, voff, soff[i] + q * 1024);
}

// contract_shared_buf with the first two stages' W fragments ALREADY in flight (wA = block 0, wB = block 1, issued by the
// caller in front of whatever it does before the loop).  Same MFMA order, same prefetch distance.
template <int NA>
__device__ __forceinline__ void hm_contract(f32x16 (&acc)[NA], __amdgpu_buffer_rsrc_t rs, int voff, const int (&soff)[NA],
                                            const float* sp, int nq, f32x4 (&wA)[NA], f32x4 (&wB)[NA]) {
    f32x4 wC[NA], sA, sB, sC;
    auto mm = [&](const f32x4 (&w)[NA], const f32x4& sf) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] = mfma32(w[i][e], sf[e], acc[i]);
    };
    const float* bp = sp;  // moving base: block q + j at bp + 8 * j
    sA = *reinterpret_cast<const f32x4*>(bp);
    sB = *reinterpret_cast<const f32x4*>(bp + 8);
    int q = 0;
    for (; q + 2 < nq; q += 3) {
        hm_ldw<NA>(wC, rs, voff, soff, q + 2);
        sC = *reinterpret_cast<const f32x4*>(bp + 16);
        SNMF_PIN();
        mm(wA, sA);
        hm_ldw<NA>(wA, rs, voff, soff, q + 3);
        sA = *reinterpret_cast<const f32x4*>(bp + 24);
        SNMF_PIN();
        mm(wB, sB);
        hm_ldw<NA>(wB, rs, voff, soff, q + 4);
        sB = *reinterpret_cast<const f32x4*>(bp + 32);
        bp += 24;
        SNMF_PIN();
        mm(wC, sC);
    }
    if (q < nq) mm(wA, sA);
    if (q + 1 < nq) mm(wB, sB);
}

// epilogue 1 of one 32-row tile: Lam -> ratio, V from REGISTERS (v[g] = rows 8g + 4h .. + 3 of frame fl), ratio -> LDS image.
// Arithmetic and summation order of rp_p1_epilogue_t.
template <bool OBJ, bool MASKED>
__device__ __forceinline__ void hm_p1_epilogue_t(const StepArgs& a, const f32x16& acc, const f32x4 (&v)[4], float* Rs, int phi, int t0,
                                                 int lane, float& dsum) {
    const int fl = lane & 31, h = lane >> 5;
    const int t = t0 + fl;
    float* rsp = Rs + fl * a.ldr + phi * 32 + 4 * h;
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
__device__ __forceinline__ void hm_p1_epilogue(const StepArgs& a, const f32x16& acc, const f32x4 (&v)[4], float* Rs, int phi, int t0,
                                               int lane, float& dsum) {
    if (!OBJ || (phi * 32 + 32 <= a.F && t0 + 32 <= a.T)) hm_p1_epilogue_t<OBJ, false>(a, acc, v, Rs, phi, t0, lane, dsum);
    else hm_p1_epilogue_t<OBJ, true>(a, acc, v, Rs, phi, t0, lane, dsum);
}

// epilogue 2 of one 32-column tile: H <- H .* num .* (1 ./ dph), H from REGISTERS (hov[g] = columns kap*32 + 8g + 4h .. + 3 of
// frame fl, fetched from the LDS block before the block was released), the result stored straight to HBM: 16 bytes per lane,
// the two lane halves of a frame adjacent.  rp_p2_epilogue's arithmetic (scalar / r-vector sparsity).
template <bool OBJ>
__device__ __forceinline__ void hm_p2_epilogue(const StepArgs& a, const f32x16& acc, const f32x4 (&hov)[4], __amdgpu_buffer_rsrc_t rso,
                                               int voff, int kap, int lane, const f32x4 (&dpf)[4], float& shsum) {
    const int h = lane >> 5;
    f32x4 spv[4];
    if (OBJ && !a.lam_is_u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) spv[g] = *reinterpret_cast<const f32x4*>(a.lamk + kap * 32 + 8 * g + 4 * h);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = hov[g][j] * acc[4 * g + j] * dpf[g][j];
        buf_store_b128(rso, voff, (kap * 32 + 8 * g) * 4, o);
    }
    if (OBJ) {
        if (a.lam_is_u) {
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

template <bool OBJ>
__global__ __launch_bounds__(256, 1) void k_hstep_m(StepArgs a) {
    constexpr int NW = 4, Tt = 32, NTHR = NW * 64;
    constexpr int PR = Tt / NW;  // H rows per wave and tile (one 1 KiB DMA piece each)
    if (a.stop && *a.stop) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int lane = threadIdx.x & 63, w = wave_index();
    const int fl = lane & 31, h = lane >> 5;
    const int rp = a.rp, Fp = a.Fp, ldh = a.ldh, ldr = a.ldr;
    const int bufsz = Tt * (ldh + ldr);  // floats per buffer: Hs [Tt][ldh] then Rs [Tt][ldr]
    float* wxs = lds + 2 * bufsz;        // [rp] extra row of W
    unsigned* cnt = reinterpret_cast<unsigned*>(wxs + rp);
    unsigned *hready = cnt, *rdone = cnt + 4;
    double acc_div = 0.0, acc_sh = 0.0;
    if (a.xr) {
        for (int k = threadIdx.x; k < rp; k += NTHR) wxs[k] = a.wx[k];
        // the 7 unused cells of the extra 8-deep k-block stay zero for the whole kernel
        for (int i = threadIdx.x; i < 2 * Tt * 8; i += NTHR) {
            const int bsel = i / (Tt * 8), ii = i - bsel * Tt * 8;
            lds[bsel * bufsz + Tt * ldh + (ii >> 3) * ldr + a.Fm + (ii & 7)] = 0.f;
        }
    }
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int nmy = (int)blockIdx.x < a.n_tiles ? (a.n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    auto tile_of = [&](int j) { return (int)blockIdx.x + j * (int)gridDim.x; };

    // ---- this wave's tiles and operand addresses (all scalar but the fixed lane offsets) ----
    const int phi0 = w, phi1 = w + NW;  // row tiles    (host: nf == 8)
    const int kap0 = w, kap1 = w + NW;  // column tiles (host: nk == 8)
    const __amdgpu_buffer_rsrc_t rst = wimage_rsrc(a.Wt4, (size_t)a.nf * rp * 32);
    const __amdgpu_buffer_rsrc_t rsk = wimage_rsrc(a.Wk4, (size_t)a.nk * a.Fq * 32);
    const int so1[2] = {phi0 * rp * 128, phi1 * rp * 128};
    const int so2[2] = {kap0 * a.Fq * 128, kap1 * a.Fq * 128};
    const int wv = lane * 16;
    const int vlane = (fl * Fp + 4 * h) * 4;  // V piece of this lane inside a tile: frame fl, rows 4h .. (+ 32 phi + 8 g: scalar)
    const int hlane = (fl * rp + 4 * h) * 4;  // H_out piece of this lane inside a tile
    const int nq2 = a.Fq / 8;
    f32x4 dp0[4], dp1[4];
    rp_p2_consts(a, kap0, lane, dp0);
    rp_p2_consts(a, kap1, lane, dp1);

    auto dma_h = [&](int tile, float* dstH) {  // this wave's PR rows of the tile's H block: HBM -> LDS, no registers
        const __amdgpu_buffer_rsrc_t rh =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Hin + (size_t)tile * Tt * rp), 0, Tt * rp * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const int t = w + NW * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dstH + t * ldh), 16, wv, t * rp * 4, 0, 0);
        }
    };
    // V pieces of this wave's two row tiles (+ its 8 frames of the extra row: lanes 16c' .. of frame 8w + 4c + (lane >> 4))
    auto ld_v = [&](int tile, f32x4 (&v0)[4], f32x4 (&v1)[4], float (&vx)[2]) {
        const __amdgpu_buffer_rsrc_t rv =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.V + (size_t)tile * Tt * Fp), 0, Tt * Fp * 4, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            v0[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vlane, (phi0 * 32 + 8 * g) * 4, 0));
            v1[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vlane, (phi1 * 32 + 8 * g) * 4, 0));
        }
        if (a.xr) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                vx[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, ((PR * w + 4 * c + (lane >> 4)) * Fp + a.Fm) * 4, 0, 0));
        }
    };

    f32x4 vc0[4], vc1[4], vn0[4], vn1[4];
    float vxc[2] = {0.f, 0.f}, vxn[2] = {0.f, 0.f};
    f32x4 wA1[2], wB1[2];
    if (nmy > 0) {
        dma_h(tile_of(0), lds);
        ld_v(tile_of(0), vc0, vc1, vxc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing else orders a ds_read behind an LDS-DMA
        rp_post(hready, w, 1u, lane);
        hm_ldw<2>(wA1, rst, wv, so1, 0);
        hm_ldw<2>(wB1, rst, wv, so1, 1);
    }
    SNMF_STAMP_DECL
    for (int j = 0; j < nmy; ++j) {
        const int tile = tile_of(j), t0 = tile * Tt;
        SNMF_STAMP(11);
        float* Hs = lds + (j & 1) * bufsz;
        float* Rs = Hs + Tt * ldh;
        const bool more = j + 1 < nmy;
        // ---------------- P1: Lam rows of row tiles w, w + 4 ----------------
        rp_await(hready, (unsigned)(j + 1), a.stop);
        SNMF_STAMP(0);
        f32x16 acc[2] = {zero16(), zero16()};
        hm_contract<2>(acc, rst, wv, so1, Hs + fl * ldh + 4 * h, a.nqk, wA1, wB1);
        SNMF_STAMP(4);
        // ---------------- the tile's HBM reads, all here (see the header) + P2's first fragments ----------------
        if (more) {
            dma_h(tile_of(j + 1), lds + ((j + 1) & 1) * bufsz);
            ld_v(tile_of(j + 1), vn0, vn1, vxn);
        }
        f32x4 wA2[2], wB2[2];
        hm_ldw<2>(wA2, rsk, wv, so2, 0);
        hm_ldw<2>(wB2, rsk, wv, so2, 1);
        SNMF_PIN();
        SNMF_STAMP(1);
        // ---------------- epilogue 1: ratio rows -> image; the extra row's 8 frames; this wave's H values of epilogue 2 ----------------
        float dsum = 0.f;
        hm_p1_epilogue<OBJ>(a, acc[0], vc0, Rs, phi0, t0, lane, dsum);
        hm_p1_epilogue<OBJ>(a, acc[1], vc1, Rs, phi1, t0, lane, dsum);
        if (OBJ) acc_div += (double)dsum;
        SNMF_STAMP(5);
        f32x4 ho0[4], ho1[4];
        {
            const float* hsp = Hs + fl * ldh + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ho0[g] = *reinterpret_cast<const f32x4*>(hsp + kap0 * 32 + 8 * g);
                ho1[g] = *reinterpret_cast<const f32x4*>(hsp + kap1 * 32 + 8 * g);
            }
        }
        if (a.xr) {  // lam_x[t] = sum_k W[Fm, k] H[k, t]: hstep_p1_xrow's arithmetic, V[Fm, t] from the prefetched registers
            float dsx = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int tl = PR * w + 4 * c + (lane >> 4), kl = lane & 15;
                const float* hrow = Hs + tl * ldh;
                float s0 = 0.f, s1 = 0.f;
                for (int k = 4 * kl; k < rp; k += 64) {
                    const f32x4 wv4 = *reinterpret_cast<const f32x4*>(wxs + k);
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + k);
                    s0 += wv4[0] * hv[0] + wv4[1] * hv[1];
                    s1 += wv4[2] * hv[2] + wv4[3] * hv[3];
                }
                const float s = row_sum_f(s0 + s1);
                if (kl == 0) {
                    const int t = t0 + tl;
                    const float v = vxc[c];
                    const float lam = fmaxf(s, kFlr);
                    if (OBJ) dsx += (t < a.T) ? div_term<BM_KL>(v, lam, a.beta, a.inv_bb1) : 0.f;
                    Rs[tl * ldr + a.Fm] = v * fast_rcp(lam);
                }
            }
            if (OBJ) acc_div += (double)dsx;
        }
        SNMF_STAMP(6);
        rp_post(rdone, w, (unsigned)(j + 1), lane);
        // ---------------- P2: W^T * ratio for column tiles w, w + 4 over every ratio row ----------------
        rp_await(rdone, (unsigned)(j + 1), a.stop);
        SNMF_STAMP(7);
        f32x16 ac2[2] = {zero16(), zero16()};
        hm_contract<2>(ac2, rsk, wv, so2, Rs + fl * ldr + 4 * h, nq2, wA2, wB2);
        SNMF_STAMP(9);
        // (the P2 loop has consumed fragments issued after this wave's DMA pieces of tile j + 1: they have landed)
        if (more) {
            rp_post(hready, w, (unsigned)(j + 2), lane);
            hm_ldw<2>(wA1, rst, wv, so1, 0);  // P1 of the next tile: its first fragments fly through epilogue 2
            hm_ldw<2>(wB1, rst, wv, so1, 1);
        }
        SNMF_PIN();
        // ---------------- epilogue 2: H update of this wave's columns, registers -> HBM ----------------
        {
            const __amdgpu_buffer_rsrc_t rso =
                __builtin_amdgcn_make_buffer_rsrc(a.Hout + (size_t)tile * Tt * rp, 0, Tt * rp * 4, 0x00020000);
            float shsum = 0.f;
            hm_p2_epilogue<OBJ>(a, ac2[0], ho0, rso, hlane, kap0, lane, dp0, shsum);
            hm_p2_epilogue<OBJ>(a, ac2[1], ho1, rso, hlane, kap1, lane, dp1, shsum);
            if (OBJ) acc_sh += (double)shsum;
        }
        SNMF_STAMP(10);
        if (more) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vc0[g] = vn0[g];
                vc1[g] = vn1[g];
            }
            vxc[0] = vxn[0];
            vxc[1] = vxn[1];
        }
    }

    SNMF_STAMP_OUT(a.prof + ((size_t)blockIdx.x * 8 + w) * 12, 12);
    SNMF_STAMP_CLK(a.prof, (size_t)blockIdx.x * 8 + w);
    if (OBJ) {
        // deterministic workgroup reduction of the two fp64 partial sums (as k_hstep)
        __syncthreads();
        double* red = reinterpret_cast<double*>(lds);  // [2][NTHR]
        red[threadIdx.x] = acc_div;
        red[NTHR + threadIdx.x] = acc_sh;
        __syncthreads();
        for (int s = NTHR / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                red[threadIdx.x] += red[threadIdx.x + s];
                red[NTHR + threadIdx.x] += red[NTHR + threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            a.part[2 * blockIdx.x] = red[0];
            a.part[2 * blockIdx.x + 1] = red[NTHR];
        }
    }
}

}  // namespace snmf
