// Dev microbenchmark (round 2): k_wstats' P3 loop in isolation.  One 256-thread workgroup per CU (one wave per SIMD, nothing
// else resident); every wave runs `reps` contractions of nq k-blocks (4 MFMAs each, ONE dependent accumulator chain) with
// the A fragments from an LDS image and the B fragments from an L2-resident operand image, exactly as contract_sb<1,true,2>
// does, and variants of it.  Reports cycles per MFMA (64 = the matrix pipe's rate).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/p3 scripts/p3_loop_probe.hip && /tmp/p3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// VAR: 0 = the product loop (two named stages of SB=2 k-blocks, fragments one stage ahead)
//      1 = W fragment 0 for every block (no global loads in the loop)   2 = LDS fragment 0   3 = both
//      4 = three named single-block stages, fragments two blocks ahead (contract_shared's loop, one chain)
//      5 = as 0 with the two k-blocks of a stage on two accumulator chains
//      6 = as 0, W through buffer_load (descriptor + SCALAR k-block offset + fixed lane offset: no address VALU), no clamps
//      7 = 6 + LDS fragments at immediate offsets from one moving base (one VALU add per 4 k-blocks)
//      8 = 7 with the loop fully unrolled (nq = 32: every LDS offset an immediate, every W offset a scalar constant)
template <int VAR>
__device__ __forceinline__ void contract_b(f32x16& acc, __amdgpu_buffer_rsrc_t rs, int voff, int soff0, const float* sp, int nq) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 wA[2], wB[2], sA[2], sB[2];
    auto ldw = [&](int q) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff0 + q * 1024, 0)); };
    if (VAR == 6) {
        auto ldstage = [&](f32x4 (&w)[2], f32x4 (&s)[2], int q0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                w[j] = ldw(q0 + j);
                s[j] = *reinterpret_cast<const f32x4*>(sp + 8 * (q0 + j));
            }
        };
        auto mmstage = [&](const f32x4 (&w)[2], const f32x4 (&s)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(s[j][e], w[j][e], acc);
        };
        ldstage(wA, sA, 0);
        for (int q = 0; q < nq; q += 4) {
            ldstage(wB, sB, q + 2);
            PIN();
            mmstage(wA, sA);
            ldstage(wA, sA, q + 4);
            PIN();
            mmstage(wB, sB);
        }
        return;
    }
    auto mmstage = [&](const f32x4 (&w)[2], const f32x4 (&s)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(s[j][e], w[j][e], acc);
    };
    const float* bp = sp;  // moving LDS base: block q+j at bp + 8*j
    wA[0] = ldw(0); wA[1] = ldw(1);
    sA[0] = *reinterpret_cast<const f32x4*>(bp); sA[1] = *reinterpret_cast<const f32x4*>(bp + 8);
    if (VAR == 8) {
#pragma unroll
        for (int q = 0; q < 32; q += 4) {
            wB[0] = ldw(q + 2); wB[1] = ldw(q + 3);
            sB[0] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 2)); sB[1] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 3));
            PIN();
            mmstage(wA, sA);
            wA[0] = ldw(q + 4); wA[1] = ldw(q + 5);
            sA[0] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 4)); sA[1] = *reinterpret_cast<const f32x4*>(sp + 8 * (q + 5));
            PIN();
            mmstage(wB, sB);
        }
        return;
    }
    for (int q = 0; q < nq; q += 4) {
        wB[0] = ldw(q + 2); wB[1] = ldw(q + 3);
        sB[0] = *reinterpret_cast<const f32x4*>(bp + 16); sB[1] = *reinterpret_cast<const f32x4*>(bp + 24);
        PIN();
        mmstage(wA, sA);
        wA[0] = ldw(q + 4); wA[1] = ldw(q + 5);
        sA[0] = *reinterpret_cast<const f32x4*>(bp + 32); sA[1] = *reinterpret_cast<const f32x4*>(bp + 40);
        bp += 32;
        PIN();
        mmstage(wB, sB);
    }
}
template <int VAR>
__device__ __forceinline__ void contract(f32x16& acc, const f32x4* __restrict__ wp, const float* sp, int nq) {
    const int last = nq - 1;
    if (VAR == 4) {
        f32x4 wA, wB, wC, sA, sB, sC;
        auto ld = [&](f32x4& w, f32x4& s, int q) {
            const int qq = q < last ? q : last;
            w = wp[(size_t)qq * 64];
            s = *reinterpret_cast<const f32x4*>(sp + 8 * qq);
        };
        auto mm = [&](const f32x4& w, const f32x4& s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma32(s[e], w[e], acc);
        };
        ld(wA, sA, 0);
        ld(wB, sB, 1);
        int q = 0;
        for (; q + 2 < nq; q += 3) {
            ld(wC, sC, q + 2);
            PIN();
            mm(wA, sA);
            ld(wA, sA, q + 3);
            PIN();
            mm(wB, sB);
            ld(wB, sB, q + 4);
            PIN();
            mm(wC, sC);
        }
        if (q < nq) mm(wA, sA);
        if (q + 1 < nq) mm(wB, sB);
        return;
    }
    f32x4 wA[2], wB[2], sA[2], sB[2];
    f32x16 acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
    auto ldstage = [&](f32x4 (&w)[2], f32x4 (&s)[2], int q0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qq = (q0 + j) < last ? (q0 + j) : last;
            w[j] = wp[(size_t)((VAR & 1) && VAR < 4 ? 0 : qq) * 64];
            s[j] = *reinterpret_cast<const f32x4*>(sp + 8 * ((VAR & 2) && VAR < 4 ? 0 : qq));
        }
    };
    auto mmstage = [&](const f32x4 (&w)[2], const f32x4 (&s)[2]) {
        if (VAR == 5) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = mfma32(s[0][e], w[0][e], acc);
                acc2 = mfma32(s[1][e], w[1][e], acc2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = mfma32(s[j][e], w[j][e], acc);
        }
    };
    ldstage(wA, sA, 0);
    for (int q = 0; q < nq; q += 4) {
        ldstage(wB, sB, q + 2);
        PIN();
        mmstage(wA, sA);
        ldstage(wA, sA, q + 4);
        PIN();
        mmstage(wB, sB);
    }
    if (VAR == 5)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void k_p3(const float* __restrict__ Wt4, const float* __restrict__ Hsrc, float* out, unsigned long long* cyc,
                                             int rp, int ldh, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 32 * ldh; i += 256) lds[i] = Hsrc[i % (32 * rp)];
    __syncthreads();
    const int fl = lane & 31, h = lane >> 5;
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wt4 + (size_t)((blockIdx.x + w) & 7) * rp * 32) + lane;
    const float* sp = lds + fl * ldh + 4 * h;
    float tot = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (VAR >= 6) {
            const int phi = (blockIdx.x + w) & 7;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt4), 0, 8 * rp * 32 * 4, 0x00020000);
            contract_b<VAR>(acc, rs, lane * 16, phi * rp * 32 * 4, sp, rp / 8);
        } else {
            contract<VAR>(acc, wp, sp, rp / 8);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
        tot += s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = tot;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <typename K>
static void run(K kern, const char* name, const float* W, const float* H, float* out, unsigned long long* cyc, int rp, int reps) {
    const int ldh = rp + 4, grid = 256;
    const size_t lds = (size_t)32 * ldh * 4 + 256;  // + slack: the unclamped variants read two k-blocks past the last row
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, W, H, out, cyc, rp, ldh, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n_mfma = (double)reps * (rp / 8) * 4;
    printf("%-72s cycles per MFMA: median %.1f  (min %.1f max %.1f)  err=%s\n", name, h[h.size() / 2] / n_mfma, h.front() / n_mfma,
           h.back() / n_mfma, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    const int rp = 256;
    std::vector<float> hw((size_t)8 * rp * 32), hh((size_t)32 * rp);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.001f * (float)(i % 977);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = 0.002f * (float)(i % 613);
    float *W, *H, *out;
    unsigned long long* cyc;
    hipMalloc(&W, hw.size() * 4);
    hipMalloc(&H, hh.size() * 4);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(H, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    for (int reps : {1, 50}) {
        printf("-- %d contraction(s) of %d MFMAs per wave\n", reps, rp / 2);
        run(k_p3<0>, "product loop (2 stages x 2 k-blocks, one chain)", W, H, out, cyc, rp, reps);
        run(k_p3<1>, "  W fragment 0 only (no global loads in the loop)", W, H, out, cyc, rp, reps);
        run(k_p3<2>, "  LDS fragment 0 only", W, H, out, cyc, rp, reps);
        run(k_p3<3>, "  neither stream", W, H, out, cyc, rp, reps);
        run(k_p3<4>, "three single-block stages, fragments two blocks ahead", W, H, out, cyc, rp, reps);
        run(k_p3<5>, "product loop, two accumulator chains", W, H, out, cyc, rp, reps);
        run(k_p3<6>, "W by buffer_load + scalar offsets, no clamps", W, H, out, cyc, rp, reps);
        run(k_p3<7>, "  + LDS fragments at immediate offsets from a moving base", W, H, out, cyc, rp, reps);
        run(k_p3<8>, "  + fully unrolled (32 k-blocks)", W, H, out, cyc, rp, reps);
    }
    return 0;
}
