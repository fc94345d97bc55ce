// Dev microbenchmark: which f32 MFMA shape delivers more FLOP/s on MI355X once the chip holds its clock down
// under load?  v_mfma_f32_32x32x2_f32 (what the solver kernels are built from) against v_mfma_f32_16x16x4_f32
// (same flops per cycle per SIMD on paper: 4096 flop / 64 cycles vs 2048 flop / 32 cycles).
// Bare loops on random operands, one or two waves per SIMD, operands either held in registers or re-read from LDS
// with one ds_read_b128 per four MFMAs (the cadence of the solver's P1/P2/P3 loops).  Reports wall TFLOP/s and the
// in-kernel clock (s_memtime over s_memrealtime, median over workgroups) -- MI355X_MICROARCH.md, DVFS give-back.
// Build + run:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/mfma_shape_bench scripts/mfma_shape_bench.hip && /tmp/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp {
    unsigned long long c0, c1, r0, r1;
};

// SHAPE 32: 32x32x2, 4 accumulators (64 VGPRs), 32 MFMAs per loop trip.  SHAPE 16: 16x16x4, 16 accumulators
// (64 VGPRs), 64 MFMAs per trip.  Both trips are 131072 flop per wave.
template <int SHAPE, bool LDSOP>
__global__ __launch_bounds__(512) void k_loop(const float* __restrict__ src, float* __restrict__ out, Stamp* st,
                                               int trips) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = src[(threadIdx.x * 8 + j + blockIdx.x * 131) & 0xfffff];
        b[j] = src[(threadIdx.x * 8 + j + 7777 + blockIdx.x * 17) & 0xfffff];
    }
    unsigned long long c0 = 0, r0 = 0;
    if (lane == 0) {
        c0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 av = {a[j], a[(j + 1) & 7], a[(j + 2) & 7], a[(j + 3) & 7]};
                if (LDSOP) av = lp[((t * 8 + j) & 31) * 64];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], b[(j + c) & 7], acc[c], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[c][i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                f32x4 av = {a[j & 7], a[(j + 1) & 7], a[(j + 2) & 7], a[(j + 3) & 7]};
                if (LDSOP) av = lp[((t * 16 + j) & 31) * 64];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[(4 * j + c) & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], b[(j + c) & 7], acc[(4 * j + c) & 15], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if (lane == 0) {
        Stamp x{c0, __builtin_amdgcn_s_memtime(), r0, __builtin_amdgcn_s_memrealtime()};
        st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = x;
    }
}

template <typename K>
static void run(K kern, const char* name, int threads, const float* src, float* out, Stamp* st, int trips) {
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // hold the load for ~2 s first so that the clock the chip settles at is the one measured
    float warm = 0.f;
    while (warm < 2000.f) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, src, out, st, trips);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        warm += ms;
    }
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, src, out, st, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int nw = grid * threads / 64;
    std::vector<Stamp> h(nw);
    hipMemcpy(h.data(), st, nw * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) {
        clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);  // GHz (s_memrealtime ticks at 100 MHz)
        cyc.push_back((double)(s.c1 - s.c0));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop = (double)nw * trips * 131072.0 * reps;
    const double mfma_cycles = (double)trips * (32 * 64);  // issue cycles one wave needs per trip, either shape
    printf("%-34s %7.1f TFLOP/s  %6.3f ms/launch  clock %.3f GHz  wave cycles %.0f (MFMA issue %.0f x waves/SIMD %d)  err=%s\n",
           name, flop / (ms * 1e-3) / 1e12, ms / reps, clk[clk.size() / 2], cyc[cyc.size() / 2], mfma_cycles, threads / 256,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    std::mt19937 g(7);
    std::uniform_real_distribution<float> u(0.01f, 1.f);
    std::vector<float> h(1 << 20);
    for (auto& x : h) x = u(g);
    float *src, *out;
    Stamp* st;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&st, 256 * 8 * sizeof(Stamp));
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int trips = 4000;  // 4000 x 2048 issue cycles = 8.2 M cycles, ~4 ms per launch at one wave per SIMD
    run(k_loop<32, false>, "32x32x2  regs  1 wave/SIMD", 256, src, out, st, trips);
    run(k_loop<16, false>, "16x16x4  regs  1 wave/SIMD", 256, src, out, st, trips);
    run(k_loop<32, true>, "32x32x2  lds   1 wave/SIMD", 256, src, out, st, trips);
    run(k_loop<16, true>, "16x16x4  lds   1 wave/SIMD", 256, src, out, st, trips);
    run(k_loop<32, false>, "32x32x2  regs  2 waves/SIMD", 512, src, out, st, trips / 2);
    run(k_loop<16, false>, "16x16x4  regs  2 waves/SIMD", 512, src, out, st, trips / 2);
    run(k_loop<32, true>, "32x32x2  lds   2 waves/SIMD", 512, src, out, st, trips / 2);
    run(k_loop<16, true>, "16x16x4  lds   2 waves/SIMD", 512, src, out, st, trips / 2);
    return 0;
}
