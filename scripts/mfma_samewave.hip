// Dev microbenchmark (round 2): does the SAME wave's other work hide under its own v_mfma_f32_32x32x2_f32?
// scripts/mfma_valu_overlap.hip showed that a neighbour wave gets nothing issued beside a dense MFMA loop and that an
// s_nop behind an MFMA only starts after the MFMA's 64 cycles.  This one puts K independent instructions of a kind
// between consecutive (independent-accumulator) MFMAs of one wave per SIMD and reports cycles per MFMA: 64 means the
// filler was free, 64 + K * issue-cost means it was serialised.
// Build + run:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/msw scripts/mfma_samewave.hip && /tmp/msw
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

// KIND: 0 none, 1 v_fma_f32, 2 v_rcp_f32, 3 ds_read_b128, 4 global_load_dwordx4 (L2 resident), 5 v_mov/v_add integer (address-like),
//       6 s_add (scalar ALU)
// CH = number of independent accumulator chains the MFMAs rotate over (1 = every MFMA depends on the one before it)
template <int KIND, int K, int WAVES, int CH = 4>
__global__ __launch_bounds__(WAVES * 64) void k_same(const float* __restrict__ src, float* __restrict__ out, Stamp* st, int trips) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    float a[8], b[8], x[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = src[(threadIdx.x * 8 + j + blockIdx.x * 131) & 0xfffff];
        b[j] = src[(threadIdx.x * 8 + j + 7777 + blockIdx.x * 17) & 0xfffff];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = src[(threadIdx.x * 16 + i) & 0xfffff] * 1e-3f + 1.f;
    int xi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xi[i] = lane + i;
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    const f32x4* gp = reinterpret_cast<const f32x4*>(src) + threadIdx.x;
    f32x4 sacc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int sc = blockIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + c) & 7], b[(j + 2 * c) & 7], acc[c % CH], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int s = (j * 4 + c) * K + k;
                    if (KIND == 1) x[s & 15] = __builtin_fmaf(x[s & 15], 0.999f, 1e-3f);
                    if (KIND == 2) x[s & 15] = __builtin_amdgcn_rcpf(x[s & 15]);
                    if (KIND == 3) sacc[s & 3] += lp[((t + s) & 31) * 64];
                    if (KIND == 4) sacc[s & 3] += gp[((t + s) & 15) * 1024];
                    if (KIND == 5) xi[s & 15] = (xi[s & 15] + 12345) ^ s;
                    if (KIND == 6) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sc));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = sacc[0][0] + sacc[1][1] + sacc[2][2] + sacc[3][3] + (float)sc;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + (float)xi[i];
    out[blockIdx.x * (WAVES * 64) + threadIdx.x] = s;
    if (lane == 0) {
        Stamp q{c0, c1, r0, r1};
        st[blockIdx.x * WAVES + w] = q;
    }
}

template <typename Kn>
static void run(Kn kern, int waves, const char* name, const float* src, float* out, Stamp* st, int trips) {
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float warm = 0.f;
    while (warm < 600.f) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, src, out, st, trips);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        warm += ms;
    }
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, src, out, st, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid * waves);
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) {
        clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
        cyc.push_back((double)(s.c1 - s.c0));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop = (double)grid * waves * trips * 131072.0 * reps;
    // cycles per MFMA per SIMD: waves/4 waves share a SIMD, so a wave's span covers (waves/4) x its own MFMAs
    printf("%-64s %7.1f TFLOP/s  clock %.3f GHz  SIMD cycles per MFMA %.1f  err=%s\n", name, flop / (ms * 1e-3) / 1e12,
           clk[clk.size() / 2], cyc[cyc.size() / 2] / ((double)trips * 32 * (waves / 4)), hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

#define RUN(KIND, K, WAVES, NAME) run(k_same<KIND, K, WAVES>, WAVES, NAME, src, out, st, trips)
#define RUNC(KIND, K, WAVES, CH, NAME) run(k_same<KIND, K, WAVES, CH>, WAVES, NAME, src, out, st, trips)
int main() {
    std::mt19937 g(7);
    std::uniform_real_distribution<float> u(0.01f, 1.f);
    std::vector<float> h(1 << 20);
    for (auto& x : h) x = u(g);
    float *src, *out;
    Stamp* st;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&st, 256 * 16 * sizeof(Stamp));
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int trips = 1500;
    RUN(0, 0, 4, "1 wave/SIMD: bare MFMA loop");
    RUN(1, 4, 4, "1 wave/SIMD: + 4 v_fma_f32 after each MFMA");
    RUN(1, 8, 4, "1 wave/SIMD: + 8 v_fma_f32");
    RUN(1, 12, 4, "1 wave/SIMD: + 12 v_fma_f32");
    RUN(1, 16, 4, "1 wave/SIMD: + 16 v_fma_f32");
    RUN(2, 1, 4, "1 wave/SIMD: + 1 v_rcp_f32");
    RUN(2, 2, 4, "1 wave/SIMD: + 2 v_rcp_f32");
    RUN(2, 4, 4, "1 wave/SIMD: + 4 v_rcp_f32");
    RUN(3, 1, 4, "1 wave/SIMD: + 1 ds_read_b128");
    RUN(3, 2, 4, "1 wave/SIMD: + 2 ds_read_b128");
    RUN(3, 4, 4, "1 wave/SIMD: + 4 ds_read_b128");
    RUN(4, 1, 4, "1 wave/SIMD: + 1 global_load_dwordx4");
    RUN(4, 2, 4, "1 wave/SIMD: + 2 global_load_dwordx4");
    RUN(5, 8, 4, "1 wave/SIMD: + 8 integer VALU (v_add / v_xor)");
    RUN(6, 8, 4, "1 wave/SIMD: + 8 s_add_u32");
    RUN(6, 16, 4, "1 wave/SIMD: + 16 s_add_u32");
    // two waves per SIMD, both running the same interleaved loop
    RUN(0, 0, 8, "2 waves/SIMD: bare MFMA loops");
    RUN(1, 8, 8, "2 waves/SIMD: + 8 v_fma_f32 after each MFMA");
    RUN(1, 16, 8, "2 waves/SIMD: + 16 v_fma_f32");
    RUN(2, 2, 8, "2 waves/SIMD: + 2 v_rcp_f32");
    RUN(3, 2, 8, "2 waves/SIMD: + 2 ds_read_b128");
    RUN(6, 16, 8, "2 waves/SIMD: + 16 s_add_u32");
    // dependent chains: every MFMA accumulates into the result of the previous one (CH = 1), or of the one before that (CH = 2)
    RUNC(0, 0, 4, 1, "1 wave/SIMD: ONE accumulator chain (all MFMAs dependent)");
    RUNC(0, 0, 4, 2, "1 wave/SIMD: two chains");
    RUNC(6, 8, 4, 1, "1 wave/SIMD: one chain + 8 s_add_u32");
    RUNC(1, 4, 4, 1, "1 wave/SIMD: one chain + 4 v_fma_f32");
    RUNC(0, 0, 8, 1, "2 waves/SIMD: one chain each");
    RUNC(0, 0, 12, 1, "3 waves/SIMD: one chain each");
    return 0;
}
