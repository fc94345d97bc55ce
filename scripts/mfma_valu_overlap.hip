// Dev microbenchmark (round 2): how much does work on a NEIGHBOUR wave of the same SIMD cost an MFMA-issuing wave?
// k_hstep_rp's tile period equals MFMA time + epilogue time of both waves of a SIMD whatever the schedule
// (profiles/r02_experiments.md); this isolates the effect.  One 512-thread workgroup per CU: waves 0-3 (one per SIMD)
// run a bare v_mfma_f32_32x32x2_f32 loop on random register operands (4 independent accumulators), waves 4-7 (their
// SIMD partners) run one of: nothing, a dense v_fma_f32 loop, a transcendental (v_rcp/v_log) loop, an LDS
// ds_read_b128 loop, an L2-resident global_load loop, or an s_sleep polling loop -- until the MFMA waves raise a flag.
// Reported: TFLOP/s of the MFMA waves and the in-kernel clock.
// Build + run:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/mvo scripts/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Stamp { unsigned long long c0, c1, r0, r1; };

// CH: independent accumulator chains of the MFMA wave (1 = every MFMA waits for the result of the one before it)
template <int MODE, int PRIO = 0, bool MF = true, int PAD = 0, int CH = 4>
__global__ __launch_bounds__(512) void k_pair(const float* __restrict__ src, float* __restrict__ out, Stamp* st, int trips,
                                               unsigned long long* polls) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    __shared__ unsigned done;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    if (threadIdx.x == 0) done = 0u;
    __syncthreads();
    if (w < 4) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = src[(threadIdx.x * 8 + j + blockIdx.x * 131) & 0xfffff];
            b[j] = src[(threadIdx.x * 8 + j + 7777 + blockIdx.x * 17) & 0xfffff];
        }
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        if (MF) {
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[c % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + c) & 7], b[(j + 2 * c) & 7], acc[c % CH], 0, 0, 0);
                        // PAD: give the issue port away while the matrix pipe works on the MFMA just issued (64 cycles):
                        // a wave that sits on a not-yet-issuable MFMA keeps every other wave of the SIMD from issuing
                        if (PAD >= 1) asm volatile("s_nop 15");
                        if (PAD >= 2) asm volatile("s_nop 15");
                        if (PAD >= 3) asm volatile("s_nop 15");
                        if (PAD == 4) __builtin_amdgcn_s_sleep(0);
                    }
            }
        } else {  // no MFMAs: just hold the neighbour in its loop for the same time (the MFMA loop takes trips * 2048 cycles)
            const unsigned long long tend = c0 + (unsigned long long)trips * 2048ull;
            while (__builtin_amdgcn_s_memtime() < tend) __builtin_amdgcn_s_sleep(8);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[c][i];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (lane == 0) {
            Stamp x{c0, __builtin_amdgcn_s_memtime(), r0, __builtin_amdgcn_s_memrealtime()};
            st[blockIdx.x * 4 + w] = x;
            __hip_atomic_fetch_add(&done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        if (MODE == 0) return;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = src[(threadIdx.x * 16 + i) & 0xfffff] * 1e-3f + 1.f;
        const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
        const f32x4* gp = reinterpret_cast<const f32x4*>(src) + threadIdx.x + (size_t)blockIdx.x * 512;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
        int it = 0;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u) {
            if (MODE == 1) {  // dense VALU: 64 independent FMAs per poll
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], 0.999f, 1e-3f);
            } else if (MODE == 2) {  // transcendental unit: 16 rcp + 16 log2 per poll
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]) + 1.5f;
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_logf(x[i]) + 2.f;
            } else if (MODE == 3) {  // LDS: 8 ds_read_b128 per poll
#pragma unroll
                for (int r = 0; r < 8; ++r) sacc += lp[((it * 8 + r) & 31) * 64];
            } else if (MODE == 4) {  // L2-resident global loads: 4 x 16 B per lane per poll
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc += gp[((it * 4 + r) & 15) * 1024];
            } else {  // MODE 5: polling with s_sleep, as the kernels' bounded waits do
                __builtin_amdgcn_s_sleep(1);
            }
            ++it;
        }
        float s = sacc[0] + sacc[1] + sacc[2] + sacc[3];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        if (s == 123.456f) out[threadIdx.x] = s;  // keep the work alive
        if (lane == 0 && blockIdx.x == 0 && w == 4) *polls = (unsigned long long)it;
    }
}

static unsigned long long* g_polls;
template <typename K>
static void run(K kern, const char* name, const float* src, float* out, Stamp* st, int trips) {
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float warm = 0.f;
    while (warm < 1500.f) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, src, out, st, trips, g_polls);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        warm += ms;
    }
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, src, out, st, trips, g_polls);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(grid * 4);
    hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (auto& s : h) {
        clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
        cyc.push_back((double)(s.c1 - s.c0));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop = (double)grid * 4 * trips * 131072.0 * reps;
    unsigned long long polls = 0;
    hipMemcpy(&polls, g_polls, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f TFLOP/s  %6.3f ms/launch  clock %.3f GHz  cycles per MFMA %.1f  neighbour loop trips %llu  err=%s\n", name,
           flop / (ms * 1e-3) / 1e12, ms / reps, clk[clk.size() / 2], cyc[cyc.size() / 2] / ((double)trips * 32), polls,
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
    hipMalloc(&g_polls, 8);
    hipMemset(g_polls, 0, 8);
    const int trips = 4000;
    run(k_pair<0>, "MFMA wave alone on its SIMD", src, out, st, trips);
    run(k_pair<5>, "neighbour: s_sleep polling loop", src, out, st, trips);
    run(k_pair<1>, "neighbour: dense v_fma_f32 loop", src, out, st, trips);
    run(k_pair<2>, "neighbour: v_rcp_f32 / v_log_f32 loop", src, out, st, trips);
    run(k_pair<3>, "neighbour: LDS ds_read_b128 loop", src, out, st, trips);
    run(k_pair<4>, "neighbour: L2-resident global_load loop", src, out, st, trips);
    // how fast does the NEIGHBOUR get through its loop beside an MFMA wave (trips in the same wall time)?
    run(k_pair<1, 0, false>, "v_fma loop, partner idle (no MFMAs)", src, out, st, trips);
    run(k_pair<1, 0, true>, "v_fma loop beside the MFMA wave", src, out, st, trips);
    run(k_pair<1, 3, true>, "v_fma loop beside the MFMA wave, s_setprio 3", src, out, st, trips);
    run(k_pair<2, 0, false>, "rcp/log loop, partner idle", src, out, st, trips);
    run(k_pair<2, 0, true>, "rcp/log loop beside the MFMA wave", src, out, st, trips);
    run(k_pair<2, 3, true>, "rcp/log loop beside the MFMA wave, s_setprio 3", src, out, st, trips);
    run(k_pair<3, 0, false>, "ds_read_b128 loop, partner idle", src, out, st, trips);
    run(k_pair<3, 0, true>, "ds_read_b128 loop beside the MFMA wave", src, out, st, trips);
    run(k_pair<3, 3, true>, "ds_read_b128 loop beside the MFMA wave, s_setprio 3", src, out, st, trips);
    // the MFMA wave yields the issue port between its MFMAs (s_nop 15 = 16 idle cycles each)
    run(k_pair<1, 0, true, 1>, "v_fma neighbour; MFMA wave: 1 x s_nop 15 after each MFMA", src, out, st, trips);
    run(k_pair<1, 0, true, 2>, "v_fma neighbour; MFMA wave: 2 x s_nop 15", src, out, st, trips);
    run(k_pair<1, 0, true, 3>, "v_fma neighbour; MFMA wave: 3 x s_nop 15", src, out, st, trips);
    run(k_pair<1, 0, true, 4>, "v_fma neighbour; MFMA wave: 3 x s_nop 15 + s_sleep 0", src, out, st, trips);
    run(k_pair<2, 0, true, 3>, "rcp/log neighbour; MFMA wave: 3 x s_nop 15", src, out, st, trips);
    run(k_pair<3, 0, true, 3>, "ds_read_b128 neighbour; MFMA wave: 3 x s_nop 15", src, out, st, trips);
    run(k_pair<5, 0, true, 3>, "s_sleep polling neighbour; MFMA wave: 3 x s_nop 15", src, out, st, trips);
    // the MFMA wave runs ONE dependent chain: its next MFMA is not issuable until the previous one has finished -- does the
    // neighbour get the issue slots then, and what does that cost the chain?
    run(k_pair<0, 0, true, 0, 1>, "one dependent chain, alone", src, out, st, trips);
    run(k_pair<1, 0, true, 0, 1>, "one dependent chain; v_fma neighbour", src, out, st, trips);
    run(k_pair<2, 0, true, 0, 1>, "one dependent chain; rcp/log neighbour", src, out, st, trips);
    run(k_pair<3, 0, true, 0, 1>, "one dependent chain; ds_read_b128 neighbour", src, out, st, trips);
    run(k_pair<4, 0, true, 0, 1>, "one dependent chain; global_load neighbour", src, out, st, trips);
    run(k_pair<5, 0, true, 0, 1>, "one dependent chain; s_sleep polling neighbour", src, out, st, trips);
    run(k_pair<1, 0, true, 0, 2>, "two chains; v_fma neighbour", src, out, st, trips);
    return 0;
}
