// scratch_probe.hip -- what does a kernel pay per LAUNCH for using scratch memory at all (a few spilled registers)?  Two kernels that
// differ only in a 76-byte per-lane private array indexed at run time (forces scratch), 256 workgroups x 512 threads, ~50 us busy.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/prof_build/scratch_probe scripts/scratch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ void busy(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
template <bool SCR>
__global__ __launch_bounds__(512) void k(float* out, int idx, unsigned long long ticks) {
    volatile float priv[19];
    if (SCR) {
        for (int i = 0; i < 19; ++i) priv[(i + idx) % 19] = (float)(threadIdx.x + i);
    }
    busy(ticks);
    float v = 1.f;
    if (SCR) v = priv[(idx + threadIdx.x) % 19];  // run-time index: the array lives in scratch
    if (v == -1.f) out[threadIdx.x] = v;
}
int main() {
    float* out;
    CK(hipMalloc(&out, 4096));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int busy_us : {0, 50}) {
        for (int scr = 0; scr < 2; ++scr) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 200; ++i) {
                    if (scr) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, s, out, i, (unsigned long long)busy_us * 100);
                    else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, s, out, i, (unsigned long long)busy_us * 100);
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("busy %d us, scratch %d: %.2f us per launch\n", busy_us, scr, best / 200 * 1000);
        }
    }
    return 0;
}
