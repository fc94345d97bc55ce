// overlap_probe.hip -- can a second kernel (stream B) fill the tail of a first one (stream A) on gfx950, gated by a
// stream memory wait on "every workgroup of A has started"?  (round 6: the H step's tail under the W statistics' ramp)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe scripts/overlap_probe.hip && /tmp/overlap_probe
// Kernel A: G workgroups x 768 threads, 135 KB of LDS (one per CU, as k_hstep_rp), each busy for 200 us + a spread of 0..30 us.
// Kernel B: G workgroups x 512 threads, 100 KB of LDS (cannot share a CU with A), each waits for A's workgroup of the same index
// to publish (sc1 store / sc1 load), checks a payload A wrote with sc1 stores, then is busy for 200 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz
__device__ __forceinline__ void busy(unsigned long long ticks) {
    const unsigned long long t0 = now();
    while (now() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(768) void kA(unsigned* started, unsigned* sig, unsigned epoch, unsigned* progress, float* payload,
                                          unsigned long long* tA, int spread) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, G = gridDim.x;
    if (threadIdx.x == 0) {
        tA[2 * b] = now();
        const unsigned old = __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)G * epoch - 1u) __hip_atomic_store(sig, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    lds[threadIdx.x] = (float)b;
    busy(20000ull + (unsigned long long)((b * 37) % 16) * spread);
    // payload: 768 floats per workgroup, write-through (sc1) stores, then the progress word
    __builtin_nontemporal_store(0.f, &lds[0]);
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(payload + (size_t)b * 768, 0, 768 * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)(epoch * 1000 + b)), rs, threadIdx.x * 4, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(progress + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tA[2 * b + 1] = now();
    }
}

__global__ __launch_bounds__(512) void kB(const unsigned* progress, unsigned epoch, const float* payload, unsigned long long* tB,
                                          int* bad, int wait_on) {
    extern __shared__ float lds[];
    const int c = blockIdx.x;
    // the workgroup waits for A's workgroup (c * 7) % G: not its own CU's predecessor
    const int src = (c * 7 + 3) % gridDim.x;
    if (threadIdx.x == 0) {
        tB[3 * c] = now();
        int spin = 0;
        while (wait_on && __hip_atomic_load(progress + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++spin > (1 << 17)) { atomicAdd(bad, 1000000); break; }
        }
        tB[3 * c + 1] = now();
    }
    __syncthreads();
    if (wait_on) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(payload) + (size_t)src * 768, 0, 768 * 4, 0x00020000);
        const float x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, 0, 16));
        if (x != (float)(epoch * 1000 + src)) atomicAdd(bad, 1);
    }
    lds[threadIdx.x] = 1.f;
    busy(20000ull);
    if (threadIdx.x == 0) tB[3 * c + 2] = now();
}

int main() {
    int dev = 0, can = -1, ncu = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    printf("CanUseStreamWaitValue = %d, CUs = %d\n", can, ncu);
    const int G = ncu;
    hipStream_t sA, sB;
    CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    unsigned *started, *progress, *sig = nullptr, *sig_dev = nullptr;
    float* payload;
    unsigned long long *tA, *tB;
    int* bad;
    CK(hipMalloc(&started, 4));
    CK(hipMalloc(&progress, G * 4));
    CK(hipMalloc(&payload, (size_t)G * 768 * 4));
    CK(hipMalloc(&tA, G * 16));
    CK(hipMalloc(&tB, G * 24));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&sig_dev, 8));
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(es));
    CK(hipFuncSetAttribute((const void*)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 135 * 1024));
    CK(hipFuncSetAttribute((const void*)kB, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1, eb;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    for (int mode = 0; mode < 4; ++mode) {
        // 0: serial on one stream.  1: gated by hipStreamWaitValue32 on signal memory.  2: the same on plain device memory.
        // 3: B on its own stream with NO gate at all, only behind the previous round (what happens without one)
        unsigned* s = mode == 2 ? sig_dev : sig;
        if (mode == 1 && es != hipSuccess) continue;
        CK(hipMemset(started, 0, 4));
        CK(hipMemset(progress, 0, G * 4));
        CK(hipMemset(bad, 0, 4));
        if (s) CK(hipMemset(s, 0, 4));
        CK(hipDeviceSynchronize());
        const int R = 12;
        bool ok = true;
        float best = 1e9f;
        std::vector<unsigned long long> hA(2 * G), hB(3 * G);
        for (int rep = 0; rep < 3 && ok; ++rep) {
            CK(hipEventRecord(e0, sA));
            for (int i = 0; i < R; ++i) {
                const unsigned epoch = (unsigned)(rep * R + i + 1);
                hipLaunchKernelGGL(kA, dim3(G), dim3(768), 135 * 1024, sA, started, s ? s : sig_dev, epoch, progress, payload, tA, 200);
                if (mode == 0) {
                    hipLaunchKernelGGL(kB, dim3(G), dim3(512), 100 * 1024, sA, progress, epoch, payload, tB, bad, 1);
                } else {
                    if (mode != 3) {
                        hipError_t ew = hipStreamWaitValue32(sB, s, epoch, hipStreamWaitValueGte, 0xffffffffu);
                        if (ew != hipSuccess) {
                            printf("mode %d: hipStreamWaitValue32: %s\n", mode, hipGetErrorString(ew));
                            ok = false;
                            break;
                        }
                    }
                    hipLaunchKernelGGL(kB, dim3(G), dim3(512), 100 * 1024, sB, progress, epoch, payload, tB, bad, 1);
                    CK(hipEventRecord(eb, sB));
                    CK(hipStreamWaitEvent(sA, eb, 0));
                }
            }
            CK(hipEventRecord(e1, sA));
            hipError_t esy = hipEventSynchronize(e1);
            if (esy != hipSuccess) { printf("mode %d: sync: %s\n", mode, hipGetErrorString(esy)); ok = false; break; }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / R);
        }
        if (!ok) { (void)hipGetLastError(); continue; }
        CK(hipDeviceSynchronize());
        int hbad = 0;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hA.data(), tA, G * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hB.data(), tB, G * 24, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a1min = ~0ull, a1max = 0, b0min = ~0ull, b0max = 0, b2max = 0, waitsum = 0;
        for (int i = 0; i < G; ++i) {
            a0 = std::min(a0, hA[2 * i]);
            a1min = std::min(a1min, hA[2 * i + 1]);
            a1max = std::max(a1max, hA[2 * i + 1]);
            b0min = std::min(b0min, hB[3 * i]);
            b0max = std::max(b0max, hB[3 * i]);
            b2max = std::max(b2max, hB[3 * i + 2]);
            waitsum += hB[3 * i + 1] - hB[3 * i];
        }
        printf("mode %d: %.1f us per A+B round | last round (us from A's first start): A ends %.1f..%.1f, B starts %.1f..%.1f, B ends %.1f, "
               "mean wait inside B %.1f us, payload mismatches %d\n",
               mode, best * 1000.f, (a1min - a0) / 100.0, (a1max - a0) / 100.0, (double)(b0min - a0) / 100.0, (double)(b0max - a0) / 100.0,
               (b2max - a0) / 100.0, waitsum / 100.0 / G, hbad);
    }
    return 0;
}
