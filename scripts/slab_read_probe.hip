// slab_read_probe.hip -- what does k_wfin's read pattern cost per round trip?  A producer kernel (256 workgroups) writes n_chunks slabs of
// rp x Fp floats, then a consumer (one workgroup per column, as k_wfin) has thread (g, e) read the column's piece e of 16 chunks --
//   strided:     the engine's layout [chunk][rp][Fp]: sixteen 16-byte loads 4*rp*Fp bytes apart (sixteen pages)
//   contiguous:  a column-major layout [k][chunk][Fp]: the same sixteen loads Fp*4 bytes apart (one or two pages)
// and stamps (100 MHz counter) the issue and the arrival of the batch; every launch follows a fresh producer launch (cold lines, other
// kernels' translations in the TLBs), like the iteration loop.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/prof_build/slab_read_probe scripts/slab_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void producer(float* slabs, size_t per_wg) {
    f32x4* p = reinterpret_cast<f32x4*>(slabs + (size_t)blockIdx.x * per_wg);
    for (size_t i = threadIdx.x; i < per_wg / 4; i += 512) p[i] = f32x4{1.f, 2.f, 3.f, (float)blockIdx.x};
}

// thread t < 8 * nE: chunk group g = t / nE (16 chunks each: n_chunks = 128), piece e = t % nE of column k = blockIdx.x
template <bool CONTIG>
__global__ __launch_bounds__(768) void consumer(const float* slabs, int rp, int Fp, int n_chunks, unsigned long long* stamps, float* sink) {
    const int k = blockIdx.x, tid = threadIdx.x, nE = Fp / 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    f32x4 x[16];
    float acc = 0.f;
    unsigned long long t1 = t0, t2 = t0;
    if (tid < 8 * nE) {
        const int g = tid / nE, e = tid - g * nE;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = g * 16 + j;
            const size_t off = CONTIG ? ((size_t)k * n_chunks + c) * Fp + 4 * e : ((size_t)c * rp + k) * Fp + 4 * e;
            x[j] = *reinterpret_cast<const f32x4*>(slabs + off);
        }
        t1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += x[j][0] + x[j][1] + x[j][2] + x[j][3];
        asm volatile("" : "+v"(acc));
        t2 = __builtin_amdgcn_s_memrealtime();
    }
    if (acc == -1.f) sink[tid] = acc;
    if (tid == 0) {
        stamps[k * 4 + 0] = t0;
        stamps[k * 4 + 1] = t1;
        stamps[k * 4 + 2] = t2;
    }
}

int main() {
    struct Shape { const char* name; int rp, Fp, r; } shapes[] = {{"Mel 64 x r=100 (rp 128, Fp 64)", 128, 64, 100}, {"C2 257 x r=256 (rp 256, Fp 288)", 256, 288, 256}};
    const int n_chunks = 128;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (const Shape& sh : shapes) {
        const size_t n = (size_t)n_chunks * sh.rp * sh.Fp;
        float *slabs, *sink;
        unsigned long long* st;
        CK(hipMalloc(&slabs, n * 4));
        CK(hipMalloc(&sink, 4096));
        CK(hipMalloc(&st, 1024 * 4 * 8));
        std::vector<unsigned long long> h(1024 * 4);
        for (int contig = 0; contig < 2; ++contig) {
            std::vector<double> issue, wait, total;
            for (int rep = 0; rep < 40; ++rep) {
                hipLaunchKernelGGL(producer, dim3(256), dim3(512), 0, s, slabs, n / 256 / 4 * 4);
                if (contig) hipLaunchKernelGGL(consumer<true>, dim3(sh.r), dim3(768), 0, s, slabs, sh.rp, sh.Fp, n_chunks, st, sink);
                else hipLaunchKernelGGL(consumer<false>, dim3(sh.r), dim3(768), 0, s, slabs, sh.rp, sh.Fp, n_chunks, st, sink);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h.data(), st, sh.r * 4 * 8, hipMemcpyDeviceToHost));
                if (rep < 5) continue;
                unsigned long long first = ~0ull, last = 0;
                double w = 0, is = 0;
                for (int k = 0; k < sh.r; ++k) {
                    first = std::min(first, h[k * 4]);
                    last = std::max(last, h[k * 4 + 2]);
                    is += (double)(h[k * 4 + 1] - h[k * 4]);
                    w += (double)(h[k * 4 + 2] - h[k * 4 + 1]);
                }
                issue.push_back(is / sh.r * 0.01);
                wait.push_back(w / sh.r * 0.01);
                total.push_back((double)(last - first) * 0.01);
            }
            std::sort(issue.begin(), issue.end()); std::sort(wait.begin(), wait.end()); std::sort(total.begin(), total.end());
            printf("%s, %s: issue %.2f us, wait for the batch %.2f us, first start -> last arrival %.2f us (medians of %zu launches)\n", sh.name,
                   contig ? "column-major [k][chunk][Fp]" : "engine layout [chunk][rp][Fp]", issue[issue.size() / 2], wait[wait.size() / 2],
                   total[total.size() / 2], issue.size());
        }
        hipFree(slabs); hipFree(sink); hipFree(st);
    }
    return 0;
}
