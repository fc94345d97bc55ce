// Standalone microbenchmark of k_hsolve_frame (256 concurrent frames, 100 fixed iterations): us per iteration of one workgroup.
// Build + run:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ise_snmf_nat_amd/csrc -o /tmp/frame_bench scripts/frame_bench.hip && /tmp/frame_bench
#include "snmf_kernels.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace snmf;
#ifndef KERN
#define KERN k_hsolve_frame
#endif
int main() {
    const int F = 513, r = 200, rp = 224, Fp = 516, n = 256, iters = 100;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(0.01f, 1.f);
    std::vector<float> W((size_t)rp * Fp, 0.f), V((size_t)n * Fp, 0.f), H((size_t)n * rp, 0.f), wx(rp, 0.f), dphv(rp, 1.f), lamk(rp, 0.f), cs(rp, 0.f);
    for (int k = 0; k < r; ++k) { double s = 0; for (int f = 0; f < F; ++f) { W[(size_t)k * Fp + f] = u(g); s += W[(size_t)k * Fp + f]; } cs[k] = s; dphv[k] = s + 5.f; lamk[k] = 5.f; wx[k] = W[(size_t)k * Fp + 512]; }
    for (int t = 0; t < n; ++t) { for (int f = 0; f < F; ++f) V[(size_t)t * Fp + f] = 50.f * u(g); for (int k = 0; k < r; ++k) H[(size_t)t * rp + k] = u(g); }
    float *dW, *dV, *dH, *dwx, *ddp, *dlk, *dcs; double *ddiv, *dcost; DevState* dst;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dV, V.size() * 4); hipMalloc(&dH, H.size() * 4); hipMalloc(&dwx, rp * 4); hipMalloc(&ddp, rp * 4); hipMalloc(&dlk, rp * 4); hipMalloc(&dcs, rp * 4);
    hipMalloc(&ddiv, (size_t)n * iters * 8); hipMalloc(&dcost, (size_t)n * iters * 8); hipMalloc(&dst, n * sizeof(DevState));
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwx, wx.data(), rp * 4, hipMemcpyHostToDevice); hipMemcpy(ddp, dphv.data(), rp * 4, hipMemcpyHostToDevice); hipMemcpy(dlk, lamk.data(), rp * 4, hipMemcpyHostToDevice); hipMemcpy(dcs, cs.data(), rp * 4, hipMemcpyHostToDevice);
    StepArgs a{}; a.V = dV; a.Hin = dH; a.Hout = dH; a.dphv = ddp; a.colsum = dcs; a.lamk = dlk; a.wx = dwx; a.F = F; a.T = 1; a.Fp = Fp; a.rp = rp; a.Fm = 512; a.xr = 1; a.beta = 1.f;
    SmallArgs sa{}; sa.max_iter = iters; sa.cost_check = 1; sa.conv_eps = 0.0; sa.divh = ddiv; sa.costh = dcost; sa.st = dst; sa.tps = 1;
    const size_t lds = (size_t)(32 + 4 * 200 + 3 * 516 + 8 * 512 + 64 * 201) * 4 + 4096;
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(n), dim3(512), lds, 0, a, sa, (const float*)dW);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) { double c; hipMemcpy(&c, dcost + iters - 1, 8, hipMemcpyDeviceToHost); printf("%-28s %.3f us/iteration  (cost[99]=%.6e) err=%s\n", name, ms * 1e3 / iters, c, hipGetErrorString(hipGetLastError())); }
        }
    };
    run(KERN<8, 25, BM_KL, true>, "obj");
    run(KERN<8, 25, BM_KL, false>, "no obj");
    return 0;
}
