// Probe of v_mfma_f32_4x4x1_16b_f32's operand layout on gfx950 (hipcc --offload-arch=gfx950 scripts/mfma_4x4x1_probe.hip -o /tmp/p && /tmp/p).
// Hypothesis (CDNA3 ISA, "4x4x1, 16 blocks"): block b = lane / 4;  A[i] is the a-operand of lane 4b + i, B[j] the b-operand of
// lane 4b + j, and D[i][j] lands in VGPR i of lane 4b + j:   d[l][i] = a[4 (l / 4) + i] * b[l]  (+ c).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d, unsigned long long* cyc) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
    // throughput: a dependent chain of 256 and 4 independent chains
    f32x4 c0 = c, c1 = c, c2 = c, c3 = c;
    const float av = a[l], bv = b[l];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c3, 0, 0, 0);
    }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    d[256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float ha[64], hb[64], hd[512];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
    float *da, *db, *dd; unsigned long long* dc;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 2048); hipMalloc(&dc, 16);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, dc);
    unsigned long long hc[2];
    hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float ref = ha[4 * (l / 4) + i] * hb[l];
            if (hd[l * 4 + i] != ref) { if (bad < 8) printf("lane %d vgpr %d: got %g, hypothesis %g\n", l, i, hd[l * 4 + i], ref); ++bad; }
        }
    printf("layout hypothesis d[l][i] = a[4(l/4)+i] * b[l]: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
    printf("cycles per 4x4x1 MFMA: dependent chain %.1f, four independent chains %.1f (s_memtime ticks, 100 MHz?: raw %llu %llu for 256 each)\n",
           hc[0] / 256.0, hc[1] / 256.0, hc[0], hc[1]);
    return bad != 0;
}
