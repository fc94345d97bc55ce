// Diagnostic (not part of the product): how fast can independent waves stream PRIVATE 32-frame tiles (H: 32 x rp floats,
// V: 32 x F floats, both contiguous per tile) from HBM into registers, by waves per CU and by access pattern?
//   pattern 0: operand layout -- lane (h, t) reads 16 bytes at row t, column quad 2q + h   (what an MFMA B operand wants)
//   pattern 1: linear         -- lane l reads 16 bytes at l * 16 + i * 1024               (fully coalesced)
// hipcc -O3 --offload-arch=gfx950 scripts/tile_stream_probe.hip -o scripts/prof_build/tile_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ void k_probe(const float* __restrict__ H, const float* __restrict__ V, int rp, int F, int n_tiles, float* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int gw = blockIdx.x * wpb + wv, nw = gridDim.x * wpb;
    const int t = lane & 31, h = lane >> 5;
    f32x4 acc = {0, 0, 0, 0};
    for (int tile = gw; tile < n_tiles; tile += nw) {
        const float* hp = H + (size_t)tile * 32 * rp;
        const float* vp = V + (size_t)tile * 32 * F;
        f32x4 x[24];
        if (PAT == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = (8 * q + 4 * h < rp) ? *reinterpret_cast<const f32x4*>(hp + t * rp + 8 * q + 4 * h) : acc;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[16 + q] = (8 * q + 4 * h < F) ? *reinterpret_cast<const f32x4*>(vp + t * F + 8 * q + 4 * h) : acc;
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = (q * 256 + lane * 4 < 32 * rp) ? *reinterpret_cast<const f32x4*>(hp + q * 256 + lane * 4) : acc;
#pragma unroll
            for (int q = 0; q < 8; ++q) x[16 + q] = (q * 256 + lane * 4 < 32 * F) ? *reinterpret_cast<const f32x4*>(vp + q * 256 + lane * 4) : acc;
        }
#pragma unroll
        for (int q = 0; q < 24; ++q) acc += x[q];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[gw] = acc[0];
}
int main() {
    const int rp = 128, F = 64, T = 288000, n_tiles = T / 32;
    float *H, *V, *out;
    hipMalloc(&H, (size_t)T * rp * 4); hipMalloc(&V, (size_t)T * F * 4); hipMalloc(&out, 1 << 20);
    hipMemset(H, 0, (size_t)T * rp * 4); hipMemset(V, 0, (size_t)T * F * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)T * (rp + F) * 4;
    for (int pat = 0; pat < 2; ++pat)
        for (int wpc : {1, 2, 4, 8, 16}) {
            const int wpb = wpc >= 4 ? 4 : wpc, blocks = 256 * (wpc / wpb);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64 * wpb), 0, 0, H, V, rp, F, n_tiles, out);
                else hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(64 * wpb), 0, 0, H, V, rp, F, n_tiles, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("pattern %d, %2d waves/CU: %.1f us, %.2f TB/s, %.2f us per tile per wave\n", pat, wpc, ms * 1e3, bytes / ms / 1e9,
                                     ms * 1e3 / ((double)n_tiles / (256.0 * wpc)));
            }
        }
    return 0;
}
