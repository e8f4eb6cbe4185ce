// Microbenchmark: what does the access pattern of the short-K epilogue (one read stream, two write streams, 128 x 128 fp32 tiles =
// 128 row segments of 512 B, rows 4 KB apart, persistent workgroups walking the 8 column tiles of a row panel) cost against the same
// bytes moved linearly?   hipcc -O3 --offload-arch=gfx950 tools/micro/tile_stream.hip -o /tmp/ts && /tmp/ts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_linear(const f32x4* __restrict__ x, f32x4* __restrict__ y1, f32x4* __restrict__ y2, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = x[i];
        y1[i] = v * 1.5f;
        y2[i] = v * 0.5f;
    }
}
// tile t of a persistent workgroup: row panel t / 8, column tile t % 8 (the 8 column tiles of a panel are consecutive)
template <int WRITES>
__global__ void __launch_bounds__(256) k_tiled(const float* __restrict__ x, float* __restrict__ y1, float* __restrict__ y2, int B, int D) {
    const int ntile = (B / 128) * (D / 128);
    const int r = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;      // 8 rows x 32 float4 per pass
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long base = (long)(t / (D / 128)) * 128 * D + (long)(t % (D / 128)) * 128;
        f32x4 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = *reinterpret_cast<const f32x4*>(x + base + (long)(p * 8 + r) * D + c4);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            *reinterpret_cast<f32x4*>(y1 + base + (long)(p * 8 + r) * D + c4) = v[p] * 1.5f;
            if (WRITES > 1) *reinterpret_cast<f32x4*>(y2 + base + (long)(p * 8 + r) * D + c4) = v[p] * 0.5f;
        }
    }
}

// the epilogue's own pattern: wave w owns the 64 x 64 quadrant (w >> 1, w & 1) of the tile and moves 32 x 32 sub-tiles, one instruction
// = 8 rows x 128 B (lane >> 3 = row, (lane & 7) * 16 B)
__global__ void __launch_bounds__(256) k_quad(const float* __restrict__ x, float* __restrict__ y1, float* __restrict__ y2, int B, int D) {
    const int ntile = (B / 128) * (D / 128);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rr0 = lane >> 3, cc = (lane & 7) * 4;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long base = (long)(t / (D / 128)) * 128 * D + (long)(t % (D / 128)) * 128 + (long)((wave >> 1) * 64 + rr0) * D + (wave & 1) * 64 + cc;
        f32x4 v[16];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[s2 * 4 + q] = *reinterpret_cast<const f32x4*>(x + base + (long)((s2 >> 1) * 32 + q * 8) * D + (s2 & 1) * 32);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<f32x4*>(y1 + base + (long)((s2 >> 1) * 32 + q * 8) * D + (s2 & 1) * 32) = v[s2 * 4 + q] * 1.5f;
                *reinterpret_cast<f32x4*>(y2 + base + (long)((s2 >> 1) * 32 + q * 8) * D + (s2 & 1) * 32) = v[s2 * 4 + q] * 0.5f;
            }
    }
}

int main() {
    const int B = 65536, D = 1024;
    const long n = (long)B * D;
    float *x, *y1, *y2;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y1, n * 4)); CK(hipMalloc(&y2, n * 4));
    CK(hipMemset(x, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %7.1f us  %6.0f GB/s\n", name, ms * 1e3 / 20, bytes / (ms / 20 * 1e-3) / 1e9);
    };
    time("linear, 1 read + 2 writes, 2048 workgroups", 3.0 * n * 4, [&]() { hipLaunchKernelGGL(k_linear, 2048, 256, 0, 0, (const f32x4*)x, (f32x4*)y1, (f32x4*)y2, n / 4); });
    for (int g : {512, 1024, 2048})
        time(g == 512 ? "tiles 128x128, 1 read + 2 writes, 512 workgroups" : g == 1024 ? "tiles 128x128, 1 read + 2 writes, 1024 workgroups" : "tiles 128x128, 1 read + 2 writes, 2048 workgroups",
             3.0 * n * 4, [&]() { hipLaunchKernelGGL(k_tiled<2>, g, 256, 0, 0, x, y1, y2, B, D); });
    for (int g : {512, 768, 1024})
        time(g == 512 ? "quadrant pattern (8 rows x 128 B per instruction), 512 wg" : g == 768 ? "quadrant pattern, 768 wg" : "quadrant pattern, 1024 wg",
             3.0 * n * 4, [&]() { hipLaunchKernelGGL(k_quad, g, 256, 0, 0, x, y1, y2, B, D); });
    time("tiles 128x128, 1 read + 1 write, 1024 workgroups", 2.0 * n * 4, [&]() { hipLaunchKernelGGL(k_tiled<1>, 1024, 256, 0, 0, x, y1, y2, B, D); });
    return 0;
}
