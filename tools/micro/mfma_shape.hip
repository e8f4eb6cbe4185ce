// Micro-benchmark: sustained exact-fp32 MFMA rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 on random data
// (operands in registers), to see whether the chip holds a different clock for the two shapes under load
// (MI355X_MICROARCH.md, "DVFS give-back" item 7 reports +15 % FLOP/s for the smaller bf16 shape).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* __restrict__ out, int iters, long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
    const long long c0 = clock64(), w0 = wall_clock64();
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {            // 16 MFMAs of 4096 flop per iteration
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 3], b[j], acc[j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        out[t] = s;
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // 32 MFMAs of 2048 flop per iteration
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 3], b[j & 3], acc[j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
        out[t] = s;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int SHAPE>
static void run(int waves_per_simd, const float* din, float* dout, long long* dclk) {
    const int blocks = 256 * waves_per_simd, iters = 20000;        // 256 CUs x (4 waves = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<SHAPE>, blocks, 256, 0, 0, din, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k<SHAPE>, blocks, 256, 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), dclk, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    const double flops = 10.0 * blocks * 4 * (double)iters * 16 * 4096;      // both shapes: 65536 flop per wave-iteration
    printf("%dx%d  %d wave(s)/SIMD : %.1f TFLOP/s   in-kernel clock %.0f MHz\n", SHAPE, SHAPE, waves_per_simd, flops / (ms * 1e-3) / 1e12,
           cs / ws * 100.0);
}

// 32x32x2 with the GEMM kernel's LDS traffic: RD ds_read_b32 per MFMA feeding the next operands (RD = 0, 1, 2)
template <int RD>
__global__ void __launch_bounds__(256) k_lds(const float* __restrict__ in, float* __restrict__ out, int iters, long long* clk) {
    __shared__ float sm[8192];
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = in[(blockIdx.x * 8192 + i) & 0xffff];
    __syncthreads();
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    const float* p = sm + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        const int base = (it & 15) * 256;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 3], b[j], acc[j], 0, 0, 0);
                if (RD >= 1) a[(u + j) & 3] = p[base + (u * 4 + j) * 64];
                if (RD >= 2) b[j] = p[base + 4096 + (u * 4 + j) * 64];
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[t] = s;
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int RD>
static void run_lds(int waves_per_simd, const float* din, float* dout, long long* dclk) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_lds<RD>, blocks, 256, 0, 0, din, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_lds<RD>, blocks, 256, 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), dclk, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    const double flops = 10.0 * blocks * 4 * (double)iters * 16 * 4096;
    printf("32x32 + %d ds_read_b32 per MFMA, %d wave(s)/SIMD : %.1f TFLOP/s   in-kernel clock %.0f MHz\n", RD, waves_per_simd,
           flops / (ms * 1e-3) / 1e12, cs / ws * 100.0);
}

int main() {
    float *din, *dout;
    long long* dclk;
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMalloc(&din, 65536 * 4); hipMalloc(&dout, 1 << 22); hipMalloc(&dclk, 4096 * 16);
    hipMemcpy(din, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) { run<32>(w, din, dout, dclk); run<16>(w, din, dout, dclk); }
    for (int w = 1; w <= 2; ++w) { run_lds<0>(w, din, dout, dclk); run_lds<1>(w, din, dout, dclk); run_lds<2>(w, din, dout, dclk); }
    return 0;
}
