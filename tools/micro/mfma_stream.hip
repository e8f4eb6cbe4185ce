// Micro-benchmark (round 4): does the fp32 MFMA SHAPE matter once the loop carries the GEMM kernels' memory activity?
// the c3 step's products run their MFMAs at ~2.03 GHz (power budget) although a bare fp32 MFMA loop holds 2.38 GHz for both shapes; the
// guide (MI355X_MICROARCH.md, DVFS give-back item 7) reports that for bf16 the smaller shape holds a higher clock under load (+12..15 % FLOP/s).
// Here: the same flops per wave-iteration on v_mfma_f32_32x32x2_f32 (16 MFMAs) or v_mfma_f32_16x16x4_f32 (32 MFMAs), with LD float4 global loads per
// lane and iteration streamed from a buffer far larger than the caches (k_gemm's 128 x 128 x 32 k-tile: 2 per 16 MFMAs of 4096 flop), written to LDS and
// read back as fragments (one ds_read_b32 per 4096 flop of MFMA, as the GEMM kernels).  Prints TFLOP/s, the in-kernel clock and the GB/s streamed.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_stream mfma_stream.hip && ./mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int LD>
__global__ void __launch_bounds__(256, 2) k(const f32x4* __restrict__ stream, long long n4, float* __restrict__ out, int iters, long long* clk) {
    __shared__ float sm[2][4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const long long per_wg = (long long)iters * 256 * (LD > 0 ? LD : 1);
    const f32x4* p = stream + ((long long)blockIdx.x * per_wg) % (n4 - per_wg > 0 ? n4 - per_wg : 1);
    for (int i = tid; i < 4096; i += 256) { sm[0][i] = (float)(i & 15) * 0.01f; sm[1][i] = (float)(i & 7) * 0.02f; }
    __syncthreads();
    f32x4 v[LD > 0 ? LD : 1];
#pragma unroll
    for (int u = 0; u < (LD > 0 ? LD : 1); ++u) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (LD > 0) {
#pragma unroll
        for (int u = 0; u < LD; ++u) v[u] = p[u * 256 + tid];
    }
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = sm[0][lane + 64 * i]; b[i] = sm[1][lane + 64 * i]; }
    f32x16 acc32[4];
    f32x4 acc16[16];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc32[j][r] = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc16[j][r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (LD > 0) {       // the tile requested one iteration ago goes to LDS, the next one is requested (register-staged double buffering)
#pragma unroll
            for (int u = 0; u < LD; ++u) *reinterpret_cast<f32x4*>(&sm[buf][(u * 1024 + tid * 4) & 4095]) = v[u];
            const f32x4* q = p + (long long)(it + 1 < iters ? it + 1 : it) * 256 * LD;
#pragma unroll
            for (int u = 0; u < LD; ++u) v[u] = q[u * 256 + tid];
        }
        if (SHAPE == 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc32[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 3], b[j], acc32[j], 0, 0, 0);
                    a[(u + j) & 3] = sm[buf ^ 1][(lane + 64 * (u * 4 + j)) & 4095];       // one fragment read per MFMA of 4096 flop
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 3], b[j & 3], acc16[j], 0, 0, 0);
                    if (j & 1) a[(u + j) & 3] = sm[buf ^ 1][(lane + 64 * (u * 8 + (j >> 1))) & 4095];      // one read per two MFMAs of 2048 flop
                }
            }
        }
        if (LD > 0) __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc32[j][r];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc16[j][r];
    for (int u = 0; u < (LD > 0 ? LD : 1); ++u) s += v[u].x;
    out[blockIdx.x * 256 + tid] = s;
    const long long c1 = clock64(), w1 = wall_clock64();
    if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int SHAPE, int LD>
static void run(const f32x4* ds, long long n4, float* dout, long long* dclk) {
    const int blocks = 512, iters = 4000;            // 256 CUs x 2 workgroups of 4 waves: two waves per SIMD, as the GEMM kernels
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SHAPE, LD>), blocks, 256, 0, 0, ds, n4, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((k<SHAPE, LD>), blocks, 256, 0, 0, ds, n4, dout, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), dclk, blocks * 2 * sizeof(long long), hipMemcpyDeviceToHost);
    double cs = 0, ws = 0;
    for (int i = 0; i < blocks; ++i) { cs += h[2 * i]; ws += h[2 * i + 1]; }
    const double flops = (double)reps * blocks * 4 * (double)iters * 16 * 4096;
    const double bytes = (double)reps * blocks * (double)iters * 256 * LD * 16;
    printf("%dx%d  %d float4 loads per lane and 16x4096 flop : %6.1f TFLOP/s   clock %4.0f MHz   %5.0f GB/s streamed\n", SHAPE, SHAPE, LD,
           flops / (ms * 1e-3) / 1e12, cs / ws * 100.0, bytes / (ms * 1e-3) / 1e9);
}

int main() {
    const long long n4 = (3ll << 30) / 16;           // 3 GiB of float4: far beyond the 256 MiB Infinity Cache
    f32x4* ds;
    float* dout;
    long long* dclk;
    if (hipMalloc(&ds, n4 * 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMalloc(&dout, 512 * 256 * 4);
    hipMalloc(&dclk, 4096 * 16);
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    for (long long off = 0; off < n4 * 16; off += (1 << 22)) hipMemcpy((char*)ds + off, h.data(), 1 << 22, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0>(ds, n4, dout, dclk); run<16, 0>(ds, n4, dout, dclk);
        run<32, 2>(ds, n4, dout, dclk); run<16, 2>(ds, n4, dout, dclk);
        run<32, 4>(ds, n4, dout, dclk); run<16, 4>(ds, n4, dout, dclk);
    }
    return 0;
}
