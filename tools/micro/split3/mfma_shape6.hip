// Which bf16 MFMA shape holds the higher clock for the six-term product at this kernel's tile (64 x 64 per wave, two workgroups of four waves per CU)?
// Operands: random bf16 pieces in registers (12 fragments per k-step, as the k-loop of k_gemm_s3 reads them), no LDS, no loads in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) k_mfma(const bf16x8* __restrict__ src, float* __restrict__ out, long long* __restrict__ dbg, int nk) {
    const int tid = threadIdx.x;
    bf16x8 a[3][4], b[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[s][i] = src[(blockIdx.x * 24 + s * 4 + i) * 256 + tid];
            b[s][i] = src[(blockIdx.x * 24 + 12 + s * 4 + i) * 256 + tid];
        }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int k = 0; k < nk; ++k) {      // one 16-deep k-tile: 24 MFMAs
#define T32(SA, SB) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[SA][i + 2 * (k & 1)], b[SB][j + 2 * (k & 1)], acc[i][j], 0, 0, 0);
            T32(0, 0) T32(0, 1) T32(0, 2) T32(1, 0) T32(1, 1) T32(2, 0)
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int k = 0; k < nk; k += 2) {   // one 32-deep k-step: 96 MFMAs = the flops of two 16-deep k-tiles above
#define T16(SA, SB) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[SA][i], b[SB][j], acc[i][j], 0, 0, 0);
            T16(0, 0) T16(0, 1) T16(0, 2) T16(1, 0) T16(1, 1) T16(2, 0)
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { dbg[2 * blockIdx.x] = c1 - c0; dbg[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const bf16x8* src, float* out, long long* dbg, int reps, const char* name) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_mfma<SHAPE>, 512, 256, 0, 0, src, out, dbg, 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_mfma<SHAPE>, 512, 256, 0, 0, src, out, dbg, 64);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(1024); CK(hipMemcpy(h.data(), dbg, 8192, hipMemcpyDeviceToHost));
    std::vector<double> g, c; for (int i = 0; i < 512; ++i) { g.push_back(h[2 * i] * 0.1 / h[2 * i + 1]); c.push_back((double)h[2 * i]); }
    std::sort(g.begin(), g.end()); std::sort(c.begin(), c.end());
    printf("%-10s %7.1f us per launch, in-kernel clock median %.3f GHz, loop cycles median %.0f\n", name, ms * 1e3 / reps, g[256], c[256]);
}
int main() {
    const size_t n = (size_t)512 * 24 * 256;
    std::vector<unsigned> h(n * 4);
    srand(1);
    for (auto& v : h) {       // two random bf16 in [-2, 2): sign, exponent 119..128 (mixed magnitudes as split pieces have), random mantissa
        unsigned lo = ((rand() & 1) << 15) | ((119 + rand() % 10) << 7) | (rand() & 127), hi = ((rand() & 1) << 15) | ((119 + rand() % 10) << 7) | (rand() & 127);
        v = lo | (hi << 16);
    }
    bf16x8* src; float* out; long long* dbg;
    CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&dbg, 8192));
    CK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice));
    for (int rd = 0; rd < 3; ++rd) {
        run<32>(src, out, dbg, 200, "32x32x16");
        run<16>(src, out, dbg, 200, "16x16x32");
    }
    return 0;
}
