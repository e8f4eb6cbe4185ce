// SENETLayer in ONE pass per direction (/root/reference/rec_now/layers/senet_layer.py:93-119: squeeze = mean over each
// field's columns, excitation = two Dense layers, scale = field * weight).  The unfused route (interact.hip k_senet_* +
// recnow_multi_dense) reads every field three times and the upstream gradient twice, and runs the tiny excitation MLP as
// five separate launches on (B, F)-sized tensors.  Here a workgroup owns SF_R consecutive batch rows: the fields (and, in
// backward, the gradient) are read ONCE into registers, the (F -> M -> F) excitation and its backward run per row out of
// LDS-resident weights, and the layer output / field gradients leave from the same registers.
//   forward : reads 4*B*F*D, writes 4*B*F*D (+ the B x (2F + M) saved activations)
//   backward: reads 8*B*F*D, writes 4*B*F*D; the weight and bias gradients are accumulated per workgroup in registers over
//             its rows and added up in a fixed order afterwards (deterministic).
// Limits: equal field widths D % 4 == 0 with D/4 a power of two, F*D/4 <= 256 (a row is one float4 per thread), F <= 64,
// M <= 64, built-in activations.  Everything else takes the unfused kernels.
#include <atomic>
#include "common.hpp"
#include "gemm.hpp"

#ifndef SF_R
#define SF_R 4                 // batch rows per workgroup step (-DSF_R=8: A/B through tools/build_variant.py)
#endif
typedef float sf_f4 __attribute__((ext_vector_type(4)));

struct SfDims {
    int F, D, M, act1, act2, has_bias;
};

static inline bool sf_ok(int F, int D, int M) {
    if (F < 1 || F > 64 || M < 1 || M > 64 || D < 4 || (D & 3)) return false;
    const int q = D / 4;
    return (q & (q - 1)) == 0 && F * q <= 256;
}
static inline int sf_grid(int64_t B) {
    int64_t g = (B + SF_R - 1) / SF_R;
    return (int)(g > 2048 ? 2048 : (g > 0 ? g : 1));
}
// LDS floats: W1 [F][M], W2 [M][F], b1 [M], b2 [F], then per-row vectors
static inline size_t sf_lds_floats(int F, int M) { return (size_t)2 * F * M + M + F + (size_t)SF_R * (4 * F + 2 * M); }

__device__ __forceinline__ void sf_load_weights(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                const float* __restrict__ b2, int F, int M, float* w1s, float* w2s, float* b1s, float* b2s) {
    for (int i = threadIdx.x; i < F * M; i += 256) {
        w1s[i] = W1[i];
        w2s[i] = W2[i];
    }
    for (int i = threadIdx.x; i < M; i += 256) b1s[i] = b1 ? b1[i] : 0.f;
    for (int i = threadIdx.x; i < F; i += 256) b2s[i] = b2 ? b2[i] : 0.f;
}

__global__ void __launch_bounds__(256)
k_senet_fused_fwd(const float* const* __restrict__ fields, SfDims dm, int64_t B, const float* __restrict__ W1, const float* __restrict__ b1,
                  const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out, float* __restrict__ sq_save,
                  float* __restrict__ h_save, float* __restrict__ w_save) {
    extern __shared__ __attribute__((aligned(16))) float sf_lds[];
    const int F = dm.F, D = dm.D, M = dm.M, Q = D / 4, QT = F * Q;
    float* w1s = sf_lds;
    float* w2s = w1s + F * M;
    float* b1s = w2s + F * M;
    float* b2s = b1s + M;
    float* sq = b2s + F;              // [SF_R][F]
    float* hs = sq + SF_R * F;        // [SF_R][M]
    float* ws = hs + SF_R * M;        // [SF_R][F]
    sf_load_weights(W1, b1, W2, b2, F, M, w1s, w2s, b1s, b2s);
    const int t = threadIdx.x;
    const bool col = t < QT;
    const int f = col ? t / Q : 0, d = col ? (t - f * Q) * 4 : 0;
    const rn_gcf xf = (rn_gcf)fields[f] + d;      // global address space: see common.hpp
    const int64_t FD = (int64_t)F * D;
    int P = 1;                                             // threads per hidden unit in the first product (power of two <= 8)
    while (P < 8 && 2 * P * SF_R * M <= 256) P *= 2;
    // Round 5: the rows of step s + 1 are REQUESTED while step s runs its excitation (a second register set, unconditional loads from clamped
    // rows -- a load under a lane condition ends in a full wait at its merge, DESIGN.md 5e -- masked when used): a step was
    // load -> wait -> squeeze -> two small products -> scale -> store, strictly in series per workgroup (3.5 TB/s with eight workgroups per CU).
    sf_f4 vn[SF_R];
    auto request = [&](int64_t r0) {
#pragma unroll
        for (int j = 0; j < SF_R; ++j) {
            const int64_t r = r0 + j < B ? r0 + j : B - 1;
            vn[j] = *reinterpret_cast<const RN_GLOBAL sf_f4*>(xf + r * D);
        }
    };
    request((int64_t)blockIdx.x * SF_R);
    for (int64_t b0 = (int64_t)blockIdx.x * SF_R; b0 < B; b0 += (int64_t)gridDim.x * SF_R) {
        sf_f4 v[SF_R];
#pragma unroll
        for (int j = 0; j < SF_R; ++j) v[j] = (col && b0 + j < B) ? vn[j] : sf_f4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                   // previous step is done with sq / hs / ws (and the weights are loaded)
#pragma unroll
        for (int j = 0; j < SF_R; ++j) {
            float s = (v[j].x + v[j].y) + (v[j].z + v[j].w);
            for (int o = Q / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (col && d == 0) sq[j * F + f] = s / (float)D;
        }
        {
            const int64_t nb0 = b0 + (int64_t)gridDim.x * SF_R;
            request(nb0 < B ? nb0 : b0);                   // (the last step re-reads its own rows: the loads stay unconditional)
        }
        __syncthreads();
        // h = act1(sq W1 + b1): P threads share one output's k range; 256 / P outputs per round (the P lanes of an output run the same rounds)
        for (int o = t / P; o < (SF_R * M + 256 / P - 1) / (256 / P) * (256 / P); o += 256 / P) {
            const int part = t % P;
            const bool act = o < SF_R * M;
            const int j = act ? o / M : 0, m = act ? o - j * M : 0;
            float a = 0.f;
            if (act)
                for (int k = part; k < F; k += P) a += sq[j * F + k] * w1s[k * M + m];
            for (int x = P / 2; x > 0; x >>= 1) a += __shfl_xor(a, x, 64);
            if (act && part == 0) {
                a = rn_act(a + b1s[m], dm.act1);
                hs[o] = a;
                if (b0 + j < B) h_save[(b0 + j) * M + m] = a;
            }
        }
        __syncthreads();
        for (int i = t; i < SF_R * F; i += 256) {          // w = act2(h W2 + b2)
            const int j = i / F, k = i - j * F;
            float a = b2s[k];
            for (int m = 0; m < M; ++m) a += hs[j * M + m] * w2s[m * F + k];
            a = rn_act(a, dm.act2);
            ws[i] = a;
            if (b0 + j < B) {
                w_save[(b0 + j) * F + k] = a;
                sq_save[(b0 + j) * F + k] = sq[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SF_R; ++j)
            if (col && b0 + j < B) *reinterpret_cast<sf_f4*>(out + (b0 + j) * FD + t * 4) = v[j] * ws[j * F + f];
    }
}

// ---- round 6: the same pass on 8-row tiles with the fields read as 512 contiguous bytes per half wave (D = 16, F % 8 == 0) -----------------------------------
// The kernel above maps thread t to (field t / 4, chunk t % 4) of ONE row: its wave loads touch 16 field tensors x 64 B each (a quarter of the 256 B that the
// four rows of a step hold per field), 4.1 TB/s.  Here a half wave reads (8 rows) x (64 B) of ONE field -- 512 contiguous bytes -- the rows change layout in an LDS
// tile ([row][F D], row stride padded by 16 floats: the eight lanes of a ds_write_b128 group are two rows x four chunks, 16 banks apart) and leave as whole rows
// (a wave stores 1 KiB of one output row).  Squeeze, excitation and the saved activations as above, per 8 rows.
#define S8_R 8
#define S8_LD 1040
static inline bool sf8_ok(int F, int D, int M) { return D == 16 && F >= 8 && F <= 64 && F % 8 == 0 && M >= 1 && M <= 64; }
static inline size_t sf8_lds_floats(int F, int M) { return (size_t)2 * F * M + M + F + (size_t)S8_R * (2 * F + M) + (size_t)S8_R * S8_LD; }

__global__ void __launch_bounds__(256)
k_senet_fused_fwd8(const float* const* __restrict__ fields, SfDims dm, int64_t B, const float* __restrict__ W1, const float* __restrict__ b1,
                   const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out, float* __restrict__ sq_save,
                   float* __restrict__ h_save, float* __restrict__ w_save) {
    extern __shared__ __attribute__((aligned(16))) float sf_lds[];
    const int F = dm.F, M = dm.M, QT = F * 4, NP = F / 8;
    constexpr int D = 16;
    float* w1s = sf_lds;
    float* w2s = w1s + F * M;
    float* b1s = w2s + F * M;
    float* b2s = b1s + M;
    float* sq = b2s + F;              // [8][F]
    float* hs = sq + S8_R * F;        // [8][M]
    float* ws = hs + S8_R * M;        // [8][F]
    float* X = ws + S8_R * F;         // [8][S8_LD]
    sf_load_weights(W1, b1, W2, b2, F, M, w1s, w2s, b1s, b2s);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, fi = lane >> 5, r = (lane >> 2) & 7, q = lane & 3;
    // this thread's field of pass p: 8 p + 2 w + fi; the pointers are fetched once (a per-lane pointer load in front of every row load would be a dependent latency)
    rn_gcf fp[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) fp[p] = (rn_gcf)fields[(p < NP ? 8 * p : 0) + 2 * w + fi] + 4 * q;
    const int64_t FD = (int64_t)F * D;
    int P = 1;
    while (P < 8 && 2 * P * S8_R * M <= 256) P *= 2;
    sf_f4 vn[8];
    auto request = [&](int64_t r0) {
        const int64_t row = r0 + r < B ? r0 + r : B - 1;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            vn[p] = *reinterpret_cast<const RN_GLOBAL sf_f4*>(fp[p] + row * D);      // unconditional: passes beyond NP re-read the fields of pass 0
    };
    request((int64_t)blockIdx.x * S8_R);
    for (int64_t b0 = (int64_t)blockIdx.x * S8_R; b0 < B; b0 += (int64_t)gridDim.x * S8_R) {
        sf_f4 v[8];
        const bool live = b0 + r < B;
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = (p < NP && live) ? vn[p] : sf_f4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                                   // the previous tile is done with X / ws (and the weights are loaded)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p < NP) {
                const int f = 8 * p + 2 * w + fi;
                *reinterpret_cast<sf_f4*>(X + r * S8_LD + f * D + 4 * q) = v[p];
                float s = (v[p].x + v[p].y) + (v[p].z + v[p].w);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 1, 64);
                if (q == 0) sq[r * F + f] = s / (float)D;
            }
        }
        {
            const int64_t nb0 = b0 + (int64_t)gridDim.x * S8_R;
            request(nb0 < B ? nb0 : b0);
        }
        __syncthreads();
        for (int o = t / P; o < (S8_R * M + 256 / P - 1) / (256 / P) * (256 / P); o += 256 / P) {      // h = act1(sq W1 + b1)
            const int part = t % P;
            const bool act = o < S8_R * M;
            const int j = act ? o / M : 0, m = act ? o - j * M : 0;
            float a = 0.f;
            if (act)
                for (int k = part; k < F; k += P) a += sq[j * F + k] * w1s[k * M + m];
            for (int x = P / 2; x > 0; x >>= 1) a += __shfl_xor(a, x, 64);
            if (act && part == 0) {
                a = rn_act(a + b1s[m], dm.act1);
                hs[o] = a;
                if (b0 + j < B) h_save[(b0 + j) * M + m] = a;
            }
        }
        __syncthreads();
        for (int i = t; i < S8_R * F; i += 256) {          // w = act2(h W2 + b2)
            const int j = i / F, k = i - j * F;
            float a = b2s[k];
            for (int m = 0; m < M; ++m) a += hs[j * M + m] * w2s[m * F + k];
            a = rn_act(a, dm.act2);
            ws[i] = a;
            if (b0 + j < B) {
                w_save[(b0 + j) * F + k] = a;
                sq_save[(b0 + j) * F + k] = sq[i];
            }
        }
        __syncthreads();
        if (t < QT) {
#pragma unroll
            for (int j = 0; j < S8_R; ++j)
                if (b0 + j < B)
                    *reinterpret_cast<sf_f4*>(out + (b0 + j) * FD + t * 4) = *reinterpret_cast<const sf_f4*>(X + j * S8_LD + t * 4) * ws[j * F + (t >> 2)];
        }
    }
}

// Weight-gradient outputs are numbered  [0, F*M): dW1[k][m];  [F*M, 2*F*M): dW2[m][k];  then db1[M], db2[F].
// NACC = weight-gradient outputs per thread (>= ceil((2*F*M + M + F) / 256); at most 34 for F = M = 64).
// The per-row vectors of a step live in LDS with a FIXED row stride SF_RS (F, M <= 64): the rows of one vector are then
// immediate offsets of one address, so a weight-gradient output costs two address registers, not two per row.
#define SF_RS 64
template <int NACC>
__global__ void __launch_bounds__(256)
k_senet_fused_bwd(const float* const* __restrict__ fields, float* const* __restrict__ dfields, SfDims dm, int64_t B,
                  const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ dout,
                  const float* __restrict__ sq_save, const float* __restrict__ h_save, const float* __restrict__ w_save,
                  float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sf_lds[];
    const int F = dm.F, D = dm.D, M = dm.M, Q = D / 4, QT = F * Q, FM = F * M;
    // weights with odd row strides: the backward products walk W2 by rows per thread (dz2 W2^T) and W1 by rows per thread
    // (dz1 W1^T); with strides F = 64 / M = 16 every lane of a wave would hit the same one or two LDS banks
    const int L1 = M + 1, L2 = F + 1;
    float* w1s = sf_lds;                  // [F][M + 1]
    float* w2s = w1s + F * L1;            // [M][F + 1]
    float* sq = w2s + M * L2;             // seven [SF_R][SF_RS] vectors
    float* hs = sq + SF_R * SF_RS;
    float* ws = hs + SF_R * SF_RS;
    float* z2 = ws + SF_R * SF_RS;        // dW = <dout, x> per field, then dz2 = dW * act2'(w)
    float* z1 = z2 + SF_R * SF_RS;        // dz1
    float* dq = z1 + SF_R * SF_RS;        // dsq
    float* ones = dq + SF_R * SF_RS;      // 1 at [j][0]: the "a" operand of the bias gradients
    for (int i = threadIdx.x; i < FM; i += 256) {
        w1s[(i / M) * L1 + i % M] = W1[i];
        w2s[(i / F) * L2 + i % F] = W2[i];
    }
    if (threadIdx.x < SF_R) ones[threadIdx.x * SF_RS] = 1.f;
    const int t = threadIdx.x;
    const bool col = t < QT;
    const int f = col ? t / Q : 0, d = col ? (t - f * Q) * 4 : 0;
    const rn_gcf xf = (rn_gcf)fields[f] + d;      // global address space: see common.hpp
    const rn_gf dxf = (rn_gf)dfields[f] + d;
    const int64_t FD = (int64_t)F * D;
    const int NOUT = 2 * FM + M + F;
    int P = 1;                                             // threads per hidden unit in dz2 W2^T (power of two <= 8)
    while (P < 8 && 2 * P * SF_R * M <= 256) P *= 2;
    // every weight-gradient output is sum_j a[j] * b[j] over the rows j of a step, a and b taken from the row vectors
    float acc[NACC];
    int pa[NACC], pb[NACC];                    // offsets into sf_lds (pointers would be generic: 64-bit, read with flat loads)
    const int o_ones = (int)(ones - sf_lds), o_sq = (int)(sq - sf_lds), o_hs = (int)(hs - sf_lds), o_z1 = (int)(z1 - sf_lds),
              o_z2 = (int)(z2 - sf_lds);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int o = t + 256 * i;
        acc[i] = 0.f;
        pa[i] = o_ones; pb[i] = o_ones;            // o >= NOUT: an unused slot
        if (o < FM) {                              // dW1[k][m] += sq[k] * dz1[m]
            const int k = o / M, m = o - k * M;
            pa[i] = o_sq + k; pb[i] = o_z1 + m;
        } else if (o < 2 * FM) {                   // dW2[m][k] += h[m] * dz2[k]
            const int m = (o - FM) / F, k = (o - FM) - m * F;
            pa[i] = o_hs + m; pb[i] = o_z2 + k;
        } else if (o < 2 * FM + M) {               // db1[m] += dz1[m]
            pb[i] = o_z1 + (o - 2 * FM);
        } else if (o < NOUT) {                     // db2[k] += dz2[k]
            pb[i] = o_z2 + (o - 2 * FM - M);
        }
    }
    // the fields and the upstream gradient of step s + 1 are requested while step s computes (as the forward kernel, round 5)
    sf_f4 vn[SF_R], gn[SF_R];
    const int tq = col ? t : QT - 1;                       // a thread beyond the row (F * D / 4 < 256) reads the last float4 and masks it
    auto request = [&](int64_t r0) {
#pragma unroll
        for (int j = 0; j < SF_R; ++j) {
            const int64_t r = r0 + j < B ? r0 + j : B - 1;
            vn[j] = *reinterpret_cast<const RN_GLOBAL sf_f4*>(xf + r * D);
            gn[j] = *reinterpret_cast<const sf_f4*>(dout + r * FD + tq * 4);
        }
    };
    request((int64_t)blockIdx.x * SF_R);
    for (int64_t b0 = (int64_t)blockIdx.x * SF_R; b0 < B; b0 += (int64_t)gridDim.x * SF_R) {
        sf_f4 v[SF_R], g[SF_R];
#pragma unroll
        for (int j = 0; j < SF_R; ++j) {
            const bool live = col && b0 + j < B;
            v[j] = live ? vn[j] : sf_f4{0.f, 0.f, 0.f, 0.f};
            g[j] = live ? gn[j] : sf_f4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                   // previous step is done with the row vectors
        for (int i = t; i < SF_R * F; i += 256) {
            const int j = i / F, k = i - j * F;
            const bool live = b0 + j < B;
            ws[j * SF_RS + k] = live ? w_save[b0 * F + i] : 0.f;        // rows of a step are consecutive in the saved arrays
            sq[j * SF_RS + k] = live ? sq_save[b0 * F + i] : 0.f;
        }
        for (int i = t; i < SF_R * M; i += 256) {
            const int j = i / M, m = i - j * M;
            hs[j * SF_RS + m] = (b0 + j < B) ? h_save[b0 * M + i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SF_R; ++j) {
            float s = (v[j].x * g[j].x + v[j].y * g[j].y) + (v[j].z * g[j].z + v[j].w * g[j].w);
            for (int o = Q / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (col && d == 0) z2[j * SF_RS + f] = s;
        }
        {
            const int64_t nb0 = b0 + (int64_t)gridDim.x * SF_R;
            request(nb0 < B ? nb0 : b0);
        }
        __syncthreads();
        for (int i = t; i < SF_R * F; i += 256) {          // dz2 = dW * act2'(w)   (rows past B: zero)
            const int j = i / F, k = i - j * F;
            z2[j * SF_RS + k] *= rn_act_grad_from_out(ws[j * SF_RS + k], dm.act2);
        }
        __syncthreads();
        // dz1 = (dz2 W2^T) * act1'(h): P threads share one output's k range; 256 / P outputs per round
        for (int o = t / P; o < (SF_R * M + 256 / P - 1) / (256 / P) * (256 / P); o += 256 / P) {
            const int part = t % P;
            const bool act = o < SF_R * M;
            const int j = act ? o / M : 0, m = act ? o - j * M : 0;
            float a = 0.f;
            if (act)
                for (int k = part; k < F; k += P) a += z2[j * SF_RS + k] * w2s[m * L2 + k];
            for (int x = P / 2; x > 0; x >>= 1) a += __shfl_xor(a, x, 64);
            if (act && part == 0) z1[j * SF_RS + m] = a * rn_act_grad_from_out(hs[j * SF_RS + m], dm.act1);
        }
        __syncthreads();
        for (int i = t; i < SF_R * F; i += 256) {          // dsq = dz1 W1^T
            const int j = i / F, k = i - j * F;
            float a = 0.f;
            for (int m = 0; m < M; ++m) a += z1[j * SF_RS + m] * w1s[k * L1 + m];
            dq[j * SF_RS + k] = a;
        }
        // weight gradients of this step's rows, into this thread's outputs (rows past B contribute zeros)
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < SF_R; ++j) a += sf_lds[pa[i] + j * SF_RS] * sf_lds[pb[i] + j * SF_RS];
            acc[i] += a;
        }
        __syncthreads();                                   // dq complete (and z1 read by everyone before the next step rewrites it)
#pragma unroll
        for (int j = 0; j < SF_R; ++j)
            if (col && b0 + j < B)
                *reinterpret_cast<RN_GLOBAL sf_f4*>(dxf + (b0 + j) * D) = g[j] * ws[j * SF_RS + f] + dq[j * SF_RS + f] / (float)D;
    }
    float* dst = part + (int64_t)blockIdx.x * NOUT;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int o = t + 256 * i;
        if (o < NOUT) dst[o] = acc[i];
    }
}

extern "C" int recnow_senet_fused_supported(int F, int D, int M) { return sf_ok(F, D, M) ? 1 : 0; }

extern "C" size_t recnow_senet_fused_workspace_bytes(int64_t B, int F, int M) {
    const int64_t nout = 2 * (int64_t)F * M + M + F;
    return rn_align((size_t)sf_grid(B) * nout * sizeof(float)) + rn_colsum_ws_bytes(sf_grid(B), nout) + rn_align((size_t)nout * sizeof(float)) + 256;
}

extern "C" int recnow_senet_fused_fwd(const float* const* fields, int F, int D, int64_t B, const float* W1, const float* b1,
                                      const float* W2, const float* b2, int M, int act1, int act2, float* out, float* sq_save,
                                      float* h_save, float* w_save, void* stream) {
    if (B < 0 || !sf_ok(F, D, M)) return B < 0 ? RECNOW_EINVAL : RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if (!fields || !W1 || !W2 || !out || !sq_save || !h_save || !w_save) return RECNOW_EINVAL;
    const SfDims dm = {F, D, M, act1, act2, (b1 || b2) ? 1 : 0};
    static const bool tile8 = []() { const char* e = getenv("RECNOW_SENET_TILE8"); return !e || e[0] != '0'; }();      // A/B switch
    if (tile8 && sf8_ok(F, D, M) && ((uintptr_t)out & 15) == 0) {
        const size_t lds = sf8_lds_floats(F, M) * sizeof(float);
        if (lds > 64 * 1024) {
            static std::atomic<bool> raised[64];
            int dev = 0;
            RN_HIP(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire)) {
                RN_HIP(hipFuncSetAttribute((const void*)k_senet_fused_fwd8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sf8_lds_floats(64, 64) * sizeof(float))));
                if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
            }
        }
        int64_t g = (B + S8_R - 1) / S8_R;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(k_senet_fused_fwd8, (int)g, 256, lds, (hipStream_t)stream, fields, dm, B, W1, b1, W2, b2, out, sq_save, h_save, w_save);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    hipLaunchKernelGGL(k_senet_fused_fwd, sf_grid(B), 256, sf_lds_floats(F, M) * sizeof(float), (hipStream_t)stream, fields, dm, B, W1,
                       b1, W2, b2, out, sq_save, h_save, w_save);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// dW1 (F, M), db1 (M), dW2 (M, F), db2 (F): db1 / db2 may be NULL (layer built without bias).
extern "C" int recnow_senet_fused_bwd(const float* const* fields, float* const* dfields, int F, int D, int64_t B, const float* W1,
                                      const float* W2, int M, int act1, int act2, const float* dout, const float* sq_save,
                                      const float* h_save, const float* w_save, float* dW1, float* db1, float* dW2, float* db2,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || !sf_ok(F, D, M)) return B < 0 ? RECNOW_EINVAL : RECNOW_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int FM = F * M, nout = 2 * FM + M + F;
    if (B == 0) {
        if (dW1) RN_HIP(hipMemsetAsync(dW1, 0, (size_t)FM * sizeof(float), st));
        if (dW2) RN_HIP(hipMemsetAsync(dW2, 0, (size_t)FM * sizeof(float), st));
        if (db1) RN_HIP(hipMemsetAsync(db1, 0, (size_t)M * sizeof(float), st));
        if (db2) RN_HIP(hipMemsetAsync(db2, 0, (size_t)F * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!fields || !dfields || !W1 || !W2 || !dout || !sq_save || !h_save || !w_save || !dW1 || !dW2 || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_senet_fused_workspace_bytes(B, F, M)) return RECNOW_EWORKSPACE;
    int nb = sf_grid(B);
    RnCarver c(ws, ws_bytes);
    float* part = c.take<float>((size_t)nb * nout);
    float* sums = c.take<float>((size_t)nout);
    void* cws = (void*)(c.base + c.off);
    const size_t cbytes = ws_bytes - c.off;
    const SfDims dm = {F, D, M, act1, act2, 0};
    const size_t lds = ((size_t)F * (M + 1) + (size_t)M * (F + 1) + 7 * SF_R * SF_RS) * sizeof(float);
    const int nacc = (nout + 255) / 256;
    // one resident wave of workgroups (rows are grid-strided): every further workgroup costs a weight fill and a slab of partial sums that
    // the column sum reads back (as the DCNLayer backward); RECNOW_SENET_RESIDENT=0 is the A/B switch: 0.85 -> 0.78 ms for the layer's step at B = 131 072; the forward measured the same either way
    static const bool resident = []() { const char* e = getenv("RECNOW_SENET_RESIDENT"); return !e || e[0] != '0'; }();
#define SF_BWD(NA) do {                                                                                                            \
        if (resident) {                                                                                                            \
            int occ = 0;                                                                                                           \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_senet_fused_bwd<NA>, 256, lds) != hipSuccess || occ < 1) occ = 1; \
            if (nb > 256 * occ) nb = 256 * occ;                                                                                    \
        }                                                                                                                          \
        hipLaunchKernelGGL(k_senet_fused_bwd<NA>, nb, 256, lds, st, fields, dfields, dm, B, W1, W2, dout, sq_save, h_save, w_save, part); \
    } while (0)
    if (nacc <= 3) SF_BWD(3);
    else if (nacc <= 5) SF_BWD(5);
    else if (nacc <= 9) SF_BWD(9);
    else if (nacc <= 17) SF_BWD(17);
    else SF_BWD(34);
#undef SF_BWD
    RN_LAUNCH_CHECK();
    int rc = rn_colsum(part, nullptr, 0, 0, nb, nout, nout, sums, 0, cws, cbytes, st);
    if (rc) return rc;
    RN_HIP(hipMemcpyAsync(dW1, sums, (size_t)FM * sizeof(float), hipMemcpyDeviceToDevice, st));
    RN_HIP(hipMemcpyAsync(dW2, sums + FM, (size_t)FM * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (db1) RN_HIP(hipMemcpyAsync(db1, sums + 2 * FM, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (db2) RN_HIP(hipMemcpyAsync(db2, sums + 2 * FM + M, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RECNOW_OK;
}
