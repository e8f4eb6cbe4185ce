// CINLayer backward, data gradients of one layer in ONE forward-sized product (reference forward: /root/reference/rec_now/layers/cin_layer.py:101-110;
// the backward is TF autodiff there, derived in SURVEY.md Appendix C7):
//
//     T[m][(f,h)] = sum_c dX_k[m][c] * W_k[c][f*H_{k-1} + h]          -- an (M x H_k) x (H_k x F*H_{k-1}) product on the matrix cores
//     dX_{k-1}[m][h] += sum_f T[m][(f,h)] * x0[m][f]                  -- VALU, on the accumulator tile
//     dx0[m][f]      += sum_h T[m][(f,h)] * X_{k-1}[m][h]             -- VALU + a 32-lane transpose-reduce, on the same tile
//
// Rounds 1-3 ran the two reductions as two more forward-sized products with the outer products (dX_k (x) x0), (dX_k (x) X_{k-1}) generated in
// the operand loads: T was formed twice.  Here it is formed once and never reaches HBM (it is M x F*H floats: 8.6 GB per layer at BASELINE
// config 4): the step executes 4.12 TFLOP per rank instead of 5.50.
//
// Geometry.  One workgroup = 128 rows m, 512 threads = 8 waves (4 x 2): wave (wm, wn) owns rows 32 wm .. + 31 and 64 of the 128 tile columns.
// The workgroup walks the column tiles of T (128 columns = 128 / HP fields) with the k-loop running on across tile boundaries: k-tiles of 32,
// A (dX_k rows, re-read from L1 / L2 for every field) and B (the W_k slice, L2-resident) double-buffered in LDS, staged through registers:
// tile t + 1 goes registers -> LDS right behind the barrier that opens tile t, the loads of tile t + 2 are issued at once (clamped index: no load
// sits under a condition).  Every H_k / 32 k-tiles a field's T tile is complete in the 32x32x2 MFMA accumulators
// (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
//   * dX_{k-1}: a second accumulator (32 registers) += T * x0[row][f]; x0 of the tile sits transposed in LDS ([f][row]: one ds_read_b128 = 4 rows);
//   * dx0: p[r] = sum over the lane's columns of T * X_{k-1} (X_{k-1} of the lane's 32 positions lives in registers for the whole kernel), then
//     the 16 values of a lane are summed over the 32 lanes that share its rows by a halving butterfly (16 shuffles instead of 80) and added to
//     LDS with ds_add_f32 -- at most two waves add to one cell and a + b = b + a in floating point, so the result does not depend on the order.
// HP = 64 (layer 1 of CIN, X_0 = x0): a column tile holds two fields, wave column wn owns field 2 ft + wn: its row sums are complete and its
// dX_{k-1} accumulator is a partial sum over the fields of its parity (joined at the end); HP = 128: one field, wave column wn owns 64 of its h.
// At the end the accumulators are joined in LDS (the operand buffers are free) and added to the outputs as whole rows.
// Bound: fp32 MFMA.  Deterministic (fixed summation orders; the two-term LDS adds commute).
#include "cin_bwd.hpp"

typedef float cb_f16 __attribute__((ext_vector_type(16)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));

#define CB_THREADS 512
#define CB_BM 128
#define CB_BK 32
#define CB_LDA (CB_BM + 1)          // A tile is transposed on its way into LDS ([k][row]): odd row stride, 4-byte writes without conflicts
#define CB_LDB 128
#define CB_A_SZ (CB_BK * CB_LDA)
#define CB_B_SZ (CB_BK * CB_LDB)
#define CB_LR(ptr) (*(const volatile __attribute__((address_space(3))) float*)(ptr))

struct CinBwdK {
    const float* dXk;      // [M][Hk]
    const float* W;        // [Hk][F * HP]
    const float* x0t;      // [M][F]
    const float* Xp;       // [M][HP]
    float* dXp;            // [M][HP]  +=
    float* dx0t;           // [M][F]   +=   (may be the same buffer as dXp: layer 1, X_0 = x0)
    int Hk, F;
};

__device__ __forceinline__ void cb_lds_add(float* p, float v) {
    const unsigned a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)p);
    asm volatile("ds_add_f32 %0, %1" ::"v"(a), "v"(v) : "memory");
}

template <int HP>
__global__ void __launch_bounds__(CB_THREADS, 2)
k_cin_bwd_fused(const CinBwdK p) {
    constexpr int G = 128 / HP;                  // fields per column tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                      // 2 x CB_A_SZ
    float* const Bs = smem + 2 * CB_A_SZ;        // 2 x CB_B_SZ
    float* const x0s = Bs + 2 * CB_B_SZ;         // [F][128]
    float* const rs = x0s + p.F * CB_BM;         // [F][128] row sums of dx0
    float* const S = smem;                       // [128][HP] after the k-loop (the operand buffers are free then)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h5 = lane >> 5, l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * CB_BM;
    const int F = p.F, Hk = p.Hk, KT = Hk / CB_BK, NFT = F / G, NT = NFT * KT;
    const int64_t ldw = (int64_t)F * HP;

    // ---- one-time staging: x0 of the tile transposed into LDS, row sums zeroed, X_{k-1} of this lane's positions into registers
    for (int i = tid; i < CB_BM * (F / 4); i += CB_THREADS) {
        const int row = i % CB_BM, f4 = i / CB_BM;
        const cb_f4 v = *reinterpret_cast<const cb_f4*>(p.x0t + (m0 + row) * F + 4 * f4);
        x0s[(4 * f4 + 0) * CB_BM + row] = v.x;
        x0s[(4 * f4 + 1) * CB_BM + row] = v.y;
        x0s[(4 * f4 + 2) * CB_BM + row] = v.z;
        x0s[(4 * f4 + 3) * CB_BM + row] = v.w;
    }
    for (int i = tid; i < F * CB_BM; i += CB_THREADS) rs[i] = 0.f;
    const int hcol0 = (G == 1 ? 64 * wn : 0) + l31;          // this lane's column of X_{k-1} / dX_{k-1} for sub-tile j: hcol0 + 32 j
    float xp[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xp[j][r] = p.Xp[(m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h5) * HP + hcol0 + 32 * j];

    // ---- operand staging: 1024 float4 per tile and operand, two per thread
    int a_r[2], a_k[2], b_k[2], b_n[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * CB_THREADS;
        a_k[i] = (idx & 7) * 4;  a_r[i] = idx >> 3;          // A: [row][k], float4 along k
        b_n[i] = (idx & 31) * 4; b_k[i] = idx >> 5;          // B: [k][n], float4 along n
    }
    // Staging slots: 0, 1 = the thread's two float4 of the A tile, 2, 3 = of the B tile.  A slot of k-tile t + 1 goes registers -> LDS (`commit`)
    // and is at once requested again for k-tile t + 2 (`issue`); the four slots are spread over the MFMA steps of k-tile t (below), so that no
    // staging section stands between the barrier and the MFMAs with both waves of every SIMD in it (the sliced schedule of gemm_kernel.hpp).
    cb_f4 ra[2], rb[2];
    auto issue = [&](int slot, int ft, int kt) {
        if (slot < 2) ra[slot] = *reinterpret_cast<const cb_f4*>(p.dXk + (m0 + a_r[slot]) * Hk + kt * CB_BK + a_k[slot]);
        else rb[slot - 2] = *reinterpret_cast<const cb_f4*>(p.W + (int64_t)(kt * CB_BK + b_k[slot - 2]) * ldw + ft * 128 + b_n[slot - 2]);
    };
    auto commit = [&](int slot, int buf) {
        if (slot < 2) {
            float* a = As + buf * CB_A_SZ + a_k[slot] * CB_LDA + a_r[slot];
            a[0] = ra[slot].x; a[CB_LDA] = ra[slot].y; a[2 * CB_LDA] = ra[slot].z; a[3 * CB_LDA] = ra[slot].w;
        } else {
            *reinterpret_cast<cb_f4*>(Bs + buf * CB_B_SZ + b_k[slot - 2] * CB_LDB + b_n[slot - 2]) = rb[slot - 2];
        }
    };

    cb_f16 acc[2], dacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; dacc[j][r] = 0.f; }

    // (column tile, k-tile) of the k-tiles t (being computed) and t + 2 (being requested), advanced without divisions; indices past the last
    // k-tile are clamped to it (loads stay unconditional: the surplus copies land in the buffer nobody reads)
    auto advance = [&](int& ft, int& kt) {
        if (ft * KT + kt < NT - 1) {
            if (++kt == KT) { kt = 0; ++ft; }
        }
    };
    int ft = 0, kt = 0, ft2 = 0, kt2 = 0;
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) issue(slot, 0, 0);
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) commit(slot, 0);
    advance(ft2, kt2);                                       // k-tile 1
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) issue(slot, ft2, kt2);
    advance(ft2, kt2);                                       // k-tile 2
    const int a_off = h5 * CB_LDA + 32 * wm + l31;
    const int b_off = h5 * CB_LDB + 64 * wn + l31;
    for (int t = 0; t < NT; ++t) {
        const int cur = t & 1;
        __syncthreads();
        const float* as = As + cur * CB_A_SZ + a_off;
        const float* bs = Bs + cur * CB_B_SZ + b_off;
        float a0 = CB_LR(as), b00 = CB_LR(bs), b01 = CB_LR(bs + 32);
#pragma unroll
        for (int s = 0; s < CB_BK / 2; ++s) {
            const int sn = s + 1 < CB_BK / 2 ? s + 1 : s;
            const float a1 = CB_LR(as + 2 * sn * CB_LDA), b10 = CB_LR(bs + 2 * sn * CB_LDB), b11 = CB_LR(bs + 2 * sn * CB_LDB + 32);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b00, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b01, acc[1], 0, 0, 0);
            if (s % 4 == 1) {                                // compile-time (unrolled): one staging slot behind MFMA steps 1, 5, 9, 13
                commit(s / 4, cur ^ 1);
                issue(s / 4, ft2, kt2);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1; b00 = b10; b01 = b11;
        }
        advance(ft2, kt2);
        const bool tile_done = kt == KT - 1;
        const int ftc = ft;
        advance(ft, kt);
        if (tile_done) {                                     // block-uniform: the T tile of column tile ftc is complete
            const int f = G == 1 ? ftc : 2 * ftc + wn;
            float pr[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const cb_f4 xv = *reinterpret_cast<const cb_f4*>(x0s + f * CB_BM + 32 * wm + 8 * q + 4 * h5);
                const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * q + e;
                    dacc[0][r] += acc[0][r] * xr[e];
                    dacc[1][r] += acc[1][r] * xr[e];
                    pr[r] = acc[0][r] * xp[0][r] + acc[1][r] * xp[1][r];
                    acc[0][r] = 0.f;
                    acc[1][r] = 0.f;
                }
            }
            // sum over the 32 lanes that hold the same rows (they differ in the column): halving butterfly, the lane keeps the half of the
            // values its own lane-index bit selects and receives the partner's sums of that half
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int w = 8 >> st;                       // values kept after this step
                const bool up = (l31 >> (4 - st)) & 1;
#pragma unroll
                for (int i = 0; i < w; ++i) {
                    const float keep = up ? pr[i + w] : pr[i];
                    const float send = up ? pr[i] : pr[i + w];
                    pr[i] = keep + __shfl_xor(send, 16 >> st, 64);
                }
            }
            pr[0] += __shfl_xor(pr[0], 1, 64);
            // the lane now holds the sum of register index ri = (bit 4, bit 3, bit 2, bit 1 of its lane index)
            const int ri = ((l31 >> 4) & 1) * 8 + ((l31 >> 3) & 1) * 4 + ((l31 >> 2) & 1) * 2 + ((l31 >> 1) & 1);
            if ((l31 & 1) == 0) cb_lds_add(rs + f * CB_BM + 32 * wm + (ri & 3) + 8 * (ri >> 2) + 4 * h5, pr[0]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ds_add_f32 above are inline asm: the compiler's wait-count pass does not see them
    __syncthreads();                                         // every operand read and every row-sum add is done
    // ---- join the dX_{k-1} accumulators in LDS (S aliases the operand buffers), then whole-row updates of the outputs
    for (int i = tid; i < CB_BM * HP; i += CB_THREADS) S[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* dst = S + (32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h5) * HP + hcol0 + 32 * j;
            if (G == 1) *dst = dacc[j][r];                   // disjoint cells
            else cb_lds_add(dst, dacc[j][r]);                // two waves (field parities) per cell: a two-term sum, order-free
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const bool same = p.dXp == p.dx0t;                       // layer 1: X_0 = x0, both gradients land in dx0 (HP == F)
    for (int i = tid; i < CB_BM * (HP / 4); i += CB_THREADS) {
        const int row = i / (HP / 4), c4 = (i % (HP / 4)) * 4;
        cb_f4 v = *reinterpret_cast<const cb_f4*>(S + row * HP + c4);
        if (same) {
            v.x += rs[(c4 + 0) * CB_BM + row]; v.y += rs[(c4 + 1) * CB_BM + row];
            v.z += rs[(c4 + 2) * CB_BM + row]; v.w += rs[(c4 + 3) * CB_BM + row];
        }
        cb_f4* g = reinterpret_cast<cb_f4*>(p.dXp + (m0 + row) * HP + c4);
        *g = *g + v;
    }
    if (!same) {
        for (int i = tid; i < CB_BM * (F / 4); i += CB_THREADS) {
            const int row = i / (F / 4), c4 = (i % (F / 4)) * 4;
            cb_f4 v;
            v.x = rs[(c4 + 0) * CB_BM + row]; v.y = rs[(c4 + 1) * CB_BM + row];
            v.z = rs[(c4 + 2) * CB_BM + row]; v.w = rs[(c4 + 3) * CB_BM + row];
            cb_f4* g = reinterpret_cast<cb_f4*>(p.dx0t + (m0 + row) * F + c4);
            *g = *g + v;
        }
    }
}

static size_t cb_lds_bytes(int F) { return (size_t)(2 * CB_A_SZ + 2 * CB_B_SZ + 2 * F * CB_BM) * sizeof(float); }

bool rn_cin_bwd_fused_supported(int64_t M, int Hk, int Hp, int F) {
    if (M <= 0 || M % CB_BM || Hk < CB_BK || Hk % CB_BK || (Hp != 64 && Hp != 128)) return false;
    if (F < 4 || F % 4 || (F * Hp) % 128) return false;      // float4 rows of x0 / dx0; whole column tiles
    if ((size_t)CB_BM * Hp * sizeof(float) > (size_t)(2 * CB_A_SZ + 2 * CB_B_SZ) * sizeof(float)) return false;
    return cb_lds_bytes(F) <= 160 * 1024;
}

int rn_cin_bwd_fused(const float* dXk, const float* W, const float* x0t, const float* Xp, float* dXp, float* dx0t, int64_t M, int Hk, int Hp,
                     int F, hipStream_t st) {
    if (!rn_cin_bwd_fused_supported(M, Hk, Hp, F)) return RECNOW_EUNSUPPORTED;
    if (!dXk || !W || !x0t || !Xp || !dXp || !dx0t) return RECNOW_EINVAL;
    if (dXp == dx0t && Hp != F) return RECNOW_EINVAL;
    if ((((uintptr_t)dXk | (uintptr_t)W | (uintptr_t)x0t | (uintptr_t)dXp | (uintptr_t)dx0t) & 15) != 0) return RECNOW_EUNSUPPORTED;
    CinBwdK k;
    k.dXk = dXk; k.W = W; k.x0t = x0t; k.Xp = Xp; k.dXp = dXp; k.dx0t = dx0t; k.Hk = Hk; k.F = F;
    const size_t lds = cb_lds_bytes(F);
    const int grid = (int)(M / CB_BM);
    // more than 64 KB of dynamic LDS: raised once per device and kernel (the attribute belongs to the function on that device)
    static bool raised[2][64];
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    const int which = Hp == 128 ? 0 : 1;
    if (dev < 0 || dev >= 64 || !raised[which][dev]) {
        if (Hp == 128) RN_HIP(hipFuncSetAttribute((const void*)k_cin_bwd_fused<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        else RN_HIP(hipFuncSetAttribute((const void*)k_cin_bwd_fused<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (dev >= 0 && dev < 64) raised[which][dev] = true;
    }
    if (Hp == 128) hipLaunchKernelGGL((k_cin_bwd_fused<128>), grid, CB_THREADS, lds, st, k);
    else hipLaunchKernelGGL((k_cin_bwd_fused<64>), grid, CB_THREADS, lds, st, k);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
