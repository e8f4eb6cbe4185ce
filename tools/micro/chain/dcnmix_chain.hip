// Chained products of DCNMixLayer, exact fp32 on v_mfma_f32_32x32x2_f32 (reference rec_now/layers/dcn_mix_layer.py:141-150 of
// layer l followed by :135-136 of layer l + 1).
//
// Between two cross layers the step runs  out = x0 * (T2g [W; b])  (K = N*S + N = 130 deep, 4096 output tiles, HBM-bound: it
// streams x0 in and out / O out) and then  T1' = act(out [U' | K'])  (K = D deep, MFMA-bound, reads `out` back).  As two launches
// the first leaves the MFMA pipe 40 % idle and the second leaves HBM idle.  Here one workgroup owns a block of 128 batch rows
// and walks the D columns in tiles of 64; per column tile it
//   P1  forms the 64 columns of O for its rows, TRANSPOSED (M = columns, N = rows): a lane's accumulator registers then hold
//       O[row = lane & 31][4 consecutive columns] x 8,
//   E1  multiplies by x0 (loaded in that very layout: float4 per lane), stores out (and O), adds the tile's share of the
//       gate logits out . K' on the VALU, and
//   P2  feeds the products straight back as the A operand of the second product: the accumulator register that holds columns
//       (c, c + 4) of a row is the A fragment of the 32x32x2 MFMA for the k-pair (c, c + 4) -- `out` never comes back from
//       memory and never passes through LDS.
// Two workgroups per CU, each wave owns 32 rows: acc(P1) 32 + acc(P2) 64 registers; x0 / out / O move in row layout (whole cache
// lines) and change layout through a wave-private XOR-swizzled LDS tile (in the accumulator layout a wave instruction touches 32
// rows x 32 B: the streams ran at 1.8 TB/s that way).
// Operand staging: the shared operands (a 32 x 64 piece of [W; b], a 16 x 128 piece of [U' | K']) are 8 KB tiles that go
// global -> registers -> LDS two steps ahead (two register sets, two LDS buffers, one barrier per step of 32 MFMAs per wave);
// the private operand (the wave's 32 rows of T2g) is loaded straight into B fragments: with the k-pairs of a 32-deep step
// chosen as (p, p + 16) a lane needs 16 consecutive floats of its row.
//
// STATUS: correct (tests/test_chain_gpu.py), OFF by default (RECNOW_CHAIN=1 switches it on from 65 536 rows).  Measured at B = 65 536,
// D = 1024 (tools/micro/chain_probe.py, in-kernel s_memtime / s_memrealtime stamps): 420-437 us per launch against 155 + 204 us for
// the two launches it replaces.  Per SIMD the two resident waves issue 8256 MFMAs = 528 k cycles; the kernel takes 616 k cycles (the
// pipe is 86 % busy) -- but at 1.83 GHz: with the HBM streams running beside the MFMAs the chip holds 1.83 GHz, against 2.08 GHz for the
// same kernel with its loads and stores compiled out (321 us) and 2.03 GHz inside k_gemm.  The step's products are bound by the clock
// the power budget allows, not by idle phases one could fill: overlapping the HBM-bound product with the MFMA-bound one trades pipe
// utilisation for clock and loses ~20 %.  DESIGN.md 5g.
#include "dcnmix_chain.hpp"
#include "gemm_kernel.hpp"
#include "prof.hpp"

#define CH_THREADS 256
#define CH_CT 64              // columns per column tile
#define CH_TILE 2048          // floats per staged operand tile (8 KB)

struct ChainF {
    const float *T2g, *Wc2, *x0, *Wc1, *gate;
    float *out, *O, *T1;
    int64_t ldt;              // row stride of T2g and T1 (LDT)
    int D, nrb, act_inner;
};

template <bool HAS_O>
__global__ void __launch_bounds__(CH_THREADS, 2)
k_mix_chain_fwd(const ChainF p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [2][CH_TILE] operand tiles | [4][2048] layout-change tiles | gate weights [D][2]
    float* stg = smem + 2 * CH_TILE + (threadIdx.x >> 6) * 2048;       // layout-change tile of this wave (8 KB)
    float* gl = smem + 2 * CH_TILE + 4 * 2048;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
    const int D = p.D, nct = D / CH_CT;
    const int64_t ldt = p.ldt;
    for (int i = threadIdx.x; i < D * 2 / 4; i += CH_THREADS) reinterpret_cast<f32x4*>(gl)[i] = reinterpret_cast<const f32x4*>(p.gate)[i];

    // ---- lane-invariant offsets (bytes, 32-bit; the host checks the ranges)
    // staged tiles: this thread's first float4 slot (the second one is 16 / 8 k-rows further: a uniform add to the base)
    unsigned o1 = (unsigned)(((threadIdx.x >> 4) * D + (threadIdx.x & 15) * 4) * 4);          // P1 tile: 32 k-rows x 64 columns of [W; b]
    unsigned o2 = (unsigned)(((threadIdx.x >> 5) * (int)ldt + (threadIdx.x & 31) * 4) * 4);   // P2 tile: 16 k-rows x 128 columns of [U' | K'] (row stride LDT)
    const unsigned row_w = (unsigned)(wave * 32 + l31);                 // this lane's row inside the block
    unsigned ot = (unsigned)((row_w * (unsigned)ldt + 16 * h) * 4);         // T2g fragments
    // x0 / out / O move in ROW layout (a wave instruction = 4 rows x 256 B: whole cache lines; the accumulator layout's 32 rows x 32 B
    // per instruction ran the step's streams at 1.8 TB/s) and change layout through a wave-private, XOR-swizzled LDS tile
    unsigned oxr = (unsigned)((((unsigned)wave * 32 + (lane >> 4)) * (unsigned)D + (lane & 15) * 4) * 4);      // row 4i + (lane >> 4): + i * 16 D bytes
    unsigned ot_tail = (unsigned)((row_w * (unsigned)ldt + h) * 4);                // T2g[row][128 + h] (the gate values: tail k-pair)
    unsigned ow = (unsigned)((h * D + l31) * 4);                            // tail rows of [W; b] (k = 128 + h)
    unsigned o_t1 = (unsigned)(((wave * 32 + 4 * h) * (unsigned)ldt + l31) * 4);      // T1' stores: row 4h of the wave's block, column l31
    unsigned rbase = (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4));      // layout-change tile, row layout: row 4i + (lane >> 4), slot lane & 15
    unsigned abase = (unsigned)(l31 * 256 + (((l31 & 15) ^ h) << 4));                           // accumulator layout: row l31, slot 8 cb + 2 q + h
    const float* as_l = smem + 16 * h * 64 + l31;                       // P1 A fragments: + buf * CH_TILE + p * 64 + cb * 32
    const float* bs_l = smem + 4 * h * 128 + l31;                       // P2 B fragments: + buf * CH_TILE + (8 * (i >> 2) + c) * 128 + sb * 32
    const float* gl_l = gl + 8 * h;                                     // gate weights of columns (.. + 4h + 0..3): + (ct * 64 + cb * 32 + 8q) * 2

    f32x4 rt[2][2];           // staged-tile register sets (tile g lives in set g & 1)
    f32x4 tq[2][4];           // T2g fragments of P1 step kt (set kt & 1)
    f32x4 x0r[4];             // x0 in flight: half a column tile (16 of the wave's rows), see ld_x0
    float wt[2], tt;          // tail k-pair (128, 129): [W; b] rows and the T2g (gate) columns
    f32x16 acc1[2], acc2[4];
    float gp0 = 0.f, gp1 = 0.f;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[sb][r] = 0.f;

    auto ld_tile1 = [&](int set, int ct, int kt) {
        const char* __restrict__ b = reinterpret_cast<const char*>(p.Wc2 + (int64_t)kt * 32 * D + ct * CH_CT);
#pragma unroll
        for (int i = 0; i < 2; ++i) { asm volatile("" : "+v"(o1)); rt[set][i] = *reinterpret_cast<const f32x4*>(b + (int64_t)i * 16 * D * 4 + o1); }
    };
    auto ld_tile2 = [&](int set, int ct, int t2) {
        const char* __restrict__ b = reinterpret_cast<const char*>(p.Wc1 + (int64_t)(ct * CH_CT + t2 * 16) * ldt);
#pragma unroll
        for (int i = 0; i < 2; ++i) { asm volatile("" : "+v"(o2)); rt[set][i] = *reinterpret_cast<const f32x4*>(b + (int64_t)i * 8 * ldt * 4 + o2); }
    };
    auto st_tile = [&](int set, int buf) {
        float* s = smem + buf * CH_TILE + threadIdx.x * 4;
        *reinterpret_cast<f32x4*>(s) = rt[set][0];
        *reinterpret_cast<f32x4*>(s + CH_THREADS * 4) = rt[set][1];
    };
    auto ld_t2g = [&](int set, int rb, int kt) {
        const char* __restrict__ b = reinterpret_cast<const char*>(p.T2g + (int64_t)rb * 128 * ldt + kt * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(ot)); tq[set][j] = *reinterpret_cast<const f32x4*>(b + ot + j * 16); }
    };
    // x0 of a column tile goes HBM -> registers (row layout) -> layout-change tile in two halves of 16 rows through the same 16 registers
    auto ld_x0 = [&](int rb, int ct, int half) {
        const char* __restrict__ b = reinterpret_cast<const char*>(p.x0 + (int64_t)rb * 128 * D + ct * CH_CT);
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(oxr)); x0r[i] = *reinterpret_cast<const f32x4*>(b + (int64_t)(4 * half + i) * 16 * D + oxr); }
    };
    auto st_x0 = [&](int half) {
        asm volatile("" : "+v"(rbase));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(stg) + ((rbase ^ (i << 6)) + (4 * half + i) * 1024)) = x0r[i];
    };
    auto ld_tail = [&](int rb, int ct) {
        const char* __restrict__ b = reinterpret_cast<const char*>(p.Wc2 + (int64_t)128 * D + ct * CH_CT);
        asm volatile("" : "+v"(ow), "+v"(ot_tail));
        wt[0] = *reinterpret_cast<const float*>(b + ow);
        wt[1] = *reinterpret_cast<const float*>(b + ow + 32 * 4);
        tt = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.T2g + (int64_t)rb * 128 * ldt + 128) + ot_tail);
    };
    // P1 step kt: 16 k-pairs (p, p + 16) x 2 column blocks
    auto p1_step = [&](int buf, int set, bool first) {
        const float* as = as_l + buf * CH_TILE;
        float a0[2][2], a1[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) a0[u][cb] = FR(as + u * 64 + cb * 32);
#pragma unroll
        for (int pp = 0; pp < 16; pp += 4) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) a1[u][cb] = FR(as + (pp + 2 + u) * 64 + cb * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float b = tq[set][(pp + u) >> 2][(pp + u) & 3];
                    if (first && pp + u == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][cb], b, z, 0, 0, 0);
                    } else {
                        acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][cb], b, acc1[cb], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if (pp + 4 < 16) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) a0[u][cb] = FR(as + (pp + 4 + u) * 64 + cb * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float b = tq[set][(pp + 2 + u) >> 2][(pp + 2 + u) & 3];
                    acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][cb], b, acc1[cb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // P2 step t2: the 16 columns 16 * t2 .. + 15 of the column tile = 8 accumulator registers of block t2 >> 1 as A fragments
    auto p2_step = [&](int buf, int t2) {
        const float* bs = bs_l + buf * CH_TILE;
        const int cb = t2 >> 1, e = t2 & 1;
        float b0[4], b1[4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) b0[sb] = FR(bs + sb * 32);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) b1[sb] = FR(bs + (8 * ((i + 1) >> 2) + ((i + 1) & 3)) * 128 + sb * 32);
            __builtin_amdgcn_sched_barrier(0);
            {
                const float a = acc1[cb][4 * (2 * e + (i >> 2)) + (i & 3)];
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) acc2[sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[sb], acc2[sb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < 8) {
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) b0[sb] = FR(bs + (8 * ((i + 2) >> 2) + ((i + 2) & 3)) * 128 + sb * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float a = acc1[cb][4 * (2 * e + ((i + 1) >> 2)) + ((i + 1) & 3)];
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) acc2[sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[sb], acc2[sb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int rb = blockIdx.x, ct = 0;
    if (rb >= p.nrb) return;
    // ---- prologue: tiles 0 and 1, T2g fragments of steps 0 and 1, x0 and the tail pair of the first column tile
    ld_tile1(0, 0, 0);
    ld_tile1(1, 0, 1);
    ld_t2g(0, rb, 0);
    ld_t2g(1, rb, 1);
    ld_x0(rb, 0, 0);
    st_x0(0);
    ld_x0(rb, 0, 1);
    ld_tail(rb, 0);
    st_tile(0, 0);
    __syncthreads();
    for (;;) {
        int ctn = ct + 1, rbn = rb;
        if (ctn == nct) {
            ctn = 0;
            rbn = rb + gridDim.x < p.nrb ? rb + gridDim.x : rb;      // the last item prefetches its own first tile again (loads stay unconditional)
        }
        // ---- P1: steps g = 0..3
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < 2) ld_tile1(kt & 1, ct, kt + 2);
            else ld_tile2(kt & 1, ct, kt - 2);
            if (kt == 0) st_x0(1);      // second half of this column tile's x0 (requested two steps ago): registers -> layout-change tile
            if (kt == 1) ld_t2g(0, rb, 2);
            if (kt == 2) ld_t2g(1, rb, 3);
            __builtin_amdgcn_sched_barrier(0);
            p1_step(kt & 1, kt & 1, kt == 0);
            st_tile((kt + 1) & 1, (kt + 1) & 1);
            __syncthreads();
        }
        // ---- E1: tail k-pair, times x0, stores, gate logits; the products stay in acc1 as P2's A fragments
        {
            // layout-change tile of this wave: [32 rows][16 float4 slots], slot (r, c4) holds columns 4 (c4 ^ (r & 15)) .. + 3 of row r.
            // Byte offsets: one lane-dependent base per layout, XORed with a constant per access (recomputed here: hoisted out of the
            // loop the 12 variants cost 12 registers the loop does not have)
            asm volatile("" : "+v"(rbase), "+v"(abase));
            // the loads of P2's first step go out BEFORE this block's stores: a counted wait on a load also waits for every older store
            ld_tile2(0, ct, 2);
            ld_x0(rbn, ctn, 0);
            __builtin_amdgcn_sched_barrier(0);
            auto row_at = [&](int i) { return reinterpret_cast<char*>(stg) + ((rbase ^ ((i & 3) << 6)) + i * 1024); };
            auto acc_at = [&](int cb, int q) { return reinterpret_cast<char*>(stg) + (abase ^ ((cb * 8 + 2 * q) << 4)); };
            f32x4 xa[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xa[j] = *reinterpret_cast<const f32x4*>(acc_at(j >> 2, j & 3));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc1[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[cb], tt, acc1[cb], 0, 0, 0);
            RN_LDS_WAVE_SYNC();
            char* __restrict__ ob = reinterpret_cast<char*>(p.out + (int64_t)rb * 128 * D + ct * CH_CT);
            char* __restrict__ Ob = HAS_O ? reinterpret_cast<char*>(p.O + (int64_t)rb * 128 * D + ct * CH_CT) : nullptr;
            if (HAS_O) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<f32x4*>(acc_at(j >> 2, j & 3)) =
                        mk4(acc1[j >> 2][4 * (j & 3)], acc1[j >> 2][4 * (j & 3) + 1], acc1[j >> 2][4 * (j & 3) + 2], acc1[j >> 2][4 * (j & 3) + 3]);
                RN_LDS_WAVE_SYNC();
                f32x4 t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const f32x4*>(row_at(i));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("" : "+v"(oxr));
                    *reinterpret_cast<f32x4*>(Ob + (int64_t)i * 16 * D + oxr) = t[i];
                }
                RN_LDS_WAVE_SYNC();
            }
            const float* gk = gl_l + ct * CH_CT * 2;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 o = mk4(acc1[cb][4 * q], acc1[cb][4 * q + 1], acc1[cb][4 * q + 2], acc1[cb][4 * q + 3]);
                    const f32x4 v = o * xa[cb * 4 + q];
                    *reinterpret_cast<f32x4*>(acc_at(cb, q)) = v;
                    acc1[cb][4 * q] = v.x; acc1[cb][4 * q + 1] = v.y; acc1[cb][4 * q + 2] = v.z; acc1[cb][4 * q + 3] = v.w;
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(gk + (cb * 32 + q * 8) * 2);
                    const f32x4 k1 = *reinterpret_cast<const f32x4*>(gk + (cb * 32 + q * 8) * 2 + 4);
                    gp0 += v.x * k0.x + v.y * k0.z + v.z * k1.x + v.w * k1.z;
                    gp1 += v.x * k0.y + v.y * k0.w + v.z * k1.y + v.w * k1.w;
                }
            RN_LDS_WAVE_SYNC();
            {
                f32x4 t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const f32x4*>(row_at(i));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("" : "+v"(oxr));
                    *reinterpret_cast<f32x4*>(ob + (int64_t)i * 16 * D + oxr) = t[i];
                }
            }
            RN_LDS_WAVE_SYNC();
        }
        // ---- P2: steps g = 4..7
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            if (t2 == 1) ld_tile2(1, ct, 3);
            if (t2 >= 2) ld_tile1(t2 & 1, ctn, t2 - 2);
            if (t2 == 2) {
                st_x0(0);
                ld_x0(rbn, ctn, 1);
            }
            if (t2 == 1) ld_tail(rbn, ctn);
            if (t2 == 2) ld_t2g(0, rbn, 0);
            if (t2 == 3) ld_t2g(1, rbn, 1);
            __builtin_amdgcn_sched_barrier(0);
            p2_step(t2 & 1, t2);
            st_tile((t2 + 1) & 1, (t2 + 1) & 1);
            __syncthreads();
        }
        if (ct == nct - 1) {
            // ---- E2: T1' rows of this block: act_inner on the 128 sub-space columns, raw gate logits behind them
            char* __restrict__ tb = reinterpret_cast<char*>(p.T1 + (int64_t)rb * 128 * ldt);
            auto store_t1 = [&](auto act) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    char* __restrict__ tr = tb + (int64_t)(8 * (r >> 2) + (r & 3)) * ldt * 4;      // (uniform) + lane offset + immediate
#pragma unroll
                    for (int sb = 0; sb < 4; ++sb) {
                        asm volatile("" : "+v"(o_t1));
                        *reinterpret_cast<float*>(tr + o_t1 + sb * 128) = act(acc2[sb][r]);
                        acc2[sb][r] = 0.f;
                    }
                }
            };
            if (p.act_inner == RECNOW_ACT_TANH) store_t1([](float v) { return rn_tanh(v); });
            else if (p.act_inner == RECNOW_ACT_RELU) store_t1([](float v) { return v > 0.f ? v : 0.f; });
            else if (p.act_inner == RECNOW_ACT_SIGMOID) store_t1([](float v) { return 1.f / (1.f + expf(-v)); });
            else store_t1([](float v) { return v; });
            gp0 += __shfl_xor(gp0, 32, 64);
            gp1 += __shfl_xor(gp1, 32, 64);
            if (h == 0) {
                f32x2 gv;
                gv.x = gp0; gv.y = gp1;
                *reinterpret_cast<f32x2*>(tb + ot + 128 * 4) = gv;          // (h == 0: ot is column 0 of this lane's row)
            }
            gp0 = gp1 = 0.f;
            if (rb + (int)gridDim.x >= p.nrb) break;
        }
        rb = rbn;
        ct = ctn;
    }
}

bool rn_mix_chain_fwd_supported(int64_t B, int D, int S, int N, int LDT) {
    return N == 2 && S == 64 && LDT == 144 && B > 0 && B % 128 == 0 && D % CH_CT == 0 && D >= 128 && D <= 4096 &&
           (int64_t)128 * D * 4 < (1ll << 31) && (int64_t)128 * LDT * 4 < (1ll << 31);
}

int rn_mix_chain_fwd(const float* T2g, const float* Wc2, const float* x0, float* out, float* O, const float* Wc1_next,
                     const float* gate_next, float* T1_next, int64_t B, int D, int LDT, int act_inner, hipStream_t st) {
    if (!rn_mix_chain_fwd_supported(B, D, 64, 2, LDT)) return RECNOW_EUNSUPPORTED;
    if (!T2g || !Wc2 || !x0 || !out || !Wc1_next || !gate_next || !T1_next) return RECNOW_EINVAL;
    ChainF p;
    p.T2g = T2g; p.Wc2 = Wc2; p.x0 = x0; p.Wc1 = Wc1_next; p.gate = gate_next;
    p.out = out; p.O = O; p.T1 = T1_next;
    p.ldt = LDT; p.D = D; p.nrb = (int)(B / 128); p.act_inner = act_inner;
    const size_t lds = (2 * CH_TILE + 4 * 2048 + (size_t)D * 2) * sizeof(float);
    const int grid = p.nrb < 512 ? p.nrb : 512;
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_CHAIN, 4.0 * B * D * 130.0, 4.0 * B * (2.0 * LDT + (O ? 3.0 : 2.0) * D), st) : nullptr;
    if (lds > 64 * 1024) {          // D > 2048: more than the default limit of dynamic LDS per workgroup
        if (O) RN_HIP(hipFuncSetAttribute((const void*)k_mix_chain_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else RN_HIP(hipFuncSetAttribute((const void*)k_mix_chain_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (O) hipLaunchKernelGGL((k_mix_chain_fwd<true>), grid, CH_THREADS, lds, st, p);
    else hipLaunchKernelGGL((k_mix_chain_fwd<false>), grid, CH_THREADS, lds, st, p);
    rn_prof_end(pr, st);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// diagnostics entry (tools/micro/chain_probe.py): the forward chain kernel alone
extern "C" int recnow_dbg_mix_chain_fwd(const float* T2g, const float* Wc2, const float* x0, float* out, float* O, const float* Wc1_next,
                                        const float* gate_next, float* T1_next, int64_t B, int D, int LDT, int act_inner, void* stream) {
    return rn_mix_chain_fwd(T2g, Wc2, x0, out, O, Wc1_next, gate_next, T1_next, B, D, LDT, act_inner, (hipStream_t)stream);
}
