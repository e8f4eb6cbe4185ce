// Exact-fp32 MFMA GEMM for gfx950: C = epilogue(opA(A) x opB(B)), batched, optional split-K.
//
// Why this shape: the reference's contractions (multi_dense_layer.py:90, dcn_mix_layer.py:135-141, MMoE/PLE experts)
// must match TF's fp32 matmul to 1e-5 relative, so bf16 MFMA is out; gfx950's v_mfma_f32_32x32x2_f32 is an exact fp32
// fma chain at 64 FLOP/clk/SIMD (157 TFLOP/s).  At that rate a 128x128x32 block tile needs 16 KB + 16 KB of LDS per
// 4096 MFMA-cycles per wave, so LDS bandwidth and even ds_read_b32 fragment loads are far from binding; what matters
// is (1) fragment reads that are bank-conflict-free, (2) global loads of the next k-tile in flight under the MFMAs of
// the current one (register-staged double buffering, one barrier per k-tile), (3) enough workgroups (split-K for the
// K = batch-rows weight-gradient GEMMs, reduced deterministically from slabs -- no float atomics).
//
// LDS images are k-major: As[k][m], Bs[k][n].  A wave64 MFMA 32x32x2 fragment read is then 32 consecutive floats for
// lanes 0-31 (k) and 32 consecutive floats for lanes 32-63 (k+1): conflict-free ds_read_b32 with no padding tricks.
// An operand that is k-contiguous in memory ([m][k]) is loaded as float4 along k (coalesced 128-B rows) and transposed
// on the LDS write; its row stride is ROWS+1 floats so those 4-B writes are conflict-free too.
#include <atomic>
#include "gemm_kernel.hpp"
#include "prof.hpp"

// sum the split-K slabs in slice order (deterministic) and apply the epilogue.  float4 per thread along N (the slabs
// are [M][N] with N % 4 == 0 whenever VEC), four slabs in flight per thread.
// Deterministic slab reduction of the split-K products.  QUAD: four adjacent lanes share one float4 of the output, each sums a
// quarter of the slabs (a contiguous range, in order), and the four partial sums are combined as (q0 + q1) + (q2 + q3): four times
// the threads and loads in flight -- an output of 1024 x 132 floats is 33 792 float4, i.e. 132 workgroups of threads that each walk
// 64 slabs with four loads in flight (10.5 us for 35 MB that the product has just left in the memory-side cache).
template <bool VEC, bool QUAD>
__device__ __forceinline__ void splitk_reduce_body(const GemmK& p, const int bid, const int nblk) {
    constexpr int W = VEC ? 4 : 1;
    constexpr int NQ = QUAD ? 4 : 1;
    const int64_t MN = (int64_t)p.M * p.npart;          // one slab
    const int64_t total = MN * p.batch / W;
    const int q = QUAD ? (threadIdx.x & 3) : 0;
    const int per = (p.splitk + NQ - 1) / NQ, k_lo = q * per, k_hi = min(p.splitk, k_lo + per);
    // QUAD: every lane of a quad runs the loop for the quad's element (whole quads are in or out: total is rounded up by the caller's grid)
    for (int64_t i = ((int64_t)bid * 256 + threadIdx.x) / NQ; i < total; i += (int64_t)nblk * 256 / NQ) {
        const int64_t e0 = i * W;
        const int b = (int)(e0 / MN);
        const int64_t mn = e0 % MN;
        const int row = (int)(mn / p.npart), col = (int)(mn % p.npart);
        const float* P = p.partial + ((int64_t)b * p.splitk) * MN + mn;
        // the slabs are summed in fp64 (the kernel is bound by reading them; 64 double adds per output are free): the K = B weight
        // gradients then carry only the fp32 accumulation INSIDE a slab (K / splitk deep), not another 64-term fp32 sum on top --
        // at K = 32 768 the full-size PLE parity test sat at 1.0e-5 of its bound with fp32 slab sums
        double s[W];
#pragma unroll
        for (int e = 0; e < W; ++e) s[e] = 0.0;
        int k = k_lo;
        if (VEC) {
            // eight slabs in flight per lane (the loop is bound by the latency of its loads: 4 in flight read 2.7 TB/s); the sums keep
            // the sequential order
            for (; k + 8 <= k_hi; k += 8) {
                float4 a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(P + (int64_t)(k + u) * MN);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s[0] += a[u].x;
                    s[1 % W] += a[u].y;
                    s[2 % W] += a[u].z;
                    s[3 % W] += a[u].w;
                }
            }
            for (; k + 4 <= k_hi; k += 4) {
                const float4 a0 = *reinterpret_cast<const float4*>(P + (int64_t)(k + 0) * MN);
                const float4 a1 = *reinterpret_cast<const float4*>(P + (int64_t)(k + 1) * MN);
                const float4 a2 = *reinterpret_cast<const float4*>(P + (int64_t)(k + 2) * MN);
                const float4 a3 = *reinterpret_cast<const float4*>(P + (int64_t)(k + 3) * MN);
                s[0] = (((s[0] + a0.x) + a1.x) + a2.x) + a3.x;
                s[1 % W] = (((s[1 % W] + a0.y) + a1.y) + a2.y) + a3.y;
                s[2 % W] = (((s[2 % W] + a0.z) + a1.z) + a2.z) + a3.z;
                s[3 % W] = (((s[3 % W] + a0.w) + a1.w) + a2.w) + a3.w;
            }
        }
        for (; k < k_hi; ++k) {
#pragma unroll
            for (int e = 0; e < W; ++e) s[e] += P[(int64_t)k * MN + e];
        }
        if (QUAD) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                s[e] += __shfl_xor(s[e], 1, 64);         // q0 + q1 | q2 + q3 (both lanes of a pair hold the same sum)
                s[e] += __shfl_xor(s[e], 2, 64);         // (q0 + q1) + (q2 + q3)
            }
            if (q != 0) continue;
        }
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float v = (float)s[e];
            const int c = col + e;
            if (c >= p.N) {                      // side-product columns ride behind the N main columns of a slab row
                if (c - p.N < p.sp_r) p.cx[(int64_t)row * p.cx_ms + (int64_t)(c - p.N) * p.cx_rs] = v;
                continue;
            }
            if (p.bias) v += p.bias[(int64_t)b * p.sBias + c];
            if (c < p.act_cols) v = rn_act(v, p.act);
            if (p.emul) {
                const float ev = p.emul[(int64_t)b * p.sE + (int64_t)row * p.lde + c];
                v *= (p.e_mode == RECNOW_OPMODE_ACTGRAD) ? rn_act_grad_from_out(ev, p.e_act) : ev;
            }
            float* dst = p.C + (int64_t)b * p.sC + (p.c_trans ? ((int64_t)c * p.ldc + row) : ((int64_t)row * p.ldc + c));
            if (p.perm_s > 0) dst = p.C + ((int64_t)(c / p.perm_s) * p.M + row) * p.perm_s + (c % p.perm_s);      // [N/s][M][s] layout
            if (p.accumulate) v += *dst;
            *dst = v;
        }
    }
}
template <bool VEC, bool QUAD>
__global__ void __launch_bounds__(256)
k_gemm_splitk_reduce(const GemmK p) {
    splitk_reduce_body<VEC, QUAD>(p, blockIdx.x, gridDim.x);
}

// One launch at the end of a DCN-v2 cross layer's backward: the slab reductions of its two K = B weight-gradient products (dW with
// dbias riding as side columns, dU with dgate) and the sum of the sub-space kernel's per-workgroup dV partials -- three launches of a
// few microseconds each otherwise.  Blocks [0, na) run reduction a, [na, na + nb) reduction b, the rest the dV sum (64 columns per
// block: 4 strided groups of partials, fixed order, then a 4-term LDS sum).
__device__ __forceinline__ void reduce_variant(const GemmK& p, int variant, int bid, int nblk) {
    if (variant == 2) splitk_reduce_body<true, true>(p, bid, nblk);
    else if (variant == 1) splitk_reduce_body<true, false>(p, bid, nblk);
    else splitk_reduce_body<false, false>(p, bid, nblk);
}
__global__ void __launch_bounds__(256)
k_layer_end_reduce(const GemmK pa, int va, int na, const GemmK pb, int vb, int nb, const float* __restrict__ dv_part, int dv_nparts, int dv_total,
                   float* __restrict__ dV) {
    const int bid = blockIdx.x;
    if (bid < na) { reduce_variant(pa, va, bid, na); return; }
    if (bid < na + nb) { reduce_variant(pb, vb, bid - na, nb); return; }
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = (bid - na - nb) * 64 + e;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < dv_total) {
        int g = q;
        for (; g + 4 * 7 < dv_nparts; g += 4 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += dv_part[(int64_t)(g + 4 * u) * dv_total + i];
        }
        for (; g < dv_nparts; g += 4) s[0] += dv_part[(int64_t)g * dv_total + i];
    }
    red[q][e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (q == 0 && i < dv_total) dV[i] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// ---- host dispatch -------------------------------------------------------------------------------------
// (tile family, small-M rule and K split: gemm_dispatch.hpp, host-only so that it also builds under the CPU sanitizers)
#include "gemm_dispatch.hpp"
#include "gemm_split.hpp"
int rn_gemm_precision();
static inline RnDispatchEnv dispatch_env() {
    static const int bm64 = []() { const char* e = getenv("RECNOW_GEMM_BM64"); return e ? atoi(e) : 256; }();        // A/B: 0 = off, N = tile bound
    static const int kb = []() { const char* e = getenv("RECNOW_GEMM_BM64_KB"); return e ? atoi(e) : -1; }();       // A/B: 0 = only to fill the chip, 1 = always
    return RnDispatchEnv{bm64, kb, rn_gemm_precision()};
}
static inline GemmCfg pick_cfg(const recnow_gemm_desc* d) { return pick_cfg(d, dispatch_env()); }

static std::atomic<int> g_gemm_staging{[]() { const char* e = getenv("RECNOW_GEMM_GLDS"); return (e && e[0] == '1') ? 1 : 0; }()};      // recnow_set_gemm_staging
static int g_gemm_precision = []() {
    const char* e = getenv("RECNOW_GEMM_PRECISION");
    return (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) ? 1 : 0;
}();
int rn_gemm_precision() { return g_gemm_precision; }
int rn_gemm_set_precision(int mode) {
    if (mode != 0 && mode != 1) return RECNOW_EINVAL;
    g_gemm_precision = mode;
    return RECNOW_OK;
}

// products whose B operand the split-precision kernel splits once per launch into global planes (weights: small, L2-resident)
// (round 6: also the (K, 128) activation operand of the K = B weight-gradient products, [k][n] rows of ldb floats -- each of the eight row-tile
// workgroups that read a k-tile of it split it again before)
static inline bool split_planes_shape(const recnow_gemm_desc* d) {
    static const bool kb = []() { const char* e = getenv("RECNOW_SPLIT_LEAN"); return e && e[0] == '2'; }();      // A/B switch, see rn_gemm_launch_split
    return d->sp_r > 0 && d->N == 128 && d->K % 16 == 0 && d->batch == 1 && ((!d->a_trans && d->K <= 4096) || (kb && d->a_trans && !d->b_trans));
}

// short-K products whose packed weights the split-precision short-K kernel splits into planes (K = 144: the ring schedule)
static inline bool shortk_planes_shape(const recnow_gemm_desc* d) {
    return d->K == 144 && d->N % 128 == 0 && d->M % 128 == 0 && d->batch == 1 && !d->a_trans && d->sp_r == 0 && d->eu_r == 0;
}
static inline void tag_used_split_shortk() {}

// workgroup slots the K split of the NEXT launches of this host thread aims at (0 = the default); see pick_split
static thread_local int g_split_slots = 0;
void rn_gemm_split_slots(int slots) { g_split_slots = slots; }

size_t rn_gemm_ws_bytes(const recnow_gemm_desc* d) {
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return 0;
    const GemmCfg c = pick_cfg(d);
    size_t b = rnd_slab_bytes_any(d, c, 256);
    if (split_planes_shape(d)) b += rn_gemm_split_planes_bytes(d->K, d->N);      // whatever the precision mode is when the product runs
    if (shortk_planes_shape(d) && b < rn_gemm_shortk_planes_bytes(d->K, d->N)) b = rn_gemm_shortk_planes_bytes(d->K, d->N);
    return b;
}

static inline bool host_aligned(const void* p, int64_t ld, int64_t sb) {
    return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0) && (sb % 4 == 0);
}
// the lean (EDGE = false) kernel needs every tile in bounds and every float4 aligned
static bool gemm_interior(const recnow_gemm_desc* d, const GemmCfg& c, int bk, int kchunk, bool split) {
    if (d->M % c.BM || d->N % c.BN || d->K % bk || kchunk % bk) return false;
    if ((int64_t)256 * d->lda >= (1ll << 30) || (int64_t)256 * d->ldb >= (1ll << 30)) return false;      // 32-bit byte offsets inside a tile
    if (!host_aligned(d->A, d->lda, d->a_batch_stride) || !host_aligned(d->B, d->ldb, d->b_batch_stride)) return false;
    if (d->a_mode == RECNOW_OPMODE_OUTER) { if (d->a_hq % 4) return false; }
    else if (d->a_mode != RECNOW_OPMODE_NONE && !host_aligned(d->A2, d->lda, d->a_batch_stride)) return false;
    if (d->b_mode == RECNOW_OPMODE_OUTER) { if (d->b_hq % 4) return false; }
    else if (d->b_mode != RECNOW_OPMODE_NONE && !host_aligned(d->B2, d->ldb, d->b_batch_stride)) return false;
    // the lean epilogue is float4 along the columns of C (the split-K slabs always qualify: N is a tile multiple)
    if (!split) {
        if (!d->c_trans && !host_aligned(d->C, d->ldc, d->c_batch_stride)) return false;
        if (d->emul && !host_aligned(d->emul, d->lde, d->e_batch_stride)) return false;
        if (d->bias && !host_aligned(d->bias, 4, d->bias_batch_stride)) return false;
    }
    return true;
}

static_assert(sizeof(GemmK) <= sizeof(((RnDeferredReduce*)nullptr)->k), "RnDeferredReduce::k holds a GemmK");
static thread_local const void* tl_planes_hint = nullptr;
void rn_gemm_planes_hint(const void* planes) { tl_planes_hint = planes; }

static int rn_gemm_impl(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st, RnDeferredReduce* defer) {
    const void* planes_ready = tl_planes_hint;      // (for THIS call only)
    tl_planes_hint = nullptr;
    if (defer) defer->valid = 0;
    if (!d || d->M < 0 || d->N < 0 || d->K < 0 || d->batch < 0) return RECNOW_EINVAL;
    if (d->M == 0 || d->N == 0 || d->batch == 0) return RECNOW_OK;
    if (!d->A || !d->B || !d->C) return RECNOW_EINVAL;
    if ((d->a_mode != RECNOW_OPMODE_NONE && !d->A2) || (d->b_mode != RECNOW_OPMODE_NONE && !d->B2)) return RECNOW_EINVAL;
    if (d->K == 0) return RECNOW_EUNSUPPORTED;
    GemmCfg c = pick_cfg(d);
    GemmK k;
    k.A = d->A; k.A2 = d->a_mode ? d->A2 : nullptr; k.B = d->B; k.B2 = d->b_mode ? d->B2 : nullptr;
    k.bias = d->bias; k.emul = d->emul; k.C = d->C; k.partial = nullptr;
    k.lda = d->lda; k.ldb = d->ldb; k.ldc = d->ldc; k.lde = d->lde;
    k.sA = d->a_batch_stride; k.sB = d->b_batch_stride; k.sC = d->c_batch_stride;
    k.sBias = d->bias_batch_stride; k.sE = d->e_batch_stride;
    k.M = d->M; k.N = d->N; k.K = d->K; k.batch = d->batch;
    k.a_mode = d->a_mode; k.a_act = d->a_act; k.b_mode = d->b_mode; k.b_act = d->b_act;
    k.act = d->act; k.act_cols = d->act_cols > 0 ? d->act_cols : d->N; k.e_mode = d->e_mode; k.e_act = d->e_act;
    k.accumulate = d->accumulate; k.c_trans = d->c_trans;
    k.a_hq = d->a_hq > 0 ? d->a_hq : 1; k.b_hq = d->b_hq > 0 ? d->b_hq : 1; k.a_ld2 = d->a_ld2; k.b_ld2 = d->b_ld2;
    k.bx = d->sp_bx; k.cx = d->sp_cx; k.bx_ks = d->sp_bx_ks; k.bx_rs = d->sp_bx_rs; k.cx_ms = d->sp_cx_ms; k.cx_rs = d->sp_cx_rs;
    k.sp_r = d->sp_r;
    k.eu_p = d->eu_p; k.eu_q = d->eu_q; k.eu_pms = d->eu_pms; k.eu_qrs = d->eu_qrs; k.eu_qns = d->eu_qns; k.eu_r = d->eu_r;
    k.npart = d->N + (d->sp_r > 0 ? 4 : 0);
    if (d->sp_r < 0 || d->sp_r > 4 || d->eu_r < 0 || d->eu_r > 4 || (d->sp_r > 0 && d->eu_r > 0)) return RECNOW_EINVAL;
    if (d->sp_r > 0 && (!d->sp_bx || !d->sp_cx || d->batch != 1)) return RECNOW_EINVAL;
    if (d->eu_r > 0 && (!d->eu_p || !d->eu_q || d->batch != 1)) return RECNOW_EINVAL;
    pick_split(d, c, &k.splitk, &k.kchunk, g_split_slots > 0 ? g_split_slots : 512);
    k.perm_s = 0;
    if (d->c_perm_s > 0) {
        if (k.splitk <= 1) return RECNOW_EUNSUPPORTED;       // the permuted store lives in the split-K reduce
        k.perm_s = d->c_perm_s;
    }
    k.trace = nullptr;
    k.cu_slots = nullptr; k.stagger_ticks = 0;
    k.tail_pairs = 8;
    static const int direct = []() { const char* e = getenv("RECNOW_GEMM_DIRECT"); return e ? atoi(e) : 1; }();      // A/B switch
    k.direct_store = direct;
    static const int gemm_prio = []() { const char* e = getenv("RECNOW_GEMM_PRIO"); return e ? atoi(e) : 0; }();     // A/B switch
    k.prio = gemm_prio;
    if (d->c_perm_s < 0 || (d->c_perm_s > 0 && (d->N % d->c_perm_s || d->batch != 1 || d->c_trans || d->accumulate))) return RECNOW_EINVAL;
    if (d->k_valid < 0 || d->k_valid > d->K) return RECNOW_EINVAL;
    static const bool sk_tail = []() { const char* e = getenv("RECNOW_SK_TAIL"); return !e || e[0] != '0'; }();      // A/B switch
    if (sk_tail && d->k_valid > 0 && d->K % 16 == 0 && d->k_valid > d->K - 16) k.tail_pairs = (d->k_valid - (d->K - 16) + 1) / 2;      // pairs of the last 16-deep tile
    k.C2 = d->C2; k.E2 = d->E2; k.ldc2 = d->ldc2; k.lde2 = d->lde2;
    k.as_in = d->as_in; k.as_out = d->as_out;
    if ((d->as_in != nullptr) != (d->as_out != nullptr)) return RECNOW_EINVAL;
    if (d->c2_mode < 0 || d->c2_mode > 6 || (d->c2_mode && d->c2_mode != 3 && d->c2_mode < 5 && !d->C2) || ((d->c2_mode == 2 || d->c2_mode >= 4) && !d->E2))
        return RECNOW_EINVAL;
    if (d->c2_mode >= 5) {      // the input gradient in one go: C = acc + E2 * E3 [+ E4 * E5] + rv (x) cv * E6
        if (!d->E3 || !d->E6 || !d->rv || !d->cv || ((uintptr_t)d->cv & 15) || d->lde2 != d->lde3 || d->emul || d->accumulate || (d->c2_mode == 5 && (!d->E4 || !d->E5)))
            return RECNOW_EINVAL;
        if (!host_aligned(d->E2, d->lde2, 0) || !host_aligned(d->E3, d->lde2, 0) || !host_aligned(d->E6, d->lde2, 0) ||
            (d->c2_mode == 5 && (!host_aligned(d->E4, d->lde2, 0) || !host_aligned(d->E5, d->lde2, 0))))
            return RECNOW_EUNSUPPORTED;
    }
    k.E4 = d->E4; k.E5 = d->E5; k.E6 = d->E6;
    if (d->c2_mode == 3 && (!d->hv || !d->hp || !d->emul || d->hp_ld < d->N / 64 || ((uintptr_t)d->hv & 15))) return RECNOW_EINVAL;
    if (d->c2_mode == 4 && (!d->E3 || !d->rv || !d->cv || ((uintptr_t)d->cv & 15) || !host_aligned(d->E3, d->lde3, 0))) return RECNOW_EINVAL;
    k.E3 = d->E3; k.lde3 = d->lde3; k.rv = d->rv; k.cv = d->cv; k.hv = d->hv; k.hp = d->hp; k.hp_ld = d->hp_ld;
    k.mid_V = d->mid_V; k.mid_T1 = d->mid_T1; k.mid_T2 = d->mid_T2; k.mid_T2g = d->mid_T2g; k.mid_ld = d->mid_ld; k.mid_act_outer = d->mid_act_outer;
    if (d->mid_V) {      // fused sub-space forward: the transposed GEMM1 of DCNMixLayer with two experts of 64 (see recnow_gemm_desc)
        if (!d->mid_T1 || !d->mid_T2 || !d->mid_T2g || d->M != 128 || d->N % 128 || d->sp_r != 2 || !d->a_trans || !d->b_trans || d->a_mode ||
            (d->b_mode != RECNOW_OPMODE_NONE && d->b_mode != RECNOW_OPMODE_MUL) ||
            d->batch != 1 || d->bias || d->emul || d->accumulate || d->c_trans || d->mid_ld % 4 || d->mid_ld < 144 ||
            (((uintptr_t)d->mid_T1 | (uintptr_t)d->mid_T2 | (uintptr_t)d->mid_T2g) & 15))
            return RECNOW_EUNSUPPORTED;
    }
    if (d->c2_mode && (d->K > 512 || d->K % 16 || d->batch != 1)) return RECNOW_EUNSUPPORTED;      // short-K kernel only
#ifdef RN_GEMM_TRACE
    if (const char* t = getenv("RECNOW_GEMM_TRACE")) k.trace = (long long*)strtoull(t, nullptr, 10);
#endif
    if (k.splitk > 1) {
        const size_t need = rn_align((size_t)k.splitk * d->batch * d->M * k.npart * sizeof(float));
        if (!ws || ws_bytes < need) return RECNOW_EWORKSPACE;
        k.partial = (float*)ws;
    }
    const long long gz = (long long)d->batch * k.splitk;
    if (gz > 65535) return RECNOW_EUNSUPPORTED;
    dim3 grid(rn_cdiv(d->M, c.BM), rn_cdiv(d->N, c.BN), (unsigned)gz);
    k.xcd_remap = (k.splitk > 1 && d->batch == 1 && grid.y == 1 && gz % 8 == 0 && grid.x > 1) ? 1 : 0;
    if (!k.xcd_remap && grid.y > 1 && grid.x % 8 == 0 && d->sp_r == 0 && !d->as_out) k.xcd_remap = 2;
    const bool a_kc = d->a_trans == 0, b_kc = d->b_trans != 0;
    int rc;
    int tag = (c.BM == 256 && c.BN == 32) ? RN_TAG_GEMM_256x32 : (c.BM == 256) ? RN_TAG_GEMM_256x64
            : (c.BN == 160) ? RN_TAG_GEMM_128x160 : (c.BM == 64) ? RN_TAG_GEMM_64x128 : RN_TAG_GEMM_128x128;
    // short-K products (K <= 256, e.g. the K = N*S+N = 130 contractions of DCN-v2) are prologue/epilogue dominated:
    // BK = 16 halves the LDS footprint so 4 workgroups per CU overlap each other's load/store phases.
    const bool short_k = d->K <= 256 || (d->c2_mode && d->K <= 512);
    const bool bk16 = short_k && c.BM == 128;
    const bool edge = !gemm_interior(d, c, bk16 ? 16 : 32, k.kchunk, k.splitk > 1);
    // lean kernels exist for the two big tile families and the (layout, operand-kind) combos the layers use; anything
    // else (and every edge shape) runs the general kernel of the same tile family.
    rc = RECNOW_EUNSUPPORTED;
    int xf = (d->sp_r > 0 ? 1 : 0) | (d->eu_r > 0 ? 2 : 0);
    if (d->mid_V && (edge || k.splitk != 1 || c.BM != 128 || c.BN != 128 || short_k)) return RECNOW_EUNSUPPORTED;
    // the persistent short-K kernel takes the plain products below (same condition as its branch)
    const bool use_shortk = !xf && !d->as_out && !edge && bk16 && c.BN == 128 && a_kc && k.splitk == 1 && d->batch == 1 && d->a_mode == 0 &&
                            d->b_mode == 0 && !d->bias && d->act == RECNOW_ACT_LINEAR && !d->c_trans &&
                            (!d->emul || d->e_mode == RECNOW_OPMODE_MUL);
    if (use_shortk) tag = RN_TAG_GEMM_SHORTK;
    static const bool split_longk = []() { const char* e = getenv("RECNOW_SPLIT_LONGK"); return !e || e[0] != '0'; }();      // A/B switch (diagnostics: which family an error comes from)
    const bool split_ok = g_gemm_precision == 1 && split_longk && xf == 1 && !d->as_out && !edge && !bk16 && c.BM == 128 && c.BN == 128 && d->b_mode == 0 &&
                          (d->a_mode == RECNOW_OPMODE_NONE || d->a_mode == RECNOW_OPMODE_MUL) &&
                          ((a_kc && !b_kc && (d->a_mode == 0 || planes_ready != nullptr)) || (a_kc && b_kc) || (!a_kc && !b_kc));      // (x0 * O_{l-1}) U: the lean kernel on the caller's planes only
    if (split_ok) tag = RN_TAG_GEMM_SPLIT;
    if (d->mid_V) tag = RN_TAG_GEMM_MIDF;      // a kernel of its own: the product's flops + the whole sub-space forward in one launch
    RnProfRecord* pr = nullptr;
    if (rn_prof_on()) {
        // algorithmic HBM bytes: every operand read once, every output written once (read-modify-write outputs count twice)
        const double mk = (double)d->M * d->K, kn = (double)d->K * d->N, mn = (double)d->M * d->N;
        const double elems = mk * (1 + (d->a_mode && d->a_mode != RECNOW_OPMODE_OUTER ? 1 : 0)) + kn * (1 + (d->b_mode && d->b_mode != RECNOW_OPMODE_OUTER ? 1 : 0)) +
                             mn * (1 + (d->emul ? 1 : 0) + (d->accumulate ? 1 : 0)) +
                             (d->c2_mode == 1 ? mn : d->c2_mode == 2 ? 3 * mn : d->c2_mode == 3 ? 0.0 : d->c2_mode == 4 ? 3 * mn : d->c2_mode == 5 ? 5 * mn : d->c2_mode == 6 ? 3 * mn : 0.0) +
                             (d->as_out ? 2 * mk : 0.0);
        pr = rn_prof_begin(tag, d->prof_flops > 0.0 ? d->prof_flops : 2.0 * d->M * d->N * (double)d->K * d->batch, 4.0 * elems * d->batch, st);
    }
    if (d->as_out) {      // A-stream side output: instantiated with the side product of dT2g; written by the first column tile
        if (xf != 1 || d->a_trans || d->a_mode != RECNOW_OPMODE_MUL || d->batch != 1 || k.splitk != 1 ||
            !host_aligned(d->as_in, d->lda, 0) || !host_aligned(d->as_out, d->lda, 0))
            return RECNOW_EUNSUPPORTED;
        xf |= 4;
    }
    if (xf) {        // side product / rank-R update exist only in the lean 128x128 kernels: the caller guarantees the shape
        if (edge || (c.BM != 128 && c.BM != 64) || c.BN != 128 || d->c2_mode) return RECNOW_EUNSUPPORTED;
        rc = RECNOW_EUNSUPPORTED;
        if (c.BM == 64) {        // small-M dispatch (pick_cfg): 64 x 128 tiles, two-wide side product, fp32 kernels only
            if (xf != 1 || bk16) return RECNOW_EUNSUPPORTED;
            rc = rn_gemm_launch_lean64x(k, a_kc, b_kc, d->a_mode, d->b_mode, 9, grid, st);
            if (rc) return rc;
        } else {
        // opt-in split precision: the long-K products with a side product (every k_gemm launch of the DCN-v2 step)
        if (split_ok) {
            void* planes = nullptr;
            if (split_planes_shape(d)) {
                const size_t used = k.splitk > 1 ? rn_align((size_t)k.splitk * d->batch * d->M * k.npart * sizeof(float)) : 0;
                if (ws && ws_bytes >= used + rn_gemm_split_planes_bytes(d->K, d->N)) planes = (char*)ws + used;
            }
            rc = rn_gemm_launch_split(k, a_kc, b_kc, d->a_mode, planes, grid, st, planes_ready);
        }
        static const bool sp_narrow = []() { const char* e = getenv("RECNOW_SP_NARROW"); return !e || e[0] != '0'; }();      // A/B switch
        if (d->mid_V) rc = rn_gemm_launch_lean128x(k, a_kc, b_kc, 32, 0, d->b_mode, 25, grid, st);      // XF 16 | 8 | 1: the fused sub-space forward
        else if (rc == RECNOW_EUNSUPPORTED && sp_narrow && xf == 1 && d->sp_r <= 2 && !bk16) {
            const bool glds = g_gemm_staging.load(std::memory_order_relaxed) == 1;      // A/B switch: LDS-DMA operand staging (XF | 32)
            if (glds && !a_kc && !b_kc && d->a_mode == 0 && d->b_mode == 0) rc = rn_gemm_launch_lean128x(k, a_kc, b_kc, 32, 0, 0, 41, grid, st);
            if (rc == RECNOW_EUNSUPPORTED) rc = rn_gemm_launch_lean128x(k, a_kc, b_kc, 32, d->a_mode, d->b_mode, 9, grid, st);
        }
        if (!d->mid_V && rc == RECNOW_EUNSUPPORTED) rc = rn_gemm_launch_lean128x(k, a_kc, b_kc, bk16 ? 16 : 32, d->a_mode, d->b_mode, xf, grid, st);
        if (rc) return rc;
        }
    } else if (use_shortk) {
        // C = (A B) [* emul] [+ C] with a short K: persistent kernel, no per-tile prologue, pipelined epilogue (gemm_shortk.hip)
        if (d->c2_mode && d->c2_mode < 5 && ((d->C2 && !host_aligned(d->C2, d->ldc2, 0)) || ((d->c2_mode == 2 || d->c2_mode == 4) && !host_aligned(d->E2, d->lde2, 0)))) return RECNOW_EUNSUPPORTED;
        rc = RECNOW_EUNSUPPORTED;
        static const bool sk_split = []() { const char* e = getenv("RECNOW_SPLIT_SHORTK"); return !e || e[0] != '0'; }();      // A/B switch
        if (g_gemm_precision == 1 && sk_split && shortk_planes_shape(d) && (planes_ready || (ws && ws_bytes >= rn_gemm_shortk_planes_bytes(d->K, d->N)))) {
            rc = rn_gemm_launch_shortk_split(k, b_kc, (d->emul ? 1 : 0) | (d->accumulate ? 2 : 0), d->c2_mode, ws, st, planes_ready);
            if (rc == RECNOW_OK) tag_used_split_shortk();
        }
        if (rc == RECNOW_EUNSUPPORTED) rc = rn_gemm_launch_shortk(k, b_kc, (d->emul ? 1 : 0) | (d->accumulate ? 2 : 0), d->c2_mode, st);
        if (rc) return rc;
    } else if (d->c2_mode) {
        return RECNOW_EUNSUPPORTED;          // the second output exists only in the short-K kernel
    } else if (!edge && c.BM == 128) {
        const int bk = bk16 ? 16 : 32;
        rc = (c.BN == 160) ? rn_gemm_launch_lean160(k, a_kc, b_kc, bk, d->a_mode, d->b_mode, grid, st)
                           : rn_gemm_launch_lean128(k, a_kc, b_kc, bk, d->a_mode, d->b_mode, grid, st);
    }
    if (!edge && c.BM == 256 && c.BN == 64) rc = rn_gemm_launch_lean64(k, a_kc, b_kc, d->a_mode, d->b_mode, grid, st);
    if (rc == RECNOW_EUNSUPPORTED) rc = rn_gemm_launch_edge(k, c.BM, c.BN, a_kc, b_kc, grid, st);
    rn_prof_end(pr, st);
    if (rc) return rc;
    if (k.splitk > 1) {
        const int64_t total = (int64_t)d->M * k.npart * d->batch;
        int g = rn_cdiv(total, 256);
        if (g > 2048) g = 2048;
        static const bool quad_reduce = []() { const char* e = getenv("RECNOW_REDUCE_QUAD"); return !e || e[0] != '0'; }();      // A/B switch
        int variant = 0, blocks = g;
        // four lanes per output (each a quarter of the slabs) only where one lane per output would leave the chip idle: with eight slabs in
        // flight per lane the one-lane form reads faster from 96 workgroups on (c3 layer-end reduction: 19.3 vs 24.1 us)
        if (d->N % 4 == 0 && quad_reduce && k.splitk >= 16 && (total / 4) % 64 == 0 && rn_cdiv(total / 4, 256) < 96) {      // whole quads per wave: the shuffles need all four lanes in the loop
            variant = 2;
            blocks = rn_cdiv(total, 256) > 4096 ? 4096 : rn_cdiv(total, 256);
        } else if (d->N % 4 == 0) {
            variant = 1;
            blocks = rn_cdiv(total / 4, 256) > 2048 ? 2048 : rn_cdiv(total / 4, 256);
        }
        if (defer) {             // the caller runs the reduction (rn_layer_end_reduce)
            memcpy(defer->k, &k, sizeof(GemmK));
            defer->variant = variant;
            defer->blocks = blocks;
            defer->valid = 1;
            return RECNOW_OK;
        }
        if (variant == 2) hipLaunchKernelGGL((k_gemm_splitk_reduce<true, true>), blocks, 256, 0, st, k);
        else if (variant == 1) hipLaunchKernelGGL((k_gemm_splitk_reduce<true, false>), blocks, 256, 0, st, k);
        else hipLaunchKernelGGL((k_gemm_splitk_reduce<false, false>), blocks, 256, 0, st, k);
        RN_LAUNCH_CHECK();
    }
    return RECNOW_OK;
}
int rn_gemm(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st) { return rn_gemm_impl(d, ws, ws_bytes, st, nullptr); }
int rn_gemm_deferred(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st, RnDeferredReduce* out) {
    if (!out) return RECNOW_EINVAL;
    return rn_gemm_impl(d, ws, ws_bytes, st, out);
}
void rn_deferred_slabs(const RnDeferredReduce* r, const float** partial, int* nsplit, int* ld, int64_t* stride) {
    GemmK k;
    memcpy(&k, r->k, sizeof(GemmK));
    *partial = k.partial;
    *nsplit = k.splitk;
    *ld = k.npart;
    *stride = (int64_t)k.M * k.npart;
}
int rn_layer_end_reduce(const RnDeferredReduce* a, const RnDeferredReduce* b, const float* dv_part, int dv_nparts, int dv_total, float* dV,
                        hipStream_t st) {
    GemmK ka, kb;
    memset(&ka, 0, sizeof(ka));
    memset(&kb, 0, sizeof(kb));
    const int na = (a && a->valid) ? a->blocks : 0, nb = (b && b->valid) ? b->blocks : 0;
    if (na) memcpy(&ka, a->k, sizeof(GemmK));
    if (nb) memcpy(&kb, b->k, sizeof(GemmK));
    const int nv = (dv_part && dV && dv_nparts > 0 && dv_total > 0) ? rn_cdiv(dv_total, 64) : 0;
    if (na + nb + nv == 0) return RECNOW_OK;
    hipLaunchKernelGGL(k_layer_end_reduce, na + nb + nv, 256, 0, st, ka, na ? a->variant : 0, na, kb, nb ? b->variant : 0, nb, dv_part, dv_nparts,
                       dv_total, dV);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// ---- deterministic column sum (bias gradients) -----------------------------------------------------------
#define CS_ROWS_PER_BLOCK 512
#define CS_MIN_ROWS 16
// rows per workgroup: 512 for tall matrices, fewer when that would leave most of the 256 CUs idle
static inline int cs_rows(int64_t M, int64_t N) {
    const int64_t colblk = (N + 255) / 256;
    int64_t r = CS_ROWS_PER_BLOCK;
    while (r > CS_MIN_ROWS && ((M + r - 1) / r) * colblk < 512) r /= 2;
    return (int)r;
}
__device__ __forceinline__ float cs_term(const float* __restrict__ X, const float* __restrict__ X2, int mode, int act, int64_t i) {
    float v = X[i];
    if (mode == RECNOW_OPMODE_MUL) v *= X2[i];
    else if (mode == RECNOW_OPMODE_ACTGRAD) v *= rn_act_grad_from_out(X2[i], act);
    return v;
}
__global__ void __launch_bounds__(256)
k_colsum_partial(const float* __restrict__ X, const float* __restrict__ X2, int mode, int act, int64_t M, int64_t N, int64_t ld,
                 float* __restrict__ part, int rows, int cw, int64_t x_bs, int64_t part_bs) {
    // blockIdx.z: one of a batch of equally shaped matrices (rn_colsum_batched: the bias gradients of all experts of a layer in ONE launch -- a
    // single (32 768 x 512) matrix is 128 workgroups, half the chip; round 5)
    X += (int64_t)blockIdx.z * x_bs;
    if (X2) X2 += (int64_t)blockIdx.z * x_bs;
    part += (int64_t)blockIdx.z * part_bs;
    // block (bx, by): columns bx*cw + (tid % cw), rows by*rows .. ; the 256/cw row lanes of a column take rows rl apart
    // (cw = 64 or 128 for matrices narrower than 256 columns: all 256 threads load), four loads in flight each, and are
    // summed in a fixed order through LDS
    // fp64 accumulators (the kernel is bound by its loads; a bias gradient is a sum of up to 512 rows per workgroup here and of the slabs
    // after it, with cancellation: summed in fp32 the full-size PLE + listwise test sat at 1.08e-5 of its 1e-5 bound)
    __shared__ double red[256];
    const int c = threadIdx.x % cw, rlane = threadIdx.x / cw, rl = 256 / cw;
    const int64_t col = (int64_t)blockIdx.x * cw + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    const int64_t r1 = min(M, r0 + rows);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (col < N) {
        int64_t r = r0 + rlane;
        for (; r + 3 * rl < r1; r += 4 * rl) {
            s0 += cs_term(X, X2, mode, act, r * ld + col);
            s1 += cs_term(X, X2, mode, act, (r + rl) * ld + col);
            s2 += cs_term(X, X2, mode, act, (r + 2 * rl) * ld + col);
            s3 += cs_term(X, X2, mode, act, (r + 3 * rl) * ld + col);
        }
        for (; r < r1; r += rl) s0 += cs_term(X, X2, mode, act, r * ld + col);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rlane == 0 && col < N) {
        double t = red[c];
        for (int u = 1; u < rl; ++u) t += red[u * cw + c];
        part[(int64_t)blockIdx.y * N + col] = (float)t;
    }
}
// out[col] = sum over slabs, fixed order: 64 columns x 16 strided slab groups per workgroup, then a 16-term LDS sum (a
// column-per-thread loop over hundreds of slabs is a chain of dependent-latency loads: 84 us for 512 slabs, measured)
__global__ void __launch_bounds__(1024)
k_colsum_final(const float* __restrict__ part, int nslab, int64_t N, float* __restrict__ out, int accumulate, int64_t part_bs, int64_t out_bs) {
    __shared__ double red[16][64];
    part += (int64_t)blockIdx.y * part_bs;
    out += (int64_t)blockIdx.y * out_bs;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + e;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (col < N) {
        int i = q;
        for (; i + 48 < nslab; i += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += part[(int64_t)(i + 16 * u) * N + col];
        }
        for (; i < nslab; i += 16) s[0] += part[(int64_t)i * N + col];
    }
    red[q][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (q == 0 && col < N) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += red[u][e];
        out[col] = accumulate ? (float)((double)out[col] + t) : (float)t;
    }
}

// narrow matrices (N < 64): one workgroup column-strip per (column, row-chunk), all 256 threads walk the rows
#define CS_NARROW_ROWS 8192
__global__ void __launch_bounds__(256)
k_colsum_narrow(const float* __restrict__ X, const float* __restrict__ X2, int mode, int act, int64_t M, int64_t N, int64_t ld,
                float* __restrict__ part) {
    __shared__ double red[16];
    const int64_t col = blockIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * CS_NARROW_ROWS, r1 = min(M, r0 + CS_NARROW_ROWS);
    double s = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        float v = X[r * ld + col];
        if (mode == RECNOW_OPMODE_MUL) v *= X2[r * ld + col];
        else if (mode == RECNOW_OPMODE_ACTGRAD) v *= rn_act_grad_from_out(X2[r * ld + col], act);
        s += v;
    }
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * N + col] = (float)s;
}

size_t rn_colsum_ws_bytes(int64_t M, int64_t N) {
    // upper bound valid for every row count <= M (callers size the workspace for an upper bound of M): rows per
    // workgroup are halved only while slabs * colblk < 512, so slabs <= max(ceil(M/512), 1024/colblk + 2)
    const int64_t m = M > 0 ? M : 1, n = N > 0 ? N : 1;
    const int64_t colblk = (n + 255) / 256;
    int64_t slabs = rn_cdiv(m, CS_ROWS_PER_BLOCK);
    const int64_t alt = 1024 / colblk + 2;
    if (alt > slabs) slabs = alt;
    return rn_align((size_t)(slabs + 1) * (size_t)n * sizeof(float));
}

int rn_colsum(const float* X, const float* X2, int mode, int act, int64_t M, int64_t N, int64_t ld, float* out,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    if (M < 0 || N < 0) return RECNOW_EINVAL;
    if (N == 0) return RECNOW_OK;
    if (!out) return RECNOW_EINVAL;
    if (M == 0) {
        if (!accumulate) RN_HIP(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!X || (mode && !X2) || !ws) return RECNOW_EINVAL;
    if (ws_bytes < rn_colsum_ws_bytes(M, N)) return RECNOW_EWORKSPACE;
    const int rows = cs_rows(M, N);
    int nslab = rn_cdiv(M, rows);
    if (N < 64) {
        nslab = rn_cdiv(M, CS_NARROW_ROWS);
        dim3 gn((unsigned)N, nslab);
        hipLaunchKernelGGL(k_colsum_narrow, gn, 256, 0, st, X, X2, mode, act, M, N, ld, (float*)ws);
    } else {
        const int cw = N <= 64 ? 64 : N <= 128 ? 128 : 256;
        dim3 g1(rn_cdiv(N, cw), nslab);
        hipLaunchKernelGGL(k_colsum_partial, g1, 256, 0, st, X, X2, mode, act, M, N, ld, (float*)ws, rows, cw, (int64_t)0, (int64_t)0);
    }
    hipLaunchKernelGGL(k_colsum_final, rn_cdiv(N, 64), 1024, 0, st, (const float*)ws, nslab, N, out, accumulate, (int64_t)0, (int64_t)0);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// `batch` matrices of one shape, x_bs floats apart (X2 likewise), out[b * out_bs + col]: two launches for all of them.  N >= 64 columns (narrower
// ones: one rn_colsum per matrix).  ws: batch * rn_colsum_ws_bytes(M, N).  The sums of a matrix are those of rn_colsum, term for term.
int rn_colsum_batched(const float* X, const float* X2, int mode, int act, int64_t M, int64_t N, int64_t ld, int batch, int64_t x_bs, float* out,
                      int64_t out_bs, void* ws, size_t ws_bytes, hipStream_t st) {
    if (M < 1 || N < 64 || batch < 1 || batch > 65535) return RECNOW_EUNSUPPORTED;
    if (!X || (mode && !X2) || !ws || !out) return RECNOW_EINVAL;
    const size_t one = rn_colsum_ws_bytes(M, N);
    if (ws_bytes < one * (size_t)batch) return RECNOW_EWORKSPACE;
    const int rows = cs_rows(M, N);
    const int nslab = rn_cdiv(M, rows);
    const int cw = N <= 64 ? 64 : N <= 128 ? 128 : 256;
    const int64_t part_bs = (int64_t)(one / sizeof(float));
    dim3 g1(rn_cdiv(N, cw), nslab, batch);
    hipLaunchKernelGGL(k_colsum_partial, g1, 256, 0, st, X, X2, mode, act, M, N, ld, (float*)ws, rows, cw, x_bs, part_bs);
    dim3 g2(rn_cdiv(N, 64), batch);
    hipLaunchKernelGGL(k_colsum_final, g2, 1024, 0, st, (const float*)ws, nslab, N, out, 0, part_bs, out_bs);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_set_gemm_precision(int mode) { return rn_gemm_set_precision(mode); }
extern "C" int recnow_set_gemm_staging(int mode) {
    if (mode != 0 && mode != 1) return RECNOW_EINVAL;
    g_gemm_staging.store(mode, std::memory_order_relaxed);
    return RECNOW_OK;
}
extern "C" int recnow_get_gemm_staging(void) { return g_gemm_staging.load(std::memory_order_relaxed); }
extern "C" int recnow_get_gemm_precision(void) { return rn_gemm_precision(); }
extern "C" size_t recnow_gemm_workspace_bytes(const recnow_gemm_desc* desc_host) {
    return desc_host ? rn_gemm_ws_bytes(desc_host) : 0;
}
extern "C" int recnow_gemm(const recnow_gemm_desc* desc_host, void* ws, size_t ws_bytes, void* stream) {
    return rn_gemm(desc_host, ws, ws_bytes, (hipStream_t)stream);
}
