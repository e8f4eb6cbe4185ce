// In-batch pairwise loss on sorted segments.  Replaces the (B,B) mask algebra, tile/transpose and boolean_mask
// passes of /root/reference/rec_now/rec_block/pairwise_loss_from_batch.py:228-291 by one thread per row walking its
// own segment.  Members of a segment are contiguous in sorted order and ascending in original row index (stable
// sort), so "ascending sorted position" == "ascending j": the pair order of tf.boolean_mask over the row-major
// flattened mask (:217, :272-273) is reproduced exactly.
//
// Integer/latency-bound: the member records (16 B each) of a segment are read by every lane of the segment at the
// same time (wave-uniform broadcast loads served by L1), work is sum_g n_g^2 candidate compares.
#include "common.hpp"
#include "scan.hpp"
#include "group_small.hpp"

struct __attribute__((aligned(16))) Member {   // one sorted row
    float label;
    float score;
    int32_t row;      // original row index
    int32_t valid;    // sample mask (1 = takes part)
};

__device__ __forceinline__ Member load_member(const float* __restrict__ scores, const float* __restrict__ labels,
                                              const uint8_t* __restrict__ mask, const int32_t* __restrict__ order, int64_t k) {
    Member m;
    m.row = order[k];
    m.label = labels[m.row];
    m.score = scores[m.row];
    m.valid = mask ? (mask[m.row] != 0) : 1;
    return m;
}

// Where a walk's member records come from: the packed array (k_pack_members), or -- UNP, the step's loss stage since round 5 -- straight from the loss's
// inputs through the sorted order (three dependent gathers instead of one 16-byte load: used to FILL the LDS stage of a workgroup, which is what the walks
// read; a walk that cannot be staged -- a group of more than 2048 rows -- pays the gathers per member).  Saves the pack launch in front of the walk.
template <bool UNP>
struct MemberSrc {
    const Member* __restrict__ mem;
    const float* __restrict__ scores;
    const float* __restrict__ labels;
    const uint8_t* __restrict__ mask;
    const int32_t* __restrict__ order;
    __device__ __forceinline__ Member operator[](int64_t k) const {
        if constexpr (UNP) return load_member(scores, labels, mask, order, k);
        else return mem[k];
    }
};

// zero_b / zero_1 (optional): the B per-row counters and the one total that the counting kernel adds into -- cleared here
// instead of by two memset launches (a loss call is a chain of few-microsecond launches; each one removed is ~4 us)
__global__ void k_pack_members(const float* __restrict__ scores, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
                               const int32_t* __restrict__ order, int64_t B, Member* __restrict__ out,
                               unsigned long long* __restrict__ zero_b, unsigned long long* __restrict__ zero_1) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < B) {
        out[k] = load_member(scores, labels, mask, order, k);
        if (zero_b) zero_b[k] = 0ull;
    }
    if (k == 0 && zero_1) *zero_1 = 0ull;
}

template <int FLAGS>
__device__ __forceinline__ bool pair_ok(const Member& a, const Member& b) {   // a = positive candidate, b = negative
    bool ok = a.valid && b.valid;
    if (FLAGS & RECNOW_PAIR_LABEL_GT) ok = ok && (a.label > b.label);
    if (FLAGS & RECNOW_PAIR_WRONG_ORDER) ok = ok && (a.score < b.score);
    return ok;
}


// The rows of a workgroup are 256 consecutive sorted positions, so the members they walk form ONE contiguous range of the
// member array: [first row's segment start, last row's segment end).  When it fits (<= PW_STAGE members, 32 KB) it is staged
// in LDS once, coalesced, and every per-row walk reads LDS (a walk is a chain of dependent 16-byte loads otherwise: ~0.4 us
// per member from L1/L2).  Block-uniform decision; oversize ranges (one huge group) fall back to global loads.
#define PW_STAGE 2048
template <typename SRC>
__device__ __forceinline__ bool stage_members(const SRC mem, const int32_t* __restrict__ seg_id,
                                              const int32_t* __restrict__ seg_first, int64_t B, Member* lds, int* base, int rows_per_block = 0) {
    *base = 0;
    const int64_t rpb = rows_per_block > 0 ? rows_per_block : (int64_t)blockDim.x;
    const int64_t k0 = (int64_t)blockIdx.x * rpb;
    if (k0 >= B) return false;
    const int64_t k1 = min(B, k0 + rpb) - 1;
    const int lo = seg_first[seg_id[k0]], hi = seg_first[seg_id[k1] + 1];
    if (hi - lo > PW_STAGE) return false;
    for (int i = threadIdx.x; i < hi - lo; i += blockDim.x) lds[i] = mem[lo + i];
    __syncthreads();
    *base = lo;
    return true;
}
// A row's walk over its segment, from LDS when the block's range was staged, else from global memory.  Two loops, so each
// reads through a pointer of a known address space (one generic pointer made every read a flat_load that waits on both
// counters), unrolled by four so that four member reads are in flight instead of one per ~40-instruction body.
#define PW_WALK(IN_LDS, STAGED, SBASE, MEM, S, E, J, O, BODY)            \
    do {                                                                  \
        if (IN_LDS) {                                                     \
            _Pragma("unroll 4") for (int J = (S); J < (E); ++J) {        \
                const Member O = (STAGED)[J - (SBASE)];                   \
                BODY                                                      \
            }                                                             \
        } else {                                                          \
            _Pragma("unroll 4") for (int J = (S); J < (E); ++J) {        \
                const Member O = (MEM)[J];                                \
                BODY                                                      \
            }                                                             \
        }                                                                 \
    } while (0)

// the same two loops with a stride of one wave: lanes of a wave share the walk of ONE row (k_pair_long)
#define PW_WALK_STRIDED(IN_LDS, STAGED, SBASE, MEM, S, E, J, O, BODY)    \
    do {                                                                  \
        if (IN_LDS) {                                                     \
            _Pragma("unroll 4") for (int J = (S); J < (E); J += 64) {    \
                const Member O = (STAGED)[J - (SBASE)];                   \
                BODY                                                      \
            }                                                             \
        } else {                                                          \
            _Pragma("unroll 4") for (int J = (S); J < (E); J += 64) {    \
                const Member O = (MEM)[J];                                \
                BODY                                                      \
            }                                                             \
        }                                                                 \
    } while (0)

// One candidate of a row's walk.  Labels are strictly ordered in at most one direction, so at most one of (me, o) / (o, me) is a
// pair; both take their terms from ONE exp(-|x|) (x = the active pair's score difference): softplus(-x) = max(-x, 0) +
// log(1 + e), sigma(-x) = e / (1 + e) or 1 / (1 + e).  No per-lane branches around the transcendentals (lanes of a wave walk
// segments of different membership: both sides would be serialised).  Hardware exp2 / log2 / rcp (1 ulp each) instead of libm
// expf / log1pf / IEEE division: ~25 instead of ~100 instructions per candidate.  e is in (0, 1], so 1 + e is in (1, 2] (never
// denormal): log(1 + e) is off by at most the rounding of 1 + e, 6e-8 absolute, on terms that sum to O(1) per pair -- two
// orders below the 1e-5 parity bar (tests compare against the fp64 oracle).
template <int FLAGS>
__device__ __forceinline__ void bpr_term(const Member& me, const Member& o, bool other, float factor, float& la, float& ga) {
    const bool fwd = other && pair_ok<FLAGS>(me, o);      // me is the positive of (me, o)
    const bool bwd = other && pair_ok<FLAGS>(o, me);      // me is the negative of (o, me)
    const float d = factor * (me.score - o.score);
    const float x = fwd ? d : -d;
    const float ex = __builtin_amdgcn_exp2f(-1.44269504f * fabsf(x));
    const float inv = __builtin_amdgcn_rcpf(1.f + ex);
    const float sg = x >= 0.f ? ex * inv : inv;                                          // sigma(-x)
    const float sp = fmaxf(-x, 0.f) + 0.69314718f * __builtin_amdgcn_logf(1.f + ex);     // softplus(-x)
    la += fwd ? sp : 0.f;
    ga += fwd ? -sg : (bwd ? sg : 0.f);
}

// ---- long segments: a wave per row ------------------------------------------------------------------
// One thread per row makes the time of a call the latency of the longest walk: a 2048-row group (the Zipf-skewed batches of
// SURVEY 8d) is 2048 dependent iterations per lane whatever the rest of the batch looks like.  Rows of segments longer than
// PW_LONG are therefore walked by a whole wave each -- lanes stride over the members (coalesced 16-byte reads), partial sums
// are joined by the fixed butterfly of wave_sum, so results stay bitwise reproducible -- and parked per sorted position for
// the thread-per-row kernels below, which skip the walk of such rows.  A workgroup owns 64 consecutive sorted rows; a segment
// lying strictly inside them is shorter than 64 rows, so only the first and the last row's segment can be long, and looking at
// those two decides block-uniformly whether there is anything to do: batches without long groups pay one empty launch.
#define PW_LONG 512
template <int FLAGS, int MODE>                     // MODE 0: pair counts;  1: BPR loss and gradient terms;  2: both in one walk
__global__ void __launch_bounds__(256)
k_pair_long(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first, int64_t B,
            float factor, int32_t* __restrict__ long_cnt, float* __restrict__ long_la, float* __restrict__ long_ga) {
    __shared__ Member staged[PW_STAGE];
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    if (k0 >= B) return;
    const int64_t kl = min(B, k0 + 64) - 1;
    const int g0 = seg_id[k0], g1 = seg_id[kl];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // the only segments of these 64 rows that can be long: the first row's and the last row's (everything below is block-uniform)
    bool staged_used = false;
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 1 && g1 == g0) break;
        const int g = phase == 0 ? g0 : g1;
        const int s = seg_first[g], e = seg_first[g + 1];
        if (e - s <= PW_LONG) continue;
        // members of the segment through LDS when they fit (32 KB): lanes then read consecutive 16-byte records at LDS speed and
        // the walk is bound by its ~25 VALU instructions per candidate; from global memory it was bound by the read latency
        const bool in_lds = e - s <= PW_STAGE;
        if (in_lds) {
            if (staged_used) __syncthreads();                       // every wave is done with the first segment
            for (int i = threadIdx.x; i < e - s; i += 256) staged[i] = mem[s + i];
            __syncthreads();
            staged_used = true;
        }
        const int64_t ka = k0 > s ? k0 : (int64_t)s, kb = kl < (int64_t)e - 1 ? kl : (int64_t)e - 1;
        for (int64_t k = ka + w; k <= kb; k += 4) {                 // waves take the segment's rows of this block in turn
            const Member me = mem[k];
            int cc = 0;
            float la = 0.f, ga = 0.f;
            PW_WALK_STRIDED(in_lds, staged, s, mem, s + lane, e, j, o, {
                if (MODE != 1) cc += (j != (int)k && pair_ok<FLAGS>(me, o)) ? 1 : 0;
                if (MODE != 0) bpr_term<FLAGS>(me, o, j != (int)k, factor, la, ga);
            });
            if (MODE != 1) {
                cc = wave_sum(cc);
                if (lane == 0) long_cnt[k] = cc;
            }
            if (MODE != 0) {
                la = wave_sum(la);
                ga = wave_sum(ga);
                if (lane == 0) {
                    long_la[k] = la;
                    long_ga[k] = ga;
                }
            }
        }
    }
}

// ---- count ---------------------------------------------------------------------------------------
template <int FLAGS>
__global__ void __launch_bounds__(256)
k_pair_count(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first,
             const int32_t* __restrict__ super_id, int64_t B, const int32_t* __restrict__ long_cnt, int32_t* __restrict__ cnt_row,
             unsigned long long* __restrict__ cnt_super, unsigned long long* __restrict__ n_pair) {
    __shared__ long long red[16];
    __shared__ Member staged[PW_STAGE];
    int sbase;
    const bool in_lds = stage_members(mem, seg_id, seg_first, B, staged, &sbase);
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long c = 0;
    if (k < B) {
        const Member me = mem[k];
        const int g = seg_id[k];
        const int s = seg_first[g], e = seg_first[g + 1];
        const bool is_long = e - s > PW_LONG;       // walked by k_pair_long
        int cc = 0;
        PW_WALK(in_lds, staged, sbase, mem, (is_long ? e : s), e, j, o, { cc += (j != (int)k && pair_ok<FLAGS>(me, o)) ? 1 : 0; });
        if (is_long) cc = long_cnt[k];
        cnt_row[me.row] = cc;
        if (cc) atomicAdd(&cnt_super[super_id[k]], (unsigned long long)cc);   // integer atomics: order-independent
        c = cc;
    }
    c = block_sum<long long>(c, red);
    if (threadIdx.x == 0 && c) atomicAdd(n_pair, (unsigned long long)c);
}

// ---- emit ----------------------------------------------------------------------------------------
template <int FLAGS>
__global__ void __launch_bounds__(256)
k_pair_emit(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first,
            int64_t B, const int64_t* __restrict__ offsets, int32_t* __restrict__ pos_idx, int32_t* __restrict__ neg_idx,
            int64_t capacity) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const Member me = mem[k];
    const int g = seg_id[k];
    const int s = seg_first[g], e = seg_first[g + 1];
    int64_t o = offsets[me.row];
    for (int j = s; j < e; ++j) {
        const Member ot = mem[j];
        if (j != (int)k && pair_ok<FLAGS>(me, ot)) {
            if (o < capacity) {
                pos_idx[o] = me.row;
                neg_idx[o] = ot.row;
            }
            ++o;
        }
    }
}

// ---- fused BPR forward + backward ------------------------------------------------------------------
// softplus(-x) = max(-x,0) + log1p(exp(-|x|))  ==  TF's max(x,0) - x + log1p(exp(-|x|)) for labels = 1
__device__ __forceinline__ float softplus_neg(float x) { return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_neg(float x) {   // sigma(-x), stable
    const float e = expf(-fabsf(x));
    return x >= 0.f ? e / (1.f + e) : 1.f / (1.f + e);
}

template <int FLAGS>
__global__ void __launch_bounds__(256)
k_pair_bpr(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first,
           const int32_t* __restrict__ super_id, const unsigned long long* __restrict__ cnt_super,
           const unsigned long long* __restrict__ n_pair, int64_t B, float factor, float power, int reduce_mean,
           const float* __restrict__ long_la, const float* __restrict__ long_ga, double* __restrict__ block_loss,
           float* __restrict__ dscores) {
    __shared__ double red[16];
    __shared__ Member staged[PW_STAGE];
    int sbase;
    const bool in_lds = stage_members(mem, seg_id, seg_first, B, staged, &sbase);
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double lsum = 0.0;
    if (k < B) {
        const Member me = mem[k];
        const int g = seg_id[k];
        const int s = seg_first[g], e = seg_first[g + 1];
        float w = 1.f;
        if (power != 0.f) {
            const float cnt = (float)cnt_super[super_id[k]];
            // cnt == 0: this row takes part in no pair, its weight is never used (avoid 0**negative = inf -> inf*0)
            w = (cnt == 0.f) ? 1.f : ((power == 1.f) ? cnt : powf(cnt, power));
        }
        const bool is_long = e - s > PW_LONG;       // walked by k_pair_long
        float la = 0.f, ga = 0.f;
        PW_WALK(in_lds, staged, sbase, mem, (is_long ? e : s), e, j, o, { bpr_term<FLAGS>(me, o, j != (int)k, factor, la, ga); });
        if (is_long) {
            la = long_la[k];
            ga = long_ga[k];
        }
        const float denom = reduce_mean ? ((float)(*n_pair) + 1.0e-10f) : 1.f;
        dscores[me.row] = w * factor * ga / denom;
        lsum = (double)(w * la);
    }
    lsum = block_sum<double>(lsum, red);
    if (threadIdx.x == 0) block_loss[blockIdx.x] = lsum;
}

// One pass for the loss without occurrence weights (click_occurance_power == 0, the default): the pair count of a row and its BPR
// terms come out of the SAME walk, so counting and loss are one launch each for the long and the short rows instead of two.  The
// gradient is left unnormalised (d sum / d score); the 1 / (P + 1e-10) of the mean is applied where the incoming gradient is
// multiplied in (k_pair_scale_grad), because P is complete only when this kernel is.
// Round 5 (second session): LPR lanes share the walk of ONE row (members s + sub, s + sub + LPR, ...): a walk is a chain of ~40 dependent-issue
// instructions per member (two transcendentals), and with one lane per row a 64-member group made the launch 64 such steps long whatever the grid --
// 12.6 us at 8192 rows, where the grid is 32 workgroups.  The lanes' gradient terms meet by two fixed-order butterfly steps (the same bits on every
// lane and run), their loss terms and counts go into the block sum as they are.  A block owns 256 / LPR consecutive sorted rows.
// SKIP_LONG: rows of long segments contribute nothing here and their dscores entry is not written (the long-row workgroups of k_pair_all do both).
template <int FLAGS, int LPR, bool SKIP_LONG = false, typename SRC = const Member*>
__device__ __forceinline__ void pair_one_body(const SRC mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first,
                                              int64_t B, float factor, const int32_t* __restrict__ long_cnt, const float* __restrict__ long_la,
                                              const float* __restrict__ long_ga, double* __restrict__ block_loss, unsigned long long* __restrict__ n_pair,
                                              float* __restrict__ dscores, Member* staged, double* red, long long* redc) {
    int sbase;
    const bool in_lds = stage_members(mem, seg_id, seg_first, B, staged, &sbase, 256 / LPR);
    const int sub = (int)threadIdx.x % LPR;
    const int64_t kr = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int64_t k = kr < B ? kr : B - 1;         // (surplus lanes of the last block walk the last row again and contribute nothing)
    double lsum = 0.0;
    long long c = 0;
    {
        const Member me = in_lds ? staged[k - sbase] : mem[k];          // (the staged range covers every row of the block)
        const int g = seg_id[k];
        const int s = seg_first[g], e = seg_first[g + 1];
        const bool is_long = e - s > PW_LONG;       // walked by k_pair_long
        int cc = 0;
        float la = 0.f, ga = 0.f;
        if (in_lds) {
#pragma unroll 4
            for (int j = (is_long ? e : s) + sub; j < e; j += LPR) {
                const Member o = staged[j - sbase];
                cc += (j != (int)k && pair_ok<FLAGS>(me, o)) ? 1 : 0;
                bpr_term<FLAGS>(me, o, j != (int)k, factor, la, ga);
            }
        } else {
#pragma unroll 4
            for (int j = (is_long ? e : s) + sub; j < e; j += LPR) {
                const Member o = mem[j];
                cc += (j != (int)k && pair_ok<FLAGS>(me, o)) ? 1 : 0;
                bpr_term<FLAGS>(me, o, j != (int)k, factor, la, ga);
            }
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) ga += __shfl_xor(ga, o, 64);
        if (is_long && !SKIP_LONG) {
            cc = sub == 0 ? long_cnt[k] : 0;
            la = sub == 0 ? long_la[k] : 0.f;
            ga = long_ga[k];
        }
        if (kr < B && !(SKIP_LONG && is_long)) {
            if (sub == 0) dscores[me.row] = factor * ga;
            lsum = (double)la;
            c = cc;
        }
    }
    lsum = block_sum<double>(lsum, red);
    c = block_sum<long long>(c, redc);
    if (threadIdx.x == 0) {
        block_loss[blockIdx.x] = lsum;
        if (c) atomicAdd(n_pair, (unsigned long long)c);      // integer atomics: order-independent
    }
}
template <int FLAGS>
__global__ void __launch_bounds__(256)
k_pair_one(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first, int64_t B, float factor,
           const int32_t* __restrict__ long_cnt, const float* __restrict__ long_la, const float* __restrict__ long_ga,
           double* __restrict__ block_loss, unsigned long long* __restrict__ n_pair, float* __restrict__ dscores) {
    __shared__ double red[16];
    __shared__ long long redc[16];
    __shared__ Member staged[PW_STAGE];
    pair_one_body<FLAGS, 1>(mem, seg_id, seg_first, B, factor, long_cnt, long_la, long_ga, block_loss, n_pair, dscores, staged, red, redc);
}
template <int FLAGS>
__global__ void __launch_bounds__(256)
k_pair_one4(const Member* __restrict__ mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first, int64_t B, float factor,
            const int32_t* __restrict__ long_cnt, const float* __restrict__ long_la, const float* __restrict__ long_ga,
            double* __restrict__ block_loss, unsigned long long* __restrict__ n_pair, float* __restrict__ dscores) {
    __shared__ double red[16];
    __shared__ long long redc[16];
    __shared__ Member staged[PW_STAGE];
    pair_one_body<FLAGS, 4>(mem, seg_id, seg_first, B, factor, long_cnt, long_la, long_ga, block_loss, n_pair, dscores, staged, red, redc);
}
// Round 5 (second session): the loss walk as ONE launch.  Workgroups [0, g_one) are k_pair_one / k_pair_one4 with the rows of long segments left out;
// workgroup g_one + b is k_pair_long's workgroup b (64 consecutive sorted rows, a wave per row of a long segment) -- but instead of parking (count, loss
// term, gradient term) per sorted row for a second kernel it writes the row's dscores entry, adds the counts to n_pair and leaves its loss terms as
// block_loss[g_one + b] (0 for the usual workgroup without a long segment).  No workgroup reads what another one writes: the batch without long groups no
// longer pays an empty launch (~6 us in front of every thread-per-row launch), the skewed batch no longer a round trip through three B-sized arrays.
template <int FLAGS, int LPR, bool UNP>
__global__ void __launch_bounds__(256)
k_pair_all(const MemberSrc<UNP> mem, const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first, int64_t B, float factor,
           double* __restrict__ block_loss, unsigned long long* __restrict__ n_pair, float* __restrict__ dscores, int g_one) {
    __shared__ double red[16];
    __shared__ long long redc[16];
    __shared__ Member staged[PW_STAGE];
    if ((int)blockIdx.x < g_one) {
        pair_one_body<FLAGS, LPR, true, MemberSrc<UNP>>(mem, seg_id, seg_first, B, factor, nullptr, nullptr, nullptr, block_loss, n_pair, dscores, staged, red,
                                                         redc);
        return;
    }
    const int64_t k0 = (int64_t)((int)blockIdx.x - g_one) * 64;
    double lsum = 0.0;
    long long csum = 0;
    if (k0 < B) {                                                       // block-uniform
        const int64_t kl = min(B, k0 + 64) - 1;
        const int g0 = seg_id[k0], g1 = seg_id[kl];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        bool staged_used = false;
        for (int phase = 0; phase < 2; ++phase) {                       // (as k_pair_long: only the first and the last row's segment can be long)
            if (phase == 1 && g1 == g0) break;
            const int g = phase == 0 ? g0 : g1;
            const int s = seg_first[g], e = seg_first[g + 1];
            if (e - s <= PW_LONG) continue;
            const bool in_lds = e - s <= PW_STAGE;
            if (in_lds) {
                if (staged_used) __syncthreads();
                for (int i = threadIdx.x; i < e - s; i += 256) staged[i] = mem[s + i];
                __syncthreads();
                staged_used = true;
            }
            const int64_t ka = k0 > s ? k0 : (int64_t)s, kb = kl < (int64_t)e - 1 ? kl : (int64_t)e - 1;
            for (int64_t k = ka + w; k <= kb; k += 4) {
                const Member me = in_lds ? staged[k - s] : mem[k];
                int cc = 0;
                float la = 0.f, ga = 0.f;
                PW_WALK_STRIDED(in_lds, staged, s, mem, s + lane, e, j, o, {
                    cc += (j != (int)k && pair_ok<FLAGS>(me, o)) ? 1 : 0;
                    bpr_term<FLAGS>(me, o, j != (int)k, factor, la, ga);
                });
                cc = wave_sum(cc);
                la = wave_sum(la);
                ga = wave_sum(ga);
                if (lane == 0) {
                    dscores[me.row] = factor * ga;
                    lsum += (double)la;                                 // (lane 0 of wave w: this wave's rows in ascending order)
                    csum += cc;
                }
            }
        }
    }
    lsum = block_sum<double>(lsum, red);
    csum = block_sum<long long>(csum, redc);
    if (threadIdx.x == 0) {
        block_loss[blockIdx.x] = lsum;
        if (csum) atomicAdd(n_pair, (unsigned long long)csum);
    }
}

// out[i] = d[i] * g[0] / (P + eps)   (P = *n_pair; n_pair == NULL: no division)
__global__ void __launch_bounds__(256)
k_pair_scale_grad(const float* __restrict__ d, const float* __restrict__ g, const unsigned long long* __restrict__ n_pair, float eps, int64_t B,
                  float* __restrict__ out) {
    const float sc = n_pair ? g[0] / ((float)(*n_pair) + eps) : g[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) out[i] = d[i] * sc;
}

__global__ void __launch_bounds__(1024)
k_loss_finalize(const double* __restrict__ part, int n, const unsigned long long* __restrict__ n_pair, int64_t p_host,
                int reduce_mean, float* __restrict__ loss) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];     // fixed order per thread, fixed tree
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) {
        float v = (float)s;
        if (reduce_mean) {
            const float P = n_pair ? (float)(*n_pair) : (float)p_host;
            v = v / (P + 1.0e-10f);
        }
        *loss = v;
    }
}

// ---- explicit-vector bpr_loss_func -----------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_bpr_vec(const float* __restrict__ pos, const float* __restrict__ neg, const float* __restrict__ weights, int64_t P,
          float factor, int reduce_mean, double* __restrict__ block_loss, float* __restrict__ dpos) {
    __shared__ double red[16];
    double lsum = 0.0;
    const float denom = reduce_mean ? ((float)P + 1.0e-10f) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = factor * (pos[i] - neg[i]);
        const float w = weights ? weights[i] : 1.f;
        lsum += (double)(w * softplus_neg(x));
        dpos[i] = -w * factor * sigmoid_neg(x) / denom;
    }
    lsum = block_sum<double>(lsum, red);
    if (threadIdx.x == 0) block_loss[blockIdx.x] = lsum;
}

// ---- dense mask / occurrence weights (API parity helpers) ---------------------------------------------
__global__ void k_pair_mask_dense(const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                                  const int32_t* __restrict__ seg_first, int64_t B, int only_upper_band, uint8_t* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int i = order[k];
    const int g = seg_id[k];
    for (int j = seg_first[g]; j < seg_first[g + 1]; ++j) {
        const int r = order[j];
        if (r == i) continue;
        if (only_upper_band && r != i + 1) continue;
        out[(int64_t)i * B + r] = 1;
    }
}

__global__ void k_occ_weight(const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                             const int32_t* __restrict__ seg_first, int64_t B, float power, float* __restrict__ w) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int g = seg_id[k];
    const float c = (float)(seg_first[g + 1] - seg_first[g]);
    w[order[k]] = (power == 1.f) ? c : powf(c, power);
}


// ---- small batches: grouping + member packing in ONE launch ---------------------------------------------------------
// B <= 8192 rows with one float32 / int32 group tensor (BASELINE config 2: pairwise_loss_from_batch at B = 8192, ~128 user groups).
// The general front end is a chain of launches of a few microseconds each (keys, solo memset, grouping, pack) whose wall time is
// launch latency.  Here ONE 1024-thread workgroup (group_small.hpp) forms the canonical keys, sorts them in LDS, derives the
// segments and writes, besides order / seg_id / seg_first, the packed member records (label, score, row, valid) that the counting
// and loss kernels walk, and clears their counters.  (Doing the pair walks in this workgroup too was tried: one CU is 1/256 of
// the chip, the loss took 0.3 ms instead of 0.08.)
__global__ void __launch_bounds__(GS_T)
k_group_pack_small(const void* __restrict__ groups, int key_dtype, const float* __restrict__ labels, const float* __restrict__ scores,
                   const uint8_t* __restrict__ mask, int B, int32_t* __restrict__ order, int32_t* __restrict__ seg_id,
                   int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg,
                   Member* __restrict__ mem, unsigned long long* __restrict__ cnt_super, unsigned long long* __restrict__ n_pair) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
    uint32_t* key0 = reinterpret_cast<uint32_t*>(gs_lds);
    uint32_t* key1 = key0 + GS_MAXB;
    uint16_t* idx0 = reinterpret_cast<uint16_t*>(key1 + GS_MAXB);
    uint16_t* idx1 = idx0 + GS_MAXB;
    uint16_t* cnt = idx1 + GS_MAXB;                                   // [16][GS_T] u16 = 32 KB
    unsigned* wsum = reinterpret_cast<unsigned*>(cnt + 16 * GS_T);    // 34 words
    uint8_t* solo = reinterpret_cast<uint8_t*>(wsum + 34);            // [B] NaN / inf flags by ORIGINAL row
    const int tid = threadIdx.x;
    unsigned vor = 0, vand = 0xffffffffu, ior = 0, iand = 0xffffffffu, bad = 0;
    // canonical keys (recnow_group_keys): -0.0 == +0.0; NaN and +-inf equal nothing, themselves included
    for (int i = tid; i < B; i += GS_T) {
        uint32_t k;
        bool so = false;
        if (key_dtype == RECNOW_KEY_F32) {
            const float v = reinterpret_cast<const float*>(groups)[i];
            k = __float_as_uint(v);
            if (v == 0.0f) k = 0u;
            so = !(fabsf(v) < INFINITY);
        } else {
            k = (uint32_t)reinterpret_cast<const int32_t*>(groups)[i];
        }
        key0[i] = k;
        idx0[i] = (uint16_t)i;
        solo[i] = so ? 1 : 0;
        vor |= k;
        vand &= k;
        uint32_t im;
        bad |= gs_int_key(k, &im) ? 0u : 1u;
        ior |= im;
        iand &= im;
        if (cnt_super) cnt_super[i] = 0ull;
    }
    if (tid == 0 && n_pair) *n_pair = 0ull;
    const unsigned varying = gs_varying_bits(key0, B, vor, vand, ior, iand, bad, wsum);      // (small-integer images of the keys when all have one)
    uint32_t* ka = key0; uint32_t* kb = key1;
    uint16_t* ia = idx0; uint16_t* ib = idx1;
    gs_radix_sort_lds(ka, kb, ia, ib, cnt, wsum, B, varying);
    const int lo = tid * GS_KPT, hi = min(B, lo + GS_KPT);
    unsigned heads = 0, nh = 0;
    for (int i = lo; i < hi; ++i) {
        bool h = true;
        if (i > 0) h = (ka[i] != ka[i - 1]) || solo[ia[i]] || solo[ia[i - 1]];
        heads |= (h ? 1u : 0u) << (i - lo);
        nh += h ? 1u : 0u;
    }
    unsigned total = 0;
    unsigned g = gs_block_exclusive_scan(nh, wsum, &total);
    for (int i = lo; i < hi; ++i) {
        const bool h = (heads >> (i - lo)) & 1u;
        if (h) { seg_first[g] = i; ++g; }
        const int row = ia[i];
        order[i] = row;
        seg_id[i] = (int32_t)g - 1;
        super_id[i] = (int32_t)g - 1;
        Member m;
        m.row = row;
        m.label = labels[row];
        m.score = scores[row];
        m.valid = mask ? (mask[row] != 0) : 1;
        mem[i] = m;
    }
    if (tid == 0) {
        seg_first[total] = B;
        n_seg[0] = (int32_t)total;
        n_seg[1] = (int32_t)total;
    }
}
static inline size_t ps_lds_bytes() { return gs_lds_bytes() + GS_MAXB; }
__global__ void k_seg_empty2(int32_t* seg_first, int32_t* n_seg, unsigned long long* n_pair) {
    seg_first[0] = 0;
    n_seg[0] = 0;
    n_seg[1] = 0;
    *n_pair = 0ull;
}

// ---- host side ---------------------------------------------------------------------------------------
#define RN_PW_T 256
#define RN_PW_LPR 4        // lanes per row of k_pair_one4
#define RN_VEC_BLOCKS 1024

extern "C" size_t recnow_pairwise_workspace_bytes(int64_t B) {
    if (B < 0) return 0;
    size_t s = rn_align((size_t)(B + 1) * sizeof(Member));
    size_t nb = 2 * (size_t)rn_cdiv(B > 0 ? B : 1, RN_PW_T / RN_PW_LPR);  // block partials of k_pair_all: RN_PW_T / RN_PW_LPR rows per thread-per-row block + 64 per long-row block
    if (nb < RN_VEC_BLOCKS) nb = RN_VEC_BLOCKS;
    s += rn_align(nb * sizeof(double));
    s += 3 * rn_align((size_t)(B > 0 ? B : 1) * sizeof(float));          // k_pair_long: counts, loss and gradient terms per sorted row
    s += rn_scan_ws_bytes(B);
    return s;
}

#define RN_DISPATCH_FLAGS(KERNEL, ...)                                                          \
    switch (flags & 3) {                                                                        \
        case 0: hipLaunchKernelGGL(KERNEL<0>, G, RN_PW_T, 0, st, __VA_ARGS__); break;           \
        case 1: hipLaunchKernelGGL(KERNEL<1>, G, RN_PW_T, 0, st, __VA_ARGS__); break;           \
        case 2: hipLaunchKernelGGL(KERNEL<2>, G, RN_PW_T, 0, st, __VA_ARGS__); break;           \
        default: hipLaunchKernelGGL(KERNEL<3>, G, RN_PW_T, 0, st, __VA_ARGS__); break;          \
    }

#define RN_DISPATCH_LONG(MODE, ...)                                                                              \
    switch (flags & 3) {                                                                                         \
        case 0: hipLaunchKernelGGL((k_pair_long<0, MODE>), rn_cdiv(B, 64), 256, 0, st, __VA_ARGS__); break;      \
        case 1: hipLaunchKernelGGL((k_pair_long<1, MODE>), rn_cdiv(B, 64), 256, 0, st, __VA_ARGS__); break;      \
        case 2: hipLaunchKernelGGL((k_pair_long<2, MODE>), rn_cdiv(B, 64), 256, 0, st, __VA_ARGS__); break;      \
        default: hipLaunchKernelGGL((k_pair_long<3, MODE>), rn_cdiv(B, 64), 256, 0, st, __VA_ARGS__); break;     \
    }

// workspace layout: members (B + 1) | per-block loss partials | long-row counts | long-row loss terms | long-row gradient terms
struct PairWs {
    Member* mem;
    double* part;
    int32_t* long_cnt;
    float *long_la, *long_ga;
};
static inline PairWs pair_ws(void* ws, size_t ws_bytes, int64_t B) {
    RnCarver c(ws, ws_bytes);
    PairWs p;
    p.mem = c.take<Member>(B + 1);
    const int G = 2 * rn_cdiv(B, RN_PW_T / RN_PW_LPR);
    p.part = c.take<double>(G > RN_VEC_BLOCKS ? G : RN_VEC_BLOCKS);
    p.long_cnt = c.take<int32_t>(B);
    p.long_la = c.take<float>(B);
    p.long_ga = c.take<float>(B);
    return p;
}

// The Member array lives at the start of the pairwise workspace; every entry point re-packs it (B x 16 B).
static int pack_members(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order, int64_t B,
                        Member* mem, hipStream_t st, int64_t* zero_b = nullptr, int64_t* zero_1 = nullptr) {
    hipLaunchKernelGGL(k_pack_members, rn_cdiv(B, 256), 256, 0, st, scores, labels, mask, order, B, mem,
                       (unsigned long long*)zero_b, (unsigned long long*)zero_1);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_pair_count(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                                 const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id, int64_t B,
                                 int flags, int32_t* cnt_row, int64_t* cnt_super, int64_t* n_pair, void* ws,
                                 size_t ws_bytes, void* stream) {
    if (B < 0 || !n_pair) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(n_pair, 0, sizeof(int64_t), st));
        return RECNOW_OK;
    }
    if (!scores || !labels || !order || !seg_id || !seg_first || !super_id || !cnt_row || !cnt_super || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_pairwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    const PairWs pw = pair_ws(ws, ws_bytes, B);
    Member* mem = pw.mem;
    // RECNOW_PAIR_MEMBERS_PACKED: recnow_group_pack_small has packed the members into `ws` and cleared the counters
    int rc = (flags & RECNOW_PAIR_MEMBERS_PACKED) ? RECNOW_OK : pack_members(scores, labels, mask, order, B, mem, st, cnt_super, n_pair);      // also clears cnt_super[0..B) and *n_pair
    if (rc) return rc;
    const int G = rn_cdiv(B, RN_PW_T);
    RN_DISPATCH_LONG(0, mem, seg_id, seg_first, B, 1.f, pw.long_cnt, pw.long_la, pw.long_ga);
    RN_DISPATCH_FLAGS(k_pair_count, mem, seg_id, seg_first, super_id, B, pw.long_cnt, cnt_row, (unsigned long long*)cnt_super,
                      (unsigned long long*)n_pair);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_pair_offsets(const int32_t* cnt_row, int64_t B, int64_t* offsets, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || !offsets) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(offsets, 0, sizeof(int64_t), st));
        return RECNOW_OK;
    }
    if (!cnt_row || !ws) return RECNOW_EINVAL;
    return rn_exclusive_scan_i32_i64(cnt_row, offsets, B, ws, ws_bytes, st);
}

extern "C" int recnow_pair_emit(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                                const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, const int64_t* offsets,
                                int32_t* pos_idx, int32_t* neg_idx, int64_t capacity, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || capacity < 0) return RECNOW_EINVAL;
    if (B == 0 || capacity == 0) return RECNOW_OK;
    if (!scores || !labels || !order || !seg_id || !seg_first || !offsets || !pos_idx || !neg_idx || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_pairwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Member* mem = (Member*)ws;
    int rc = pack_members(scores, labels, mask, order, B, mem, st);
    if (rc) return rc;
    const int G = rn_cdiv(B, RN_PW_T);
    RN_DISPATCH_FLAGS(k_pair_emit, mem, seg_id, seg_first, B, offsets, pos_idx, neg_idx, capacity);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_pair_bpr_fwdbwd(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                                      const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id,
                                      const int64_t* cnt_super, const int64_t* n_pair, int64_t B, int flags, float factor,
                                      float power, int reduce_mean, float* loss, float* dscores, void* ws, size_t ws_bytes,
                                      void* stream) {
    if (B < 0 || !loss) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return RECNOW_OK;
    }
    if (!scores || !labels || !order || !seg_id || !seg_first || !super_id || !n_pair || !dscores || !ws) return RECNOW_EINVAL;
    if (power != 0.f && !cnt_super) return RECNOW_EINVAL;
    // the fused walk visits each unordered candidate {i, j} once and assumes at most ONE of (i,j) / (j,i) is a pair, which the
    // label (label_i > label_j) or wrong-order (s_i < s_j) predicate guarantees; without either both directions would be pairs
    if ((flags & (RECNOW_PAIR_LABEL_GT | RECNOW_PAIR_WRONG_ORDER)) == 0) return RECNOW_EINVAL;
    if (ws_bytes < recnow_pairwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    const PairWs pw = pair_ws(ws, ws_bytes, B);
    Member* mem = pw.mem;
    const int G = rn_cdiv(B, RN_PW_T);
    double* part = pw.part;
    // RECNOW_PAIR_MEMBERS_PACKED: `ws` still holds the members recnow_pair_count packed from these very inputs
    int rc = (flags & RECNOW_PAIR_MEMBERS_PACKED) ? RECNOW_OK : pack_members(scores, labels, mask, order, B, mem, st);
    if (rc) return rc;
    RN_DISPATCH_LONG(1, mem, seg_id, seg_first, B, factor, pw.long_cnt, pw.long_la, pw.long_ga);
    RN_DISPATCH_FLAGS(k_pair_bpr, mem, seg_id, seg_first, super_id, (const unsigned long long*)cnt_super,
                      (const unsigned long long*)n_pair, B, factor, power, reduce_mean, pw.long_la, pw.long_ga, part, dscores);
    hipLaunchKernelGGL(k_loss_finalize, 1, 1024, 0, st, part, G, (const unsigned long long*)n_pair, (int64_t)0, reduce_mean, loss);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// part_out != NULL (library-internal: the loss stage of recnow_dcn_mix_step): the per-workgroup loss partials are NOT summed here -- the caller's next
// kernel (k_step_dscore, 1024 threads: the same sum in the same order as k_loss_finalize) does it, one launch less per step; *part_out / *nparts_out
// receive the partials.  B must be > 0 then.
int rn_pair_bpr_onepass(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                        const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, float factor, int reduce_mean,
                        float* loss, float* dscores_unnorm, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream,
                        const double** part_out, int* nparts_out) {
    if (B < 0 || !loss || !n_pair) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        RN_HIP(hipMemsetAsync(n_pair, 0, sizeof(int64_t), st));
        return RECNOW_OK;
    }
    if (!scores || !labels || !order || !seg_id || !seg_first || !dscores_unnorm || !ws) return RECNOW_EINVAL;
    if ((flags & (RECNOW_PAIR_LABEL_GT | RECNOW_PAIR_WRONG_ORDER)) == 0) return RECNOW_EINVAL;      // as recnow_pair_bpr_fwdbwd
    if (ws_bytes < recnow_pairwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    const PairWs pw = pair_ws(ws, ws_bytes, B);
    const int G = rn_cdiv(B, RN_PW_T);
    static const bool one_launch = []() { const char* e = getenv("RECNOW_PAIR_ALL"); return !e || e[0] != '0'; }();      // A/B switch: 0 = k_pair_long + k_pair_one(4)
    // RN_PAIR_UNPACKED (internal: the step's loss stage): no pack launch -- the walk's workgroups fill their LDS stages from the inputs through `order`;
    // RN_PAIR_NPAIR_ZEROED: an earlier launch of the caller has cleared *n_pair (the step's grouping launch), else a fill does it here.
    const bool unpacked = one_launch && (flags & RN_PAIR_UNPACKED) != 0 && !(flags & RECNOW_PAIR_MEMBERS_PACKED);
    // RECNOW_PAIR_MEMBERS_PACKED: recnow_group_pack_small has packed the members into `ws` and cleared *n_pair
    int rc = RECNOW_OK;
    if (unpacked) {
        if (!(flags & RN_PAIR_NPAIR_ZEROED)) RN_HIP(hipMemsetAsync(n_pair, 0, sizeof(int64_t), st));
    } else if (!(flags & RECNOW_PAIR_MEMBERS_PACKED)) {
        rc = pack_members(scores, labels, mask, order, B, pw.mem, st, nullptr, n_pair);
    }
    if (rc) return rc;
    // Four lanes per row while the one-lane grid would leave most of the chip idle (B <= 32 768: at most 128 workgroups).  Measured (tools/layer_bench.py, GPU
    // time of the loss fwd+bwd, one box): B = 8192 / 128 groups 66 -> 62 us, Zipf-skewed 130 -> 117 us, the 8192-row step 0.609 -> 0.602 ms; B = 65 536 / 1024 groups
    // 81 -> 86 us (four times the block sums and staging for a grid that already fills the chip): one lane per row stays there.  RECNOW_PAIR_LPR=1 / =4 force a form.
    static const int lpr_env = []() { const char* e = getenv("RECNOW_PAIR_LPR"); return e ? atoi(e) : 0; }();
    const bool quad = lpr_env == 4 || (lpr_env != 1 && B <= 32768);
    int nparts = G;
    if (one_launch) {
        const int g_one = rn_cdiv(B, quad ? RN_PW_T / RN_PW_LPR : RN_PW_T), g_all = g_one + rn_cdiv(B, 64);
        nparts = g_all;
#define RN_PAIR_ALL(F)                                                                                                                              \
    do {                                                                                                                                            \
        if (unpacked) {                                                                                                                             \
            const MemberSrc<true> src{pw.mem, scores, labels, mask, order};                                                                         \
            if (quad) hipLaunchKernelGGL((k_pair_all<F, RN_PW_LPR, true>), g_all, RN_PW_T, 0, st, src, seg_id, seg_first, B, factor, pw.part,         \
                                         (unsigned long long*)n_pair, dscores_unnorm, g_one);                                                        \
            else hipLaunchKernelGGL((k_pair_all<F, 1, true>), g_all, RN_PW_T, 0, st, src, seg_id, seg_first, B, factor, pw.part,                      \
                                    (unsigned long long*)n_pair, dscores_unnorm, g_one);                                                             \
        } else {                                                                                                                                    \
            const MemberSrc<false> src{pw.mem, nullptr, nullptr, nullptr, nullptr};                                                                 \
            if (quad) hipLaunchKernelGGL((k_pair_all<F, RN_PW_LPR, false>), g_all, RN_PW_T, 0, st, src, seg_id, seg_first, B, factor, pw.part,        \
                                         (unsigned long long*)n_pair, dscores_unnorm, g_one);                                                        \
            else hipLaunchKernelGGL((k_pair_all<F, 1, false>), g_all, RN_PW_T, 0, st, src, seg_id, seg_first, B, factor, pw.part,                     \
                                    (unsigned long long*)n_pair, dscores_unnorm, g_one);                                                             \
        }                                                                                                                                           \
    } while (0)
        switch (flags & 3) {
            case 0: RN_PAIR_ALL(0); break;
            case 1: RN_PAIR_ALL(1); break;
            case 2: RN_PAIR_ALL(2); break;
            default: RN_PAIR_ALL(3); break;
        }
#undef RN_PAIR_ALL
    } else {
        RN_DISPATCH_LONG(2, pw.mem, seg_id, seg_first, B, factor, pw.long_cnt, pw.long_la, pw.long_ga);
        if (quad) {
            const int G = rn_cdiv(B, RN_PW_T / RN_PW_LPR);          // (shadows the one-lane grid inside the dispatch macro)
            nparts = G;
            RN_DISPATCH_FLAGS(k_pair_one4, pw.mem, seg_id, seg_first, B, factor, pw.long_cnt, pw.long_la, pw.long_ga, pw.part,
                              (unsigned long long*)n_pair, dscores_unnorm);
        } else {
            RN_DISPATCH_FLAGS(k_pair_one, pw.mem, seg_id, seg_first, B, factor, pw.long_cnt, pw.long_la, pw.long_ga, pw.part,
                              (unsigned long long*)n_pair, dscores_unnorm);
        }
    }
    if (part_out) {
        *part_out = pw.part;
        *nparts_out = nparts;
    } else {
        hipLaunchKernelGGL(k_loss_finalize, 1, 1024, 0, st, pw.part, nparts, (const unsigned long long*)n_pair, (int64_t)0, reduce_mean, loss);
    }
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_pair_bpr_onepass(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                                      const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, float factor, int reduce_mean,
                                      float* loss, float* dscores_unnorm, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream) {
    return rn_pair_bpr_onepass(scores, labels, mask, order, seg_id, seg_first, B, flags & ~(RN_PAIR_UNPACKED | RN_PAIR_NPAIR_ZEROED), factor, reduce_mean, loss,
                               dscores_unnorm, n_pair, ws, ws_bytes, stream, nullptr, nullptr);
}

extern "C" int recnow_pair_scale_grad(const float* dscores_unnorm, const float* g, const int64_t* n_pair, float eps, int64_t B, float* out,
                                      void* stream) {
    if (B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!dscores_unnorm || !g || !out) return RECNOW_EINVAL;
    int G = rn_cdiv(B, 256);
    if (G > 2048) G = 2048;
    hipLaunchKernelGGL(k_pair_scale_grad, G, 256, 0, (hipStream_t)stream, dscores_unnorm, g, (const unsigned long long*)n_pair, eps, B, out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_bpr_loss_fwdbwd(const float* pos, const float* neg, const float* weights, int64_t P, float factor,
                                      int reduce_mean, float* loss, float* dpos, void* ws, size_t ws_bytes, void* stream) {
    if (P < 0 || !loss) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (P == 0) {
        RN_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return RECNOW_OK;
    }
    if (!pos || !neg || !dpos || !ws) return RECNOW_EINVAL;
    if (ws_bytes < rn_align(RN_VEC_BLOCKS * sizeof(double))) return RECNOW_EWORKSPACE;
    double* part = (double*)ws;
    int G = rn_cdiv(P, 256);
    if (G > RN_VEC_BLOCKS) G = RN_VEC_BLOCKS;
    hipLaunchKernelGGL(k_bpr_vec, G, 256, 0, st, pos, neg, weights, P, factor, reduce_mean, part, dpos);
    hipLaunchKernelGGL(k_loss_finalize, 1, 1024, 0, st, part, G, (const unsigned long long*)nullptr, P, reduce_mean, loss);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_pair_mask_dense(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                                      int only_upper_band, uint8_t* mask_out, void* stream) {
    if (B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!order || !seg_id || !seg_first || !mask_out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_pair_mask_dense, rn_cdiv(B, 256), 256, 0, (hipStream_t)stream, order, seg_id, seg_first, B,
                       only_upper_band, mask_out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_occurance_power_weight(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                                             float power, float* w_out, void* stream) {
    if (B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!order || !seg_id || !seg_first || !w_out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_occ_weight, rn_cdiv(B, 256), 256, 0, (hipStream_t)stream, order, seg_id, seg_first, B, power, w_out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}


// Front end of the pairwise loss for a small batch (B <= 8192, ONE float32 or int32 group tensor) in one launch: canonical keys,
// stable sort, segments AND the packed member records + cleared counters that recnow_pair_count / recnow_pair_bpr_fwdbwd expect
// when called with RECNOW_PAIR_MEMBERS_PACKED on the same workspace.  Equivalent to recnow_group_keys + recnow_group_segments +
// the packing pass of recnow_pair_count.
// The single workgroup needs ps_lds_bytes() (~139 KB) of dynamic LDS: asked of the device once (a partition or a part with less LDS
// per workgroup answers 0 here and the callers take the general route), and the kernel's LDS limit is raised once, not per call.
static bool pack_small_lds_ok() {
    static int cached[64];      // 0 = not asked, 1 = fits, 2 = does not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!cached[dev]) {
        // raising the kernel's dynamic-LDS limit IS the query: the runtime refuses a size above what a workgroup of this device
        // may own (the device attribute only reports the 64 KB default limit)
        const bool ok = hipFuncSetAttribute((const void*)k_group_pack_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ps_lds_bytes()) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        cached[dev] = ok ? 1 : 2;
    }
    return cached[dev] == 1;
}
extern "C" int recnow_pairwise_small_supported(int64_t B, int key_dtype) {
    if (!(B >= 0 && B <= GS_MAXB && (key_dtype == RECNOW_KEY_F32 || key_dtype == RECNOW_KEY_I32))) return 0;
    return (B == 0 || pack_small_lds_ok()) ? 1 : 0;
}
extern "C" int recnow_group_pack_small(const void* groups, int key_dtype, const float* labels, const float* scores,
                                       const uint8_t* mask, int64_t B, int32_t* order, int32_t* seg_id, int32_t* seg_first,
                                       int32_t* super_id, int32_t* n_seg, int64_t* cnt_super, int64_t* n_pair, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (B < 0 || !seg_first || !n_seg || !n_pair) return RECNOW_EINVAL;
    if (!recnow_pairwise_small_supported(B, key_dtype)) return RECNOW_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        hipLaunchKernelGGL(k_seg_empty2, 1, 1, 0, st, seg_first, n_seg, (unsigned long long*)n_pair);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (!groups || !labels || !scores || !order || !seg_id || !super_id || !cnt_super || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_pairwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    const PairWs pw = pair_ws(ws, ws_bytes, B);
    const size_t lds = ps_lds_bytes();            // the limit was raised by recnow_pairwise_small_supported above
    hipLaunchKernelGGL(k_group_pack_small, 1, GS_T, lds, st, groups, key_dtype, labels, scores, mask, (int)B, order, seg_id, seg_first,
                       super_id, n_seg, pw.mem, (unsigned long long*)cnt_super, (unsigned long long*)n_pair);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}


// ---- the default pairwise_loss in one call (include/recnow.h: recnow_pairwise_loss) ----------------------------------------------
struct PairLossWs {
    int32_t *order, *seg_id, *seg_first, *super_id, *n_seg;
    int64_t* cnt_super;
    uint32_t* words;
    uint8_t* solo;
    float* dsu;
    void* grp; size_t grp_bytes;
    void* pair; size_t pair_bytes;
    int n_words;
    size_t total;
};
static PairLossWs pair_loss_carve(void* ws, int64_t B, int key_dtype) {
    PairLossWs w;
    RnCarver c(ws, 0);
    const int64_t Bp = B > 0 ? B : 1;
    w.n_words = recnow_key_words(key_dtype);
    if (w.n_words < 1) w.n_words = 1;
    w.order = c.take<int32_t>(Bp);
    w.seg_id = c.take<int32_t>(Bp);
    w.seg_first = c.take<int32_t>(Bp + 1);
    w.super_id = c.take<int32_t>(Bp);
    w.n_seg = c.take<int32_t>(2);
    w.cnt_super = c.take<int64_t>(Bp);
    w.words = c.take<uint32_t>((size_t)w.n_words * Bp);
    w.solo = c.take<uint8_t>(Bp);
    w.dsu = c.take<float>(Bp);
    w.grp_bytes = recnow_group_segments_workspace_bytes(B, w.n_words);
    w.grp = c.take<char>(w.grp_bytes);
    w.pair_bytes = recnow_pairwise_workspace_bytes(B);
    w.pair = c.take<char>(w.pair_bytes);
    w.total = c.off;
    return w;
}
extern "C" size_t recnow_pairwise_loss_workspace_bytes(int64_t B, int key_dtype) {
    if (B < 0) return 0;
    return pair_loss_carve(nullptr, B, key_dtype).total + 256;
}
// dscores = unnormalised pair gradients x 1 / (P + eps) (mean) or as they are (sum); {loss, (float) P} for the caller
__global__ void __launch_bounds__(256)
k_pair_norm_grad(const float* __restrict__ d, const unsigned long long* __restrict__ n_pair, int reduce_mean, float eps, int64_t B,
                 float* __restrict__ out, float* __restrict__ loss, float* __restrict__ out2, const int32_t* __restrict__ n_seg) {
    const float P = (float)(*n_pair);
    // n_seg[0] < 0: the cooperative grouping launch timed out (scan_sort.hip) and left the identity grouping: NaN, never a silent zero
    const bool bad = n_seg[0] < 0;
    const float sc = bad ? __int_as_float(0x7fc00000) : (reduce_mean ? 1.f / (P + eps) : 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (bad) loss[0] = sc;
        if (out2) { out2[0] = bad ? sc : loss[0]; out2[1] = P; }
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) out[i] = d[i] * sc;
}
struct RnTileFwd;
int rn_group_mid_raw(const void* group, int dtype, int64_t B, uint8_t* solo, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                     int32_t* n_seg, void* ws, size_t ws_bytes, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed);       // scan_sort.hip
extern "C" int recnow_pairwise_loss(const void* groups, int key_dtype, const float* labels, const float* scores, const uint8_t* mask,
                                    int64_t B, int flags, float factor, int reduce_mean, float* loss, int64_t* n_pair, float* out2,
                                    float* dscores, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || !loss || !n_pair) return RECNOW_EINVAL;
    if (recnow_key_words(key_dtype) < 1) return RECNOW_EINVAL;
    if ((flags & (RECNOW_PAIR_LABEL_GT | RECNOW_PAIR_WRONG_ORDER)) == 0 || (flags & ~3)) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        RN_HIP(hipMemsetAsync(n_pair, 0, sizeof(int64_t), st));
        if (out2) RN_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!groups || !labels || !scores || !dscores || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_pairwise_loss_workspace_bytes(B, key_dtype)) return RECNOW_EWORKSPACE;
    const PairLossWs w = pair_loss_carve(ws, B, key_dtype);
    int rc;
    int f = flags;
    if (recnow_pairwise_small_supported(B, key_dtype)) {
        if ((rc = recnow_group_pack_small(groups, key_dtype, labels, scores, mask, B, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.cnt_super,
                                          n_pair, w.pair, w.pair_bytes, stream)))
            return rc;
        f |= RECNOW_PAIR_MEMBERS_PACKED;
    } else if ((rc = rn_group_mid_raw(groups, key_dtype, B, w.solo, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp, w.grp_bytes, st, nullptr, nullptr, nullptr, nullptr)) !=
               RECNOW_EUNSUPPORTED) {
        // (one float32 / int32 id tensor: keys and solo flags formed inside the cooperative grouping launch, scan_sort.hip)
        if (rc) return rc;
    } else {
        RN_HIP(hipMemsetAsync(w.solo, 0, (size_t)B, st));
        if ((rc = recnow_group_keys(groups, key_dtype, B, w.words, w.solo, stream))) return rc;
        if ((rc = recnow_group_segments(w.words, w.solo, B, w.n_words, w.n_words, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp, w.grp_bytes,
                                        stream)))
            return rc;
    }
    if ((rc = recnow_pair_bpr_onepass(scores, labels, mask, w.order, w.seg_id, w.seg_first, B, f, factor, reduce_mean, loss, w.dsu, n_pair, w.pair,
                                      w.pair_bytes, stream)))
        return rc;
    int G = rn_cdiv(B, 256);
    if (G > 2048) G = 2048;
    hipLaunchKernelGGL(k_pair_norm_grad, G, 256, 0, st, w.dsu, (const unsigned long long*)n_pair, reduce_mean, 1.0e-10f, B, dscores, loss, out2, w.n_seg);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
