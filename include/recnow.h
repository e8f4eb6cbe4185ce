/*
 * librecnow_hip.so -- C ABI of the MI355X (gfx950) implementation of rec_now's in-batch ranking-loss and
 * feature-interaction hot path.
 *
 * The reference (ChaoLiangTHU/rec_now) is pure Python on TensorFlow: it has NO FFI/plugin layer, so there is no
 * reference C interface to copy.  Each entry point below therefore cites the reference PYTHON function/lines whose
 * arithmetic it replaces; the Python mirror of the reference's own call signatures lives in rec_now_amd/ and reaches
 * these symbols through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory (hipMalloc'ed / torch CUDA tensor storage) unless
 *     the parameter name ends in _host.  The caller owns every buffer; the library never allocates or frees.
 *   - every entry point is asynchronous on `stream` (a hipStream_t passed as void*), stateless and re-entrant -- with the three
 *     process-wide settings and the one host-side note listed under "Library state" below.
 *   - return value: 0 = ok, negative = RECNOW_E*, positive = a forwarded hipError_t.  No C++ exception crosses.
 *   - scratch memory: `*_workspace_bytes(...)` is queried first, the caller passes `ws`/`ws_bytes`.
 *   - all matrices are row-major, fp32 unless stated; index outputs are int32, counts are int64.
 *   - reductions are deterministic (no floating-point atomics anywhere): same inputs -> same bits.
 */
#ifndef RECNOW_H_
#define RECNOW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECNOW_OK 0
#define RECNOW_EINVAL (-1)       /* bad argument (null pointer, negative size, unsupported combination) */
#define RECNOW_EWORKSPACE (-2)   /* ws_bytes smaller than *_workspace_bytes() */
#define RECNOW_EUNSUPPORTED (-3) /* shape outside what the kernels implement (documented per function) */

/* activations fused into kernels (keras names: None/'linear', 'relu', 'tanh', 'sigmoid') */
#define RECNOW_ACT_LINEAR 0
#define RECNOW_ACT_RELU 1
#define RECNOW_ACT_TANH 2
#define RECNOW_ACT_SIGMOID 3

/* group-id dtypes accepted by recnow_group_keys */
#define RECNOW_KEY_F32 0
#define RECNOW_KEY_F64 1
#define RECNOW_KEY_I32 2
#define RECNOW_KEY_I64 3

/* pair predicate flags */
#define RECNOW_PAIR_LABEL_GT 1    /* keep (i,j) only if label_i > label_j   (pairwise_loss_from_batch.py:189)     */
#define RECNOW_PAIR_WRONG_ORDER 2 /* keep (i,j) only if score_i < score_j   (pairwise_loss_from_batch.py:197-203) */
#define RECNOW_PAIR_MEMBERS_PACKED 256 /* recnow_pair_bpr_fwdbwd (and recnow_pair_count after recnow_group_pack_small): the workspace is the one recnow_pair_count just
                                          used with the same scores / labels / mask / order -- its packed rows are reused */

int recnow_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Grouping: replaces the dense (B,B) same-group mask of rec_now/rec_block/pairwise_loss_from_batch.py:16-40,43-74
 * and tf.unique_with_counts of listwise_loss_from_batch.py:109 by canonical keys + stable radix sort + segments.
 * ---------------------------------------------------------------------------------------------------------- */

/* Number of 32-bit key words one group tensor of `dtype` contributes (1 or 2); <0 on bad dtype. */
int recnow_key_words(int dtype);

/* Canonicalise one group-id tensor into key words.  Float semantics follow `g_i - g_j == 0.0`
 * (pairwise_loss_from_batch.py:33-35): -0.0 == +0.0; NaN and +-inf rows pair with nobody -> solo[i] |= 1.
 * words: [recnow_key_words(dtype)][B] (word-major, most significant word first).  solo: [B], OR-accumulated. */
int recnow_group_keys(const void* group, int dtype, int64_t B, uint32_t* words, uint8_t* solo, void* stream);

size_t recnow_group_segments_workspace_bytes(int64_t B, int n_words);

/* Stable LSD radix sort of rows by the composite key (all words, lexicographic) and segment detection.
 *   words        [n_words][B]  composite key, words of groups[0] first (n_words_first of them)
 *   solo         [B]           rows that pair with nobody (each becomes a segment of its own)
 * outputs
 *   order        [B]    original row index at sorted position k (ascending row index inside a segment)
 *   seg_id       [B]    segment (= group) index of sorted position k, segments numbered in sorted order
 *   seg_first    [B+1]  first sorted position of segment g; seg_first[n_seg] = B
 *   super_id     [B]    index of the groups[0]-only segment containing sorted position k
 *   n_seg        [2]    {number of segments, number of groups[0]-only segments}
 */
int recnow_group_segments(const uint32_t* words, const uint8_t* solo, int64_t B, int n_words, int n_words_first,
                          int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id, int32_t* n_seg,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * In-batch pairwise loss: rec_now/rec_block/pairwise_loss_from_batch.py
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_pairwise_workspace_bytes(int64_t B);

/* Per-row count of valid partners (replaces the mask algebra of pairwise_loss_from_batch.py:254-264).
 *   valid(i,j) = same segment, i != j, mask_i && mask_j, [label_i > label_j], [score_i < score_j]
 *   cnt_row   [B]  c_i by ORIGINAL row index
 *   cnt_super [B]  number of surviving pairs whose positive row lies in groups[0]-segment s  (:282-291)
 *   n_pair    [1]
 * mask may be NULL (all rows valid). */
int recnow_pair_count(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                      const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id, int64_t B, int flags,
                      int32_t* cnt_row, int64_t* cnt_super, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream);

/* Exclusive scan of cnt_row over ORIGINAL row order -> offsets[B+1] (offsets[B] = n_pair). */
int recnow_pair_offsets(const int32_t* cnt_row, int64_t B, int64_t* offsets, void* ws, size_t ws_bytes, void* stream);

/* Materialise the pair list in the reference's order: row-major over the dense mask = ascending i, then ascending j
 * (tf.boolean_mask of the flattened (B,B) mask, pairwise_loss_from_batch.py:217,272-273).  Bit-exact integer path.
 * pos_idx/neg_idx: [capacity]; pairs beyond capacity are not written. */
int recnow_pair_emit(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                     const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, const int64_t* offsets,
                     int32_t* pos_idx, int32_t* neg_idx, int64_t capacity, void* ws, size_t ws_bytes, void* stream);

/* Fused BPR/logistic pairwise loss + gradient without materialising pairs (pairwise_loss_from_batch.py:96-127 applied
 * to the pair set of :254-274; occurrence weights of :130-151,282-291).
 *   loss   = sum_p w_p * softplus(-factor*(s_i - s_j)) / (float(P) + 1e-10)      (or the raw sum if !reduce_mean)
 *   w_p    = cnt_super[super(i_p)] ** power   (power == 0 -> 1)
 *   dscores[k] = d loss / d scores[k]   (pair set and weights are constants, :264,:270)
 * loss: [1] fp32, dscores: [B].
 * flags must contain RECNOW_PAIR_LABEL_GT or RECNOW_PAIR_WRONG_ORDER (else RECNOW_EINVAL): the fused walk relies on at most
 * one of (i,j) / (j,i) being a pair.  Pair sets without either predicate go through recnow_pair_emit + recnow_bpr_loss_fwdbwd. */
int recnow_pair_bpr_fwdbwd(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                           const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id,
                           const int64_t* cnt_super, const int64_t* n_pair, int64_t B, int flags, float factor,
                           float power, int reduce_mean, float* loss, float* dscores, void* ws, size_t ws_bytes,
                           void* stream);

/* The same loss WITHOUT occurrence weights (click_occurance_power == 0, the reference's default) in one walk per row: pair counts and
 * BPR terms come out of the same pass (no recnow_pair_count before it).  n_pair (out, [1]) = P; loss = sum / (P + 1e-10) or the raw
 * sum; dscores_unnorm[k] = d(SUM of pair losses) / d scores[k], i.e. NOT divided by P: multiply the incoming gradient in with
 * recnow_pair_scale_grad, which applies 1 / (P + eps) as well (n_pair = NULL there for the raw sum).  flags as recnow_pair_bpr_fwdbwd
 * (RECNOW_PAIR_MEMBERS_PACKED after recnow_group_pack_small on the same ws). */
int recnow_pair_bpr_onepass(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                            const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, float factor, int reduce_mean,
                            float* loss, float* dscores_unnorm, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream);
/* out[i] = dscores_unnorm[i] * g[0] / (float(*n_pair) + eps); g a DEVICE scalar (the gradient of the loss); n_pair == NULL: no division. */
int recnow_pair_scale_grad(const float* dscores_unnorm, const float* g, const int64_t* n_pair, float eps, int64_t B, float* out,
                           void* stream);

/* pairwise_loss(outputs, labels, groups) with the reference's defaults (pairloss_func = bpr_loss_func, click_occurance_power = 0,
 * one group tensor; rec_block/pairwise_loss_from_batch.py:228-279) as ONE call: grouping (the single-launch front end when
 * recnow_pairwise_small_supported, else keys + radix sort + segments), the one-walk loss and the gradient
 *   dscores[k] = d loss / d scores[k]   (already divided by P + 1e-10 when reduce_mean),
 * so that a host framework's backward is one multiply by the incoming gradient.  One workspace holds every intermediate
 * (recnow_pairwise_loss_workspace_bytes); out2 (2 floats, optional) receives {loss, (float) P} beside loss / n_pair.
 * flags: RECNOW_PAIR_LABEL_GT [| RECNOW_PAIR_WRONG_ORDER].  mask may be NULL. */
size_t recnow_pairwise_loss_workspace_bytes(int64_t B, int key_dtype);
int recnow_pairwise_loss(const void* groups, int key_dtype, const float* labels, const float* scores, const uint8_t* mask, int64_t B,
                         int flags, float factor, int reduce_mean, float* loss, int64_t* n_pair, float* out2, float* dscores, void* ws,
                         size_t ws_bytes, void* stream);

/* Front end of the loss for a SMALL batch in one launch (BASELINE config 2: pairwise_loss_from_batch at B = 8192): canonical keys
 * of ONE float32 / int32 group tensor (key_dtype RECNOW_KEY_F32 / RECNOW_KEY_I32), stable sort and segments by a single
 * 1024-thread workgroup on LDS-resident keys, B <= 8192 (recnow_pairwise_small_supported), plus what recnow_pair_count would do
 * first: the packed member records in `ws` (recnow_pairwise_workspace_bytes) and cleared cnt_super[0..B) / *n_pair.  Follow with
 * recnow_pair_count and recnow_pair_bpr_fwdbwd on the SAME ws with RECNOW_PAIR_MEMBERS_PACKED in flags.  Same outputs as
 * recnow_group_keys + recnow_group_segments on these ids. */
int recnow_pairwise_small_supported(int64_t B, int key_dtype);
int recnow_group_pack_small(const void* groups, int key_dtype, const float* labels, const float* scores, const uint8_t* mask,
                            int64_t B, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id, int32_t* n_seg,
                            int64_t* cnt_super, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream);

/* bpr_loss_func on explicit (P,) vectors (pairwise_loss_from_batch.py:96-127).  weights may be NULL.
 * dpos = d loss/d outputs_pos, dneg = -dpos.  P may be 0 (loss = 0). */
int recnow_bpr_loss_fwdbwd(const float* pos, const float* neg, const float* weights, int64_t P, float factor,
                           int reduce_mean, float* loss, float* dpos, void* ws, size_t ws_bytes, void* stream);

/* Dense (B,B) bool mask, for API parity with generate_pair_mask (pairwise_loss_from_batch.py:43-74).
 * only_upper_band keeps only j == i+1 (band_part(m,0,1) minus the diagonal, :38-39).  mask_out must be zero-filled. */
int recnow_pair_mask_dense(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                           int only_upper_band, uint8_t* mask_out, void* stream);

/* occurance_power_weight (pairwise_loss_from_batch.py:130-151): w[i] = (size of i's segment) ** power. */
int recnow_occurance_power_weight(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                                  float power, float* w_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * In-batch listwise loss: rec_now/rec_block/listwise_loss_from_batch.py:89-173 on sorted segments (no (G,B) matrix).
 *   valid group g: has a label > th and a (label - th) < 0 (:135-137);  p_i = y_i / sum_g y (:144)
 *   row g of the reference's dense logits = members' logits + (B - n_g) entries equal to pad_logit
 *       (pad_logit = value_of_masked_logit when do_mask_logits, else 0; :139-140)
 *   l_g = lse(row g) * sum_i p_i - sum_i p_i s_i (:167);  loss = mean over valid groups of w_g l_g, NaN -> 0 (:168-172)
 * Segment arrays are sized B and indexed by segment; valid_rank[g] = index of g among the valid groups in
 * FIRST-OCCURRENCE order (the row order of the reference's outputs, tf.unique :109) or -1.
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_listwise_workspace_bytes(int64_t B);
int recnow_listwise_segments(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_first,
                             const int32_t* n_seg, int64_t B, float pos_neg_th, float pad_logit, int32_t* seg_valid,
                             float* seg_lse, float* seg_ysum, float* seg_psum, float* seg_pdot, int32_t* valid_rank,
                             int32_t* n_valid, void* ws, size_t ws_bytes, void* stream);
/* loss [1]; dbase[i] = w_g * (softmax_i * psum_g - p_i) (0 outside valid groups): d loss/d logits = dbase / n_valid
 * (do_reduce) or dbase * upstream[row_rank] (per-list losses);  row_rank[i] = valid rank of i's group or -1;
 * group_loss[r] = weighted loss of the r-th valid list.  weights: [n_valid] or NULL. */
int recnow_listwise_loss_fwdbwd(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_id,
                                const int32_t* seg_first, const int32_t* seg_valid, const float* seg_lse,
                                const float* seg_ysum, const float* seg_psum, const float* seg_pdot,
                                const int32_t* valid_rank, const int32_t* n_valid, const float* weights, int64_t B,
                                float* loss, float* dbase, int32_t* row_rank, float* group_loss, void* stream);
/* to_listwise_sample + listwise_loss_via_softmax_cross_entropy_with_logits(do_reduce=True) fused on sorted segments, as ONE call
 * (rec_block/listwise_loss_from_batch.py:89-173): grouping of `groups` (one id tensor), the per-list statistics, the mean loss over
 * the valid lists (0 when there is none: nan_to_zero) and  dlogits[i] = d loss / d logits[i]  (already divided by the number of valid
 * lists), so that a host framework's backward is one multiply.  out2 (2 floats): {loss, (float) number of valid lists}.
 * weights: [>= number of valid lists] in first-occurrence order of the valid groups, or NULL.  One workspace
 * (recnow_listwise_loss_workspace_bytes) holds every intermediate. */
size_t recnow_listwise_loss_workspace_bytes(int64_t B, int key_dtype);
int recnow_listwise_loss(const void* groups, int key_dtype, const float* labels, const float* logits, const float* weights, int64_t B,
                         float pos_neg_th, float pad_logit, float* out2, float* dlogits, void* ws, size_t ws_bytes, void* stream);
/* Dense (n_valid,B) outputs of to_listwise_sample (:131-148) for API parity.  The caller pre-fills mask_out = 0,
 * labels_out = 0, logits_out = pad_logit; members of valid groups are scattered in. */
int recnow_listwise_dense(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_id,
                          const float* seg_ysum, const int32_t* valid_rank, int64_t B, uint8_t* mask_out,
                          float* labels_out, float* logits_out, void* stream);
int recnow_listwise_dense_bwd(const float* ddense, const int32_t* row_rank, int64_t B, float* dlogits, void* stream);
/* tf.nn.softmax_cross_entropy_with_logits over the rows of dense (G,N) matrices (:167): row_loss = lse*sum(p) - p.s;
 * backward dlogits[g][j] = grow[g] * (exp(s - lse_g) * psum_g - p).  HBM-bound streaming over G*N. */
int recnow_softmax_ce_rows_fwd(const float* labels, const float* logits, int64_t G, int64_t N, float* row_loss,
                               float* row_lse, float* row_psum, void* stream);
int recnow_softmax_ce_rows_bwd(const float* labels, const float* logits, const float* row_lse, const float* row_psum,
                               const float* grow, int64_t G, int64_t N, float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * FMLayer: rec_now/layers/fm_layer.py:24-42.   HBM-bound (12*B*F*D bytes fwd+bwd).
 *   y[b] = 0.5 * sum_d [ (sum_f x_f[b][d])^2 - sum_f x_f[b][d]^2 ]
 * fields: DEVICE array of F device pointers, each a contiguous (B,D) fp32 tensor (the reference's list input);
 * y: [B]; S: [B*D] field sums saved for backward (NULL = forward only).
 * ---------------------------------------------------------------------------------------------------------- */
int recnow_fm_fwd(const float* const* fields, int F, int64_t B, int D, float* y, float* S, void* stream);
/* dfields[f][b][d] = gy[b] * (S[b][d] - x_f[b][d]);  dfields: DEVICE array of F device pointers to (B,D) buffers. */
int recnow_fm_bwd(const float* const* fields, float* const* dfields, int F, int64_t B, int D, const float* S,
                  const float* gy, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: every product/accumulate is an fp32 fma, no reduced precision),
 * the dense contraction under MultiDenseLayer (multi_dense_layer.py:90), DCNMixLayer (dcn_mix_layer.py:135-141),
 * MMOELayer / PLELayer experts and gates.  Bound: fp32 MFMA peak (157 TFLOP/s), not HBM.
 *     C[b] = epilogue( opA(A[b]) x opB(B[b]) ),   b = 0 .. batch-1
 *     epilogue(v) = act(v + bias[n]) * emul[m][n]  (+ C[m][n] when accumulate)
 * Operand modes fuse the elementwise producer of a gradient GEMM into the operand load:
 *     MUL:      operand[i] * second[i]                       ACTGRAD:  operand[i] * act'(second[i]), act' through the
 *                                                                      activation OUTPUT (1-y^2, y(1-y), y>0)
 * A batch stride of 0 broadcasts that operand.  K-splitting (deterministic slab reduce) is chosen internally when
 * M*N is small and K is large (weight gradients, K = batch rows).
 * ---------------------------------------------------------------------------------------------------------- */
#define RECNOW_OPMODE_NONE 0
#define RECNOW_OPMODE_MUL 1
#define RECNOW_OPMODE_ACTGRAD 2
#define RECNOW_OPMODE_OUTER 3   /* operand(row,col) = second[row][col / hq] * first[row][col % hq]: CIN's outer product
                                 (cin_layer.py:103) generated in the operand load; row = the operand's non-k index for A
                                 in [M][K] storage / B in [N][K] storage, and the k index for [K][M] / [K][N] storage */

typedef struct recnow_gemm_desc {
    const float* A;  const float* A2; int64_t lda; int64_t a_batch_stride; int a_trans; int a_mode; int a_act; int a_pad;
    const float* B;  const float* B2; int64_t ldb; int64_t b_batch_stride; int b_trans; int b_mode; int b_act; int b_pad;
    float* C; int64_t ldc; int64_t c_batch_stride;
    int M, N, K, batch;               /* logical A: (M,K), B: (K,N), C: (M,N) */
    const float* bias; int64_t bias_batch_stride;
    const float* emul; int64_t lde; int64_t e_batch_stride;
    int act;                          /* RECNOW_ACT_* applied to columns < act_cols (act_cols <= 0: all columns) */
    int act_cols;
    int e_mode; int e_act;            /* e_mode 0/MUL: v *= emul[m][n];  ACTGRAD: v *= act'(emul[m][n]) (e_act) */
    int accumulate;
    int a_hq; int b_hq;               /* OUTER mode: inner width hq (A2/B2 hold the [row][col / hq] factor) */
    int64_t a_ld2; int64_t b_ld2;     /* OUTER mode: row stride of A2 / B2 */
    int c_trans;                      /* 1: store the result transposed, C[n][m] (ldc = row stride of that layout) */
    /* Side product (lean 128x128 kernels, batch 1): sp_r <= 4 extra output columns computed on the VALU from the A tile
     * in LDS:  sp_cx[m*sp_cx_ms + r*sp_cx_rs] = sum_k A'[m][k] * sp_bx[k*sp_bx_ks + r*sp_bx_rs]  (A' = A after its
     * operand mode).  DCN-v2 uses it for the N gate columns so that N*S + N = 130 runs as exactly 128 MFMA columns. */
    const float* sp_bx; float* sp_cx; int64_t sp_bx_ks, sp_bx_rs, sp_cx_ms, sp_cx_rs; int sp_r; int sp_pad;
    /* Rank-R epilogue update (lean 128x128 kernels, batch 1): before bias-activation / emul,
     *   v[m][n] += sum_{r < eu_r <= 4} eu_p[m*eu_pms + r] * eu_q[r*eu_qrs + n*eu_qns]  (K = 128 + 2 as exactly 128). */
    const float* eu_p; const float* eu_q; int64_t eu_pms, eu_qrs, eu_qns; int eu_r; int eu_pad;
    double prof_flops;                /* algorithmic flops of this product for the measurement hook (0: 2*M*N*K*batch);
                                         callers that zero-pad K or move columns to a side product state the true count */
    /* Second output of the same accumulators (short-K persistent kernel only: K <= 512, K % 16 == 0, M, N multiples of
     * 128, A [M][K], batch 1, no bias / activation / transposed store; anything else returns RECNOW_EUNSUPPORTED):
     *   c2_mode 1:  C2[m][n]  = acc                    (DCN-v2 forward keeps O next to y = x * O)
     *   c2_mode 2:  C2[m][n] += acc * E2[m][n]         (DCN-v2 backward: dx += g_l * O_l in the kernel that produces g_l)
     * `acc` is the raw product A B, before emul / accumulate are applied for C. */
    float* C2; int64_t ldc2; const float* E2; int64_t lde2; int c2_mode; int c2_pad;
    /* Elementwise side output of the A stream (lean 128x128 kernels with a side product, A [M][K], a_mode MUL, batch 1,
     * no split-K; written by the workgroups of the first column tile):
     *   as_out[m][k] = A[m][k] * as_in[m][k]      (both with A's leading dimension lda)
     * DCN-v2 backward: the dT2g product streams g = dL/dy anyway and writes dx = g * O on the way. */
    const float* as_in; float* as_out;
    /* More second-output forms of the short-K kernel (same shape rules as c2_mode 1 / 2), used by the model-level fused step
     * "cross layers + scoring head" (recnow_dcn_mix_score_fwd/bwd):
     *   c2_mode 3 (emul required, b_trans 0): C2 = acc; C = acc * emul is NOT stored; instead its row-dot with the column
     *              vector hv (N) leaves as partials hp[m * hp_ld + 2 * (n / 128) + ((n % 128) / 64)], hp_ld >= N / 64: a
     *              Dense(1) head folded into the epilogue of the product that forms its input.
     *   c2_mode 4 (b_trans 1, no emul): C = acc; C2[m][n] = acc * E2[m][n] + rv[m] * cv[n] * E3[m][n] (written, not read). */
    const float* E3; int64_t lde3; const float* rv; const float* cv; const float* hv; float* hp; int hp_ld; int hp_pad;
    /* k_valid (0 = K): the caller guarantees that A's columns / B's rows k_valid .. K-1 are ZERO (a depth padded to the k-tile);
     * the short-K kernel then skips the MFMA steps of the padding (DCN-v2: 130 of 144). */
    int k_valid; int k_pad;
    /* c_perm_s > 0 (split-K products only, batch 1, no transposed store / accumulate; else RECNOW_EINVAL / EUNSUPPORTED): the result
     * is stored as [N / c_perm_s][M][c_perm_s] instead of [M][N], C[((n / s) * M + m) * s + n % s] -- DCN-v2's dU (N, D, S) straight
     * from the product x_l^T dA (D x N*S), without an unpack pass. */
    int c_perm_s; int c_perm_pad;
    /* Fused sub-space forward of DCNMixLayer (rec_now/layers/dcn_mix_layer.py:135-138,146-147; lean 128x128 kernel, batch 1, no split-K):
     * the product is formulated TRANSPOSED -- A = [U | K]^T stored [K][M] with M = N_e * S = 128 (a_trans 1), B = x_l^T stored [N][K]
     * (b_trans 1), so "N" is the batch -- the side product (sp_bx = K, sp_r = 2) is taken from the B tile, and the epilogue continues
     * from the accumulators: H1 = act(acc) -> mid_T1 [row][s]; per expert C = H1 V on the MFMA with the accumulator registers as the A
     * fragments (k pairs (s, s + 4): no LDS round trip), H2 = mid_act_outer(C) -> mid_T2, G = softmax(logits), mid_T2g = [G * H2 | G | 0],
     * logits -> mid_T1[:, 128 + e].  mid_V (2, 64, 64); mid_T1 / mid_T2 / mid_T2g (N x mid_ld).  C is not written. */
    const float* mid_V; float* mid_T1; float* mid_T2; float* mid_T2g; int64_t mid_ld; int mid_act_outer; int mid_pad;
    /* c2_mode 5 / 6 (ABI 4; short-K kernel, K = 144, b_trans 1, no emul / accumulate; C2 unused): the LAST product of a backward pass writes the
     * whole input gradient in one go,
     *   C[m][n] = acc + E2 * E3 [+ E4 * E5] + rv[m] * cv[n] * E6          (elementwise products; c2_mode 6 leaves the bracket out)
     * all five tensors with the leading dimension lde2 (= lde3), every one read once, C written once and never read.  DCN-v2 (round 5): d loss / d x
     * = g_0 + g_1 * O_0 + g_2 * O_1 + dscore (x) w_head * O_2 is accumulated ONCE, by the product that forms g_0, instead of as a read-modify-write in
     * the product of every layer: 8 instead of 10 passes over a (B, D) tensor per step. */
    const float* E4; const float* E5; const float* E6;
    /* a_trans = 0: A stored [M][K] (lda = row stride);  1: stored [K][M]
     * b_trans = 0: B stored [K][N] (ldb = row stride);  1: stored [N][K] */
} recnow_gemm_desc;

size_t recnow_gemm_workspace_bytes(const recnow_gemm_desc* desc_host);
int recnow_gemm(const recnow_gemm_desc* desc_host, void* ws, size_t ws_bytes, void* stream);
/* Process-wide arithmetic of the long-K 128-column products with a side product (the K = 1024 / K = B products of DCNMixLayer):
 *   0 (default) exact fp32 MFMA -- an fp32 fma chain, what TF's fp32 matmul computes up to summation order;
 *   1 "bf16x3": every fp32 operand element split into three bf16 pieces, six bf16 MFMA terms per product, fp32 accumulation:
 *     relative error of a product <= 2^-25, results within the 1e-5 parity bound but NOT bit-identical to mode 0 (opt-in).
 * Also read once from the environment (RECNOW_GEMM_PRECISION=bf16x3).  Shapes without a split kernel run mode 0 regardless. */
int recnow_set_gemm_precision(int mode);
int recnow_get_gemm_precision(void);
/* Process-wide operand staging of the long-K products whose two operands are plain [k][row] tensors (the K = B weight-gradient products dU_l of
 * DCNMixLayer): 0 (default) global -> registers -> LDS, 1 LDS-DMA (global_load_lds_dwordx4, no staging registers).  Same LDS image, same k order:
 * the results are bit-identical; kept as an A/B switch (profiles/r05_glds_ab.md: neutral).  Also read once from RECNOW_GEMM_GLDS=1.  Returns RECNOW_OK. */
int recnow_set_gemm_staging(int mode);
int recnow_get_gemm_staging(void);

/* ------------------------------------------------------------------------------------------------------------
 * MultiDenseLayer: rec_now/layers/multi_dense_layer.py:80-94.   y[n] = act(x[n] @ kernel[n] + bias[n])
 *   x: (B,D) when x_batched == 0 (broadcast to every n, :88-89) else (N,B,D); kernel (N,D,U); bias (N,1,U) or NULL;
 *   y: (N,B,U).  act must be a RECNOW_ACT_* code (other activations are applied by the caller on the linear output).
 * Backward (derived from the forward lines; the reference leaves it to TF autodiff):
 *   dZ = dy * act'(y);  dkernel[n] = x[n]^T dZ[n];  dbias[n] = colsum dZ[n];  dx[n] = dZ[n] kernel[n]^T (summed over n
 *   when x was broadcast).  Any of dx / dkernel / dbias may be NULL (not needed).
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_multi_dense_workspace_bytes(int64_t B, int D, int U, int N);
int recnow_multi_dense_fwd(const float* x, int x_batched, const float* kernel, const float* bias, int64_t B, int D, int U,
                           int N, int act, float* y, void* ws, size_t ws_bytes, void* stream);
int recnow_multi_dense_bwd(const float* x, int x_batched, const float* kernel, const float* y, const float* dy, int64_t B,
                           int D, int U, int N, int act, float* dx, float* dkernel, float* dbias, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Gate mixing of MMOELayer (mmoe_layer.py:109-117) and PLELayer (ple_layer.py:274-293):
 *   g[t][b][:] = softmax(logits[t][b][:]);   out[t][b][:] = sum_n g[t][b][n] * E_n[b][:]
 * experts: DEVICE array of N device pointers, expert n is a contiguous (B,U) matrix (so PLE's "own + shared" expert
 * subsets need no concat copy).  logits, gates: (T,B,N); out: (T,B,U).  HBM-bound.
 * Backward: dexperts[n] (+)= sum_t g*dout;  dlogits = g * (dg - sum_n g*dg), dg[t][b][n] = dout[t][b][:] . E_n[b][:].
 * accumulate_dexperts != 0 adds into dexperts (an expert feeding several gates).
 * ---------------------------------------------------------------------------------------------------------- */
int recnow_moe_mix_fwd(const float* logits, const float* const* experts, int T, int64_t B, int N, int U, float* gates,
                       float* out, void* stream);
int recnow_moe_mix_bwd(const float* gates, const float* const* experts, const float* dout, int T, int64_t B, int N, int U,
                       float* dlogits, float* const* dexperts, int accumulate_dexperts, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * DCNLayer (DCN-v1 cross, reference variant without residual): rec_now/layers/dcn_layer.py:79-103
 *   x_{l+1} = act(x0 * (x_l . w_l) + b_l),  l = 0..L-1;  all L layers fused in one pass over x0 (HBM-bound: 8*B*D
 *   bytes forward, 20*B*D backward incl. the recompute read).  Any L, any D, any alignment: the fused register kernels
 *   cover L <= 4 and D <= 4096 (D % 4 == 0, 16-byte aligned rows; else D <= 1024); other shapes run the same math as
 *   streaming kernels over the per-row scalars (rows kernel + columns kernel, see csrc/dcn.hip).
 *   kernels: (L,D) (row l = kernel_l[:,0]); biases: (L,D) or NULL (use_bias=False); y: (B,D).
 *   csave: optional (B,L) output of the forward, the per-row scalars c_l = x_l . w_l.  Given to the backward, x_l is
 *   elementwise in x0 (one dot-reduce per layer, a wave per row); csave == NULL: the backward recomputes the forward per row
 *   from x0 (nothing saved).  dx (B,D), dkernels (L,D), dbiases (L,D) or NULL.
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_dcn_workspace_bytes(int64_t B, int D, int L);
int recnow_dcn_fwd(const float* x, const float* kernels, const float* biases, int64_t B, int D, int L, int act, float* y,
                   float* csave, void* stream);
int recnow_dcn_bwd(const float* x, const float* kernels, const float* biases, const float* dy, const float* csave, int64_t B,
                   int D, int L, int act, float* dx, float* dkernels, float* dbiases, void* ws, size_t ws_bytes, void* stream);
/* One cross layer with its own layer input (one iteration of the loop at rec_now/layers/dcn_layer.py:91-100):
 *   z = act(x0 * (x_l . w) + b);  c_out (B) = x_l . w is kept for the backward.  Used when the activation between the layers
 *   is a user callable (keras.activations.get accepts any, dcn_layer.py:30 via keras.layers.Dense): act = RECNOW_ACT_LINEAR
 *   here and the callable runs on z.  Any D, any alignment.  w, b: (D); b may be NULL.
 *   backward: dz is the gradient w.r.t. z; z may be NULL when act is linear.  dx0, dxl (B,D) are written (not accumulated);
 *   dw (D), db (D, may be NULL). */
size_t recnow_dcn_step_workspace_bytes(int64_t B, int D);
int recnow_dcn_step_fwd(const float* x0, const float* xl, const float* w, const float* b, int64_t B, int D, int act, float* z,
                        float* c_out, void* stream);
int recnow_dcn_step_bwd(const float* x0, const float* xl, const float* w, const float* z, const float* c, const float* dz,
                        int64_t B, int D, int act, float* dx0, float* dxl, float* dw, float* db, void* ws, size_t ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * DCNMixLayer (DCN-v2 mixture of low-rank experts, reference variant without residual):
 * rec_now/layers/dcn_mix_layer.py:114-151.  Per layer l (x_0 = x):
 *   A = x_l U_n -> act_inner -> (. V_n) -> act_outer -> (. W_n + b_n) -> * x   ;   gates = softmax(x_l K)
 *   x_{l+1} = sum_n gates[:,n] * (...)_n
 * Weight pointer arrays are HOST arrays of L DEVICE pointers:
 *   U[l]: (N,D,S) origin_to_sub_kernels_of_layer{l};  V[l]: (N,S,S) sub_to_sub_...;  W[l]: (N,S,D) sub_to_origin_...;
 *   bias[l]: (1,N,D) bias_of_layer{l};  gate[l]: (D,N) gate_of_layer{l} kernel (Dense, use_bias=False).
 * The two D-sized contractions per layer run on the exact-fp32 MFMA GEMM with the gate logits folded in as extra
 * columns and the gate-weighted bias folded in as extra K rows; `saved` keeps the small (B x ~(N*S+N)) activations
 * and the layer inputs for backward.
 *
 * Library state (the whole of it).
 *   - Process-wide settings: recnow_set_gemm_precision, recnow_set_gemm_staging, recnow_prof_* (one launching thread per process is the model).
 *   - One host-side NOTE per `saved` buffer of the DCN-v2 entries, keyed by the buffer's ADDRESS (a table of 256 entries under a mutex,
 *     csrc/dcnmix.hip): which weight packs / piece planes the forward that filled it left there, because the route rule (batch size,
 *     precision mode, RECNOW_TILE) is read per call and may differ between a forward and its backward.  Contract: every forward call
 *     (recnow_dcn_mix_fwd / _score_fwd / the FORWARD phase of recnow_dcn_mix_step) RE-WRITES the note of its `saved` before any backward can read
 *     it, so a recycled address -- another shape, another allocation -- never meets a stale note; a backward whose `saved` has no note (filled
 *     by another copy of the library, or more than 256 distinct buffers ago) packs what its route needs itself: slower by one pack launch,
 *     never wrong.  The note holds no device data and no pointer is dereferenced through it.
 *   - Thread-local hand-offs between the phases of ONE recnow_dcn_mix_step call (packs written by the GROUP phase's front kernel, the piece
 *     planes of the next product): consumed inside the call that set them.
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_dcn_mix_saved_bytes(int64_t B, int D, int S, int N, int L);
size_t recnow_dcn_mix_workspace_bytes(int64_t B, int D, int S, int N, int L);
int recnow_dcn_mix_fwd(const float* x, const float* const* U_host, const float* const* V_host, const float* const* W_host,
                       const float* const* bias_host, const float* const* gate_host, int64_t B, int D, int S, int N, int L,
                       int act_inner, int act_outer, float* y, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                       void* stream, int need_dx);
/* need_dx: 0 when the caller will not ask for the gradient w.r.t. x (x is data, as under tf.GradientTape.gradient(loss,
 * weights)): the forward then keeps nothing that only dx needs.  recnow_dcn_mix_bwd must be given dx == NULL after such a
 * forward, and may be given dx == NULL after any forward: the products and passes that only feed dx are not launched. */
int recnow_dcn_mix_bwd(const float* x, const float* const* U_host, const float* const* V_host, const float* const* W_host,
                       const float* const* bias_host, const float* const* gate_host, const float* dy, const void* saved,
                       size_t saved_bytes, int64_t B, int D, int S, int N, int L, int act_inner, int act_outer, float* dx,
                       float* const* dU_host, float* const* dV_host, float* const* dW_host, float* const* dbias_host,
                       float* const* dgate_host, void* ws, size_t ws_bytes, void* stream, void* stream2);

/* Model-level fusion of the north-star step: the L cross layers followed by a Dense(1) scoring head,
 *   scores[m] = DCNMix(x)[m] . head_w + head_b        (rec_now/layers/dcn_mix_layer.py:149-150 -> multi_dense_layer.py:90-92
 *                                                       with units = 1, num_dnn = 1)
 * The head is folded into the epilogue of the last layer's output product (the layer output is never stored) and its rank-one
 * gradient dscore (x) head_w is never materialised in the backward (see csrc/dcnmix.hip).  Same saved / workspace sizes as
 * recnow_dcn_mix_fwd/bwd.  recnow_dcn_mix_score_supported: 1 when the shape takes the fused route (N*S and D multiples of 128,
 * B a multiple of 256, S in {32, 64}, L <= 8), else 0 -- the caller then composes recnow_dcn_mix_* with recnow_multi_dense_*.
 * head_w (D), head_b (1) or NULL, scores (B), dscores (B), dhead_w (D), dhead_b (1) or NULL.
 * layer_events_host: optional HOST array of L hipEvent_t (entries may be NULL): event l is recorded when every weight gradient
 * of layer l has been issued (layers are walked L-1 .. 0; the head's gradients are complete at event L-1), so a data-parallel
 * caller can all-reduce a layer's gradients while the lower layers' backward still runs. */
int recnow_dcn_mix_score_supported(int64_t B, int D, int S, int N, int L);
/* 1 when a step / score / layer call of this shape runs the row-block persistent kernels (csrc/dcnmix_tile.hip: two experts of 64,
 * D in {256, 512, 1024}, batches up to 16 384 rows; RECNOW_TILE=0 / =1 and the precision mode are read at every call): the ONE
 * statement of that rule -- callers that place other work around the cross layers (rec_now_amd/step.py: where the grouping of the
 * batch runs) ask here instead of restating it.  2 (round 6): split-precision mode with the FORWARD pass of all cross layers as one
 * row-block launch on the bf16 MFMA (csrc/dcnmix_tile_split.hip; RECNOW_TILE_SPLIT=0 / =1) and the backward pass one launch per product. */
int recnow_dcn_mix_tile_route(int64_t B, int D, int S, int N, int L);
int recnow_dcn_mix_score_fwd(const float* x, const float* const* U_host, const float* const* V_host, const float* const* W_host,
                             const float* const* bias_host, const float* const* gate_host, const float* head_w, const float* head_b,
                             int64_t B, int D, int S, int N, int L, int act_inner, int act_outer, float* scores, void* saved,
                             size_t saved_bytes, void* ws, size_t ws_bytes, void* stream, int need_dx);
int recnow_dcn_mix_score_bwd(const float* x, const float* const* U_host, const float* const* V_host, const float* const* W_host,
                             const float* const* bias_host, const float* const* gate_host, const float* head_w, const float* dscores,
                             const void* saved, size_t saved_bytes, int64_t B, int D, int S, int N, int L, int act_inner,
                             int act_outer, float* dx, float* const* dU_host, float* const* dV_host, float* const* dW_host,
                             float* const* dbias_host, float* const* dgate_host, float* dhead_w, float* dhead_b, void* ws,
                             size_t ws_bytes, void* stream, void* stream2, void* const* layer_events_host);
/* stream2: optional second hipStream_t (NULL or == stream: single-stream).  When given, the weight-gradient products and
 * the dx recompute run on it concurrently with the data-gradient chain on `stream`, ordered by events created and
 * destroyed inside the call; on return all of stream2's work is ordered before later work submitted to `stream`. */

/* The whole north-star step -- x -> L cross layers -> Dense(1) head -> in-batch pairwise (BPR) loss -> every gradient -- enqueued by
 * ONE call per phase, on buffers the caller allocates once (at 8192 rows per GPU, the 8-GPU shard of the metric's batch,
 * the ~60 launches of a step take less GPU time than a host framework needs to enqueue them one by one; a phase is one call, and,
 * because nothing is allocated and no event of the library is recorded inside, each phase can be captured into a HIP graph and
 * replayed).  Reference path: rec_now/layers/dcn_mix_layer.py:114-151 -> multi_dense_layer.py:80-94 ->
 * rec_block/pairwise_loss_from_batch.py:228-279 (pairloss_func = bpr_loss_func, no occurrence weights, optional mask).
 *
 * phases (bit mask, executed in this order within one call):
 *   RECNOW_STEP_GROUP     canonical keys + radix sort + segments of `groups` (score independent: may run on another stream,
 *                         under the forward pass; order it before RECNOW_STEP_LOSS with an event);
 *   RECNOW_STEP_FORWARD   recnow_dcn_mix_score_fwd -> scores;
 *   RECNOW_STEP_LOSS      loss, n_pair, stats, d loss / d scores (normalised by 1 / (P + 1e-10) when reduce_mean, else the
 *                         gradient of the loss SUM: a data-parallel caller divides by the global pair count after its all-reduce);
 *   RECNOW_STEP_BACKWARD  recnow_dcn_mix_score_bwd for the cross layers layer_hi .. layer_lo (descending; the head's gradients
 *                         belong to layer L-1).  A caller that all-reduces per layer issues one call (one graph) per layer.
 * All of a step's phases must use the same descriptor contents and workspace.  Shapes: recnow_dcn_mix_score_supported.
 * Any split of the phases over calls gives the same results; GROUP | FORWARD | LOSS in ONE call is the cheapest form at shard sizes (round 5): the
 * grouping launch then also writes the row-block kernels' weight packs on its spare workgroups and clears the pair counter, and the pair walk
 * fills its LDS stages straight from scores / labels / mask (no pack launch, no fill). */
#define RECNOW_STEP_GROUP 1
#define RECNOW_STEP_FORWARD 2
#define RECNOW_STEP_LOSS 4
#define RECNOW_STEP_BACKWARD 8
typedef struct recnow_dcn_mix_step_desc {
    int64_t B;
    int D, S, N, L;
    int act_inner, act_outer;           /* RECNOW_ACT_* of the cross layers */
    int group_dtype;                    /* RECNOW_KEY_* of `groups` */
    int only_use_wrong_order_pair;      /* pairwise_loss(only_use_wrong_order_pair=...) */
    int reduce_mean;                    /* 1: loss = sum / (P + 1e-10) (the reference's), 0: the sum */
    float factor;                       /* bpr_loss_func(factor=...) */
    const float* x;                     /* (B, D) */
    const float* labels;                /* (B) */
    const void* groups;                 /* (B) ids */
    const uint8_t* mask;                /* (B) or NULL */
    const float* const* U_host;         /* HOST arrays of L DEVICE pointers, as recnow_dcn_mix_fwd */
    const float* const* V_host;
    const float* const* W_host;
    const float* const* bias_host;
    const float* const* gate_host;
    const float* head_w;                /* (D) */
    const float* head_b;                /* (1) or NULL */
    float* scores;                      /* out (B) */
    float* loss;                        /* out (1) */
    int64_t* n_pair;                    /* out (1) */
    float* stats;                       /* out (2) or NULL: {loss as stored, (float) n_pair}, e.g. the tail of a gradient bucket */
    float* dx;                          /* out (B, D) or NULL: x is data */
    float* const* dU_host;              /* HOST arrays of L DEVICE pointers: the gradients */
    float* const* dV_host;
    float* const* dW_host;
    float* const* dbias_host;
    float* const* dgate_host;
    float* dhead_w;                     /* out (D) */
    float* dhead_b;                     /* out (1) or NULL */
    void* ws;                           /* recnow_dcn_mix_step_workspace_bytes; holds the step's state between the phases */
    size_t ws_bytes;
    void* stream2;                      /* optional second hipStream_t of RECNOW_STEP_BACKWARD (as recnow_dcn_mix_score_bwd's stream2): the
                                           weight-gradient products run on it beside the data-gradient chain; NULL: one stream */
    void* const* layer_events_host;     /* optional HOST array of L hipEvent_t (entries may be NULL), as recnow_dcn_mix_score_bwd: event l is
                                           recorded once every weight gradient of layer l has been issued.  Must be NULL while the call
                                           is being captured into a graph (an event recorded inside a capture belongs to the graph). */
    int64_t B_pad;                      /* 0 or == B: every buffer has B rows.  > B (ABI 4): RAGGED batch on the fast route -- the per-rank
                                           batches of whole groups that data parallelism produces (rec_now_amd/dp.py shard_rows_by_group)
                                           are not multiples of 256.  x, dx and scores then have B_pad rows of storage, B_pad a multiple of
                                           256 that recnow_dcn_mix_score_supported accepts; rows [B, B_pad) of x must be ZERO (written once
                                           by the caller); the layers run over B_pad rows, the grouping and the loss over the first B
                                           (labels, groups, mask: B elements), d loss / d scores of the padding rows is zero, so they add
                                           exactly 0.0f to every gradient; scores[B..B_pad) = head_b and dx rows >= B = 0 are written.
                                           Workspace: recnow_dcn_mix_step_workspace_bytes(B_pad, ...). */
} recnow_dcn_mix_step_desc;
size_t recnow_dcn_mix_step_workspace_bytes(int64_t B, int D, int S, int N, int L, int group_dtype);
int recnow_dcn_mix_step(const recnow_dcn_mix_step_desc* desc_host, int phases, int layer_hi, int layer_lo, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * CINLayer (xDeepFM Compressed Interaction Network): rec_now/layers/cin_layer.py:72-122
 *   X_k[b,d,c] = sum_{f,h} W_k[c, f*H_{k-1}+h] * x0[b,d,f] * X_{k-1}[b,d,h]   (X_0 = x0, H_0 = F)          (:103-109)
 *   out = sum over the channels of all kept layers -> (B,D)            (sum_channel, :116-117)
 *       | concat of the kept layers' channels, transposed -> (B, ctot*D) (:119-121); kept = [x0,] X_1..X_L (:112-113)
 * emb: (B, F*D) (concat of the field embeddings, :88-91); weights_host: HOST array of L DEVICE pointers, W_k stored
 * (H_k, F*H_{k-1}) = weight_of_layer{k}[0,0]; hidden_host: HOST int[L].  An implicit GEMM over M = B*D rows on the
 * exact-fp32 MFMA kernel; the (B,D,F,H) outer product is generated in the operand load and never stored.
 * `saved` keeps x0 transposed and X_1..X_L for backward.
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_cin_saved_bytes(int64_t B, int D, int F, const int* hidden_host, int L);
size_t recnow_cin_workspace_bytes(int64_t B, int D, int F, const int* hidden_host, int L);
int recnow_cin_fwd(const float* emb, const float* const* weights_host, int64_t B, int D, int F, const int* hidden_host, int L,
                   int output_input, int sum_channel, float* out, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                   void* stream);
int recnow_cin_bwd(const float* const* weights_host, const float* dout, const void* saved, size_t saved_bytes, int64_t B, int D,
                   int F, const int* hidden_host, int L, int output_input, int sum_channel, float* demb,
                   float* const* dweights_host, void* ws, size_t ws_bytes, void* stream);

/* ============================================================================================================
 * SURVEY.md section 8f rows (neighbours of the hot path).
 * ========================================================================================================== */

/* InnerPNNLayer: rec_now/layers/inner_pnn_layer.py:25-53.  out[b][p] = <x_r[b], x_c[b]> over the F(F-1)/2 field pairs
 * r < c in r-major order (:41-45).  fields / dfields: DEVICE arrays of F device pointers to contiguous (B,D) fp32
 * tensors (the reference's list input); out, dout: (B, F(F-1)/2).  D <= 64.
 * Backward: dx_f[b][:] = sum_{g != f} dout[b][p(f,g)] * x_g[b][:]. */
int recnow_inner_pnn_fwd(const float* const* fields, int F, int64_t B, int D, float* out, void* stream);
int recnow_inner_pnn_bwd(const float* const* fields, float* const* dfields, int F, int64_t B, int D, const float* dout,
                         void* stream);

/* SENETLayer: rec_now/layers/senet_layer.py:93-119.  Fields may have different widths: dims[f], offs[f] (first column of
 * field f in the concatenation, total = sum dims) are DEVICE int32 arrays.
 *   squeeze (:104-110):      sq[b][f] = mean_d x_f[b][d]                                  sq: (B,F)
 *   scale   (:112-117):      out[b][offs[f]+d] = x_f[b][d] * w[b][f]                      w: (B,F) excitation, out: (B,total)
 *   scale_bwd_w:             dw[b][f] = sum_d dout[b][offs[f]+d] * x_f[b][d]
 *   scale_bwd_x:             dx_f[b][d] = dout[b][offs[f]+d] * w[b][f] + dsq[b][f] / dims[f]   (dsq = gradient w.r.t. sq)
 * uniform_d > 0: the caller's promise that every field is uniform_d wide and all tensors are 16-byte aligned (selects the
 * float4 path when uniform_d/4 is a power of two); 0: general widths, staged through LDS (one concatenated row must fit:
 * total <= 12287).
 * The two Dense layers between squeeze and scale (:46-66) are recnow_multi_dense_* with N = 1. */
int recnow_senet_squeeze(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                         float* sq, int uniform_d, void* stream);
int recnow_senet_scale_fwd(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                           const float* w, float* out, int uniform_d, void* stream);
int recnow_senet_scale_bwd_w(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                             const float* dout, float* dw, int uniform_d, void* stream);
int recnow_senet_scale_bwd_x(float* const* dfields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                             const float* w, const float* dout, const float* dsq, int uniform_d, void* stream);

/* SENETLayer in one pass per direction (senet_layer.py:93-119), for equal field widths D (D % 4 == 0, D/4 a power of two,
 * F*D/4 <= 256), F <= 64, hidden width M <= 64 and RECNOW_ACT_* activations: squeeze, the F -> M -> F excitation and the
 * scaling read the fields once.  W1 (F,M), b1 (M), W2 (M,F), b2 (F) row-major (b1 / b2 may be NULL).
 *   fwd: out (B, F*D); saves sq (B,F), h (B,M) = act1(sq W1 + b1), w (B,F) = act2(h W2 + b2) for the backward pass.
 *   bwd: dfields[f] (B,D), dW1, db1, dW2, db2 (db1 / db2 may be NULL); deterministic (fixed-order partial sums).
 * recnow_senet_fused_supported returns 1 when (F, D, M) is inside these limits. */
int recnow_senet_fused_supported(int F, int D, int M);
size_t recnow_senet_fused_workspace_bytes(int64_t B, int F, int M);
int recnow_senet_fused_fwd(const float* const* fields, int F, int D, int64_t B, const float* W1, const float* b1, const float* W2,
                           const float* b2, int M, int act1, int act2, float* out, float* sq_save, float* h_save, float* w_save,
                           void* stream);
int recnow_senet_fused_bwd(const float* const* fields, float* const* dfields, int F, int D, int64_t B, const float* W1,
                           const float* W2, int M, int act1, int act2, const float* dout, const float* sq_save,
                           const float* h_save, const float* w_save, float* dW1, float* db1, float* dW2, float* db2, void* ws,
                           size_t ws_bytes, void* stream);

/* attention_by_dot_product: rec_now/rec_block/attention.py:12-38.  user (B,L,D), doc (B,D), D <= 256:
 *   s_l = <user[b][l], doc[b]> (max(.,0) when filter_neg, :31-32);  mat[b] = sum_l user[b][l] * s_l;  score_sum[b] = sum_l s_l
 * Backward recomputes s; dmat (B,D) / dsum (B) may be NULL (no gradient from that output). */
int recnow_attention_dot_fwd(const float* user, const float* doc, int64_t B, int L, int D, int filter_neg, float* mat,
                             float* score_sum, void* stream);
int recnow_attention_dot_bwd(const float* user, const float* doc, const float* dmat, const float* dsum, int64_t B, int L,
                             int D, int filter_neg, float* duser, float* ddoc, void* stream);

/* focal_crossentropy_loss: rec_now/rec_block/focal_loss.py:12-66.  alpha <= 0 / gamma <= 0 switch the respective factor
 * off (the reference's `if alpha:` / `if gamma:`).  loss_elem (B) and/or loss_mean (1) may be NULL; the mean is summed in
 * double in a fixed order (ws: recnow_focal_loss_workspace_bytes).
 * Backward: dlogits[i] = dloss_i/dlogit_i * (gelem ? gelem[i] : 1) * (gscalar ? *gscalar : 1) * scale  (scale = 1/B for the
 * mean); stop_weight_gradient as :60-61. */
size_t recnow_focal_loss_workspace_bytes(int64_t B);
int recnow_focal_loss_fwd(const float* labels, const float* logits, int64_t B, float alpha, float gamma, float* loss_elem,
                          float* loss_mean, void* ws, size_t ws_bytes, void* stream);
int recnow_focal_loss_bwd(const float* labels, const float* logits, int64_t B, float alpha, float gamma,
                          int stop_weight_gradient, const float* gelem, const float* gscalar, float scale, float* dlogits,
                          void* stream);

/* Pooled embedding lookup by slot: rec_now/rec_block/embedding_util.py:239-324 (and :138-195 for the slot -> target map).
 *   recnow_slot_targets:   seg[i] = index of slots[i] in targets[0..T) or -1;  key[i] (optional) = ids[i] if seg[i] >= 0 else
 *                          INT64_MIN (sorts after every id >= 0 as an unsigned radix key in recnow_group_segments, and costs
 *                          one extra 8-bit pass where all-ones cost five).  slot_dtype: RECNOW_KEY_I32/I64
 *                          (slots and targets share it; targets is a DEVICE array).  N = B*C entries.
 *   recnow_embed_pool_fwd: out[b][t][:] = sum_{c: seg[b][c]==t} weights[b][c] * table[rows[b][c]][:]  (mean != 0: divided by the
 *                          number of pooled entries, empty segments 0 - tf.math.unsorted_segment_mean).  rows: row index into
 *                          `table` per entry (the ids themselves, or the inverse index of recnow_embed_unique); weights, cnt
 *                          (B,T entry counts, needed by the 'mean' backward) may be NULL.  No atomics: fixed summation order.
 *                          V = rows of `table`: an entry whose row index is outside [0, V) adds a zero row (it still counts
 *                          for 'mean'), the same entries recnow_embed_scatter_rows drops in the backward.
 *   recnow_embed_pool_bwd_weights: dweights[b][c] = <dout[b][seg[b][c]], table[rows[b][c]]> (/ cnt for 'mean'), 0 for entries
 *                          that are not pooled or whose row is outside the table: the gradient TF autodiff gives `weights`.
 *   recnow_embed_unique:   from the sort of `key` (recnow_group_keys(I64) + recnow_group_segments): unique[s] = id of sorted
 *                          segment s, inverse[entry] = s, *n_unique = number of real ids (the INT64_MIN segment excluded).
 *   recnow_embed_rows_bwd: drows[s][:] = sum over the entries of segment s of w * dout[b][t][:] (/cnt): the gradient of row
 *                          unique[s]; row_ids[s] = that id (INT64_MIN for the unpooled segment and for unused slots s >= n_seg).
 *                          drows / row_ids hold N slots.  The entry range is cut into fixed chunks (a hot id may own millions of
 *                          entries); pieces are joined in ascending chunk order (ws: recnow_embed_rows_bwd_workspace_bytes).
 *   recnow_embed_scatter_rows: dtable[row_ids[s]][:] = drows[s][:] into a zero-initialised dense (V,D) gradient. */
/* id_limit / key32 (ABI 4): id_limit = V > 0 says the ids index a table of V rows (V < 2^31 - 1): entries that are not pooled AND ids outside
 * [0, V) then carry the key V instead of INT64_MIN -- it sorts last and is dropped like every id outside the table -- and key32 (optional,
 * N int32) receives the same keys as 32-bit values: the sort of the backward pass runs on ONE key word (three 8-bit digit passes for
 * V = 2^20 instead of four on two words).  id_limit = 0, key32 = NULL: the INT64_MIN form (ids of unknown range: the unique path). */
int recnow_slot_targets(const void* slots, int slot_dtype, const void* targets, int T, const int64_t* ids, int64_t N,
                        int32_t* seg, int64_t* key, int64_t id_limit, int32_t* key32, void* stream);
int recnow_embed_pool_fwd(const float* table, int D, int64_t V, const int64_t* rows, const int32_t* seg, const float* weights,
                          int64_t B, int C, int T, int mean, float* out, float* cnt, void* stream);
int recnow_embed_pool_bwd_weights(const float* table, int D, int64_t V, const int64_t* rows, const int32_t* seg, const float* cnt,
                                  const float* dout, int64_t B, int C, int T, int mean, float* dweights, void* stream);
int recnow_embed_unique(const int64_t* key, const int32_t* order, const int32_t* seg_id, const int32_t* seg_first,
                        const int32_t* n_seg, int64_t N, int64_t* unique, int64_t* inverse, int32_t* n_unique, void* stream);
size_t recnow_embed_rows_bwd_workspace_bytes(int64_t N, int D);
int recnow_embed_rows_bwd(const int64_t* key, const int32_t* order, const int32_t* seg_id, const int32_t* seg_first,
                          const int32_t* n_seg, const int32_t* seg, const float* weights, const float* cnt, const float* dout,
                          int64_t N, int C, int T, int D, int mean, float* drows, int64_t* row_ids, void* ws, size_t ws_bytes,
                          void* stream);
/* n_seg (ABI 4, optional DEVICE pointer: the segment count of recnow_group_segments): only the first n_seg[0] slots are read. */
int recnow_embed_scatter_rows(const float* drows, const int64_t* row_ids, int64_t n_slots, int D, int64_t V, float* dtable,
                              const int32_t* n_seg, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Measurement hook (bench.py): per-launch HIP-event timing of the GEMM kernels on the launch stream.
 * recnow_prof_enable(capacity > 0) arms `capacity` launch slots, (0) disables.  recnow_prof_collect synchronises and
 * returns per-kernel-family totals in HOST arrays of 16 entries indexed by tag: 1 = k_gemm<128,128>, 2 = k_gemm<128,160>,
 * 3 = k_gemm<256,64>, 4 = k_gemm<256,32>, 5 = k_gemm_shortk, 6 = k_mix_mid_fwd, 7 = k_mix_mid_bwd, 8 = k_gemm_split, 9 = the fused GEMM1,
 * 10 / 11 = k_mix_tile_fwd / _bwd, 12 = k_gemm<64,128> (csrc/prof.hpp): launches, total
 * milliseconds, total algorithmic flops (2*M*N*K*batch) and (bytes_host, may be NULL) total algorithmic HBM bytes: every
 * operand read once, every output written once, read-modify-write outputs twice.
 * ---------------------------------------------------------------------------------------------------------- */
int recnow_prof_enable(int capacity);
/* Time only every n-th GEMM launch (default 1 = all): the two timing events around a launch cost ~2 us of stream
 * serialisation each; with n coprime to the launches per step every launch position is sampled equally often. */
int recnow_prof_sample_every(int n);
int recnow_prof_collect(int* count_host, double* ms_host, double* flops_host, double* bytes_host);
/* The recorded launches one by one instead of per-tag totals: tag_host / t0_ms_host / t1_ms_host (HOST arrays of `capacity` entries) get
 * the tag and the interval of every record, in milliseconds after the first record's start; returns the number written or a negative
 * code; synchronises and rewinds like recnow_prof_collect.  After recnow_prof_sample_every(1) the records are EVERY hooked launch plus
 * the phase tags 13 (grouping), 14 (loss stage), 15 (packs, layer-end reductions): the timeline from which a caller tells overlapping
 * launches of two streams apart. */
int recnow_prof_intervals(int* tag_host, double* t0_ms_host, double* t1_ms_host, int capacity);
/* (ABI 6) Entries of the per-tag arrays recnow_prof_collect fills -- size the host arrays from this -- and the number of records that found the pool full
 * since the last recnow_prof_enable / the last call (non-zero: the totals and intervals under-report; re-arm with a larger capacity). */
int recnow_prof_tag_count(void);
int recnow_prof_dropped(void);

/* HIP events owned through the C ABI (timing disabled): the layer_events_host of recnow_dcn_mix_score_bwd.  A host framework
 * whose event type is created lazily (torch.cuda.Event) cannot hand a handle over before the first record. */
int recnow_event_create(void** event_out);
int recnow_event_destroy(void* event);
int recnow_event_record(void* event, void* stream);
int recnow_stream_wait_event(void* stream, void* event);
/* x[0..n) *= 1 / (count[0] + eps), count a DEVICE scalar: the in-place normalisation of a reduced gradient bucket by the global
 * pair count (pairwise_loss_from_batch.py:125-126, P + 1e-10) without a host round trip.  x 16-byte aligned.
 * stats_out (optional, 2 floats): receives (loss_sum[0] / (count[0] + eps), count[0]) = the mean loss the reference returns
 * (pairwise_loss_from_batch.py:279) and the pair count, from the same launch. */
int recnow_scale_by_inv_count(float* x, int64_t n, const float* count, float eps, const float* loss_sum, float* stats_out,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECNOW_H_ */
