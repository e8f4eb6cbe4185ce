/* TEST INFRASTRUCTURE ONLY (oracle/): never linked or called by the product library.
 *
 * Plain-C restatement of the reference's in-batch pair enumeration and BPR pairwise loss in the reference's own
 * O(B^2) formulation, without the (B,B) temporaries, so it also runs at B = 65536 where the dense torch/numpy oracle
 * cannot (/root/reference/rec_now/rec_block/pairwise_loss_from_batch.py):
 *   same group      : g_i - g_j == 0.0 in float, i != j                         (:31-37)
 *   sample mask     : mask_i && mask_j                                          (:154-172)
 *   label condition : label_i > label_j                                         (:189)
 *   wrong order     : score_i < score_j (optional)                              (:197-203)
 *   pair order      : row-major over (i, j) = tf.boolean_mask of the flattened mask   (:217, :272-273)
 *   occurrence      : weight = (#surviving pairs whose positive row has the same groups[0] id) ** power   (:130-151,:282-291)
 *   loss            : sum_p w_p * softplus(-factor*(s_i - s_j)) / (float(P) + 1e-10)  (:117-126)
 * Pinned by tests/test_oracle_golden.py against the reference's literal goldens (0.5415076 / 1.3132617) and against
 * oracle/dense_ref.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int valid_pair(const float* g, const float* label, const float* score, const uint8_t* mask, int64_t i, int64_t j, int flags) {
    if (i == j) return 0;
    if (!((g[i] - g[j]) == 0.0f)) return 0;               /* NaN / inf differences compare false */
    if (mask && !(mask[i] && mask[j])) return 0;
    if ((flags & 1) && !(label[i] > label[j])) return 0;
    if ((flags & 2) && !(score[i] < score[j])) return 0;
    return 1;
}

/* Writes up to `cap` pairs in the reference's order; returns the total number of pairs. */
int64_t oracle_pair_indices(const float* g, const float* label, const float* score, const uint8_t* mask, int64_t B, int flags,
                            int32_t* pos, int32_t* neg, int64_t cap) {
    int64_t n = 0;
    for (int64_t i = 0; i < B; ++i)
        for (int64_t j = 0; j < B; ++j)
            if (valid_pair(g, label, score, mask, i, j, flags)) {
                if (n < cap) { pos[n] = (int32_t)i; neg[n] = (int32_t)j; }
                ++n;
            }
    return n;
}

/* loss and d loss / d score (double accumulation); returns the pair count. */
int64_t oracle_pairwise_bpr(const float* g, const float* label, const float* score, const uint8_t* mask, int64_t B, int flags,
                            double factor, double power, double* loss_out, double* dscore) {
    int64_t P = 0;
    /* pass 1: pairs per positive row, then per groups[0] value (here: the single group tensor) */
    int64_t* crow = (int64_t*)calloc((size_t)(B > 0 ? B : 1), sizeof(int64_t));
    for (int64_t i = 0; i < B; ++i) {
        int64_t c = 0;
        for (int64_t j = 0; j < B; ++j) c += valid_pair(g, label, score, mask, i, j, flags);
        crow[i] = c;
        P += c;
    }
    double* w = (double*)malloc((size_t)(B > 0 ? B : 1) * sizeof(double));
    for (int64_t i = 0; i < B; ++i) {
        w[i] = 1.0;
        if (power != 0.0 && crow[i] > 0) {
            int64_t cnt = 0;                                   /* pairs whose positive row shares i's group id */
            for (int64_t k = 0; k < B; ++k)
                if ((g[i] - g[k]) == 0.0f) cnt += crow[k];
            w[i] = pow((double)cnt, power);
        }
    }
    const double denom = (double)(float)P + 1.0e-10;
    double sum = 0.0;
    for (int64_t i = 0; i < B; ++i) dscore[i] = 0.0;
    for (int64_t i = 0; i < B; ++i) {
        if (!crow[i]) continue;
        for (int64_t j = 0; j < B; ++j) {
            if (!valid_pair(g, label, score, mask, i, j, flags)) continue;
            const double x = factor * ((double)score[i] - (double)score[j]);
            const double sp = (x > 0 ? 0.0 : -x) + log1p(exp(-fabs(x)));      /* softplus(-x) */
            const double sg = 1.0 / (1.0 + exp(x));                           /* sigma(-x)    */
            sum += w[i] * sp;
            dscore[i] -= w[i] * factor * sg / denom;
            dscore[j] += w[i] * factor * sg / denom;
        }
    }
    *loss_out = sum / denom;
    free(crow);
    free(w);
    return P;
}

/* ---- segment-based CPU port (bench.py's cpu_baseline at BASELINE.json's batch size) ---------------------------------------
 * The reference's (B,B) formulation (:30-37,:90-93) cannot run at B = 65536 (>= 100 GB of temporaries), and the quadratic
 * loops above take minutes there.  This is the same pair set and loss computed the way a CPU implementation would: sort the
 * rows by group id (stable: members stay in ascending row order), then only the n_g^2 candidates inside each group are
 * visited, groups in parallel (OpenMP).  NaN / infinite ids pair with nobody ((g_i - g_j) == 0 is false for them), -0.0 and
 * +0.0 are the same group.  Checked against oracle_pairwise_bpr in tests/test_oracle_golden.py. */
typedef struct { float key; int32_t row; } KeyRow;
static int cmp_keyrow(const void* a, const void* b) {
    const KeyRow* x = (const KeyRow*)a;
    const KeyRow* y = (const KeyRow*)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->row > y->row) - (x->row < y->row);
}

int64_t oracle_pairwise_bpr_grouped(const float* g, const float* label, const float* score, const uint8_t* mask, int64_t B, int flags,
                                    double factor, double power, double* loss_out, double* dscore) {
    KeyRow* kr = (KeyRow*)malloc((size_t)(B > 0 ? B : 1) * sizeof(KeyRow));
    int64_t n = 0;
    for (int64_t i = 0; i < B; ++i) {
        dscore[i] = 0.0;
        if (!((g[i] - g[i]) == 0.0f)) continue;                 /* NaN / inf: no partner */
        kr[n].key = g[i] == 0.0f ? 0.0f : g[i];                 /* -0.0 -> +0.0 */
        kr[n].row = (int32_t)i;
        ++n;
    }
    qsort(kr, (size_t)n, sizeof(KeyRow), cmp_keyrow);
    int64_t* first = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t ng = 0;
    for (int64_t k = 0; k < n; ++k)
        if (k == 0 || kr[k].key != kr[k - 1].key) first[ng++] = k;
    first[ng] = n;
    int64_t* cnt_g = (int64_t*)calloc((size_t)(ng > 0 ? ng : 1), sizeof(int64_t));
    int64_t P = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : P)
    for (int64_t s = 0; s < ng; ++s) {
        int64_t c = 0;
        for (int64_t a = first[s]; a < first[s + 1]; ++a)
            for (int64_t b = first[s]; b < first[s + 1]; ++b) c += valid_pair(g, label, score, mask, kr[a].row, kr[b].row, flags);
        cnt_g[s] = c;
        P += c;
    }
    const double denom = (double)(float)P + 1.0e-10;
    double sum = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : sum)
    for (int64_t s = 0; s < ng; ++s) {
        if (!cnt_g[s]) continue;
        const double w = power != 0.0 ? pow((double)cnt_g[s], power) : 1.0;
        for (int64_t a = first[s]; a < first[s + 1]; ++a) {
            const int64_t i = kr[a].row;
            for (int64_t b = first[s]; b < first[s + 1]; ++b) {
                const int64_t j = kr[b].row;
                if (!valid_pair(g, label, score, mask, i, j, flags)) continue;
                const double x = factor * ((double)score[i] - (double)score[j]);
                const double sp = (x > 0 ? 0.0 : -x) + log1p(exp(-fabs(x)));
                const double sg = 1.0 / (1.0 + exp(x));
                sum += w * sp;
                dscore[i] -= w * factor * sg / denom;             /* rows of a group belong to one thread: no race */
                dscore[j] += w * factor * sg / denom;
            }
        }
    }
    *loss_out = sum / denom;
    free(kr);
    free(first);
    free(cnt_g);
    return P;
}
