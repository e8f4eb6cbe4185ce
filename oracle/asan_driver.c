/* TEST INFRASTRUCTURE ONLY (oracle/): sanitizer driver of the plain-C oracle.  `make -C oracle asan` builds it together with
 * pairs_oracle.c under -fsanitize=address,undefined; tests/test_oracle_golden.py runs it on the CPU box.  It walks the entry points over
 * the reference's literal pairwise case (/root/reference/tests/rec_block/test_pairwise_loss_from_batch.py:33-74: goldens 0.5415076 and
 * 1.3132617), the empty batch, one row, NaN / infinite / signed-zero group ids, a mask, and seeded random batches on which the
 * segment-based port must agree with the quadratic restatement.  Any out-of-bounds access, leak or undefined operation aborts. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t oracle_pair_indices(const float*, const float*, const float*, const uint8_t*, int64_t, int, int32_t*, int32_t*, int64_t);
int64_t oracle_pairwise_bpr(const float*, const float*, const float*, const uint8_t*, int64_t, int, double, double, double*, double*);
int64_t oracle_pairwise_bpr_grouped(const float*, const float*, const float*, const uint8_t*, int64_t, int, double, double, double*, double*);

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static void random_case(int64_t B, int n_groups, int flags, double factor, double power, int with_mask) {
    float* g = (float*)malloc((size_t)(B + 1) * sizeof(float));
    float* y = (float*)malloc((size_t)(B + 1) * sizeof(float));
    float* s = (float*)malloc((size_t)(B + 1) * sizeof(float));
    uint8_t* m = (uint8_t*)malloc((size_t)(B + 1));
    double* d1 = (double*)malloc((size_t)(B + 1) * sizeof(double));
    double* d2 = (double*)malloc((size_t)(B + 1) * sizeof(double));
    for (int64_t i = 0; i < B; ++i) {
        g[i] = (float)(rnd() % (uint32_t)n_groups);
        y[i] = (float)(rnd() % 3);
        s[i] = (float)((int)(rnd() % 2001) - 1000) / 250.0f;
        m[i] = (uint8_t)(rnd() % 4 != 0);
    }
    double l1 = 0, l2 = 0;
    const int64_t p1 = oracle_pairwise_bpr(g, y, s, with_mask ? m : NULL, B, flags, factor, power, &l1, d1);
    const int64_t p2 = oracle_pairwise_bpr_grouped(g, y, s, with_mask ? m : NULL, B, flags, factor, 0.0, &l2, d2);
    CHECK(p1 == p2);
    if (power == 0.0) {
        CHECK(fabs(l1 - l2) <= 1e-12 * (1.0 + fabs(l1)));
        for (int64_t i = 0; i < B; ++i) CHECK(fabs(d1[i] - d2[i]) <= 1e-12);
    }
    int32_t* pos = (int32_t*)malloc((size_t)(p1 + 1) * sizeof(int32_t));
    int32_t* neg = (int32_t*)malloc((size_t)(p1 + 1) * sizeof(int32_t));
    CHECK(oracle_pair_indices(g, y, s, with_mask ? m : NULL, B, flags, pos, neg, p1) == p1);
    for (int64_t k = 1; k < p1; ++k) CHECK(pos[k] > pos[k - 1] || (pos[k] == pos[k - 1] && neg[k] > neg[k - 1]));      /* row-major order */
    CHECK(oracle_pair_indices(g, y, s, with_mask ? m : NULL, B, flags, pos, neg, p1 / 2) == p1);                          /* capped write */
    free(g); free(y); free(s); free(m); free(d1); free(d2); free(pos); free(neg);
}

int main(void) {
    /* the reference's literal case */
    const float g[5] = {1, 1, 2, 2, 2}, s[5] = {0, 1, 2, 3, 4}, y[5] = {1.1f, 0, 0, 1, 1};
    const uint8_t m[5] = {1, 1, 0, 0, 0};
    double loss = 0, d[5];
    CHECK(oracle_pairwise_bpr(g, y, s, NULL, 5, 1, 1.0, -0.5, &loss, d) == 3 && fabs(loss - 0.5415076) < 1e-4);
    oracle_pairwise_bpr(g, y, s, m, 5, 1, 1.0, -0.5, &loss, d);
    CHECK(fabs(loss - 1.3132617) < 1e-4);
    /* empty batch and one row: zero pairs, loss 0 (0 / 1e-10) */
    double d1[1] = {7.0};
    CHECK(oracle_pairwise_bpr(g, y, s, NULL, 0, 1, 1.0, 0.0, &loss, d1) == 0 && loss == 0.0);
    CHECK(oracle_pairwise_bpr_grouped(g, y, s, NULL, 0, 1, 1.0, 0.0, &loss, d1) == 0 && loss == 0.0);
    CHECK(oracle_pairwise_bpr_grouped(g, y, s, NULL, 1, 1, 1.0, 0.0, &loss, d1) == 0 && d1[0] == 0.0);
    CHECK(oracle_pair_indices(g, y, s, NULL, 0, 1, NULL, NULL, 0) == 0);
    /* NaN / inf ids pair with nobody, -0.0 == +0.0 */
    const float gq[6] = {NAN, NAN, INFINITY, INFINITY, -0.0f, 0.0f}, yq[6] = {1, 0, 1, 0, 1, 0}, sq[6] = {0, 1, 0, 1, 0, 1};
    double dq[6];
    CHECK(oracle_pairwise_bpr(gq, yq, sq, NULL, 6, 1, 1.0, 0.0, &loss, dq) == 1);
    CHECK(oracle_pairwise_bpr_grouped(gq, yq, sq, NULL, 6, 1, 1.0, 0.0, &loss, dq) == 1 && dq[0] == 0.0 && dq[2] == 0.0 && dq[4] < 0.0 && dq[5] > 0.0);
    /* seeded random batches: quadratic restatement == segment port, pair order, capped index writes */
    random_case(257, 9, 1, 1.0, 0.0, 0);
    random_case(300, 1, 1, 2.5, 0.0, 1);
    random_case(64, 64, 3, 1.0, 0.0, 1);
    random_case(513, 31, 1, 1.0, -0.5, 0);
    random_case(129, 5, 3, 0.5, 2.0, 1);
    if (fails) { printf("asan driver: %d check(s) failed\n", fails); return 1; }
    printf("asan driver ok\n");
    return 0;
}
