// TEST INFRASTRUCTURE: the host-only dispatch logic of the MFMA GEMM (rec_now_amd/csrc/gemm_dispatch.hpp: tile family, 64-row small-M rule, K split, slab
// workspace) swept over shapes on the CPU under -fsanitize=address,undefined (sanitizer build; `make -C tools/san`, run by
// tests/test_oracle_golden.py).  Checked for every shape: the K slices tile [0, K) exactly (k-chunks are multiples of 32, the last slice is not empty),
// the grid's z extent fits a launch, the slab size never overflows size_t arithmetic for the sizes the layers use, the 64-row choice is only made
// for row counts it divides, and the split of the hot-path products is what DESIGN.md says it is.
#include <initializer_list>
#include <stdio.h>
#include <string.h>
#include "../../rec_now_amd/csrc/gemm_dispatch.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s  (M %d N %d K %d batch %d sp_r %d)\n", __FILE__, __LINE__, #c, d.M, d.N, d.K, d.batch, d.sp_r); ++fails; } } while (0)

static void one(int M, int N, int K, int batch, int sp_r, int c2_mode, const RnDispatchEnv& env, int* out_bm = nullptr, int* out_s = nullptr, int slots = 512) {
    recnow_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.batch = batch; d.sp_r = sp_r; d.c2_mode = c2_mode;
    const GemmCfg c = pick_cfg(&d, env);
    CHECK(c.BM == 64 || c.BM == 128 || c.BM == 256);
    CHECK(c.BN == 32 || c.BN == 64 || c.BN == 128 || c.BN == 160);
    if (c.BM == 64) CHECK(M % 64 == 0 && N == 128 && sp_r >= 1 && sp_r <= 2 && env.precision == 0 && batch == 1);
    int s = 0, kc = 0;
    pick_split(&d, c, &s, &kc, slots);
    CHECK(s >= 1 && kc >= 32 && kc % 32 == 0);
    CHECK(rnd_slab_bytes(&d, s, 256) <= rnd_slab_bytes_any(&d, c, 256));      // what the workspace query reserves holds either slot target
    CHECK((long long)(s - 1) * kc < K);                    // the last slice starts inside K
    CHECK((long long)s * kc >= K);                         // the slices cover K
    if (c2_mode) CHECK(s == 1);                            // fused second outputs need the whole K in one workgroup
    CHECK((long long)batch * s <= 65535 || K > (1 << 24)); // gridDim.z
    const size_t slab = rnd_slab_bytes(&d, s, 256);
    CHECK((s > 1) == (slab > 0));
    if (s > 1) CHECK(slab >= (size_t)s * batch * M * N * sizeof(float) && slab % 256 == 0);
    if (out_bm) *out_bm = c.BM;
    if (out_s) *out_s = s;
}

int main(void) {
    const RnDispatchEnv envs[] = {{256, -1, 0}, {0, -1, 0}, {64, 0, 0}, {256, 1, 0}, {256, -1, 1}};
    const int Ms[] = {1, 7, 64, 100, 128, 192, 256, 1000, 1024, 4096, 8192, 16384, 32768, 65536, 262144, 2097152};
    const int Ns[] = {1, 8, 32, 33, 64, 65, 128, 130, 144, 160, 161, 256, 1024, 4096, 8192};
    const int Ks[] = {1, 31, 32, 33, 130, 144, 256, 512, 1000, 1024, 4096, 8192, 16384, 32768, 65536, 262144};
    for (const RnDispatchEnv& env : envs)
        for (int M : Ms)
            for (int N : Ns)
                for (int K : Ks)
                    for (int batch : {1, 3})
                        for (int sp_r : {0, 2, 4})
                            for (int c2 : {0, 1}) {
                                if (sp_r && batch != 1) continue;
                                one(M, N, K, batch, sp_r, c2, env);
                                one(M, N, K, batch, sp_r, c2, env, nullptr, nullptr, 256);      // products launched as concurrent pairs (dcnmix_bwd_tile)
                            }
    // the hot-path products, as DESIGN.md 5g / 5h state them (default switches)
    const RnDispatchEnv def = {256, -1, 0};
    int bm = 0, s = 0;
    recnow_gemm_desc d;
    memset(&d, 0, sizeof(d));
    one(65536, 128, 1024, 1, 2, 0, def, &bm, &s); d.M = 65536; d.N = 128; d.K = 1024; d.batch = 1; d.sp_r = 2; CHECK(bm == 128 && s == 1);      // GEMM1 at the metric's batch: 512 tiles, no split
    one(1024, 128, 65536, 1, 2, 0, def, &bm, &s); d.M = 1024; d.K = 65536; CHECK(bm == 128 && s == 64);                                        // dU / dW: 8 tiles x 64 slabs
    one(8192, 128, 1024, 1, 2, 0, def, &bm, &s); d.M = 8192; d.K = 1024; CHECK(bm == 64 && s == 4);                                            // the 8-GPU shard: 128 tiles x 4
    one(1024, 128, 8192, 1, 2, 0, def, &bm, &s); d.M = 1024; d.K = 8192; CHECK(bm == 64 && s == 32);
    one(1024, 128, 8192, 1, 2, 0, def, &bm, &s, 256); CHECK(bm == 64 && s == 16);                                                              // ... as one of a concurrent pair: 16 tiles x 16 slices
    one(16384, 128, 1024, 1, 2, 0, def, &bm, &s); d.M = 16384; d.K = 1024; CHECK(bm == 64 && s == 2);
    one(32768, 128, 1024, 1, 2, 0, def, &bm, &s); d.M = 32768; CHECK(bm == 64 && s == 1);
    one(32768, 4096, 512, 1, 0, 0, def, &bm, &s); d.M = 32768; d.N = 4096; d.K = 512; d.sp_r = 0; CHECK(bm == 128 && s == 1);                  // PLE expert layer
    one(512, 4096, 32768, 1, 0, 0, def, &bm, &s); d.M = 512; d.K = 32768; CHECK(bm == 128 && s == 4);                                          // its weight gradient: slabs of 8192 terms (accuracy)
    if (fails) { printf("dispatch sanitizer driver: %d check(s) failed\n", fails); return 1; }
    printf("dispatch sanitizer driver ok\n");
    return 0;
}
