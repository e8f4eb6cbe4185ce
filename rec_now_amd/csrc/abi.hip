#include "common.hpp"
// Bumped whenever an exported signature or recnow_gemm_desc changes incompatibly; rec_now_amd/_lib.py refuses a library whose
// version differs from the one its SIGNATURES table was written for.  2: recnow_prof_collect (4 arrays), recnow_embed_pool_fwd
// (V), recnow_gemm_desc (second outputs, side products); 3 (round 3): recnow_dcn_mix_step + its descriptor, recnow_pairwise_loss,
// recnow_listwise_loss, the packed weights kept in recnow_dcn_mix_saved_bytes; 4 (round 5): recnow_dcn_mix_step_desc.B_pad (ragged per-rank
// batches on the fast route), recnow_dcn_mix_tile_route; 5 (round 5, second session): recnow_set_gemm_staging / recnow_get_gemm_staging (new symbols: a
// version-4 build would fail to bind them); 6 (round 6): recnow_prof_tag_count / recnow_prof_dropped, the split-precision piece planes in
// recnow_dcn_mix_saved_bytes.
extern "C" int recnow_abi_version(void) { return 6; }
