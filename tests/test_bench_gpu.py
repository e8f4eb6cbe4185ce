"""bench.py end to end on the GPU (as the driver runs it, with few steps): the default line carries every field of the contract, and the
step captured into a HIP graph (`--graph`, SURVEY 8f.1 diagnostic) replays to the same loss as the eager step -- a guard for the capture
itself (hipStreamEndCapture crashed when the warm-up had run on the legacy default stream)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    """The environment of a bench.py child WITHOUT the launcher variables another test of this pytest process may have exported (the
    one-rank RCCL fixture of test_step_gpu.py sets WORLD_SIZE / RANK / MASTER_*): with them bench.py would believe a launcher started it."""
    return {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_WORLD_SIZE', 'GROUP_RANK')}


def _bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline'] + list(flags),
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_line_fields_and_graph_replay(dev):
    eager = _bench()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline'):
        assert key in eager, key
    assert eager['n_gpus'] == 1 and eager['steps'] == 2 and eager['vs_baseline'] is None
    # round 6: the default arithmetic at 65 536 rows is the six-term 3 x bf16 split (same 1e-5 gate); the exact-fp32 step is timed in the same run and
    # is the LAST key of the line
    assert eager['config']['gemm_precision'] == 'bf16x3' and eager['dtype'].startswith('f32 via 3 x bf16 six-term split')
    assert list(eager)[-1] == 'exact_f32' and eager['exact_f32']['gemm_precision'] == 'f32' and eager['exact_f32']['ms_per_step'] > 0
    r = eager['roofline']
    assert r['bound'] == 'mfma' and 0.2 < r['frac'] < 1.0 and r['unit'] == 'TFLOP/s'
    assert abs(r['peak'] - 16 * 157.3 / 6) < 1e-6 and r['kernel'] == 'k_gemm_split'      # the split products' ceiling: dense bf16 peak / 6 terms
    exact = _bench('--gemm-precision', 'f32')
    assert exact['dtype'] == 'f32' and list(exact)[-1] == 'split_precision' and exact['roofline']['peak'] == 157.3
    assert exact['parity']['ok'] and exact['parity_max_rel'] <= 1e-5
    assert eager['parity']['ok'] and eager['parity_max_rel'] <= 1e-5          # in-run parity vs the fp64 subset oracle (north_star tolerance)
    graph = _bench('--graph')
    assert graph['config']['hip_graph'] is True
    assert graph['config']['loss'] == eager['config']['loss']                 # same kernels, same order: bit-identical loss


def test_self_launch_one_rank_force_dist_is_self_verifying(dev):
    """`bench.py --gpus N` starts its own ranks (child processes through torch.distributed.run, the parent never touches the GPU).  Driven here
    with one rank over a real RCCL group: the line carries `rccl_ranks`, and the cross-rank gate -- the reduced loss and pair count against
    oracle/pairs_oracle.c on the all-gathered batch, every all-reduced weight gradient against the sum of the per-rank fp64 oracles -- holds."""
    # 8177 rows: a RAGGED per-rank batch (whole groups per rank are never multiples of 256) on padded storage -- the same kernels as the 8192-row shard
    # (profiles/r05_bench_rows8177.json against rows8192.json: 0.642 vs 0.644 ms on one box), self-verifying like every N > 1 line
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--launch', '--rows', '8177',
                          '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['rccl_ranks'] == 1 and line['n_gpus'] == 1
    par = line['parity']
    assert par['ok'] and par['parity_max_rel'] <= 1e-5
    assert par['gathered_batch_pairs_oracle']['pairs_equal'] and par['gathered_rows'] == 8177 and line['config']['rows_per_rank'] == [8177]
    assert 'ragged batch on padded storage (8177 -> 8192 rows)' in line['config']['route']
    full = par['oracle_fp64_all_ranks']
    assert full['pairs_equal'] and len([k for k in full if k.startswith('cross.')]) == 15 and 'head.kernel' in full
    # the N > 1 line describes itself: the dominant kernel by EXCLUSIVE time is a row-block kernel at this shard size, the route says where the
    # grouping ran, and the price of the collectives is measured
    roof = line['roofline']
    # (8192 rows: the six K = B weight-gradient products on 64-row tiles, or one of the two row-block launches -- never the 128 x 128 family, which
    # does not run at this size; round 4's line named it)
    assert roof['kernel'].startswith('k_mix_tile_') or roof['kernel'].startswith('k_gemm<64,128'), roof['kernel']
    assert any(k.startswith('k_mix_tile_fwd') for k in roof['exclusive_ms_per_step']) and any(k.startswith('k_mix_tile_bwd') for k in roof['exclusive_ms_per_step'])
    assert 'row-block persistent kernels' in line['config']['route'] and 'on the main stream in front of the forward pass' in line['config']['route']
    excl = roof['exclusive_ms_per_step']
    assert any(k.startswith('grouping') for k in excl) and any(k.startswith('loss stage') for k in excl)
    assert 0.5 * line['ms_per_step'] < roof['exclusive_covered_ms_per_step'] < 1.5 * line['ms_per_step']
    assert abs(sum(excl.values()) - roof['exclusive_covered_ms_per_step']) < 1e-6 * max(1.0, roof['exclusive_covered_ms_per_step'])
    assert 'comm_exposed_ms' in line and line['comm']['with_collectives_ms'] > 0 and line['comm']['without_collectives_ms'] > 0
    assert line['ms_per_step_per_rank']['min'] <= line['ms_per_step_per_rank']['max'] and len(line['ms_per_step_per_rank']['ranks']) == 1


def test_more_ranks_than_gpus_fails_cleanly(dev):
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n)], capture_output=True, text=True, timeout=120, cwd=ROOT, env=_clean_env())
    assert out.returncode != 0
    assert 'only %d GPU(s) visible' % (n - 1) in out.stderr


def test_two_ranks_share_one_gpu_over_gloo_and_verify_across_ranks(dev):
    """The N > 1 step with REAL multi-rank semantics on a one-GPU box: two ranks (two processes on GPU 0, `--backend gloo --oversubscribe`) own the two
    shards of ONE global batch of 8192 rows split by `dp.shard_rows_by_group` (`--shard hash`: whole groups per rank, ragged shard sizes), the step route writes its gradients into the layer-wise
    reducer's buckets, every bucket is all-reduced across the two processes, and the cross-rank gate holds the reduced loss / pair count to the C pair
    oracle on the ALL-GATHERED batch and all 17 all-reduced weight gradients to the sum of the two ranks' fp64 oracles."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--oversubscribe', '--rows', '4096',
                          '--shard', 'hash', '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['backend'] == 'gloo' and line['shard'] == 'hash'
    # --shard hash: ONE global batch of 8192 rows split by dp.shard_rows_by_group -- whole groups per rank, RAGGED shards on padded storage
    rr = line['config']['rows_per_rank']
    assert sum(rr) == 8192 and len(rr) == 2 and all(r % 256 for r in rr), rr
    assert 'ragged batch on padded storage' in line['config']['route']
    assert 'comm_exposed_ms' in line and len(line['ms_per_step_per_rank']['ranks']) == 2
    par = line['parity']
    assert par['gathered_rows'] == 8192 and par['gathered_batch_pairs_oracle']['pairs_equal']
    full = par['oracle_fp64_all_ranks']
    assert full['pairs_equal'] and len([k for k in full if k.startswith('cross.')]) == 15
    assert par['ok'] and par['parity_max_rel'] <= 1e-5, par


@pytest.mark.parametrize('config', ['c4', 'c5'])
def test_model_configs_two_ranks_over_gloo_verify_across_ranks(dev, config):
    """`bench.py --config c4 | c5` (VERDICT round 5, row e''): the data-parallel MODEL steps of BASELINE.json's configs[3] / configs[4] -- CIN || FM -> head ->
    pairwise; PLE -> 3 heads -> listwise -- with two real ranks (two processes on GPU 0 over gloo), every rank its own whole groups / lists, the gradients
    all-reduced bucket by bucket from autograd's post-accumulate hooks (dp.OverlappedGradientReducer).  The line's gate: the reduced loss against the
    oracle's loss stage on the gathered batch, outputs and d loss / d x per rank and every all-reduced weight gradient against the sum of the ranks' fp64
    oracles.  /root/reference/rec_now/layers/cin_layer.py:72-122, ple_layer.py:295-321, rec_block/listwise_loss_from_batch.py:89-173."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--gpus', '2', '--backend', 'gloo', '--oversubscribe',
                          '--rows', '2048', '--steps', '2', '--warmup', '1'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=_clean_env())
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline', 'parity', 'comm_exposed_ms'):
        assert key in line, key
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['config']['rows_per_rank'] == [2048, 2048] and line['scaling'] == 'weak'
    assert ('configs[3]' if config == 'c4' else 'configs[4]') in line['config']['workload']
    par = line['parity']
    assert par['ok'] and par['parity_max_rel'] <= 1e-5, par
    full = par['oracle_fp64_all_ranks']
    if config == 'c4':
        assert par['gathered_batch_pairs_oracle']['pairs_gpu'] == par['gathered_batch_pairs_oracle']['pairs_oracle'] > 0
        assert {'cin.1', 'cin.2', 'cin.3', 'head.kernel', 'head.bias', 'outputs', 'dx', 'loss'} <= set(full)
    else:
        assert par['listwise_stage_oracle']['lists_gpu'] == par['listwise_stage_oracle']['lists_oracle'] > 0
        assert len([k for k in full if k.startswith('ple.')]) >= 4 * 2 * 2 * 2 + 4 and 'head.kernel' in full
    assert line['roofline']['bound'] == 'mfma' and len(line['comm']['buckets']) >= 1
