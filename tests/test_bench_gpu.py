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


def _bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline'] + list(flags),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_line_fields_and_graph_replay(dev):
    eager = _bench()
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline'):
        assert key in eager, key
    assert eager['n_gpus'] == 1 and eager['steps'] == 2 and eager['vs_baseline'] is None and eager['dtype'] == 'f32'
    r = eager['roofline']
    assert r['bound'] == 'mfma' and 0.3 < r['frac'] < 1.0 and r['unit'] == 'TFLOP/s'
    assert eager['parity']['ok'] and eager['parity_max_rel'] <= 1e-5          # in-run parity vs the fp64 subset oracle (north_star tolerance)
    graph = _bench('--graph')
    assert graph['config']['hip_graph'] is True
    assert graph['config']['loss'] == eager['config']['loss']                 # same kernels, same order: bit-identical loss
