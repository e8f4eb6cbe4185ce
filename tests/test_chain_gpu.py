"""The chained products of DCNMixLayer (csrc/dcnmix_chain.hip: out = x0 * (T2g [W; b]) of layer l and T1 of layer l + 1 in one kernel,
the first product's accumulator registers as the second product's A fragments; reference
/root/reference/rec_now/layers/dcn_mix_layer.py:141-150 followed by :135-136) are opt-in (slower than the two launches they replace on
this chip: DESIGN.md 5g).  Here they are forced on for every batch (`RECNOW_CHAIN=2`, read once per process: hence a subprocess) and
the oracle sweeps of the fused node and of the step route run through them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chained_products_on_small_batches(dev):
    env = dict(os.environ, RECNOW_CHAIN='2')
    out = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu',
                          os.path.join(ROOT, 'tests', 'test_fused_gpu.py'), os.path.join(ROOT, 'tests', 'test_step_gpu.py'),
                          '-k', 'vs_oracle_and_unfused or equals_autograd_route'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout
