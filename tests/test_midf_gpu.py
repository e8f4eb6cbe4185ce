"""The sub-space forward fused into the (transposed) GEMM1 epilogue of DCNMixLayer (`recnow_gemm_desc.mid_V`, csrc/gemm_kernel.hpp
`gemm_midf_epilogue`; reference /root/reference/rec_now/layers/dcn_mix_layer.py:135-138,146-147) is only dispatched from 65 536 rows on
(512 batch tiles), where tests/test_northstar_gpu.py holds it to the fp64 oracle.  Here the same kernel is forced on for every batch
(`RECNOW_MIDF=2`, read once per process: hence a subprocess) and the small-shape oracle sweeps of the layer, of the fused node and of the
step route run through it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_subspace_forward_on_small_batches(dev):
    env = dict(os.environ, RECNOW_MIDF='2')
    out = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu',
                          os.path.join(ROOT, 'tests', 'test_fused_gpu.py'), os.path.join(ROOT, 'tests', 'test_step_gpu.py'),
                          '-k', 'vs_oracle_and_unfused or equals_autograd_route'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout
