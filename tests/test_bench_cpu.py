"""bench.py's self-launcher on a host without a GPU: `--gpus N` must fail cleanly, naming the visible device count, before anything is started."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_enough_devices_names_the_visible_count():
    import torch
    n_vis = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n_vis + 2)], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert 'only %d GPU(s) visible' % n_vis in out.stderr and 'nothing launched' in out.stderr
