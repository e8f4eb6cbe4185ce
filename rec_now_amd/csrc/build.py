"""Builds librecnow_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'librecnow_hip.so')
OBJ_DIR = os.path.join(HERE, 'build')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
if os.environ.get('RECNOW_TRACE') == '1':        # diagnostic build: per-workgroup timestamps in the GEMM kernels (tools/gemm_trace.py)
    FLAGS.append('-DRN_GEMM_TRACE')


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(verbose=True, force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HERE, '*.hip')))
    deps = sorted(glob.glob(os.path.join(HERE, '*.hpp'))) + [os.path.join(PKG, '..', 'include', 'recnow.h')]
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _newer(s, o, deps):
            jobs.append(['hipcc'] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT):
        run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
