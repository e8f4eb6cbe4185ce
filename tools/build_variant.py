#!/usr/bin/env python
"""Builds an A/B variant of the library beside the product build:

    python tools/build_variant.py NAME -DMACRO[=V] ...      ->  rec_now_amd/librecnow_hip.NAME.so   (objects under csrc/build.NAME/)

Select it at run time with RECNOW_LIB_PATH=rec_now_amd/librecnow_hip.NAME.so (rec_now_amd/_lib.py).  The variant .so is git-ignored and
travels to the GPU box with the snapshot, like the product library."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'rec_now_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(CSRC, 'build.' + name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(ROOT, 'rec_now_amd', 'librecnow_hip.%s.so' % name)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + '.o') for s in srcs]

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('%s\n%s' % (' '.join(cmd), r.stderr))

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, [['hipcc'] + FLAGS + extra + ['-c', s, '-o', o] for s, o in zip(srcs, objs)]))
    run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    print(out)


if __name__ == '__main__':
    main()
