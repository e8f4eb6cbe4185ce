"""Register / LDS / scratch footprint of every kernel of one HIP source (device-only compile to assembly, the .amdhsa metadata notes).

    python tools/isa/kernel_regs.py rec_now_amd/csrc/gemm_split.hip [-DRN_...]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1]
out = os.path.join(tempfile.mkdtemp(), 'k.s')
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-function', '--cuda-device-only', '-S', src, '-o', out,
                '-I' + os.path.join(ROOT, 'include')] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
meta = txt[txt.rfind('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    f = lambda k: (re.search(r'\.%s:\s*(\S+)' % k, blk) or [None, '?'])[1]      # noqa: E731
    name = f('name')
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()[:110]
    except OSError:
        pass
    print('%-112s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill %s' % (name, f('vgpr_count'), blk.split()[0], f('sgpr_count'), f('group_segment_fixed_size'),
                                                                      f('private_segment_fixed_size'), f('vgpr_spill_count')))
