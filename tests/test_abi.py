"""CPU: the C-ABI shared library loads and exports every symbol include/recnow.h declares; the ctypes table binds
every one of them.  No compute call is made (no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'recnow.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(recnow_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
    from rec_now_amd import _lib
    names = _declared()
    assert len(names) >= 10
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), 'librecnow_hip.so does not export %s' % n
        assert n in _lib.SIGNATURES, '%s is not bound in rec_now_amd/_lib.py' % n
    for n in _lib.SIGNATURES:
        assert n in names, '%s is bound but not declared in include/recnow.h' % n
    assert lib.recnow_abi_version() == _lib.ABI_VERSION


def test_cpu_tensor_is_refused_loudly():
    import pytest
    import torch
    from rec_now_amd.layers.fm_layer import FMLayer
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        FMLayer()([torch.zeros(2, 4), torch.zeros(2, 4)])


def test_small_device_tables_are_uploaded_once_per_content():
    """_lib.const_array / block_ptr_array: host logic of the pointer / width tables of the list-of-tensor kernels (the device
    argument may be the CPU here: only the caching and the address arithmetic are under test)."""
    import torch
    from rec_now_amd import _lib
    cpu = torch.device('cpu')
    a = _lib.const_array([3, 5, 7], torch.int32, cpu)
    assert a.dtype == torch.int32 and a.tolist() == [3, 5, 7]
    assert _lib.const_array((3, 5, 7), torch.int32, cpu) is a                 # same content -> the same upload
    assert _lib.const_array([3, 5, 8], torch.int32, cpu) is not a
    assert _lib.const_array([3, 5, 7], torch.int64, cpu) is not a
    buf = torch.zeros(4, 6, 5)
    p = _lib.block_ptr_array(buf, 4)
    assert p.dtype == torch.int64 and p.tolist() == [buf[f].data_ptr() for f in range(4)]
    assert _lib.ptr_array([buf[1], buf[3]], cpu).tolist() == [buf[1].data_ptr(), buf[3].data_ptr()]


def test_every_module_of_the_package_imports():
    """A syntax error in a host module must show up in the CPU suite, not first on the GPU box."""
    import importlib
    import pkgutil
    import rec_now_amd
    names = [m.name for m in pkgutil.walk_packages(rec_now_amd.__path__, 'rec_now_amd.') if '.csrc.build' not in m.name and not m.name.endswith('librecnow_hip')]
    assert len(names) > 15
    for n in names:
        importlib.import_module(n)


def test_size_queries_cover_the_row_block_route():
    """Host-only size queries (no device call): for the shapes the row-block persistent kernels take (csrc/dcnmix_tile.hip: two experts of 64,
    D = 256 / 512 / 1024, whole 32-row blocks) `saved` holds their fragment-ordered weight packs -- P1 .. P4 of D x 128 floats and V^T per layer --
    and the workspace one dT1 block per layer plus one dV partial per layer and workgroup; other shapes do not pay for them."""
    from rec_now_amd import _lib
    lib = _lib.load()
    B, D, S, N, L = 8192, 1024, 64, 2, 3
    ldt = 144
    act = B * ldt * 4
    packs = L * (4 * D * 128 + 2 * 64 * 64) * 4
    # three experts of 64: no row-block kernels, no extra buffers; the same sizes otherwise (N S + N = 195 -> another leading dimension, so compare
    # each shape with its own lower bound instead)
    sv = lib.recnow_dcn_mix_saved_bytes(B, D, S, N, L)
    ws = lib.recnow_dcn_mix_workspace_bytes(B, D, S, N, L)
    base_saved = L * 3 * act + (L - 1) * B * D * 4 + L * B * D * 4 + (2 * L + 1) * D * ldt * 4
    # round 6: + the split-precision piece planes of the packed weights (per layer two of D / 8 x 128 and two of 144 / 8 x D units of 16 B, three pieces each)
    planes = L * 2 * (D // 8 * 128 * 48 + 144 // 8 * D * 48)
    # round 6, second session: + the fragment-ordered piece planes of the split-precision row-block forward (csrc/dcnmix_tile_split.hip: per layer three
    # pieces of D x 16 + D x 18 units of 16 B)
    tile_planes = L * 3 * (D * 16 + D * 18) * 16
    assert sv >= base_saved + packs + planes + tile_planes
    assert sv < base_saved + packs + planes + tile_planes + (1 << 20)             # alignment slack only
    assert ws >= L * act + L * 256 * N * S * S * 4 + 3 * act + 2 * B * D * 4
    # a width without an instantiation (D = 1152: exact-128 path, leading dimension 144) and a batch off the exact path (B % 256 != 0: leading
    # dimension 160, no O_l, no packs at all): what the formulation itself keeps plus alignment slack -- the packs (7 MB / 6 MB) are not in there
    d2 = 1152
    base2 = L * 3 * act + (2 * L - 1) * B * d2 * 4 + (2 * L + 1) * d2 * ldt * 4 + L * 2 * (d2 // 8 * 128 * 48 + 144 // 8 * d2 * 48)      # (+ the piece planes: exact path)
    assert base2 <= lib.recnow_dcn_mix_saved_bytes(B, d2, S, N, L) < base2 + (1 << 20)
    b3 = 8200
    base3 = L * 3 * b3 * 160 * 4 + (L - 1) * b3 * D * 4
    assert base3 <= lib.recnow_dcn_mix_saved_bytes(b3, D, S, N, L) < base3 + (1 << 20)
