"""CPU: the C-ABI library loads and exports every symbol include/crb_hip.h declares (no compute calls)."""
import ctypes
import os


def test_header_symbols_exported():
    import crbhip
    protos = crbhip.parse_header()
    assert len(protos) >= 19
    L = ctypes.CDLL(crbhip.lib_path)
    for name in protos:
        assert hasattr(L, name), name


def test_product_library_exports_no_measurement_entry_point():
    """the knobs that select kernel variants or make kernels skip work (wrong results by design) live in
    libcrbhip_measure.so only (include/crb_hip_measure.h, csrc built with -DCRB_MEASURE); nothing in a long-lived process
    can flip the product library into such a mode"""
    import subprocess
    import crbhip
    measure = crbhip.parse_header(crbhip._lib.measure_header_path)
    assert 'crb_sparse_conv_set_wgrad_mode' in measure and 'crb_sparse_conv_bf16x3_set_mode' in measure
    assert not crbhip._lib.MEASURE and crbhip.lib_path.endswith('libcrbhip.so')
    exported = subprocess.check_output(['nm', '-D', '--defined-only', crbhip.lib_path], text=True)
    names = {ln.split()[-1] for ln in exported.splitlines() if ln.strip()}
    for name in measure:
        assert name not in names, name
    for pat in ('set_mode', 'set_wgrad', 'set_subtiles', 'set_tiles', 'conv_timing'):
        assert not [n for n in names if n.startswith('crb_') and pat in n], pat
    assert not set(measure) & set(crbhip.parse_header())
    mlib = crbhip.lib_path.replace('libcrbhip.so', 'libcrbhip_measure.so')
    if os.path.exists(mlib):                                     # the tools' build has both sets
        L = ctypes.CDLL(mlib)
        for name in list(measure) + list(crbhip.parse_header()):
            assert hasattr(L, name), name


def test_abi_version_and_pure_host_queries():
    import crbhip
    assert crbhip.lib.crb_abi_version() >= 1
    assert crbhip.lib.crb_hash_capacity_for(1000) == 8192          # small tables: every key % 8 class holds all keys
    assert crbhip.lib.crb_hash_capacity_for(300000) == 1048576
    assert crbhip.lib.crb_voxelize_workspace_bytes(20000, 1, 16000, 5) > 0
    assert crbhip.lib.crb_sparse_conv_supported(64, 64) == 1
    assert crbhip.lib.crb_sparse_conv_supported(7, 9) == 0


def test_header_has_no_torch_types():
    import crbhip
    import re
    src = open(crbhip.header_path).read()
    code = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    assert 'at::' not in code and 'torch' not in code and 'Tensor' not in code


def test_missing_gpu_fails_loudly():
    """product ops refuse CPU tensors instead of silently falling back"""
    import pytest
    import torch
    import crbhip
    from crbhip import voxel
    with pytest.raises(crbhip.CrbHipError):
        voxel.voxelize(torch.zeros(10, 4), torch.tensor([0, 10], dtype=torch.int32), [0, -40, -3, 70.4, 40, 1],
                       [0.05, 0.05, 0.1], 100, 5)


def test_ptr_keeps_temporaries_alive_until_check():
    """`ptr(t.contiguous())` may receive a temporary; _lib keeps it referenced until check() so that the allocator cannot
    hand its block to the next temporary of the same argument list"""
    import gc
    import weakref
    import torch
    from crbhip import _lib
    t = torch.arange(12).view(3, 4).t()                     # non-contiguous -> .contiguous() makes a temporary
    tmp = t.contiguous()
    ref = weakref.ref(tmp)
    p = _lib.ptr(tmp)
    del tmp
    gc.collect()
    assert ref() is not None and p.value == ref().data_ptr()
    _lib.check(0, 'noop')
    gc.collect()
    assert ref() is None
