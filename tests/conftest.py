import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'crb-active-3ddet_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def dev():
    import torch
    return torch.device('cuda', 0)


def require_measure_lib():
    """kernels that live in the measurement library only (first Winograd design, split-bf16 gather-GEMM): their tests run in a
    process that loaded libcrbhip_measure.so (CRB_MEASURE_LIB=1 before importing crbhip) - see
    tests/test_measure_lib_gpu.py, which starts that process; in the normal test process they are skipped"""
    import crbhip
    if not crbhip._lib.MEASURE:
        pytest.skip('measurement-library kernel: covered by tests/test_measure_lib_gpu.py in its own process')
