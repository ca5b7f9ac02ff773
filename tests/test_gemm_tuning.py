"""Vendor-GEMM look-up table (crbhip/gemm_tuning.py, tunableop_gfx950.csv): the file is well formed and only holds shapes whose
row counts do not depend on the data; on the GPU the looked-up solutions give the library-default results (f32 GEMM, same
products, another summation order: 2e-5 of the output scale)."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSV = os.path.join(ROOT, 'crb-active-3ddet_amd', 'tunableop_gfx950.csv')
FIXED_ROWS = {320000, 524288, 563200, 1048576, 7077888}      # 16 x 20,000 points; keypoint / RoI-grid groups; BEV cells


def test_lookup_table_is_well_formed_and_data_independent():
    lines = [l.strip() for l in open(CSV) if l.strip()]
    validators = [l for l in lines if l.startswith('Validator')]
    assert {v.split(',')[1] for v in validators} >= {'PT_VERSION', 'GCN_ARCH_NAME', 'HIPBLASLT_VERSION', 'ROCBLAS_VERSION'}
    assert any('gfx950' in v for v in validators)
    entries = [l for l in lines if not l.startswith('Validator')]
    assert len(entries) >= 40
    for l in entries:
        op, sig, sol, t = l.split(',')
        assert op.startswith('Gemm') and float(t) > 0
        dims = [int(x) for x in sig.split('_')[1:4]]
        assert all(d < 100000 or d in FIXED_ROWS for d in dims), sig


def test_lookups_are_off_without_a_device_or_on_request(monkeypatch):
    from crbhip import gemm_tuning
    monkeypatch.setenv('CRB_TUNABLEOP', '0')
    assert gemm_tuning.use_tuned_gemms() is False
    monkeypatch.delenv('CRB_TUNABLEOP')
    monkeypatch.setenv('PYTORCH_TUNABLEOP_ENABLED', '0')        # the user's own TunableOp configuration wins
    assert gemm_tuning.use_tuned_gemms() is False


@pytest.mark.gpu
def test_looked_up_solutions_compute_the_same_gemms():
    import pcdet
    import torch.cuda.tunable as tunable
    assert pcdet.TUNED_GEMMS and tunable.is_enabled() and not tunable.tuning_is_enabled()
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    for rows, k, n in ((563200, 512, 72), (563200, 128, 256), (32768, 640, 128), (524288, 64, 64)):
        a = torch.randn(rows, k, device=dev, generator=g)
        w = torch.randn(n, k, device=dev, generator=g) * 0.05
        b = torch.randn(n, device=dev, generator=g)
        y = torch.nn.functional.linear(a, w, b)
        tunable.enable(False)
        try:
            ref = torch.nn.functional.linear(a, w, b)
        finally:
            tunable.enable(True)
        assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
