"""Vendor-GEMM solution choices for the data-independent dense shapes of the hot path (PyTorch TunableOp).

The dense row GEMMs of the path (anchor-head 1x1 convolutions as (B*H*W, 512) x (512, 72), the stride-1 up-sampling branch,
the shared MLPs of the set-abstraction scales over 7.08 M grouped rows, the RoI head's 27,648 -> 256 layer ...) go to
rocBLAS / hipBLASLt through torch. Their default heuristics pick poorly for these tall, skinny f32 shapes (N = 16 ... 72):
the picks measured best on an MI355X (one tuning run of bench.py with PYTORCH_TUNABLEOP_TUNING=1, entries whose row count
depends on the data removed) are kept in tunableop_gfx950.csv and looked up at run time — no tuning happens in the product,
shapes that are not in the file (and library versions other than the file's validators) take the library default.
PV-RCNN training 137 -> 150 frames/s, CRB scoring 537 -> 552, SECOND 247.2 -> 248.8 with it.
CRB_TUNABLEOP=0 leaves torch untouched; a user who configures TunableOp himself (PYTORCH_TUNABLEOP_ENABLED set) is not
overridden."""
import os

RESULTS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tunableop_gfx950.csv')


def use_tuned_gemms():
    """-> True when the look-up table was installed"""
    if os.environ.get('CRB_TUNABLEOP', '1') == '0' or 'PYTORCH_TUNABLEOP_ENABLED' in os.environ:
        return False
    import torch
    if not torch.cuda.is_available() or not os.path.exists(RESULTS):
        return False
    try:
        import torch.cuda.tunable as tunable
        tunable.set_filename(RESULTS, insert_device_ordinal=False)
        tunable.enable(True)
        tunable.tuning_enable(False)            # look-ups only
        os.environ.setdefault('PYTORCH_TUNABLEOP_VERBOSE', '0')
        if hasattr(tunable, 'write_file_on_exit'):
            tunable.write_file_on_exit(False)
        return True
    except Exception as e:                      # an older torch without the module: library defaults
        import warnings
        warnings.warn('crbhip: TunableOp look-ups not installed (%s)' % e)
        return False
