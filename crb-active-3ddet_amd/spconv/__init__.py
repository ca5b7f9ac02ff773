"""MI355X-native stand-in for the `spconv` package surface the reference imports
(pcdet/utils/spconv_utils.py:3-6, pcdet/datasets/processor/data_processor.py:17-26).
All arithmetic runs in hand-written gfx950 kernels behind libcrbhip.so; this package is the thin host mirror."""
__version__ = '2.1.21+crbhip'
from . import pytorch, utils  # noqa: F401
