"""stub of the `cumm` namespace: the reference only needs cumm.tensorview.from_numpy
(pcdet/datasets/processor/data_processor.py:9-12,53-54)."""
from . import tensorview  # noqa: F401
