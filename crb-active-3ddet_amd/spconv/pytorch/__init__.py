from .core import SparseConvTensor  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401
from .conv import (SparseConvolution, SubMConv3d, SparseConv3d, SparseInverseConv3d, SubMConv2d,  # noqa: F401
                   SparseConv2d, set_arithmetic)
from . import conv  # noqa: F401
