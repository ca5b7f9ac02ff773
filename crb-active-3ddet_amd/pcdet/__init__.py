"""MI355X-native host mirror of the `pcdet` surface on the CRB hot path: same module paths, class and
function names, argument meaning and return contracts as the reference, over hand-written gfx950 kernels
(libcrbhip.so). Only what SECOND / PV-RCNN fwd+bwd and the CRB acquisition pass import is provided."""
__version__ = '0.5.2+crbhip'

from crbhip.gemm_tuning import use_tuned_gemms as _use_tuned_gemms

TUNED_GEMMS = _use_tuned_gemms()       # vendor-GEMM solution look-ups for the fixed dense shapes (CRB_TUNABLEOP=0: off)

