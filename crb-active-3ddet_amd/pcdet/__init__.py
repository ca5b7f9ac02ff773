"""MI355X-native host mirror of the `pcdet` surface on the CRB hot path: same module paths, class and
function names, argument meaning and return contracts as the reference, over hand-written gfx950 kernels
(libcrbhip.so). Only what SECOND / PV-RCNN fwd+bwd and the CRB acquisition pass import is provided."""
__version__ = '0.5.2+crbhip'
