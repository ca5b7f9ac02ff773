from .detector3d_template import Detector3DTemplate
from .second_net import SECONDNet

__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'SECONDNet': SECONDNet,
}

try:
    from .pv_rcnn import PVRCNN
    __all__['PVRCNN'] = PVRCNN
except ImportError:   # PV-RCNN pieces land after SECOND
    pass


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
