__all__ = {}
try:
    from .roi_head_template import RoIHeadTemplate
    from .pvrcnn_head import PVRCNNHead
    from .partA2_head import PartA2FCHead
    __all__.update({'RoIHeadTemplate': RoIHeadTemplate, 'PVRCNNHead': PVRCNNHead, 'PartA2FCHead': PartA2FCHead})
except ImportError:
    pass
