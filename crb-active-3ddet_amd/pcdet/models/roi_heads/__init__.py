__all__ = {}
try:
    from .roi_head_template import RoIHeadTemplate
    from .pvrcnn_head import PVRCNNHead
    __all__.update({'RoIHeadTemplate': RoIHeadTemplate, 'PVRCNNHead': PVRCNNHead})
except ImportError:
    pass
