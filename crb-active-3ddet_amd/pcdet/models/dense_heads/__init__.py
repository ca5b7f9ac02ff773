"""dense / point heads on the SECOND and PV-RCNN paths, looked up by the NAME field of the model config"""
from . import anchor_head_single, anchor_head_template, point_head_simple, point_head_template
from .anchor_head_single import AnchorHeadSingle
from .anchor_head_template import AnchorHeadTemplate
from .point_head_simple import PointHeadSimple
from .point_head_template import PointHeadTemplate

__all__ = {cls.__name__: cls for cls in (AnchorHeadTemplate, AnchorHeadSingle, PointHeadTemplate, PointHeadSimple)}
