from .anchor_head_single import AnchorHeadSingle
from .anchor_head_template import AnchorHeadTemplate

__all__ = {
    'AnchorHeadTemplate': AnchorHeadTemplate,
    'AnchorHeadSingle': AnchorHeadSingle,
}
